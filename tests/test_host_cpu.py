"""CPU: host logic that needs no GPU — the pbtxt reader, the model zoo against the reference's own
model files (when mounted), graph building / sizing / parameter layout, optimizer schedules."""
import math
import os

import pytest

from convnet_amd import models, pbtxt
from convnet_amd.convnet import ConvNet
from convnet_amd.optimizer import SGDOptimizer

REF = "/root/reference/examples"


def test_pbtxt_reader_defaults_presence_and_errors():
    m = pbtxt.parse('name: "x"  seed: 7 # comment\nlayer { name: "a" num_channels: 3 }\n'
                    'edge { source: "a" dest: "b" edge_type: CONVOLUTIONAL kernel_size: 3 grad_check_epsilon: 0.01 grad_check_epsilon: 1e-3\n'
                    '  weight_optimizer { epsilon: 0.5 } }')
    assert m.name == "x" and m.seed == 7 and m.max_iter == -1
    e = m.edge[0]
    assert e.edge_type == "CONVOLUTIONAL" and e.kernel_size == 3 and e.stride == 1 and e.padding == 0
    assert e.has_kernel_size() and not e.has_kernel_size_y() and e.kernel_size_y == 0
    assert e.grad_check_epsilon == [0.01, 0.001]
    assert e.weight_optimizer.epsilon == 0.5 and e.weight_optimizer.final_momentum == 0.0
    assert not e.has_bias_optimizer() and e.bias_optimizer.gradient_clip == -1.0
    assert m.layer[0].activation == "LINEAR" and m.layer[0].loss_function == "CROSS_ENTROPY_MULTINOMIAL"
    with pytest.raises(ValueError):
        pbtxt.parse('name: "x" bogus_field: 3')
    d = pbtxt.Optimizer()
    d.epsilon = 0.1
    d.l2_decay = 0.5
    o = d.copy()
    o.MergeFrom(e.weight_optimizer)      # default-optimizer merge, src/convnet.cc:36-50
    assert o.epsilon == 0.5 and o.l2_decay == 0.5


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted")
@pytest.mark.parametrize("gen,path", [(models.alexnet, "imagenet/CLS_net_20140621074703.pbtxt"), (models.mnist_conv, "mnist-conv/net.pbtxt"),
                                      (models.alexnet_nin, "imagenet/CLS_net_20140801232522.pbtxt")])
def test_generated_models_equal_reference_pbtxt(gen, path):
    ref, mine = pbtxt.read(os.path.join(REF, path)), pbtxt.parse(gen())

    def sig(m):
        out = [(l.name, l.num_channels, l.activation, l.dropprob, l.image_size_y, l.image_size_x) for l in m.layer]
        for e in m.edge:
            w, b = e.weight_optimizer, e.bias_optimizer
            out.append((e.source, e.dest, e.edge_type, e.kernel_size, e.stride, e.padding, e.shared_bias, e.initialization, e.init_wt,
                        e.init_bias, w.epsilon, w.final_momentum, w.momentum_transition_timescale, w.l2_decay, w.weight_norm_limit, w.weight_norm_constraint,
                        b.epsilon, b.final_momentum, b.l2_decay, e.add_scale, e.pow_scale, e.frac_of_filters_response_norm))
        return out
    assert sig(ref) == sig(mine)


def test_alexnet_graph_sizes_params_and_work_match_the_survey():
    net = ConvNet(models.alexnet())
    sizes = {l.GetName(): (l.GetSizeY(), l.GetNumChannels()) for l in net.layers_}
    assert sizes["hidden1_conv"] == (110, 96) and sizes["hidden1_maxpool"] == (55, 96) and sizes["hidden2_conv"] == (26, 256)
    assert sizes["hidden2_rnorm"] == (13, 256) and sizes["hidden5_conv"] == (11, 256) and sizes["hidden5_maxpool"] == (6, 256)
    assert [l.GetName() for l in net.layers_][0] == "input" and net.layers_[-1].GetName() == "output"
    assert sum(e.GetParameterMemoryRequirement() for e in net.edges_) == 62357608
    fwd, train = models.count_macs(net)
    assert fwd == 1125565568 and train == 3205941504
    rn = net.GetEdgeByName("hidden1_maxpool:hidden1_rnorm")
    assert rn.num_filters_response_norm_ == 24
    d = net.GetEdgeByName("input:hidden1_conv").conv_desc_
    assert (d.padding_y, d.padding_x, d.stride_y, d.kernel_size_x) == (-1, -1, 2, 7)           # stored negated


def test_mnist_and_vgg_graphs():
    net = ConvNet(models.mnist_conv())
    assert [(l.GetSizeY(), l.GetNumChannels()) for l in net.layers_] == [(28, 1), (25, 48), (11, 48), (8, 128), (3, 128), (1, 10)]
    fwd, _ = models.count_macs(net)
    assert fwd == 480000 + 6291456 + 11520
    v = ConvNet(models.vgg())
    assert v.GetLayerByName("pool5").GetSizeY() == 7 and len([e for e in v.edges_ if e.GetParameterMemoryRequirement()]) == 16


def test_sgd_schedules():
    import numpy as np
    c = pbtxt.Optimizer()
    c.epsilon, c.initial_momentum, c.final_momentum, c.momentum_transition_timescale = 0.01, 0.5, 0.9, 2000
    o = SGDOptimizer.__new__(SGDOptimizer)
    from convnet_amd.optimizer import Optimizer
    Optimizer.__init__(o, c)
    o.initial_momentum_, o.final_momentum_, o.momentum_transition_timescale_ = 0.5, 0.9, 2000
    o.step_ = 0
    assert o.GetMomentum() == 0.5
    o.step_ = 2000
    assert abs(o.GetMomentum() - (0.5 + 0.4 * (1 - math.exp(-1)))) < 1e-7                     # src/optimizer.cc:158-165, in float
    assert o.GetMomentum() == float(np.float32(o.GetMomentum()))                                # ... and representable as one
    c2 = pbtxt.Optimizer()
    c2.epsilon, c2.epsilon_decay, c2.epsilon_decay_timescale, c2.minimum_epsilon = 1.0, "INVERSE_T", 10, 0.2
    Optimizer.__init__(o, c2)
    o.step_ = 10
    assert o.GetDecayedEpsilon() == 0.5
    o.step_ = 1000
    assert o.GetDecayedEpsilon() == float(np.float32(0.2))   # float minimum_epsilon_


def test_hdf5_io_layout_and_reference_written_file(tmp_path):
    """The HDF5 layer without a GPU: dataset shape convention (a column-major (rows, cols) matrix is a
    row-major (cols, rows) dataset, util.cc:128-175), int attributes, missing-attribute default, size mismatch; and, when
    the reference tree is mounted, a file the reference's own tools wrote (examples/imagenet/pixel_mean.h5)."""
    import numpy as np
    from convnet_amd import hdf5io
    p = str(tmp_path / "t.h5")
    a = np.arange(12, dtype=np.float32)
    with hdf5io.File(p, "w") as f:
        f.WriteHDF5CPU(a, 4, 3, "x:y:weight")          # Matrix(rows=3, cols=4).WriteHDF5 passes (size[1], size[0])
        f.WriteHDF5IntAttr("x:y:weight_step", 41)
    with hdf5io.File(p) as f:
        assert f.ReadHDF5Shape("x:y:weight") == (3, 4) and f.Has("x:y:weight") and not f.Has("x:y:bias")
        assert np.array_equal(f.ReadHDF5CPU(12, "x:y:weight"), a)
        assert f.ReadHDF5IntAttr("x:y:weight_step", 0) == 41 and f.ReadHDF5IntAttr("__current_iter__", 7) == 7
        with pytest.raises(ValueError):
            f.ReadHDF5CPU(11, "x:y:weight")
        with pytest.raises(KeyError):
            f.ReadHDF5Shape("nope")
    ref = os.path.join(REF, "imagenet/pixel_mean.h5")
    if os.path.exists(ref):
        with hdf5io.File(ref) as f:
            assert f.ReadHDF5Shape("pixel_mean") == (1, 3)
            assert np.allclose(f.ReadHDF5CPU(3, "pixel_mean"), [122.77497, 115.91181, 102.984184], rtol=1e-6)
            assert np.allclose(f.ReadHDF5CPU(3, "pixel_std"), [70.58011, 68.60053, 72.02416], rtol=1e-6)


def test_check_reduce_learning_rate_follows_the_reference_rule():
    """src/convnet.cc:788-817: compare the means of the older and the newer half of the last reduce_lr_num_steps
    validation values; reduce when the improvement is below reduce_lr_threshold."""
    text = models.mnist_conv().replace("print_after: 100", "print_after: 100\nreduce_lr_num_steps: 4\nreduce_lr_threshold: 0.01")
    net = ConvNet(text)
    assert not net.CheckReduceLearningRate([0.1, 0.2, 0.3])                       # fewer than num_steps values
    assert not net.CheckReduceLearningRate([0.1, 0.2, 0.3, 0.4])                  # accuracy still rising (0.15 -> 0.35)
    assert net.CheckReduceLearningRate([0.0, 0.5, 0.50, 0.51, 0.505, 0.507])      # last four: 0.505 -> 0.506: flat
    assert net.CheckReduceLearningRate([0.6, 0.6, 0.5, 0.5])                      # got worse
    net.model_.smaller_is_better = True
    assert not net.CheckReduceLearningRate([0.6, 0.6, 0.5, 0.5])                  # an error metric that still falls


def test_round3_bench_line_prices_the_dominant_kernel_on_the_pipe_it_executes_on():
    """profiles/r03_bench_n1.json — the line `python bench.py` printed on the MI355X at the end of round 3: driver contract fields,
    the roofline against the EXECUTED pipe (2500 dense bf16 TFLOP/s / 6 MFMA flops per fp32 product = 416.67; VERDICT r02 item 4:
    no fraction may exceed 1), traffic measured live by the two PMC child passes, and the strong-scaling batch sweep beside it."""
    import json
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    d = json.load(open(os.path.join(root, "r03_bench_n1.json")))
    for k, t in [("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                 ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict), ("roofline", dict),
                 ("cpu_baseline", dict), ("rccl_ranks", int)]:
        assert isinstance(d[k], t), k
    assert d["vs_baseline"] is None and d["scaling"] == "weak" and d["dtype"] == "f32" and d["n_gpus"] == 1 and d["rccl_ranks"] == 0
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] == pytest.approx(d["config"]["global_batch"] / (d["ms_per_step"] * 1e-3), rel=1e-3)
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["kernel"].endswith(",split,pre>")
    assert r["peak"] == pytest.approx(2500.0 / 6, abs=0.01) and r["frac"] == pytest.approx(r["achieved"] / r["peak"], abs=1e-3)
    assert r["achieved"] == pytest.approx(r["flops_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e12, rel=2e-2)
    assert r["pipe"]["frac"] == pytest.approx(r["frac"], abs=1e-3) and r["pipe"]["peak"] == 2500.0
    fracs = [r["frac"], r["executed_frac"], r["model_frac"], r["one_stream"]["frac"], r["all_mfma_kernels"]["frac"]] + \
            [f["frac"] for f in r["families"].values()]
    assert all(0 < f <= 1 for f in fracs), fracs
    assert r["one_stream"]["achieved"] > r["achieved"]
    assert r["traffic"] > 0 and r["traffic_source"].startswith("measured in this run") and r["traffic"] == r["traffic_read"] + r["traffic_write"]
    # the same kernel's average duration in the rocprofv3 --kernel-trace --stats summary of the same command agrees with the HIP events
    import csv
    with open(os.path.join(root, "r03_bench_kernel_stats.csv")) as f:
        row = next(x for x in csv.DictReader(f) if "ggp_kernel<2, 2, 2, 128, true, true>" in x["Name"])
    assert float(row["AverageNs"]) * 1e-6 == pytest.approx(r["avg_launch_ms"], rel=0.08)
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["unit"] == "images/sec" and c["sample"]
    assert d["fp32_mfma_path"]["matrix_path"] == "fp32" and d["fp32_mfma_path"]["value"] < d["value"] and d["fp32_mfma_path"]["model_frac"] < 1
    assert d["ref_host"]["value"] > 0
    # per-GPU batches of strong scaling (global 256 over 2 / 4 / 8 GPUs), before and after the flat column space
    rate = {b: json.load(open(os.path.join(root, f"r03_bench_b{b}.json")))["value"] for b in (128, 64, 32)}
    before = {b: json.load(open(os.path.join(root, f"r03_before_flat_columns_bench_b{b}.json")))["value"] for b in (128, 64, 32)}
    assert rate[32] > 1.5 * before[32] and rate[64] > 1.15 * before[64] and rate[128] >= 0.98 * before[128]
    assert rate[128] > 0.9 * d["value"] and rate[64] > 0.7 * d["value"] and rate[32] > 0.55 * d["value"]


@pytest.mark.parametrize("which", ["r01_bench_n1.json", "r02_bench_n1.json", "r02_fp32path_bench_n1.json"])
def test_committed_bench_line_follows_the_driver_contract(which):
    """profiles/rNN_bench_n1.json is the line `python bench.py` printed on the MI355X: every field the driver and the
    judge read is present, typed, and self-consistent."""
    import json
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", which)
    d = json.load(open(path))
    for k, t in [("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                 ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict),
                 ("roofline", dict), ("cpu_baseline", dict)]:
        assert isinstance(d[k], t), k
    assert d["vs_baseline"] is None and d["scaling"] == "weak" and d["dtype"] == "f32" and d["unit"] == "images/sec"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] == pytest.approx(d["config"]["global_batch"] / (d["ms_per_step"] * 1e-3), rel=1e-3)
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3 and "traffic" in r
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], abs=1e-3)
    assert r["achieved"] == pytest.approx(r["flops_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e12, rel=2e-2)
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["unit"] == "images/sec" and c["sample"]
    if which == "r02_bench_n1.json":
        # the default matrix path forms fp32 products on the bf16 pipe: the line carries the fp32-instruction run beside it, the
        # pipe-level accounting, and the dominant kernel's rate without a co-running kernel of the second stream
        assert "bf16" in d["arithmetic"] and d["fp32_mfma_path"]["matrix_path"] == "fp32" and d["fp32_mfma_path"]["value"] < d["value"]
        assert r["pipe"]["peak"] == 2500.0 and r["pipe"]["executed_per_algorithmic_flop"] == 6
        assert r["one_stream"]["achieved"] > r["achieved"] and r["traffic"] > 0 and r["kernel"].endswith(",split,pre>")
        assert d["ref_host"]["value"] > 0


def test_nin_model_graph_and_work_count():
    """CLS_net_20140801232522 (CONV_ONETOONE layers): layer sizes, parameter count of a 1x1 edge and MACs per image with the
    1x1 convolutions counted per pixel."""
    net = ConvNet(models.alexnet_nin())
    sizes = {l.GetName(): (l.GetSizeY(), l.GetNumChannels()) for l in net.layers_}
    assert sizes["hidden2_conv"] == (27, 256) and sizes["hidden2_conv_nin1"] == (27, 256) and sizes["hidden2_maxpool"] == (14, 256)
    assert sizes["hidden4_conv_nin1"] == (14, 768) and sizes["hidden5_conv_nin2"] == (12, 512) and sizes["hidden5_maxpool"] == (6, 512)
    e = net.GetEdgeByName("hidden3_conv:hidden3_conv_nin1")
    assert e.GetParameterMemoryRequirement() == 768 * (384 + 1)
    fwd, train = models.count_macs(net)
    assert fwd == 2035639424 and train == 5936163072


def test_pmc_traffic_summary_is_reproducible_from_the_committed_counter_files():
    """profiles/r01_pmc_traffic_bench.json (what bench.py reports as roofline.traffic) == tools/pmc_traffic.py over the two
    committed rocprofv3 counter files, and the calibration kernels behave as DESIGN.md section 5 states."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prof = os.path.join(root, "profiles")
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "pmc_traffic.py"),
                          os.path.join(prof, "r01_pmc_FETCH_SIZE_counter_collection.csv"),
                          os.path.join(prof, "r01_pmc_WRITE_SIZE_counter_collection.csv")], capture_output=True, text=True, check=True)
    got = json.loads(out.stdout)
    with open(os.path.join(prof, "r01_pmc_traffic_bench.json")) as f:
        want = json.load(f)
    assert got["kernels"] == want["kernels"]
    k = got["kernels"]
    # rnorm1 forward reads its 290 400 KiB input once and writes as much: the doubled FETCH_SIZE must equal WRITE_SIZE
    rn = k["rnorm_fwd_lds_kernel<64>"]
    assert rn["write_size_kib_raw"] == 290400.0 and abs(2 * rn["fetch_size_kib_raw"] / 290400.0 - 1.0) < 0.01
    dom = k["gg_kernel<2, 2, 2, 128, false, true, false>"]
    assert dom["launches"] == 40 and 3.0e8 < dom["traffic_bytes"] < 4.0e8


@pytest.mark.skipif(not os.path.exists("/root/reference/proto/convnet_config.proto"), reason="reference tree not mounted")
def test_pbtxt_schema_matches_the_reference_proto_field_for_field():
    """convnet_amd/pbtxt.py restates proto/convnet_config.proto by hand: every field of Layer / Edge / Optimizer / Model (and
    LayerSlice) must exist with the proto's default, so a reference pbtxt means the same thing to both hosts.  The proto is
    parsed with the same reader the seam build uses to generate its C++ config classes (oracle/seam/gen_config_pb.py)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen_config_pb", os.path.join(root, "oracle", "seam", "gen_config_pb.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    from convnet_amd import pbtxt
    with open("/root/reference/proto/convnet_config.proto") as f:
        msgs = {m.name: m for m in gen.flatten(gen.parse(f.read()))}
    out_of_scope = {"Model": {"subnet", "train_dataset", "valid_dataset"}}   # sub-net merging and dataset configs: not on the path
    for name, cls in (("Layer", pbtxt.Layer), ("Edge", pbtxt.Edge), ("Optimizer", pbtxt.Optimizer), ("Model", pbtxt.Model),
                      ("LayerSlice", pbtxt.LayerSlice)):
        proto = {f["name"]: f for f in msgs[name].fields}
        enums = {e: [v for v, _ in vals] for m in msgs.values() for e, vals in m.enums}
        assert set(proto) - set(cls.FIELDS) == out_of_scope.get(name, set()), name
        assert not set(cls.FIELDS) - set(proto), name
        for key, mine in cls.FIELDS.items():
            f = proto[key]
            if f["label"] == "repeated":
                assert mine is list or (isinstance(mine, tuple) and mine[0] is list), (name, key)
                continue
            if isinstance(mine, type):                       # sub-message
                assert f["type"] == mine.__name__, (name, key)
                continue
            d = f["default"]
            if f["type"] == "string":
                want = "" if d is None else d.strip('"')
            elif f["type"] == "bool":
                want = d == "true"
            elif f["type"] in ("float", "double"):
                want = 0.0 if d is None else float(d)
            elif f["type"] in gen.SCALARS:
                want = 0 if d is None else int(d)
            else:                                            # enum: declared default, else its first value (proto2)
                want = d if d is not None else enums[f["type"].split(".")[-1]][0]
            if isinstance(want, float):
                assert abs(float(mine) - want) <= 1e-9 * max(1.0, abs(want)), (name, key, mine, want)
            else:
                assert mine == want and type(mine) is type(want), (name, key, mine, want)


def test_bench_kernel_name_matching_and_one_stream_fields():
    """bench.py pairs its kernel-family names with rocprofv3's full template argument lists (roofline.traffic lookup) and derives the
    dominant kernel's one-stream rate from the library's per-launch profile rows."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    same = b._same_kernel
    assert same("ggp_kernel<2,2,2,128,split,pre>", "void chip::ggp_kernel<2, 2, 2, 128, true, true>(chip::GGParams, chip::GGClassTable)")
    assert same("ggp_kernel<2,2,2,128,split>", "ggp_kernel<2, 2, 2, 128, true, false>")
    assert not same("ggp_kernel<2,2,2,128,split>", "ggp_kernel<2, 2, 2, 128, true, true>")
    assert same("ggp_kernel<2,2,2,128>", "ggp_kernel<2, 2, 2, 128, false, false>") and same("ggp_kernel<2,2,2,128>", "ggp_kernel<2, 2, 2, 128>")
    assert not same("ggp_kernel<2,2,2,128>", "ggp_kernel<1, 4, 3, 64>")
    assert same("wg_kernel<2,2,5,3,x16,split>", "wg_kernel<2, 2, 5, 3, true, 16, true>") and not same("wg_kernel<2,2,5,3,x16>", "wg_kernel<2, 2, 5, 3, true, 16, true>")
    assert same("wg_kernel<2,2,2,2>", "wg_kernel<2, 2, 2, 2, true, 32>") and not same("wg_kernel<2,2,2,2>", "wg_kernel<2, 2, 2, 2, true, 16>")
    assert same("gg_kernel<2,2,2,128,kc,split>", "gg_kernel<2, 2, 2, 128, true, true, false, true>")
    assert same("gg_kernel<1,4,3,64,rc>", "gg_kernel<1, 4, 3, 64, false, true, false>") and not same("gg_kernel<1,4,3,64,rc>", "gg_kernel<1, 4, 3, 64, true, true, false>")
    assert not same("sgd_kernel", "map2_kernel<add_scalar::{lambda")     # truncated names in the PMC file must not raise
    assert same("gpw_kernel<128x512,split,raw>", "chip::gpw_kernel(chip::GGParams, chip::GGClassTable)") and same("gpw_kernel<128x512,split,raw>", "chip::gpw_kernel")
    assert not same("gpw_kernel<128x512,split,raw>", "chip::gpw_tail_fix_kernel(chip::GGParams)")
    assert same("wgw_kernel<256x192,split>", "void chip::wgw_kernel<3>(chip::WGParams)") and not same("wgw_kernel<256x192,split>", "void chip::wgw_kernel<4>(chip::WGParams)")
    assert same("wgw_kernel<256x256,split>", "chip::wgw_kernel<4>") and not same("wgw_kernel<256x256,split>", "chip::wgw_kernel<3>")
    assert same("gpp_kernel<2,2,2,128,split,raw>", "void chip::gpp_kernel<2, 2, 2, 128, true>(chip::GGParams, chip::GGClassTable)")
    assert b.kernel_peak("gpw_kernel<128x512,split,raw>") == pytest.approx(416.6667, rel=1e-5) and b.kernel_peak("wgw_kernel<256x256,split>") == pytest.approx(416.6667, rel=1e-5)
    rows = [{"kernel": "k", "ms": 2.0, "flops": 4e11, "launches": 4}, {"kernel": "k", "ms": 2.0, "flops": 4e11, "launches": 4},
            {"kernel": "other", "ms": 9.0, "flops": 1.0, "launches": 1}]
    one = b.one_stream_fields(rows, "k")["one_stream"]
    assert one["achieved"] == 200.0 and one["launches"] == 8 and one["avg_launch_ms"] == 0.5 and abs(one["frac"] - 200.0 / 157.3) < 1e-4
    assert b.one_stream_fields(rows, "absent") == {} and b.one_stream_fields(None, "k") == {}
    # a bf16-split build is priced on the pipe it executes on: 2500 dense bf16 TFLOP/s / 6 MFMA flops per fp32 product — no
    # fraction on the bench line is taken against the fp32 instruction's peak any more (a fraction that can exceed 1)
    rows = [{"kernel": "ggp_kernel<2,2,2,128,split,pre>", "ms": 2.0, "flops": 4e11, "launches": 4}]
    one = b.one_stream_fields(rows, "ggp_kernel<2,2,2,128,split,pre>")["one_stream"]
    assert abs(one["frac"] - 200.0 / (2500.0 / 6)) < 1e-4 and one["frac"] < 1
    assert b.kernel_peak("wg_kernel<2,2,2,2,split>") == pytest.approx(416.6667, rel=1e-5) and b.kernel_peak("wg_kernel<2,2,2,2>") == 157.3


@pytest.mark.parametrize("gen", ["alexnet", "alexnet_nin", "vgg", "mnist_conv", "lenet5"])
def test_pbtxt_writer_round_trips_every_model(gen):
    """WritePbtxt (src/util.cc:104-112) counterpart: the text the writer emits parses back to the same message — explicit presence,
    repeated fields, nested optimizers, enums bare and strings quoted."""
    m = pbtxt.parse(getattr(models, gen)())
    m.timestamp.append("20140621074703")
    m.checkpoint_dir = 'dir with "quotes"'
    text = pbtxt.dump(m)
    back = pbtxt.parse(text)
    assert pbtxt.dump(back) == text and back.timestamp == ["20140621074703"] and back.checkpoint_dir == 'dir with "quotes"'
    assert 'activation: RECTIFIED_LINEAR' in text and 'name: "' in text
    assert [l.name for l in back.layer] == [l.name for l in m.layer] and [(e.source, e.dest) for e in back.edge] == [(e.source, e.dest) for e in m.edge]
    assert back.edge[0].has_weight_optimizer() == m.edge[0].has_weight_optimizer()


def test_timestamp_model_stamps_every_run_and_writes_the_pbtxt(tmp_path):
    """ConvNet::TimestampModel (src/convnet.cc:830-838): a new stamp per Train() on root (a resumed model keeps its old stamps and
    gets a new checkpoint name), <dir>/<name>_<stamp>.pbtxt written, log file names set."""
    net = ConvNet(models.mnist_conv())
    net.model_.checkpoint_dir = str(tmp_path)
    first = net.TimestampModel()
    ck1 = net.GetCheckpointFilename()
    second = net.TimestampModel()
    assert net.model_.timestamp == [first, second] and second > first and net.GetCheckpointFilename() != ck1
    for ts in (first, second):
        assert os.path.exists(os.path.join(str(tmp_path), f"{net.model_.name}_{ts}.pbtxt"))
    assert pbtxt.read(os.path.join(str(tmp_path), f"{net.model_.name}_{second}.pbtxt")).timestamp == [first, second]
    assert net.log_file_.endswith(f"{second}_train.log") and net.val_log_file_.endswith(f"{second}_valid.log")
