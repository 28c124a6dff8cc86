"""GPU: whole-net tests through the ConvNet driver.
 * Fprop(false) activations of a small AlexNet-shaped net == the CPU oracle run layer by layer on the
   same weights and inputs (tolerance: the reference's 1e-4, py/test_conv.py:382-392).
 * fused entry points == the reference's unfused Matrix-call sequence (same net, same data).
 * the GradChecker port runs the reference's flow (the strict gate is tests/test_grad_check_strict.py).
 * one SGD step leaves parameters equal between fused/unfused; training reduces the loss.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle  # noqa: E402
from oracle import Geom  # noqa: E402
from golden_cases import rel_err  # noqa: E402

TOL = 1e-4


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available()
    from convnet_amd.matrix import Matrix
    Matrix.SetupCUDADevice(0)
    return Matrix


def small_alexnet(dropprob=0.0, grad_check=False):
    """AlexNet topology (conv-pool-rnorm x2, conv x3, pool, fc x3) at 35x35 input / thin channels."""
    from convnet_amd import models
    gc = models._gc(grad_check, 6)
    s = models._header("tiny_alex")
    L, C, P, R, F = models._layer, models._conv, models._pool, models._rnorm, models._fc
    s += L("input", 3, size=35)
    s += L("c1", 16, "RECTIFIED_LINEAR") + L("p1", 16) + L("r1", 16, "RECTIFIED_LINEAR")
    s += L("c2", 24, "RECTIFIED_LINEAR") + L("p2", 24) + L("r2", 24, "RECTIFIED_LINEAR")
    s += L("c3", 32, "RECTIFIED_LINEAR") + L("c4", 32, "RECTIFIED_LINEAR") + L("c5", 24, "RECTIFIED_LINEAR") + L("p5", 24)
    s += L("f6", 48, "RECTIFIED_LINEAR", dropprob) + L("f7", 40, "RECTIFIED_LINEAR", dropprob) + L("output", 10, "SOFTMAX")
    s += C("input", "c1", 5, 2, 1, grad_check=gc) + P("c1", "p1", 3, 2, 1) + R("p1", "r1", 0.05, 0.75, 0.5)
    s += C("r1", "c2", 3, 1, 0, init_bias=1.0, grad_check=gc) + P("c2", "p2", 3, 2, 1) + R("p2", "r2", 0.05, 0.75, 0.25)
    s += C("r2", "c3", 3, 1, 1, grad_check=gc) + C("c3", "c4", 3, 1, 1, init_bias=1.0, grad_check=gc) + C("c4", "c5", 3, 1, 0, init_bias=1.0, grad_check=gc)
    s += P("c5", "p5", 3, 2, 1)
    s += F("p5", "f6", grad_check=gc) + F("f6", "f7", grad_check=gc) + F("f7", "output", grad_check=gc)
    return s


def build(text, batch, fused, seed_data=5, cls=None):
    from convnet_amd.convnet import ConvNet
    from convnet_amd.datahandler import SyntheticDataHandler
    net = (cls or ConvNet)(text, fused=fused)
    net.SetBatchsize(batch)
    data = SyntheticDataHandler(net, batch, seed=seed_data, num_batches=1)
    net.SetupDataset(data)
    net.AllocateMemory(False)
    return net


def copy_params(src, dst):
    dst.parameters_.Set(src.parameters_)


def test_fprop_activations_match_cpu_oracle_layer_by_layer(gpu):
    from convnet_amd.edge import ConvEdge, FCEdge, MaxPoolEdge, ResponseNormEdge
    N = 8
    net = build(small_alexnet(), N, fused=False)
    for l in net.layers_:
        l.ResetAddOrOverwrite()
    net.GetBatch(net.train_dataset_)
    net.Fprop(False)
    # replay on the oracle, edge by edge, with the device's parameters
    acts = {net.input_layers_[0].GetName(): net.input_layers_[0].GetState().ToNumpy()}
    for l in net.layers_:
        if l.IsInput():
            continue
        e = l.incoming_edge_[0]
        src = e.GetSource()
        x = acts[src.GetName()]
        C, H, W = src.GetNumChannels(), src.GetSizeY(), src.GetSizeX()
        if isinstance(e, ConvEdge):
            d = e.conv_desc_
            g = Geom(N, C, H, W, d.num_output_channels, d.kernel_size_y, d.kernel_size_x, d.stride_y, d.stride_x, -d.padding_y, -d.padding_x)
            w = e.GetWeight().ToNumpy().reshape(g.filt_shape())
            y = oracle.port.conv_up(g, x.reshape(g.in_shape()), w)
            y = oracle.port.add_row_vec(y.reshape(g.F, -1), e.GetBias().ToNumpy().reshape(-1)).reshape(-1)
        elif isinstance(e, MaxPoolEdge):
            d = e.conv_desc_
            g = Geom(N, C, H, W, C, d.kernel_size_y, d.kernel_size_x, d.stride_y, d.stride_x, -d.padding_y, -d.padding_x)
            y = oracle.port.max_pool(g, x.reshape(g.in_shape())).reshape(-1)
        elif isinstance(e, ResponseNormEdge):
            y = oracle.port.rnorm(x.reshape(C, H, W, N), e.num_filters_response_norm_, e.add_scale_, e.pow_scale_, e.blocked_).reshape(-1)
        elif isinstance(e, FCEdge):
            Fo = l.GetNumChannels()
            w = e.GetWeight().ToNumpy()             # (D, F) numpy view of (F, D) col-major
            y = oracle.port.dot(np.ascontiguousarray(x.reshape(-1, N)), w, np.zeros((Fo, N), np.float32), 0.0, 1.0, False, True)
            y = oracle.port.add_row_vec(y, e.GetBias().ToNumpy().reshape(-1)).reshape(-1)
        if l.is_relu:
            y = oracle.port.lower_bound(y, 0.0)
        if l.IsOutput():
            y = oracle.port.softmax_row_major(y.reshape(l.GetNumChannels(), N)).reshape(-1)
        acts[l.GetName()] = y
        got = l.GetState().ToNumpy().reshape(-1)
        assert rel_err(got, y) < TOL, (l.GetName(), rel_err(got, y))


def nets(which):
    from convnet_amd import models
    if which == "tiny_alex":
        return small_alexnet()
    if which == "nin67":   # the reference's second ImageNet model (CONV_ONETOONE layers) at 67x67, 10 classes, no dropout
        return models.alexnet_nin(image_size=67, num_classes=10, dropout=False)
    return {"mnist_conv": models.mnist_conv, "lenet5": models.lenet5}[which]()


@pytest.mark.parametrize("which,fused", [("tiny_alex", False), ("tiny_alex", True), ("lenet5", False), ("mnist_conv", True),
                                         ("nin67", False), ("nin67", True)])
def test_bprop_gradients_match_cpu_oracle_whole_net(gpu, which, fused):
    """Analytic-vs-analytic: every layer derivative and every weight/bias gradient of one
    Fprop/ComputeDeriv/Bprop equals the CPU oracle's (conv dgrad+wgrad, pool undo with ties,
    response-norm undo, FC, 1x1 conv, shared-bias gradients), fused and unfused."""
    from oracle_net import forward_backward
    text = nets(which)
    N = 8
    net = build(text, N, fused=fused)
    for l in net.layers_:
        l.ResetAddOrOverwrite()
    net.GetBatch(net.train_dataset_)
    x = net.input_layers_[0].GetState().ToNumpy()
    labels = net.output_layers_[0].GetData().ToNumpy().reshape(-1)
    net.Fprop(True)
    net.ComputeDeriv()
    net.Bprop()
    # The deep NIN net (~190k ReLU units per image behind reductions up to 6912 terms) always has a few units whose
    # pre-activation is within fp32 rounding of 0 and gate differently on GPU and CPU; each flip colours everything
    # upstream of it (seen: 1 unit of 98,304 -> 10 % of one image's conv1 derivative).  For that net the backward ops are
    # teacher-forced: every op gets the device's own states / incoming derivatives and must reproduce the device's output.
    force = None
    if which == "nin67":
        force = ({l.GetName(): l.GetState().ToNumpy().reshape(-1) for l in net.layers_},
                 {l.GetName(): l.GetDeriv().ToNumpy().reshape(-1) for l in net.layers_ if not l.IsInput()})
    acts, derivs, grads = forward_backward(net, x, labels, force=force)
    for l in net.layers_:
        assert rel_err(l.GetState().ToNumpy().reshape(-1), acts[l.GetName()]) < TOL, ("state", l.GetName())
        if l.GetName() in derivs and not l.IsInput():
            assert rel_err(l.GetDeriv().ToNumpy().reshape(-1), derivs[l.GetName()]) < TOL, ("deriv", l.GetName())
    for e in net.edges_:
        if e.GetName() in grads:
            dw, db = grads[e.GetName()]
            assert rel_err(e.GetGradWeight().ToNumpy().reshape(-1), dw) < TOL, ("dW", e.GetName())
            assert rel_err(e.GetGradBias().ToNumpy().reshape(-1), db) < TOL, ("db", e.GetName())


@pytest.mark.parametrize("which", ["tiny_alex", "mnist_conv", "lenet5", "nin67"])
def test_fused_equals_unfused_forward_backward_and_update(gpu, which):
    text = nets(which)
    N = 32
    a, b = build(text, N, fused=False), build(text, N, fused=True)
    copy_params(a, b)
    for net in (a, b):
        for l in net.layers_:
            l.ResetAddOrOverwrite()
        net.GetBatch(net.train_dataset_)
        net.Fprop(True)       # no dropout in these nets: deterministic
        net.ComputeDeriv()
        net.Bprop()
    for la, lb in zip(a.layers_, b.layers_):
        assert rel_err(la.GetState().ToNumpy(), lb.GetState().ToNumpy()) < 1e-5, la.GetName()
    ga, gb = a.grad_parameters_.ToNumpy(), b.grad_parameters_.ToNumpy()
    assert rel_err(ga, gb) < 1e-5
    # fused mode counted correct predictions on device; unfused computes them per step on the host
    assert abs(b.ReadCorrectCount() - a.GetLoss()[0]) < 0.5
    a.UpdateWeights()
    b.UpdateWeights()
    assert rel_err(a.parameters_.ToNumpy(), b.parameters_.ToNumpy()) < 1e-6


@pytest.mark.parametrize("which,batch", [("lenet5", 16), ("tiny_alex", 8)])
def test_grad_checker_port_runs_the_reference_flow(gpu, which, batch):
    """apps/run_grad_check.cc's flow (random fill of inputs by the back-end's RNG, label 0) through the port: every flagged edge
    is checked, the analytic gradient is finite and somewhere non-zero, and at least one check passes the reference's rule.
    The parity GATE — pass wherever the reference's CPU run passes, at the same point — is tests/test_grad_check_strict.py."""
    from convnet_amd import models
    from convnet_amd.grad_check import GradChecker
    text = {"tiny_alex": small_alexnet(grad_check=True), "lenet5": models.lenet5(grad_check=True)}[which]
    net = build(text, batch, fused=False, cls=GradChecker)
    res = net.Run()
    assert len(res) >= 5
    passed = 0
    for name, r in res.items():
        for what in ("weights", "bias"):
            ok, a, numerical = r[what]
            assert np.all(np.isfinite(a)) and len(numerical) >= 1
            passed += bool(ok)
    assert passed > 0 and any(np.any(r["weights"][1]) for r in res.values())


def test_training_fits_a_fixed_batch_and_dropout_net_stays_finite(gpu):
    """TrainOneBatch end to end (fused path): SGD+momentum memorises one fixed synthetic batch
    (CE loss falls, accuracy rises); the same net with dropout 0.4 trains without NaN/Inf."""
    N = 64
    net = build(small_alexnet(dropprob=0.0), N, fused=True)
    out = net.output_layers_[0]

    def ce_loss():
        for l in net.layers_:
            l.ResetAddOrOverwrite()
        net.GetBatch(net.train_dataset_)
        net.Fprop(False)
        return out.GetLoss() / N

    loss0 = ce_loss()
    net.TrainOneBatch()
    first = net.ReadCorrectCount()
    for _ in range(150):
        net.TrainOneBatch()
    net.ReadCorrectCount()
    net.TrainOneBatch()
    last = net.ReadCorrectCount()
    loss1 = ce_loss()
    assert loss1 < 0.9 * loss0, (loss0, loss1)     # eps=0.01, 151 steps: 2.57 -> 2.11 measured
    assert last >= first + 4, (first, last)
    drop = build(small_alexnet(dropprob=0.4), N, fused=True)
    for _ in range(20):
        drop.TrainOneBatch()
    assert np.isfinite(drop.parameters_.ToNumpy()).all()
    assert 0 <= drop.ReadCorrectCount() <= 20 * N


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["update", "wgrad", "both"])
def test_side_stream_updates_are_bit_identical_to_serial_updates(gpu, mode):
    """overlap_update enqueues each edge's optimizer step on a second stream during Bprop, overlap_wgrad each edge's weight
    gradient; weights, momentum history and gradients after several steps must equal the serial run bit for bit (dropout off:
    the RNG stream is shared state)."""
    from convnet_amd.convnet import ConvNet
    from convnet_amd.datahandler import SyntheticDataHandler
    nets = []
    for overlap in (False, True):
        net = ConvNet(small_alexnet(dropprob=0.0), fused=True, overlap_update=overlap and mode in ("update", "both"),
                      overlap_wgrad=overlap and mode in ("wgrad", "both"))
        net.SetBatchsize(32)
        net.SetupDataset(SyntheticDataHandler(net, 32, seed=11, num_batches=2))
        net.AllocateMemory(False)
        nets.append(net)
    copy_params(nets[0], nets[1])
    for _ in range(6):
        for net in nets:
            net.TrainOneBatch()
    assert nets[1].side_stream_ is not None and nets[0].side_stream_ is None
    import torch
    torch.cuda.synchronize()
    for name in ("parameters_", "history_", "grad_parameters_"):
        a, b = getattr(nets[0], name).ToNumpy().reshape(-1), getattr(nets[1], name).ToNumpy().reshape(-1)
        for (off, n), (off1, n1) in zip(nets[0].edge_slices_.values(), nets[1].edge_slices_.values()):
            assert (off, n) == (off1, n1)   # the 128-float padding between slices is never written: skip it
            bad = np.flatnonzero(a[off:off + n] != b[off:off + n])
            assert bad.size == 0, (name, off, bad.size, bad[:8].tolist())


@pytest.mark.gpu
def test_chunk_datahandler_stages_batches_like_the_reference_pipeline(gpu):
    """GPU-side input staging: chunk on the GPU as (dims, cases), per-batch random crop + flip +
    transpose by extract_patches into the input layer, mean/std normalisation at load, in-place column shuffle at chunk
    wrap with images and labels kept paired.  Labels are the case ids, so every staged image can be re-derived on the
    CPU from the ORIGINAL chunk with the offsets the device sampled — across two shuffles."""
    from convnet_amd.convnet import ConvNet
    from convnet_amd.datahandler import ChunkDataHandler
    rng = np.random.default_rng(3)
    cases, colors, S, crop, bs = 21, 3, 41, 35, 8
    images = rng.integers(0, 256, (cases, colors * S * S)).astype(np.float32)
    labels = np.arange(cases, dtype=np.float32) % 10
    ids = np.arange(cases, dtype=np.float32)
    mean, std = rng.random(colors * S * S).astype(np.float32) * 255, (rng.random(colors * S * S).astype(np.float32) + 0.5) * 60
    net = ConvNet(small_alexnet(), fused=True)
    net.SetBatchsize(bs)
    dh = ChunkDataHandler(images, ids, bs, S, crop, colors, translate=True, flip=True, mean=mean, std=std, randomize=True, seed=4)
    net.SetupDataset(dh)
    net.AllocateMemory(False)
    normed = oracle.port.div_by_col_vec(oracle.port.add_col_mult(images.copy(), mean, -1.0), std)
    seen = []
    for step in range(7):      # 7 * 8 = 56 cases > 2 chunks: two shuffles
        net.GetBatch(dh)
        got = net.input_layers_[0].GetState().ToNumpy().reshape(colors, crop, crop, bs)
        case = net.output_layers_[0].GetData().ToNumpy().reshape(-1).astype(int)
        wo, ho, fl = (m.ToNumpy().reshape(-1) for m in (dh.width_offset_, dh.height_offset_, dh.flip_bit_))
        assert (wo >= 0).all() and (wo < S - crop + 1).all() and (ho >= 0).all() and (ho < S - crop + 1).all()
        ref = oracle.port.extract_patches(np.ascontiguousarray(normed[case]), wo, ho, fl, S, S, crop, crop)
        assert np.array_equal(got, ref), step
        seen.extend(case.tolist())
    assert len(set(seen[:16])) == 16 and set(seen) == set(range(cases))     # a chunk pass never repeats a case
    assert seen[:8] == list(range(8)) and sorted(seen[16:32]) != seen[16:32]  # first pass in order, later passes shuffled
    # and a real training step runs off it
    dh2 = ChunkDataHandler(images, labels, bs, S, crop, colors, mean=mean, std=std, seed=5)
    net.SetupDataset(dh2)
    for _ in range(3):
        net.TrainOneBatch()
    assert np.isfinite(net.parameters_.ToNumpy()).all()


@pytest.mark.gpu
def test_hdf5_checkpoint_resume_is_bit_exact_and_uses_the_reference_layout(gpu, tmp_path):
    """ConvNet::Save / Load (src/convnet.cc:666-684,737-751): train 3 steps, save, load into a differently initialised
    net, continue both for 2 steps on the same data -> bit-identical parameters, momentum history, iteration and
    optimizer step counters.  Dataset / attribute names are the reference's."""
    from convnet_amd import hdf5io
    a = build(small_alexnet(), 16, fused=True, seed_data=9)
    for _ in range(3):
        a.TrainOneBatch()
    path = str(tmp_path / "ckpt.h5")
    a.Save(path)
    with hdf5io.File(path) as f:
        e = a.GetEdgeByName("input:c1")
        assert f.ReadHDF5Shape("input:c1:weight") == (e.GetWeight().GetRows(), e.GetWeight().GetCols())
        assert f.Has("input:c1:bias") and f.Has("input:c1:weight_gradient_history") and f.Has("f7:output:bias_gradient_history")
        assert f.ReadHDF5IntAttr("input:c1:weight_step", -1) == 3 and f.ReadHDF5IntAttr("__current_iter__", -1) == 3
        assert f.ReadHDF5IntAttr("__lr_reduce_counter__", -1) == 0
        assert np.array_equal(f.ReadHDF5CPU(e.GetWeight().GetNumEls(), "input:c1:weight"), e.GetWeight().ToNumpy().reshape(-1))
    b = build(small_alexnet(), 16, fused=True, seed_data=9)
    b.parameters_.Mult(0.5)            # make sure Load really overwrites
    b.Load(path)
    assert b.current_iter_ == 3 and b.GetEdgeByName("f6:f7").weight_optimizer_.step_ == 3
    for net in (a, b):
        net.train_dataset_.pos_ = 0
        for _ in range(2):
            net.TrainOneBatch()
    for ea, eb in zip(a.edges_, b.edges_):
        if ea.GetParameterMemoryRequirement():
            assert np.array_equal(ea.GetWeight().ToNumpy(), eb.GetWeight().ToNumpy()), ea.GetName()
            assert np.array_equal(ea.GetBias().ToNumpy(), eb.GetBias().ToNumpy()), ea.GetName()
            assert np.array_equal(ea.weight_optimizer_.gradient_history_.ToNumpy(), eb.weight_optimizer_.gradient_history_.ToNumpy())
    # an fprop-only net (no optimizer state allocated) loads the same file
    from convnet_amd.convnet import ConvNet
    from convnet_amd.datahandler import SyntheticDataHandler
    c = ConvNet(small_alexnet(), fused=True)
    c.SetBatchsize(16)
    c.SetupDataset(SyntheticDataHandler(c, 16, seed=9, num_batches=1))
    c.AllocateMemory(True)
    c.Load(path)
    assert np.array_equal(c.GetEdgeByName("c4:c5").GetWeight().ToNumpy(), np.asarray(hdf5io.File(path).ReadHDF5CPU(
        c.GetEdgeByName("c4:c5").GetWeight().GetNumEls(), "c4:c5:weight")).reshape(c.GetEdgeByName("c4:c5").GetWeight().ToNumpy().shape))


@pytest.mark.gpu
def test_train_loop_validate_polyak_lr_schedule_checkpoint_and_feature_extraction(gpu, tmp_path):
    """ConvNet::Train / Validate / ExtractFeatures (src/convnet.cc:571-657,866-1011) on a tiny net: cadence of the
    train / validation log, Polyak queue, learning-rate cut on a flat validation curve, checkpoint + resume from
    __current_iter__, and the (cases, dims) feature file."""
    from convnet_amd import hdf5io
    from convnet_amd.convnet import ConvNet
    from convnet_amd.datahandler import ChunkDataHandler
    extra = (f'print_after: 5\nvalidate_after: 10\nsave_after: 20\nreduce_lr_factor: 0.5\nreduce_lr_num_steps: 2\nreduce_lr_max: 3\n'
             f'reduce_lr_threshold: 1.0\npolyak_after: 2\npolyak_queue_size: 3\ncheckpoint_dir: "{tmp_path}"\ntimestamp: "t0"')
    text = small_alexnet().replace("print_after: 100", extra)
    rng = np.random.default_rng(2)
    S, colors, bs = 35, 3, 16
    def data(cases, seed, randomize=False):
        r = np.random.default_rng(seed)
        return ChunkDataHandler(r.standard_normal((cases, colors * S * S)).astype(np.float32), r.integers(0, 10, cases), bs, S, S, colors,
                                translate=False, flip=False, randomize=randomize, seed=seed)
    net = ConvNet(text, fused=True)
    net.SetBatchsize(bs)
    net.SetupDataset(data(64, 1, True))
    net.SetupValidationDataset(data(40, 2))
    net.AllocateMemory(False)
    lines = []
    eps0 = net.GetEdgeByName("f6:f7").weight_optimizer_.epsilon_
    hist = net.Train(max_iter=30, log=lines.append)
    assert [t[0] for t in hist["train"]] == [5, 10, 15, 20, 25, 30] and all(0.0 <= t[1] <= 1.0 for t in hist["train"])
    assert [v[0] for v in hist["val"]] == [10, 20, 30] and all(0.0 <= v[1] <= 1.0 for v in hist["val"])
    # threshold 1.0 makes every comparison "flat": the first qualifying validation (2 values, step 20) arms, step 30 cuts
    assert hist["lr_reductions"] == 1 and net.lr_reduce_counter_ == 1
    assert net.GetEdgeByName("f6:f7").weight_optimizer_.epsilon_ == pytest.approx(eps0 * 0.5)
    assert net.polyak_queue_full_ and len(net.polyak_parameters_) == 3
    # ConvNet::Train stamps every run (src/convnet.cc:875, TimestampModel :830-838): the model came in with the stamp "t0" of an
    # earlier run, this run appended its own, checkpoints under it and wrote the stamped model next to the checkpoint
    ckpt = net.GetCheckpointFilename()
    stamps = net.model_.timestamp
    assert len(stamps) == 2 and stamps[0] == "t0" and ckpt.endswith(f"tiny_alex_{stamps[1]}.h5") and os.path.exists(ckpt)
    stamped = ckpt[:-3] + ".pbtxt"
    assert os.path.exists(stamped) and not os.path.exists(os.path.join(str(tmp_path), "tiny_alex_t0.h5"))
    with hdf5io.File(ckpt) as f:
        assert f.ReadHDF5IntAttr("__current_iter__", -1) == 30 and f.ReadHDF5IntAttr("__lr_reduce_counter__", -1) == 1
    # resume the reference's way — from the stamped model file: a fresh net picks up iteration, optimizer steps and the reduced
    # learning rate, and trains on
    net2 = ConvNet(stamped, fused=True)
    net2.SetBatchsize(bs)
    net2.SetupDataset(data(64, 1, True))
    net2.AllocateMemory(False)
    net2.Load()
    assert net2.current_iter_ == 30 and net2.GetEdgeByName("f6:f7").weight_optimizer_.epsilon_ == pytest.approx(eps0 * 0.5)
    assert np.array_equal(net2.GetEdgeByName("c4:c5").GetWeight().ToNumpy(), net.GetEdgeByName("c4:c5").GetWeight().ToNumpy())
    hist2 = net2.Train(max_iter=35, log=lines.append, checkpoint=False)
    assert [t[0] for t in hist2["train"]] == [35] and net2.current_iter_ == 35
    # validation is deterministic and leaves the weights alone
    v1, v2 = net2.Validate(data(40, 2)), net2.Validate(data(40, 2))
    assert v1 == v2 and len(v1) == 1
    # features: 40 cases through batches of 16 (last batch contributes 8), one case per row
    out = str(tmp_path / "feat.h5")
    val = data(40, 2)
    net2.ExtractFeatures(val, ["f7", "output"], out)
    with hdf5io.File(out) as f:
        assert f.ReadHDF5Shape("f7") == (40, 40) and f.ReadHDF5Shape("output") == (10, 40)    # (dims, cases) column-major view
        probs = f.ReadHDF5CPU(400, "output").reshape(40, 10)
    assert np.allclose(probs.sum(axis=1), 1.0, atol=1e-5)
    val.Seek(0)
    for l in net2.layers_:
        l.ResetAddOrOverwrite()
    val.GetBatch(net2.data_layers_)
    net2.Fprop(False)
    assert np.array_equal(probs[:16], net2.GetLayerByName("output").GetState().ToNumpy().T)
    val.Seek(32)               # the straddling batch: cases 32..39 then 0..7
    for l in net2.layers_:
        l.ResetAddOrOverwrite()
    val.GetBatch(net2.data_layers_)
    net2.Fprop(False)
    last = net2.GetLayerByName("output").GetState().ToNumpy().T
    assert np.array_equal(probs[32:], last[:8]) and np.array_equal(probs[:8], last[8:])
