"""GPU: the data-parallel path with the REAL ConvNet (SURVEY §4 DP invariants).

 * two processes, one model replica each (both on cuda:0 of this 1-GPU box, gloo transport standing in for RCCL), each training
   on its half of a global batch: replicas stay bit-identical after 3 steps, the exchanged gradient of P x (B/P) equals the
   1 x B gradient (src/convnet.cc:429-431: sum over ranks / num_processes), also for a net with a TIED edge and with
   the side-stream optimizer overlap; the 3-step result equals the reference's own single-process host at batch B;
 * one rank, backend nccl (= RCCL): the overlap machinery itself (events, communication stream, bucket ranges of a merging DAG)
   leaves a training run bit-identical to the run without exchange — through torch.distributed AND through the library's own
   exchange entries (convnet_hip_comm_*, the path a C/C++ host uses);
 * the exchange entries called directly through the C ABI, and driven by the reference's unmodified C++ host through the
   subclass of oracle/seam/seam_host.cc."""
import ctypes
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import ref_host  # noqa: E402
from golden_cases import rel_err  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def tied_net():
    """input -> a -> b -> c with the b->c convolution TIED to a->b (same 8->8 3x3 filters used twice), pool, fc, softmax."""
    from convnet_amd import models
    L, C, P, F = models._layer, models._conv, models._pool, models._fc
    s = models._header("tied")
    s += L("input", 3, size=12)
    s += L("a", 8, "RECTIFIED_LINEAR") + L("b", 8, "RECTIFIED_LINEAR") + L("c", 8, "RECTIFIED_LINEAR") + L("p", 8)
    s += L("f", 16, "RECTIFIED_LINEAR") + L("output", 5, "SOFTMAX")
    s += C("input", "a", 3, 1, 1) + C("a", "b", 3, 1, 1) + C("b", "c", 3, 1, 1, grad_check='  tied_to: "a:b"\n')
    s += P("c", "p", 2, 2, 0) + F("p", "f") + F("f", "output")
    return s


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


_CHILD = r'''
import os, sys
sys.path[:0] = [{here!r}, {root!r}]
import numpy as np, torch, torch.distributed as dist
import ref_host
from convnet_amd.convnet import ConvNet
from convnet_amd.matrix import Matrix
from test_data_parallel_gpu import SliceData, net_text
rank, world, port, which, B, overlap, side, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], int(sys.argv[5]), sys.argv[6] == "1", sys.argv[7] == "1", sys.argv[8]
assert torch.cuda.is_available()
Matrix.SetupCUDADevice(0)
exchange = None
if world > 1:
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", port
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from convnet_amd.data_parallel import GradientExchange
    exchange = GradientExchange(bucket_bytes=4096, overlap=overlap)
net = ConvNet(net_text(which), fused=True, process_id=rank, num_processes=world, exchange=exchange, overlap_update=side)
net.SetBatchsize(B // world)
net.SetupDataset(SliceData(net, B, rank, world, seed=7, num_batches=2))
net.AllocateMemory(False)
p0 = np.load(out + ".p0.npy")
net.parameters_.FromNumpy(p0)
net.TrainOneBatch()
torch.cuda.synchronize()
g1 = net.grad_parameters_.ToNumpy().reshape(-1).copy()     # the exchanged (averaged) gradient of step 1, L2 decay etc. applied later
for _ in range(2):
    net.TrainOneBatch()
torch.cuda.synchronize()
np.savez(out + f".r{{rank}}.npz", g1=g1, p3=net.parameters_.ToNumpy().reshape(-1))
if world > 1:
    dist.destroy_process_group()
'''


def net_text(which):
    from test_net_gpu import small_alexnet
    from test_reference_host import dag_net
    return {"tiny_alex": small_alexnet, "tied": tied_net, "dag": dag_net}[which]()


class SliceData:
    """Rank r's columns [r*B/P, (r+1)*B/P) of the data shim's global hash batches (seam_datahandler.h / ref_host.hash_batch)."""

    def __init__(self, net, B, rank, world, seed, num_batches):
        from convnet_amd.matrix import Matrix
        self.batch_size_, self.pos_, self.batches_ = B // world, 0, []
        lo, hi = rank * (B // world), (rank + 1) * (B // world)
        for b in range(num_batches):
            per = {}
            for l in net.data_layers_:
                m = Matrix()
                if l.IsInput():
                    dims = l.GetSizeY() * l.GetSizeX() * l.GetSizeT() * l.GetNumChannels()
                    m.AllocateGPUMemory(hi - lo, dims)
                    m.FromNumpy(np.ascontiguousarray(ref_host.hash_batch(seed, b, dims * B, True).reshape(dims, B)[:, lo:hi]))
                else:
                    m.AllocateGPUMemory(hi - lo, 1)
                    m.FromNumpy(ref_host.hash_batch(seed, b, B, False, l.GetNumChannels())[lo:hi])
                per[l.GetName()] = m
            self.batches_.append(per)

    def GetBatchSize(self):
        return self.batch_size_

    def GetDataSetSize(self):
        return self.batch_size_ * len(self.batches_)

    def Seek(self, row):
        self.pos_ = row // self.batch_size_

    def Sync(self):
        pass

    def GetBatch(self, data_layers):
        b = self.batches_[self.pos_ % len(self.batches_)]
        self.pos_ += 1
        for l in data_layers:
            (l.GetState() if l.IsInput() else l.GetData()).Set(b[l.GetName()])


def _p0(which, tmp, B):
    """Parameters from the integer hash at He scale, laid out by the reference's own host (describe)."""
    host = ref_host.RefHost(ref_host.CPU_SO)
    m, d = ref_host.write_configs(tmp, net_text(which), B, 2, 7, which)
    layers, edges, total = host.describe(m, d)
    slices, end = ref_host.slices_from_describe(layers, edges)
    assert end == total
    return host, m, d, ref_host.golden_params(total, 7, slices)


def _run(world, which, B, overlap, side, out, p0):
    np.save(out + ".p0.npy", p0)
    port = str(_free_port())
    code = _CHILD.format(here=HERE, root=ROOT)
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), str(world), port, which, str(B), "1" if overlap else "0", "1" if side else "0", out],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    for p in procs:
        so, se = p.communicate(timeout=300)
        assert p.returncode == 0, se[-3000:]
    return [np.load(out + f".r{r}.npz") for r in range(world)]


@pytest.mark.parametrize("which,overlap,side", [("tiny_alex", True, False), ("tied", True, True), ("dag", False, False)])
def test_two_replicas_of_the_real_convnet_stay_identical_and_match_one_process(which, overlap, side, tmp_path):
    if not os.path.exists(ref_host.CPU_SO):
        pytest.skip("oracle/_ref/libref_host_cpu.so not built")
    B = 8
    host, m, d, p0 = _p0(which, tmp_path, B)
    two = _run(2, which, B, overlap, side, str(tmp_path / "two"), p0)
    one = _run(1, which, B, overlap, False, str(tmp_path / "one"), p0)[0]
    # (1) replicas bit-identical: gradient after the exchange and parameters after 3 momentum steps
    assert np.array_equal(two[0]["g1"], two[1]["g1"]) and np.array_equal(two[0]["p3"], two[1]["p3"])
    # (2) P x (B/P) == 1 x B: the mean of the two half-batch gradients is the full-batch gradient (convnet.cc:429-431)
    assert rel_err(two[0]["g1"], one["g1"]) < 1e-5, rel_err(two[0]["g1"], one["g1"])
    assert rel_err(two[0]["p3"], one["p3"]) < 1e-5
    # (3) and equals the reference's own single-process host on its CPU path at batch B (tied edges included)
    p3_ref, _, _ = host.train(m, d, 3, p0)
    assert rel_err(two[0]["p3"], p3_ref) < 1e-4, rel_err(two[0]["p3"], p3_ref)
    assert not np.array_equal(p3_ref, p0)


@pytest.fixture(scope="module")
def nccl_one_rank():
    import torch
    import torch.distributed as dist
    assert torch.cuda.is_available()
    from convnet_amd.matrix import Matrix
    Matrix.SetupCUDADevice(0)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


@pytest.mark.parametrize("transport", ["torch", "abi"])
@pytest.mark.parametrize("which,side", [("tiny_alex", False), ("dag", True), ("tied", True), ("tiny_alex", "wgrad"), ("tied", "wgrad"), ("dag", "both")])
def test_one_rank_rccl_overlap_path_is_bit_identical_to_no_exchange(nccl_one_rank, transport, which, side):
    """Events, communication stream, bucket ranges (several per bucket on the DAG) and the per-edge waits, with a world of one:
    the all-reduce is the identity, so any difference from the plain run is a synchronisation or range bug."""
    import torch
    from convnet_amd.convnet import ConvNet
    from convnet_amd.data_parallel import GradientExchange
    from convnet_amd import _lib
    B = 8
    runs = []
    for ex in (None, "x"):
        exchange = GradientExchange(bucket_bytes=2048, overlap=True, transport=transport) if ex else None
        net = ConvNet(net_text(which), fused=True, exchange=exchange, overlap_update=side in (True, "both") and ex is not None,
                      overlap_wgrad=side in ("wgrad", "both") and ex is not None)   # "wgrad": weight gradients (and their all-reduce posts) on the second stream
        net.SetBatchsize(B)
        net.SetupDataset(SliceData(net, B, 0, 1, seed=9, num_batches=2))
        net.AllocateMemory(False)
        if ex:
            assert len(exchange.buckets_) >= 2
            net.parameters_.FromNumpy(runs[0][0])
        else:
            runs.append((net.parameters_.ToNumpy().reshape(-1).copy(),))
        for _ in range(3):
            net.TrainOneBatch()
        torch.cuda.synchronize()
        if transport == "abi" and ex:
            assert _lib.lib.convnet_hip_comm_sync() == 0
        runs.append(net.parameters_.ToNumpy().reshape(-1).copy())
        if ex:
            if transport == "abi":     # a torch collective while the library's communicator is live: drained first, then legal
                assert exchange.SumScalars([3.0, 4.5]) == [3.0, 4.5]
            exchange.Close()           # destroys the library's communicator (idempotent; also registered atexit)
            exchange.Close()
            assert _lib.lib.convnet_hip_comm_size() == 1
    assert np.array_equal(runs[1], runs[2]) and not np.array_equal(runs[0][0], runs[2])


def test_abi_transport_without_overlap_is_serial_and_slot_plan_is_checked_up_front(nccl_one_rank):
    """`overlap=False` is honoured by the C-ABI transport (the compute stream waits for each bucket right after posting
    it), and a bucket plan that needs more slots than the library has is refused at Register, not half-way through a backward pass."""
    import torch
    from convnet_amd import _lib
    from convnet_amd.convnet import ConvNet
    from convnet_amd.data_parallel import GradientExchange
    assert _lib.lib.convnet_hip_comm_max_slots() == 256
    runs = []
    for overlap in (True, False):
        ex = GradientExchange(bucket_bytes=2048, overlap=overlap, transport="abi")
        net = ConvNet(net_text("dag"), fused=True, exchange=ex, overlap_update=False, overlap_wgrad=False)
        net.SetBatchsize(8)
        net.SetupDataset(SliceData(net, 8, 0, 1, seed=9, num_batches=2))
        net.AllocateMemory(False)
        if runs:
            net.parameters_.FromNumpy(runs[0][0])
        p0 = net.parameters_.ToNumpy().reshape(-1).copy()
        for _ in range(2):
            net.TrainOneBatch()
        torch.cuda.synchronize()
        runs.append((p0, net.parameters_.ToNumpy().reshape(-1).copy()))
        ex.Close()
    assert np.array_equal(runs[0][1], runs[1][1])

    class TooMany(GradientExchange):
        def _flat_ranges(self, bucket):
            return [(i, i + 1) for i in range(300)]
    ex = TooMany(bucket_bytes=1 << 30, overlap=True, transport="abi")
    net = ConvNet(net_text("tiny_alex"), fused=True, exchange=None)
    net.SetBatchsize(8)
    net.SetupDataset(SliceData(net, 8, 0, 1, seed=9, num_batches=2))
    net.AllocateMemory(False)
    with pytest.raises(RuntimeError, match="slots per step"):
        ex.Register(net)
    ex.Close()


_CHILD_RCCL = r'''
import os, sys
sys.path[:0] = [{here!r}, {root!r}]
import numpy as np, torch, torch.distributed as dist
from convnet_amd.convnet import ConvNet
from convnet_amd.matrix import Matrix
from convnet_amd.data_parallel import GradientExchange
from test_data_parallel_gpu import SliceData, net_text
rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
Matrix.SetupCUDADevice(rank)
os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", port
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
res = {{}}
p0 = np.load(out + ".p0.npy")
for transport in ("torch", "abi"):
    ex = GradientExchange(bucket_bytes=4096, overlap=True, transport=transport)
    net = ConvNet(net_text("tiny_alex"), fused=True, process_id=rank, num_processes=world, exchange=ex, overlap_update=True, overlap_wgrad=True)
    net.SetBatchsize(8 // world)
    net.SetupDataset(SliceData(net, 8, rank, world, seed=7, num_batches=2))
    net.AllocateMemory(False)
    net.parameters_.FromNumpy(p0)
    for _ in range(3):
        net.TrainOneBatch()
    sums = ex.SumScalars([float(rank + 1), 10.0])       # a torch collective beside the library's live communicator
    torch.cuda.synchronize()
    res[transport] = net.parameters_.ToNumpy().reshape(-1).copy()
    res[transport + "_sum"] = np.array(sums)
    ex.Close()
np.savez(out + f".r{{rank}}.npz", **res)
dist.destroy_process_group()
'''


def test_two_gpu_rccl_torch_and_abi_transports_are_bitwise_equal(tmp_path):
    """Needs two GPUs (skipped on the 1-GPU test box): two RCCL ranks, one per GPU, the same run through torch.distributed's
    all_reduce(AVG) and through the library's convnet_hip_comm_* entries (sum, then a true division by the rank count): parameters
    after 3 overlapped steps bit-identical across ranks AND across transports; a torch collective (SumScalars) issued while the
    library's own communicator is live completes (two communicators in one process)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs: RCCL refuses two ranks on one device")
    from convnet_amd import pbtxt
    from convnet_amd.convnet import ConvNet
    net = ConvNet(net_text("tiny_alex"))
    net.SetBatchsize(4)
    rng = np.random.default_rng(3)
    out = str(tmp_path / "rccl")
    # parameter count without touching the GPU in this process: built graph + the reference's slice layout
    total = sum(((n + 127) // 128) * 128 for n in (e.GetParameterMemoryRequirement() for e in net.edges_) if n)
    np.save(out + ".p0.npy", (rng.standard_normal(total) * 0.05).astype(np.float32))
    port = str(_free_port())
    code = _CHILD_RCCL.format(here=HERE, root=ROOT)
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), "2", port, out], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(2)]
    for p in procs:
        so, se = p.communicate(timeout=300)
        assert p.returncode == 0, se[-3000:]
    r0, r1 = np.load(out + ".r0.npz"), np.load(out + ".r1.npz")
    for t in ("torch", "abi"):
        assert np.array_equal(r0[t], r1[t]), f"replicas diverged ({t})"
        assert np.array_equal(r0[t + "_sum"], [3.0, 20.0])
    assert np.array_equal(r0["torch"], r0["abi"]), "both transports sum, then divide by the rank count: bit for bit"


def test_exchange_entries_through_the_c_abi_one_rank():
    """include/convnet_hip.h convnet_hip_comm_*: id, init (dlopens librccl), broadcast, all-reduce of a sub-range posted behind
    compute-stream work, device-side wait, sync, destroy; error paths (bad slot, out-of-range, wait on an empty slot, no comm)."""
    import torch
    assert torch.cuda.is_available()
    from convnet_amd import _lib
    from convnet_amd.matrix import Matrix
    Matrix.SetupCUDADevice(0)
    lib = _lib.lib
    m = Matrix()
    m.AllocateGPUMemory(1, 5000)
    x = np.arange(5000, dtype=np.float32)
    m.FromNumpy(x)
    assert lib.convnet_hip_comm_allreduce_avg(m.GetMat(), 0, 10, 0) != 0         # no communicator yet: refused, not crashed
    idbuf = ctypes.create_string_buffer(128)
    assert lib.convnet_hip_comm_unique_id(idbuf) == 0 and any(idbuf.raw)
    assert lib.convnet_hip_comm_init(0, 1, idbuf.raw) == 0
    assert (lib.convnet_hip_comm_rank(), lib.convnet_hip_comm_size()) == (0, 1)
    assert lib.convnet_hip_comm_init(0, 1, idbuf.raw) != 0                          # double init refused
    assert lib.convnet_hip_comm_broadcast(m.GetMat(), 0) == 0
    m.Mult(2.0)                                                                       # compute-stream work the post must wait for
    assert lib.convnet_hip_comm_allreduce_avg(m.GetMat(), 128, 3000, 3) == 0
    assert lib.convnet_hip_comm_wait(3) == 0
    m.Add(1.0)                                                                        # ordered after the exchange by the wait
    assert np.array_equal(m.ToNumpy().reshape(-1), 2 * x + 1)
    assert lib.convnet_hip_comm_wait(4) != 0                                          # nothing posted there
    assert lib.convnet_hip_comm_allreduce_avg(m.GetMat(), 4000, 2000, 1) != 0         # beyond the matrix
    assert lib.convnet_hip_comm_allreduce_avg(m.GetMat(), 0, 10, 999) != 0            # bad slot
    assert lib.convnet_hip_comm_sync() == 0 and lib.convnet_hip_comm_destroy() == 0
    assert lib.convnet_hip_comm_destroy() == 0                                        # idempotent


def test_reference_cpp_host_trains_through_the_exchange_entries(tmp_path):
    """The reference's unmodified C++ host with the data-parallel SUBCLASS of oracle/seam/seam_host.cc (Bprop posts each edge's
    gradient slice, UpdateWeights waits per edge — INTEGRATION.md §4), rank 0 of 1: bit-identical to its own plain loop, with
    several buckets in flight."""
    import torch
    assert torch.cuda.is_available()
    from convnet_amd import _lib
    from test_net_gpu import small_alexnet
    if not os.path.exists(ref_host.HIP_SO):
        pytest.skip("oracle/_ref/libref_host_hip.so not built")
    ctypes.CDLL(_lib.LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    host = ref_host.RefHost(ref_host.HIP_SO)
    m, d = ref_host.write_configs(tmp_path, small_alexnet(), 8, 2, 5, "dp")
    layers, edges, total = host.describe(m, d)
    slices, _ = ref_host.slices_from_describe(layers, edges)
    p0 = ref_host.golden_params(total, 5, slices)
    plain, _, _ = host.train(m, d, 3, p0)
    for bucket_bytes in (1, 16384, 1 << 30):
        dp = host.train_dp(m, d, 3, p0, rank=0, nranks=1, comm_id=None, bucket_bytes=bucket_bytes)
        assert np.array_equal(dp, plain), bucket_bytes
    assert not np.array_equal(plain, p0)


@pytest.mark.parametrize("N", [256, 64])
def test_bench_configuration_two_streams_and_exchange_are_bit_identical_to_the_serial_run(nccl_one_rank, N):
    """What bench.py times — the real AlexNet 224 x 224 at the benchmark batch, fused entry points, weight gradients AND optimizer steps
    on the second stream — against the same net run serially on one stream, and the same again with a 1-rank gradient exchange on both
    transports: parameters, momentum history and gradients after three steps must agree bit for bit (dropout off: the RNG stream is
    shared state).  Stream-ordering bugs (per-stream scratch arenas, split-K slabs, tail-fix buffers, filter planes) depend on sizes
    and timing, so this runs at the full geometry; tests/test_net_gpu.py holds the small-net version."""
    import torch
    from convnet_amd import models
    from convnet_amd.convnet import ConvNet
    from convnet_amd.datahandler import SyntheticDataHandler
    from convnet_amd.data_parallel import GradientExchange
    from convnet_amd import _lib
    text = models.alexnet(dropprob=0.0)

    def run(overlap, transport=None, start=None):
        ex = GradientExchange(bucket_bytes=8 << 20, overlap=True, transport=transport) if transport else None
        net = ConvNet(text, fused=True, exchange=ex, overlap_update=overlap, overlap_wgrad=overlap)
        net.SetBatchsize(N)
        net.SetupDataset(SyntheticDataHandler(net, N, seed=11, num_batches=2))
        net.AllocateMemory(False)
        if start is not None:
            net.parameters_.FromNumpy(start)
        first = net.parameters_.ToNumpy().reshape(-1).copy()
        for _ in range(3):
            net.TrainOneBatch()
        torch.cuda.synchronize()
        if transport == "abi":
            assert _lib.lib.convnet_hip_comm_sync() == 0
        out = {k: getattr(net, k).ToNumpy().reshape(-1).copy() for k in ("parameters_", "history_", "grad_parameters_")}
        slices = list(net.edge_slices_.values())
        if ex:
            ex.Close()
        del net
        torch.cuda.empty_cache()
        return first, out, slices

    first, serial, slices = run(False)
    assert not np.array_equal(first, serial["parameters_"])
    for overlap, transport in ((True, None), (True, "torch"), (True, "abi")):
        _, got, _ = run(overlap, transport, start=first)
        for name in serial:
            for off, n in slices:   # the 128-float padding between slices is never written: skip it
                bad = np.flatnonzero(serial[name][off:off + n] != got[name][off:off + n])
                assert bad.size == 0, (N, overlap, transport, name, off, int(bad.size), bad[:8].tolist())


def test_bench_launches_two_ranks_and_prints_the_scale_line(tmp_path):
    """`python bench.py --gpus 2` as the driver's SCALE run starts it, on this 1-GPU box: --share-device puts both ranks on GPU 0 and
    swaps RCCL for gloo (RCCL refuses two ranks on one device), everything else is the N-rank path — the self-launch through
    torch.distributed.run, the headline run with the overlapped exchange, the legs beside it, the ONE JSON line.  Asserts what a SCALE
    record needs: the top-level `value` / `config` are BASELINE's configuration 4 — a GLOBAL batch of 256 split over
    the ranks, 128 per GPU here — the weak run (256 per GPU, global 512) is nested as `weak`, `exchange_ms_exposed` = the step with the
    exchange minus the same step without it, `rccl_ranks` counts RCCL ranks only (0 here: gloo carried the exchange), and the replicas
    hold bit-identical parameters after the timed steps."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-device", "--steps", "3", "--warmup", "1",
                          "--no-cpu-baseline", "--no-ref-host", "--no-other-path", "--no-live-traffic"],
                         capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["steps"] == 3 and d["warmup"] == 1
    assert d["config"]["batch_per_gpu"] == 128 and d["config"]["global_batch"] == 256 and d["config"]["parallelism"].startswith("dp2")
    assert abs(d["value"] - 256 * 3 / (d["ms_per_step"] * 3e-3)) < 0.01 * d["value"]      # whole-job images/s over all ranks
    assert d["config4_value"] == d["value"] and d["compute_only_ms_per_step"] > 0
    assert abs(d["exchange_ms_exposed"] - (d["ms_per_step"] - d["compute_only_ms_per_step"])) < 2e-3
    w = d["weak"]
    assert w["scaling"] == "weak" and w["global_batch"] == 512 and w["batch_per_gpu"] == 256 and w["n_gpus"] == 2 and w["value"] > 0
    assert d["replicas_identical"] is True
    assert d["rccl_ranks"] == 0 and w["rccl_ranks"] == 0 and "share_device" in d      # gloo, not RCCL, carried the exchange
    assert d["roofline"] and d["roofline"]["frac"] <= 1.0
