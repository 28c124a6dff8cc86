"""Test helper: ctypes view of the reference's WHOLE host (ConvNet / GradChecker / Layer / Edge / Optimizer, compiled unmodified,
oracle/Makefile target `host`, oracle/seam/seam_host.cc) in its two builds

    oracle/_ref/libref_host_cpu.so   over the reference's CPU Matrix (CPUMatrix.cc + eigenmat)   — the whole-net oracle
    oracle/_ref/libref_host_hip.so   over the reference's GPU Matrix (matrix.cc) + libconvnet_hip.so  — the drop-in demonstration

and the numpy restatement of the batches oracle/seam/seam_datahandler.h feeds them."""
import ctypes
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPU_SO = os.path.join(ROOT, "oracle", "_ref", "libref_host_cpu.so")
HIP_SO = os.path.join(ROOT, "oracle", "_ref", "libref_host_hip.so")


def hash_batch(seed, batch_index, n, is_input, classes=0):
    """seam_datahandler.h GetBatch: lowbias32 of (base + flat element index); inputs uniform, zero mean, unit variance."""
    base = (seed * 0x9E3779B1 + batch_index * 0x85EBCA77 + (0x1234567 if is_input else 0x7654321)) & 0xFFFFFFFF
    x = (np.arange(n, dtype=np.uint64) + base).astype(np.uint32)
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x7FEB352D)
    x ^= x >> np.uint32(15)
    x *= np.uint32(0x846CA68B)
    x ^= x >> np.uint32(16)
    if is_input:
        return (((x >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0) - np.float32(0.5)) * np.float32(3.4641016)).astype(np.float32)
    return (x % np.uint32(classes)).astype(np.float32)


class RefHost:
    def __init__(self, so_path, omp_threads=8):
        self.lib = ctypes.CDLL(so_path)
        if so_path == CPU_SO:
            # eigenmat's OpenMP loops over these tiny layers: a team of every core of a big (or quota-limited) host costs far
            # more in fork/join than the work itself
            gomp = ctypes.CDLL("libgomp.so.1")
            gomp.omp_set_num_threads(max(1, min(omp_threads, os.cpu_count() or 1)))
        self.lib.seam_host_train.restype = ctypes.c_long
        self.lib.seam_host_train.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long,
                                             ctypes.c_void_p, ctypes.c_void_p]
        self.lib.seam_host_batch.restype = ctypes.c_long
        self.lib.seam_host_batch.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long]
        self.lib.seam_host_checkpoint.restype = ctypes.c_long
        self.lib.seam_host_checkpoint.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p,
                                                  ctypes.c_long]
        self.lib.seam_host_reduce_lr.restype = None
        self.lib.seam_host_reduce_lr.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        self.lib.seam_host_describe.restype = ctypes.c_long
        self.lib.seam_host_describe.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_long]
        self.lib.seam_host_sgd.restype = None
        self.lib.seam_host_sgd.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        self.lib.seam_host_grad_check.restype = None
        self.lib.seam_host_grad_check.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p]
        self.lib.seam_host_grad_check_fixed.restype = ctypes.c_int
        self.lib.seam_host_grad_check_fixed.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int]

    def _call(self, model, data, steps, p_in, cap):
        out = np.zeros(cap, np.float32) if cap else None
        metric = ctypes.c_float()
        loss = np.zeros(max(steps, 1), np.float32)
        n = self.lib.seam_host_train(str(model).encode(), str(data).encode(), steps, None if p_in is None else p_in.ctypes.data,
                                     None if out is None else out.ctypes.data, cap, ctypes.byref(metric), loss.ctypes.data)
        assert cap == 0 or n <= cap
        return (None if out is None else out[:n]), metric.value, loss[:max(steps, 0)], n

    def init_params(self, model, data, cap=1 << 24):
        """The reference's own random initialisation (Edge::Initialize), flat with its 128-float slice alignment."""
        return self._call(model, data, -1, None, cap)[0].copy()

    def gradient(self, model, data, params):
        """One Fprop(train) / ComputeDeriv / Bprop on batch 0: the flat gradient buffer."""
        return self._call(model, data, 0, np.ascontiguousarray(params), params.size)[0].copy()

    def train(self, model, data, steps, params=None):
        """`steps` x ConvNet::TrainOneBatch: (flat parameters, summed correct count, per-step loss).  params=None: start from
        the reference's own initialisation and do not fetch the parameters (returns their count instead)."""
        if params is None:
            _, m, l, n = self._call(model, data, steps, None, 0)
            return n, m, l.copy()
        p, m, l, _ = self._call(model, data, steps, np.ascontiguousarray(params), params.size)
        return p.copy(), m, l.copy()

    def batch(self, model, data, index, dims, batch):
        x, y = np.zeros(dims * batch, np.float32), np.zeros(batch, np.float32)
        n = self.lib.seam_host_batch(str(model).encode(), str(data).encode(), index, x.ctypes.data, x.size, y.ctypes.data, y.size)
        assert n == x.size
        return x, y

    def checkpoint(self, model, data, steps, params, path):
        """`steps` x TrainOneBatch from `params`, then the reference's ConvNet::Save(path); returns the saved flat parameters."""
        p = np.ascontiguousarray(params, np.float32)
        out = np.zeros_like(p)
        n = self.lib.seam_host_checkpoint(str(model).encode(), str(data).encode(), steps, p.ctypes.data, str(path).encode(), out.ctypes.data, out.size)
        assert n == out.size
        return out

    def reduce_lr_decisions(self, model, errors):
        """ConvNet::CheckReduceLearningRate after each validation of the history `errors`."""
        e = np.ascontiguousarray(errors, np.float32)
        out = np.zeros(e.size, np.int32)
        self.lib.seam_host_reduce_lr(str(model).encode(), e.ctypes.data, e.size, out.ctypes.data)
        return out.astype(bool)

    def describe(self, model, data):
        """(layers, edges, flat_size) as the reference builds the net: layers = [(name, size_y, size_x, channels, is_input,
        is_output)] in its topological order, edges = [(source, dest, parameter floats)] in edge order."""
        buf = ctypes.create_string_buffer(1 << 20)
        n = self.lib.seam_host_describe(str(model).encode(), str(data).encode(), buf, len(buf))
        assert 0 < n < len(buf)
        layers, edges, total = [], [], None
        for line in buf.value.decode().splitlines():
            f = line.split()
            if f[0] == "layer":
                layers.append((f[1], int(f[2]), int(f[3]), int(f[4]), bool(int(f[5])), bool(int(f[6]))))
            elif f[0] == "edge":
                edges.append((f[1], f[2], int(f[3])))
            else:
                total = int(f[1])
        return layers, edges, total

    def sgd(self, optimizer_text, params, grads):
        """The reference's Optimizer (ChooseOptimizer on the text config) stepping one (rows, cols) column-major parameter with
        grads[t]: the parameter after every step, shape (steps, cols, rows) like `grads`."""
        steps, cols, rows = grads.shape
        out = np.zeros_like(grads)
        p, g = np.ascontiguousarray(params, np.float32), np.ascontiguousarray(grads, np.float32)
        self.lib.seam_host_sgd(optimizer_text.encode(), rows, cols, steps, p.ctypes.data, g.ctypes.data, out.ctypes.data)
        return out

    def grad_check(self, model, batch, out_h5):
        """apps/run_grad_check.cc: GradChecker::Run writing <edge>_{weights,bias}_{analytical,numerical} to `out_h5`."""
        self.lib.seam_host_grad_check(str(model).encode(), batch, str(out_h5).encode())


    def train_dp(self, model, data, steps, params, rank=0, nranks=1, comm_id=None, bucket_bytes=8 << 20):
        """`steps` x TrainOneBatch of the data-parallel subclass host (seam_host.cc SeamDPNet: gradient slices posted to the
        library's RCCL exchange entries from Bprop, waited for in UpdateWeights) — libref_host_hip.so only."""
        fn = self.lib.seam_host_train_dp
        fn.restype = ctypes.c_long
        fn.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int,
                       ctypes.c_char_p, ctypes.c_long, ctypes.c_void_p]
        p = np.ascontiguousarray(params, np.float32)
        out = np.zeros_like(p)
        buckets = ctypes.c_int(0)
        n = fn(str(model).encode(), str(data).encode(), steps, p.ctypes.data, out.ctypes.data, out.size, rank, nranks, comm_id, bucket_bytes,
               ctypes.byref(buckets))
        assert n == out.size, n
        return out

    def grad_check_fixed(self, model, data, params, out_h5):
        """GradChecker on the data shim's batch 0 at the given parameters (seam_host.cc SeamGradChecker::RunFixed): the reference's
        own compiled pass/fail verdicts, [(weights_passed, bias_passed)] per grad_check edge in edge order; arrays go to `out_h5`."""
        p = np.ascontiguousarray(params, np.float32)
        flags = np.full(256, -1, np.int32)
        n = self.lib.seam_host_grad_check_fixed(str(model).encode(), str(data).encode(), p.ctypes.data, str(out_h5).encode(), flags.ctypes.data,
                                                flags.size)
        assert 0 < n <= flags.size // 2
        return [(bool(flags[2 * i]), bool(flags[2 * i + 1])) for i in range(n)]


def write_configs(tmp, model_text, batch, num_batches, seed, name="net"):
    m = os.path.join(str(tmp), name + ".pbtxt")
    d = os.path.join(str(tmp), name + "_data.pbtxt")
    with open(m, "w") as f:
        f.write(model_text)
    with open(d, "w") as f:   # seam_datahandler.h reads: batch_size, max_dataset_size (cases), chunk_size (= the data seed)
        f.write(f"batch_size: {batch}\nmax_dataset_size: {batch * num_batches}\nchunk_size: {seed}\n")
    return m, d


def read_grad_check(path, edge_names):
    """GradChecker output file -> {edge: {"weights"|"bias": (analytical[k], numerical[n_eps, k])}} (grad_check.cc:105-131)."""
    from convnet_amd import hdf5io
    out = {}
    with hdf5io.File(path, "r") as f:
        for e in edge_names:
            out[e] = {}
            for kind in ("weights", "bias"):
                rows, cols = f.ReadHDF5Shape(f"{e}_{kind}_analytical")
                a = np.asarray(f.ReadHDF5CPU(rows * cols, f"{e}_{kind}_analytical"), np.float32).reshape(-1)
                rows, cols = f.ReadHDF5Shape(f"{e}_{kind}_numerical")
                n = np.asarray(f.ReadHDF5CPU(rows * cols, f"{e}_{kind}_numerical"), np.float32).reshape(-1, a.size)
                out[e][kind] = (a, n)
    return out


def grad_check_passes(a, n):
    """GradChecker::GradCheck's criterion (grad_check.cc:41-64): some epsilon with mean |diff/scale| over non-zero entries < 1 %."""
    best = np.inf
    for row in n:
        diff, scale = a - row, (a + row) / 2
        nz = ~((scale == 0) & (diff == 0))
        if nz.any():
            with np.errstate(divide="ignore", invalid="ignore"):
                best = min(best, float(np.mean(np.abs(diff[nz] / scale[nz]))))
        else:
            best = 0.0
    return best < 0.01, best


def h5_listing(path):
    """(dataset names, root attribute names) of an HDF5 file, straight from libhdf5 (no h5py in this image)."""
    from convnet_amd import hdf5io
    L = hdf5io._lib()
    hid = ctypes.c_int64
    for name, res, args in (("H5Gget_num_objs", ctypes.c_int, [hid, ctypes.POINTER(ctypes.c_uint64)]),
                            ("H5Lget_name_by_idx", ctypes.c_ssize_t, [hid, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_uint64, ctypes.c_char_p,
                                                                      ctypes.c_size_t, hid]),
                            ("H5Aget_num_attrs", ctypes.c_int, [hid]),
                            ("H5Aget_name_by_idx", ctypes.c_ssize_t, [hid, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_uint64, ctypes.c_char_p,
                                                                      ctypes.c_size_t, hid])):
        f = getattr(L, name)
        f.restype, f.argtypes = res, args
    with hdf5io.File(str(path), "r") as f:
        n = ctypes.c_uint64(0)
        assert L.H5Gget_num_objs(f.id, ctypes.byref(n)) >= 0
        buf = ctypes.create_string_buffer(512)
        names, attrs = [], []
        for i in range(n.value):
            assert L.H5Lget_name_by_idx(f.id, b".", 0, 0, i, buf, len(buf), 0) >= 0
            names.append(buf.value.decode())
        L.H5Gopen2.restype, L.H5Gopen2.argtypes = hid, [hid, ctypes.c_char_p, hid]
        L.H5Gclose.restype, L.H5Gclose.argtypes = ctypes.c_int, [hid]
        root = L.H5Gopen2(f.id, b"/", 0)       # attributes written "on the file" live on its root group
        assert root >= 0
        for i in range(L.H5Aget_num_attrs(root)):
            assert L.H5Aget_name_by_idx(root, b".", 0, 0, i, buf, len(buf), 0) >= 0
            attrs.append(buf.value.decode())
        L.H5Gclose(root)
    return sorted(names), sorted(attrs)


def golden_params(total, seed, slices):
    """Deterministic integer-hash parameters for the full-size golden runs (no RNG library involved, identical on every
    machine): unit-variance uniform hash values scaled per edge slice to sqrt(2 / fan_in) so the 8-layer net neither
    saturates nor dies.  ``slices`` = [(offset, floats, dest_channels)] of the edges with parameters; fan_in = floats /
    dest_channels - 1 (weights F x K plus F biases share the slice, src/convnet.cc:271-296)."""
    p = np.zeros(total, np.float32)
    u = hash_batch(seed + 17, 3, total, True)
    for off, n, F in slices:
        fan_in = n // F - 1
        p[off:off + n] = u[off:off + n] * np.float32(np.sqrt(2.0 / fan_in))
    return p


def slices_from_describe(layers, edges):
    """[(offset, floats, dest_channels)] with the reference's 128-float slice alignment, from RefHost.describe()."""
    chans = {name: c for name, _, _, c, _, _ in layers}
    out, off = [], 0
    for _, dst, n in edges:
        if n == 0:
            continue
        out.append((off, n, chans[dst]))
        off += (n + 127) // 128 * 128
    return out, off


def grad_check_criterion(a, n):
    """GradChecker::GradCheck's running criterion, epsilon by epsilon, in the reference's float arithmetic INCLUDING its carry-over
    of diff_sum and of the non-zero count between epsilons (grad_check.cc:41-64): [value after epsilon 0, after epsilon 1, ...] up to
    and including the first value < 0.01 (where the reference stops).  NaN (0/0: every entry exactly zero) never passes."""
    f32 = np.float32
    diff_sum, non_zero, out = f32(0), 0, []
    for row in n:
        for k in range(a.size):
            diff = f32(a[k]) - f32(row[k])
            scale = (f32(a[k]) + f32(row[k])) / f32(2)
            if not (scale == 0 and diff == 0):
                with np.errstate(divide="ignore", invalid="ignore"):
                    diff_sum = f32(diff_sum + abs(diff / scale))
                non_zero += 1
        with np.errstate(divide="ignore", invalid="ignore"):
            diff_sum = f32(diff_sum / f32(non_zero)) if non_zero else f32("nan")
        out.append(float(diff_sum))
        if diff_sum < 0.01:
            break
    return out
