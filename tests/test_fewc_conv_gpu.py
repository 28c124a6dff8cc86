"""gfc_kernel (convnet_amd/csrc/fewc_conv.hip): conv fprop of the few-channel, wide-filter, stride-2 layer (AlexNet conv1: 3 x 7 x 7 / 2)
as a patch-resident gather-GEMM with the filter bank resident in LDS, persistent blocks and a producer wave — against the CPU oracle
(the reference's conv_up, cudamat_conv_gemm.cu:545-640) on geometries chosen for ITS mechanisms: several tiles per block (the patch
refill and both barriers), two column groups with a ragged last one, fewer than 96 filters, no padding / wide padding (border rows and
columns from the zero page), rectangular images, the fused bias + ReLU epilogue.  Every case asserts
that gfc_kernel is what ran.  The full conv1 geometry at 256 images runs in tests/test_full_geometry_gpu.py on the default path.
Tolerance: the reference's own kernel-test metric, max|a-b| / mean|a+b| < 1e-4 (py/test_conv.py:382-392)."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle  # noqa: E402
from oracle import Geom  # noqa: E402
from golden_cases import rel_err  # noqa: E402

TOL = 1e-4


@pytest.fixture(scope="module")
def hip():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from convnet_amd.matrix import Matrix
    from hip_adapter import HipImpl
    Matrix.SetupCUDADevice(0)
    Matrix.InitRandom(42)
    from convnet_amd import _lib
    _lib.lib.convnet_hip_set_matrix_path(1)
    return HipImpl()


def last_kernel():
    from convnet_amd import _lib
    info = _lib.KernelInfo()
    _lib.lib.convnet_hip_last_kernel_info(ctypes.byref(info))
    return info.name.decode()


def rnd(rng, shape):
    return rng.standard_normal(shape).astype(np.float32)


CASES = [
    Geom(N=32, C=3, H=31, W=31, F=96, Ky=7, Kx=7, sy=2, sx=2, pady=1, padx=1),     # 14 x 14 outputs: two column groups (8 + 6), 28 tiles
    Geom(N=64, C=3, H=25, W=25, F=64, Ky=7, Kx=7, sy=2, sx=2),                     # no padding, 64 of 96 rows, two image blocks
    Geom(N=32, C=3, H=21, W=45, F=80, Ky=7, Kx=7, sy=2, sx=2, pady=3, padx=3),     # rectangular, three columns / rows of padding, 23 outputs per row
    Geom(N=256, C=3, H=63, W=63, F=96, Ky=7, Kx=7, sy=2, sx=2, pady=1, padx=1),    # 30 x 30 x 8 image blocks = 960 tiles: every CU a run of several tiles
    Geom(N=96, C=3, H=224, W=224, F=96, Ky=7, Kx=7, sy=2, sx=2, pady=1, padx=1),   # conv1's own image at 96 images: 4 620 tiles
]
_id = lambda g: f"N{g.N}H{g.H}W{g.W}F{g.F}p{g.pady}"  # noqa: E731


@pytest.mark.parametrize("g", CASES, ids=_id)
def test_fewc_fprop_vs_oracle(hip, g):
    rng = np.random.default_rng(51)
    x, w = rnd(rng, g.in_shape()), rnd(rng, g.filt_shape())
    t0 = rnd(rng, g.out_shape())
    got = hip.conv_up(g, x, w, t0.copy(), 0.0)
    assert last_kernel() == "gfc_kernel(fprop)", last_kernel()
    ref = oracle.port.conv_up(g, x, w, t0.copy(), 0.0)
    assert rel_err(got, ref) < TOL
    if g.N * g.H * g.W <= 2 * 10 ** 6:
        # accumulating into the target (a layer with several incoming edges) is left to the gather kernels
        got = hip.conv_up(g, x, w, t0.copy(), 1.0)
        assert last_kernel() == "gg_kernel(fprop)", last_kernel()
        assert rel_err(got, ref + t0) < TOL


def test_fewc_fused_bias_relu(hip):
    g = CASES[0]
    rng = np.random.default_rng(52)
    x, w, b = rnd(rng, g.in_shape()), rnd(rng, g.filt_shape()), rnd(rng, (g.F,))
    fused = hip.conv_up_bias_relu(g, x, w, b, relu=True)
    assert last_kernel() == "gfc_kernel(fprop)"
    y = oracle.port.conv_up(g, x, w)
    y = oracle.port.add_row_vec(y.reshape(g.F, -1), b).reshape(g.out_shape())
    assert rel_err(fused, oracle.port.lower_bound(y, 0.0)) < TOL
    # ... and bit for bit what the unfused sequence of library calls gives (the epilogue adds and clamps in the same order)
    from convnet_amd.matrix import Matrix
    from hip_adapter import _mat, _desc
    xm = _mat(x, g.N, g.W * g.H * g.C, (g.N, g.W, g.H, g.C))
    wm = _mat(w, g.F, g.K, (g.F, g.Kx, g.Ky, g.C))
    bm = _mat(b, 1, g.F)
    tm = _mat(np.zeros(g.out_shape(), np.float32), g.N, g.Mx * g.My * g.F, (g.N, g.Mx, g.My, g.F))
    Matrix.ConvUp(xm, wm, tm, _desc(g), 0.0)
    tm.Reshape(-1, g.F)
    tm.AddRowVec(bm)
    tm.Reshape(g.N, -1)
    tm.LowerBound(0.0)
    assert np.array_equal(fused, tm.ToNumpy().reshape(g.out_shape()))


def test_other_shapes_stay_on_the_gather_kernels(hip):
    """N % 32 != 0, more than 96 filters, another filter size: the generic-k path of ggp_kernel / gg_kernel, as before"""
    rng = np.random.default_rng(53)
    for g in (Geom(N=48, C=3, H=31, W=31, F=96, Ky=7, Kx=7, sy=2, sx=2, pady=1, padx=1),
              Geom(N=32, C=3, H=31, W=31, F=128, Ky=7, Kx=7, sy=2, sx=2, pady=1, padx=1),
              Geom(N=32, C=3, H=20, W=20, F=64, Ky=3, Kx=3, pady=1, padx=1)):
        x, w = rnd(rng, g.in_shape()), rnd(rng, g.filt_shape())
        got = hip.conv_up(g, x, w)
        assert last_kernel() == "gg_kernel(fprop)", (g, last_kernel())
        assert rel_err(got, oracle.port.conv_up(g, x, w)) < TOL


# ---- the weight gradient of the same layer class: wg_kernel's 160 x 96 tile of 16 x 16 MFMAs (gather_gemm.hip) -----------------------------
# Against the CPU oracle (the reference's conv_outp, cudamat_conv_gemm.cu:827-960) with the bias row riding along, on geometries chosen
# for the walk of the reduction: several image chunks per pixel, a ragged last chunk, rectangular maps, wide padding, other tap shapes,
# slab ranges that begin in the middle of an output row.  (Round 6 tried a patch-resident A operand for this tile — a ring of input
# columns, 42 staged rows per pixel instead of 147; parity green on these cases at the first run, 16 % SLOWER: the tile is bound by the
# split arithmetic of its wave, not by staging — profiles/r06_conv1_wgrad_ring.txt — and the build was removed.)
WGRAD_CASES = [
    Geom(N=32, C=3, H=31, W=31, F=96, Ky=7, Kx=7, sy=2, sx=2, pady=1, padx=1),     # one image chunk, 14 x 14 outputs
    Geom(N=96, C=3, H=21, W=45, F=80, Ky=7, Kx=7, sy=2, sx=2, pady=3, padx=3),     # three image chunks per pixel, rectangular, wide padding, 80 of 96 filters
    Geom(N=44, C=3, H=25, W=25, F=96, Ky=7, Kx=7, sy=2, sx=2),                     # N % 32 != 0: the second chunk is 12 images; no padding
    Geom(N=64, C=5, H=15, W=19, F=96, Ky=4, Kx=7, sy=2, sx=2, pady=1, padx=1),     # 5 channels x 4 x 7 taps: K = 140
    Geom(N=32, C=3, H=12, W=12, F=96, Ky=7, Kx=7, pady=3, padx=3),                 # stride 1
    Geom(N=256, C=3, H=63, W=63, F=96, Ky=7, Kx=7, sy=2, sx=2, pady=1, padx=1),    # 30 x 30 outputs x 8 chunks = 7 200 chunks over ~500 blocks
]


@pytest.mark.parametrize("g", WGRAD_CASES, ids=_id)
def test_fewc_wgrad_vs_oracle(hip, g):
    from convnet_amd import _lib
    from hip_adapter import conv_outp_bias
    rng = np.random.default_rng(54)
    x, dy = rnd(rng, g.in_shape()), rnd(rng, g.out_shape())
    dw0, db0 = rnd(rng, g.filt_shape()), rnd(rng, (g.F,))
    for st, so in ((0.0, 1.0), (1.0, 0.5)):
        _lib.profile_enable(True)
        dw, db = conv_outp_bias(g, x, dy, dw0.copy(), db0.copy(), st, so)
        names = [r["kernel"] for r in _lib.profile_report()]
        _lib.profile_enable(False)
        assert any(n.startswith("wg_kernel<2,2,5,3,x16") for n in names), names
        assert rel_err(dw, oracle.port.conv_outp(g, x, dy, dw0.copy(), st, so)) < TOL
        ref_db = st * db0 + so * dy.reshape(g.F, -1).astype(np.float64).sum(axis=1)
        assert rel_err(db, ref_db.astype(np.float32)) < TOL
