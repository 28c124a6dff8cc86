"""GPU parity tests proper: the HIP hot path, called through the C ABI (class Matrix -> ctypes ->
libconvnet_hip.so), against (1) the committed golden outputs of the reference's own CPU code and
(2) the CPU oracle on seeded inputs, using the reference's acceptance metric and tolerance:
max|a-b| / mean|a+b| < 1e-4 (py/test_conv.py:382-392).  Selection ops must be bit-exact.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle  # noqa: E402
from oracle import Geom  # noqa: E402
from golden_cases import compute_all, rel_err  # noqa: E402

TOL = 1e-4   # the reference's own kernel-test tolerance (py/test_conv.py:387-392)
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hotpath_golden.npz")


@pytest.fixture(scope="module")
def hip():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from convnet_amd.matrix import Matrix
    from hip_adapter import HipImpl
    Matrix.SetupCUDADevice(0)
    Matrix.InitRandom(42)
    return HipImpl()


@pytest.fixture(params=["split", "fp32"])
def matrix_path(request, hip):
    """Both ways the GEMM kernels form their fp32 products (include/convnet_hip.h: convnet_hip_set_matrix_path): on the bf16
    matrix pipe from exact three-way operand splits (the default) and with the fp32 matrix instruction.  Same tolerances."""
    from convnet_amd import _lib
    L = _lib.lib
    L.convnet_hip_set_matrix_path(1 if request.param == "split" else 0)
    yield request.param
    L.convnet_hip_set_matrix_path(1)


def rnd(rng, shape):
    return rng.standard_normal(shape).astype(np.float32)


def test_golden_vectors(hip, matrix_path):
    golden = dict(np.load(GOLDEN))
    got = compute_all(hip)
    assert set(got) == set(golden)
    for k in sorted(golden):
        if k.endswith("/max") or k in ("softmax/grad", "softmax/correct"):
            if k == "softmax/grad":
                assert rel_err(got[k], golden[k]) < 1e-6, k
            else:
                assert np.array_equal(got[k], golden[k]), k
        else:
            assert rel_err(got[k], golden[k]) < TOL, (k, rel_err(got[k], golden[k]))


def test_golden_sgd_fused_is_bit_exact(hip):
    """The one-pass SGD kernel keeps every reference statement a separately rounded fp32 op."""
    from hip_adapter import HipImpl
    from golden_cases import inputs
    golden = dict(np.load(GOLDEN))
    g0, w0, h0 = inputs(403, (30, 12), (30, 12), (30, 12))
    HipImpl(fused=True).sgd_step(g0, w0, h0, 5e-4, 0.9, 0.01, 0.7, 0.8, 0.0)
    assert np.array_equal(g0, golden["sgd/grad"]) and np.array_equal(h0, golden["sgd/hist"])
    assert rel_err(w0, golden["sgd/param"]) < 1e-6  # row-norm limit uses a different summation order


CONV_CASES = [
    # py/test_conv.py Test2D (:394-418)
    Geom(N=128, C=32, H=12, W=12, F=64, Ky=3, Kx=3, sy=2, sx=2, pady=1, padx=1),
    # ragged everything: N%4!=0, F%4!=0, rectangular image/kernel/stride
    Geom(N=5, C=3, H=17, W=15, F=7, Ky=5, Kx=3, sy=2, sx=1, pady=2, padx=1),
    Geom(N=100, C=1, H=28, W=28, F=48, Ky=4, Kx=4),                                    # mnist-conv conv1, bs=100
    Geom(N=100, C=48, H=11, W=11, F=128, Ky=4, Kx=4),                                  # mnist-conv conv2
    Geom(N=8, C=3, H=64, W=64, F=96, Ky=7, Kx=7, sy=2, sx=2, pady=1, padx=1),          # AlexNet conv1 (cropped image)
    Geom(N=8, C=96, H=27, W=27, F=256, Ky=5, Kx=5, sy=2, sx=2),                         # AlexNet conv2 (cropped)
    Geom(N=16, C=256, H=13, W=13, F=384, Ky=3, Kx=3, pady=1, padx=1),                  # AlexNet conv3 (exact)
    Geom(N=132, C=16, H=7, W=7, F=40, Ky=3, Kx=3, pady=1, padx=1),                     # N just over one 128 block
    Geom(N=4, C=8, H=9, W=9, F=200, Ky=3, Kx=3, sy=3, sx=3),                            # stride == kernel
    Geom(N=4, C=8, H=10, W=10, F=33, Ky=2, Kx=2, sy=3, sx=3),                           # stride > kernel (uncovered pixels)
    Geom(N=6, C=3, H=21, W=19, F=96, Ky=7, Kx=7, sy=2, sx=2, pady=1, padx=1),          # conv1 shape, N%4!=0: scalar path of the 16x16 wgrad tile
]


@pytest.mark.parametrize("g", CONV_CASES, ids=lambda g: f"N{g.N}C{g.C}H{g.H}F{g.F}k{g.Ky}s{g.sy}p{g.pady}")
def test_conv_up_down_outp_vs_oracle(hip, matrix_path, g):
    rng = np.random.default_rng(11)
    x, w, dy = rnd(rng, g.in_shape()), rnd(rng, g.filt_shape()), rnd(rng, g.out_shape())
    for st in (0.0, 1.0):
        t0 = rnd(rng, g.out_shape())
        assert rel_err(hip.conv_up(g, x, w, t0.copy(), st), oracle.port.conv_up(g, x, w, t0.copy(), st)) < TOL
        t0 = rnd(rng, g.in_shape())
        assert rel_err(hip.conv_down(g, dy, w, t0.copy(), st), oracle.port.conv_down(g, dy, w, t0.copy(), st)) < TOL
        t0 = rnd(rng, g.filt_shape())
        so = 0.37 / g.N
        assert rel_err(hip.conv_outp(g, x, dy, t0.copy(), st, so), oracle.port.conv_outp(g, x, dy, t0.copy(), st, so)) < TOL


def test_conv_fused_bias_relu_equals_unfused_sequence(hip, matrix_path):
    g = Geom(N=16, C=8, H=9, W=9, F=24, Ky=3, Kx=3, pady=1, padx=1)
    rng = np.random.default_rng(12)
    x, w, b = rnd(rng, g.in_shape()), rnd(rng, g.filt_shape()), rnd(rng, (g.F,))
    fused = hip.conv_up_bias_relu(g, x, w, b, relu=True)
    y = oracle.port.conv_up(g, x, w)
    y = oracle.port.add_row_vec(y.reshape(g.F, -1), b).reshape(g.out_shape())  # (N*My*Mx, F) + row vec: conv_edge.cc:146-148
    y = oracle.port.lower_bound(y, 0.0)
    assert rel_err(fused, y) < TOL


POOL_CASES = [
    Geom(N=128, C=32, H=12, W=12, F=32, Ky=3, Kx=3, sy=2, sx=2, pady=1, padx=1),
    Geom(N=7, C=5, H=11, W=11, F=5, Ky=3, Kx=3, sy=2, sx=2, pady=1, padx=1),
    Geom(N=100, C=48, H=25, W=25, F=48, Ky=4, Kx=4, sy=2, sx=2),                     # mnist-conv pool1
    Geom(N=16, C=96, H=110, W=110, F=96, Ky=3, Kx=3, sy=2, sx=2, pady=1, padx=1),    # AlexNet pool1 (exact spatial)
    Geom(N=6, C=3, H=9, W=7, F=3, Ky=3, Kx=2, sy=2, sx=1, pady=1, padx=0),           # rectangular
    # few channels x few images on a map of >= 400 pixels: the row grid takes the XCD order, the 2 x 2-block undo grid does not
    Geom(N=16, C=3, H=24, W=24, F=3, Ky=3, Kx=3, sy=2, sx=2, pady=1, padx=1),
    Geom(N=128, C=3, H=20, W=20, F=3, Ky=3, Kx=3, sy=2, sx=2, pady=1, padx=1),
]


@pytest.mark.parametrize("g", POOL_CASES, ids=lambda g: f"N{g.N}C{g.C}H{g.H}k{g.Ky}s{g.sy}p{g.pady}")
def test_pooling_vs_oracle(hip, g):
    rng = np.random.default_rng(13)
    x = np.maximum(rnd(rng, g.in_shape()), 0)   # post-ReLU input: exact ties at 0 everywhere
    dy = rnd(rng, g.pooled_shape())
    mp = hip.max_pool(g, x)
    assert np.array_equal(mp, oracle.port.max_pool(g, x))
    assert rel_err(hip.avg_pool(g, x), oracle.port.avg_pool(g, x)) < 1e-6
    for st in (0.0, 1.0):
        t0 = rnd(rng, g.in_shape())
        assert rel_err(hip.max_pool_undo(g, x, dy, mp, t0.copy(), st), oracle.port.max_pool_undo(g, x, dy, mp, t0.copy(), st)) < 1e-6
        assert rel_err(hip.avg_pool_undo(g, dy, t0.copy(), st), oracle.port.avg_pool_undo(g, dy, t0.copy(), st)) < 1e-6


def test_max_pool_undo_routes_gradient_to_every_tie(hip):
    """an argmax-index implementation is NOT parity-equivalent."""
    g = Geom(N=4, C=1, H=4, W=4, F=1, Ky=2, Kx=2, sy=2, sx=2)
    x = np.zeros(g.in_shape(), np.float32)          # every window is a 4-way tie
    dy = np.ones(g.pooled_shape(), np.float32)
    mp = hip.max_pool(g, x)
    out = hip.max_pool_undo(g, x, dy, mp)
    assert np.array_equal(out, np.ones(g.in_shape(), np.float32))


@pytest.mark.parametrize("shape,size_f,blocked", [
    ((32, 6, 6, 128), 8, False),        # Test2D rnorm: sizeF=8, add .005, pow .75
    ((96, 5, 5, 16), 24, False),        # AlexNet rnorm1 channel geometry
    ((256, 3, 3, 8), 64, False),        # AlexNet rnorm2
    ((20, 2, 3, 5), 5, True),
    ((7, 2, 2, 3), 3, False),           # ragged location count
    # the fast kernels of round 6 (pool_norm.hip: rnorm_fwd_fast_kernel / rnorm_undo_fast_kernel; windows 24 and 64, 16-byte rows)
    ((96, 9, 7, 12), 24, False),        # 12 channels per lane, 64-location tiles: 756 locations = 11 whole tiles and a ragged one
    ((96, 27, 27, 32), 24, False),      # 365 tiles over the persistent blocks of eight XCD ranges (several tiles per block: the register prefetch)
    ((100, 5, 5, 16), 24, False),       # > 96 channels: 6 per lane, 17 lane groups, two spare channels on the zero rows
    ((250, 3, 5, 8), 64, False),        # 8 per lane, 32 groups, six spare channels
    ((64, 4, 4, 8), 64, False),         # the window spans every channel: all of it clipped at one end or the other
    ((96, 5, 5, 15), 24, False),        # 375 locations: rows not 16-byte multiples -> the LDS-tiled kernels
    ((96, 5, 5, 16), 24, True),         # blocked windows -> the LDS-tiled kernels
])
def test_response_norm_vs_oracle(hip, shape, size_f, blocked):
    rng = np.random.default_rng(14)
    x, dy = rnd(rng, shape), rnd(rng, shape)
    assert rel_err(hip.rnorm(x, size_f, 0.005, 0.75, blocked), oracle.port.rnorm(x, size_f, 0.005, 0.75, blocked)) < TOL
    assert rel_err(hip.rnorm_undo(dy, x, size_f, 0.005, 0.75, blocked), oracle.port.rnorm_undo(dy, x, size_f, 0.005, 0.75, blocked)) < TOL


@pytest.mark.parametrize("N,D,F", [(256, 1152, 1000), (100, 1152, 10), (9, 37, 11), (128, 4096, 512), (32, 260, 132)])
def test_fc_dot_three_forms_vs_oracle(hip, matrix_path, N, D, F):
    rng = np.random.default_rng(15)
    x, w, dy = rnd(rng, (D, N)), rnd(rng, (D, F)) * 0.1, rnd(rng, (F, N))
    for beta in (0.0, 1.0):
        t0 = rnd(rng, (F, N))   # fc_edge.cc:54  out = in * W^T
        assert rel_err(hip.dot(x, w, t0.copy(), beta, 1.0, False, True), oracle.port.dot(x, w, t0.copy(), beta, 1.0, False, True)) < TOL
        t0 = rnd(rng, (D, N))   # fc_edge.cc:66  d_in = d_out * W
        assert rel_err(hip.dot(dy, w, t0.copy(), beta, 1.0), oracle.port.dot(dy, w, t0.copy(), beta, 1.0)) < TOL
        t0 = rnd(rng, (D, F))   # fc_edge.cc:74  dW = d_out^T * in / N
        assert rel_err(hip.dot(dy, x, t0.copy(), beta, 1.0 / N, True, False), oracle.port.dot(dy, x, t0.copy(), beta, 1.0 / N, True, False)) < TOL


@pytest.mark.parametrize("m,n,k", [(37, 50, 61), (128, 96, 256), (5, 3, 2)])
def test_dot_full_contract_all_transposes_and_alpha(hip, m, n, k):
    """dot's whole contract (cudamat.cu:2130-2152: every transpose combination, any alpha and beta), not only fc_edge.cc's three
    uses: T,T and alpha != 1 on the N,x side run on the library's general kernel."""
    rng = np.random.default_rng(23)
    for ta in (False, True):
        for tb in (False, True):
            a = rnd(rng, (m, k) if ta else (k, m))      # numpy (cols, rows) view of a column-major (rows, cols) matrix
            b = rnd(rng, (k, n) if tb else (n, k))
            for beta, alpha in ((0.0, 1.0), (0.5, -1.75), (1.0, 0.3)):
                t0 = rnd(rng, (n, m))
                got = hip.dot(a, b, t0.copy(), beta, alpha, ta, tb)
                want = oracle.port.dot(a, b, t0.copy(), beta, alpha, ta, tb)
                assert rel_err(got, want) < TOL, (ta, tb, beta, alpha, rel_err(got, want))


def test_linearity_and_adjointness_at_full_alexnet_conv3_size(hip, matrix_path):
    """Size-independent properties at BASELINE's full layer size (N=256), where the CPU oracle would
    take minutes: <conv(x,w), dy> == <x, convT(dy,w)> == <w, wgrad(x,dy)> (the three kernels are
    mutually adjoint), checked in float64 on the host."""
    g = Geom(N=256, C=256, H=13, W=13, F=384, Ky=3, Kx=3, pady=1, padx=1)
    rng = np.random.default_rng(16)
    x, w, dy = rnd(rng, g.in_shape()), rnd(rng, g.filt_shape()) * 0.05, rnd(rng, g.out_shape())
    y = hip.conv_up(g, x, w)
    dx = hip.conv_down(g, dy, w)
    dw = hip.conv_outp(g, x, dy)
    a = float(np.vdot(y.astype(np.float64), dy.astype(np.float64)))
    b = float(np.vdot(x.astype(np.float64), dx.astype(np.float64)))
    c = float(np.vdot(w.astype(np.float64), dw.astype(np.float64)))
    assert abs(a - b) / abs(a) < 1e-5 and abs(a - c) / abs(a) < 1e-5, (a, b, c)
    # spot-check 64 output elements exactly against a float64 dot product of the gathered patch
    for _ in range(64):
        n, f, oy, ox = rng.integers(g.N), rng.integers(g.F), rng.integers(g.My), rng.integers(g.Mx)
        acc = 0.0
        for ky in range(g.Ky):
            for kx in range(g.Kx):
                iy, ix = oy + ky - 1, ox + kx - 1
                if 0 <= iy < g.H and 0 <= ix < g.W:
                    acc += float(np.dot(x[:, iy, ix, n].astype(np.float64), w[:, ky, kx, f].astype(np.float64)))
        assert abs(acc - y[f, oy, ox, n]) < 1e-4 * max(1.0, abs(acc))


def test_softmax_family_and_fused(hip):
    from convnet_amd.matrix import Matrix
    from hip_adapter import _mat
    rng = np.random.default_rng(17)
    N, K = 256, 1000
    z = (3 * rnd(rng, (K, N)))
    labels = rng.integers(0, K, N).astype(np.float32)
    p_ref = oracle.port.softmax_row_major(z.copy())
    # probabilities span 1e-9..0.5: compare element-wise relative (expf ulp + summation order).  Every assertion names what it
    # saw: this test once failed as the first GPU test of a fresh box (NOTES.md) and the message was lost.
    p_hip = hip.softmax_row_major(z.copy())
    assert np.allclose(p_hip, p_ref, rtol=1e-5, atol=1e-12), ("softmax", float(np.abs(p_hip / p_ref - 1).max()), int(np.isnan(p_hip).sum()))
    c_hip, c_ref = hip.softmax_correct_row_major(p_ref, labels), oracle.port.softmax_correct_row_major(p_ref, labels)
    assert np.array_equal(c_hip, c_ref), ("correct", np.flatnonzero(c_hip != c_ref)[:8])
    e = rel_err(hip.softmax_ce_row_major(p_ref, labels), oracle.port.softmax_ce_row_major(p_ref, labels))
    assert e < 1e-5, ("cross entropy", e)
    g_hip, g_ref = hip.softmax_grad_row_major(p_ref, labels), oracle.port.softmax_grad_row_major(p_ref, labels)
    assert np.array_equal(g_hip, g_ref), ("CE derivative", float(np.abs(g_hip - g_ref).max()))
    # fused: softmax + CE-deriv + correct count
    Z, L = _mat(z, N, K), _mat(labels, N, 1)
    P, D, C = _mat(np.zeros_like(z), N, K), _mat(np.zeros_like(z), N, K), _mat(np.zeros(1, np.float32), 1, 1)
    Matrix.SoftmaxCEGradCorrect(Z, L, P, D, C)
    assert np.allclose(P.ToNumpy().reshape(z.shape), p_ref, rtol=1e-5, atol=1e-12)
    assert np.allclose(D.ToNumpy().reshape(z.shape), oracle.port.softmax_grad_row_major(p_ref, labels), rtol=1e-5, atol=1e-7)
    assert C.ToNumpy().reshape(-1)[0] == oracle.port.softmax_correct_row_major(p_ref, labels).sum()


def test_elementwise_reductions_and_views(hip):
    from convnet_amd.matrix import Matrix
    from hip_adapter import _mat
    rng = np.random.default_rng(18)
    a = rnd(rng, (37, 50))   # (rows=50, cols=37)
    b = rnd(rng, (37,))
    assert rel_err(hip.add_row_vec(a.copy(), b), oracle.port.add_row_vec(a.copy(), b)) < 1e-7
    for axis, n in ((0, 37), (1, 50)):
        t0 = rnd(rng, (n,))
        assert rel_err(hip.sum_by_axis(a, t0.copy(), axis, 0.5, 1.0), oracle.port.sum_by_axis(a, t0.copy(), axis, 0.5, 1.0)) < 1e-5
    big = rnd(rng, (3, 5000))   # long columns: block-per-column path
    t0 = np.zeros(3, np.float32)
    assert rel_err(hip.sum_by_axis(big, t0.copy(), 0, 1.0, 0.0), oracle.port.sum_by_axis(big, t0.copy(), 0, 1.0, 0.0)) < 1e-5
    # many short columns (the reference's two-step shared-bias gradient, src/conv_edge.cc:213-218): float4 wave path, with a
    # column count that is not a multiple of the 32 columns a block takes, and one shape that must stay on the scalar path
    for rows, cols in ((256, 4099), (132, 5000), (130, 4100)):
        many = rnd(rng, (cols, rows))
        t0 = rnd(rng, (cols,))
        assert rel_err(hip.sum_by_axis(many, t0.copy(), 0, 0.25, 0.5), oracle.port.sum_by_axis(many, t0.copy(), 0, 0.25, 0.5)) < 1e-5
    assert np.array_equal(hip.lower_bound(a.copy(), 0.0), oracle.port.lower_bound(a.copy(), 0.0))
    assert np.array_equal(hip.upper_bound_mod(a.copy(), 0.4), oracle.port.upper_bound_mod(a.copy(), 0.4))
    st = np.maximum(rnd(rng, a.shape), 0)
    assert np.array_equal(hip.relu_deriv(a.copy(), st), oracle.port.relu_deriv(a.copy(), st))
    for lim, con in ((0.8, False), (1.5, True)):
        assert rel_err(hip.normlimit_rows(a.copy(), lim, con), oracle.port.normlimit_rows(a.copy(), lim, con)) < 1e-6
    # views: slice + reshape share memory (cudamat.cu:587-626)
    M = _mat(a, 50, 37)
    S = Matrix()
    M.GetSlice(S, 5, 9)
    S.Set(7.0)
    back = M.ToNumpy()
    assert np.all(back[5:9] == 7.0) and np.array_equal(back[:5], a[:5]) and np.array_equal(back[9:], a[9:])
    assert abs(M.Sum() - float(back.astype(np.float64).sum())) < 1e-2
    assert M.ReadValue(3, 2) == back[2, 3]
    M.WriteValue(3, 2, -1.5)
    assert M.ReadValue(3 + 50 * 2) == -1.5


def test_dropout_statistics_and_relu_dropout(hip):
    from convnet_amd.matrix import Matrix
    from hip_adapter import _mat
    n = 1 << 20
    x = np.random.default_rng(19).standard_normal(n).astype(np.float32)
    M = _mat(x, n, 1)
    M.Dropout(0.4, 0.0, 1.0 / 0.6)
    y = M.ToNumpy().reshape(-1)
    dropped = (y == 0)
    assert abs(dropped.mean() - 0.4) < 5e-3
    assert np.allclose(y[~dropped], x[~dropped] * np.float32(1.0 / 0.6), rtol=1e-6)
    M2 = _mat(x, n, 1)
    M2.ReluDropout(0.4, 1.0 / 0.6)
    y2 = M2.ToNumpy().reshape(-1)
    kept = y2 != 0
    assert np.all(x[kept] > 0) and np.allclose(y2[kept], x[kept] * np.float32(1.0 / 0.6), rtol=1e-6)
    pos = x > 0
    assert abs((y2[pos] == 0).mean() - 0.4) < 5e-3
    R = Matrix()
    R.AllocateGPUMemory(n, 1)
    R.FillWithRandn()
    r = R.ToNumpy().reshape(-1)
    assert abs(r.mean()) < 5e-3 and abs(r.std() - 1) < 5e-3
    R.FillWithRand()
    r = R.ToNumpy().reshape(-1)
    assert r.min() >= 0 and r.max() < 1 and abs(r.mean() - 0.5) < 2e-3


@pytest.mark.gpu
def test_event_trio_orders_two_streams(hip):
    """cuda_create_event / cuda_record_event / cuda_synchronize_event (cudamat.cu:70-91): a consumer on a second
    stream that waits on the producer's event sees the produced values (Matrix::SetReady / WaitTillReady)."""
    import torch
    from convnet_amd.matrix import Matrix
    a, b = Matrix(256, 4096), Matrix(256, 4096)
    side = torch.cuda.Stream()
    for k in range(5):
        a.Set(float(k))
        for _ in range(20):
            a.Add(1.0)              # a long-ish chain on the main stream
        a.SetReady()
        with Matrix.OnStream(side):
            a.WaitTillReady()
            b.Set(a)                # copy on the side stream, ordered after the chain by the event only
            b.SetReady()
        b.WaitTillReady()           # main stream waits for the copy before reading / overwriting
        assert float(b.Sum()) == pytest.approx((k + 20) * 256 * 4096, rel=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("g", [CONV_CASES[0], CONV_CASES[1], CONV_CASES[4], CONV_CASES[5], CONV_CASES[6], CONV_CASES[10]],
                         ids=lambda g: f"N{g.N}C{g.C}F{g.F}K{g.K}")
def test_conv_outp_bias_equals_outp_plus_two_step_sum(hip, matrix_path, g):
    """convOutpBias = convOutp + the shared-bias gradient (conv_edge.cc:210-221), whether the bias row rides in the
    weight-gradient tile (K+1 fits the padded tile: conv1's 147+1 <= 160) or the library falls back to a column sum
    (K a multiple of the tile: conv3's 2304)."""
    from hip_adapter import conv_outp_bias
    rng = np.random.default_rng(21)
    x, dy = rnd(rng, g.in_shape()), rnd(rng, g.out_shape())
    so = 0.41 / g.N
    for st in (0.0, 1.0):
        dw0, db0 = rnd(rng, g.filt_shape()), rnd(rng, (g.F,))
        dw, db = conv_outp_bias(g, x, dy, dw0.copy(), db0.copy(), st, so)
        assert rel_err(dw, oracle.port.conv_outp(g, x, dy, dw0.copy(), st, so)) < TOL
        ref_db = st * db0 + so * dy.reshape(g.F, -1).astype(np.float64).sum(axis=1)
        assert np.allclose(db, ref_db, rtol=2e-5, atol=1e-5 * np.abs(ref_db).max()), np.abs(db - ref_db).max()


@pytest.mark.gpu
def test_input_staging_is_bit_exact_vs_oracle_incl_ragged_sizes(hip):
    """extract_patches / shuffleColumns / copy_transpose are byte moves and the col/row-vector ops single fp32
    operations: bit-exact.  Sizes that are not multiples of the 32 x 32 transpose tile, flips on and off, crops at both
    borders, an odd column count (the unpaired last index of shuffleColumns stays put)."""
    rng = np.random.default_rng(31)
    for n, colors, W, H, pw, ph in [(37, 3, 45, 41, 33, 32), (5, 1, 9, 9, 9, 9), (64, 3, 40, 40, 32, 32)]:
        im = rng.standard_normal((n, colors * H * W)).astype(np.float32)
        wo = rng.integers(0, W - pw + 1, n).astype(np.float32)
        ho = rng.integers(0, H - ph + 1, n).astype(np.float32)
        wo[0], ho[0] = W - pw, H - ph
        fl = (rng.random(n) > 0.5).astype(np.float32)
        assert np.array_equal(hip.extract_patches(im, wo, ho, fl, W, H, pw, ph), oracle.port.extract_patches(im, wo, ho, fl, W, H, pw, ph))
        perm = rng.permutation(n).astype(np.float32)
        assert np.array_equal(hip.shuffle_columns(im.copy(), perm), oracle.port.shuffle_columns(im.copy(), perm))
        assert np.array_equal(hip.copy_transpose(im), im.T)
        mean, std = rng.standard_normal(im.shape[1]).astype(np.float32), (rng.random(im.shape[1]) + 0.5).astype(np.float32)
        assert np.array_equal(hip.div_by_col_vec(hip.add_col_mult(im.copy(), mean, -1.0), std),
                              oracle.port.div_by_col_vec(oracle.port.add_col_mult(im.copy(), mean, -1.0), std))
        got, ref = hip.normalize_columns(im.copy()), oracle.port.normalize_columns(im.copy())
        assert np.allclose(got, ref, rtol=0, atol=1e-5)      # column mean: parallel vs sequential fp32 sum


@pytest.mark.gpu
def test_conv_up_tail_split_more_tiles_than_slots(hip, matrix_path):
    """648 block tiles on 512 resident slots: tiles 512..647 are computed as K-split pieces whose raw sums
    gg_tail_fix_kernel adds in fixed order before the normal epilogue (accumulate / bias / ReLU).  Same result as the
    whole-K path (CONVNET_GG_NO_TAIL_SPLIT, a -DCONVNET_DIAG build knob, is the A/B switch used when measuring)."""
    g = Geom(N=256, C=64, H=18, W=18, F=256, Ky=3, Kx=3, pady=1, padx=1)
    rng = np.random.default_rng(41)
    x, w, b = rnd(rng, g.in_shape()), rnd(rng, g.filt_shape()), rnd(rng, (g.F,))
    ref = oracle.port.conv_up(g, x, w)
    assert rel_err(hip.conv_up(g, x, w), ref) < TOL
    t0 = rnd(rng, g.out_shape())
    assert rel_err(hip.conv_up(g, x, w, t0.copy(), 1.0), t0 + ref) < TOL
    fused = hip.conv_up_bias_relu(g, x, w, b, relu=True)
    y = np.maximum(ref + b.reshape(g.F, 1, 1, 1), 0.0)
    assert rel_err(fused, y) < TOL


def _random_geoms(seed, count):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < count:
        Ky, Kx = int(rng.integers(1, 6)), int(rng.integers(1, 6))
        sy, sx = int(rng.integers(1, 4)), int(rng.integers(1, 4))
        pady, padx = int(rng.integers(0, min(3, Ky))), int(rng.integers(0, min(3, Kx)))
        H, W = int(rng.integers(Ky, 19)), int(rng.integers(Kx, 19))
        N = int(rng.choice([1, 2, 3, 4, 7, 8, 12, 31, 64, 100, 129]))
        C, F = int(rng.choice([1, 2, 3, 4, 5, 8, 16, 33])), int(rng.choice([1, 3, 4, 16, 31, 64, 96, 100, 130]))
        g = Geom(N=N, C=C, H=H, W=W, F=F, Ky=Ky, Kx=Kx, sy=sy, sx=sx, pady=pady, padx=padx)
        if g.My >= 1 and g.Mx >= 1 and g.N * g.C * g.H * g.W * g.F * Ky * Kx < 4e8:
            out.append(g)
    return out


@pytest.mark.gpu
def test_conv_random_geometries_vs_oracle(hip, matrix_path):
    """Seeded sweep over 48 random geometries (rectangular images / kernels / strides, paddings, every row-tile class,
    vector and scalar paths, accumulate on and off) for fprop, dgrad, wgrad and the fused wgrad+bias."""
    from hip_adapter import conv_outp_bias
    rng = np.random.default_rng(77)
    for i, g in enumerate(_random_geoms(1234, 48)):
        x, w, dy = rnd(rng, g.in_shape()), rnd(rng, g.filt_shape()), rnd(rng, g.out_shape())
        st = float(i % 2)
        t0 = rnd(rng, g.out_shape())
        assert rel_err(hip.conv_up(g, x, w, t0.copy(), st), oracle.port.conv_up(g, x, w, t0.copy(), st)) < TOL, ("up", g)
        t0 = rnd(rng, g.in_shape())
        assert rel_err(hip.conv_down(g, dy, w, t0.copy(), st), oracle.port.conv_down(g, dy, w, t0.copy(), st)) < TOL, ("down", g)
        dw0, db0 = rnd(rng, g.filt_shape()), rnd(rng, (g.F,))
        dw, db = conv_outp_bias(g, x, dy, dw0.copy(), db0.copy(), st, 0.5 / g.N)
        assert rel_err(dw, oracle.port.conv_outp(g, x, dy, dw0.copy(), st, 0.5 / g.N)) < TOL, ("outp", g)
        ref_db = st * db0 + (0.5 / g.N) * dy.reshape(g.F, -1).astype(np.float64).sum(axis=1)
        assert np.allclose(db, ref_db, rtol=3e-5, atol=2e-5 * max(1e-6, np.abs(ref_db).max())), ("db", g)


@pytest.mark.gpu
def test_pooling_random_square_geometries_vs_oracle(hip):
    """Seeded sweep of square pooling set-ups (the CPU oracle only defines pool-undo for square maps / windows, SURVEY §8c
    quirks): K 2-4, stride 1-3, padding 0..K-1, ragged and vector N — generic kernels and the fixed 3x3s2 / 2x2s2 ones."""
    rng = np.random.default_rng(99)
    done = 0
    while done < 30:
        K, s = int(rng.integers(2, 5)), int(rng.integers(1, 4))
        pad = int(rng.integers(0, K))
        H = int(rng.integers(K, 24))
        N, C = int(rng.choice([1, 3, 4, 8, 20, 64, 130])), int(rng.choice([1, 2, 5, 16]))
        g = Geom(N=N, C=C, H=H, W=H, F=C, Ky=K, Kx=K, sy=s, sx=s, pady=pad, padx=pad)
        if g.My < 1 or (g.My - 1) * s - pad >= H:     # last window must start inside the image
            continue
        done += 1
        x = np.maximum(rnd(rng, g.in_shape()), 0)
        dy = rnd(rng, g.pooled_shape())
        mp = hip.max_pool(g, x)
        assert np.array_equal(mp, oracle.port.max_pool(g, x)), g
        assert rel_err(hip.avg_pool(g, x), oracle.port.avg_pool(g, x)) < 1e-6, g
        st = float(done % 2)
        t0 = rnd(rng, g.in_shape())
        assert rel_err(hip.max_pool_undo(g, x, dy, mp, t0.copy(), st), oracle.port.max_pool_undo(g, x, dy, mp, t0.copy(), st)) < 1e-6, g
        assert rel_err(hip.avg_pool_undo(g, dy, t0.copy(), st), oracle.port.avg_pool_undo(g, dy, t0.copy(), st)) < 1e-6, g


@pytest.mark.gpu
def test_conv_up_three_blocks_per_cu_build(hip, matrix_path):
    """1600 block tiles (>= 2 rounds of 768 slots) selects the 3-blocks-per-CU build of gg_kernel (k-row-major B stage,
    wave-uniform tap decode); 1600 = 2 x 768 + 64 also leaves a tail that is K-split.  Padding, accumulate, bias + ReLU."""
    g = Geom(N=256, C=36, H=40, W=40, F=128, Ky=3, Kx=3, pady=1, padx=1)
    rng = np.random.default_rng(43)
    x, w, b = rnd(rng, g.in_shape()), rnd(rng, g.filt_shape()), rnd(rng, (g.F,))
    ref = oracle.port.conv_up(g, x, w)
    assert rel_err(hip.conv_up(g, x, w), ref) < TOL
    t0 = rnd(rng, g.out_shape())
    assert rel_err(hip.conv_up(g, x, w, t0.copy(), 1.0), t0 + ref) < TOL
    assert rel_err(hip.conv_up_bias_relu(g, x, w, b, relu=True), np.maximum(ref + b.reshape(g.F, 1, 1, 1), 0.0)) < TOL


@pytest.mark.gpu
def test_conv_down_mask_with_tail_split(hip, matrix_path):
    """Stride-1 dgrad with the fused ReLU' + dropout' epilogue (convDownMask) at 648 tiles: whole-K blocks and K-split tail
    tiles must apply accumulate -> mask -> post-scale identically."""
    from convnet_amd.matrix import Matrix
    from hip_adapter import _desc, _mat
    g = Geom(N=256, C=256, H=18, W=18, F=64, Ky=3, Kx=3, pady=1, padx=1)
    rng = np.random.default_rng(47)
    dy, w = rnd(rng, g.out_shape()), rnd(rng, g.filt_shape())
    state = np.maximum(rnd(rng, g.in_shape()), 0)          # the source layer's post-ReLU state: ~half zeros
    ref = oracle.port.conv_down(g, dy, w)
    for st in (0.0, 1.0):
        t0 = rnd(rng, g.in_shape())
        D = hip._act(dy, g.N, g.Mx, g.My, g.F)
        W = _mat(w, g.F, g.K, (g.F, g.Kx, g.Ky, g.C))
        S = hip._act(state, g.N, g.W, g.H, g.C)
        T = hip._act(t0, g.N, g.W, g.H, g.C)
        Matrix.ConvDownMask(D, W, S, T, _desc(g), st, 1.25)
        want = np.where(state > 0, (st * t0 + ref) * 1.25, 0.0).astype(np.float32)
        assert rel_err(T.ToNumpy().reshape(g.in_shape()), want) < TOL


def test_sgd_multi_equals_one_call_per_tensor(hip):
    """sgd_momentum_step_multi (round 6): several tensors, each with its own hyper-parameters, in one launch per 16 — bit for bit what
    one sgd_momentum_step per tensor gives (and that entry is pinned to the reference's golden vectors above): 19 tensors (two launches),
    sizes from 1 to 1.3 M elements, unaligned ones, l2 / clip on and off."""
    from convnet_amd.matrix import Matrix
    from hip_adapter import _mat
    rng = np.random.default_rng(23)
    sizes = [(96, 147), (1, 96), (256, 2400), (1, 256), (384, 2304), (1, 384), (384, 3456), (1, 1), (256, 3456), (1, 1000), (7, 13), (3, 5),
             (64, 64), (1, 4096), (33, 77), (128, 9), (1, 3), (1024, 1024), (5, 1)]
    hyper = [(0.0005 * (i % 3), 0.0 if i % 4 else 0.5, 0.01 / (1 + i % 5), 0.9 - 0.1 * (i % 2)) for i in range(len(sizes))]
    data = [(rnd(rng, (c, r)), rnd(rng, (c, r)), rnd(rng, (c, r))) for (r, c) in sizes]

    def run(multi):
        mats = [tuple(_mat(a.copy(), r, c) for a in d) for d, (r, c) in zip(data, sizes)]
        items = [(g, w, h, *hp) for (g, w, h), hp in zip(mats, hyper)]
        if multi:
            Matrix.SGDMomentumStepMulti(items)
        else:
            for it in items:
                Matrix.SGDMomentumStep(*it)
        return [tuple(m.ToNumpy() for m in t) for t in mats]
    one, many = run(False), run(True)
    for (g1, w1, h1), (g2, w2, h2) in zip(one, many):
        assert np.array_equal(g1, g2) and np.array_equal(w1, w2) and np.array_equal(h1, h2)
