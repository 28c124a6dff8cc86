"""Deferred epilogues (include/convnet_hip.h: convnet_hip_set_deferred_epilogues; csrc/state.hip, common.h: PendingOp): with the switch on,
the reference host's UNFUSED call sequences — convUp, reshape, add_row_vec(bias), lower_bound_scalar(0) (src/conv_edge.cc:138-149 +
src/layer.cc:549-551); convDown / MaxPoolUndo then apply_rectified_linear_deriv (src/layer.cc:556-558); ResponseNormCrossMap then
lower_bound_scalar — run as ONE fused launch each, with the eager sequence's results bit for bit, and any other call in between
launches the parked one first."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import Geom  # noqa: E402


@pytest.fixture(scope="module")
def lib():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from convnet_amd.matrix import Matrix
    Matrix.SetupCUDADevice(0)
    Matrix.InitRandom(42)
    from convnet_amd import _lib
    _lib.lib.convnet_hip_set_matrix_path(1)
    yield _lib
    _lib.lib.convnet_hip_set_deferred_epilogues(0)


def kernels_of(lib, fn, defer):
    """runs fn() with the switch set, returns (result, number of element-wise calls the library absorbed into a parked call)"""
    lib.lib.convnet_hip_set_deferred_epilogues(1 if defer else 0)
    before = lib.lib.convnet_hip_deferred_absorbed()
    out = fn()
    n = lib.lib.convnet_hip_deferred_absorbed() - before
    lib.lib.convnet_hip_set_deferred_epilogues(0)
    return out, n


def mats(g, rng):
    from hip_adapter import _mat
    x = rng.standard_normal(g.in_shape()).astype(np.float32)
    w = rng.standard_normal(g.filt_shape()).astype(np.float32)
    b = rng.standard_normal((g.F,)).astype(np.float32)
    return (_mat(x, g.N, g.W * g.H * g.C, (g.N, g.W, g.H, g.C)), _mat(w, g.F, g.K, (g.F, g.Kx, g.Ky, g.C)), _mat(b, 1, g.F))


@pytest.mark.parametrize("g", [Geom(N=64, C=32, H=13, W=13, F=96, Ky=3, Kx=3, pady=1, padx=1),
                               Geom(N=32, C=3, H=31, W=31, F=96, Ky=7, Kx=7, sy=2, sx=2, pady=1, padx=1)], ids=["c32k3", "conv1"])
def test_forward_sequence_is_one_launch_with_the_eager_bits(lib, g):
    from convnet_amd.matrix import Matrix
    from hip_adapter import _mat, _desc
    rng = np.random.default_rng(61)
    xm, wm, bm = mats(g, rng)

    def seq():
        t = _mat(np.zeros(g.out_shape(), np.float32), g.N, g.Mx * g.My * g.F, (g.N, g.Mx, g.My, g.F))
        Matrix.ConvUp(xm, wm, t, _desc(g), 0.0)          # conv_edge.cc:142
        t.Reshape(-1, g.F)
        t.AddRowVec(bm)                                   # :147
        t.Reshape(g.N, -1)
        t.LowerBound(0.0)                                 # layer.cc:550
        return t.ToNumpy()
    eager, k_eager = kernels_of(lib, seq, False)
    fused, k_fused = kernels_of(lib, seq, True)
    assert np.array_equal(eager, fused)
    assert k_eager == 0 and k_fused == 2, (k_eager, k_fused)   # the bias and the ReLU joined the convolution
    assert (fused >= 0).all() and (fused > 0).any()


def test_backward_sequences_and_response_norm(lib):
    from convnet_amd.matrix import Matrix
    from hip_adapter import _mat, _desc
    rng = np.random.default_rng(62)
    g = Geom(N=64, C=96, H=13, W=13, F=32, Ky=3, Kx=3, pady=1, padx=1)
    dy = rng.standard_normal(g.out_shape()).astype(np.float32)
    w = rng.standard_normal(g.filt_shape()).astype(np.float32)
    state = rng.standard_normal(g.in_shape()).astype(np.float32)
    dym = _mat(dy, g.N, g.Mx * g.My * g.F, (g.N, g.Mx, g.My, g.F))
    wm = _mat(w, g.F, g.K, (g.F, g.Kx, g.Ky, g.C))
    sm = _mat(state, g.N, g.W * g.H * g.C, (g.N, g.W, g.H, g.C))

    def conv_down_relu():
        t = _mat(np.zeros(g.in_shape(), np.float32), g.N, g.W * g.H * g.C, (g.N, g.W, g.H, g.C))
        Matrix.ConvDown(dym, wm, t, _desc(g), 0.0)
        t.ApplyDerivativeOfReLU(sm)                       # layer.cc:557
        return t.ToNumpy()
    eager, k_eager = kernels_of(lib, conv_down_relu, False)
    fused, k_fused = kernels_of(lib, conv_down_relu, True)
    assert np.array_equal(eager, fused) and k_eager == 0 and k_fused == 1, (k_eager, k_fused)

    p = Geom(N=64, C=16, H=13, W=13, F=16, Ky=3, Kx=3, sy=2, sx=2, pady=1, padx=1)
    x = rng.standard_normal(p.in_shape()).astype(np.float32)
    xm = _mat(x, p.N, p.W * p.H * p.C, (p.N, p.W, p.H, p.C))
    ym = _mat(np.zeros(p.pooled_shape(), np.float32), p.N, p.Mx * p.My * p.C, (p.N, p.Mx, p.My, p.C))
    Matrix.ConvMaxPool(xm, ym, _desc(p, True))
    gm = _mat(rng.standard_normal(p.pooled_shape()).astype(np.float32), p.N, p.Mx * p.My * p.C, (p.N, p.Mx, p.My, p.C))

    def pool_undo_relu():
        t = _mat(np.zeros(p.in_shape(), np.float32), p.N, p.W * p.H * p.C, (p.N, p.W, p.H, p.C))
        Matrix.ConvMaxPoolUndo(xm, gm, ym, t, _desc(p, True), 0.0)
        t.ApplyDerivativeOfReLU(xm)
        return t.ToNumpy()
    eager, k_eager = kernels_of(lib, pool_undo_relu, False)
    fused, k_fused = kernels_of(lib, pool_undo_relu, True)
    assert np.array_equal(eager, fused) and k_eager == 0 and k_fused == 1, (k_eager, k_fused)

    def rnorm_relu():
        t = _mat(np.zeros(p.in_shape(), np.float32), p.N, p.W * p.H * p.C, (p.N, p.W, p.H, p.C))
        Matrix.ConvResponseNormCrossMap(xm, t, p.C, 5, 0.001, 0.75, False)
        t.LowerBound(0.0)
        return t.ToNumpy()
    eager, k_eager = kernels_of(lib, rnorm_relu, False)
    fused, k_fused = kernels_of(lib, rnorm_relu, True)
    assert np.array_equal(eager, fused) and k_eager == 0 and k_fused == 1, (k_eager, k_fused)


def test_anything_else_flushes_the_parked_call_first(lib):
    """a read-back right behind convUp; an element-wise call on ANOTHER matrix; a bias of the wrong length; a non-zero bound: the
    parked convolution is launched unfused, then the call runs as written"""
    from convnet_amd.matrix import Matrix
    from hip_adapter import _mat, _desc
    rng = np.random.default_rng(63)
    g = Geom(N=64, C=32, H=9, W=9, F=96, Ky=3, Kx=3, pady=1, padx=1)
    xm, wm, bm = mats(g, rng)
    other = _mat(rng.standard_normal((g.F, 40)).astype(np.float32), 40, g.F)

    def seq():
        t = _mat(np.zeros(g.out_shape(), np.float32), g.N, g.Mx * g.My * g.F, (g.N, g.Mx, g.My, g.F))
        Matrix.ConvUp(xm, wm, t, _desc(g), 0.0)
        a = t.ToNumpy().copy()                            # read-back: flush
        Matrix.ConvUp(xm, wm, t, _desc(g), 0.0)
        other.AddRowVec(bm)                               # another matrix: flush, then add there
        b = t.ToNumpy().copy()
        Matrix.ConvUp(xm, wm, t, _desc(g), 0.0)
        t.LowerBound(0.5)                                 # not a ReLU: flush, then clamp at 0.5
        c = t.ToNumpy().copy()
        return a, b, c
    e, _ = kernels_of(lib, seq, False)
    f, n = kernels_of(lib, seq, True)
    assert n == 0
    assert np.array_equal(e[0], f[0]) and np.array_equal(e[1], f[1]) and np.array_equal(e[2], f[2])
    assert np.array_equal(e[0], e[1]) and (e[2] >= 0.5).all()
