"""GPU: the REFERENCE'S OWN host code on this library (both seams at once).

oracle/_ref/libref_matrix_seam.so is the reference's GPU `class Matrix` — src/matrix.{h,cc} compiled unmodified against the
reference's own cudamat headers (oracle/Makefile target `seam`; the only stand-in is a 6-line <cublas.h>) — linked to
convnet_amd/lib/libconvnet_hip.so.  These tests call Matrix::ConvUp / ConvDown / ConvOutp / ConvMaxPool(Undo) /
ConvAvgPool(Undo) / ConvResponseNormCrossMap(Undo) / Dot / AddRowVec / SumRows / LowerBound / ApplyDerivativeOfReLU exactly
as the reference's edges and layers do, and compare with the CPU oracle.  The .so is built where /root/reference is mounted
and travels with the snapshot; without it the tests skip."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle  # noqa: E402
from oracle import Geom  # noqa: E402
from golden_cases import rel_err  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEAM = os.path.join(ROOT, "oracle", "_ref", "libref_matrix_seam.so")
TOL = 1e-4

F, I = ctypes.c_float, ctypes.c_int


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


@pytest.fixture(scope="module")
def seam():
    if not os.path.exists(SEAM):
        pytest.skip("oracle/_ref/libref_matrix_seam.so not built (needs the reference tree at build time)")
    import torch
    assert torch.cuda.is_available()
    from convnet_amd import _lib     # loads libconvnet_hip.so after torch (one HIP runtime)
    ctypes.CDLL(_lib.LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    lib = ctypes.CDLL(SEAM)
    lib.seam_init(0)
    return lib


def rnd(rng, shape):
    return rng.standard_normal(shape).astype(np.float32)


@pytest.mark.parametrize("g", [Geom(N=8, C=3, H=32, W=32, F=96, Ky=7, Kx=7, sy=2, sx=2, pady=1, padx=1),
                               Geom(N=16, C=96, H=13, W=13, F=128, Ky=5, Kx=5, sy=2, sx=2),
                               Geom(N=12, C=32, H=9, W=9, F=48, Ky=3, Kx=3, pady=1, padx=1),
                               Geom(N=5, C=3, H=17, W=15, F=7, Ky=5, Kx=3, sy=2, sx=1, pady=2, padx=1)],
                         ids=lambda g: f"N{g.N}C{g.C}F{g.F}k{g.Ky}s{g.sy}")
def test_reference_matrix_conv_ops_run_on_this_library(seam, g):
    rng = np.random.default_rng(5)
    x, w, dy = rnd(rng, g.in_shape()), rnd(rng, g.filt_shape()), rnd(rng, g.out_shape())

    def conv(op, a, b, out, st, so):
        seam.seam_conv(I(op), _p(a), _p(b), _p(out), I(g.N), I(g.C), I(g.H), I(g.W), I(g.F), I(g.Ky), I(g.Kx), I(g.sy), I(g.sx),
                       I(g.pady), I(g.padx), I(g.My), I(g.Mx), F(st), F(so))
        return out
    for st in (0.0, 1.0):
        t0 = rnd(rng, g.out_shape())
        assert rel_err(conv(0, x, w, t0.copy(), st, 1.0), oracle.port.conv_up(g, x, w, t0.copy(), st)) < TOL
        t0 = rnd(rng, g.in_shape())
        assert rel_err(conv(1, dy, w, t0.copy(), st, 1.0), oracle.port.conv_down(g, dy, w, t0.copy(), st)) < TOL
        t0 = rnd(rng, g.filt_shape())
        assert rel_err(conv(2, x, dy, t0.copy(), st, 0.3 / g.N), oracle.port.conv_outp(g, x, dy, t0.copy(), st, 0.3 / g.N)) < TOL


@pytest.mark.parametrize("g", [Geom(N=16, C=24, H=26, W=26, F=24, Ky=3, Kx=3, sy=2, sx=2, pady=1, padx=1),
                               Geom(N=6, C=5, H=12, W=12, F=5, Ky=2, Kx=2, sy=2, sx=2)], ids=lambda g: f"N{g.N}k{g.Ky}")
def test_reference_matrix_pooling_runs_on_this_library(seam, g):
    rng = np.random.default_rng(6)
    x = np.maximum(rnd(rng, g.in_shape()), 0)
    dy = rnd(rng, g.pooled_shape())

    def pool(op, xin, dyin, yin, out, st=0.0):
        seam.seam_pool(I(op), _p(xin), _p(dyin), _p(yin), _p(out), I(g.N), I(g.C), I(g.H), I(g.W), I(g.Ky), I(g.sy), I(g.pady), I(g.My),
                       I(g.Mx), F(st))
        return out
    zeros = np.zeros(g.pooled_shape(), np.float32)
    mp = pool(0, x, zeros, zeros, zeros.copy())
    assert np.array_equal(mp, oracle.port.max_pool(g, x))
    assert rel_err(pool(1, x, zeros, zeros, zeros.copy()), oracle.port.avg_pool(g, x)) < 1e-6
    for st in (0.0, 1.0):
        t0 = rnd(rng, g.in_shape())
        assert rel_err(pool(2, x, dy, mp, t0.copy(), st), oracle.port.max_pool_undo(g, x, dy, mp, t0.copy(), st)) < 1e-6
        assert rel_err(pool(3, x, dy, mp, t0.copy(), st), oracle.port.avg_pool_undo(g, dy, t0.copy(), st)) < 1e-6


def test_reference_matrix_rnorm_dot_and_layer_helpers_run_on_this_library(seam):
    rng = np.random.default_rng(7)
    N, C, P = 12, 32, 25
    x, dy = rnd(rng, (C, P, 1, N)), rnd(rng, (C, P, 1, N))
    out = np.zeros_like(x)
    seam.seam_rnorm(I(0), _p(x), _p(dy), _p(out), I(N), I(C), I(P), I(8), F(0.005), F(0.75), I(0))
    assert rel_err(out, oracle.port.rnorm(x, 8, 0.005, 0.75, False)) < TOL
    seam.seam_rnorm(I(1), _p(x), _p(dy), _p(out), I(N), I(C), I(P), I(8), F(0.005), F(0.75), I(0))
    assert rel_err(out, oracle.port.rnorm_undo(dy, x, 8, 0.005, 0.75, False)) < TOL
    # FC: out = in * W^T (fc_edge.cc:54), d_in = d_out * W (:66), dW = d_out^T * in / N (:75)
    D, Fo = 200, 70
    a, w, d = rnd(rng, (D, N)), rnd(rng, (D, Fo)), rnd(rng, (Fo, N))     # numpy (cols, rows)
    c = np.zeros((Fo, N), np.float32)
    seam.seam_dot(_p(a), I(N), I(D), I(0), _p(w), I(Fo), I(D), I(1), _p(c), I(N), I(Fo), F(0.0), F(1.0))
    assert rel_err(c, oracle.port.dot(a, w, np.zeros((Fo, N), np.float32), 0.0, 1.0, False, True)) < TOL
    c = np.zeros((D, N), np.float32)
    seam.seam_dot(_p(d), I(N), I(Fo), I(0), _p(w), I(Fo), I(D), I(0), _p(c), I(N), I(D), F(0.0), F(1.0))
    assert rel_err(c, oracle.port.dot(d, w, np.zeros((D, N), np.float32), 0.0, 1.0)) < TOL
    c = rnd(rng, (D, Fo))
    ref = oracle.port.dot(d, a, c.copy(), 1.0, 1.0 / N, True, False)
    seam.seam_dot(_p(d), I(N), I(Fo), I(1), _p(a), I(N), I(D), I(0), _p(c), I(Fo), I(D), F(1.0), F(1.0 / N))
    assert rel_err(c, ref) < TOL
    # AddRowVec / SumRows / LowerBound / ReLU'
    m, v = rnd(rng, (Fo, N)), rnd(rng, (Fo,))
    ref = oracle.port.add_row_vec(m.copy(), v)
    seam.seam_misc(I(0), _p(m), I(N), I(Fo), _p(v), F(0), F(0))
    assert np.allclose(m, ref, rtol=1e-6)
    t = rnd(rng, (Fo,))
    ref = oracle.port.sum_by_axis(m, t.copy(), 0, 0.5, 1.0)
    seam.seam_misc(I(1), _p(m), I(N), I(Fo), _p(t), F(1.0), F(0.5))
    assert np.allclose(t, ref, rtol=1e-5, atol=1e-5)
    ref = oracle.port.lower_bound(m.copy(), 0.0)
    seam.seam_misc(I(2), _p(m), I(N), I(Fo), _p(v), F(0.0), F(0))
    assert np.array_equal(m, ref)
    deriv = rnd(rng, (Fo, N))
    ref = oracle.port.relu_deriv(deriv.copy(), m)
    seam.seam_misc(I(3), _p(deriv), I(N), I(Fo), _p(m), F(0), F(0))
    assert np.array_equal(deriv, ref)


def test_reference_matrix_output_layer_and_sgd_sequence_run_on_this_library(seam):
    """SoftmaxLayer::ApplyActivation + CrossEntropyMultinomial (deriv, correct count, CE) and the SGDOptimizer::Optimize call
    sequence (AddMult l2, UpperBoundMod clip, Mult eps, history update, parameter update, NormLimitByAxis) issued by the
    reference's Matrix methods: softmax to 1e-5, the SGD step to an ulp against the oracle."""
    rng = np.random.default_rng(8)
    N, classes = 37, 10
    logits = (3 * rnd(rng, (classes, N))).astype(np.float32)
    labels = rng.integers(0, classes, N).astype(np.float32)
    probs = logits.copy()
    deriv, correct, ce = np.zeros((classes, N), np.float32), np.zeros(N, np.float32), np.zeros(N, np.float32)
    seam.seam_softmax(_p(probs), _p(labels), _p(deriv), _p(correct), _p(ce), I(N), I(classes))
    p_ref = oracle.port.softmax_row_major(logits.copy())
    assert np.allclose(probs, p_ref, rtol=1e-5, atol=1e-7)
    assert np.allclose(deriv, oracle.port.softmax_grad_row_major(p_ref.copy(), labels), rtol=1e-5, atol=1e-6)
    assert np.array_equal(correct, np.asarray(oracle.port.softmax_correct_row_major(p_ref, labels)).reshape(-1))
    assert np.allclose(ce, np.asarray(oracle.port.softmax_ce_row_major(p_ref, labels)).reshape(-1), rtol=1e-5, atol=1e-6)
    rows, cols = 30, 12
    g0, w0, h0 = rnd(rng, (cols, rows)), rnd(rng, (cols, rows)), rnd(rng, (cols, rows))
    g, w, h = g0.copy(), w0.copy(), h0.copy()
    seam.seam_sgd(_p(g), _p(w), _p(h), I(rows), I(cols), F(5e-4), F(0.9), F(0.01), F(0.7), F(0.0))
    gr, wr, hr = g0.copy(), w0.copy(), h0.copy()
    oracle.port.sgd_step(gr, wr, hr, 5e-4, 0.9, 0.01, 0.7, 0.0, 0.0)
    # the unfused Matrix calls are single fp32 ops each; add_mult (g += l2*w) contracts to an fma on the GPU (as cublasSaxpy does
    # on the reference's own GPU path), so this sequence agrees with the CPU oracle to an ulp, not bit for bit — the
    # bit-exact guarantee is the fused sgd_momentum_step entry (tests/test_hip_parity.py)
    assert np.allclose(g, gr, rtol=3e-7, atol=1e-9) and np.allclose(h, hr, rtol=3e-7, atol=1e-9) and np.allclose(w, wr, rtol=3e-7, atol=1e-9)
    g, w, h = g0.copy(), (3 * w0).copy(), h0.copy()
    seam.seam_sgd(_p(g), _p(w), _p(h), I(rows), I(cols), F(0.0), F(0.0), F(0.01), F(0.5), F(2.0))
    gr, wr, hr = g0.copy(), (3 * w0).copy(), h0.copy()
    oracle.port.sgd_step(gr, wr, hr, 0.0, 0.0, 0.01, 0.5, 2.0, 0.0)
    assert np.allclose(w, wr, rtol=2e-6, atol=1e-7) and np.abs(np.linalg.norm(w.reshape(cols, rows), axis=0)).max() <= 2.0 * (1 + 1e-5)
