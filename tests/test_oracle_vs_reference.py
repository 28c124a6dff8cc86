"""Pins oracle/convnet_oracle.c (the portable restatement) against the reference's own compiled
CPU path (oracle/_ref, built from /root/reference by oracle/Makefile).  Skipped where the
reference build is not present; the committed golden vectors (test_oracle_golden.py) cover that
case.  Shapes include py/test_conv.py's Test2D case (N=128, 12x12x32 -> 64, k3 s2 p1,
sizeF=8, add_scale=.005, pow_scale=.75; py/test_conv.py:394-418).
"""
import numpy as np
import pytest

import oracle
from oracle import Geom

pytestmark = pytest.mark.skipif(oracle.ref is None, reason="oracle/_ref not built (no /root/reference)")

CONV_CASES = [
    Geom(N=128, C=32, H=12, W=12, F=64, Ky=3, Kx=3, sy=2, sx=2, pady=1, padx=1),   # Test2D
    Geom(N=5, C=3, H=17, W=15, F=7, Ky=5, Kx=3, sy=2, sx=1, pady=2, padx=1),        # ragged, rectangular
    Geom(N=16, C=1, H=28, W=28, F=48, Ky=4, Kx=4),                                   # mnist-conv conv1
    Geom(N=4, C=8, H=9, W=9, F=16, Ky=3, Kx=3, sy=1, sx=1, pady=1, padx=1),
    Geom(N=3, C=3, H=23, W=23, F=6, Ky=7, Kx=7, sy=2, sx=2, pady=1, padx=1),         # conv1-like
]
POOL_CASES = [
    Geom(N=128, C=32, H=12, W=12, F=32, Ky=3, Kx=3, sy=2, sx=2, pady=1, padx=1),
    Geom(N=7, C=5, H=11, W=11, F=5, Ky=3, Kx=3, sy=2, sx=2, pady=1, padx=1),         # AlexNet pool geometry
    Geom(N=6, C=4, H=25, W=25, F=4, Ky=4, Kx=4, sy=2, sx=2),                          # mnist-conv pool
    Geom(N=3, C=2, H=8, W=8, F=2, Ky=2, Kx=2, sy=2, sx=2),
]


def rel(a, b):
    # the reference's own metric: max|a-b| / mean|a+b| (py/test_conv.py:382-385)
    return float(np.abs(a - b).max() / (np.abs(a + b).mean() + 1e-30))


def rnd(rng, shape):
    return rng.standard_normal(shape).astype(np.float32)


@pytest.mark.parametrize("g", CONV_CASES)
def test_conv_up_down_outp(g):
    rng = np.random.default_rng(1)
    x, w, dy = rnd(rng, g.in_shape()), rnd(rng, g.filt_shape()), rnd(rng, g.out_shape())
    for st in (0.0, 1.0):
        t0 = rnd(rng, g.out_shape())
        a = oracle.port.conv_up(g, x, w, t0.copy(), st, 1.0)
        b = oracle.ref.conv_up(g, x, w, t0.copy(), st, 1.0)
        assert rel(a, b) < 1e-6
        t0 = rnd(rng, g.in_shape())
        a = oracle.port.conv_down(g, dy, w, t0.copy(), st, 1.0)
        b = oracle.ref.conv_down(g, dy, w, t0.copy(), st, 1.0)
        assert rel(a, b) < 1e-6
        t0 = rnd(rng, g.filt_shape())
        a = oracle.port.conv_outp(g, x, dy, t0.copy(), st, 0.37)
        b = oracle.ref.conv_outp(g, x, dy, t0.copy(), st, 0.37)
        assert rel(a, b) < 1e-6


@pytest.mark.parametrize("g", POOL_CASES)
def test_pools(g):
    rng = np.random.default_rng(2)
    x = np.maximum(rnd(rng, g.in_shape()), 0)  # post-ReLU: many exact ties at 0
    dy = rnd(rng, g.pooled_shape())
    a, b = oracle.port.max_pool(g, x), oracle.ref.max_pool(g, x)
    assert np.array_equal(a, b)
    a2, b2 = oracle.port.avg_pool(g, x), oracle.ref.avg_pool(g, x)
    assert rel(a2, b2) < 1e-6
    for st in (0.0, 1.0):
        t0 = rnd(rng, g.in_shape())
        u, v = oracle.port.max_pool_undo(g, x, dy, a, t0.copy(), st), oracle.ref.max_pool_undo(g, x, dy, b, t0.copy(), st)
        assert rel(u, v) < 1e-6
        if g.H == g.W and g.Ky == g.Kx:  # reference CPU avg-undo assumes square maps
            u, v = oracle.port.avg_pool_undo(g, dy, t0.copy(), st), oracle.ref.avg_pool_undo(g, dy, t0.copy(), st)
            assert rel(u, v) < 1e-6


@pytest.mark.parametrize("shape,size_f,blocked", [((32, 6, 6, 128), 8, False), ((96, 3, 3, 5), 24, False),
                                                  ((20, 2, 3, 4), 5, True), ((7, 2, 2, 3), 3, False)])
def test_rnorm(shape, size_f, blocked):
    rng = np.random.default_rng(3)
    x, dy = rnd(rng, shape), rnd(rng, shape)
    a, b = oracle.port.rnorm(x, size_f, 0.005, 0.75, blocked), oracle.ref.rnorm(x, size_f, 0.005, 0.75, blocked)
    assert rel(a, b) < 1e-6
    a, b = oracle.port.rnorm_undo(dy, x, size_f, 0.005, 0.75, blocked), oracle.ref.rnorm_undo(dy, x, size_f, 0.005, 0.75, blocked)
    assert rel(a, b) < 2e-6


def test_dense_ops():
    rng = np.random.default_rng(4)
    N, D, F = 9, 37, 11
    x = rnd(rng, (D, N))      # (N, D) col-major
    w = rnd(rng, (D, F))      # (F, D) col-major
    dy = rnd(rng, (F, N))     # (N, F) col-major
    # fc fwd: out = in * W^T (src/fc_edge.cc:54)
    for beta in (0.0, 1.0):
        t0 = rnd(rng, (F, N))
        assert rel(oracle.port.dot(x, w, t0.copy(), beta, 1.0, False, True), oracle.ref.dot(x, w, t0.copy(), beta, 1.0, False, True)) < 1e-5
        t0 = rnd(rng, (D, N))  # dgrad: d_in = d_out * W (fc_edge.cc:66)
        assert rel(oracle.port.dot(dy, w, t0.copy(), beta, 1.0), oracle.ref.dot(dy, w, t0.copy(), beta, 1.0)) < 1e-5
        t0 = rnd(rng, (D, F))  # wgrad: dW = d_out^T * in (fc_edge.cc:74)
        assert rel(oracle.port.dot(dy, x, t0.copy(), beta, 0.25, True, False), oracle.ref.dot(dy, x, t0.copy(), beta, 0.25, True, False)) < 1e-5
    b = rnd(rng, (F,))
    assert rel(oracle.port.add_row_vec(dy.copy(), b), oracle.ref.add_row_vec(dy.copy(), b)) < 1e-7
    for axis, n in ((0, F), (1, N)):
        t0 = rnd(rng, (n,))
        assert rel(oracle.port.sum_by_axis(dy, t0.copy(), axis, 0.5, 1.0), oracle.ref.sum_by_axis(dy, t0.copy(), axis, 0.5, 1.0)) < 1e-5
    assert np.array_equal(oracle.port.lower_bound(dy.copy(), 0.0), oracle.ref.lower_bound(dy.copy(), 0.0))
    assert np.array_equal(oracle.port.upper_bound_mod(dy.copy(), 0.4), oracle.ref.upper_bound_mod(dy.copy(), 0.4))
    st = np.maximum(rnd(rng, (F, N)), 0)
    assert np.array_equal(oracle.port.relu_deriv(dy.copy(), st), oracle.ref.relu_deriv(dy.copy(), st))


def test_softmax_family():
    rng = np.random.default_rng(5)
    N, K = 13, 10
    z = (3 * rnd(rng, (K, N)))
    labels = rng.integers(0, K, N).astype(np.float32)
    p, q = oracle.port.softmax_row_major(z.copy()), oracle.ref.softmax_row_major(z.copy())
    assert rel(p, q) < 1e-6
    assert np.array_equal(oracle.port.softmax_grad_row_major(q, labels), oracle.ref.softmax_grad_row_major(q, labels))
    assert np.array_equal(oracle.port.softmax_correct_row_major(q, labels), oracle.ref.softmax_correct_row_major(q, labels))
    assert rel(oracle.port.softmax_ce_row_major(q, labels), oracle.ref.softmax_ce_row_major(q, labels)) < 1e-6


@pytest.mark.parametrize("limit,constraint", [(0.0, 0.0), (0.8, 0.0), (0.0, 1.5)])
def test_sgd_step(limit, constraint):
    rng = np.random.default_rng(6)
    F, D = 12, 30
    g0, w0, h0 = rnd(rng, (D, F)), rnd(rng, (D, F)), rnd(rng, (D, F))
    a = [g0.copy(), w0.copy(), h0.copy()]
    b = [g0.copy(), w0.copy(), h0.copy()]
    oracle.port.sgd_step(*a, 5e-4, 0.9, 0.01, 0.7, limit, constraint)
    oracle.ref.sgd_step(*b, 5e-4, 0.9, 0.01, 0.7, limit, constraint)
    for u, v in zip(a, b):
        assert rel(u, v) < 1e-6


def test_random_conv_geometries_port_equals_reference():
    """Seeded sweep over 40 geometries (rectangular maps and kernels, stride 1-3 per axis, padding up to kernel/2, N and F not
    multiples of 4): the restatement must follow the reference's CPU conv for all three passes, with and without accumulation."""
    rng = np.random.default_rng(2024)
    done = 0
    while done < 40:
        Ky, Kx = int(rng.integers(1, 6)), int(rng.integers(1, 6))
        sy, sx = int(rng.integers(1, 4)), int(rng.integers(1, 4))
        pady, padx = int(rng.integers(0, Ky // 2 + 1)), int(rng.integers(0, Kx // 2 + 1))
        H, W = int(rng.integers(Ky, Ky + 9)), int(rng.integers(Kx, Kx + 9))
        g = Geom(N=int(rng.integers(1, 10)), C=int(rng.integers(1, 7)), H=H, W=W, F=int(rng.integers(1, 10)), Ky=Ky, Kx=Kx, sy=sy, sx=sx,
                 pady=pady, padx=padx)
        if g.My < 1 or g.Mx < 1:
            continue
        done += 1
        x, w, dy = rnd(rng, g.in_shape()), rnd(rng, g.filt_shape()), rnd(rng, g.out_shape())
        st = float(done % 2)
        t0 = rnd(rng, g.out_shape())
        assert rel(oracle.port.conv_up(g, x, w, t0.copy(), st, 1.0), oracle.ref.conv_up(g, x, w, t0.copy(), st, 1.0)) < 2e-6, g
        t0 = rnd(rng, g.in_shape())
        assert rel(oracle.port.conv_down(g, dy, w, t0.copy(), st, 1.0), oracle.ref.conv_down(g, dy, w, t0.copy(), st, 1.0)) < 2e-6, g
        t0 = rnd(rng, g.filt_shape())
        assert rel(oracle.port.conv_outp(g, x, dy, t0.copy(), st, 0.5), oracle.ref.conv_outp(g, x, dy, t0.copy(), st, 0.5)) < 2e-6, g


def test_random_square_pool_and_rnorm_port_equals_reference():
    """Seeded sweep: 30 square pooling set-ups (the reference's CPU undo assumes square maps and kernels) with post-ReLU ties, and
    20 response-norm set-ups (window 1..C, blocked and sliding)."""
    rng = np.random.default_rng(77)
    done = 0
    while done < 30:
        K, s = int(rng.integers(2, 5)), int(rng.integers(1, 4))
        pad = int(rng.integers(0, K // 2 + 1))
        H = int(rng.integers(K, K + 10))
        C = int(rng.integers(1, 6))
        g = Geom(N=int(rng.integers(1, 9)), C=C, H=H, W=H, F=C, Ky=K, Kx=K, sy=s, sx=s, pady=pad, padx=pad)
        if g.My < 1 or (g.My - 1) * s - pad >= H:     # every window must touch the map
            continue
        done += 1
        x = np.maximum(rnd(rng, g.in_shape()), 0)
        dy = rnd(rng, g.pooled_shape())
        a, b = oracle.port.max_pool(g, x), oracle.ref.max_pool(g, x)
        assert np.array_equal(a, b), g
        assert rel(oracle.port.avg_pool(g, x), oracle.ref.avg_pool(g, x)) < 1e-6, g
        st = float(done % 2)
        t0 = rnd(rng, g.in_shape())
        assert rel(oracle.port.max_pool_undo(g, x, dy, a, t0.copy(), st), oracle.ref.max_pool_undo(g, x, dy, b, t0.copy(), st)) < 1e-6, g
        assert rel(oracle.port.avg_pool_undo(g, dy, t0.copy(), st), oracle.ref.avg_pool_undo(g, dy, t0.copy(), st)) < 1e-6, g
    for i in range(20):
        C = int(rng.integers(2, 40))
        size_f = int(rng.integers(1, C + 1))
        blocked = bool(i % 3 == 0)
        x = rnd(rng, (C, int(rng.integers(1, 5)), int(rng.integers(1, 5)), int(rng.integers(1, 9))))
        dy = rnd(rng, x.shape)
        a, b = oracle.port.rnorm(x, size_f, 0.005, 0.75, blocked), oracle.ref.rnorm(x, size_f, 0.005, 0.75, blocked)
        assert rel(a, b) < 2e-6, (C, size_f, blocked)
        u, v = oracle.port.rnorm_undo(dy, x, size_f, 0.005, 0.75, blocked), oracle.ref.rnorm_undo(dy, x, size_f, 0.005, 0.75, blocked)
        assert rel(u, v) < 1e-5, (C, size_f, blocked)
