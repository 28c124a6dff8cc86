"""wgw_kernel (convnet_amd/csrc/wgrad_wide.hip, convnet_hip_set_wgrad_tile(1)): the weight gradients on a 256 x 256 / 256 x 192 tile with
four waves of 128 x 128, against the CPU oracle (the reference's conv_outp, cudamat_conv_gemm.cu:827-960, and dot TN) and against
wg_kernel.  Tolerance as every kernel test: max|a-b| / mean|a+b| < 1e-4 (py/test_conv.py:382-392).

Written at the end of round 4, first run on the MI355X in round 5 (all cases green at the first attempt; conv2-5 weight gradients
24-29 % faster than wg_kernel, profiles/r05_wide_kernels.md) and the library default since (convnet_hip_set_wgrad_tile(1))."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle  # noqa: E402
from oracle import Geom  # noqa: E402
from golden_cases import rel_err  # noqa: E402
from fp64_ref import ref_outp  # noqa: E402

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


@pytest.fixture
def wide(hip):
    """wgw_kernel selected (include/convnet_hip.h: wgrad tile 1 — the default; set explicitly so that the file does not depend on it)"""
    from convnet_amd import _lib
    _lib.lib.convnet_hip_set_wgrad_tile(1)
    yield
    _lib.lib.convnet_hip_set_wgrad_tile(1)


def last_kernel_timer_names():
    """names the HIP-event profiler saw since it was switched on (the kernels that really ran)"""
    from convnet_amd import _lib
    return [r["kernel"] for r in _lib.profile_report()]


def rnd(rng, shape):
    return rng.standard_normal(shape).astype(np.float32)


# K = C*Ky*Kx >= 256 and F >= 192 and N % 32 == 0, or the launch stays on wg_kernel
CASES = [
    Geom(N=32, C=32, H=9, W=9, F=192, Ky=3, Kx=3, pady=1, padx=1),             # one k tile (288 rows, partial second), 192-filter tile, border taps
    Geom(N=64, C=48, H=7, W=10, F=256, Ky=3, Kx=3, pady=1, padx=1),            # 256-filter tile, rectangular image, two image chunks per pixel
    Geom(N=32, C=64, H=11, W=11, F=200, Ky=3, Kx=3),                           # ragged filter tile (200 of 256), no padding, 81 chunks
    Geom(N=96, C=16, H=12, W=12, F=224, Ky=5, Kx=5, sy=2, sx=2, pady=2, padx=2),  # stride 2, 5 x 5 taps, three chunks per pixel
    Geom(N=32, C=29, H=8, W=8, F=192, Ky=3, Kx=3, pady=1, padx=1),             # K = 261: five rows in the second k tile, spare row for the bias
    Geom(N=32, C=32, H=9, W=9, F=198, Ky=3, Kx=3, pady=1, padx=1),             # F % 4 != 0: the direct 4-byte write-out
    Geom(N=256, C=256, H=13, W=13, F=384, Ky=3, Kx=3, pady=1, padx=1),         # conv3 itself: 9 x 2 tiles of 256 x 192
    Geom(N=256, C=384, H=13, W=13, F=256, Ky=3, Kx=3),                         # conv5 itself: 14 tiles of 256 x 256
]
_id = lambda g: f"N{g.N}C{g.C}H{g.H}W{g.W}F{g.F}k{g.Ky}s{g.sy}p{g.pady}"  # noqa: E731


@pytest.mark.parametrize("g", CASES, ids=_id)
def test_wide_wgrad_vs_oracle(hip, wide, g):
    from convnet_amd import _lib
    rng = np.random.default_rng(41)
    x, dy = rnd(rng, g.in_shape()), rnd(rng, g.out_shape())
    for st, so in (((0.0, 1.0),) if g.N * g.C * g.F > 10 ** 7 else ((0.0, 1.0), (1.0, 0.5))):   # (full-size layers: one pass of the CPU oracle)
        t0 = rnd(rng, g.filt_shape())
        _lib.profile_enable(True)
        got = hip.conv_outp(g, x, dy, t0.copy(), st, so)
        names = last_kernel_timer_names()
        _lib.profile_enable(False)
        assert any(n.startswith("wgw_kernel") for n in names), names
        assert rel_err(got, oracle.port.conv_outp(g, x, dy, t0.copy(), st, so)) < TOL


@pytest.mark.parametrize("g", CASES[:6], ids=_id)
def test_wide_wgrad_with_bias_row(hip, wide, g):
    from hip_adapter import conv_outp_bias
    rng = np.random.default_rng(42)
    x, dy = rnd(rng, g.in_shape()), rnd(rng, g.out_shape())
    dw0, db0 = rnd(rng, g.filt_shape()), rnd(rng, (g.F,))
    dw, db = conv_outp_bias(g, x, dy, dw0.copy(), db0.copy(), 1.0, 0.25)
    assert rel_err(dw, oracle.port.conv_outp(g, x, dy, dw0.copy(), 1.0, 0.25)) < TOL
    ref_db = db0 + 0.25 * dy.reshape(g.F, -1).astype(np.float64).sum(axis=1)
    assert rel_err(db, ref_db.astype(np.float32)) < TOL


def test_fc_wgrad_stays_on_wg_kernel(hip, wide):
    """dot TN (fc_edge.cc:74) has 8 chunks of reduction per tile at 256 images: the wide tile does not take it"""
    from convnet_amd import _lib
    rng = np.random.default_rng(43)
    N, D, F = 64, 512, 320
    x, dy = rnd(rng, (D, N)), rnd(rng, (F, N))
    t0 = rnd(rng, (D, F))
    _lib.profile_enable(True)
    got = hip.dot(dy, x, t0.copy(), 1.0, 1.0 / N, True, False)
    names = last_kernel_timer_names()
    _lib.profile_enable(False)
    assert any(n.startswith("wg_kernel") for n in names) and not any(n.startswith("wgw_kernel") for n in names), names
    assert rel_err(got, oracle.port.dot(dy, x, t0.copy(), 1.0, 1.0 / N, True, False)) < TOL


def test_wide_agrees_with_wg_kernel(hip):
    from convnet_amd import _lib
    g = Geom(N=64, C=64, H=13, W=13, F=256, Ky=3, Kx=3, pady=1, padx=1)
    rng = np.random.default_rng(44)
    x, dy = rnd(rng, g.in_shape()), rnd(rng, g.out_shape())
    outs = []
    for mode in (0, 1):
        _lib.lib.convnet_hip_set_wgrad_tile(mode)
        outs.append(hip.conv_outp(g, x, dy))
    _lib.lib.convnet_hip_set_wgrad_tile(1)
    assert rel_err(outs[1], outs[0]) < 1e-5   # same operand splits and products; the split-K partition differs


def test_wide_single_block_epilogue_on_hardware(hip, wide):
    """>= 256 tiles: one block per tile, no slabs — scaleTargets / scaleOutput and the bias row in the kernel's OWN write-out (the AlexNet
    layers never get there: their 10-28 tiles are cut in K; the emulation covered this path, tests/test_emulated_kernels.py).  A layer
    that size is beyond a whole-tensor pass of the CPU oracle, so the check is a float64 evaluation of the reference's definition
    (cudamat_conv_gemm.cu:827-960) at sampled weights — every k-tile and filter-tile corner, the spare bias row's neighbours and random
    ones — plus the whole bias gradient in float64; wg_kernel runs beside it on the same data only to show that the two partitions of
    the reduction agree to rounding."""
    from convnet_amd import _lib
    from hip_adapter import conv_outp_bias
    g = Geom(N=32, C=500, H=8, W=8, F=4096, Ky=3, Kx=3, pady=1, padx=1)   # K = 4500: 18 k-tiles (a spare row for the bias) x 16 filter tiles = 288
    rng = np.random.default_rng(45)
    x, dy = rnd(rng, g.in_shape()), rnd(rng, g.out_shape())
    dw0, db0 = rnd(rng, g.filt_shape()), rnd(rng, (g.F,))
    outs = []
    for mode in (0, 1):
        _lib.lib.convnet_hip_set_wgrad_tile(mode)
        _lib.profile_enable(True)
        dw, db = conv_outp_bias(g, x, dy, dw0.copy(), db0.copy(), 1.0, 0.5)
        names = last_kernel_timer_names()
        _lib.profile_enable(False)
        assert any(n.startswith("wgw_kernel" if mode else "wg_kernel") for n in names), names
        if mode:
            assert not any("reduce" in n for n in names), names   # one block per tile: nothing to reduce
        outs.append((dw, db))
    _lib.lib.convnet_hip_set_wgrad_tile(1)
    dw, db = outs[1]
    scale = float(np.abs(dw - dw0).mean())
    picks = [(c, ky, kx, f) for c in (0, 28, 56, 499) for (ky, kx) in ((0, 0), (1, 1), (2, 2)) for f in (0, 255, 256, 4095)]   # tile corners: k = 9c + 3ky + kx
    picks += [(int(rng.integers(g.C)), int(rng.integers(3)), int(rng.integers(3)), int(rng.integers(g.F))) for _ in range(48)]
    for (c, ky, kx, f) in picks:
        want = float(dw0[c, ky, kx, f]) + 0.5 * ref_outp(g, x, dy, c, ky, kx, f)
        assert abs(want - dw[c, ky, kx, f]) < TOL * scale, ("wgrad", c, ky, kx, f, want, dw[c, ky, kx, f])
    ref_db = db0 + 0.5 * dy.reshape(g.F, -1).astype(np.float64).sum(axis=1)
    assert rel_err(db, ref_db.astype(np.float32)) < TOL
    assert rel_err(dw, outs[0][0]) < 1e-5 and rel_err(db, outs[0][1]) < 1e-5
