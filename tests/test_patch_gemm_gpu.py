"""gpp_kernel / gpw_kernel (convnet_amd/csrc/patch_gemm.hip) — the patch-resident gather-GEMM on pre-split source planes that runs conv fprop /
dgrad of the 3x3 and 5x5 layers on the default (bf16-split) matrix path — against the CPU oracle (the reference's conv_up /
conv_down, cudamat_conv_gemm.cu:545-825) on geometries chosen for ITS mechanisms: tiles that wrap from one image row to the next,
from one 64-image block to the next, the ragged last tile, tap groups of a stride-2 row, partial row tiles, border tap rows.
Every case asserts that the patch kernel is what ran (convnet_hip_last_kernel_info), so a silent fallback cannot pass.
Tolerance: the reference's own kernel-test metric, max|a-b| / mean|a+b| < 1e-4 (py/test_conv.py:382-392)."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle  # noqa: E402
from oracle import Geom  # noqa: E402
from golden_cases import rel_err  # noqa: E402
from fp64_ref import ref_up  # noqa: E402

TOL = 1e-4
DEFAULT_MODE = 3   # include/convnet_hip.h: gpw_kernel where its launch policy takes the launch, ggp_kernel elsewhere


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


@pytest.fixture(params=["raw", "planes"])
def patch_mode(request, hip):
    """Both builds of gpp_kernel (include/convnet_hip.h: convnet_hip_set_patch_mode): the source slab staged as raw fp32 and split
    by the consumers (default), or read from bf16 planes written by act_planes_kernel."""
    from convnet_amd import _lib
    _lib.lib.convnet_hip_set_patch_mode(1 if request.param == "raw" else 2)
    yield request.param
    _lib.lib.convnet_hip_set_patch_mode(DEFAULT_MODE)


def last_kernel():
    from convnet_amd import _lib
    info = _lib.KernelInfo()
    _lib.lib.convnet_hip_last_kernel_info(ctypes.byref(info))
    return info.name.decode()


def rnd(rng, shape):
    return rng.standard_normal(shape).astype(np.float32)


# fprop needs C % 16 == 0 and F > 64; dgrad needs F % 16 == 0 and C > 64; both N % 64 == 0 and an output grid >= 4 wide
FPROP = [
    Geom(N=64, C=32, H=9, W=9, F=96, Ky=3, Kx=3, pady=1, padx=1),            # 9-wide rows: every third tile wraps
    Geom(N=128, C=80, H=13, W=13, F=144, Ky=3, Kx=3, pady=1, padx=1),         # conv3/4 grid, two image blocks, partial row tile
    Geom(N=64, C=96, H=6, W=6, F=128, Ky=3, Kx=3),                            # 4-wide output grid, no padding
    Geom(N=64, C=80, H=15, W=15, F=96, Ky=5, Kx=5, sy=2, sx=2),               # conv2 type: tap groups {0,2,4} and {1,3} of a stride-2 row
    Geom(N=64, C=16, H=12, W=12, F=80, Ky=4, Kx=4, sy=2, sx=2, pady=1, padx=1),  # two groups of two
    Geom(N=64, C=48, H=7, W=10, F=72, Ky=3, Kx=3, pady=1, padx=1),            # rectangular image
    Geom(N=64, C=32, H=8, W=8, F=100, Ky=1, Kx=3, padx=1),                    # one tap row
    Geom(N=64, C=32, H=8, W=8, F=100, Ky=2, Kx=2),                            # group of two taps
    Geom(N=192, C=16, H=5, W=5, F=72, Ky=3, Kx=3, pady=1, padx=1),            # 75 units: ragged last tile, three image blocks
    Geom(N=64, C=32, H=11, W=11, F=96, Ky=3, Kx=3, pady=2, padx=2),           # padding wider than the reach of one tap
]
DGRAD = [
    Geom(N=64, C=96, H=9, W=9, F=32, Ky=3, Kx=3, pady=1, padx=1),
    Geom(N=128, C=144, H=13, W=13, F=80, Ky=3, Kx=3, pady=1, padx=1),
    Geom(N=64, C=128, H=13, W=13, F=48, Ky=3, Kx=3),                          # conv5 type: pad 0, 11 x 11 derivatives into 13 x 13
    Geom(N=64, C=72, H=7, W=10, F=48, Ky=3, Kx=3, pady=1, padx=1),
    Geom(N=64, C=100, H=8, W=8, F=32, Ky=2, Kx=2),
    Geom(N=192, C=72, H=5, W=5, F=16, Ky=3, Kx=3, pady=1, padx=1),
]
_id = lambda g: f"N{g.N}C{g.C}H{g.H}W{g.W}F{g.F}k{g.Ky}x{g.Kx}s{g.sy}p{g.pady}"  # noqa: E731


@pytest.mark.parametrize("g", FPROP, ids=_id)
def test_patch_fprop_vs_oracle(hip, patch_mode, g):
    rng = np.random.default_rng(21)
    x, w = rnd(rng, g.in_shape()), rnd(rng, g.filt_shape())
    for st in (0.0, 1.0):
        t0 = rnd(rng, g.out_shape())
        got = hip.conv_up(g, x, w, t0.copy(), st)
        assert last_kernel() == "gpp_kernel(fprop)", last_kernel()
        assert rel_err(got, oracle.port.conv_up(g, x, w, t0.copy(), st)) < TOL


@pytest.mark.parametrize("g", DGRAD, ids=_id)
def test_patch_dgrad_vs_oracle(hip, patch_mode, g):
    rng = np.random.default_rng(22)
    dy, w = rnd(rng, g.out_shape()), rnd(rng, g.filt_shape())
    for st in (0.0, 1.0):
        t0 = rnd(rng, g.in_shape())
        got = hip.conv_down(g, dy, w, t0.copy(), st)
        assert last_kernel() == "gpp_kernel(dgrad)", last_kernel()
        assert rel_err(got, oracle.port.conv_down(g, dy, w, t0.copy(), st)) < TOL


def test_patch_fused_bias_relu(hip, patch_mode):
    g = Geom(N=64, C=32, H=9, W=9, F=96, Ky=3, Kx=3, pady=1, padx=1)
    rng = np.random.default_rng(23)
    x, w, b = rnd(rng, g.in_shape()), rnd(rng, g.filt_shape()), rnd(rng, (g.F,))
    fused = hip.conv_up_bias_relu(g, x, w, b, relu=True)
    assert last_kernel() == "gpp_kernel(fprop)"
    y = oracle.port.conv_up(g, x, w)
    y = oracle.port.add_row_vec(y.reshape(g.F, -1), b).reshape(g.out_shape())
    assert rel_err(fused, oracle.port.lower_bound(y, 0.0)) < TOL


def test_patch_modes_agree_with_ggp_kernel(hip):
    """Same exact operand splits, same six products, fp32 accumulation in a different order (taps innermost per tap row): the three
    kernels agree to accumulation rounding, far inside the oracle tolerance."""
    from convnet_amd import _lib
    g = Geom(N=64, C=64, H=13, W=13, F=128, Ky=3, Kx=3, pady=1, padx=1)
    rng = np.random.default_rng(24)
    x, w = rnd(rng, g.in_shape()), rnd(rng, g.filt_shape())
    outs = []
    for mode in (0, 1, 2):
        _lib.lib.convnet_hip_set_patch_mode(mode)
        outs.append(hip.conv_up(g, x, w))
        assert last_kernel() == ("gg_kernel(fprop)" if mode == 0 else "gpp_kernel(fprop)")
    _lib.lib.convnet_hip_set_patch_mode(DEFAULT_MODE)
    assert rel_err(outs[1], outs[0]) < 1e-5 and rel_err(outs[2], outs[0]) < 1e-5
    assert np.array_equal(outs[1], outs[2])   # raw and planes builds: identical operands, identical order


# ---- gpw_kernel (patch modes 3 / 4: 128 rows x 8 units per block, four waves staging for themselves; 3 x 3 stride-1 gathers with
# output rows >= 8 wide).  Mode 3 — the library default — adds a launch policy (patch_gemm.hip: wide_plan: the blocks must fill their
# rounds on the chip, else ggp_kernel); these cases are about the kernel's mechanisms at sizes the policy would leave alone, so they
# run in mode 4 (no policy); the policy itself: test_wide_launch_policy.  First run on the MI355X in round 5
# (profiles/r05_wide_kernels.md: the write-out of the 256 accumulator registers needed an explicit element-wise AGPR read).
WIDE_FPROP = [
    Geom(N=64, C=32, H=9, W=9, F=96, Ky=3, Kx=3, pady=1, padx=1),             # a wrap in almost every tile
    Geom(N=128, C=80, H=13, W=13, F=144, Ky=3, Kx=3, pady=1, padx=1),          # conv3/4 grid, two image blocks, partial row tile, ragged last tile
    Geom(N=64, C=16, H=10, W=10, F=72, Ky=3, Kx=3),                            # pad 0, 8-wide output rows
    Geom(N=64, C=48, H=7, W=12, F=72, Ky=3, Kx=3, pady=1, padx=1),             # rectangular
    Geom(N=64, C=32, H=11, W=11, F=96, Ky=3, Kx=3, pady=2, padx=2),            # whole tap rows outside the image
    Geom(N=64, C=32, H=8, W=8, F=100, Ky=1, Kx=3, padx=1),                     # one tap row
    Geom(N=256, C=384, H=13, W=13, F=384, Ky=3, Kx=3, pady=1, padx=1),         # conv4 itself: 254 tiles, 216 chunks each
    Geom(N=256, C=192, H=13, W=13, F=256, Ky=3, Kx=3, pady=1, padx=1),         # 170 tiles: split-K
]
WIDE_DGRAD = [
    Geom(N=64, C=96, H=9, W=9, F=32, Ky=3, Kx=3, pady=1, padx=1),
    Geom(N=128, C=144, H=13, W=13, F=80, Ky=3, Kx=3, pady=1, padx=1),
    Geom(N=64, C=128, H=13, W=13, F=48, Ky=3, Kx=3),                           # conv5 type: pad 0, 11 x 11 derivatives into 13 x 13
    Geom(N=256, C=384, H=13, W=13, F=256, Ky=3, Kx=3, pady=1, padx=1),         # conv5's dgrad at full size
]


@pytest.fixture
def wide_mode(hip):
    """gpw_kernel wherever the shape allows (include/convnet_hip.h: patch mode 4)"""
    from convnet_amd import _lib
    _lib.lib.convnet_hip_set_patch_mode(4)
    yield
    _lib.lib.convnet_hip_set_patch_mode(DEFAULT_MODE)


@pytest.mark.parametrize("g", WIDE_FPROP, ids=_id)
def test_wide_fprop_vs_oracle(hip, wide_mode, g):
    rng = np.random.default_rng(31)
    x, w = rnd(rng, g.in_shape()), rnd(rng, g.filt_shape())
    for st in ((0.0,) if g.N * g.C * g.F > 10 ** 7 else (0.0, 1.0)):   # (the full-size layers: one pass of the CPU oracle)
        t0 = rnd(rng, g.out_shape())
        got = hip.conv_up(g, x, w, t0.copy(), st)
        assert last_kernel() == "gpw_kernel(fprop)", last_kernel()
        assert rel_err(got, oracle.port.conv_up(g, x, w, t0.copy(), st)) < TOL


@pytest.mark.parametrize("g", WIDE_DGRAD, ids=_id)
def test_wide_dgrad_vs_oracle(hip, wide_mode, g):
    rng = np.random.default_rng(32)
    dy, w = rnd(rng, g.out_shape()), rnd(rng, g.filt_shape())
    for st in ((0.0,) if g.N * g.C * g.F > 10 ** 7 else (0.0, 1.0)):
        t0 = rnd(rng, g.in_shape())
        got = hip.conv_down(g, dy, w, t0.copy(), st)
        assert last_kernel() == "gpw_kernel(dgrad)", last_kernel()
        assert rel_err(got, oracle.port.conv_down(g, dy, w, t0.copy(), st)) < TOL


def test_wide_agrees_with_gpp_raw(hip):
    """Same operand splits, same six products, same order along k (cb, tap row, tap) — up to the split-K partition, which each kernel
    picks for its own tile count: agreement to accumulation rounding."""
    from convnet_amd import _lib
    g = Geom(N=64, C=64, H=13, W=13, F=128, Ky=3, Kx=3, pady=1, padx=1)
    rng = np.random.default_rng(33)
    x, w = rnd(rng, g.in_shape()), rnd(rng, g.filt_shape())
    outs = []
    for mode in (1, 4):
        _lib.lib.convnet_hip_set_patch_mode(mode)
        outs.append(hip.conv_up(g, x, w))
        assert last_kernel() == ("gpp_kernel(fprop)" if mode == 1 else "gpw_kernel(fprop)")
    _lib.lib.convnet_hip_set_patch_mode(DEFAULT_MODE)
    assert rel_err(outs[1], outs[0]) < 1e-5


def test_wide_launch_policy(hip):
    """Mode 3, the default: gpw_kernel takes the launches that run in ONE round filling the chip (conv4 at 256 images: 255 tiles on 256
    CUs; conv5 fprop: 122 tiles in two K-ranges) and — since round 6 — the K-split ones from 0.6 fill (conv3 at 64 images: 66 tiles in
    three K-ranges, measured 7-12 % ahead), leaves the others to ggp_kernel (conv3 dgrad at 256 images: 170 whole-K tiles measured 401 us
    against 366 with ggp_kernel's tail split; conv4 at 128 images: 129 tiles = half the chip; small problems) — and both give the oracle's result."""
    from convnet_amd import _lib
    assert _lib.lib.convnet_hip_get_patch_mode() == DEFAULT_MODE
    rng = np.random.default_rng(34)
    takes = [
        (Geom(N=256, C=384, H=13, W=13, F=256, Ky=3, Kx=3), "fprop", True),                   # conv5 fprop: 122 tiles x 2 K-ranges
        (Geom(N=256, C=256, H=13, W=13, F=384, Ky=3, Kx=3, pady=1, padx=1), "dgrad", False),  # conv3 dgrad: 170 tiles
        (Geom(N=64, C=256, H=13, W=13, F=384, Ky=3, Kx=3, pady=1, padx=1), "fprop", True),    # conv3 at 64 images: 66 tiles x 3 K-ranges
        (Geom(N=128, C=384, H=13, W=13, F=384, Ky=3, Kx=3, pady=1, padx=1), "fprop", False),  # conv4 at 128 images: 129 tiles
        (Geom(N=64, C=32, H=9, W=9, F=96, Ky=3, Kx=3, pady=1, padx=1), "fprop", False),       # 11 tiles
    ]
    for g, which, wide_expected in takes:
        w = rnd(rng, g.filt_shape())
        if which == "fprop":
            x = rnd(rng, g.in_shape())
            got, ref = hip.conv_up(g, x, w), None
            assert last_kernel() == ("gpw_kernel(fprop)" if wide_expected else "gg_kernel(fprop)"), (g, last_kernel())
            if g.N * g.C * g.F < 10 ** 7:
                ref = oracle.port.conv_up(g, x, w)
        else:
            dy = rnd(rng, g.out_shape())
            got, ref = hip.conv_down(g, dy, w), None
            assert last_kernel() == ("gpw_kernel(dgrad)" if wide_expected else "gg_kernel(dgrad)"), (g, last_kernel())
        if ref is not None:
            assert rel_err(got, ref) < TOL


def test_wide_tail_split_on_hardware(hip):
    """More tiles than CUs and a partial last round: gpw_kernel cuts only the last round's tiles in K and gpw_tail_fix_kernel sums
    their partial tiles (VGG-size layers take this path; the emulation covered it, tests/test_emulated_kernels.py).  1 058 tiles on 256
    CUs: four whole rounds and 34 tail tiles.  The layer is 80 GFLOP — beyond a whole-tensor pass of the CPU oracle — so the check is a
    float64 evaluation of the reference's definition (cudamat_conv_gemm.cu:545-640) at sampled outputs: the image corners and borders,
    the LAST tiles of the launch (the K-cut ones: the highest filter block's last pixels) and random ones; the default gather kernel
    runs beside it on the same data only to show that the two K partitions agree to rounding."""
    from convnet_amd import _lib
    g = Geom(N=64, C=64, H=92, W=92, F=128, Ky=3, Kx=3, pady=1, padx=1)
    rng = np.random.default_rng(35)
    x, w = rnd(rng, g.in_shape()), rnd(rng, g.filt_shape())
    outs = []
    for mode in (0, 3):
        _lib.lib.convnet_hip_set_patch_mode(mode)
        _lib.profile_enable(True)
        outs.append(hip.conv_up(g, x, w))
        names = [r["kernel"] for r in _lib.profile_report()]
        _lib.profile_enable(False)
        if mode == 3:   # (the launch policy takes it: one K-range, many rounds)
            assert last_kernel() == "gpw_kernel(fprop)" and any(n.startswith("gpw_kernel") for n in names), (last_kernel(), names)
            assert any(n == "gg_tail_fix_kernel" for n in names), names
    _lib.lib.convnet_hip_set_patch_mode(DEFAULT_MODE)
    y = outs[1]
    scale = float(np.abs(y).mean())
    picks = [(f, oy, ox, n) for f in (0, 127) for (oy, ox) in ((0, 0), (0, 91), (91, 0), (91, 91), (90, 84), (91, 88), (45, 46)) for n in (0, 63)]
    picks += [(int(rng.integers(g.F)), int(rng.integers(80, g.My)), int(rng.integers(g.Mx)), int(rng.integers(g.N))) for _ in range(32)]   # the last rows: the tail tiles
    picks += [(int(rng.integers(g.F)), int(rng.integers(g.My)), int(rng.integers(g.Mx)), int(rng.integers(g.N))) for _ in range(32)]
    for (f, oy, ox, n) in picks:
        want = ref_up(g, x, w, f, oy, ox, n)
        assert abs(want - y[f, oy, ox, n]) < TOL * scale, ("fprop", f, oy, ox, n, want, y[f, oy, ox, n])
    assert rel_err(outs[1], outs[0]) < 1e-5


# ---- gpv_kernel (round 6; patch modes 3 / 4): gpw_kernel's tile for tap rows cut into groups of three and two taps — 5 x 5 stride 2
# forward ({0,2,4} / {1,3}: AlexNet's conv2), the stride classes of its input gradient (3- and 2-tap rows, all classes in one launch),
# and a 96-row build for 65..96-row problems.  Mechanisms: superchunks of three and two chunks, two slots per wave in the first chunk of
# a two-chunk superchunk, K-ranges that begin inside a tap row's groups, the quarter piece of the 96-row filter chunk.  Against the
# CPU oracle; every case asserts from the kernel timers that gpv_kernel is what ran.
VAR_FPROP = [
    Geom(N=64, C=32, H=23, W=23, F=96, Ky=5, Kx=5, sy=2, sx=2),                  # conv2's form on the 96-row build, 10-wide rows, split-K
    Geom(N=128, C=80, H=21, W=21, F=144, Ky=5, Kx=5, sy=2, sx=2),                # 128-row build, partial row tile, 9-wide rows (wraps), two image blocks
    Geom(N=64, C=16, H=21, W=25, F=96, Ky=5, Kx=5, sy=2, sx=2, pady=2, padx=2),  # padded: border columns from the zero page, tap rows skipped
    Geom(N=64, C=16, H=20, W=20, F=128, Ky=4, Kx=4, sy=2, sx=2, pady=1, padx=1), # groups of two and two
    Geom(N=64, C=32, H=10, W=10, F=128, Ky=2, Kx=2),                             # one group of two
    Geom(N=64, C=32, H=9, W=9, F=72, Ky=3, Kx=3, pady=1, padx=1),                # 3 x 3 stride 1 on the 96-row build
    Geom(N=192, C=16, H=19, W=19, F=80, Ky=5, Kx=5, sy=2, sx=2),                 # 8-wide rows, three image blocks
    Geom(N=256, C=96, H=55, W=55, F=256, Ky=5, Kx=5, sy=2, sx=2),                # conv2 itself: 676 tiles, tail split
]
VAR_DGRAD = [
    Geom(N=64, C=96, H=19, W=19, F=32, Ky=5, Kx=5, sy=2, sx=2),                  # classes 3x3, 3x2, 2x3, 2x2 on the 96-row build
    Geom(N=64, C=132, H=17, W=21, F=32, Ky=5, Kx=5, sy=2, sx=2, pady=2, padx=2), # 128-row build, two row tiles, classes start at different pixels
    Geom(N=128, C=96, H=23, W=23, F=48, Ky=5, Kx=5, sy=2, sx=2),                 # two image blocks
    Geom(N=64, C=96, H=20, W=20, F=16, Ky=4, Kx=4, sy=2, sx=2, pady=1, padx=1),  # four classes of 2 x 2 taps
    Geom(N=256, C=96, H=55, W=55, F=256, Ky=5, Kx=5, sy=2, sx=2),                # conv2's input gradient itself
]


def _ran(hip_call):
    """run hip_call with the kernel timers on; returns (result, names of the kernels that ran)"""
    from convnet_amd import _lib
    _lib.profile_enable(True)
    out = hip_call()
    names = [r["kernel"] for r in _lib.profile_report()]
    _lib.profile_enable(False)
    return out, names


@pytest.mark.parametrize("g", VAR_FPROP, ids=_id)
def test_group_tile_fprop_vs_oracle(hip, wide_mode, g):
    rng = np.random.default_rng(41)
    x, w = rnd(rng, g.in_shape()), rnd(rng, g.filt_shape())
    for st in ((0.0,) if g.N * g.C * g.F > 10 ** 6 else (0.0, 1.0)):
        t0 = rnd(rng, g.out_shape())
        got, names = _ran(lambda: hip.conv_up(g, x, w, t0.copy(), st))
        assert any(n.startswith("gpv_kernel") for n in names), names
        assert rel_err(got, oracle.port.conv_up(g, x, w, t0.copy(), st)) < TOL


@pytest.mark.parametrize("g", VAR_DGRAD, ids=_id)
def test_group_tile_dgrad_vs_oracle(hip, wide_mode, g):
    """(mode 4: in the default mode a strided input gradient stays on ggp_kernel, which measured faster — patch_classes_ok)"""
    rng = np.random.default_rng(42)
    dy, w = rnd(rng, g.out_shape()), rnd(rng, g.filt_shape())
    for st in ((0.0,) if g.N * g.C * g.F > 10 ** 6 else (0.0, 1.0)):
        t0 = rnd(rng, g.in_shape())
        got, names = _ran(lambda: hip.conv_down(g, dy, w, t0.copy(), st))
        assert any(n.startswith("gpv_kernel") for n in names), names
        assert rel_err(got, oracle.port.conv_down(g, dy, w, t0.copy(), st)) < TOL


def test_group_tile_is_the_default_for_conv2(hip):
    """mode 3 with its launch policy: AlexNet's conv2 at 256 images runs forward on gpv_kernel (676 tiles = two whole rounds + a tail
    split), and the fused bias + ReLU epilogue equals the unfused sequence bit for bit"""
    from convnet_amd import _lib
    assert _lib.lib.convnet_hip_get_patch_mode() == DEFAULT_MODE
    g = Geom(N=256, C=96, H=55, W=55, F=256, Ky=5, Kx=5, sy=2, sx=2)
    rng = np.random.default_rng(43)
    x, w, b = rnd(rng, g.in_shape()), rnd(rng, g.filt_shape()), rnd(rng, (g.F,))
    y, names = _ran(lambda: hip.conv_up(g, x, w))
    assert any(n.startswith("gpv_kernel<128x512") for n in names) and "gg_tail_fix_kernel" in names, names
    fused = hip.conv_up_bias_relu(g, x, w, b, relu=True)
    unfused = np.maximum(y + b.reshape(-1, 1, 1, 1), 0.0).astype(np.float32)
    assert np.array_equal(fused, unfused)
    dy = rnd(rng, g.out_shape())
    _, names = _ran(lambda: hip.conv_down(g, dy, w))
    assert any(n.startswith("ggp_kernel<1,4,3,64") for n in names), names   # (the input gradient: measured faster on the one-pixel tiles)
