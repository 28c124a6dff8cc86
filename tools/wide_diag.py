"""Where does gpw_kernel (patch mode 4: without its launch policy) differ from the default gather kernel?  Runs each geometry on both and prints the
structure of the mismatch (which rows, pixels, images).  GPU only; no oracle involved (the default kernel is the reference here,
it is parity-green against the oracle).  python tools/wide_diag.py [fprop|dgrad] [mode ...]"""
import ctypes
import os
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
from oracle import Geom  # noqa: E402  (geometry record only)
from hip_adapter import HipImpl  # noqa: E402
from convnet_amd import _lib  # noqa: E402
from convnet_amd.matrix import Matrix  # noqa: E402

Matrix.SetupCUDADevice(0)
Matrix.InitRandom(42)
_lib.lib.convnet_hip_set_matrix_path(1)
hip = HipImpl()

CASES = [
    Geom(N=64, C=32, H=9, W=9, F=96, Ky=3, Kx=3, pady=1, padx=1),
    Geom(N=128, C=80, H=13, W=13, F=144, Ky=3, Kx=3, pady=1, padx=1),
    Geom(N=64, C=80, H=13, W=13, F=144, Ky=3, Kx=3, pady=1, padx=1),     # one image block
    Geom(N=128, C=80, H=13, W=13, F=128, Ky=3, Kx=3, pady=1, padx=1),    # whole row tile
    Geom(N=128, C=32, H=13, W=13, F=144, Ky=3, Kx=3, pady=1, padx=1),    # two channel blocks
    Geom(N=128, C=16, H=13, W=13, F=128, Ky=3, Kx=3, pady=1, padx=1),    # one channel block: no split-K
    Geom(N=128, C=80, H=9, W=9, F=144, Ky=3, Kx=3, pady=1, padx=1),
    Geom(N=64, C=16, H=10, W=10, F=72, Ky=3, Kx=3),
    Geom(N=64, C=32, H=16, W=16, F=128, Ky=3, Kx=3, pady=1, padx=1),     # rows of 16: tiles never wrap mid-row
    Geom(N=256, C=384, H=13, W=13, F=384, Ky=3, Kx=3, pady=1, padx=1),
]


def describe(name, got, ref):
    err = np.abs(got - ref)
    scale = np.abs(ref).mean()
    bad = err > 1e-4 * scale * 10
    print(f"  {name}: max err / mean {err.max() / scale:.3g}, bad {bad.sum()} of {bad.size}, nan {np.isnan(got).sum()}")
    if not bad.any():
        return
    F, My, Mx, N = got.shape
    f_bad = np.where(bad.any(axis=(1, 2, 3)))[0]
    px_bad = np.argwhere(bad.any(axis=(0, 3)))
    n_bad = np.where(bad.any(axis=(0, 1, 2)))[0]

    def runs(a):
        a = list(a)
        out, s = [], None
        for i, v in enumerate(a):
            if s is None:
                s = p = v
            elif v == p + 1:
                p = v
            else:
                out.append((s, p)); s = p = v
        if s is not None:
            out.append((s, p))
        return out[:12]
    print(f"    rows {runs(f_bad)} ({len(f_bad)} of {F})")
    print(f"    images {runs(n_bad)} ({len(n_bad)} of {N})")
    print(f"    pixels ({len(px_bad)} of {My * Mx}): flat {runs(sorted(int(y * Mx + x) for y, x in px_bad))}")
    # (image block, pixel) units -> tile index of 8 units
    units = sorted({(int(n // 64) * My * Mx + int(y * Mx + x)) for y, x in px_bad for n in n_bad[::16]})
    print(f"    tiles (unit // 8) {runs(sorted({u // 8 for u in units}))}; unit % 8 {sorted({u % 8 for u in units})}")
    i = np.unravel_index(np.argmax(err), err.shape)
    print(f"    worst at f={i[0]} y={i[1]} x={i[2]} n={i[3]}: got {got[i]:.5g} ref {ref[i]:.5g}")


which = sys.argv[1] if len(sys.argv) > 1 else "fprop"
modes = [int(a) for a in sys.argv[2:]] or [4]
rng = np.random.default_rng(5)
for g in CASES:
    print(f"N{g.N} C{g.C} H{g.H}x{g.W} F{g.F} k{g.Ky} p{g.pady}:", flush=True)
    if which == "fprop":
        a, w = rng.standard_normal(g.in_shape()).astype(np.float32), rng.standard_normal(g.filt_shape()).astype(np.float32)
        run = lambda: hip.conv_up(g, a, w)  # noqa: E731
    else:
        a, w = rng.standard_normal(g.out_shape()).astype(np.float32), rng.standard_normal(g.filt_shape()).astype(np.float32)
        run = lambda: hip.conv_down(g, a, w)  # noqa: E731
    _lib.lib.convnet_hip_set_patch_mode(0)
    ref = run()
    for m in modes:
        _lib.lib.convnet_hip_set_patch_mode(m)
        _lib.profile_enable(True)
        got = run()
        names = [(r["kernel"], r.get("launches")) for r in _lib.profile_report()]
        _lib.profile_enable(False)
        info = _lib.KernelInfo()
        _lib.lib.convnet_hip_last_kernel_info(ctypes.byref(info))
        describe(f"mode {m} {names} grid {info.grid_blocks} split_k {info.split_k}", got, ref)
        got2 = run()
        print(f"    repeat run identical: {np.array_equal(got, got2, equal_nan=True)}")
_lib.lib.convnet_hip_set_patch_mode(0)
