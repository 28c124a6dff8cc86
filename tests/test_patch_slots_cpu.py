"""The slab / slot scheme of gpp_kernel (convnet_amd/csrc/patch_gemm.hip) restated in numpy and checked against the CPU oracle's
conv_up — host-side logic only, no GPU: units (64-image block, pixel) in flat order, tiles of four consecutive units that may wrap
to the next image row or image block, slot base S(j) (+1 inside a row, +3 across a wrap), one slab per (16-channel block, tap row,
tap group), tap groups of a stride-2 row ({0,2,4} and {1,3}), out-of-image slots = zeros, tap rows no unit of the tile has skipped,
at most 8 slots.  The -m gpu tests (tests/test_patch_gemm_gpu.py) run the kernel itself on the same mechanisms."""
import numpy as np
import pytest

import oracle
from oracle import Geom

P, NS = 4, 8   # kPatchP units per tile, slots per slab


def patch_conv(x, w, g):
    """x (C,H,W,N), w (C,Ky,Kx,F) as the oracle lays them out; returns (F,My,Mx,N)."""
    C, H, W, N = x.shape
    F = w.shape[-1] if w.ndim == 4 else g.F
    GX, G, IB = g.Mx, g.My * g.Mx, N // 64
    units = IB * G
    ng = g.sx
    gcnt = [(g.Kx - r + g.sx - 1) // g.sx for r in range(ng)]
    assert all(2 <= c <= 3 for c in gcnt)
    out = np.zeros((F, g.My, g.Mx, N), np.float64)
    max_slot = 0
    for ct in range((units + P - 1) // P):
        U = [ct * P + j for j in range(P)]
        ok = [u < units for u in U]
        ib = [u // G if o else 0 for u, o in zip(U, ok)]
        m = [u - i * G if o else 0 for u, i, o in zip(U, ib, ok)]
        oy = [mm // GX for mm in m]
        ox = [mm - o * GX for mm, o in zip(m, oy)]
        S = [0] * P
        for j in range(1, P):
            S[j] = S[j - 1] + (0 if not ok[j] else 1 if (ib[j] == ib[j - 1] and oy[j] == oy[j - 1]) else 3)
        ys0 = [o * g.sy - g.pady for o, k in zip(oy, ok) if k]
        a_lo, a_hi = max(0, -max(ys0)), min(g.Ky - 1, H - 1 - min(ys0))
        for cb in range(C // 16):
            for a in range(a_lo, a_hi + 1):
                for grp in range(ng):
                    slab = np.zeros((NS, 16, 64))
                    for s in range(NS):
                        for j in range(P):
                            i = s - S[j]
                            if ok[j] and 0 <= i < gcnt[grp]:
                                ys, xs = oy[j] * g.sy - g.pady + a, ox[j] * g.sx - g.padx + grp + i * g.sx
                                slab[s] = x[cb * 16:cb * 16 + 16, ys, xs, ib[j] * 64:ib[j] * 64 + 64] if (0 <= ys < H and 0 <= xs < W) else 0.0
                    for i in range(gcnt[grp]):
                        b = grp + i * g.sx
                        for j in range(P):
                            if ok[j]:
                                max_slot = max(max_slot, S[j] + i)
                                out[:, oy[j], ox[j], ib[j] * 64:ib[j] * 64 + 64] += w[cb * 16:cb * 16 + 16, a, b, :].T.astype(np.float64) @ slab[S[j] + i]
    assert max_slot < NS
    return out


CASES = [
    Geom(N=64, C=16, H=9, W=9, F=8, Ky=3, Kx=3, pady=1, padx=1),          # rows of 9: tiles wrap
    Geom(N=128, C=16, H=6, W=6, F=8, Ky=3, Kx=3),                          # 4-wide grid, image-block wrap, ragged last tile
    Geom(N=64, C=32, H=15, W=15, F=8, Ky=5, Kx=5, sy=2, sx=2),             # stride 2: tap groups {0,2,4}, {1,3}
    Geom(N=64, C=16, H=12, W=12, F=8, Ky=4, Kx=4, sy=2, sx=2, pady=1, padx=1),
    Geom(N=64, C=16, H=7, W=10, F=8, Ky=3, Kx=3, pady=1, padx=1),
    Geom(N=64, C=16, H=8, W=8, F=8, Ky=2, Kx=2),
    Geom(N=64, C=16, H=11, W=11, F=8, Ky=3, Kx=3, pady=2, padx=2),
]


@pytest.mark.parametrize("g", CASES, ids=lambda g: f"N{g.N}C{g.C}H{g.H}W{g.W}k{g.Ky}s{g.sy}p{g.pady}")
def test_slot_scheme_reproduces_the_convolution(g):
    rng = np.random.default_rng(3)
    x = rng.standard_normal(g.in_shape()).astype(np.float32)
    w = rng.standard_normal(g.filt_shape()).astype(np.float32)
    ref = oracle.port.conv_up(g, x, w).astype(np.float64)
    wl = w.reshape(g.C, g.Ky, g.Kx, g.F) if w.shape != (g.C, g.Ky, g.Kx, g.F) else w
    got = patch_conv(x, wl, g)
    assert np.abs(got - ref).max() < 2e-4 * np.abs(ref).max()
