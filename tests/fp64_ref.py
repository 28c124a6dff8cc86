"""float64 evaluations of the reference's conv definitions at single output elements (layouts of cudamat_conv_gemm.cuh:5-10, sizes of
src/edge.cc:108-114; conv_up / conv_down / conv_outp = cudamat_conv_gemm.cu:545-640 / 650-790 / 827-960) — the checker for layers
too large for a whole-tensor pass of the CPU oracle.  numpy only; test infrastructure."""
import numpy as np


def ref_up(g, x, w, f, oy, ox, n):
    acc = 0.0
    for ky in range(g.Ky):
        for kx in range(g.Kx):
            iy, ix = oy * g.sy + ky - g.pady, ox * g.sx + kx - g.padx
            if 0 <= iy < g.H and 0 <= ix < g.W:
                acc += float(np.dot(x[:, iy, ix, n].astype(np.float64), w[:, ky, kx, f].astype(np.float64)))
    return acc


def ref_down(g, dy, w, c, iy, ix, n):
    acc = 0.0
    for ky in range(g.Ky):
        for kx in range(g.Kx):
            ty, tx = iy + g.pady - ky, ix + g.padx - kx
            if ty % g.sy or tx % g.sx:
                continue
            oy, ox = ty // g.sy, tx // g.sx
            if 0 <= oy < g.My and 0 <= ox < g.Mx:
                acc += float(np.dot(dy[:, oy, ox, n].astype(np.float64), w[c, ky, kx, :].astype(np.float64)))
    return acc


def ref_outp(g, x, dy, c, ky, kx, f):
    # all output locations whose tap (ky,kx) falls inside the image
    oy = np.arange(g.My)
    ox = np.arange(g.Mx)
    iy, ix = oy * g.sy + ky - g.pady, ox * g.sx + kx - g.padx
    my, mx = (iy >= 0) & (iy < g.H), (ix >= 0) & (ix < g.W)
    xs = x[c][np.ix_(iy[my], ix[mx])].astype(np.float64)        # (oy, ox, N)
    ds = dy[f][np.ix_(oy[my], ox[mx])].astype(np.float64)
    return float((xs * ds).sum())
