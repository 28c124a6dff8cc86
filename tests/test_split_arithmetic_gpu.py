"""The bf16-split matrix path's arithmetic, pinned ON THE GPU, on the PRODUCT kernels.

The default matrix path (include/convnet_hip.h: convnet_hip_set_matrix_path(1)) forms every fp32 product on the bf16 pipe from exact
three-way operand splits, six of the nine cross terms; path 0 uses the fp32 matrix instruction.  The reference's GEMM is plain fp32
sgemm (cudamat/cudamat.cu:2130-2152 cublasSgemm; eigenmat/eigenmat.cc:2284-2298).  The 1e-4 max/mean parity tolerance cannot tell a
three-term split from a two-term one (~1e-5), so this file measures the error of BOTH paths against float64 on the exact conv2 /
conv4 / fc6 shapes at N = 256 — fprop, dgrad and wgrad kernels — and on adversarial data:

   err(path) = max over sampled outputs of |out - exact| / sum_k |a_k b_k|       (error relative to the magnitude that was summed)

for both.  What is asserted, per data family (BOUNDS below; measured table: profiles/r03_split_arithmetic.txt):
  * N(0,1) data — the training regime — and magnitudes up to the top of the range the split carries ("huge"):
    err(split) <= 1.25 * err(fp32 path), or both below one unit of 2^-24 (fc6: K = 9216 random-sign terms average out);
  * a wide dynamic range INSIDE every dot product (per-k scales 2^-20 .. 2^20): <= 2 x the fp32 path.  The bf16 instruction adds 16
    products and the accumulator in one step; with terms 2^40 apart that step loses more low bits than eight 2-term fp32 steps;
  * heavy cancellation (the terms of a dot product cancel in pairs to ~2^-12 of their magnitude, so only EXACT products survive):
    the fp32 instruction forms exact products, the split drops three of nine cross terms (<= 2^-23 of a product): the split's error
    is 2-9 x the fp32 path's there (0.05-0.12 against 0.006-0.06 units), and asserted below 0.25 x 2^-24 of sum|ab| — 1/20 of what
    either path's accumulation rounding costs on ordinary data;
  * tiny magnitudes (|x| ~ 2^-116) whose second and third split terms are fp32 / bf16 DENORMALS: the matrix pipe flushes
    denormal bf16 inputs, so those terms are lost: error up to 2^-20 of sum|ab| (asserted <= 32 x 2^-24).  Values below ~2^-110
    do not occur in a network whose activations are O(1); documented in include/convnet_hip.h.
Special values (inf, NaN, finite values above the bf16 range) are covered at the end: what matches the fp32 path and the one
documented deviation.

`tools/split_gemm.hip` is the stand-alone probe of the same arithmetic (one tile, LDS-resident); its output is in profiles/."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle  # noqa: E402
from oracle import Geom  # noqa: E402

BF16_MAX = np.array([0x7F7F0000], np.uint32).view(np.float32)[0]         # 3.3895314e38
BF16_RNE_LIMIT = np.array([0x7F7F7FFF], np.uint32).view(np.float32)[0]   # 3.3961773e38: the largest fp32 that rounds to a finite bf16
                                                                          # (0x7F7F8000 is a tie that rounds to even = bf16 inf)


@pytest.fixture(scope="module")
def hip():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from convnet_amd.matrix import Matrix
    from hip_adapter import HipImpl
    Matrix.SetupCUDADevice(0)
    return HipImpl()


# per data family: (largest allowed err(split) / err(fp32 path),  absolute floor in units of 2^-24 of sum|ab| below which the ratio is
# not asked — both errors are then rounding noise of the last bit —,  absolute cap on err(split) in the same units)
BOUNDS = {"normal": (1.25, 1.0, 8.0), "huge": (1.25, 1.0, 8.0), "dynamic_range": (2.0, 2.0, 32.0), "cancellation": (16.0, 0.0, 0.25),
          "tiny": (8.0, 0.0, 32.0)}


def _check(tag, kind, es, ef, table):
    u = 2.0 ** -24
    table.append(f"{tag:12s} {kind:14s} split {es / u:8.3f} x 2^-24   fp32 path {ef / u:8.3f} x 2^-24 of sum|ab|   ratio {es / max(ef, 1e-300):5.2f}")
    print(table[-1])
    ratio, floor, cap = BOUNDS[kind]
    assert np.isfinite(es) and np.isfinite(ef), (tag, kind, es, ef)
    assert es <= cap * u, (tag, kind, "absolute cap", es / u, cap)
    assert es <= max(ratio * ef, floor * u), (tag, kind, "ratio to the fp32 path", es / u, ef / u, ratio)


def _both_paths(fn):
    """fn() on the bf16-split path and on the fp32-instruction path."""
    from convnet_amd import _lib
    out = {}
    try:
        for name, v in (("split", 1), ("fp32", 0)):
            _lib.lib.convnet_hip_set_matrix_path(v)
            out[name] = fn()
    finally:
        _lib.lib.convnet_hip_set_matrix_path(1)
    return out["split"], out["fp32"]


TABLE = []   # every measured row of this process (tools/profile_round.sh copies the -s output into profiles/)

CONV2 = Geom(N=256, C=96, H=55, W=55, F=256, Ky=5, Kx=5, sy=2, sx=2)
CONV4 = Geom(N=256, C=384, H=13, W=13, F=384, Ky=3, Kx=3, pady=1, padx=1)


def _data(kind, rng, g):
    """(x, w, dy) for one adversarial family; layouts: x (C,H,W,N), w (C,Ky,Kx,F), dy (F,My,Mx,N) (tests/oracle Geom)."""
    x = rng.standard_normal(g.in_shape(), dtype=np.float32)
    w = rng.standard_normal(g.filt_shape(), dtype=np.float32) * np.float32(0.05)
    dy = rng.standard_normal(g.out_shape(), dtype=np.float32)
    if kind == "normal":
        return x, w, dy
    if kind == "dynamic_range":
        # a different power of two per channel / filter: every dot product mixes terms 2^40 apart
        x *= np.exp2(rng.integers(-20, 21, (g.C, 1, 1, 1))).astype(np.float32)
        w *= np.exp2(rng.integers(-20, 21, (g.C, 1, 1, 1))).astype(np.float32)
        dy *= np.exp2(rng.integers(-20, 21, (g.F, 1, 1, 1))).astype(np.float32)
        return x, w, dy
    if kind == "cancellation":
        # channel 2i+1 repeats channel 2i up to 2^-12 (input) and with the opposite sign (filters): each pair of terms of a forward
        # dot product cancels to ~2^-12 of its magnitude — whatever survives is the low bits of the products.  Likewise over filter
        # pairs for the backward direction and over image pairs for the weight gradient.
        x[1::2] = x[0::2][:x[1::2].shape[0]] * (1 + np.float32(2.0 ** -12) * rng.standard_normal(x[1::2].shape, dtype=np.float32))
        w[1::2] = -w[0::2][:w[1::2].shape[0]]
        w[..., 1::2] = -w[..., 0::2] * (1 + np.float32(2.0 ** -12) * rng.standard_normal(w[..., 1::2].shape, dtype=np.float32))
        x[..., 1::2] = x[..., 0::2]
        dy[1::2] = dy[0::2][:dy[1::2].shape[0]]
        dy[..., 1::2] = -dy[..., 0::2] * (1 + np.float32(2.0 ** -12) * rng.standard_normal(dy[..., 1::2].shape, dtype=np.float32))
        return x, w, dy
    if kind == "tiny":
        # |x| ~ 2^-116: h is normal, the residuals m and l are fp32 denormals (bf16 denormals as MFMA operands)
        return x * np.float32(2.0 ** -116), w * np.float32(2.0 ** 20), dy * np.float32(2.0 ** -116)
    if kind == "huge":
        # magnitudes up to the largest the split carries (3.39e38 rounds to a finite bf16); filters small enough that sums stay finite
        with np.errstate(over="ignore"):   # the few samples beyond 3.4 sigma overflow to inf and are clipped back
            x = np.clip(x * np.float32(1e38), -BF16_RNE_LIMIT, BF16_RNE_LIMIT)
            dy = np.clip(dy * np.float32(1e38), -BF16_RNE_LIMIT, BF16_RNE_LIMIT)
        return x, w * np.float32(2.0 ** -20), dy
    raise ValueError(kind)


def _fprop_err(g, x, w, y, pix, rng):
    """max |y - exact| / sum|ab| over the sampled output pixels (all filters, all images)."""
    worst = 0.0
    for (oy, ox) in pix:
        acc = np.zeros((g.F, g.N)); mag = np.zeros((g.F, g.N))
        for ky in range(g.Ky):
            for kx in range(g.Kx):
                iy, ix = oy * g.sy + ky - g.pady, ox * g.sx + kx - g.padx
                if 0 <= iy < g.H and 0 <= ix < g.W:
                    a, b = w[:, ky, kx, :].astype(np.float64), x[:, iy, ix, :].astype(np.float64)   # (C,F), (C,N)
                    acc += a.T @ b
                    mag += np.abs(a).T @ np.abs(b)
        e = float((np.abs(y[:, oy, ox, :] - acc) / np.maximum(mag, 1e-300)).max())
        worst = e if not np.isfinite(e) else max(worst, e)
        if not np.isfinite(worst):
            break
    return worst


def _dgrad_err(g, dy, w, dx, pix, rng):
    worst = 0.0
    for (iy, ix) in pix:
        acc = np.zeros((g.C, g.N)); mag = np.zeros((g.C, g.N))
        for ky in range(g.Ky):
            for kx in range(g.Kx):
                ty, tx = iy + g.pady - ky, ix + g.padx - kx
                if ty % g.sy or tx % g.sx:
                    continue
                oy, ox = ty // g.sy, tx // g.sx
                if 0 <= oy < g.My and 0 <= ox < g.Mx:
                    a, b = w[:, ky, kx, :].astype(np.float64), dy[:, oy, ox, :].astype(np.float64)   # (C,F), (F,N)
                    acc += a @ b
                    mag += np.abs(a) @ np.abs(b)
        m = mag > 0
        if m.any():
            e = float((np.abs(dx[:, iy, ix, :] - acc)[m] / mag[m]).max())
            worst = e if not np.isfinite(e) else max(worst, e)
            if not np.isfinite(worst):
                break
    return worst


def _wgrad_err(g, x, dy, dw, taps, rng):
    worst = 0.0
    oy, ox = np.arange(g.My), np.arange(g.Mx)
    for (c, ky, kx) in taps:
        iy, ix = oy * g.sy + ky - g.pady, ox * g.sx + kx - g.padx
        my, mx = (iy >= 0) & (iy < g.H), (ix >= 0) & (ix < g.W)
        xs = x[c][np.ix_(iy[my], ix[mx])].astype(np.float64).reshape(-1)               # (pixels*N)
        ds = dy[:, oy[my]][:, :, ox[mx]].astype(np.float64).reshape(g.F, -1)           # (F, pixels*N)
        acc, mag = ds @ xs, np.abs(ds) @ np.abs(xs)
        e = float((np.abs(dw[c, ky, kx, :] - acc) / np.maximum(mag, 1e-300)).max())
        worst = e if not np.isfinite(e) else max(worst, e)
        if not np.isfinite(worst):
            break
    return worst


@pytest.mark.parametrize("kind", ["normal", "dynamic_range", "cancellation", "tiny", "huge"])
@pytest.mark.parametrize("layer", ["conv2", "conv4"])
def test_split_path_error_vs_float64_is_within_a_quarter_of_the_fp32_instruction_path(hip, layer, kind):
    g = {"conv2": CONV2, "conv4": CONV4}[layer]
    rng = np.random.default_rng({"conv2": 20, "conv4": 40}[layer] + len(kind))
    x, w, dy = _data(kind, rng, g)
    opix = [(0, 0), (g.My - 1, g.Mx - 1), (0, g.Mx - 1)] + [(int(rng.integers(g.My)), int(rng.integers(g.Mx))) for _ in range(5)]
    ipix = [(0, 0), (g.H - 1, g.W - 1), (1, g.W - 2)] + [(int(rng.integers(g.H)), int(rng.integers(g.W))) for _ in range(5)]
    taps = [(int(rng.integers(g.C)), int(rng.integers(g.Ky)), int(rng.integers(g.Kx))) for _ in range(4)]
    rows = []
    ys, yf = _both_paths(lambda: hip.conv_up(g, x, w))
    rows.append(("fprop", _fprop_err(g, x, w, ys, opix, rng), _fprop_err(g, x, w, yf, opix, rng)))
    ds, df = _both_paths(lambda: hip.conv_down(g, dy, w))
    rows.append(("dgrad", _dgrad_err(g, dy, w, ds, ipix, rng), _dgrad_err(g, dy, w, df, ipix, rng)))
    if kind != "huge":   # a weight gradient sums 173 k (conv2) / 43 k (conv4) products of two huge operands: it overflows on any path
        dyw = dy * np.float32(2.0 ** 116) if kind == "tiny" else dy   # tiny x tiny underflows on any path: tiny inputs, normal derivatives
        ws, wf = _both_paths(lambda: hip.conv_outp(g, x, dyw))
        rows.append(("wgrad", _wgrad_err(g, x, dyw, ws, taps, rng), _wgrad_err(g, x, dyw, wf, taps, rng)))
    for op, es, ef in rows:
        _check(f"{layer} {op}", kind, es, ef, TABLE)


@pytest.mark.parametrize("kind", ["normal", "dynamic_range", "cancellation"])
def test_split_path_error_vs_float64_fc6_exact_shape(hip, kind):
    """fc6 at N = 256: out(N, 4096) = in(N, 9216) W(4096, 9216)^T, din = dout W, dW = dout^T in (fc_edge.cc:60-110) — all 1 M outputs
    of the forward product against float64."""
    N, D, F = 256, 9216, 4096
    rng = np.random.default_rng(66 + len(kind))
    a = rng.standard_normal((D, N), dtype=np.float32)        # activations, numpy (cols, rows) of the column-major (N, D)
    w = rng.standard_normal((D, F), dtype=np.float32) * np.float32(0.02)
    if kind == "dynamic_range":
        s = np.exp2(rng.integers(-20, 21, (D, 1))).astype(np.float32)
        a, w = a * s, w * np.exp2(rng.integers(-20, 21, (D, 1))).astype(np.float32)
    elif kind == "cancellation":
        a[1::2] = a[0::2] * (1 + np.float32(2.0 ** -12) * rng.standard_normal(a[1::2].shape, dtype=np.float32))
        w[1::2] = -w[0::2]
    ys, yf = _both_paths(lambda: hip.dot(a, w, np.zeros((F, N), np.float32), 0.0, 1.0, False, True))
    exact = w.astype(np.float64).T @ a.astype(np.float64)          # (F, N)
    mag = np.abs(w).astype(np.float64).T @ np.abs(a).astype(np.float64)
    es, ef = float((np.abs(ys - exact) / mag).max()), float((np.abs(yf - exact) / mag).max())
    _check("fc6 fprop", kind, es, ef, TABLE)


def test_special_values_on_both_paths(hip):
    """What a NaN / inf / out-of-bf16-range operand does.  fp32 path: IEEE (cublasSgemm's behaviour, cudamat.cu:2130-2152).
    Split path, as documented in include/convnet_hip.h (convnet_hip_set_matrix_path):
      * NaN operands poison exactly the outputs they poison on the fp32 path;
      * finite operands up to 3.396e38 (everything that rounds to a finite bf16) behave like any other value;
      * the FILTER operand of the conv / FC forward and backward kernels (pre-split into bf16 planes by filter_planes_kernel /
        dgrad_filter_planes_kernel) saturates: +-inf and finite values above the bf16 range enter as +-3.39e38;
      * the ACTIVATION / DERIVATIVE operand is split in the kernels' inner loops without a range check: a value with
        |x| > 3.396e38 (incl. +-inf) rounds to a bf16 inf whose residual x - inf is NaN, so the outputs it touches are non-finite —
        NaN where the fp32 path gives +-inf (or NaN).  The set of non-finite outputs is the same on both paths."""
    g = Geom(N=32, C=32, H=9, W=9, F=64, Ky=3, Kx=3, pady=1, padx=1)
    rng = np.random.default_rng(7)
    x = rng.standard_normal(g.in_shape(), dtype=np.float32)
    w = rng.standard_normal(g.filt_shape(), dtype=np.float32)
    # (1) NaN in one activation, one filter weight
    xn, wn = x.copy(), w.copy()
    xn[3, 4, 4, 5] = np.nan
    wn[7, 1, 1, 9] = np.nan
    ys, yf = _both_paths(lambda: hip.conv_up(g, xn, wn))
    assert np.array_equal(np.isnan(ys), np.isnan(yf)) and np.isnan(ys).sum() > 0
    ok = ~np.isnan(yf)
    assert np.abs(ys[ok] - yf[ok]).max() < 1e-4 * np.abs(yf[ok]).mean()
    # (2) values at the edge of the range the split carries, on the in-loop (activation) operand
    xb = x.copy()
    xb[2, 3, 3, 1], xb[2, 3, 4, 1] = BF16_RNE_LIMIT, -BF16_MAX
    wb = w * np.float32(2.0 ** -30)
    ys, yf = _both_paths(lambda: hip.conv_up(g, xb, wb))
    assert np.all(np.isfinite(ys)) and np.all(np.isfinite(yf))
    assert np.abs(ys - yf).max() <= 2e-6 * np.abs(yf).max()
    # (3) +-inf and an above-range finite value in the FILTER operand: saturates to +-bf16 max on the split path
    wi = w.copy()
    wi[5, 0, 0, 11], wi[6, 2, 2, 12], wi[8, 1, 0, 13] = np.inf, -np.inf, np.float32(3.40e38)
    xs = x * np.float32(2.0 ** -40)    # keep the saturated products finite
    ys, yf = _both_paths(lambda: hip.conv_up(g, xs, wi))
    assert np.all(np.isfinite(ys)), "filter planes saturate"
    assert np.isinf(yf[11]).any() and np.isinf(yf[12]).any()                           # IEEE on the fp32 path
    untouched = [f for f in range(g.F) if f not in (11, 12, 13)]
    assert np.abs(ys[untouched] - yf[untouched]).max() < 1e-4 * np.abs(yf[untouched]).mean()
    wsat = wi.copy()
    wsat[5, 0, 0, 11], wsat[6, 2, 2, 12], wsat[8, 1, 0, 13] = BF16_MAX, -BF16_MAX, BF16_MAX
    ysat, _ = _both_paths(lambda: hip.conv_up(g, xs, wsat))
    assert np.array_equal(ys, ysat), "an out-of-range filter value acts exactly as +-bf16 max"
    # (4) +-inf / above-range finite in the ACTIVATION operand: the documented deviation
    xi = x.copy()
    xi[1, 2, 2, 3], xi[4, 6, 6, 8], xi[9, 0, 8, 20] = np.inf, -np.inf, np.float32(3.40e38)
    ys, yf = _both_paths(lambda: hip.conv_up(g, xi, w * np.float32(2.0 ** -10)))
    bad_s, bad_f = ~np.isfinite(ys), ~np.isfinite(yf)
    touched_inf = bad_f.copy()
    assert touched_inf.sum() > 0 and np.all(bad_s[touched_inf]), "every output the fp32 path makes non-finite is non-finite on the split path"
    # the above-range FINITE value is finite on the fp32 path and NaN on the split path: the only outputs where the two differ in kind
    extra = bad_s & ~bad_f
    assert set(np.unique(np.nonzero(extra)[3])) <= {20}, "only image 20 (the 3.40e38 activation) may differ in finiteness"
    ok = ~bad_s
    assert np.abs(ys[ok] - yf[ok]).max() < 1e-4 * np.abs(yf[ok]).mean()
    # (5) an FC WEIGHT with +-inf: FC weights are split inside the loops like an activation, so the outputs they touch are
    # non-finite on both paths (NaN on the split path where the fp32 path gives +-inf) — no saturation, a diverged run stays visible
    rng = np.random.default_rng(8)
    a = rng.standard_normal((96, 64), dtype=np.float32)      # numpy (D, N) == column-major Matrix (N images, D)
    wfc = rng.standard_normal((96, 40), dtype=np.float32)    # numpy (D, F) == Matrix (F, D)
    wfc[17, 3], wfc[2, 9] = np.inf, -np.inf
    t0 = np.zeros((40, 64), np.float32)                      # numpy (F, N) == Matrix (N, F)
    ys, yf = _both_paths(lambda: hip.dot(a, wfc, t0.copy(), 0.0, 1.0, False, True))
    bad_s, bad_f = ~np.isfinite(ys), ~np.isfinite(yf)
    assert np.array_equal(bad_s, bad_f) and bad_f[3].all() and bad_f[9].all() and bad_f.sum() == 2 * 64
    assert np.abs(ys[~bad_s] - yf[~bad_s]).max() < 1e-4 * np.abs(yf[~bad_f]).mean()


def test_default_path_of_the_c_abi_gives_ieee_values_for_non_finite_operands(hip):
    """on the library's DEFAULT matrix path (0) a +-inf / above-bf16-range activation gives the very VALUES IEEE
    fp32 arithmetic gives (+-inf where the sum is infinite, NaN only for inf - inf), checked against numpy on the affected outputs."""
    from convnet_amd import _lib
    g = Geom(N=32, C=16, H=7, W=7, F=32, Ky=3, Kx=3, pady=1, padx=1)
    rng = np.random.default_rng(9)
    x = rng.standard_normal(g.in_shape(), dtype=np.float32)
    w = np.abs(rng.standard_normal(g.filt_shape(), dtype=np.float32)) + np.float32(0.5)   # positive weights: no inf - inf
    x[1, 2, 2, 3], x[4, 5, 5, 8], x[9, 0, 6, 20] = np.inf, -np.inf, np.float32(3.40e38)
    _lib.lib.convnet_hip_set_matrix_path(0)
    try:
        y = hip.conv_up(g, x, w)
    finally:
        _lib.lib.convnet_hip_set_matrix_path(1)
    ref = oracle.port.conv_up(g, x, w)
    assert np.array_equal(np.isposinf(y), np.isposinf(ref)) and np.array_equal(np.isneginf(y), np.isneginf(ref))
    assert np.isposinf(y[:, :, :, 3]).any() and np.isneginf(y[:, :, :, 8]).any() and not np.isnan(y).any()
    fin = np.isfinite(ref)
    assert np.abs(y[fin] - ref[fin]).max() < 1e-4 * np.abs(ref[fin & (np.abs(ref) < 1e30)]).mean() or np.allclose(y[fin], ref[fin], rtol=1e-4)
