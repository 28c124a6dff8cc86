"""CPU restatement of the bf16-split product arithmetic the GEMM kernels use by default (`split8` / `split_mac` in
convnet_amd/csrc/gather_gemm.hip, `convnet_hip_set_matrix_path` in include/convnet_hip.h).

What the kernels claim and this file checks with numpy, no GPU needed:
  * the three-way split x = h + m + l (h = rne_bf16(x), m = rne_bf16(x - h), l = x - h - m) is EXACT for every finite fp32 value in
    the normal range, and every term is a bf16 (8 significant bits);
  * the three dropped cross products (ml, lm, ll) are together <= 2^-23 of |a b|;
  * a reduction accumulated in fp32 from the six kept products per element pair is as close to the double-precision result as the
    plain fp32 FMA-free dot product is (the accumulation rounding dominates both).
The device-side counterpart is tools/split_gemm.hip (one 128 x 512 tile, K = 3456, against double on the MI355X)."""
import numpy as np


def rne_bf16(x):
    """fp32 -> nearest bf16 (ties to even), returned as fp32 (what v_cvt_pk_bf16_f32 followed by the widening shift gives)."""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    rounded = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return rounded.astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, np.float32)
    h = rne_bf16(x)
    r1 = (x - h).astype(np.float32)
    m = rne_bf16(r1)
    l = (r1 - m).astype(np.float32)
    return h, m, l


def is_bf16(v):
    return np.all((np.asarray(v, np.float32).view(np.uint32) & 0xFFFF) == 0)


def samples(rng, n):
    mant = rng.standard_normal(n).astype(np.float32)
    expo = rng.integers(-60, 60, n)
    x = (mant * np.exp2(expo)).astype(np.float32)
    edge = np.array([0.0, -0.0, 1.0, -1.0, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24, 255.5, 3.0e38, -3.0e38, 1.1754944e-38 * 2 ** 20,
                     np.float32(np.pi), np.float32(1.0 / 3.0), 0.99609375, 1.00390625, 1.99999988], np.float32)
    return np.concatenate([x, edge])


def test_three_way_split_is_exact_and_every_term_is_a_bf16():
    rng = np.random.default_rng(0)
    x = samples(rng, 400000)
    h, m, l = split3(x)
    assert is_bf16(h) and is_bf16(m) and is_bf16(l)
    # exact in real arithmetic: check in float64 (three bf16 terms of decreasing magnitude add without rounding there)
    assert np.array_equal(h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64), x.astype(np.float64))
    # the residuals shrink by 2^-8 per level (round to nearest): |m| <= 2^-8 |x|, |l| <= 2^-16 |x|
    ax = np.abs(x.astype(np.float64))
    assert np.all(np.abs(m.astype(np.float64)) <= ax * 2.0 ** -8 * (1 + 2.0 ** -7))
    assert np.all(np.abs(l.astype(np.float64)) <= ax * 2.0 ** -16 * (1 + 2.0 ** -7))


def test_dropped_cross_terms_are_below_2_pow_minus_23_of_the_product():
    rng = np.random.default_rng(1)
    a, b = samples(rng, 200000), samples(rng, 200000)
    (ah, am, al), (bh, bm, bl) = split3(a), split3(b)
    f = lambda v: v.astype(np.float64)
    kept = f(ah) * f(bh) + f(ah) * f(bm) + f(am) * f(bh) + f(ah) * f(bl) + f(al) * f(bh) + f(am) * f(bm)
    exact = f(a) * f(b)
    dropped = np.abs(exact - kept)     # = |am*bl + al*bm + al*bl| exactly (all products of bf16 pairs are exact in float64)
    assert np.all(dropped <= np.abs(exact) * 2.0 ** -23)
    # typical size is far smaller than the bound
    nz = np.abs(exact) > 0
    assert np.median(dropped[nz] / np.abs(exact[nz])) < 2.0 ** -27


def _dot_fp32_sequential(a, b):
    acc = np.zeros(a.shape[1:], np.float32)
    for k in range(a.shape[0]):
        acc = (acc + (a[k] * b[k]).astype(np.float32)).astype(np.float32)
    return acc


def _dot_split(a, b, depth=16):
    """The kernels' order: per chunk of `depth` k values, six partial sums (one per kept product, each summed exactly like the MFMA's
    fp32 accumulate is modelled here: products exact, one fp32 rounding per instruction), smallest first."""
    acc = np.zeros(a.shape[1:], np.float32)
    for k0 in range(0, a.shape[0], depth):
        (ah, am, al), (bh, bm, bl) = split3(a[k0:k0 + depth]), split3(b[k0:k0 + depth])
        for x, y in ((am, bm), (ah, bl), (al, bh), (ah, bm), (am, bh), (ah, bh)):
            part = (x.astype(np.float64) * y.astype(np.float64)).sum(axis=0)
            acc = (acc.astype(np.float64) + part).astype(np.float32)
    return acc


def test_split_reduction_is_as_accurate_as_an_fp32_dot_product():
    rng = np.random.default_rng(2)
    for K in (4, 64, 3456):
        a = (rng.standard_normal((K, 4096)) * 0.05).astype(np.float32)
        b = rng.standard_normal((K, 4096)).astype(np.float32)
        ref = (a.astype(np.float64) * b.astype(np.float64)).sum(axis=0)
        mag = np.abs(a.astype(np.float64) * b.astype(np.float64)).sum(axis=0)
        e_fp32 = np.abs(_dot_fp32_sequential(a, b) - ref) / mag
        e_split = np.abs(_dot_split(a, b) - ref) / mag
        # never worse than a few ulps of the summed magnitudes, and not worse than the sequential fp32 sum at long reductions
        assert e_split.max() <= 4 * 2.0 ** -24, (K, e_split.max() / 2.0 ** -24)
        if K >= 64:
            assert np.sqrt((e_split ** 2).mean()) <= np.sqrt((e_fp32 ** 2).mean()), (K, e_split.mean(), e_fp32.mean())


def _planes_from_bank(bank):
    """bank[k, row] (k-rows of 16-deep chunks) -> planes[chunk, plane, lh, row, j]: k-slot j of k-group lh is k-row 2j + lh of the chunk
    (the layout ggp_kernel<..., SPLIT, APRE> stages: DESIGN.md section 2.1b)."""
    K, R = bank.shape
    out = np.zeros((K // 16, 3, 2, R, 8), np.float32)
    for lh in range(2):
        for j in range(8):
            h, m, l = split3(bank[2 * j + lh::16])          # rows (chunk, row)
            out[:, 0, lh, :, j], out[:, 1, lh, :, j], out[:, 2, lh, :, j] = h, m, l
    return out


def test_filter_plane_layouts_equal_relayout_then_split():
    """filter_planes_kernel / dgrad_filter_planes_kernel (csrc/gather_gemm.hip) index the reference's filter bank
    W[f + F*(kx + Kx*(ky + Ky*c))] directly; restated here and compared with the two-step definition: tap-major re-layout of the bank
    (filter_tapmajor_kernel / dgrad_filter_kernel's tap_major order), then the per-chunk plane split."""
    rng = np.random.default_rng(5)
    F, C, Ky, Kx = 32, 48, 3, 5
    W = rng.standard_normal(F * Kx * Ky * C).astype(np.float32)
    w = lambda f, ky, kx, c: W[f + F * (kx + Kx * (ky + Ky * c))]
    TYX = Ky * Kx
    # forward: k = (cb*TYX + tap)*16 + c16, rows f
    bank = np.zeros((C * TYX, F), np.float32)
    for cb in range(C // 16):
        for tap in range(TYX):
            for c16 in range(16):
                bank[(cb * TYX + tap) * 16 + c16] = [W[f + F * (tap + TYX * (16 * cb + c16))] for f in range(F)]
    want = _planes_from_bank(bank)
    got = np.zeros_like(want)
    for chunk in range(C // 16 * TYX):
        tap, cb = chunk % TYX, chunk // TYX
        for lh in range(2):
            for f in range(F):
                x = np.array([W[f + F * (tap + TYX * (16 * cb + 2 * j + lh))] for j in range(8)], np.float32)
                got[chunk, 0, lh, f], got[chunk, 1, lh, f], got[chunk, 2, lh, f] = split3(x)
    assert np.array_equal(got, want)
    # input gradient, stride class (cy, cx) of a stride-(2, 2) convolution: k = (fb*TYXc + tap)*16 + f16, rows c
    sy = sx = 2
    for cy, cx in ((0, 0), (1, 0), (1, 1)):
        TYc, TXc = -(-(Ky - cy) // sy), -(-(Kx - cx) // sx)
        TYXc = TYc * TXc
        bank = np.zeros((F * TYXc, C), np.float32)
        for fb in range(F // 16):
            for a in range(TYc):
                for b in range(TXc):
                    for f16 in range(16):
                        bank[(fb * TYXc + b + TXc * a) * 16 + f16] = [w(16 * fb + f16, cy + sy * a, cx + sx * b, c) for c in range(C)]
        want = _planes_from_bank(bank)
        got = np.zeros_like(want)
        for chunk in range(F // 16 * TYXc):
            tap, fb = chunk % TYXc, chunk // TYXc
            a, b = tap // TXc, tap % TXc
            for lh in range(2):
                for c in range(C):
                    base = F * ((cx + sx * b) + Kx * ((cy + sy * a) + Ky * c)) + 16 * fb + lh
                    got[chunk, 0, lh, c], got[chunk, 1, lh, c], got[chunk, 2, lh, c] = split3(W[base:base + 16:2])
        assert np.array_equal(got, want), (cy, cx)
