"""gpv_kernel (convnet_amd/csrc/patch_gemm.hip: gpw_kernel's 8-unit tile for tap rows cut into GROUPS of three or two taps — the
5 x 5 stride-2 convolution of AlexNet's second layer forward, {0,2,4} / {1,3}, and the four stride classes of its input gradient,
3- and 2-tap rows) — its control logic restated in Python and run against the CPU oracle; no GPU.

What is modelled, as in tests/test_patch_wide_cpu.py: the unit / slot bookkeeping, the per-lane slot description PER GROUP, the
superchunk walk (channel block, tap row, group) with its three iterators (MFMAs; filter chunk two chunks ahead; slab one superchunk
ahead), the two slab buffers changing roles per superchunk, and the rule each wave waits by: everything it issued BEFORE this chunk's
loads (3 filter pieces + one slot, or + two slots in the first chunk of a two-chunk superchunk, whose successor's first-needed slots
must all be under way by then).  Loads land either the moment they are issued (a write into a region still being read shows up) or as
late as the issuing wave's wait allows (a read before its data shows up); LDS starts as NaN."""
import numpy as np
import pytest

import oracle
from oracle import Geom

P, NS, WAVES = 8, 12, 4
ZERO_SLOT = -2


def groups_of(TX, ssx, d):
    """tap groups of one tap row: group g = taps gb0[g] + i*d*ssx, i < cnt[g] (patch_shape_ok / gpv_kernel)"""
    ng = ssx
    cnt = [(TX - g + ssx - 1) // ssx if TX > g else 0 for g in range(ng)]
    gb0 = [g if d > 0 else g + (cnt[g] - 1) * ssx for g in range(ng)]
    return ng, cnt, gb0


def run_tile(par, src, bank, col_tile, lazy, sc_range=None, mutate=None):
    """One block of gpv_kernel.  par: the GGParams / class fields the kernel reads; src [KC][SH][SW][N]; bank[q] = filter chunk q as a
    [R][16] matrix (q = cb*TYX + a*TX + b).  sc_range: (first, end) superchunk of a K-range (split-K / tail split), None = the whole
    reduction with border tap rows skipped.  Returns {unit j: [R][64] accumulator}."""
    G, GX, IB = par["G"], par["GX"], par["IB"]
    SH, SW, ssy, ssx, y0, x0, d = par["SH"], par["SW"], par["ssy"], par["ssx"], par["y0"], par["x0"], par["dir"]
    TX, TYX = par["TX"], par["TYX"]
    TYn, units = TYX // TX, IB * G
    ng, cnt, gb0 = groups_of(TX, ssx, d)
    assert all(c in (2, 3) for c in cnt)
    # ---- units
    ok, ib, oy, ox, S = [], [], [], [], []
    for j in range(P):
        U = col_tile * P + j
        ok.append(U < units)
        ib.append(U // G if ok[j] else 0)
        m = U - ib[j] * G if ok[j] else 0
        oy.append(m // GX)
        ox.append(m - oy[j] * GX)
        S.append(0 if j == 0 else S[j - 1] + (0 if not ok[j] else 1 if (ib[j] == ib[j - 1] and oy[j] == oy[j - 1]) else 3))
    ys0 = [oy[j] * ssy + y0 for j in range(P) if ok[j]]
    ys_f, ys_l = min(ys0), max(ys0)
    # first-needed order (over three tap slots, whatever the group) -> my_ord[wave][step]; per-"lane" slot description
    my_ord = [[-1] * 3 for _ in range(WAVES)]
    slot = {}   # s -> (source row of tap row 0, source column of group 0, image, smallest tap slot i that reads it)
    seen, no = set(), 0
    for i in range(3):
        for j in range(P):
            sl = S[j] + i
            if ok[j] and sl < NS and sl not in seen:
                seen.add(sl)
                my_ord[no % 4][no // 4] = sl
                no += 1
            if ok[j]:
                assert sl < NS
                imin = min(i, slot[sl][3]) if sl in slot else i
                slot[sl] = (oy[j] * ssy + y0, ox[j] * ssx + x0 + d * gb0[0] + i * ssx, ib[j] * 64, imin)
    assert no <= 12

    def slot_desc(a, g):
        out = {}
        for s, (sy, sx, sib, imin) in slot.items():
            if imin >= cnt[g]:
                continue   # nobody reads it in this group
            ys, xs = sy + d * a, sx + d * (gb0[g] - gb0[0])
            out[s] = (ys, xs, sib) if (0 <= xs < SW and 0 <= ys < SH) else ZERO_SLOT
        return out
    # ---- reduction range
    a_lo, a_hi = 0, TYn - 1
    if sc_range is None:
        if d > 0:
            a_lo, a_hi = max(0, -ys_l), min(TYn - 1, SH - 1 - ys_f)
        else:
            a_lo, a_hi = max(0, ys_f - (SH - 1)), min(TYn - 1, ys_l)
    nrow = max(0, a_hi - a_lo + 1)
    nsc_all = (par["KC"] // 16) * nrow * ng
    sc_beg, sc_end = (0, nsc_all) if sc_range is None else (min(nsc_all, sc_range[0]), min(nsc_all, sc_range[1]))
    nchunks = sum(cnt[sc % ng] for sc in range(sc_beg, sc_end))
    R = bank[0].shape[0]
    acc = {j: np.zeros((R, 64)) for j in range(P) if ok[j]}
    if nchunks == 0:
        return acc
    g_beg, r_beg, cb_beg = sc_beg % ng, (sc_beg // ng) % nrow, sc_beg // (ng * nrow)
    # ---- LDS and in-flight loads
    A_lds = [np.full((R, 16), np.nan) for _ in range(3)]
    B_lds = [[np.full((16, 64), np.nan) for _ in range(NS)] for _ in range(2)]
    inflight = [[] for _ in range(WAVES)]   # per wave: (landing closure, batch id)

    def issue(w, fn, batch_id):
        if lazy:
            inflight[w].append((fn, batch_id))
        else:
            fn()

    def wait_all_before(w, batch_id):   # s_waitcnt vmcnt(this chunk's own load count)
        keep = []
        for fn, b in inflight[w]:
            if b < batch_id:
                fn()
            else:
                keep.append((fn, b))
        inflight[w] = keep

    dstep = d * ssx
    # byte steps of the filter pointer as chunk-index steps: next tap; what a finished group adds (to the next group of the row, or
    # from the last group to the first of the next row); what a finished channel block adds
    x_g = []
    for g in range(ng):
        last = gb0[g] + (cnt[g] - 1) * dstep
        nxt = gb0[g + 1] if g + 1 < ng else TX + gb0[0]
        x_g.append(nxt - last - dstep)
    a_cbs_x = TYX - nrow * TX
    st = dict(a_q=cb_beg * TYX + (a_lo + r_beg) * TX + gb0[g_beg], A_i=0, A_g=g_beg, A_r=r_beg, A_left=nchunks,
              f=[0, 1, 2], B_g=g_beg, B_r=r_beg, B_cb=cb_beg)

    def issue_a(batch_id):
        q, stage = st["a_q"], st["f"][0]
        assert 0 <= q < len(bank), q
        for w in range(WAVES):   # a quarter of the chunk each: rows stand in for the pieces
            lo, hi = (w * R) // 4, ((w + 1) * R) // 4

            def land(lo=lo, hi=hi, q=q, stage=stage):
                A_lds[stage][lo:hi] = bank[q][lo:hi]
            issue(w, land, batch_id)
        st["f"] = st["f"][1:] + st["f"][:1]
        st["A_left"] -= 1
        more = 1 if st["A_left"] > 0 else 0
        g_old = st["A_g"]
        i1 = st["A_i"] + 1
        w1 = 1 if i1 >= cnt[g_old] else 0
        st["A_i"] = i1 * (1 - w1)
        g1 = g_old + w1
        wg = 1 if g1 >= ng else 0
        st["A_g"] = g1 - ng * wg
        r1 = st["A_r"] + wg
        w2 = 1 if r1 >= nrow else 0
        st["A_r"] = r1 - nrow * w2
        st["a_q"] += more * (dstep + w1 * x_g[g_old] + w2 * a_cbs_x)

    def slab_next(step):
        g1 = st["B_g"] + step
        wg = 1 if g1 >= ng else 0
        st["B_g"] = g1 - ng * wg
        r1 = st["B_r"] + wg
        w = 1 if r1 >= nrow else 0
        st["B_r"] = r1 - nrow * w
        st["B_cb"] += w

    def issue_slot(w, sl, buf, cb, desc, enable, batch_id):
        if not enable or sl < 0 or sl not in desc:
            return   # dump region
        dsc = desc[sl]

        def land(sl=sl, buf=buf, cb=cb, dsc=dsc):
            if dsc == ZERO_SLOT:
                B_lds[buf][sl] = np.zeros((16, 64))
            else:
                ys, xs, sib = dsc
                B_lds[buf][sl] = src[16 * cb:16 * cb + 16, ys, xs, sib:sib + 64].astype(np.float64)
        issue(w, land, batch_id)

    # prologue (batch ids -3, -2): slab 0, filter chunks 0 and 1; vmcnt(0)
    desc0 = slot_desc(a_lo + st["B_r"], st["B_g"])
    for w in range(WAVES):
        for q in range(3):
            issue_slot(w, my_ord[w][q], 0, st["B_cb"], desc0, True, -3)
    slab_next(1)
    issue_a(-2)
    issue_a(-2)
    for w in range(WAVES):
        wait_all_before(w, 10 ** 9)
    # consumer state
    stage, bufsel, ti, sc, g_cur = 0, 0, 0, sc_beg, g_beg
    o = [list(my_ord[w]) for w in range(WAVES)]

    def read_chunk():   # load_a + read_b right behind a barrier
        a = A_lds[stage].copy()
        b = {j: B_lds[bufsel][S[j] + ti].copy() for j in acc}
        return a, b
    cur = read_chunk()
    for c in range(nchunks):
        cnt_cur = cnt[g_cur]
        nsl = 2 if (cnt_cur == 2 and ti == 0) else 1
        if mutate == "one_slot_always":
            nsl = 1
        # this chunk's loads: the filter chunk two ahead, one or two slots of the next slab into the idle buffer
        issue_a(c)
        desc = slot_desc(a_lo + st["B_r"], st["B_g"])
        for w in range(WAVES):
            for _ in range(nsl):
                issue_slot(w, o[w][0], bufsel ^ 1 if mutate != "same_buffer" else bufsel, st["B_cb"], desc, sc + 1 < sc_end, c)
                o[w] = o[w][1:] + o[w][:1]
        # MFMAs of this chunk (operands were read behind the previous barrier)
        a, b = cur
        for j in acc:
            acc[j] += a @ b[j]
        # advance(), wait, barrier, reads of the next chunk
        t1 = ti + 1
        wv = 1 if t1 >= cnt_cur else 0
        ti = t1 * (1 - wv)
        bufsel ^= wv
        sc += wv
        g1 = g_cur + wv
        g_cur = g1 - ng * (1 if g1 >= ng else 0)
        slab_next(wv)
        stage = (stage + 1) % 3
        for w in range(WAVES):
            wait_all_before(w, c if mutate != "wait_one_chunk_late" else c - 1)
        cur = read_chunk() if c + 1 < nchunks else None
    return acc


def conv_by_tiles(g, x, w, dgrad, lazy, ranges=1, mutate=None):
    """fprop of g, or the input gradient of g stride class by stride class, assembled from gpv_kernel tiles; x / w in the oracle's layouts.
    ranges > 1: every tile's reduction is cut into that many K-ranges (superchunk granularity) whose partial tiles are summed."""
    C, F = g.C, g.F
    wl = w.reshape(C, g.Ky, g.Kx, F)
    jobs = []   # (par, bank, output placement)
    if not dgrad:
        par = dict(G=g.My * g.Mx, GX=g.Mx, IB=g.N // 64, N=g.N, SH=g.H, SW=g.W, ssy=g.sy, ssx=g.sx, y0=-g.pady, x0=-g.padx, dir=1,
                   TX=g.Kx, TYX=g.Ky * g.Kx, KC=C)
        bank = [wl[16 * cb:16 * cb + 16, a, b, :].T.astype(np.float64) for cb in range(C // 16) for a in range(g.Ky) for b in range(g.Kx)]
        out = np.zeros((F, g.My, g.Mx, g.N))
        jobs.append((par, bank, (0, 0, 1, 1)))
    else:
        out = np.zeros((C, g.H, g.W, g.N))
        for cy in range(g.sy):
            for cx in range(g.sx):
                # conv_down_impl (gather_gemm.hip): input rows iy with (iy + pad) % sy == cy share the tap set ky = cy + sy*a
                TYc = -(-(g.Ky - cy) // g.sy) if cy < g.Ky else 0
                TXc = -(-(g.Kx - cx) // g.sx) if cx < g.Kx else 0
                jy0 = max(0, -(-(g.pady - cy) // g.sy))
                jx0 = max(0, -(-(g.padx - cx) // g.sx))
                iy0, ix0 = cy - g.pady + g.sy * jy0, cx - g.padx + g.sx * jx0
                if iy0 >= g.H or ix0 >= g.W or TYc * TXc == 0:
                    continue
                GY, GX = (g.H - 1 - iy0) // g.sy + 1, (g.W - 1 - ix0) // g.sx + 1
                par = dict(G=GY * GX, GX=GX, IB=g.N // 64, N=g.N, SH=g.My, SW=g.Mx, ssy=1, ssx=1, y0=jy0, x0=jx0, dir=-1,
                           TX=TXc, TYX=TYc * TXc, KC=F)
                bank = [wl[:, cy + g.sy * a, cx + g.sx * b, 16 * fb:16 * fb + 16].astype(np.float64)
                        for fb in range(F // 16) for a in range(TYc) for b in range(TXc)]
                jobs.append((par, bank, (iy0, ix0, g.sy, g.sx)))
    for par, bank, (py0, px0, psy, psx) in jobs:
        units = par["IB"] * par["G"]
        ng, _, _ = groups_of(par["TX"], par["ssx"], par["dir"])
        nsc = (par["KC"] // 16) * (par["TYX"] // par["TX"]) * ng
        for ct in range((units + P - 1) // P):
            if ranges == 1:
                parts = [run_tile(par, x, bank, ct, lazy, None, mutate)]
            else:
                cps = -(-nsc // ranges)
                parts = [run_tile(par, x, bank, ct, lazy, (k * cps, (k + 1) * cps), mutate) for k in range(ranges)]
            for j in parts[0]:
                v = sum(p[j] for p in parts)
                U = ct * P + j
                ib, m = divmod(U, par["G"])
                out[:, py0 + psy * (m // par["GX"]), px0 + psx * (m % par["GX"]), ib * 64:ib * 64 + 64] = v
    return out


FPROP = [
    Geom(N=64, C=16, H=23, W=23, F=8, Ky=5, Kx=5, sy=2, sx=2),                   # conv2's form: 10-wide output rows, groups of 3 and 2
    Geom(N=128, C=32, H=21, W=21, F=8, Ky=5, Kx=5, sy=2, sx=2),                  # 9-wide rows: a wrap in almost every tile; image-block wrap
    Geom(N=64, C=16, H=19, W=25, F=8, Ky=5, Kx=5, sy=2, sx=2, pady=2, padx=2),   # rectangular, padding: border columns and whole tap rows outside
    Geom(N=64, C=16, H=20, W=20, F=8, Ky=4, Kx=4, sy=2, sx=2, pady=1, padx=1),   # groups of 2 and 2
    Geom(N=64, C=16, H=12, W=12, F=8, Ky=3, Kx=3, pady=1, padx=1),               # the 3 x 3 stride-1 case gpw_kernel runs (one group of 3)
    Geom(N=64, C=16, H=12, W=12, F=8, Ky=2, Kx=2),                               # one group of 2: every chunk pattern is (two slots, one slot)
]
DGRAD = [
    Geom(N=64, C=8, H=23, W=23, F=32, Ky=5, Kx=5, sy=2, sx=2),                   # conv2's form: classes 3x3, 3x2, 2x3, 2x2
    Geom(N=64, C=8, H=21, W=25, F=16, Ky=5, Kx=5, sy=2, sx=2, pady=2, padx=2),   # padded: the classes start at different input pixels
    Geom(N=128, C=8, H=13, W=13, F=16, Ky=3, Kx=3, pady=1, padx=1),              # stride 1: one class of 3-tap rows
    Geom(N=64, C=8, H=20, W=20, F=16, Ky=4, Kx=4, sy=2, sx=2, pady=1, padx=1),   # four 2 x 2 classes
]
_id = lambda g: f"N{g.N}C{g.C}H{g.H}W{g.W}F{g.F}k{g.Ky}x{g.Kx}s{g.sy}p{g.pady}"  # noqa: E731


def _data(g, seed, dgrad):
    rng = np.random.default_rng(seed)
    a = rng.standard_normal(g.out_shape() if dgrad else g.in_shape()).astype(np.float32)
    w = rng.standard_normal(g.filt_shape()).astype(np.float32)
    ref = (oracle.port.conv_down(g, a, w) if dgrad else oracle.port.conv_up(g, a, w)).astype(np.float64)
    return a, w, ref


@pytest.mark.parametrize("lazy", [False, True], ids=["eager", "lazy"])
@pytest.mark.parametrize("g", FPROP, ids=_id)
def test_group_tile_schedule_fprop(g, lazy):
    x, w, ref = _data(g, 5, False)
    got = conv_by_tiles(g, x, w, False, lazy)
    assert not np.isnan(got).any()
    assert np.abs(got - ref).max() < 2e-4 * np.abs(ref).max()


@pytest.mark.parametrize("lazy", [False, True], ids=["eager", "lazy"])
@pytest.mark.parametrize("g", DGRAD, ids=_id)
def test_group_tile_schedule_dgrad(g, lazy):
    dy, w, ref = _data(g, 6, True)
    got = conv_by_tiles(g, dy, w, True, lazy)
    assert not np.isnan(got).any()
    assert np.abs(got - ref).max() < 2e-4 * np.abs(ref).max()


@pytest.mark.parametrize("ranges", [2, 3, 7])
@pytest.mark.parametrize("lazy", [False, True], ids=["eager", "lazy"])
def test_group_tile_schedule_k_ranges(lazy, ranges):
    """split-K / tail split: a range may begin with the second group of a tap row and end inside a channel block"""
    g = FPROP[0]
    x, w, ref = _data(g, 7, False)
    got = conv_by_tiles(g, x, w, False, lazy, ranges)
    assert not np.isnan(got).any()
    assert np.abs(got - ref).max() < 2e-4 * np.abs(ref).max()


@pytest.mark.parametrize("mutate,lazy", [("one_slot_always", True), ("same_buffer", False), ("wait_one_chunk_late", True)])
def test_group_tile_schedule_detects_seeded_faults(mutate, lazy):
    """the model must be able to fail: one slot per chunk also in two-chunk superchunks (the next slab's first-needed slots are late),
    the next slab written into the buffer being read, the wait letting the previous chunk's loads stay in flight"""
    g = FPROP[0]
    x, w, ref = _data(g, 8, False)
    got = conv_by_tiles(g, x, w, False, lazy, 1, mutate)
    assert np.isnan(got).any() or np.abs(got - ref).max() > 1e-2 * np.abs(ref).max()
