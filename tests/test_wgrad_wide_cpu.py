"""wgw_kernel (convnet_amd/csrc/wgrad_wide.hip, wgrad tile 1): its chunk loop as a data-level model driven by the REAL schedule —
convnet_amd/csrc/wgrad_wide_schedule.h is plain C++, compiled here with g++ and dumped — against numpy.  No GPU.  The kernel was written
with the round's GPU budget spent; its schedule rules are static_asserts in that header, this is the other half: two LDS stages that
change roles every chunk, staging loads that land either at once or as late as the wait in front of the chunk barrier allows (LDS and
registers start as NaN: anything read before it was written, or overwritten before it was used, poisons the sum), the prologue, the
(pixel, image chunk) walk one chunk ahead with its zero-page loads past the end, one raw register set per operand, the filter-column
slots, and that a step's operands do not change while its six products run.

One wave is modelled: 4 row tiles x NTL filter tiles of ONE row each; a chunk is 32 images = 2 halves x 4 pairs x 4 images."""
import json
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "convnet_amd", "csrc")
READ_B, PAIR_B, READ_A, PAIR_A, FETCH, WALK_BEGIN, WALK_STEP = 1, 2, 3, 4, 5, 6, 7

DUMP = r'''
#include <cstdio>
#include "wgrad_wide_schedule.h"
template <int NTL> void dump() {
  using S = chip::wgw::Schedule<NTL>;
  constexpr S s{};
  static_assert(chip::wgw::schedule_ok<NTL>(), "rules");
  std::printf("{\"NTL\":%d,\"COLS\":%d,\"G\":%d,\"NSLOT\":%d,\"GB\":%d,\"NA\":%d,\"NB\":%d,\"units\":[", NTL, S::COLS, S::G, S::NSLOT, S::GB, S::NA, S::NB);
  for (int g = 0; g < S::G; ++g) {
    std::printf("%s[", g ? "," : "");
    for (int i = 0; i < s.n[g]; ++i) std::printf("%s[%d,%d,%d,%d]", i ? "," : "", s.u[g][i].kind, s.u[g][i].a, s.u[g][i].b, s.u[g][i].c);
    std::printf("]");
  }
  std::printf("]}\n");
}
int main() { dump<3>(); dump<4>(); }
'''


@pytest.fixture(scope="module")
def schedules(tmp_path_factory):
    d = tmp_path_factory.mktemp("wgw")
    src, exe = d / "dump.cc", d / "dump"
    src.write_text(DUMP)
    subprocess.run(["g++", "-std=c++17", "-I", CSRC, str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    return {s["NTL"]: s for s in map(json.loads, out.strip().splitlines())}


def run_wave(S, A, B, lazy):
    """A [chunks][4 tiles][32 images], B [chunks][NTL tiles][32 images] -> acc [4][NTL] = sum over chunks and images of A*B"""
    NTL, COLS, NSLOT, GB, NA = S["NTL"], S["COLS"], S["NSLOT"], S["GB"], S["NA"]
    nchunks = A.shape[0]
    nan = np.nan
    As = np.full((2, 4, 32), nan)
    Bs = np.full((2, NTL, 32), nan)
    fa = np.full((2, 4, 4, 4), nan)        # [half][tile][pair][4 images]
    fb = np.full((NSLOT, 4, 4), nan)       # [slot][pair][4 images]
    ra = np.full(16, nan)                  # raw pieces of one half of one row: 16 images
    rb = np.full(16, nan)
    acc = np.zeros((4, NTL))
    pending = []

    def images(h, q):                      # the 4 images pair q of half h covers
        return slice(16 * h + 4 * q, 16 * h + 4 * q + 4)

    # the walk: chunk to stage next, chunks left
    st = dict(next=0, left=nchunks, live=False, cur_chunk=None, stage=0)

    def walk_begin():
        st["live"] = st["left"] > 0
        st["cur_chunk"] = st["next"]

    def walk_step():
        st["left"] -= 1
        st["next"] += 1

    def fetch(i):
        stage, live, c = st["stage"], st["live"], st["cur_chunk"]
        if i < NA:
            t, h = i % 4, i // 4
            val = A[c, t, 16 * h:16 * h + 16].copy() if live else np.zeros(16)

            def land(stage=stage, t=t, h=h, val=val):
                As[stage, t, 16 * h:16 * h + 16] = val
        else:
            j = i - NA
            u, h = j % NTL, j // NTL
            val = B[c, u, 16 * h:16 * h + 16].copy() if live else np.zeros(16)

            def land(stage=stage, u=u, h=h, val=val):
                Bs[stage, u, 16 * h:16 * h + 16] = val
        if lazy:
            pending.append(land)
        else:
            land()

    def wait_all():
        for f in pending:
            f()
        pending.clear()

    cur, nxt = 0, 1                        # stage the MFMAs read / the stage being filled

    def unit(x):
        nonlocal ra, rb
        kind, a, b, c = x
        if kind == READ_B:
            col = a % COLS
            h, u = col // NTL, col % NTL
            rb = Bs[nxt if a == COLS else cur, u, 16 * h:16 * h + 16].copy()
        elif kind == PAIR_B:
            fb[a % NSLOT, b] = rb[4 * b:4 * b + 4]
        elif kind == READ_A:
            ra = As[nxt if a == 0 else cur, b, 16 * a:16 * a + 16].copy()
        elif kind == PAIR_A:
            fa[a, b, c] = ra[4 * c:4 * c + 4]
        elif kind == FETCH:
            fetch(a)
        elif kind == WALK_BEGIN:
            walk_begin()
        elif kind == WALK_STEP:
            walk_step()

    # prologue
    walk_begin()
    for i in range(NA + S["NB"]):
        fetch(i)
    walk_step()
    wait_all()
    st["stage"] = 1
    for t in range(4):
        for q in range(4):
            fa[0, t, q] = As[0, t, images(0, q)]
    for q in range(4):
        fb[0, q] = Bs[0, 0, images(0, q)]
    # chunks
    for c in range(nchunks):
        snap = None
        for g in range(S["G"]):
            j, k = divmod(g, 6)
            h, u = j // NTL, j % NTL
            if g == GB:
                wait_all()                 # vmcnt(0), then the barrier
            for x in S["units"][g]:
                unit(x)
            ops = (fa[h].copy(), fb[j % NSLOT].copy())
            if k == 0:
                snap = ops
                acc[:, u] += np.einsum("tqi,qi->t", ops[0], ops[1])
            else:                          # the step's operands are the same for all six products
                assert np.array_equal(snap[0], ops[0], equal_nan=True) and np.array_equal(snap[1], ops[1], equal_nan=True), (c, g)
        cur, nxt = nxt, cur
        st["stage"] ^= 1
    return acc


@pytest.mark.parametrize("lazy", [False, True], ids=["eager", "lazy"])
@pytest.mark.parametrize("nchunks", [1, 2, 5])
@pytest.mark.parametrize("ntl", [3, 4])
def test_chunk_loop_of_the_real_schedule(schedules, ntl, nchunks, lazy):
    S = schedules[ntl]
    rng = np.random.default_rng(7)
    A = rng.standard_normal((nchunks, 4, 32))
    B = rng.standard_normal((nchunks, ntl, 32))
    got = run_wave(S, A, B, lazy)
    ref = np.einsum("cti,cui->tu", A, B)
    assert not np.isnan(got).any()
    assert np.abs(got - ref).max() < 1e-9 * max(1.0, np.abs(ref).max())
