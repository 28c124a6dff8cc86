// The per-chunk schedule of wgw_kernel (wgrad_wide.hip) and its compile-time checker.  Plain C++17, no HIP: the kernel includes it, and
// tests/test_wgrad_wide_cpu.py compiles it with g++ to run the schedule through a data-level model of the chunk loop.
#pragma once

namespace chip {
namespace wgw {

// ---- the per-chunk schedule ----------------------------------------------------------------------------------------------------
// Units of work beside the MFMAs.  Columns j = 0 .. COLS-1 of a chunk are (half h = j / NTL, filter tile u = j % NTL); step j runs
// column j's 24 MFMAs from fa[h][0..3] and fb[j % NSLOT]; sub-step g = 6*j + k runs product k.  Column COLS means column 0 of the NEXT chunk.
enum Kind : int { kNone = 0, kReadB, kPairB, kReadA, kPairA, kFetch, kWalkBegin, kWalkStep };
struct Unit {
  int kind, a, b, c;   // ReadB(col) PairB(col, q) ReadA(half, tile) PairA(half, tile, q) Fetch(piece)
};
constexpr int kMaxUnits = 5;

// (A variant with the staging loads spread evenly over the chunk instead of issued in its first sub-steps was written for the first
// A/B on hardware and measured the same to 0.3 % — conv2 weight gradients 955 vs 958 us, profiles/r05_wide_kernels.md; gone.)
template <int NTL>
struct Schedule {
  static constexpr int COLS = 2 * NTL, G = 6 * COLS, NSLOT = NTL == 4 ? 4 : 3, GB = 6 * (COLS - 2);   // barrier in front of sub-step GB
  static constexpr int NA = 8, NB = 2 * NTL, NF = NA + NB;
  int n[G] = {};
  Unit u[G][kMaxUnits] = {};
  constexpr void add(int g, Unit x) { u[g][n[g]++] = x; }
  constexpr Schedule() {
    // staging of the next chunk: one piece per sub-step from the start (they need the rest of the chunk to land)
    add(0, Unit{kWalkBegin, 0, 0, 0});
    int last = 0;
    for (int i = 0; i < NF; ++i) {
      last = i;
      add(last, Unit{kFetch, i, 0, 0});
    }
    add(last + 1, Unit{kWalkStep, 0, 0, 0});
    // filter columns: column j + 1 during step j; the last two before the barrier (both in step COLS - 3); the next chunk's column 0 in the last step
    for (int j = 0; j + 3 < COLS; ++j) {
      add(6 * j, Unit{kReadB, j + 1, 0, 0});
      for (int q = 0; q < 4; ++q) add(6 * j + 1 + q, Unit{kPairB, j + 1, q, 0});
    }
    {
      const int g0 = 6 * (COLS - 3);
      add(g0, Unit{kReadB, COLS - 2, 0, 0});
      add(g0 + 1, Unit{kPairB, COLS - 2, 0, 0});
      add(g0 + 1, Unit{kPairB, COLS - 2, 1, 0});
      add(g0 + 2, Unit{kPairB, COLS - 2, 2, 0});
      add(g0 + 2, Unit{kPairB, COLS - 2, 3, 0});
      add(g0 + 3, Unit{kReadB, COLS - 1, 0, 0});
      add(g0 + 4, Unit{kPairB, COLS - 1, 0, 0});
      add(g0 + 4, Unit{kPairB, COLS - 1, 1, 0});
      add(g0 + 5, Unit{kPairB, COLS - 1, 2, 0});
      add(g0 + 5, Unit{kPairB, COLS - 1, 3, 0});
    }
    {
      const int g0 = 6 * (COLS - 1);
      add(g0, Unit{kReadB, COLS, 0, 0});
      for (int q = 0; q < 4; ++q) add(g0 + 1 + q, Unit{kPairB, COLS, q, 0});
    }
    // row tiles of the second half, during the first: a read and four pairs per tile, in sequence, over 6*NTL sub-steps
    {
      int g = 0;
      for (int t = 0; t < 4; ++t) {
        add(g, Unit{kReadA, 1, t, 0});
        const int left = 6 * NTL - (g + 1) - (3 - t) * 3;   // sub-steps this tile's pairs may take (later tiles need >= 3 each)
        if (left >= 4) {
          for (int q = 0; q < 4; ++q) add(g + 1 + q, Unit{kPairA, 1, t, q});
          g += 5;
        } else {
          add(g + 1, Unit{kPairA, 1, t, 0});
          add(g + 1, Unit{kPairA, 1, t, 1});
          add(g + 2, Unit{kPairA, 1, t, 2});
          add(g + 2, Unit{kPairA, 1, t, 3});
          g += 3;
        }
      }
    }
    // row tiles of the NEXT chunk's first half, behind the barrier: three sub-steps per tile
    for (int t = 0; t < 4; ++t) {
      const int g0 = GB + 3 * t;
      add(g0, Unit{kReadA, 0, t, 0});
      add(g0 + 1, Unit{kPairA, 0, t, 0});
      add(g0 + 1, Unit{kPairA, 0, t, 1});
      add(g0 + 2, Unit{kPairA, 0, t, 2});
      add(g0 + 2, Unit{kPairA, 0, t, 3});
    }
  }
};

// The rules the kernel relies on, checked at compile time.
template <int NTL>
constexpr bool schedule_ok() {
  using S = Schedule<NTL>;
  constexpr S s{};
  int readB[S::COLS + 1] = {}, readA[2][4] = {}, pairB[S::COLS + 1][4] = {}, pairA[2][4][4] = {}, fetch[S::NF] = {};
  for (int j = 0; j <= S::COLS; ++j) {
    readB[j] = -1;
    for (int q = 0; q < 4; ++q) pairB[j][q] = -1;
  }
  for (int h = 0; h < 2; ++h)
    for (int t = 0; t < 4; ++t) {
      readA[h][t] = -1;
      for (int q = 0; q < 4; ++q) pairA[h][t][q] = -1;
    }
  for (int i = 0; i < S::NF; ++i) fetch[i] = -1;
  int walk_begin = -1, walk_step = -1;
  int lastReadB = -1, lastReadA = -1;   // one raw register set each: a read may only follow the pairs of the read before it
  for (int g = 0; g < S::G; ++g) {
    if (s.n[g] > kMaxUnits) return false;
    for (int i = 0; i < s.n[g]; ++i) {
      const Unit x = s.u[g][i];
      if (x.kind == kReadB) {
        if (readB[x.a] >= 0) return false;
        if (lastReadB >= 0)
          for (int q = 0; q < 4; ++q)
            if (pairB[lastReadB][q] < 0) return false;   // ... whose pairs are all placed earlier (program order)
        readB[x.a] = g;
        lastReadB = x.a;
      } else if (x.kind == kPairB) {
        if (pairB[x.a][x.b] >= 0 || readB[x.a] < 0 || readB[x.a] >= g || lastReadB != x.a) return false;   // >= 1 sub-step after its read
        pairB[x.a][x.b] = g;
      } else if (x.kind == kReadA) {
        if (readA[x.a][x.b] >= 0) return false;
        if (lastReadA >= 0)
          for (int q = 0; q < 4; ++q)
            if (pairA[lastReadA >> 2][lastReadA & 3][q] < 0) return false;
        readA[x.a][x.b] = g;
        lastReadA = x.a * 4 + x.b;
      } else if (x.kind == kPairA) {
        if (pairA[x.a][x.b][x.c] >= 0 || readA[x.a][x.b] < 0 || readA[x.a][x.b] >= g || lastReadA != x.a * 4 + x.b) return false;
        pairA[x.a][x.b][x.c] = g;
      } else if (x.kind == kFetch) {
        if (fetch[x.a] >= 0 || walk_begin < 0 || walk_step >= 0) return false;
        fetch[x.a] = g;
      } else if (x.kind == kWalkBegin) {
        walk_begin = g;
      } else if (x.kind == kWalkStep) {
        walk_step = g;
      }
    }
  }
  if (walk_begin != 0 || walk_step < 0) return false;
  for (int i = 0; i < S::NF; ++i)
    if (fetch[i] < 0 || fetch[i] >= 6 * NTL) return false;   // staged early: at least the last step in front of the barrier is their time to land
  for (int j = 1; j <= S::COLS; ++j) {
    for (int q = 0; q < 4; ++q) {
      if (pairB[j][q] < 0) return false;
      if (j < S::COLS && pairB[j][q] >= 6 * j) return false;                       // column j split before step j
      if (pairB[j][q] < 6 * (j - S::NSLOT + 1)) return false;                        // ... into a slot whose previous column is done
    }
    // reads of the current buffers in front of the barrier, of the next buffers behind it
    if (j < S::COLS ? readB[j] >= S::GB : readB[j] < S::GB) return false;
  }
  if (S::COLS % S::NSLOT != 0) return false;                                          // the next chunk's column 0 lands in slot 0
  for (int t = 0; t < 4; ++t)
    for (int q = 0; q < 4; ++q) {
      if (pairA[1][t][q] < 0 || pairA[1][t][q] >= 6 * NTL) return false;              // second half ready when its steps start
      if (pairA[0][t][q] < S::GB) return false;                                       // next first half: behind the barrier (and after the first half's last use)
    }
  for (int t = 0; t < 4; ++t)
    if (readA[1][t] >= S::GB || readA[0][t] < S::GB) return false;
  return true;
}
static_assert(schedule_ok<3>() && schedule_ok<4>(), "wgw_kernel: schedule breaks a rule");
template <int NTL>
inline constexpr Schedule<NTL> kSchedule{};

}  // namespace wgw
}  // namespace chip
