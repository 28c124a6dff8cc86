// wgw_kernel: the weight-gradient outer-GEMM (wg_kernel, gather_gemm.hip; the reference's _convOutpGemm, cudamat_conv_gemm.cu:827-960,
// and dot TN) on a 256 x 256 (or 256 x 192) tile with FOUR waves of 128 x 128 (128 x 96) — one block per CU, one wave per SIMD.
//
// Why (NOTES): wg_kernel's 128 x 128 tile stages 32 KB per 1 536 MFMA cycles and the vector-memory path of a CU moves
// ~10 B/clk: 0.47 of the matrix pipe, the 0.46 measured; its 64 x 64 waves also spend 7.7 VALU per MFMA on the operand split, the
// whole VALU budget of two waves per SIMD.  A 128 x 128 wave tile halves both: 64 KB per 6 144 MFMA cycles = 10.7 B/clk (12.4 for the
// 192-filter tile), ~4 VALU per MFMA.  Same arithmetic as wg_kernel's bf16-split build: same operand layout in LDS (rows of 32 images,
// XOR-swizzled at the source), same k-slot to image map, same six products in split_mac's order — the per-accumulator sums differ from
// wg_kernel's only by the split-K partition.
//
// Structure: no producer wave (512 registers per wave leave no room for a fifth).  A chunk = 32 images of one output pixel; its 2*NTL
// (half, filter tile) steps of 24 MFMAs are cut into sub-steps of four MFMAs (one product over the four row tiles: four independent
// accumulators between two MFMAs on the same one), each fenced and carrying a hand-made share of the other work from a compile-time
// schedule (wgw::Schedule, checked by static_assert): the 16-byte staging loads of the next chunk (its own per-lane address and border
// test, as in wg_kernel) in the first half of the chunk, the split of the next filter column, the split of the second half's four
// row tiles, and — behind the chunk barrier, which sits two steps before the end — the split of the NEXT chunk's first half, so that no
// wave reads the current buffers after the barrier and the next chunk may overwrite them at once.  Two LDS stages of 64 KB.
// NOT YET RUN ON HARDWARE (written at the end of round 4 with the GPU budget spent): opt-in, convnet_hip_set_wgrad_tile(1).  Runs
// correctly in the CPU emulation of this source (tests/test_emulated_kernels.py); the schedule: tests/test_wgrad_wide_cpu.py.
#include <algorithm>
#include <string>

#include "gather_gemm.h"
#include "wgrad_wide_schedule.h"

namespace chip {

template <int NTL>
__global__ __launch_bounds__(256, 1) void wgw_kernel(const WGParams p) {
  using S = wgw::Schedule<NTL>;
  constexpr int WN = 2, MT = 4, NT = 256;
  constexpr int KT = 256, FT = WN * NTL * 32;
  constexpr int A_STAGE = KT * WG_NB, B_STAGE = FT * WG_NB;   // floats: rows of 32 images
  constexpr int NA = S::NA, NB = S::NB, NSLOT = S::NSLOT, COLS = S::COLS;
  static_assert(NA * NT == KT * 8 && NB * NT == FT * 8, "16-byte pieces per thread");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                 // [2][A_STAGE]
  float* Bs = smem + 2 * A_STAGE;   // [2][B_STAGE]

  // tile and K-range of this block: wg_kernel's XCD-aware order
  const int tiles = p.k_tiles * p.f_tiles;
  const int total_blocks = tiles * p.splits;
  const int L = xcd_remap(blockIdx.x, total_blocks);
  if (L >= total_blocks || (int)blockIdx.x >= ((total_blocks + 7) >> 3) * 8) return;
  const int split = L / tiles, tile_id = L - split * tiles;
  const int f_tile = tile_id % p.f_tiles, k_tile = tile_id / p.f_tiles;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int kc0 = k_tile * KT, f0 = f_tile * FT;
  const int N = p.N, SH = p.SH, SW = p.SW;

  // ---- staging slots of this thread (fixed for the whole kernel): wg_kernel's, for the direct-to-LDS path -------------------------
  // piece idx = tid + it*256 of a stage: row idx >> 3, LDS 16-byte slot idx & 7 of the row, holding images piece (idx & 7) ^ ((row >> 1) & 7)
  unsigned a_const[NA], b_const[NB];
  int a_ta[NA], a_tb[NA], a_alt[NA];
  bool a_ok[NA], b_ok[NB];
#pragma unroll
  for (int it = 0; it < NA; ++it) {
    const int idx = tid + it * NT, row = idx >> 3, c4 = idx & 7;
    const int k = kc0 + row;
    a_ok[it] = k < p.K;
    a_alt[it] = (k == p.K && p.bias_dst) ? 32 : 0;   // ones for the bias row, zeros otherwise
    const int kk = a_ok[it] ? k : 0;
    const int ch = kk / p.TYX, tap = kk - ch * p.TYX;
    a_ta[it] = tap / p.TX;
    a_tb[it] = tap - a_ta[it] * p.TX;
    a_const[it] = (unsigned)(ch * SH * SW + a_ta[it] * SW + a_tb[it]) * (unsigned)N + 4u * (unsigned)(c4 ^ ((row >> 1) & 7));
  }
#pragma unroll
  for (int it = 0; it < NB; ++it) {
    const int idx = tid + it * NT, row = idx >> 3, c4 = idx & 7;
    const int f = f0 + row;
    b_ok[it] = f < p.F;
    b_const[it] = (unsigned)(b_ok[it] ? f : 0) * (unsigned)p.M * (unsigned)N + 4u * (unsigned)(c4 ^ ((row >> 1) & 7));
  }

  const int cbeg = split * p.chunks_per_split;
  const int cend = min(cbeg + p.chunks_per_split, p.chunks_total);
  auto sgpr = [](int v) __attribute__((always_inline)) { return __builtin_amdgcn_readfirstlane(v); };

  f32x16 acc[MT][NTL];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int u = 0; u < NTL; ++u)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][u][e] = 0.f;

  if (cend > cbeg) {
    // ---- the (pixel, image chunk) walk of the staging, one chunk ahead: wave-uniform, integer arithmetic only ---------------------
    int w_m = sgpr(cbeg / p.nchunk);
    int w_nc = sgpr(cbeg - w_m * p.nchunk);
    int w_oy = sgpr(w_m / p.GX);
    int w_ox = sgpr(w_m - w_oy * p.GX);
    int left = cend - cbeg;          // chunks not yet staged
    int ysb = 0, xsb = 0;            // source row / column of tap (0, 0) of the chunk being staged
    unsigned ua = 0, ub = 0;         // its uniform offsets into the source / the derivatives
    unsigned live = 0;               // all ones while a chunk is left to stage, else 0: past the end every piece is the zero page
    float* stage_a = As;             // where the chunk being staged goes
    float* stage_b = Bs;
    auto walk_begin = [&]() __attribute__((always_inline)) {
      ysb = w_oy * p.ssy + p.y0;
      xsb = w_ox * p.ssx + p.x0;
      const unsigned nb = (unsigned)w_nc * WG_NB;
      ua = (unsigned)(ysb * SW + xsb) * (unsigned)N + nb;
      ub = (unsigned)w_m * (unsigned)N + nb;
      live = 0u - ((unsigned)(-left) >> 31);
    };
    auto walk_step = [&]() __attribute__((always_inline)) {
      --left;
      const int n1 = w_nc + 1, wp = 1 - (int)((unsigned)(n1 - p.nchunk) >> 31);   // wp = 1: next pixel
      w_nc = n1 - p.nchunk * wp;
      w_m += wp;
      const int x1 = w_ox + wp, wr = 1 - (int)((unsigned)(x1 - p.GX) >> 31);      // wr = 1: next pixel row
      w_ox = x1 - p.GX * wr;
      w_oy += wr;
    };
    auto fetch_piece = [&](auto IT) __attribute__((always_inline)) {
      constexpr int i = decltype(IT)::value;
      if constexpr (i < NA) {
        // (no short-circuit, both addresses computed: a lazily evaluated side comes back as a divergent branch and cuts the sub-step in two)
        const bool ok = a_ok[i] & ((unsigned)(ysb + a_ta[i]) < (unsigned)SH) & ((unsigned)(xsb + a_tb[i]) < (unsigned)SW) & (live != 0u);
        const float* const s1 = p.src + (a_const[i] + ua);
        const float* const s0 = p.zero + a_alt[i];
        const float* src = ok ? s1 : s0;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(stage_a + 4 * (64 * wave + i * NT)), 16, 0, 0);
      } else {
        constexpr int j = i - NA;
        const bool ok = b_ok[j] & (live != 0u);
        const float* const s1 = p.dout + (b_const[j] + ub);
        const float* src = ok ? s1 : p.zero;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(stage_b + 4 * (64 * wave + j * NT)), 16, 0, 0);
      }
    };

    // ---- consumer state ----------------------------------------------------------------------------------------------------------
    // fragment rows of this lane: row tile t of the wave = rows wm*128 + t*32 + li of the A stage, filter tile u = rows wn*NTL*32 + u*32 + li
    // of the B stage; the two 16-byte pieces of half h are LDS slots (2*(2h) + lh) ^ swz and (2*(2h + 1) + lh) ^ swz of the row
    const int swz = (li >> 1) & 7;
    int pofs[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 2; ++i) pofs[h][i] = 4 * ((2 * (2 * h + i) + lh) ^ swz);
    const float* a_cur = As + (wm * MT * 32 + li) * WG_NB;        // this chunk's stage
    const float* b_cur = Bs + (wn * NTL * 32 + li) * WG_NB;
    const float* a_nxt = a_cur + A_STAGE;                          // the stage being filled
    const float* b_nxt = b_cur + B_STAGE;
    Split8 fa[2][MT], fb[NSLOT];
    f32x4 ra0, ra1, rb0, rb1;   // raw pieces between a read unit and its pair units
    auto pair = [&](const f32x4& v0, const f32x4& v1, int q, Split8& f) __attribute__((always_inline)) {
      // pair q of split8 over x = {v0[0..3], v1[0..3]}
      const float x0 = q < 2 ? v0[2 * q] : v1[2 * q - 4], x1 = q < 2 ? v0[2 * q + 1] : v1[2 * q - 3];
      const unsigned H = pk_bf16(x0, x1);
      const float r0 = x0 - __uint_as_float(H << 16), r1 = x1 - __uint_as_float(H & 0xffff0000u);
      const unsigned M = pk_bf16(r0, r1);
      const float s0 = r0 - __uint_as_float(M << 16), s1 = r1 - __uint_as_float(M & 0xffff0000u);
      f.h[q] = H;
      f.m[q] = M;
      f.l[q] = pk_bf16(s0, s1);
    };
    auto do_unit = [&](auto GG, auto II) __attribute__((always_inline)) {
      constexpr wgw::Unit x = wgw::kSchedule<NTL>.u[decltype(GG)::value][decltype(II)::value];
      if constexpr (x.kind == wgw::kReadB) {
        constexpr int col = x.a % COLS, h = col / NTL, u = col % NTL;
        const float* rowp = (x.a == COLS ? b_nxt : b_cur) + u * 32 * WG_NB;
        rb0 = ld4(rowp + pofs[h][0]);
        rb1 = ld4(rowp + pofs[h][1]);
      } else if constexpr (x.kind == wgw::kPairB) {
        pair(rb0, rb1, x.b, fb[x.a % NSLOT]);
      } else if constexpr (x.kind == wgw::kReadA) {
        const float* rowp = (x.a == 0 ? a_nxt : a_cur) + x.b * 32 * WG_NB;
        ra0 = ld4(rowp + pofs[x.a][0]);
        ra1 = ld4(rowp + pofs[x.a][1]);
      } else if constexpr (x.kind == wgw::kPairA) {
        pair(ra0, ra1, x.c, fa[x.a][x.b]);
      } else if constexpr (x.kind == wgw::kFetch) {
        fetch_piece(std::integral_constant<int, x.a>{});
      } else if constexpr (x.kind == wgw::kWalkBegin) {
        walk_begin();
      } else if constexpr (x.kind == wgw::kWalkStep) {
        walk_step();
      }
    };

    // ---- prologue: stage chunk cbeg, split its first half and its column 0 --------------------------------------------------------
    walk_begin();
    static_for<0, NA + NB>([&](auto IT) __attribute__((always_inline)) { fetch_piece(IT); });
    walk_step();
    __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0) lgkmcnt(0)
    __syncthreads();
    stage_a = As + A_STAGE;
    stage_b = Bs + B_STAGE;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const float* rowp = a_cur + t * 32 * WG_NB;
      const f32x4 v0 = ld4(rowp + pofs[0][0]), v1 = ld4(rowp + pofs[0][1]);
#pragma unroll
      for (int q = 0; q < 4; ++q) pair(v0, v1, q, fa[0][t]);
    }
    {
      const f32x4 v0 = ld4(b_cur + pofs[0][0]), v1 = ld4(b_cur + pofs[0][1]);
#pragma unroll
      for (int q = 0; q < 4; ++q) pair(v0, v1, q, fb[0]);
    }

    // ---- the chunks --------------------------------------------------------------------------------------------------------------
    for (int c = cbeg; c < cend; ++c) {
      static_for<0, S::G>([&](auto GG) __attribute__((always_inline)) {
        constexpr int g = decltype(GG)::value, j = g / 6, k = g % 6, h = j / NTL, u = j % NTL;
        if constexpr (g == S::GB) {
          // everything this wave staged has landed; behind the barrier every wave's has, and no wave reads this chunk's stages again
          // (the operands of the two remaining columns are in registers): the next chunk may overwrite them from its first sub-step
          __builtin_amdgcn_s_waitcnt(0x0070);
          __syncthreads();
        }
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, wgw::kSchedule<NTL>.n[g]>([&](auto I) __attribute__((always_inline)) { do_unit(GG, I); });
        const Split8& b = fb[j % NSLOT];
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          const Split8& a = fa[h][t];
          const u32x4& av = k == 0 || k == 4 ? a.m : k == 2 ? a.l : a.h;   // (m,m) (h,l) (l,h) (h,m) (m,h) (h,h): split_mac's order
          const u32x4& bw = k == 0 || k == 3 ? b.m : k == 1 ? b.l : b.h;
          acc[t][u] = mma_bf16(av, bw, acc[t][u]);
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
          __builtin_amdgcn_sched_group_barrier(0x004, 3, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      // the stages change roles
      {
        const float* t0 = a_cur;
        a_cur = a_nxt;
        a_nxt = t0;
        const float* t1 = b_cur;
        b_cur = b_nxt;
        b_nxt = t1;
        const ptrdiff_t da = (ptrdiff_t)A_STAGE - 2 * (stage_a - As), db = (ptrdiff_t)B_STAGE - 2 * (stage_b - Bs);   // 0 <-> STAGE
        stage_a += da;
        stage_b += db;
      }
    }
    __builtin_amdgcn_s_waitcnt(0x0070);   // (the zero-page loads past the end) before the LDS is released
  }

  // ---- write-out --------------------------------------------------------------------------------------------------------------------
  const bool fin = p.splits == 1;
  const int KB = p.K + (p.bias_dst ? 1 : 0);
  float* out = fin ? p.dst : p.partial + (size_t)split * KB * p.F;
  if (p.wide) {
    // As wg_kernel's wide write-out, for the bigger tile: the accumulator layout gives a lane ONE filter column and 16 k-rows per
    // 32 x 32 tile — 256 four-byte stores per lane here, and stores are issue-bound when every wave of the chip stores at once
    // (MI355X_MICROARCH.md: hundreds of cycles per store instruction in a store tail).  Each wave transposes its 128 x FW sub-tile
    // through its quarter of the (now idle) stages in two passes of 64 rows — ds_write_b32 with the lanes along f, ds_read_b128 along
    // f — and stores 16 bytes per lane: 64 stores instead of 256.  One wave's LDS operations execute in order: no barrier inside.
    constexpr int FW = 32 * NTL, Q = FW / 4;                        // sub-tile width in floats / in 16-byte pieces
    constexpr int WS = (2 * (A_STAGE + B_STAGE)) / 4;               // floats of LDS per wave
    static_assert(64 * FW <= WS, "half a sub-tile per pass");
    __syncthreads();                                                 // every wave is done with the stages
    float* ws = smem + wave * WS;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
      for (int tl = 0; tl < 2; ++tl)
#pragma unroll
        for (int u = 0; u < NTL; ++u)
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) ws[(tl * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh) * FW + u * 32 + li] = acc[2 * pass + tl][u][reg];
      CHIP_WAVE_LOCKSTEP();
#pragma unroll
      for (int it = 0; it < Q; ++it) {                               // 64 rows x Q pieces = 64 * Q / 64 lanes
        const int idx = it * 64 + lane, row = idx / Q, c4 = idx - row * Q;
        f32x4 v = ld4(ws + row * FW + 4 * c4);
        const int k = kc0 + wm * 128 + pass * 64 + row, f = f0 + wn * FW + 4 * c4;
        if (k >= KB || f >= p.F) continue;
        float* dp = (fin && k == p.K) ? p.bias_dst + f : out + (size_t)k * p.F + f;
        if (fin) {
          v = v * p.scaleOutput;
          if (p.scaleTargets != 0.f) v = p.scaleTargets * ld4(dp) + v;
        }
        st4(dp, v);
      }
      CHIP_WAVE_LOCKSTEP();                                          // (the next pass overwrites ws)
    }
    return;
  }
  // direct form (a lane holds one filter column and 16 k-rows per tile): filter counts that are not a multiple of four, unaligned targets
#pragma unroll
  for (int u = 0; u < NTL; ++u) {
    const int f = f0 + (wn * NTL + u) * 32 + li;
    if (f >= p.F) continue;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int k = kc0 + (wm * MT + t) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
        if (k >= KB) continue;
        float* dp = (fin && k == p.K) ? p.bias_dst + f : out + (size_t)k * p.F + f;
        float v = acc[t][u][reg];
        if (fin) {
          v *= p.scaleOutput;
          if (p.scaleTargets != 0.f) v = p.scaleTargets * (*dp) + v;
        }
        *dp = v;
      }
    }
  }
}

namespace {

int g_wgrad_tile = -1;
inline int wgrad_tile() {
  if (g_wgrad_tile < 0) g_wgrad_tile = CHIP_KNOB("CONVNET_WG_TILE", 1);
  return g_wgrad_tile;
}

template <int NTL>
void wgw_launch(WGParams& p, const char* op, double flops, double exec) {
  constexpr int KT = 256, FT = 64 * NTL;
  const size_t lds = sizeof(float) * 2 * (KT + FT) * WG_NB;
  static bool once = false;
  if (!once) {
    CHIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(wgw_kernel<NTL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    once = true;
  }
  p.k_tiles = divup(p.K, KT);
  if (p.bias_dst && divup(p.K + 1, KT) != p.k_tiles) p.bias_dst = nullptr;   // no padding row to spare: caller sums separately
  p.f_tiles = divup(p.F, FT);
  p.zero = zero_page();
  p.wide = (p.F % 4 == 0 && (reinterpret_cast<uintptr_t>(p.dst) & 15) == 0 && (!p.bias_dst || (reinterpret_cast<uintptr_t>(p.bias_dst) & 15) == 0)) ? 1 : 0;   // 16-byte write-out
  const int tiles = p.k_tiles * p.f_tiles;
  const size_t total = (size_t)(p.K + (p.bias_dst ? 1 : 0)) * p.F;
  // one full round of resident blocks (ONE per CU), as wg_launch_cfg: floor, not ceil
  constexpr int kBlocks = 256;
  int splits = 1;
  if (tiles < kBlocks) {
    splits = kBlocks / tiles;
    const int max_by_len = p.chunks_total / 16 > 0 ? p.chunks_total / 16 : 1;
    if (splits > max_by_len) splits = max_by_len;
    const size_t max_by_bytes = (size_t(256) << 20) / (total * sizeof(float)) + 1;
    if ((size_t)splits > max_by_bytes) splits = (int)max_by_bytes;
    if (splits < 1) splits = 1;
  }
#ifdef CONVNET_EMU
  if (const char* e = getenv("CONVNET_EMU_WG_SPLITS")) splits = atoi(e);   // tests/emu: reach the single-block-per-tile epilogue at emulation sizes
#endif
  p.chunks_per_split = divup(p.chunks_total, splits);
  splits = divup(p.chunks_total, p.chunks_per_split);
  p.splits = splits;
  const int groups = splits > 64 ? 32 : 1;
  p.partial = splits > 1 ? static_cast<float*>(workspace(sizeof(float) * total * (splits + (groups > 1 ? groups : 0)))) : nullptr;
  dim3 grid(((tiles * splits + 7) / 8) * 8), block(256);
  {
    KernelTimer timer(NTL == 4 ? "wgw_kernel<256x256,split>" : "wgw_kernel<256x192,split>",
                      op, flops, 0.0, exec);
    hipLaunchKernelGGL((wgw_kernel<NTL>), grid, block, lds, stream(), p);
  }
  if (splits > 1) wg_reduce_launch(p, total, splits, groups, op);
}

}  // namespace

// Takes the weight-gradient launch when the wide tile is selected and applies: the bf16-split products (matrix path 1), the 16-byte
// staging path, whole 32-image chunks, at least one full tile of rows and 192 filters, a reduction of >= 64 chunks.  Returns false
// otherwise (wg_kernel runs).
bool wgw_try(WGParams& p, bool vec, bool split_products, const char* op, double flops, double exec) {
  if (!wgrad_tile() || !vec || !split_products) return false;
  if (p.N % WG_NB != 0 || p.K < 256 || p.F < 192) return false;
  if (p.chunks_total < CHIP_DIAG_KNOB("CONVNET_WGW_MIN_CHUNKS", 64)) return false;   // an FC weight gradient: a handful of chunks per 256 x 256 outputs, write-out-bound either way
  const int pad256 = divup(p.F, 256) * 256, pad192 = divup(p.F, 192) * 192;
  if (pad192 < pad256) wgw_launch<3>(p, op, flops, exec);
  else wgw_launch<4>(p, op, flops, exec);
  return true;
}

}  // namespace chip

extern "C" {
void convnet_hip_set_wgrad_tile(int mode) { chip::g_wgrad_tile = mode < 0 ? 0 : mode > 1 ? 1 : mode; }
int convnet_hip_get_wgrad_tile(void) { return chip::wgrad_tile(); }
}
