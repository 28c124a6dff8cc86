// Implicit-GEMM kernels for the conv_edge / fc_edge hot path on gfx950: fp32 operands, fp32 accumulation, fp32 results.
//
// Everything the reference lowers to im2col + cublasSgemm + scatter (cudamat_conv_gemm.cu:545-960:
// _convUpGemm / _convDownGemm / _convOutpGemm) and cublasSgemm for FC (cudamat.cu:2130-2152) is
// expressed here as two kernel families operating directly on the reference's CHWN layout
// (image index fastest), with no im2col buffer and no atomics:
//
//   gg_kernel   "gather-GEMM":  out[row r, (pixel m, image n)] = sum_k A[r,k] * src[n, tap(m,k)]
//               - fprop  (r = filter,        k = (c,ky,kx), A = W            )   convUpGemm
//               - dgrad  (r = input channel, k = (f,a,b),   A = per-stride-class re-laid W)  convDownGemm
//               - FC fwd / FC dgrad as the 1-pixel case (dot NT / NN)
//   wg_kernel   "outer-GEMM":   dW[k=(c,ky,kx), f] = sum_{m,n} patch(src)[n,k,m] * dout[n,f,m]
//               - conv wgrad (convOutpGemm) and FC wgrad (dot TN), split over (m,n) with a
//                 deterministic second-stage reduce (what the reference's partial_sum was for,
//                 src/conv_edge.cc:191-205).
//
// MFMA mapping (v_mfma_f32_32x32x2_f32: D[32x32] += A[32x2]*B[2x32], lane l holds A[l&31][l>>5],
// B[l>>5][l&31]; D col = l&31, row = (reg&3)+8*(reg>>2)+4*(l>>5)).  The image index n is the
// contiguous dimension of every activation, so it is mapped to the D *column* (lane) dimension:
// loads of 4 consecutive images per lane are one ds_read_b128 / global dwordx4 and stores are
// 512 contiguous bytes per half-wave.  64 FLOP/clk/SIMD = 157.3 TFLOP/s chip peak (fp32 matrix).
//
// Two matrix paths, chosen per launch (matrix_path(), include/convnet_hip.h: convnet_hip_set_matrix_path): the products are formed
// either by that fp32 instruction or — the default — on the bf16 pipe from EXACT three-way operand splits, six
// v_mfma_f32_32x32x16_bf16 per 32x32x16 block (Split8 / split8 / split_mac below; same C/D layout, so tiles, epilogues and launch
// logic are shared).  ggp_kernel is gg_kernel with a producer wave; its split build can read the A operand as pre-split planes.
#include <algorithm>
#include <array>
#include <cmath>
#include <map>
#include <vector>
#include <cstdlib>
#include <string>
#include <type_traits>

#include "common.h"

#include "gather_gemm.h"

namespace chip {

// CW = columns per wave-column — consecutive (pixel, image) columns of the flat column space, GGParams::NP — (128: 4 interleaved 32-column MFMA column tiles per wave, ds_read_b128;
// 64: 2 tiles, ds_read_b64 — used with MT=3 so a 96-row problem (conv1 fprop, conv2 dgrad) fills its tile).
// O3 = the 3-blocks-per-CU build (launch bound 3 waves/SIMD + the k-row-major B stage that makes it fit): chosen by the
// host only for launches with enough tiles to fill >= 2 rounds of 768 slots, where it gains 2-6 %; on ~512-tile launches
// the blocks spread 3/1 over the CUs and it loses, so the 2-block build stays the default.
template <int WR, int WC, int MT, int CW, bool A_KCONTIG, bool VEC, bool O3 = false, bool SPLIT = false>
__global__ __launch_bounds__(WR* WC * 64, (O3 ? 3 : 2)) void gg_kernel(const GGParams pin, const GGClassTable ct) {
  constexpr int NT = WR * WC * 64;
  constexpr int NTC = CW / 32, CW4 = CW / 4;
  using fvec = __attribute__((ext_vector_type(NTC))) float;
  constexpr int ROWS = WR * MT * 32;
  constexpr int APITCH = A_KCONTIG ? (BK + 4) : ROWS;
  constexpr int A_STAGE = A_KCONTIG ? ROWS * APITCH : BK * ROWS;
  constexpr int B_STAGE = WC * BK * CW;
  constexpr int NA = ((A_KCONTIG ? ROWS * (BK / 4) : BK * (ROWS / 4)) + NT - 1) / NT;
  constexpr int NB = (WC * BK * CW4 + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                 // [2][A_STAGE]
  float* Bs = smem + 2 * A_STAGE;   // [2][B_STAGE]

  // fields a stride class overrides live in scalars; everything else is read from the kernarg struct in place
  const GGParams& p = pin;
  GGTile T;
  if (!gg_select_tile(p, ct, T)) return;
  const float* const pA = T.A;
  const int pK = T.K, pGX = T.GX, pG = T.G, pTX = T.TX, pTYX = T.TYX, py0 = T.y0, px0 = T.x0, pdy0 = T.dy0, pdx0 = T.dx0, pncols = T.ncols;
  const int L = T.L, tsplit = T.tsplit;   // tsplit >= 0: this block computes one K-range of a tail tile
  const int row_tile = L % p.row_tiles, col_tile = L / p.row_tiles;
  const int split = blockIdx.y;

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wr = wave / WC, wc = wave % WC;
  const int li = lane & 31, lh = lane >> 5;
  const int r0 = row_tile * ROWS;
  const int N = p.N;
  unsigned long long tr_begin_k = 0;
#ifdef CONVNET_GG_TRACE
  tr_begin_k = trace_clock();
#endif

  // ---- per-thread constants for the B (source) staging slots -----------------------------------
  // VEC (direct-to-LDS) lays the B stage out k-row major, [krow][wave-column][image]: one wave instruction (64 lanes x
  // 16 B) then fills exactly one k-row (or half of one), so the k-row — and with it the whole (channel, tap_y, tap_x)
  // decode — is WAVE-UNIFORM and lives in SGPRs/SALU, while a lane's (wave-column, image quad) is the same for all of
  // its slots.  That removes ~28 VGPRs of per-slot state and most of the staging VALU work.  The scalar path keeps the
  // [wave-column][krow][image] layout.
  constexpr bool KM = O3;
  static_assert(!O3 || VEC, "the 3-block build is a vector-path variant");
  constexpr int KROW_LANES = WC * CW4;   // lanes (16-B pieces) per k-row: 64 or 128
  static_assert(!KM || (NT % KROW_LANES == 0 && (KROW_LANES == 64 || KROW_LANES == 128)), "k-row must align with wave instructions");
  constexpr int BROW = KM ? WC * CW : CW;   // floats between consecutive k-rows of one wave-column in LDS
  int b_ys0[NB], b_xs0[NB], b_n[NB], b_lds[NB], b_krow[NB];
  bool b_ok[NB];
#pragma unroll
  for (int it = 0; it < NB; ++it) {
    const int idx = tid + it * NT;
    const int sub = tid % KROW_LANES;   // KM: the same for every slot of this lane
    const int wcol = KM ? sub / CW4 : idx / (BK * CW4), krow = KM ? 0 : (idx / CW4) % BK, c4 = KM ? sub % CW4 : idx % CW4;
    const int colid = col_tile * WC + wcol;
    const bool ok = idx < WC * BK * CW4 && colid < pncols;
    const int q = colid * CW + 4 * c4;   // flat column (GGParams::NP)
    const int mq = q / p.NP;
    const bool in = ok && mq < pG;
    const int m = in ? mq : 0;
    const int oy = m / pGX, ox = m - oy * pGX;
    b_ys0[it] = oy * p.ssy + py0;
    b_xs0[it] = ox * p.ssx + px0;
    b_n[it] = in ? q - mq * p.NP : N;   // N: out of range for every guard below
    b_ok[it] = in;
    b_krow[it] = krow;
    b_lds[it] = (wcol * BK + krow) * CW + 4 * c4;
  }

  const int cps = tsplit >= 0 ? p.tail_cps : p.chunks_per_split;
  const int kbeg = (tsplit >= 0 ? tsplit : split) * cps * BK;
  int kend = kbeg + cps * BK;
  if (kend > pK) kend = pK;
  const int nchunks = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;

  f32x4 ra[NA], rb[NB];

  // ---- VEC fast path: incremental tap decode + branch-free loads ---------------------------------
  // The slot -> (k-row, column) assignment never changes, so the (channel, tap_y, tap_x) decode of
  // a slot's k is carried from chunk to chunk (k advances by BK = dch*TYX + da*TX + db each chunk,
  // one conditional carry per digit) instead of being re-divided, and out-of-range taps / rows /
  // images read a 16-byte zero page instead of branching.  ~25 VALU per slot per chunk.
  const int TYn = pTYX / pTX;
  const int dch = BK / pTYX, drem = BK - dch * pTYX;
  const int da = drem / pTX, db = drem - da * pTX;
  int s_k[NB], s_ch[NB], s_a[NB], s_b[NB];
  const float* a_ptr[NA];
  int a_k[NA];
  bool a_ok[NA];
  if (VEC) {
#pragma unroll
    for (int it = 0; it < NB; ++it) {
      const int k = kbeg + (KM ? (64 * __builtin_amdgcn_readfirstlane(tid >> 6) + it * NT) / KROW_LANES   // wave-uniform k-row
                               : b_krow[it]);
      const int ch = k / pTYX, tap = k - ch * pTYX;
      s_k[it] = k;
      s_ch[it] = ch;
      s_a[it] = tap / pTX;
      s_b[it] = tap - s_a[it] * pTX;
      b_ok[it] = b_ok[it] && b_n[it] < N;
    }
#pragma unroll
    for (int it = 0; it < NA; ++it) {
      const int idx = tid + it * NT;
      if (!A_KCONTIG) {
        const int krow = idx / (ROWS / 4), c4 = idx % (ROWS / 4);
        const int r = r0 + 4 * c4;
        // the pointer is the whole slot state: rows past R (or lanes past the tile) start AT the end sentinel and only
        // move further, in-range rows cross it exactly when their k reaches kend (see fetch_a_piece)
        // (3-block build only: saves the k counter and the in-range flag; the 64-bit compare costs the 2-block build 1 %)
        a_k[it] = O3 ? 0 : kbeg + krow;
        a_ok[it] = idx < BK * (ROWS / 4) && r < p.R;
        a_ptr[it] = (a_ok[it] || !O3) ? pA + (size_t)p.lda * (kbeg + krow) + r : pA + (size_t)p.lda * kend + p.R;
      } else {
        const int row = idx / (BK / 4), c4 = idx % (BK / 4);
        const int r = r0 + row;
        a_k[it] = kbeg + 4 * c4;
        a_ok[it] = idx < ROWS * (BK / 4) && r < p.R;
        a_ptr[it] = pA + (size_t)p.lda * (a_ok[it] ? r : 0) + a_k[it];
      }
    }
  }
  // Direct-to-LDS staging (global_load_lds_dwordx4): both tiles are stored lane-linear (LDS float offset
  // = 4 * slot index), so one wave-instruction fills 1 KiB = two 512-byte rows with no VGPR round trip, no
  // ds_write pass and nothing to wait for before the stage's closing barrier (whose fence drains vmcnt).
  // Ablation on conv4: the register-staged ds_write pass cost 7 % of the kernel.  Out-of-range lanes read
  // the zero page; lanes past the tile are masked off.  (The k-contiguous A tile is padded, not
  // lane-linear, so it keeps the register path.)
  constexpr bool GLDS_A = VEC && !A_KCONTIG, GLDS_B = VEC;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  // one staging slot each (it must be a compile-time constant after unrolling: the slot state lives in registers)
  // scalars for the staging lambdas (they capture these, not the parameter struct)
  const float* const q_zero = p.zero;
  const float* const q_src = p.src;
  const int q_lda = p.lda, q_dir = p.dir, q_SH = p.SH, q_SW = p.SW;
  const float* const a_end = pA + (size_t)p.lda * kend;   // r-contiguous A: first address of row k = kend
  auto fetch_a_piece = [&](auto IT, int buf) __attribute__((always_inline)) {
    constexpr int it = decltype(IT)::value;
    const float* const ap = a_ptr[it];   // rvalues: a conditional on two lvalues selects an ADDRESS and keeps both in memory
    const bool ok = ((A_KCONTIG || !O3) ? (a_ok[it] && a_k[it] < kend) : ap < a_end);
    const float* src = ok ? ap + 0 : q_zero + 0;
    if (GLDS_A) {
      if (tid + it * NT < BK * (ROWS / 4))
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(As + buf * A_STAGE + 4 * (64 * wave_u + it * NT)), 16, 0, 0);
    } else {
      ra[it] = ld4(src);
    }
    a_k[it] += BK;
    a_ptr[it] += A_KCONTIG ? (size_t)BK : (size_t)q_lda * BK;
  };
  auto fetch_b_piece = [&](auto IT, int buf) __attribute__((always_inline)) {
    constexpr int it = decltype(IT)::value;
    // lane constants are slot-independent in the k-row-major layout (slot 0 holds them); s_* are wave-uniform
    constexpr int lc = KM ? 0 : it;
    const int ys = b_ys0[lc] + q_dir * s_a[it], xs = b_xs0[lc] + q_dir * s_b[it];
    const bool ok = b_ok[lc] && s_k[it] < kend && (unsigned)ys < (unsigned)q_SH && (unsigned)xs < (unsigned)q_SW;
    const unsigned off = (unsigned)((s_ch[it] * q_SH + ys) * q_SW + xs) * (unsigned)N + (unsigned)b_n[lc];
    const float* src = ok ? q_src + off : q_zero;
    if (GLDS_B) {
      if (tid + it * NT < WC * BK * CW4)
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(Bs + buf * B_STAGE + 4 * (64 * wave_u + it * NT)), 16, 0, 0);
    } else {
      rb[it] = ld4(src);
    }
    s_k[it] += BK;
    int b = s_b[it] + db, a = s_a[it] + da, ch = s_ch[it] + dch;
    if (b >= pTX) { b -= pTX; a += 1; }
    if (a >= TYn) { a -= TYn; ch += 1; }
    s_b[it] = b; s_a[it] = a; s_ch[it] = ch;
  };
  auto fetch_vec = [&](int buf) __attribute__((always_inline)) {
    static_for<0, NA>([&](auto IT) __attribute__((always_inline)) { fetch_a_piece(IT, buf); });
    static_for<0, NB>([&](auto IT) __attribute__((always_inline)) { fetch_b_piece(IT, buf); });
  };

  auto fetch = [&](int k0, int buf) __attribute__((always_inline)) {
    if (VEC) {
      fetch_vec(buf);   // stateful: called with k0 = kbeg, kbeg+BK, ... in order
      return;
    }
#pragma unroll
    for (int it = 0; it < NA; ++it) {
      const int idx = tid + it * NT;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (!A_KCONTIG) {
        const int krow = idx / (ROWS / 4), c4 = idx % (ROWS / 4);
        const int k = k0 + krow, r = r0 + 4 * c4;
        if (idx < BK * (ROWS / 4) && k < kend) {
          const float* ap = pA + (size_t)p.lda * k + r;
          if (VEC) {
            if (r < p.R) v = ld4(ap);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (r + e < p.R) v[e] = ap[e];
          }
        }
      } else {
        const int row = idx / (BK / 4), c4 = idx % (BK / 4);
        const int k = k0 + 4 * c4, r = r0 + row;
        if (idx < ROWS * (BK / 4) && r < p.R) {
          const float* ap = pA + (size_t)p.lda * r + k;
          if (VEC) {
            if (k < kend) v = ld4(ap);  // VEC implies K%4==0 and split boundaries %4==0
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (k + e < kend) v[e] = ap[e];
          }
        }
      }
      ra[it] = v;
    }
#pragma unroll
    for (int it = 0; it < NB; ++it) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      const int k = k0 + b_krow[it];
      if (b_ok[it] && k < kend) {
        const int ch = k / pTYX;
        const int tap = k - ch * pTYX;
        const int a = tap / pTX, b = tap - a * pTX;
        const int ys = b_ys0[it] + p.dir * a, xs = b_xs0[it] + p.dir * b;
        if (ys >= 0 && ys < p.SH && xs >= 0 && xs < p.SW) {
          const float* sp = p.src + ((size_t)(ch * p.SH + ys) * p.SW + xs) * N + b_n[it];
          if (VEC) {
            if (b_n[it] < N) v = ld4(sp);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (b_n[it] + e < N) v[e] = sp[e];
          }
        }
      }
      rb[it] = v;
    }
  };

  auto stash = [&](int buf) __attribute__((always_inline)) {
    float* as = As + buf * A_STAGE;
    float* bs = Bs + buf * B_STAGE;
    if (!GLDS_A) {
#pragma unroll
      for (int it = 0; it < NA; ++it) {
        const int idx = tid + it * NT;
        if (!A_KCONTIG) {
          if (idx < BK * (ROWS / 4)) st4(as + 4 * idx, ra[it]);
        } else {
          const int row = idx / (BK / 4), c4 = idx % (BK / 4);
          if (idx < ROWS * (BK / 4)) st4(as + row * APITCH + 4 * c4, ra[it]);
        }
      }
    }
    if (!GLDS_B) {
#pragma unroll
      for (int it = 0; it < NB; ++it) {
        const int idx = tid + it * NT;
        if (idx < WC * BK * CW4) st4(bs + b_lds[it], rb[it]);
      }
    }
  };

  f32x16 acc[MT][NTC];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int u = 0; u < NTC; ++u)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][u][e] = 0.f;

  if (nchunks > 0) {
    fetch(kbeg, 0);
    stash(0);
  }
  __syncthreads();

  // With both tiles staged direct-to-LDS there is no register hand-over, so the next chunk's loads need not be
  // issued as one block: slot i's address arithmetic (~25 VALU) + load goes between the MFMAs of k-step i, where
  // it co-executes with the matrix pipe instead of delaying the chunk's first MFMA.
  // Measured on conv4 (N=256): 126 TFLOP/s with the block fetch, 121 spread, 120 spread + sched_group_barrier
  // interleave — the two resident waves of a SIMD already cover each other's staging phase, so SPREAD stays off.
  constexpr bool SPREAD = false && GLDS_A && GLDS_B;
  const int prio = p.prio & 15;   // 0: no priority changes; 1: MFMA phase high; 2: staging phase high (see GGParams::prio)
  constexpr int NP = NA + NB, PPS = (NP + BK / 2 - 1) / (BK / 2);
  unsigned long long tr_begin = tr_begin_k, tr_loop = 0, tr_stage = 0, tr_mfma = 0, tr_sync = 0, tr_min_m = ~0ull, tr_max_m = 0, tr_min_t = ~0ull, tr_max_t = 0;
#ifdef CONVNET_GG_TRACE
  tr_loop = trace_clock();
#endif
  if (prio == 2) __builtin_amdgcn_s_setprio(2);
  for (int c = 0; c < nchunks; ++c) {
    const int buf = c & 1;
    const bool more = c + 1 < nchunks;
    unsigned long long tr0 = 0, tr1 = 0, tr2 = 0;
    if constexpr (kTrace) tr0 = trace_clock();
    if (!SPREAD && more) fetch(kbeg + (c + 1) * BK, buf ^ 1);
    if constexpr (kTrace) tr1 = trace_clock();
    const float* as = As + buf * A_STAGE;
    const float* bs = Bs + buf * B_STAGE + (KM ? wc * CW : wc * BK * CW) + NTC * li;
    if constexpr (SPLIT) {
      // bf16-split products (Split8 below): the chunk is one MFMA deep; a lane's k-slot j is the LDS k-row the fp32 loop gives it
      // in its j-th MFMA of the chunk (r-contiguous A: 2j + lh; k-contiguous A: 8*(j/4) + 4*lh + j%4), for both operands.
      Split8 fa[MT];
      fvec rb[8];
      if constexpr (!A_KCONTIG) {
        const float* ar = as + wr * MT * 32 + li;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          float x[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) x[j] = ar[(2 * j + lh) * ROWS + t * 32];
          split8(x, fa[t]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) rb[j] = *reinterpret_cast<const fvec*>(bs + (2 * j + lh) * BROW);
      } else {
        const float* ar = as + (wr * MT * 32 + li) * APITCH + 4 * lh;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          const f32x4 v0 = ld4(ar + t * 32 * APITCH), v1 = ld4(ar + t * 32 * APITCH + 8);
          const float x[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
          split8(x, fa[t]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) rb[j] = *reinterpret_cast<const fvec*>(bs + (8 * (j / 4) + 4 * lh + (j & 3)) * BROW);
      }
#pragma unroll
      for (int u = 0; u < NTC; ++u) {
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = rb[j][u];
        Split8 fb;
        split8(x, fb);
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          acc[t][u] = split_mac(fa[t], fb, acc[t][u]);
        }
      }
    } else if (!A_KCONTIG) {
      const float* ar = as + wr * MT * 32 + li;
      // fragments for step kk+1 are requested before the MFMAs of step kk
      float a[2][MT];
      fvec b4[2];
#pragma unroll
      for (int t = 0; t < MT; ++t) a[0][t] = ar[lh * ROWS + t * 32];
      b4[0] = *reinterpret_cast<const fvec*>(bs + lh * BROW);
      if (prio == 1) __builtin_amdgcn_s_setprio(2);
      else if (prio == 2) __builtin_amdgcn_s_setprio(0);
      static_for<0, BK / 2>([&](auto KK) __attribute__((always_inline)) {
        constexpr int kk = decltype(KK)::value;
        constexpr int cur = kk & 1, nxt = cur ^ 1;
        if constexpr (kk + 1 < BK / 2) {
          const int krow = 2 * (kk + 1) + lh;
#pragma unroll
          for (int t = 0; t < MT; ++t) a[nxt][t] = ar[krow * ROWS + t * 32];
          b4[nxt] = *reinterpret_cast<const fvec*>(bs + krow * BROW);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (SPREAD && more) {
          static_for<0, PPS>([&](auto Q) __attribute__((always_inline)) {
            constexpr int pi = kk * PPS + decltype(Q)::value;
            if constexpr (pi < NA) fetch_a_piece(std::integral_constant<int, pi>{}, buf ^ 1);
            else if constexpr (pi < NP) fetch_b_piece(std::integral_constant<int, pi - NA>{}, buf ^ 1);
          });
        }
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
          for (int u = 0; u < NTC; ++u)
            acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][t], b4[cur][u], acc[t][u], 0, 0, 0);
        if (SPREAD) {
          // issue order inside this k-step: one MFMA, then a few of the staging slot's VALU/SALU ops, repeated —
          // the address arithmetic runs while the matrix pipe works instead of before or after the MFMA block
#pragma unroll
          for (int i = 0; i < MT * NTC; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x006, 5, 0);
          }
        }
      });
      if (prio == 1) __builtin_amdgcn_s_setprio(0);
      else if (prio == 2) __builtin_amdgcn_s_setprio(2);
    } else {
      const float* ar = as + (wr * MT * 32 + li) * APITCH + 4 * lh;
#pragma unroll
      for (int q = 0; q < BK / 8; ++q) {
        f32x4 a4[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t) a4[t] = ld4(ar + t * 32 * APITCH + 8 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const fvec b4 = *reinterpret_cast<const fvec*>(bs + (8 * q + 4 * lh + e) * BROW);
#pragma unroll
          for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int u = 0; u < NTC; ++u)
              acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[t][e], b4[u], acc[t][u], 0, 0, 0);
        }
      }
    }
    if constexpr (kTrace) tr2 = trace_clock();
    if (c + 1 < nchunks) stash(buf ^ 1);
    __syncthreads();
    if constexpr (kTrace) {
      const unsigned long long tr3 = trace_clock();
      tr_stage += tr1 - tr0; tr_mfma += tr2 - tr1; tr_sync += tr3 - tr2;
      tr_min_m = min(tr_min_m, tr2 - tr1); tr_max_m = max(tr_max_m, tr2 - tr1);
      tr_min_t = min(tr_min_t, tr3 - tr0); tr_max_t = max(tr_max_t, tr3 - tr0);
    }
  }
  if (prio == 2) __builtin_amdgcn_s_setprio(0);
#ifdef CONVNET_GG_TRACE
  const unsigned long long tr_loop_end = trace_clock();
  auto trace_out = [&](unsigned long long t_end) {
    if (g_gg_trace && lane == 0) {
      unsigned long long* o = g_gg_trace + 16 * ((size_t)blockIdx.x * (NT / 64) + wave);
      o[0] = __builtin_amdgcn_s_getreg((15 << 11) | 4);   // HW_ID low 16 bits: wave slot, SIMD, CU
      o[1] = __builtin_amdgcn_s_getreg((3 << 11) | 20);   // XCC_ID
      o[2] = tr_begin; o[3] = t_end; o[4] = (unsigned long long)nchunks; o[5] = tr_stage; o[6] = tr_mfma; o[7] = tr_sync;
      o[8] = tr_min_m; o[9] = tr_max_m; o[10] = tr_min_t; o[11] = tr_max_t; o[12] = tr_loop; o[13] = tr_loop_end; o[14] = (unsigned long long)L;
    }
  };
#endif

  // ---- epilogue ---------------------------------------------------------------------------------
  if (tsplit >= 0) {   // one K-range of a tail tile: raw sums, register order (gg_tail_fix_kernel finishes the tile)
    float* pp = p.tail_partial + ((size_t)(L - p.tail_first) * p.tail_splits + tsplit) * (size_t)(ROWS * WC * CW);
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        fvec v;
#pragma unroll
        for (int u = 0; u < NTC; ++u) v[u] = acc[t][u][reg];
        *reinterpret_cast<fvec*>(pp + ((size_t)(t * 16 + reg) * NT + tid) * NTC) = v;
      }
    return;
  }
  gg_epilogue<WR, WC, MT, CW, VEC>(p, acc, row_tile, col_tile, split, pncols, pGX, pG, pdy0, pdx0);
#ifdef CONVNET_GG_TRACE
  trace_out(trace_clock());
#endif
}

// -------------------------------------------------------------------------------------------------
// ggp_kernel: gg_kernel's r-contiguous direct-to-LDS build with a PRODUCER wave.
//
// Measured on gg_kernel (profiles/r02_gg_loop_diagnosis.md): with all staging removed the same main loop runs 11-14 % faster; about
// half of that is the staging instructions themselves — ~130 VALU of address arithmetic and 6 LDS-DMA issues per wave per chunk,
// which an in-order wave executes INSTEAD of feeding the matrix pipe — and half is waiting for the loads at the chunk-closing
// vmcnt(0).  Here a fifth wave does all of it: it computes every address, issues all 24 `global_load_lds` of a chunk into a
// THREE-stage LDS ring two chunks ahead, and arrives at the chunk barrier once the NEXT chunk has landed.  The four consumer
// waves run nothing but fragment reads, MFMAs and that one barrier per chunk; they carry no staging state (≈150 VGPRs, so the
// ten waves of the two resident blocks fit 3/3/2/2 on the four SIMDs).
//
// The B stage is k-row major ([krow][wave-column][image], 64 lanes x 16 B = exactly one k-row), so a producer instruction's
// (channel, tap_y, tap_x) is wave-uniform and lives in SGPRs; a lane's (wave-column, image quad) never changes.  Same MFMA order,
// same epilogue, same tile selection as gg_kernel: results are bit-identical to it.
// -------------------------------------------------------------------------------------------------
// APRE (with SPLIT): the A operand arrives ALREADY split — filter_planes_rt_kernel / dgrad_filter_planes_rt_kernel (patch_gemm.hip) write the re-laid filter
// bank as bf16 planes [chunk][row tile][plane h/m/l][k-group lh][row][8 x bf16 = k-slots j of k-rows 2j + lh], so a lane's A fragment of one plane is ONE ds_read_b128
// and a third of the loop's split VALU is gone (the bank is a few MB and is rewritten every call anyway).
template <int WR, int WC, int MT, int CW, bool SPLIT = false, bool APRE = false>
__global__ __launch_bounds__(WR* WC * 64 + 64, ((SPLIT && (MT * (CW / 32) > 6 || (MT * (CW / 32) == 6 && !APRE))) ? 2 : 3)) void ggp_kernel(const GGParams pin, const GGClassTable ct) {
  static_assert(!APRE || SPLIT, "pre-split A is a variant of the bf16-split build");
  constexpr int NC = WR * WC * 64;   // consumer threads
  constexpr int NTC = CW / 32, CW4 = CW / 4;
  using fvec = __attribute__((ext_vector_type(NTC))) float;
  constexpr int ROWS = WR * MT * 32;
  // A stage in floats: 16 k-rows x ROWS fp32, or (APRE) 3 planes x 2 k-groups x ROWS x 16 bytes
  constexpr int A_STAGE = APRE ? 6 * ROWS * 4 : BK * ROWS, B_STAGE = WC * BK * CW, ST = 3;
  constexpr int NA = A_STAGE / 4 / 64, NB = WC * BK * CW4 / 64;   // producer wave-instructions per chunk (16 bytes per lane)
  static_assert(WC * CW4 == 64, "one producer instruction = one k-row of the B stage");
  static_assert((A_STAGE / 4) % 64 == 0 && NB == BK, "whole wave-instructions");
  constexpr int BROW = WC * CW;      // floats between consecutive k-rows of the B stage
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                  // [ST][A_STAGE]
  float* Bs = smem + ST * A_STAGE;   // [ST][B_STAGE]

  const GGParams& p = pin;
  GGTile T;
  if (!gg_select_tile(p, ct, T)) return;
  const int L = T.L, tsplit = T.tsplit;
  const int row_tile = L % p.row_tiles, col_tile = L / p.row_tiles;
  const int split = blockIdx.y;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int r0 = row_tile * ROWS;
  const int N = p.N;

  const int cps = tsplit >= 0 ? p.tail_cps : p.chunks_per_split;
  const int kbeg = (tsplit >= 0 ? tsplit : split) * cps * BK;
  int kend = kbeg + cps * BK;
  if (kend > T.K) kend = T.K;
  int nchunks = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;

  // Border-tap skipping.  In tap-major order a whole run of KC/BK chunks belongs to one tap; when this block owns the whole
  // reduction, the taps whose source pixel lies outside the image for EVERY output pixel of the tile contribute exact zeros and
  // are left out — producer and consumers walk the rectangle [a_lo, a_hi] x [b_lo, b_hi] of taps that exist for some pixel of the
  // tile (a tile is WC*CW/NP consecutive pixels: one at N = 256, eight at N = 32; a lane whose own pixel lacks a tap of the
  // rectangle reads the zero page, as before).  (A dgrad gather otherwise issues 1.13x (conv2) to 1.40x (conv5) the algorithmic
  // MACs on the zero page, a padded 3x3 fprop 1.11x.)  It pays on launches of several rounds; a single round ends with its
  // slowest tile.
  const int TYn = T.TYX / T.TX;
  int a_lo = 0, a_hi = TYn - 1, b_lo = 0, b_hi = T.TX - 1;
  const bool skip = tsplit < 0 && p.splits == 1 && !p.gk;
  if (skip) {
    // pixel range of the tile (row-major): rows oy_f..oy_l; columns ox_f..ox_l when it stays inside one row, else the whole row
    const int m_f = (col_tile * WC * CW) / p.NP, m_l = min(T.G - 1, ((col_tile + 1) * WC * CW - 1) / p.NP);
    const int oy_f = m_f / T.GX, oy_l = m_l / T.GX;
    const int ox_f = oy_f == oy_l ? m_f - oy_f * T.GX : 0, ox_l = oy_f == oy_l ? m_l - oy_l * T.GX : T.GX - 1;
    const int ys_f = oy_f * p.ssy + T.y0, ys_l = oy_l * p.ssy + T.y0, xs_f = ox_f * p.ssx + T.x0, xs_l = ox_l * p.ssx + T.x0;
    if (p.dir > 0) {   // tap a exists for source row ys0 iff 0 <= ys0 + a < SH: union over ys0 in [ys_f, ys_l]
      a_lo = max(0, -ys_l); a_hi = min(TYn - 1, p.SH - 1 - ys_f);
      b_lo = max(0, -xs_l); b_hi = min(T.TX - 1, p.SW - 1 - xs_f);
    } else {           // 0 <= ys0 - a < SH
      a_lo = max(0, ys_f - (p.SH - 1)); a_hi = min(TYn - 1, ys_l);
      b_lo = max(0, xs_f - (p.SW - 1)); b_hi = min(T.TX - 1, xs_l);
    }
    const int na = a_hi - a_lo + 1, nb2 = b_hi - b_lo + 1;
    nchunks = (na > 0 && nb2 > 0) ? na * nb2 * (p.KC / BK) : 0;
  }

  // vmcnt(n) alone (expcnt / lgkmcnt untouched): low four bits in [3:0], high two in [15:14]
  constexpr int kLoads = NA + NB;
  static_assert(kLoads < 64, "vmcnt immediate");
  constexpr int WAIT_ONE_CHUNK_IN_FLIGHT = (kLoads & 15) | ((kLoads >> 4) << 14) | 0x0F70;
  constexpr int WAIT_ALL_LOADS = 0x0F70;

  if (wave == WR * WC) {
    // ================================ producer wave ================================
    __builtin_amdgcn_s_setprio(3);
    // lane constants of the B stage: this lane's (wave-column, image quad) and the source pixel of tap (0,0)
    const int wcol = lane / CW4, c4 = lane % CW4;
    const int colid = col_tile * WC + wcol;
    const int q = colid * CW + 4 * c4;   // flat column (GGParams::NP)
    const int mq = q / p.NP;
    const bool col_ok = colid < T.ncols && mq < T.G;
    const int m = col_ok ? mq : 0;
    const int oy = m / T.GX, ox = m - oy * T.GX;
    const int ys0 = oy * p.ssy + T.y0, xs0 = ox * p.ssx + T.x0;
    const int bn = q - mq * p.NP;
    const bool b_ok = col_ok && bn < N;
    // Reduction order k = (cb*TYX + tap)*BK + c16 (KC % BK == 0): the BK k-rows of a chunk are BK consecutive channels of ONE tap, so
    // every lane's source pixel (and whether it exists) is fixed for the chunk and k-row `it` is a constant plane stride away — the
    // producer spends one 64-bit add per load instead of a (channel, tap_y, tap_x) decode with carries.
    const int TX = T.TX, KC = p.KC, TYX = T.TYX;
    int cb = 0, ta = a_lo, tb = b_lo;   // 16-channel block, tap
    (void)KC;
    if (!skip) {   // a K-range of a split starts inside the chunk sequence (chunk index = cb*TYX + tap)
      const int ci = kbeg / BK;
      cb = ci / TYX;
      const int tap = ci - cb * TYX;
      ta = tap / TX;
      tb = tap - ta * TX;
    }
    const float* const zero = p.zero;
    const float* const src = p.src;
    const int dir = p.dir, SH = p.SH, SW = p.SW, lda = p.lda, R = p.R;
    const unsigned plane_bytes = (unsigned)SH * (unsigned)SW * (unsigned)N * 4u;   // < 2^31 floats per tensor (conv_geo)
    const float* bptr;        // this lane's element of k-row 0 of the next chunk to issue
    unsigned bstride;         // bytes between consecutive k-rows for this lane (0 on the zero page)
    unsigned bvoff;           // this lane's byte offset inside ONE channel plane of the source (fast path below)
    const bool plane_small = (size_t)SH * SW * N < (size_t(1) << 30);   // ... which must fit 32 bits
    bool ball;                // every lane of the wave has the tap: no zero-page lane in this chunk
    auto retap = [&]() __attribute__((always_inline)) {
      const int ys = ys0 + dir * ta, xs = xs0 + dir * tb;
      const bool ok = b_ok && (unsigned)ys < (unsigned)SH && (unsigned)xs < (unsigned)SW;
      const unsigned poff = (unsigned)(ys * SW + xs) * (unsigned)N + (unsigned)bn;
      const unsigned off = (unsigned)(cb * BK * SH * SW) * (unsigned)N + poff;
      bptr = ok ? src + off : zero;
      bstride = ok ? plane_bytes : 0u;
      bvoff = poff * 4u;
      ball = plane_small && __builtin_amdgcn_ballot_w64(ok) == ~0ull;
    };
    retap();
    // A stage: lane-linear [krow][ROWS]; instruction `it` covers 16-byte pieces 64*it .. 64*it+63 of the chunk.
    // APRE: the planes of a (chunk, row tile) are one contiguous run in the order they take in LDS (filter_planes_rt_kernel), so the
    // whole stage is ONE wave-uniform base + 16 bytes per lane, M0 rewritten once per four pieces (the instruction's immediate offset
    // advances source and destination together; rows past R are zeros in the planes).  An LDS-DMA instruction that rewrites M0 and
    // carries a 64-bit per-lane address costs the issuing wave ~2.7x one that does not (tools/dma_issue).
    unsigned a_off[APRE ? 1 : NA];   // byte offset of this lane's piece of instruction `it` from the chunk's first filter row
    bool a_ok[APRE ? 1 : NA];
    if constexpr (!APRE) {
#pragma unroll
      for (int it = 0; it < NA; ++it) {
        const int idx = lane + 64 * it;
        const int krow = idx / (ROWS / 4), q = idx - krow * (ROWS / 4);
        a_ok[it] = r0 + 4 * q < R;
        a_off[it] = (unsigned)(krow * lda + r0 + 4 * q) * 4u;
      }
    }
    const unsigned a_lane = (unsigned)lane * 16u;
    const char* const abase0 = reinterpret_cast<const char*>(T.A) + (APRE ? (size_t)row_tile * (6 * ROWS * 16) : (size_t)0);
    const size_t a_chunk_bytes = APRE ? (size_t)p.row_tiles * (6 * ROWS * 16) : (size_t)lda * (BK * 4);   // chunk index = cb*TYX + tap

#ifdef CONVNET_DIAG
    const int diag = p.prio >> 4;   // timing diagnostics (results wrong): 1 = no source staging after the prologue, 2 = no filter staging after it
#else
    constexpr int diag = 0;
#endif
    int issued = 0;
    auto issue = [&](int stage) __attribute__((always_inline)) {
      const char* abase = abase0 + a_chunk_bytes * (size_t)(cb * TYX + ta * TX + tb);   // wave-uniform
      const bool skip_a = (diag & 2) && issued >= 2, skip_b = (diag & 1) && issued >= 2;
      ++issued;
      if (skip_a) {
      } else if constexpr (APRE) {
        const unsigned lda0 = lds_addr(As + stage * A_STAGE);
#pragma unroll
        for (int it = 0; it < NA; it += 4) {
          if (it + 3 < NA) lds_dma4(a_lane, abase, abase, abase, abase, lda0 + 1024u * it);
          else
#pragma unroll
            for (int j = it; j < NA; ++j) lds_dma1(a_lane, abase + 1024 * (j - it), lda0 + 1024u * j);
          abase += 4096;
        }
      } else {
#pragma unroll
        for (int it = 0; it < NA; ++it) {
          const float* ap = a_ok[it] ? reinterpret_cast<const float*>(abase + a_off[it]) : zero;
          __builtin_amdgcn_global_load_lds((gbl_ptr_t)ap, (lds_ptr_t)(As + stage * A_STAGE + 4 * 64 * it), 16, 0, 0);
        }
      }
      if (skip_b) {
      } else if (ball) {
        // every lane reads real data: k-row `it` is one wave-uniform base (SALU) + this lane's 32-bit offset inside a channel plane,
        // and M0 is rewritten once per four k-rows — the instruction's immediate offset moves the LDS destination by 1 KB per k-row,
        // the uniform base absorbs the same 1 KB on the source side.  No VALU, a quarter of the M0 writes.
        const char* sb = reinterpret_cast<const char*>(src) + (size_t)(cb * BK) * plane_bytes;   // wave-uniform: k-row 0 of the chunk
        const size_t d1 = (size_t)plane_bytes - 1024;
        const unsigned lds0 = lds_addr(Bs + stage * B_STAGE);
#pragma unroll
        for (int it = 0; it < NB; it += 4) {
          lds_dma4(bvoff, sb, sb + d1, sb + 2 * d1, sb + 3 * d1, lds0 + 1024u * it);
          sb += 4 * (size_t)plane_bytes;
        }
      } else {
        const char* bp = reinterpret_cast<const char*>(bptr);
        const unsigned bstep = bstride;
#pragma unroll
        for (int it = 0; it < NB; ++it) {
          __builtin_amdgcn_global_load_lds((gbl_ptr_t)bp, (lds_ptr_t)(Bs + stage * B_STAGE + 4 * 64 * it), 16, 0, 0);
          bp += bstep;
        }
      }
      // next chunk: the next tap of the rectangle for the same 16 channels, then the next channel block.  Taps innermost keeps
      // what neighbouring tiles fetch for tap t+1 one chunk — not one whole tap run — away from what they fetched for tap t.
      // (A/B, same box: channel blocks innermost — a whole run of KC/16 chunks per tap — is 5 % slower on conv2-5 and fetches
      // 1.5x more through the fabric: what a neighbouring tile read for tap t is 16 MB of traffic away when this tile needs it for t+1)
      if (++tb > b_hi) {
        tb = b_lo;
        if (++ta > a_hi) {
          ta = a_lo;
          ++cb;
        }
      }
      retap();
    };

    // Generic k order (GGParams::gk): chunk ci is k-rows 16*ci .. 16*ci + 15 of the bank's own order; each k-row's source is the
    // lane's pixel position plus a per-k-row constant from a table the wave reads with scalar loads.  Tiles whose every lane has
    // every tap inside the image (all but the border pixels) issue the chunk as four lds_dma4 groups; the others per k-row and lane.
    int gci = kbeg / BK;
    typedef const __attribute__((address_space(4))) unsigned* const_u32_ptr_t;   // constant address space: scalar (s_load) reads
    const const_u32_ptr_t ktab = (const_u32_ptr_t)p.ktab;
    const unsigned gvoff = (unsigned)((ys0 * SW + xs0) * N + bn) * 4u;
    const bool gfast = p.gk && __builtin_amdgcn_ballot_w64(b_ok && ys0 >= 0 && ys0 + TYn <= SH && xs0 >= 0 && xs0 + TX <= SW) == ~0ull;
    auto issue_gk = [&](int stage) __attribute__((always_inline)) {
      if constexpr (APRE) {
        const char* abase = abase0 + a_chunk_bytes * (size_t)gci;
        const unsigned lda0 = lds_addr(As + stage * A_STAGE);
#pragma unroll
        for (int it = 0; it < NA; it += 4) {
          if (it + 3 < NA) lds_dma4(a_lane, abase, abase, abase, abase, lda0 + 1024u * it);
          else
#pragma unroll
            for (int j = it; j < NA; ++j) lds_dma1(a_lane, abase + 1024 * (j - it), lda0 + 1024u * j);
          abase += 4096;
        }
      }
      const const_u32_ptr_t tk = ktab + 16 * gci;   // wave-uniform
      const char* const sbase = reinterpret_cast<const char*>(src);
      const unsigned lds0 = lds_addr(Bs + stage * B_STAGE);
      if (gfast) {
#pragma unroll
        for (int it = 0; it < NB; it += 4)
          lds_dma4(gvoff, sbase + (unsigned)__builtin_amdgcn_readfirstlane((int)tk[it]),
                   sbase + (unsigned)__builtin_amdgcn_readfirstlane((int)tk[it + 1]) - 1024,
                   sbase + (unsigned)__builtin_amdgcn_readfirstlane((int)tk[it + 2]) - 2048,
                   sbase + (unsigned)__builtin_amdgcn_readfirstlane((int)tk[it + 3]) - 3072, lds0 + 1024u * it);
      } else {
        const const_u32_ptr_t tt = tk + T.K;   // packed taps
#pragma unroll
        for (int it = 0; it < NB; ++it) {
          const int a = (int)(tt[it] >> 16), b = (int)(tt[it] & 0xffffu);
          const bool ok = b_ok && (unsigned)(ys0 + a) < (unsigned)SH && (unsigned)(xs0 + b) < (unsigned)SW;
          const char* gp = ok ? sbase + ((ptrdiff_t)tk[it] + ((ptrdiff_t)(ys0 * SW + xs0) * N + bn) * 4) : reinterpret_cast<const char*>(zero);
          __builtin_amdgcn_global_load_lds((gbl_ptr_t)gp, (lds_ptr_t)(Bs + stage * B_STAGE + 4 * 64 * it), 16, 0, 0);
        }
      }
      ++gci;
    };
    if (p.gk) {
      if (nchunks > 0) issue_gk(0);
      if (nchunks > 1) issue_gk(1);
      if (nchunks > 1) __builtin_amdgcn_s_waitcnt(WAIT_ONE_CHUNK_IN_FLIGHT); else __builtin_amdgcn_s_waitcnt(WAIT_ALL_LOADS);
      __builtin_amdgcn_s_barrier();
      int fill = 2;
      for (int c = 0; c < nchunks; ++c) {
        const bool more2 = c + 2 < nchunks;
        if (more2) issue_gk(fill);
        fill = fill == ST - 1 ? 0 : fill + 1;
        if (more2) __builtin_amdgcn_s_waitcnt(WAIT_ONE_CHUNK_IN_FLIGHT); else __builtin_amdgcn_s_waitcnt(WAIT_ALL_LOADS);
        __builtin_amdgcn_s_barrier();
      }
      return;
    }
    if (nchunks > 0) issue(0);
    if (nchunks > 1) issue(1);
    if (nchunks > 1) __builtin_amdgcn_s_waitcnt(WAIT_ONE_CHUNK_IN_FLIGHT); else __builtin_amdgcn_s_waitcnt(WAIT_ALL_LOADS);
    __builtin_amdgcn_s_barrier();
    int fill = 2;   // stage of chunk c + 2
    for (int c = 0; c < nchunks; ++c) {
      const bool more2 = c + 2 < nchunks;
      if (more2) issue(fill);
      fill = fill == ST - 1 ? 0 : fill + 1;
      // chunk c+1 must have landed before the consumers are released into it
      if (more2) __builtin_amdgcn_s_waitcnt(WAIT_ONE_CHUNK_IN_FLIGHT); else __builtin_amdgcn_s_waitcnt(WAIT_ALL_LOADS);
      __builtin_amdgcn_s_barrier();
    }
    return;
  }

  // ================================ consumer waves ================================
  const int wr = wave / WC, wc = wave % WC;
  const int li = lane & 31, lh = lane >> 5;
  f32x16 acc[MT][NTC];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int u = 0; u < NTC; ++u)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][u][e] = 0.f;

  __syncthreads();   // chunk 0 has landed (the fences of __syncthreads keep LDS reads on their side of the barrier)
  // (Tried: closing barrier in front of the last k-step with the next chunk's first fragments requested right behind it — 123 vs
  // 130 TFLOP/s on conv4; the compiler's own placement, barrier after the first MFMA of the last k-step, is the better one.)
  int stage = 0;
  if constexpr (SPLIT) {
    // bf16-split products (Split8).  One chunk = one MFMA deep: k-slot (lh, j) of the instruction takes LDS k-row 2j + lh for both
    // operands (any common bijection of the chunk's 16 k-rows onto the 16 k-slots gives the same sum) — the fp32 loop's reads.
    // Software pipeline, because the split is ~5.5 VALU per operand element and VALU only overlaps MFMAs of the SAME wave here
    // (one consumer wave per SIMD): column u's 6*MT MFMAs run with the split of column u+1 in their shadows; the chunk barrier
    // sits in front of the LAST column, whose MFMAs cover the next chunk's A and column-0 reads and splits.  A fragments
    // ping-pong between two register sets over a loop unrolled by two chunks.
    // this chunk's A fragments, split: from fp32 k-rows (read + split here) or, APRE, three b128 reads of the bf16 planes
    auto load_a = [&](int st, Split8 (&fa)[MT]) __attribute__((always_inline)) {
      if constexpr (APRE) {
        const u32x4* ap = reinterpret_cast<const u32x4*>(As + st * A_STAGE) + lh * ROWS + wr * MT * 32 + li;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          fa[t].h = ap[t * 32];
          fa[t].m = ap[2 * ROWS + t * 32];
          fa[t].l = ap[4 * ROWS + t * 32];
        }
      } else {
        const float* ar = As + st * A_STAGE + wr * MT * 32 + li;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          float ra[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) ra[j] = ar[(2 * j + lh) * ROWS + t * 32];
          split8(ra, fa[t]);
        }
      }
    };
    auto read_col = [&](int st, int u, float (&x)[8]) __attribute__((always_inline)) {
      const float* bs = Bs + st * B_STAGE + wc * CW + NTC * li + u;
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = bs[(2 * j + lh) * BROW];
    };
    Split8 fa0[MT], fa1[MT], fb[2];
    float rc[8];   // raw column in flight
    if (nchunks > 0) {
      load_a(0, fa0);
      read_col(0, 0, rc);
      split8(rc, fb[0]);
      read_col(0, 1 % NTC, rc);
    }
    // one chunk: fa = this chunk's A fragments, fan = where the next chunk's go; on entry fb[0] = column 0 split, rc = raw column 1
    auto chunk = [&](Split8 (&fa)[MT], Split8 (&fan)[MT]) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u + 1 < NTC; ++u) {
        __builtin_amdgcn_sched_barrier(0);
        split8(rc, fb[(u + 1) & 1]);
        if (u + 2 < NTC) read_col(stage, u + 2, rc);
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[t][u] = split_mac(fa[t], fb[u & 1], acc[t][u]);
#pragma unroll
        for (int i = 0; i < 6 * MT; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      stage = stage == ST - 1 ? 0 : stage + 1;
      __syncthreads();   // every consumer has read this chunk out of LDS; the producer has the next one landed
      load_a(stage, fan);
      read_col(stage, 0, rc);
      split8(rc, fb[NTC & 1]);
#pragma unroll
      for (int t = 0; t < MT; ++t) acc[t][NTC - 1] = split_mac(fa[t], fb[(NTC - 1) & 1], acc[t][NTC - 1]);
#pragma unroll
      for (int i = 0; i < 6 * MT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, (44 * (APRE ? 1 : MT + 1) + 6 * MT - 1) / (6 * MT), 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      read_col(stage, 1 % NTC, rc);
    };
    static_assert(NTC % 2 == 0, "column parity of fb is carried across chunks");
    // (no condition between the two chunks of an iteration: the compiler would sink the next chunk's splits below it, out of the
    // MFMA shadows; an odd chunk is peeled off in front instead.  The last chunk reads and splits one stage of garbage, unused.)
    int c = 0;
    if (nchunks & 1) {
      chunk(fa0, fa1);
      c = 1;
    } else {
#pragma unroll
      for (int t = 0; t < MT; ++t) fa1[t] = fa0[t];
    }
    for (; c < nchunks; c += 2) {
      chunk(fa1, fa0);
      chunk(fa0, fa1);
    }
  } else
  for (int c = 0; c < nchunks; ++c) {
    const float* ar = As + stage * A_STAGE + wr * MT * 32 + li;
    const float* bs = Bs + stage * B_STAGE + wc * CW + NTC * li;
    float a[2][MT];
    fvec b4[2];
#pragma unroll
    for (int t = 0; t < MT; ++t) a[0][t] = ar[lh * ROWS + t * 32];
    b4[0] = *reinterpret_cast<const fvec*>(bs + lh * BROW);
    static_for<0, BK / 2>([&](auto KK) __attribute__((always_inline)) {
      constexpr int kk = decltype(KK)::value;
      constexpr int cur = kk & 1, nxt = cur ^ 1;
      if constexpr (kk + 1 < BK / 2) {
        const int krow = 2 * (kk + 1) + lh;
#pragma unroll
        for (int t = 0; t < MT; ++t) a[nxt][t] = ar[krow * ROWS + t * 32];
        b4[nxt] = *reinterpret_cast<const fvec*>(bs + krow * BROW);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int u = 0; u < NTC; ++u)
          acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][t], b4[cur][u], acc[t][u], 0, 0, 0);
    });
    stage = stage == ST - 1 ? 0 : stage + 1;
    __syncthreads();   // every consumer is done with this stage; the producer has the next chunk in LDS
  }

  // ---- epilogue: gg_kernel's ---------------------------------------------------------------------
  if (tsplit >= 0) {
    float* pp = p.tail_partial + ((size_t)(L - p.tail_first) * p.tail_splits + tsplit) * (size_t)(ROWS * WC * CW);
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        fvec v;
#pragma unroll
        for (int u = 0; u < NTC; ++u) v[u] = acc[t][u][reg];
        *reinterpret_cast<fvec*>(pp + ((size_t)(t * 16 + reg) * NC + tid) * NTC) = v;
      }
    return;
  }
  gg_epilogue<WR, WC, MT, CW, true>(p, acc, row_tile, col_tile, split, T.ncols, T.GX, T.G, T.dy0, T.dx0);
}

// dst = scaleTargets*dst + sum_s slab[s]  (+bias[row], relu) over a full dst extent.
__global__ void gg_reduce_kernel(float* __restrict__ dst, const float* __restrict__ partial, const float* __restrict__ bias,
                                 size_t total, size_t slab, int splits, size_t per_row, float scaleTargets, int relu,
                                 const float* __restrict__ mask, float post_scale) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += partial[(size_t)k * slab + i];
    if (scaleTargets != 0.f) s = scaleTargets * dst[i] + s;
    if (bias) s += bias[i / per_row];
    if (relu) s = s > 0.f ? s : 0.f;
    if (mask) s = mask[i] > 0.f ? s * post_scale : 0.f;
    dst[i] = s;
  }
}

// Re-lay the filter bank for one stride class of the input-gradient gather:
// Wt[c + C*(b + TXc*(a + TYc*f))] = W[f + F*((cx + s_x*b) + Kx*((cy + s_y*a) + Ky*c))].
// tap_major (ggp_kernel's order): Wt[c + C*(f16 + 16*((b + TXc*a) + TYXc*fb))], f = 16*fb + f16 — reduction index
// k = (fb*TYXc + tap)*16 + f16 instead of f*TYXc + tap.
__global__ void dgrad_filter_kernel(const float* __restrict__ W, float* __restrict__ Wt, int F, int C, int Ky, int Kx, int cy,
                                    int cx, int sy, int sx, int TYc, int TXc, int tap_major) {
  const size_t total = (size_t)C * TXc * TYc * F;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = i % C;
    size_t r = i / C;
    int a, b, f;
    if (tap_major) {
      const int f16 = r % 16;
      r /= 16;
      const int tap = r % (TXc * TYc);
      const int fb = r / (TXc * TYc);
      f = fb * 16 + f16;
      a = tap / TXc;
      b = tap - a * TXc;
    } else {
      b = r % TXc;
      r /= TXc;
      a = r % TYc;
      f = r / TYc;
    }
    Wt[i] = W[(size_t)f + (size_t)F * ((cx + sx * b) + Kx * ((cy + sy * a) + Ky * c))];
  }
}

// Forward filters in ggp_kernel's reduction order: Wt[f + F*(c16 + 16*(tap + TYX*cb))] = W[f + F*(tap + TYX*(16*cb + c16))].
__global__ void filter_tapmajor_kernel(const float* __restrict__ W, float* __restrict__ Wt, int F, int C, int TYX) {
  const size_t total = (size_t)F * C * TYX;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int f = i % F;
    size_t r = i / F;
    const int c16 = r % 16;
    r /= 16;
    const int tap = r % TYX;
    const int c = (int)(r / TYX) * 16 + c16;
    Wt[i] = W[(size_t)f + (size_t)F * (tap + (size_t)TYX * c)];
  }
}

// -------------------------------------------------------------------------------------------------
// wg_kernel: dW[k, f] over (pixel, image) reduction.
// -------------------------------------------------------------------------------------------------
// (WGParams, WG_NB, WG_PITCH: gather_gemm.h — shared with wgw_kernel, wgrad_wide.hip)
// TS = MFMA tile edge: 32 (v_mfma_f32_32x32x2, 16 accumulator registers per tile) or 16 (v_mfma_f32_16x16x4, 4 per
// tile; same FLOP/clk).  The 16-wide tiles let a 160 x 96 problem (conv1: 147 taps x 96 filters) split evenly
// over 2x2 waves (80 x 48 each), which no arrangement of 32-wide tiles can: the 5-wave 32x32 config ran at 70
// TFLOP/s with 10 waves on 4 SIMDs.
template <int WM, int WN, int MT, int NTL, bool VEC, int TS, bool SPLIT = false>
__global__ __launch_bounds__(WM* WN * 64) void wg_kernel(const WGParams p) {
  constexpr int NT = WM * WN * 64;
  constexpr int KT = WM * MT * TS;   // k-columns (D rows) per block
  constexpr int FT = WN * NTL * TS;  // filters (D cols / lanes) per block
  constexpr int LH = 64 / TS;        // k-groups of a wave: lane = li + TS*lh supplies k index lh of each MFMA
  using facc = __attribute__((ext_vector_type(TS == 32 ? 16 : 4))) float;
  // VEC: tiles are staged direct-to-LDS (global_load_lds_dwordx4), which writes lane-linear — rows of exactly 32 floats,
  // no padding.  Bank conflicts of the row-strided ds_read_b128 fragment reads are avoided by an XOR swizzle applied at
  // the SOURCE: LDS 16-byte slot c of row r holds images piece c ^ ((r >> 1) & 7) of that row, so the 16 lanes of a
  // b128 access group (rows with all 16 combinations of (r & 1, (r >> 1) & 7)) hit 16 distinct 4-bank groups.
  constexpr int PITCH = VEC ? WG_NB : WG_PITCH;
  constexpr int A_STAGE = KT * PITCH, B_STAGE = FT * PITCH;
  constexpr int NA = (KT * 8 + NT - 1) / NT, NB = (FT * 8 + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + 2 * A_STAGE;

  // XCD-aware order: each XCD (blocks b%8) owns a contiguous run of (split, tile) pairs with the tile
  // index fastest, so the blocks sharing one L2 walk the SAME pixel range and re-use each other's
  // input / deriv rows from L2 instead of every XCD streaming the whole tensors.
  const int tiles = p.k_tiles * p.f_tiles;
  const int total_blocks = tiles * p.splits;
  const int L = xcd_remap(blockIdx.x, total_blocks);
  if (L >= total_blocks || (int)blockIdx.x >= ((total_blocks + 7) >> 3) * 8) return;
  const int split = L / tiles, tile_id = L - split * tiles;
  const int f_tile = tile_id % p.f_tiles, k_tile = tile_id / p.f_tiles;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane % TS, lh = lane / TS;
  const int kc0 = k_tile * KT, f0 = f_tile * FT;
  const int N = p.N;

  // A slots: one k-column each (fixed for the whole kernel)
  int a_choff[NA], a_ta[NA], a_tb[NA], a_n[NA], a_lds[NA], a_alt[NA];
  bool a_ok[NA];
#pragma unroll
  for (int it = 0; it < NA; ++it) {
    const int idx = tid + it * NT;
    const int row = idx >> 3, c4 = idx & 7;
    const int k = kc0 + row;
    const bool ok = idx < KT * 8 && k < p.K;
    a_alt[it] = (idx < KT * 8 && k == p.K && p.bias_dst) ? 32 : 0;   // ones for the bias row, zeros otherwise
    const int kk = ok ? k : 0;
    const int ch = kk / p.TYX, tap = kk - ch * p.TYX;
    a_ta[it] = tap / p.TX;
    a_tb[it] = tap - a_ta[it] * p.TX;
    a_choff[it] = ch * p.SH * p.SW;
    a_n[it] = 4 * (VEC ? (c4 ^ ((row >> 1) & 7)) : c4);
    a_ok[it] = ok;
    a_lds[it] = row * WG_PITCH + 4 * c4;
  }
  int b_f[NB], b_n[NB], b_lds[NB];
  bool b_ok[NB];
#pragma unroll
  for (int it = 0; it < NB; ++it) {
    const int idx = tid + it * NT;
    const int row = idx >> 3, c4 = idx & 7;
    b_f[it] = f0 + row;
    b_ok[it] = idx < FT * 8 && b_f[it] < p.F;
    b_n[it] = 4 * (VEC ? (c4 ^ ((row >> 1) & 7)) : c4);
    b_lds[it] = row * WG_PITCH + 4 * c4;
  }

  const int cbeg = split * p.chunks_per_split;
  int cend = cbeg + p.chunks_per_split;
  if (cend > p.chunks_total) cend = p.chunks_total;

  f32x4 ra[NA], rb[NB];

  // ---- VEC fast path: the (pixel, image-chunk) walk is carried incrementally (no per-chunk division),
  // every slot's address is  const(slot) + uniform(chunk)  and out-of-range loads hit the zero page.
  unsigned a_const[NA], b_const[NB];
  int w_m = 0, w_nc = 0, w_oy = 0, w_ox = 0;   // wave-uniform walk state
  if (VEC) {
    w_m = cbeg / p.nchunk;
    w_nc = cbeg - w_m * p.nchunk;
    w_oy = w_m / p.GX;
    w_ox = w_m - w_oy * p.GX;
#pragma unroll
    for (int it = 0; it < NA; ++it) a_const[it] = (unsigned)(a_choff[it] + a_ta[it] * p.SW + a_tb[it]) * (unsigned)N + (unsigned)a_n[it];
#pragma unroll
    for (int it = 0; it < NB; ++it) b_const[it] = (unsigned)(b_ok[it] ? b_f[it] : 0) * (unsigned)p.M * (unsigned)N + (unsigned)b_n[it];
  }
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  auto fetch_vec = [&](int buf) {
    const int ysb = w_oy * p.ssy + p.y0, xsb = w_ox * p.ssx + p.x0;
    const int nb = w_nc * WG_NB;
    const unsigned ua = (unsigned)(ysb * p.SW + xsb) * (unsigned)N + (unsigned)nb;
    const unsigned ub = (unsigned)w_m * (unsigned)N + (unsigned)nb;
#pragma unroll
    for (int it = 0; it < NA; ++it) {
      const bool ok = a_ok[it] && (unsigned)(ysb + a_ta[it]) < (unsigned)p.SH && (unsigned)(xsb + a_tb[it]) < (unsigned)p.SW && nb + a_n[it] < N;
      const float* src = ok ? p.src + (a_const[it] + ua) : p.zero + a_alt[it];
      if (tid + it * NT < KT * 8)
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(As + buf * A_STAGE + 4 * (64 * wave_u + it * NT)), 16, 0, 0);
    }
#pragma unroll
    for (int it = 0; it < NB; ++it) {
      const bool ok = b_ok[it] && nb + b_n[it] < N;
      const float* src = ok ? p.dout + (b_const[it] + ub) : p.zero;
      if (tid + it * NT < FT * 8)
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(Bs + buf * B_STAGE + 4 * (64 * wave_u + it * NT)), 16, 0, 0);
    }
    if (++w_nc == p.nchunk) {
      w_nc = 0;
      ++w_m;
      if (++w_ox == p.GX) {
        w_ox = 0;
        ++w_oy;
      }
    }
  };

  auto fetch = [&](int c, int buf) {
    if (VEC) {
      fetch_vec(buf);   // stateful: called for c = cbeg, cbeg+1, ... in order; lands in LDS stage `buf` directly
      return;
    }
    const int m = c / p.nchunk, nc = c - m * p.nchunk;
    const int oy = m / p.GX, ox = m - oy * p.GX;
    const int ysb = oy * p.ssy + p.y0, xsb = ox * p.ssx + p.x0;
    const int nb = nc * WG_NB;
#pragma unroll
    for (int it = 0; it < NA; ++it) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      const int ys = ysb + a_ta[it], xs = xsb + a_tb[it];
      const int n = nb + a_n[it];
      if (a_alt[it]) {
        v = f32x4{1.f, 1.f, 1.f, 1.f};   // images past N meet zero dout columns
      } else if (a_ok[it] && ys >= 0 && ys < p.SH && xs >= 0 && xs < p.SW) {
        const float* sp = p.src + ((size_t)(a_choff[it] + ys * p.SW + xs)) * N + n;
        if (VEC) {
          if (n < N) v = ld4(sp);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n + e < N) v[e] = sp[e];
        }
      }
      ra[it] = v;
    }
#pragma unroll
    for (int it = 0; it < NB; ++it) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      const int n = nb + b_n[it];
      if (b_ok[it]) {
        const float* dp = p.dout + ((size_t)b_f[it] * p.M + m) * N + n;
        if (VEC) {
          if (n < N) v = ld4(dp);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n + e < N) v[e] = dp[e];
        }
      }
      rb[it] = v;
    }
  };
  auto stash = [&](int buf) {
    if (VEC) return;   // already in LDS
    float* as = As + buf * A_STAGE;
    float* bs = Bs + buf * B_STAGE;
#pragma unroll
    for (int it = 0; it < NA; ++it)
      if (tid + it * NT < KT * 8) st4(as + a_lds[it], ra[it]);
#pragma unroll
    for (int it = 0; it < NB; ++it)
      if (tid + it * NT < FT * 8) st4(bs + b_lds[it], rb[it]);
  };

  facc acc[MT][NTL];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int u = 0; u < NTL; ++u)
#pragma unroll
      for (int e = 0; e < (TS == 32 ? 16 : 4); ++e) acc[t][u][e] = 0.f;

  if (cend > cbeg) {
    fetch(cbeg, 0);
    stash(0);
  }
  __syncthreads();
  const int swz = VEC ? ((li >> 1) & 7) : 0;   // (row >> 1) & 7 of every fragment row this lane reads (tile bases are multiples of 16)
  const int prio = p.prio;
  for (int c = cbeg; c < cend; ++c) {
    const int buf = (c - cbeg) & 1;
    if (prio == 2) __builtin_amdgcn_s_setprio(2);
    if (c + 1 < cend) fetch(c + 1, buf ^ 1);
    if (prio == 2) __builtin_amdgcn_s_setprio(0);
    else if (prio == 1) __builtin_amdgcn_s_setprio(2);
    const float* ar = As + buf * A_STAGE + (wm * MT * TS + li) * PITCH;
    const float* br = Bs + buf * B_STAGE + (wn * NTL * TS + li) * PITCH;
    if constexpr (SPLIT) {
      // bf16-split products (see Split8): one MFMA is 8*LH images deep, a lane's 8 k-slots are the two b128 pieces the fp32 loop
      // reads in iterations q = 2h and 2h+1 — the same images for both operands, so the sum is the same.  The stage is walked as
      // H * NTL column steps (half-stage h, filter tile u) of 6*MT MFMAs each; the split of the NEXT step's operands (its B tile,
      // and the A tiles when it opens a new half) is pinned into the shadows of this step's MFMAs.
      constexpr int H = WG_NB / (8 * LH), STEPS = H * NTL;
      auto split_row = [&](const float* rowp, int h, Split8& o) __attribute__((always_inline)) {
        const int piece0 = 4 * ((LH * (2 * h) + lh) ^ swz), piece1 = 4 * ((LH * (2 * h + 1) + lh) ^ swz);
        const f32x4 v0 = ld4(rowp + piece0), v1 = ld4(rowp + piece1);
        const float x[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        split8(x, o);
      };
      Split8 fa[2][MT], fb[2];
#pragma unroll
      for (int t = 0; t < MT; ++t) split_row(ar + t * TS * PITCH, 0, fa[0][t]);
      split_row(br, 0, fb[0]);
      static_for<0, STEPS>([&](auto S) __attribute__((always_inline)) {
        constexpr int s = decltype(S)::value, h = s / NTL, u = s % NTL;
        constexpr bool more = s + 1 < STEPS;
        constexpr int h2 = (s + 1) / NTL, u2 = (s + 1) % NTL;
        constexpr bool new_half = more && u2 == 0;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (new_half) {
#pragma unroll
          for (int t = 0; t < MT; ++t) split_row(ar + t * TS * PITCH, h2, fa[h2 & 1][t]);
        }
        if constexpr (more) split_row(br + u2 * TS * PITCH, h2, fb[(s + 1) & 1]);
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[t][u] = split_mac(fa[h & 1][t], fb[s & 1], acc[t][u]);
        if constexpr (more) {
          constexpr int valu = 46 * (new_half ? MT + 1 : 1), per = (valu + 6 * MT - 1) / (6 * MT);
#pragma unroll
          for (int i = 0; i < 6 * MT; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, per, 0);
          }
        }
      });
      __builtin_amdgcn_sched_barrier(0);
    } else
#pragma unroll
    for (int q = 0; q < WG_NB / (4 * LH); ++q) {   // one b128 per lane = 4*LH images of the stage
      const int piece = 4 * ((LH * q + lh) ^ swz);
      f32x4 a4[MT], b4[NTL];
#pragma unroll
      for (int t = 0; t < MT; ++t) a4[t] = ld4(ar + t * TS * PITCH + piece);
#pragma unroll
      for (int u = 0; u < NTL; ++u) b4[u] = ld4(br + u * TS * PITCH + piece);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
          for (int u = 0; u < NTL; ++u) {
            if constexpr (TS == 32)
              acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[t][e], b4[u][e], acc[t][u], 0, 0, 0);
            else
              acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[t][e], b4[u][e], acc[t][u], 0, 0, 0);
          }
    }
    if (prio == 1) __builtin_amdgcn_s_setprio(0);
    if (c + 1 < cend) stash(buf ^ 1);
    __syncthreads();
  }

  const bool fin = p.splits == 1;
  const int KB = p.K + (p.bias_dst ? 1 : 0);
  float* out = fin ? p.dst : p.partial + (size_t)split * KB * p.F;
  if constexpr (TS == 32 && WM == 2 && WN == 2 && MT == 2 && NTL == 2) {
    if (p.wide) {
      // Wide write-out.  The accumulator layout gives a lane ONE filter column and 16 k-rows per tile, so the direct write-out is 64
      // four-byte stores per lane, 128 contiguous bytes per half-wave.  Each wave transposes its 64 x 64 sub-tile through its quarter
      // of the (now idle) staging buffers — ds_write_b32 with the lanes along f, ds_read_b128 along f — and stores 16 bytes per lane,
      // four 256-byte rows per instruction: 16 stores instead of 64.  LDS ops of one wave execute in order: no barrier.  Measured:
      // fc6 wgrad 157.6 -> 140.0 us at N = 256; at 32 images per GPU (one 32-image chunk of MFMA work per 64 KB tile) the conv and FC
      // weight-gradient kernels lose 14-15 %.
      float* ws = smem + wave * 4096;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) ws[(t * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh) * 64 + u * 32 + li] = acc[t][u][reg];
      CHIP_WAVE_LOCKSTEP();
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int idx = it * 64 + lane, row = idx >> 4, c4 = idx & 15;
        f32x4 v = ld4(ws + row * 64 + 4 * c4);
        const int k = kc0 + wm * 64 + row, f = f0 + wn * 64 + 4 * c4;
        if (k >= KB || f >= p.F) continue;
        float* dp = (fin && k == p.K) ? p.bias_dst + f : out + (size_t)k * p.F + f;
        if (fin) {
          v = v * p.scaleOutput;
          if (p.scaleTargets != 0.f) v = p.scaleTargets * ld4(dp) + v;
        }
        st4(dp, v);
      }
      return;
    }
  }
#pragma unroll
  for (int u = 0; u < NTL; ++u) {
    const int f = f0 + (wn * NTL + u) * TS + li;
    if (f >= p.F) continue;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
#pragma unroll
      for (int reg = 0; reg < (TS == 32 ? 16 : 4); ++reg) {
        // D row held by (lane group lh, register reg): 32x32: (reg&3) + 8*(reg>>2) + 4*lh; 16x16: 4*lh + reg
        const int k = kc0 + (wm * MT + t) * TS + (TS == 32 ? (reg & 3) + 8 * (reg >> 2) + 4 * lh : 4 * lh + reg);
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

// first level of a two-level slab reduce (many splits, few elements: conv1's 512 x 14 112): group g sums
// its `per` consecutive slabs into stage[g]; the plain reduce then finishes over the groups.
__global__ void wg_reduce_group_kernel(float* __restrict__ stage, const float* __restrict__ partial, size_t total, int splits, int per) {
  const int g = blockIdx.y;
  const int k0 = g * per, k1 = min(splits, k0 + per);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int k = k0; k < k1; ++k) s += partial[(size_t)k * total + i];
    stage[(size_t)g * total + i] = s;
  }
}

// elements [0, main) go to dst, [main, total) to dst2 (the fused bias-gradient row; main == total without it)
__global__ void wg_reduce_kernel(float* __restrict__ dst, float* __restrict__ dst2, const float* __restrict__ partial, size_t total,
                                 size_t main, int splits, float scaleTargets, float scaleOutput) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += partial[(size_t)k * total + i];
    s *= scaleOutput;
    float* d = i < main ? dst + i : dst2 + (i - main);
    if (scaleTargets != 0.f) s = scaleTargets * (*d) + s;
    *d = s;
  }
}

// -------------------------------------------------------------------------------------------------
// dot() outside the three fc_edge.cc shapes: target = beta*target + alpha*op(mat1)*op(mat2) for ANY transpose combination and
// any alpha (the full contract of cudamat.cu:2130-2152, which hands all four cases to cublasSgemm).  Off the hot path — T,T and
// alpha != 1 on the N,x side never occur in the training step — so this is a plain 32x32 LDS-tiled fp32 FMA kernel, not MFMA.
// op1(i,l) = t1 ? mat1[l + ld1*i] : mat1[i + ld1*l];  op2(l,j) = t2 ? mat2[j + ld2*l] : mat2[l + ld2*j]   (column-major).
// -------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) dot_generic_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ c, int m, int n,
                                                          int k, int ld1, int ld2, int t1, int t2, float beta, float alpha) {
  __shared__ float As[32][33], Bs[32][33];   // As[l][i], Bs[l][j]
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  const int i0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int l0 = 0; l0 < k; l0 += 32) {
    for (int r = ty; r < 32; r += 8) {
      // tile element (l = l0 + r or l0 + tx, chosen so the global read is contiguous in tx)
      if (t1) {   // contiguous in l
        const int i = i0 + r, l = l0 + tx;
        As[tx][r] = (i < m && l < k) ? a[(size_t)l + (size_t)ld1 * i] : 0.f;
      } else {    // contiguous in i
        const int i = i0 + tx, l = l0 + r;
        As[r][tx] = (i < m && l < k) ? a[(size_t)i + (size_t)ld1 * l] : 0.f;
      }
      if (t2) {   // contiguous in j
        const int j = j0 + tx, l = l0 + r;
        Bs[r][tx] = (j < n && l < k) ? b[(size_t)j + (size_t)ld2 * l] : 0.f;
      } else {    // contiguous in l
        const int j = j0 + r, l = l0 + tx;
        Bs[tx][r] = (j < n && l < k) ? b[(size_t)l + (size_t)ld2 * j] : 0.f;
      }
    }
    __syncthreads();
#pragma unroll 8
    for (int l = 0; l < 32; ++l) {
      const float av = As[l][tx];
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = fmaf(av, Bs[l][ty + 8 * q], acc[q]);
    }
    __syncthreads();
  }
  const int i = i0 + tx;
  if (i < m) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int j = j0 + ty + 8 * q;
      if (j < n) {
        float* d = c + (size_t)i + (size_t)m * j;
        *d = (beta != 0.f ? beta * (*d) : 0.f) + alpha * acc[q];
      }
    }
  }
}

// -------------------------------------------------------------------------------------------------
// host-side dispatch
// -------------------------------------------------------------------------------------------------
namespace {

constexpr int kTargetBlocks = 512;  // ~2 resident blocks per CU

// tag + algorithmic flops of the call being dispatched (consumed by KernelTimer in the launchers)
const char* t_op = "";
double t_flops = 0.0;
double t_exec = 0.0;   // MFMA work the launch issues when it differs from the algorithmic t_flops (0 = same)

// Issue-priority scheme of gg_kernel's main loop (CONVNET_GG_PRIO overrides, for A/B runs).  Measured per layer (AlexNet, N=256):
// raising the STAGING phase (2) beats raising the MFMA phase (1, round 1's choice) and no priority (0) by ~1 % — the co-resident
// wave's address arithmetic and LDS-DMA issue finish sooner and its MFMA phase starts earlier.
inline int gg_prio_mode() {
  return CHIP_DIAG_KNOB("CONVNET_GG_PRIO", 2);
}

// The producer-wave build of the gather-GEMM (ggp_kernel) for the launches that have one (r-contiguous A, vector path,
// 64 pieces per k-row).  On by default; CONVNET_GG_PRODUCER=0 restores gg_kernel everywhere (A/B runs).
inline bool gg_producer_mode() {
  return CHIP_DIAG_KNOB("CONVNET_GG_PRODUCER", 1) != 0;
}

// Which matrix instruction the GEMM kernels form their products with (matrix_path(), csrc/state.hip): 1 = bf16-split (default;
// Split8 above), 0 = v_mfma_f32_32x32x2_f32.  CONVNET_GG_SPLIT=0/1 sets the initial value, convnet_hip_set_matrix_path() changes it
// at run time (tests and bench.py run both); CONVNET_WG_SPLIT overrides it for wg_kernel alone (A/B runs).
inline bool gg_split_mode() { return matrix_path() != 0; }
inline bool wg_split_mode() {
  const int v = CHIP_DIAG_KNOB("CONVNET_WG_SPLIT", -1);
  return v < 0 ? gg_split_mode() : v != 0;
}

// ggp_kernel exists for the tile shapes whose B stage has 64 sixteen-byte pieces per k-row (gg_run picks those for R > 32).
inline bool ggp_shape_ok(int R, int KC) { return gg_producer_mode() && R > 32 && KC > 0 && KC % BK == 0 && !CHIP_DIAG_KNOB("CONVNET_GG_ROWS64", 0); }

template <typename Kern>
int resident_slots(Kern kern, int threads, size_t lds) {
  int n = 0;
  CHIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(kern), threads, lds));
  if (n < 1) n = 1;
  if (CHIP_DIAG_KNOB("CONVNET_GG_VERBOSE", 0)) fprintf(stderr, "libconvnet_hip: %d resident blocks per CU (%d threads, %zu B LDS)\n", n, threads, lds);
  return n * 256;   // 256 CUs
}

inline int wg_prio_mode() {
  return CHIP_DIAG_KNOB("CONVNET_WG_PRIO", 2);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <typename Kern>
void allow_big_lds(Kern kern, size_t lds) {
  // >64 KiB of dynamic LDS needs an explicit opt-in once per kernel.
  if (lds > 64 * 1024) CHIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
}

// Launch one gather-GEMM.  `dst_elems` > 0 means the launch covers the whole destination matrix,
// which makes a split-K (slab per split + deterministic reduce) legal.
static const GGClassTable kNoClasses = {};

// One launch over every stride class of a strided dgrad.  `cls` carries A, K, GX, G, TX, TYX, y0, x0, dy0, dx0
// per class; ncols / col_tiles / tile_end are filled here for the chosen tile shape.
template <int WR, int WC, int MT, int CW>
void gg_launch_classes(GGParams& p, GGClassTable& ct, bool vec) {
  constexpr int ROWS = WR * MT * 32;
  constexpr int A_STAGE = BK * ROWS, B_STAGE = WC * BK * CW;
  const size_t lds = sizeof(float) * 2 * (A_STAGE + B_STAGE);
  p.NP = vec ? p.N : divup(p.N, CW) * CW;
  p.row_tiles = divup(p.R, ROWS);
  p.zero = zero_page();
  p.prio = gg_prio_mode() | (CHIP_DIAG_KNOB("CONVNET_GGP_DIAG", 0) << 4);
  p.splits = 1;
  p.chunks_per_split = 1 << 24;
  p.partial = nullptr;
  p.slab = 0;
  std::sort(ct.c, ct.c + ct.n, [](const GGClass& a, const GGClass& b) { return a.K > b.K; });
  int end = 0;
  for (int i = 0; i < ct.n; ++i) {
    ct.c[i].ncols = divup(ct.c[i].G * p.NP, CW);
    ct.c[i].col_tiles = divup(ct.c[i].ncols, WC);
    end += p.row_tiles * ct.c[i].col_tiles;
    ct.c[i].tile_end = end;
  }
  static const std::string kname_g = "gg_kernel<" + std::to_string(WR) + "," + std::to_string(WC) + "," + std::to_string(MT) + "," + std::to_string(CW) + ",rc>";
  static const std::string kname_p = "ggp_kernel<" + std::to_string(WR) + "," + std::to_string(WC) + "," + std::to_string(MT) + "," + std::to_string(CW) + ">";
  static const std::string kname_s0 = kname_p.substr(0, kname_p.size() - 1) + ",split>";
  static const std::string kname_s1 = kname_p.substr(0, kname_p.size() - 1) + ",split,pre>";   // A read as pre-split bf16 planes
  const std::string& kname_s = p.apre ? kname_s1 : kname_s0;
  static const std::string kname_gs = kname_g.substr(0, kname_g.size() - 1) + ",split>";
  const std::string& kname = p.KC > 0 ? (gg_split_mode() ? kname_s : kname_p) : ((vec && gg_split_mode()) ? kname_gs : kname_g);
  KernelTimer timer(kname.c_str(), t_op, t_flops, 0.0, p.KC > 0 ? 0.0 : t_exec);
  dim3 grid(end), block(WR * WC * 64);
  if (p.KC > 0) {
    CHIP_REQUIRE(vec && WC * (CW / 4) == 64);
    if constexpr (WC * (CW / 4) == 64) {
      const size_t lds3 = lds / 2 * 3;
      if (gg_split_mode() && p.apre) {
        const size_t lds3p = sizeof(float) * 3 * (6 * ROWS * 4 + B_STAGE);
        allow_big_lds(ggp_kernel<WR, WC, MT, CW, true, true>, lds3p);
        hipLaunchKernelGGL((ggp_kernel<WR, WC, MT, CW, true, true>), grid, dim3(WR * WC * 64 + 64), lds3p, stream(), p, ct);
      } else if (gg_split_mode()) {
        allow_big_lds(ggp_kernel<WR, WC, MT, CW, true>, lds3);
        hipLaunchKernelGGL((ggp_kernel<WR, WC, MT, CW, true>), grid, dim3(WR * WC * 64 + 64), lds3, stream(), p, ct);
      } else {
        allow_big_lds(ggp_kernel<WR, WC, MT, CW>, lds3);
        hipLaunchKernelGGL((ggp_kernel<WR, WC, MT, CW>), grid, dim3(WR * WC * 64 + 64), lds3, stream(), p, ct);
      }
    }
  } else if (vec && gg_split_mode()) {
    allow_big_lds(gg_kernel<WR, WC, MT, CW, false, true, false, true>, lds);
    hipLaunchKernelGGL((gg_kernel<WR, WC, MT, CW, false, true, false, true>), grid, block, lds, stream(), p, ct);
  } else if (vec) {
    allow_big_lds(gg_kernel<WR, WC, MT, CW, false, true>, lds);
    hipLaunchKernelGGL((gg_kernel<WR, WC, MT, CW, false, true>), grid, block, lds, stream(), p, ct);
  } else {
    allow_big_lds(gg_kernel<WR, WC, MT, CW, false, false>, lds);
    hipLaunchKernelGGL((gg_kernel<WR, WC, MT, CW, false, false>), grid, block, lds, stream(), p, ct);
  }
}

template <int WR, int WC, int MT, int CW, bool AK>
void gg_launch_cfg(GGParams& p, bool vec, size_t dst_elems) {
  constexpr int ROWS = WR * MT * 32;
  constexpr int A_STAGE = AK ? ROWS * (BK + 4) : BK * ROWS;
  constexpr int B_STAGE = WC * BK * CW;
  p.NP = vec ? p.N : divup(p.N, CW) * CW;
  p.ncols = divup(p.G * p.NP, CW);
  // CONVNET_GG_LDS_PAD (diagnostic): extra dynamic LDS per block, e.g. 65536 to force ONE resident block per CU
  const size_t lds_pad = (size_t)CHIP_DIAG_KNOB("CONVNET_GG_LDS_PAD", 0);
  const size_t lds = sizeof(float) * 2 * (A_STAGE + B_STAGE) + lds_pad;
  p.row_tiles = divup(p.R, ROWS);
  p.col_tiles = divup(p.ncols, WC);
  p.zero = zero_page();
  p.prio = gg_prio_mode() | (CHIP_DIAG_KNOB("CONVNET_GGP_DIAG", 0) << 4);
  const int tiles = p.row_tiles * p.col_tiles;
  const int kchunks = divup(p.K, BK);
  // 3 blocks per CU (768 slots) for launches with at least two such rounds of tiles; only the 128-row r-contiguous
  // vector build has the variant (it is the one whose register count sits between the 2- and 3-block limits).
  const bool no_o3 = CHIP_DIAG_KNOB("CONVNET_GG_NO_O3", 0) != 0;
  const bool o3 = !no_o3 && !gg_split_mode() && vec && !AK && WR == 2 && WC == 2 && MT == 2 && CW == 128 && tiles >= 2 * 768 && p.KC == 0;
  int slots_launch = o3 ? 768 : kTargetBlocks;
  if constexpr (!AK && WC * (CW / 4) == 64) {
    if (p.KC > 0) {
      static const int pslots_s = resident_slots(ggp_kernel<WR, WC, MT, CW, true>, WR * WC * 64 + 64, sizeof(float) * 3 * (A_STAGE + B_STAGE));
      static const int pslots_p = resident_slots(ggp_kernel<WR, WC, MT, CW, true, true>, WR * WC * 64 + 64, sizeof(float) * 3 * (6 * ROWS * 4 + B_STAGE));
      static const int pslots_f = resident_slots(ggp_kernel<WR, WC, MT, CW>, WR * WC * 64 + 64, sizeof(float) * 3 * (A_STAGE + B_STAGE));
      slots_launch = gg_split_mode() ? (p.apre ? pslots_p : pslots_s) : pslots_f;
    }
  }
  // FLOP/s of one resident block, for the two estimates below: fp32 MFMA = half a CU (64 FLOP/clk/SIMD, 2 blocks per CU) at 80 %;
  // bf16-split = the measured ~190 TFLOP/s-equivalent of a full chip over its resident blocks.
  const double block_rate = gg_split_mode() ? 190e12 / slots_launch : 64.0 * 4 * 2.2e9 * 0.8 / 2;
  // Split-K factor by wave quantisation: every block of a launch takes the same time, so a grid of b
  // blocks on `slots` resident-block slots runs ceil(b/slots) rounds and wastes the empty part of the
  // last one (338 tiles on 512 slots = 66 % busy; 3 K-splits = 1014 blocks = 99 %).  Pick the split with
  // the best estimated time = rounds * (work per block) + slab-reduce traffic; needs a launch that owns
  // the whole destination (dst_elems > 0).
  int splits = 1;
  int tail_s_instead = 0;   // > 0: split-K was cancelled in favour of a tail split with this many K-ranges (the choice below must honour it)
  if (dst_elems > 0 && kchunks >= 16) {
    const double slots = slots_launch;
    const double flops = 2.0 * ROWS * (WC * (double)CW) * (double)p.K;           // per tile, all K
    double best_t = 1e30;
    for (int sp = 1; sp <= 16 && kchunks / sp >= 8; ++sp) {
      const double rounds = std::ceil(tiles * (double)sp / slots);
      double t = rounds * (flops / sp) / block_rate;
      if (sp > 1) t += sizeof(float) * (double)dst_elems * (2.0 * sp + 1) / 4.0e12 + 4e-6;
      if (t < best_t * 0.97) {
        best_t = t;
        splits = sp;
      }
    }
    // ... against the tail split below (whole-K blocks for the full rounds, only the last round's tiles cut): the same round count
    // with a fix-up over a few tiles instead of a reduce over the whole destination (conv3 dgrad: 338 tiles on 256 slots, 3 slabs of
    // 44 MB reduced in 35 us vs 82 tail tiles fixed in ~10)
    if (splits > 1 && tiles > slots_launch && tiles % slots_launch != 0 && kchunks >= 32 && !CHIP_DIAG_KNOB("CONVNET_GG_NO_TAIL_SPLIT", 0)) {
      const int full = (tiles / slots_launch) * slots_launch, rem = tiles - full;
      const double t_round = flops / block_rate, tile_bytes = sizeof(float) * (double)ROWS * WC * CW;
      for (int s = 2; s <= 8 && kchunks / s >= 8; ++s) {
        const double t = (full / slots + std::ceil(rem * (double)s / slots) / s) * t_round + rem * (s + 1.0) * tile_bytes / 4.0e12 + 6e-6;
        if (t < best_t) {
          splits = 1;
          tail_s_instead = s;
          break;
        }
      }
    }
  }
  p.chunks_per_split = kchunks > 0 ? divup(kchunks, splits) : 1;
  splits = kchunks > 0 ? divup(kchunks, p.chunks_per_split) : 1;
  p.splits = splits;
  p.slab = dst_elems;
  p.partial = splits > 1 ? static_cast<float*>(workspace(sizeof(float) * dst_elems * splits)) : nullptr;
  // Tail split: more tiles than slots and a partial last round (conv2 fprop: 1352 tiles = 2.64 rounds of 512).  Cut only
  // the last round's tiles into s K-ranges so that round is full as well: 2 + ceil(328*3/512)/3 = 2.67 rounds instead of 3.
  p.tail_splits = 1;
  p.tail_partial = nullptr;
  const int slots = slots_launch;
  const bool no_tail = CHIP_DIAG_KNOB("CONVNET_GG_NO_TAIL_SPLIT", 0) != 0;
  if (!no_tail && splits == 1 && dst_elems > 0 && tiles > slots && tiles % slots != 0 && kchunks >= 32) {
    const int full = (tiles / slots) * slots, rem = tiles - full;
    const double tile_bytes = sizeof(float) * (double)ROWS * WC * CW;
    const double t_round = 2.0 * ROWS * (WC * (double)CW) * (double)p.K / block_rate;   // one whole-K block
    double best = 0.95;   // cost of the last round today = 1 round; require a 5 % gain on it
    int best_s = 1;
    for (int s = 2; s <= 8 && kchunks / s >= 8; ++s) {
      const double cost = std::ceil(rem * (double)s / slots) / s + (rem * (s + 1.0) * tile_bytes / 4.0e12 + 6e-6) / t_round;
      if (cost < best) {
        best = cost;
        best_s = s;
      }
    }
    if (best_s == 1 && tail_s_instead > 1) best_s = tail_s_instead;   // (the two estimates differ in form: never fall between them)
    if (best_s > 1) {
      p.tail_first = full;
      p.tail_cps = divup(kchunks, best_s);
      p.tail_splits = divup(kchunks, p.tail_cps);
      p.tail_tf8 = full / 8;
      p.tail_tt8 = divup(rem * p.tail_splits, 8);
      p.tail_partial = static_cast<float*>(workspace(sizeof(float) * (size_t)rem * p.tail_splits * ROWS * WC * CW));
    }
  }
  dim3 grid(p.tail_splits > 1 ? 8 * (p.tail_tf8 + p.tail_tt8) : ((tiles + 7) / 8) * 8, splits);
  dim3 block(WR * WC * 64);
  static const std::string kname_g = "gg_kernel<" + std::to_string(WR) + "," + std::to_string(WC) + "," + std::to_string(MT) + "," + std::to_string(CW) + "," + (AK ? "kc" : "rc") + ">";
  static const std::string kname_p = "ggp_kernel<" + std::to_string(WR) + "," + std::to_string(WC) + "," + std::to_string(MT) + "," + std::to_string(CW) + ">";
  static const std::string kname_s0 = kname_p.substr(0, kname_p.size() - 1) + ",split>";
  static const std::string kname_s1 = kname_p.substr(0, kname_p.size() - 1) + ",split,pre>";   // A read as pre-split bf16 planes
  const std::string& kname_s = p.apre ? kname_s1 : kname_s0;
  static const std::string kname_gs = kname_g.substr(0, kname_g.size() - 1) + ",split>";
  const std::string& kname = p.KC > 0 ? (gg_split_mode() ? kname_s : kname_p) : ((vec && gg_split_mode()) ? kname_gs : kname_g);
  {
    // ggp_kernel leaves out the border taps no pixel of the tile has when the block owns its whole reduction: executed <= algorithmic then
    const bool skips = p.KC > 0 && splits == 1;
    KernelTimer timer(kname.c_str(), t_op, t_flops, 0.0, skips ? 0.0 : t_exec);
    if constexpr (!AK && WR == 2 && WC == 2 && MT == 2 && CW == 128) {
      if (o3) {
        allow_big_lds(gg_kernel<WR, WC, MT, CW, AK, true, true>, lds);
        hipLaunchKernelGGL((gg_kernel<WR, WC, MT, CW, AK, true, true>), grid, block, lds, stream(), p, kNoClasses);
      }
    }
    bool donep = false;
    if (p.KC > 0) CHIP_REQUIRE(!AK && vec && WC * (CW / 4) == 64);
    if constexpr (!AK && WC * (CW / 4) == 64) {
      if (p.KC > 0) {
        const size_t lds3 = sizeof(float) * 3 * (A_STAGE + B_STAGE) + lds_pad;
        if (gg_split_mode() && p.apre) {
          const size_t lds3p = sizeof(float) * 3 * (6 * ROWS * 4 + B_STAGE) + lds_pad;
          allow_big_lds(ggp_kernel<WR, WC, MT, CW, true, true>, lds3p);
          hipLaunchKernelGGL((ggp_kernel<WR, WC, MT, CW, true, true>), grid, dim3(WR * WC * 64 + 64), lds3p, stream(), p, kNoClasses);
        } else if (gg_split_mode()) {
          allow_big_lds(ggp_kernel<WR, WC, MT, CW, true>, lds3);
          hipLaunchKernelGGL((ggp_kernel<WR, WC, MT, CW, true>), grid, dim3(WR * WC * 64 + 64), lds3, stream(), p, kNoClasses);
        } else {
          allow_big_lds(ggp_kernel<WR, WC, MT, CW>, lds3);
          hipLaunchKernelGGL((ggp_kernel<WR, WC, MT, CW>), grid, dim3(WR * WC * 64 + 64), lds3, stream(), p, kNoClasses);
        }
        donep = true;
      }
    }
    if (o3 || donep) {
    } else if (vec && gg_split_mode()) {
      allow_big_lds(gg_kernel<WR, WC, MT, CW, AK, true, false, true>, lds);
      hipLaunchKernelGGL((gg_kernel<WR, WC, MT, CW, AK, true, false, true>), grid, block, lds, stream(), p, kNoClasses);
    } else if (vec) {
      allow_big_lds(gg_kernel<WR, WC, MT, CW, AK, true>, lds);
      hipLaunchKernelGGL((gg_kernel<WR, WC, MT, CW, AK, true>), grid, block, lds, stream(), p, kNoClasses);
    } else {
      allow_big_lds(gg_kernel<WR, WC, MT, CW, AK, false>, lds);
      hipLaunchKernelGGL((gg_kernel<WR, WC, MT, CW, AK, false>), grid, block, lds, stream(), p, kNoClasses);
    }
  }
  if (p.tail_splits > 1) {
    const int rem = tiles - p.tail_first;
    KernelTimer timer("gg_tail_fix_kernel", t_op, 0.0, sizeof(float) * (double)rem * (p.tail_splits + 1) * ROWS * WC * CW);
    if (vec) hipLaunchKernelGGL((gg_tail_fix_kernel<WR, WC, MT, CW, true>), dim3(rem * kTailFixParts), block, 0, stream(), p);
    else hipLaunchKernelGGL((gg_tail_fix_kernel<WR, WC, MT, CW, false>), dim3(rem * kTailFixParts), block, 0, stream(), p);
  }
  if (splits > 1) {
    KernelTimer timer("gg_reduce_kernel", t_op, 0.0, sizeof(float) * (double)dst_elems * (splits + 1));
    size_t nb = (dst_elems + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(gg_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, stream(), p.dst, p.partial, p.bias, dst_elems, p.slab,
                       splits, (size_t)p.DP * p.N, p.scaleTargets, p.relu, p.mask, p.post_scale);
  }
}

}  // namespace

void wg_reduce_launch(const WGParams& p, size_t total, int splits, int groups, const char* op) {   // (wgw_kernel's; wg_launch_cfg carries its own copy)
  KernelTimer timer("wg_reduce_kernel", op, 0.0, sizeof(float) * (double)total * (splits + 1));
  size_t nb = (total + 255) / 256;
  if (nb > 4096) nb = 4096;
  const float* slabs = p.partial;
  int nslabs = splits;
  if (groups > 1) {
    float* stage = p.partial + (size_t)splits * total;
    const int per = divup(splits, groups);
    hipLaunchKernelGGL(wg_reduce_group_kernel, dim3((unsigned)nb, divup(splits, per)), dim3(256), 0, stream(), stage, p.partial, total, splits, per);
    slabs = stage;
    nslabs = divup(splits, per);
  }
  hipLaunchKernelGGL(wg_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, stream(), p.dst, p.bias_dst, slabs, total, (size_t)p.K * p.F, nslabs,
                     p.scaleTargets, p.scaleOutput);
}

void gg_reduce_launch(const GGParams& p, size_t dst_elems, int splits, const char* op) {
  KernelTimer timer("gg_reduce_kernel", op, 0.0, sizeof(float) * (double)dst_elems * (splits + 1));
  size_t nb = (dst_elems + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(gg_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, stream(), p.dst, p.partial, p.bias, dst_elems, p.slab, splits,
                     (size_t)p.DP * p.N, p.scaleTargets, p.relu, p.mask, p.post_scale);
}

namespace {

template <bool AK>
void gg_run(GGParams& p, bool vec, size_t dst_elems) {
  // pick the row tile (128/64/32) that pads the fewest rows; ties go to the larger tile.  (The 96-row
  // 6-wave config measured 68 TFLOP/s vs 106 for the 128-row one, so 96-row problems — conv1 fprop,
  // conv2 dgrad — run 25 % padded on the 128-row kernel: 80 effective TFLOP/s.)
  // row tile by problem height: 128 rows (2x2 waves of 64x128), 96 rows (4 waves of 96x64: conv1 fprop,
  // conv2 dgrad), 64 and 32 rows for small layers.
  const int force64 = CHIP_DIAG_KNOB("CONVNET_GG_ROWS64", 0);   // experiment knob: 64-row tiles everywhere
  const bool no_skinny = CHIP_DIAG_KNOB("CONVNET_GG_NO_SKINNY", 0) != 0;
  // An FC layer at a small per-GPU batch (strong scaling of a global batch: 128 / 64 / 32 images per GPU) is a GEMM with <= 128
  // columns, bound by streaming the weight matrix once: four waves stacked along the rows, one 64-column wave-column, so a
  // 32-image batch pads 2x instead of 8x and every weight row still reaches LDS with 16-byte pieces.
  if (p.skinny && vec && !no_skinny) gg_launch_cfg<4, 1, 1, 64, AK>(p, vec, dst_elems);
  else if (force64) gg_launch_cfg<2, 2, 1, 128, AK>(p, vec, dst_elems);
  else if (p.R > 96 || (p.R > 64 && p.R <= 96 && (!vec)))
    gg_launch_cfg<2, 2, 2, 128, AK>(p, vec, dst_elems);
  else if (p.R > 64)
    gg_launch_cfg<1, 4, 3, 64, AK>(p, vec, dst_elems);
  else if (p.R > 32)
    gg_launch_cfg<2, 2, 1, 128, AK>(p, vec, dst_elems);
  else
    gg_launch_cfg<1, 4, 1, 128, AK>(p, vec, dst_elems);
}

void gg_run_classes(GGParams& p, GGClassTable& ct, bool vec) {   // same tile choice as gg_run
  if (p.R > 96 || (p.R > 64 && p.R <= 96 && (!vec)))
    gg_launch_classes<2, 2, 2, 128>(p, ct, vec);
  else if (p.R > 64)
    gg_launch_classes<1, 4, 3, 64>(p, ct, vec);
  else if (p.R > 32)
    gg_launch_classes<2, 2, 1, 128>(p, ct, vec);
  else
    gg_launch_classes<1, 4, 1, 128>(p, ct, vec);
}

template <int WM, int WN, int MT, int NTL, int TS = 32>
void wg_launch_cfg(WGParams& p, bool vec) {
  constexpr int KT = WM * MT * TS, FT = WN * NTL * TS;
  const size_t lds = sizeof(float) * 2 * (KT + FT) * (vec ? WG_NB : WG_PITCH);
  p.k_tiles = divup(p.K, KT);
  if (p.bias_dst && divup(p.K + 1, KT) != p.k_tiles) p.bias_dst = nullptr;   // no padding row to spare: caller sums separately
  p.f_tiles = divup(p.F, FT);
  p.zero = zero_page();
  p.prio = wg_prio_mode();
  const int tiles = p.k_tiles * p.f_tiles;
  const size_t total = (size_t)(p.K + (p.bias_dst ? 1 : 0)) * p.F;
  const bool no_wide = CHIP_DIAG_KNOB("CONVNET_WG_NO_WIDE", 0) != 0;
  p.wide = (!no_wide && vec && TS == 32 && WM == 2 && WN == 2 && MT == 2 && NTL == 2 && p.F % 4 == 0 && aligned16(p.dst) &&
            (!p.bias_dst || aligned16(p.bias_dst))) ? 1 : 0;
  // one full round of resident blocks (2 per CU): floor, not ceil — 568 blocks on 512 slots take two
  // rounds and leave the chip half empty (measured: 1.05 waves/SIMD, 46 % MFMA busy).
  int splits = 1;
  if (tiles < kTargetBlocks) {
    splits = kTargetBlocks / tiles;
    const int max_by_len = p.chunks_total / 16 > 0 ? p.chunks_total / 16 : 1;
    if (splits > max_by_len) splits = max_by_len;
    const size_t max_by_bytes = (size_t(256) << 20) / (total * sizeof(float)) + 1;
    if ((size_t)splits > max_by_bytes) splits = (int)max_by_bytes;
    if (splits > 1024) splits = 1024;
    if (splits < 1) splits = 1;
  }
  p.chunks_per_split = divup(p.chunks_total, splits);
  splits = divup(p.chunks_total, p.chunks_per_split);
  p.splits = splits;
  const int groups = splits > 64 ? 32 : 1;   // two-level reduce when the slab count dwarfs the tile
  p.partial = splits > 1 ? static_cast<float*>(workspace(sizeof(float) * total * (splits + (groups > 1 ? groups : 0)))) : nullptr;
  dim3 grid(((tiles * splits + 7) / 8) * 8), block(WM * WN * 64);
  static const std::string kname_f = "wg_kernel<" + std::to_string(WM) + "," + std::to_string(WN) + "," + std::to_string(MT) + "," + std::to_string(NTL) + (TS == 16 ? ",x16>" : ">");
  static const std::string kname_s = kname_f.substr(0, kname_f.size() - 1) + ",split>";
  const bool split_products = vec && wg_split_mode();
  const std::string& kname = split_products ? kname_s : kname_f;
  {
    KernelTimer timer(kname.c_str(), t_op, t_flops, 0.0, t_exec);
    if (split_products) {
      allow_big_lds(wg_kernel<WM, WN, MT, NTL, true, TS, true>, lds);
      hipLaunchKernelGGL((wg_kernel<WM, WN, MT, NTL, true, TS, true>), grid, block, lds, stream(), p);
    } else if (vec) {
      allow_big_lds(wg_kernel<WM, WN, MT, NTL, true, TS>, lds);
      hipLaunchKernelGGL((wg_kernel<WM, WN, MT, NTL, true, TS>), grid, block, lds, stream(), p);
    } else {
      allow_big_lds(wg_kernel<WM, WN, MT, NTL, false, TS>, lds);
      hipLaunchKernelGGL((wg_kernel<WM, WN, MT, NTL, false, TS>), grid, block, lds, stream(), p);
    }
  }
  if (splits > 1) {
    KernelTimer timer("wg_reduce_kernel", t_op, 0.0, sizeof(float) * (double)total * (splits + 1));
    size_t nb = (total + 255) / 256;
    if (nb > 4096) nb = 4096;
    const float* slabs = p.partial;
    int nslabs = splits;
    if (groups > 1) {
      float* stage = p.partial + (size_t)splits * total;
      const int per = divup(splits, groups);
      hipLaunchKernelGGL(wg_reduce_group_kernel, dim3((unsigned)nb, divup(splits, per)), dim3(256), 0, stream(), stage, p.partial, total, splits, per);
      slabs = stage;
      nslabs = divup(splits, per);
    }
    hipLaunchKernelGGL(wg_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, stream(), p.dst, p.bias_dst, slabs, total,
                       (size_t)p.K * p.F, nslabs, p.scaleTargets, p.scaleOutput);
  }
}

void wg_launch(WGParams& p, bool vec) {
  if (wgw_try(p, vec, vec && wg_split_mode(), t_op, t_flops, t_exec)) return;   // the wide tile (wgrad_wide.hip), when selected
  // filter tile: fewest padded filters among 128/96/64/32; k tile 128, or 160 when that pads less.
  int ft = 128, pad = divup(p.F, 128) * 128;
  const int cands[3] = {96, 64, 32};
  for (int c : cands) {
    const int q = divup(p.F, c) * c;
    if (q < pad) {
      ft = c;
      pad = q;
    }
  }
  const bool k160 = divup(p.K, 160) * 160 < divup(p.K, 128) * 128;
  if (ft == 128) wg_launch_cfg<2, 2, 2, 2>(p, vec);
  else if (ft == 96 && k160) wg_launch_cfg<2, 2, 5, 3, 16>(p, vec);   // 160 x 96 as 2x2 waves of 80 x 48 (16x16x4 MFMA)
  else if (ft == 96) wg_launch_cfg<4, 1, 1, 3>(p, vec);
  else if (ft == 64) wg_launch_cfg<4, 1, 1, 2>(p, vec);
  else wg_launch_cfg<4, 1, 1, 1>(p, vec);
}

struct ConvGeo {
  int N, C, H, W, F, Ky, Kx, sy, sx, py, px, My, Mx;
};

ConvGeo conv_geo(const Shape4D* img, const Shape4D* flt, const Shape4D* out, const ConvDesc& d, const cudamat* mi,
                 const cudamat* mf, const cudamat* mo) {
  // Same consistency checks as the reference (cudamat_conv_gemm.cu:586-610).
  ConvGeo g;
  g.N = img->shape[0]; g.W = img->shape[1]; g.H = img->shape[2]; g.C = img->shape[3];
  g.Mx = out->shape[1]; g.My = out->shape[2]; g.F = out->shape[3];
  g.Ky = d.kernel_size_y; g.Kx = d.kernel_size_x; g.sy = d.stride_y; g.sx = d.stride_x;
  g.py = d.padding_y; g.px = d.padding_x;
  CHIP_REQUIRE(out->shape[0] == g.N);
  CHIP_REQUIRE(d.num_input_channels == g.C && d.num_output_channels == g.F);
  CHIP_REQUIRE(d.num_groups == 1);
  CHIP_REQUIRE(d.input_channel_begin == 0 && (d.input_channel_end == 0 || d.input_channel_end == g.C));
  CHIP_REQUIRE(d.output_channel_begin == 0 && (d.output_channel_end == 0 || d.output_channel_end == g.F));
  CHIP_REQUIRE(flt->shape[0] == g.F && flt->shape[1] == g.Kx && flt->shape[2] == g.Ky && flt->shape[3] == g.C);
  CHIP_REQUIRE(mi->size[0] == g.N && mi->size[1] == g.H * g.W * g.C);
  CHIP_REQUIRE(mo->size[0] == g.N && mo->size[1] == g.My * g.Mx * g.F);
  CHIP_REQUIRE(mf->size[0] == g.F && mf->size[1] == g.Ky * g.Kx * g.C);
  CHIP_REQUIRE(g.My == (g.H - 2 * g.py - g.Ky) / g.sy + 1 && g.Mx == (g.W - 2 * g.px - g.Kx) / g.sx + 1);
  CHIP_REQUIRE(d.kernel_size_t <= 1);
  // the kernels index activations with 32-bit element offsets: refuse loudly instead of wrapping (2^31 floats = 8.6 GB per tensor)
  CHIP_REQUIRE((size_t)g.N * g.H * g.W * g.C < (1ull << 31) && (size_t)g.N * g.My * g.Mx * g.F < (1ull << 31));
  return g;
}

// Row-tile height gg_run / gg_run_classes pick for an R-row problem on the producer-wave kernels (the pre-split filter planes are laid
// out per row tile of this height, filter_planes_rt_kernel).
inline int gg_tile_rows(int R) { return CHIP_DIAG_KNOB("CONVNET_GG_ROWS64", 0) ? 64 : R > 96 ? 128 : R > 64 ? 96 : R > 32 ? 64 : 32; }

inline bool gg_presplit_mode() {
  return gg_split_mode() && !CHIP_DIAG_KNOB("CONVNET_GG_NO_PRESPLIT", 0);
}

// k-row table of ggp_kernel's generic-k mode (GGParams::ktab), built once per geometry and kept on the device
const unsigned* gk_table(int C, int H, int W, int N, int Ky, int Kx) {
  static std::map<std::array<int, 7>, unsigned*> cache;   // (keyed by device too: the table lives in that device's memory)
  int dev = 0;
  CHIP_CHECK(hipGetDevice(&dev));
  const std::array<int, 7> key = {dev, C, H, W, N, Ky, Kx};
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  const int K = C * Ky * Kx, KP = divup(K, BK) * BK;
  std::vector<unsigned> tab(2 * (size_t)KP);
  for (int k = 0; k < KP; ++k) {
    const int kk = std::min(k, K - 1), ch = kk / (Ky * Kx), tap = kk % (Ky * Kx), a = tap / Kx, b = tap % Kx;
    tab[k] = (unsigned)(((size_t)(ch * H + a) * W + b) * N * 4);
    tab[KP + k] = ((unsigned)a << 16) | (unsigned)b;
  }
  unsigned* d = nullptr;
  CHIP_CHECK(hipMalloc((void**)&d, tab.size() * sizeof(unsigned)));
  CHIP_CHECK(hipMemcpy(d, tab.data(), tab.size() * sizeof(unsigned), hipMemcpyHostToDevice));
  cache[key] = d;
  return d;
}

void conv_up_impl(cudamat* images, cudamat* filters, cudamat* bias, cudamat* targets, Shape4D* is, Shape4D* fs, Shape4D* ts,
                  const ConvDesc& d, float scaleTargets, int relu) {
  const ConvGeo g = conv_geo(is, fs, ts, d, images, filters, targets);
  GGParams p{};
  p.A = filters->data_device; p.src = images->data_device; p.dst = targets->data_device;
  p.bias = bias ? bias->data_device : nullptr;
  p.R = g.F; p.K = g.C * g.Ky * g.Kx; p.N = g.N; p.lda = g.F;
  p.GX = g.Mx; p.G = g.My * g.Mx; p.TX = g.Kx; p.TYX = g.Ky * g.Kx;
  p.SH = g.H; p.SW = g.W; p.ssy = g.sy; p.ssx = g.sx; p.y0 = g.py; p.x0 = g.px; p.dir = 1;
  p.DW = g.Mx; p.DP = g.My * g.Mx; p.dsy = 1; p.dsx = 1; p.dy0 = 0; p.dx0 = 0;
  p.scaleTargets = scaleTargets; p.relu = relu;
  const bool vec = g.N % 4 == 0 && g.F % 4 == 0 && aligned16(p.A) && aligned16(p.src) && aligned16(p.dst);
  if (vec && gg_presplit_mode() && gg_producer_mode() &&
      gfc_try(images->data_device, filters->data_device, p.bias, targets->data_device, g.N, g.C, g.H, g.W, g.F, g.Ky, g.Kx, g.sy, g.sx, -g.py, -g.px,
              g.My, g.Mx, scaleTargets, relu, 2.0 * g.N * p.G * (double)g.F * p.K))
    return;
  if (vec && gg_presplit_mode() && gg_producer_mode() && g.C % BK != 0 && g.F > 32 && CHIP_KNOB("CONVNET_GG_GK", 1) &&
      (size_t)g.C * g.H * g.W * g.N < (size_t(1) << 30)) {
    // few input channels (conv1: C = 3): the producer-wave kernel in the bank's own k order, k-row sources from a table
    const int K = g.C * g.Ky * g.Kx, KP = divup(K, BK) * BK, TH = gg_tile_rows(g.F);
    void* planes = workspace_aux((size_t)96 * (KP / 16) * divup(g.F, TH) * TH);
    filter_planes_gk_launch(filters->data_device, planes, g.F, K, KP, TH, "conv_fprop");
    p.A = static_cast<const float*>(planes);
    p.apre = 1;
    p.gk = 1;
    p.ktab = gk_table(g.C, g.H, g.W, g.N, g.Ky, g.Kx);
    p.K = KP;
    p.KC = KP;   // selects the producer-wave kernels in the launchers; the reduction itself is the table's
    t_op = "conv_fprop";
    t_flops = 2.0 * g.N * p.G * (double)g.F * K;
    t_exec = 0.0;
    gg_run<false>(p, vec, (size_t)g.N * p.DP * g.F);
    note_kernel("gg_kernel(fprop)", t_flops, p.row_tiles * p.col_tiles, p.splits);
    return;
  }
  if (vec && ggp_shape_ok(g.F, g.C) && gg_presplit_mode()) {
    // patch-resident gather on pre-split source planes (patch_gemm.hip) where the geometry has one
    p.KC = g.C;
    if (patch_shape_ok(p, (size_t)g.N * p.DP * g.F)) {
      t_op = "conv_fprop";
      t_flops = 2.0 * g.N * p.G * (double)g.F * p.K;
      const PatchBank bank{filters->data_device, g.F, g.C, g.Ky, g.Kx, 0, 0, 1, 1, g.Ky, g.Kx, false};
      patch_run(p, (size_t)g.N * p.DP * g.F, t_op, t_flops, bank);
      note_kernel(convnet_hip_get_patch_mode() >= 3 ? "gpw_kernel(fprop)" : "gpp_kernel(fprop)", t_flops, p.row_tiles * p.col_tiles, p.splits);
      return;
    }
    p.KC = 0;
  }
  if (vec && ggp_shape_ok(g.F, g.C)) {
    // producer-wave kernel: the reduction runs tap-major over a re-laid copy of the filter bank (a few MB, ~5 us)
    const size_t welems = (size_t)g.F * p.K;
    if (gg_presplit_mode()) {
      // the consumers read the bank as ready-made bf16 planes: tap-major re-layout and exact three-way split in one pass
      const int TH = gg_tile_rows(g.F);
      void* planes = workspace_aux((size_t)96 * (g.C / 16) * p.TYX * divup(g.F, TH) * TH);
      const PatchBank bank{filters->data_device, g.F, g.C, g.Ky, g.Kx, 0, 0, 1, 1, g.Ky, g.Kx, false};
      filter_planes_rt_launch(bank, planes, p.TYX, TH, "conv_fprop");
      p.A = static_cast<const float*>(planes);
      p.apre = 1;
    } else if (p.TYX > 1) {
      float* wt = static_cast<float*>(workspace_aux(sizeof(float) * welems));
      int nb = (int)((welems + 255) / 256);
      if (nb > 2048) nb = 2048;
      KernelTimer timer("filter_tapmajor_kernel", "conv_fprop", 0.0, 8.0 * welems);
      hipLaunchKernelGGL(filter_tapmajor_kernel, dim3(nb), dim3(256), 0, stream(), filters->data_device, wt, g.F, g.C, p.TYX);
      p.A = wt;
    }
    p.KC = g.C;
  }
  t_op = "conv_fprop";
  t_flops = 2.0 * g.N * p.G * (double)g.F * p.K;
  t_exec = 0.0;
  gg_run<false>(p, vec, (size_t)g.N * p.DP * g.F);
  note_kernel("gg_kernel(fprop)", 2.0 * g.N * p.G * (double)g.F * p.K, p.row_tiles * p.col_tiles, p.splits);
}

}  // namespace
}  // namespace chip

using namespace chip;

#ifdef CONVNET_GG_TRACE
extern "C" void convnet_hip_debug_set_gg_trace(unsigned long long* dev_buf) {
  CHIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(chip::g_gg_trace), &dev_buf, sizeof dev_buf));
}
#endif

extern "C" {

// deferred epilogues (common.h: PendingOp): the parked forms of convUp / convDown
static void conv_down_impl(cudamat* derivs, cudamat* filters, cudamat* targets, Shape4D* ds, Shape4D* fs, Shape4D* ts, const ConvDesc& d,
                           float scaleTargets, cudamat* mask, float post_scale);
static void launch_conv_up(PendingOp& o) {
  conv_up_impl(&o.m[0], &o.m[1], o.has_bias ? &o.bias : nullptr, &o.m[2], &o.s[0], &o.s[1], &o.s[2], o.desc, o.scaleTargets, o.relu);
}
static void launch_conv_down(PendingOp& o) {
  conv_down_impl(&o.m[0], &o.m[1], &o.m[2], &o.s[0], &o.s[1], &o.s[2], o.desc, o.scaleTargets, o.has_mask ? &o.mask : nullptr, 1.0f);
}
static bool park_conv(int kind, void (*launch)(PendingOp&), cudamat* a, cudamat* b, cudamat* c, Shape4D* sa, Shape4D* sb, Shape4D* sc, const ConvDesc& d,
                      float scaleTargets) {
  if (!defer_begin(kind, launch)) return false;
  // the call's consistency checks run NOW, so that a bad convUp / convDown fails at the call that made it and not inside whichever later
  // call launches the parked one: kind 1 = (images, filters, targets), kind 2 = (derivs, filters, targets)
  if (kind == 1) (void)conv_geo(sa, sb, sc, d, a, b, c);
  else (void)conv_geo(sc, sb, sa, d, c, b, a);
  PendingOp& o = pending();
  o.m[0] = *a; o.m[1] = *b; o.m[2] = *c;
  o.s[0] = *sa; o.s[1] = *sb; o.s[2] = *sc;
  o.desc = d;
  o.scaleTargets = scaleTargets;
  return true;
}

void convUpGemm(cudamat* images, cudamat* filters, cudamat* targets, Shape4D* is, Shape4D* fs, Shape4D* ts, ConvDesc d,
                float scaleTargets) {
  if (park_conv(1, launch_conv_up, images, filters, targets, is, fs, ts, d, scaleTargets)) return;
  conv_up_impl(images, filters, nullptr, targets, is, fs, ts, d, scaleTargets, 0);
}

void convUp(cudamat* images, cudamat* filters, cudamat* targets, Shape4D* is, Shape4D* fs, Shape4D* ts, ConvDesc d,
            float scaleTargets) {
  if (park_conv(1, launch_conv_up, images, filters, targets, is, fs, ts, d, scaleTargets)) return;
  conv_up_impl(images, filters, nullptr, targets, is, fs, ts, d, scaleTargets, 0);
}

void convUpBiasAct(cudamat* images, cudamat* filters, cudamat* bias, cudamat* targets, Shape4D* is, Shape4D* fs, Shape4D* ts,
                   ConvDesc d, float scaleTargets, int relu) {
  if (bias) CHIP_REQUIRE(bias->size[0] * bias->size[1] == d.num_output_channels);
  conv_up_impl(images, filters, bias, targets, is, fs, ts, d, scaleTargets, relu);
}

static void conv_down_impl(cudamat* derivs, cudamat* filters, cudamat* targets, Shape4D* ds, Shape4D* fs, Shape4D* ts, const ConvDesc& d,
                           float scaleTargets, cudamat* mask, float post_scale) {
  const ConvGeo g = conv_geo(ts, fs, ds, d, targets, filters, derivs);
  if (mask) CHIP_REQUIRE(numel(mask) == numel(targets));
  // one launch per stride class (cy,cx): input rows iy with (iy - pad) % sy == cy share the tap set
  // ky = cy + sy*a; their sources are oy = (iy - pad - cy)/sy - a  (pad = ConvDesc padding, <= 0).
  const size_t wt_floats = ((size_t)g.C * g.F * g.Ky * g.Kx + 64 * (size_t)g.sy * g.sx + 63) / 64 * 64;
  // + room for the class banks as bf16 planes per row tile (rows padded to the tile height): 24 floats per (16 filters, tap, row)
  const int TH = gg_tile_rows(g.C);
  const size_t pl_floats = (size_t)24 * (g.F / 16 + 1) * g.Ky * g.Kx * divup(g.C, TH) * TH + 64 * (size_t)g.sy * g.sx;
  float* wt = static_cast<float*>(workspace_aux(sizeof(float) * (wt_floats + pl_floats)));
  size_t poff = 0;
  size_t woff = 0;
  double flops = 0;
  int blocks = 0;
  bool patched = false;
  GGParams base{};
  base.src = derivs->data_device; base.dst = targets->data_device; base.bias = nullptr;
  base.R = g.C; base.N = g.N; base.lda = g.C;
  base.SH = g.My; base.SW = g.Mx; base.ssy = 1; base.ssx = 1; base.dir = -1;
  base.DW = g.W; base.DP = g.H * g.W; base.dsy = g.sy; base.dsx = g.sx;
  base.scaleTargets = scaleTargets; base.relu = 0;
  base.mask = mask ? mask->data_device : nullptr; base.post_scale = post_scale;
  const bool vec = g.N % 4 == 0 && g.C % 4 == 0 && aligned16(base.src) && aligned16(base.dst) && aligned16(base.mask);
  GGClassTable ct{};
  const bool tapm = vec && ggp_shape_ok(g.C, g.F);   // producer-wave kernel: k = tap*F + f over tap-major class filters
  if (tapm) base.KC = g.F;
  const bool pre = tapm && gg_presplit_mode();        // ... reading the class banks as bf16 planes (dgrad_filter_planes_kernel)
  float* const planes = wt + wt_floats;
  if (pre) base.apre = 1;
  const bool multi = g.sy * g.sx > 1 && g.sy * g.sx <= kMaxClasses;   // all classes in one launch (no wave-quantisation per class)
  t_op = "conv_dgrad";
  // Work accounting.  ALGORITHMIC = the transposed convolution's MACs, 2*N*My*Mx*F*C*Ky*Kx (every output pixel meets every tap
  // once — the same count as fprop).  EXECUTED = what the gather issues: every INPUT pixel of a class runs the class's whole tap
  // set, border pixels on the zero page (conv5, pad 0: 169*9 vs 121*9 taps = 1.40x; conv2: 1.13x).  bench.py's roofline uses the
  // algorithmic figure; the executed one travels beside it.
  const double alg_flops = 2.0 * g.N * (double)g.My * g.Mx * (double)g.F * g.C * g.Ky * g.Kx;
  double exec_total = 0;
  for (int cy = 0; cy < g.sy; ++cy)
    for (int cx = 0; cx < g.sx; ++cx) {
      const int TYc = cy < g.Ky ? divup(g.Ky - cy, g.sy) : 0, TXc = cx < g.Kx ? divup(g.Kx - cx, g.sx) : 0;
      const int ny = -g.py - cy, nx = -g.px - cx;
      const int jy0 = ny > 0 ? (ny + g.sy - 1) / g.sy : 0, jx0 = nx > 0 ? (nx + g.sx - 1) / g.sx : 0;
      const int iy0 = cy + g.py + g.sy * jy0, ix0 = cx + g.px + g.sx * jx0;
      if (iy0 >= g.H || ix0 >= g.W) continue;
      exec_total += 2.0 * g.N * (double)(((g.H - 1 - iy0) / g.sy + 1) * ((g.W - 1 - ix0) / g.sx + 1)) * g.C * (double)(g.F * TYc * TXc);
    }
  for (int cy = 0; cy < g.sy; ++cy) {
    for (int cx = 0; cx < g.sx; ++cx) {
      const int TYc = cy < g.Ky ? divup(g.Ky - cy, g.sy) : 0;
      const int TXc = cx < g.Kx ? divup(g.Kx - cx, g.sx) : 0;
      // iy = cy + pad + sy*j >= 0  ->  j >= ceil((-pad - cy)/sy)
      auto first_j = [](int c, int pad, int s) {
        const int need = -pad - c;
        return need > 0 ? (need + s - 1) / s : 0;
      };
      const int jy0 = first_j(cy, g.py, g.sy), jx0 = first_j(cx, g.px, g.sx);
      const int iy0 = cy + g.py + g.sy * jy0, ix0 = cx + g.px + g.sx * jx0;
      if (iy0 >= g.H || ix0 >= g.W) continue;
      const int GY = (g.H - 1 - iy0) / g.sy + 1, GX = (g.W - 1 - ix0) / g.sx + 1;
      float* wc = wt + woff;
      const size_t welems = (size_t)g.C * g.F * TYc * TXc;
      woff += (welems + 63) / 64 * 64;
      GGClass k{};
      k.A = wc; k.K = g.F * TYc * TXc;
      if (!multi && pre && welems > 0) {
        // a stride-1 convolution: one class, a stride-1 gather over the derivatives -> the patch-resident kernel where it applies
        GGParams p = base;
        p.K = k.K; p.GX = GX; p.G = GY * GX; p.TX = TXc; p.TYX = TYc * TXc;
        p.y0 = jy0; p.x0 = jx0; p.dy0 = iy0; p.dx0 = ix0;
        const bool whole = g.sy == 1 && g.sx == 1 && GY == g.H && GX == g.W;
        if (patch_shape_ok(p, whole ? (size_t)g.N * g.H * g.W * g.C : 0)) {
          const double cflops = 2.0 * g.N * p.G * (double)g.C * p.K;
          flops += cflops;
          t_flops = exec_total > 0 ? alg_flops * (cflops / exec_total) : 0.0;
          const PatchBank bank{filters->data_device, g.F, g.C, g.Ky, g.Kx, cy, cx, g.sy, g.sx, TYc, TXc, true};
          patch_run(p, whole ? (size_t)g.N * g.H * g.W * g.C : 0, t_op, t_flops, bank);
          patched = true;
          blocks += p.row_tiles * p.col_tiles;
          continue;
        }
      }
      if (pre && welems > 0) {
        // class bank straight to bf16 planes (re-layout + exact split in one pass)
        float* pc = planes + poff;
        poff += ((size_t)24 * (g.F / 16) * TYc * TXc * divup(g.C, TH) * TH + 63) / 64 * 64;
        const PatchBank bank{filters->data_device, g.F, g.C, g.Ky, g.Kx, cy, cx, g.sy, g.sx, TYc, TXc, true};
        filter_planes_rt_launch(bank, pc, TYc * TXc, TH, "conv_dgrad");
        k.A = pc;
      } else if (welems > 0) {
        int nb = (int)((welems + 255) / 256);
        if (nb > 2048) nb = 2048;
        KernelTimer timer("dgrad_filter_kernel", "conv_dgrad", 0.0, 8.0 * welems);
        hipLaunchKernelGGL(dgrad_filter_kernel, dim3(nb), dim3(256), 0, stream(), filters->data_device, wc, g.F, g.C, g.Ky,
                           g.Kx, cy, cx, g.sy, g.sx, TYc, TXc, tapm ? 1 : 0);
      }
      k.GX = GX; k.G = GY * GX; k.TX = TXc > 0 ? TXc : 1; k.TYX = TYc * TXc > 0 ? TYc * TXc : 1;
      k.y0 = jy0; k.x0 = jx0; k.dy0 = iy0; k.dx0 = ix0;
      const double cflops = 2.0 * g.N * k.G * (double)g.C * k.K;
      flops += cflops;
      if (multi) {
        ct.c[ct.n++] = k;
        continue;
      }
      GGParams p = base;
      p.A = k.A; p.K = k.K; p.GX = k.GX; p.G = k.G; p.TX = k.TX; p.TYX = k.TYX;
      p.y0 = k.y0; p.x0 = k.x0; p.dy0 = k.dy0; p.dx0 = k.dx0;
      t_flops = exec_total > 0 ? alg_flops * (cflops / exec_total) : 0.0;
      t_exec = cflops;
      // a stride-1 convolution has a single class that owns every input pixel: split-K is legal there
      const bool whole = g.sy == 1 && g.sx == 1 && GY == g.H && GX == g.W;
      gg_run<false>(p, vec, whole ? (size_t)g.N * g.H * g.W * g.C : 0);
      blocks += p.row_tiles * p.col_tiles;
    }
  }
  if (multi && ct.n > 0) {
    t_flops = alg_flops;
    t_exec = flops;
    if (pre && patch_classes_ok(base, ct)) {
      patch_run_classes(base, ct, t_op, t_flops, t_exec);   // the wide tile over 3- and 2-tap rows (patch_gemm.hip: gpv_kernel)
      patched = true;
    } else {
      gg_run_classes(base, ct, vec);
    }
    blocks = ct.c[ct.n - 1].tile_end;
  }
  note_kernel(!patched ? "gg_kernel(dgrad)" : convnet_hip_get_patch_mode() >= 3 ? "gpw_kernel(dgrad)" : "gpp_kernel(dgrad)", alg_flops, blocks, 1);
}

void convDownGemm(cudamat* derivs, cudamat* filters, cudamat* targets, Shape4D* ds, Shape4D* fs, Shape4D* ts, ConvDesc d,
                  float scaleTargets) {
  if (park_conv(2, launch_conv_down, derivs, filters, targets, ds, fs, ts, d, scaleTargets)) return;
  conv_down_impl(derivs, filters, targets, ds, fs, ts, d, scaleTargets, nullptr, 1.0f);
}

void convDown(cudamat* derivs, cudamat* filters, cudamat* targets, Shape4D* ds, Shape4D* fs, Shape4D* ts, ConvDesc d,
              float scaleTargets) {
  if (park_conv(2, launch_conv_down, derivs, filters, targets, ds, fs, ts, d, scaleTargets)) return;
  conv_down_impl(derivs, filters, targets, ds, fs, ts, d, scaleTargets, nullptr, 1.0f);
}

void convDownMask(cudamat* derivs, cudamat* filters, cudamat* state, cudamat* targets, Shape4D* ds, Shape4D* fs, Shape4D* ts, ConvDesc d,
                  float scaleTargets, float post_scale) {
  conv_down_impl(derivs, filters, targets, ds, fs, ts, d, scaleTargets, state, post_scale);
}

static void conv_outp_impl(cudamat* images, cudamat* derivs, cudamat* targets, cudamat* bias_grad, Shape4D* is, Shape4D* ds, Shape4D* ts,
                           const ConvDesc& d, float scaleTargets, float scaleOutput) {
  const ConvGeo g = conv_geo(is, ts, ds, d, images, targets, derivs);
  WGParams p{};
  if (bias_grad) {
    CHIP_REQUIRE(bias_grad->on_device && (size_t)numel(bias_grad) == (size_t)g.F);
    p.bias_dst = bias_grad->data_device;
  }
  p.src = images->data_device; p.dout = derivs->data_device; p.dst = targets->data_device;
  p.K = g.C * g.Ky * g.Kx; p.F = g.F; p.N = g.N;
  p.GX = g.Mx; p.M = g.My * g.Mx; p.TX = g.Kx; p.TYX = g.Ky * g.Kx; p.SH = g.H; p.SW = g.W;
  p.ssy = g.sy; p.ssx = g.sx; p.y0 = g.py; p.x0 = g.px;
  p.nchunk = divup(g.N, WG_NB); p.chunks_total = p.M * p.nchunk;
  p.scaleTargets = scaleTargets; p.scaleOutput = scaleOutput;
  const bool vec = g.N % 4 == 0 && aligned16(p.src) && aligned16(p.dout);
  t_op = "conv_wgrad";
  t_flops = 2.0 * g.N * p.M * (double)g.F * p.K;
  t_exec = 0.0;
  wg_launch(p, vec);
  note_kernel("wg_kernel(wgrad)", 2.0 * g.N * p.M * (double)g.F * p.K, p.k_tiles * p.f_tiles, p.splits);
  if (bias_grad && !p.bias_dst) {
    // the tile had no spare row: db = scaleTargets*db + scaleOutput * colsum over (N*M, F) view of derivs
    cudamat view = *derivs;
    view.size[0] = g.N * p.M;
    view.size[1] = g.F;
    view.is_trans = 0;
    view.owns_data = 0;
    CHIP_REQUIRE(sum_by_axis(&view, bias_grad, 0, scaleOutput, scaleTargets) == 0);
  }
}

void convOutpGemm(cudamat* images, cudamat* derivs, cudamat* targets, Shape4D* is, Shape4D* ds, Shape4D* ts, ConvDesc d,
                  float scaleTargets, float scaleOutput) {
  conv_outp_impl(images, derivs, targets, nullptr, is, ds, ts, d, scaleTargets, scaleOutput);
}

void convOutpBias(cudamat* images, cudamat* derivs, cudamat* targets, cudamat* bias_grad, Shape4D* is, Shape4D* ds, Shape4D* ts,
                  ConvDesc d, float scaleTargets, float scaleOutput) {
  conv_outp_impl(images, derivs, targets, bias_grad, is, ds, ts, d, scaleTargets, scaleOutput);
}

void convOutp(cudamat* images, cudamat* derivs, cudamat* targets, Shape4D* is, Shape4D* ds, Shape4D* ts, ConvDesc d,
              int /*partialSumY*/, int /*partialSumX*/, float scaleTargets, float scaleOutput) {
  convOutpGemm(images, derivs, targets, is, ds, ts, d, scaleTargets, scaleOutput);
}

// target = beta*target + alpha*op(mat1)*op(mat2)   (cudamat.cu:2130-2152).
// The three shapes fc_edge.cc uses keep the image index as the contiguous dimension of the
// activation operand, so they are the 1-pixel cases of the conv kernels:
//   NT: out(N,F)  = in(N,D)  * W(F,D)^T      -> gg_kernel, A = W   (r-contiguous)
//   NN: din(N,D)  = dout(N,F)* W(F,D)        -> gg_kernel, A = W   (k-contiguous)
//   TN: dW(F,D)   = dout(N,F)^T * in(N,D)    -> wg_kernel
static int dot_impl(cudamat* mat1, cudamat* mat2, cudamat* bias, cudamat* target, float beta, float alpha, int relu, cudamat* mask,
                    float post_scale) {
  if (!mat1->on_device || !mat2->on_device || !target->on_device) return ERROR_NOT_ON_DEVICE;
  const int t1 = mat1->is_trans, t2 = mat2->is_trans;
  const int m = t1 ? mat1->size[1] : mat1->size[0], k1 = t1 ? mat1->size[0] : mat1->size[1];
  const int k2 = t2 ? mat2->size[1] : mat2->size[0], n = t2 ? mat2->size[0] : mat2->size[1];
  if (m != target->size[0] || n != target->size[1] || k1 != k2) return ERROR_INCOMPATIBLE_DIMENSIONS;
  if (target->is_trans) return ERROR_TRANSPOSED;
  const int K = k1;
  if ((t1 && t2) || (!t1 && alpha != 1.0f)) {
    // not one of fc_edge.cc's three shapes: the general kernel keeps dot()'s full contract; the fused extras have no such form
    if (bias || relu || mask) return ERROR_UNSUPPORTED;
    if (m == 0 || n == 0) return 0;
    KernelTimer timer("dot_generic_kernel", "dot_generic", 2.0 * m * (double)n * K, 0.0);
    hipLaunchKernelGGL(dot_generic_kernel, dim3(divup(m, 32), divup(n, 32)), dim3(256), 0, stream(), mat1->data_device, mat2->data_device,
                       target->data_device, m, n, K, mat1->size[0], mat2->size[0], t1, t2, beta, alpha);
    return launch_status();
  }
  if (!t1) {
    // activations (m = N images) x weights
    GGParams p{};
    p.src = mat1->data_device; p.dst = target->data_device; p.bias = bias ? bias->data_device : nullptr;
    p.A = mat2->data_device;
    p.R = n; p.K = K; p.N = m; p.lda = mat2->size[0];
    p.GX = 1; p.G = 1; p.TX = 1; p.TYX = 1; p.SH = 1; p.SW = 1; p.ssy = 1; p.ssx = 1; p.y0 = 0; p.x0 = 0; p.dir = 1;
    p.DW = 1; p.DP = 1; p.dsy = 1; p.dsx = 1; p.dy0 = 0; p.dx0 = 0;
    p.scaleTargets = beta; p.relu = relu;
    if (mask && numel(mask) != numel(target)) return ERROR_INCOMPATIBLE_DIMENSIONS;
    p.mask = mask ? mask->data_device : nullptr; p.post_scale = post_scale;
    const bool base_vec = m % 4 == 0 && aligned16(p.src) && aligned16(p.dst) && aligned16(p.A) && aligned16(p.mask);
    t_op = t2 ? "fc_fprop" : "fc_dgrad";
    t_flops = 2.0 * m * (double)n * K;
    t_exec = 0.0;
    // the 128-row x 64-column tile of a small per-GPU batch, for layers that fill its rows: an FC head with few outputs (R <= 64)
    // keeps the 32- / 64-row tiles instead of padding to 128 rows
    p.skinny = m <= 128 && n > 64 && !CHIP_DIAG_KNOB("CONVNET_GG_NO_SKINNY", 0);
    if (t2) {   // NT: A[r=f + F*k=d]
      const bool v = base_vec && n % 4 == 0;
      if (v && !p.skinny && ggp_shape_ok(n, K)) p.KC = K;   // one tap: tap-major IS channel-major, no re-layout
      gg_run<false>(p, v, (size_t)m * n);
    } else {    // NN: A[k=f + F*r=d]
      gg_run<true>(p, base_vec && K % 4 == 0 && mat2->size[0] % 4 == 0, (size_t)m * n);
    }
    note_kernel(t2 ? "gg_kernel(fc NT)" : "gg_kernel(fc NN)", 2.0 * m * (double)n * K, p.row_tiles * p.col_tiles, p.splits);
    return launch_status();
  }
  if (t1 && !t2) {
    if (bias || relu || mask) return ERROR_UNSUPPORTED;
    // TN: target(m=F, n=D)[f + F*d] = sum_i mat1[i + N*f] * mat2[i + N*d], i over N images
    WGParams p{};
    p.src = mat2->data_device; p.dout = mat1->data_device; p.dst = target->data_device;
    p.K = n; p.F = m; p.N = K;
    p.GX = 1; p.M = 1; p.TX = 1; p.TYX = 1; p.SH = 1; p.SW = 1; p.ssy = 1; p.ssx = 1; p.y0 = 0; p.x0 = 0;
    p.nchunk = divup(K, WG_NB); p.chunks_total = p.nchunk;
    p.scaleTargets = beta; p.scaleOutput = alpha;
    const bool vec = K % 4 == 0 && aligned16(p.src) && aligned16(p.dout);
    t_op = "fc_wgrad";
    t_flops = 2.0 * m * (double)n * K;
    t_exec = 0.0;
    wg_launch(p, vec);
    note_kernel("wg_kernel(fc TN)", 2.0 * m * (double)n * K, p.k_tiles * p.f_tiles, p.splits);
    return launch_status();
  }
  return ERROR_UNSUPPORTED;   // unreachable: every transpose combination is handled above
}

int dotBiasAct(cudamat* mat1, cudamat* mat2, cudamat* bias, cudamat* target, float beta, float alpha, int relu) {
  return dot_impl(mat1, mat2, bias, target, beta, alpha, relu, nullptr, 1.0f);
}

int dotMask(cudamat* mat1, cudamat* mat2, cudamat* state, cudamat* target, float beta, float alpha, float post_scale) {
  return dot_impl(mat1, mat2, nullptr, target, beta, alpha, 0, state, post_scale);
}

int dot(cudamat* mat1, cudamat* mat2, cudamat* target, float beta, float alpha) {
  return dot_impl(mat1, mat2, nullptr, target, beta, alpha, 0, nullptr, 1.0f);
}

}  // extern "C"
