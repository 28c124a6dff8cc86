// Feasibility probe for a COOPERATIVE-SPLIT gather-GEMM block (round 3): does an 8-wave, one-block-per-CU tile whose B operand is
// split to bf16 planes ONCE per block (through LDS) beat ggp_kernel's per-wave in-register split?
//
//   C[R x N] = A[R x K] * B[K x N],  fp32 in / out, products on the bf16 pipe from exact three-way splits (six MFMAs per block).
//   A arrives pre-split (the product's filter_planes layout: [chunk][plane*2 + lh][row][8 x bf16], k-slot j of group lh = k-row 2j + lh);
//   B is fp32 [k][n] (n contiguous), staged raw with global_load_lds, split by all 512 threads (one split8 per thread per chunk),
//   written to LDS as planes, read back as ds_read_b128 MFMA operands.
//
// Block = 8 waves (WR x WC), tile (WR*MT*32) x 256, 1 block per CU (<= 256 VGPRs).  No producer wave: every wave issues its share
// of the chunk's LDS-DMA (A: 6*ROWS/64 pieces, B: 16 pieces).  Per 16-deep chunk and wave: 48 MFMAs, 46 split VALU, 8 ds_read_b32,
// 3 ds_write_b128, 18 ds_read_b128, ~5 DMA pieces (ggp_kernel: 48 MFMAs, ~180 VALU, 32 + 6 LDS reads, and a fifth wave staging).
//
// Usage: coop_probe [R] [K] [N] [KP] [mode]     KP > 0: B row index = k % KP (re-reads like a tap walk);  mode bit 0: no staging after
// the prologue (consumer-only ceiling), bit 1: no split (planes of chunk 0 reused).  Results are only checked when mode == 0.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/coop_probe tools/coop_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int BK = 16, COLS = 256;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Split8 { u32x4 h, m, l; };
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
  f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ void split8(const float (&x)[8], Split8& s) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float x0 = x[2 * q], x1 = x[2 * q + 1];
    const unsigned H = pk_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(H << 16), r1 = x1 - __uint_as_float(H & 0xffff0000u);
    const unsigned M = pk_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(M << 16), s1 = r1 - __uint_as_float(M & 0xffff0000u);
    s.h[q] = H; s.m[q] = M; s.l[q] = pk_bf16(s0, s1);
  }
}
__device__ __forceinline__ f32x16 mma(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mac6(const Split8& a, const Split8& b, f32x16 v) {
  v = mma(a.m, b.m, v); v = mma(a.h, b.l, v); v = mma(a.l, b.h, v); v = mma(a.h, b.m, v); v = mma(a.m, b.h, v); v = mma(a.h, b.h, v);
  return v;
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

template <int WR, int WC, int MT, int NTC>
__global__ void __launch_bounds__(512, 1) coop_gemm(const u32x4* __restrict__ Ap, const float* __restrict__ B, float* __restrict__ C, int R, int K,
                                                   int N, int KP, int mode) {
  static_assert(WR * WC == 8 && WC * NTC * 32 == COLS, "8 waves, 256 columns");
  constexpr int ROWS = WR * MT * 32;
  constexpr int A_STAGE = 6 * ROWS * 16, RAW_STAGE = BK * COLS * 4, PL_STAGE = 6 * COLS * 16;   // bytes
  constexpr int NA = 6 * ROWS / 64;                                                              // A pieces per chunk
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* const sA = lds;                      // [2][A_STAGE]
  char* const sR = lds + 2 * A_STAGE;        // [2][RAW_STAGE]
  char* const sP = sR + 2 * RAW_STAGE;       // [2][PL_STAGE]

  const int row_tiles = R / ROWS;
  const int row_tile = blockIdx.x % row_tiles, col_tile = blockIdx.x / row_tiles;
  const int r0 = row_tile * ROWS, c0 = col_tile * COLS;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int wr = wave / WC, wc = wave % WC;
  const int li = lane & 31, lh = lane >> 5;
  const int nch = K / BK;

  auto issue_a = [&](int chunk, int st) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < (NA + 7) / 8; ++j) {
      const int p = wave + 8 * j;
      if (p < NA) {
        const int e = 64 * p + lane, pl2 = e / ROWS, row = e - pl2 * ROWS;
        const u32x4* src = Ap + ((size_t)chunk * 6 + pl2) * R + r0 + row;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(sA + st * A_STAGE + 1024 * p), 16, 0, 0);
      }
    }
  };
  auto issue_raw = [&](int chunk, int st) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int kr = 2 * wave + j;
      int k = chunk * BK + kr;
      if (KP > 0) k %= KP;
      const float* src = B + (size_t)k * N + c0 + 4 * lane;
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(sR + st * RAW_STAGE + 1024 * kr), 16, 0, 0);
    }
  };
  // this thread's share of the chunk's B split: column `scol`, k-group `slh` (k-rows 2j + slh)
  const int scol = tid & (COLS - 1), slh = tid >> 8;
  auto split_chunk = [&](int rst, int pst) __attribute__((always_inline)) {
    const float* rp = reinterpret_cast<const float*>(sR + rst * RAW_STAGE) + slh * COLS + scol;
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = rp[2 * j * COLS];
    Split8 s;
    split8(x, s);
    u32x4* pp = reinterpret_cast<u32x4*>(sP + pst * PL_STAGE) + slh * COLS + scol;
    pp[0] = s.h;
    pp[2 * COLS] = s.m;
    pp[4 * COLS] = s.l;
  };

  f32x16 acc[MT][NTC];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int u = 0; u < NTC; ++u)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][u][e] = 0.f;

  issue_a(0, 0);
  issue_raw(0, 0);
  if (nch > 1) issue_raw(1, 1);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  split_chunk(0, 0);
  __syncthreads();

  for (int i = 0; i < nch; ++i) {
    const int st = i & 1;
    const bool stage_on = !(mode & 1) || i == 0;
    if (stage_on) {
      if (i + 1 < nch) issue_a(i + 1, st ^ 1);
      if (i + 2 < nch) issue_raw(i + 2, st);
    }
    if (i + 1 < nch && !(mode & 2)) split_chunk(st ^ 1, st ^ 1);
    const int ast = (mode & 1) ? 0 : st, pst = (mode & 2) ? 0 : st;
    const u32x4* ap = reinterpret_cast<const u32x4*>(sA + ast * A_STAGE) + lh * ROWS + wr * MT * 32 + li;
    const u32x4* bp = reinterpret_cast<const u32x4*>(sP + pst * PL_STAGE) + lh * COLS + wc * NTC * 32 + li;
    Split8 fa[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      fa[t].h = ap[t * 32];
      fa[t].m = ap[2 * ROWS + t * 32];
      fa[t].l = ap[4 * ROWS + t * 32];
    }
    Split8 fb[2];
    fb[0].h = bp[0]; fb[0].m = bp[2 * COLS]; fb[0].l = bp[4 * COLS];
#pragma unroll
    for (int u = 0; u < NTC; ++u) {
      if (u + 1 < NTC) {
        fb[(u + 1) & 1].h = bp[(u + 1) * 32];
        fb[(u + 1) & 1].m = bp[2 * COLS + (u + 1) * 32];
        fb[(u + 1) & 1].l = bp[4 * COLS + (u + 1) * 32];
      }
#pragma unroll
      for (int t = 0; t < MT; ++t) acc[t][u] = mac6(fa[t], fb[u & 1], acc[t][u]);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
  }

#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int u = 0; u < NTC; ++u)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = r0 + wr * MT * 32 + t * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
        const int col = c0 + wc * NTC * 32 + u * 32 + li;
        C[(size_t)row * N + col] = acc[t][u][reg];
      }
}

// PING-PONG variant: the two waves of a SIMD (w and w + 4: a workgroup's waves go to the SIMDs cyclically) alternate — one runs
// the 48 MFMAs of its chunk from REGISTER-resident fragments while its partner does everything else for its next chunk: 18
// ds_read_b128 of fragments, its share of the LDS-DMA, one split8.  Two barriers per chunk.  Group 0 (waves 0-3) computes chunk i in
// phase 2i and loads chunk i+1 in phase 2i+1; group 1 loads chunk i in phase 2i and computes it in phase 2i+1.
//   split of chunk j:   columns 0-127 by group 0 in phase 2j-3, columns 128-255 by group 1 in phase 2j-2   (raw(j) landed before)
//   A(j): issued by group 0 in phase 2j-3;   raw(j): issued by group 1 in phase 2j-6
template <int WR, int WC, int MT, int NTC>
__global__ void __launch_bounds__(512, 1) pp_gemm(const u32x4* __restrict__ Ap, const float* __restrict__ B, float* __restrict__ C, int R, int K,
                                                 int N, int KP, int mode) {
  static_assert(WR * WC == 8 && WC * NTC * 32 == COLS, "8 waves, 256 columns");
  constexpr int ROWS = WR * MT * 32;
  constexpr int A_STAGE = 6 * ROWS * 16, RAW_STAGE = BK * COLS * 4, PL_STAGE = 6 * COLS * 16;
  constexpr int NA = 6 * ROWS / 64, NAW = (NA + 3) / 4;   // A pieces per chunk; per wave of group 0
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* const sA = lds;                      // [2][A_STAGE]
  constexpr int D = 4;                        // raw ring depth = how many chunks ahead of its MFMAs a raw chunk is requested
  char* const sR = lds + 2 * A_STAGE;        // [D][RAW_STAGE]
  char* const sP = sR + D * RAW_STAGE;       // [2][PL_STAGE]
  const int row_tiles = R / ROWS;
  const int row_tile = blockIdx.x % row_tiles, col_tile = blockIdx.x / row_tiles;
  const int r0 = row_tile * ROWS, c0 = col_tile * COLS;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int grp = wave >> 2, gw = wave & 3;   // group, wave within group
  const int wr = wave / WC, wc = wave % WC;
  const int li = lane & 31, lh = lane >> 5;
  const int nch = K / BK;

  auto issue_a = [&](int chunk) __attribute__((always_inline)) {   // group 0: all NA pieces of the chunk over its 4 waves
#pragma unroll
    for (int j = 0; j < NAW; ++j) {
      const int p = gw + 4 * j;
      if (p < NA) {
        const int e = 64 * p + lane, pl2 = e / ROWS, row = e - pl2 * ROWS;
        const u32x4* src = Ap + ((size_t)chunk * 6 + pl2) * R + r0 + row;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(sA + (chunk & 1) * A_STAGE + 1024 * p), 16, 0, 0);
      }
    }
  };
  auto issue_raw = [&](int chunk) __attribute__((always_inline)) {   // group 1: 16 k-rows over its 4 waves
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kr = 4 * gw + j;
      int k = chunk * BK + kr;
      if (KP > 0) k %= KP;
      const float* src = B + (size_t)k * N + c0 + 4 * lane;
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(sR + (chunk % D) * RAW_STAGE + 1024 * kr), 16, 0, 0);
    }
  };
  // split share: group g takes columns 128 g .. 128 g + 127; thread = (column, k-group)
  const int scol = 128 * grp + (tid & 127), slh = (tid >> 7) & 1;
  auto split_chunk = [&](int chunk) __attribute__((always_inline)) {
    const float* rp = reinterpret_cast<const float*>(sR + (chunk % D) * RAW_STAGE) + slh * COLS + scol;
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = rp[2 * j * COLS];
    Split8 s;
    split8(x, s);
    u32x4* pp = reinterpret_cast<u32x4*>(sP + (chunk & 1) * PL_STAGE) + slh * COLS + scol;
    pp[0] = s.h;
    pp[2 * COLS] = s.m;
    pp[4 * COLS] = s.l;
  };
  Split8 fa[MT], fb[NTC];
  auto load_frags = [&](int chunk) __attribute__((always_inline)) {
    const u32x4* ap = reinterpret_cast<const u32x4*>(sA + (chunk & 1) * A_STAGE) + lh * ROWS + wr * MT * 32 + li;
    const u32x4* bp = reinterpret_cast<const u32x4*>(sP + (chunk & 1) * PL_STAGE) + lh * COLS + wc * NTC * 32 + li;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      fa[t].h = ap[t * 32];
      fa[t].m = ap[2 * ROWS + t * 32];
      fa[t].l = ap[4 * ROWS + t * 32];
    }
#pragma unroll
    for (int u = 0; u < NTC; ++u) {
      fb[u].h = bp[u * 32];
      fb[u].m = bp[2 * COLS + u * 32];
      fb[u].l = bp[4 * COLS + u * 32];
    }
  };
  f32x16 acc[MT][NTC];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int u = 0; u < NTC; ++u)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][u][e] = 0.f;
  auto mfma_phase = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < NTC; ++u)
#pragma unroll
      for (int t = 0; t < MT; ++t) acc[t][u] = mac6(fa[t], fb[u], acc[t][u]);
  };

  // ---- prologue: chunks 0 and 1 fully prepared the slow way ----
  if (grp == 0) { issue_a(0); if (nch > 1) issue_a(1); }
  else { for (int c = 0; c < D && c < nch; ++c) issue_raw(c); }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  split_chunk(0);                 // both groups: their column halves
  if (nch > 1) split_chunk(1);
  __syncthreads();
  if (grp == 0) load_frags(0);    // group 0 enters phase 0 with chunk 0 in registers
  // NOTE planes(1) sits in stage 1, A(1) in stage 1: consumed by load phases for chunk 1 below

  for (int i = 0; i < nch; ++i) {
    // ---- phase 2i: group 0 computes chunk i; group 1 loads chunk i, issues raw(i+3), splits its half of chunk i+2
    __builtin_amdgcn_sched_barrier(0);
    if (grp == 0) {
      mfma_phase();
    } else {
      load_frags(i);
      if (!(mode & 1) && i + D < nch) issue_raw(i + D);
      if (!(mode & 2) && i + 2 < nch) split_chunk(i + 2);
    }
    __builtin_amdgcn_sched_barrier(0);
    // group 1 just issued raw(i+3) (4 pieces): leave them in flight, everything older (raw(i+2), split next phase by group 0) has landed;
    // group 0's A(i+1) (issued a phase ago) must be in LDS before group 1 reads it next phase... it is read in phase 2i+2: waited below
    if (grp == 0) __builtin_amdgcn_s_waitcnt(0x0F70);   // end of its MFMA phase: A(i+1), issued a phase ago, is in LDS
    __syncthreads();
    // ---- phase 2i+1: group 1 computes chunk i; group 0 loads chunk i+1, issues A(i+2), splits its half of chunk i+3 ... see header:
    // its half of chunk j is split in phase 2j-3, i.e. j = i + 2 here
    __builtin_amdgcn_sched_barrier(0);
    if (grp == 1) {
      mfma_phase();
    } else {
      if (i + 1 < nch) load_frags(i + 1);
      if (!(mode & 1) && i + 2 < nch) issue_a(i + 2);
      if (!(mode & 2) && i + 2 < nch) split_chunk(i + 2);
    }
    __builtin_amdgcn_sched_barrier(0);
    // group 1, end of its MFMA phase: raw(i+3) (split from the next phase on) is in LDS; raw(i+4), issued a phase ago, may still fly
    if (grp == 1) __builtin_amdgcn_s_waitcnt(4 | 0x0F70);
    __syncthreads();
  }

#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int u = 0; u < NTC; ++u)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = r0 + wr * MT * 32 + t * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
        const int col = c0 + wc * NTC * 32 + u * 32 + li;
        C[(size_t)row * N + col] = acc[t][u][reg];
      }
}

static unsigned short rne_bf16(float x) {
  unsigned u;
  memcpy(&u, &x, 4);
  u += 0x7FFF + ((u >> 16) & 1);
  return (unsigned short)(u >> 16);
}
static float bf2f(unsigned short h) {
  unsigned u = (unsigned)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

template <int WR, int WC, int MT, int NTC, bool PP>
static void run(int R, int K, int N, int KP, int mode) {
  constexpr int ROWS = WR * MT * 32;
  if (R % ROWS || N % COLS || K % BK) { fprintf(stderr, "R %% %d, N %% 256, K %% 16 must be 0\n", ROWS); exit(1); }
  std::vector<float> A((size_t)R * K), Bm((size_t)(KP > 0 ? KP : K) * N);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; };
  for (auto& v : A) v = rnd() * 0.1f;
  for (auto& v : Bm) v = rnd();
  // A planes: [chunk][pl*2+lh][row][8 x bf16], slot j of group lh = k-row 2j + lh
  std::vector<unsigned short> P((size_t)(K / BK) * 6 * R * 8);
  for (int c = 0; c < K / BK; ++c)
    for (int lh = 0; lh < 2; ++lh)
      for (int r = 0; r < R; ++r)
        for (int j = 0; j < 8; ++j) {
          const float x = A[(size_t)r * K + c * BK + 2 * j + lh];
          const unsigned short h = rne_bf16(x);
          const float r1 = x - bf2f(h);
          const unsigned short m = rne_bf16(r1);
          const unsigned short l = rne_bf16(r1 - bf2f(m));
          const unsigned short pl[3] = {h, m, l};
          for (int q = 0; q < 3; ++q) P[((((size_t)c * 3 + q) * 2 + lh) * R + r) * 8 + j] = pl[q];
        }
  u32x4* dA; float *dB, *dC;
  CK(hipMalloc(&dA, P.size() * 2)); CK(hipMalloc(&dB, Bm.size() * 4)); CK(hipMalloc(&dC, (size_t)R * N * 4));
  CK(hipMemcpy(dA, P.data(), P.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, Bm.data(), Bm.size() * 4, hipMemcpyHostToDevice));
  const size_t ldsb = 2 * (6 * ROWS * 16) + (PP ? 4 : 2) * (BK * COLS * 4) + 2 * (6 * COLS * 16);
  auto kern = PP ? pp_gemm<WR, WC, MT, NTC> : coop_gemm<WR, WC, MT, NTC>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
  const int blocks = (R / ROWS) * (N / COLS);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), ldsb, 0, dA, dB, dC, R, K, N, KP, mode);
  CK(hipDeviceSynchronize());
  const int reps = 5;
  CK(hipEventRecord(e0));
  for (int it = 0; it < reps; ++it) hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), ldsb, 0, dA, dB, dC, R, K, N, KP, mode);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const double tf = 2.0 * R * (double)K * N / (ms * 1e-3) / 1e12;
  printf(PP ? "pingpong " : "coop     ");
  printf("<%d,%d,%d,%d> tile %dx256  R=%d K=%d N=%d KP=%d mode=%d  blocks=%d lds=%zu  %.3f ms  %.1f TFLOP/s-eq (%.3f of 416.7)\n", WR, WC, MT, NTC,
         ROWS, R, K, N, KP, mode, blocks, ldsb, ms, tf, tf / 416.7);
  if (mode == 0) {
    std::vector<float> Cm((size_t)R * N);
    CK(hipMemcpy(Cm.data(), dC, Cm.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int t = 0; t < 200; ++t) {
      s = s * 1664525u + 1013904223u;
      const int r = (s >> 8) % R;
      s = s * 1664525u + 1013904223u;
      const int n = (s >> 8) % N;
      double acc = 0, mag = 0;
      for (int k = 0; k < K; ++k) {
        const double a = A[(size_t)r * K + k], b = Bm[(size_t)(KP > 0 ? k % KP : k) * N + n];
        acc += a * b;
        mag += fabs(a * b);
      }
      worst = fmax(worst, fabs(Cm[(size_t)r * N + n] - acc) / mag);
    }
    printf("   max |err| / sum|ab| over 200 samples: %.3f x 2^-24 %s\n", worst / ldexp(1.0, -24), worst < 64 * ldexp(1.0, -24) ? "OK" : "WRONG");
  }
  hipFree(dA); hipFree(dB); hipFree(dC);
}

int main(int argc, char** argv) {
  const int R = argc > 1 ? atoi(argv[1]) : 512, K = argc > 2 ? atoi(argv[2]) : 3456, N = argc > 3 ? atoi(argv[3]) : 65536;
  const int KP = argc > 4 ? atoi(argv[4]) : 384, mode = argc > 5 ? atoi(argv[5]) : 0;
  const int shape = argc > 6 ? atoi(argv[6]) : 0;
  if (shape == 0) run<4, 2, 2, 4, false>(R, K, N, KP, mode);
  else if (shape == 1) run<2, 4, 3, 2, false>(R, K, N, KP, mode);
  else if (shape == 2) run<4, 2, 2, 4, true>(R, K, N, KP, mode);
  else run<2, 4, 3, 2, true>(R, K, N, KP, mode);
  return 0;
}
