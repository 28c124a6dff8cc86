// gfc_kernel: conv fprop of a layer with FEW input channels and a wide strided filter — AlexNet's conv1 (3 channels, 7 x 7, stride 2:
// 110 x 110 x 96 outputs, K = 147) — as a patch-resident gather-GEMM.  Replaces _convUpGemm for that shape
// (/root/reference/cudamat/cudamat_conv_gemm.cu:545-640: kExpand im2col + cublasSgemm); the reference's direct back-end has dedicated
// C <= 3 kernels for the same reason (cudamat_conv_filteracts.cu:39-222).
//
// Why (profiles/r04_kernel_experiments.md): on ggp_kernel's generic-k path a tile is ONE output pixel x 256 images and every
// chunk stages 16 KB of source k-rows + 9 KB of filter planes per 1 152 MFMA cycles — 21.7 B/clk asked of a vector-memory path that moves
// ~10: 0.30 of the matrix pipe, each input pixel fetched ~12 times.  And the layer WRITES 1.19 GB: at the same ~10 B/clk per CU that is
// ~200 us on its own (measured: a first version of this kernel — 8 pixels per tile, the write-out a phase of its own — ran 610 us, 406
// without the stores, 378 without stores and refill), so the stores have to ride inside the MFMA stream.  Here
//   * the whole filter bank stays in LDS for the life of the block: 160 k-slots x 96 rows as bf16 planes = 90 KB, loaded once;
//   * a tile is 4 neighbouring output pixels of one row x 32 images x 96 filters (a wave: one pixel, 96 x 32, three accumulators), and
//     its source PATCH — all 21 (channel, tap row) rows x 13 input columns x 32 images (rows padded to 16 columns = two 1 KB LDS-DMA
//     pieces: 42 KB) — is staged once per tile: every MFMA operand is read from it at a per-lane base + a compile-time immediate;
//   * blocks are persistent (one per CU, a contiguous run of tiles each); a fifth wave, the producer, refills the patch half by half
//     for the NEXT tile while the four consumer waves work on this one (two barriers per tile) — its loads are the only vector-memory
//     reads of the block, the consumers' only vector-memory instructions are stores;
//   * the accumulators are double-buffered: while tile T accumulates, tile T-1 goes out between its MFMAs — as twelve 16-byte stores
//     per lane after a 4 x 4 transpose inside each quad of lanes (the MFMA layout gives a lane one image and 16 rows).
// k-slots: chunk c (of 10), k-group lh, slot j  <->  patch row r = 2c + (j >> 2)  (r = channel * 7 + tap row),  kx = 2 (j & 3) + lh.
// kx == 7 does not exist: those 20 spare slots carry the 7 taps of patch row 20 (spare index s = 2c + (j >> 2) < 7: kx = s) and zeros
// otherwise (filter planes zero, the source read points into a zeroed LDS region — no 0 x inf from a neighbour's pixel): 147 real
// k-slots in 160.  Arithmetic: the exact three-way bf16 split and six products of every gather-GEMM here (gather_gemm.h: split_mac).
#include <algorithm>

#include "gather_gemm.h"

namespace chip {
namespace gfc {

constexpr int KX = 7, KY = 7, CH = 3, S = 2, NR = CH * KY, P = 4, IMG = 32;
constexpr int PW = S * (P - 1) + KX;          // 13 input columns under 4 output pixels
constexpr int XB = IMG * 4;                   // bytes of one (row, column): 32 images
constexpr int RP = 2;                         // 1 KB LDS-DMA pieces per patch row: 16 columns, the last three padding
constexpr int ROWB = RP * 1024;
static_assert(PW * XB <= ROWB, "patch row");
constexpr int NCH = 10, ROWS = 96;
constexpr int H0_ROWS = 11, H1_ROWS = 10;     // patch rows 0..9 and 20 | rows 10..19
constexpr int H0_BYTES = H0_ROWS * ROWB, H1_BYTES = H1_ROWS * ROWB;
constexpr int BIAS_OFF = 0, BIAS_BYTES = 512; // the bias row (96 floats; zeros without one): read per store from LDS — a global load among the
                                              // consumers' stores would make every use wait for the stores in front of it
constexpr int A_OFF = BIAS_BYTES;
constexpr int A_CHUNK = 6 * ROWS * 16;        // planes h / m / l x k-group x 96 rows x 16 bytes
constexpr int A_BYTES = NCH * A_CHUNK;        // 92 160
constexpr int A_PIECES = A_BYTES / 1024;      // 90
constexpr int PATCH_OFF = A_OFF + A_BYTES;
constexpr int ZERO_OFF = PATCH_OFF + H0_BYTES + H1_BYTES;
constexpr int ZERO_BYTES = 1024 + 256;        // a lane base spans < 1 024 bytes
constexpr int LDS_BYTES = ZERO_OFF + ZERO_BYTES;
static_assert(LDS_BYTES <= 160 * 1024, "LDS");
static_assert(A_BYTES % 1024 == 0, "filter bank in whole pieces");
constexpr int NGROUP = 3 * 4;                 // 16-byte stores per lane and tile: one per four accumulator registers
constexpr int DUMP_BYTES = 4096;              // where the stores of rows past F (and of the tile before the first) go

// byte offset of patch row r inside the patch region
__host__ __device__ constexpr int row_off(int r) { return r < 10 ? r * ROWB : r == 20 ? 10 * ROWB : H0_BYTES + (r - 10) * ROWB; }
// filter column (kx + KX * r) of k-slot (c, lh, j), or -1 for a zero slot
__host__ __device__ constexpr int slot_k(int c, int lh, int j) {
  const int r = 2 * c + (j >> 2), kx = 2 * (j & 3) + lh;
  if (kx < KX) return kx + KX * r;
  return r < KX ? r + KX * 20 : -1;   // spare slot r: tap kx = r of patch row 20
}

struct Params {
  const float* src;
  const u32x4* planes;   // [chunk][plane][lh][96 rows]
  float* dst;
  float* dump;           // DUMP_BYTES of scratch
  const float* bias;     // nullable
  const float* zero;
  int F, N, H, W, My, Mx, pady, padx;
  int XG, IB, tiles;
  int relu;
  int diag;   // experiments (CONVNET_GFC_DIAG, -DCONVNET_DIAG builds): 1 = every store into the dump area, 2 = no patch refill after the first tile, 4 = no store instructions
};

}  // namespace gfc

// the bank as the planes gfc_kernel keeps in LDS: rows past F and zero slots are zeros (split8_sat: the filter operand saturates)
__global__ void gfc_planes_kernel(const float* __restrict__ W, u32x4* __restrict__ out, int F) {
  using namespace gfc;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= NCH * 2 * ROWS) return;
  const int f = i % ROWS, lh = (i / ROWS) & 1, c = i / (2 * ROWS);
  Split8 sp = {};
  if (f < F) {
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = slot_k(c, lh, j);
      x[j] = k >= 0 ? W[(size_t)f + (size_t)F * k] : 0.f;
    }
    split8_sat(x, sp);
  }
  u32x4* o = out + ((size_t)c * 6 + lh) * ROWS + f;
  o[0] = sp.h;
  o[2 * ROWS] = sp.m;
  o[4 * ROWS] = sp.l;
}

// RELU: the fused ReLU of the write-out as a compile-time choice (one v_max per value; as a run-time flag it is a compare, a mask merge and a
// select per value in a loop that is bound by instruction issue)
template <bool RELU>
__global__ __launch_bounds__(320, 1) void gfc_kernel(const gfc::Params p) {
  using namespace gfc;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* const lds = reinterpret_cast<char*>(smem);
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int N = p.N;

  // this block's run of tiles: T = (ib * My + oy) * XG + xg, contiguous per block (a block walks along an image row, then down)
  const int G = gridDim.x, q = p.tiles / G, rem = p.tiles % G, b = blockIdx.x;
  const int T0 = b * q + min(b, rem), T1 = T0 + q + (b < rem ? 1 : 0);
  if (T0 >= T1) return;
  int xg = T0 % p.XG, oy = (T0 / p.XG) % p.My, ib = T0 / (p.XG * p.My);
  auto next_tile = [&]() __attribute__((always_inline)) {
    if (++xg == p.XG) {
      xg = 0;
      if (++oy == p.My) {
        oy = 0;
        ++ib;
      }
    }
  };

  // ---- prologue: the filter bank (all five waves), zeros, then the role split ---------------------------------------------------
  for (int i = wave; i < A_PIECES; i += 5) {
    const char* g = reinterpret_cast<const char*>(p.planes) + (size_t)i * 1024 + lane * 16;
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)(lds + A_OFF + i * 1024), 16, 0, 0);
  }
  if (tid * 16 < ZERO_BYTES) *reinterpret_cast<u32x4*>(lds + ZERO_OFF + tid * 16) = u32x4{0, 0, 0, 0};
  if (tid < BIAS_BYTES / 4) *reinterpret_cast<float*>(lds + BIAS_OFF + tid * 4) = p.bias && tid < p.F ? p.bias[tid] : 0.f;

  if (wave == 4) {
    // ================================ producer ================================
    // piece k of patch row r: 16-byte unit l = (column 8 k + (l >> 3), image quad l & 7).  The row's source is a wave-uniform base (scalar
    // arithmetic: channel plane + tap row), the lane adds its column and quad; lanes outside the image read the zero page.
    const int lx = lane >> 3;
    const unsigned lane_off = (unsigned)lx * (unsigned)(N * 4) + (unsigned)(lane & 7) * 16u;
    const size_t plane = (size_t)p.H * p.W * N * 4, rowb = (size_t)p.W * N * 4;
    auto issue_half = [&](auto HH) __attribute__((always_inline)) {
      constexpr int h = decltype(HH)::value;
      constexpr int n = h ? H1_ROWS : H0_ROWS;
      const int y0 = S * oy - p.pady, x0 = S * P * xg - p.padx;
      const char* const origin = reinterpret_cast<const char*>(p.src) + ((ptrdiff_t)y0 * p.W + x0) * (ptrdiff_t)(N * 4) + (ptrdiff_t)ib * (IMG * 4);
      bool xok[RP];
#pragma unroll
      for (int k = 0; k < RP; ++k) xok[k] = 8 * k + lx < PW && (unsigned)(x0 + 8 * k + lx) < (unsigned)p.W;
#pragma unroll
      for (int idx = 0; idx < n; ++idx) {
        const int r = h ? 10 + idx : idx == 10 ? 20 : idx;
        const int ch = r / KY, ky = r % KY;
        const bool yok = (unsigned)(y0 + ky) < (unsigned)p.H;   // wave-uniform
        const char* const rowp = origin + (size_t)ch * plane + (size_t)ky * rowb;
#pragma unroll
        for (int k = 0; k < RP; ++k) {
          const char* g = yok && xok[k] ? rowp + (size_t)(8 * k) * (size_t)(N * 4) + lane_off : reinterpret_cast<const char*>(p.zero);
          __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)(lds + PATCH_OFF + (h ? H0_BYTES : 0) + idx * ROWB + k * 1024), 16, 0, 0);
        }
      }
    };
    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, 1>;
    issue_half(H0{});
    issue_half(H1{});
    __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0) lgkmcnt(0): bank, first patch
    __syncthreads();                      // start
    for (int T = T0; T < T1; ++T) {
      const bool more = T + 1 < T1 && !(p.diag & 2);
      // S_a: half 1 of this tile has landed (issued behind the previous S_b, or in the prologue); half 0 is free after it
      __builtin_amdgcn_s_waitcnt(0x0070);
      __syncthreads();
      next_tile();
      if (more) issue_half(H0{});
      // S_b: half 0 of the next tile has landed; half 1 is free after it
      __builtin_amdgcn_s_waitcnt(0x0070);
      __syncthreads();
      if (more) issue_half(H1{});
    }
    __builtin_amdgcn_s_waitcnt(0x0070);
    return;
  }

  // ================================ consumers ================================
  // wave w: output pixel w of the tile, 96 rows x 32 images; lane (li, lh): image li, k-group lh
  const int li = lane & 31, lh = lane >> 5;
  const unsigned base_l = (unsigned)(PATCH_OFF + (S * wave) * XB + li * 4 + lh * XB);
  const unsigned a_l = (unsigned)(A_OFF + (lh * ROWS + li) * 16);
  const unsigned lh_mask = lh ? 0xFFFFFFFFu : 0u;

  f32x16 acc[2][3];             // this tile's sums and the previous tile's, on their way out
  // filter fragments: the h plane of this chunk and the next (it is used up to the chunk's last product); the m and l planes are read for
  // the next chunk into the same registers as soon as their last product has issued (product order below) — 48 registers instead of 72
  u32x4 fah[2][3], fam[3], fal[3];
  Split8 fb[2];                 // the split source column of this chunk and the next
  float bv[8];
  // (CHIP_HERE: a compiler-level memory barrier, no instruction.  Between the two barriers of a tile nothing else stops the optimizer
  // from hoisting the LDS reads of four chunks to the top of the tile — it did, and spilled them)
#ifndef CONVNET_EMU
#define CHIP_HERE() asm volatile("" ::: "memory")
#else
#define CHIP_HERE() ((void)0)
#endif
  auto read_a = [&](auto CC, auto PL, u32x4 (&dst)[3]) __attribute__((always_inline)) {   // plane PL (0 h, 1 m, 2 l) of chunk CC
    constexpr int c = decltype(CC)::value, pl = decltype(PL)::value;
    CHIP_HERE();
#pragma unroll
    for (int t = 0; t < 3; ++t) dst[t] = *reinterpret_cast<const u32x4*>(lds + a_l + c * A_CHUNK + pl * (2 * ROWS * 16) + t * 512);
  };
  // the raw source values of chunk c (one image per lane, eight k-slots) into bv
  auto read_b = [&](auto CC) __attribute__((always_inline)) {
    constexpr int c = decltype(CC)::value;
    unsigned lhm = lh_mask;
    CHIP_HERE();
#ifndef CONVNET_EMU
    asm volatile("" : "+v"(lhm));
#endif
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = 2 * c + (j >> 2);
      const int imm_n = row_off(r) + 2 * (j & 3) * XB;   // kx = 2 (j & 3) + lh: the lane base carries lh
      if ((j & 3) != 3) {
        bv[j] = *reinterpret_cast<const float*>(lds + base_l + imm_n);
      } else {
        // lh = 0: kx = 6 of row r; lh = 1: the spare slot — tap r of patch row 20, or zeros
        // (as base + immediate + a masked constant, the mask made opaque per chunk: left to itself the compiler keeps all twenty
        // selected addresses of a tile in registers across the loop and spills them)
        const int imm_s = r < KX ? row_off(20) + (r - 1) * XB : (ZERO_OFF - PATCH_OFF);
        bv[j] = *reinterpret_cast<const float*>(lds + (base_l + (lhm & (unsigned)(imm_s - imm_n))) + imm_n);
      }
    }
  };
  // pair q of split8 of the raw values in bv (gather_gemm.h: split8, one pair at a time so that a step can carry one)
  auto split_pair = [&](int s, int q) __attribute__((always_inline)) {
    const float x0 = bv[2 * q], x1 = bv[2 * q + 1];
    const unsigned H = pk_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(H << 16), r1 = x1 - __uint_as_float(H & 0xffff0000u);
    const unsigned M = pk_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(M << 16), s1 = r1 - __uint_as_float(M & 0xffff0000u);
    fb[s].h[q] = H;
    fb[s].m[q] = M;
    fb[s].l[q] = pk_bf16(s0, s1);
    // "this pair is complete HERE" (its first use is a chunk away: unpinned, the optimizer sinks all four pairs to the end of the chunk)
#ifndef CONVNET_EMU
    asm volatile("" ::"v"(fb[s].h[q]), "v"(fb[s].m[q]), "v"(fb[s].l[q]));
#endif
  };
  // product k over the three accumulators.  The six products of split_mac in an order that retires the l plane of the filter after
  // k = 1 and the m plane after k = 3: (h,l) (l,h) (m,m) (m,h) (h,m) (h,h) — small terms first, the leading one last, as there.
  auto mac_step = [&](auto K, int s, f32x16 (&a)[3]) __attribute__((always_inline)) {
    constexpr int k = decltype(K)::value;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const u32x4& av = k == 1 ? fal[t] : k == 2 || k == 3 ? fam[t] : fah[s][t];
      const u32x4& bw = k == 0 ? fb[s].l : k == 2 || k == 4 ? fb[s].m : fb[s].h;
      a[t] = mma_bf16(av, bw, a[t]);
    }
    // "these three MFMAs are HERE": the accumulators pass through an empty volatile statement, which keeps its place among the
    // loads, stores and fences of the step (an MFMA has no side effect: without this the optimizer is free to collect the MFMAs of
    // four chunks behind the tile's first barrier and spill every operand they wait for — it did)
#ifndef CONVNET_EMU
    asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]));
#endif
  };

  // ---- write-out of one tile, 16 bytes per lane.  The MFMA layout gives a lane ONE image (column li) and 16 rows per accumulator; four
  // registers 4g .. 4g + 3 are four consecutive rows.  A 4 x 4 transpose inside each quad of lanes (two DPP butterfly stages, 16 VALU)
  // turns them into four consecutive IMAGES of one row per lane: lane q = li & 3 of a quad ends up with row 32 t + 8 g + q + 4 lh, images
  // 4 (li >> 2) .. + 3 — one global_store_dwordx4 per group instead of four dword stores (a vector-memory instruction costs a busy CU
  // ~50-60 cycles whatever its width: 12 stores per lane and tile instead of 48).
  // Address = a wave-uniform row base (scalar: 32 t + 8 g) + the lane's 32-bit byte offset (pixel, image quad, its q + 4 lh rows).
  // Row groups past F (F % 8 == 0) and a tile that does not exist (the one before the first, a pixel past the end of the image row)
  // store into the dump area instead: no branch, the store always happens.
  struct Out {
    unsigned lane_bytes;   // of the real destination
    int flim;              // rows below flim exist (0: nothing of this tile does)
  };
  const size_t fstride = (size_t)p.My * p.Mx * N;   // floats per output row
  const int q4 = li & 3;
  auto out_of = [&](int xg_, int oy_, int ib_, bool exists) __attribute__((always_inline)) {
    const int ox = P * xg_ + wave;
    Out o;
    o.lane_bytes = (unsigned)((((size_t)oy_ * p.Mx + ox) * N + ib_ * IMG + 4 * (li >> 2) + (size_t)(q4 + 4 * lh) * fstride) * 4);
    o.flim = exists && ox < p.Mx ? p.F : 0;
#ifdef CONVNET_DIAG
    if (p.diag & 1) o.flim = 0;
#endif
    return o;
  };
  const unsigned bias_q = (unsigned)(BIAS_OFF + (q4 + 4 * lh) * 4);
  auto bias_of = [&](auto GG) __attribute__((always_inline)) {   // the bias of this lane's row in group GG = 4 t + g, from LDS
    constexpr int gi = decltype(GG)::value;
    return *reinterpret_cast<const float*>(lds + bias_q + (32 * (gi >> 2) + 8 * (gi & 3)) * 4);
  };
  // lane-pair and lane-quad exchanges (quad_perm [1,0,3,2] and [2,3,0,1])
  auto swap1 = [](float v) __attribute__((always_inline)) {
#ifndef CONVNET_EMU
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
#else
    return emu::shfl_xor(v, 1);
#endif
  };
  auto swap2 = [](float v) __attribute__((always_inline)) {
#ifndef CONVNET_EMU
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
#else
    return emu::shfl_xor(v, 2);
#endif
  };
  const bool odd1 = (li & 1) != 0, odd2 = (li & 2) != 0;
  auto store_group = [&](auto GG, const Out& o, const f32x16 (&a)[3], float bias) __attribute__((always_inline)) {
    constexpr int gi = decltype(GG)::value, t = gi >> 2, g = gi & 3;
    constexpr int cf = 32 * t + 8 * g;
    // M[lane q][r] = a[t][4g + r]  ->  v[j] = M[j][q]
    float x0 = a[t][4 * g], x1 = a[t][4 * g + 1], x2 = a[t][4 * g + 2], x3 = a[t][4 * g + 3];
    {   // stage 1, partner lane ^ 1: (x0, x1) and (x2, x3)
      const float r01 = swap1(odd1 ? x0 : x1), r23 = swap1(odd1 ? x2 : x3);
      x0 = odd1 ? r01 : x0; x1 = odd1 ? x1 : r01;
      x2 = odd1 ? r23 : x2; x3 = odd1 ? x3 : r23;
    }
    {   // stage 2, partner lane ^ 2: (x0, x1) <-> (x2, x3)
      const float ra = swap2(odd2 ? x0 : x2), rb = swap2(odd2 ? x1 : x3);
      x0 = odd2 ? ra : x0; x1 = odd2 ? rb : x1;
      x2 = odd2 ? x2 : ra; x3 = odd2 ? x3 : rb;
    }
    f32x4 v = {x0 + bias, x1 + bias, x2 + bias, x3 + bias};
    if constexpr (RELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);   // (NaN -> 0 like `v > 0 ? v : 0`; no branch: one would cut the chunk's basic block)
    }
    const bool real = cf < o.flim;                                                     // wave-uniform
    char* const rowb = real ? reinterpret_cast<char*>(p.dst) + (size_t)cf * fstride * 4 : reinterpret_cast<char*>(p.dump);
    const unsigned off = real ? o.lane_bytes : (unsigned)lane * 16u;
#ifdef CONVNET_DIAG
    if (p.diag & 4) return;   // (diag 4: the transposes without the store instruction)
#endif
    *reinterpret_cast<f32x4*>(rowb + off) = v;
  };

  // One chunk = six fenced steps of three MFMAs.  Step 0 carries the next chunk's source reads and its h plane (the m and l planes follow
  // in steps 4 and 2), steps 1-4 one pair each of its split
  // (a pair is ~11 VALU: under four per MFMA), and step 5 (no split there) one 16-byte store group of the PREVIOUS tile — 12 groups over
  // 10 chunks: the last two ride in step 0 of chunks 8 and 9 — with one consumer wave per SIMD nothing else hides them.
  auto chunk_body = [&](auto CC, bool has_next, f32x16 (&cur)[3], const f32x16 (&prev)[3], const Out& po) __attribute__((always_inline)) {
    constexpr int c = decltype(CC)::value, s = c & 1;
    using CN = std::integral_constant<int, (c + 1) % NCH>;
    __builtin_amdgcn_sched_barrier(0);
    if (has_next) {
      read_b(CN{});
      read_a(CN{}, std::integral_constant<int, 0>{}, fah[s ^ 1]);
    }
    CHIP_HERE();
    // the bias values of this chunk's store groups, read with the operands: a read next to its use would wait on LDS
    const float bs0 = bias_of(std::integral_constant<int, c>{});
    float bs1 = 0.f;
    if constexpr (c >= 8) bs1 = bias_of(std::integral_constant<int, c + 2>{});
    static_for<0, 6>([&](auto KK) __attribute__((always_inline)) {
      constexpr int k = decltype(KK)::value;
      if constexpr (k >= 1 && k <= 4) {
        if (has_next) split_pair(s ^ 1, k - 1);
      }
      if constexpr (k == 0 && c >= 8) store_group(std::integral_constant<int, c + 2>{}, po, prev, bs1);
      if constexpr (k == 5) store_group(std::integral_constant<int, c>{}, po, prev, bs0);
      if constexpr (k == 2) {
        if (has_next) read_a(CN{}, std::integral_constant<int, 2>{}, fal);   // the l plane's last product (k = 1) has issued
      }
      if constexpr (k == 4) {
        if (has_next) read_a(CN{}, std::integral_constant<int, 1>{}, fam);   // the m plane's (k = 3)
      }
      mac_step(KK, s, cur);
      // inside the step: an MFMA first, then a third of the step's other work behind each (the matrix pipe runs while the VALU / LDS
      // instructions issue; left alone the scheduler puts the 35 VALU of a store group in FRONT of the step's MFMAs)
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);
        __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  auto run_tile = [&](bool more, f32x16 (&cur)[3], const f32x16 (&prev)[3], const Out& po) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) cur[t][e] = 0.f;
    static_for<0, NCH>([&](auto CC) __attribute__((always_inline)) {
      constexpr int c = decltype(CC)::value;
      if constexpr (c == 4) __syncthreads();          // S_a: every wave holds chunk 4's operands (the last of half 0); half 1 has landed
      if constexpr (c == NCH - 1) __syncthreads();    // S_b: ... chunk 9's (the last of half 1); half 0 of the next tile has landed
      if constexpr (c + 1 < NCH) chunk_body(CC, true, cur, prev, po);
      else chunk_body(CC, more, cur, prev, po);
    });
  };

  __builtin_amdgcn_s_waitcnt(0x0070);   // this wave's share of the bank
  __syncthreads();                      // start: bank, zeros and the first patch are in LDS
  {
    using C0 = std::integral_constant<int, 0>;
    read_b(C0{});
    read_a(C0{}, std::integral_constant<int, 0>{}, fah[0]);
    read_a(C0{}, std::integral_constant<int, 1>{}, fam);
    read_a(C0{}, std::integral_constant<int, 2>{}, fal);
  }
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4) split_pair(0, q4);
  Out po = out_of(0, 0, 0, false);      // "the tile before the first": its stores go to the dump area
  int T = T0;
  for (;;) {
    // two tiles per turn, so that which accumulator set is filled and which is written out is a compile-time matter
    run_tile(T + 1 < T1, acc[0], acc[1], po);
    po = out_of(xg, oy, ib, true);
    next_tile();
    if (++T >= T1) {
      static_for<0, NGROUP>([&](auto GG) __attribute__((always_inline)) { store_group(GG, po, acc[0], bias_of(GG)); });
      break;
    }
    run_tile(T + 1 < T1, acc[1], acc[0], po);
    po = out_of(xg, oy, ib, true);
    next_tile();
    if (++T >= T1) {
      static_for<0, NGROUP>([&](auto GG) __attribute__((always_inline)) { store_group(GG, po, acc[1], bias_of(GG)); });
      break;
    }
  }
}

// Takes conv_up_impl's launch when the shape is conv1's kind: the bf16-split products, 16-byte-aligned operands, N % 32 == 0,
// 3 channels x 7 x 7, stride 2 both ways, F <= 96 and a multiple of 8, a source small enough for 32-bit plane offsets.  Returns false otherwise.
bool gfc_try(const float* images, const float* filters, const float* bias, float* targets, int N, int C, int H, int W, int F, int Ky, int Kx,
             int sy, int sx, int pady, int padx, int My, int Mx, float scaleTargets, int relu, double flops) {
  using namespace gfc;
  if (!CHIP_KNOB("CONVNET_GG_FEWC", 1) || matrix_path() != 1) return false;
  if (Kx != KX || Ky != KY || C != CH || sy != S || sx != S || F > ROWS || F < 8 || F % 8 != 0 || N % IMG != 0 || My < 1 || Mx < 1) return false;
  if (pady < 0 || padx < 0 || (size_t)C * H * W * N * 4 >= (size_t(1) << 32) || (size_t)F * My * Mx * N * 4 >= (size_t(1) << 32)) return false;
  if (scaleTargets != 0.f) return false;   // accumulating into the target (a layer with several incoming edges): the gather kernels
  if ((reinterpret_cast<uintptr_t>(images) | reinterpret_cast<uintptr_t>(targets)) & 15) return false;
  static bool once = false;
  static int cus = 256;
  if (!once) {
    CHIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gfc_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    CHIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gfc_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
#ifndef CONVNET_EMU
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
#else
    cus = 4;   // tests/emu: a small "chip", so that a block's run is several tiles (the refill of the patch, both barriers)
#endif
    once = true;
  }
  u32x4* planes = static_cast<u32x4*>(workspace_aux(A_BYTES + DUMP_BYTES));
  {
    KernelTimer timer("filter_planes_kernel", "conv_fprop", 0.0, 4.0 * F * C * Ky * Kx + A_BYTES);
    hipLaunchKernelGGL(gfc_planes_kernel, dim3(divup(NCH * 2 * ROWS, 256)), dim3(256), 0, stream(), filters, planes, F);
  }
  Params p{};
  p.src = images; p.planes = planes; p.dst = targets; p.bias = bias; p.zero = zero_page();
  p.dump = reinterpret_cast<float*>(reinterpret_cast<char*>(planes) + A_BYTES);
  p.F = F; p.N = N; p.H = H; p.W = W; p.My = My; p.Mx = Mx; p.pady = pady; p.padx = padx;
  p.XG = divup(Mx, P); p.IB = N / IMG; p.tiles = p.XG * My * p.IB;
  p.relu = relu; p.diag = CHIP_DIAG_KNOB("CONVNET_GFC_DIAG", 0);
  const int grid = std::min(cus, p.tiles);
  {
    KernelTimer timer("gfc_kernel<96x128,split>", "conv_fprop", flops, 0.0, 0.0);
    if (relu) hipLaunchKernelGGL(gfc_kernel<true>, dim3(grid), dim3(320), LDS_BYTES, stream(), p);
    else hipLaunchKernelGGL(gfc_kernel<false>, dim3(grid), dim3(320), LDS_BYTES, stream(), p);
  }
  note_kernel("gfc_kernel(fprop)", flops, grid, 1);
  return true;
}

}  // namespace chip
