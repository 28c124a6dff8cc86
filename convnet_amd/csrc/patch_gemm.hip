// gpp_kernel: the patch-resident, pre-split gather-GEMM for the 3x3 / 5x5 convolutions (conv_edge fprop and dgrad; the reference
// lowers both to im2col + cublasSgemm, cudamat_conv_gemm.cu:545-680 / 682-825, and its direct kernels re-read every image pixel
// per neighbouring module, cudamat_conv_filteracts.cu:985-1140).
//
// ggp_kernel (gather_gemm.hip) gives a block ONE output pixel x 256 images, so every (tap, 16-channel chunk) is a fresh 16 KB
// fetch of raw fp32 source through the CU's vector-memory path, split into bf16 terms by every consumer wave that touches it.
// Here both are gone:
//   * the block tile is kPatchP = 4 neighbouring output pixels x 64 images.  For one tap ROW and one 16-channel block the source
//     pixels the tile needs for ALL taps of that row ("slab": P + taps - 1 pixels, 6 for a 3x3) are staged once; tap b of pixel j
//     reads slot S(j) + b, a shifted LDS address, not a new load.  A 3x3 row costs 6 slots for 12 pixel-taps;
//   * the source arrives ALREADY split: act_planes_kernel writes every activation / derivative tensor once as bf16 planes
//     [plane h/m/l][channel block][k-group][y][x][image] (8 bf16 = 16 bytes per (pixel, image): one MFMA B operand), exactly as
//     filter_planes_kernel does for the filter bank.  Consumers run ds_read_b128 + MFMA + one barrier per chunk, no VALU.
// Structure as ggp_kernel: four consumer waves (2 x 2, 64 rows x 128 columns each) + one producer wave that owns all staging: A
// planes of chunk c+2 into a three-stage ring, the next slab into the idle half of a double buffer, spread over the current
// slab's chunks; every piece is 1 KB = one slot of one (plane, k-group) = one wave-uniform source address + 16 bytes per lane.
// Units: the column space is (64-image block, pixel) units in flat order (GGParams::patch); a tile is 4 consecutive units, so it may
// wrap from one image row to the next (or from one image block to the next): the slot base S(j) then jumps by 3 and the slab needs 8
// slots instead of 6.  Out-of-image slots are zero-filled by the producer (LDS stores, no memory traffic); slots no pixel reads are
// left alone.  Tap rows no pixel of the tile has are skipped.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <string>

#include "gather_gemm.h"

#ifndef CONVNET_GPV_LDS_MAP
#define CONVNET_GPV_LDS_MAP 0   // gpv_kernel: 0 = filter ring, then the two slabs, then the dump slot; 1 = ring stages and slabs interleaved
#endif
#ifndef CONVNET_GPV_FILT_LATE
#define CONVNET_GPV_FILT_LATE 1   // gpv_kernel: the filter pieces in the split-free steps of columns 1 and 2 (0: under column 0, as gpw_kernel)
#endif

namespace chip {

// One pass over an activation / derivative tensor (CHWN fp32, C % 16 == 0, N % 64 == 0): exact three-way bf16 split into the layout
// gpp_kernel stages from: u32x4 [channel block cb][pixel][64-image block ib][region q = plane*2 + k-group lh][image % 64], 8 bf16 =
// channels 16*cb + 2*j + lh of one (pixel, image) — the six 1 KB regions of a (cb, pixel, ib) "slot" are contiguous, so the
// producer moves a slot with ONE address and immediate offsets.  Thread = (cb, lh, pixel, 4 images): reads 8 channels x 4 images,
// writes 3 x 64 B.  Values above the bf16 range saturate to +-bf16 max (split8_sat), as the filter operand does; NaN stays NaN.
__global__ void __launch_bounds__(256) act_planes_kernel(const float* __restrict__ src, u32x4* __restrict__ out, int CB, int HW, int N4) {
  const size_t total = (size_t)CB * 2 * HW * N4;
  const size_t cstride = (size_t)HW * N4 * 4;   // floats per channel
  const int IB = N4 / 16;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int n4 = (int)(i % N4);
    size_t r = i / N4;
    const int pix = (int)(r % HW);
    r /= HW;
    const int lh = (int)(r & 1), cb = (int)(r >> 1);
    const float* sp = src + ((size_t)(16 * cb + lh) * HW + pix) * (size_t)(N4 * 4) + 4 * n4;
    f32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = ld4(sp + 2 * j * cstride);
    u32x4* o = out + ((((size_t)cb * HW + pix) * IB + (n4 >> 4)) * 6 + lh) * 64 + 4 * (n4 & 15);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = v[j][e];
      Split8 s;
      split8_sat(x, s);
      o[e] = s.h;
      o[128 + e] = s.m;
      o[256 + e] = s.l;
    }
  }
}

// The filter bank for ggp_kernel's pre-split build and gpp_kernel: the exact three-way bf16 split of the re-laid bank (Split8), laid
// out per row tile of TH rows: u32x4 [chunk][row tile rt][region q = plane*2 + lh][row % TH] — the 6*TH*16 bytes a block stages per
// chunk are contiguous, in the order they take in LDS, so the producer wave moves them with one address and immediate offsets.
// Rows past R (last tile) are zeros.  forward bank: rows f, chunk = tap + TYX*cb, k-slot j of k-group lh = channel 16*cb + 2*j + lh.
__global__ void filter_planes_rt_kernel(const float* __restrict__ W, u32x4* __restrict__ out, int F, int C, int TYX, int TH) {
  const int RT = (F + TH - 1) / TH;
  const size_t total = (size_t)(C / 16) * TYX * 2 * RT * TH;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int f = (int)(i % (RT * TH));
    const size_t r = i / (RT * TH);
    const int lh = (int)(r & 1);
    const size_t chunk = r >> 1;
    const int tap = (int)(chunk % TYX), cb = (int)(chunk / TYX);
    Split8 sp = {};
    if (f < F) {
      float x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = W[(size_t)f + (size_t)F * (tap + (size_t)TYX * (16 * cb + 2 * j + lh))];
      split8_sat(x, sp);
    }
    u32x4* o = out + ((chunk * RT + f / TH) * 6 + lh) * TH + f % TH;
    o[0] = sp.h;
    o[2 * TH] = sp.m;
    o[4 * TH] = sp.l;
  }
}
// one stride class of the input-gradient bank: rows c, chunk = tap + TYXc*fb, k-slot j of k-group lh = filter 16*fb + 2*j + lh
__global__ void dgrad_filter_planes_rt_kernel(const float* __restrict__ W, u32x4* __restrict__ out, int F, int C, int Ky, int Kx, int cy, int cx,
                                              int sy, int sx, int TYc, int TXc, int TH) {
  const int TYXc = TYc * TXc, RT = (C + TH - 1) / TH;
  const size_t total = (size_t)(F / 16) * TYXc * 2 * RT * TH;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % (RT * TH));
    const size_t r = i / (RT * TH);
    const int lh = (int)(r & 1);
    const size_t chunk = r >> 1;
    const int tap = (int)(chunk % TYXc), fb = (int)(chunk / TYXc);
    const int a = tap / TXc, b = tap - a * TXc;
    Split8 sp = {};
    if (c < C) {
      const float* wp = W + (size_t)F * ((cx + sx * b) + Kx * ((cy + sy * a) + (size_t)Ky * c)) + 16 * fb + lh;
      float x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = wp[2 * j];
      split8_sat(x, sp);
    }
    u32x4* o = out + ((chunk * RT + c / TH) * 6 + lh) * TH + c % TH;
    o[0] = sp.h;
    o[2 * TH] = sp.m;
    o[4 * TH] = sp.l;
  }
}

// the forward bank in its OWN k order (k = ch*TYX + tap, any channel count), K padded to KP = 16 * chunks: rows f, chunk = k / 16,
// k-slot j of k-group lh = k-row 16*chunk + 2*j + lh; rows past F and k-rows past K are zeros.  For ggp_kernel's generic-k mode (conv1).
__global__ void filter_planes_gk_kernel(const float* __restrict__ W, u32x4* __restrict__ out, int F, int K, int KP, int TH) {
  const int RT = (F + TH - 1) / TH;
  const size_t total = (size_t)(KP / 16) * 2 * RT * TH;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int f = (int)(i % (RT * TH));
    const size_t r = i / (RT * TH);
    const int lh = (int)(r & 1);
    const size_t chunk = r >> 1;
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 16 * (int)chunk + 2 * j + lh;
      x[j] = (f < F && k < K) ? W[(size_t)f + (size_t)F * k] : 0.f;
    }
    Split8 sp;
    split8_sat(x, sp);
    u32x4* o = out + ((chunk * RT + f / TH) * 6 + lh) * TH + f % TH;
    o[0] = sp.h;
    o[2 * TH] = sp.m;
    o[4 * TH] = sp.l;
  }
}
void filter_planes_gk_launch(const float* W, void* out, int F, int K, int KP, int TH, const char* op) {
  const size_t work = (size_t)(KP / 16) * 2 * divup(F, TH) * TH;
  int nb = (int)((work + 255) / 256);
  if (nb > 2048) nb = 2048;
  KernelTimer timer("filter_planes_kernel", op, 0.0, 10.0 * (double)F * K);
  hipLaunchKernelGGL(filter_planes_gk_kernel, dim3(nb), dim3(256), 0, stream(), W, static_cast<u32x4*>(out), F, K, KP, TH);
}

void filter_planes_rt_launch(const PatchBank& b, void* out, int TYX, int TH, const char* op) {
  const int R = b.dgrad ? b.C : b.F, KCn = b.dgrad ? b.F : b.C;
  const size_t work = (size_t)(KCn / 16) * TYX * 2 * divup(R, TH) * TH;
  int nb = (int)((work + 255) / 256);
  if (nb > 2048) nb = 2048;
  KernelTimer timer(b.dgrad ? "dgrad_filter_planes_kernel" : "filter_planes_kernel", op, 0.0, 10.0 * (double)R * KCn * TYX);
  if (b.dgrad)
    hipLaunchKernelGGL(dgrad_filter_planes_rt_kernel, dim3(nb), dim3(256), 0, stream(), b.W, static_cast<u32x4*>(out), b.F, b.C, b.Ky, b.Kx, b.cy,
                       b.cx, b.sy, b.sx, b.TYc, b.TXc, TH);
  else
    hipLaunchKernelGGL(filter_planes_rt_kernel, dim3(nb), dim3(256), 0, stream(), b.W, static_cast<u32x4*>(out), b.F, b.C, TYX, TH);
}

// LDS position (16-byte units inside a 1 KB slot) of image `img` of the slot's 64: rotated inside each group of 16 by the group
// index, so that the consumers' ds_read_b128 — lane li reads image NTC*(li % 16) + u, lanes 16..31 the next slot — touch 16
// distinct bank quads in every 16-lane access group (natural order: 4-way conflict).  The producer applies it through its per-lane
// source address; global memory keeps the natural image order.
__device__ __forceinline__ int slot_pos(int img) { return (img & 48) | ((img + (img >> 4)) & 15); }
__device__ __forceinline__ int slot_img(int pos) { return (pos & 48) | ((pos - (pos >> 4)) & 15); }

// BRAW: the source slab is staged as RAW fp32 ([slot][16 k-rows][64 images], 4 KB per slot, straight from the CHWN tensor — no
// planes pass, 4 pieces per slot instead of 6) and split by the consumers in the MFMA shadows exactly as ggp_kernel does; !BRAW: bf16
// planes from act_planes_kernel, consumers without VALU.
template <int WR, int WC, int MT, int CW, bool BRAW>
__global__ __launch_bounds__(WR* WC * 64 + 64, 2) void gpp_kernel(const GGParams pin, const GGClassTable ct) {
  constexpr int NC = WR * WC * 64;   // consumer threads
  constexpr int NTC = CW / 32;
  using fvec = __attribute__((ext_vector_type(NTC))) float;
  constexpr int ROWS = WR * MT * 32;
  constexpr int A_STAGE = 6 * ROWS * 4;   // floats: 3 planes x 2 k-groups x ROWS x 16 bytes
  constexpr int STA = 3;                  // A ring
  constexpr int NS = 8;                   // slots of a slab
  constexpr int PS = BRAW ? 4 : 6;        // 1 KB pieces per slot: 16 k-rows x 64 images x 4 B, or 6 (plane, k-group) regions
  constexpr int SLAB = PS * NS * 256;     // floats per slab
  constexpr int NA = A_STAGE / 4 / 64;    // producer instructions per A chunk
  constexpr int NBP = PS * NS;            // ... per slab
  constexpr int UW = CW / 64;             // units per wave-column
  static_assert(WC * UW == kPatchP && NTC == 4, "tile = kPatchP units of 64 images; slot_pos is the NTC = 4 swizzle");
  static_assert(NA + NBP < 64 && NA % 2 == 0, "vmcnt immediate; batch sizes are even");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                   // [STA][A_STAGE]
  float* Bs = smem + STA * A_STAGE;   // [2][SLAB]

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

  // ---- the tile's units (wave-uniform) ----------------------------------------------------------------------------------
  const int G = T.G, GX = T.GX, units = p.IB * G;
  int u_ib[kPatchP], u_oy[kPatchP], u_ox[kPatchP], S[kPatchP];
  bool u_ok[kPatchP];
  int ys_f = 1 << 30, ys_l = -(1 << 30);
#pragma unroll
  for (int j = 0; j < kPatchP; ++j) {
    const int U = col_tile * kPatchP + j;
    u_ok[j] = U < units;
    const int ib = u_ok[j] ? U / G : 0, m = u_ok[j] ? U - ib * G : 0;
    u_ib[j] = ib;
    u_oy[j] = m / GX;
    u_ox[j] = m - u_oy[j] * GX;
    if (u_ok[j]) {
      const int ys0 = u_oy[j] * p.ssy + T.y0;
      ys_f = min(ys_f, ys0);
      ys_l = max(ys_l, ys0);
    }
    // slot base: the next pixel of the same image row shares all but one slot with its neighbour; anything else starts a fresh run
    S[j] = j == 0 ? 0 : S[j - 1] + (!u_ok[j] ? 0 : (u_ib[j] == u_ib[j - 1] && u_oy[j] == u_oy[j - 1]) ? 1 : 3);
  }

  // ---- reduction range in SUPERCHUNKS: (16-channel block cb, tap row a, tap group g); taps of the group innermost -------------
  const int TX = T.TX, TYX = T.TYX, TYn = TYX / TX;
  const int ng = p.ng, gc0 = p.gcnt[0], gc1 = p.gcnt[1], gbase0 = p.gb0[0], gbase1 = p.gb0[1];
  int a_lo = 0, a_hi = TYn - 1;
  const bool skip = tsplit < 0 && p.splits == 1;
  if (skip) {   // tap rows that exist for some pixel of the tile (ggp_kernel's border-tap skipping, rows only)
    if (p.dir > 0) { a_lo = max(0, -ys_l); a_hi = min(TYn - 1, p.SH - 1 - ys_f); }
    else { a_lo = max(0, ys_f - (p.SH - 1)); a_hi = min(TYn - 1, ys_l); }
  }
  const int nrow = max(0, a_hi - a_lo + 1);
  const int nsc_all = (p.KC / BK) * nrow * ng;
  int sc_beg = 0, sc_end = nsc_all;
  if (!skip) {
    const int cps = tsplit >= 0 ? p.tail_cps : p.chunks_per_split;
    sc_beg = min(nsc_all, (tsplit >= 0 ? tsplit : split) * cps);
    sc_end = min(nsc_all, sc_beg + cps);
  }
  const int nsc = sc_end - sc_beg;
  // chunks of the range: superchunk sc belongs to group sc % ng
  int nchunks;
  if (ng == 1) nchunks = nsc * gc0;
  else {
    const int n1 = (sc_end >> 1) - (sc_beg >> 1);   // odd indices in [sc_beg, sc_end)
    nchunks = (nsc - n1) * gc0 + n1 * gc1;
  }


  if (wave == WR * WC) {
    // ================================ producer wave ================================
    __builtin_amdgcn_s_setprio(3);
    if (nchunks == 0) {
      __builtin_amdgcn_s_barrier();
      return;
    }
    const char* const planes = reinterpret_cast<const char*>(p.planes);
    const unsigned lane_off = (unsigned)slot_img(lane) * 16u;   // planes: this lane's 16 bytes of a region's 1 KB source run
    // raw: a piece is 4 k-rows (channels) x 64 images; lane = (k-row of the piece, image quad)
    const size_t ch_bytes = (size_t)p.SH * p.SW * N * 4;
    const size_t lane_off_raw = (size_t)(lane >> 4) * ch_bytes + (size_t)(lane & 15) * 16;
    const char* const rawsrc = reinterpret_cast<const char*>(p.src);
    const int dir = p.dir, SH = p.SH, SW = p.SW, ssy = p.ssy, ssx = p.ssx;
#ifdef CONVNET_DIAG
    const int diag = p.prio;   // timing diagnostics (results wrong): 1 = no slab loads after the prologue, 2 = no A loads after it
#else
    constexpr int diag = 0;
#endif
    // Every LDS-DMA instruction of this wave is `global_load_lds_dwordx4 v_lane, s[base] offset:imm` with M0 rewritten once per FOUR
    // pieces: source and destination advance by the same 1 KB per piece (the instruction's immediate offset applies to both), because
    // act_planes_kernel / filter_planes_rt_kernel lay a slot's six regions and a chunk's twelve pieces out contiguously.  An
    // instruction that rewrites M0 costs the issuing wave ~2.7x one that does not (tools/dma_issue, profiles/r02_kernel_experiments.md).
    const unsigned a_lane = (unsigned)lane * 16u;
    const char* const abase0 = reinterpret_cast<const char*>(T.A) + (size_t)row_tile * (6 * ROWS * 16);
    const size_t a_chunk_bytes = (size_t)p.row_tiles * (6 * ROWS * 16);

    // A iterator: two chunks ahead of the consumers.  (cb, a, g, i) -> filter chunk cb*TYX + a*TX + b, b = gb0[g] + i*dir*ssx
    int A_cb, A_a, A_g, A_i, A_left = nchunks;
    {
      A_g = sc_beg % ng;
      const int r = sc_beg / ng;
      A_a = a_lo + (nrow > 0 ? r % nrow : 0);
      A_cb = nrow > 0 ? r / nrow : 0;
      A_i = 0;
    }
    auto issue_a = [&](int stage) __attribute__((always_inline)) {
      const int b = (A_g ? gbase1 : gbase0) + A_i * dir * ssx;
      const char* abase = abase0 + a_chunk_bytes * (size_t)(A_cb * TYX + A_a * TX + b);   // wave-uniform
      const unsigned lda0 = lds_addr(As + stage * A_STAGE);
#pragma unroll
      for (int it = 0; it < NA; it += 4) {
        lds_dma4(a_lane, abase, abase, abase, abase, lda0 + 1024u * it);
        abase += 4096;
      }
      --A_left;
      if (++A_i == (A_g ? gc1 : gc0)) {
        A_i = 0;
        if (++A_g == ng) {
          A_g = 0;
          if (++A_a > a_hi) {
            A_a = a_lo;
            ++A_cb;
          }
        }
      }
    };

    // slab iterator: one superchunk ahead.  Lane s < NS describes slot s: the source pixel of (unit j, tap slot i) with S[j] + i == s
    int B_cb, B_a, B_g;
    {
      B_g = sc_beg % ng;
      const int r = sc_beg / ng;
      B_a = a_lo + (nrow > 0 ? r % nrow : 0);
      B_cb = nrow > 0 ? r / nrow : 0;
    }
    constexpr unsigned kNoSlot = 0xFFFFFFFFu;     // no pixel of the tile reads this slot for this group: left alone
    constexpr unsigned kZeroSlot = 0xFFFFFFFEu;   // read, but outside the image: filled with zeros
    const int IBn = p.IB;
    auto slot_desc = [&]() __attribute__((always_inline)) {   // -> planes: (pixel*IB + ib) of the slot's source, in slots of 6 KB; raw: float index of (pixel, image ib*64) in a channel
      const int s = lane & (NS - 1);
      const int cnt = B_g ? gc1 : gc0, b0 = B_g ? gbase1 : gbase0;
      unsigned off = kNoSlot;
#pragma unroll
      for (int j = 0; j < kPatchP; ++j) {
        const int i = s - S[j];
        if (u_ok[j] && i >= 0 && i < cnt) {
          const int ys = u_oy[j] * ssy + T.y0 + dir * B_a;
          const int xs = u_ox[j] * ssx + T.x0 + dir * b0 + i * ssx;
          const bool in = (unsigned)ys < (unsigned)SH && (unsigned)xs < (unsigned)SW;
          off = !in ? kZeroSlot : BRAW ? (unsigned)((ys * SW + xs) * N + u_ib[j] * 64) : (unsigned)((ys * SW + xs) * IBn + u_ib[j]);
        }
      }
      return off;
    };
    auto slab_next = [&]() __attribute__((always_inline)) {
      if (++B_g == ng) {
        B_g = 0;
        if (++B_a > a_hi) {
          B_a = a_lo;
          ++B_cb;
        }
      }
    };
    // slots [S0, S1) of a slab; slot s lands at LDS offset PS*s KB (planes: regions q = plane*2 + k-group 1 KB apart, as in memory;
    // raw: k-rows 256 B apart).  Returns the number of loads issued (PS per in-image slot).
    const size_t cb_bytes = BRAW ? 16 * ch_bytes : (size_t)SH * SW * IBn * 6144;   // one 16-channel block of the source
    const unsigned zero_lds = lds_addr(Bs) + (unsigned)lane * 16u;
    // Issue order of a slab's slots: first the (up to four) slots tap slot 0 reads, S[j]; then S[j] + 1, then S[j] + 2.  The first
    // four must have landed when the superchunk starts; the rest is first read by its SECOND chunk, so it may be issued as late as
    // the last chunk of the superchunk before: every chunk of a 3-tap superchunk then carries two slots (12 pieces) of the next
    // slab beside its 12 A pieces — the same smooth stream the A ring has — instead of 4 / 2 / 0.
    int ord[NS];
    {
      unsigned seen = 0;
      int no = 0;
#pragma unroll
      for (int q = 0; q < NS; ++q) ord[q] = -1;
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < kPatchP; ++j) {
          const int sl = S[j] + i;
          if (u_ok[j] && sl < NS && !((seen >> sl) & 1)) {
            seen |= 1u << sl;
#pragma unroll
            for (int q = 0; q < NS; ++q)
              if (q == no) ord[q] = sl;
            ++no;
          }
        }
    }
    // positions [P0, P1) of that order
    auto issue_slab = [&](auto P0c, auto P1c, int buf, int cb, unsigned soff) __attribute__((always_inline)) {
      constexpr int P0 = decltype(P0c)::value, P1 = decltype(P1c)::value;
      int n = 0;
      static_for<P0, P1>([&](auto PP) __attribute__((always_inline)) {
        const int sl = ord[decltype(PP)::value];
        if (sl < 0) return;
        const unsigned so = (unsigned)__builtin_amdgcn_readlane((int)soff, sl);
        if (so == kNoSlot) return;
        float* const ld = Bs + buf * SLAB + 256 * PS * sl;
        if (so == kZeroSlot) {
          const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
          for (int q = 0; q < PS; ++q)
            lds_store16(zero_lds + (unsigned)((buf * SLAB + 256 * (PS * sl + q)) * 4), z);
          return;
        }
        if constexpr (BRAW) {
          // four pieces of 4 k-rows x 64 images: one M0 write, SGPR bases 4 channel planes apart (minus the 1 KB the immediate offset adds)
          const char* const base = rawsrc + (size_t)cb * cb_bytes + (size_t)so * 4;   // wave-uniform: k-row 0 of the slot
          const size_t d4 = 4 * ch_bytes - 1024;
          lds_dma4((unsigned)lane_off_raw, base, base + d4, base + 2 * d4, base + 3 * d4, lds_addr(ld));
        } else {
          const char* const base = planes + (size_t)cb * cb_bytes + (size_t)so * 6144;   // wave-uniform
          const unsigned l0 = lds_addr(ld);
          lds_dma4(lane_off, base, base, base, base, l0);
          lds_dma2(lane_off, base + 4096, base + 4096, l0 + 4096u);
        }
        n += PS;
      });
      return n;
    };
    using I0 = std::integral_constant<int, 0>;
    using I2 = std::integral_constant<int, 2>;
    using I4 = std::integral_constant<int, 4>;
    using I8 = std::integral_constant<int, NS>;
    // wait until at most n of this wave's loads are in flight (n = what the current batch issued, always even), and for its LDS
    // zero-fills.  vmcnt: low four bits in [3:0], high two in [15:14]; expcnt untouched; lgkmcnt(0) in [11:8].
    auto wait_all_but = [&](int n) __attribute__((always_inline)) {
#define GPP_VMCNT(n) (((n) & 15) | (((n) >> 4) << 14) | 0x0070)
#define GPP_CASE(k) case k: __builtin_amdgcn_s_waitcnt(GPP_VMCNT(k)); break;
      switch (n) {
        GPP_CASE(0) GPP_CASE(2) GPP_CASE(4) GPP_CASE(6) GPP_CASE(8) GPP_CASE(10) GPP_CASE(12) GPP_CASE(14) GPP_CASE(16) GPP_CASE(18)
        GPP_CASE(20) GPP_CASE(22) GPP_CASE(24) GPP_CASE(26) GPP_CASE(28) GPP_CASE(30) GPP_CASE(32) GPP_CASE(34) GPP_CASE(36) GPP_CASE(38)
        GPP_CASE(40) GPP_CASE(42) GPP_CASE(44) GPP_CASE(46) GPP_CASE(48) GPP_CASE(50) GPP_CASE(52) GPP_CASE(54) GPP_CASE(56) GPP_CASE(58)
        default: __builtin_amdgcn_s_waitcnt(GPP_VMCNT(60)); break;
      }
#undef GPP_CASE
#undef GPP_VMCNT
    };

    // prologue: slab 0, A of chunks 0 and 1
    {
      const unsigned so = slot_desc();
      issue_slab(I0{}, I8{}, 0, B_cb, so);
      slab_next();
    }
    issue_a(0);
    const bool two = A_left > 0;
    if (two) issue_a(1);
    wait_all_but(two ? NA : 0);
    __builtin_amdgcn_s_barrier();
    int fill = 2, buf = 0;
    for (int sc = sc_beg; sc < sc_end; ++sc) {
      const int cnt = ((sc % ng) ? gc1 : gc0);
      const bool has_next = sc + 1 < sc_end && !(diag & 1);
      unsigned so = 0;
      const int ncb = B_cb;
      if (has_next) {
        so = slot_desc();
        slab_next();
      }
      for (int t = 0; t < cnt; ++t) {
        int n = 0;
        if (A_left > 0 && !(diag & 2)) {
          issue_a(fill);
          n += NA;
        }
        fill = fill == STA - 1 ? 0 : fill + 1;
        if (has_next) {
          // the next slab goes into the buffer the previous superchunk used: tap slot 0's slots before the last chunk, the rest in it
          if (cnt == 2) {
            if (t == 0) n += issue_slab(I0{}, I4{}, buf ^ 1, ncb, so);
            else n += issue_slab(I4{}, I8{}, buf ^ 1, ncb, so);
          } else {
            if (t == 0) n += issue_slab(I0{}, I2{}, buf ^ 1, ncb, so);
            else if (t == 1) n += issue_slab(I2{}, I4{}, buf ^ 1, ncb, so);
            else n += issue_slab(I4{}, I8{}, buf ^ 1, ncb, so);
          }
        }
        wait_all_but(n);   // everything issued before this chunk's batch has landed: the next chunk's A and what it reads of the slabs
        __builtin_amdgcn_s_barrier();
      }
      buf ^= 1;
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

  // this lane's unit and its slot base; its NTC columns are images NTC*(li % 16) + u of that unit
  int S_l = 0;
  {
    const int j = wc * UW + (NTC * li) / 64;
#pragma unroll
    for (int q = 0; q < kPatchP; ++q)
      if (q == j) S_l = S[q];
  }
  // planes: u32x4 units inside a slab: slot base (6 regions of 64 per slot) + k-group region + swizzled image position
  // raw: float units: slot base (16 k-rows of 64 per slot) + k-row lh + image
  int boff[NTC];
#pragma unroll
  for (int u = 0; u < NTC; ++u)
    boff[u] = BRAW ? S_l * 1024 + lh * 64 + (NTC * li) % 64 + u : S_l * 384 + lh * 64 + slot_pos((NTC * li) % 64 + u);
  const u32x4* const Bs4 = reinterpret_cast<const u32x4*>(Bs);

  __syncthreads();   // slab 0 and A chunk 0 have landed
  if (nchunks > 0) {
    auto load_a = [&](int st, Split8 (&fa)[MT]) __attribute__((always_inline)) {
      const u32x4* ap = reinterpret_cast<const u32x4*>(As + st * A_STAGE) + lh * ROWS + wr * MT * 32 + li;
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        fa[t].h = ap[t * 32];
        fa[t].m = ap[2 * ROWS + t * 32];
        fa[t].l = ap[4 * ROWS + t * 32];
      }
    };
    int stage = 0, buf = 0, ti = 0, g = sc_beg % ng, cnt = g ? gc1 : gc0;
    // the next chunk: next tap slot of this slab, or slot 0 of the other buffer
    auto advance = [&]() __attribute__((always_inline)) {
      if (++ti == cnt) {
        ti = 0;
        buf ^= 1;
        if (++g == ng) g = 0;
        cnt = g ? gc1 : gc0;
      }
      stage = stage == STA - 1 ? 0 : stage + 1;
    };
    Split8 fa0[MT], fa1[MT], fb[2];
    static_assert(NTC % 2 == 0, "column parity of fb is carried across chunks");
    if constexpr (BRAW) {
      // ggp_kernel's consumer: a column's eight k-rows are read raw (k-slot (lh, j) = k-row 2j + lh of the slot) and split while
      // the previous column's MFMAs run; the chunk barrier sits in front of the last column.
      auto read_col = [&](int u, float (&x)[8]) __attribute__((always_inline)) {
        const float* bs = Bs + buf * SLAB + ti * 1024 + boff[u];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = bs[2 * j * 64];
      };
      float rc[8];
      load_a(0, fa0);
      read_col(0, rc);
      split8(rc, fb[0]);
      read_col(1, rc);
      auto chunk = [&](Split8 (&fa)[MT], Split8 (&fan)[MT]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u + 1 < NTC; ++u) {
          __builtin_amdgcn_sched_barrier(0);
          split8(rc, fb[(u + 1) & 1]);
          if (u + 2 < NTC) read_col(u + 2, rc);
#pragma unroll
          for (int t = 0; t < MT; ++t) acc[t][u] = split_mac(fa[t], fb[u & 1], acc[t][u]);
#pragma unroll
          for (int i = 0; i < 6 * MT; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        advance();
        __syncthreads();   // every consumer has read this chunk's A out of LDS; the producer has the next chunk (and slab) landed
        load_a(stage, fan);
        read_col(0, rc);
        split8(rc, fb[NTC & 1]);
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[t][NTC - 1] = split_mac(fa[t], fb[(NTC - 1) & 1], acc[t][NTC - 1]);
#pragma unroll
        for (int i = 0; i < 6 * MT; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, (44 + 6 * MT - 1) / (6 * MT), 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        read_col(1, rc);
      };
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
    } else {
      auto load_b = [&](int u, Split8& f) __attribute__((always_inline)) {
        const u32x4* bp = Bs4 + buf * (SLAB / 4) + ti * 384 + boff[u];
        f.h = bp[0];
        f.m = bp[128];
        f.l = bp[256];
      };
      load_a(0, fa0);
      load_b(0, fb[0]);
      // one chunk = one tap of the slab: column u's 6*MT MFMAs run with column u+1's three plane reads in their shadow; the chunk
      // barrier sits in front of the LAST column, whose MFMAs cover the next chunk's A and column-0 reads.
      auto chunk = [&](Split8 (&fa)[MT], Split8 (&fan)[MT]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u + 1 < NTC; ++u) {
          __builtin_amdgcn_sched_barrier(0);
          load_b(u + 1, fb[(u + 1) & 1]);
#pragma unroll
          for (int t = 0; t < MT; ++t) acc[t][u] = split_mac(fa[t], fb[u & 1], acc[t][u]);
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x008, 6 * MT - 3, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        advance();
        __syncthreads();   // every consumer has read this chunk's A out of LDS; the producer has the next chunk (and slab) landed
        load_a(stage, fan);
        load_b(0, fb[NTC & 1]);
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[t][NTC - 1] = split_mac(fa[t], fb[(NTC - 1) & 1], acc[t][NTC - 1]);
#pragma unroll
        for (int i = 0; i < 3 * MT + 3; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      };
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
    }
  }

  // ---- epilogue: gg_kernel's, with the unit column mapping (GGParams::patch) -------------------------------------------------
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

// gpw_kernel: the WIDE patch tile — 128 rows x kWideP = 8 units (512 columns) per block, four waves of 128 x 128 (256 accumulator
// registers each, so one block per CU and one wave per SIMD), NO producer wave.  3-tap rows of a stride-1 gather only (the 3 x 3 layers).
//
// Why this shape (profiles/r04_kernel_experiments.md §1-2): a CU moves ~1 KB of LDS-DMA per ~100 cycles however the pieces are
// addressed, so a tile is matrix-pipe-bound only under ~15 pieces per 1 536 MFMA cycles.  ggp_kernel's 128 x 256 needs 28, gpp_kernel's
// 24 (planes) / 20 (raw); here a chunk is 3 072 MFMA cycles per SIMD for 12 filter pieces + a third of a 12-slot raw slab (4 KB per
// slot): 28 per 3 072 = 14.  The source stays raw fp32 — no planes pass — and each wave splits its own 128 columns (32 values per
// lane per chunk, 1.8 VALU per MFMA; the 64 x 128 waves of ggp_kernel pay 3.7).  The filter planes are gpp_kernel's (row tiles of 128).
// Staging is issued by the four waves themselves, with gpp_kernel's producer instructions (one M0 write and immediate-offset pieces per
// group): per chunk every wave moves a quarter (3 KB) of the filter chunk two ahead and one slot of the next slab.  With one wave per
// SIMD nothing hides a wave's bookkeeping, so a chunk is ONE basic block: the iterators step with selects, a slot that is outside
// the image is a load of the zero page and a slot nobody reads a load into a dump region — every wave issues exactly 7 loads per
// chunk and waits with a constant count in front of the chunk barrier.
// Row tiles of 128 are exact for 384 and 256 rows; conv3/4 at 256 images are 254 tiles for 256 CUs.
// Patch mode 3, the default where its launch policy applies (patch_run).  First run on the MI355X in round 5 (parity green, conv4
// 509 -> 452-475 us; profiles/r05_wide_kernels.md); the two variants written for that first A/B — staging loads in back-to-back groups, a
// two-stage filter ring below 128 KB in case M0 were narrower than the LDS — measured the same to 1 % and are gone.  The CPU emulation
// of this source (tests/test_emulated_kernels.py) and the schedule model (tests/test_patch_wide_cpu.py) stay as GPU-less checks.
__global__ __launch_bounds__(256, 1) void gpw_kernel(const GGParams pin, const GGClassTable ct) {
  constexpr int WC = 4, MT = 4, CW = 128, NTC = CW / 32, P = kWideP, NS = kWideNS;
  using fvec = __attribute__((ext_vector_type(NTC))) float;
  constexpr int NC = WC * 64;
  constexpr int ROWS = MT * 32;
  constexpr int A_STAGE = 6 * ROWS * 4;   // floats: 3 planes x 2 k-groups x ROWS x 16 bytes
  constexpr int STA = 3;                  // A ring: filter chunks staged two ahead
  constexpr int SLAB = NS * 1024;         // floats per slab: a slot is 16 k-rows x 64 images of fp32
  static_assert(A_STAGE * 4 == WC * 3 * 1024, "three 1 KB pieces of a filter chunk per wave");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                   // [STA][A_STAGE]
  float* Bs = smem + STA * A_STAGE;   // [2][SLAB], then a 4 KB dump slot

  const GGParams& p = pin;
  GGTile T;
  if (!gg_select_tile(p, ct, T)) return;
  const int L = T.L, tsplit = T.tsplit;
  const int row_tile = L % p.row_tiles, col_tile = L / p.row_tiles;
  const int split = blockIdx.y;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int N = p.N;
  const int dir = p.dir, SH = p.SH, SW = p.SW;

  // ---- the tile's units (wave-uniform), reduced at once to what the loop needs ---------------------------------------------------
  //   S_l     this lane's slot base (its unit: wave wc owns units 2*wc and 2*wc + 1, lanes li < 16 the first)
  //   my_ord  this wave's three slots of a slab, in first-needed order: the (up to eight) slots tap slot 0 reads, then what tap
  //           slot 1 adds, then tap slot 2; position q of that order belongs to wave q % 4, step q / 4
  //   lane s < NS describes slot s: source row / column of (unit j, tap slot i) with S[j] + i == s for tap row 0, image block
  const int G = T.G, GX = T.GX, units = p.IB * G;
  int ys_f = 1 << 30, ys_l = -(1 << 30);
  int S_l = 0, my_ord[3] = {-1, -1, -1};
  int sy_l = 0, sx_l = 0, sib_l = 0;
  bool valid_l = false;
  {
    int S[P], oy[P], ox[P], ib[P];
    bool ok[P];
#pragma unroll
    for (int j = 0; j < P; ++j) {
      const int U = col_tile * P + j;
      ok[j] = U < units;
      ib[j] = ok[j] ? U / G : 0;
      const int m = ok[j] ? U - ib[j] * G : 0;
      oy[j] = m / GX;
      ox[j] = m - oy[j] * GX;
      if (ok[j]) {
        const int ys0 = oy[j] * p.ssy + T.y0;
        ys_f = min(ys_f, ys0);
        ys_l = max(ys_l, ys0);
      }
      // the next pixel of the same image row shares all but one slot with its neighbour; anything else starts a fresh run
      S[j] = j == 0 ? 0 : S[j - 1] + (!ok[j] ? 0 : (ib[j] == ib[j - 1] && oy[j] == oy[j - 1]) ? 1 : 3);
    }
    const int ju = wave * 2 + ((li >> 4) & 1), s = lane & 15;
    unsigned seen = 0;
    int no = 0;
#pragma unroll
    for (int j = 0; j < P; ++j)
      if (j == ju) S_l = S[j];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < P; ++j) {
        const int sl = S[j] + i;
        if (ok[j] && sl < NS && !((seen >> sl) & 1)) {
          seen |= 1u << sl;
#pragma unroll
          for (int q = 0; q < 3; ++q)
            if (no == 4 * q + wave) my_ord[q] = sl;
          ++no;
        }
        if (ok[j] && sl == s) {
          valid_l = true;
          sy_l = oy[j] * p.ssy + T.y0;
          sx_l = ox[j] * p.ssx + T.x0 + dir * p.gb0[0] + i * p.ssx;
          sib_l = ib[j] * 64;
        }
      }
  }
  const bool xin_l = valid_l && (unsigned)sx_l < (unsigned)SW;
  // Integer division has no scalar form: everything derived from U / G, m / GX lives in VGPRs although it is wave-uniform, and drags
  // the loop's bookkeeping onto the vector ALU.  What the loop keeps goes back to SGPRs here.
  auto sgpr = [](int v) __attribute__((always_inline)) { return __builtin_amdgcn_readfirstlane(v); };
#pragma unroll
  for (int q = 0; q < 3; ++q) my_ord[q] = sgpr(my_ord[q]);
  ys_f = sgpr(ys_f);
  ys_l = sgpr(ys_l);

  // ---- reduction range in superchunks (16-channel block cb, tap row a); three chunks (taps) each --------------------------------
  const int TX = T.TX, TYX = T.TYX, TYn = TYX / TX;
  int a_lo = 0, a_hi = TYn - 1;
  const bool skip = tsplit < 0 && p.splits == 1;
  if (skip) {   // tap rows that exist for some pixel of the tile
    if (dir > 0) { a_lo = max(0, -ys_l); a_hi = min(TYn - 1, SH - 1 - ys_f); }
    else { a_lo = max(0, ys_f - (SH - 1)); a_hi = min(TYn - 1, ys_l); }
  }
  const int nrow = max(0, a_hi - a_lo + 1);
  const int nsc_all = (p.KC / BK) * nrow;
  int sc_beg = 0, sc_end = nsc_all;
  if (!skip) {
    const int cps = tsplit >= 0 ? p.tail_cps : p.chunks_per_split;
    sc_beg = min(nsc_all, (tsplit >= 0 ? tsplit : split) * cps);
    sc_end = min(nsc_all, sc_beg + cps);
  }
  const int nchunks = 3 * (sc_end - sc_beg);
  const int cb_beg = nrow > 0 ? sgpr(sc_beg / nrow) : 0, r_beg = nrow > 0 ? sgpr(sc_beg % nrow) : 0;   // first superchunk: channel block, tap row - a_lo

  f32x16 acc[MT][NTC];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int u = 0; u < NTC; ++u)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][u][e] = 0.f;

  if (nchunks > 0) {
    // ================================ staging (every wave its share) ================================
    const size_t ch_bytes = (size_t)SH * SW * N * 4;
    // a slot piece is 4 k-rows (channels) x 64 images: lane = (k-row of the piece, image quad)
    const unsigned lane_off_raw = (unsigned)((size_t)(lane >> 4) * ch_bytes + (size_t)(lane & 15) * 16);
    const char* const rawsrc = reinterpret_cast<const char*>(p.src);
    const char* const zero_page = reinterpret_cast<const char*>(p.zero);
    const unsigned a_lane = (unsigned)lane * 16u;
    const char* const abase0 = reinterpret_cast<const char*>(T.A) + (size_t)row_tile * (6 * ROWS * 16) + 3072u * wave;
    const size_t a_chunk_bytes = (size_t)p.row_tiles * (6 * ROWS * 16);
    const unsigned lds_a = lds_addr(As) + 3072u * wave;
    const unsigned lds_b = lds_addr(Bs);
    const unsigned lds_dump = lds_b + 2u * SLAB * 4u;
    const int gb = p.gb0[0], dstep = dir * p.ssx;   // tap of tap slot i: gb + i*dstep

    // Everything below steps with selects between values that are already computed — no lazily evaluated side, nothing the compiler
    // turns into a branch (check the ISA after an edit: `s_cbranch` between two `s_barrier`s of the loop means a chunk is no longer one block).
    // filter iterator, two chunks ahead of the MFMAs: tap slot i of tap row a of channel block cb is filter chunk cb*TYX + a*TX + gb
    // + i*dstep; a running pointer and the three byte steps (next tap / next tap row / next channel block).  Past the end it stays on
    // the last chunk (those loads go to a stage nobody reads any more).
    // (as differences: a select between two captured variables becomes a load through a selected ADDRESS and sends the whole closure to scratch)
    const ptrdiff_t a_tap = (ptrdiff_t)a_chunk_bytes * dstep, a_row_x = (ptrdiff_t)a_chunk_bytes * (TX - 3 * dstep),
                    a_cbs_x = (ptrdiff_t)a_chunk_bytes * (TYX - (a_hi - a_lo + 1) * TX);
    const char* a_ptr = abase0 + a_chunk_bytes * (size_t)(cb_beg * TYX + (a_lo + r_beg) * TX + gb);   // wave-uniform
    // counters as plain integer arithmetic (0/1 flags, masks): booleans with && / ?: come back from the optimizer as branches
    int A_i = 0, A_r = r_beg, A_left = nchunks;   // tap slot, tap row - a_lo, chunks not yet issued
    unsigned lds_f0 = lds_a, lds_f1 = lds_a + A_STAGE * 4u, lds_f2 = lds_a + 2u * A_STAGE * 4u;   // the ring stage to fill next first
    // (in two parts, so that a chunk can place them in different steps: the address into SGPRs, then the loads and the stepping)
    const char* a_cur = nullptr;
    unsigned a_lds = 0;
    auto issue_a_addr = [&]() __attribute__((always_inline)) {
      a_cur = uniform_ptr(a_ptr);
      a_lds = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_f0);
    };
    auto issue_a_go = [&]() __attribute__((always_inline)) { lds_dma3_rfl(a_lane, a_cur, a_cur, a_cur, a_lds); };   // (prologue)
    auto issue_a_piece = [&](auto J) __attribute__((always_inline)) { lds_dma_piece_rfl<decltype(J)::value>(a_lane, a_cur, a_lds); };
    auto issue_a_step = [&]() __attribute__((always_inline)) {
      const unsigned f = lds_f0;
      lds_f0 = lds_f1;
      lds_f1 = lds_f2;
      lds_f2 = f;
      --A_left;
      const int more = (int)((unsigned)(-A_left) >> 31);   // 1 while chunks are left
      const int i1 = A_i + 1, w1 = (i1 * 11) >> 5;          // w1 = 1 when the tap row is complete (i1 == 3)
      A_i = i1 - 3 * w1;
      const int r1 = A_r + w1, w2 = 1 - (int)((unsigned)(r1 - nrow) >> 31);   // w2 = 1 when the channel block is complete (r1 == nrow)
      A_r = r1 - nrow * w2;
      const ptrdiff_t d = a_tap + (-(ptrdiff_t)w1 & a_row_x) + (-(ptrdiff_t)w2 & a_cbs_x);   // next tap / + next row / + next block
      a_ptr += -(ptrdiff_t)more & d;
    };
    auto issue_a = [&]() __attribute__((always_inline)) {
      issue_a_addr();
      issue_a_go();
      issue_a_step();
    };
    // slab iterator, one superchunk ahead: tap row and the source pointer of its channel block
    int B_r = A_r;   // tap row - a_lo
    const size_t cb_bytes = 16 * ch_bytes;   // one 16-channel block of the source
    const char* slab_src = rawsrc + (size_t)cb_beg * cb_bytes;
    auto slab_next = [&](int step) __attribute__((always_inline)) {   // step: 0 / 1
      const int r1 = B_r + step, w = 1 - (int)((unsigned)(r1 - nrow) >> 31);
      B_r = r1 - nrow * w;
      slab_src += -(ptrdiff_t)w & (ptrdiff_t)cb_bytes;
    };
    constexpr unsigned kNoSlot = 0xFFFFFFFFu;     // no unit of the tile reads this slot: the load goes to the dump region
    constexpr unsigned kZeroSlot = 0xFFFFFFFEu;   // read, but outside the image: loaded from the zero page
    auto slot_desc = [&](int a) __attribute__((always_inline)) {   // -> float index of (pixel, image ib*64) inside a channel plane
      const int ys = sy_l + dir * a;
      const unsigned off = (unsigned)((ys * SW + sx_l) * N + sib_l);
      const bool in = xin_l && (unsigned)ys < (unsigned)SH;
      const unsigned o1 = in ? off : kZeroSlot;
      return valid_l ? o1 : kNoSlot;
    };
    const ptrdiff_t d4 = (ptrdiff_t)(4 * ch_bytes) - 1024;   // four channel planes on, minus the 1 KB the immediate offset adds
    // slot sl (< 0: none) of the slab described by soff from source block `src` into the slab buffer at LDS address ldbuf — in three
    // parts for the same reason: what kind of load it is, its four addresses, the loads
    struct SlotIssue {
      unsigned so, voff, ld, real;   // real: 0 / 1
      const char *p0, *p1, *p2, *p3;
    } si;
    // 0 / 1 flags and masks instead of booleans (see the counters above): none = nobody reads the slot, real = it is inside the image
    auto slot_kind = [&](int sl, unsigned ldbuf, unsigned soff, int enable) __attribute__((always_inline)) {
      const unsigned neg = (unsigned)sl >> 31;
      const int slc = sl & ~(-(int)neg);                                     // max(sl, 0)
      si.so = (unsigned)__builtin_amdgcn_readlane((int)soff, slc);
      const unsigned is_no = 1u - min(si.so + 1u, 1u), is_zero = 1u - min(si.so + 2u, 1u);   // so == kNoSlot, so == kZeroSlot
      const unsigned none = (1u - (unsigned)enable) | neg | is_no;
      si.real = (1u - none) & (1u - is_zero);
      const unsigned ldr = ldbuf + (unsigned)slc * 4096u;
      si.ld = ldr ^ ((ldr ^ lds_dump) & (0u - none));                        // none ? dump : slot
      si.voff = lane_off_raw & (0u - si.real);
    };
    auto slot_addr = [&](const char* src) __attribute__((always_inline)) {
      const ptrdiff_t m = -(ptrdiff_t)si.real;
      const char* const rbase = src + (size_t)si.so * 4;   // wave-uniform: k-row 0 of the slot
      const char* const base = zero_page + ((rbase - zero_page) & m);
      const ptrdiff_t st = (ptrdiff_t)-1024 + ((d4 + 1024) & m);
      si.p0 = uniform_ptr(base);
      si.p1 = uniform_ptr(base + st);
      si.p2 = uniform_ptr(base + 2 * st);
      si.p3 = uniform_ptr(base + 3 * st);
    };
    auto slot_go = [&]() __attribute__((always_inline)) {   // (prologue)
      lds_dma4_rfl(si.voff, si.p0, si.p1, si.p2, si.p3, (unsigned)__builtin_amdgcn_readfirstlane((int)si.ld));
    };
    auto slot_piece = [&](auto J) __attribute__((always_inline)) {
      constexpr int j = decltype(J)::value;
      lds_dma_piece_rfl<j>(si.voff, j == 0 ? si.p0 : j == 1 ? si.p1 : j == 2 ? si.p2 : si.p3, (unsigned)__builtin_amdgcn_readfirstlane((int)si.ld));
    };
    auto issue_slot = [&](int sl, unsigned ldbuf, const char* src, unsigned soff, int enable) __attribute__((always_inline)) {
      slot_kind(sl, ldbuf, soff, enable);
      slot_addr(src);
      slot_go();
    };

    // ================================ consumer state ================================
    const int boff = S_l * 1024 + lh * 64 + NTC * (li & 15);   // floats inside a slab: slot base + k-row lh + first image
    auto load_a = [&](int st, Split8 (&fa)[MT]) __attribute__((always_inline)) {
      const u32x4* ap = reinterpret_cast<const u32x4*>(As + st * A_STAGE) + lh * ROWS + li;
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        fa[t].h = ap[t * 32];
        fa[t].m = ap[2 * ROWS + t * 32];
        fa[t].l = ap[4 * ROWS + t * 32];
      }
    };
    int stage = 0, ti = 0, sc = sc_beg;
    unsigned bufsel = 0;   // 0 / 1: the slab buffer the MFMAs read
    // k-slot (lh, j) of a column = k-row 2j + lh of the slot; one 16-byte read per k-row brings this lane's NTC images
    f32x4 bv[8];
    auto read_b = [&]() __attribute__((always_inline)) {
      const float* bs = Bs + bufsel * SLAB + ti * 1024 + boff;
#pragma unroll
      for (int j = 0; j < 8; ++j) bv[j] = ld4(bs + 2 * j * 64);
    };
    // this chunk's share of the staging: a quarter of the filter chunk two ahead and one slot of the next slab into the idle buffer
    // (tap slot 0's slots — positions 0..7 — in the first two chunks, the rest in the last: they are first read by the SECOND chunk
    // of the next superchunk).  Seven loads, always.  o0 is this chunk's slot of the wave's three; they rotate with the chunks.
    int o0 = my_ord[0], o1 = my_ord[1], o2 = my_ord[2];
    auto next_slot_kind = [&]() __attribute__((always_inline)) {
      slot_kind(o0, lds_b + (bufsel ^ 1u) * (SLAB * 4u), slot_desc(a_lo + B_r), (int)((unsigned)(sc + 1 - sc_end) >> 31));   // sc + 1 < sc_end
      const int o = o0;
      o0 = o1;
      o1 = o2;
      o2 = o;
    };
    auto advance = [&]() __attribute__((always_inline)) {
      const int t1 = ti + 1, w = (t1 * 11) >> 5;   // w = 1 when the superchunk is complete
      ti = t1 - 3 * w;
      bufsel ^= (unsigned)w;
      sc += w;
      slab_next(w);
      const int s1 = stage + 1, ws = 1 - (int)((unsigned)(s1 - STA) >> 31);   // ws = 1: wrap
      stage = s1 - STA * ws;
    };

    // prologue: slab 0, filter chunks 0 and 1
    {
      const unsigned so = slot_desc(a_lo + B_r);
#pragma unroll
      for (int q = 0; q < 3; ++q) issue_slot(my_ord[q], lds_b, slab_src, so, 1);
      slab_next(1);
    }
    issue_a();
    issue_a();   // (two ahead)
    __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0) lgkmcnt(0)
    __syncthreads();

    Split8 fa0[MT], fa1[MT], fb[2];
    static_assert(NTC % 2 == 0, "column parity of fb is carried across chunks");
    // A chunk is written out as 4 columns x 6 STEPS of MT = 4 MFMAs, each step fenced (sched_barrier) and carrying its share of the
    // other work by hand: the scheduler's group patterns hold inside a step of this size, not over a whole column (it left 20-50
    // VALU in one lump at region borders: ~12 % of a chunk with the matrix pipe idle).  A step = one of the six products of split_mac,
    // in its order, over the four row tiles: four independent accumulators between two MFMAs on the same one (with ONE wave per SIMD no
    // other wave fills that wait; per accumulator the order of the six is split_mac's — same bits).
    auto mac_step = [&](auto K, const Split8 (&fa)[MT], const Split8& b, int u) __attribute__((always_inline)) {
      constexpr int k = decltype(K)::value;
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const u32x4& av = k == 0 || k == 4 ? fa[t].m : k == 2 ? fa[t].l : fa[t].h;   // (m,m) (h,l) (l,h) (h,m) (m,h) (h,h)
        const u32x4& bw = k == 0 || k == 3 ? b.m : k == 1 ? b.l : b.h;
        acc[t][u] = mma_bf16(av, bw, acc[t][u]);
      }
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x004, 3, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    // pair q of split8 for column u of the slab values in bv
    auto split_pair = [&](int u, int q, Split8& f) __attribute__((always_inline)) {
      const float x0 = bv[2 * q][u], x1 = bv[2 * q + 1][u];
      const unsigned H = pk_bf16(x0, x1);
      const float r0 = x0 - __uint_as_float(H << 16), r1 = x1 - __uint_as_float(H & 0xffff0000u);
      const unsigned M = pk_bf16(r0, r1);
      const float s0 = r0 - __uint_as_float(M << 16), s1 = r1 - __uint_as_float(M & 0xffff0000u);
      f.h[q] = H;
      f.m[q] = M;
      f.l[q] = pk_bf16(s0, s1);
    };
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>;
    using K3 = std::integral_constant<int, 3>;
    using K4 = std::integral_constant<int, 4>;
    using K5 = std::integral_constant<int, 5>;
    // one chunk = one tap of the slab.  Columns 0..2: the split of the next column in steps 0..3 (a pair each), the staging issue
    // under column 0, the counters of the next chunk under column 2; the chunk barrier; column 3 with the next chunk's LDS reads in
    // steps 0-1 and the split of its column 0 in steps 2..5.
    auto chunk = [&](Split8 (&fa)[MT], Split8 (&fan)[MT]) __attribute__((always_inline)) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u + 1 < NTC; ++u) {
        Split8& fn = fb[(u + 1) & 1];
        const Split8& fc = fb[u & 1];
        // The seven staging loads go out ONE per step (a piece costs the issuing wave ~60 cycles; back to back they stack): the filter
        // chunk's three under column 0, the slot's four under column 1.  Steps 4 and 5 carry no split: the staging bookkeeping (column
        // 0: which load the slot is, its addresses; column 1: the filter iterator) and the counters of the NEXT chunk (column 2).
        split_pair(u + 1, 0, fn);
        if (u == 0) issue_a_addr();
        if (u == 1) slot_piece(K0{});
        mac_step(K0{}, fa, fc, u);
        split_pair(u + 1, 1, fn);
        if (u == 0) issue_a_piece(K0{});
        if (u == 1) slot_piece(K1{});
        mac_step(K1{}, fa, fc, u);
        split_pair(u + 1, 2, fn);
        if (u == 0) issue_a_piece(K1{});
        if (u == 1) slot_piece(K2{});
        mac_step(K2{}, fa, fc, u);
        split_pair(u + 1, 3, fn);
        if (u == 0) issue_a_piece(K2{});
        if (u == 1) slot_piece(K3{});
        mac_step(K3{}, fa, fc, u);
        if (u == 0) next_slot_kind();
        if (u == 1) issue_a_step();
        if (u == NTC - 2) advance();
        mac_step(K4{}, fa, fc, u);
        if (u == 0) slot_addr(slab_src);
        mac_step(K5{}, fa, fc, u);
      }
      // the last column's split is complete HERE (the compiler otherwise sinks it towards its use, out of the MFMA shadow)
      CHIP_PIN_SPLIT8(fb[(NTC - 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      // everything this wave issued before this chunk's batch has landed: vmcnt(7) lgkmcnt(0)
      __builtin_amdgcn_s_waitcnt(0x0077);
      __syncthreads();                      // ... and every other wave's; every wave has read this chunk's A and slab slots out of LDS
      {
        const Split8& fc = fb[(NTC - 1) & 1];
        Split8& fn = fb[NTC & 1];
        read_b();
        mac_step(K0{}, fa, fc, NTC - 1);
        load_a(stage, fan);
        mac_step(K1{}, fa, fc, NTC - 1);
        split_pair(0, 0, fn);
        mac_step(K2{}, fa, fc, NTC - 1);
        split_pair(0, 1, fn);
        mac_step(K3{}, fa, fc, NTC - 1);
        split_pair(0, 2, fn);
        mac_step(K4{}, fa, fc, NTC - 1);
        split_pair(0, 3, fn);
        mac_step(K5{}, fa, fc, NTC - 1);
        CHIP_PIN_SPLIT8(fn);   // (as above: complete here)
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    load_a(0, fa0);
    read_b();
#pragma unroll
    for (int q = 0; q < 4; ++q) split_pair(0, q, fb[0]);
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
    __builtin_amdgcn_s_waitcnt(0x0070);   // the last two batches (dump / free-stage loads) before the LDS is released
  }

  // ---- epilogue: gg_kernel's, with the unit column mapping (GGParams::patch); every accumulator read straight out of its AGPR
  acc_settle();
  if (tsplit >= 0) {
    float* pp = p.tail_partial + ((size_t)(L - p.tail_first) * p.tail_splits + tsplit) * (size_t)(ROWS * WC * CW);
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        fvec v;
#pragma unroll
        for (int u = 0; u < NTC; ++u) v[u] = acc_elem<true>(acc[t][u][reg]);
        *reinterpret_cast<fvec*>(pp + ((size_t)(t * 16 + reg) * NC + tid) * NTC) = v;
      }
    return;
  }
  gg_epilogue<1, WC, MT, CW, true, true>(p, acc, row_tile, col_tile, split, T.ncols, T.GX, T.G, T.dy0, T.dx0);
}

// gg_tail_fix_kernel for gpw_kernel's / gpv_kernel's tile: the same four blocks per tail tile, each summing the tail_splits partial tiles of a
// quarter of the accumulator registers in fixed order and running the normal epilogue on them — with the quarter a COMPILE-TIME constant per
// branch: with 256 accumulator registers per lane the runtime range of gg_tail_fix_kernel indexes the array dynamically (2 KB of scratch).
template <int MT, int PART>
__device__ __forceinline__ void gpw_tail_fix_part(const GGParams& p, int tile) {
  constexpr int WC = 4, CW = 128, NT = WC * 64, NTC = CW / 32, ROWS = MT * 32;
  constexpr int reg_lo = PART * (16 / kTailFixParts), reg_hi = reg_lo + 16 / kTailFixParts;
  using fvec = __attribute__((ext_vector_type(NTC))) float;
  const int L = p.tail_first + tile;
  const int tid = threadIdx.x;
  f32x16 acc[MT][NTC];
  const float* pp = p.tail_partial + (size_t)tile * p.tail_splits * (size_t)(ROWS * WC * CW);
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int reg = reg_lo; reg < reg_hi; ++reg) {
      fvec v = *reinterpret_cast<const fvec*>(pp + ((size_t)(t * 16 + reg) * NT + tid) * NTC);
      for (int sp = 1; sp < p.tail_splits; ++sp)
        v += *reinterpret_cast<const fvec*>(pp + (size_t)sp * (ROWS * WC * CW) + ((size_t)(t * 16 + reg) * NT + tid) * NTC);
#pragma unroll
      for (int u = 0; u < NTC; ++u) acc[t][u][reg] = v[u];
    }
  gg_epilogue<1, WC, MT, CW, true>(p, acc, L % p.row_tiles, L / p.row_tiles, 0, p.ncols, p.GX, p.G, p.dy0, p.dx0, reg_lo, reg_hi);
}
template <int MT>
__global__ __launch_bounds__(256) void gpw_tail_fix_kernel(const GGParams p) {
  const int tile = blockIdx.x / kTailFixParts;
  switch (blockIdx.x % kTailFixParts) {
    case 0: gpw_tail_fix_part<MT, 0>(p, tile); break;
    case 1: gpw_tail_fix_part<MT, 1>(p, tile); break;
    case 2: gpw_tail_fix_part<MT, 2>(p, tile); break;
    default: gpw_tail_fix_part<MT, 3>(p, tile); break;
  }
}

// gpv_kernel: gpw_kernel's tile and staging scheme for tap rows cut into GROUPS of three or two taps — what AlexNet's second layer needs:
//   * its 5 x 5 stride-2 forward pass (the reference special-cases the same class, cudamat_conv_filteracts.cu:985-1140): a tap row is the
//     groups {0,2,4} and {1,3}; inside a group neighbouring output pixels' taps coincide exactly as in a stride-1 row (slot i of pixel
//     j+1 = slot i+1 of pixel j), so a group is a slab of 8 + cnt - 1 slots and a SUPERCHUNK of cnt chunks;
//   * the four stride classes of its input gradient (cudamat_conv_imgacts.cu:355-392 is the reference's gather form), each a stride-1
//     gather with 3- or 2-tap rows, all classes in ONE launch (GGClassTable, as ggp_kernel runs them), on an MT = 3 build: 96 rows.
// Everything gpw_kernel's header says holds; what is new:
//   * the superchunk walk is (16-channel block, tap row, group), the chunk count per superchunk 3 or 2 (runtime, wave-uniform);
//   * a 3-chunk superchunk gives every wave one slot of the next slab per chunk (7 loads per chunk, as gpw_kernel); a 2-chunk superchunk
//     must bring the next slab's first-needed slots (positions 0..7 of the order: what tap slot 0 reads) under way in its FIRST chunk,
//     so that chunk carries TWO slots per wave (11 loads) and the second one (7).  The chunk body is a template over that count — the
//     wait in front of the chunk barrier needs it as an immediate — and the loop picks the body with one scalar branch per chunk;
//   * the slab buffers still alternate per superchunk; the first-needed order is computed over three tap slots whatever the group:
//     a slot only tap slot 2 reads is "nobody's" in a 2-tap group (a dump load);
//   * MT = 3: a filter chunk is 9 216 B, a wave's share two pieces and a quarter (lds_dma_quarter_rfl: 16 lanes).
// Restated with adversarial load landing in tests/test_patch_var_cpu.py, run on the CPU by tests/test_emulated_kernels.py.
template <int MT>
__global__ __launch_bounds__(256, 1) void gpv_kernel(const GGParams pin, const GGClassTable ct) {
  constexpr int WC = 4, CW = 128, NTC = CW / 32, P = kWideP, NS = kWideNS;
  using fvec = __attribute__((ext_vector_type(NTC))) float;
  constexpr int NC = WC * 64;
  constexpr int ROWS = MT * 32;
  constexpr int A_STAGE = 6 * ROWS * 4;   // floats: 3 planes x 2 k-groups x ROWS x 16 bytes
  constexpr int STA = 3;                  // A ring: filter chunks staged two ahead
  constexpr int SLAB = NS * 1024;         // floats per slab: a slot is 16 k-rows x 64 images of fp32
  constexpr unsigned A_WAVE = A_STAGE * 4 / WC;   // bytes of a filter chunk per wave: 3 072 (MT = 4), 2 304 (MT = 3)
  static_assert(MT == 4 || MT == 3, "a wave's share of a filter chunk is three pieces, or two and a quarter");
  extern __shared__ __attribute__((aligned(16))) float smem[];
#if CONVNET_GPV_LDS_MAP == 1
  // filter stage 0 | slab 0 | filter stage 1 | slab 1 | filter stage 2 | dump slot
  constexpr int A_STRIDE = A_STAGE + SLAB, B_STRIDE = A_STAGE + SLAB;
  float* As = smem;
  float* Bs = smem + A_STAGE;
  float* const dump_slot = smem + STA * A_STAGE + 2 * SLAB;
#else
  constexpr int A_STRIDE = A_STAGE, B_STRIDE = SLAB;
  float* As = smem;                   // [STA][A_STAGE]
  float* Bs = smem + STA * A_STAGE;   // [2][SLAB], then a 4 KB dump slot
  float* const dump_slot = smem + STA * A_STAGE + 2 * SLAB;
#endif

  const GGParams& p = pin;
  GGTile T;
  if (!gg_select_tile(p, ct, T)) return;
  const int L = T.L, tsplit = T.tsplit;
  const int row_tile = L % p.row_tiles, col_tile = L / p.row_tiles;
  const int split = blockIdx.y;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int N = p.N;
  const int dir = p.dir, SH = p.SH, SW = p.SW, ssx = p.ssx;
  auto sgpr = [](int v) __attribute__((always_inline)) { return __builtin_amdgcn_readfirstlane(v); };

  // ---- tap groups of a tap row (patch_shape_ok's rule, from the tile's own TX: the stride classes differ in it) --------------------
  //   group g: taps gb0[g] + i*dstep, i < cnt[g]; ng = ssx groups.  One group: both entries equal.
  const int TX = sgpr(T.TX), TYX = sgpr(T.TYX), TYn = sgpr(TYX / TX);
  const int ng = ssx;
  const int cnt0 = sgpr(ssx == 1 ? TX : (TX + 1) >> 1), cnt1 = sgpr(ssx == 1 ? TX : TX >> 1);
  const int gb00 = sgpr(dir > 0 ? 0 : (cnt0 - 1) * ssx), gb01 = sgpr(ssx == 1 ? gb00 : dir > 0 ? 1 : 1 + (cnt1 - 1) * ssx);
  const int dstep = dir * ssx;

  // ---- the tile's units (wave-uniform), reduced at once to what the loop needs: as gpw_kernel, plus per lane s < NS the smallest
  //      tap slot that reads slot s (imin: a slot only tap slot 2 reads does not exist for a 2-tap group)
  const int G = T.G, GX = T.GX, units = p.IB * G;
  int ys_f = 1 << 30, ys_l = -(1 << 30);
  int S_l = 0, my_ord[3] = {-1, -1, -1};
  int sy_l = 0, sx_l = 0, sib_l = 0, imin_l = 3;
  {
    int S[P], oy[P], ox[P], ib[P];
    bool ok[P];
#pragma unroll
    for (int j = 0; j < P; ++j) {
      const int U = col_tile * P + j;
      ok[j] = U < units;
      ib[j] = ok[j] ? U / G : 0;
      const int m = ok[j] ? U - ib[j] * G : 0;
      oy[j] = m / GX;
      ox[j] = m - oy[j] * GX;
      if (ok[j]) {
        const int ys0 = oy[j] * p.ssy + T.y0;
        ys_f = min(ys_f, ys0);
        ys_l = max(ys_l, ys0);
      }
      S[j] = j == 0 ? 0 : S[j - 1] + (!ok[j] ? 0 : (ib[j] == ib[j - 1] && oy[j] == oy[j - 1]) ? 1 : 3);
    }
    const int ju = wave * 2 + ((li >> 4) & 1), s = lane & 15;
    unsigned seen = 0;
    int no = 0;
#pragma unroll
    for (int j = 0; j < P; ++j)
      if (j == ju) S_l = S[j];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < P; ++j) {
        const int sl = S[j] + i;
        if (ok[j] && sl < NS && !((seen >> sl) & 1)) {
          seen |= 1u << sl;
#pragma unroll
          for (int q = 0; q < 3; ++q)
            if (no == 4 * q + wave) my_ord[q] = sl;
          ++no;
        }
        if (ok[j] && sl == s) {
          imin_l = min(imin_l, i);
          sy_l = oy[j] * p.ssy + T.y0;
          sx_l = ox[j] * ssx + T.x0 + dir * gb00 + i * ssx;   // group 0's source column; group 1's is dir*(gb01 - gb00) further
          sib_l = ib[j] * 64;
        }
      }
  }
  // per group: does a unit read this lane's slot, and is its source column inside the image
  const int sx1_l = sx_l + dir * (gb01 - gb00);
  const bool valid0_l = imin_l < cnt0, valid1_l = imin_l < cnt1;
  const bool xin0_l = valid0_l && (unsigned)sx_l < (unsigned)SW, xin1_l = valid1_l && (unsigned)sx1_l < (unsigned)SW;
#pragma unroll
  for (int q = 0; q < 3; ++q) my_ord[q] = sgpr(my_ord[q]);
  ys_f = sgpr(ys_f);
  ys_l = sgpr(ys_l);

  // ---- reduction range in superchunks (16-channel block cb, tap row a, group g); cnt[g] chunks (taps) each --------------------------
  int a_lo = 0, a_hi = TYn - 1;
  const bool skip = tsplit < 0 && p.splits == 1;
  if (skip) {   // tap rows that exist for some pixel of the tile
    if (dir > 0) { a_lo = max(0, -ys_l); a_hi = min(TYn - 1, SH - 1 - ys_f); }
    else { a_lo = max(0, ys_f - (SH - 1)); a_hi = min(TYn - 1, ys_l); }
  }
  a_lo = sgpr(a_lo);
  a_hi = sgpr(a_hi);
  const int nrow = max(0, a_hi - a_lo + 1);
  const int nsc_all = (p.KC / BK) * nrow * ng;
  int sc_beg = 0, sc_end = nsc_all;
  if (!skip) {
    const int cps = tsplit >= 0 ? p.tail_cps : p.chunks_per_split;
    sc_beg = min(nsc_all, (tsplit >= 0 ? tsplit : split) * cps);
    sc_end = min(nsc_all, sc_beg + cps);
  }
  sc_beg = sgpr(sc_beg);   // (what comes out of an integer division lives in VGPRs although it is wave-uniform, and drags its users along)
  sc_end = sgpr(sc_end);
  const int n_g1 = ng == 2 ? (sc_end >> 1) - (sc_beg >> 1) : 0;   // superchunks of group 1 (odd indices) in the range
  const int nchunks = (sc_end - sc_beg - n_g1) * cnt0 + n_g1 * cnt1;
  // first superchunk: group, tap row - a_lo, channel block
  const int g_beg = ng == 2 ? (sc_beg & 1) : 0;
  const int rr = ng == 2 ? sc_beg >> 1 : sc_beg;
  const int cb_beg = nrow > 0 ? sgpr(rr / nrow) : 0, r_beg = nrow > 0 ? sgpr(rr % nrow) : 0;

  f32x16 acc[MT][NTC];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int u = 0; u < NTC; ++u)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][u][e] = 0.f;

  if (nchunks > 0) {
    // ================================ staging (every wave its share) ================================
    const size_t ch_bytes = (size_t)SH * SW * N * 4;
    const unsigned lane_off_raw = (unsigned)((size_t)(lane >> 4) * ch_bytes + (size_t)(lane & 15) * 16);
    const char* const rawsrc = reinterpret_cast<const char*>(p.src);
    const char* const zero_page = reinterpret_cast<const char*>(p.zero);
    const unsigned a_lane = (unsigned)lane * 16u;
    const char* const abase0 = reinterpret_cast<const char*>(T.A) + (size_t)row_tile * (6 * ROWS * 16) + A_WAVE * wave;
    const size_t a_chunk_bytes = (size_t)p.row_tiles * (6 * ROWS * 16);
    const unsigned lds_a = lds_addr(As) + A_WAVE * wave;
    const unsigned lds_b = lds_addr(Bs);
    const unsigned lds_dump = lds_addr(dump_slot);

#ifdef CONVNET_DIAG
    // timing diagnostics (results wrong), CONVNET_GPP_DIAG: 1 = no slot loads after the prologue, 2 = no filter loads after it, 4 = no dump loads
    const int dg = p.prio, dg_slot = 1 - (dg & 1), dg_filt = 1 - ((dg >> 1) & 1), dg_nodump = (dg >> 2) & 1;
    int dg_ring_dump = 0;   // (32: set behind the prologue, which then fills all three ring stages)
#endif
    // Everything below steps with selects between values that are already computed (gpw_kernel's rules: no lazily evaluated side, 0/1
    // flags and masks instead of booleans, differences instead of selected addresses).
    // filter iterator, two chunks ahead of the MFMAs: tap slot i of group g of tap row a of channel block cb is filter chunk
    // cb*TYX + a*TX + gb0[g] + i*dstep; a running pointer and its byte steps: next tap; what a finished group adds on top (to the
    // next group of the row, or from the last group to the first of the next row); what a finished channel block adds on top.
    const ptrdiff_t a_tap = (ptrdiff_t)a_chunk_bytes * dstep;
    const ptrdiff_t a_x0 = (ptrdiff_t)a_chunk_bytes * ((ng == 2 ? gb01 : TX + gb00) - (gb00 + (cnt0 - 1) * dstep) - dstep);
    const ptrdiff_t a_x1 = ng == 2 ? (ptrdiff_t)a_chunk_bytes * (TX + gb00 - (gb01 + (cnt1 - 1) * dstep) - dstep) : a_x0;
    const ptrdiff_t a_cbs_x = (ptrdiff_t)a_chunk_bytes * (TYX - (a_hi - a_lo + 1) * TX);
    const char* a_ptr = abase0 + a_chunk_bytes * (size_t)(cb_beg * TYX + (a_lo + r_beg) * TX + (g_beg ? gb01 : gb00));   // wave-uniform
#ifdef CONVNET_DIAG
    if (dg & 8) a_ptr += (size_t)((blockIdx.x >> 3) & 7) * 4096;     // every block of an XCD group reads its filter chunks 4 KB further on (wrong data, same amount)
    if (dg & 16) a_ptr += (size_t)((blockIdx.x >> 3) & 7) * 256;     // ... 256 B further on
#endif
    int A_i = 0, A_g = g_beg, A_cnt = g_beg ? cnt1 : cnt0, A_r = r_beg, A_left = nchunks;
    unsigned lds_f0 = lds_a, lds_f1 = lds_a + A_STRIDE * 4u, lds_f2 = lds_a + 2u * A_STRIDE * 4u;   // the ring stage to fill next first
    const char* a_cur = nullptr;
    unsigned a_lds = 0;
    auto issue_a_addr = [&]() __attribute__((always_inline)) {
      a_cur = uniform_ptr(a_ptr);
      a_lds = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_f0);
#ifdef CONVNET_DIAG
      // 32: the filter pieces land in the dump slot; 64: they read the zero page (every lane the same 16 bytes); 128: always the range's first chunk
      {
        const unsigned m32 = 0u - (unsigned)dg_ring_dump;
        a_lds = (unsigned)__builtin_amdgcn_readfirstlane((int)(a_lds ^ ((a_lds ^ lds_dump) & m32)));
        const ptrdiff_t m64 = -(ptrdiff_t)((dg >> 6) & 1), m128 = -(ptrdiff_t)((dg >> 7) & 1);
        const char* q = a_cur + ((abase0 - a_cur) & m128);
        q = q + ((zero_page - q) & m64);
        a_cur = uniform_ptr(q);
      }
#endif
    };
    auto issue_a_piece = [&](auto J) __attribute__((always_inline)) {
      constexpr int j = decltype(J)::value;
#ifdef CONVNET_DIAG
      if constexpr (MT == 4) {
        const unsigned m32 = 0u - (unsigned)dg_ring_dump, m64 = 0u - ((unsigned)(dg >> 6) & 1u);
        lds_dma_piece_if_rfl<j>(sgpr(dg_filt), a_lane & ~m64, uniform_ptr(a_cur - ((ptrdiff_t)(1024 * j) & -(ptrdiff_t)((dg >> 6) & 1))),
                                (unsigned)__builtin_amdgcn_readfirstlane((int)(a_lds - ((1024u * j) & m32))));
        return;
      }
#endif
      if constexpr (MT == 3 && j == 2) lds_dma_quarter_rfl<j>(a_lane, a_cur, a_lds);
      else lds_dma_piece_rfl<j>(a_lane, a_cur, a_lds);
    };
    auto issue_a_step = [&]() __attribute__((always_inline)) {
      const unsigned f = lds_f0;
      lds_f0 = lds_f1;
      lds_f1 = lds_f2;
      lds_f2 = f;
      --A_left;
      const int more = (int)((unsigned)(-A_left) >> 31);            // 1 while chunks are left
      const int i1 = A_i + 1, w1 = 1 - (int)((unsigned)(i1 - A_cnt) >> 31);   // w1 = 1 when the group is complete
      A_i = i1 & (w1 - 1);
      const ptrdiff_t xg = a_x0 + (-(ptrdiff_t)A_g & (a_x1 - a_x0));           // what THIS group's end adds
      const int g1 = A_g + w1, wg = 1 - (int)((unsigned)(g1 - ng) >> 31);      // wg = 1 when the tap row is complete
      A_g = g1 - (ng & -wg);
      A_cnt = cnt0 + ((cnt1 - cnt0) & -A_g);
      const int r1 = A_r + wg, w2 = 1 - (int)((unsigned)(r1 - nrow) >> 31);    // w2 = 1 when the channel block is complete
      A_r = r1 - (nrow & -w2);
      const ptrdiff_t d = a_tap + (-(ptrdiff_t)w1 & xg) + (-(ptrdiff_t)w2 & a_cbs_x);
      a_ptr += -(ptrdiff_t)more & d;
    };
    auto issue_a = [&]() __attribute__((always_inline)) {   // (prologue)
      issue_a_addr();
      issue_a_piece(std::integral_constant<int, 0>{});
      issue_a_piece(std::integral_constant<int, 1>{});
      issue_a_piece(std::integral_constant<int, 2>{});
      issue_a_step();
    };
    // slab iterator, one superchunk ahead: group, tap row and the source pointer of its channel block
    int B_g = g_beg, B_r = r_beg;
    const size_t cb_bytes = 16 * ch_bytes;   // one 16-channel block of the source
    const char* slab_src = rawsrc + (size_t)cb_beg * cb_bytes;
    auto slab_next = [&](int step) __attribute__((always_inline)) {   // step: 0 / 1
      const int g1 = B_g + step, wg = 1 - (int)((unsigned)(g1 - ng) >> 31);
      B_g = g1 - (ng & -wg);
      const int r1 = B_r + wg, w = 1 - (int)((unsigned)(r1 - nrow) >> 31);
      B_r = r1 - (nrow & -w);
      slab_src += -(ptrdiff_t)w & (ptrdiff_t)cb_bytes;
    };
    constexpr unsigned kNoSlot = 0xFFFFFFFFu;     // no unit of the tile reads this slot: the load goes to the dump region
    constexpr unsigned kZeroSlot = 0xFFFFFFFEu;   // read, but outside the image: loaded from the zero page
    auto slot_desc = [&](int a, int g) __attribute__((always_inline)) {   // -> float index of (pixel, image ib*64) inside a channel plane
      const int ys = sy_l + dir * a;
      const int sx = g ? sx1_l : sx_l;
      const bool xin = g ? xin1_l : xin0_l, valid = g ? valid1_l : valid0_l;
      const unsigned off = (unsigned)((ys * SW + sx) * N + sib_l);
      const bool in = xin && (unsigned)ys < (unsigned)SH;
      const unsigned o1 = in ? off : kZeroSlot;
      return valid ? o1 : kNoSlot;
    };
    const ptrdiff_t d4 = (ptrdiff_t)(4 * ch_bytes) - 1024;   // four channel planes on, minus the 1 KB the immediate offset adds
    struct SlotIssue {
      unsigned so, voff, ld, real, none;   // real, none: 0 / 1
      const char *p0, *p1, *p2, *p3;
    } si;
    auto slot_kind = [&](int sl, unsigned ldbuf, unsigned soff, int enable) __attribute__((always_inline)) {
      const unsigned neg = (unsigned)sl >> 31;
      const int slc = sl & ~(-(int)neg);                                     // max(sl, 0)
      si.so = (unsigned)__builtin_amdgcn_readlane((int)soff, slc);
      const unsigned is_no = 1u - min(si.so + 1u, 1u), is_zero = 1u - min(si.so + 2u, 1u);   // so == kNoSlot, so == kZeroSlot
      const unsigned none = (1u - (unsigned)enable) | neg | is_no;
      si.real = (1u - none) & (1u - is_zero);
      si.none = none;
      const unsigned ldr = ldbuf + (unsigned)slc * 4096u;
      si.ld = ldr ^ ((ldr ^ lds_dump) & (0u - none));                        // none ? dump : slot
      si.voff = lane_off_raw & (0u - si.real);
    };
    auto slot_addr = [&](const char* src) __attribute__((always_inline)) {
      const ptrdiff_t m = -(ptrdiff_t)si.real;
      const char* const rbase = src + (size_t)si.so * 4;   // wave-uniform: k-row 0 of the slot
      const char* const base = zero_page + ((rbase - zero_page) & m);
      const ptrdiff_t st = (ptrdiff_t)-1024 + ((d4 + 1024) & m);
      si.p0 = uniform_ptr(base);
      si.p1 = uniform_ptr(base + st);
      si.p2 = uniform_ptr(base + 2 * st);
      si.p3 = uniform_ptr(base + 3 * st);
    };
    auto slot_go = [&]() __attribute__((always_inline)) {   // (prologue)
      lds_dma4_rfl(si.voff, si.p0, si.p1, si.p2, si.p3, (unsigned)__builtin_amdgcn_readfirstlane((int)si.ld));
    };
    auto slot_piece = [&](auto J) __attribute__((always_inline)) {
      constexpr int j = decltype(J)::value;
#ifdef CONVNET_DIAG
      lds_dma_piece_if_rfl<j>(sgpr(dg_slot & (1 - (dg_nodump & (int)si.none))), si.voff, j == 0 ? si.p0 : j == 1 ? si.p1 : j == 2 ? si.p2 : si.p3, (unsigned)__builtin_amdgcn_readfirstlane((int)si.ld));
      return;
#endif
      lds_dma_piece_rfl<j>(si.voff, j == 0 ? si.p0 : j == 1 ? si.p1 : j == 2 ? si.p2 : si.p3, (unsigned)__builtin_amdgcn_readfirstlane((int)si.ld));
    };
    auto slot_piece_if = [&](int on, auto J) __attribute__((always_inline)) {
      constexpr int j = decltype(J)::value;
#ifdef CONVNET_DIAG
      on = sgpr(on & dg_slot & (1 - (dg_nodump & (int)si.none)));
#endif
      lds_dma_piece_if_rfl<j>(on, si.voff, j == 0 ? si.p0 : j == 1 ? si.p1 : j == 2 ? si.p2 : si.p3, (unsigned)__builtin_amdgcn_readfirstlane((int)si.ld));
    };

    // ================================ consumer state ================================
    const int boff = S_l * 1024 + lh * 64 + NTC * (li & 15);   // floats inside a slab: slot base + k-row lh + first image
    auto load_a = [&](int st, Split8 (&fa)[MT]) __attribute__((always_inline)) {
      const u32x4* ap = reinterpret_cast<const u32x4*>(As + st * A_STRIDE) + lh * ROWS + li;
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        fa[t].h = ap[t * 32];
        fa[t].m = ap[2 * ROWS + t * 32];
        fa[t].l = ap[4 * ROWS + t * 32];
      }
    };
    int stage = 0, ti = 0, sc = sc_beg, g_cur = g_beg, cnt_cur = g_beg ? cnt1 : cnt0;
    unsigned bufsel = 0;   // 0 / 1: the slab buffer the MFMAs read
    f32x4 bv[8];
    auto read_b = [&]() __attribute__((always_inline)) {
      const float* bs = Bs + bufsel * B_STRIDE + ti * 1024 + boff;
#pragma unroll
      for (int j = 0; j < 8; ++j) bv[j] = ld4(bs + 2 * j * 64);
    };
    // the wave's three slots of a slab rotate with its slot loads: one per chunk of a 3-chunk superchunk, two and one in a 2-chunk one
    int o0 = my_ord[0], o1 = my_ord[1], o2 = my_ord[2];
    auto next_slot_kind = [&](int on) __attribute__((always_inline)) {   // on = 0: no slot this time — a dump load's description, no rotation
      slot_kind(o0, lds_b + (bufsel ^ 1u) * (B_STRIDE * 4u), slot_desc(a_lo + B_r, B_g), on & (int)((unsigned)(sc + 1 - sc_end) >> 31));   // sc + 1 < sc_end
      const int m = -on, r0 = o0 ^ o1, r1 = o1 ^ o2, r2 = o2 ^ o0;
      o0 ^= r0 & m;   // on: (o0, o1, o2) <- (o1, o2, o0)
      o1 ^= r1 & m;
      o2 ^= r2 & m;
    };
    auto advance = [&]() __attribute__((always_inline)) {
      const int t1 = ti + 1, w = 1 - (int)((unsigned)(t1 - cnt_cur) >> 31);   // w = 1 when the superchunk is complete
      ti = t1 & (w - 1);
      bufsel ^= (unsigned)w;
      sc += w;
      const int g1 = g_cur + w, wg = 1 - (int)((unsigned)(g1 - ng) >> 31);
      g_cur = g1 - (ng & -wg);
      cnt_cur = cnt0 + ((cnt1 - cnt0) & -g_cur);
      slab_next(w);
      const int s1 = stage + 1, ws = 1 - (int)((unsigned)(s1 - STA) >> 31);   // ws = 1: wrap
      stage = s1 - (STA & -ws);
    };

    // prologue: slab 0, filter chunks 0 and 1
    {
      const unsigned so = slot_desc(a_lo + B_r, B_g);
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        slot_kind(my_ord[q], lds_b, so, 1);
        slot_addr(slab_src);
        slot_go();
      }
      slab_next(1);
    }
    issue_a();
    issue_a();   // (two ahead)
#ifdef CONVNET_DIAG
    if (dg & 32) {   // real data in the third stage too, then no more writes into the ring
      const char* keep_p = a_ptr;
      const int k0 = A_i, k1 = A_g, k2 = A_cnt, k3 = A_r, k4 = A_left;
      issue_a();
      a_ptr = keep_p; A_i = k0; A_g = k1; A_cnt = k2; A_r = k3; A_left = k4;
      const unsigned f = lds_f2; lds_f2 = lds_f1; lds_f1 = lds_f0; lds_f0 = f;
      dg_ring_dump = 1;
    }
#endif
    __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0) lgkmcnt(0)
    __syncthreads();

    Split8 fa0[MT], fa1[MT], fb[2];
    static_assert(NTC % 2 == 0, "column parity of fb is carried across chunks");
    // gpw_kernel's fenced steps: one of the six products of split_mac, in its order, over the MT row tiles
    auto mac_step = [&](auto K, const Split8 (&fa)[MT], const Split8& b, int u) __attribute__((always_inline)) {
      constexpr int k = decltype(K)::value;
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const u32x4& av = k == 0 || k == 4 ? fa[t].m : k == 2 ? fa[t].l : fa[t].h;   // (m,m) (h,l) (l,h) (h,m) (m,h) (h,h)
        const u32x4& bw = k == 0 || k == 3 ? b.m : k == 1 ? b.l : b.h;
        acc[t][u] = mma_bf16(av, bw, acc[t][u]);
      }
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x004, 3, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    auto split_pair = [&](int u, int q, Split8& f) __attribute__((always_inline)) {
      const float x0 = bv[2 * q][u], x1 = bv[2 * q + 1][u];
      const unsigned H = pk_bf16(x0, x1);
      const float r0 = x0 - __uint_as_float(H << 16), r1 = x1 - __uint_as_float(H & 0xffff0000u);
      const unsigned M = pk_bf16(r0, r1);
      const float s0 = r0 - __uint_as_float(M << 16), s1 = r1 - __uint_as_float(M & 0xffff0000u);
      f.h[q] = H;
      f.m[q] = M;
      f.l[q] = pk_bf16(s0, s1);
    };
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>;
    using K3 = std::integral_constant<int, 3>;
    using K4 = std::integral_constant<int, 4>;
    using K5 = std::integral_constant<int, 5>;
    // one chunk = one tap of the slab.  Column 0: the filter chunk's three pieces, the first slot's kind and addresses; column 1: that
    // slot's four pieces, the filter iterator, the SECOND slot's kind and addresses; column 2: the second slot's pieces, the counters
    // of the next chunk; the chunk barrier; column 3 with the next chunk's LDS reads and the split of its column 0.
    // The second slot exists only in the first chunk of a two-chunk superchunk (`two`, scalar).  Two chunk bodies chosen by a branch
    // would be the obvious form — and the register allocator then fails to give the 256 accumulators the same registers on both
    // paths (it spilled accumulator tuples at every merge: 1.1 KB of scratch) — so there is ONE body; the second slot's description is
    // always computed (disabled: a dump slot, no rotation of the wave's slots) and only its four load instructions and the count of
    // the closing wait depend on `two` — with the scalar branch INSIDE the asm statement (lds_dma_piece_if_rfl, wait_vm_7_or_11):
    // as C++ `if`s the same five branches made the compiler keep every loop iterator in vector registers (+100 VALU per chunk).
    auto chunk = [&](Split8 (&fa)[MT], Split8 (&fan)[MT]) __attribute__((always_inline)) {
      const int two = sgpr((int)(1u - min((unsigned)(((cnt_cur - 2) | ti)), 1u)));   // cnt_cur == 2 && ti == 0
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u + 1 < NTC; ++u) {
        Split8& fn = fb[(u + 1) & 1];
        const Split8& fc = fb[u & 1];
#if CONVNET_GPV_FILT_LATE
        // Where a staging load sits matters more than how many there are (measured, profiles/r06_gpv_kernel.md: with the filter pieces
        // under column 0 — in the wake of the chunk's twenty ds_read_b128 — they cost ~230 cycles each and the kernel 15 %; the slot
        // pieces under column 1 ~20): column 0 carries bookkeeping only, the slot pieces sit under columns 1 and 2, the filter pieces
        // in the split-free steps 4 and 5 of columns 1 and 2.
        if (u == 2) slot_piece_if(two, K0{});
        split_pair(u + 1, 0, fn);
        if (u == 1) slot_piece(K0{});
        mac_step(K0{}, fa, fc, u);
        if (u == 2) slot_piece_if(two, K1{});
        split_pair(u + 1, 1, fn);
        if (u == 1) slot_piece(K1{});
        mac_step(K1{}, fa, fc, u);
        if (u == 2) slot_piece_if(two, K2{});
        split_pair(u + 1, 2, fn);
        if (u == 0) next_slot_kind(1);
        if (u == 1) slot_piece(K2{});
        mac_step(K2{}, fa, fc, u);
        if (u == 2) slot_piece_if(two, K3{});
        split_pair(u + 1, 3, fn);
        if (u == 0) slot_addr(slab_src);
        if (u == 1) slot_piece(K3{});
        mac_step(K3{}, fa, fc, u);
        if (u == 0) issue_a_addr();
        if (u == 1) {
          issue_a_piece(K0{});
          next_slot_kind(two);
        }
        if (u == 2) {
          issue_a_piece(K2{});
          advance();
        }
        mac_step(K4{}, fa, fc, u);
        if (u == 1) {
          issue_a_piece(K1{});
          slot_addr(slab_src);
        }
        if (u == 2) issue_a_step();
        mac_step(K5{}, fa, fc, u);
#else
        if (u == 2) slot_piece_if(two, K0{});
        split_pair(u + 1, 0, fn);
        if (u == 0) issue_a_addr();
        if (u == 1) slot_piece(K0{});
        mac_step(K0{}, fa, fc, u);
        if (u == 2) slot_piece_if(two, K1{});
        split_pair(u + 1, 1, fn);
        if (u == 0) issue_a_piece(K0{});
        if (u == 1) slot_piece(K1{});
        mac_step(K1{}, fa, fc, u);
        if (u == 2) slot_piece_if(two, K2{});
        split_pair(u + 1, 2, fn);
        if (u == 0) issue_a_piece(K1{});
        if (u == 1) slot_piece(K2{});
        mac_step(K2{}, fa, fc, u);
        if (u == 2) slot_piece_if(two, K3{});
        split_pair(u + 1, 3, fn);
        if (u == 0) issue_a_piece(K2{});
        if (u == 1) slot_piece(K3{});
        mac_step(K3{}, fa, fc, u);
        if (u == 0) next_slot_kind(1);
        if (u == 1) issue_a_step();
        if (u == NTC - 2) advance();
        mac_step(K4{}, fa, fc, u);
        if (u == 0) slot_addr(slab_src);
        if (u == 1) {
          next_slot_kind(two);
          slot_addr(slab_src);
        }
        mac_step(K5{}, fa, fc, u);
#endif
      }
      CHIP_PIN_SPLIT8(fb[(NTC - 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      // everything this wave issued before this chunk's batch (7 loads, or 11) has landed: vmcnt(7 | 11) lgkmcnt(0)
#ifdef CONVNET_DIAG
      if (!(dg & 512)) wait_vm_7_or_11(two);     // 512: no wait for the loads, 256: no chunk barrier (what the synchronisation costs)
      if (!(dg & 256)) __syncthreads();
#else
      wait_vm_7_or_11(two);
      __syncthreads();                      // ... and every other wave's; every wave has read this chunk's A and slab slots out of LDS
#endif
      {
        const Split8& fc = fb[(NTC - 1) & 1];
        Split8& fn = fb[NTC & 1];
        read_b();
        mac_step(K0{}, fa, fc, NTC - 1);
        load_a(stage, fan);
        mac_step(K1{}, fa, fc, NTC - 1);
        split_pair(0, 0, fn);
        mac_step(K2{}, fa, fc, NTC - 1);
        split_pair(0, 1, fn);
        mac_step(K3{}, fa, fc, NTC - 1);
        split_pair(0, 2, fn);
        mac_step(K4{}, fa, fc, NTC - 1);
        split_pair(0, 3, fn);
        mac_step(K5{}, fa, fc, NTC - 1);
        CHIP_PIN_SPLIT8(fn);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    auto run = [&](Split8 (&fa)[MT], Split8 (&fan)[MT]) __attribute__((always_inline)) { chunk(fa, fan); };
    load_a(0, fa0);
    read_b();
#pragma unroll
    for (int q = 0; q < 4; ++q) split_pair(0, q, fb[0]);
    int c = 0;
    if (nchunks & 1) {
      run(fa0, fa1);
      c = 1;
    } else {
#pragma unroll
      for (int t = 0; t < MT; ++t) fa1[t] = fa0[t];
    }
    for (; c < nchunks; c += 2) {
      run(fa1, fa0);
      run(fa0, fa1);
    }
    __builtin_amdgcn_s_waitcnt(0x0070);   // the last two batches (dump / free-stage loads) before the LDS is released
  }

  // ---- epilogue: gg_kernel's, with the unit column mapping (GGParams::patch); every accumulator read straight out of its AGPR
  acc_settle();
  if (tsplit >= 0) {
    float* pp = p.tail_partial + ((size_t)(L - p.tail_first) * p.tail_splits + tsplit) * (size_t)(ROWS * WC * CW);
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        fvec v;
#pragma unroll
        for (int u = 0; u < NTC; ++u) v[u] = acc_elem<true>(acc[t][u][reg]);
        *reinterpret_cast<fvec*>(pp + ((size_t)(t * 16 + reg) * NC + tid) * NTC) = v;
      }
    return;
  }
  gg_epilogue<1, WC, MT, CW, true, true>(p, acc, row_tile, col_tile, split, T.ncols, T.GX, T.G, T.dy0, T.dx0);
}

namespace {

int g_patch_mode = -1;
inline int patch_mode() {
  if (g_patch_mode < 0) g_patch_mode = CHIP_KNOB("CONVNET_GG_PATCH", 3);
  return g_patch_mode;
}

template <typename Kern>
int patch_slots(Kern kern, int threads, size_t lds) {
  CHIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int n = 0;
  CHIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(kern), threads, lds));
#ifdef CONVNET_EMU
  if (const char* e = getenv("CONVNET_EMU_SLOTS")) return atoi(e);   // tests/emu: a small "chip", so that small problems reach the tail split
#endif
  return (n < 1 ? 1 : n) * 256;
}

// gpw_kernel's / gpv_kernel's split-K by wave quantisation (gg_launch_cfg's rule; the unit of a K-range is the superchunk) and their
// LAUNCH POLICY.  A 128 x 512 tile with 512-register waves owns its CU, so a launch is rounds of `slots` blocks and nothing fills a partial one.
// Measured on the MI355X (profiles/r05_wide_kernels.md): one round that fills the chip beats ggp_kernel by 7-12 % (conv3 fprop, conv4,
// conv5 with and without two K-ranges), a K-split over TWO rounds loses to ggp_kernel's tail split (conv3 dgrad: 170 tiles, 3 ranges,
// 435 + 37 us against 343 + 23) — every extra round pays the 256 KB write-out and the prologue again.  So: take the launch when the
// blocks fill >= 85 % of their rounds and either nothing is split or everything runs in one round; otherwise the caller's ggp_kernel.
struct WidePlan {
  int splits;
  bool take;
};
// row tile of the wide kernels: 96 rows (gpv_kernel<3>) for 65..96-row problems — conv2's input gradient — else 128
inline int wide_rows(int R) { return R > 64 && R <= 96 ? 96 : 128; }
// which of them runs a single-problem launch: gpw_kernel keeps what it was validated on (one group of three taps, 128-row tiles)
inline bool wide_is_var(const GGParams& p) { return !(p.ng == 1 && p.gcnt[0] == 3 && wide_rows(p.R) == 128); }
inline WidePlan wide_plan(const GGParams& p, size_t dst_elems, int slots) {
  const int CB = p.KC / BK, TYn = p.TYX / p.TX, rows = wide_rows(p.R);
  const int tiles = divup(p.R, rows) * divup((p.N / 64) * p.G, kWideP);
  const int nsc = CB * TYn * p.ng, kchunks = CB * p.TYX;
  const double block_rate = 230e12 / slots;
  const double fl = 2.0 * rows * (double)(kWideP * 64) * (double)p.K;
  auto finish = [&](int sp) {
    const int cps = divup(nsc, sp);
    return divup(nsc, cps);
  };
  // Round 6 (profiles/r06_batch_policy.txt): ONE round first.  Blocks that all run at once finish together whatever share of the chip
  // they fill — and on a power-limited part a half-empty chip clocks its busy CUs higher (conv4 at 128 images, 129 tiles: 1.35 us per
  // chunk against 2.05 with every CU busy).  Measured against ggp_kernel: gpw_kernel wins at 0.77 fill in three K-ranges (conv3 / conv4 /
  // conv5 at 64 images: 66 tiles, 7-12 %) and at 0.67 in two (conv3 dgrad at 128 images: 219 vs 251 us), loses at 0.66 in ONE K-range
  // (conv3 dgrad at 256 images: 170 whole-K tiles, 401 vs 343 + 23 us with ggp_kernel's tail split) and at 0.50 (129 tiles at 128 images).
  if (tiles <= slots) {
    int best_sp = 1;
    double best_t = 1e30;
    for (int sp = 1; sp <= 16 && tiles * sp <= slots && (sp == 1 || (dst_elems > 0 && kchunks / sp >= 8 && nsc / sp >= 1)); ++sp) {
      double t = (fl / sp) / block_rate;
      if (sp > 1) t += sizeof(float) * (double)dst_elems * (2.0 * sp + 1) / 4.0e12 + 4e-6;
      if (t < best_t * 0.97) {
        best_t = t;
        best_sp = sp;
      }
    }
    const int sp = finish(best_sp);
    const double fill = (double)tiles * sp / slots;
    if (fill >= (wide_is_var(p) ? 0.85 : sp > 1 ? 0.60 : 0.75)) return {sp, true};
  }
  int splits = 1;
  if (dst_elems > 0 && kchunks >= 16) {
    double best_t = 1e30;
    for (int sp = 1; sp <= 16 && kchunks / sp >= 8 && nsc / sp >= 1; ++sp) {
      const double rounds = std::ceil(tiles * (double)sp / slots);
      double t = rounds * (fl / sp) / block_rate;
      if (sp > 1) t += sizeof(float) * (double)dst_elems * (2.0 * sp + 1) / 4.0e12 + 4e-6;
      if (t < best_t * 0.97) {
        best_t = t;
        splits = sp;
      }
    }
  }
  splits = finish(splits);
  const long long blocks = (long long)tiles * splits;
  const long long rounds = (blocks + slots - 1) / slots;
  const bool fill = (double)blocks >= 0.85 * (double)(rounds * slots) || (splits == 1 && rounds >= 4);   // (many rounds: the tail split evens the last)
  return {splits, fill && (splits == 1 || rounds == 1)};
}
constexpr size_t wide_lds(int rows) { return sizeof(float) * (3 * (6 * rows * 4) + 2 * (kWideNS * 1024) + 1024); }   // filter ring + two slabs + the dump slot
constexpr size_t kWideLds = wide_lds(128);
inline int wide_slots(int rows = 128, bool var = false) {
  static const int n = patch_slots(gpw_kernel, 256, kWideLds);
  if (!var) return n;
  static const int n4 = patch_slots(gpv_kernel<4>, 256, wide_lds(128)), n3 = patch_slots(gpv_kernel<3>, 256, wide_lds(96));
  return rows == 96 ? n3 : n4;
}

}  // namespace

// Can this gather (GGParams filled by conv_up_impl / conv_down_impl for ggp_kernel's tap-major pre-split path: KC > 0, apre) run on
// gpp_kernel / gpw_kernel / gpv_kernel?  Fills the tap groups.  A tap row is cut into ssx groups of taps that are ssx apart (one group for
// a stride-1 gather): inside a group neighbouring pixels' taps coincide, slot i of pixel j+1 = slot i+1 of pixel j.
// dst_elems: the size of the whole destination when the launch may be cut in K (0: it may not).
bool patch_shape_ok(GGParams& p, size_t dst_elems) {
  if (!patch_mode() || matrix_path() == 0 || p.KC <= 0 || p.KC % BK != 0) return false;
  if (p.N % 64 != 0 || p.GX < 4 || p.R <= 64) return false;
  if ((size_t)p.SH * p.SW * p.N >= (size_t(1) << 28)) return false;   // the raw build's lane offset spans 3 channel planes in 32 bits
  if (p.ssx < 1 || p.ssx > 2 || (p.dir < 0 && p.ssx != 1)) return false;
  p.ng = p.ssx;
  for (int r = 0; r < p.ng; ++r) {
    const int cnt = p.TX > r ? (p.TX - r + p.ssx - 1) / p.ssx : 0;
    if (cnt < 2 || cnt > 3) return false;
    p.gcnt[r] = cnt;
    p.gb0[r] = p.dir > 0 ? r : r + (cnt - 1) * p.ssx;
  }
  if (p.ng == 1) { p.gcnt[1] = p.gcnt[0]; p.gb0[1] = p.gb0[0]; }
  // the wide kernels: their 12 slots hold ONE wrap per tile: output rows of >= 8 pixels.  gpw_kernel: one group of three taps (the
  // 3 x 3 stride-1 layers); gpv_kernel: any groups of three and two (5 x 5 stride 2: {0,2,4} / {1,3}), 96-row tiles
  if (patch_mode() >= 3) {
    if (p.GX < kWideP) return false;
    return patch_mode() == 4 || wide_plan(p, dst_elems, wide_slots(wide_rows(p.R), wide_is_var(p))).take;   // (mode 4: the parity tests' small shapes, A/B runs)
  }
  return true;
}

// The split of the filter bank (and, for the planes build, of the whole source tensor: C = p.KC channels of SH x SW x N) into planes,
// then the gather-GEMM on them.  `op` / `flops` feed the kernel timers (algorithmic work of the call).
void patch_run(GGParams& p, size_t dst_elems, const char* op, double flops, const PatchBank& bank) {
  constexpr int WR = 2, WC = 2, MT = 2, CW = 128;   // gpp_kernel
  static_assert(128 == WR * MT * 32, "row tile");
  // CONVNET_GG_PATCH / convnet_hip_set_patch_mode: 1 = gpp_kernel on a raw fp32 slab, split by the consumers; 2 = gpp_kernel on bf16
  // planes of the source tensor (one more pass); 3 = the wide kernels (8 units x 128 / 96 rows, raw slab, no producer wave: gpw_kernel,
  // gpv_kernel) where wide_plan takes the launch, 4 = wherever the shape allows
  const int mode = patch_mode();
  const bool wide = mode >= 3, braw = mode != 2;
  const bool var = wide && wide_is_var(p);
  const int ROWS = wide ? wide_rows(p.R) : 128;
  const int PU = wide ? kWideP : kPatchP;           // units per tile
  const int TCOLS = PU * 64;                        // columns per tile
  const int threads = wide ? 256 : WR * WC * 64 + 64;
  const int C = p.KC, HW = p.SH * p.SW, N = p.N, CB = C / BK;
  const size_t elems = (size_t)C * HW * N;
  const int RT = divup(p.R, ROWS);
  {
    // filter bank -> [chunk][row tile][plane, k-group][ROWS rows] bf16 planes
    const size_t welems = (size_t)CB * p.TYX * RT * ROWS * 16;
    u32x4* ap = static_cast<u32x4*>(workspace_aux(welems * 6 + (CHIP_DIAG_KNOB("CONVNET_GPP_DIAG", 0) ? 65536 : 0)));
    filter_planes_rt_launch(bank, ap, p.TYX, ROWS, op);
    p.A = reinterpret_cast<const float*>(ap);
    p.apre = 1;
  }
  u32x4* planes = nullptr;
  if (!braw) {
    planes = static_cast<u32x4*>(workspace_planes(elems * 6));
    KernelTimer timer("act_planes_kernel", op, 0.0, 10.0 * elems);
    const size_t work = (size_t)CB * 2 * HW * (N / 4);
    size_t nb = (work + 255) / 256;
    if (nb > 16384) nb = 16384;
    hipLaunchKernelGGL(act_planes_kernel, dim3((unsigned)nb), dim3(256), 0, stream(), p.src, planes, CB, HW, N / 4);
  }
  p.patch = 1;
  p.IB = N / 64;
  p.planes = planes;
  p.NP = N;
  p.ncols = 0;
  p.row_tiles = RT;
  p.col_tiles = divup(p.IB * p.G, PU);
  p.zero = zero_page();
  p.prio = CHIP_DIAG_KNOB("CONVNET_GPP_DIAG", 0);
  constexpr size_t lds_p = sizeof(float) * (3 * (6 * 128 * 4) + 2 * (6 * 8 * 256)), lds_r = sizeof(float) * (3 * (6 * 128 * 4) + 2 * (4 * 8 * 256));
  const size_t lds_w = wide_lds(ROWS);
  static const int slots_p = patch_slots(gpp_kernel<WR, WC, MT, CW, false>, WR * WC * 64 + 64, lds_p);
  static const int slots_r = patch_slots(gpp_kernel<WR, WC, MT, CW, true>, WR * WC * 64 + 64, lds_r);
  const int slots = wide ? wide_slots(ROWS, var) : braw ? slots_r : slots_p;
  const int tiles = p.row_tiles * p.col_tiles;
  const int TYn = p.TYX / p.TX;
  const int nsc = CB * TYn * p.ng;                    // superchunks of a whole reduction
  const int kchunks = CB * p.TYX;                     // 16-deep chunks
  const double block_rate = 230e12 / slots;
  // split-K by wave quantisation, as gg_launch_cfg; the unit of a K-range is the superchunk
  int splits = 1;
  if (wide) {
    splits = wide_plan(p, dst_elems, slots).splits;
  } else if (dst_elems > 0 && kchunks >= 16) {
    const double fl = 2.0 * ROWS * (double)TCOLS * (double)p.K;
    double best_t = 1e30;
    for (int sp = 1; sp <= 16 && kchunks / sp >= 8 && nsc / sp >= 1; ++sp) {
      const double rounds = std::ceil(tiles * (double)sp / slots);
      double t = rounds * (fl / sp) / block_rate;
      if (sp > 1) t += sizeof(float) * (double)dst_elems * (2.0 * sp + 1) / 4.0e12 + 4e-6;
      if (t < best_t * 0.97) {
        best_t = t;
        splits = sp;
      }
    }
  }
  p.chunks_per_split = divup(nsc, splits);
  splits = divup(nsc, p.chunks_per_split);
  p.splits = splits;
  p.slab = dst_elems;
  p.partial = splits > 1 ? static_cast<float*>(workspace(sizeof(float) * dst_elems * splits)) : nullptr;
  p.tail_splits = 1;
  p.tail_partial = nullptr;
  if (splits == 1 && dst_elems > 0 && tiles > slots && tiles % slots != 0 && kchunks >= 32) {
    const int full = (tiles / slots) * slots, rem = tiles - full;
    const double tile_bytes = sizeof(float) * (double)ROWS * TCOLS;
    const double t_round = 2.0 * ROWS * (double)TCOLS * (double)p.K / block_rate;
    double best = 0.95;
    int best_s = 1;
    for (int s = 2; s <= 8 && kchunks / s >= 8; ++s) {
      const double cost = std::ceil(rem * (double)s / slots) / s + (rem * (s + 1.0) * tile_bytes / 4.0e12 + 6e-6) / t_round;
      if (cost < best) {
        best = cost;
        best_s = s;
      }
    }
#ifdef CONVNET_EMU
    if (const char* e = getenv("CONVNET_EMU_TAIL")) best_s = atoi(e);   // tests/emu: the cost model never picks it at emulation sizes
#endif
    if (best_s > 1) {
      p.tail_first = full;
      p.tail_cps = divup(nsc, best_s);
      p.tail_splits = divup(nsc, p.tail_cps);
      p.tail_tf8 = full / 8;
      p.tail_tt8 = divup(rem * p.tail_splits, 8);
      p.tail_partial = static_cast<float*>(workspace(sizeof(float) * (size_t)rem * p.tail_splits * ROWS * TCOLS));
    }
  }
  dim3 grid(p.tail_splits > 1 ? 8 * (p.tail_tf8 + p.tail_tt8) : ((tiles + 7) / 8) * 8, splits);
  static const GGClassTable kNone = {};
  {
    // (",split" in a timer name is how bench.py prices the kernel on the bf16 pipe: kernel_peak)
    KernelTimer timer(var ? (ROWS == 96 ? "gpv_kernel<96x512,split,raw>" : "gpv_kernel<128x512,split,raw>")
                      : wide ? "gpw_kernel<128x512,split,raw>" : braw ? "gpp_kernel<2,2,2,128,split,raw>" : "gpp_kernel<2,2,2,128,split,planes>", op, flops, 0.0, 0.0);
    if (var && ROWS == 96) hipLaunchKernelGGL(gpv_kernel<3>, grid, dim3(threads), lds_w, stream(), p, kNone);
    else if (var) hipLaunchKernelGGL(gpv_kernel<4>, grid, dim3(threads), lds_w, stream(), p, kNone);
    else if (wide) hipLaunchKernelGGL(gpw_kernel, grid, dim3(threads), lds_w, stream(), p, kNone);
    else if (braw) hipLaunchKernelGGL((gpp_kernel<WR, WC, MT, CW, true>), grid, dim3(threads), lds_r, stream(), p, kNone);
    else hipLaunchKernelGGL((gpp_kernel<WR, WC, MT, CW, false>), grid, dim3(threads), lds_p, stream(), p, kNone);
  }
  if (p.tail_splits > 1) {
    const int rem = tiles - p.tail_first;
    KernelTimer timer("gg_tail_fix_kernel", op, 0.0, sizeof(float) * (double)rem * (p.tail_splits + 1) * ROWS * TCOLS);
    if (wide && ROWS == 96) hipLaunchKernelGGL(gpw_tail_fix_kernel<3>, dim3(rem * kTailFixParts), dim3(256), 0, stream(), p);
    else if (wide) hipLaunchKernelGGL(gpw_tail_fix_kernel<4>, dim3(rem * kTailFixParts), dim3(256), 0, stream(), p);
    else hipLaunchKernelGGL((gg_tail_fix_kernel<WR, WC, MT, CW, true>), dim3(rem * kTailFixParts), dim3(WR * WC * 64), 0, stream(), p);
  }
  if (splits > 1) gg_reduce_launch(p, dst_elems, splits, op);
}

// All stride classes of a strided input gradient in ONE gpv_kernel launch (conv2: 5 x 5 stride 2 -> classes of 3x3, 3x2, 2x3, 2x2
// taps, each a stride-1 gather over the derivative map; ggp_kernel's gg_launch_classes is the fallback).  `base` as conv_down_impl
// fills it for the tap-major pre-split path (KC = filters, apre, dir = -1, ssx = 1); the classes carry their banks as bf16 planes per
// row tile of wide_rows(R) rows (gg_tile_rows gives the same height).  Fills ncols / col_tiles / tile_end.  False: not this shape.
bool patch_classes_ok(const GGParams& base, const GGClassTable& ct) {
  // Mode 4 only (parity tests, A/B runs).  Measured on the MI355X (profiles/r06_gpv_kernel.md): conv2's input gradient runs 1 273 us here against
  // 1 173 on ggp_kernel<1,4,3,64> — an 8-pixel tile cannot leave out the border taps of its edge pixels (7 % more MFMA work than the one-pixel
  // tiles, which skip them per pixel), and the 96-row build issues 6.6 other instructions per MFMA.  The default keeps ggp_kernel.
  if (patch_mode() < 4 || matrix_path() == 0 || !base.apre || base.KC <= 0 || base.KC % BK != 0) return false;
  if (base.N % 64 != 0 || base.R <= 64 || base.dir >= 0 || base.ssx != 1 || base.ssy != 1) return false;
  if ((size_t)base.SH * base.SW * base.N >= (size_t(1) << 28)) return false;
  if (ct.n < 1) return false;
  for (int i = 0; i < ct.n; ++i)
    if (ct.c[i].TX < 2 || ct.c[i].TX > 3 || ct.c[i].GX < kWideP || ct.c[i].TYX % ct.c[i].TX != 0) return false;
  return true;
}
void patch_run_classes(GGParams& p, GGClassTable& ct, const char* op, double flops, double exec) {
  const int ROWS = wide_rows(p.R);
  p.patch = 1;
  p.IB = p.N / 64;
  p.planes = nullptr;
  p.NP = p.N;
  p.ncols = 0;
  p.row_tiles = divup(p.R, ROWS);
  p.zero = zero_page();
  p.prio = CHIP_DIAG_KNOB("CONVNET_GPP_DIAG", 0);
  p.splits = 1;
  p.chunks_per_split = 1 << 24;
  p.partial = nullptr;
  p.slab = 0;
  p.tail_splits = 1;
  p.tail_partial = nullptr;
  p.ng = 1;
  std::sort(ct.c, ct.c + ct.n, [](const GGClass& a, const GGClass& b) { return a.K > b.K; });
  int end = 0;
  for (int i = 0; i < ct.n; ++i) {
    ct.c[i].ncols = 0;
    ct.c[i].col_tiles = divup(p.IB * ct.c[i].G, kWideP);
    end += p.row_tiles * ct.c[i].col_tiles;
    ct.c[i].tile_end = end;
  }
  p.col_tiles = ct.c[0].col_tiles;
  wide_slots(ROWS, true);   // (the kernel's LDS opt-in)
  KernelTimer timer(ROWS == 96 ? "gpv_kernel<96x512,split,raw>" : "gpv_kernel<128x512,split,raw>", op, flops, 0.0, exec);
  if (ROWS == 96) hipLaunchKernelGGL(gpv_kernel<3>, dim3(end), dim3(256), wide_lds(96), stream(), p, ct);
  else hipLaunchKernelGGL(gpv_kernel<4>, dim3(end), dim3(256), wide_lds(128), stream(), p, ct);
}

}  // namespace chip

extern "C" {
void convnet_hip_set_patch_mode(int mode) { chip::g_patch_mode = mode < 0 ? 0 : mode > 4 ? 4 : mode; }
int convnet_hip_get_patch_mode(void) { return chip::patch_mode(); }
}
