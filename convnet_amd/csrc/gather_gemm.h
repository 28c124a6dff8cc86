// Shared device helpers of the gather-GEMM translation units (gather_gemm.hip, patch_gemm.hip): launch parameter blocks, the
// exact bf16 three-way split (Split8), the tile selection and the block-tile write-out.  gfx950 only.
#pragma once
#include <cstring>
#include <type_traits>

#include "common.h"

namespace chip {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int BK = 16;  // reduction depth per LDS stage (gg_kernel)

// Diagnostic build only (tools/gg_trace.cc compiles this file with -DCONVNET_GG_TRACE; the library never does): per-block phase
// timing of gg_kernel's main loop with s_memtime — where a chunk's wall time goes (staging issue / MFMA phase / closing wait +
// barrier), how the two co-resident blocks of a CU share the matrix pipe, and how far block end times spread.
#ifdef CONVNET_GG_TRACE
constexpr bool kTrace = CONVNET_GG_TRACE >= 2;          // 2: per-chunk phases (perturbs the loop by ~20 %); 1: block-level times only
__device__ unsigned long long* g_gg_trace = nullptr;   // 16 words per (block, wave)
#else
constexpr bool kTrace = false;
#endif
__device__ __forceinline__ unsigned long long trace_clock() {
#ifdef CONVNET_GG_TRACE
  return __builtin_amdgcn_s_memtime();
#else
  return 0;
#endif
}

struct GGParams {
  const float* A;
  const float* src;
  float* dst;
  const float* bias;  // per output row, nullable
  float* partial;     // split-K slabs, nullable
  const float* zero;  // >= 16 bytes of zeros: target of out-of-range loads (branch-free fast path)
  int R, K, N;
  int lda;            // A[r + lda*k] (r-contiguous) or A[k + lda*r] (k-contiguous)
  int GX, G;          // output pixel grid of this launch: G = GY*GX pixels, m = oy*GX + ox
  int TX, TYX;        // taps: k = ch*TYX + a*TX + b   (channel-major, KC == 0)
  int apre;           // ggp_kernel split build: A is the pre-split bf16-plane image of the bank (filter_planes_kernel / dgrad_filter_planes_kernel)
  int KC;             // > 0 (ggp_kernel): reduction order k = ((cb*TYX + tap)*BK + c16, channel ch = cb*BK + c16 of KC — every chunk of BK
                      // k-rows is ONE tap of one 16-channel block, taps innermost — over a filter bank re-laid to match
  int SH, SW;         // source image
  int ssy, ssx, y0, x0, dir;       // source row of tap a: oy*ssy + y0 + dir*a
  int DW, DP;         // dest image width, pixels per channel (DH*DW)
  int dsy, dsx, dy0, dx0;          // dest pixel: (oy*dsy + dy0, ox*dsx + dx0)
  int NP;             // column pitch of one output pixel.  The column space of the GEMM is FLAT: column q = m*NP + n is image n of
                      // output pixel m, and wave-column `colid` owns columns [colid*CW, colid*CW + CW).  Vector path: NP = N (N % 4 == 0,
                      // so a 16-byte piece never straddles pixels) — a wave-column spans CW/N pixels when N < CW, and no MFMA column
                      // is padding at any batch size (round 2 gave every pixel ceil(N/CW) wave-columns of its own: at 32 images per GPU
                      // three quarters of every MFMA column were zeros).  Scalar path: NP = ceil(N/CW)*CW, the padded form.
  int ncols;          // ceil(G*NP / CW) wave-columns in total
  int row_tiles, col_tiles;
  int chunks_per_split;  // in BK units
  int splits;
  size_t slab;        // floats per split slab (= dst extent)
  float scaleTargets;
  int relu;
  const float* mask;  // nullable; same layout as dst: out = mask > 0 ? out * post_scale : 0  (fused ReLU' [+dropout'])
  float post_scale;
  // Tail split (tail_splits > 1): tiles [0, tail_first) are whole-K blocks that fill complete rounds of the resident
  // block slots; the remaining tiles — the partial last round — are each cut into tail_splits K-ranges so the last round
  // is full too.  Their raw accumulators go to tail_partial in register order and gg_tail_fix_kernel sums them and runs
  // the normal epilogue.  Block b: XCD k = b & 7 does its run of tail_tf8 full tiles, then its tail_tt8 tail pieces.
  int tail_first, tail_splits, tail_cps, tail_tf8, tail_tt8;
  float* tail_partial;
  int prio;            // issue priority scheme of the main loop (gg_prio_mode())
  int skinny;          // host only: the whole column space is <= 128 columns (an FC layer at <= 128 images per GPU): gg_run picks the
                       // 128-row x 64-column tile instead of padding a 256-column one with zeros
  // gpp_kernel (patch_gemm.hip): the column space is tiled by UNITS, a unit = (64-image block ib, output pixel m), unit index
  // U = ib*G + m; a block tile is kPatchP consecutive units (so normally kPatchP neighbouring pixels of one row for the same 64
  // images) and `col_tile` counts those.  The source operand is read from its bf16 planes (act_planes_kernel).
  // ggp_kernel with a GENERIC reduction order (gk): k-rows in the filter bank's own order k = ch*TYX + a*TX + b (any channel count —
  // conv1's C = 3), K padded to KP = a multiple of 16.  ktab[k] = byte offset of k-row k's source from the output pixel's own
  // source position ((ch*SH + a)*SW + b)*N*4 (fprop, dir = +1); ktab[KP + k] = a << 16 | b for the border tiles' per-lane range check.
  // Padding rows repeat row K - 1 (their filter planes are zeros).
  const unsigned* ktab;
  int gk;
  int patch;           // 1: gpp_kernel's column mapping in gg_epilogue / gg_tail_fix_kernel
  int IB;              // N / 64
  const void* planes;  // u32x4 [channel block cb][SH][SW][IB][region = plane*2 + k-group lh][64 images]: 8 bf16 = channels 16*cb + 2*j + lh of one (pixel, image)
  int ng, gcnt[2], gb0[2];           // tap groups of one tap row: group g = taps gb0[g] + i*dir*ssx, i < gcnt[g] (slot i of the patch)
};
constexpr int kPatchP = 4;   // units (pixels) per gpp_kernel block tile: 4 x 64 images = 256 columns
constexpr int kWideP = 8;    // ... per gpw_kernel block tile: 512 columns
constexpr int kWideNS = 12;  // slots of its slab: 8 units + 2 more taps of a 3-tap row + 2 for ONE wrap (output rows >= 8 wide)

// A strided dgrad is one gather-GEMM per stride class (conv_down_impl); the classes differ only in the fields
// below.  Passing them as a table lets ONE launch cover all classes: block b belongs to the class whose
// [tile_end[c-1], tile_end[c]) range holds b (classes sorted by K, largest first, so the long blocks are
// dispatched first and the short ones fill the tail).  n == 0: ordinary single-problem launch.
struct GGClass {
  const float* A;
  int K, GX, G, TX, TYX, y0, x0, dy0, dx0, ncols, col_tiles, tile_end;
};
constexpr int kMaxClasses = 16;
struct GGClassTable {
  int n;
  GGClass c[kMaxClasses];
};

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// The LDS byte address of a pointer into the block's dynamic shared memory (what M0 and the ds instructions take), and the
// address-space pointer types of the staging builtin.  CONVNET_EMU: the kernels compiled as host code and run on the CPU
// (tests/emu/hip/hip_runtime.h — test infrastructure for kernels that have not been on hardware yet); LDS is an ordinary array there.
// Where a wave exchanges data with ITSELF through LDS and relies on its lanes running in lockstep (legal on the hardware: one wave's LDS
// operations execute in order) the emulation, whose lanes are independent fibers, needs to be told; nothing on the device.
#ifdef CONVNET_EMU
#define CHIP_WAVE_LOCKSTEP() emu::wave_sync()
#else
#define CHIP_WAVE_LOCKSTEP() ((void)0)
#endif
#ifndef CONVNET_EMU
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)(lds_ptr_t)p; }
#else
extern float smem[];   // every kernel's `extern __shared__ float smem[]`
typedef void* lds_ptr_t;
typedef const void* gbl_ptr_t;
inline unsigned lds_addr(const void* p) { return (unsigned)(reinterpret_cast<const char*>(p) - reinterpret_cast<const char*>(smem)); }
inline char* lds_ptr(unsigned a) { return reinterpret_cast<char*>(smem) + a; }
#endif

// LDS-DMA staging pieces of a producer wave, written out: `global_load_lds_dwordx4 v_off, s[base] offset:imm` — a wave-uniform
// 64-bit base in SGPRs plus a 32-bit per-lane byte offset, M0 (the LDS destination) written ONCE for up to four 1 KB pieces whose
// destinations are 1 KB apart (the immediate offset applies to both addresses, so source j is passed as s_j with its own 1024*j
// already subtracted by the caller, or the same pointer four times when the source advances by 1 KB too).  The compiler's own
// selection of __builtin_amdgcn_global_load_lds builds a 64-bit VGPR address per piece (one or two VALU each, issued between the
// co-resident consumer wave's MFMAs) and rewrites M0 per piece; tools/dma_issue: 27 vs 10 cycles per instruction to issue.
// M0 is a reserved register the compiler re-materialises before each of its own uses (-Wno-inline-asm for the clobber note).
// A wave-uniform pointer, pinned to SGPRs: the compiler is free to keep a uniform value in VGPRs (it does after a select), and the
// "s" constraint of the staging instructions below does not move it back.
__device__ __forceinline__ const char* uniform_ptr(const char* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
}
#ifndef CONVNET_EMU
__device__ __forceinline__ void lds_dma4(unsigned voff, const char* s0, const char* s1, const char* s2, const char* s3, unsigned lds) {
  asm volatile("s_mov_b32 m0, %5\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %0, %1\n\t"
               "global_load_lds_dwordx4 %0, %2 offset:1024\n\t"
               "global_load_lds_dwordx4 %0, %3 offset:2048\n\t"
               "global_load_lds_dwordx4 %0, %4 offset:3072"
               ::"v"(voff), "s"(s0), "s"(s1), "s"(s2), "s"(s3), "s"(lds)
               : "memory", "m0");
}
__device__ __forceinline__ void lds_dma3(unsigned voff, const char* s0, const char* s1, const char* s2, unsigned lds) {
  asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %0, %1\n\t"
               "global_load_lds_dwordx4 %0, %2 offset:1024\n\t"
               "global_load_lds_dwordx4 %0, %3 offset:2048"
               ::"v"(voff), "s"(s0), "s"(s1), "s"(s2), "s"(lds)
               : "memory", "m0");
}
// The same groups for SGPR operands that may have been written by a VALU instruction (v_readfirstlane: uniform_ptr) right in front of the
// statement: a vector-memory instruction reading such an SGPR needs five wait states, and the compiler does not pad what is inside an
// asm string.  (ggp_kernel's / gpp_kernel's producers compute their bases with scalar instructions and use the plain forms above.)
__device__ __forceinline__ void lds_dma4_rfl(unsigned voff, const char* s0, const char* s1, const char* s2, const char* s3, unsigned lds) {
  asm volatile("s_nop 4\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %0, %1\n\t"
               "global_load_lds_dwordx4 %0, %2 offset:1024\n\t"
               "global_load_lds_dwordx4 %0, %3 offset:2048\n\t"
               "global_load_lds_dwordx4 %0, %4 offset:3072"
               ::"v"(voff), "s"(s0), "s"(s1), "s"(s2), "s"(s3), "s"(lds)
               : "memory", "m0");
}
__device__ __forceinline__ void lds_dma3_rfl(unsigned voff, const char* s0, const char* s1, const char* s2, unsigned lds) {
  asm volatile("s_nop 4\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %0, %1\n\t"
               "global_load_lds_dwordx4 %0, %2 offset:1024\n\t"
               "global_load_lds_dwordx4 %0, %3 offset:2048"
               ::"v"(voff), "s"(s0), "s"(s1), "s"(s2), "s"(lds)
               : "memory", "m0");
}
// ONE piece, for streams that place their pieces between MFMAs one at a time (a piece costs the issuing wave ~60 cycles, back to back
// they stack: MI355X_MICROARCH.md): piece J of a group whose M0 base is `lds` and whose source j is passed with 1024*j already subtracted.
template <int J>
__device__ __forceinline__ void lds_dma_piece_rfl(unsigned voff, const char* s, unsigned lds) {
  static_assert(J >= 0 && J < 4, "immediate offset");
  if constexpr (J == 0) asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(s), "s"(lds) : "memory", "m0");
  if constexpr (J == 1) asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024" ::"v"(voff), "s"(s), "s"(lds) : "memory", "m0");
  if constexpr (J == 2) asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:2048" ::"v"(voff), "s"(s), "s"(lds) : "memory", "m0");
  if constexpr (J == 3) asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:3072" ::"v"(voff), "s"(s), "s"(lds) : "memory", "m0");
}
// A QUARTER piece: lanes 0..15 only (256 bytes) — the tail of a 96-row filter chunk's per-wave share (9 216 B / 4 waves = 2 304 B = two
// pieces and a quarter).  EXEC is narrowed around the one instruction and restored inside the statement.
template <int J>
__device__ __forceinline__ void lds_dma_quarter_rfl(unsigned voff, const char* s, unsigned lds) {
  static_assert(J >= 0 && J < 4, "immediate offset");
  unsigned long long keep;
  if constexpr (J == 0) asm volatile("s_nop 4\n\ts_mov_b32 m0, %3\n\ts_mov_b64 %0, exec\n\ts_mov_b64 exec, 0xffff\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b64 exec, %0" : "=&s"(keep) : "v"(voff), "s"(s), "s"(lds) : "memory", "m0");
  if constexpr (J == 1) asm volatile("s_nop 4\n\ts_mov_b32 m0, %3\n\ts_mov_b64 %0, exec\n\ts_mov_b64 exec, 0xffff\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\ts_mov_b64 exec, %0" : "=&s"(keep) : "v"(voff), "s"(s), "s"(lds) : "memory", "m0");
  if constexpr (J == 2) asm volatile("s_nop 4\n\ts_mov_b32 m0, %3\n\ts_mov_b64 %0, exec\n\ts_mov_b64 exec, 0xffff\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048\n\ts_mov_b64 exec, %0" : "=&s"(keep) : "v"(voff), "s"(s), "s"(lds) : "memory", "m0");
  if constexpr (J == 3) asm volatile("s_nop 4\n\ts_mov_b32 m0, %3\n\ts_mov_b64 %0, exec\n\ts_mov_b64 exec, 0xffff\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072\n\ts_mov_b64 exec, %0" : "=&s"(keep) : "v"(voff), "s"(s), "s"(lds) : "memory", "m0");
}
// A piece that exists only when the wave-uniform flag `on` is set, the scalar branch around it INSIDE the statement: written as
// `if (on) piece` the compiler moved every loop iterator of gpv_kernel onto the vector ALU (+100 VALU per chunk); what it cannot see
// it cannot restructure.  Likewise the closing wait whose count depends on the flag.
template <int J>
__device__ __forceinline__ void lds_dma_piece_if_rfl(int on, unsigned voff, const char* s, unsigned lds) {
  static_assert(J >= 0 && J < 4, "immediate offset");
  if constexpr (J == 0) asm volatile("s_cmp_eq_u32 %3, 0\n\ts_cbranch_scc1 1f\n\ts_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n1:" ::"v"(voff), "s"(s), "s"(lds), "s"(on) : "memory", "m0", "scc");
  if constexpr (J == 1) asm volatile("s_cmp_eq_u32 %3, 0\n\ts_cbranch_scc1 1f\n\ts_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024\n1:" ::"v"(voff), "s"(s), "s"(lds), "s"(on) : "memory", "m0", "scc");
  if constexpr (J == 2) asm volatile("s_cmp_eq_u32 %3, 0\n\ts_cbranch_scc1 1f\n\ts_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:2048\n1:" ::"v"(voff), "s"(s), "s"(lds), "s"(on) : "memory", "m0", "scc");
  if constexpr (J == 3) asm volatile("s_cmp_eq_u32 %3, 0\n\ts_cbranch_scc1 1f\n\ts_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:3072\n1:" ::"v"(voff), "s"(s), "s"(lds), "s"(on) : "memory", "m0", "scc");
}
// s_waitcnt vmcnt(on ? 11 : 7) lgkmcnt(0)
__device__ __forceinline__ void wait_vm_7_or_11(int on) {
  asm volatile("s_waitcnt vmcnt(11) lgkmcnt(0)\n\ts_cmp_lg_u32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_waitcnt vmcnt(7)\n1:" ::"s"(on) : "memory", "scc");
}
__device__ __forceinline__ void lds_dma2(unsigned voff, const char* s0, const char* s1, unsigned lds) {
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %0, %1\n\t"
               "global_load_lds_dwordx4 %0, %2 offset:1024"
               ::"v"(voff), "s"(s0), "s"(s1), "s"(lds)
               : "memory", "m0");
}
__device__ __forceinline__ void lds_dma1(unsigned voff, const char* s0, unsigned lds) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(s0), "s"(lds) : "memory", "m0");
}

// 16 bytes per lane to an LDS byte address (the producer's zero fill)
__device__ __forceinline__ void lds_store16(unsigned addr, u32x4 v) { asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
// "this Split8 is complete HERE": an empty statement that uses its three registers (the compiler otherwise sinks a split towards its use)
#define CHIP_PIN_SPLIT8(f) asm volatile("" ::"v"((f).h), "v"((f).m), "v"((f).l))
#else   // CONVNET_EMU: the same data movement as an immediate copy — piece j: 16 bytes per lane from s_j + 1024*j + voff to lds + 1024*j + 16*lane
inline void emu_piece(unsigned voff, const char* s, unsigned lds, int j) {
  std::memcpy(lds_ptr(lds + 1024u * j) + 16 * emu::lane_id(), s + 1024 * j + voff, 16);
}
inline void lds_dma4(unsigned voff, const char* s0, const char* s1, const char* s2, const char* s3, unsigned lds) {
  emu_piece(voff, s0, lds, 0); emu_piece(voff, s1, lds, 1); emu_piece(voff, s2, lds, 2); emu_piece(voff, s3, lds, 3);
}
inline void lds_dma3(unsigned voff, const char* s0, const char* s1, const char* s2, unsigned lds) {
  emu_piece(voff, s0, lds, 0); emu_piece(voff, s1, lds, 1); emu_piece(voff, s2, lds, 2);
}
inline void lds_dma4_rfl(unsigned voff, const char* s0, const char* s1, const char* s2, const char* s3, unsigned lds) { lds_dma4(voff, s0, s1, s2, s3, lds); }
inline void lds_dma3_rfl(unsigned voff, const char* s0, const char* s1, const char* s2, unsigned lds) { lds_dma3(voff, s0, s1, s2, lds); }
template <int J>
inline void lds_dma_piece_rfl(unsigned voff, const char* s, unsigned lds) { emu_piece(voff, s, lds, J); }
template <int J>
inline void lds_dma_quarter_rfl(unsigned voff, const char* s, unsigned lds) {
  if (emu::lane_id() < 16) emu_piece(voff, s, lds, J);
}
template <int J>
inline void lds_dma_piece_if_rfl(int on, unsigned voff, const char* s, unsigned lds) {
  if (on) emu_piece(voff, s, lds, J);
}
inline void wait_vm_7_or_11(int) {}
inline void lds_dma2(unsigned voff, const char* s0, const char* s1, unsigned lds) { emu_piece(voff, s0, lds, 0); emu_piece(voff, s1, lds, 1); }
inline void lds_dma1(unsigned voff, const char* s0, unsigned lds) { emu_piece(voff, s0, lds, 0); }
inline void lds_store16(unsigned addr, u32x4 v) { std::memcpy(lds_ptr(addr), &v, 16); }
#define CHIP_PIN_SPLIT8(f) ((void)0)
#endif
// compile-time loop: f(integral_constant<int, I>) for I in [B, E) — keeps register-array indices constant
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// XCD-aware block -> tile map: hardware places block b on XCD b%8 (observed; speed only).  Give
// each XCD a contiguous run of logical tiles, ordered row-tile-fastest, so blocks that share a
// source-column tile run on one XCD's L2 back to back.
__device__ __forceinline__ int xcd_remap(int b, int total) {
  const int per = (total + 7) >> 3;
  return (b & 7) * per + (b >> 3);
}

// The write-out of one block tile: accumulate into / overwrite the destination with the fused bias, ReLU and mask
// options (fin), or store the raw sums into this split's slab.  Shared by gg_kernel and gg_tail_fix_kernel.
// reg_lo / reg_hi: the accumulator registers (of every row tile) this call writes out — all 16 from the GEMM kernels, a quarter from
// each of the four blocks gg_tail_fix_kernel gives a tail tile.
// acc_elem<true>: one accumulator element read out of ITS accumulation register by an explicit v_accvgpr_read.  gpw_kernel's 256
// accumulators live in AGPRs; left to itself the compiler (ROCm 7.2) copies whole 16-register tuples into VGPRs for the write-out,
// spills some of them, and its reload drops element 2 of a spilled tuple (found on the MI355X: one row of one row tile wrong in one
// of four images whenever the split-K write-out ran — profiles/r05_wide_kernels.md).  Element by element there is nothing to spill.
template <bool AGPR>
__device__ __forceinline__ float acc_elem(float x) {
#ifndef CONVNET_EMU
  if constexpr (AGPR) {
    float r;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(r) : "a"(x));
    return r;
  }
#endif
  return x;
}
// in front of the first acc_elem<true>: the last MFMA's result registers are not readable for 18 cycles and the compiler's hazard
// padding does not look into inline asm
__device__ __forceinline__ void acc_settle() {
#ifndef CONVNET_EMU
  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
#endif
}

template <int WR, int WC, int MT, int CW, bool VEC, bool ACC_AGPR = false>
__device__ __forceinline__ void gg_epilogue(const GGParams& p, f32x16 (&acc)[MT][CW / 32], int row_tile, int col_tile, int split,
                                            int pncols, int pGX, int pG, int pdy0, int pdx0, int reg_lo = 0, int reg_hi = 16) {
  constexpr int NTC = CW / 32;
  using fvec = __attribute__((ext_vector_type(NTC))) float;
  constexpr int ROWS = WR * MT * 32;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wr = wave / WC, wc = wave % WC;
  const int li = lane & 31, lh = lane >> 5;
  const int r0 = row_tile * ROWS;
  const int N = p.N;
  int m, n;
  if (p.patch) {
    // unit tile: wave-column wc holds CW/64 units, lane li the images NTC*li % 64 .. + NTC - 1 of unit (NTC*li)/64 of them
    // units per tile: kPatchP (gpp_kernel), kWideP for gpw_kernel's <1, 4, 4, 128> and gpv_kernel's <1, 4, 3 | 4, 128> — spelled so that every
    // other instantiation keeps the constant (and the machine code) it was validated with
    constexpr int PU = (WR == 1 && WC == 4 && (MT == 4 || MT == 3) && CW == 128) ? kWideP : kPatchP;
    const int U = col_tile * PU + wc * (CW / 64) + (NTC * li) / 64;
    const int ib = U / pG;
    if (ib >= p.IB) return;
    m = U - ib * pG;
    n = ib * 64 + (NTC * li) % 64;
  } else {
    const int colid = col_tile * WC + wc;
    if (colid >= pncols) return;
    const int q = colid * CW + NTC * li;   // flat column (GGParams::NP)
    m = q / p.NP;
    n = q - m * p.NP;
    if (m >= pG || n >= N) return;
  }
  const int oy = m / pGX, ox = m - oy * pGX;
  const int dpix = (oy * p.dsy + pdy0) * p.DW + ox * p.dsx + pdx0;
  float* base = (p.splits > 1 ? p.partial + (size_t)split * p.slab : p.dst) + (size_t)dpix * N + n;
  const bool fin = p.splits == 1;
#pragma unroll
  for (int t = 0; t < MT; ++t) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = r0 + wr * MT * 32 + t * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
      if (row >= p.R || reg < reg_lo || reg >= reg_hi) continue;
      fvec v;
#pragma unroll
      for (int u = 0; u < NTC; ++u) v[u] = acc_elem<ACC_AGPR>(acc[t][u][reg]);
      float* dp = base + (size_t)row * p.DP * N;
      if (fin) {
        const float bv = p.bias ? p.bias[row] : 0.f;
        if (VEC) {
          if (p.scaleTargets != 0.f) {
            const fvec o = *reinterpret_cast<const fvec*>(dp);
            v = p.scaleTargets * o + v;
          }
          v = v + bv;
          if (p.relu) {
#pragma unroll
            for (int e = 0; e < NTC; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
          }
          if (p.mask) {
            const fvec mk = *reinterpret_cast<const fvec*>(p.mask + (dp - p.dst));
#pragma unroll
            for (int e = 0; e < NTC; ++e) v[e] = mk[e] > 0.f ? v[e] * p.post_scale : 0.f;
          }
          *reinterpret_cast<fvec*>(dp) = v;
        } else {
#pragma unroll
          for (int e = 0; e < NTC; ++e) {
            if (n + e < N) {
              float x = v[e];
              if (p.scaleTargets != 0.f) x = p.scaleTargets * dp[e] + x;
              x += bv;
              if (p.relu) x = x > 0.f ? x : 0.f;
              if (p.mask) x = p.mask[(dp - p.dst) + e] > 0.f ? x * p.post_scale : 0.f;
              dp[e] = x;
            }
          }
        }
      } else {
        if (VEC) {
          *reinterpret_cast<fvec*>(dp) = v;
        } else {
#pragma unroll
          for (int e = 0; e < NTC; ++e)
            if (n + e < N) dp[e] = v[e];
        }
      }
    }
  }
}

// -------------------------------------------------------------------------------------------------
// fp32 products on the bf16 matrix pipe (the default matrix path; convnet_hip_set_matrix_path / CONVNET_GG_SPLIT=0 select the fp32
// instruction instead): every operand value is split EXACTLY into three bf16
// terms, x = h + m + l with h = rne8(x), m = rne8(x - h), l = x - h - m (the second residual has at most 8 significant bits), and
// a*b is accumulated in fp32 as hh + hm + mh + hl + lh + mm by six v_mfma_f32_32x32x16_bf16.  The three dropped cross terms
// (ml, lm, ll) are below 2^-23 of the product.  tools/split_gemm.hip measures it against double on conv4's reduction length:
// max error 4.08 x 2^-24 of sum|ab| vs 4.55 x 2^-24 for v_mfma_f32_32x32x2_f32 on the same data — the fp32 accumulation rounding
// dominates both.  Six 32-cycle instructions replace eight 64-cycle ones per 32 x 32 x 16 block: 2.67x the matrix-pipe rate,
// paid for with ~5.5 VALU per operand element for the split.
struct Split8 {
  u32x4 h, m, l;   // 8 bf16 each: one A or B operand of the MFMA
};
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  f32x2_t v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ void split8(const float (&x)[8], Split8& s) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float x0 = x[2 * q], x1 = x[2 * q + 1];
    const unsigned H = pk_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(H << 16), r1 = x1 - __uint_as_float(H & 0xffff0000u);
    const unsigned M = pk_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(M << 16), s1 = r1 - __uint_as_float(M & 0xffff0000u);
    s.h[q] = H;
    s.m[q] = M;
    s.l[q] = pk_bf16(s0, s1);
  }
}
// Range of the split.  h = rne8(x) is finite for |x| <= 0x7F7F7FFF (3.396e38); above it — the top 0.2 % of the fp32 range and +-inf —
// h is a bf16 inf and the residual x - h is NaN.  The in-loop split8 above carries no range check (one more VALU per element in
// loops that are VALU-limited): such an ACTIVATION / DERIVATIVE value makes the outputs it touches NaN where the fp32 instruction
// gives +-inf or a huge finite number (documented in include/convnet_hip.h, pinned by tests/test_split_arithmetic_gpu.py).  The
// FILTER operand is split outside the loops (filter_planes_kernel, dgrad_filter_planes_kernel) and saturates instead: +-inf and
// above-range finite values enter as +-bf16 max (3.3895e38); NaN stays NaN.
__device__ __forceinline__ void split8_sat(float (&x)[8], Split8& s) {
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (fabsf(x[j]) > __uint_as_float(0x7F7F7FFFu)) x[j] = copysignf(__uint_as_float(0x7F7F0000u), x[j]);   // false for NaN
  split8(x, s);
}
__device__ __forceinline__ f32x16 mma_bf16(u32x4 a, u32x4 b, f32x16 c) {
  typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mma_bf16(u32x4 a, u32x4 b, f32x4 c) {   // 16 x 16 tile, 32 k-slots: lane (li, lh) holds k = 8*lh .. 8*lh + 7
  typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// acc += a*b from the split operands: hh + hm + mh + hl + lh + mm, smallest first.  (Measured with m*l and l*m added — every dropped
// term then is l*l <= 2^-32 of the product: 13-15 % slower on every layer, same parity results; not kept.)
template <typename Acc>
__device__ __forceinline__ Acc split_mac(const Split8& a, const Split8& b, Acc v) {
  v = mma_bf16(a.m, b.m, v);
  v = mma_bf16(a.h, b.l, v);
  v = mma_bf16(a.l, b.h, v);
  v = mma_bf16(a.h, b.m, v);
  v = mma_bf16(a.m, b.h, v);
  v = mma_bf16(a.h, b.h, v);
  return v;
}

// Which tile does this block compute, and with which per-class fields?  Shared by gg_kernel and ggp_kernel.
struct GGTile {
  const float* A;
  int K, GX, G, TX, TYX, y0, x0, dy0, dx0, ncols, col_tiles;
  int L, tsplit;   // logical tile; tsplit >= 0: one K-range of a tail tile
};
__device__ __forceinline__ bool gg_select_tile(const GGParams& p, const GGClassTable& ct, GGTile& t) {
  t.A = p.A; t.K = p.K; t.GX = p.GX; t.G = p.G; t.TX = p.TX; t.TYX = p.TYX; t.y0 = p.y0; t.x0 = p.x0; t.dy0 = p.dy0; t.dx0 = p.dx0;
  t.ncols = p.ncols; t.col_tiles = p.col_tiles; t.tsplit = -1;
  if (ct.n > 0) {
    const int b = blockIdx.x;
    if (b >= ct.c[ct.n - 1].tile_end) return false;
    int c = 0;
    while (b >= ct.c[c].tile_end) ++c;
    const int cbeg = c > 0 ? ct.c[c - 1].tile_end : 0;
    {
      // XCD-aware order inside the class (hardware places block b on XCD b%8): the blocks of this class that land on one XCD
      // take a CONTIGUOUS run of its logical tiles, so neighbouring pixels — which gather overlapping taps — share one L2.
      // Without it every XCD saw pixels 8 apart and conv2's dgrad fetched each deriv element once per tap (4.2 GiB for a
      // 169 MiB tensor, profiles/r01_pmc_traffic_bench.json).  Exact counts, no padding blocks: residue r = i%8 owns
      // q + (r < m) tiles starting at r*q + min(r, m).
      const int i = b - cbeg, T = ct.c[c].tile_end - cbeg;
      const int q = T >> 3, m = T & 7, r = i & 7;
      t.L = r * q + (r < m ? r : m) + (i >> 3);
    }
    const GGClass& k = ct.c[c];
    t.A = k.A; t.K = k.K; t.GX = k.GX; t.G = k.G; t.TX = k.TX; t.TYX = k.TYX;
    t.y0 = k.y0; t.x0 = k.x0; t.dy0 = k.dy0; t.dx0 = k.dx0; t.ncols = k.ncols; t.col_tiles = k.col_tiles;
  } else if (p.tail_splits > 1) {
    const int k = blockIdx.x & 7, i = blockIdx.x >> 3;
    if (i < p.tail_tf8) {
      t.L = k * p.tail_tf8 + i;
    } else {
      const int j = k * p.tail_tt8 + (i - p.tail_tf8);
      if (j >= (p.row_tiles * t.col_tiles - p.tail_first) * p.tail_splits) return false;
      t.L = p.tail_first + j / p.tail_splits;
      t.tsplit = j % p.tail_splits;
    }
  } else {
    const int tiles = p.row_tiles * t.col_tiles;
    const int per = (tiles + 7) >> 3;
    t.L = xcd_remap(blockIdx.x, tiles);
    if (t.L >= tiles || (int)blockIdx.x >= per * 8) return false;
  }
  return true;
}

// Sums the tail_splits partial tiles of one tail tile in fixed order and applies the normal epilogue.  FOUR blocks per tile, each
// takes accumulator registers 4q .. 4q+3 of every row tile (a quarter of the tile's rows): a launch of `rem` blocks left most of
// the chip idle (conv2 fprop: 72 tiles, 61 us for 37 MB).
constexpr int kTailFixParts = 4;
template <int WR, int WC, int MT, int CW, bool VEC>
__global__ __launch_bounds__(WR* WC * 64) void gg_tail_fix_kernel(const GGParams p) {
  constexpr int NT = WR * WC * 64, NTC = CW / 32, ROWS = WR * MT * 32;
  using fvec = __attribute__((ext_vector_type(NTC))) float;
  const int tile = blockIdx.x / kTailFixParts, part = blockIdx.x % kTailFixParts;
  const int L = p.tail_first + tile;
  const int tid = threadIdx.x;
  const int reg_lo = part * (16 / kTailFixParts), reg_hi = reg_lo + 16 / kTailFixParts;
  f32x16 acc[MT][NTC];
  const float* pp = p.tail_partial + (size_t)tile * p.tail_splits * (size_t)(ROWS * WC * CW);
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      if (reg < reg_lo || reg >= reg_hi) continue;
      fvec v = *reinterpret_cast<const fvec*>(pp + ((size_t)(t * 16 + reg) * NT + tid) * NTC);
      for (int sp = 1; sp < p.tail_splits; ++sp)
        v += *reinterpret_cast<const fvec*>(pp + (size_t)sp * (ROWS * WC * CW) + ((size_t)(t * 16 + reg) * NT + tid) * NTC);
#pragma unroll
      for (int u = 0; u < NTC; ++u) acc[t][u][reg] = v[u];
    }
  gg_epilogue<WR, WC, MT, CW, VEC>(p, acc, L % p.row_tiles, L / p.row_tiles, 0, p.ncols, p.GX, p.G, p.dy0, p.dx0, reg_lo, reg_hi);
}

// gpp_kernel (patch_gemm.hip): the patch-resident gather-GEMM on pre-split source planes.  patch_shape_ok() decides whether a gather
// prepared for ggp_kernel's pre-split tap-major path can take it (and fills the tap groups); patch_run() splits the source tensor and launches.
struct PatchBank {   // where the filter bank of the call comes from: forward filters, or one stride class of the input-gradient bank
  const float* W;
  int F, C, Ky, Kx, cy, cx, sy, sx, TYc, TXc;
  bool dgrad;
};
// the bank as bf16 planes per row tile of TH rows (patch_gemm.hip: filter_planes_rt_kernel); out holds 96 * (KC/16) * TYX * ceil(R/TH)*TH bytes
void filter_planes_rt_launch(const PatchBank& bank, void* out, int TYX, int TH, const char* op);
void filter_planes_gk_launch(const float* W, void* out, int F, int K, int KP, int TH, const char* op);   // generic k order (conv1)
bool patch_shape_ok(GGParams& p, size_t dst_elems);
void patch_run(GGParams& p, size_t dst_elems, const char* op, double flops, const PatchBank& bank);
// every stride class of a strided input gradient in one gpv_kernel launch (the classes' banks already as bf16 planes per row tile)
bool patch_classes_ok(const GGParams& base, const GGClassTable& ct);
void patch_run_classes(GGParams& p, GGClassTable& ct, const char* op, double flops, double exec);

// gfc_kernel (fewc_conv.hip): conv fprop of a few-channel, wide-filter, stride-2 layer (AlexNet conv1) as a patch-resident gather-GEMM
// with the filter bank resident in LDS; takes the launch and returns true where the shape is of that kind
bool gfc_try(const float* images, const float* filters, const float* bias, float* targets, int N, int C, int H, int W, int F, int Ky, int Kx,
             int sy, int sx, int pady, int padx, int My, int Mx, float scaleTargets, int relu, double flops);

// -------------------------------------------------------------------------------------------------
// wg_kernel / wgw_kernel: dW[k, f] over the (pixel, image) reduction (gather_gemm.hip, wgrad_wide.hip).
// -------------------------------------------------------------------------------------------------
struct WGParams {
  const float* src;   // layer input  (N, SH*SW*C)
  const float* dout;  // output deriv (N, M*F), M = GY*GX
  float* dst;         // dW (F, K)  column-major: dst[f + F*k]
  float* partial;     // [splits][K (+1 with bias_dst)][F]
  const float* zero;  // zero page for out-of-range loads; floats [32, 36) of the page hold 1.0f
  float* bias_dst;    // nullable: db (1, F).  The bias gradient is the dW row of a virtual tap k == K whose input is
                      // the constant 1 (db[f] = sum over pixels, images of dout) — it rides in a padding row of the tile.
  int K, F, N;
  int GX, M;
  int TX, TYX;
  int SH, SW;
  int ssy, ssx, y0, x0;
  int nchunk;         // ceil(N/32) image chunks per pixel
  int chunks_total;   // M*nchunk
  int chunks_per_split;
  int splits;
  int k_tiles, f_tiles;
  float scaleTargets, scaleOutput;
  int prio;           // issue priority scheme (wg_prio_mode()): 0 none, 1 MFMA phase high, 2 staging phase high
  int wide;           // the 128 x 128 tile's write-out goes through LDS and leaves as 16-byte stores (F % 4 == 0, 16-byte aligned targets)
};

constexpr int WG_NB = 32;          // images per stage
constexpr int WG_PITCH = WG_NB + 4;  // conflict-free ds_read_b128 across rows

// slab reduce of the weight-gradient kernels (gather_gemm.hip): dst (and the fused bias row) = scaleTargets*dst + scaleOutput * sum of
// the `splits` slabs of `total` floats in p.partial, in fixed order; two levels when groups > 1 (stage area behind the slabs)
void wg_reduce_launch(const WGParams& p, size_t total, int splits, int groups, const char* op);
// wgw_kernel (wgrad_wide.hip): takes the launch and returns true when the wide tile is selected (convnet_hip_set_wgrad_tile) and applies
bool wgw_try(WGParams& p, bool vec, bool split_products, const char* op, double flops, double exec);

// split-K second stage (gather_gemm.hip): dst = scaleTargets*dst + sum of the slabs, with the fused bias / ReLU / mask options of p
void gg_reduce_launch(const GGParams& p, size_t dst_elems, int splits, const char* op);

}  // namespace chip
