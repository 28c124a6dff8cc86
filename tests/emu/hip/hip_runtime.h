// Stand-in for <hip/hip_runtime.h>: compiles convnet_amd/csrc/*.hip as plain host C++ (clang) and runs the kernels on the CPU, one block
// at a time, every thread of the block a fiber (ucontext) on one OS thread.  Test infrastructure (tests/test_emulated_kernels.py), not
// a product path: it exists to run kernels FUNCTIONALLY that have not been on hardware yet (gpw_kernel, wgw_kernel) — calibrated on one
// that has (gpp_kernel).  What is modelled: thread / block indices, dynamic LDS (one array), __syncthreads / s_barrier, wave-collectives
// (readfirstlane, readlane, ballot, the bf16 MFMAs with the register layout the kernels assume), LDS-DMA as an immediate copy (the
// staging helpers of gather_gemm.h have a CONVNET_EMU branch).  Not modelled: timing, s_waitcnt (loads land at once — the scheduling
// models of tests/test_patch_wide_cpu.py / test_wgrad_wide_cpu.py cover late landing), hazards, bank conflicts.
#pragma once
#define CONVNET_EMU 1
#include <ucontext.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef void* hipStream_t;
typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
inline const char* hipGetErrorString(hipError_t) { return "emu"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, const void*, int, size_t) { *n = 1; return hipSuccess; }

namespace emu {

struct Wave {
  int arrived = 0;
  unsigned gen = 0;
  alignas(16) unsigned char slot[64][64];   // what each lane publishes for a collective
};
struct Block;
struct Fiber {
  ucontext_t uc;
  std::unique_ptr<char[]> stack;
  dim3 tid;
  int lane = 0, wave = 0;
  bool done = false;
  Block* blk = nullptr;
};
struct Block {
  dim3 bid, bdim, gdim;
  std::vector<Fiber> fibers;
  std::vector<Wave> waves;
  int barrier_arrived = 0;
  unsigned barrier_gen = 0;
  ucontext_t sched;
  const std::function<void()>* fn = nullptr;
};
inline Fiber*& cur() {
  static Fiber* f = nullptr;
  return f;
}
inline void yield() { swapcontext(&cur()->uc, &cur()->blk->sched); }
inline int live_lanes(Block* b, int w) {
  int n = 0;
  for (auto& f : b->fibers) n += (f.wave == w && !f.done) ? 1 : 0;
  return n;
}
inline void wave_sync() {
  Fiber* f = cur();
  Wave& w = f->blk->waves[f->wave];
  const unsigned g = w.gen;
  if (++w.arrived >= live_lanes(f->blk, f->wave)) {
    w.arrived = 0;
    ++w.gen;
  } else {
    while (w.gen == g) yield();
  }
}
inline void block_barrier() {
  Fiber* f = cur();
  Block* b = f->blk;
  int live = 0;
  for (auto& x : b->fibers) live += x.done ? 0 : 1;
  const unsigned g = b->barrier_gen;
  if (++b->barrier_arrived >= live) {
    b->barrier_arrived = 0;
    ++b->barrier_gen;
  } else {
    while (b->barrier_gen == g) yield();
  }
}
inline void trampoline() {
  Fiber* f = cur();
  (*f->blk->fn)();
  f->done = true;
  // a lane that leaves may complete a collective / barrier the others wait in
  Block* b = f->blk;
  Wave& w = b->waves[f->wave];
  if (w.arrived > 0 && w.arrived >= live_lanes(b, f->wave)) { w.arrived = 0; ++w.gen; }
  int live = 0;
  for (auto& x : b->fibers) live += x.done ? 0 : 1;
  if (b->barrier_arrived > 0 && b->barrier_arrived >= live) { b->barrier_arrived = 0; ++b->barrier_gen; }
  swapcontext(&f->uc, &b->sched);
}
inline void launch(dim3 grid, dim3 block, size_t /*lds*/, const std::function<void()>& fn) {
  constexpr size_t kStack = 256 * 1024;
  for (unsigned bz = 0; bz < grid.z; ++bz)
   for (unsigned by = 0; by < grid.y; ++by)
    for (unsigned bx = 0; bx < grid.x; ++bx) {
      Block b;
      b.bid = dim3(bx, by, bz);
      b.bdim = block;
      b.gdim = grid;
      b.fn = &fn;
      const int nt = (int)block.x;
      b.fibers.resize(nt);
      b.waves.resize((nt + 63) / 64);
      for (int t = 0; t < nt; ++t) {
        Fiber& f = b.fibers[t];
        f.tid = dim3(t);
        f.lane = t & 63;
        f.wave = t >> 6;
        f.blk = &b;
        f.stack.reset(new char[kStack]);
        getcontext(&f.uc);
        f.uc.uc_stack.ss_sp = f.stack.get();
        f.uc.uc_stack.ss_size = kStack;
        f.uc.uc_link = nullptr;
        makecontext(&f.uc, (void (*)())trampoline, 0);
      }
      for (;;) {
        int running = 0;
        for (auto& f : b.fibers) {
          if (f.done) continue;
          ++running;
          cur() = &f;
          swapcontext(&b.sched, &f.uc);
        }
        if (!running) break;
      }
      cur() = nullptr;
    }
}

// ---- collectives ------------------------------------------------------------------------------------------------------------------
template <typename T>
inline T publish_and_read(const T& mine, int from_lane) {
  static_assert(sizeof(T) <= 64, "slot");
  Fiber* f = cur();
  Wave& w = f->blk->waves[f->wave];
  std::memcpy(w.slot[f->lane], &mine, sizeof(T));
  wave_sync();
  T out;
  std::memcpy(&out, w.slot[from_lane], sizeof(T));
  wave_sync();
  return out;
}
inline int first_live_lane() {
  Fiber* f = cur();
  for (auto& x : f->blk->fibers)
    if (x.wave == f->wave && !x.done) return x.lane;
  return 0;
}
inline int readfirstlane(int v) { return publish_and_read(v, first_live_lane()); }
inline int readlane(int v, int lane) { return publish_and_read(v, lane & 63); }
inline unsigned long long ballot(bool p) {
  Fiber* f = cur();
  Wave& w = f->blk->waves[f->wave];
  const int mine = p ? 1 : 0;
  std::memcpy(w.slot[f->lane], &mine, sizeof mine);
  wave_sync();
  unsigned long long m = 0;
  for (auto& x : f->blk->fibers)
    if (x.wave == f->wave && !x.done) {
      int v;
      std::memcpy(&v, w.slot[x.lane], sizeof v);
      if (v) m |= 1ull << x.lane;
    }
  wave_sync();
  return m;
}
inline float bf16_to_float(unsigned short h) {
  const unsigned u = (unsigned)h << 16;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}
// v_mfma_f32_32x32x16_bf16 with the layout the kernels assume: lane = li + 32*lh; A operand = row li, k-slots 8*lh .. 8*lh + 7; B operand
// = column li, the same k-slots; D register r of lane (li, lh) = row (r & 3) + 8*(r >> 2) + 4*lh, column li.
template <typename A8, typename C16>
inline C16 mfma_32x32x16(const A8& a, const A8& b, C16 c) {
  static_assert(sizeof(A8) == 16, "8 bf16");
  Fiber* f = cur();
  Wave& w = f->blk->waves[f->wave];
  std::memcpy(w.slot[f->lane], &a, 16);
  std::memcpy(w.slot[f->lane] + 16, &b, 16);
  wave_sync();
  const int li = f->lane & 31, lh = f->lane >> 5;
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
    float s = 0.f;
    for (int kg = 0; kg < 2; ++kg) {
      unsigned short av[8], bv[8];
      std::memcpy(av, w.slot[row + 32 * kg], 16);
      std::memcpy(bv, w.slot[li + 32 * kg] + 16, 16);
      for (int e = 0; e < 8; ++e) s += bf16_to_float(av[e]) * bf16_to_float(bv[e]);
    }
    c[r] += s;
  }
  wave_sync();
  return c;
}
// v_mfma_f32_32x32x2_f32: lane = li + 32*lh supplies A[row li][k = lh] and B[k = lh][column li]; D as above
template <typename C16>
inline C16 mfma_32x32x2_f32(float a, float b, C16 c) {
  Fiber* f = cur();
  Wave& w = f->blk->waves[f->wave];
  std::memcpy(w.slot[f->lane], &a, 4);
  std::memcpy(w.slot[f->lane] + 4, &b, 4);
  wave_sync();
  const int li = f->lane & 31, lh = f->lane >> 5;
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
    float s = 0.f;
    for (int k = 0; k < 2; ++k) {
      float av, bv;
      std::memcpy(&av, w.slot[row + 32 * k], 4);
      std::memcpy(&bv, w.slot[li + 32 * k] + 4, 4);
      s += av * bv;
    }
    c[r] += s;
  }
  wave_sync();
  return c;
}
// v_mfma_f32_16x16x4_f32: lane = li + 16*lh (lh = k 0..3); D register r of lane (li, lh) = row 4*lh + r, column li
template <typename C4>
inline C4 mfma_16x16x4_f32(float a, float b, C4 c) {
  Fiber* f = cur();
  Wave& w = f->blk->waves[f->wave];
  std::memcpy(w.slot[f->lane], &a, 4);
  std::memcpy(w.slot[f->lane] + 4, &b, 4);
  wave_sync();
  const int li = f->lane & 15, lh = f->lane >> 4;
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * lh + r;
    float s = 0.f;
    for (int k = 0; k < 4; ++k) {
      float av, bv;
      std::memcpy(&av, w.slot[row + 16 * k], 4);
      std::memcpy(&bv, w.slot[li + 16 * k] + 4, 4);
      s += av * bv;
    }
    c[r] += s;
  }
  wave_sync();
  return c;
}
// v_mfma_f32_16x16x32_bf16: lane = li + 16*lh (lh 0..3) holds row / column li, k-slots 8*lh .. 8*lh + 7; D register r = row 4*lh + r, column li
template <typename A8, typename C4>
inline C4 mfma_16x16x32(const A8& a, const A8& b, C4 c) {
  static_assert(sizeof(A8) == 16, "8 bf16");
  Fiber* f = cur();
  Wave& w = f->blk->waves[f->wave];
  std::memcpy(w.slot[f->lane], &a, 16);
  std::memcpy(w.slot[f->lane] + 16, &b, 16);
  wave_sync();
  const int li = f->lane & 15, lh = f->lane >> 4;
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * lh + r;
    float s = 0.f;
    for (int kg = 0; kg < 4; ++kg) {
      unsigned short av[8], bv[8];
      std::memcpy(av, w.slot[row + 16 * kg], 16);
      std::memcpy(bv, w.slot[li + 16 * kg] + 16, 16);
      for (int e = 0; e < 8; ++e) s += bf16_to_float(av[e]) * bf16_to_float(bv[e]);
    }
    c[r] += s;
  }
  wave_sync();
  return c;
}
template <typename G, typename L>
inline void global_load_lds(G g, L l, int size) {   // each lane: `size` bytes from its own address to the wave-uniform LDS base + lane*size
  std::memcpy((char*)(uintptr_t)l + (size_t)cur()->lane * size, (const void*)(uintptr_t)g, (size_t)size);
}
inline int lane_id() { return cur()->lane; }
template <typename T>
inline T shfl_xor(T v, int mask) { return publish_and_read(v, (cur()->lane ^ mask) & 63); }
inline int syncthreads_or(int pred) {
  Block* b = cur()->blk;
  static int acc[2];
  const unsigned g = b->barrier_gen & 1;
  if (pred) acc[g] = 1;
  block_barrier();
  const int r = acc[g];
  block_barrier();
  acc[g] = 0;   // every fiber clears it after both barriers: nobody reads this generation's slot again before the next use two barriers on
  return r;
}

}  // namespace emu

#define threadIdx (emu::cur()->tid)
#define blockIdx (emu::cur()->blk->bid)
#define blockDim (emu::cur()->blk->bdim)
#define gridDim (emu::cur()->blk->gdim)
#define hipLaunchKernelGGL(kern, grid, block, lds, strm, ...) emu::launch(grid, block, lds, [&] { kern(__VA_ARGS__); })
#define __syncthreads() emu::block_barrier()
#define __builtin_amdgcn_s_barrier() emu::block_barrier()
#define __builtin_amdgcn_readfirstlane(v) emu::readfirstlane((int)(v))
#define __builtin_amdgcn_readlane(v, l) emu::readlane((int)(v), (int)(l))
#define __builtin_amdgcn_ballot_w64(p) emu::ballot(p)
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(a, b, c) ((void)0)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_s_memtime() 0ull
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) emu::mfma_32x32x16(a, b, c)
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z) emu::mfma_16x16x32(a, b, c)
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) emu::mfma_32x32x2_f32(a, b, c)
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) emu::mfma_16x16x4_f32(a, b, c)
#define __builtin_amdgcn_s_getreg(x) 0u
#define __builtin_amdgcn_global_load_lds(g, l, size, off, aux) emu::global_load_lds(g, l, size)

template <typename T>
inline T __shfl_xor(T v, int mask, int width = 64) { return emu::shfl_xor(v, mask); }   // (width 64 only)
#define __syncthreads_or(p) emu::syncthreads_or(p)
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline float atomicAdd(float* p, float v) { const float o = *p; *p = o + v; return o; }   // one OS thread: fibers never interleave inside this
inline int atomicAdd(int* p, int v) { const int o = *p; *p = o + v; return o; }
inline unsigned atomicAdd(unsigned* p, unsigned v) { const unsigned o = *p; *p = o + v; return o; }
inline float __log2f(float x) { return std::log2(x); }
inline float __powf(float x, float y) { return std::pow(x, y); }
inline float __expf(float x) { return std::exp(x); }
inline float __uint_as_float(unsigned u) { float f; std::memcpy(&f, &u, 4); return f; }
inline unsigned __float_as_uint(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }
// (fabsf / copysignf: the C library's, via <cmath>)
inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }
inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
inline size_t min(size_t a, size_t b) { return a < b ? a : b; }
inline size_t max(size_t a, size_t b) { return a > b ? a : b; }
inline unsigned long long min(unsigned long long a, unsigned long long b) { return a < b ? a : b; }
inline unsigned long long max(unsigned long long a, unsigned long long b) { return a > b ? a : b; }
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
inline hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n ? n : 1); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemset(void* d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memcpy(d, s, n); return hipSuccess; }
