// Where do the cycles of the bf16-split consumer loop go?  One wave per SIMD (256-thread blocks, one per CU), wave tile 64 x 128,
// operands resident in LDS; per 16-deep chunk: 24 LDS reads, 3 x 8 + ... splits (~264 VALU) and 48 v_mfma_f32_32x32x16_bf16.
// MODE 0: MFMAs only (operands split once);  1: splits only (results folded into a checksum);  2: both, compiler order;
// 3: both, splits of column u+1 pinned into the MFMA shadows of column u.  Prints shader cycles per chunk (s_memtime) and the
// shader clock during the run (cycles / wall time).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/split_probe tools/split_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int BK = 16, ROWS = 128, BROW = 256;

struct Split8 { u32x4 h, m, l; };
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
  f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
template <int SUBMODE>
__device__ __forceinline__ float sub(float a, float b) {
  if constexpr (SUBMODE == 1) {
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
  } else {
    return a - b;
  }
}
template <int SUBMODE>
__device__ __forceinline__ void split8(const float (&x)[8], Split8& s) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float x0 = x[2 * q], x1 = x[2 * q + 1];
    const unsigned H = pk_bf16(x0, x1);
    const float r0 = sub<SUBMODE>(x0, __uint_as_float(H << 16)), r1 = sub<SUBMODE>(x1, __uint_as_float(H & 0xffff0000u));
    const unsigned M = pk_bf16(r0, r1);
    const float s0 = sub<SUBMODE>(r0, __uint_as_float(M << 16)), s1 = sub<SUBMODE>(r1, __uint_as_float(M & 0xffff0000u));
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

template <int MODE, int SUBMODE>
__global__ void __launch_bounds__(256, 1) probe(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ out,
                                               unsigned long long* __restrict__ ticks, int chunks) {
  extern __shared__ float lds[];
  float* sA = lds;
  float* sB = lds + BK * ROWS;
  for (int e = threadIdx.x; e < BK * ROWS; e += 256) sA[e] = A[e];
  for (int e = threadIdx.x; e < BK * BROW; e += 256) sB[e] = B[e];
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wr = w >> 1, wc = w & 1;
  const int li = lane & 31, lh = lane >> 5;
  f32x16 acc[2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;
  unsigned chk = 0;
  const float* ar = sA + wr * 64 + li;
  const float* bs = sB + wc * 128 + 4 * li;
  Split8 ca[2], cb[4];
  {
    float x[8];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = ar[(2 * j + lh) * ROWS + t * 32];
      split8<SUBMODE>(x, ca[t]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = bs[(2 * j + lh) * BROW + u];
      split8<SUBMODE>(x, cb[u]);
    }
  }
  const unsigned long long t0 = __builtin_readcyclecounter();
  const unsigned long long w0 = wall_clock64();
  for (int c = 0; c < chunks; ++c) {
    if constexpr (MODE == 0) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t][u] = mac6(ca[t], cb[u], acc[t][u]);
    } else {
      float ra[2][8];
      f32x4 rb[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int t = 0; t < 2; ++t) ra[t][j] = ar[(2 * j + lh) * ROWS + t * 32 + (c & 1)];
        rb[j] = *reinterpret_cast<const f32x4*>(bs + (2 * j + lh) * BROW + 4 * (c & 1));
      }
      Split8 fa[2], fb[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) split8<SUBMODE>(ra[t], fa[t]);
      auto col = [&](int u, Split8& o) __attribute__((always_inline)) {
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = rb[j][u];
        split8<SUBMODE>(x, o);
      };
      if constexpr (MODE == 1) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          col(u, fb[0]);
          chk ^= fb[0].h[0] ^ fb[0].m[1] ^ fb[0].l[2] ^ fb[0].h[3] ^ fb[0].m[0] ^ fb[0].l[1] ^ fb[0].h[2] ^ fb[0].m[3] ^ fb[0].l[0] ^ fb[0].h[1] ^ fb[0].m[2] ^ fb[0].l[3];
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
          chk ^= fa[t].h[0] ^ fa[t].m[1] ^ fa[t].l[2] ^ fa[t].h[3] ^ fa[t].m[0] ^ fa[t].l[1] ^ fa[t].h[2] ^ fa[t].m[3] ^ fa[t].l[0] ^ fa[t].h[1] ^ fa[t].m[2] ^ fa[t].l[3];
      } else {
        col(0, fb[0]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if constexpr (MODE == 3) __builtin_amdgcn_sched_barrier(0);
          if (u + 1 < 4) col(u + 1, fb[(u + 1) & 1]);
#pragma unroll
          for (int t = 0; t < 2; ++t) acc[t][u] = mac6(fa[t], fb[u & 1], acc[t][u]);
          if constexpr (MODE == 3) {
            if (u + 1 < 4) {
#pragma unroll
              for (int i = 0; i < 12; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
              }
            }
          }
        }
        if constexpr (MODE == 3) __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  const unsigned long long w1 = wall_clock64();
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[t][u][r];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)chk;
  if (threadIdx.x == 0) {
    ticks[2 * blockIdx.x] = t1 - t0;
    ticks[2 * blockIdx.x + 1] = w1 - w0;
  }
}

template <int MODE, int SUBMODE>
static void run(const char* what, const float* dA, const float* dB, float* dO, unsigned long long* dT, int chunks) {
  const size_t lds = sizeof(float) * BK * (ROWS + BROW) + 90 * 1024;   // one block per CU
  hipFuncSetAttribute((const void*)probe<MODE, SUBMODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  probe<MODE, SUBMODE><<<256, 256, lds>>>(dA, dB, dO, dT, chunks);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<MODE, SUBMODE><<<256, 256, lds>>>(dA, dB, dO, dT, chunks);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> t(512);
  hipMemcpy(t.data(), dT, 512 * 8, hipMemcpyDeviceToHost);
  double cyc = 0, wall = 0;
  for (int b = 0; b < 256; ++b) { cyc += t[2 * b]; wall += t[2 * b + 1]; }
  cyc /= 256; wall /= 256;
  const double ghz = cyc / (wall * 10.0);   // wall_clock64 ticks at 100 MHz
  const double tf = 2.0 * 128 * 256 * 16 * (double)chunks * 256 / ms * 1e-9;
  printf("%-46s %8.1f cycles/chunk  %6.3f GHz  %7.3f ms  %6.1f TFLOP/s-eq\n", what, cyc / chunks, ghz, ms, MODE == 1 ? 0.0 : tf);
}

int main() {
  std::vector<float> A(BK * ROWS + 64), B(BK * BROW + 64);
  srand(3);
  for (auto& v : A) v = rand() / (float)RAND_MAX - 0.5f;
  for (auto& v : B) v = rand() / (float)RAND_MAX - 0.5f;
  float *dA, *dB, *dO;
  unsigned long long* dT;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dO, 256 * 256 * 4); hipMalloc(&dT, 512 * 8);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  const int chunks = 4000;
  run<0, 0>("48 MFMAs only", dA, dB, dO, dT, chunks);
  run<1, 0>("reads + splits only (a - b)", dA, dB, dO, dT, chunks);
  run<1, 1>("reads + splits only (asm v_sub_f32)", dA, dB, dO, dT, chunks);
  run<2, 0>("both, compiler order", dA, dB, dO, dT, chunks);
  run<2, 1>("both, compiler order, asm sub", dA, dB, dO, dT, chunks);
  run<3, 0>("both, next column's split pinned under MFMAs", dA, dB, dO, dT, chunks);
  run<3, 1>("both, pinned, asm sub", dA, dB, dO, dT, chunks);
  return 0;
}
