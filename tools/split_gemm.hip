// fp32 GEMM on the bf16 matrix pipe by exact three-way operand splitting (x = h + m + l, each a bf16; six of the nine
// cross products kept: hh, hm, mh, hl, lh, mm; the dropped ones are <= 2^-23 of the product with round-to-nearest splits).
// Feasibility probe for gfx950: (1) numerics of one 128 x 512 tile against a double-precision CPU result, beside the same
// tile on v_mfma_f32_32x32x2_f32; (2) the rate of the consumer loop alone (operands resident in LDS, no staging), which
// bounds what a producer-wave kernel built on it could reach.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/split_gemm tools/split_gemm.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 16;       // one bf16 MFMA deep
constexpr int ROWS = 128;    // block tile rows (A operand: [k][ROWS] in LDS)
constexpr int COLS = 512;    // block tile columns (B operand: [k][COLS]); 4 waves x 128

struct Split8 {
  u32x4 h, m, l;             // 8 bf16 each (the MFMA's A/B operand registers)
};

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
  f32x2 v = {a, b};
  bf16x2 r = __builtin_convertvector(v, bf16x2);
  return __builtin_bit_cast(unsigned, r);
}

// exact: x = h + m + l with h = rne8(x), m = rne8(x - h), l = x - h - m (fits 8 bits)
__device__ __forceinline__ void split8(const float (&x)[8], Split8& s) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float x0 = x[2 * p], x1 = x[2 * p + 1];
    const unsigned H = pk_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(H << 16), r1 = x1 - __uint_as_float(H & 0xffff0000u);
    const unsigned M = pk_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(M << 16), s1 = r1 - __uint_as_float(M & 0xffff0000u);
    s.h[p] = H;
    s.m[p] = M;
    s.l[p] = pk_bf16(s0, s1);
  }
}

__device__ __forceinline__ f32x16 mma(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// One block = 4 waves side by side, wave tile (MTW*32) x 128 ... MTW = 4: 128 x 128 per wave (256 accumulator registers, one
// wave per SIMD); MTW = 2 uses the upper 64 rows only (to compare a two-waves-per-SIMD shape).
// resident = 1: the chunk loop re-reads LDS stage 0 `chunks` times (rate probe); 0: stages A/B from global per chunk (numerics).
template <int MTW, bool SPLIT>
__global__ void __launch_bounds__(256, 1) tile_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ D,
                                                      int chunks, int resident) {
  extern __shared__ float lds[];
  float* sA = lds;                 // [BK][ROWS]
  float* sB = lds + BK * ROWS;     // [BK][COLS]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = lane & 31, kg = lane >> 5;
  f32x16 acc[MTW][4];
#pragma unroll
  for (int a = 0; a < MTW; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  for (int c = 0; c < chunks; ++c) {
    if (!resident || c == 0) {
      __syncthreads();
      const float* gA = A + (size_t)(resident ? 0 : c) * BK * ROWS;
      const float* gB = B + (size_t)(resident ? 0 : c) * BK * COLS;
      for (int e = threadIdx.x; e < BK * ROWS; e += 256) sA[e] = gA[e];
      for (int e = threadIdx.x; e < BK * COLS; e += 256) sB[e] = gB[e];
      __syncthreads();
    }
    if constexpr (SPLIT) {
      Split8 fa[MTW];
#pragma unroll
      for (int a = 0; a < MTW; ++a) {
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = sA[(8 * kg + j) * ROWS + a * 32 + i];
        split8(x, fa[a]);
      }
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = sB[(8 * kg + j) * COLS + w * 128 + b * 32 + i];
        Split8 fb;
        split8(x, fb);
#pragma unroll
        for (int a = 0; a < MTW; ++a) {
          f32x16 t = acc[a][b];
          t = mma(fa[a].m, fb.m, t);      // small terms first
          t = mma(fa[a].h, fb.l, t);
          t = mma(fa[a].l, fb.h, t);
          t = mma(fa[a].h, fb.m, t);
          t = mma(fa[a].m, fb.h, t);
          t = mma(fa[a].h, fb.h, t);
          acc[a][b] = t;
        }
      }
    } else {
      // the fp32 matrix instruction on the same tile: lane (i, kg) supplies k = 2*s + kg of k-step s
#pragma unroll
      for (int s = 0; s < BK / 2; ++s) {
        float fa[MTW], fb[4];
#pragma unroll
        for (int a = 0; a < MTW; ++a) fa[a] = sA[(2 * s + kg) * ROWS + a * 32 + i];
#pragma unroll
        for (int b = 0; b < 4; ++b) fb[b] = sB[(2 * s + kg) * COLS + w * 128 + b * 32 + i];
#pragma unroll
        for (int a = 0; a < MTW; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a], fb[b], acc[a][b], 0, 0, 0);
      }
    }
  }
  // D[row][col], row-major 128 x 512 per block; C/D map: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  float* Dt = D + (size_t)blockIdx.x * ROWS * COLS;
#pragma unroll
  for (int a = 0; a < MTW; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg, col = w * 128 + b * 32 + i;
        Dt[(size_t)row * COLS + col] = acc[a][b][r];
      }
}

template <int MTW, bool SPLIT>
static double rate(const float* dA, const float* dB, float* dD, int chunks, int blocks_per_cu) {
  const size_t lds = sizeof(float) * BK * (ROWS + COLS) + (blocks_per_cu == 1 ? 81920 : 0);   // pad: force one block per CU
  hipFuncSetAttribute((const void*)tile_kernel<MTW, SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  tile_kernel<MTW, SPLIT><<<grid, 256, lds>>>(dA, dB, dD, chunks, 1);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  tile_kernel<MTW, SPLIT><<<grid, 256, lds>>>(dA, dB, dD, chunks, 1);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = 2.0 * (MTW * 32) * COLS * BK * (double)chunks * grid;
  const double tf = flops / ms * 1e-9;
  printf("%-6s wave tile %3d x 128, %d block(s)/CU, %d chunks: %8.3f ms  %7.1f TFLOP/s (fp32-equivalent)\n", SPLIT ? "split" : "fp32",
         MTW * 32, blocks_per_cu, chunks, ms, tf);
  return tf;
}

int main() {
  const int K = 3456, chunks = K / BK;   // conv4's reduction length
  std::vector<float> A((size_t)K * ROWS), B((size_t)K * COLS);
  srand(7);
  auto rnd = [] { return (float)((rand() / (double)RAND_MAX) * 2.0 - 1.0); };
  for (auto& v : A) v = rnd() * 0.05f;
  for (auto& v : B) v = rnd();
  float *dA, *dB, *dD;
  hipMalloc(&dA, A.size() * 4);
  hipMalloc(&dB, B.size() * 4);
  hipMalloc(&dD, sizeof(float) * ROWS * COLS * 512);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);

  // ---- numerics: one tile, the whole K, against double; the fp32 instruction's error beside it
  std::vector<double> ref((size_t)ROWS * COLS, 0.0);
  std::vector<double> mag((size_t)ROWS * COLS, 0.0);
  for (int k = 0; k < K; ++k)
    for (int r = 0; r < ROWS; ++r) {
      const double a = A[(size_t)k * ROWS + r];
      for (int c = 0; c < COLS; ++c) {
        const double p = a * B[(size_t)k * COLS + c];
        ref[(size_t)r * COLS + c] += p;
        mag[(size_t)r * COLS + c] += std::fabs(p);
      }
    }
  std::vector<float> out((size_t)ROWS * COLS);
  const size_t lds = sizeof(float) * BK * (ROWS + COLS);
  for (int split = 0; split < 2; ++split) {
    if (split) {
      hipFuncSetAttribute((const void*)tile_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      tile_kernel<4, true><<<1, 256, lds>>>(dA, dB, dD, chunks, 0);
    } else {
      hipFuncSetAttribute((const void*)tile_kernel<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      tile_kernel<4, false><<<1, 256, lds>>>(dA, dB, dD, chunks, 0);
    }
    hipDeviceSynchronize();
    hipMemcpy(out.data(), dD, out.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0, sum2 = 0, worst_abs = 0;
    for (size_t e = 0; e < out.size(); ++e) {
      const double err = std::fabs(out[e] - ref[e]);
      const double rel = err / mag[e];        // relative to sum |a b|: the scale fp32 summation error is bounded against
      worst = rel > worst ? rel : worst;
      worst_abs = err > worst_abs ? err : worst_abs;
      sum2 += rel * rel;
    }
    printf("%-6s K=%d: max |err| / sum|ab| = %.3e (= %.2f x 2^-24), rms %.3e, max |err| %.3e\n", split ? "split" : "fp32", K, worst,
           worst / 5.9604645e-8, std::sqrt(sum2 / out.size()), worst_abs);
  }

  // ---- rate of the consumer loop alone
  rate<4, false>(dA, dB, dD, 2000, 1);
  rate<2, false>(dA, dB, dD, 2000, 2);
  rate<4, true>(dA, dB, dD, 2000, 1);
  rate<2, true>(dA, dB, dD, 2000, 1);
  rate<2, true>(dA, dB, dD, 2000, 2);
  return 0;
}
