// Sustained fp32-MFMA ceiling probe for gfx950: register-only v_mfma_f32_32x32x2_f32 chains, no memory traffic.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/mfma_peak tools/mfma_peak.hip ; run on the GPU box.
// Prints TFLOP/s for 1, 2 and 4 waves per SIMD; DESIGN.md quotes this as the practical ceiling beside the 157.3 paper peak.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ void __launch_bounds__(256) mfma_loop(float* out, int iters, float a, float b) {
  f32x16 acc[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c)
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[c][i];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int CHAINS>
static void run(int wg_per_cu, int iters) {
  float* out;
  hipMalloc(&out, 4096);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int grid = 256 * wg_per_cu;
  mfma_loop<CHAINS><<<grid, 256>>>(out, iters, 1.f, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  mfma_loop<CHAINS><<<grid, 256>>>(out, iters, 1.f, 1.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = 2.0 * 32 * 32 * 2 * CHAINS * (double)iters * 4 * grid;
  printf("chains=%d waves/SIMD=%d iters=%d  %.3f ms  %.1f TFLOP/s\n", CHAINS, wg_per_cu, iters, ms, flops / ms * 1e-9);
  hipFree(out);
}

int main() {
  run<8>(1, 20000);
  run<8>(2, 20000);
  run<4>(4, 20000);
  run<8>(2, 200000);
  return 0;
}
