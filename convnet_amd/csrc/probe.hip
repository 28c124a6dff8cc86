// convnet_hip_probe_matrix_pipe: what the chip sustains on the instruction the default GEMM kernels execute, with nothing else in the
// way — measured, for the roofline report (bench.py: roofline.power_ceiling; tools/power_ceiling.py).
//
// The bf16-split kernels (gather_gemm.h: split_mac) issue six v_mfma_f32_32x32x16_bf16 per 32 x 32 x 16 block of fp32 products; the
// instruction's nominal rate is one per 32 cycles per SIMD at 2.4 GHz = 2.5 PFLOP/s dense (MI355X_MICROARCH.md).  The chip clocks to its
// POWER budget, though (same guide, "DVFS give-back": an attention kernel ran 1.90-1.95 GHz on random data and 2.30 GHz on zeros), so the
// rate a kernel of these instructions can reach on real operands is a property of the part, not of the kernel.  This probe pins it:
// one wave per SIMD (256-thread blocks that own their CU, as gpw_kernel / gpv_kernel / wgw_kernel do), sixteen 32 x 32 accumulators
// (the 128 x 128 wave tile), the six products in split_mac's order over four A and four B fragments held in registers — NO memory
// traffic, NO split arithmetic, NO barriers: every issue slot an MFMA, back to back on independent accumulators.  Operands are the
// h / m / l planes of N(0,1) values (what a training step feeds the pipe) or zeros.  Reported: executed bf16 TFLOP/s, the same in
// algorithmic fp32 units (/ 6), and the effective clock two ways — shader cycles (s_memtime) over wall time (100 MHz counter) inside
// the kernel, and MFMA issue rate x 32 cycles.
#include <vector>

#include "gather_gemm.h"

namespace chip {

__global__ __launch_bounds__(256, 1) void matrix_pipe_probe_kernel(const u32x4* __restrict__ frag, float* __restrict__ sink, unsigned long long* __restrict__ ticks, int iters) {
  extern __shared__ float probe_lds[];   // sized by the launcher so that a block owns its CU
  const int lane = threadIdx.x & 63;
  Split8 a[4], b[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    a[t].h = frag[(0 * 8 + t) * 64 + lane];
    a[t].m = frag[(1 * 8 + t) * 64 + lane];
    a[t].l = frag[(2 * 8 + t) * 64 + lane];
    b[t].h = frag[(0 * 8 + 4 + t) * 64 + lane];
    b[t].m = frag[(1 * 8 + 4 + t) * 64 + lane];
    b[t].l = frag[(2 * 8 + 4 + t) * 64 + lane];
  }
  f32x16 acc[4][4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][u][e] = 0.f;
  const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      // one product of the six over the four row tiles, then the next: four independent accumulators between two MFMAs on the same one
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t][u] = mma_bf16(a[t].m, b[u].m, acc[t][u]);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t][u] = mma_bf16(a[t].h, b[u].l, acc[t][u]);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t][u] = mma_bf16(a[t].l, b[u].h, acc[t][u]);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t][u] = mma_bf16(a[t].h, b[u].m, acc[t][u]);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t][u] = mma_bf16(a[t].m, b[u].h, acc[t][u]);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t][u] = mma_bf16(a[t].h, b[u].h, acc[t][u]);
    }
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < 16; ++e) s += acc[t][u][e];
  sink[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) {
    ticks[2 * blockIdx.x] = c1 - c0;
    ticks[2 * blockIdx.x + 1] = w1 - w0;
  }
}

}  // namespace chip

extern "C" int convnet_hip_probe_matrix_pipe(int random_operands, double seconds, double* out4) {
  using namespace chip;
  if (!out4 || seconds <= 0) return ERROR_GENERIC;
  // operand fragments: 3 planes x (4 A + 4 B fragments) x 64 lanes x 8 bf16 — the exact three-way split of N(0,1) values, or zeros
  std::vector<unsigned> host(3 * 8 * 64 * 4, 0u);
  if (random_operands) {
    unsigned long long st = 0x9E3779B97F4A7C15ull;
    auto uni = [&]() {   // xorshift64*: a fixed sequence, no libc state
      st ^= st >> 12; st ^= st << 25; st ^= st >> 27;
      return (double)((st * 0x2545F4914F6CDD1Dull) >> 11) / 9007199254740992.0;
    };
    auto bf = [](float x) {   // round to nearest even bf16, returned widened back
      unsigned u;
      memcpy(&u, &x, 4);
      u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
      float r;
      memcpy(&r, &u, 4);
      return r;
    };
    auto bits = [](float x) {
      unsigned u;
      memcpy(&u, &x, 4);
      return u >> 16;
    };
    for (int f = 0; f < 8; ++f)
      for (int l = 0; l < 64; ++l)
        for (int q = 0; q < 4; ++q) {
          unsigned w[3] = {0, 0, 0};
          for (int hf = 0; hf < 2; ++hf) {
            const double u1 = uni() + 1e-12, u2 = uni();
            const float x = (float)(std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2));
            const float h = bf(x), m = bf(x - h), lo = bf(x - h - m);
            w[0] |= bits(h) << (16 * hf);
            w[1] |= bits(m) << (16 * hf);
            w[2] |= bits(lo) << (16 * hf);
          }
          for (int pl = 0; pl < 3; ++pl) host[((pl * 8 + f) * 64 + l) * 4 + q] = w[pl];
        }
  }
  u32x4* frag = nullptr;
  float* sink = nullptr;
  unsigned long long* ticks = nullptr;
  int cus = 256;
  hipDeviceProp_t prop;
  int dev = 0;
  CHIP_CHECK(hipGetDevice(&dev));
  if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
  CHIP_CHECK(hipMalloc((void**)&frag, host.size() * 4));
  CHIP_CHECK(hipMalloc((void**)&sink, (size_t)cus * 256 * 4));
  CHIP_CHECK(hipMalloc((void**)&ticks, (size_t)cus * 16));
  CHIP_CHECK(hipMemcpy(frag, host.data(), host.size() * 4, hipMemcpyHostToDevice));
  const size_t lds = 100 * 1024;   // more than half a CU's LDS: one block per CU, one wave per SIMD
  CHIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(matrix_pipe_probe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipStream_t st = stream();
  hipEvent_t e0, e1;
  CHIP_CHECK(hipEventCreate(&e0));
  CHIP_CHECK(hipEventCreate(&e1));
  // calibrate the iteration count on a short run, then one launch of about `seconds`
  int iters = 2000;
  float ms = 0.f;
  for (int pass = 0; pass < 2; ++pass) {
    CHIP_CHECK(hipEventRecord(e0, st));
    hipLaunchKernelGGL(matrix_pipe_probe_kernel, dim3(cus), dim3(256), lds, st, frag, sink, ticks, iters);
    CHIP_CHECK(hipEventRecord(e1, st));
    CHIP_CHECK(hipEventSynchronize(e1));
    CHIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (pass == 0) {
      const double want = seconds * 1e3 / (ms > 0.f ? ms : 1.f) * iters;
      iters = want > 2.0e8 ? 200000000 : want < 100 ? 100 : (int)want;
    }
  }
  std::vector<unsigned long long> t(2 * (size_t)cus);
  CHIP_CHECK(hipMemcpy(t.data(), ticks, t.size() * 8, hipMemcpyDeviceToHost));
  double cyc = 0, wall = 0;
  for (int b = 0; b < cus; ++b) { cyc += (double)t[2 * b]; wall += (double)t[2 * b + 1]; }
  const double mfmas = 96.0 * iters;                                        // per wave
  const double flops = mfmas * 2.0 * 32 * 32 * 16 * 4.0 * cus;              // 4 waves per CU
  out4[0] = flops / (ms * 1e-3) / 1e12;                                     // executed bf16 TFLOP/s
  out4[1] = out4[0] / 6.0;                                                  // the same in algorithmic fp32 TFLOP/s of the split kernels
  out4[2] = wall > 0 ? cyc / (wall * 10.0) : 0.0;                           // GHz: shader cycles over the 100 MHz wall counter
  out4[3] = mfmas * 32.0 / (ms * 1e-3) / 1e9;                               // GHz the issue rate implies at 32 cycles per MFMA per SIMD
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  hipFree(frag);
  hipFree(sink);
  hipFree(ticks);
  return launch_status();
}
