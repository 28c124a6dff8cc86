// How fast can ONE wave issue LDS-DMA loads (global_load_lds_dwordx4, 1 KiB per wave-instruction), and what serialises them?
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/dma_issue tools/dma_issue.hip ; run on the GPU box.
// Variants: M0 changes on every instruction (what gg_kernel / wg_kernel did), M0 constant with the instruction's immediate
// offset selecting the LDS slot, plain global_load_dwordx4 into VGPRs for comparison; 1 wave per CU and 8 waves per CU; source
// either L2-resident (64 KiB footprint) or streaming from HBM.  Reports cycles (s_memtime) per instruction to ISSUE the batch and
// until the batch has LANDED (vmcnt(0)).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE, int NLOAD>
__global__ void __launch_bounds__(256) probe(const float* __restrict__ src, size_t stride_floats, unsigned long long* out, int reps, float* sink) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* lds = smem + wave * (NLOAD * 256);
  const float* g = src + ((size_t)blockIdx.x * (blockDim.x >> 6) + wave) * stride_floats + 4 * lane;
  unsigned long long t_issue = 0, t_land = 0;
  f4 acc = {0, 0, 0, 0};
  for (int r = 0; r < reps; ++r) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE == 0) {        // M0 changes every instruction
#pragma unroll
      for (int i = 0; i < NLOAD; ++i) __builtin_amdgcn_global_load_lds((gbl_ptr_t)(g + 256 * i), (lds_ptr_t)(lds + 256 * i), 16, 0, 0);
    } else if (MODE == 1) { // M0 constant per group of 4: immediate offset 0/1024/2048/3072 picks the slot (applied to both addresses)
#pragma unroll
      for (int i = 0; i < NLOAD; i += 4) {
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(g + 256 * i), (lds_ptr_t)(lds + 256 * i), 16, 0, 0);
        if (i + 1 < NLOAD) __builtin_amdgcn_global_load_lds((gbl_ptr_t)(g + 256 * i), (lds_ptr_t)(lds + 256 * i), 16, 1024, 0);
        if (i + 2 < NLOAD) __builtin_amdgcn_global_load_lds((gbl_ptr_t)(g + 256 * i), (lds_ptr_t)(lds + 256 * i), 16, 2048, 0);
        if (i + 3 < NLOAD) __builtin_amdgcn_global_load_lds((gbl_ptr_t)(g + 256 * i), (lds_ptr_t)(lds + 256 * i), 16, 3072, 0);
      }
    } else {                // plain loads into registers
#pragma unroll
      for (int i = 0; i < NLOAD; ++i) acc += *reinterpret_cast<const f4*>(g + 256 * i);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    if (MODE == 2) asm volatile("" ::"v"(acc));
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    t_issue += t1 - t0;
    t_land += t2 - t0;
    g += NLOAD * 256;   // next batch (footprint wraps on the host side via stride/reps)
  }
  if (lane == 0) {
    out[2 * (blockIdx.x * (blockDim.x >> 6) + wave)] = t_issue;
    out[2 * (blockIdx.x * (blockDim.x >> 6) + wave) + 1] = t_land;
  }
  if (acc[0] == 123.456f) sink[0] = acc[0] + smem[threadIdx.x];
}

template <int MODE, int NLOAD>
static void run(const char* name, int waves_per_block, int blocks, bool hot, const float* buf, size_t buf_floats) {
  const int reps = 64;
  const size_t per_wave = (size_t)NLOAD * 256 * (hot ? 1 : reps);
  const size_t stride = hot ? 0 : per_wave;   // hot: every wave re-reads the same 24 KiB
  if (!hot && (size_t)blocks * waves_per_block * per_wave > buf_floats) { printf("%s: buffer too small\n", name); return; }
  unsigned long long* out;
  float* sink;
  hipMalloc((void**)&out, (size_t)blocks * waves_per_block * 16);
  hipMalloc((void**)&sink, 64);
  const size_t lds = (size_t)waves_per_block * NLOAD * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE, NLOAD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int it = 0; it < 2; ++it) probe<MODE, NLOAD><<<blocks, waves_per_block * 64, lds>>>(hot ? buf : buf, hot ? 0 : stride, out, hot ? reps : reps, sink);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h((size_t)blocks * waves_per_block * 2);
  hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
  double si = 0, sl = 0;
  for (size_t i = 0; i < h.size(); i += 2) { si += h[i]; sl += h[i + 1]; }
  const double n = (double)(h.size() / 2) * reps * NLOAD;
  printf("%-34s %d wave(s)/CU x %d CUs, %s: issue %.0f cyc/instr, landed %.0f cyc/instr (batch of %d: %.0f / %.0f)\n", name, waves_per_block, blocks,
         hot ? "L2-hot" : "HBM stream", si / n, sl / n, NLOAD, si / n * NLOAD, sl / n * NLOAD);
  hipFree(out);
  hipFree(sink);
}

int main() {
  const size_t buf_floats = (size_t)1 << 30;   // 4 GiB
  float* buf;
  if (hipMalloc((void**)&buf, buf_floats * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(buf, 0, buf_floats * 4);
  for (int hot = 1; hot >= 0; --hot) {
    run<0, 24>("LDS-DMA, M0 per instruction", 1, 256, hot, buf, buf_floats);
    run<1, 24>("LDS-DMA, M0 per 4 (imm offset)", 1, 256, hot, buf, buf_floats);
    run<2, 24>("global_load_dwordx4 -> VGPR", 1, 256, hot, buf, buf_floats);
    run<0, 6>("LDS-DMA x6, M0 per instruction", 4, 256, hot, buf, buf_floats);
    run<1, 6>("LDS-DMA x6, M0 per 4 (imm offset)", 4, 256, hot, buf, buf_floats);
    run<2, 6>("global_load_dwordx4 x6 -> VGPR", 4, 256, hot, buf, buf_floats);
  }
  return 0;
}
