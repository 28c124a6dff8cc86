// Where do the two co-resident 256-thread blocks of a CU land, and which wave-buffer slots do their waves get?
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/hwid_census tools/hwid_census.hip ; run on the GPU box.
// Each wave records HW_REG_HW_ID (wave slot [3:0], SIMD [5:4], CU [11:8], SH [12], SE [15:13]) and HW_REG_XCC_ID; the host
// prints (a) whether the four waves of a block share one slot number, (b) whether the two blocks of a CU differ in slot
// parity, (c) which block indices share a CU.  Used to design gg_kernel's start stagger.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>

__global__ void __launch_bounds__(256, 2) census(unsigned* out, int spin) {
  extern __shared__ float smem[];
  const unsigned hw = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 0xf;          // WAVE_ID
  const unsigned hw_all = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);           // low 16 bits of HW_ID
  const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xf;         // XCC_ID
  const unsigned long long t0 = __builtin_readcyclecounter();
  // stay resident long enough that every slot of the chip is occupied at once
  float s = 0.f;
  for (int i = 0; i < spin; ++i) { __builtin_amdgcn_s_sleep(8); s += 1.f; }
  if (s < 0) smem[threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) {
    unsigned* o = out + 4 * (blockIdx.x * 4 + (threadIdx.x >> 6));
    o[0] = hw; o[1] = hw_all; o[2] = xcc; o[3] = (unsigned)(t0 & 0xffffffffu);
  }
}

int main() {
  const int blocks = 1024;
  unsigned* d;
  hipMalloc(&d, blocks * 16 * sizeof(unsigned));
  hipFuncSetAttribute(reinterpret_cast<const void*>(census), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
  census<<<blocks, 256, 72 * 1024>>>(d, 2000);
  hipDeviceSynchronize();
  std::vector<unsigned> h(blocks * 16);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  int same_slot = 0;
  std::map<unsigned, std::vector<int>> cu_blocks;   // key: xcc, se, sh, cu
  std::map<int, int> slot_hist;
  for (int b = 0; b < blocks; ++b) {
    bool same = true;
    for (int w = 0; w < 4; ++w) {
      same &= h[4 * (b * 4 + w)] == h[4 * (b * 4)];
      slot_hist[h[4 * (b * 4 + w)]]++;
    }
    same_slot += same;
    const unsigned id = h[4 * (b * 4) + 1];
    const unsigned key = (h[4 * (b * 4) + 2] << 16) | (id & 0xff00);
    cu_blocks[key].push_back(b);
  }
  printf("blocks whose 4 waves share one slot number: %d / %d\n", same_slot, blocks);
  for (auto& kv : slot_hist) printf("slot %d: %d waves\n", kv.first, kv.second);
  printf("distinct (xcc,se,sh,cu): %zu\n", cu_blocks.size());
  int shown = 0, parity_diff = 0, pairs = 0;
  for (auto& kv : cu_blocks) {
    if (shown < 12) {
      printf("cu %06x:", kv.first);
      for (int b : kv.second) printf(" b%d(slot %u,simd0 %u)", b, h[4 * (b * 4)], (h[4 * (b * 4) + 1] >> 4) & 3);
      printf("\n");
      ++shown;
    }
    if (kv.second.size() >= 2) {
      ++pairs;
      parity_diff += (h[4 * (kv.second[0] * 4)] & 1) != (h[4 * (kv.second[1] * 4)] & 1);
    }
  }
  printf("CUs whose first two blocks differ in slot parity: %d / %d\n", parity_diff, pairs);
  return 0;
}
