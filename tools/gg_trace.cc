// Per-block phase trace of gg_kernel on one AlexNet layer (diagnostic; see the CONVNET_GG_TRACE block in csrc/gather_gemm.hip).
// Build (the library sources with tracing compiled in, linked straight into this tool):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCONVNET_GG_TRACE -Wno-comment -o tools/gg_trace tools/gg_trace.cc \
//         convnet_amd/csrc/{state,gather_gemm,pool_norm,elementwise,input_staging,comm}.hip -ldl
// Run on the GPU box:  tools/gg_trace [conv3|conv4|conv5] [fprop|dgrad]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../include/convnet_hip.h"

extern "C" void convnet_hip_debug_set_gg_trace(unsigned long long* dev_buf);

static cudamat mat(int rows, int cols, float scale) {
  cudamat m;
  memset(&m, 0, sizeof m);
  m.size[0] = rows; m.size[1] = cols; m.on_device = 1; m.owns_data = 1;
  const size_t n = (size_t)rows * cols;
  hipMalloc((void**)&m.data_device, n * sizeof(float));
  std::vector<float> h(n);
  unsigned s = 12345u + (unsigned)n;
  for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = scale * ((int)(s >> 8) % 2001 - 1000) * 1e-3f; }
  hipMemcpy(m.data_device, h.data(), n * sizeof(float), hipMemcpyHostToDevice);
  return m;
}

int main(int argc, char** argv) {
  const std::string layer = argc > 1 ? argv[1] : "conv4", dir = argc > 2 ? argv[2] : "fprop";
  int C = 384, H = 13, F = 384, K = 3, pad = 1, stride = 1;
  if (layer == "conv3") C = 256;
  if (layer == "conv5") { F = 256; pad = 0; }
  if (layer == "conv2") { C = 96; H = 55; F = 256; K = 5; pad = 0; stride = 2; }
  if (layer == "conv1") { C = 3; H = 224; F = 96; K = 7; pad = 1; stride = 2; }
  const int N = 256, M = (H + 2 * pad - K) / stride + 1;
  convnet_hip_init(0);
  cudamat x = mat(N, H * H * C, 1.f), w = mat(F, K * K * C, 0.05f), y = mat(N, M * M * F, 1.f);
  Shape4D xs = {{N, H, H, C}}, ws = {{F, K, K, C}}, ys = {{N, M, M, F}};
  ConvDesc d;
  memset(&d, 0, sizeof d);
  d.num_input_channels = C; d.num_output_channels = F; d.kernel_size_y = d.kernel_size_x = K; d.kernel_size_t = 1;
  d.stride_y = d.stride_x = stride; d.stride_t = 1; d.padding_y = d.padding_x = -pad; d.num_groups = 1;
  const int kBlocks = 16384;
  unsigned long long* tr;
  hipMalloc((void**)&tr, (size_t)kBlocks * 4 * 16 * 8);
  hipMemset(tr, 0, (size_t)kBlocks * 4 * 16 * 8);
  auto run = [&]() {
    if (dir == "fprop") convUpGemm(&x, &w, &y, &xs, &ws, &ys, d, 0.f);
    else convDownGemm(&y, &w, &x, &ys, &ws, &xs, d, 0.f);
  };
  for (int i = 0; i < 3; ++i) run();
  hipDeviceSynchronize();
  convnet_hip_debug_set_gg_trace(tr);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  run();
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h((size_t)kBlocks * 4 * 16);
  hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost);
  struct Rec { unsigned hw, xcc; unsigned long long b, e, n, st, mf, sy, minm, maxm, mint, maxt, lb, le; };
  std::vector<Rec> r;
  for (int b = 0; b < kBlocks; ++b) {
    const unsigned long long* o = &h[(size_t)b * 4 * 16];   // wave 0 of the block
    if (o[3] == 0) continue;
    r.push_back({(unsigned)o[0], (unsigned)o[1] & 15, o[2], o[3], o[4], o[5], o[6], o[7], o[8], o[9], o[10], o[11], o[12], o[13]});
  }
  if (r.empty()) { printf("no trace records\n"); return 1; }
  // s_memtime counters of different XCDs are not aligned: every time is taken relative to the first block start on its own XCD
  std::map<unsigned, unsigned long long> x0, x1;
  for (auto& q : r) {
    if (!x0.count(q.xcc) || q.b < x0[q.xcc]) x0[q.xcc] = q.b;
    if (!x1.count(q.xcc) || q.e > x1[q.xcc]) x1[q.xcc] = q.e;
  }
  double span = 0;
  for (auto& kv : x0) span = std::max(span, (double)(x1[kv.first] - kv.second));
  printf("%s %s: %zu traced blocks, event time %.1f us, longest XCD span %.0f ticks (%.0f ticks/us)\n", layer.c_str(), dir.c_str(), r.size(), ms * 1e3, span,
         span / (ms * 1e3));
  auto pct = [](std::vector<double> v, double p) { std::sort(v.begin(), v.end()); return v[(size_t)(p * (v.size() - 1))]; };
  std::vector<double> start, end, dur, pro, epi, loop;
  for (auto& q : r) {
    start.push_back((double)(q.b - x0[q.xcc])); end.push_back((double)(q.e - x0[q.xcc])); dur.push_back((double)(q.e - q.b));
    pro.push_back((double)(q.lb - q.b)); epi.push_back((double)(q.e - q.le)); loop.push_back((double)(q.le - q.lb) / (double)std::max(1ull, q.n));
  }
  printf("block start      p0 %.0f  p50 %.0f  p95 %.0f  max %.0f\n", pct(start, 0), pct(start, .5), pct(start, .95), pct(start, 1));
  printf("block end        p0 %.0f  p5 %.0f  p50 %.0f  p95 %.0f  max %.0f   (mean block duration / span %.3f)\n", pct(end, 0), pct(end, .05), pct(end, .5),
         pct(end, .95), pct(end, 1), [&] { double s = 0; for (double d : dur) s += d; return s / dur.size() / span; }());
  printf("prologue p50 %.0f p95 %.0f ; epilogue p50 %.0f p95 %.0f ; loop ticks per chunk p5 %.0f p50 %.0f p95 %.0f\n", pct(pro, .5), pct(pro, .95), pct(epi, .5),
         pct(epi, .95), pct(loop, .05), pct(loop, .5), pct(loop, .95));
  for (int slot = 0; slot < 2; ++slot) {
    double n = 0, st = 0, mf = 0, sy = 0, cnt = 0, minm = 0, maxm = 0, mint = 0, maxt = 0, d = 0;
    for (auto& q : r) {
      if ((int)(q.hw & 1) != slot) continue;
      cnt += 1; n += q.n; st += q.st; mf += q.mf; sy += q.sy; minm += q.minm; maxm += q.maxm; mint += q.mint; maxt += q.maxt; d += (double)(q.e - q.b);
      st += 0;
    }
    if (cnt == 0) continue;
    double lp = 0;
    for (auto& q : r) if ((int)(q.hw & 1) == slot) lp += (double)(q.le - q.lb);
    printf("wave slot %d: loop ticks per chunk %.0f\n", slot, lp / n);
    printf("wave slot %d: %4.0f blocks, %.1f chunks each; per chunk: staging %.0f  mfma-phase %.0f  wait+barrier %.0f  (total %.0f); per-block min/max mfma-phase %.0f / %.0f, "
           "min/max chunk %.0f / %.0f; block duration %.0f\n", slot, cnt, n / cnt, st / n, mf / n, sy / n, (st + mf + sy) / n, minm / cnt, maxm / cnt, mint / cnt, maxt / cnt, d / cnt);
  }
  printf("span per XCD:");
  for (auto& kv : x0) printf("  x%u %.0f", kv.first, (double)(x1[kv.first] - kv.second));
  printf("\n");
  return 0;
}
