// Runs the library's GEMM-shaped kernels FUNCTIONALLY on the CPU — the real sources (gather_gemm.hip, patch_gemm.hip, wgrad_wide.hip)
// compiled as host C++ against tests/emu/hip/hip_runtime.h — on small problems against a straightforward double-precision reference:
// the default kernels through the C ABI (convUp / convDown / convOutp[Bias] / dot: gg_kernel, ggp_kernel incl. its generic-k mode and
// the stride-class table, wg_kernel in both tile sizes, the slab reduces) as the calibration of the harness — they are green on hardware
// — and the opt-in ones (gpp_kernel, gpw_kernel and variants, wgw_kernel and variant) through their launchers (patch_run, wgw_try).
// Prints one line per case; exit status 0 only if all pass.
#include <hip/hip_runtime.h>

#include <random>
#include <string>
#include <vector>

#include "../../convnet_amd/csrc/gather_gemm.h"   // (the .hip files are compiled one by one against the same stand-in header and linked in:
                                                  // tests/test_emulated_kernels.py)

namespace chip {
alignas(16) float smem[40960 + 4096];   // 160 KB + slack: the block's LDS
alignas(16) float rn_smem[40960];          // pool_norm.hip names its dynamic LDS array differently
hipStream_t stream() { return nullptr; }
static std::vector<char> g_ws[3];
static void* arena(int i, size_t bytes) {
  if (g_ws[i].size() < bytes + 64) g_ws[i].resize(bytes + 64);
  return (void*)(((uintptr_t)g_ws[i].data() + 63) & ~(uintptr_t)63);
}
void* workspace(size_t b) { return arena(0, b); }
void* workspace_aux(size_t b) { return arena(1, b); }
void* workspace_planes(size_t b) { return arena(2, b); }
const float* zero_page() {
  alignas(64) static float z[64] = {};
  z[32] = z[33] = z[34] = z[35] = 1.f;
  return z;
}
int matrix_path() { return 1; }
void set_last_error(const char*) {}
const char* g_last_kernel = "";
void note_kernel(const char* name, double, int, int) { g_last_kernel = name; }
KernelTimer::KernelTimer(const char*, const char*, double, double, double) : slot(-1) {}
KernelTimer::~KernelTimer() {}
// deferred epilogues (state.hip): off here — every entry launches when it is called
bool defer_begin(int, void (*)(PendingOp&)) { return false; }
PendingOp& pending() {
  static PendingOp p = {};
  return p;
}
void flush_pending() {}
long g_absorbed = 0;
}  // namespace chip

extern "C" int copy_on_device(cudamat* src, cudamat* dst) {   // state.hip's
  std::memcpy(dst->data_device, src->data_device, sizeof(float) * (size_t)src->size[0] * src->size[1]);
  return 0;
}

using namespace chip;

struct Geo {
  int N, C, H, W, F, Ky, Kx, sy, sx, pad;   // pad >= 0; the library's ConvDesc carries it negated
  int My() const { return (H + 2 * pad - Ky) / sy + 1; }
  int Mx() const { return (W + 2 * pad - Kx) / sx + 1; }
};
static std::vector<float> rnd(size_t n, unsigned seed) {
  std::mt19937 g(seed);
  std::normal_distribution<float> d(0.f, 1.f);
  std::vector<float> v(n + 16);   // + slack for 16-byte alignment of the start
  for (auto& x : v) x = d(g);
  return v;
}
static float* al16(std::vector<float>& v) { return (float*)(((uintptr_t)v.data() + 15) & ~(uintptr_t)15); }
static double rel_err(const float* a, const std::vector<double>& b) {   // the reference's kernel-test metric: max|a-b| / mean|a+b|
  double mx = 0, mean = 0;
  for (size_t i = 0; i < b.size(); ++i) {
    mx = std::max(mx, std::fabs((double)a[i] - b[i]));
    mean += std::fabs((double)a[i] + b[i]);
  }
  return mx / (mean / b.size());
}
static int g_fail = 0;
static void verdict(const std::string& name, double err, bool ran) {
  const bool ok = ran && err < 1e-4;
  std::printf("%s %s rel_err=%.3g%s\n", ok ? "PASS" : "FAIL", name.c_str(), err, ran ? "" : " (the kernel under test did not take the launch)");
  std::fflush(stdout);
  if (!ok) ++g_fail;
}

// x[c][y][x][n], w[f + F*(tap + TYX*c)], y[f][oy][ox][n]
static void fprop_case(const Geo& g, int mode, const char* tag) {
  const int My = g.My(), Mx = g.Mx(), TYX = g.Ky * g.Kx;
  auto xv = rnd((size_t)g.C * g.H * g.W * g.N, 1), wv = rnd((size_t)g.F * TYX * g.C, 2), yv = rnd((size_t)g.F * My * Mx * g.N, 3);
  float *x = al16(xv), *w = al16(wv), *y = al16(yv);
  std::vector<double> ref((size_t)g.F * My * Mx * g.N);
  for (int f = 0; f < g.F; ++f)
    for (int oy = 0; oy < My; ++oy)
      for (int ox = 0; ox < Mx; ++ox)
        for (int n = 0; n < g.N; ++n) {
          double s = 0;
          for (int c = 0; c < g.C; ++c)
            for (int a = 0; a < g.Ky; ++a)
              for (int b = 0; b < g.Kx; ++b) {
                const int ys = oy * g.sy - g.pad + a, xs = ox * g.sx - g.pad + b;
                if (ys < 0 || ys >= g.H || xs < 0 || xs >= g.W) continue;
                s += (double)x[((size_t)(c * g.H + ys) * g.W + xs) * g.N + n] * w[f + (size_t)g.F * (a * g.Kx + b + TYX * c)];
              }
          ref[((size_t)(f * My + oy) * Mx + ox) * g.N + n] = s;
        }
  convnet_hip_set_patch_mode(mode);
  GGParams p{};   // conv_up_impl (gather_gemm.hip)
  p.A = w; p.src = x; p.dst = y; p.bias = nullptr;
  p.R = g.F; p.K = g.C * TYX; p.N = g.N; p.lda = g.F;
  p.GX = Mx; p.G = My * Mx; p.TX = g.Kx; p.TYX = TYX;
  p.SH = g.H; p.SW = g.W; p.ssy = g.sy; p.ssx = g.sx; p.y0 = -g.pad; p.x0 = -g.pad; p.dir = 1;
  p.DW = Mx; p.DP = My * Mx; p.dsy = 1; p.dsx = 1; p.dy0 = 0; p.dx0 = 0;
  p.scaleTargets = 0.f; p.relu = 0;
  p.KC = g.C;
  const bool ok = patch_shape_ok(p, (size_t)g.N * p.DP * g.F);
  if (ok) {
    const PatchBank bank{w, g.F, g.C, g.Ky, g.Kx, 0, 0, 1, 1, g.Ky, g.Kx, false};
    patch_run(p, (size_t)g.N * p.DP * g.F, "conv_fprop", 0.0, bank);
  }
  verdict(std::string(tag) + " fprop N" + std::to_string(g.N) + " C" + std::to_string(g.C) + " " + std::to_string(g.H) + "x" + std::to_string(g.W) + " F" +
              std::to_string(g.F) + " k" + std::to_string(g.Ky) + " p" + std::to_string(g.pad) + " splits=" + std::to_string(p.splits) + " tail_splits=" +
              std::to_string(p.tail_splits),
          ok ? rel_err(y, ref) : 1.0, ok);
}

// dy[f][oy][ox][n], w as above, dx[c][iy][ix][n]; stride 1
static void dgrad_case(const Geo& g, int mode, const char* tag) {
  const int My = g.My(), Mx = g.Mx(), TYX = g.Ky * g.Kx;
  auto dv = rnd((size_t)g.F * My * Mx * g.N, 4), wv = rnd((size_t)g.F * TYX * g.C, 5), ov = rnd((size_t)g.C * g.H * g.W * g.N, 6);
  float *dy = al16(dv), *w = al16(wv), *dx = al16(ov);
  std::vector<double> ref((size_t)g.C * g.H * g.W * g.N);
  for (int c = 0; c < g.C; ++c)
    for (int iy = 0; iy < g.H; ++iy)
      for (int ix = 0; ix < g.W; ++ix)
        for (int n = 0; n < g.N; ++n) {
          double s = 0;
          for (int f = 0; f < g.F; ++f)
            for (int a = 0; a < g.Ky; ++a)
              for (int b = 0; b < g.Kx; ++b) {
                const int oy = iy + g.pad - a, ox = ix + g.pad - b;
                if (oy < 0 || oy >= My || ox < 0 || ox >= Mx) continue;
                s += (double)dy[((size_t)(f * My + oy) * Mx + ox) * g.N + n] * w[f + (size_t)g.F * (a * g.Kx + b + TYX * c)];
              }
          ref[((size_t)(c * g.H + iy) * g.W + ix) * g.N + n] = s;
        }
  convnet_hip_set_patch_mode(mode);
  GGParams p{};   // conv_down_impl, the one stride class of a stride-1 convolution
  p.src = dy; p.dst = dx; p.bias = nullptr;
  p.R = g.C; p.N = g.N; p.lda = g.C;
  p.SH = My; p.SW = Mx; p.ssy = 1; p.ssx = 1; p.dir = -1;
  p.DW = g.W; p.DP = g.H * g.W; p.dsy = 1; p.dsx = 1;
  p.scaleTargets = 0.f; p.relu = 0; p.mask = nullptr; p.post_scale = 1.f;
  p.KC = g.F; p.apre = 1;
  p.K = g.F * TYX; p.GX = g.W; p.G = g.H * g.W; p.TX = g.Kx; p.TYX = TYX;
  p.y0 = g.pad; p.x0 = g.pad; p.dy0 = 0; p.dx0 = 0;
  const bool ok = patch_shape_ok(p, (size_t)g.N * g.H * g.W * g.C);
  if (ok) {
    const PatchBank bank{w, g.F, g.C, g.Ky, g.Kx, 0, 0, 1, 1, g.Ky, g.Kx, true};
    patch_run(p, (size_t)g.N * g.H * g.W * g.C, "conv_dgrad", 0.0, bank);
  }
  verdict(std::string(tag) + " dgrad N" + std::to_string(g.N) + " C" + std::to_string(g.C) + " " + std::to_string(g.H) + "x" + std::to_string(g.W) + " F" +
              std::to_string(g.F) + " k" + std::to_string(g.Ky) + " p" + std::to_string(g.pad),
          ok ? rel_err(dx, ref) : 1.0, ok);
}

// dW[f + F*k], k = c*TYX + tap; optional bias-gradient row
static void wgrad_case(const Geo& g, bool with_bias, float scaleTargets, float scaleOutput, int tile_mode = 1) {
  const int My = g.My(), Mx = g.Mx(), TYX = g.Ky * g.Kx, K = g.C * TYX;
  auto xv = rnd((size_t)g.C * g.H * g.W * g.N, 7), dv = rnd((size_t)g.F * My * Mx * g.N, 8), wv = rnd((size_t)g.F * K, 9), bv = rnd(g.F, 10);
  float *x = al16(xv), *dy = al16(dv), *dw = al16(wv), *db = al16(bv);
  std::vector<double> ref((size_t)g.F * K), refb(g.F);
  for (int f = 0; f < g.F; ++f) {
    for (int c = 0; c < g.C; ++c)
      for (int a = 0; a < g.Ky; ++a)
        for (int b = 0; b < g.Kx; ++b) {
          double s = 0;
          for (int oy = 0; oy < My; ++oy)
            for (int ox = 0; ox < Mx; ++ox) {
              const int ys = oy * g.sy - g.pad + a, xs = ox * g.sx - g.pad + b;
              if (ys < 0 || ys >= g.H || xs < 0 || xs >= g.W) continue;
              for (int n = 0; n < g.N; ++n)
                s += (double)x[((size_t)(c * g.H + ys) * g.W + xs) * g.N + n] * dy[((size_t)(f * My + oy) * Mx + ox) * g.N + n];
            }
          const size_t i = f + (size_t)g.F * (a * g.Kx + b + TYX * c);
          ref[i] = scaleTargets * dw[i] + scaleOutput * s;
        }
    double sb = 0;
    for (size_t i = 0; i < (size_t)My * Mx * g.N; ++i) sb += dy[(size_t)f * My * Mx * g.N + i];
    refb[f] = scaleTargets * db[f] + scaleOutput * sb;
  }
  convnet_hip_set_wgrad_tile(tile_mode);
  WGParams p{};   // conv_outp_impl (gather_gemm.hip)
  p.bias_dst = with_bias ? db : nullptr;
  p.src = x; p.dout = dy; p.dst = dw;
  p.K = K; p.F = g.F; p.N = g.N;
  p.GX = Mx; p.M = My * Mx; p.TX = g.Kx; p.TYX = TYX; p.SH = g.H; p.SW = g.W;
  p.ssy = g.sy; p.ssx = g.sx; p.y0 = -g.pad; p.x0 = -g.pad;
  p.nchunk = divup(g.N, WG_NB); p.chunks_total = p.M * p.nchunk;
  p.scaleTargets = scaleTargets; p.scaleOutput = scaleOutput;
  const bool ok = wgw_try(p, true, true, "conv_wgrad", 0.0, 0.0);
  double err = ok ? rel_err(dw, ref) : 1.0;
  if (ok && with_bias && p.bias_dst) err = std::max(err, rel_err(db, refb));
  verdict(std::string("wgw") + " wgrad N" + std::to_string(g.N) + " C" + std::to_string(g.C) + " " + std::to_string(g.H) + "x" + std::to_string(g.W) + " F" + std::to_string(g.F) +
              " k" + std::to_string(g.Ky) + " s" + std::to_string(g.sy) + " p" + std::to_string(g.pad) + (with_bias ? (p.bias_dst ? " +bias row" : " (no spare row for the bias)") : "") +
              " splits=" + std::to_string(p.splits),
          err, ok);
}

// ---- the default kernels through the C ABI ------------------------------------------------------------------------------------------
static cudamat mat(float* d, int rows, int cols) {
  cudamat m{};
  m.data_device = d; m.on_device = 1; m.size[0] = rows; m.size[1] = cols; m.owns_data = 0;
  return m;
}
static ConvDesc desc(const Geo& g) {
  ConvDesc d{};
  d.num_input_channels = g.C; d.num_output_channels = g.F; d.kernel_size_y = g.Ky; d.kernel_size_x = g.Kx; d.kernel_size_t = 1;
  d.stride_y = g.sy; d.stride_x = g.sx; d.stride_t = 1; d.padding_y = -g.pad; d.padding_x = -g.pad; d.num_groups = 1;
  return d;
}
static std::string gname(const Geo& g) {
  return "N" + std::to_string(g.N) + " C" + std::to_string(g.C) + " " + std::to_string(g.H) + "x" + std::to_string(g.W) + " F" + std::to_string(g.F) + " k" +
         std::to_string(g.Ky) + " s" + std::to_string(g.sy) + " p" + std::to_string(g.pad);
}
static void abi_conv_case(const Geo& g, const char* which, int patch_mode = 0) {
  convnet_hip_set_patch_mode(patch_mode);
  convnet_hip_set_wgrad_tile(0);
  const int My = g.My(), Mx = g.Mx(), TYX = g.Ky * g.Kx, K = g.C * TYX;
  auto xv = rnd((size_t)g.C * g.H * g.W * g.N, 11), wv = rnd((size_t)g.F * K, 12), yv = rnd((size_t)g.F * My * Mx * g.N, 13), bv = rnd(g.F, 14);
  float *x = al16(xv), *w = al16(wv), *y = al16(yv), *b = al16(bv);
  cudamat mx = mat(x, g.N, g.H * g.W * g.C), mw = mat(w, g.F, K), my = mat(y, g.N, My * Mx * g.F), mb = mat(b, 1, g.F);
  Shape4D sx{{g.N, g.W, g.H, g.C}}, sw{{g.F, g.Kx, g.Ky, g.C}}, sy{{g.N, Mx, My, g.F}};
  const std::string what = which;
  if (what == "up") {
    std::vector<double> ref((size_t)g.F * My * Mx * g.N);
    for (int f = 0; f < g.F; ++f)
      for (int oy = 0; oy < My; ++oy)
        for (int ox = 0; ox < Mx; ++ox)
          for (int n = 0; n < g.N; ++n) {
            double s = 0;
            for (int c = 0; c < g.C; ++c)
              for (int a = 0; a < g.Ky; ++a)
                for (int bb = 0; bb < g.Kx; ++bb) {
                  const int ys = oy * g.sy - g.pad + a, xs = ox * g.sx - g.pad + bb;
                  if (ys < 0 || ys >= g.H || xs < 0 || xs >= g.W) continue;
                  s += (double)x[((size_t)(c * g.H + ys) * g.W + xs) * g.N + n] * w[f + (size_t)g.F * (a * g.Kx + bb + TYX * c)];
                }
            ref[((size_t)(f * My + oy) * Mx + ox) * g.N + n] = s;
          }
    convUp(&mx, &mw, &my, &sx, &sw, &sy, desc(g), 0.f);
    verdict("abi convUp " + gname(g) + " [" + chip::g_last_kernel + "]", rel_err(y, ref), true);
  } else if (what == "down") {
    std::vector<double> ref((size_t)g.C * g.H * g.W * g.N);
    for (int c = 0; c < g.C; ++c)
      for (int iy = 0; iy < g.H; ++iy)
        for (int ix = 0; ix < g.W; ++ix)
          for (int n = 0; n < g.N; ++n) {
            double s = 0;
            for (int f = 0; f < g.F; ++f)
              for (int a = 0; a < g.Ky; ++a)
                for (int bb = 0; bb < g.Kx; ++bb) {
                  const int ty = iy + g.pad - a, tx = ix + g.pad - bb;
                  if (ty < 0 || tx < 0 || ty % g.sy || tx % g.sx) continue;
                  const int oy = ty / g.sy, ox = tx / g.sx;
                  if (oy >= My || ox >= Mx) continue;
                  s += (double)y[((size_t)(f * My + oy) * Mx + ox) * g.N + n] * w[f + (size_t)g.F * (a * g.Kx + bb + TYX * c)];
                }
            ref[((size_t)(c * g.H + iy) * g.W + ix) * g.N + n] = s;
          }
    convDown(&my, &mw, &mx, &sy, &sw, &sx, desc(g), 0.f);
    verdict("abi convDown " + gname(g) + " [" + chip::g_last_kernel + "]", rel_err(x, ref), patch_mode == 0 || std::string(chip::g_last_kernel).find("gpw") == 0);
  } else {   // "outp": weight gradients + bias gradient
    std::vector<double> ref((size_t)g.F * K), refb(g.F);
    for (int f = 0; f < g.F; ++f) {
      for (int c = 0; c < g.C; ++c)
        for (int a = 0; a < g.Ky; ++a)
          for (int bb = 0; bb < g.Kx; ++bb) {
            double s = 0;
            for (int oy = 0; oy < My; ++oy)
              for (int ox = 0; ox < Mx; ++ox) {
                const int ys = oy * g.sy - g.pad + a, xs = ox * g.sx - g.pad + bb;
                if (ys < 0 || ys >= g.H || xs < 0 || xs >= g.W) continue;
                for (int n = 0; n < g.N; ++n)
                  s += (double)x[((size_t)(c * g.H + ys) * g.W + xs) * g.N + n] * y[((size_t)(f * My + oy) * Mx + ox) * g.N + n];
              }
            const size_t i = f + (size_t)g.F * (a * g.Kx + bb + TYX * c);
            ref[i] = 1.0 * w[i] + 0.5 * s;
          }
      double sb = 0;
      for (size_t i = 0; i < (size_t)My * Mx * g.N; ++i) sb += y[(size_t)f * My * Mx * g.N + i];
      refb[f] = 1.0 * b[f] + 0.5 * sb;
    }
    convOutpBias(&mx, &my, &mw, &mb, &sx, &sy, &sw, desc(g), 1.f, 0.5f);
    verdict("abi convOutpBias " + gname(g), std::max(rel_err(w, ref), rel_err(b, refb)), true);
  }
}
// the three products of an FC layer (fc_edge.cc:54-74): dot(mat1, mat2, target, beta, alpha) with cudamat's is_trans flags
static void abi_dot_case(int N, int D, int F) {
  auto xv = rnd((size_t)N * D, 21), wv = rnd((size_t)F * D, 22), dv = rnd((size_t)N * F, 23), tv = rnd((size_t)std::max(N, F) * std::max(D, F), 24);
  float *x = al16(xv), *w = al16(wv), *dy = al16(dv), *t = al16(tv);
  double worst = 0;
  {   // out(N,F) = in(N,D) * W(F,D)^T
    cudamat mx = mat(x, N, D), mw = mat(w, F, D), mt = mat(t, N, F);
    mw.is_trans = 1;
    std::vector<double> ref((size_t)N * F);
    for (int n = 0; n < N; ++n)
      for (int f = 0; f < F; ++f) {
        double s = 0;
        for (int d = 0; d < D; ++d) s += (double)x[n + (size_t)N * d] * w[f + (size_t)F * d];
        ref[n + (size_t)N * f] = s;
      }
    if (dot(&mx, &mw, &mt, 0.f, 1.f)) worst = 1;
    worst = std::max(worst, rel_err(t, ref));
  }
  {   // d_in(N,D) = d_out(N,F) * W(F,D)
    cudamat md = mat(dy, N, F), mw = mat(w, F, D), mt = mat(t, N, D);
    std::vector<double> ref((size_t)N * D);
    for (int n = 0; n < N; ++n)
      for (int d = 0; d < D; ++d) {
        double s = 0;
        for (int f = 0; f < F; ++f) s += (double)dy[n + (size_t)N * f] * w[f + (size_t)F * d];
        ref[n + (size_t)N * d] = s;
      }
    if (dot(&md, &mw, &mt, 0.f, 1.f)) worst = 1;
    worst = std::max(worst, rel_err(t, ref));
  }
  {   // dW(F,D) = d_out(N,F)^T * in(N,D) / N
    cudamat md = mat(dy, N, F), mx = mat(x, N, D), mt = mat(t, F, D);
    md.is_trans = 1;
    std::vector<double> ref((size_t)F * D);
    for (int f = 0; f < F; ++f)
      for (int d = 0; d < D; ++d) {
        double s = 0;
        for (int n = 0; n < N; ++n) s += (double)dy[n + (size_t)N * f] * x[n + (size_t)N * d];
        ref[f + (size_t)F * d] = s / N;
      }
    if (dot(&md, &mx, &mt, 0.f, 1.f / N)) worst = 1;
    worst = std::max(worst, rel_err(t, ref));
  }
  verdict("abi dot NT / NN / TN N" + std::to_string(N) + " D" + std::to_string(D) + " F" + std::to_string(F), worst, true);
}

// ---- the HBM-bound kernels through the C ABI, against the CPU oracle (oracle/liboracle.so: the C restatement of the reference) ---------
extern "C" {
void oracle_max_pool(const float*, float*, int N, int C, int H, int W, int Ky, int Kx, int sy, int sx, int pady, int padx, int My, int Mx, float scaleTargets, float scaleOutput);
void oracle_max_pool_undo(const float* images, const float* maxGrads, const float* maxActs, float* targets, int N, int C, int H, int W, int Ky, int Kx, int sy,
                          int sx, int pady, int padx, int My, int Mx, float scaleTargets);
void oracle_rnorm(const float* images, float* targets, int num_locs, int C, int sizeF, float addScale, float powScale, int blocked);
void oracle_rnorm_undo(const float* outGrads, const float* inputs, float* targets, int num_locs, int C, int sizeF, float addScale, float powScale, int blocked);
void oracle_sgd_step(float* grad, float* param, float* history, int rows, int cols, float l2_decay, float gradient_clip, float epsilon, float momentum,
                     float norm_limit, float norm_constraint);
}
static double rel_err_f(const float* a, const float* b, size_t n) {
  std::vector<double> r(b, b + n);
  return rel_err(a, r);
}
static void pool_case(int N, int C, int H, int K, int st) {
  const int M = (H - K) / st + 1;
  Geo g{N, C, H, H, C, K, K, st, st, 0};
  auto xv = rnd((size_t)C * H * H * N, 31), yv = rnd((size_t)C * M * M * N, 32), gv = rnd((size_t)C * M * M * N, 33), dv = rnd((size_t)C * H * H * N, 34);
  float *x = al16(xv), *y = al16(yv), *gr = al16(gv), *dx = al16(dv);
  std::vector<float> yr((size_t)C * M * M * N), dr((size_t)C * H * H * N);
  cudamat mx = mat(x, N, H * H * C), my = mat(y, N, M * M * C), mg = mat(gr, N, M * M * C), md = mat(dx, N, H * H * C);
  Shape4D sx{{N, H, H, C}}, sy{{N, M, M, C}};
  MaxPoolGemm(&mx, &my, &sx, &sy, desc(g), 0.f, 1.f);
  oracle_max_pool(x, yr.data(), N, C, H, H, K, K, st, st, 0, 0, M, M, 0.f, 1.f);
  const double e1 = rel_err_f(y, yr.data(), yr.size());
  MaxPoolUndoGemm(&mx, &mg, &my, &md, &sx, &sy, desc(g), 0.f);
  oracle_max_pool_undo(x, gr, yr.data(), dr.data(), N, C, H, H, K, K, st, st, 0, 0, M, M, 0.f);
  const double e2 = rel_err_f(dx, dr.data(), dr.size());
  verdict("abi MaxPool + MaxPoolUndo N" + std::to_string(N) + " C" + std::to_string(C) + " " + std::to_string(H) + "x" + std::to_string(H) + " k" + std::to_string(K) +
              " s" + std::to_string(st),
          std::max(e1, e2), true);
}
// MaxPoolMask + MaxPoolUndoMask (3 x 3 stride 2) against the oracle's MaxPool + MaxPoolUndo; quantised inputs: every window holds ties
static void pool_mask_case(int N, int C, int H) {
  const int K = 3, st = 2, M = (H - K) / st + 1;
  Geo g{N, C, H, H, C, K, K, st, st, 0};
  auto xv = rnd((size_t)C * H * H * N, 35), yv = rnd((size_t)C * M * M * N, 36), gv = rnd((size_t)C * M * M * N, 37), dv = rnd((size_t)C * H * H * N, 38);
  for (float& v : xv) v = std::floor(2.f * v);
  std::vector<float> mv((size_t)C * M * M * N / 2 + 8);
  float *x = al16(xv), *y = al16(yv), *gr = al16(gv), *dx = al16(dv), *mk = al16(mv);
  std::vector<float> yr((size_t)C * M * M * N), dr((size_t)C * H * H * N);
  cudamat mx = mat(x, N, H * H * C), my = mat(y, N, M * M * C), mg = mat(gr, N, M * M * C), md = mat(dx, N, H * H * C), mm = mat(mk, N, M * M * C / 2);
  Shape4D sx{{N, H, H, C}}, sy{{N, M, M, C}};
  const int rc1 = MaxPoolMask(&mx, &my, &mm, &sx, &sy, desc(g));
  oracle_max_pool(x, yr.data(), N, C, H, H, K, K, st, st, 0, 0, M, M, 0.f, 1.f);
  const double e1 = rel_err_f(y, yr.data(), yr.size());
  const int rc2 = MaxPoolUndoMask(&mg, &mm, &md, &sx, &sy, desc(g), 0.f, 0);
  oracle_max_pool_undo(x, gr, yr.data(), dr.data(), N, C, H, H, K, K, st, st, 0, 0, M, M, 0.f);
  const double e2 = rel_err_f(dx, dr.data(), dr.size());
  // ... and with the ReLU' of the layer below fused: the derivative survives where the input itself is positive
  const int rc3 = MaxPoolUndoMask(&mg, &mm, &md, &sx, &sy, desc(g), 0.f, 1);
  for (size_t i = 0; i < dr.size(); ++i) dr[i] = x[i] > 0.f ? dr[i] : 0.f;
  const double e3 = rel_err_f(dx, dr.data(), dr.size());
  verdict("abi MaxPoolMask + MaxPoolUndoMask N" + std::to_string(N) + " C" + std::to_string(C) + " " + std::to_string(H) + "x" + std::to_string(H),
          std::max(e1, std::max(e2, e3)), rc1 == 0 && rc2 == 0 && rc3 == 0);
}
static void rnorm_case(int N, int C, int HW, int sizeF) {
  const int locs = HW * N;
  auto xv = rnd((size_t)C * locs, 41), yv = rnd((size_t)C * locs, 42), gv = rnd((size_t)C * locs, 43), dv = rnd((size_t)C * locs, 44);
  float *x = al16(xv), *y = al16(yv), *gr = al16(gv), *dx = al16(dv);
  std::vector<float> yr((size_t)C * locs), dr((size_t)C * locs);
  cudamat mx = mat(x, N, HW * C), my = mat(y, N, HW * C), mg = mat(gr, N, HW * C), md = mat(dx, N, HW * C);
  ResponseNormCrossMapGemm(&mx, &my, C, sizeF, 0.001f, 0.75f, false);
  oracle_rnorm(x, yr.data(), locs, C, sizeF, 0.001f, 0.75f, 0);
  const double e1 = rel_err_f(y, yr.data(), yr.size());
  ResponseNormCrossMapUndoGemm(&mg, &mx, &md, C, sizeF, 0.001f, 0.75f, false);
  oracle_rnorm_undo(gr, x, dr.data(), locs, C, sizeF, 0.001f, 0.75f, 0);
  const double e2 = rel_err_f(dx, dr.data(), dr.size());
  verdict("abi ResponseNormCrossMap + Undo N" + std::to_string(N) + " C" + std::to_string(C) + " pixels " + std::to_string(HW) + " size " + std::to_string(sizeF), std::max(e1, e2), true);
}
static void sgd_case(int rows, int cols) {
  const size_t n = (size_t)rows * cols;
  auto gv = rnd(n, 51), pv = rnd(n, 52), hv = rnd(n, 53);
  float *g = al16(gv), *p = al16(pv), *h = al16(hv);
  std::vector<float> g2(g, g + n), p2(p, p + n), h2(h, h + n);
  cudamat mg = mat(g, rows, cols), mp = mat(p, rows, cols), mh = mat(h, rows, cols);
  const int rc = sgd_momentum_step(&mg, &mp, &mh, 0.0005f, 0.f, 0.01f, 0.9f);
  oracle_sgd_step(g2.data(), p2.data(), h2.data(), rows, cols, 0.0005f, 0.f, 0.01f, 0.9f, 0.f, 0.f);
  verdict("abi sgd_momentum_step " + std::to_string(rows) + " x " + std::to_string(cols), std::max(rel_err_f(p, p2.data(), n), rel_err_f(h, h2.data(), n)), rc == 0);
}

int main(int argc, char** argv) {
  const std::string what = argc > 1 ? argv[1] : "quick";   // abi | gpp | gpw | gpv | gpvtail | gpwtail | wgw | wgwvar | wgwfin | quick (a subset of each, ~1 minute) | all
  const bool all = what == "all", quick = what == "quick";   // ("all" does not include gpwtail: its 8-slot chip is a process-wide setting)
  if (what == "abi" || all || quick) {   // the default kernels through the C ABI: the calibration of the harness (green on hardware)
    abi_conv_case(Geo{64, 16, 9, 9, 128, 3, 3, 1, 1, 1}, "up");      // ggp_kernel<2,2,2,128>, pre-split filter planes
    abi_conv_case(Geo{32, 16, 9, 9, 128, 3, 3, 1, 1, 1}, "outp");    // wg_kernel<2,2,2,2>, K = 144: bias row in the second k tile
    abi_conv_case(Geo{64, 3, 15, 15, 96, 7, 7, 2, 2, 1}, "up");      // conv1 type: gfc_kernel (patch-resident, filter bank in LDS), 12 tiles on 4 blocks
    if (!quick) {
      abi_conv_case(Geo{48, 3, 15, 15, 96, 7, 7, 2, 2, 1}, "up");      // conv1 type, N % 32 != 0: generic-k order on ggp_kernel<1,4,3,64>
      abi_conv_case(Geo{32, 3, 27, 27, 80, 7, 7, 2, 2, 1}, "up");      // conv1 type: gfc_kernel, two column groups (12 = 8 + 4 pixels), 80 of 96 rows, 24 tiles on 4 blocks
      abi_conv_case(Geo{64, 128, 9, 9, 32, 3, 3, 1, 1, 1}, "down");    // one stride class
      abi_conv_case(Geo{64, 96, 11, 11, 32, 5, 5, 2, 2, 0}, "down");   // conv2 type: four stride classes in one launch
      abi_conv_case(Geo{32, 3, 15, 15, 96, 7, 7, 2, 2, 1}, "outp");    // conv1 type: the 160 x 96 tile of 16 x 16 MFMAs, bias row in a padding row
      abi_conv_case(Geo{96, 3, 21, 17, 80, 7, 7, 2, 2, 1}, "outp");    // ... three image chunks per pixel, rectangular, 80 of 96 filters
      abi_conv_case(Geo{32, 5, 15, 19, 96, 4, 7, 2, 2, 1}, "outp");    // ... 5 channels x 4 x 7 taps (20 bands, K = 140)
      abi_conv_case(Geo{32, 3, 12, 12, 96, 7, 7, 1, 1, 3}, "outp");    // ... stride 1, three columns of padding on either side
      abi_dot_case(64, 256, 128);
      pool_case(32, 32, 21, 3, 2);        // pool1 type (3 x 3 stride 2; 441 pixels: the 2 x 2-block undo kernel)
      pool_case(64, 16, 7, 3, 2);         // a small map: the per-output undo kernel
      pool_case(32, 8, 43, 3, 2);         // 21 x 21 outputs: the 2 x 2-block forward kernel with an odd last row and column
      pool_mask_case(32, 8, 21);          // the mask pair: 10 x 10 windows on a 21 x 21 map, every window with ties
      pool_mask_case(16, 4, 12);          // even map: the last input row and column lie outside every window
      rnorm_case(32, 96, 25, 5);          // rnorm1 type: 96 channels, window 5
      rnorm_case(16, 256, 9, 5);
      rnorm_case(32, 96, 9, 24);          // rnorm1 itself: window 24 = the fast kernels' 6-channel segments
      rnorm_case(16, 256, 4, 64);         // rnorm2 itself: window 64, 8-channel segments
      rnorm_case(16, 100, 9, 24);         // 17 lane groups of 6 channels: two spare channels on the zero rows
      rnorm_case(8, 70, 4, 64);           // 9 groups of 8, two spare channels; 32 locations = half a 64-location tile
      sgd_case(96, 147);
    }
  }
  if (what == "gpp" || all || quick) {   // an opt-in kernel that is green on hardware too
    fprop_case(Geo{64, 16, 9, 9, 96, 3, 3, 1, 1, 1}, 1, "gpp(raw)");
    if (!quick) dgrad_case(Geo{64, 96, 6, 6, 16, 3, 3, 1, 1, 1}, 1, "gpp(raw)");
  }
  if (what == "gpw" || all || quick) {
    fprop_case(Geo{64, 16, 9, 9, 96, 3, 3, 1, 1, 1}, 4, "gpw");     // 9-wide rows: a wrap in almost every tile, ragged last tile
    if (!quick) fprop_case(Geo{128, 32, 8, 8, 130, 3, 3, 1, 1, 1}, 4, "gpw");   // two image blocks, two channel blocks (split-K), partial second row tile
    if (!quick) fprop_case(Geo{64, 16, 10, 10, 72, 3, 3, 1, 1, 0}, 4, "gpw");   // pad 0: 8-wide output rows
    dgrad_case(Geo{64, 96, 9, 9, 16, 3, 3, 1, 1, 1}, 4, "gpw");
    if (!quick) dgrad_case(Geo{64, 72, 10, 10, 32, 3, 3, 1, 1, 0}, 4, "gpw");   // conv5 type: 8 x 8 derivatives into 10 x 10
  }
  if (what == "gpv" || all || quick) {   // gpv_kernel: tap rows in groups of three and two (patch_gemm.hip)
    fprop_case(Geo{64, 16, 23, 23, 96, 5, 5, 2, 2, 0}, 4, "gpv");      // conv2's form ({0,2,4} / {1,3}) on the 96-row build: 10-wide output rows, ragged last tile
    abi_conv_case(Geo{64, 96, 19, 19, 16, 5, 5, 2, 2, 0}, "down", 4);  // conv2's input gradient: classes of 3x3, 3x2, 2x3, 2x2 taps in one launch, 96 rows
    if (!quick) {
      fprop_case(Geo{64, 16, 21, 21, 130, 5, 5, 2, 2, 0}, 4, "gpv");   // 128-row build, partial second row tile, 9-wide rows: a wrap in almost every tile
      fprop_case(Geo{128, 32, 19, 19, 96, 5, 5, 2, 2, 0}, 4, "gpv");   // two image blocks, two channel blocks
      fprop_case(Geo{64, 16, 21, 25, 96, 5, 5, 2, 2, 2}, 4, "gpv");    // padding 2: border columns on the zero page, whole tap rows skipped, rectangular
      fprop_case(Geo{64, 16, 20, 20, 128, 4, 4, 2, 2, 1}, 4, "gpv");   // 4 x 4 stride 2: groups of two and two
      fprop_case(Geo{64, 16, 10, 10, 128, 2, 2, 1, 1, 0}, 4, "gpv");   // one group of two: every superchunk is (two slots, one slot)
      fprop_case(Geo{64, 16, 9, 9, 72, 3, 3, 1, 1, 1}, 4, "gpv");      // gpw_kernel's 3 x 3 stride-1 case on the 96-row build
      abi_conv_case(Geo{64, 132, 17, 21, 16, 5, 5, 2, 2, 2}, "down", 4);   // 128-row build, two row tiles, padded: classes start at different pixels
      abi_conv_case(Geo{64, 96, 20, 20, 16, 4, 4, 2, 2, 1}, "down", 4);    // four classes of 2 x 2 taps
    }
  }
  if (what == "gpvtail") {   // 13 tiles on an 8-slot "chip", the last round's 5 tiles cut in 3 K-ranges (ranges begin inside a tap row's groups)
    setenv("CONVNET_EMU_SLOTS", "8", 1);
    setenv("CONVNET_EMU_TAIL", "3", 1);
    fprop_case(Geo{64, 32, 23, 23, 96, 5, 5, 2, 2, 0}, 4, "gpv(tail split)");
  }
  if (what == "gpwtail") {   // 11 tiles on an 8-slot "chip", the last round's 3 tiles cut in 3 K-ranges: tail split + gpw_tail_fix_kernel
    setenv("CONVNET_EMU_SLOTS", "8", 1);   // (read once, at the first patch_run of the mode: run this leg in its own process)
    setenv("CONVNET_EMU_TAIL", "3", 1);
    fprop_case(Geo{64, 64, 9, 9, 96, 3, 3, 1, 1, 1}, 4, "gpw(tail split)");
  }
  if (what == "wgw" || all || quick) {
    wgrad_case(Geo{32, 32, 9, 9, 192, 3, 3, 1, 1, 1}, false, 0.f, 1.f);    // 256 x 192 tile, two k tiles (288 rows), border taps
    if (!quick) wgrad_case(Geo{64, 29, 8, 8, 200, 3, 3, 1, 1, 1}, true, 1.f, 0.5f);    // 256 x 256 tile ragged in f, K = 261: bias row in the second k tile, two chunks per pixel
    if (!quick) wgrad_case(Geo{64, 16, 12, 12, 224, 5, 5, 2, 2, 2}, false, 0.f, 1.f);  // stride 2, 5 x 5, two chunks per pixel
    if (!quick) wgrad_case(Geo{32, 32, 9, 9, 198, 3, 3, 1, 1, 1}, false, 0.f, 1.f);    // F % 4 != 0: the direct write-out
  }
  if (what == "wgwfin") {   // one block per tile (no slabs): scaleTargets / scaleOutput and the bias row in the kernel's own epilogue, both write-outs
    setenv("CONVNET_EMU_WG_SPLITS", "1", 1);
    wgrad_case(Geo{64, 29, 8, 8, 200, 3, 3, 1, 1, 1}, true, 1.f, 0.5f);
    wgrad_case(Geo{32, 29, 8, 8, 198, 3, 3, 1, 1, 1}, true, 1.f, 0.5f);
  }
  std::printf("%s\n", g_fail ? "SOME FAILED" : "ALL PASSED");
  return g_fail ? 1 : 0;
}
