// Runs gpp_kernel (calibration: it is green on hardware), gpw_kernel and wgw_kernel FUNCTIONALLY on the CPU — the real kernel sources
// compiled as host C++ against tests/emu/hip/hip_runtime.h — through their own host launchers (patch_run, wgw_try), on small
// convolutions, against a straightforward double-precision reference.  Prints one line per case; exit status 0 only if all pass.
#include <hip/hip_runtime.h>

#include <random>
#include <string>
#include <vector>

#include "../../convnet_amd/csrc/patch_gemm.hip"
#include "../../convnet_amd/csrc/wgrad_wide.hip"

namespace chip {
alignas(16) float smem[40960 + 4096];   // 160 KB + slack: the block's LDS
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
void note_kernel(const char*, double, int, int) {}
KernelTimer::KernelTimer(const char*, const char*, double, double, double) : slot(-1) {}
KernelTimer::~KernelTimer() {}
void gg_reduce_launch(const GGParams& p, size_t dst_elems, int splits, const char*) {   // gg_reduce_kernel on the host
  const size_t per_row = (size_t)p.DP * p.N;
  for (size_t i = 0; i < dst_elems; ++i) {
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += p.partial[(size_t)k * p.slab + i];
    if (p.scaleTargets != 0.f) s = p.scaleTargets * p.dst[i] + s;
    if (p.bias) s += p.bias[i / per_row];
    if (p.relu) s = s > 0.f ? s : 0.f;
    if (p.mask) s = p.mask[i] > 0.f ? s * p.post_scale : 0.f;
    p.dst[i] = s;
  }
}
void wg_reduce_launch(const WGParams& p, size_t total, int splits, int, const char*) {   // wg_reduce_kernel on the host
  const size_t main = (size_t)p.K * p.F;
  for (size_t i = 0; i < total; ++i) {
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += p.partial[(size_t)k * total + i];
    s *= p.scaleOutput;
    float* d = i < main ? p.dst + i : p.bias_dst + (i - main);
    *d = p.scaleTargets != 0.f ? p.scaleTargets * (*d) + s : s;
  }
}
}  // namespace chip

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
  const bool ok = patch_shape_ok(p);
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
  const bool ok = patch_shape_ok(p);
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
  verdict(std::string(tile_mode == 2 ? "wgw(spread)" : "wgw") + " wgrad N" + std::to_string(g.N) + " C" + std::to_string(g.C) + " " + std::to_string(g.H) + "x" + std::to_string(g.W) + " F" + std::to_string(g.F) +
              " k" + std::to_string(g.Ky) + " s" + std::to_string(g.sy) + " p" + std::to_string(g.pad) + (with_bias ? (p.bias_dst ? " +bias row" : " (no spare row for the bias)") : "") +
              " splits=" + std::to_string(p.splits),
          err, ok);
}

int main(int argc, char** argv) {
  const std::string what = argc > 1 ? argv[1] : "quick";   // gpp | gpw | wgw | quick (a subset of each, ~1 minute) | all
  const bool all = what == "all", quick = what == "quick";   // ("all" does not include gpwtail: its 8-slot chip is a process-wide setting)
  if (what == "gpp" || all || quick) {   // calibration of the harness on a kernel that is green on hardware
    fprop_case(Geo{64, 16, 9, 9, 96, 3, 3, 1, 1, 1}, 1, "gpp(raw)");
    if (!quick) dgrad_case(Geo{64, 96, 6, 6, 16, 3, 3, 1, 1, 1}, 1, "gpp(raw)");
  }
  if (what == "gpw" || all || quick) {
    fprop_case(Geo{64, 16, 9, 9, 96, 3, 3, 1, 1, 1}, 3, "gpw");     // 9-wide rows: a wrap in almost every tile, ragged last tile
    if (!quick) fprop_case(Geo{128, 32, 8, 8, 130, 3, 3, 1, 1, 1}, 3, "gpw");   // two image blocks, two channel blocks (split-K), partial second row tile
    if (!quick) fprop_case(Geo{64, 16, 10, 10, 72, 3, 3, 1, 1, 0}, 3, "gpw");   // pad 0: 8-wide output rows
    dgrad_case(Geo{64, 96, 9, 9, 16, 3, 3, 1, 1, 1}, 3, "gpw");
    if (!quick) dgrad_case(Geo{64, 72, 10, 10, 32, 3, 3, 1, 1, 0}, 3, "gpw");   // conv5 type: 8 x 8 derivatives into 10 x 10
  }
  if (what == "gpwvar" || all) {   // its two variants: grouped staging loads (mode 4), two-stage filter ring (mode 5)
    fprop_case(Geo{64, 16, 9, 9, 96, 3, 3, 1, 1, 1}, 4, "gpw(grouped)");
    dgrad_case(Geo{64, 96, 9, 9, 16, 3, 3, 1, 1, 1}, 4, "gpw(grouped)");
    fprop_case(Geo{64, 16, 9, 9, 96, 3, 3, 1, 1, 1}, 5, "gpw(ring2)");
    fprop_case(Geo{128, 32, 8, 8, 130, 3, 3, 1, 1, 1}, 5, "gpw(ring2)");
    dgrad_case(Geo{64, 96, 9, 9, 16, 3, 3, 1, 1, 1}, 5, "gpw(ring2)");
  }
  if (what == "gpwtail") {   // 11 tiles on an 8-slot "chip", the last round's 3 tiles cut in 3 K-ranges: tail split + gpw_tail_fix_kernel
    setenv("CONVNET_EMU_SLOTS", "8", 1);   // (read once, at the first patch_run of the mode: run this leg in its own process)
    setenv("CONVNET_EMU_TAIL", "3", 1);
    fprop_case(Geo{64, 64, 9, 9, 96, 3, 3, 1, 1, 1}, 3, "gpw(tail split)");
  }
  if (what == "wgw" || all || quick) {
    wgrad_case(Geo{32, 32, 9, 9, 192, 3, 3, 1, 1, 1}, false, 0.f, 1.f);    // 256 x 192 tile, two k tiles (288 rows), border taps
    if (!quick) wgrad_case(Geo{64, 29, 8, 8, 200, 3, 3, 1, 1, 1}, true, 1.f, 0.5f);    // 256 x 256 tile ragged in f, K = 261: bias row in the second k tile, two chunks per pixel
    if (!quick) wgrad_case(Geo{64, 16, 12, 12, 224, 5, 5, 2, 2, 2}, false, 0.f, 1.f);  // stride 2, 5 x 5, two chunks per pixel
  }
  if (what == "wgwvar" || all) {   // the staging loads spread over the chunk (wgrad tile 2)
    wgrad_case(Geo{32, 32, 9, 9, 192, 3, 3, 1, 1, 1}, false, 0.f, 1.f, 2);
    wgrad_case(Geo{64, 29, 8, 8, 200, 3, 3, 1, 1, 1}, true, 1.f, 0.5f, 2);
  }
  std::printf("%s\n", g_fail ? "SOME FAILED" : "ALL PASSED");
  return g_fail ? 1 : 0;
}
