// Max/avg pooling (+undo) and cross-map response normalisation (+undo) on the CHWN layout.
// Reference: cudamat/cudamat_conv_gemm.cu:153-300 (kPool/kAvgPoolUndo/kMaxPoolUndo, scatter +
// atomicAdd) and :438-540 (kCrossMapDenoms/kCrossMapRNorm/kCrossMapRNormUndo); semantics pinned by
// the CPU oracle (src/CPUMatrix.cc:574-827, eigenmat/cpumat_conv.cc:462-560).
// These are HBM-bound: every kernel is a pure *gather* (one lane owns 4 consecutive images of one
// output element, float4 in / float4 out, no atomics), so the traffic is the algorithmic
// read-once/write-once bytes plus L2-absorbed window overlap.
#include <cfloat>
#include <cstdlib>

#include "common.h"

namespace chip {

using f32x4 = __attribute__((ext_vector_type(4))) float;

struct PoolGeo {
  int N, C, H, W, Ky, Kx, sy, sx, py, px, My, Mx;
  int nvec;  // ceil(N/4)
  // XCD-aware block order of the fixed-window kernels (0 = plain 3-D grid): see pool_block()
  int xper, xtotal, xrows, xbx;
};

// Fixed-window kernels, block -> (x-block, row, channel).  A 3x3 stride-2 window shares an input row with the window below
// it, and with the plain (x, row, channel) grid those two blocks sit 14 block ids apart, i.e. on different XCDs (hardware
// places block b on XCD b%8): the shared row is fetched twice from the fabric — measured 1.67x the algorithmic read bytes for
// pool1 forward and undo (profiles/r01_pmc_traffic_bench.json), which puts both kernels on the memory system's ceiling.
// With the 1-D launch each XCD owns a contiguous run of logical blocks ordered ROW-fastest, so the rows a block shares with
// its vertical neighbours are requested by the same XCD back to back and the second request hits its L2.
__device__ __forceinline__ bool pool_block(const PoolGeo& g, int& bx, int& row, int& c) {
  if (g.xper == 0) {
    bx = blockIdx.x; row = blockIdx.y; c = blockIdx.z;
    return true;
  }
  const int b = blockIdx.x, slot = b >> 3;
  const int L = (b & 7) * g.xper + slot;
  if (slot >= g.xper || L >= g.xtotal) return false;
  row = L % g.xrows;
  const int t = L / g.xrows;
  bx = t % g.xbx;
  c = t / g.xbx;
  return true;
}

__device__ __forceinline__ f32x4 ldv(const float* p, int n, int N, bool vec) {
  if (vec) return *reinterpret_cast<const f32x4*>(p);
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (n + e < N) v[e] = p[e];
  return v;
}
__device__ __forceinline__ void stv(float* p, f32x4 v, int n, int N, bool vec) {
  if (vec) {
    *reinterpret_cast<f32x4*>(p) = v;
    return;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (n + e < N) p[e] = v[e];
}

// work item = (channel c, output pixel m, image quad q); q fastest so a wave reads 1 KiB runs.
template <bool MAX>
__global__ void pool_fwd_kernel(const float* __restrict__ in, float* __restrict__ out, PoolGeo g, float st, float so, bool vec) {
  const size_t total = (size_t)g.C * g.My * g.Mx * g.nvec;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int q = (int)(i % g.nvec);
    size_t r = i / g.nvec;
    const int ox = (int)(r % g.Mx);
    r /= g.Mx;
    const int oy = (int)(r % g.My);
    const int c = (int)(r / g.My);
    const int n = 4 * q;
    const int y0 = max(0, oy * g.sy + g.py), y1 = min(g.H, oy * g.sy + g.py + g.Ky);
    const int x0 = max(0, ox * g.sx + g.px), x1 = min(g.W, ox * g.sx + g.px + g.Kx);
    f32x4 acc = MAX ? f32x4{-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX} : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int y = y0; y < y1; ++y)
      for (int x = x0; x < x1; ++x) {
        const f32x4 v = ldv(in + ((size_t)(c * g.H + y) * g.W + x) * g.N + n, n, g.N, vec);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = MAX ? (acc[e] < v[e] ? v[e] : acc[e]) : acc[e] + v[e];
      }
    if (!MAX) {
      const float cnt = (float)((y1 - y0) * (x1 - x0));
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = acc[e] / cnt;
    }
    float* op = out + ((size_t)(c * g.My + oy) * g.Mx + ox) * g.N + n;
    f32x4 res = so * acc;
    if (st != 0.f) res = st * ldv(op, n, g.N, vec) + res;
    stv(op, res, n, g.N, vec);
  }
}

__device__ __forceinline__ void cover(int i, int pad, int k, int s, int M, int& lo, int& hi) {
  // pooled coordinates whose window covers input coordinate i (src/CPUMatrix.cc:669-673)
  lo = (i - pad < k) ? 0 : (i - pad - k) / s + 1;
  hi = min(M, 1 + (i - pad) / s);
}

// work item = (c, input pixel, image quad).  MAX: d_in += d_out[o] for every covering o whose max
// equals this input (all ties count — SURVEY.md fact 9).  AVG: d_in += d_out[o]/|clipped window o|.
template <bool MAX>
__global__ void pool_undo_kernel(const float* __restrict__ images, const float* __restrict__ grads, const float* __restrict__ acts,
                                 float* __restrict__ out, PoolGeo g, float st, bool vec, bool relu_mask) {
  const size_t total = (size_t)g.C * g.H * g.W * g.nvec;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int q = (int)(i % g.nvec);
    size_t r = i / g.nvec;
    const int ix = (int)(r % g.W);
    r /= g.W;
    const int iy = (int)(r % g.H);
    const int c = (int)(r / g.H);
    const int n = 4 * q;
    int oy0, oy1, ox0, ox1;
    cover(iy, g.py, g.Ky, g.sy, g.My, oy0, oy1);
    cover(ix, g.px, g.Kx, g.sx, g.Mx, ox0, ox1);
    const bool inside = ix < g.px + g.sx * (g.Mx - 1) + g.Kx && iy < g.py + g.sy * (g.My - 1) + g.Ky;
    const size_t t = ((size_t)(c * g.H + iy) * g.W + ix) * g.N + n;
    f32x4 img = {0.f, 0.f, 0.f, 0.f};
    if (MAX) img = ldv(images + t, n, g.N, vec);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (inside)
      for (int oy = oy0; oy < oy1; ++oy) {
        const float ry = fminf((float)g.H, (float)(g.py + oy * g.sy + g.Ky)) - fmaxf(0.f, (float)(g.py + oy * g.sy));
        for (int ox = ox0; ox < ox1; ++ox) {
          const size_t o = ((size_t)(c * g.My + oy) * g.Mx + ox) * g.N + n;
          const f32x4 gv = ldv(grads + o, n, g.N, vec);
          if (MAX) {
            const f32x4 av = ldv(acts + o, n, g.N, vec);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += (img[e] == av[e]) ? gv[e] : 0.f;
          } else {
            const float rx = fminf((float)g.W, (float)(g.px + ox * g.sx + g.Kx)) - fmaxf(0.f, (float)(g.px + ox * g.sx));
            const float inv = 1.0f / (rx * ry);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += gv[e] * inv;
          }
        }
      }
    if (st != 0.f) acc = st * ldv(out + t, n, g.N, vec) + acc;
    if (MAX && relu_mask) {   // fused ReLU' of the layer below: its state IS `images`
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = img[e] > 0.f ? acc[e] : 0.f;
    }
    stv(out + t, acc, n, g.N, vec);
  }
}

// Fixed-window variants (square K x K, stride S known at compile time; N % 4 == 0): grid = (W*nvec/256, rows,
// C) so no 64-bit index division, and every window load is issued up front from a clamped address
// (select, not branch) so a lane has K*K (fwd) or 2*ceil(K/S)^2 (undo) independent 16-byte loads in flight.
// Same visiting order as the generic kernels, so the fp32 sums are bit-identical to them.
template <bool MAX, int K, int S>
__global__ void __launch_bounds__(256) pool_fwd_fixed_kernel(const float* __restrict__ in, float* __restrict__ out, PoolGeo g, float st,
                                                             float so) {
  int bx, oy, c;
  if (!pool_block(g, bx, oy, c)) return;
  const int j = bx * 256 + threadIdx.x;
  if (j >= g.Mx * g.nvec) return;
  const int ox = j / g.nvec, n = 4 * (j - ox * g.nvec);
  const int ys = oy * S + g.py, xs = ox * S + g.px;
  const float* plane = in + (size_t)c * g.H * g.W * g.N + n;
  f32x4 v[K][K];
  bool ok[K][K];
#pragma unroll
  for (int dy = 0; dy < K; ++dy)
#pragma unroll
    for (int dx = 0; dx < K; ++dx) {
      const int y = ys + dy, x = xs + dx;
      ok[dy][dx] = y >= 0 && y < g.H && x >= 0 && x < g.W;
      const size_t o = ok[dy][dx] ? ((size_t)y * g.W + x) * g.N : 0;
      v[dy][dx] = *reinterpret_cast<const f32x4*>(plane + o);
    }
  f32x4 acc = MAX ? f32x4{-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX} : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int dy = 0; dy < K; ++dy)
#pragma unroll
    for (int dx = 0; dx < K; ++dx)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (MAX) acc[e] = (ok[dy][dx] && acc[e] < v[dy][dx][e]) ? v[dy][dx][e] : acc[e];
        else acc[e] = ok[dy][dx] ? acc[e] + v[dy][dx][e] : acc[e];
      }
  if (!MAX) {
    const int y0 = max(0, ys), y1 = min(g.H, ys + K), x0 = max(0, xs), x1 = min(g.W, xs + K);
    const float cnt = (float)((y1 - y0) * (x1 - x0));
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = acc[e] / cnt;
  }
  float* op = out + ((size_t)(c * g.My + oy) * g.Mx + ox) * g.N + n;
  f32x4 res = so * acc;
  if (st != 0.f) res = st * *reinterpret_cast<const f32x4*>(op) + res;
  *reinterpret_cast<f32x4*>(op) = res;
}

// 3 x 3 stride-2 max pooling on 2 x 2 OUTPUT blocks: the four windows of a block cover 5 x 5 inputs, 25 loads for four outputs instead
// of 36 (the per-output kernel above; the vector-memory path, not HBM, is what a pooling kernel saturates first — the same trade as
// pool_undo_max32_block_kernel).  A maximum is a selection: same bits whatever the visiting order.
__global__ void __launch_bounds__(256) pool_fwd_max32_block_kernel(const float* __restrict__ in, float* __restrict__ out, PoolGeo g, float st, float so) {
  int bx, by, c;
  if (!pool_block(g, bx, by, c)) return;
  const int MB = (g.Mx + 1) >> 1;
  const int j = bx * 256 + threadIdx.x;
  if (j >= MB * g.nvec) return;
  const int xb = j / g.nvec, n = 4 * (j - xb * g.nvec);
  const int oy0 = 2 * by, ox0 = 2 * xb;
  const int ys = oy0 * 2 + g.py, xs = ox0 * 2 + g.px;
  const float* plane = in + (size_t)c * g.H * g.W * g.N + n;
  f32x4 v[5][5];
  bool ok[5][5];
#pragma unroll
  for (int dy = 0; dy < 5; ++dy)
#pragma unroll
    for (int dx = 0; dx < 5; ++dx) {
      const int y = ys + dy, x = xs + dx;
      ok[dy][dx] = y >= 0 && y < g.H && x >= 0 && x < g.W;
      const size_t o = ok[dy][dx] ? ((size_t)y * g.W + x) * g.N : 0;
      v[dy][dx] = *reinterpret_cast<const f32x4*>(plane + o);
    }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int oy = oy0 + a, ox = ox0 + b;
      if (oy >= g.My || ox >= g.Mx) continue;
      f32x4 acc = {-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc[e] = (ok[2 * a + dy][2 * b + dx] && acc[e] < v[2 * a + dy][2 * b + dx][e]) ? v[2 * a + dy][2 * b + dx][e] : acc[e];
      float* op = out + ((size_t)(c * g.My + oy) * g.Mx + ox) * g.N + n;
      f32x4 res = so * acc;
      if (st != 0.f) res = st * *reinterpret_cast<const f32x4*>(op) + res;
      *reinterpret_cast<f32x4*>(op) = res;
    }
}

template <bool MAX, int K, int S>
__global__ void __launch_bounds__(256) pool_undo_fixed_kernel(const float* __restrict__ images, const float* __restrict__ grads,
                                                              const float* __restrict__ acts, float* __restrict__ out, PoolGeo g, float st,
                                                              bool relu_mask) {
  constexpr int CV = (K + S - 1) / S;   // pooled coordinates that can cover one input coordinate
  int bx, iy, c;
  if (!pool_block(g, bx, iy, c)) return;
  const int j = bx * 256 + threadIdx.x;
  if (j >= g.W * g.nvec) return;
  const int ix = j / g.nvec, n = 4 * (j - ix * g.nvec);
  const size_t t = ((size_t)(c * g.H + iy) * g.W + ix) * g.N + n;
  f32x4 img = {0.f, 0.f, 0.f, 0.f};
  if (MAX) img = *reinterpret_cast<const f32x4*>(images + t);
  const int ty = iy - g.py, tx = ix - g.px;
  const int oyh = ty >= 0 ? ty / S : -1, oxh = tx >= 0 ? tx / S : -1;
  const size_t pplane = (size_t)c * g.My * g.Mx * g.N + n;
  f32x4 gv[CV][CV], av[CV][CV];
  float wt[CV][CV];
  bool ok[CV][CV];
#pragma unroll
  for (int a = 0; a < CV; ++a) {
    const int oy = oyh - (CV - 1 - a);
    const bool vy = oy >= 0 && oy < g.My && oy * S + K > ty;
    const float ry = fminf((float)g.H, (float)(g.py + oy * S + K)) - fmaxf(0.f, (float)(g.py + oy * S));
#pragma unroll
    for (int b = 0; b < CV; ++b) {
      const int ox = oxh - (CV - 1 - b);
      ok[a][b] = vy && ox >= 0 && ox < g.Mx && ox * S + K > tx;
      const size_t o = pplane + (ok[a][b] ? ((size_t)oy * g.Mx + ox) * g.N : 0);
      gv[a][b] = *reinterpret_cast<const f32x4*>(grads + o);
      if (MAX) av[a][b] = *reinterpret_cast<const f32x4*>(acts + o);
      else {
        const float rx = fminf((float)g.W, (float)(g.px + ox * S + K)) - fmaxf(0.f, (float)(g.px + ox * S));
        wt[a][b] = 1.0f / (rx * ry);
      }
    }
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a < CV; ++a)
#pragma unroll
    for (int b = 0; b < CV; ++b)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (MAX) acc[e] += (ok[a][b] && img[e] == av[a][b][e]) ? gv[a][b][e] : 0.f;
        else acc[e] = ok[a][b] ? acc[e] + gv[a][b][e] * wt[a][b] : acc[e];
      }
  if (st != 0.f) acc = st * *reinterpret_cast<const f32x4*>(out + t) + acc;
  if (MAX && relu_mask) {
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = img[e] > 0.f ? acc[e] : 0.f;
  }
  *reinterpret_cast<f32x4*>(out + t) = acc;
}

// 3 x 3 stride-2 max-pool undo, one thread per 2 x 2 block of input pixels (x 4 images): the four pixels of a block are covered by
// the SAME two pooled rows and two pooled columns (pixel tx is covered by ox with 2*ox <= tx < 2*ox + 3: the pair (tx0, tx0 + 1)
// always meets {c, c + 1}, c = (tx0 + 1)/2 - 1), so 4 gradient + 4 activation loads serve four outputs instead of sixteen each
// — 3 loads per output instead of 9.  Same candidate order per pixel as pool_undo_fixed_kernel (pooled row ascending, then column;
// a window that does not cover the pixel adds 0.f), so the sums are bit-identical to it.
__global__ void __launch_bounds__(256) pool_undo_max32_block_kernel(const float* __restrict__ images, const float* __restrict__ grads,
                                                                    const float* __restrict__ acts, float* __restrict__ out, PoolGeo g, float st,
                                                                    bool relu_mask) {
  int bx, by, c;
  if (!pool_block(g, bx, by, c)) return;
  const int WB = (g.W + 1) >> 1;
  const int j = bx * 256 + threadIdx.x;
  if (j >= WB * g.nvec) return;
  const int xb = j / g.nvec, n = 4 * (j - xb * g.nvec);
  const int iy0 = 2 * by, ix0 = 2 * xb;
  const int ty0 = iy0 - g.py, tx0 = ix0 - g.px;   // >= 0: py, px are the negated paddings
  const int oy0 = (ty0 + 1) / 2 - 1, ox0 = (tx0 + 1) / 2 - 1;
  const size_t pplane = (size_t)c * g.My * g.Mx * g.N + n;
  f32x4 gv[2][2], av[2][2];
  bool pok[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int oy = oy0 + a, ox = ox0 + b;
      pok[a][b] = oy >= 0 && oy < g.My && ox >= 0 && ox < g.Mx;
      const size_t o = pplane + (pok[a][b] ? ((size_t)oy * g.Mx + ox) * g.N : 0);
      gv[a][b] = *reinterpret_cast<const f32x4*>(grads + o);
      av[a][b] = *reinterpret_cast<const f32x4*>(acts + o);
    }
  f32x4 img[2][2];
  bool iok[2][2];
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      iok[dy][dx] = iy0 + dy < g.H && ix0 + dx < g.W;
      const size_t t = ((size_t)(c * g.H + (iok[dy][dx] ? iy0 + dy : iy0)) * g.W + (iok[dy][dx] ? ix0 + dx : ix0)) * g.N + n;
      img[dy][dx] = *reinterpret_cast<const f32x4*>(images + t);
    }
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      if (!iok[dy][dx]) continue;
      const int ty = ty0 + dy, tx = tx0 + dx;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int oy = oy0 + a, ox = ox0 + b;
          const bool cov = pok[a][b] && 2 * oy <= ty && ty < 2 * oy + 3 && 2 * ox <= tx && tx < 2 * ox + 3;
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] += (cov && img[dy][dx][e] == av[a][b][e]) ? gv[a][b][e] : 0.f;
        }
      float* op = out + ((size_t)(c * g.H + iy0 + dy) * g.W + ix0 + dx) * g.N + n;
      if (st != 0.f) acc = st * *reinterpret_cast<const f32x4*>(op) + acc;
      if (relu_mask) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = img[dy][dx][e] > 0.f ? acc[e] : 0.f;
      }
      *reinterpret_cast<f32x4*>(op) = acc;
    }
}

// ---- 3 x 3 stride-2 max pooling with a WINDOW MASK (round 6; MaxPoolMask / MaxPoolUndoMask, include/convnet_hip.h) ---------------------
// The undo of a max pooling needs the layer's input only to find which inputs equal their window's maximum (every tie counts,
// cudamat_conv_gemm.cu:220-262): for AlexNet's pool1 that is a 1.19 GB tensor read again, plus the 0.3 GB of maxima, to route 0.3 GB of
// derivatives.  The forward pass has both in registers: it records, per pooled element, a 16-bit mask — bit 3*dy + dx set when input
// (dy, dx) of the window is inside the image and EQUAL to the maximum (the same float == the undo kernels apply), bit 9 set when the
// maximum is positive (what the fused ReLU' of the layer below asks of a matching input: x == max and x > 0  <=>  x == max and max > 0).
// The undo then reads masks (2 B per pooled element) and derivatives only: pool1 2.98 -> 1.64 GB.  Same 2 x 2 blocks, same candidate
// order per pixel as pool_fwd_max32_block_kernel / pool_undo_max32_block_kernel: results bit-identical to MaxPool + MaxPoolUndo(Relu).
// Mask layout: uint16 [c][oy][ox][image] (the pooled tensor's own order), four images = one 8-byte store per lane.
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;

__global__ void __launch_bounds__(256) pool_fwd_max32_mask_kernel(const float* __restrict__ in, float* __restrict__ out, u32x2* __restrict__ mask, PoolGeo g) {
  int bx, by, c;
  if (!pool_block(g, bx, by, c)) return;
  const int MB = (g.Mx + 1) >> 1;
  const int j = bx * 256 + threadIdx.x;
  if (j >= MB * g.nvec) return;
  const int xb = j / g.nvec, n = 4 * (j - xb * g.nvec);
  const int oy0 = 2 * by, ox0 = 2 * xb;
  const int ys = oy0 * 2 + g.py, xs = ox0 * 2 + g.px;
  const float* plane = in + (size_t)c * g.H * g.W * g.N + n;
  f32x4 v[5][5];
  bool ok[5][5];
#pragma unroll
  for (int dy = 0; dy < 5; ++dy)
#pragma unroll
    for (int dx = 0; dx < 5; ++dx) {
      const int y = ys + dy, x = xs + dx;
      ok[dy][dx] = y >= 0 && y < g.H && x >= 0 && x < g.W;
      const size_t o = ok[dy][dx] ? ((size_t)y * g.W + x) * g.N : 0;
      v[dy][dx] = *reinterpret_cast<const f32x4*>(plane + o);
    }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int oy = oy0 + a, ox = ox0 + b;
      if (oy >= g.My || ox >= g.Mx) continue;
      f32x4 acc = {-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc[e] = (ok[2 * a + dy][2 * b + dx] && acc[e] < v[2 * a + dy][2 * b + dx][e]) ? v[2 * a + dy][2 * b + dx][e] : acc[e];
      unsigned m[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        m[e] = acc[e] > 0.f ? 1u << 9 : 0u;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx)
            m[e] |= (ok[2 * a + dy][2 * b + dx] && v[2 * a + dy][2 * b + dx][e] == acc[e]) ? 1u << (3 * dy + dx) : 0u;
      }
      const size_t o = ((size_t)(c * g.My + oy) * g.Mx + ox) * g.N + n;
      *reinterpret_cast<f32x4*>(out + o) = acc;
      mask[o >> 2] = u32x2{m[0] | (m[1] << 16), m[2] | (m[3] << 16)};
    }
}

__global__ void __launch_bounds__(256) pool_undo_max32_mask_kernel(const float* __restrict__ grads, const u32x2* __restrict__ mask, float* __restrict__ out,
                                                                   PoolGeo g, float st, bool relu) {
  int bx, by, c;
  if (!pool_block(g, bx, by, c)) return;
  const int WB = (g.W + 1) >> 1;
  const int j = bx * 256 + threadIdx.x;
  if (j >= WB * g.nvec) return;
  const int xb = j / g.nvec, n = 4 * (j - xb * g.nvec);
  const int iy0 = 2 * by, ix0 = 2 * xb;
  const int ty0 = iy0 - g.py, tx0 = ix0 - g.px;   // >= 0: py, px are the negated paddings
  const int oy0 = (ty0 + 1) / 2 - 1, ox0 = (tx0 + 1) / 2 - 1;
  const size_t pplane = (size_t)c * g.My * g.Mx * g.N + n;
  f32x4 gv[2][2];
  unsigned mk[2][2][4];   // per image: the window's mask, reduced to "counts at all" by bit 9 when the ReLU' is fused
  bool pok[2][2];
  const unsigned need = relu ? 1u << 9 : 0u;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int oy = oy0 + a, ox = ox0 + b;
      pok[a][b] = oy >= 0 && oy < g.My && ox >= 0 && ox < g.Mx;
      const size_t o = pplane + (pok[a][b] ? ((size_t)oy * g.Mx + ox) * g.N : 0);
      gv[a][b] = *reinterpret_cast<const f32x4*>(grads + o);
      const u32x2 w = mask[o >> 2];
      const unsigned m4[4] = {w[0] & 0xffffu, w[0] >> 16, w[1] & 0xffffu, w[1] >> 16};
#pragma unroll
      for (int e = 0; e < 4; ++e) mk[a][b][e] = (m4[e] & need) == need ? m4[e] : 0u;
    }
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      if (!(iy0 + dy < g.H && ix0 + dx < g.W)) continue;
      const int ty = ty0 + dy, tx = tx0 + dx;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int oy = oy0 + a, ox = ox0 + b;
          const bool cov = pok[a][b] && 2 * oy <= ty && ty < 2 * oy + 3 && 2 * ox <= tx && tx < 2 * ox + 3;
          const unsigned bit = cov ? 1u << (3 * (ty - 2 * oy) + (tx - 2 * ox)) : 0u;
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] += (mk[a][b][e] & bit) ? gv[a][b][e] : 0.f;
        }
      float* op = out + ((size_t)(c * g.H + iy0 + dy) * g.W + ix0 + dx) * g.N + n;
      if (st != 0.f) acc = st * *reinterpret_cast<const f32x4*>(op) + acc;
      *reinterpret_cast<f32x4*>(op) = acc;
    }
}

// ---- cross-map response norm --------------------------------------------------------------------------
// One lane owns 4 consecutive "locations" (a location = one (pixel, image); locations are
// contiguous in memory) and walks the channel axis with the reference's sliding-window update
// (subtract the channel that leaves, add the one that enters: cpumat_conv.cc:476-490), so the
// fp32 rounding sequence per location is the oracle's.
__device__ __forceinline__ void win_fwd(int j, int C, int sizeF, bool blocked, int& start, int& end) {
  start = blocked ? (j / sizeF) * sizeF : -sizeF / 2 + j;
  end = min(C, start + sizeF);
  start = max(0, start);
}
__device__ __forceinline__ void win_bwd(int j, int C, int sizeF, bool blocked, int& start, int& end) {
  start = blocked ? (j / sizeF) * sizeF : -sizeF + sizeF / 2 + j + 1;
  end = min(C, start + sizeF);
  start = max(0, start);
}

__global__ void rnorm_fwd_kernel(const float* __restrict__ in, float* __restrict__ out, size_t locs, int C, int sizeF, float addScale,
                                 float powScale, bool blocked, bool vec, int cseg, int nseg, bool relu) {
  const size_t nq = (locs + 3) >> 2;
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq * nseg; q += (size_t)gridDim.x * blockDim.x) {
    const int seg = (int)(q / nq);
    const size_t l = (q % nq) << 2;
    const int j0 = seg * cseg, j1 = min(C, j0 + cseg);
    const int rem = (int)min((size_t)4, locs - l);
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    int ps, pe;
    {
      int e0;
      win_fwd(j0, C, sizeF, blocked, ps, e0);
      pe = ps;   // the first iteration adds the whole window of j0 directly
    }
    for (int j = j0; j < j1; ++j) {
      int s, e;
      win_fwd(j, C, sizeF, blocked, s, e);
      for (int i = ps; i < s; ++i) {
        const f32x4 v = ldv(in + (size_t)i * locs + l, 0, rem, vec);
        sum = sum - v * v;
      }
      for (int i = pe; i < e; ++i) {
        const f32x4 v = ldv(in + (size_t)i * locs + l, 0, rem, vec);
        sum = sum + v * v;
      }
      const f32x4 x = ldv(in + (size_t)j * locs + l, 0, rem, vec);
      f32x4 y;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        y[t] = x[t] * powf(1.f + addScale * sum[t], -powScale);
        if (relu) y[t] = fmaxf(y[t], 0.f);
      }
      stv(out + (size_t)j * locs + l, y, 0, rem, vec);
      ps = s;
      pe = e;
    }
  }
}

// pass 1 of undo: den = (1+a*sum)^(-b-1); prod = dout*in*den; scaled = dout*den^(b/(b+1))
__global__ void rnorm_undo1_kernel(const float* __restrict__ dout, const float* __restrict__ in, float* __restrict__ prod,
                                   float* __restrict__ scaled, size_t locs, int C, int sizeF, float addScale, float powScale, bool blocked,
                                   bool vec, int cseg, int nseg) {
  const size_t nq = (locs + 3) >> 2;
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq * nseg; q += (size_t)gridDim.x * blockDim.x) {
    const int seg = (int)(q / nq);
    const size_t l = (q % nq) << 2;
    const int j0 = seg * cseg, j1 = min(C, j0 + cseg);
    const int rem = (int)min((size_t)4, locs - l);
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    int ps, pe;
    {
      int e0;
      win_fwd(j0, C, sizeF, blocked, ps, e0);
      pe = ps;   // the first iteration adds the whole window of j0 directly
    }
    for (int j = j0; j < j1; ++j) {
      int s, e;
      win_fwd(j, C, sizeF, blocked, s, e);
      for (int i = ps; i < s; ++i) {
        const f32x4 v = ldv(in + (size_t)i * locs + l, 0, rem, vec);
        sum = sum - v * v;
      }
      for (int i = pe; i < e; ++i) {
        const f32x4 v = ldv(in + (size_t)i * locs + l, 0, rem, vec);
        sum = sum + v * v;
      }
      const f32x4 x = ldv(in + (size_t)j * locs + l, 0, rem, vec);
      const f32x4 d = ldv(dout + (size_t)j * locs + l, 0, rem, vec);
      f32x4 pr, sc;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float den = powf(1.f + addScale * sum[t], -powScale - 1.f);
        pr[t] = d[t] * x[t] * den;
        sc[t] = d[t] * powf(den, powScale / (powScale + 1.f));
      }
      stv(prod + (size_t)j * locs + l, pr, 0, rem, vec);
      stv(scaled + (size_t)j * locs + l, sc, 0, rem, vec);
      ps = s;
      pe = e;
    }
  }
}

// pass 2: target_j = scaled_j - 2ab * in_j * sum_{i in win^-1(j)} prod_i
__global__ void rnorm_undo2_kernel(const float* __restrict__ in, const float* __restrict__ prod, const float* __restrict__ scaled,
                                   float* __restrict__ out, size_t locs, int C, int sizeF, float addScale, float powScale, bool blocked,
                                   bool vec, int cseg, int nseg) {
  const size_t nq = (locs + 3) >> 2;
  const float k2 = 2 * addScale * powScale;
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq * nseg; q += (size_t)gridDim.x * blockDim.x) {
    const int seg = (int)(q / nq);
    const size_t l = (q % nq) << 2;
    const int j0 = seg * cseg, j1 = min(C, j0 + cseg);
    const int rem = (int)min((size_t)4, locs - l);
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    int ps, pe;
    {
      int e0;
      win_bwd(j0, C, sizeF, blocked, ps, e0);
      pe = ps;   // the first iteration adds the whole window of j0 directly
    }
    for (int j = j0; j < j1; ++j) {
      int s, e;
      win_bwd(j, C, sizeF, blocked, s, e);
      for (int i = ps; i < s; ++i) sum = sum - ldv(prod + (size_t)i * locs + l, 0, rem, vec);
      for (int i = pe; i < e; ++i) sum = sum + ldv(prod + (size_t)i * locs + l, 0, rem, vec);
      const f32x4 x = ldv(in + (size_t)j * locs + l, 0, rem, vec);
      const f32x4 sc = ldv(scaled + (size_t)j * locs + l, 0, rem, vec);
      f32x4 y;
#pragma unroll
      for (int t = 0; t < 4; ++t) y[t] = sc[t] - k2 * x[t] * sum[t];
      stv(out + (size_t)j * locs + l, y, 0, rem, vec);
      ps = s;
      pe = e;
    }
  }
}

// ---- LDS-tiled response norm: read-once / write-once ------------------------------------------------
// A block stages ALL channels of LT consecutive locations in LDS (C x LT floats; every row is a
// contiguous 4*LT-byte run of the tensor), lanes = (location, channel group) slide their windows over
// LDS, so HBM sees exactly the algorithmic bytes: 8 B/element forward, 12 B/element backward (the global-
// memory walkers above re-read each element ~3x through L2 and, backward, round-trip two temporaries).
template <int LT>
__device__ __forceinline__ void rn_stage(const float* __restrict__ g, float* __restrict__ s, size_t locs, size_t l0, int C, bool vec) {
  constexpr int L4 = LT / 4;
  for (int idx = threadIdx.x; idx < C * L4; idx += blockDim.x) {
    const int c = idx / L4, q = idx - c * L4;
    const size_t l = l0 + 4 * q;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (l < locs) v = ldv(g + (size_t)c * locs + l, 0, (int)min((size_t)4, locs - l), vec);
    *reinterpret_cast<f32x4*>(s + c * LT + 4 * q) = v;
  }
}

// Block -> location-tile map of the LDS response-norm kernels.  A tile is LT consecutive locations = LT*4 bytes per channel row;
// with LT = 16 that is HALF a 128-byte line, and the hardware puts consecutive blocks on different XCDs (block b on XCD b%8), so
// neighbouring tiles fetched every line into two L2s (measured on rnorm1's undo: 1.19 GB fetched for 0.59 GB of input).  Each XCD
// therefore takes a contiguous run of tiles.  `tiles` = real tile count; the grid is rounded up to a multiple of 8.
__device__ __forceinline__ long rn_tile(unsigned tiles, bool xcd) {
  if (!xcd) return blockIdx.x < tiles ? (long)blockIdx.x : -1;
  const unsigned per = (tiles + 7) >> 3;
  const unsigned L = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  return L < tiles ? (long)L : -1;
}

template <int LT>
__global__ void rnorm_fwd_lds_kernel(const float* __restrict__ in, float* __restrict__ out, size_t locs, int C, int sizeF, float addScale,
                                     float powScale, bool blocked, bool vec, bool relu, unsigned tiles, bool xcd) {
  extern __shared__ __attribute__((aligned(16))) float rn_smem[];
  float* xs = rn_smem;   // [C][LT]
  const long tile = rn_tile(tiles, xcd);
  if (tile < 0) return;
  const size_t l0 = (size_t)tile * LT;
  rn_stage<LT>(in, xs, locs, l0, C, vec);
  __syncthreads();
  const int l = threadIdx.x % LT, g = threadIdx.x / LT, G = blockDim.x / LT;
  const int cg = (C + G - 1) / G, j0 = g * cg, j1 = min(C, j0 + cg);
  if (l0 + l >= locs || j0 >= j1) return;
  int ps, pe;
  {
    int e0;
    win_fwd(j0, C, sizeF, blocked, ps, e0);
    pe = ps;
  }
  float sum = 0.f;
  for (int j = j0; j < j1; ++j) {
    int s, e;
    win_fwd(j, C, sizeF, blocked, s, e);
    for (int i = ps; i < s; ++i) { const float v = xs[i * LT + l]; sum -= v * v; }
    for (int i = pe; i < e; ++i) { const float v = xs[i * LT + l]; sum += v * v; }
    // u^(-b) = exp2(-b * log2(u)), u >= 1: two quarter-rate transcendentals instead of ~100 VALU of powf
    // (the reference's own GPU path uses __powf, cudamat_conv_gemm.cu:458); relative error ~1e-6.
    const float y = xs[j * LT + l] * exp2f(-powScale * __log2f(1.f + addScale * sum));
    out[(size_t)j * locs + l0 + l] = relu ? fmaxf(y, 0.f) : y;
    ps = s;
    pe = e;
  }
}

template <int LT>
__global__ void __launch_bounds__(512) rnorm_undo_lds_kernel(const float* __restrict__ dout, const float* __restrict__ in, float* __restrict__ out, size_t locs, int C,
                                      int sizeF, float addScale, float powScale, bool blocked, bool vec, unsigned tiles, bool xcd) {
  extern __shared__ __attribute__((aligned(16))) float rn_smem[];
  float* xs = rn_smem;            // [C][LT] inputs
  float* ds = xs + C * LT;        // [C][LT] out-grads, then "scaled" = dout * den^(b/(b+1))
  float* ps_ = ds + C * LT;       // [C][LT] prod   = dout * in * den,  den = (1 + a*S)^(-b-1)
  const long tile = rn_tile(tiles, xcd);
  if (tile < 0) return;
  const size_t l0 = (size_t)tile * LT;
  rn_stage<LT>(in, xs, locs, l0, C, vec);
  rn_stage<LT>(dout, ds, locs, l0, C, vec);
  __syncthreads();
  const int l = threadIdx.x % LT, g = threadIdx.x / LT, G = blockDim.x / LT;
  const int cg = (C + G - 1) / G, j0 = g * cg, j1 = min(C, j0 + cg);
  const bool live = l0 + l < locs && j0 < j1;
  if (live) {
    int ps, pe;
    {
      int e0;
      win_fwd(j0, C, sizeF, blocked, ps, e0);
      pe = ps;
    }
    float sum = 0.f;
    for (int j = j0; j < j1; ++j) {
      int s, e;
      win_fwd(j, C, sizeF, blocked, s, e);
      for (int i = ps; i < s; ++i) { const float v = xs[i * LT + l]; sum -= v * v; }
      for (int i = pe; i < e; ++i) { const float v = xs[i * LT + l]; sum += v * v; }
      const float lg = __log2f(1.f + addScale * sum);
      const float den = exp2f((-powScale - 1.f) * lg);          // (1 + a*S)^(-b-1)
      const float d = ds[j * LT + l];
      ps_[j * LT + l] = d * xs[j * LT + l] * den;
      ds[j * LT + l] = d * exp2f(-powScale * lg);               // d * den^(b/(b+1)) = d * (1 + a*S)^(-b)
      ps = s;
      pe = e;
    }
  }
  __syncthreads();
  if (!live) return;
  const float k2 = 2 * addScale * powScale;
  int ps, pe;
  {
    int e0;
    win_bwd(j0, C, sizeF, blocked, ps, e0);
    pe = ps;
  }
  float sum = 0.f;
  for (int j = j0; j < j1; ++j) {
    int s, e;
    win_bwd(j, C, sizeF, blocked, s, e);
    for (int i = ps; i < s; ++i) sum -= ps_[i * LT + l];
    for (int i = pe; i < e; ++i) sum += ps_[i * LT + l];
    out[(size_t)j * locs + l0 + l] = ds[j * LT + l] - k2 * xs[j * LT + l] * sum;
    ps = s;
    pe = e;
  }
}

// ---- the AlexNet form of the two kernels (round 6): window and segment known at compile time, pipelined ----------------------------------
// AlexNet's windows are a quarter of the channels (frac_of_filters_response_norm 0.25: 24 of 96, 64 of 256).  Measured on rnorm1's undo,
// the LDS-tiled kernels above are ISSUE-bound, not latency-bound: ~130 instructions per element (window bounds, two variable-length
// slide loops and their LDS addresses per channel, a 24- or 64-term first window per 6 or 8 channels, the scalar tail paths of ldv);
// giving that kernel loads in flight throughout (a persistent block that fetches its next tile into registers) moved it by 10 % only.
// Here, for an even window SZ with SZ/2 a multiple of the CG channels a lane owns:
//   * the channel loop is unrolled and the slides are two LDS reads at compile-time offsets, in the oracle's order (subtract the channel
//     that leaves, add the one that enters, cpumat_conv.cc:476-490).  Rows of zeros below channel 0 and above channel C - 1 stand in for
//     the clipped window ends (x - 0*0 and x + 0*0 are exact): no bounds anywhere;
//   * a lane's FIRST window is the sum of SZ / CG segment sums — every lane squares and adds its own CG channels once and leaves the sum
//     in LDS — instead of SZ terms per lane (a different association of the same terms; the slides start from it);
//   * the block is persistent over a strided run of its XCD's tiles and fetches the NEXT tile into registers right after the current one
//     has been put into LDS (loads in flight during the whole compute and write-out), from per-lane pointers set up once;
//   * results leave through LDS as 16-byte vectors: a quarter of the store instructions;
//   * when C is not a multiple of CG the last lanes' spare channels read the zero rows, compute zeros and write them back into zero rows
//     (the arrays are CG rows longer than C); only rows < C are copied out.
// Launch: blockDim.x = ceil(C / CG) * LT, gridDim.x = 8 * (blocks per XCD); 16-byte rows (vec), !blocked only.
template <int NV>
struct RnLanes {          // this lane's 16-byte pieces of a [C][LT] tile: offset in the tensor (channel * locs + 4 * quad), or -1
  long long off[NV];
  int last_q[NV];         // quad index inside the tile (for the ragged last tile)
};
template <int NV>
__device__ __forceinline__ RnLanes<NV> rn_lanes(size_t locs, int C, int L4) {
  RnLanes<NV> r;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int idx = threadIdx.x + k * blockDim.x;
    const int c = idx / L4, q = idx - c * L4;
    r.off[k] = c < C ? (long long)((size_t)c * locs + 4 * q) : -1;
    r.last_q[k] = 4 * q;
  }
  return r;
}
template <int NV>
__device__ __forceinline__ void rn_fetch(const float* __restrict__ g, const RnLanes<NV>& ln, f32x4 (&r)[NV], size_t locs, size_t l0) {
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (ln.off[k] >= 0 && l0 + ln.last_q[k] < locs) v = *reinterpret_cast<const f32x4*>(g + ln.off[k] + l0);
    r[k] = v;
  }
}
template <int NV>
__device__ __forceinline__ void rn_put(float* __restrict__ s, const f32x4 (&r)[NV], int n4) {
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int idx = threadIdx.x + k * blockDim.x;   // element 4 * idx of the [C][LT] array
    if (idx < n4) *reinterpret_cast<f32x4*>(s + 4 * idx) = r[k];
  }
}
template <int NV>
__device__ __forceinline__ void rn_copy_out(const float* __restrict__ s, float* __restrict__ out, const RnLanes<NV>& ln, size_t locs, size_t l0) {
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int idx = threadIdx.x + k * blockDim.x;
    if (ln.off[k] >= 0 && l0 + ln.last_q[k] < locs) *reinterpret_cast<f32x4*>(out + ln.off[k] + l0) = *reinterpret_cast<const f32x4*>(s + 4 * idx);
  }
}
struct RnRun {
  unsigned t, hi, step;
};
__device__ __forceinline__ RnRun rn_run(unsigned tiles) {
  const unsigned per = (tiles + 7) >> 3, x = blockIdx.x & 7, lo = x * per;
  return {lo + (blockIdx.x >> 3), min(tiles, lo + per), gridDim.x >> 3};
}
// zero rows below channel 0 / from channel C on (the spare channels of the last lanes and everything a window reaches above them)
constexpr int rn_halo_lo(int SZ) { return SZ / 2; }
constexpr int rn_halo_hi(int CG, int SZ) { return CG + SZ / 2 + 1; }
// 2^x by the hardware instruction alone (v_exp_f32): exp2f() wraps it in a range fix for results below 2^-126, which a power
// (1 + a*S)^(-b) only reaches for sums of squares near the top of the fp32 range
__device__ __forceinline__ float rn_exp2(float x) {
#ifdef CONVNET_EMU
  return exp2f(x);
#else
  return __builtin_amdgcn_exp2f(x);
#endif
}

// first window of a lane = 2 * NQ sub-segment sums on either side of its first channel; a lane owns NSUB sub-segments of SS channels
template <int CG, int SZ>
struct RnSeg {
  static constexpr int H = SZ / 2, SS = CG < H ? CG : H, NSUB = CG / SS, NQ = H / SS;
  static_assert(SZ % 2 == 0 && CG % SS == 0 && H % SS == 0, "the first window is a whole number of sub-segments on either side");
};

template <int LT, int CG, int SZ>
__global__ void __launch_bounds__(512) rnorm_fwd_fast_kernel(const float* __restrict__ in, float* __restrict__ out, size_t locs, int C, float addScale,
                                                             float powScale, bool relu, unsigned tiles) {
  using Seg = RnSeg<CG, SZ>;
  constexpr int NV = (CG + 3) / 4, L4 = LT / 4, HLO = rn_halo_lo(SZ), HHI = rn_halo_hi(CG, SZ), H = Seg::H, SS = Seg::SS, NSUB = Seg::NSUB, NQ = Seg::NQ;
  extern __shared__ __attribute__((aligned(16))) float rn_smem[];
  const int G = blockDim.x / LT;
  float* xs = rn_smem + HLO * LT;                              // rows -HLO .. C + HHI - 1
  float* ys = rn_smem + (HLO + C + HHI) * LT;                  // rows 0 .. C + CG - 1
  float* sx = ys + (C + CG) * LT + NQ * LT;                    // sub-segment sums of x^2: rows -NQ .. G * NSUB + NQ - 1
  RnRun run = rn_run(tiles);
  if (run.t >= run.hi) return;
  for (int i = threadIdx.x; i < HLO * LT; i += blockDim.x) xs[i - HLO * LT] = 0.f;
  for (int i = threadIdx.x; i < HHI * LT; i += blockDim.x) xs[C * LT + i] = 0.f;
  for (int i = threadIdx.x; i < NQ * LT; i += blockDim.x) sx[i - NQ * LT] = sx[G * NSUB * LT + i] = 0.f;
  const int l = threadIdx.x % LT, g = threadIdx.x / LT, j0 = g * CG;
  const RnLanes<NV> ln = rn_lanes<NV>(locs, C, L4);
  f32x4 rx[NV];
  rn_fetch<NV>(in, ln, rx, locs, (size_t)run.t * LT);
  for (; run.t < run.hi; run.t += run.step) {
    const size_t l0 = (size_t)run.t * LT;
    rn_put<NV>(xs, rx, C * L4);
    __syncthreads();
    if (run.t + run.step < run.hi) rn_fetch<NV>(in, ln, rx, locs, (size_t)(run.t + run.step) * LT);
    float xo[CG];
#pragma unroll
    for (int u = 0; u < NSUB; ++u) {
      float seg = 0.f;
#pragma unroll
      for (int c = u * SS; c < (u + 1) * SS; ++c) {
        xo[c] = xs[(j0 + c) * LT + l];
        seg += xo[c] * xo[c];
      }
      sx[(g * NSUB + u) * LT + l] = seg;
    }
    __syncthreads();
    float sum = 0.f;   // window of channel j0: [j0 - H, j0 + H)
#pragma unroll
    for (int q = -NQ; q < NQ; ++q) sum += sx[(g * NSUB + q) * LT + l];
#pragma unroll
    for (int c = 0; c < CG; ++c) {
      if (c > 0) {
        const float a = xs[(j0 + c - 1 - H) * LT + l], b = xs[(j0 + c - 1 + H) * LT + l];
        sum -= a * a;
        sum += b * b;
      }
      // u^(-b) = exp2(-b * log2(u)), u >= 1 (see rnorm_fwd_lds_kernel)
      const float y = xo[c] * rn_exp2(-powScale * __log2f(1.f + addScale * sum));
      ys[(j0 + c) * LT + l] = relu ? fmaxf(y, 0.f) : y;
    }
    __syncthreads();
    rn_copy_out<NV>(ys, out, ln, locs, l0);
    // (the next put writes xs, which nobody reads any more; the next tile's sx / ys writes come behind the next barriers)
  }
}

template <int LT, int CG, int SZ>
__global__ void __launch_bounds__(512) rnorm_undo_fast_kernel(const float* __restrict__ dout, const float* __restrict__ in, float* __restrict__ out,
                                                              size_t locs, int C, float addScale, float powScale, unsigned tiles) {
  using Seg = RnSeg<CG, SZ>;
  constexpr int NV = (CG + 3) / 4, L4 = LT / 4, HLO = rn_halo_lo(SZ), HHI = rn_halo_hi(CG, SZ), H = Seg::H, SS = Seg::SS, NSUB = Seg::NSUB, NQ = Seg::NQ;
  extern __shared__ __attribute__((aligned(16))) float rn_smem[];
  const int G = blockDim.x / LT, R = HLO + C + HHI;
  float* xs = rn_smem + HLO * LT;                 // rows -HLO .. C + HHI - 1: inputs, then the results
  float* ps_ = rn_smem + (R + HLO) * LT;          // rows -HLO .. C + HHI - 1: out-grads, then (element by element, each by the lane that owns it)
                                                  // prod = dout * in * den,  den = (1 + a*S)^(-b-1)
  float* sx = rn_smem + 2 * R * LT + NQ * LT;     // sub-segment sums of x^2: rows -NQ .. G * NSUB + NQ - 1
  float* sp = sx + (G * NSUB + 2 * NQ) * LT;      // ... of prod
  RnRun run = rn_run(tiles);
  if (run.t >= run.hi) return;
  for (int i = threadIdx.x; i < HLO * LT; i += blockDim.x) xs[i - HLO * LT] = ps_[i - HLO * LT] = 0.f;
  for (int i = threadIdx.x; i < HHI * LT; i += blockDim.x) xs[C * LT + i] = ps_[C * LT + i] = 0.f;
  for (int i = threadIdx.x; i < NQ * LT; i += blockDim.x) sx[i - NQ * LT] = sx[G * NSUB * LT + i] = sp[i - NQ * LT] = sp[G * NSUB * LT + i] = 0.f;
  const int l = threadIdx.x % LT, g = threadIdx.x / LT, j0 = g * CG;
  const float k2 = 2 * addScale * powScale;
  const RnLanes<NV> ln = rn_lanes<NV>(locs, C, L4);
  f32x4 rx[NV], rd[NV];
  rn_fetch<NV>(in, ln, rx, locs, (size_t)run.t * LT);
  rn_fetch<NV>(dout, ln, rd, locs, (size_t)run.t * LT);
  for (; run.t < run.hi; run.t += run.step) {
    const size_t l0 = (size_t)run.t * LT;
    rn_put<NV>(xs, rx, C * L4);
    rn_put<NV>(ps_, rd, C * L4);
    __syncthreads();
    if (run.t + run.step < run.hi) {
      rn_fetch<NV>(in, ln, rx, locs, (size_t)(run.t + run.step) * LT);
      rn_fetch<NV>(dout, ln, rd, locs, (size_t)(run.t + run.step) * LT);
    }
    float xo[CG], sc[CG];
#pragma unroll
    for (int u = 0; u < NSUB; ++u) {
      float seg = 0.f;
#pragma unroll
      for (int c = u * SS; c < (u + 1) * SS; ++c) {
        xo[c] = xs[(j0 + c) * LT + l];
        seg += xo[c] * xo[c];
      }
      sx[(g * NSUB + u) * LT + l] = seg;
    }
    __syncthreads();
    {
      float sum = 0.f, pseg = 0.f;   // forward window of channel j0: [j0 - H, j0 + H)
#pragma unroll
      for (int q = -NQ; q < NQ; ++q) sum += sx[(g * NSUB + q) * LT + l];
#pragma unroll
      for (int c = 0; c < CG; ++c) {
        if (c > 0) {
          const float a = xs[(j0 + c - 1 - H) * LT + l], b = xs[(j0 + c - 1 + H) * LT + l];
          sum -= a * a;
          sum += b * b;
        }
        const float lg = __log2f(1.f + addScale * sum);
        const float den = rn_exp2((-powScale - 1.f) * lg);        // (1 + a*S)^(-b-1)
        const float d = ps_[(j0 + c) * LT + l];                   // (the spare channels of the last lanes: a zero row)
        const float pr = d * xo[c] * den;
        ps_[(j0 + c) * LT + l] = pr;
        pseg += pr;
        sc[c] = d * rn_exp2(-powScale * lg);                      // d * (1 + a*S)^(-b)
        if ((c + 1) % SS == 0) {
          sp[(g * NSUB + c / SS) * LT + l] = pseg;
          pseg = 0.f;
        }
      }
    }
    __syncthreads();
    {
      float sum = 0.f;   // [j0 - H, j0 + H), then one slide to the backward window of channel j0: [j0 - H + 1, j0 + H + 1)
#pragma unroll
      for (int q = -NQ; q < NQ; ++q) sum += sp[(g * NSUB + q) * LT + l];
#pragma unroll
      for (int c = 0; c < CG; ++c) {
        sum -= ps_[(j0 + c - H) * LT + l];
        sum += ps_[(j0 + c + H) * LT + l];
        xs[(j0 + c) * LT + l] = sc[c] - k2 * xo[c] * sum;   // (every lane has read its inputs: the rows now carry the result)
      }
    }
    __syncthreads();
    rn_copy_out<NV>(xs, out, ln, locs, l0);
    __syncthreads();   // ... before the next tile's inputs overwrite them
  }
}

namespace {

inline bool a16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Channel segmentation of the response-norm walk: a lane slides its window over `cseg` channels only
// (its first window is summed directly), so (location quads) x (segments) lanes are in flight instead of
// one long dependent chain per location.  Segments are kept >= the window so the direct first sum
// costs at most as much as the slide it replaces.
inline void rnorm_segments(size_t quads, int C, int sizeF, int& cseg, int& nseg) {
  const size_t want = size_t(1) << 20;   // ~1M lanes: 4 waves per SIMD on 256 CUs
  int n = (int)((want + quads - 1) / quads);
  int max_n = C / (sizeF > 8 ? sizeF : 8);
  if (max_n < 1) max_n = 1;
  if (n > max_n) n = max_n;
  if (n < 1) n = 1;
  cseg = (C + n - 1) / n;
  nseg = (C + cseg - 1) / cseg;
}

inline int grid_for(size_t items) {
  size_t b = (items + 255) / 256;
  if (b > 4096) b = 4096;
  return b ? (int)b : 1;
}

// Launch geometry of the persistent response-norm kernels: blocks per XCD = twice what its 32 CUs hold at once (the runtime's occupancy
// figure for this kernel, block size and LDS: registers, LDS and wave slots), never more than the XCD has tiles.
inline unsigned rn_pipe_grid(const void* kernel, unsigned tiles, size_t smem, int threads) {
  struct Key { const void* k; size_t smem; int threads; int n; };
  static thread_local Key cache[8] = {};
  int n = 0;
  for (const Key& e : cache)
    if (e.k == kernel && e.smem == smem && e.threads == threads) n = e.n;
  if (!n) {
    if (smem > 64 * 1024) CHIP_CHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    CHIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, threads, smem));
    if (n < 1) n = 1;
    static thread_local int next = 0;
    cache[next++ & 7] = {kernel, smem, threads, n};
  }
  n *= 2;   // twice what is resident: the second half starts as the first blocks finish their (shorter) runs — measured 2-4 % ahead of exactly resident
  if (const int f = CHIP_DIAG_KNOB("CONVNET_RNORM_PIPE_BPC", 0)) n = f;
  const unsigned per = (tiles + 7) / 8;
  return 8 * std::min(per, 32u * (unsigned)n);
}

// Shape of the fast kernels: window 24 (AlexNet's rnorm1: a quarter of 96 channels) with 6 / 12 / 24 channels per lane, window 64
// (rnorm2: a quarter of 256) with 8, LT locations per tile, ceil(C / CG) * LT threads.  false: none (the LDS-tiled kernels take the call).
struct RnFast {
  int CG, LT, threads, SZ;
};
inline bool rn_fast_shape(int C, int sizeF, bool blocked, bool vec, bool undo, RnFast& f) {
  if (!CHIP_KNOB("CONVNET_RNORM_FAST", 1) || blocked || !vec || C < sizeF) return false;
  f.SZ = sizeF;
  if (sizeF == 24) {
    // rows of LT * 4 bytes: 128-byte rows run at 4.2 TB/s, 64-byte ones at 3.1 (rnorm1, profiles/r06_rnorm.txt)
    f.CG = C <= 96 ? 12 : 6;
    if (const int k = CHIP_DIAG_KNOB("CONVNET_RNORM_FAST_CG", 0)) f.CG = k;
  } else if (sizeF == 64) {
    f.CG = 8;
  } else {
    return false;
  }
  const int G = divup(C, f.CG);
  f.LT = 128;
  while (f.LT > 16 && G * f.LT > 512) f.LT /= 2;
  if (f.LT == 128 && f.CG != 24) f.LT = 64;
  f.threads = G * f.LT;
  (void)undo;
  return f.threads <= 512;
}
inline size_t rn_fast_rows(const RnFast& f, int C, bool undo) {
  const int H = f.SZ / 2, SS = f.CG < H ? f.CG : H;
  const int R = rn_halo_lo(f.SZ) + C + rn_halo_hi(f.CG, f.SZ), seg = divup(C, f.CG) * (f.CG / SS) + 2 * (H / SS);
  return undo ? (size_t)2 * R + 2 * seg : (size_t)R + C + f.CG + seg;
}

bool rnorm_fwd_fast(const float* in, float* out, size_t locs, int C, int sizeF, float addScale, float powScale, bool blocked, bool vec, bool relu) {
  RnFast f;
  if (!rn_fast_shape(C, sizeF, blocked, vec, false, f)) return false;
  const size_t smem = sizeof(float) * (size_t)f.LT * rn_fast_rows(f, C, false);
  if (smem > 160 * 1024) return false;
  const unsigned tiles = (unsigned)((locs + f.LT - 1) / f.LT);
  const dim3 block(f.threads);
#define RN_F(L, G, S)                                                                                                           \
  do {                                                                                                                          \
    const dim3 grid(rn_pipe_grid((const void*)rnorm_fwd_fast_kernel<L, G, S>, tiles, smem, f.threads));                          \
    hipLaunchKernelGGL((rnorm_fwd_fast_kernel<L, G, S>), grid, block, smem, stream(), in, out, locs, C, addScale, powScale, relu, tiles); \
  } while (0)
  if (f.SZ == 24 && f.CG == 24 && f.LT == 128) RN_F(128, 24, 24);
  else if (f.SZ == 24 && f.CG == 12 && f.LT == 64) RN_F(64, 12, 24);
  else if (f.SZ == 24 && f.CG == 12 && f.LT == 32) RN_F(32, 12, 24);
  else if (f.SZ == 24 && f.CG == 6 && f.LT == 64) RN_F(64, 6, 24);
  else if (f.SZ == 24 && f.CG == 6 && f.LT == 32) RN_F(32, 6, 24);
  else if (f.SZ == 24 && f.CG == 6 && f.LT == 16) RN_F(16, 6, 24);
  else if (f.SZ == 64 && f.LT == 64) RN_F(64, 8, 64);
  else if (f.SZ == 64 && f.LT == 32) RN_F(32, 8, 64);
  else if (f.SZ == 64 && f.LT == 16) RN_F(16, 8, 64);
  else return false;
#undef RN_F
  return true;
}

bool rnorm_undo_fast(const float* dout, const float* in, float* out, size_t locs, int C, int sizeF, float addScale, float powScale, bool blocked, bool vec) {
  RnFast f;
  if (!rn_fast_shape(C, sizeF, blocked, vec, true, f)) return false;
  const size_t smem = sizeof(float) * (size_t)f.LT * rn_fast_rows(f, C, true);
  if (smem > 160 * 1024) return false;
  const unsigned tiles = (unsigned)((locs + f.LT - 1) / f.LT);
  const dim3 block(f.threads);
#define RN_U(L, G, S)                                                                                                           \
  do {                                                                                                                          \
    const dim3 grid(rn_pipe_grid((const void*)rnorm_undo_fast_kernel<L, G, S>, tiles, smem, f.threads));                         \
    hipLaunchKernelGGL((rnorm_undo_fast_kernel<L, G, S>), grid, block, smem, stream(), dout, in, out, locs, C, addScale, powScale, tiles); \
  } while (0)
  if (f.SZ == 24 && f.CG == 24 && f.LT == 128) RN_U(128, 24, 24);
  else if (f.SZ == 24 && f.CG == 12 && f.LT == 64) RN_U(64, 12, 24);
  else if (f.SZ == 24 && f.CG == 12 && f.LT == 32) RN_U(32, 12, 24);
  else if (f.SZ == 24 && f.CG == 6 && f.LT == 64) RN_U(64, 6, 24);
  else if (f.SZ == 24 && f.CG == 6 && f.LT == 32) RN_U(32, 6, 24);
  else if (f.SZ == 24 && f.CG == 6 && f.LT == 16) RN_U(16, 6, 24);
  else if (f.SZ == 64 && f.LT == 64) RN_U(64, 8, 64);
  else if (f.SZ == 64 && f.LT == 32) RN_U(32, 8, 64);
  else if (f.SZ == 64 && f.LT == 16) RN_U(16, 8, 64);
  else return false;
#undef RN_U
  return true;
}

PoolGeo pool_geo(const Shape4D* in, const Shape4D* out, const ConvDesc& d, const cudamat* mi, const cudamat* mo) {
  PoolGeo g;
  g.N = in->shape[0]; g.W = in->shape[1]; g.H = in->shape[2]; g.C = in->shape[3];
  g.Mx = out->shape[1]; g.My = out->shape[2];
  g.Ky = d.kernel_size_y; g.Kx = d.kernel_size_x; g.sy = d.stride_y; g.sx = d.stride_x; g.py = d.padding_y; g.px = d.padding_x;
  g.nvec = divup(g.N, 4);
  g.xper = g.xtotal = g.xrows = g.xbx = 0;
  CHIP_REQUIRE(out->shape[0] == g.N && out->shape[3] == g.C);
  CHIP_REQUIRE(d.num_input_channels == g.C && d.num_output_channels == g.C);  // cudamat_conv_gemm.cu:1150-1160
  CHIP_REQUIRE(mi->size[0] == g.N && mi->size[1] == g.H * g.W * g.C);
  CHIP_REQUIRE(mo->size[0] == g.N && mo->size[1] == g.My * g.Mx * g.C);
  CHIP_REQUIRE(g.My == (g.H - 2 * g.py - g.Ky) / g.sy + 1 && g.Mx == (g.W - 2 * g.px - g.Kx) / g.sx + 1);
  return g;
}

// 10*K+S when a compile-time (K, S) instantiation exists and its 3-D grid is legal, else 0 (generic kernel)
inline int fixed_window(const PoolGeo& g, bool vec) {
  if (!vec || g.Ky != g.Kx || g.sy != g.sx || g.H > 65535 || g.C > 65535) return 0;
  if (g.Ky == 3 && g.sy == 2) return 32;
  if (g.Ky == 2 && g.sy == 2) return 22;
  return 0;
}

// Switches a fixed-window launch to the XCD-aware 1-D order (CONVNET_POOL_NO_XCD=1 keeps the plain 3-D grid, for A/B runs).
inline dim3 pool_xcd_grid(PoolGeo& g, int xblocks, int rows, dim3 plain) {
  const bool off = CHIP_DIAG_KNOB("CONVNET_POOL_NO_XCD", 0) != 0;
  const long long total = (long long)xblocks * rows * g.C;
  g.xper = g.xtotal = g.xrows = g.xbx = 0;   // a second call on the same geometry (the 2 x 2-block undo) starts clean
  if (off || total < 64 || total > (1ll << 30)) return plain;
  g.xbx = xblocks; g.xrows = rows; g.xtotal = (int)total; g.xper = (int)((total + 7) / 8);
  return dim3(8 * g.xper);
}

template <bool MAX>
void pool_fwd(cudamat* images, cudamat* targets, Shape4D* is, Shape4D* ts, const ConvDesc& d, float st, float so) {
  PoolGeo g = pool_geo(is, ts, d, images, targets);
  const bool vec = g.N % 4 == 0 && a16(images->data_device) && a16(targets->data_device);
  const size_t total = (size_t)g.C * g.My * g.Mx * g.nvec;
  // algorithmic bytes: read input once, write output once
  KernelTimer timer(MAX ? "pool_fwd_kernel<max>" : "pool_fwd_kernel<avg>", "pool_fwd", 0.0, 4.0 * g.N * g.C * ((double)g.H * g.W + (double)g.My * g.Mx));
  const int fx = fixed_window(g, vec);
  dim3 fgrid(divup(g.Mx * g.nvec, 256), g.My, g.C);
  if (fx) fgrid = pool_xcd_grid(g, fgrid.x, g.My, fgrid);
  if (fx == 32 && MAX && g.My * g.Mx >= 400 && CHIP_KNOB("CONVNET_POOL_FWD_BLOCK", 1)) {
    // one thread per 2 x 2 output block: 25 loads for four outputs instead of 36 (small maps keep the per-output kernel's parallelism)
    const int hb = (g.My + 1) / 2;
    dim3 bgrid(divup(((g.Mx + 1) / 2) * g.nvec, 256), hb, g.C);
    bgrid = pool_xcd_grid(g, bgrid.x, hb, bgrid);
    hipLaunchKernelGGL(pool_fwd_max32_block_kernel, bgrid, dim3(256), 0, stream(), images->data_device, targets->data_device, g, st, so);
  } else if (fx == 32)
    hipLaunchKernelGGL((pool_fwd_fixed_kernel<MAX, 3, 2>), fgrid, dim3(256), 0, stream(), images->data_device, targets->data_device, g, st, so);
  else if (fx == 22)
    hipLaunchKernelGGL((pool_fwd_fixed_kernel<MAX, 2, 2>), fgrid, dim3(256), 0, stream(), images->data_device, targets->data_device, g, st, so);
  else
    hipLaunchKernelGGL(pool_fwd_kernel<MAX>, dim3(grid_for(total)), dim3(256), 0, stream(), images->data_device, targets->data_device, g,
                       st, so, vec);
}

template <bool MAX>
void pool_undo(cudamat* images, cudamat* grads, cudamat* acts, cudamat* targets, Shape4D* in_shape, Shape4D* pooled_shape,
               const ConvDesc& d, float st, bool relu_mask = false) {
  PoolGeo g = pool_geo(in_shape, pooled_shape, d, targets, grads);
  const bool vec = g.N % 4 == 0 && a16(grads->data_device) && a16(targets->data_device) &&
                   (!MAX || (a16(images->data_device) && a16(acts->data_device)));
  const size_t total = (size_t)g.C * g.H * g.W * g.nvec;
  KernelTimer timer(MAX ? "pool_undo_kernel<max>" : "pool_undo_kernel<avg>", "pool_undo", 0.0,
                    4.0 * g.N * g.C * ((MAX ? 2.0 : 1.0) * g.H * g.W + (MAX ? 2.0 : 1.0) * g.My * g.Mx + (st != 0.f ? (double)g.H * g.W : 0.0)));
  const float* im = MAX ? images->data_device : nullptr;
  const float* ac = MAX ? acts->data_device : nullptr;
  const int fx = fixed_window(g, vec);
  dim3 fgrid(divup(g.W * g.nvec, 256), g.H, g.C);
  if (fx) fgrid = pool_xcd_grid(g, fgrid.x, g.H, fgrid);
  if (fx == 32 && MAX && g.H * g.W >= 400 && CHIP_KNOB("CONVNET_POOL_UNDO_BLOCK", 1)) {   // (11 x 11 maps: 17.7 vs 16.5 us)
    // one thread per 2 x 2 input block: a third of the loads per output
    const int hb = (g.H + 1) / 2;
    dim3 bgrid(divup(((g.W + 1) / 2) * g.nvec, 256), hb, g.C);
    bgrid = pool_xcd_grid(g, bgrid.x, hb, bgrid);
    hipLaunchKernelGGL(pool_undo_max32_block_kernel, bgrid, dim3(256), 0, stream(), im, grads->data_device, ac, targets->data_device, g, st, relu_mask);
  } else if (fx == 32)
    hipLaunchKernelGGL((pool_undo_fixed_kernel<MAX, 3, 2>), fgrid, dim3(256), 0, stream(), im, grads->data_device, ac, targets->data_device, g,
                       st, relu_mask);
  else if (fx == 22)
    hipLaunchKernelGGL((pool_undo_fixed_kernel<MAX, 2, 2>), fgrid, dim3(256), 0, stream(), im, grads->data_device, ac, targets->data_device, g,
                       st, relu_mask);
  else
    hipLaunchKernelGGL(pool_undo_kernel<MAX>, dim3(grid_for(total)), dim3(256), 0, stream(), im, grads->data_device, ac,
                       targets->data_device, g, st, vec, relu_mask);
}

}  // namespace
}  // namespace chip

using namespace chip;

extern "C" {

void MaxPoolGemm(cudamat* images, cudamat* targets, Shape4D* is, Shape4D* ts, ConvDesc d, float scaleTargets, float scaleOutput) {
  pool_fwd<true>(images, targets, is, ts, d, scaleTargets, scaleOutput);
}
void AvgPoolGemm(cudamat* images, cudamat* targets, Shape4D* is, Shape4D* ts, ConvDesc d, float scaleTargets, float scaleOutput) {
  pool_fwd<false>(images, targets, is, ts, d, scaleTargets, scaleOutput);
}
void MaxPool(cudamat* images, cudamat* targets, Shape4D* is, Shape4D* ts, ConvDesc d) { pool_fwd<true>(images, targets, is, ts, d, 0.f, 1.f); }
void AvgPool(cudamat* images, cudamat* targets, Shape4D* is, Shape4D* ts, ConvDesc d) { pool_fwd<false>(images, targets, is, ts, d, 0.f, 1.f); }

// deferred epilogues (common.h: PendingOp): the parked form of MaxPoolUndo — the ReLU' of the layer below can join it
static void launch_pool_undo(PendingOp& o) {
  pool_undo<true>(&o.m[0], &o.m[1], &o.m[2], &o.m[3], &o.s[0], &o.s[1], o.desc, o.scaleTargets, o.has_mask != 0);
}
static bool park_pool_undo(cudamat* images, cudamat* maxGrads, cudamat* maxActs, cudamat* targets, Shape4D* images_shape, Shape4D* maxGrads_shape,
                           const ConvDesc& d, float scaleTargets) {
  if (!defer_begin(3, launch_pool_undo)) return false;
  PendingOp& o = pending();
  o.m[0] = *images; o.m[1] = *maxGrads; o.m[2] = *maxActs; o.m[3] = *targets;
  o.s[0] = *images_shape; o.s[1] = *maxGrads_shape;
  o.desc = d;
  o.scaleTargets = scaleTargets;
  return true;
}
void MaxPoolUndoGemm(cudamat* images, cudamat* maxGrads, cudamat* maxActs, cudamat* targets, Shape4D* images_shape, Shape4D* maxGrads_shape,
                     ConvDesc d, float scaleTargets) {
  if (park_pool_undo(images, maxGrads, maxActs, targets, images_shape, maxGrads_shape, d, scaleTargets)) return;
  pool_undo<true>(images, maxGrads, maxActs, targets, images_shape, maxGrads_shape, d, scaleTargets);
}
void MaxPoolUndoRelu(cudamat* images, cudamat* maxGrads, cudamat* maxActs, cudamat* targets, Shape4D* images_shape,
                     Shape4D* maxGrads_shape, ConvDesc d, float scaleTargets) {
  pool_undo<true>(images, maxGrads, maxActs, targets, images_shape, maxGrads_shape, d, scaleTargets, true);
}
void MaxPoolUndo(cudamat* images, cudamat* maxGrads, cudamat* maxActs, cudamat* targets, Shape4D* images_shape, Shape4D* maxGrads_shape,
                 ConvDesc d, float scaleTargets) {
  if (park_pool_undo(images, maxGrads, maxActs, targets, images_shape, maxGrads_shape, d, scaleTargets)) return;
  pool_undo<true>(images, maxGrads, maxActs, targets, images_shape, maxGrads_shape, d, scaleTargets);
}
// ---- max pooling with a window mask (the kernels' header has the format) ------------------------------------------------------------------
static bool mask_geo_ok(const PoolGeo& g, const cudamat* images, const cudamat* pooled_a, const cudamat* pooled_b, const cudamat* mask) {
  const bool vec = g.N % 4 == 0 && a16(images->data_device) && a16(pooled_a->data_device) && (!pooled_b || a16(pooled_b->data_device)) &&
                   a16(mask->data_device);
  // (negated paddings: py, px <= 0 is what the 2 x 2-block undo assumes)
  return fixed_window(g, vec) == 32 && g.py <= 0 && g.px <= 0 && numel(mask) * 2 >= (size_t)g.N * g.C * g.My * g.Mx;
}
int MaxPoolMask(cudamat* images, cudamat* targets, cudamat* mask, Shape4D* images_shape, Shape4D* targets_shape, ConvDesc conv_desc) {
  if (!images->on_device || !targets->on_device || !mask->on_device) return ERROR_NOT_ON_DEVICE;
  PoolGeo g = pool_geo(images_shape, targets_shape, conv_desc, images, targets);
  if (!mask_geo_ok(g, images, targets, nullptr, mask)) return ERROR_UNSUPPORTED;
  KernelTimer timer("pool_fwd_mask_kernel<max>", "pool_fwd", 0.0, (double)g.N * g.C * (4.0 * g.H * g.W + 6.0 * g.My * g.Mx));
  const int hb = (g.My + 1) / 2;
  dim3 bgrid(divup(((g.Mx + 1) / 2) * g.nvec, 256), hb, g.C);
  bgrid = pool_xcd_grid(g, bgrid.x, hb, bgrid);
  hipLaunchKernelGGL(pool_fwd_max32_mask_kernel, bgrid, dim3(256), 0, stream(), images->data_device, targets->data_device,
                     reinterpret_cast<u32x2*>(mask->data_device), g);
  return launch_status();
}
int MaxPoolUndoMask(cudamat* maxGrads, cudamat* mask, cudamat* targets, Shape4D* targets_shape, Shape4D* maxGrads_shape, ConvDesc conv_desc,
                    float scaleTargets, int relu) {
  if (!maxGrads->on_device || !targets->on_device || !mask->on_device) return ERROR_NOT_ON_DEVICE;
  // MaxPoolUndoRelu masks the WHOLE result, accumulated target included, by input > 0 — known from the masks only where the input is some
  // window's maximum: the fused ReLU' is offered for an overwriting undo only
  if (relu && scaleTargets != 0.f) return ERROR_UNSUPPORTED;
  PoolGeo g = pool_geo(targets_shape, maxGrads_shape, conv_desc, targets, maxGrads);
  if (!mask_geo_ok(g, targets, maxGrads, nullptr, mask)) return ERROR_UNSUPPORTED;
  KernelTimer timer("pool_undo_mask_kernel<max>", "pool_undo", 0.0,
                    (double)g.N * g.C * ((scaleTargets != 0.f ? 8.0 : 4.0) * g.H * g.W + 6.0 * g.My * g.Mx));
  const int hb = (g.H + 1) / 2;
  dim3 bgrid(divup(((g.W + 1) / 2) * g.nvec, 256), hb, g.C);
  bgrid = pool_xcd_grid(g, bgrid.x, hb, bgrid);
  hipLaunchKernelGGL(pool_undo_max32_mask_kernel, bgrid, dim3(256), 0, stream(), maxGrads->data_device,
                     reinterpret_cast<const u32x2*>(mask->data_device), targets->data_device, g, scaleTargets, relu != 0);
  return launch_status();
}
void AvgPoolUndoGemm(cudamat* avgGrads, cudamat* targets, Shape4D* avgGrads_shape, Shape4D* targets_shape, ConvDesc d, float scaleTargets) {
  pool_undo<false>(nullptr, avgGrads, nullptr, targets, targets_shape, avgGrads_shape, d, scaleTargets);
}
void AvgPoolUndo(cudamat* avgGrads, cudamat* targets, Shape4D* avgGrads_shape, Shape4D* targets_shape, ConvDesc d, float scaleTargets) {
  pool_undo<false>(nullptr, avgGrads, nullptr, targets, targets_shape, avgGrads_shape, d, scaleTargets);
}

static void rnorm_fwd_impl(cudamat* images, cudamat* targets, int numFilters, int sizeF, float addScale, float powScale, bool blocked, bool relu) {
  const size_t total = numel(images);
  CHIP_REQUIRE(numel(targets) == total && numFilters > 0 && total % numFilters == 0 && sizeF > 0);
  const size_t locs = total / numFilters;
  const bool vec = locs % 4 == 0 && a16(images->data_device) && a16(targets->data_device);
  KernelTimer timer("rnorm_fwd_kernel", "rnorm_fwd", 0.0, 8.0 * total);
  {
    const int C = numFilters;
    int LT = C <= 192 ? 64 : (C <= 384 ? 32 : (C <= 768 ? 16 : 0));
    if (const int f = CHIP_DIAG_KNOB("CONVNET_RNORM_FWD_LT", 0)) LT = f;   // tuning knob (tools/pool_bench.py, -DCONVNET_DIAG builds)
    if (rnorm_fwd_fast(images->data_device, targets->data_device, locs, C, sizeF, addScale, powScale, blocked, vec, relu)) return;
    if (LT) {   // LDS-tiled, read-once/write-once
      const size_t smem = sizeof(float) * (size_t)C * LT;
      const bool xcd = !CHIP_DIAG_KNOB("CONVNET_RNORM_NO_XCD", 0);   // A/B switch for the XCD-contiguous tile order
      const unsigned tiles = (unsigned)((locs + LT - 1) / LT);
      const dim3 grid(xcd ? (tiles + 7) / 8 * 8 : tiles), block(256);
#define RN_FWD(L) hipLaunchKernelGGL(rnorm_fwd_lds_kernel<L>, grid, block, smem, stream(), images->data_device, targets->data_device, locs, C, sizeF, addScale, powScale, blocked, vec, relu, tiles, xcd)
      if (LT == 64) RN_FWD(64);
      else if (LT == 32) RN_FWD(32);
      else if (LT == 16) RN_FWD(16);
      else RN_FWD(8);
#undef RN_FWD
      return;
    }
  }
  int cseg, nseg;
  rnorm_segments((locs + 3) / 4, numFilters, sizeF, cseg, nseg);
  hipLaunchKernelGGL(rnorm_fwd_kernel, dim3(grid_for((locs + 3) / 4 * nseg)), dim3(256), 0, stream(), images->data_device, targets->data_device, locs,
                     numFilters, sizeF, addScale, powScale, blocked, vec, cseg, nseg, relu);
}
static void launch_rnorm_fwd(PendingOp& o) { rnorm_fwd_impl(&o.m[0], &o.m[1], o.i0, o.i1, o.f0, o.f1, o.b0, o.relu != 0); }
void ResponseNormCrossMapGemm(cudamat* images, cudamat* targets, int numFilters, int sizeF, float addScale, float powScale, bool blocked) {
  if (defer_begin(4, launch_rnorm_fwd)) {   // (the ReLU of a RECTIFIED_LINEAR destination layer can join it)
    PendingOp& o = pending();
    o.m[0] = *images; o.m[1] = *targets;
    o.i0 = numFilters; o.i1 = sizeF; o.f0 = addScale; o.f1 = powScale; o.b0 = blocked;
    return;
  }
  rnorm_fwd_impl(images, targets, numFilters, sizeF, addScale, powScale, blocked, false);
}
void ResponseNormCrossMapRelu(cudamat* images, cudamat* targets, int numFilters, int sizeF, float addScale, float powScale, bool blocked) {
  rnorm_fwd_impl(images, targets, numFilters, sizeF, addScale, powScale, blocked, true);
}
void ResponseNormCrossMap(cudamat* images, cudamat* targets, int numFilters, int sizeF, float addScale, float powScale, bool blocked) {
  ResponseNormCrossMapGemm(images, targets, numFilters, sizeF, addScale, powScale, blocked);
}

void ResponseNormCrossMapUndoGemm(cudamat* outGrads, cudamat* inputs, cudamat* targets, int numFilters, int sizeF, float addScale,
                                  float powScale, bool blocked) {
  const size_t total = numel(inputs);
  CHIP_REQUIRE(numel(targets) == total && numel(outGrads) == total && numFilters > 0 && total % numFilters == 0 && sizeF > 0);
  const size_t locs = total / numFilters;
  {
    const int C = numFilters;
    // measured (N=256): C=96 LT 8/16/32 -> 484/291/324 us, C=256 LT 8/16/32 -> 104/83/166 us: more, smaller blocks per CU
    // (the three phases of a block do not overlap) beat longer contiguous rows
    int LT = C <= 64 ? 64 : (C <= 256 ? 16 : (C <= 512 ? 8 : 0));
    if (const int f = CHIP_DIAG_KNOB("CONVNET_RNORM_UNDO_LT", 0)) LT = f;   // tuning knob (tools/pool_bench.py, -DCONVNET_DIAG builds)
    if (LT) {
      const bool vec = locs % 4 == 0 && a16(outGrads->data_device) && a16(inputs->data_device) && a16(targets->data_device);
      KernelTimer timer("rnorm_undo_kernels", "rnorm_undo", 0.0, 12.0 * total);
      if (rnorm_undo_fast(outGrads->data_device, inputs->data_device, targets->data_device, locs, C, sizeF, addScale, powScale, blocked, vec)) return;
      const size_t smem = sizeof(float) * 3 * (size_t)C * LT;
      const bool xcd = !CHIP_DIAG_KNOB("CONVNET_RNORM_NO_XCD", 0);   // A/B switch for the XCD-contiguous tile order
      const unsigned tiles = (unsigned)((locs + LT - 1) / LT);
      const dim3 grid(xcd ? (tiles + 7) / 8 * 8 : tiles), block(CHIP_DIAG_KNOB("CONVNET_RNORM_UNDO_THREADS", 0) ? CHIP_DIAG_KNOB("CONVNET_RNORM_UNDO_THREADS", 0) : C > 128 ? 512 : 256);   // (rnorm2, C = 256: 75 -> 58 us with 8 instead of 16 channels per thread; C = 96: no difference)
      CHIP_REQUIRE(smem <= 160 * 1024);
#define RN_UNDO(L)                                                                                                              \
  do {                                                                                                                          \
    if (smem > 64 * 1024)   /* >64 KiB of dynamic LDS needs an explicit opt-in */                                                \
      CHIP_CHECK(hipFuncSetAttribute((const void*)rnorm_undo_lds_kernel<L>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
    hipLaunchKernelGGL(rnorm_undo_lds_kernel<L>, grid, block, smem, stream(), outGrads->data_device, inputs->data_device,       \
                       targets->data_device, locs, C, sizeF, addScale, powScale, blocked, vec, tiles, xcd);                      \
  } while (0)
      if (LT == 64) RN_UNDO(64);
      else if (LT == 32) RN_UNDO(32);
      else if (LT == 16) RN_UNDO(16);
      else RN_UNDO(8);
#undef RN_UNDO
      return;
    }
  }
  const size_t padded = (total + 63) / 64 * 64;
  float* prod = static_cast<float*>(workspace(sizeof(float) * padded * 2));
  float* scaled = prod + padded;
  const bool vec = locs % 4 == 0 && a16(outGrads->data_device) && a16(inputs->data_device) && a16(targets->data_device);
  int cseg, nseg;
  rnorm_segments((locs + 3) / 4, numFilters, sizeF, cseg, nseg);
  const int grid = grid_for((locs + 3) / 4 * nseg);
  KernelTimer timer("rnorm_undo_kernels", "rnorm_undo", 0.0, 12.0 * total);
  hipLaunchKernelGGL(rnorm_undo1_kernel, dim3(grid), dim3(256), 0, stream(), outGrads->data_device, inputs->data_device, prod, scaled, locs,
                     numFilters, sizeF, addScale, powScale, blocked, vec, cseg, nseg);
  hipLaunchKernelGGL(rnorm_undo2_kernel, dim3(grid), dim3(256), 0, stream(), inputs->data_device, prod, scaled, targets->data_device, locs,
                     numFilters, sizeF, addScale, powScale, blocked, vec, cseg, nseg);
}
void ResponseNormCrossMapUndo(cudamat* outGrads, cudamat* inputs, cudamat* /*acts*/, cudamat* targets, int numFilters, int sizeF,
                              float addScale, float powScale, bool blocked) {
  ResponseNormCrossMapUndoGemm(outGrads, inputs, targets, numFilters, sizeF, addScale, powScale, blocked);
}

}  // extern "C"
