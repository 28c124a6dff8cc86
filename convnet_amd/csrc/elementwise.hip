// HBM-bound pieces of the hot path: elementwise ops, axis reductions, output-layer kernels,
// the SGD update and the RNG-driven ops.  Reference: cudamat/cudamat.cu (host wrappers) +
// cudamat/cudamat_kernels.cu (kernels); semantics pinned by eigenmat/eigenmat.cc (the CPU oracle).
// All kernels stream float4 per lane where the pointers allow it and grid-stride over ~2048 blocks.
#include <cfloat>
#include <cmath>

#include "common.h"

namespace chip {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kThreads = 256;
constexpr int kMaxBlocks = 2048;

inline int blocks_for(size_t work_items) {
  size_t b = (work_items + kThreads - 1) / kThreads;
  if (b > kMaxBlocks) b = kMaxBlocks;
  if (b < 1) b = 1;
  return (int)b;
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// out[i] = f(i, a[i], b[i]); a/b may alias out.  Vector body + scalar tail.
template <typename F>
__global__ void map2_kernel(float* __restrict__ out, const float* a, const float* b, size_t n, bool vec, F f) {
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  if (vec) {
    const size_t n4 = n >> 2;
    for (size_t i = tid; i < n4; i += stride) {
      const f32x4 x = reinterpret_cast<const f32x4*>(a)[i];
      const f32x4 y = b ? reinterpret_cast<const f32x4*>(b)[i] : x;
      f32x4 r;
#pragma unroll
      for (int e = 0; e < 4; ++e) r[e] = f(x[e], y[e]);
      reinterpret_cast<f32x4*>(out)[i] = r;
    }
    for (size_t i = (n4 << 2) + tid; i < n; i += stride) out[i] = f(a[i], b ? b[i] : a[i]);
  } else {
    for (size_t i = tid; i < n; i += stride) out[i] = f(a[i], b ? b[i] : a[i]);
  }
}

template <typename F>
int map2(cudamat* out, const cudamat* a, const cudamat* b, F f) {
  const size_t n = numel(a);
  if (!a->on_device || !out->on_device || (b && !b->on_device)) return ERROR_NOT_ON_DEVICE;
  if (numel(out) != n || (b && numel(b) != n)) return ERROR_INCOMPATIBLE_DIMENSIONS;
  if (n == 0) return 0;
  const bool vec = al16(out->data_device) && al16(a->data_device) && (!b || al16(b->data_device));
  KernelTimer timer("map2_kernel", "elementwise", 0.0, 4.0 * n * (b ? 3 : 2));
  hipLaunchKernelGGL(map2_kernel<F>, dim3(blocks_for(n / 4 + 1)), dim3(kThreads), 0, stream(), out->data_device,
                     a->data_device, b ? b->data_device : nullptr, n, vec, f);
  return launch_status();
}

// ---- block reductions ------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) sh[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
  return t;
}

// target[j] = p*target[j] + mult * sum_i g(mat[i + rows*j]) : one block per column (contiguous).
template <bool SQ>
__global__ void colsum_block_kernel(const float* __restrict__ mat, float* __restrict__ target, int rows, float mult, float p) {
  __shared__ float sh[8];
  const float* col = mat + (size_t)blockIdx.x * rows;
  float s = 0.f;
  if (((reinterpret_cast<uintptr_t>(col) & 15) == 0) && (rows & 3) == 0) {
    const f32x4* c4 = reinterpret_cast<const f32x4*>(col);
    for (int i = threadIdx.x; i < (rows >> 2); i += blockDim.x) {
      const f32x4 v = c4[i];
      s += SQ ? (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]) : (v[0] + v[1]) + (v[2] + v[3]);
    }
  } else {
    for (int i = threadIdx.x; i < rows; i += blockDim.x) s += SQ ? col[i] * col[i] : col[i];
  }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) target[blockIdx.x] = (p != 0.f ? p * target[blockIdx.x] : 0.f) + mult * s;
}

// few, very long columns (shared-bias gradient: 96 columns x 774,400 rows for conv1): split every column
// over gridDim.y blocks into a slab of partial sums, then one tiny deterministic finishing pass.
template <bool SQ>
__global__ void colsum_split_kernel(const float* __restrict__ mat, float* __restrict__ part, int rows, int per) {
  __shared__ float sh[8];
  const float* col = mat + (size_t)blockIdx.x * rows;
  const int beg = blockIdx.y * per, end = min(rows, beg + per);
  float s = 0.f;
  if (((reinterpret_cast<uintptr_t>(col) & 15) == 0) && (rows & 3) == 0 && (per & 3) == 0) {
    const f32x4* c4 = reinterpret_cast<const f32x4*>(col);
    for (int i = (beg >> 2) + threadIdx.x; i < (end >> 2); i += blockDim.x) {
      const f32x4 v = c4[i];
      s += SQ ? (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]) : (v[0] + v[1]) + (v[2] + v[3]);
    }
  } else {
    for (int i = beg + threadIdx.x; i < end; i += blockDim.x) s += SQ ? col[i] * col[i] : col[i];
  }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = s;
}

__global__ void colsum_finish_kernel(const float* __restrict__ part, float* __restrict__ target, int cols, int splits, float mult, float p) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= cols) return;
  float s = 0.f;
  for (int k = 0; k < splits; ++k) s += part[(size_t)k * cols + j];
  target[j] = (p != 0.f ? p * target[j] : 0.f) + mult * s;
}

// short columns: one wave per column, 4 columns per block
template <bool SQ>
__global__ void colsum_wave_kernel(const float* __restrict__ mat, float* __restrict__ target, int rows, int cols, float mult, float p) {
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6), l = threadIdx.x & 63;
  if (j >= cols) return;
  const float* col = mat + (size_t)j * rows;
  float s = 0.f;
  for (int i = l; i < rows; i += 64) s += SQ ? col[i] * col[i] : col[i];
  s = wave_sum(s);
  if (l == 0) target[j] = (p != 0.f ? p * target[j] : 0.f) + mult * s;
}

// many short columns (the reference's two-step shared-bias gradient sums a (batch, pixels*filters) matrix over the batch first,
// src/conv_edge.cc:213-218: 1.16 M columns of 256 for conv1): float4 loads, 8 columns per wave so 8 loads are in flight before
// the first cross-lane reduction.
template <bool SQ>
__global__ void colsum_wave4_kernel(const float* __restrict__ mat, float* __restrict__ target, int rows, int cols, float mult, float p) {
  const int l = threadIdx.x & 63;
  const int j0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 8;
  const int r4 = rows >> 2;
  float s[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    s[u] = 0.f;
    if (j0 + u < cols) {
      const f32x4* c4 = reinterpret_cast<const f32x4*>(mat + (size_t)(j0 + u) * rows);
      for (int i = l; i < r4; i += 64) {
        const f32x4 v = c4[i];
        s[u] += SQ ? (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]) : (v[0] + v[1]) + (v[2] + v[3]);
      }
    }
  }
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const float t = wave_sum(s[u]);
    if (l == 0 && j0 + u < cols) target[j0 + u] = (p != 0.f ? p * target[j0 + u] : 0.f) + mult * t;
  }
}

// target[i] = p*target[i] + mult * sum_j g(mat[i + rows*j]) : one thread per row, coalesced over rows.
template <bool SQ>
__global__ void rowsum_kernel(const float* __restrict__ mat, float* __restrict__ target, int rows, int cols, float mult, float p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows) return;
  float s = 0.f;
  for (int j = 0; j < cols; ++j) {
    const float v = mat[(size_t)i + (size_t)rows * j];
    s += SQ ? v * v : v;
  }
  target[i] = (p != 0.f ? p * target[i] : 0.f) + mult * s;
}

template <bool SQ>
int axis_sum(cudamat* mat, cudamat* target, int axis, float mult, float p) {
  if (!mat->on_device || !target->on_device) return ERROR_NOT_ON_DEVICE;
  if (mat->is_trans) return ERROR_TRANSPOSED;
  const int rows = mat->size[0], cols = mat->size[1];
  if (axis == 0) {
    if (target->size[0] != 1 || target->size[1] != cols) return ERROR_INCOMPATIBLE_DIMENSIONS;
    if (rows >= 32768 && cols < 1024) {
      int splits = 2048 / cols;
      if (splits > rows / 4096) splits = rows / 4096;
      if (splits < 2) splits = 2;
      int per = ((divup(rows, splits) + 3) / 4) * 4;
      splits = divup(rows, per);
      float* part = static_cast<float*>(workspace(sizeof(float) * (size_t)splits * cols));
      KernelTimer timer(SQ ? "colsum_split_kernel<sq>" : "colsum_split_kernel", "colsum", 0.0, 4.0 * rows * cols);
      hipLaunchKernelGGL(colsum_split_kernel<SQ>, dim3(cols, splits), dim3(256), 0, stream(), mat->data_device, part, rows, per);
      hipLaunchKernelGGL(colsum_finish_kernel, dim3(divup(cols, 256)), dim3(256), 0, stream(), part, target->data_device, cols, splits, mult, p);
    } else if (rows >= 2048)
      hipLaunchKernelGGL(colsum_block_kernel<SQ>, dim3(cols), dim3(256), 0, stream(), mat->data_device, target->data_device, rows, mult, p);
    else if ((rows & 3) == 0 && rows >= 128 && cols >= 4096 && (reinterpret_cast<uintptr_t>(mat->data_device) & 15) == 0)
      hipLaunchKernelGGL(colsum_wave4_kernel<SQ>, dim3(divup(cols, 32)), dim3(256), 0, stream(), mat->data_device, target->data_device, rows, cols, mult, p);
    else
      hipLaunchKernelGGL(colsum_wave_kernel<SQ>, dim3(divup(cols, 4)), dim3(256), 0, stream(), mat->data_device, target->data_device, rows, cols, mult, p);
  } else if (axis == 1) {
    if (target->size[1] != 1 || target->size[0] != rows) return ERROR_INCOMPATIBLE_DIMENSIONS;
    hipLaunchKernelGGL(rowsum_kernel<SQ>, dim3(divup(rows, 256)), dim3(256), 0, stream(), mat->data_device, target->data_device, rows, cols, mult, p);
  } else {
    return ERROR_UNSUPPORTED;
  }
  return launch_status();
}

// Whole-matrix reductions return a float to the host (vdot/sum_all/euclid_norm do in the reference
// too: cublasSdot etc.) — a device->host sync, used off the per-step path only.
template <int MODE>  // 0: sum a, 1: sum a*b, 2: sum a*a
__global__ void reduce_all_kernel(const float* __restrict__ a, const float* __restrict__ b, size_t n, float* __restrict__ out) {
  __shared__ float sh[8];
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    s += MODE == 0 ? a[i] : (MODE == 1 ? a[i] * b[i] : a[i] * a[i]);
  s = block_sum(s, sh);
  if (threadIdx.x == 0) out[blockIdx.x] = s;
}

template <int MODE>
float reduce_all(const cudamat* a, const cudamat* b, int* err) {
  *err = 0;
  const size_t n = numel(a);
  const int nb = blocks_for(n) > 256 ? 256 : blocks_for(n);
  float* part = static_cast<float*>(workspace(sizeof(float) * 256));
  hipLaunchKernelGGL(reduce_all_kernel<MODE>, dim3(nb), dim3(kThreads), 0, stream(), a->data_device, b ? b->data_device : nullptr, n, part);
  float host[256];
  if (hipMemcpyAsync(host, part, sizeof(float) * nb, hipMemcpyDeviceToHost, stream()) != hipSuccess ||
      hipStreamSynchronize(stream()) != hipSuccess) {
    *err = CUDA_ERROR;
    return 0.f;
  }
  double t = 0;
  for (int i = 0; i < nb; ++i) t += host[i];
  return (float)t;
}

// ---- row-vector broadcast: mat[:, j] + mult*vec[j] ----------------------------------------------------
__global__ void add_row_kernel(float* __restrict__ out, const float* __restrict__ mat, const float* __restrict__ vec, int rows,
                               size_t n, float mult, bool vec4) {
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  if (vec4) {  // rows % 4 == 0: the 4 lanes of a float4 share a column
    for (size_t i = tid; i < (n >> 2); i += stride) {
      const float b = mult * vec[(i << 2) / rows];
      f32x4 v = reinterpret_cast<const f32x4*>(mat)[i];
      v = v + b;
      reinterpret_cast<f32x4*>(out)[i] = v;
    }
  } else {
    for (size_t i = tid; i < n; i += stride) out[i] = mat[i] + mult * vec[i / rows];
  }
}

int add_row(cudamat* mat, cudamat* vec, cudamat* target, float mult) {
  if (!mat->on_device || !vec->on_device || !target->on_device) return ERROR_NOT_ON_DEVICE;
  if (mat->is_trans) return ERROR_TRANSPOSED;
  if (mat->size[1] != vec->size[1] || vec->size[0] != 1 || mat->size[0] != target->size[0] || mat->size[1] != target->size[1])
    return ERROR_INCOMPATIBLE_DIMENSIONS;
  const size_t n = numel(mat);
  const bool v4 = (mat->size[0] & 3) == 0 && al16(mat->data_device) && al16(target->data_device);
  hipLaunchKernelGGL(add_row_kernel, dim3(blocks_for(n / 4 + 1)), dim3(kThreads), 0, stream(), target->data_device, mat->data_device,
                     vec->data_device, mat->size[0], n, mult, v4);
  return launch_status();
}

// ---- row-norm limit (axis=1: one output unit's incoming weights), eigenmat.cc:918-968 -----------------
// A (rows x cols) weight matrix is tiled as 256 rows x `chunk` columns per block; a lane owns one row of the
// tile (loads coalesce across lanes) and keeps that row's partial sum of squares privately, so the
// per-row norms come out deterministic (slab of partials + fixed-order finish, no atomics).
//   pass 1 (optionally fused with the SGD update, which touches every weight anyway):  partial[chunk][row]
//   pass 2: factor[row] = (constraint || norm_row > limit) ? limit / norm_row : 1
//   pass 3: rescale — a block exits without touching memory when none of its 256 rows needs scaling, which
//           is the steady state of weight_norm_limit (src/optimizer.cc:75-81).
__device__ __forceinline__ void sgd_one(float& g, float& w, float& h, float l2, float clip, float eps, float mom);

template <bool DO_SGD>
__global__ void rows_sq_kernel(float* __restrict__ g, float* __restrict__ w, float* __restrict__ h, int rows, int cols, int chunk,
                               float* __restrict__ partial, float l2, float clip, float eps, float mom) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const int c0 = blockIdx.y * chunk, c1 = min(cols, c0 + chunk);
  float s = 0.f;
  for (int c = c0; c < c1; ++c) {
    const size_t i = (size_t)r + (size_t)rows * c;
    float wv = w[i];
    if (DO_SGD) {
      float gv = g[i], hv = h[i];
      sgd_one(gv, wv, hv, l2, clip, eps, mom);
      g[i] = gv;
      h[i] = hv;
      w[i] = wv;
    }
    s += wv * wv;
  }
  partial[(size_t)blockIdx.y * rows + r] = s;
}

// rows % 4 == 0 and 16-byte aligned bases: a lane owns 4 consecutive rows (one float4 per tensor per column),
// two columns in flight, so each lane keeps 6 independent 16-byte loads outstanding instead of 3 scalar ones.
template <bool DO_SGD>
__global__ void rows_sq4_kernel(float* __restrict__ g, float* __restrict__ w, float* __restrict__ h, int rows, int cols, int chunk,
                                float* __restrict__ partial, float l2, float clip, float eps, float mom) {
  const int r = 4 * (blockIdx.x * blockDim.x + threadIdx.x);
  if (r >= rows) return;
  const int c0 = blockIdx.y * chunk, c1 = min(cols, c0 + chunk);
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  auto one = [&](int c) {
    const size_t i = (size_t)r + (size_t)rows * c;
    f32x4 wv = *reinterpret_cast<f32x4*>(w + i);
    if (DO_SGD) {
      f32x4 gv = *reinterpret_cast<f32x4*>(g + i), hv = *reinterpret_cast<f32x4*>(h + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = gv[e], b = wv[e], d = hv[e];
        sgd_one(a, b, d, l2, clip, eps, mom);
        gv[e] = a; wv[e] = b; hv[e] = d;
      }
      *reinterpret_cast<f32x4*>(g + i) = gv;
      *reinterpret_cast<f32x4*>(h + i) = hv;
      *reinterpret_cast<f32x4*>(w + i) = wv;
    }
    s += wv * wv;
  };
  int c = c0;
  for (; c + 1 < c1; c += 2) {
    one(c);
    one(c + 1);
  }
  if (c < c1) one(c);
  *reinterpret_cast<f32x4*>(partial + (size_t)blockIdx.y * rows + r) = s;
}

__global__ void row_factor_kernel(const float* __restrict__ partial, int nchunks, int rows, float norm, int constraint, float* __restrict__ factor) {
  // block = 64 rows x 4 chunk lanes; lane q sums chunks q, q+4, ... then the 4 partials combine in fixed order
  __shared__ float sh[4][64];
  const int r = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
  float s = 0.f;
  if (r < rows)
    for (int k = q; k < nchunks; k += 4) s += partial[(size_t)k * rows + r];
  sh[q][threadIdx.x & 63] = s;
  __syncthreads();
  if (q != 0 || r >= rows) return;
  s = sqrtf((sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]));
  factor[r] = (constraint == 1 || s > norm) ? norm / s : 1.f;
}

__global__ void row_scale_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols, int chunk, const float* __restrict__ factor) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  const float f = r < rows ? factor[r] : 1.f;
  const bool in_place = src == dst;
  if (in_place && !__syncthreads_or(f != 1.f)) return;
  if (r >= rows) return;
  const int c0 = blockIdx.y * chunk, c1 = min(cols, c0 + chunk);
  for (int c = c0; c < c1; ++c) {
    const size_t i = (size_t)r + (size_t)rows * c;
    dst[i] = src[i] * f;
  }
}

// grad/history may be null (plain norm limit).  Returns launch status.
inline int rows_normlimit(float* g, float* w_in, float* w_out, float* h, int rows, int cols, float norm, int constraint, bool do_sgd,
                          float l2, float clip, float eps, float mom) {
  int chunk = 64;
  while ((long)divup(rows, 256) * divup(cols, chunk) > 4096 && chunk < cols) chunk *= 2;
  const int nchunks = divup(cols, chunk);
  float* partial = static_cast<float*>(workspace(sizeof(float) * ((size_t)nchunks * rows + rows)));
  float* factor = partial + (size_t)nchunks * rows;
  dim3 grid(divup(rows, 256), nchunks);
  const bool v4 = (rows & 3) == 0 && al16(w_in) && (!do_sgd || (al16(g) && al16(h)));
  // (one timer over the three launches: the fused SGD + row-norm pass reads g, w, h, writes h, w, then re-reads and re-writes w)
  KernelTimer timer(do_sgd ? "rows_sq_kernel<sgd> + row_scale_kernel" : "rows_sq_kernel + row_scale_kernel", do_sgd ? "sgd_normlimit" : "normlimit", 0.0,
                    4.0 * (double)rows * cols * (do_sgd ? 7 : 3));
  if (v4) {
    const dim3 g4(divup(rows / 4, 64), nchunks);
    if (do_sgd)
      hipLaunchKernelGGL(rows_sq4_kernel<true>, g4, dim3(64), 0, stream(), g, w_in, h, rows, cols, chunk, partial, l2, clip, eps, mom);
    else
      hipLaunchKernelGGL(rows_sq4_kernel<false>, g4, dim3(64), 0, stream(), nullptr, w_in, nullptr, rows, cols, chunk, partial, 0.f, 0.f, 0.f, 0.f);
  } else if (do_sgd)
    hipLaunchKernelGGL(rows_sq_kernel<true>, grid, dim3(256), 0, stream(), g, w_in, h, rows, cols, chunk, partial, l2, clip, eps, mom);
  else
    hipLaunchKernelGGL(rows_sq_kernel<false>, grid, dim3(256), 0, stream(), nullptr, w_in, nullptr, rows, cols, chunk, partial, 0.f, 0.f, 0.f, 0.f);
  hipLaunchKernelGGL(row_factor_kernel, dim3(divup(rows, 64)), dim3(256), 0, stream(), partial, nchunks, rows, norm, constraint, factor);
  hipLaunchKernelGGL(row_scale_kernel, grid, dim3(256), 0, stream(), w_in, w_out, rows, cols, chunk, factor);
  return launch_status();
}

__global__ void normlimit_cols_kernel(const float* __restrict__ mat, float* __restrict__ target, int rows, float norm, int constraint) {
  __shared__ float sh[8];
  const float* col = mat + (size_t)blockIdx.x * rows;
  float* out = target + (size_t)blockIdx.x * rows;
  float s = 0.f;
  for (int i = threadIdx.x; i < rows; i += blockDim.x) s += col[i] * col[i];
  s = sqrtf(block_sum(s, sh));
  const float sc = (constraint == 1 || s > norm) ? norm / s : 1.f;
  for (int i = threadIdx.x; i < rows; i += blockDim.x) out[i] = col[i] * sc;
}

// ---- output layer -------------------------------------------------------------------------------------
// "row_major" in the reference = one case per matrix ROW (eigenmat.cc:1093-1131).  A block owns 32
// consecutive rows (128 contiguous bytes per column) and G = 32 column groups (1024 threads: the 256 x 1000
// output layer is 8 blocks, so the per-lane column loop, not bandwidth, sets the time).
struct SoftmaxOut {
  float* probs;    // may alias logits
  float* deriv;    // nullable: probs with 1 subtracted at the label, times deriv_scale
  float* correct;  // nullable: 1x1 accumulator (+= number of argmax==label rows)
};

template <int G>   // column groups per block: 32 rows x G lanes-groups, G*32 threads
__global__ void __launch_bounds__(G * 32) softmax_rows_kernel(const float* logits, const float* __restrict__ labels, SoftmaxOut o, int rows, int cols, float deriv_scale) {
  __shared__ float red[G][33];
  __shared__ int redi[G][33];
  const int r = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int row = blockIdx.x * 32 + r;
  const bool ok = row < rows;
  // pass 1: max and first argmax (strict '<' as eigenmat.cc:1278)
  float mx = -FLT_MAX;
  int am = 0x7fffffff;
  if (ok)
    for (int j = g; j < cols; j += G) {
      const float v = logits[(size_t)j * rows + row];
      if (v > mx) { mx = v; am = j; }
    }
  red[g][r] = mx;
  redi[g][r] = am;
  __syncthreads();
  mx = red[0][r];
  am = redi[0][r];
#pragma unroll
  for (int k = 1; k < G; ++k) {
    const float v = red[k][r];
    const int a = redi[k][r];
    if (v > mx || (v == mx && a < am)) { mx = v; am = a; }
  }
  __syncthreads();
  // pass 2: exp and sum
  float s = 0.f;
  if (ok)
    for (int j = g; j < cols; j += G) {
      const size_t x = (size_t)j * rows + row;
      const float e = expf(logits[x] - mx);
      o.probs[x] = e;
      s += e;
    }
  red[g][r] = s;
  __syncthreads();
  s = 0.f;
#pragma unroll
  for (int k = 0; k < G; ++k) s += red[k][r];
  // pass 3: normalise (+ CE derivative)
  const int label = (ok && labels) ? (int)labels[row] : -1;
  if (ok)
    for (int j = g; j < cols; j += G) {
      const size_t x = (size_t)j * rows + row;
      const float pr = o.probs[x] / s;
      o.probs[x] = pr;
      if (o.deriv) o.deriv[x] = deriv_scale * (j == label ? pr - 1.0f : pr);
    }
  if (o.correct) {
    __syncthreads();
    if (g == 0) red[0][r] = (ok && am == label) ? 1.f : 0.f;
    __syncthreads();
    if (threadIdx.x == 0) {
      float c = 0.f;
      for (int k = 0; k < 32; ++k) c += red[0][k];
      atomicAdd(o.correct, c);  // integer-valued partial counts: exact and order-independent in fp32
    }
  }
}


// per-row gathers on an already-normalised probability matrix
__global__ void softmax_grad_kernel(const float* __restrict__ labels, float* __restrict__ target, int rows) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < rows) target[(size_t)i + (size_t)rows * (int)labels[i]] -= 1.0f;
}

// 32 rows x 32 column-lanes per block: lane c scans columns c, c+32, ... (coalesced over the 32 rows), then the 32 candidates of a
// row are reduced in LDS.  First maximum wins, as a sequential scan with `best < v` would pick it (cudamat.cu softmax-correct kernel).
__global__ void __launch_bounds__(1024) softmax_correct_kernel(const float* __restrict__ mat, const float* __restrict__ labels, float* __restrict__ target, int rows, int cols) {
  __shared__ float sv[32][33];
  __shared__ int si[32][33];
  const int r = threadIdx.x & 31, c = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + r;
  float best = -INFINITY;
  int am = 0x7fffffff;
  if (i < rows)
    for (int j = c; j < cols; j += 32) {
      const float v = mat[(size_t)j * rows + i];
      if (am == 0x7fffffff || best < v) { best = v; am = j; }
    }
  sv[r][c] = best;
  si[r][c] = am;
  __syncthreads();
  if (c == 0 && i < rows) {
    for (int k = 1; k < 32; ++k) {
      const float v = sv[r][k];
      const int j = si[r][k];
      if (j != 0x7fffffff && (best < v || (v == best && j < am))) { best = v; am = j; }
    }
    target[i] = ((int)labels[i] == am) ? 1.f : 0.f;
  }
}

__global__ void softmax_ce_kernel(const float* __restrict__ mat, const float* __restrict__ labels, float* __restrict__ target, int rows, float tiny) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < rows) target[i] = -logf(mat[(size_t)rows * (int)labels[i] + i] + tiny);
}

// ---- Philox-4x32-10 (counter-based; key = (seed, call#), counter = element index / 4) -----------------
struct Philox {
  unsigned k0, k1;
  __device__ void round(unsigned (&c)[4], unsigned ka, unsigned kb) const {
    const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
    const unsigned h0 = (unsigned)(p0 >> 32), l0 = (unsigned)p0, h1 = (unsigned)(p1 >> 32), l1 = (unsigned)p1;
    c[0] = h1 ^ c[1] ^ ka; c[1] = l1; c[2] = h0 ^ c[3] ^ kb; c[3] = l0;
  }
  __device__ void gen(unsigned long long idx, unsigned (&c)[4]) const {
    c[0] = (unsigned)idx; c[1] = (unsigned)(idx >> 32); c[2] = 0x9E3779B9u; c[3] = 0xBB67AE85u;
    unsigned ka = k0, kb = k1;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      round(c, ka, kb);
      ka += 0x9E3779B9u;
      kb += 0xBB67AE85u;
    }
  }
};

__device__ __forceinline__ float u01(unsigned x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }  // [0,1)

// MODE 0: uniform fill; 1: normal fill (Box-Muller); 2: dropout(p,val,scale) in place; 3: bernoulli(target = u < mat);
// 4: relu then dropout(p, 0, scale)
template <int MODE>
__global__ void rng_kernel(float* __restrict__ out, const float* __restrict__ in, size_t n, Philox ph, float p, float val, float scale) {
  const size_t n4 = (n + 3) >> 2;
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (size_t)gridDim.x * blockDim.x) {
    unsigned c[4];
    ph.gen(q, c);
    float u[4] = {u01(c[0]), u01(c[1]), u01(c[2]), u01(c[3])};
    if (MODE == 1) {
      const float r0 = sqrtf(-2.f * logf(1.f - u[0])), r1 = sqrtf(-2.f * logf(1.f - u[2]));
      const float t0 = 6.2831853f * u[1], t1 = 6.2831853f * u[3];
      u[0] = r0 * cosf(t0); u[1] = r0 * sinf(t0); u[2] = r1 * cosf(t1); u[3] = r1 * sinf(t1);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const size_t i = (q << 2) + e;
      if (i >= n) break;
      if (MODE <= 1) out[i] = u[e];
      else if (MODE == 2) out[i] = (p > u[e]) ? val : in[i] * scale;
      else if (MODE == 3) out[i] = (u[e] < in[i]) ? 1.f : 0.f;
      else { const float x = in[i] > 0.f ? in[i] : 0.f; out[i] = (p > u[e]) ? 0.f : x * scale; }
    }
  }
}

struct RngHost { unsigned long long seed, counter; };

template <int MODE>
int rng_launch(rnd_struct* st, float* out, const float* in, size_t n, float p, float val, float scale) {
  if (!st || !st->dev_words) return ERROR_GENERIC;
  RngHost* h = reinterpret_cast<RngHost*>(st->dev_words);
  Philox ph;
  ph.k0 = (unsigned)(h->seed * 0x9E3779B97F4A7C15ull >> 32) ^ (unsigned)h->counter;
  ph.k1 = (unsigned)h->seed ^ (unsigned)(h->counter >> 32) ^ 0x85EBCA6Bu;
  h->counter++;
  if (n == 0) return 0;
  KernelTimer timer(MODE == 2 ? "rng_kernel<dropout>" : MODE == 4 ? "rng_kernel<relu_dropout>" : "rng_kernel", "rng", 0.0, 4.0 * n * (in ? 2 : 1));
  hipLaunchKernelGGL(rng_kernel<MODE>, dim3(blocks_for(n / 4 + 1)), dim3(kThreads), 0, stream(), out, in, n, ph, p, val, scale);
  return launch_status();
}

// One fused pass of SGDOptimizer::Optimize.  Explicit __fmul_rn/__fadd_rn keep each reference
// statement a separately rounded fp32 op (no fma contraction), so the update is bit-identical
// to the reference's multi-pass sequence.
__device__ __forceinline__ void sgd_one(float& g, float& w, float& h, float l2, float clip, float eps, float mom) {
  // hipcc defaults to -ffp-contract=fast, and __fmul_rn/__fadd_rn are header functions compiled
  // under that default (they still fuse).  Plain operators under contract(off) do not.
#pragma clang fp contract(off)
  if (l2 > 0.f) {
    const float t = w * l2;
    g = g + t;
  }
  if (clip > 0.f) g = g > clip ? clip : (g < -clip ? -clip : g);
  g = g * eps;
  const float hm = h * mom;
  h = hm + g;
  w = w - h;
}

__device__ __forceinline__ void sgd_range(float* __restrict__ g, float* __restrict__ w, float* __restrict__ h, size_t n, bool vec, float l2, float clip,
                                          float eps, float mom, size_t tid, size_t stride) {
  size_t done = 0;
  if (vec) {
    const size_t n4 = n >> 2;
    for (size_t i = tid; i < n4; i += stride) {
      f32x4 gv = reinterpret_cast<f32x4*>(g)[i], wv = reinterpret_cast<f32x4*>(w)[i], hv = reinterpret_cast<f32x4*>(h)[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = gv[e], b = wv[e], c = hv[e];
        sgd_one(a, b, c, l2, clip, eps, mom);
        gv[e] = a; wv[e] = b; hv[e] = c;
      }
      reinterpret_cast<f32x4*>(g)[i] = gv;
      reinterpret_cast<f32x4*>(w)[i] = wv;
      reinterpret_cast<f32x4*>(h)[i] = hv;
    }
    done = n4 << 2;
  }
  for (size_t i = done + tid; i < n; i += stride) sgd_one(g[i], w[i], h[i], l2, clip, eps, mom);
}
__global__ void sgd_kernel(float* __restrict__ g, float* __restrict__ w, float* __restrict__ h, size_t n, bool vec, float l2, float clip,
                           float eps, float mom) {
  sgd_range(g, w, h, n, vec, l2, clip, eps, mom, (size_t)blockIdx.x * blockDim.x + threadIdx.x, (size_t)gridDim.x * blockDim.x);
}
// Several tensors in ONE launch (sgd_momentum_step_multi): AlexNet's step updates 13 small tensors (the convolution banks and every
// bias) with ~7 us launches of a few MB each; blockIdx.y picks the tensor, every tensor with its own hyper-parameters.  Same sgd_one per
// element: bit-identical to one sgd_momentum_step per tensor.
constexpr int kSgdMulti = 16;
struct SgdItem {
  float *g, *w, *h;
  unsigned long long n;
  float l2, clip, eps, mom;
  int vec, pad_;
};
struct SgdBatch {
  SgdItem it[kSgdMulti];
};
__global__ void sgd_multi_kernel(const SgdBatch b) {
  const SgdItem& t = b.it[blockIdx.y];
  sgd_range(t.g, t.w, t.h, (size_t)t.n, t.vec != 0, t.l2, t.clip, t.eps, t.mom, (size_t)blockIdx.x * blockDim.x + threadIdx.x,
            (size_t)gridDim.x * blockDim.x);
}

}  // namespace chip

using namespace chip;

extern "C" {

// ---- deferred epilogues (common.h: PendingOp): does this element-wise call continue the parked one? ---------------------------------
namespace {
inline bool same_mat(const cudamat* a, const cudamat& b) { return a->on_device && a->data_device == b.data_device && numel(a) == numel(&b); }
// add_row_vec(T viewed as (N*M, F), bias(1, F), T) behind convUp(..., T): the shared bias of src/conv_edge.cc:145-148
bool absorb_bias(cudamat* mat, cudamat* vec, cudamat* target) {
  PendingOp& o = pending();
  if (o.kind != 1 || o.has_bias || o.relu) return false;   // (mat and target are both checked against the parked call's target below)
  const int F = o.desc.num_output_channels;
  if (!same_mat(mat, o.m[2]) || !same_mat(target, o.m[2]) || mat->is_trans || !vec->on_device || (int)numel(vec) != F || mat->size[1] != F) return false;
  o.bias = *vec;
  o.has_bias = 1;
  ++g_absorbed;
  return true;
}
// lower_bound_scalar(T, 0, T) behind convUp(..., T) [+ bias] or ResponseNormCrossMap(..., T): the ReLU of src/layer.cc:549-551
bool absorb_relu(cudamat* mat, float val, cudamat* target) {
  PendingOp& o = pending();
  if ((o.kind != 1 && o.kind != 4) || o.relu || val != 0.f) return false;
  const cudamat& t = o.kind == 1 ? o.m[2] : o.m[1];
  if (!same_mat(mat, t) || !same_mat(target, t)) return false;
  o.relu = 1;
  flush_pending();   // nothing more can join
  ++g_absorbed;
  return true;
}
// apply_rectified_linear_deriv(D, S, D) behind convDown(..., D) or MaxPoolUndo(images = S, ..., D): the ReLU' of src/layer.cc:556-558
bool absorb_relu_deriv(cudamat* deriv, cudamat* state, cudamat* target) {
  PendingOp& o = pending();
  if ((o.kind != 2 && o.kind != 3) || o.has_mask || !state->on_device || numel(state) != numel(deriv)) return false;
  const cudamat& t = o.kind == 2 ? o.m[2] : o.m[3];
  if (!same_mat(deriv, t) || !same_mat(target, t)) return false;
  if (o.kind == 3 && !same_mat(state, o.m[0])) return false;   // the pooling kernel masks with its own input
  o.mask = *state;
  o.has_mask = 1;
  flush_pending();
  ++g_absorbed;
  return true;
}
}  // namespace

int add_row_vec(cudamat* mat, cudamat* vec, cudamat* target) {
  if (absorb_bias(mat, vec, target)) return 0;
  return add_row(mat, vec, target, 1.0f);
}
int add_row_mult(cudamat* mat, cudamat* vec, cudamat* target, float mult) { return add_row(mat, vec, target, mult); }

int sum_by_axis(cudamat* mat, cudamat* target, int axis, float mult, float p) { return axis_sum<false>(mat, target, axis, mult, p); }
int sqsum_by_axis(cudamat* mat, cudamat* target, int axis, float mult, float p) { return axis_sum<true>(mat, target, axis, mult, p); }

float sum_all(cudamat* mat, int* err_code) { return reduce_all<0>(mat, nullptr, err_code); }
float vdot(cudamat* mat1, cudamat* mat2, int* err_code) {
  if (numel(mat1) != numel(mat2)) { *err_code = ERROR_INCOMPATIBLE_DIMENSIONS; return 0.f; }
  return reduce_all<1>(mat1, mat2, err_code);
}
float euclid_norm(cudamat* mat, int* err_code) { return sqrtf(reduce_all<2>(mat, nullptr, err_code)); }

int normlimit_by_axis(cudamat* mat, cudamat* target, int axis, float norm, int constraint) {
  if (!mat->on_device || !target->on_device) return ERROR_NOT_ON_DEVICE;
  if (mat->is_trans) return ERROR_TRANSPOSED;
  if (mat->size[0] != target->size[0] || mat->size[1] != target->size[1]) return ERROR_INCOMPATIBLE_DIMENSIONS;
  const int rows = mat->size[0], cols = mat->size[1];
  if (axis == 0)
    hipLaunchKernelGGL(normlimit_cols_kernel, dim3(cols), dim3(256), 0, stream(), mat->data_device, target->data_device, rows, norm, constraint);
  else
    return rows_normlimit(nullptr, mat->data_device, target->data_device, nullptr, rows, cols, norm, constraint, false, 0.f, 0.f, 0.f, 0.f);
  return launch_status();
}

// SGDOptimizer::Optimize + ApplyConstraints (src/optimizer.cc:174-200,75-81) for one (rows x cols) tensor:
// the update pass also produces the per-row norms the constraint needs.
int sgd_momentum_step_normlimit(cudamat* grad, cudamat* param, cudamat* history, float l2_decay, float gradient_clip, float epsilon,
                                float momentum, float norm, int constraint) {
  const size_t n = numel(param);
  if (!grad->on_device || !param->on_device || !history->on_device) return ERROR_NOT_ON_DEVICE;
  if (numel(grad) != n || numel(history) != n) return ERROR_INCOMPATIBLE_DIMENSIONS;
  if (param->is_trans) return ERROR_TRANSPOSED;
  if (n == 0) return 0;
  return rows_normlimit(grad->data_device, param->data_device, param->data_device, history->data_device, param->size[0], param->size[1], norm,
                        constraint, true, l2_decay, gradient_clip, epsilon, momentum);
}

int lower_bound_scalar(cudamat* mat, float val, cudamat* target) {
  if (absorb_relu(mat, val, target)) return 0;
  return map2(target, mat, nullptr, [val] __device__(float x, float) { return x > val ? x : val; });
}
int upper_bound_mod_scalar(cudamat* mat, float val, cudamat* target) {
  return map2(target, mat, nullptr, [val] __device__(float x, float) { return x > val ? val : (x < -val ? -val : x); });
}
int apply_rectified_linear_deriv(cudamat* mat1, cudamat* mat2, cudamat* target) {
  if (absorb_relu_deriv(mat1, mat2, target)) return 0;
  return map2(target, mat1, mat2, [] __device__(float d, float s) { return s > 0.f ? d : 0.f * d; });
}
int assign_scalar(cudamat* mat, float alpha) {
  return map2(mat, mat, nullptr, [alpha] __device__(float, float) { return alpha; });
}
int add_scalar(cudamat* mat, float alpha, cudamat* target) {
  return map2(target, mat, nullptr, [alpha] __device__(float x, float) { return x + alpha; });
}
int mult_by_scalar(cudamat* mat, float alpha, cudamat* target, float scale_targets) {
  if (scale_targets == 0.f) return map2(target, mat, nullptr, [alpha] __device__(float x, float) { return x * alpha; });
  return map2(target, mat, target, [alpha, scale_targets] __device__(float x, float t) { return scale_targets * t + x * alpha; });
}
int divide_by_scalar(cudamat* mat, float alpha, cudamat* target) {
  return map2(target, mat, nullptr, [alpha] __device__(float x, float) { return x / alpha; });
}
int add_mult(cudamat* mat1, cudamat* mat2, float alpha) {
  return map2(mat1, mat1, mat2, [alpha] __device__(float x, float y) { return x + alpha * y; });
}
int add_elementwise(cudamat* mat1, cudamat* mat2, cudamat* target) {
  return map2(target, mat1, mat2, [] __device__(float x, float y) { return x + y; });
}
int subtract_elementwise(cudamat* mat1, cudamat* mat2, cudamat* target) {
  return map2(target, mat1, mat2, [] __device__(float x, float y) { return x - y; });
}
int mult_elementwise(cudamat* mat1, cudamat* mat2, cudamat* target, float scale_targets) {
  if (scale_targets != 0.f) return ERROR_UNSUPPORTED;
  return map2(target, mat1, mat2, [] __device__(float x, float y) { return x * y; });
}
int apply_sqrt(cudamat* mat, cudamat* target) {
  return map2(target, mat, nullptr, [] __device__(float x, float) { return sqrtf(x); });
}

int softmax_row_major_multi(cudamat* mat, int numslices, cudamat* target) {
  if (!mat->on_device || !target->on_device) return ERROR_NOT_ON_DEVICE;
  if (mat->is_trans) return ERROR_TRANSPOSED;
  const size_t len = numel(mat);
  if (numslices <= 0 || len % numslices != 0 || numel(target) != len) return ERROR_INCOMPATIBLE_DIMENSIONS;
  const int rows = (int)(len / numslices), cols = numslices;
  SoftmaxOut o{target->data_device, nullptr, nullptr};
  hipLaunchKernelGGL(softmax_rows_kernel<32>, dim3(divup(rows, 32)), dim3(1024), 0, stream(), mat->data_device, nullptr, o, rows, cols, 1.0f);
  return launch_status();
}
int softmax_row_major(cudamat* mat, cudamat* target) { return softmax_row_major_multi(mat, mat->size[1], target); }

int apply_softmax_grad_row_major(cudamat* mat, cudamat* labels, cudamat* target) {
  if (!mat->on_device || !labels->on_device || !target->on_device) return ERROR_NOT_ON_DEVICE;
  if (numel(labels) != (size_t)mat->size[0] || numel(target) != numel(mat)) return ERROR_INCOMPATIBLE_DIMENSIONS;
  if (target->data_device != mat->data_device) {
    int rc = copy_on_device(mat, target);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(softmax_grad_kernel, dim3(divup(mat->size[0], 256)), dim3(256), 0, stream(), labels->data_device, target->data_device, mat->size[0]);
  return launch_status();
}

int get_softmax_correct_row_major(cudamat* mat, cudamat* labels, cudamat* target) {
  if (!mat->on_device || !labels->on_device || !target->on_device) return ERROR_NOT_ON_DEVICE;
  if (mat->is_trans) return ERROR_TRANSPOSED;
  if (target->size[0] != mat->size[0] || target->size[1] != 1 || numel(labels) != (size_t)mat->size[0]) return ERROR_INCOMPATIBLE_DIMENSIONS;
  hipLaunchKernelGGL(softmax_correct_kernel, dim3(divup(mat->size[0], 32)), dim3(1024), 0, stream(), mat->data_device, labels->data_device,
                     target->data_device, mat->size[0], mat->size[1]);
  return launch_status();
}

int get_softmax_cross_entropy_row_major(cudamat* mat, cudamat* labels, cudamat* target, float tiny) {
  if (!mat->on_device || !labels->on_device || !target->on_device) return ERROR_NOT_ON_DEVICE;
  if (mat->is_trans) return ERROR_TRANSPOSED;
  if (target->size[0] != mat->size[0] || target->size[1] != 1 || labels->size[0] != mat->size[0] || labels->size[1] != 1)
    return ERROR_INCOMPATIBLE_DIMENSIONS;
  hipLaunchKernelGGL(softmax_ce_kernel, dim3(divup(mat->size[0], 256)), dim3(256), 0, stream(), mat->data_device, labels->data_device,
                     target->data_device, mat->size[0], tiny);
  return launch_status();
}

int softmax_ce_grad_correct(cudamat* logits, cudamat* labels, cudamat* probs, cudamat* deriv, cudamat* correct_accum, float deriv_scale) {
  if (!logits->on_device || !labels->on_device || !probs->on_device) return ERROR_NOT_ON_DEVICE;
  if (numel(probs) != numel(logits) || (deriv && numel(deriv) != numel(logits)) || numel(labels) != (size_t)logits->size[0])
    return ERROR_INCOMPATIBLE_DIMENSIONS;
  SoftmaxOut o{probs->data_device, deriv ? deriv->data_device : nullptr, correct_accum ? correct_accum->data_device : nullptr};
  KernelTimer timer("softmax_rows_kernel", "softmax_ce", 0.0, 4.0 * numel(logits) * (deriv ? 3 : 2));
  hipLaunchKernelGGL(softmax_rows_kernel<32>, dim3(divup(logits->size[0], 32)), dim3(1024), 0, stream(), logits->data_device, labels->data_device, o,
                     logits->size[0], logits->size[1], deriv_scale);
  return launch_status();
}

// g += l2*w; clip; g *= eps; h = mom*h + g; w -= h   (src/optimizer.cc:174-200, one pass, same op order)
int sgd_momentum_step(cudamat* grad, cudamat* param, cudamat* history, float l2_decay, float gradient_clip, float epsilon, float momentum) {
  const size_t n = numel(param);
  if (!grad->on_device || !param->on_device || !history->on_device) return ERROR_NOT_ON_DEVICE;
  if (numel(grad) != n || numel(history) != n) return ERROR_INCOMPATIBLE_DIMENSIONS;
  if (n == 0) return 0;
  const bool vec = al16(grad->data_device) && al16(param->data_device) && al16(history->data_device);
  KernelTimer timer("sgd_kernel", "sgd", 0.0, 20.0 * n);   // reads g, w, h; writes h, w (>= 20 bytes per parameter)
  hipLaunchKernelGGL(sgd_kernel, dim3(blocks_for(n / 4 + 1)), dim3(kThreads), 0, stream(), grad->data_device, param->data_device,
                     history->data_device, n, vec, l2_decay, gradient_clip, epsilon, momentum);
  return launch_status();
}

// sgd_momentum_step on `count` tensors, each with its own hyper-parameters, in ceil(count / 16) launches
int sgd_momentum_step_multi(int count, cudamat** grads, cudamat** params, cudamat** histories, const float* l2_decay, const float* gradient_clip,
                            const float* epsilon, const float* momentum) {
  if (count < 0 || (count > 0 && (!grads || !params || !histories || !l2_decay || !gradient_clip || !epsilon || !momentum))) return ERROR_GENERIC;
  for (int i = 0; i < count; ++i) {
    if (!grads[i]->on_device || !params[i]->on_device || !histories[i]->on_device) return ERROR_NOT_ON_DEVICE;
    if (numel(grads[i]) != numel(params[i]) || numel(histories[i]) != numel(params[i])) return ERROR_INCOMPATIBLE_DIMENSIONS;
  }
  for (int base = 0; base < count; base += kSgdMulti) {
    SgdBatch b{};
    int m = 0;
    size_t most = 0, total = 0;
    for (int i = base; i < count && i < base + kSgdMulti; ++i) {
      const size_t n = numel(params[i]);
      if (n == 0) continue;
      SgdItem& t = b.it[m++];
      t.g = grads[i]->data_device; t.w = params[i]->data_device; t.h = histories[i]->data_device;
      t.n = n;
      t.l2 = l2_decay[i]; t.clip = gradient_clip[i]; t.eps = epsilon[i]; t.mom = momentum[i];
      t.vec = al16(t.g) && al16(t.w) && al16(t.h);
      most = n > most ? n : most;
      total += n;
    }
    if (m == 0) continue;
    KernelTimer timer("sgd_multi_kernel", "sgd", 0.0, 20.0 * total);
    hipLaunchKernelGGL(sgd_multi_kernel, dim3(blocks_for(most / 4 + 1), m), dim3(kThreads), 0, stream(), b);
  }
  return launch_status();
}

int init_random(rnd_struct* rnd_state, int seed) {
  RngHost* h = new RngHost{(unsigned long long)(unsigned)seed, 0ull};
  rnd_state->dev_mults = nullptr;
  rnd_state->dev_words = reinterpret_cast<unsigned long long*>(h);
  return 0;
}
int fill_with_rand(rnd_struct* st, cudamat* mat) { return rng_launch<0>(st, mat->data_device, nullptr, numel(mat), 0, 0, 0); }
int fill_with_randn(rnd_struct* st, cudamat* mat) { return rng_launch<1>(st, mat->data_device, nullptr, numel(mat), 0, 0, 0); }
int sample_bernoulli(rnd_struct* st, cudamat* mat, cudamat* target) {
  if (numel(mat) != numel(target)) return ERROR_INCOMPATIBLE_DIMENSIONS;
  return rng_launch<3>(st, target->data_device, mat->data_device, numel(mat), 0, 0, 0);
}
int dropout(rnd_struct* st, cudamat* mat, float dropprob, float val, float scale) {
  return rng_launch<2>(st, mat->data_device, mat->data_device, numel(mat), dropprob, val, scale);
}
int relu_dropout(rnd_struct* st, cudamat* mat, float dropprob, float scale) {
  return rng_launch<4>(st, mat->data_device, mat->data_device, numel(mat), dropprob, 0.f, scale);
}

}  // extern "C"
