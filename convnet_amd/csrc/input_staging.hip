// GPU-side input staging: the step right before the hot path.  The reference's DataHandler keeps a chunk
// of the dataset on the GPU as a (dims, cases) matrix — one case per COLUMN, [colour][row][col] contiguous — and every
// GetBatch turns a slice of it into the CHWN batch the conv kernels read (src/datahandler.cc:146-198,496-532):
//   extract_patches  random crop + horizontal flip + transpose to CHWN   (cudamat.cu:2699-2742, kExtractPatches2 :1655)
//   copy_transpose   the no-jitter case                                   (state.hip)
//   shuffleColumns   in-place pairwise column swaps by a permutation      (cudamat.cu:2655, kShuffleColumns :947)
//   add_col_vec / add_col_mult / div_by_col_vec / normalize_by_axis       mean/std normalisation (DataIterator::Preprocess)
//   add_to_each_pixel / mult_by_row_vec / div_by_row_vec                  PCA colour noise (DataIterator::AddPCANoise)
// Semantics pinned by eigenmat/eigenmat.cc:325-370,499-560,970-1005,1962-1988,2046-2090 (the CPU oracle).
// All HBM-bound byte shuffling: coalesced on both sides (LDS tile transpose where the two sides disagree), no GEMMs.
#include "common.h"

namespace chip {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kThreads = 256;
inline int blocks_for(size_t items) {
  size_t b = (items + kThreads - 1) / kThreads;
  if (b > 4096) b = 4096;
  return b ? (int)b : 1;
}

// tile = 32 images x 32 patch columns of one (colour, patch row).  Read side: lane tx walks the source columns of image
// ty (128 contiguous bytes, reversed when flipped); write side: lane tx walks the images of patch column ty (128
// contiguous bytes of the CHWN batch).
__global__ void __launch_bounds__(256) extract_patches_kernel(const float* __restrict__ images, float* __restrict__ patches,
                                                              const float* __restrict__ wo, const float* __restrict__ ho,
                                                              const float* __restrict__ flip, int N, int W, int H, int pw, int ph,
                                                              int colors) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  const int c0 = blockIdx.x * 32;
  const int row = blockIdx.y % ph, color = blockIdx.y / ph;
  const int n0 = blockIdx.z * 32;
  for (int j = ty; j < 32; j += 8) {
    const int n = n0 + j, dc = c0 + tx;
    if (n < N && dc < pw) {
      int sc = (int)wo[n] + dc;
      if (flip[n] > 0.5f) sc = W - sc - 1;
      const int sr = (int)ho[n] + row;
      tile[j][tx] = images[(size_t)sc + (size_t)W * (sr + (size_t)H * (color + (size_t)colors * n))];
    }
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int dc = c0 + j, n = n0 + tx;
    if (n < N && dc < pw) patches[(size_t)n + (size_t)N * (dc + (size_t)pw * (row + (size_t)ph * color))] = tile[tx][j];
  }
}

// one block per column pair (2j, 2j+1): swap columns idx[2j] and idx[2j+1] of the matrix in place
__global__ void shuffle_columns_kernel(float* __restrict__ m, const float* __restrict__ idx, int height, int width) {
  const int c = 2 * blockIdx.x;
  if (c + 1 >= width) return;   // odd tail: the reference copies the column onto itself
  float* a = m + (size_t)height * (int)idx[c];
  float* b = m + (size_t)height * (int)idx[c + 1];
  if (a == b) return;
  const bool v4 = (height & 3) == 0 && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0;
  if (v4) {
    for (int i = threadIdx.x; i < (height >> 2); i += blockDim.x) {
      const f32x4 x = reinterpret_cast<f32x4*>(a)[i], y = reinterpret_cast<f32x4*>(b)[i];
      reinterpret_cast<f32x4*>(a)[i] = y;
      reinterpret_cast<f32x4*>(b)[i] = x;
    }
  } else {
    for (int i = threadIdx.x; i < height; i += blockDim.x) {
      const float x = a[i], y = b[i];
      a[i] = y;
      b[i] = x;
    }
  }
}

// target[i + h*j] = f(mat[i + h*j], i, j)
template <typename F>
__global__ void rowcol_map_kernel(const float* __restrict__ mat, float* __restrict__ target, int h, size_t n, F f) {
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x) {
    const size_t j = k / h;
    const int i = (int)(k - j * h);
    target[k] = f(mat[k], i, (int)j);
  }
}

template <typename F>
int rowcol_map(cudamat* mat, cudamat* target, F f) {
  const size_t n = numel(mat);
  if (n == 0) return 0;
  hipLaunchKernelGGL(rowcol_map_kernel<F>, dim3(blocks_for(n)), dim3(kThreads), 0, stream(), mat->data_device, target->data_device,
                     mat->size[0], n, f);
  return hipGetLastError() == hipSuccess ? 0 : CUDA_ERROR;
}

// column j: target = mat - mean(mat[:, j])   (normalize_by_axis axis 0, eigenmat.cc:982-997)
__global__ void center_columns_kernel(const float* __restrict__ mat, float* __restrict__ target, int height) {
  __shared__ float sh[4];
  const float* col = mat + (size_t)blockIdx.x * height;
  float* out = target + (size_t)blockIdx.x * height;
  float s = 0.f;
  for (int i = threadIdx.x; i < height; i += blockDim.x) s += col[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  const float mean = ((sh[0] + sh[1]) + (sh[2] + sh[3])) / (float)height;
  for (int i = threadIdx.x; i < height; i += blockDim.x) out[i] = col[i] - mean;
}

int check_colvec(const cudamat* mat, const cudamat* vec, const cudamat* target) {
  if (!mat->on_device || !vec->on_device || !target->on_device) return ERROR_NOT_ON_DEVICE;
  if (mat->is_trans) return ERROR_TRANSPOSED;
  if (mat->size[0] != vec->size[0] || vec->size[1] != 1 || mat->size[0] != target->size[0] || mat->size[1] != target->size[1])
    return ERROR_INCOMPATIBLE_DIMENSIONS;
  return 0;
}
int check_rowvec(const cudamat* mat, const cudamat* vec, const cudamat* target) {
  if (!mat->on_device || !vec->on_device || !target->on_device) return ERROR_NOT_ON_DEVICE;
  if (mat->is_trans) return ERROR_TRANSPOSED;
  if (mat->size[1] != vec->size[1] || vec->size[0] != 1 || mat->size[0] != target->size[0] || mat->size[1] != target->size[1])
    return ERROR_INCOMPATIBLE_DIMENSIONS;
  return 0;
}

}  // namespace
}  // namespace chip

using namespace chip;

extern "C" {

int extract_patches(cudamat* images, cudamat* patches, cudamat* width_offset, cudamat* height_offset, cudamat* flip, int img_width,
                    int img_height, int patch_width, int patch_height) {
  if (!images->on_device || !patches->on_device || !width_offset->on_device || !height_offset->on_device || !flip->on_device)
    return ERROR_NOT_ON_DEVICE;
  if (img_width <= 0 || img_height <= 0 || patch_width <= 0 || patch_height <= 0) return ERROR_INCOMPATIBLE_DIMENSIONS;
  const int num_images = images->size[1];
  const int num_colors = images->size[0] / (img_width * img_height);
  if (patches->size[1] != num_colors * patch_width * patch_height || patches->size[0] != num_images) return ERROR_INCOMPATIBLE_DIMENSIONS;
  if ((int)numel(width_offset) != num_images || (int)numel(height_offset) != num_images || (int)numel(flip) != num_images)
    return ERROR_INCOMPATIBLE_DIMENSIONS;
  if (num_images == 0 || num_colors == 0) return 0;
  if (patch_height * num_colors > 65535 || divup(num_images, 32) > 65535) return ERROR_UNSUPPORTED;
  KernelTimer timer("extract_patches_kernel", "input_staging", 0.0, 8.0 * numel(patches));
  const dim3 grid(divup(patch_width, 32), patch_height * num_colors, divup(num_images, 32));
  hipLaunchKernelGGL(extract_patches_kernel, grid, dim3(256), 0, stream(), images->data_device, patches->data_device,
                     width_offset->data_device, height_offset->data_device, flip->data_device, num_images, img_width, img_height,
                     patch_width, patch_height, num_colors);
  return hipGetLastError() == hipSuccess ? 0 : CUDA_ERROR;
}

int shuffleColumns(cudamat* source, cudamat* rand_perm_indices) {
  if (!source->on_device || !rand_perm_indices->on_device) return ERROR_NOT_ON_DEVICE;
  const int h = source->size[0], w = source->size[1];
  if (rand_perm_indices->size[0] != 1 || rand_perm_indices->size[1] != w) return ERROR_INCOMPATIBLE_DIMENSIONS;
  if (w < 2 || h == 0) return 0;
  KernelTimer timer("shuffle_columns_kernel", "input_staging", 0.0, 8.0 * numel(source));
  hipLaunchKernelGGL(shuffle_columns_kernel, dim3(w / 2), dim3(256), 0, stream(), source->data_device, rand_perm_indices->data_device, h, w);
  return hipGetLastError() == hipSuccess ? 0 : CUDA_ERROR;
}

int add_col_mult(cudamat* mat, cudamat* vec, cudamat* target, float mult) {
  if (const int e = check_colvec(mat, vec, target)) return e;
  const float* v = vec->data_device;
  return rowcol_map(mat, target, [=] __device__(float x, int i, int) { return x + mult * v[i]; });
}
int add_col_vec(cudamat* mat, cudamat* vec, cudamat* target) { return add_col_mult(mat, vec, target, 1.0f); }

int div_by_col_vec(cudamat* mat, cudamat* vec, cudamat* target) {
  if (const int e = check_colvec(mat, vec, target)) return e;
  const float* v = vec->data_device;
  return rowcol_map(mat, target, [=] __device__(float x, int i, int) { return x / v[i]; });
}

int mult_by_row_vec(cudamat* mat, cudamat* vec, cudamat* target) {
  if (const int e = check_rowvec(mat, vec, target)) return e;
  const float* v = vec->data_device;
  return rowcol_map(mat, target, [=] __device__(float x, int, int j) { return x * v[j]; });
}

int div_by_row_vec(cudamat* mat, cudamat* vec, cudamat* target) {
  if (const int e = check_rowvec(mat, vec, target)) return e;
  const float* v = vec->data_device;
  return rowcol_map(mat, target, [=] __device__(float x, int, int j) { return x / v[j]; });
}

// target[i] = mat1[i] + mult * mat2[i % height + height * (i / num_pix)],  num_pix = height*width / num_colors
// (mat1 = a (cases, colours*pixels) batch, mat2 = (cases, colours) per-case colour noise; eigenmat.cc:346-366)
int add_to_each_pixel(cudamat* mat1, cudamat* mat2, cudamat* target, float mult) {
  if (!mat1->on_device || !mat2->on_device || !target->on_device) return ERROR_NOT_ON_DEVICE;
  if (mat1->is_trans || mat2->is_trans) return ERROR_TRANSPOSED;
  if (mat1->size[0] != mat2->size[0] || mat2->size[1] == 0 || mat1->size[1] % mat2->size[1] != 0 || mat1->size[0] != target->size[0] ||
      mat1->size[1] != target->size[1])
    return ERROR_INCOMPATIBLE_DIMENSIONS;
  const int height = mat1->size[0];
  const int cols_per_color = mat1->size[1] / mat2->size[1];   // i / num_pix == j / cols_per_color
  const float* v = mat2->data_device;
  return rowcol_map(mat1, target, [=] __device__(float x, int i, int j) { return x + mult * v[i + (size_t)height * (j / cols_per_color)]; });
}

int normalize_by_axis(cudamat* mat, cudamat* target, int axis) {
  if (!mat->on_device || !target->on_device) return ERROR_NOT_ON_DEVICE;
  if (mat->is_trans) return ERROR_TRANSPOSED;
  if (target->size[0] != mat->size[0] || target->size[1] != mat->size[1]) return ERROR_INCOMPATIBLE_DIMENSIONS;
  if (axis != 0) return ERROR_UNSUPPORTED;   // as eigenmat.cc:999-1002
  if (numel(mat) == 0) return 0;
  hipLaunchKernelGGL(center_columns_kernel, dim3(mat->size[1]), dim3(256), 0, stream(), mat->data_device, target->data_device, mat->size[0]);
  return hipGetLastError() == hipSuccess ? 0 : CUDA_ERROR;
}

}  // extern "C"
