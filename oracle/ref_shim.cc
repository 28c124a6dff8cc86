// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
//
// extern "C" trampoline around the *unmodified* reference CPU path
// (eigenmat/eigenmat.cc, eigenmat/cpumat_conv.cc, src/CPUMatrix.cc) compiled from where those
// sources lie under /root/reference by oracle/Makefile into oracle/_ref/libconvnet_ref.so.
// Nothing from the reference is copied here: this file only packs raw float pointers into the
// reference's own `eigenmat` / `Matrix` / `ConvDesc` / `Shape4D` types and calls its functions.
//
// Used for: (1) pinning oracle/convnet_oracle.c (the portable restatement), (2) generating
// tests/golden/*.npz, (3) bench.py's cpu_baseline leg (kind "reference").
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may load the result.
#include <cstring>
#include <cstdlib>
#include <new>
#include <cstdio>
#include <string>

#include "CPUMatrix.h"   // reference src/ (class Matrix, CPU flavour)
#include "eigenmat.h"    // reference eigenmat/
#include "cpumat_conv.h"

// ---- the four symbols src/util.h would have provided (util.h needs protobuf + CImg) ----------
std::string GetStringError(int err_code) {
  char buf[64];
  snprintf(buf, sizeof buf, "eigenmat error %d", err_code);
  return std::string(buf);
}
void WriteHDF5CPU(hid_t, float*, int, int, const std::string&) {}
void ReadHDF5CPU(hid_t, float*, int, const std::string&) {}
void ReadHDF5Shape(hid_t, const std::string&, int*, int*) {}

// The reference's convUp/convDown multiply an UNINITIALISED `new float[]` temp by beta=0
// (eigenmat/cpumat_conv.cc:186-201 + eigenmat/eigenmat.cc:2296 `beta * C[i]`): heap garbage that
// happens to be NaN/Inf turns into NaN outputs.  We cannot (and must not) edit the reference, so
// this library zero-fills array-new instead (linked -Bsymbolic so the reference objects bind to
// these definitions); 0*0 is what the algorithm means.
void* operator new[](size_t n) {
  void* p = calloc(1, n ? n : 1);
  if (!p) abort();
  return p;
}
void operator delete[](void* p) noexcept { free(p); }
void operator delete[](void* p, size_t) noexcept { free(p); }

namespace {

eigenmat wrap(float* p, int rows, int cols, int trans = 0) {
  eigenmat m;
  m.data = p;
  m.size[0] = rows;
  m.size[1] = cols;
  m.is_trans = trans;
  m.owns_data = 0;
  return m;
}

Shape4D shape4(int a, int b, int c, int d) {
  Shape4D s;
  s.shape[0] = a; s.shape[1] = b; s.shape[2] = c; s.shape[3] = d;
  return s;
}

// desc[] = {C, F, Ky, Kx, sy, sx, pady, padx}; pads are the *negated* values the reference
// stores in ConvDesc (src/edge.cc:97-99).
ConvDesc make_desc(const int* d) {
  ConvDesc cd;
  memset(&cd, 0, sizeof cd);
  cd.num_input_channels = d[0];
  cd.num_output_channels = d[1];
  cd.kernel_size_y = d[2];
  cd.kernel_size_x = d[3];
  cd.kernel_size_t = 1;
  cd.stride_y = d[4];
  cd.stride_x = d[5];
  cd.stride_t = 1;
  cd.padding_y = d[6];
  cd.padding_x = d[7];
  cd.padding_t = 0;
  cd.input_channel_begin = 0;
  cd.input_channel_end = d[0];
  cd.output_channel_begin = 0;
  cd.output_channel_end = d[1];
  cd.num_groups = 1;
  return cd;
}

// A reference `Matrix` holding a copy of caller data (its members are private; it owns storage).
struct Held {
  Matrix m;
  float* user;
  size_t n;
  Held(float* p, int rows, int cols, int s0, int s1, int s2, int s3) : user(p), n((size_t)rows * cols) {
    m.AllocateGPUMemory(rows, cols);
    memcpy(m.GetHostData(), p, n * sizeof(float));
    m.SetShape4D(s0, s1, s2, s3);
  }
  void back() { memcpy(user, m.GetHostData(), n * sizeof(float)); }
};

}  // namespace

extern "C" {

void ref_conv_up(float* images, float* filters, float* targets, int N, int H, int W, int My,
                 int Mx, const int* desc, float scaleTargets, float scaleOutput) {
  ConvDesc cd = make_desc(desc);
  const int C = desc[0], F = desc[1], Ky = desc[2], Kx = desc[3];
  eigenmat im = wrap(images, N, H * W * C), fl = wrap(filters, F, Ky * Kx * C),
           tg = wrap(targets, N, My * Mx * F);
  Shape4D si = shape4(N, W, H, C), sf = shape4(F, Kx, Ky, C), st = shape4(N, Mx, My, F);
  convUp(&im, &fl, &tg, si, sf, st, cd, scaleTargets, scaleOutput, true);
}

void ref_conv_down(float* derivs, float* filters, float* targets, int N, int H, int W, int My,
                   int Mx, const int* desc, float scaleTargets, float scaleOutput) {
  ConvDesc cd = make_desc(desc);
  const int C = desc[0], F = desc[1], Ky = desc[2], Kx = desc[3];
  eigenmat dv = wrap(derivs, N, My * Mx * F), fl = wrap(filters, F, Ky * Kx * C),
           tg = wrap(targets, N, H * W * C);
  Shape4D sd = shape4(N, Mx, My, F), sf = shape4(F, Kx, Ky, C), st = shape4(N, W, H, C);
  convDown(&dv, &fl, &tg, sd, sf, st, cd, scaleTargets, scaleOutput, true);
}

void ref_conv_outp(float* images, float* derivs, float* targets, int N, int H, int W, int My,
                   int Mx, const int* desc, float scaleTargets, float scaleOutput) {
  ConvDesc cd = make_desc(desc);
  const int C = desc[0], F = desc[1], Ky = desc[2], Kx = desc[3];
  eigenmat im = wrap(images, N, H * W * C), dv = wrap(derivs, N, My * Mx * F),
           tg = wrap(targets, F, Ky * Kx * C);
  Shape4D si = shape4(N, W, H, C), sd = shape4(N, Mx, My, F), st = shape4(F, Kx, Ky, C);
  convOutp(&im, &dv, &tg, si, sd, st, cd, scaleTargets, scaleOutput, true);
}

// Pooling lives in the reference's CPU `Matrix` class (src/CPUMatrix.cc:574-827).
void ref_max_pool(float* images, float* targets, int N, int H, int W, int My, int Mx,
                  const int* desc) {
  ConvDesc cd = make_desc(desc);
  const int C = desc[0];
  Held in(images, N, H * W * C, N, W, H, C), out(targets, N, My * Mx * C, N, Mx, My, C);
  Matrix::ConvMaxPool(in.m, out.m, cd);
  out.back();
}

void ref_avg_pool(float* images, float* targets, int N, int H, int W, int My, int Mx,
                  const int* desc) {
  ConvDesc cd = make_desc(desc);
  const int C = desc[0];
  Held in(images, N, H * W * C, N, W, H, C), out(targets, N, My * Mx * C, N, Mx, My, C);
  Matrix::ConvAvgPool(in.m, out.m, cd);
  out.back();
}

// images: layer input (N,H,W,C); maxGrads/maxActs: pooled grid (N,My,Mx,C); targets like images.
void ref_max_pool_undo(float* images, float* maxGrads, float* maxActs, float* targets, int N, int H,
                       int W, int My, int Mx, const int* desc, float scaleTargets) {
  ConvDesc cd = make_desc(desc);
  const int C = desc[0];
  Held in(images, N, H * W * C, N, W, H, C), dout(maxGrads, N, My * Mx * C, N, Mx, My, C),
      out(maxActs, N, My * Mx * C, N, Mx, My, C), din(targets, N, H * W * C, N, W, H, C);
  Matrix::ConvMaxPoolUndo(in.m, dout.m, out.m, din.m, cd, scaleTargets);
  din.back();
}

void ref_avg_pool_undo(float* avgGrads, float* targets, int N, int H, int W, int My, int Mx,
                       const int* desc, float scaleTargets) {
  ConvDesc cd = make_desc(desc);
  const int C = desc[0];
  Held dout(avgGrads, N, My * Mx * C, N, Mx, My, C), din(targets, N, H * W * C, N, W, H, C);
  Matrix::ConvAvgPoolUndo(dout.m, din.m, cd, scaleTargets, 1.0f);
  din.back();
}

void ref_rnorm(float* images, float* targets, int N, int locs_per_image, int C, int sizeF,
               float addScale, float powScale, int blocked) {
  eigenmat im = wrap(images, N, locs_per_image * C), tg = wrap(targets, N, locs_per_image * C);
  ResponseNormCrossMap(&im, &tg, C, sizeF, addScale, powScale, blocked != 0);
}

void ref_rnorm_undo(float* outGrads, float* inputs, float* targets, int N, int locs_per_image,
                    int C, int sizeF, float addScale, float powScale, int blocked) {
  eigenmat og = wrap(outGrads, N, locs_per_image * C), in = wrap(inputs, N, locs_per_image * C),
           tg = wrap(targets, N, locs_per_image * C);
  ResponseNormCrossMapUndo(&og, &in, &tg, C, sizeF, addScale, powScale, blocked != 0);
}

// target = beta*target + alpha*op(a)*op(b)   (eigenmat dot(); note Matrix::Dot swaps the names)
int ref_dot(float* a, int a_rows, int a_cols, int a_trans, float* b, int b_rows, int b_cols,
            int b_trans, float* target, int t_rows, int t_cols, float beta, float alpha) {
  eigenmat ma = wrap(a, a_rows, a_cols, a_trans), mb = wrap(b, b_rows, b_cols, b_trans),
           mt = wrap(target, t_rows, t_cols);
  return dot(&ma, &mb, &mt, beta, alpha);
}

int ref_add_row_vec(float* mat, float* vec, int rows, int cols) {
  eigenmat m = wrap(mat, rows, cols), v = wrap(vec, 1, cols);
  return add_row_vec(&m, &v, &m);
}

// target = p*target + mult*sum(mat, axis)
int ref_sum_by_axis(float* mat, int rows, int cols, float* target, int axis, float mult, float p) {
  eigenmat m = wrap(mat, rows, cols);
  eigenmat t = axis == 0 ? wrap(target, 1, cols) : wrap(target, rows, 1);
  return sum_by_axis(&m, &t, axis, mult, p);
}

int ref_lower_bound_scalar(float* mat, int len, float val) {
  eigenmat m = wrap(mat, len, 1);
  return lower_bound_scalar(&m, val, &m);
}

int ref_upper_bound_mod_scalar(float* mat, int len, float val) {
  eigenmat m = wrap(mat, len, 1);
  return upper_bound_mod_scalar(&m, val, &m);
}

int ref_relu_deriv(float* deriv, float* state, int len) {
  eigenmat d = wrap(deriv, len, 1), s = wrap(state, len, 1);
  return apply_rectified_linear_deriv(&d, &s, &d);
}

int ref_softmax_row_major(float* mat, int rows, int cols) {
  eigenmat m = wrap(mat, rows, cols);
  return softmax_row_major(&m);
}

int ref_softmax_grad_row_major(float* mat, float* labels, float* target, int rows, int cols) {
  eigenmat m = wrap(mat, rows, cols), l = wrap(labels, rows, 1), t = wrap(target, rows, cols);
  return apply_softmax_grad_row_major(&m, &l, &t);
}

int ref_softmax_correct_row_major(float* mat, float* labels, float* target, int rows, int cols) {
  eigenmat m = wrap(mat, rows, cols), l = wrap(labels, rows, 1), t = wrap(target, rows, 1);
  return get_softmax_correct_row_major(&m, &l, &t);
}

int ref_softmax_ce_row_major(float* mat, float* labels, float* target, int rows, int cols,
                             float tiny) {
  eigenmat m = wrap(mat, rows, cols), l = wrap(labels, rows, 1), t = wrap(target, rows, 1);
  return get_softmax_cross_entropy_row_major(&m, &l, &t, tiny);
}

int ref_normlimit_by_axis(float* mat, int rows, int cols, int axis, float norm, int constraint) {
  eigenmat m = wrap(mat, rows, cols);
  return normlimit_by_axis(&m, &m, axis, norm, constraint);
}

int ref_add_mult(float* a, float* b, int len, float alpha) {
  eigenmat ma = wrap(a, len, 1), mb = wrap(b, len, 1);
  return add_mult(&ma, &mb, alpha);
}

int ref_mult_by_scalar(float* a, int len, float alpha) {
  eigenmat ma = wrap(a, len, 1);
  return mult_by_scalar(&ma, alpha, &ma);
}

// The reference's SGD step is host logic over Matrix ops (src/optimizer.cc:174-200); replay that
// exact op sequence with the reference's own eigenmat primitives (non-Nesterov path).
void ref_sgd_step(float* grad, float* param, float* history, int rows, int cols, float l2_decay,
                  float gradient_clip, float epsilon, float momentum, float norm_limit,
                  float norm_constraint) {
  const int len = rows * cols;
  eigenmat g = wrap(grad, len, 1), p = wrap(param, len, 1), h = wrap(history, len, 1);
  if (l2_decay > 0) add_mult(&g, &p, l2_decay);
  if (gradient_clip > 0) upper_bound_mod_scalar(&g, gradient_clip, &g);
  mult_by_scalar(&g, epsilon, &g);
  mult_by_scalar(&h, momentum, &h);
  add_elementwise(&h, &g, &h);
  add_mult(&p, &h, -1.0f);
  eigenmat p2 = wrap(param, rows, cols);
  if (norm_constraint > 0) normlimit_by_axis(&p2, &p2, 1, norm_constraint, 1);
  else if (norm_limit > 0) normlimit_by_axis(&p2, &p2, 1, norm_limit, 0);
}

// ---- input staging (eigenmat.cc:73,325-370,499-560,970-1005,1962-1988,2046-2090) ------------------------------
// images: (dims, num_images) one case per column; patches: (num_images, colours*ph*pw) = CHWN batch
int ref_extract_patches(float* images, int dims, int num_images, float* patches, float* wo, float* ho, float* flip, int img_w,
                        int img_h, int pw, int ph) {
  const int colors = dims / (img_w * img_h);
  eigenmat im = wrap(images, dims, num_images), pt = wrap(patches, num_images, colors * pw * ph), w = wrap(wo, 1, num_images),
           h = wrap(ho, 1, num_images), f = wrap(flip, 1, num_images);
  return extract_patches(&im, &pt, &w, &h, &f, img_w, img_h, pw, ph);
}
int ref_shuffle_columns(float* mat, int rows, int cols, float* perm) {
  eigenmat m = wrap(mat, rows, cols), p = wrap(perm, 1, cols);
  return shuffleColumns(&m, &p);
}
int ref_add_col_mult(float* mat, int rows, int cols, float* vec, float mult) {
  eigenmat m = wrap(mat, rows, cols), v = wrap(vec, rows, 1);
  return add_col_mult(&m, &v, &m, mult);
}
int ref_div_by_col_vec(float* mat, int rows, int cols, float* vec) {
  eigenmat m = wrap(mat, rows, cols), v = wrap(vec, rows, 1);
  return div_by_col_vec(&m, &v, &m);
}
int ref_mult_by_row_vec(float* mat, int rows, int cols, float* vec) {
  eigenmat m = wrap(mat, rows, cols), v = wrap(vec, 1, cols);
  return mult_by_row_vec(&m, &v, &m);
}
int ref_normalize_columns(float* mat, int rows, int cols) {
  eigenmat m = wrap(mat, rows, cols);
  return normalize_by_axis(&m, &m, 0);
}
int ref_add_to_each_pixel(float* mat1, int rows, int cols, float* mat2, int colors, float mult) {
  eigenmat a = wrap(mat1, rows, cols), b = wrap(mat2, rows, colors);
  return add_to_each_pixel(&a, &b, &a, mult);
}
int ref_copy_transpose(float* src, int rows, int cols, float* dst) {
  eigenmat a = wrap(src, rows, cols), b = wrap(dst, cols, rows);
  return copy_transpose(&a, &b);
}

int ref_version() { return 2; }

}  // extern "C"
