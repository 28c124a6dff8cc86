// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
//
// Seam #1 over seam #2: the reference's own GPU `class Matrix` (src/matrix.{h,cc}, compiled UNMODIFIED
// from where it lies under /root/reference, against the reference's own cudamat headers) linked to THIS repo's
// libconvnet_hip.so.  oracle/Makefile target `seam` builds oracle/_ref/libref_matrix_seam.so; tests/test_reference_seam.py
// drives the reference's Matrix::ConvUp / ConvDown / ConvOutp / ConvMaxPool(Undo) / ConvAvgPool(Undo) /
// ConvResponseNormCrossMap(Undo) / Dot / AddRowVec / SumRows / … through the trampolines below on the MI355X and compares
// with the CPU oracle — i.e. the reference's host code running on the new kernels with no source change.
//
// This file holds the util.h functions matrix.cc calls and the extern "C" trampolines; seam_stubs.c holds abort-stubs for the
// cudamat symbols outside the hot path that this library does not export (ctypes loads with RTLD_NOW).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "matrix.h"   // reference src/matrix.h (GPU flavour, -DUSE_GEMM)
#include "seam_pools.h"

std::string GetStringError(int err_code) {
  char buf[64];
  snprintf(buf, sizeof buf, "cudamat error %d", err_code);
  return std::string(buf);
}
void WriteHDF5CPU(hid_t, float*, int, int, const std::string&) {}
void ReadHDF5CPU(hid_t, float*, int, const std::string&) {}
void ReadHDF5Shape(hid_t, const std::string&, int*, int*) {}

extern "C" int hipGetDevice(int* dev);
extern "C" cudaError_t cudaGetDevice(int* dev) { return hipGetDevice(dev); }

namespace {
void load(Matrix& m, const float* src, int rows, int cols) {
  m.AllocateGPUMemory(rows, cols);
  memcpy(m.GetHostData(), src, sizeof(float) * (size_t)rows * cols);
  m.CopyToDevice();
}
void store(Matrix& m, float* dst) {
  m.CopyToHost();
  memcpy(dst, m.GetHostData(), sizeof(float) * m.GetNumEls());
}
ConvDesc desc(int C, int F, int Ky, int Kx, int sy, int sx, int pady, int padx) {
  ConvDesc d;
  memset(&d, 0, sizeof d);
  d.num_input_channels = C; d.num_output_channels = F;
  d.kernel_size_y = Ky; d.kernel_size_x = Kx; d.kernel_size_t = 1;
  d.stride_y = sy; d.stride_x = sx; d.stride_t = 1;
  d.padding_y = -pady; d.padding_x = -padx; d.padding_t = 0;   // stored negated, src/edge.cc:97-99
  d.input_channel_begin = 0; d.input_channel_end = C; d.output_channel_begin = 0; d.output_channel_end = F;
  d.num_groups = 1;
  return d;
}
}  // namespace

extern "C" {

void seam_init(int device) {
  static bool done = false;
  if (done) return;
  done = true;
  Matrix::SetupCUDADevice(device);   // cuda_set_device + cublas_init + temp/ones bookkeeping (matrix.cc:486-532)
  SeamPools::ReleaseAtExit();
}

// op 0: ConvUp(images a, filters b); 1: ConvDown(derivs a, filters b); 2: ConvOutp(images a, derivs b).  `out` carries the
// initial target when scale_targets != 0.
void seam_conv(int op, const float* a, const float* b, float* out, int N, int C, int H, int W, int F, int Ky, int Kx, int sy, int sx,
               int pady, int padx, int My, int Mx, float scale_targets, float scale_outputs) {
  const ConvDesc d = desc(C, F, Ky, Kx, sy, sx, pady, padx);
  Matrix x, w, y;
  const int K = Ky * Kx * C;
  if (op == 0) {
    load(x, a, N, H * W * C); load(w, b, F, K); load(y, out, N, My * Mx * F);
    x.SetShape4D(N, W, H, C); w.SetShape4D(F, Kx, Ky, C); y.SetShape4D(N, Mx, My, F);
    Matrix::ConvUp(x, w, y, d, scale_targets);
    store(y, out);
  } else if (op == 1) {
    load(y, a, N, My * Mx * F); load(w, b, F, K); load(x, out, N, H * W * C);
    x.SetShape4D(N, W, H, C); w.SetShape4D(F, Kx, Ky, C); y.SetShape4D(N, Mx, My, F);
    Matrix::ConvDown(y, w, x, d, scale_targets);
    store(x, out);
  } else {
    load(x, a, N, H * W * C); load(y, b, N, My * Mx * F); load(w, out, F, K);
    x.SetShape4D(N, W, H, C); w.SetShape4D(F, Kx, Ky, C); y.SetShape4D(N, Mx, My, F);
    Matrix::ConvOutp(x, y, w, d, My, Mx, scale_targets, scale_outputs);
    store(w, out);
  }
}

// op 0: ConvMaxPool; 1: ConvAvgPool; 2: ConvMaxPoolUndo(x, dy, y -> out); 3: ConvAvgPoolUndo(dy -> out (input-shaped)).
void seam_pool(int op, const float* xin, const float* dyin, const float* yin, float* out, int N, int C, int H, int W, int K, int s,
               int pad, int My, int Mx, float scale_targets) {
  const ConvDesc d = desc(C, C, K, K, s, s, pad, pad);
  Matrix x, y, dy, dx;
  if (op == 0 || op == 1) {
    load(x, xin, N, H * W * C); load(y, out, N, My * Mx * C);
    x.SetShape4D(N, W, H, C); y.SetShape4D(N, Mx, My, C);
    if (op == 0) Matrix::ConvMaxPool(x, y, d); else Matrix::ConvAvgPool(x, y, d);
    store(y, out);
  } else if (op == 2) {
    load(x, xin, N, H * W * C); load(dy, dyin, N, My * Mx * C); load(y, yin, N, My * Mx * C); load(dx, out, N, H * W * C);
    x.SetShape4D(N, W, H, C); dx.SetShape4D(N, W, H, C); y.SetShape4D(N, Mx, My, C); dy.SetShape4D(N, Mx, My, C);
    Matrix::ConvMaxPoolUndo(x, dy, y, dx, d, scale_targets);
    store(dx, out);
  } else {
    load(dy, dyin, N, My * Mx * C); load(dx, out, N, H * W * C);
    dx.SetShape4D(N, W, H, C); dy.SetShape4D(N, Mx, My, C);
    Matrix::ConvAvgPoolUndo(dy, dx, d, scale_targets);   // (pooled gradient, input-sized target): avgpool_edge.cc:61-63
    store(dx, out);
  }
}

void seam_rnorm(int undo, const float* xin, const float* dyin, float* out, int N, int C, int P, int sizeF, float add_scale,
                float pow_scale, int blocked) {
  Matrix x, y, dy;
  load(x, xin, N, P * C); load(y, out, N, P * C);
  if (!undo) {
    Matrix::ConvResponseNormCrossMap(x, y, C, sizeF, add_scale, pow_scale, blocked != 0);
  } else {
    load(dy, dyin, N, P * C);
    Matrix acts;   // unused by the GEMM build (matrix.cc:967-969)
    Matrix::ConvResponseNormCrossMapUndo(dy, x, acts, y, C, sizeF, add_scale, pow_scale, blocked != 0);
  }
  store(y, out);
}

// c = alpha*c + beta*op(a)*op(b)  (Matrix::Dot, matrix.cc:678-689)
void seam_dot(const float* a, int ar, int ac, int ta, const float* b, int br, int bc, int tb, float* c, int cr, int cc, float alpha,
              float beta) {
  Matrix A, B, Cm;
  load(A, a, ar, ac); load(B, b, br, bc); load(Cm, c, cr, cc);
  Matrix::Dot(A, B, Cm, alpha, beta, ta != 0, tb != 0);
  store(Cm, c);
}

// FC / output-layer helpers: op 0 AddRowVec(m += v), 1 SumRows(v = alpha*v + beta*colsum(m)) (out = v), 2 LowerBound(m, alpha),
// 3 ApplyDerivativeOfReLU(m=deriv, v=state)
void seam_misc(int op, float* m, int rows, int cols, float* v, float alpha, float beta) {
  Matrix M, V;
  load(M, m, rows, cols);
  if (op == 0) { load(V, v, 1, cols); M.AddRowVec(V); store(M, m); }
  else if (op == 1) { load(V, v, 1, cols); M.SumRows(V, alpha, beta); store(V, v); }
  else if (op == 2) { M.LowerBound(alpha); store(M, m); }
  else { load(V, v, rows, cols); M.ApplyDerivativeOfReLU(V); store(M, m); }
}

// Output layer + optimizer pieces exactly as SoftmaxLayer / CrossEntropyMultinomial / SGDOptimizer call them (layer.cc:570,
// loss_functions.cc:81-120, optimizer.cc:174-200).  probs (N, classes) holds logits on entry; labels (N, 1).
// out_deriv (N, classes), out_correct / out_ce (N, 1).
void seam_softmax(float* probs, const float* labels, float* out_deriv, float* out_correct, float* out_ce, int N, int classes) {
  Matrix state, gt, deriv, correct, ce;
  load(state, probs, N, classes); load(gt, labels, N, 1); load(deriv, out_deriv, N, classes); load(correct, out_correct, N, 1);
  load(ce, out_ce, N, 1);
  state.ApplySoftmax();
  Matrix::SoftmaxCEDeriv(state, gt, deriv);
  Matrix::SoftmaxCorrect(state, gt, correct);
  Matrix::SoftmaxCE(state, gt, ce);
  store(state, probs); store(deriv, out_deriv); store(correct, out_correct); store(ce, out_ce);
}

// One SGD step on a (rows, cols) tensor with the reference's Matrix-call sequence (optimizer.cc:174-200), then the row-norm limit.
void seam_sgd(float* grad, float* param, float* hist, int rows, int cols, float l2, float clip, float eps, float mom, float norm_limit) {
  Matrix g, w, h;
  load(g, grad, rows, cols); load(w, param, rows, cols); load(h, hist, rows, cols);
  if (l2 > 0) g.Add(w, l2);
  if (clip > 0) g.UpperBoundMod(clip);
  g.Mult(eps);
  h.Mult(mom);
  h.Add(g);
  w.Add(h, -1);
  if (norm_limit > 0) w.NormLimitByAxis(1, norm_limit, false);
  store(g, grad); store(w, param); store(h, hist);
}

}  // extern "C"
