/* TEST INFRASTRUCTURE — NOT PRODUCT CODE.  Plain-C CPU restatement of the reference's
 * data-parallel training hot path (TorontoDeepLearning/convnet), used ONLY as the parity checker
 * by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product path
 * (convnet_amd/csrc/ *.hip behind include/convnet_hip.h) never links or calls this file.
 *
 * Parity status: PINNED.  Every function here is checked against the reference's own compiled CPU
 * path (oracle/_ref/libconvnet_ref.so = eigenmat/eigenmat.cc + eigenmat/cpumat_conv.cc +
 * src/CPUMatrix.cc built unmodified by oracle/Makefile) in tests/test_oracle_vs_reference.py,
 * and against committed golden vectors (tests/golden/, produced from that same reference build by
 * tests/golden/make_golden.py).  The reference repo ships no stored golden vectors of its own
 *  — its tests compare GPU vs CPU on random inputs with tolerance 1e-4
 * (py/test_conv.py:382-392).
 *
 * Conventions (identical to the reference):
 *  - every matrix is column-major float32; an activation is (N, X*Y*C) with image index fastest:
 *    element (n,c,y,x) at n + N*(x + W*(y + H*c))          (cudamat/cudamat_conv_gemm.cuh:5-10)
 *  - filters are (F, Kx*Ky*C): element (f,c,ky,kx) at f + F*(kx + Kx*(ky + Ky*c))
 *  - pady/padx are the NEGATED paddings the reference stores in ConvDesc (src/edge.cc:97-99):
 *    first input row of output row oy is oy*sy + pady.
 *  - accumulation order follows the reference so results agree to fp32 round-off.
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define IDX4(n, x, y, c, N, W, H) ((size_t)(n) + (size_t)(N) * ((size_t)(x) + (size_t)(W) * ((size_t)(y) + (size_t)(H) * (size_t)(c))))

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* ---- convolution ------------------------------------------------------------------------- */

/* targets = scaleTargets*targets + scaleOutput*conv(images, filters).
 * Follows eigenmat/cpumat_conv.cc:109-220 (convUp: per-module expand + sgemm): for each output
 * location the patch is walked in (c, ky, kx) order and summed sequentially in fp32, exactly the
 * order of the reference's naive sgemm inner loop (eigenmat/eigenmat.cc:2284-2298). */
void oracle_conv_up(const float* images, const float* filters, float* targets, int N, int C, int H,
                    int W, int F, int Ky, int Kx, int sy, int sx, int pady, int padx, int My, int Mx,
                    float scaleTargets, float scaleOutput) {
#pragma omp parallel
  {
    float* acc = (float*)malloc(sizeof(float) * (size_t)N);
#pragma omp for collapse(2) schedule(static)
    for (int m = 0; m < My * Mx; ++m) {
      for (int f = 0; f < F; ++f) {
        const int oy = m / Mx, ox = m % Mx;
        for (int n = 0; n < N; ++n) acc[n] = 0.f;
        for (int c = 0; c < C; ++c)
          for (int ky = 0; ky < Ky; ++ky) {
            const int iy = oy * sy + pady + ky;
            for (int kx = 0; kx < Kx; ++kx) {
              const int ix = ox * sx + padx + kx;
              /* out-of-image taps contribute a literal 0*w term in the reference (expand() writes
               * zeros, cpumat_conv.cc:46-52); adding +0 never changes an fp32 sum, so skip. */
              if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
              const float w = filters[(size_t)f + (size_t)F * (kx + Kx * (ky + Ky * c))];
              const float* src = images + IDX4(0, ix, iy, c, N, W, H);
              for (int n = 0; n < N; ++n) acc[n] += src[n] * w;
            }
          }
        float* dst = targets + IDX4(0, ox, oy, f, N, Mx, My);
        for (int n = 0; n < N; ++n) dst[n] = scaleTargets * dst[n] + scaleOutput * acc[n];
      }
    }
    free(acc);
  }
}

/* targets(N,H,W,C) = scaleTargets*targets + sum over modules of col2im(derivs x filters).
 * Follows eigenmat/cpumat_conv.cc:222-338 (convDown): the reference first scales the whole target,
 * then for module 0,1,2... computes the (N x KxKyC) product (sequential sum over f) and adds it
 * into the image (contract(), :63-107).  The per-pixel accumulation order is therefore "modules
 * ascending, each contribution already summed over f" — reproduced here per input pixel. */
void oracle_conv_down(const float* derivs, const float* filters, float* targets, int N, int C, int H,
                      int W, int F, int Ky, int Kx, int sy, int sx, int pady, int padx, int My,
                      int Mx, float scaleTargets, float scaleOutput) {
#pragma omp parallel
  {
    float* part = (float*)malloc(sizeof(float) * (size_t)N);
#pragma omp for collapse(2) schedule(static)
    for (int c = 0; c < C; ++c) {
      for (int p = 0; p < H * W; ++p) {
        const int iy = p / W, ix = p % W;
        float* dst = targets + IDX4(0, ix, iy, c, N, W, H);
        for (int n = 0; n < N; ++n) dst[n] *= scaleTargets;
        for (int oy = 0; oy < My; ++oy) {
          const int ky = iy - (oy * sy + pady);
          if (ky < 0 || ky >= Ky) continue;
          for (int ox = 0; ox < Mx; ++ox) {
            const int kx = ix - (ox * sx + padx);
            if (kx < 0 || kx >= Kx) continue;
            for (int n = 0; n < N; ++n) part[n] = 0.f;
            for (int f = 0; f < F; ++f) {
              const float w = filters[(size_t)f + (size_t)F * (kx + Kx * (ky + Ky * c))];
              const float* src = derivs + IDX4(0, ox, oy, f, N, Mx, My);
              for (int n = 0; n < N; ++n) part[n] += src[n] * w;
            }
            for (int n = 0; n < N; ++n) dst[n] += scaleOutput * part[n];
          }
        }
      }
    }
    free(part);
  }
}

/* targets(F, KxKyC) = scaleTargets*targets + scaleOutput * sum_{modules,n} derivs (x) patch(images).
 * Follows eigenmat/cpumat_conv.cc:340-460 (convOutp): scale target once, then for each module m a
 * beta=1 sgemm adds scaleOutput * (sum over n, sequential) — so the order is modules ascending,
 * inner sum over images. */
void oracle_conv_outp(const float* images, const float* derivs, float* targets, int N, int C, int H,
                      int W, int F, int Ky, int Kx, int sy, int sx, int pady, int padx, int My,
                      int Mx, float scaleTargets, float scaleOutput) {
  const int K = C * Ky * Kx;
#pragma omp parallel for collapse(2) schedule(static)
  for (int k = 0; k < K; ++k) {
    for (int f = 0; f < F; ++f) {
      const int kx = k % Kx, ky = (k / Kx) % Ky, c = k / (Kx * Ky);
      float t = targets[(size_t)f + (size_t)F * k] * scaleTargets;
      for (int oy = 0; oy < My; ++oy) {
        const int iy = oy * sy + pady + ky;
        for (int ox = 0; ox < Mx; ++ox) {
          const int ix = ox * sx + padx + kx;
          float res = 0.f;
          if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
            const float* a = derivs + IDX4(0, ox, oy, f, N, Mx, My);
            const float* b = images + IDX4(0, ix, iy, c, N, W, H);
            for (int n = 0; n < N; ++n) res += a[n] * b[n];
          }
          t = 1.0f * t + scaleOutput * res;
        }
      }
      targets[(size_t)f + (size_t)F * k] = t;
    }
  }
}

/* ---- pooling ----------------------------------------------------------------------------- */

/* Max over the window clipped to the image; follows src/CPUMatrix.cc:574-640 for the value and
 * cudamat/cudamat_conv_gemm.cu:153-201 (kPool) for the scaleTargets/scaleOutput epilogue (the CPU
 * class hard-codes 0/1).  An all-padding window yields -FLT_MAX on the CPU path. */
void oracle_max_pool(const float* images, float* targets, int N, int C, int H, int W, int Ky, int Kx,
                     int sy, int sx, int pady, int padx, int My, int Mx, float scaleTargets,
                     float scaleOutput) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int c = 0; c < C; ++c)
    for (int m = 0; m < My * Mx; ++m) {
      const int oy = m / Mx, ox = m % Mx;
      const int y0 = imax(0, oy * sy + pady), y1 = imin(H, oy * sy + pady + Ky);
      const int x0 = imax(0, ox * sx + padx), x1 = imin(W, ox * sx + padx + Kx);
      float* dst = targets + IDX4(0, ox, oy, c, N, Mx, My);
      for (int n = 0; n < N; ++n) {
        float r = -FLT_MAX;
        for (int y = y0; y < y1; ++y)
          for (int x = x0; x < x1; ++x) {
            const float v = images[IDX4(n, x, y, c, N, W, H)];
            if (r < v) r = v;
          }
        dst[n] = scaleTargets * dst[n] + scaleOutput * r;
      }
    }
}

/* Mean over the CLIPPED window (divide by the number of valid taps): src/CPUMatrix.cc:700-762. */
void oracle_avg_pool(const float* images, float* targets, int N, int C, int H, int W, int Ky, int Kx,
                     int sy, int sx, int pady, int padx, int My, int Mx, float scaleTargets,
                     float scaleOutput) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int c = 0; c < C; ++c)
    for (int m = 0; m < My * Mx; ++m) {
      const int oy = m / Mx, ox = m % Mx;
      const int y0 = imax(0, oy * sy + pady), y1 = imin(H, oy * sy + pady + Ky);
      const int x0 = imax(0, ox * sx + padx), x1 = imin(W, ox * sx + padx + Kx);
      float* dst = targets + IDX4(0, ox, oy, c, N, Mx, My);
      for (int n = 0; n < N; ++n) {
        float r = 0.f;
        int cnt = 0;
        for (int y = y0; y < y1; ++y)
          for (int x = x0; x < x1; ++x) {
            r += images[IDX4(n, x, y, c, N, W, H)];
            ++cnt;
          }
        r /= cnt;
        dst[n] = scaleTargets * dst[n] + scaleOutput * r;
      }
    }
}

/* Range of output (pooled) coordinates whose window covers input coordinate i.
 * Same integer formulas as src/CPUMatrix.cc:669-673 (written there with padding_x only). */
static void cover_range(int i, int pad, int k, int s, int M, int* lo, int* hi) {
  *lo = (i - pad < k) ? 0 : (i - pad - k) / s + 1;
  *hi = imin(M, 1 + (i - pad) / s);
}

/* d_in[p] = scaleTargets*d_in[p] + sum_{o covers p, in[p]==out[o]} d_out[o]   (ties all count):
 * src/CPUMatrix.cc:642-698; same tie rule on the GPU, cudamat_conv_gemm.cu:285-298. */
void oracle_max_pool_undo(const float* images, const float* maxGrads, const float* maxActs,
                          float* targets, int N, int C, int H, int W, int Ky, int Kx, int sy, int sx,
                          int pady, int padx, int My, int Mx, float scaleTargets) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int c = 0; c < C; ++c)
    for (int p = 0; p < H * W; ++p) {
      const int iy = p / W, ix = p % W;
      int oy0, oy1, ox0, ox1;
      cover_range(iy, pady, Ky, sy, My, &oy0, &oy1);
      cover_range(ix, padx, Kx, sx, Mx, &ox0, &ox1);
      const int inside = ix < padx + sx * (Mx - 1) + Kx && iy < pady + sy * (My - 1) + Ky;
      for (int n = 0; n < N; ++n) {
        const size_t t = IDX4(n, ix, iy, c, N, W, H);
        float r = 0.f;
        if (inside)
          for (int oy = oy0; oy < oy1; ++oy)
            for (int ox = ox0; ox < ox1; ++ox) {
              const size_t o = IDX4(n, ox, oy, c, N, Mx, My);
              if (images[t] == maxActs[o]) r += maxGrads[o];
            }
        targets[t] = scaleTargets * targets[t] + r;
      }
    }
}

/* d_in[p] = scaleTargets*d_in[p] + sum_{o covers p} d_out[o] / (clipped window size of o):
 * src/CPUMatrix.cc:764-827 (region size in float, reciprocal then multiply). */
void oracle_avg_pool_undo(const float* avgGrads, float* targets, int N, int C, int H, int W, int Ky,
                          int Kx, int sy, int sx, int pady, int padx, int My, int Mx,
                          float scaleTargets) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int c = 0; c < C; ++c)
    for (int p = 0; p < H * W; ++p) {
      const int iy = p / W, ix = p % W;
      int oy0, oy1, ox0, ox1;
      cover_range(iy, pady, Ky, sy, My, &oy0, &oy1);
      cover_range(ix, padx, Kx, sx, Mx, &ox0, &ox1);
      const int inside = ix < padx + sx * (Mx - 1) + Kx && iy < pady + sy * (My - 1) + Ky;
      for (int n = 0; n < N; ++n) {
        float r = 0.f;
        if (inside)
          for (int oy = oy0; oy < oy1; ++oy) {
            const float ry = fminf((float)H, (float)(pady + oy * sy + Ky)) - fmaxf(0.f, (float)(pady + oy * sy));
            for (int ox = ox0; ox < ox1; ++ox) {
              const float rx = fminf((float)W, (float)(padx + ox * sx + Kx)) - fmaxf(0.f, (float)(padx + ox * sx));
              const float inv = 1.0f / (rx * ry);
              r += avgGrads[IDX4(n, ox, oy, c, N, Mx, My)] * inv;
            }
          }
        const size_t t = IDX4(n, ix, iy, c, N, W, H);
        targets[t] = scaleTargets * targets[t] + r;
      }
    }
}

/* ---- cross-map response normalisation --------------------------------------------------------- */

/* out_j = in_j * (1 + addScale * sum_{i in win(j)} in_i^2)^(-powScale), sliding-window update of
 * the sum exactly as eigenmat/cpumat_conv.cc:462-494 (subtract leaving, add entering, fp32). */
void oracle_rnorm(const float* images, float* targets, int num_locs, int C, int sizeF, float addScale,
                  float powScale, int blocked) {
#pragma omp parallel for schedule(static)
  for (int loc = 0; loc < num_locs; ++loc) {
    float sum = 0.f;
    int ps = 0, pe = 0;
    for (int j = 0; j < C; ++j) {
      int start = blocked ? (j / sizeF) * sizeF : -sizeF / 2 + j;
      const int end = imin(C, start + sizeF);
      start = imax(0, start);
      for (int i = ps; i < start; ++i) { const float v = images[(size_t)i * num_locs + loc]; sum -= v * v; }
      for (int i = pe; i < end; ++i) { const float v = images[(size_t)i * num_locs + loc]; sum += v * v; }
      const size_t idx = (size_t)j * num_locs + loc;
      targets[idx] = images[idx] * powf(1 + addScale * sum, -powScale);
      ps = start; pe = end;
    }
  }
}

/* eigenmat/cpumat_conv.cc:496-560: denoms = (1+a*sum)^(-b-1); then
 * d_in_j = d_out_j*denoms_j^(b/(b+1)) - 2ab*in_j*sum_{i in win^-1(j)} d_out_i*in_i*denoms_i. */
void oracle_rnorm_undo(const float* outGrads, const float* inputs, float* targets, int num_locs, int C,
                       int sizeF, float addScale, float powScale, int blocked) {
#pragma omp parallel
  {
    float* den = (float*)malloc(sizeof(float) * (size_t)C);
#pragma omp for schedule(static)
    for (int loc = 0; loc < num_locs; ++loc) {
      float sum = 0.f;
      int ps = 0, pe = 0;
      for (int j = 0; j < C; ++j) {
        int start = blocked ? (j / sizeF) * sizeF : -sizeF / 2 + j;
        const int end = imin(C, start + sizeF);
        start = imax(0, start);
        for (int i = ps; i < start; ++i) { const float v = inputs[(size_t)i * num_locs + loc]; sum -= v * v; }
        for (int i = pe; i < end; ++i) { const float v = inputs[(size_t)i * num_locs + loc]; sum += v * v; }
        den[j] = powf(1 + addScale * sum, -powScale - 1);
        ps = start; pe = end;
      }
      sum = 0.f; ps = 0; pe = 0;
      for (int j = 0; j < C; ++j) {
        int start = blocked ? (j / sizeF) * sizeF : -sizeF + sizeF / 2 + j + 1;
        const int end = imin(C, start + sizeF);
        start = imax(0, start);
        for (int i = ps; i < start; ++i) { const size_t x = (size_t)i * num_locs + loc; sum -= outGrads[x] * inputs[x] * den[i]; }
        for (int i = pe; i < end; ++i) { const size_t x = (size_t)i * num_locs + loc; sum += outGrads[x] * inputs[x] * den[i]; }
        const size_t idx = (size_t)j * num_locs + loc;
        targets[idx] = outGrads[idx] * powf(den[j], powScale / (powScale + 1)) - 2 * addScale * powScale * inputs[idx] * sum;
        ps = start; pe = end;
      }
    }
    free(den);
  }
}

/* ---- dense ops used by fc_edge / layer / loss / optimizer -------------------------------------- */

/* target(m x n) = beta*target + alpha*op(a)*op(b), column-major, op = transpose when *_trans.
 * Semantics of eigenmat dot() (eigenmat/eigenmat.cc:1605-1631); a is (a_rows x a_cols) as stored.
 * Summation over k is sequential fp32 (Eigen blocks differently: agreement is to round-off). */
void oracle_dot(const float* a, int a_rows, int a_cols, int a_trans, const float* b, int b_rows,
                int b_cols, int b_trans, float* target, float beta, float alpha) {
  const int m = a_trans ? a_cols : a_rows, k = a_trans ? a_rows : a_cols;
  const int n = b_trans ? b_rows : b_cols;
  (void)b_cols;
#pragma omp parallel for collapse(2) schedule(static)
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < m; ++i) {
      double s = 0.0; /* double accumulate: this entry is a reference point, not an order replay */
      for (int l = 0; l < k; ++l) {
        const float av = a_trans ? a[(size_t)l + (size_t)a_rows * i] : a[(size_t)i + (size_t)a_rows * l];
        const float bv = b_trans ? b[(size_t)j + (size_t)b_rows * l] : b[(size_t)l + (size_t)b_rows * j];
        s += (double)av * bv;
      }
      const size_t t = (size_t)i + (size_t)m * j;
      target[t] = beta * target[t] + alpha * (float)s;
    }
}

/* mat[:, j] += vec[j]  (eigenmat add_row_vec, eigenmat.cc:459-477) */
void oracle_add_row_vec(float* mat, const float* vec, int rows, int cols) {
  for (int j = 0; j < cols; ++j)
    for (int i = 0; i < rows; ++i) mat[(size_t)i + (size_t)rows * j] += vec[j];
}

/* target = p*target + mult*sum(mat, axis); axis 0 sums each column (eigenmat.cc:887-916). */
void oracle_sum_by_axis(const float* mat, int rows, int cols, float* target, int axis, float mult, float p) {
  if (axis == 0) {
    for (int j = 0; j < cols; ++j) {
      double s = 0;
      for (int i = 0; i < rows; ++i) s += mat[(size_t)i + (size_t)rows * j];
      target[j] = p * target[j] + mult * (float)s;
    }
  } else {
    for (int i = 0; i < rows; ++i) {
      double s = 0;
      for (int j = 0; j < cols; ++j) s += mat[(size_t)i + (size_t)rows * j];
      target[i] = p * target[i] + mult * (float)s;
    }
  }
}

/* ReLU forward = lower_bound_scalar(0) (src/layer.cc:549-551, eigenmat.cc:700-713). */
void oracle_lower_bound(float* mat, size_t len, float val) {
  for (size_t i = 0; i < len; ++i) mat[i] = mat[i] > val ? mat[i] : val;
}

/* deriv *= (state > 0)  (eigenmat.cc:1805-1820). */
void oracle_relu_deriv(float* deriv, const float* state, size_t len) {
  for (size_t i = 0; i < len; ++i) deriv[i] = deriv[i] * (state[i] > 0 ? 1 : 0);
}

/* clip to [-val, val] (eigenmat.cc:656-683). */
void oracle_upper_bound_mod(float* mat, size_t len, float val) {
  for (size_t i = 0; i < len; ++i) {
    const float c = mat[i];
    mat[i] = c > val ? val : (c < -val ? -val : c);
  }
}

/* One case per matrix ROW ("row_major" in the reference's naming), rows x cols column-major:
 * eigenmat.cc:1093-1131 (max, exp(x-max), divide by sum; fp32 throughout, expf like C++ exp(float)). */
void oracle_softmax_row_major(float* mat, int rows, int cols) {
  for (int i = 0; i < rows; ++i) {
    float mx = mat[i], sum = 0.f;
    for (int j = 1; j < cols; ++j) { const float c = mat[(size_t)j * rows + i]; if (mx < c) mx = c; }
    for (int j = 0; j < cols; ++j) { const size_t x = (size_t)j * rows + i; mat[x] = expf(mat[x] - mx); sum += mat[x]; }
    for (int j = 0; j < cols; ++j) mat[(size_t)j * rows + i] /= sum;
  }
}

/* target = mat; target[i, label_i] -= 1 (eigenmat.cc:1175-1195). */
void oracle_softmax_grad_row_major(const float* mat, const float* labels, float* target, int rows, int cols) {
  if (target != mat) memcpy(target, mat, sizeof(float) * (size_t)rows * cols);
  for (int i = 0; i < rows; ++i) target[(size_t)i + (size_t)rows * (int)labels[i]] -= 1.0f;
}

/* first-argmax == label (strict <, eigenmat.cc:1258-1289). */
void oracle_softmax_correct_row_major(const float* mat, const float* labels, float* target, int rows, int cols) {
  for (int i = 0; i < rows; ++i) {
    int am = 0;
    for (int j = 1; j < cols; ++j)
      if (mat[(size_t)am * rows + i] < mat[(size_t)j * rows + i]) am = j;
    target[i] = ((int)labels[i] == am) ? 1.f : 0.f;
  }
}

/* -log(p[label] + tiny) (eigenmat.cc:1211-1232; tiny=1e-10 from src/CPUMatrix.cc SoftmaxCE). */
void oracle_softmax_ce_row_major(const float* mat, const float* labels, float* target, int rows, int cols, float tiny) {
  (void)cols;
  for (int i = 0; i < rows; ++i) target[i] = -logf(mat[(size_t)rows * (int)labels[i] + i] + tiny);
}

/* Rescale each ROW (axis=1: one output unit's incoming weights) to norm when it exceeds it, or
 * always when constraint (eigenmat.cc:918-968). */
void oracle_normlimit_rows(float* mat, int rows, int cols, float norm, int constraint) {
  for (int r = 0; r < rows; ++r) {
    float s = 0.f;
    for (int j = 0; j < cols; ++j) { const float v = mat[(size_t)r + (size_t)rows * j]; s += v * v; }
    s = sqrtf(s);
    s = (constraint == 1 || s > norm) ? (norm / s) : 1;
    for (int j = 0; j < cols; ++j) mat[(size_t)r + (size_t)rows * j] *= s;
  }
}

/* One SGD+momentum step, the op sequence of SGDOptimizer::Optimize (src/optimizer.cc:174-200,
 * non-Nesterov): g += l2*w; clip; g *= eps; h = mom*h; h += g; w -= h; row-norm constraint.
 * Each statement is a separate fp32 pass in the reference, so no fma contraction here. */
void oracle_sgd_step(float* grad, float* param, float* history, int rows, int cols, float l2_decay,
                     float gradient_clip, float epsilon, float momentum, float norm_limit,
                     float norm_constraint) {
  const size_t len = (size_t)rows * cols;
  for (size_t i = 0; i < len; ++i) {
    volatile float g = grad[i];
    if (l2_decay > 0) { volatile float t = param[i] * l2_decay; g = g + t; }
    if (gradient_clip > 0) g = g > gradient_clip ? gradient_clip : (g < -gradient_clip ? -gradient_clip : g);
    g = g * epsilon;
    volatile float h = history[i] * momentum;
    h = h + g;
    volatile float hm = h * -1.0f;
    grad[i] = g;
    history[i] = h;
    param[i] = param[i] + hm;
  }
  if (norm_constraint > 0) oracle_normlimit_rows(param, rows, cols, norm_constraint, 1);
  else if (norm_limit > 0) oracle_normlimit_rows(param, rows, cols, norm_limit, 0);
}

/* Dropout with an EXPLICIT uniform draw per element (the reference's RNG streams differ between
 * its own back-ends, SURVEY.md fact 10, so the draw is an input here): eigenmat.cc:277-293. */
void oracle_dropout(float* mat, const float* uniform, size_t len, float dropprob, float val, float scale) {
  for (size_t i = 0; i < len; ++i) {
    if (dropprob > uniform[i]) mat[i] = val; else mat[i] *= scale;
  }
}

/* ---- input staging (the DataHandler's GPU-side calls) ------------------------------------------------------------
 * images (dims, num_images): one case per column, [colour][row][col] contiguous; patches = CHWN batch.
 * eigenmat.cc:2046-2090 (extract_patches), :1962-1988 (shuffleColumns), :325-344, :519-535, :499-517, :970-997, :346-366. */
void oracle_extract_patches(const float* images, int num_images, int colors, float* patches, const float* wo, const float* ho,
                            const float* flip, int img_w, int img_h, int pw, int ph) {
  for (int n = 0; n < num_images; ++n)
    for (int dc = 0; dc < pw; ++dc) {
      int sc = (int)wo[n] + dc;
      if (flip[n] > 0.5f) sc = img_w - sc - 1;
      for (int dr = 0; dr < ph; ++dr) {
        const int sr = (int)ho[n] + dr;
        for (int c = 0; c < colors; ++c)
          patches[(size_t)n + (size_t)num_images * (dc + (size_t)pw * (dr + (size_t)ph * c))] =
              images[(size_t)sc + (size_t)img_w * (sr + (size_t)img_h * (c + (size_t)colors * n))];
      }
    }
}

void oracle_shuffle_columns(float* mat, int height, int width, const float* perm) {
  for (int c = 0; c + 1 < width; c += 2) {
    float* a = mat + (size_t)height * (int)perm[c];
    float* b = mat + (size_t)height * (int)perm[c + 1];
    for (int i = 0; i < height; ++i) { const float t = a[i]; a[i] = b[i]; b[i] = t; }
  }
}

void oracle_add_col_mult(float* mat, int h, int w, const float* vec, float mult) {
  for (int j = 0; j < w; ++j)
    for (int i = 0; i < h; ++i) { volatile float t = vec[i] * mult; mat[(size_t)i + (size_t)h * j] += t; }
}

void oracle_div_by_col_vec(float* mat, int h, int w, const float* vec) {
  for (int j = 0; j < w; ++j)
    for (int i = 0; i < h; ++i) mat[(size_t)i + (size_t)h * j] /= vec[i];
}

void oracle_mult_by_row_vec(float* mat, int h, int w, const float* vec) {
  for (int j = 0; j < w; ++j)
    for (int i = 0; i < h; ++i) mat[(size_t)i + (size_t)h * j] *= vec[j];
}

void oracle_normalize_columns(float* mat, int h, int w) {
  for (int j = 0; j < w; ++j) {
    float* col = mat + (size_t)h * j;
    float s = 0.f;
    for (int i = 0; i < h; ++i) s += col[i];
    s /= h;
    for (int i = 0; i < h; ++i) col[i] -= s;
  }
}

void oracle_add_to_each_pixel(float* mat1, int height, int width, const float* mat2, int colors, float mult) {
  const size_t num_pix = (size_t)height * width / colors;
  for (size_t i = 0; i < (size_t)height * width; ++i) {
    volatile float t = mult * mat2[i % height + (size_t)height * (i / num_pix)];
    mat1[i] += t;
  }
}

void oracle_copy_transpose(const float* src, int rows, int cols, float* dst) {
  for (int j = 0; j < cols; ++j)
    for (int i = 0; i < rows; ++i) dst[(size_t)j + (size_t)cols * i] = src[(size_t)i + (size_t)rows * j];
}

int oracle_version(void) { return 2; }
