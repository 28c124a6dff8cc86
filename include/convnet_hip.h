/* convnet_hip.h — C ABI of libconvnet_hip.so: the MI355X (gfx950) implementation of
 * TorontoDeepLearning/convnet's data-parallel training hot path.
 *
 * This is the reference's "seam #2" (SURVEY.md §8b): the extern "C" cudamat interface that
 * src/matrix.cc and the cudamat ctypes modules bind.  Every entry point below keeps the reference's symbol name,
 * argument order, argument meaning and error convention, so a build of the reference that links
 * this library instead of libcudamat.so + libcudamat_conv_gemm.so needs no source change in
 * the reference's edge / layer / optimizer.cc / convnet.cc.  Each declaration
 * cites the reference interface it replaces.
 *
 * Plain C: pointers, ints, floats.  No torch / HIP types cross this boundary (hipStream_t is passed
 * as void*).  All matrices are column-major fp32; activations are (num_images, X*Y*C) with the
 * image index fastest ("CHWN"), filters (F, Kx*Ky*C)            — cudamat_conv_gemm.cuh:5-10.
 * ConvDesc.padding_* hold the NEGATED pbtxt padding               — src/edge.cc:97-99.
 *
 * Error convention (cudamat.cuh:3-11): functions returning int give 0 or a negative ERROR_* code;
 * the conv/pool/norm functions return void and abort the process on inconsistent shapes, as the
 * reference does (cudamat_conv_gemm.cu:35-42,586-610).
 */
#ifndef CONVNET_HIP_H_
#define CONVNET_HIP_H_

#include <stdbool.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ERROR_INCOMPATIBLE_DIMENSIONS -1
#define CUBLAS_ERROR -2   /* kept for numbering; never returned (no BLAS library is used) */
#define CUDA_ERROR -3     /* any HIP runtime failure */
#define VIEW_ERROR -4
#define ERROR_TRANSPOSED -5
#define ERROR_GENERIC -6
#define ERROR_TRANSPOSEDNESS -7
#define ERROR_NOT_ON_DEVICE -8
#define ERROR_UNSUPPORTED -9

/* Bit-identical to `struct cudamat` (cudamat/cudamat.cuh:28-37; ctypes mirror cudamat.py:127-135).
 * `tex_obj` was a cudaTextureObject_t (64-bit); it is kept as an opaque 64-bit slot. */
typedef struct cudamat {
  float* data_host;
  float* data_device;
  int on_device;
  int on_host;
  int size[2];      /* size[0] = rows = leading dimension */
  int is_trans;     /* 0 or 1; honoured by dot() only */
  int owns_data;
  unsigned long long tex_obj;
} cudamat;
typedef cudamat hipmat;

/* cudamat/cudamat.cuh:86-107 (== eigenmat/common.h:4-27) */
typedef struct Shape4D {
  int shape[4];     /* {num_images, size_x, size_y, channels} — src/layer.cc:257-258 */
} Shape4D;

typedef struct ConvDesc {
  int num_input_channels;
  int num_output_channels;
  int kernel_size_y;
  int kernel_size_x;
  int kernel_size_t;
  int stride_y;
  int stride_x;
  int stride_t;
  int padding_y;
  int padding_x;
  int padding_t;
  int input_channel_begin;
  int input_channel_end;
  int output_channel_begin;
  int output_channel_end;
  int num_groups;
} ConvDesc;

/* ---- library state (new; the reference used the legacy default stream + per-call cudaMalloc) ---
 * All work is enqueued on ONE current stream per host thread-less global (the reference is not
 * thread-safe either).  Scratch (split-K partials, dgrad filter images) comes from a
 * library-owned arena that only grows; nothing in the ABI passes a workspace.
 * ONE DEVICE PER PROCESS (the reference's model: one process per GPU, src/convnet_cpu.cc / src/convnet.cc:407-450 over MPI ranks).  The
 * zero page, the scratch arenas, the launch settings cached per kernel (hipFuncSetAttribute, occupancy, CU count) and the wide kernels'
 * slot counts are process-wide and belong to the device that was current at the first call; cuda_set_device to ANOTHER device after
 * work has been issued is not supported (the tables that ARE keyed by device — the generic-k table, the exchange's stream and events —
 * do not change that). */
int convnet_hip_init(int device_id);                 /* cuda_set_device + cublas_init (cudamat.cuh:93-96) */
void convnet_hip_shutdown(void);                     /* cublas_shutdown */
void convnet_hip_set_stream(void* hip_stream);       /* hipStream_t; NULL = default stream */
void* convnet_hip_get_stream(void);
int convnet_hip_reserve_workspace(size_t bytes);     /* optional pre-size (avoids growth mid-run) */
const char* convnet_hip_version(void);
/* How the GEMM-shaped kernels (conv fprop / dgrad / wgrad, dot) form their fp32 products.  Operands, accumulation and results
 * are fp32 either way.
 *   0 (the library's default): v_mfma_f32_32x32x2_f32 — exact fp32 products, IEEE behaviour for inf / NaN / huge operands, what a
 *     host that links this library in place of cudamat + cublasSgemm (cudamat.cu:2130-2152) gets unless it asks otherwise.
 *   1 (what bench.py, the Python trainer and the tests select explicitly — 1.45x the training throughput): on the bf16 matrix pipe
 *     from EXACT three-way splits x = h + m + l (each term a bf16), six of the nine cross
 *     products per operand pair (hh, hm, mh, hl, lh, mm; the dropped ml, lm, ll are <= 2^-23 of a product), fp32 accumulate in
 *     v_mfma_f32_32x32x16_bf16.  Error against double, measured on the product kernels at the exact conv2 / conv4 / fc6 shapes
 *     (tests/test_split_arithmetic_gpu.py, profiles/r03_split_arithmetic.txt; unit 2^-24 of sum|ab|): N(0,1) data 4.2-4.7 vs
 *     4.4-6.1 for path 0 — accumulation rounding dominates both; terms 2^40 apart inside one dot product 14-19 vs 11-14; terms that
 *     cancel in pairs to 2^-12 (only exact products survive) 0.11 vs 0.05.
 *     Where path 1 is NOT plain fp32 arithmetic (a split product cannot be: inf * w = inf*w_h + inf*w_m is inf - inf or inf * 0
 *     whenever the weight's second term has the other sign or is zero):
 *       - an ACTIVATION / DERIVATIVE operand with |x| > 3.396e38 (0x7F7F7FFF; the top 0.2 % of the fp32 range, and +-inf) makes
 *         the outputs it touches NaN (its first split term is a bf16 inf, the residual inf - inf) where path 0 gives +-inf or a
 *         huge finite number.  NaN operands behave as on path 0.
 *       - a conv FILTER operand (pre-split outside the loops, filter_planes_rt_kernel) above that range (or +-inf) saturates to
 *         +-3.3895e38 (bf16 max); an FC weight or a weight-gradient operand (split inside the loops) gives NaN like an activation.
 *       - operands below ~2^-110 in magnitude: their second / third split terms are denormals, which the matrix pipe flushes;
 *         the product then carries 8-16 instead of 24 significant bits (measured <= 2^-20 of sum|ab| at |x| ~ 2^-116).
 * Initial value: environment CONVNET_GG_SPLIT (0/1), else 0.  May be changed between calls at any time. */
void convnet_hip_set_matrix_path(int path);
int convnet_hip_get_matrix_path(void);
/* Which gather-GEMM kernel runs conv fprop / dgrad of the 3x3 / 5x5 layers on matrix path 1 (N % 64 == 0, 16-channel blocks; same
 * arithmetic, same results to rounding — a schedule choice, for A/B runs and tests):
 *   0: ggp_kernel — one output pixel x 256 images per block, every tap a fresh fetch of the source;
 *   1: gpp_kernel, raw  — 4 neighbouring pixels x 64 images per block; the source pixels of a whole tap row staged once, as fp32;
 *   2: gpp_kernel, planes — the same tile reading the source from bf16 planes written by one extra pass (act_planes_kernel);
 *   3: gpw_kernel — 8 neighbouring pixels x 64 images x 128 rows per block, raw fp32 source, the block's four waves stage for
 *      themselves: 3x3 stride-1 gathers with output rows >= 8 pixels whose blocks fill their rounds on the chip (its launch policy,
 *      patch_gemm.hip: wide_plan); every other shape runs as in mode 0.  The default since round 5 (first run on the MI355X there:
 *      parity green, conv3 fprop / conv4 / conv5 7-12 % faster than mode 0).
 *   4: as 3 without the launch policy — gpw_kernel wherever the shape allows (the parity tests' small geometries, A/B runs).
 * Initial value: environment CONVNET_GG_PATCH, else 3. */
void convnet_hip_set_patch_mode(int mode);
int convnet_hip_get_patch_mode(void);
/* Which kernel runs the weight gradients (conv wgrad, FC wgrad) on matrix path 1 — a schedule choice like the one above:
 *   0: wg_kernel — 128 x 128 tile, four waves of 64 x 64, two blocks per CU;
 *   1: wgw_kernel — 256 x 256 (or 256 x 192) tile, four waves of 128 x 128, one block per CU: conv weight gradients with N % 32 == 0,
 *      K >= 256, F >= 192 and >= 64 chunks of reduction; other shapes (conv1, the FC layers) stay on wg_kernel.  The default since
 *      round 5 (first run on the MI355X there: parity green, conv2-5 weight gradients 24-29 % faster).
 * Initial value: environment CONVNET_WG_TILE, else 1. */
void convnet_hip_set_wgrad_tile(int mode);
int convnet_hip_get_wgrad_tile(void);
/* Deferred epilogues, for hosts that issue the reference's UNFUSED call sequence (the reference's own src/*.cc): with the switch on,
 * convUp / convUpGemm, convDown / convDownGemm, MaxPoolUndo / MaxPoolUndoGemm and ResponseNormCrossMap(Gemm) are parked — one call
 * deep — instead of launched, and the element-wise pass the reference issues right behind them on the same matrix is absorbed into the
 * parked call's fused epilogue: add_row_vec of the shared bias and lower_bound_scalar(.., 0, ..) behind a convolution
 * (src/conv_edge.cc:145-148, src/layer.cc:549-551), lower_bound_scalar behind a response normalisation, apply_rectified_linear_deriv
 * behind convDown or MaxPoolUndo (src/layer.cc:556-558).  ANY other library call launches the parked one first, so results are those
 * of the eager sequence (bit for bit on finite data: the fused epilogues add, clamp and mask in the same order; a masked-out
 * derivative is +0 where the eager pass gives 0 * d).  Initial value: environment CONVNET_DEFER_EPILOGUES, else 0. */
void convnet_hip_set_deferred_epilogues(int on);
int convnet_hip_get_deferred_epilogues(void);
long convnet_hip_deferred_absorbed(void);              /* element-wise calls absorbed into a parked call since the library was loaded */
const char* get_last_cuda_error(void);               /* cudamat.cuh:109 */
int cuda_set_device(int deviceId);                   /* cudamat.cuh:116 */
void cuda_sync_threads(void);                        /* cudamat.cuh:123 — synchronises the current stream */
/* Event trio the reference's Matrix::SetReady / WaitTillReady use for cross-stream ordering (cudamat.cuh:110-112,
 * cudamat.cu:70-91).  `t` points at a hipEvent_t (an opaque pointer, spelled void* here so this header needs no HIP
 * include).  record: on the library's current stream; synchronize: the CURRENT STREAM waits (hipStreamWaitEvent), the
 * host does not block — exactly the reference's behaviour.  Return 0 on success, non-zero on failure. */
int cuda_create_event(void** t);
int cuda_record_event(void** t);
int cuda_synchronize_event(void** t);
int cublas_init(void);                               /* cudamat.cuh:93 — no BLAS handle here: returns 0 */
int cublas_shutdown(void);                           /* cudamat.cuh:94 — frees the scratch arenas */
int destroy_tex(cudamat* mat);                       /* cudamat.cuh:127 — called by the reference's Matrix destructor; no textures here */

/* ---- memory / views (cudamat.cuh:124-153) -------------------------------------------------------- */
int allocate_device_memory(cudamat* mat);
int free_device_memory(cudamat* mat);
int copy_to_host(cudamat* mat);
int copy_to_device(cudamat* mat);
int copy_to_host_slice(cudamat* mat, size_t start, size_t end);     /* column range */
int copy_to_device_slice(cudamat* mat, size_t start, size_t end);
int copy_on_device(cudamat* mat1, cudamat* mat2);                   /* mat2 = mat1 */
int copy_transpose(cudamat* source, cudamat* target);
int reshape(cudamat* mat, int m, int n);                            /* -1 allowed for one dim */
int get_slice(cudamat* source, cudamat* target, unsigned int first_col, unsigned int last_col);
void init_from_array(cudamat* mat, float* data, int m, int n);
int init_empty(cudamat* mat, int m, int n);
int write_at(cudamat* mat, int row, int col, float val);
float read_from(cudamat* mat, int row, int col, int* err_code);

/* ---- GPU-side input staging: the DataHandler step right before the hot path (datahandler.cc:146-198,496-532) --------
 * The dataset chunk lives on the GPU as (dims, cases): one case per COLUMN, [colour][row][col] contiguous. */
/* random crop + horizontal flip + transpose into the (cases, colours*ph*pw) CHWN batch (cudamat.cuh:265, cudamat.cu:2699);
 * width_offset / height_offset / flip hold one float per case (flip > 0.5 mirrors the source column). */
int extract_patches(cudamat* images, cudamat* patches, cudamat* width_offset, cudamat* height_offset, cudamat* flip,
                    int img_width, int img_height, int patch_width, int patch_height);
/* in place: for every column pair (2j, 2j+1) swap columns idx[2j] and idx[2j+1] (cudamat.cu:2655, kShuffleColumns) */
int shuffleColumns(cudamat* source, cudamat* rand_perm_indices);
int add_col_vec(cudamat* mat, cudamat* vec, cudamat* target);                 /* cudamat.cuh:165 */
int add_col_mult(cudamat* mat, cudamat* vec, cudamat* target, float mult);    /* mean subtraction: mult = -1 */
int div_by_col_vec(cudamat* mat, cudamat* vec, cudamat* target);              /* cudamat.cuh:176: divide by std */
int mult_by_row_vec(cudamat* mat, cudamat* vec, cudamat* target);
int div_by_row_vec(cudamat* mat, cudamat* vec, cudamat* target);
int add_to_each_pixel(cudamat* mat1, cudamat* mat2, cudamat* target, float mult);   /* PCA colour noise */
int normalize_by_axis(cudamat* mat, cudamat* target, int axis);               /* axis 0: subtract each column's mean */

/* ---- convolution: cudamat/cudamat_conv_gemm.cuh:36-49 ----------------------------------------------
 * targets = scaleTargets*targets + conv(...).  scaleTargets is the reference's 0/1 accumulate flag but
 * any value is honoured.  Implemented as implicit-GEMM on fp32 MFMA (no im2col buffer).
 * Restrictions, the reference's own: num_groups == 1 (cudamat_conv_gemm.cu:599-600 asserts the same and the host
 * never passes anything else, src/edge.cc:104), whole channel ranges (input/output_channel_begin/end = 0/0 or
 * 0/channels), kernel_size_t <= 1.  These entries are void in the reference's ABI, so a violated restriction or a
 * shape mismatch cannot be returned as a code: it prints "check failed: <condition>" with file:line to stderr and
 * aborts, as the reference's assert() does.  The int-returning entries return the reference's error codes instead. */
void convUpGemm(cudamat* images, cudamat* filters, cudamat* targets,
                Shape4D* images_shape, Shape4D* filters_shape,
                Shape4D* targets_shape, ConvDesc conv_desc,
                float scaleTargets);
void convDownGemm(cudamat* derivs, cudamat* filters, cudamat* targets,
                  Shape4D* derivs_shape, Shape4D* filters_shape,
                  Shape4D* targets_shape, ConvDesc conv_desc,
                  float scaleTargets);
void convOutpGemm(cudamat* images, cudamat* derivs, cudamat* targets,
                  Shape4D* images_shape, Shape4D* derivs_shape,
                  Shape4D* targets_shape, ConvDesc conv_desc,
                  float scaleTargets, float scaleOutput);
/* convnet2-style names of the same operations (cudamat/cudamat_conv.cuh:10-29); partialSum* are
 * accepted and ignored (the split-K factor is chosen internally and reduced deterministically). */
void convUp(cudamat* images, cudamat* filters, cudamat* targets,
            Shape4D* images_shape, Shape4D* filters_shape, Shape4D* targets_shape,
            ConvDesc conv_desc, float scaleTargets);
void convDown(cudamat* derivs, cudamat* filters, cudamat* targets,
              Shape4D* derivs_shape, Shape4D* filters_shape, Shape4D* targets_shape,
              ConvDesc conv_desc, float scaleTargets);
void convOutp(cudamat* images, cudamat* derivs, cudamat* targets,
              Shape4D* images_shape, Shape4D* derivs_shape, Shape4D* targets_shape,
              ConvDesc conv_desc, int partialSumY, int partialSumX, float scaleTargets,
              float scaleOutput);

/* ---- pooling: cudamat_conv_gemm.cuh:72-92 (and cudamat_conv.cuh:58-70) ------------------------------ */
void MaxPoolGemm(cudamat* images, cudamat* targets, Shape4D* images_shape,
                 Shape4D* targets_shape, ConvDesc conv_desc, float scaleTargets,
                 float scaleOutput);
void MaxPoolUndoGemm(cudamat* images, cudamat* maxGrads, cudamat* maxActs,
                     cudamat* targets, Shape4D* images_shape,
                     Shape4D* maxGrads_shape, ConvDesc conv_desc,
                     float scaleTargets);
void AvgPoolGemm(cudamat* images, cudamat* targets, Shape4D* images_shape,
                 Shape4D* targets_shape, ConvDesc conv_desc, float scaleTargets,
                 float scaleOutput);
void AvgPoolUndoGemm(cudamat* avgGrads, cudamat* targets,
                     Shape4D* avgGrads_shape, Shape4D* targets_shape,
                     ConvDesc conv_desc, float scaleTargets);
void MaxPool(cudamat* images, cudamat* targets, Shape4D* images_shape,
             Shape4D* targets_shape, ConvDesc conv_desc);
void AvgPool(cudamat* images, cudamat* targets, Shape4D* images_shape,
             Shape4D* targets_shape, ConvDesc conv_desc);
void MaxPoolUndo(cudamat* images, cudamat* maxGrads, cudamat* maxActs,
                 cudamat* targets, Shape4D* images_shape, Shape4D* maxGrads_shape,
                 ConvDesc conv_desc, float scaleTargets);
void AvgPoolUndo(cudamat* avgGrads, cudamat* targets, Shape4D* avgGrads_shape,
                 Shape4D* targets_shape, ConvDesc conv_desc, float scaleTargets);

/* ---- cross-map response normalisation: cudamat_conv_gemm.cuh:100-106, cudamat_conv.cuh:35-42 --------- */
void ResponseNormCrossMapGemm(cudamat* images, cudamat* targets, int numFilters, int sizeF,
                              float addScale, float powScale, bool blocked);
void ResponseNormCrossMapUndoGemm(cudamat* outGrads, cudamat* inputs, cudamat* targets,
                                  int numFilters, int sizeF, float addScale, float powScale,
                                  bool blocked);
void ResponseNormCrossMap(cudamat* images, cudamat* targets, int numFilters, int sizeF,
                          float addScale, float powScale, bool blocked);
void ResponseNormCrossMapUndo(cudamat* outGrads, cudamat* inputs, cudamat* acts, cudamat* targets,
                              int numFilters, int sizeF, float addScale, float powScale,
                              bool blocked);

/* ---- dense ops (cudamat.cuh:170-263) ------------------------------------------------------------- */
/* target = beta*target + alpha*op(mat1)*op(mat2), op = transpose iff is_trans (cudamat.cu:2130-2152).
 * fc_edge.cc's three uses (NT fwd, NN dgrad, TN wgrad, alpha on the TN side) run on fp32 MFMA; the remaining cases of the
 * reference's contract (T,T; alpha != 1 with a non-transposed mat1) run on a plain LDS-tiled fp32 kernel. */
int dot(cudamat* mat1, cudamat* mat2, cudamat* target, float beta, float alpha);
float vdot(cudamat* mat1, cudamat* mat2, int* err_code);
int add_row_vec(cudamat* mat, cudamat* vec, cudamat* target);
int add_row_mult(cudamat* mat, cudamat* vec, cudamat* target, float mult);
int sum_by_axis(cudamat* mat, cudamat* target, int axis, float mult, float p);   /* target = p*target + mult*sum */
int sqsum_by_axis(cudamat* mat, cudamat* target, int axis, float mult, float p);
float sum_all(cudamat* mat, int* err_code);
float euclid_norm(cudamat* mat, int* err_code);
int normlimit_by_axis(cudamat* mat, cudamat* target, int axis, float norm, int constraint);
int lower_bound_scalar(cudamat* mat, float val, cudamat* target);
int upper_bound_mod_scalar(cudamat* mat, float val, cudamat* target);
int apply_rectified_linear_deriv(cudamat* mat1, cudamat* mat2, cudamat* target);
int assign_scalar(cudamat* mat, float alpha);
int add_scalar(cudamat* mat, float alpha, cudamat* target);
int mult_by_scalar(cudamat* mat, float alpha, cudamat* target, float scale_targets);
int divide_by_scalar(cudamat* mat, float alpha, cudamat* target);
int add_mult(cudamat* mat1, cudamat* mat2, float alpha);                         /* mat1 += alpha*mat2 */
int add_elementwise(cudamat* mat1, cudamat* mat2, cudamat* target);
int subtract_elementwise(cudamat* mat1, cudamat* mat2, cudamat* target);
int mult_elementwise(cudamat* mat1, cudamat* mat2, cudamat* target, float scale_targets);
int apply_sqrt(cudamat* mat, cudamat* target);

/* ---- output layer (cudamat.cuh:249-262) ------------------------------------------------------------ */
int softmax_row_major(cudamat* mat, cudamat* target);
int softmax_row_major_multi(cudamat* mat, int numslices, cudamat* target);
int apply_softmax_grad_row_major(cudamat* mat, cudamat* labels, cudamat* target);
int get_softmax_correct_row_major(cudamat* mat, cudamat* labels, cudamat* target);
int get_softmax_cross_entropy_row_major(cudamat* mat, cudamat* labels, cudamat* target, float tiny);

/* ---- RNG (cudamat.cuh:117-119,154-164).  The reference's GPU (multiply-with-carry) and CPU
 * (std::default_random_engine) streams already differ from each other; this
 * library uses a counter-based Philox-4x32-10 keyed by (seed, call counter, element index). --------- */
typedef struct rnd_struct {
  unsigned int* dev_mults;          /* unused; layout kept (cudamat.cuh:50-53) */
  unsigned long long* dev_words;    /* points at a host-side {seed, counter} pair owned by the library */
} rnd_struct;
int init_random(rnd_struct* rnd_state, int seed);
int fill_with_rand(rnd_struct* rnd_state, cudamat* mat);
int fill_with_randn(rnd_struct* rnd_state, cudamat* mat);
int sample_bernoulli(rnd_struct* rnd_state, cudamat* mat, cudamat* target);
int dropout(rnd_struct* rnd_state, cudamat* mat, float dropprob, float val, float scale);

/* ---- fused entry points (new; SURVEY.md §7 "offer fused entry points in the ABI") ------------------
 * Same arithmetic as the unfused sequences they replace, fewer passes over HBM.
 *  convUpBiasAct      : convUpGemm + reshape/add_row_vec(shared bias) [+ lower_bound_scalar(0)]
 *                        (src/conv_edge.cc:138-149 + src/layer.cc:549-551)
 *  dotBiasAct         : dot(NT) + add_row_vec [+ ReLU]   (src/fc_edge.cc:51-61)
 *  sgd_momentum_step  : SGDOptimizer::Optimize, non-Nesterov (src/optimizer.cc:174-200) in one pass
 *                        (row-norm constraint excluded: call normlimit_by_axis after).
 *  softmax_ce_grad_correct: softmax + SoftmaxCEDeriv + SoftmaxCorrect accumulation
 *                        (src/layer.cc:570-572, src/loss_functions.cc:81-83,114-120); `correct_accum`
 *                        is a 1x1 device matrix accumulated with atomics-free per-call add, so the
 *                        host can read it every print_after steps instead of every step.
 *  relu_dropout       : lower_bound_scalar(0) then dropout(p, 0, scale) (src/layer.cc:391,549). */
void convUpBiasAct(cudamat* images, cudamat* filters, cudamat* bias, cudamat* targets,
                   Shape4D* images_shape, Shape4D* filters_shape, Shape4D* targets_shape,
                   ConvDesc conv_desc, float scaleTargets, int relu);
int dotBiasAct(cudamat* mat1, cudamat* mat2, cudamat* bias, cudamat* target, float beta, float alpha,
               int relu);
int sgd_momentum_step(cudamat* grad, cudamat* param, cudamat* history, float l2_decay,
                      float gradient_clip, float epsilon, float momentum);
int sgd_momentum_step_normlimit(cudamat* grad, cudamat* param, cudamat* history, float l2_decay,
                                float gradient_clip, float epsilon, float momentum, float norm,
                                int constraint);
/* sgd_momentum_step on `count` tensors in one launch per 16 of them (the arrays hold one entry per tensor): the many small tensors of a
 * step — convolution banks, every bias — otherwise cost a launch each.  Element for element the same update: bit-identical to `count`
 * calls of sgd_momentum_step. */
int sgd_momentum_step_multi(int count, cudamat** grads, cudamat** params, cudamat** histories, const float* l2_decay,
                            const float* gradient_clip, const float* epsilon, const float* momentum);   /* + ApplyConstraints (src/optimizer.cc:75-81), axis=1 */
int softmax_ce_grad_correct(cudamat* logits, cudamat* labels, cudamat* probs, cudamat* deriv,
                            cudamat* correct_accum, float deriv_scale);
int relu_dropout(rnd_struct* rnd_state, cudamat* mat, float dropprob, float scale);
/* Backward fusions: the producing kernel applies the consumer layer's ReLU' (and dropout' scale) in
 * its epilogue instead of a separate read-modify-write pass over the derivative
 * (Layer::ApplyDerivativeofDropout + ReLULayer::ApplyDerivativeOfActivation, src/layer.cc:399-413,556-558):
 *   targets = (state > 0) ? post_scale * (scaleTargets*targets + op(...)) : 0.
 * Valid when the layer has a single outgoing edge (src/convnet.cc:390-404 applies them after ALL edges). */
void convDownMask(cudamat* derivs, cudamat* filters, cudamat* state, cudamat* targets,
                  Shape4D* derivs_shape, Shape4D* filters_shape, Shape4D* targets_shape,
                  ConvDesc conv_desc, float scaleTargets, float post_scale);
int dotMask(cudamat* mat1, cudamat* mat2, cudamat* state, cudamat* target, float beta, float alpha,
            float post_scale);
void MaxPoolUndoRelu(cudamat* images, cudamat* maxGrads, cudamat* maxActs, cudamat* targets,
                     Shape4D* images_shape, Shape4D* maxGrads_shape, ConvDesc conv_desc,
                     float scaleTargets);
/* MaxPoolEdge::ComputeUp / ComputeDown (src/maxpool_edge.cc:27-45) with a window mask instead of a second pass over the layer's input:
 * MaxPoolMask = MaxPool (scaleTargets 0, scaleOutput 1) that ALSO writes, per pooled element, 16 bits into `mask`: bit 3*dy + dx set when
 * input (dy, dx) of its window lies inside the image and equals the maximum (float ==, every tie: what kMaxPoolUndo tests,
 * cudamat_conv_gemm.cu:220-262), bit 9 set when the maximum is > 0.  `mask` is any device matrix of at least numel(targets) / 2 floats,
 * 16-byte aligned, laid out as uint16 in the pooled tensor's own element order.  MaxPoolUndoMask = MaxPoolUndo (relu == 0) or
 * MaxPoolUndoRelu (relu != 0) computed from (maxGrads, mask) alone — neither the layer's input nor its maxima are read — and bit-identical
 * to them as long as `mask` is what MaxPoolMask wrote for the tensors the undo would have been given.  3 x 3 windows, stride 2,
 * padding >= 0, N % 4 == 0 only, and relu != 0 only with scaleTargets == 0 (MaxPoolUndoRelu masks the accumulated target too, by
 * input > 0, which the masks hold only for inputs that are a maximum): anything else returns ERROR_UNSUPPORTED and touches nothing
 * (call MaxPool / MaxPoolUndo / MaxPoolUndoRelu). */
int MaxPoolMask(cudamat* images, cudamat* targets, cudamat* mask, Shape4D* images_shape, Shape4D* targets_shape, ConvDesc conv_desc);
int MaxPoolUndoMask(cudamat* maxGrads, cudamat* mask, cudamat* targets, Shape4D* targets_shape, Shape4D* maxGrads_shape,
                    ConvDesc conv_desc, float scaleTargets, int relu);
/* ResponseNormEdge::ComputeUp + the ReLU of a RECTIFIED_LINEAR destination layer (layer.cc:549) in one pass. */
void ResponseNormCrossMapRelu(cudamat* images, cudamat* targets, int numFilters, int sizeF,
                              float addScale, float powScale, bool blocked);
/* ConvEdge::ComputeOuter (conv_edge.cc:183-221) in one call: dW as convOutpGemm AND the shared-bias gradient
 * bias_grad(1,F) = scaleTargets*bias_grad + scaleOutput * sum over images and output pixels of derivs — the reference's
 * two-step SumRows (:210-221).  The bias row rides as a virtual tap with constant input 1 in the weight-gradient tile
 * when the tile has a spare row (conv1: 147 -> 148 of 160), else the library falls back to a column sum. */
void convOutpBias(cudamat* images, cudamat* derivs, cudamat* targets, cudamat* bias_grad,
                  Shape4D* images_shape, Shape4D* derivs_shape, Shape4D* targets_shape,
                  ConvDesc conv_desc, float scaleTargets, float scaleOutput);

/* ---- introspection for the roofline report: flops of the last MFMA launch family ----------------- */
typedef struct ConvnetHipKernelInfo {
  const char* name;      /* kernel family of the last conv/dot call */
  double flops;          /* algorithmic flops of that call (2*M*N*K) */
  int grid_blocks;
  int split_k;
} ConvnetHipKernelInfo;
void convnet_hip_last_kernel_info(ConvnetHipKernelInfo* out);
/* Per-launch HIP-event timing on the library stream (used by bench.py's roofline leg).  While enabled
 * every conv/FC/pool/norm launch is bracketed by two hipEventRecord calls; the report synchronises,
 * aggregates by (kernel, op) into text lines "kernel|op|launches|total_ms|total_flops|total_bytes|total_executed_flops"
 * (flops = algorithmic work; executed = MFMA work issued, larger for dgrad gathers that run border taps on the zero page). */
void convnet_hip_profile_enable(int on);
size_t convnet_hip_profile_report(char* buf, size_t cap);
/* What the chip sustains on the instruction the default GEMM kernels execute (v_mfma_f32_32x32x16_bf16, six per fp32 product block),
 * with nothing else in the way: one wave per SIMD, sixteen accumulators, register operands — the h / m / l planes of N(0,1) values
 * (random_operands = 1) or zeros (0) — for about `seconds`.  The part clocks to its power budget, so this is the ceiling a kernel of
 * these instructions has on this box, beside the nominal 2.5 PFLOP/s (csrc/probe.hip).  out4 = {executed bf16 TFLOP/s, the same / 6 =
 * algorithmic fp32 TFLOP/s, GHz by the shader-cycle counter over wall time, GHz the MFMA issue rate implies}.  Returns 0 or an error code. */
int convnet_hip_probe_matrix_pipe(int random_operands, double seconds, double* out4);

/* ---- data-parallel gradient exchange (csrc/comm.hip): replaces ConvNet::Accumulate + ConvNet::Broadcast ------------------
 * Reference (src/convnet.cc:407-450, behind USE_MPI): after Bprop the whole flat gradient goes device -> host, rank 0
 * MPI_Recv-sums every rank's copy, divides by num_processes_ (:431), copies it back and MPI_Bcasts it; UpdateWeights (:440-450)
 * then steps every edge.  Here: one process per GPU, RCCL over xGMI, every slice (or bucket of slices) of the flat gradient is
 * all-reduced on a communication stream from the moment it is final, and the compute stream waits — on the device, not the
 * host — right before the optimizer step that consumes it.  librccl is dlopen'ed by convnet_hip_comm_init.
 *
 *   rank 0:   convnet_hip_comm_unique_id(id);  ship the 128 bytes to the other ranks (MPI_Bcast / a file / a socket)
 *   all:      convnet_hip_comm_init(rank, nranks, id);  convnet_hip_comm_broadcast(&parameters, 0);      // convnet.cc:309
 *   Bprop(output, input, edge), after edge.ComputeOuter:  convnet_hip_comm_allreduce_avg(&grad_parameters, off, n, slot);
 *   UpdateWeights, before edge->UpdateWeights():          convnet_hip_comm_wait(slot);
 * `slot` (0..255) names the done-event of one post; it stays waitable until the same slot is posted again.
 * grad[off .. off+n) becomes sum over ranks / nranks (true division, as :431).  Returns 0 or a cudamat error code
 * (get_last_cuda_error() has the RCCL text). */
#define CONVNET_HIP_COMM_ID_BYTES 128
int convnet_hip_comm_unique_id(char* id_out);
int convnet_hip_comm_init(int rank, int nranks, const char* id_in);
int convnet_hip_comm_rank(void);
int convnet_hip_comm_size(void);
int convnet_hip_comm_max_slots(void);   /* slots [0, max_slots) exist (256): plan the posts of one step against it up front */
int convnet_hip_comm_broadcast(cudamat* mat, int root);
int convnet_hip_comm_allreduce_avg(cudamat* flat, size_t offset, size_t count, int slot);
int convnet_hip_comm_wait(int slot);
int convnet_hip_comm_sync(void);      /* host-blocking drain of the communication stream */
int convnet_hip_comm_destroy(void);

#ifdef __cplusplus
}
#endif
#endif  /* CONVNET_HIP_H_ */
