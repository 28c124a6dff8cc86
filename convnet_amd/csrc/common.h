// Internal helpers shared by the HIP translation units of libconvnet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "../../include/convnet_hip.h"

namespace chip {

// One current stream for the whole library (the reference ran everything on stream 0 of the
// current device, cudamat.cu; it is equally non-thread-safe).
hipStream_t stream();
// Grow-only scratch arena.  Returns a device pointer valid until the next call that asks for more
// than the current capacity (growth synchronises the stream first, so in-flight users stay valid).
void* workspace(size_t bytes);
void* workspace_aux(size_t bytes);   // independent second arena (dgrad filter images)
void* workspace_planes(size_t bytes);   // independent third arena (bf16 planes of a gather-GEMM's source tensor)
// 256 bytes of device zeros (allocated once): where branch-free kernels point out-of-range loads.
const float* zero_page();
// 1: GEMM kernels form fp32 products on the bf16 matrix pipe from exact three-way operand splits (default); 0: fp32 MFMA.
int matrix_path();
void set_last_error(const char* msg);
void note_kernel(const char* name, double flops, int blocks, int split_k);

// Optional per-launch timing with HIP events on the library stream (bench.py's roofline leg).
// Off by default: when off, KernelTimer is two predictable branches.
struct KernelTimer {
  // flops = ALGORITHMIC work of the call (what bench.py's roofline divides by the event time); executed = the MFMA work the
  // launch actually issues when that differs (dgrad gathers run border taps on the zero page); 0 = same as flops
  KernelTimer(const char* name, const char* op, double flops, double bytes, double executed = 0.0);
  ~KernelTimer();
  int slot;
};

// Deferred epilogues (state.hip; convnet_hip_set_deferred_epilogues, include/convnet_hip.h): with the switch on, convUp / convDown /
// MaxPoolUndo / ResponseNormCrossMap are not launched when they are called but parked — ONE call deep — until the next library call.
// If that call is the element-wise pass the reference's host issues right behind them on the same matrix (add_row_vec of the shared bias
// and lower_bound_scalar(0): src/conv_edge.cc:138-149, src/layer.cc:549-551; apply_rectified_linear_deriv: src/layer.cc:556-558), it is
// absorbed into the parked call's fused epilogue; any other call flushes the parked one first (stream() does it, so no entry point can
// overtake it).  Off by default: the unfused entries then run exactly when called.
struct PendingOp {
  int kind;            // 0 none, 1 conv up, 2 conv down, 3 max-pool undo, 4 response norm
  cudamat m[4];        // copies of the call's operands (the host may reshape its structs in between)
  Shape4D s[3];
  ConvDesc desc;
  float scaleTargets;
  int i0, i1;          // response norm: numFilters, sizeF
  float f0, f1;        //                addScale, powScale
  bool b0;             //                blocked
  cudamat bias, mask;  // what has been absorbed so far
  int has_bias, relu, has_mask;
  void (*launch)(PendingOp&);
};
bool defer_begin(int kind, void (*launch)(PendingOp&));   // false: switch off (the caller launches at once); true: fill pending()
PendingOp& pending();
void flush_pending();
extern long g_absorbed;

[[noreturn]] inline void fatal(const char* what, const char* file, int line) {
  // Same policy as the reference's conv back-end: shape/HIP errors are unrecoverable
  // (cudamat_conv_gemm.cu:35-42 getLastCudaError -> exit(EXIT_FAILURE)).
  fprintf(stderr, "libconvnet_hip: fatal: %s (%s:%d)\n", what, file, line);
  exit(EXIT_FAILURE);
}

#define CHIP_CHECK(expr)                                                            \
  do {                                                                              \
    hipError_t _e = (expr);                                                         \
    if (_e != hipSuccess) ::chip::fatal(hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

#define CHIP_REQUIRE(cond)                                       \
  do {                                                           \
    if (!(cond)) ::chip::fatal("check failed: " #cond, __FILE__, __LINE__); \
  } while (0)

inline int launch_status() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_last_error(hipGetErrorString(e));
    return CUDA_ERROR;
  }
  return 0;
}

// Launcher switches read from the environment ONCE per process.  None of them changes a result: they select between kernels /
// schedules that compute the same thing (A/B runs).  The product build honours only the documented few (CHIP_KNOB:
// CONVNET_GG_PATCH, CONVNET_GG_SPLIT); the experiment knobs of earlier rounds exist only in a -DCONVNET_DIAG build and are
// compile-time constants otherwise.
inline int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e && *e ? atoi(e) : dflt;
}
#define CHIP_KNOB(name, dflt) ([] { static const int v_ = ::chip::env_int(name, dflt); return v_; }())
#ifdef CONVNET_DIAG
#define CHIP_DIAG_KNOB(name, dflt) CHIP_KNOB(name, dflt)
#else
#define CHIP_DIAG_KNOB(name, dflt) (dflt)
#endif

inline int divup(int a, int b) { return (a + b - 1) / b; }
inline size_t numel(const cudamat* m) { return (size_t)m->size[0] * (size_t)m->size[1]; }

}  // namespace chip
