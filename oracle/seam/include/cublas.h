/* TEST INFRASTRUCTURE (oracle/seam) — stands in for NVIDIA's <cublas.h> so that the reference's UNMODIFIED
 * cudamat/cudamat.cuh + cudamat_conv_gemm.cuh + src/matrix.h + src/matrix.cc compile with g++ against this repo's
 * library.  It supplies exactly the four CUDA names those files mention outside the cudamat ABI
 * (cudamat.cuh:36,110-112; matrix.h:226; matrix.cc:536-537).  Not used by the product. */
#pragma once
#include <stddef.h>
typedef void* cudaEvent_t;                       /* matrix.h:226 `cudaEvent_t ready_` — opaque handle, as hipEvent_t is */
typedef unsigned long long cudaTextureObject_t;  /* cudamat.cuh:36 — the opaque 64-bit slot of struct cudamat */
typedef int cudaError_t;
#define cudaSuccess 0
#ifdef __cplusplus
extern "C"
#endif
cudaError_t cudaGetDevice(int* dev);             /* matrix.cc:536 — forwarded to hipGetDevice in seam_matrix.cc */
