// EXPERIMENT (not part of the library): can ONE wave per SIMD keep the fp32 MFMA pipe busy when it owns a 128x128 accumulator
// tile (256 AGPRs) and issues the next chunk's staging in the issue shadow of the current chunk's MFMAs?  See NOTES.md
// ("largest remaining lever").  Plain GEMM in the operand layouts the conv kernels use:
//     A[k][r]  (weights, r contiguous; M rows)      B[k][n]  (images, n contiguous; N columns)      C[r][n]  (n contiguous)
// Block = 4 waves (one per SIMD) = 256x256 tile, wave (wr, wc) = 128x128; BK = 16; 3 LDS stages of 32 KB filled with
// global_load_lds (16 B per lane, 8 per thread per chunk, one issued per k-step); fragments read one k-step ahead with two
// ds_read_b128 per k-step (rows/cols permuted so that a lane's four tiles are contiguous); one barrier per chunk.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/wave1_gemm tools/wave1_gemm.hip ; run on the GPU box:
//     tools/wave1_gemm [M N K [variant]] | [variant]     (defaults 4096 4096 4096, variant 2; prints TFLOP/s and the max error against a sampled fp64 check)
// Status (end of round 1): first untuned version, run once on an MI355X: 4096^3 in 1.144 ms = 120.2 TFLOP/s, results correct
// (max rel err 4e-5 on 64 fp64-checked samples).  See NOTES.md for what to try next.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int BK = 16, TM = 256, TN = 256, STAGES = 3;
constexpr int A_STAGE = BK * TM, B_STAGE = BK * TN, STAGE = A_STAGE + B_STAGE;   // floats

// VARIANT 0: staging / barrier guarded by wave-uniform branches (the version measured at 120.2 TFLOP/s).  The branches split
//            every k-step into its own basic block and the compiler then waits lgkmcnt(0) — i.e. also for the two fragment
//            reads it has just issued — in every second k-step.
// VARIANT 1: branch-free chunk body: past the end of K the staging re-reads the last chunk into a stage nobody reads again, the
//            barrier and the look-ahead fragment read run unconditionally; one basic block per chunk, exact waitcnts.
template <int VARIANT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
wave1_gemm(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int r0 = blockIdx.y * TM, c0 = blockIdx.x * TN;
  const int nchunks = K / BK;

  // staging: piece p = tid + 256*i (i < 4) of the A tile and of the B tile; krow = p/64, 16-byte column = p%64
  const float* a_src[4];
  const float* b_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = tid + 256 * i, krow = p >> 6, c4 = p & 63;
    a_src[i] = A + (size_t)krow * M + r0 + 4 * c4;
    b_src[i] = B + (size_t)krow * N + c0 + 4 * c4;
  }
  auto stage_piece = [&](int i, int stage) __attribute__((always_inline)) {   // i in 0..7: 4 A pieces then 4 B pieces
    float* base = smem + stage * STAGE + (i < 4 ? 0 : A_STAGE) + 4 * (64 * wave + 256 * (i & 3));
    const float* src = i < 4 ? a_src[i] : b_src[i - 4];
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)base, 16, 0, 0);
  };
  auto advance = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a_src[i] += (size_t)BK * M;
      b_src[i] += (size_t)BK * N;
    }
  };

  f32x16 acc[4][4];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[rt][ct][e] = 0.f;

  // prologue: chunks 0 and 1 in flight, chunk 0 landed
#pragma unroll
  for (int i = 0; i < 8; ++i) stage_piece(i, 0);
  advance();
  if (nchunks > 1) {
#pragma unroll
    for (int i = 0; i < 8; ++i) stage_piece(i, 1);
    advance();
    __builtin_amdgcn_s_waitcnt(0x0f78);   // vmcnt(8): the first eight (chunk 0) have landed
  } else {
    __builtin_amdgcn_s_waitcnt(0x0f70);
  }
  __syncthreads();

  // fragment addresses: lane j = lane%32 owns rows 4j..4j+3 (one per row tile) and cols 4j..4j+3, k = 2*ks + lane/32
  const int fj = lane & 31, fk = lane >> 5;
  auto frag = [&](int stage, int ks, f32x4& fa, f32x4& fb) __attribute__((always_inline)) {
    const float* s = smem + stage * STAGE;
    fa = *reinterpret_cast<const f32x4*>(s + (2 * ks + fk) * TM + wr * 128 + 4 * fj);
    fb = *reinterpret_cast<const f32x4*>(s + A_STAGE + (2 * ks + fk) * TN + wc * 128 + 4 * fj);
  };
  f32x4 fa, fb, na, nb;
  frag(0, 0, fa, fb);

  if constexpr (VARIANT == 0) {
  for (int c = 0; c < nchunks; ++c) {
    const int st = c % STAGES, st_next = (c + 1) % STAGES, st_fill = (c + 2) % STAGES;
    const bool fill = c + 2 < nchunks;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      if (ks < 7) {
        frag(st, ks + 1, na, nb);
      } else if (c + 1 < nchunks) {
        // chunk c+1 must have landed and every wave must be done reading
        // stage st_fill's previous contents before anyone overwrites it next iteration
        // (a bare s_barrier: __syncthreads() would add a full vmcnt(0) fence and wait for the loads just issued for chunk c+2)
        // at this point seven pieces of chunk c+2 have been issued after the eight of chunk c+1: vmcnt(7) = chunk c+1 complete
        if (fill) __builtin_amdgcn_s_waitcnt(0x0077); else __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(7|0) + lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        frag(st_next, 0, na, nb);
      }
      if (fill) stage_piece(ks, st_fill);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[rt], fb[ct], acc[rt][ct], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      fa = na;
      fb = nb;
    }
    if (fill) advance();
  }
  } else if constexpr (VARIANT == 1) {
  int st = 0;   // stage of chunk c; chunk c+1 is in st+1, chunk c+2 is being filled into st+2 (mod 3)
  for (int c = 0; c < nchunks; ++c) {
    const int st_next = st == STAGES - 1 ? 0 : st + 1, st_fill = st_next == STAGES - 1 ? 0 : st_next + 1;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      if (ks < 7) {
        frag(st, ks + 1, na, nb);
      } else {
        __builtin_amdgcn_s_waitcnt(0x0077);   // vmcnt(7): chunk c+1 complete (7 pieces of chunk c+2 are younger); lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        frag(st_next, 0, na, nb);             // past the last chunk: reads a stale stage, never used
      }
      stage_piece(ks, st_fill);               // past the end of K: re-reads the last chunk into a stage nobody reads again
      __builtin_amdgcn_sched_barrier(0);      // keep the look-ahead reads ABOVE this step's MFMAs (the scheduler sinks them to their use)
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[rt], fb[ct], acc[rt][ct], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      fa = na;
      fb = nb;
    }
    if (c + 3 < nchunks) advance();           // the source pointers stop at the last chunk
    st = st_next;
  }
  __builtin_amdgcn_s_waitcnt(0x0f70);         // the trailing dummy loads must land before the LDS is released
  } else {
  // VARIANT 2: as 1, but the look-ahead fragment reads and the staging load sit in the MIDDLE of the step's 16 MFMAs.  The
  // compiler waits lgkmcnt(0) before the first use of a fragment whatever else is in flight (it treats the LDS-DMA loads as
  // LDS traffic); with the reads issued 8 MFMAs = 512 pipe cycles before that wait, it never finds anything outstanding.
  int st = 0;
  for (int c = 0; c < nchunks; ++c) {
    const int st_next = st == STAGES - 1 ? 0 : st + 1, st_fill = st_next == STAGES - 1 ? 0 : st_next + 1;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[rt], fb[ct], acc[rt][ct], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (ks < 7) {
        frag(st, ks + 1, na, nb);
      } else {
        __builtin_amdgcn_s_waitcnt(0x0077);   // vmcnt(7): chunk c+1 complete; lgkmcnt(0): this wave is done reading stage st
        __builtin_amdgcn_s_barrier();
        frag(st_next, 0, na, nb);
      }
      stage_piece(ks, st_fill);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int rt = 2; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[rt], fb[ct], acc[rt][ct], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      fa = na;
      fb = nb;
    }
    if (c + 3 < nchunks) advance();
    st = st_next;
  }
  __builtin_amdgcn_s_waitcnt(0x0f70);
  }

  // epilogue: logical row m of row tile rt = physical row 4m + rt; logical col n = lane%32 of col tile ct = physical col 4n + ct
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int m = (e & 3) + 4 * fk + 8 * (e >> 2);
      const f32x4 v = {acc[rt][0][e], acc[rt][1][e], acc[rt][2][e], acc[rt][3][e]};
      *reinterpret_cast<f32x4*>(C + (size_t)(r0 + wr * 128 + 4 * m + rt) * N + c0 + wc * 128 + 4 * fj) = v;
    }
}

int main(int argc, char** argv) {
  const int M = argc > 3 ? atoi(argv[1]) : 4096, N = argc > 3 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 4096;
  if (M % TM || N % TN || K % BK) {
    fprintf(stderr, "M, N must be multiples of 256 and K of 16\n");
    return 2;
  }
  std::vector<float> hA((size_t)K * M), hB((size_t)K * N), hC((size_t)M * N);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) * (1.0f / 16777216.0f)) - 0.5f; };
  for (auto& v : hA) v = rnd();
  for (auto& v : hB) v = rnd();
  float *A, *B, *C;
  hipMalloc(&A, hA.size() * 4); hipMalloc(&B, hB.size() * 4); hipMalloc(&C, hC.size() * 4);
  hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
  const size_t lds = sizeof(float) * STAGES * STAGE;
  const int variant = argc > 4 ? atoi(argv[4]) : (argc == 2 ? atoi(argv[1]) : 2);
  auto kernel = variant == 0 ? wave1_gemm<0> : variant == 1 ? wave1_gemm<1> : wave1_gemm<2>;
  hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  dim3 grid(N / TN, M / TM), block(256);
  hipLaunchKernelGGL(kernel, grid, block, lds, 0, A, B, C, M, N, K);
  if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 10;
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kernel, grid, block, lds, 0, A, B, C, M, N, K);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int t = 0; t < 64; ++t) {
    const int r = (int)(((unsigned)t * 2654435761u) % (unsigned)M), n = (int)(((unsigned)t * 40503u + 17u) % (unsigned)N);
    double ref = 0;
    for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)k * M + r] * hB[(size_t)k * N + n];
    worst = fmax(worst, fabs(ref - hC[(size_t)r * N + n]) / (fabs(ref) + 1e-3));
  }
  printf("wave1_gemm<%d> %dx%dx%d: %.3f ms  %.1f TFLOP/s  blocks=%d  max rel err (64 samples) %.2e\n", variant, M, N, K, ms,
         2.0 * M * N * (double)K / ms * 1e-9, grid.x * grid.y, worst);
  return worst < 1e-3 ? 0 : 1;
}
