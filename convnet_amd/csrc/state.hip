// Library state, memory plumbing and views: the non-compute part of the cudamat ABI
// (reference cudamat/cudamat.cu:40-160,360-640), re-done over the HIP runtime.
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "common.h"

namespace chip {
namespace {
hipStream_t g_stream = nullptr;
std::string g_last_error;
ConvnetHipKernelInfo g_info = {"none", 0.0, 0, 1};
}  // namespace

// ---- deferred epilogues (common.h: PendingOp) -------------------------------------------------------------------------------------
namespace {
int g_defer = -1;   // -1: not decided (first use reads CONVNET_DEFER_EPILOGUES; default off)
PendingOp g_pending = {};
bool g_flushing = false;
inline bool defer_on() {
  if (g_defer < 0) g_defer = env_int("CONVNET_DEFER_EPILOGUES", 0) != 0 ? 1 : 0;
  return g_defer != 0;
}
}  // namespace
long g_absorbed = 0;   // element-wise calls absorbed into a parked call so far (elementwise.hip counts; tests read it)
PendingOp& pending() { return g_pending; }
void flush_pending() {
  if (g_pending.kind == 0 || g_flushing) return;
  PendingOp o = g_pending;
  g_pending.kind = 0;
  g_flushing = true;   // (the launch itself asks for the stream)
  o.launch(o);
  g_flushing = false;
}
bool defer_begin(int kind, void (*launch)(PendingOp&)) {
  if (!defer_on() || g_flushing) return false;
  flush_pending();
  g_pending = PendingOp{};
  g_pending.kind = kind;
  g_pending.launch = launch;
  return true;
}

// Every launch, copy, event and synchronisation of the library asks for the stream here: a parked call goes out first.
hipStream_t stream() {
  flush_pending();
  return g_stream;
}

int g_matrix_path = -1;   // -1: not decided yet (first use reads CONVNET_GG_SPLIT; default 0 = IEEE fp32 products)
int matrix_path() {
  if (g_matrix_path < 0) {
    const char* e = getenv("CONVNET_GG_SPLIT");
    g_matrix_path = (e && *e) ? (atoi(e) != 0 ? 1 : 0) : 0;
  }
  return g_matrix_path;
}

// Scratch arenas are per stream: a host that drives a second stream through convnet_hip_set_stream (e.g.
// optimizer updates beside the backward pass) gets its own split-K slabs, so concurrent launches on two
// streams never share a base pointer.  Grow-only; growth waits for that stream's in-flight users first and
// rounds up generously so a training run reaches steady state after the first step.
namespace {
struct Arena {
  void* p = nullptr;
  size_t cap = 0;
};
std::map<hipStream_t, Arena> g_arena[3];

void* arena_get(int which, size_t bytes, size_t slack) {
  flush_pending();
  Arena& a = g_arena[which][g_stream];
  if (bytes <= a.cap) return a.p;
  CHIP_CHECK(hipStreamSynchronize(stream()));
  if (a.p) CHIP_CHECK(hipFree(a.p));
  a.cap = ((bytes + slack) >> 20) << 20;
  CHIP_CHECK(hipMalloc(&a.p, a.cap));
  return a.p;
}
}  // namespace

void* workspace(size_t bytes) { return arena_get(0, bytes, size_t(1) << 22); }

// Second, independent arena: the dgrad filter images live here while the same call may take
// split-K slabs from workspace() (two simultaneous users must not share one base pointer).
void* workspace_aux(size_t bytes) { return arena_get(1, bytes, size_t(1) << 20); }
// Third arena: the bf16 planes of the source tensor of one gather-GEMM call (patch_gemm.hip), alive beside the other two.
void* workspace_planes(size_t bytes) { return arena_get(2, bytes, size_t(1) << 22); }

const float* zero_page() {
  static float* z = nullptr;
  if (!z) {
    CHIP_CHECK(hipMalloc((void**)&z, 256));
    CHIP_CHECK(hipMemset(z, 0, 256));
    const float ones[4] = {1.f, 1.f, 1.f, 1.f};   // floats [32, 36): the constant-1 input of wg_kernel's bias row
    CHIP_CHECK(hipMemcpy(z + 32, ones, sizeof ones, hipMemcpyHostToDevice));
  }
  return z;
}

void set_last_error(const char* msg) { g_last_error = msg ? msg : ""; }

void note_kernel(const char* name, double flops, int blocks, int split_k) {
  g_info.name = name;
  g_info.flops = flops;
  g_info.grid_blocks = blocks;
  g_info.split_k = split_k;
}

// ---- per-launch HIP-event timing ------------------------------------------------------------------
namespace {
struct ProfRec {
  std::string name, op;
  double flops, bytes, executed;
  hipEvent_t start, stop;
};
bool g_prof_on = false;
std::vector<ProfRec> g_prof;
std::vector<hipEvent_t> g_event_pool;
hipEvent_t get_event() {
  if (!g_event_pool.empty()) {
    hipEvent_t e = g_event_pool.back();
    g_event_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  CHIP_CHECK(hipEventCreate(&e));
  return e;
}
}  // namespace

KernelTimer::KernelTimer(const char* name, const char* op, double flops, double bytes, double executed) : slot(-1) {
  if (!g_prof_on) return;
  ProfRec r{name, op, flops, bytes, executed > 0.0 ? executed : flops, get_event(), get_event()};
  CHIP_CHECK(hipEventRecord(r.start, stream()));
  g_prof.push_back(r);
  slot = (int)g_prof.size() - 1;
}
KernelTimer::~KernelTimer() {
  if (slot >= 0) CHIP_CHECK(hipEventRecord(g_prof[slot].stop, stream()));
}

__global__ void transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols) {
  // dst (cols x rows, col-major) = src^T, src (rows x cols, col-major).  32x32 LDS tile (+1 pad).
  __shared__ float tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int j = ty; j < 32; j += 8) {
    const int r = bx + tx, c = by + j;
    if (r < rows && c < cols) tile[j][tx] = src[(size_t)r + (size_t)rows * c];
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = by + tx, r = bx + j;
    if (r < rows && c < cols) dst[(size_t)c + (size_t)cols * r] = tile[tx][j];
  }
}

}  // namespace chip

using namespace chip;

extern "C" {

int convnet_hip_init(int device_id) {
  if (hipSetDevice(device_id) != hipSuccess) return CUDA_ERROR;
  return 0;
}

void convnet_hip_shutdown(void) {
  flush_pending();   // (a parked call uses the arenas freed below)
  hipDeviceSynchronize();
  for (auto& per_stream : g_arena) {
    for (auto& kv : per_stream)
      if (kv.second.p) hipFree(kv.second.p);
    per_stream.clear();
  }
}

void convnet_hip_set_stream(void* s) {
  flush_pending();   // a parked call belongs to the stream it was made on
  g_stream = (hipStream_t)s;
}
void convnet_hip_set_deferred_epilogues(int on) {
  flush_pending();
  g_defer = on != 0 ? 1 : 0;
}
int convnet_hip_get_deferred_epilogues(void) { return defer_on() ? 1 : 0; }
long convnet_hip_deferred_absorbed(void) { return chip::g_absorbed; }
void* convnet_hip_get_stream(void) {
  flush_pending();   // a host that enqueues its own work on the stream must find the parked call in front of it
  return (void*)g_stream;
}

int convnet_hip_reserve_workspace(size_t bytes) {
  workspace(bytes);
  return 0;
}

const char* convnet_hip_version(void) { return "convnet_hip 0.2 (gfx950; fp32 products via bf16-split or fp32 MFMA)"; }
void convnet_hip_set_matrix_path(int path) { chip::g_matrix_path = path != 0 ? 1 : 0; }
int convnet_hip_get_matrix_path(void) { return chip::matrix_path(); }
const char* get_last_cuda_error(void) {
  flush_pending();
  return g_last_error.c_str();
}

int cuda_set_device(int deviceId) {
  flush_pending();   // a parked call holds pointers of the device that was current when it was made
  return hipSetDevice(deviceId) == hipSuccess ? 0 : CUDA_ERROR;
}

void cuda_sync_threads(void) { CHIP_CHECK(hipStreamSynchronize(stream())); }

static int event_status(hipError_t err) {
  if (err != hipSuccess) {
    set_last_error(hipGetErrorString(err));
    printf("%s\n", hipGetErrorString(err));
  }
  return err != hipSuccess;
}
int cuda_create_event(void** t) { return event_status(hipEventCreateWithFlags(reinterpret_cast<hipEvent_t*>(t), hipEventDisableTiming)); }
int cuda_record_event(void** t) { return event_status(hipEventRecord(*reinterpret_cast<hipEvent_t*>(t), stream())); }
int cuda_synchronize_event(void** t) { return event_status(hipStreamWaitEvent(stream(), *reinterpret_cast<hipEvent_t*>(t), 0)); }
// The reference's Matrix destructor releases the texture view of every matrix (src/matrix.cc:~Matrix -> cudamat.cu destroy_tex);
// this library never binds textures (tex_obj stays 0), so there is nothing to release.
int destroy_tex(cudamat* mat) {
  if (mat) mat->tex_obj = 0;
  return 0;
}
int cublas_init(void) { return 0; }
int cublas_shutdown(void) {
  convnet_hip_shutdown();
  return 0;
}

void convnet_hip_last_kernel_info(ConvnetHipKernelInfo* out) {
  flush_pending();   // "the last conv / dot call" includes a parked one
  *out = g_info;
}

void convnet_hip_profile_enable(int on) { g_prof_on = on != 0; }

// Synchronises the stream, aggregates the recorded launches by (kernel, op) into `buf` as text lines
// "kernel|op|launches|total_ms|total_flops|total_bytes|total_executed_flops" and clears the records.  Returns the
// number of bytes written (0 if nothing was recorded or the buffer is too small).
size_t convnet_hip_profile_report(char* buf, size_t cap) {
  if (g_prof.empty()) return 0;
  hipStreamSynchronize(stream());
  struct Agg { long n = 0; double ms = 0, flops = 0, bytes = 0, executed = 0; };
  std::map<std::string, Agg> agg;
  for (auto& r : g_prof) {
    float ms = 0.f;
    hipEventElapsedTime(&ms, r.start, r.stop);
    Agg& a = agg[r.name + "|" + r.op];
    a.n++; a.ms += ms; a.flops += r.flops; a.bytes += r.bytes; a.executed += r.executed;
    g_event_pool.push_back(r.start);
    g_event_pool.push_back(r.stop);
  }
  g_prof.clear();
  std::string out;
  char line[512];
  for (auto& kv : agg) {
    snprintf(line, sizeof line, "%s|%ld|%.6f|%.6e|%.6e|%.6e\n", kv.first.c_str(), kv.second.n, kv.second.ms, kv.second.flops, kv.second.bytes,
             kv.second.executed);
    out += line;
  }
  if (out.size() + 1 > cap) return 0;
  memcpy(buf, out.c_str(), out.size() + 1);
  return out.size();
}

int allocate_device_memory(cudamat* mat) {
  const size_t bytes = numel(mat) * sizeof(float);
  if (hipMalloc((void**)&mat->data_device, bytes ? bytes : 4) != hipSuccess) {
    set_last_error("hipMalloc failed");
    return CUDA_ERROR;
  }
  mat->on_device = 1;
  return 0;
}

int free_device_memory(cudamat* mat) {
  flush_pending();   // (a parked call may read or write this matrix)
  if (mat->owns_data && mat->on_device) {
    if (hipFree(mat->data_device) != hipSuccess) return CUDA_ERROR;
    mat->on_device = 0;
    mat->data_device = nullptr;
  }
  return 0;
}

int copy_to_host(cudamat* mat) {
  if (!mat->on_device) return ERROR_NOT_ON_DEVICE;
  const size_t bytes = numel(mat) * sizeof(float);
  if (hipMemcpyAsync(mat->data_host, mat->data_device, bytes, hipMemcpyDeviceToHost, stream()) != hipSuccess) return CUDA_ERROR;
  if (hipStreamSynchronize(stream()) != hipSuccess) return CUDA_ERROR;
  mat->on_host = 1;
  return 0;
}

int copy_to_device(cudamat* mat) {
  if (!mat->on_device) {
    int rc = allocate_device_memory(mat);
    if (rc) return rc;
  }
  const size_t bytes = numel(mat) * sizeof(float);
  if (hipMemcpyAsync(mat->data_device, mat->data_host, bytes, hipMemcpyHostToDevice, stream()) != hipSuccess) return CUDA_ERROR;
  if (hipStreamSynchronize(stream()) != hipSuccess) return CUDA_ERROR;
  return 0;
}

int copy_to_host_slice(cudamat* mat, size_t start, size_t end) {
  if (!mat->on_device) return ERROR_NOT_ON_DEVICE;
  if (end > (size_t)mat->size[1] || start > end) return ERROR_INCOMPATIBLE_DIMENSIONS;
  const size_t off = start * mat->size[0], bytes = (end - start) * mat->size[0] * sizeof(float);
  if (hipMemcpyAsync(mat->data_host + off, mat->data_device + off, bytes, hipMemcpyDeviceToHost, stream()) != hipSuccess) return CUDA_ERROR;
  return hipStreamSynchronize(stream()) == hipSuccess ? 0 : CUDA_ERROR;
}

int copy_to_device_slice(cudamat* mat, size_t start, size_t end) {
  // cudamat.cu:325-347: ERROR_GENERIC for an empty / out-of-range slice, and — like copy_to_device — the first copy of a
  // host-only matrix allocates its device memory (the reference's Matrix::AllocateGPUMemory relies on it, matrix.cc:105-106)
  if (end <= start || end > (size_t)mat->size[1]) return ERROR_GENERIC;
  if (!mat->on_device) {
    const int rc = allocate_device_memory(mat);
    if (rc) return rc;
  }
  const size_t off = start * mat->size[0], bytes = (end - start) * mat->size[0] * sizeof(float);
  if (hipMemcpyAsync(mat->data_device + off, mat->data_host + off, bytes, hipMemcpyHostToDevice, stream()) != hipSuccess) return CUDA_ERROR;
  return hipStreamSynchronize(stream()) == hipSuccess ? 0 : CUDA_ERROR;
}

int copy_on_device(cudamat* mat1, cudamat* mat2) {
  if (mat1->size[0] != mat2->size[0] || mat1->size[1] != mat2->size[1]) return ERROR_INCOMPATIBLE_DIMENSIONS;
  if (hipMemcpyAsync(mat2->data_device, mat1->data_device, numel(mat1) * sizeof(float), hipMemcpyDeviceToDevice, stream()) != hipSuccess) return CUDA_ERROR;
  return 0;
}

int copy_transpose(cudamat* source, cudamat* target) {
  if (source->size[0] != target->size[1] || source->size[1] != target->size[0]) return ERROR_INCOMPATIBLE_DIMENSIONS;
  const int rows = source->size[0], cols = source->size[1];
  dim3 grid(divup(rows, 32), divup(cols, 32));
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, stream(), source->data_device, target->data_device, rows, cols);
  return launch_status();
}

int reshape(cudamat* mat, int m, int n) {
  if (m < 0 && n < 0) return ERROR_GENERIC;
  const long long total = (long long)mat->size[0] * mat->size[1];
  if (m < 0) m = (int)(total / n);
  if (n < 0) n = (int)(total / m);
  if (total != (long long)m * n) return ERROR_INCOMPATIBLE_DIMENSIONS;
  mat->size[0] = m;
  mat->size[1] = n;
  return 0;
}

int get_slice(cudamat* source, cudamat* target, unsigned int first_col, unsigned int last_col) {
  if (source->is_trans) return ERROR_TRANSPOSED;
  if (!source->on_device) return ERROR_NOT_ON_DEVICE;
  if (last_col > (unsigned)source->size[1] || first_col >= last_col) return ERROR_INCOMPATIBLE_DIMENSIONS;
  const size_t rows = source->size[0];
  target->data_host = source->data_host ? source->data_host + first_col * rows : nullptr;
  target->data_device = source->data_device + first_col * rows;
  target->on_device = 1;
  target->on_host = 0;
  target->size[0] = source->size[0];
  target->size[1] = last_col - first_col;
  target->is_trans = 0;
  target->owns_data = 0;
  target->tex_obj = 0;
  return 0;
}

void init_from_array(cudamat* mat, float* data, int m, int n) {
  mat->data_host = data;
  mat->data_device = nullptr;
  mat->size[0] = m;
  mat->size[1] = n;
  mat->on_device = 0;
  mat->on_host = 1;
  mat->is_trans = 0;
  mat->owns_data = 1;
  mat->tex_obj = 0;
}

int init_empty(cudamat* mat, int m, int n) {
  mat->data_host = nullptr;
  mat->size[0] = m;
  mat->size[1] = n;
  mat->on_host = 0;
  mat->is_trans = 0;
  mat->owns_data = 1;
  mat->tex_obj = 0;
  return allocate_device_memory(mat);
}

int write_at(cudamat* mat, int row, int col, float val) {
  if (row < 0 || col < 0 || row >= mat->size[0] || col >= mat->size[1]) return ERROR_INCOMPATIBLE_DIMENSIONS;
  if (hipMemcpyAsync(mat->data_device + (size_t)col * mat->size[0] + row, &val, sizeof(float), hipMemcpyHostToDevice, stream()) != hipSuccess) return CUDA_ERROR;
  return hipStreamSynchronize(stream()) == hipSuccess ? 0 : CUDA_ERROR;
}

float read_from(cudamat* mat, int row, int col, int* err_code) {
  *err_code = 0;
  if (row < 0 || col < 0 || row >= mat->size[0] || col >= mat->size[1]) {
    *err_code = ERROR_INCOMPATIBLE_DIMENSIONS;
    return 0.f;
  }
  float v = 0.f;
  if (hipMemcpyAsync(&v, mat->data_device + (size_t)col * mat->size[0] + row, sizeof(float), hipMemcpyDeviceToHost, stream()) != hipSuccess ||
      hipStreamSynchronize(stream()) != hipSuccess)
    *err_code = CUDA_ERROR;
  return v;
}

}  // extern "C"
