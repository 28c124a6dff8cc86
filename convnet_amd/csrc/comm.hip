// Data-parallel gradient exchange behind the C ABI: RCCL all-reduce over xGMI on a communication stream, overlapped with the
// backward pass.  Replaces ConvNet::Accumulate + ConvNet::Broadcast (src/convnet.cc:407-450): the reference D2H-copies the whole
// flat gradient, MPI_Recv-sums every rank's copy on rank 0, divides by num_processes_ (:431), copies back and MPI_Bcasts — all
// after Bprop has returned.  Here a C/C++ host posts each gradient slice (or bucket of slices) the moment it is final:
//
//   convnet_hip_comm_allreduce_avg(grad, offset, count, slot)   right after the edge's ComputeOuter
//        -> event on the library's compute stream, the comm stream waits for it, ncclAllReduce(sum) in place, then
//           grad[i] /= nranks (a true division, like the reference, not a multiply by 1/n), event `slot` recorded
//   convnet_hip_comm_wait(slot)                                  right before that edge's optimizer step
//        -> the compute stream waits for the slot's event (no host sync)
//
// librccl is dlopen'ed at comm_init, so the library keeps a single link-time dependency (libamdhip64) and single-GPU users never
// load it.  One communicator per process = per GPU (one process per GPU, as the reference's MPI ranks).
#include <dlfcn.h>

#include <cstring>
#include <string>

#include "common.h"

// The handful of RCCL declarations this file uses, stated locally so the library builds where the RCCL development headers are
// not installed (it never links librccl: every entry point is looked up with dlsym at comm_init).  Values are the NCCL ABI's
// (rccl.h: ncclSuccess 0, ncclSum 0, ncclFloat 7, NCCL_UNIQUE_ID_BYTES 128).
struct ncclComm;
typedef ncclComm* ncclComm_t;
struct ncclUniqueId { char internal[128]; };
typedef int ncclResult_t;
typedef int ncclDataType_t;
typedef int ncclRedOp_t;
constexpr ncclResult_t ncclSuccess = 0;
constexpr ncclRedOp_t ncclSum = 0;
constexpr ncclDataType_t ncclFloat = 7;
// rccl_abi_check.hip checks these values against <rccl/rccl.h> at build time wherever that header is installed.

namespace chip {
namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
ncclComm_t g_comm = nullptr;
int g_rank = 0, g_nranks = 1;
hipStream_t g_comm_stream = nullptr;
int g_comm_stream_device = -1;   // the device the stream and the slot events were created on
constexpr int kSlots = 256;
hipEvent_t g_done[kSlots] = {}, g_ready[kSlots] = {};   // per slot: "slice is final" (compute stream) / "all-reduce done" (comm stream)
bool g_posted[kSlots] = {};

int fail(const char* what, const char* detail) {
  std::string m = std::string(what) + ": " + (detail ? detail : "");
  set_last_error(m.c_str());
  fprintf(stderr, "libconvnet_hip: %s\n", m.c_str());
  return CUDA_ERROR;
}

template <typename F>
bool sym(F& f, const char* name) {
  f = reinterpret_cast<F>(dlsym(g_rccl.handle, name));
  return f != nullptr;
}

int load_rccl() {
  if (g_rccl.handle) return 0;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (g_rccl.handle) break;
  }
  if (!g_rccl.handle) return fail("dlopen(librccl)", dlerror());
  if (!sym(g_rccl.GetUniqueId, "ncclGetUniqueId") || !sym(g_rccl.CommInitRank, "ncclCommInitRank") ||
      !sym(g_rccl.CommDestroy, "ncclCommDestroy") || !sym(g_rccl.AllReduce, "ncclAllReduce") ||
      !sym(g_rccl.Broadcast, "ncclBroadcast") || !sym(g_rccl.GetErrorString, "ncclGetErrorString"))
    return fail("librccl", "missing symbol");
  return 0;
}

int nccl_ok(ncclResult_t r, const char* what) {
  if (r == ncclSuccess) return 0;
  return fail(what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "rccl error");
}

__global__ void comm_divide_kernel(float* __restrict__ p, size_t n, float d) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = p[i] / d;
}

}  // namespace
}  // namespace chip

using namespace chip;

extern "C" {

int convnet_hip_comm_unique_id(char* id_out) {
  if (int rc = load_rccl()) return rc;
  ncclUniqueId id;
  if (int rc = nccl_ok(g_rccl.GetUniqueId(&id), "ncclGetUniqueId")) return rc;
  static_assert(sizeof(id.internal) == CONVNET_HIP_COMM_ID_BYTES, "id size");
  memcpy(id_out, id.internal, sizeof id.internal);
  return 0;
}

int convnet_hip_comm_init(int rank, int nranks, const char* id_in) {
  if (g_comm) return fail("convnet_hip_comm_init", "communicator already initialised");
  if (rank < 0 || nranks < 1 || rank >= nranks || !id_in) return ERROR_GENERIC;
  if (int rc = load_rccl()) return rc;
  ncclUniqueId id;
  memcpy(id.internal, id_in, sizeof id.internal);
  if (int rc = nccl_ok(g_rccl.CommInitRank(&g_comm, nranks, id, rank), "ncclCommInitRank")) return rc;
  g_rank = rank;
  g_nranks = nranks;
  // a stream that does not implicitly synchronise with the legacy default stream the reference host computes on.  Created ONCE per
  // process and kept across communicator lifetimes: every stream a process creates takes the next of GPU_MAX_HW_QUEUES hardware
  // queues, and a host that re-initialises the exchange (bench.py's strong-scaling leg after its weak run) got a comm stream that
  // shared a queue with its compute streams — 12.6 ms per step instead of 10.6 with a world of one.
  // (... per DEVICE: a host that moved to another GPU between communicators — cuda_set_device — gets a stream and events of its own there)
  int dev = 0;
  CHIP_CHECK(hipGetDevice(&dev));
  if (g_comm_stream && dev != g_comm_stream_device) {
    hipStreamSynchronize(g_comm_stream);
    hipStreamDestroy(g_comm_stream);
    for (int i = 0; i < kSlots; ++i) {
      hipEventDestroy(g_done[i]);
      hipEventDestroy(g_ready[i]);
    }
    g_comm_stream = nullptr;
  }
  if (!g_comm_stream) {
    g_comm_stream_device = dev;
    CHIP_CHECK(hipStreamCreateWithFlags(&g_comm_stream, hipStreamNonBlocking));
    for (int i = 0; i < kSlots; ++i) {
      CHIP_CHECK(hipEventCreateWithFlags(&g_done[i], hipEventDisableTiming));
      CHIP_CHECK(hipEventCreateWithFlags(&g_ready[i], hipEventDisableTiming));
    }
  }
  for (int i = 0; i < kSlots; ++i) g_posted[i] = false;
  return 0;
}

int convnet_hip_comm_rank(void) { return g_rank; }
int convnet_hip_comm_size(void) { return g_nranks; }
// slots [0, max_slots) exist; a host plans its posts per step against this number BEFORE the backward pass starts
int convnet_hip_comm_max_slots(void) { return kSlots; }

// ConvNet::Broadcast (src/convnet.cc:407-413): root's copy of `mat` to every rank, ordered after the compute stream's work
// and complete (host-synchronised) on return — it runs once, after initialisation.
int convnet_hip_comm_broadcast(cudamat* mat, int root) {
  if (!g_comm) return fail("convnet_hip_comm_broadcast", "no communicator (convnet_hip_comm_init)");
  if (!mat || !mat->on_device) return ERROR_NOT_ON_DEVICE;
  CHIP_CHECK(hipStreamSynchronize(stream()));
  if (int rc = nccl_ok(g_rccl.Broadcast(mat->data_device, mat->data_device, numel(mat), ncclFloat, root, g_comm, g_comm_stream), "ncclBroadcast"))
    return rc;
  CHIP_CHECK(hipStreamSynchronize(g_comm_stream));
  return 0;
}

int convnet_hip_comm_allreduce_avg(cudamat* flat, size_t offset, size_t count, int slot) {
  if (!g_comm) return fail("convnet_hip_comm_allreduce_avg", "no communicator (convnet_hip_comm_init)");
  if (!flat || !flat->on_device) return ERROR_NOT_ON_DEVICE;
  if (slot < 0 || slot >= kSlots) return ERROR_GENERIC;
  if (offset + count > numel(flat)) return ERROR_INCOMPATIBLE_DIMENSIONS;
  if (count == 0) return 0;
  float* p = flat->data_device + offset;
  CHIP_CHECK(hipEventRecord(g_ready[slot], stream()));         // the slice is final on the compute stream
  CHIP_CHECK(hipStreamWaitEvent(g_comm_stream, g_ready[slot], 0));
  // One rank: the mean over ranks is the identity and nothing is sent.  (RCCL would still run its oneRankReduce kernel over the
  // slice — measured 0.54 ms of device time per AlexNet step, 250 MB at ~0.7 TB/s, beside the backward pass.)  The events and
  // stream waits below stay, so a one-rank run exercises the same ordering as an N-rank one.
  if (g_nranks > 1) {
    if (int rc = nccl_ok(g_rccl.AllReduce(p, p, count, ncclFloat, ncclSum, g_comm, g_comm_stream), "ncclAllReduce")) return rc;
    size_t nb = (count + 255) / 256;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(comm_divide_kernel, dim3((unsigned)nb), dim3(256), 0, g_comm_stream, p, count, (float)g_nranks);
  }
  CHIP_CHECK(hipEventRecord(g_done[slot], g_comm_stream));
  g_posted[slot] = true;
  return launch_status();
}

int convnet_hip_comm_wait(int slot) {
  if (slot < 0 || slot >= kSlots) return ERROR_GENERIC;
  if (!g_posted[slot]) return fail("convnet_hip_comm_wait", "nothing was posted in this slot");
  CHIP_CHECK(hipStreamWaitEvent(stream(), g_done[slot], 0));
  return 0;
}

// Host-blocking drain of the communication stream (before reading gradients on the host); clears the slots.
int convnet_hip_comm_sync(void) {
  if (!g_comm) return 0;
  CHIP_CHECK(hipStreamSynchronize(g_comm_stream));
  for (bool& b : g_posted) b = false;
  return 0;
}

int convnet_hip_comm_destroy(void) {
  if (!g_comm) return 0;
  hipStreamSynchronize(g_comm_stream);
  g_rccl.CommDestroy(g_comm);
  g_comm = nullptr;
  for (int i = 0; i < kSlots; ++i) g_posted[i] = false;   // the stream and the events stay for the next communicator (see comm_init)
  g_rank = 0;
  g_nranks = 1;
  return 0;
}

}  // extern "C"
