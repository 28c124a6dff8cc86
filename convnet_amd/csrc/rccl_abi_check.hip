// Build-time check of the NCCL ABI constants comm.hip states by hand (it never includes the RCCL headers: the library builds without
// them and finds every entry point with dlsym).  Where <rccl/rccl.h> is installed — this image: /opt/rocm/include/rccl — the values
// are compared here; elsewhere this translation unit is empty.  No code, no symbols.
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
static_assert((int)ncclSuccess == 0, "comm.hip: ncclSuccess");
static_assert((int)ncclSum == 0, "comm.hip: ncclSum");
static_assert((int)ncclFloat == 7, "comm.hip: ncclFloat");
static_assert(NCCL_UNIQUE_ID_BYTES == 128 && sizeof(ncclUniqueId) == 128, "comm.hip: ncclUniqueId is 128 bytes");
#endif
