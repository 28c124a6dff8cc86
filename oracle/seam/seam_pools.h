// TEST INFRASTRUCTURE (oracle/seam).  The reference's GPU Matrix keeps static pools (`ones_`, `temp_`, src/matrix.cc:11) and a
// static bookkeeping map (`gpu_memory_`, matrix.cc:15).  The map is defined after the pools, so at process exit it is
// destroyed FIRST, and ~Matrix of every pooled matrix (matrix.cc:36-44) then searches the dead map and writes into its freed
// nodes: heap corruption ("corrupted size vs. prev_size") when the host process exits normally instead of through exit(1).
// A latent defect of the reference, not of this library; the seam libraries avoid it by emptying the pools from an atexit
// handler registered after the statics were constructed (so it runs before their destructors), while the map is alive.
#pragma once
#include <cstdlib>

#include "matrix.h"

struct SeamPools : Matrix {   // the pools are protected statics
  static void Release() {
    temp_.clear();
    ones_.clear();
    gpu_memory_.clear();
  }
  static void ReleaseAtExit() { atexit(&SeamPools::Release); }
};
