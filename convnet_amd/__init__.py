"""convnet_amd — MI355X-native implementation of TorontoDeepLearning/convnet's data-parallel
training hot path (conv fprop/dgrad/wgrad, pooling, response norm, FC, SGD, gradient exchange).

Layout:
  csrc/           hand-written HIP (gfx950) kernels + the C ABI (include/convnet_hip.h)
  _lib.py         ctypes binding of lib/libconvnet_hip.so
  matrix.py       class Matrix  (mirror of the reference's src/matrix.h)
  pbtxt.py        protobuf text-format reader for proto/convnet_config.proto models
  edge.py, layer.py, optimizer.py, loss_functions.py   operator classes (src/*_edge.cc, layer.cc, ...)
  convnet.py      ConvNet driver (src/convnet.cc): build/sort/alloc, Fprop/Bprop/TrainOneBatch
  data_parallel.py  RCCL gradient exchange (replaces the MPI Accumulate/Broadcast of convnet.cc:407-450)
"""
import os as _os

# A training step drives up to three HIP streams of its own (compute, weight gradients + optimizer steps, gradient exchange) and
# RCCL adds its internal ones.  The HIP runtime multiplexes every stream of a process onto GPU_MAX_HW_QUEUES hardware queues
# (default 4); two streams that land on one queue run strictly one after the other.  Measured on the MI355X with a 1-rank exchange:
# 12.1 ms per AlexNet step at the default, 11.3 ms with 8 queues (11.1 ms without the exchange) — the "0.7-0.8 ms a 1-rank exchange
# costs" of round 2 was the second stream sharing a queue with the first.  Must be in the environment before the HIP runtime
# initialises (the first device call), so it is set at package import; an explicit setting wins.  C/C++ hosts: INTEGRATION.md §4.
if "GPU_MAX_HW_QUEUES" not in _os.environ:
    import sys as _sys
    _t = _sys.modules.get("torch")
    if _t is not None and _t.cuda.is_initialized():
        # too late: the HIP runtime read the variable when it initialised (import convnet_amd before the first device call, or export it)
        import warnings as _warnings
        _warnings.warn("convnet_amd imported after the HIP runtime initialised: GPU_MAX_HW_QUEUES stays at the runtime's default (4); "
                       "the second stream of the training step may serialise behind the first (export GPU_MAX_HW_QUEUES=8)")
    else:
        _os.environ["GPU_MAX_HW_QUEUES"] = "8"

__version__ = "0.1"
