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
__version__ = "0.1"
