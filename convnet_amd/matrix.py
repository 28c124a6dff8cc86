"""class Matrix — host-side mirror of the reference's GPU ``class Matrix`` (src/matrix.h:18-234,
src/matrix.cc) for the data-parallel hot path, implemented over libconvnet_hip.so.

Same method names, argument order and argument meaning as the reference, so the operator classes
(edge.py / layer.py / optimizer.py) and the tests read like the reference's C++.  PyTorch is used
for exactly two things: owning device memory (``torch.empty`` -> ``data_ptr()``) and streams.
All arithmetic happens in the HIP library through its C ABI; there is no torch or CPU fallback.

Errors follow the reference: a non-zero ABI code prints ``GetStringError`` and raises
(``cerr << ...; exit(1)`` in src/matrix.cc:175-179 — here ``MatrixError`` so tests can see it).
"""
import contextlib
import ctypes
import os

import numpy as np
import torch

from . import _lib
from ._lib import ConvDesc, Shape4D, cudamat, lib


class MatrixError(RuntimeError):
    pass


def _chk(err_code, what):
    if err_code != 0:
        raise MatrixError(f"Error: {what} : {_lib.GetStringError(err_code)}")


class Matrix:
    # reference statics: temp_/ones_ pools, rnd_ state (src/matrix.cc:10-16)
    _temp = None
    _temp_size = 0
    _ones = None
    _ones_size = 0
    _rnd = None
    _device = None

    def __init__(self, rows=0, cols=0, on_gpu=True):
        self.mat_ = cudamat()
        self.mat_t_ = cudamat()
        self.shape_ = Shape4D()
        self._t = None       # torch storage (kept alive)
        self._host = None    # numpy mirror for GetHostData()
        self.name_ = ""
        if rows * cols > 0:
            self.AllocateGPUMemory(rows, cols)

    # ---- device / stream statics (src/matrix.cc:490-560) ------------------------------------------
    @staticmethod
    def SetupCUDADevice(gpu_id):
        torch.cuda.set_device(gpu_id)
        Matrix._device = torch.device("cuda", gpu_id)
        _chk(lib.convnet_hip_init(gpu_id), "Could not set device")
        Matrix.UseCurrentStream()
        # The library's default at the C ABI is the IEEE fp32 matrix instruction (path 0); this host — trainer, tests, bench —
        # selects the bf16-split products (include/convnet_hip.h: convnet_hip_set_matrix_path) unless the environment already chose.
        # Once per process: a later SetupCUDADevice (a second net, a test suite's re-init) must not undo a path the host has chosen
        # explicitly through the library in between.
        if not os.environ.get("CONVNET_GG_SPLIT") and not Matrix._path_chosen:
            lib.convnet_hip_set_matrix_path(1)
        Matrix._path_chosen = True

    _path_chosen = False

    _shared_streams = {}

    @staticmethod
    def SharedStream(role):
        """One HIP stream per (device, role) for the whole process — "side" (weight gradients / optimizer steps) and "comm" (the
        gradient exchange).  Every stream a process creates is multiplexed onto GPU_MAX_HW_QUEUES hardware queues in creation
        order, and two streams that share a queue run strictly one after the other; nets built one after another in one process
        (bench.py's strong-scaling leg, test suites) therefore reuse the same two streams instead of adding a new pair per net
        (measured: the third net of a process ran its exchange-enabled step at 13.9 ms instead of 11.0)."""
        key = (torch.cuda.current_device(), role)
        s = Matrix._shared_streams.get(key)
        if s is None:
            s = Matrix._shared_streams[key] = torch.cuda.Stream()
        return s

    @staticmethod
    def UseCurrentStream():
        """Route library launches to torch's current stream (the reference used stream 0)."""
        lib.convnet_hip_set_stream(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))

    @staticmethod
    @contextlib.contextmanager
    def OnStream(stream):
        """Launch the library calls made inside the block on ``stream`` (a torch.cuda.Stream) — the reference's
        cross-GPU ordering tools (cuda_record_event / StreamWaitEvent, cudamat.cu:65-81) applied to a second
        stream of one GPU.  The library keeps separate scratch arenas per stream."""
        prev = lib.convnet_hip_get_stream()
        lib.convnet_hip_set_stream(ctypes.c_void_p(stream.cuda_stream))
        try:
            with torch.cuda.stream(stream):
                yield
        finally:
            lib.convnet_hip_set_stream(ctypes.c_void_p(prev))

    @staticmethod
    def SetDevice(gpu_id):
        pass  # one process per GPU: the device is fixed after SetupCUDADevice

    @staticmethod
    def GetDevice():
        return Matrix._device.index if Matrix._device is not None else 0

    @staticmethod
    def SyncAllDevices():
        torch.cuda.synchronize()

    @staticmethod
    def InitRandom(seed):
        Matrix._rnd = _lib.rnd_struct()
        _chk(lib.init_random(ctypes.byref(Matrix._rnd), int(seed)), "init_random")

    # ---- allocation / views -----------------------------------------------------------------------
    def _dev(self):
        if Matrix._device is None:
            Matrix.SetupCUDADevice(torch.cuda.current_device())
        return Matrix._device

    def _bind(self, t, rows, cols):
        self._t = t
        m = self.mat_
        m.data_host = None
        m.data_device = t.data_ptr()
        m.on_device = 1
        m.on_host = 0
        m.size[0] = rows
        m.size[1] = cols
        m.is_trans = 0
        m.owns_data = 0  # torch owns the storage
        m.tex_obj = 0
        self._host = None
        self.SetupTranspose()

    def AllocateGPUMemory(self, rows, cols, name=""):
        rows, cols = int(rows), int(cols)
        self.name_ = name
        if self._t is not None and rows == self.mat_.size[0] and cols == self.mat_.size[1]:
            return
        self._bind(torch.empty(max(rows * cols, 1), dtype=torch.float32, device=self._dev()), rows, cols)

    def AllocateMainMemory(self, rows, cols):
        self._host = np.zeros((int(cols), int(rows)), np.float32)
        self.mat_.data_host = self._host.ctypes.data_as(_lib.c_float_p)
        self.mat_.on_host = 1

    def SetReady(self):
        """Matrix::SetReady (src/matrix.cc:607-617): record this matrix's event on the current stream."""
        if getattr(self, "_ready", None) is None:
            self._ready = ctypes.c_void_p()
            _chk(lib.cuda_create_event(ctypes.byref(self._ready)), "cuda_create_event")
        _chk(lib.cuda_record_event(ctypes.byref(self._ready)), "cuda_record_event")

    def WaitTillReady(self):
        """Matrix::WaitTillReady (src/matrix.cc:619-631): the current stream (not the host) waits for SetReady."""
        if getattr(self, "_ready", None) is not None:
            _chk(lib.cuda_synchronize_event(ctypes.byref(self._ready)), "cuda_synchronize_event")

    def SetupTranspose(self):
        ctypes.memmove(ctypes.byref(self.mat_t_), ctypes.byref(self.mat_), ctypes.sizeof(cudamat))
        self.mat_t_.is_trans = 1 - self.mat_.is_trans

    def GetMat(self):
        return ctypes.byref(self.mat_)

    def GetMatTranspose(self):
        return ctypes.byref(self.mat_t_)

    def SetShape4D(self, d1, d2, d3, d4):
        s = self.shape_.shape
        s[0], s[1], s[2], s[3] = int(d1), int(d2), int(d3), int(d4)

    def SetShape4D_like(self, mat):
        self.SetShape4D(*mat.shape_.shape)

    def GetShape4D(self):
        return self.shape_

    def GetSlice(self, slice_, start, end):
        """Columns [start, end) as a view sharing memory (cudamat.cu:604-626)."""
        _chk(lib.get_slice(self.GetMat(), slice_.GetMat(), int(start), int(end)), "get_slice")
        rows = self.mat_.size[0]
        slice_._t = self._t[start * rows:end * rows] if self._t is not None else None
        slice_._host = None
        slice_.SetupTranspose()

    def Reshape(self, rows, cols):
        _chk(lib.reshape(self.GetMat(), int(rows), int(cols)), "reshape")
        self.SetupTranspose()

    def GetRows(self):
        return self.mat_.size[0]

    def GetCols(self):
        return self.mat_.size[1]

    def GetNumEls(self):
        return self.mat_.size[0] * self.mat_.size[1]

    def tensor(self):
        """The device storage as a flat torch tensor (for torch.distributed collectives only)."""
        return self._t[: self.GetNumEls()]

    # ---- host <-> device ----------------------------------------------------------------------------
    def GetHostData(self):
        """numpy array of shape (cols, rows): the column-major bytes of the matrix."""
        if self._host is None or self._host.size != self.GetNumEls():
            self._host = np.zeros(self.GetNumEls(), np.float32)
        h = self._host.reshape(self.mat_.size[1], self.mat_.size[0])
        return h

    def CopyToHost(self):
        h = self.GetHostData()
        self.mat_.data_host = h.ctypes.data_as(_lib.c_float_p)
        _chk(lib.copy_to_host(self.GetMat()), "copy_to_host")
        return h

    def CopyToDevice(self):
        h = self.GetHostData()
        self.mat_.data_host = h.ctypes.data_as(_lib.c_float_p)
        _chk(lib.copy_to_device(self.GetMat()), "copy_to_device")

    def FromNumpy(self, arr):
        """Convenience for tests/benchmarks: load column-major bytes from a numpy array."""
        a = np.ascontiguousarray(arr, np.float32).reshape(-1)
        assert a.size == self.GetNumEls(), (a.size, self.GetNumEls())
        self.GetHostData().reshape(-1)[:] = a
        self.CopyToDevice()

    def ToNumpy(self):
        return self.CopyToHost().copy()

    # ---- HDF5 (src/matrix.cc:419-435): a column-major (rows, cols) matrix is a row-major (cols, rows) dataset ----------
    def WriteHDF5(self, file, name):
        h = self.CopyToHost()
        file.WriteHDF5CPU(h, self.mat_.size[1], self.mat_.size[0], name)

    def ReadHDF5(self, file, name):
        self.GetHostData().reshape(-1)[:] = file.ReadHDF5CPU(self.GetNumEls(), name)
        self.CopyToDevice()

    def AllocateAndReadHDF5(self, file, name):
        rows, cols = file.ReadHDF5Shape(name)
        self.AllocateGPUMemory(rows, cols)
        self.ReadHDF5(file, name)

    def ReadValue(self, *idx):
        row, col = idx if len(idx) == 2 else (idx[0] % self.mat_.size[0], idx[0] // self.mat_.size[0])
        err = ctypes.c_int(0)
        v = lib.read_from(self.GetMat(), int(row), int(col), ctypes.byref(err))
        _chk(err.value, "Could not read value")
        return v

    def WriteValue(self, *args):
        if len(args) == 3:
            row, col, val = args
        else:
            row, col, val = args[0] % self.mat_.size[0], args[0] // self.mat_.size[0], args[1]
        _chk(lib.write_at(self.GetMat(), int(row), int(col), float(val)), "Could not write value")

    # ---- elementwise ----------------------------------------------------------------------------------
    def Set(self, val):
        if isinstance(val, Matrix):
            _chk(lib.copy_on_device(val.GetMat(), self.GetMat()), "Could not set to val")
        else:
            _chk(lib.assign_scalar(self.GetMat(), float(val)), "Could not set to scalar")

    def Add(self, m, alpha=None):
        if isinstance(m, Matrix):
            if alpha is None:
                _chk(lib.add_elementwise(self.GetMat(), m.GetMat(), self.GetMat()), "add")
            else:
                _chk(lib.add_mult(self.GetMat(), m.GetMat(), float(alpha)), "add_mult")
        else:
            _chk(lib.add_scalar(self.GetMat(), float(m), self.GetMat()), "add_scalar")

    def AddRowVec(self, v, alpha=None):
        if alpha is None:
            _chk(lib.add_row_vec(self.GetMat(), v.GetMat(), self.GetMat()), "add_row_vec")
        else:
            _chk(lib.add_row_mult(self.GetMat(), v.GetMat(), self.GetMat(), float(alpha)), "add_row_mult")

    # ---- input staging (src/matrix.cc: AddColVec, DivideByColVec, MultByRowVec, NormalizeColumnwise, ShuffleColumns,
    #      AddToEachPixel, ExtractPatches, CopyTranspose — the DataHandler's GPU-side calls) ------------------------
    def AddColVec(self, v, alpha=1.0):
        _chk(lib.add_col_mult(self.GetMat(), v.GetMat(), self.GetMat(), float(alpha)), "add_col_mult")

    def DivideByColVec(self, v):
        _chk(lib.div_by_col_vec(self.GetMat(), v.GetMat(), self.GetMat()), "div_by_col_vec")

    def MultByRowVec(self, v):
        _chk(lib.mult_by_row_vec(self.GetMat(), v.GetMat(), self.GetMat()), "mult_by_row_vec")

    def DivideByRowVec(self, v):
        _chk(lib.div_by_row_vec(self.GetMat(), v.GetMat(), self.GetMat()), "div_by_row_vec")

    def NormalizeColumnwise(self):
        _chk(lib.normalize_by_axis(self.GetMat(), self.GetMat(), 0), "normalize_by_axis")

    def ShuffleColumns(self, rand_perm_indices):
        _chk(lib.shuffleColumns(self.GetMat(), rand_perm_indices.GetMat()), "shuffleColumns")

    def AddToEachPixel(self, v, mult):
        _chk(lib.add_to_each_pixel(self.GetMat(), v.GetMat(), self.GetMat(), float(mult)), "add_to_each_pixel")

    @staticmethod
    def ExtractPatches(source, dest, width_offset, height_offset, flip_bit, image_size_y, image_size_x, patch_size_y, patch_size_x):
        # argument order of src/matrix.cc / CPUMatrix.cc:920-933: (y, x) sizes in, (x, y) passed to the kernel
        _chk(lib.extract_patches(source.GetMat(), dest.GetMat(), width_offset.GetMat(), height_offset.GetMat(), flip_bit.GetMat(),
                                 int(image_size_x), int(image_size_y), int(patch_size_x), int(patch_size_y)), "Error extracting patches")

    def Mult(self, val):
        if isinstance(val, Matrix):
            _chk(lib.mult_elementwise(self.GetMat(), val.GetMat(), self.GetMat(), 0.0), "mult")
        else:
            _chk(lib.mult_by_scalar(self.GetMat(), float(val), self.GetMat(), 0.0), "mult_by_scalar")

    def Divide(self, val):
        _chk(lib.divide_by_scalar(self.GetMat(), float(val), self.GetMat()), "divide_by_scalar")

    def Subtract(self, m, target):
        _chk(lib.subtract_elementwise(self.GetMat(), m.GetMat(), target.GetMat()), "subtract")

    def LowerBound(self, val):
        _chk(lib.lower_bound_scalar(self.GetMat(), float(val), self.GetMat()), "lower_bound_scalar")

    def UpperBoundMod(self, val):
        _chk(lib.upper_bound_mod_scalar(self.GetMat(), float(val), self.GetMat()), "upper_bound_mod_scalar")

    def Sqrt(self):
        _chk(lib.apply_sqrt(self.GetMat(), self.GetMat()), "sqrt")

    def ApplyDerivativeOfReLU(self, state):
        _chk(lib.apply_rectified_linear_deriv(self.GetMat(), state.GetMat(), self.GetMat()), "relu deriv")

    def ApplySoftmax(self):
        _chk(lib.softmax_row_major(self.GetMat(), self.GetMat()), "softmax")

    def Dropout(self, dropprob, fill_value, scale_factor):
        _chk(lib.dropout(ctypes.byref(Matrix._rnd), self.GetMat(), float(dropprob), float(fill_value), float(scale_factor)), "dropout")

    def ReluDropout(self, dropprob, scale_factor):
        """Fused LowerBound(0) + Dropout(p, 0, scale) (one pass; same arithmetic)."""
        _chk(lib.relu_dropout(ctypes.byref(Matrix._rnd), self.GetMat(), float(dropprob), float(scale_factor)), "relu_dropout")

    def FillWithRand(self):
        _chk(lib.fill_with_rand(ctypes.byref(Matrix._rnd), self.GetMat()), "Could not fill with rand")

    def FillWithRandn(self):
        _chk(lib.fill_with_randn(ctypes.byref(Matrix._rnd), self.GetMat()), "Could not fill with randn")

    def SampleBernoulli(self, val):
        self.Set(val)
        _chk(lib.sample_bernoulli(ctypes.byref(Matrix._rnd), self.GetMat(), self.GetMat()), "sample_bernoulli")

    # ---- reductions ------------------------------------------------------------------------------------
    def Sum(self):
        err = ctypes.c_int(0)
        v = lib.sum_all(self.GetMat(), ctypes.byref(err))
        _chk(err.value, "sum_all")
        return v

    def SumRows(self, target, alpha, beta):
        """target = alpha*target + beta*colsum(self)   (src/matrix.cc:755-758)"""
        _chk(lib.sum_by_axis(self.GetMat(), target.GetMat(), 0, float(beta), float(alpha)), "sum_by_axis")

    def SumCols(self, target, alpha, beta):
        _chk(lib.sum_by_axis(self.GetMat(), target.GetMat(), 1, float(beta), float(alpha)), "sum_by_axis")

    def SqSumAxis(self, target, axis, beta, alpha):
        _chk(lib.sqsum_by_axis(self.GetMat(), target.GetMat(), int(axis), float(beta), float(alpha)), "sqsum_by_axis")

    def NormLimitByAxis(self, axis, val, constraint):
        _chk(lib.normlimit_by_axis(self.GetMat(), self.GetMat(), int(axis), float(val), int(bool(constraint))), "normlimit_by_axis")

    def EuclidNorm(self):
        err = ctypes.c_int(0)
        v = lib.euclid_norm(self.GetMat(), ctypes.byref(err))
        _chk(err.value, "euclid_norm")
        return v

    def VDot(self, m):
        err = ctypes.c_int(0)
        v = lib.vdot(self.GetMat(), m.GetMat(), ctypes.byref(err))
        _chk(err.value, "vdot")
        return v

    def CopyTranspose(self, m):
        _chk(lib.copy_transpose(self.GetMat(), m.GetMat()), "copy_transpose")

    # ---- static operators (src/matrix.cc:678-989) ---------------------------------------------------
    @staticmethod
    def Dot(a, b, c, alpha, beta, transpose_a=False, transpose_b=False):
        """c = alpha*c + beta*op(a)*op(b) — note the reference's reversed naming (src/matrix.cc:678-689)."""
        am = a.GetMatTranspose() if transpose_a else a.GetMat()
        bm = b.GetMatTranspose() if transpose_b else b.GetMat()
        _chk(lib.dot(am, bm, c.GetMat(), float(alpha), float(beta)), "dot")

    @staticmethod
    def DotBiasAct(a, b, bias, c, alpha, beta, transpose_a=False, transpose_b=False, relu=False):
        am = a.GetMatTranspose() if transpose_a else a.GetMat()
        bm = b.GetMatTranspose() if transpose_b else b.GetMat()
        _chk(lib.dotBiasAct(am, bm, bias.GetMat() if bias is not None else None, c.GetMat(), float(alpha), float(beta), int(relu)), "dotBiasAct")

    @staticmethod
    def ConvUp(input, w, output, conv_desc, scale_targets):
        lib.convUpGemm(input.GetMat(), w.GetMat(), output.GetMat(), ctypes.byref(input.shape_), ctypes.byref(w.shape_),
                       ctypes.byref(output.shape_), conv_desc, float(scale_targets))

    @staticmethod
    def ConvUpBiasAct(input, w, bias, output, conv_desc, scale_targets, relu):
        lib.convUpBiasAct(input.GetMat(), w.GetMat(), bias.GetMat() if bias is not None else None, output.GetMat(),
                          ctypes.byref(input.shape_), ctypes.byref(w.shape_), ctypes.byref(output.shape_), conv_desc,
                          float(scale_targets), int(relu))

    @staticmethod
    def ConvDown(deriv_output, w, deriv_input, conv_desc, scale_targets):
        lib.convDownGemm(deriv_output.GetMat(), w.GetMat(), deriv_input.GetMat(), ctypes.byref(deriv_output.shape_),
                         ctypes.byref(w.shape_), ctypes.byref(deriv_input.shape_), conv_desc, float(scale_targets))

    @staticmethod
    def ConvDownMask(deriv_output, w, state, deriv_input, conv_desc, scale_targets, post_scale=1.0):
        """ConvDown with the source layer's ReLU' (and dropout' scale) fused into the epilogue."""
        lib.convDownMask(deriv_output.GetMat(), w.GetMat(), state.GetMat(), deriv_input.GetMat(), ctypes.byref(deriv_output.shape_),
                         ctypes.byref(w.shape_), ctypes.byref(deriv_input.shape_), conv_desc, float(scale_targets), float(post_scale))

    @staticmethod
    def DotMask(a, b, state, c, alpha, beta, post_scale=1.0):
        _chk(lib.dotMask(a.GetMat(), b.GetMat(), state.GetMat(), c.GetMat(), float(alpha), float(beta), float(post_scale)), "dotMask")

    @staticmethod
    def ConvOutpBias(input, deriv_output, dw, db, conv_desc, scale_targets, scale_outputs):
        """ConvOutp + the shared-bias gradient (two-step SumRows of conv_edge.cc:210-221) in one library call."""
        lib.convOutpBias(input.GetMat(), deriv_output.GetMat(), dw.GetMat(), db.GetMat(), ctypes.byref(input.shape_),
                         ctypes.byref(deriv_output.shape_), ctypes.byref(dw.shape_), conv_desc, float(scale_targets),
                         float(scale_outputs))

    @staticmethod
    def ConvMaxPoolUndoRelu(input, deriv_output, output, deriv_input, conv_desc, scale_targets):
        lib.MaxPoolUndoRelu(input.GetMat(), deriv_output.GetMat(), output.GetMat(), deriv_input.GetMat(),
                            ctypes.byref(input.shape_), ctypes.byref(deriv_output.shape_), conv_desc, float(scale_targets))

    @staticmethod
    def ConvOutp(input, deriv_output, dw, conv_desc, partial_sum_y, partial_sum_x, scale_targets, scale_outputs):
        lib.convOutpGemm(input.GetMat(), deriv_output.GetMat(), dw.GetMat(), ctypes.byref(input.shape_),
                         ctypes.byref(deriv_output.shape_), ctypes.byref(dw.shape_), conv_desc, float(scale_targets),
                         float(scale_outputs))

    @staticmethod
    def ConvMaxPool(input, output, conv_desc):
        lib.MaxPoolGemm(input.GetMat(), output.GetMat(), ctypes.byref(input.shape_), ctypes.byref(output.shape_), conv_desc, 0.0, 1.0)

    @staticmethod
    def ConvMaxPoolMask(input, output, mask, conv_desc):
        """MaxPool that also records the window masks (include/convnet_hip.h: MaxPoolMask).  False: this geometry has no mask kernel and
        nothing was computed — call ConvMaxPool."""
        if not hasattr(lib, "MaxPoolMask"):   # (an older build of the library under CONVNET_HIP_LIB)
            return False
        return lib.MaxPoolMask(input.GetMat(), output.GetMat(), mask.GetMat(), ctypes.byref(input.shape_), ctypes.byref(output.shape_), conv_desc) == 0

    @staticmethod
    def ConvMaxPoolUndoMask(deriv_output, mask, deriv_input, conv_desc, scale_targets, relu):
        _chk(lib.MaxPoolUndoMask(deriv_output.GetMat(), mask.GetMat(), deriv_input.GetMat(), ctypes.byref(deriv_input.shape_),
                                 ctypes.byref(deriv_output.shape_), conv_desc, float(scale_targets), int(bool(relu))), "MaxPoolUndoMask")

    @staticmethod
    def ConvMaxPoolUndo(input, deriv_output, output, deriv_input, conv_desc, scale_targets):
        lib.MaxPoolUndoGemm(input.GetMat(), deriv_output.GetMat(), output.GetMat(), deriv_input.GetMat(),
                            ctypes.byref(input.shape_), ctypes.byref(deriv_output.shape_), conv_desc, float(scale_targets))

    @staticmethod
    def ConvAvgPool(input, output, conv_desc):
        lib.AvgPoolGemm(input.GetMat(), output.GetMat(), ctypes.byref(input.shape_), ctypes.byref(output.shape_), conv_desc, 0.0, 1.0)

    @staticmethod
    def ConvAvgPoolUndo(input, deriv_output, conv_desc, scale_targets):
        # (avgGrads, targets): src/matrix.cc:944-954
        lib.AvgPoolUndoGemm(input.GetMat(), deriv_output.GetMat(), ctypes.byref(input.shape_), ctypes.byref(deriv_output.shape_),
                            conv_desc, float(scale_targets))

    @staticmethod
    def ConvResponseNormCrossMap(input, output, numFilters, sizeF, addScale, powScale, blocked, relu=False):
        fn = lib.ResponseNormCrossMapRelu if relu else lib.ResponseNormCrossMapGemm
        fn(input.GetMat(), output.GetMat(), int(numFilters), int(sizeF), float(addScale), float(powScale), bool(blocked))

    @staticmethod
    def ConvResponseNormCrossMapUndo(outGrads, inputs, acts, targets, numFilters, sizeF, addScale, powScale, blocked):
        lib.ResponseNormCrossMapUndoGemm(outGrads.GetMat(), inputs.GetMat(), targets.GetMat(), int(numFilters), int(sizeF),
                                         float(addScale), float(powScale), bool(blocked))

    @staticmethod
    def SoftmaxCEDeriv(state, gt, deriv):
        _chk(lib.apply_softmax_grad_row_major(state.GetMat(), gt.GetMat(), deriv.GetMat()), "softmax grad")

    @staticmethod
    def SoftmaxCorrect(state, gt, output):
        _chk(lib.get_softmax_correct_row_major(state.GetMat(), gt.GetMat(), output.GetMat()), "softmax correct")

    @staticmethod
    def SoftmaxCE(state, gt, output):
        _chk(lib.get_softmax_cross_entropy_row_major(state.GetMat(), gt.GetMat(), output.GetMat(), 1e-10), "softmax ce")

    @staticmethod
    def SoftmaxCEGradCorrect(logits, gt, probs, deriv, correct_accum, deriv_scale=1.0):
        _chk(lib.softmax_ce_grad_correct(logits.GetMat(), gt.GetMat(), probs.GetMat(), deriv.GetMat() if deriv is not None else None,
                                         correct_accum.GetMat() if correct_accum is not None else None, float(deriv_scale)), "softmax fused")

    @staticmethod
    def SGDMomentumStep(grad, param, history, l2_decay, gradient_clip, epsilon, momentum):
        _chk(lib.sgd_momentum_step(grad.GetMat(), param.GetMat(), history.GetMat(), float(l2_decay), float(gradient_clip),
                                   float(epsilon), float(momentum)), "sgd step")

    @staticmethod
    def SGDMomentumStepMulti(items):
        """items: (grad, param, history, l2_decay, gradient_clip, epsilon, momentum) per tensor — sgd_momentum_step on all of them in one
        launch per 16 (include/convnet_hip.h: sgd_momentum_step_multi); an older build of the library gets one call per tensor."""
        if not items:
            return
        if not hasattr(lib, "sgd_momentum_step_multi") or len(items) == 1:
            for it in items:
                Matrix.SGDMomentumStep(*it)
            return
        n = len(items)
        MP = ctypes.POINTER(_lib.cudamat)
        arr = lambda k: (MP * n)(*[ctypes.pointer(it[k].mat_) for it in items])   # noqa: E731
        flt = lambda k: (ctypes.c_float * n)(*[float(it[k]) for it in items])    # noqa: E731
        _chk(lib.sgd_momentum_step_multi(n, arr(0), arr(1), arr(2), flt(3), flt(4), flt(5), flt(6)), "sgd step (multi)")

    @staticmethod
    def SGDMomentumStepNormLimit(grad, param, history, l2_decay, gradient_clip, epsilon, momentum, norm, constraint):
        _chk(lib.sgd_momentum_step_normlimit(grad.GetMat(), param.GetMat(), history.GetMat(), float(l2_decay), float(gradient_clip),
                                             float(epsilon), float(momentum), float(norm), int(bool(constraint))), "sgd step + norm limit")

    # ---- temp / ones pools (src/matrix.cc:633-676) -------------------------------------------------
    @staticmethod
    def RegisterTempMemory(size, why=""):
        if size > Matrix._temp_size:
            Matrix._temp_size = int(size)

    @staticmethod
    def RegisterOnes(size):
        if size > Matrix._ones_size:
            Matrix._ones_size = int(size)

    @staticmethod
    def GetTemp(rows, cols, temp):
        size = int(rows) * int(cols)
        if Matrix._temp is None or Matrix._temp.GetNumEls() < max(size, Matrix._temp_size):
            Matrix._temp = Matrix()
            Matrix._temp.AllocateGPUMemory(1, max(size, Matrix._temp_size), "temp")
        Matrix._temp.Reshape(1, -1)
        Matrix._temp.GetSlice(temp, 0, size)
        temp.Reshape(rows, cols)

    @staticmethod
    def GetOnes(rows, cols, ones):
        size = int(rows) * int(cols)
        if Matrix._ones is None or Matrix._ones.GetNumEls() < max(size, Matrix._ones_size):
            Matrix._ones = Matrix()
            Matrix._ones.AllocateGPUMemory(1, max(size, Matrix._ones_size), "ones")
            Matrix._ones.Set(1.0)
        Matrix._ones.Reshape(1, -1)
        Matrix._ones.GetSlice(ones, 0, size)
        ones.Reshape(rows, cols)


def make_conv_desc(C, F, Ky, Kx, sy=1, sx=1, pady=0, padx=0):
    """ConvDesc from pbtxt-style (positive) paddings; stores the negation like src/edge.cc:97-99."""
    d = ConvDesc()
    d.num_input_channels, d.num_output_channels = C, F
    d.kernel_size_y, d.kernel_size_x, d.kernel_size_t = Ky, Kx, 1
    d.stride_y, d.stride_x, d.stride_t = sy, sx, 1
    d.padding_y, d.padding_x, d.padding_t = -pady, -padx, 0
    d.input_channel_begin, d.input_channel_end = 0, C
    d.output_channel_begin, d.output_channel_end = 0, F
    d.num_groups = 1
    return d
