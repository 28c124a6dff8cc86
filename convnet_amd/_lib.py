"""ctypes binding of libconvnet_hip.so — the same kind of binding the reference ships for its own
C ABI (cudamat/cudamat.py:9-135, cudamat/cudamat_conv_gemm.py:4-117).

The product path has NO CPU fallback: if the HIP library is missing, import raises."""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libconvnet_hip.so")
# A/B runs of tools/ against another BUILD of the same library (e.g. the previous round's, lib/libconvnet_hip_r03.so): entry points
# that build lacks are skipped.  Never set by the product, the tests or bench.py's measured legs.
_ALT_LIB = os.environ.get("CONVNET_HIP_LIB")
if _ALT_LIB:
    LIB_PATH = _ALT_LIB if os.path.isabs(_ALT_LIB) else os.path.join(_HERE, "lib", _ALT_LIB)
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "convnet_hip.h")

c_float_p = ctypes.POINTER(ctypes.c_float)


class cudamat(ctypes.Structure):
    """Mirror of ``struct cudamat`` (include/convnet_hip.h; reference cudamat/cudamat.py:127-135)."""
    _fields_ = [("data_host", c_float_p),
                ("data_device", ctypes.c_void_p),
                ("on_device", ctypes.c_int),
                ("on_host", ctypes.c_int),
                ("size", ctypes.c_int * 2),
                ("is_trans", ctypes.c_int),
                ("owns_data", ctypes.c_int),
                ("tex_obj", ctypes.c_ulonglong)]


class Shape4D(ctypes.Structure):
    _fields_ = [("shape", ctypes.c_int * 4)]


class ConvDesc(ctypes.Structure):
    """Mirror of ``struct ConvDesc`` (reference cudamat/cudamat.py:137-190)."""
    _fields_ = [(n, ctypes.c_int) for n in (
        "num_input_channels", "num_output_channels", "kernel_size_y", "kernel_size_x", "kernel_size_t",
        "stride_y", "stride_x", "stride_t", "padding_y", "padding_x", "padding_t",
        "input_channel_begin", "input_channel_end", "output_channel_begin", "output_channel_end", "num_groups")]

    def copy(self):
        c = ConvDesc()
        ctypes.memmove(ctypes.byref(c), ctypes.byref(self), ctypes.sizeof(ConvDesc))
        return c


class rnd_struct(ctypes.Structure):
    _fields_ = [("dev_mults", ctypes.c_void_p), ("dev_words", ctypes.c_void_p)]


class KernelInfo(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char_p), ("flops", ctypes.c_double), ("grid_blocks", ctypes.c_int), ("split_k", ctypes.c_int)]


def declared_symbols(header_path=HEADER_PATH):
    """Every function name declared in include/convnet_hip.h."""
    text = open(header_path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    body = text[text.index('extern "C" {'):]
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{]*\)\s*;", body)
    return sorted(set(n for n in names if n not in ("defined",)))


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the product path.")
    # torch must be imported BEFORE the dlopen: the library's libamdhip64 dependency then resolves (by
    # SONAME) to the HIP runtime torch already loaded, so both share ONE runtime — required, because torch
    # streams and device pointers are handed straight to the library.  Loading ours first pulls in
    # /opt/rocm's copy and hipSetDevice later fails with two runtimes in the process.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    P, F, I = ctypes.POINTER, ctypes.c_float, ctypes.c_int
    M, S = P(cudamat), P(Shape4D)

    def sig(name, res, *args):
        if _ALT_LIB and not hasattr(lib, name):
            return
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = list(args)

    sig("convnet_hip_init", I, I)
    sig("convnet_hip_shutdown", None)
    sig("convnet_hip_set_stream", None, ctypes.c_void_p)
    sig("convnet_hip_get_stream", ctypes.c_void_p)
    sig("convnet_hip_reserve_workspace", I, ctypes.c_size_t)
    sig("convnet_hip_version", ctypes.c_char_p)
    sig("convnet_hip_set_matrix_path", None, I)
    sig("convnet_hip_get_matrix_path", I)
    sig("convnet_hip_set_patch_mode", None, I)
    sig("convnet_hip_get_patch_mode", I)
    sig("convnet_hip_set_deferred_epilogues", None, I)
    sig("convnet_hip_get_deferred_epilogues", I)
    sig("convnet_hip_deferred_absorbed", ctypes.c_long)
    sig("convnet_hip_set_wgrad_tile", None, I)
    sig("convnet_hip_get_wgrad_tile", I)
    sig("get_last_cuda_error", ctypes.c_char_p)
    sig("cuda_set_device", I, I)
    sig("cuda_sync_threads", None)
    for ev_fn in ("cuda_create_event", "cuda_record_event", "cuda_synchronize_event"):
        sig(ev_fn, I, P(ctypes.c_void_p))
    sig("cublas_init", I)
    sig("cublas_shutdown", I)
    sig("destroy_tex", I, M)
    sig("convnet_hip_last_kernel_info", None, P(KernelInfo))
    sig("convnet_hip_profile_enable", None, I)
    sig("convnet_hip_profile_report", ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t)
    sig("convnet_hip_probe_matrix_pipe", I, I, ctypes.c_double, ctypes.POINTER(ctypes.c_double))
    # data-parallel exchange (csrc/comm.hip)
    sig("convnet_hip_comm_unique_id", I, ctypes.c_char_p)
    sig("convnet_hip_comm_init", I, I, I, ctypes.c_char_p)
    sig("convnet_hip_comm_rank", I)
    sig("convnet_hip_comm_size", I)
    sig("convnet_hip_comm_max_slots", I)
    sig("convnet_hip_comm_broadcast", I, P(cudamat), I)
    sig("convnet_hip_comm_allreduce_avg", I, P(cudamat), ctypes.c_size_t, ctypes.c_size_t, I)
    sig("convnet_hip_comm_wait", I, I)
    sig("convnet_hip_comm_sync", I)
    sig("convnet_hip_comm_destroy", I)
    for n in ("allocate_device_memory", "free_device_memory", "copy_to_host", "copy_to_device"):
        sig(n, I, M)
    sig("copy_to_host_slice", I, M, ctypes.c_size_t, ctypes.c_size_t)
    sig("copy_to_device_slice", I, M, ctypes.c_size_t, ctypes.c_size_t)
    sig("copy_on_device", I, M, M)
    sig("copy_transpose", I, M, M)
    sig("reshape", I, M, I, I)
    sig("get_slice", I, M, M, ctypes.c_uint, ctypes.c_uint)
    sig("init_from_array", None, M, c_float_p, I, I)
    sig("init_empty", I, M, I, I)
    sig("write_at", I, M, I, I, F)
    sig("read_from", F, M, I, I, P(I))
    sig("extract_patches", I, M, M, M, M, M, I, I, I, I)
    sig("shuffleColumns", I, M, M)
    for n in ("add_col_vec", "div_by_col_vec", "mult_by_row_vec", "div_by_row_vec"):
        sig(n, I, M, M, M)
    sig("add_col_mult", I, M, M, M, F)
    sig("add_to_each_pixel", I, M, M, M, F)
    sig("normalize_by_axis", I, M, M, I)
    for n in ("convUpGemm", "convDownGemm", "convUp", "convDown"):
        sig(n, None, M, M, M, S, S, S, ConvDesc, F)
    sig("convOutpGemm", None, M, M, M, S, S, S, ConvDesc, F, F)
    sig("convOutp", None, M, M, M, S, S, S, ConvDesc, I, I, F, F)
    sig("convUpBiasAct", None, M, M, M, M, S, S, S, ConvDesc, F, I)
    sig("convDownMask", None, M, M, M, M, S, S, S, ConvDesc, F, F)
    sig("dotMask", I, M, M, M, M, F, F, F)
    sig("MaxPoolUndoRelu", None, M, M, M, M, S, S, ConvDesc, F)
    sig("MaxPoolMask", I, M, M, M, S, S, ConvDesc)
    sig("MaxPoolUndoMask", I, M, M, M, S, S, ConvDesc, F, I)
    sig("convOutpBias", None, M, M, M, M, S, S, S, ConvDesc, F, F)
    for n in ("MaxPoolGemm", "AvgPoolGemm"):
        sig(n, None, M, M, S, S, ConvDesc, F, F)
    for n in ("MaxPool", "AvgPool"):
        sig(n, None, M, M, S, S, ConvDesc)
    for n in ("MaxPoolUndoGemm", "MaxPoolUndo"):
        sig(n, None, M, M, M, M, S, S, ConvDesc, F)
    for n in ("AvgPoolUndoGemm", "AvgPoolUndo"):
        sig(n, None, M, M, S, S, ConvDesc, F)
    for n in ("ResponseNormCrossMapGemm", "ResponseNormCrossMap", "ResponseNormCrossMapRelu"):
        sig(n, None, M, M, I, I, F, F, ctypes.c_bool)
    sig("ResponseNormCrossMapUndoGemm", None, M, M, M, I, I, F, F, ctypes.c_bool)
    sig("ResponseNormCrossMapUndo", None, M, M, M, M, I, I, F, F, ctypes.c_bool)
    sig("dot", I, M, M, M, F, F)
    sig("dotBiasAct", I, M, M, M, M, F, F, I)
    sig("vdot", F, M, M, P(I))
    sig("add_row_vec", I, M, M, M)
    sig("add_row_mult", I, M, M, M, F)
    sig("sum_by_axis", I, M, M, I, F, F)
    sig("sqsum_by_axis", I, M, M, I, F, F)
    sig("sum_all", F, M, P(I))
    sig("euclid_norm", F, M, P(I))
    sig("normlimit_by_axis", I, M, M, I, F, I)
    sig("lower_bound_scalar", I, M, F, M)
    sig("upper_bound_mod_scalar", I, M, F, M)
    sig("apply_rectified_linear_deriv", I, M, M, M)
    sig("assign_scalar", I, M, F)
    sig("add_scalar", I, M, F, M)
    sig("mult_by_scalar", I, M, F, M, F)
    sig("divide_by_scalar", I, M, F, M)
    sig("add_mult", I, M, M, F)
    sig("add_elementwise", I, M, M, M)
    sig("subtract_elementwise", I, M, M, M)
    sig("mult_elementwise", I, M, M, M, F)
    sig("apply_sqrt", I, M, M)
    sig("softmax_row_major", I, M, M)
    sig("softmax_row_major_multi", I, M, I, M)
    sig("apply_softmax_grad_row_major", I, M, M, M)
    sig("get_softmax_correct_row_major", I, M, M, M)
    sig("get_softmax_cross_entropy_row_major", I, M, M, M, F)
    sig("softmax_ce_grad_correct", I, M, M, M, M, M, F)
    sig("sgd_momentum_step", I, M, M, M, F, F, F, F)
    sig("sgd_momentum_step_normlimit", I, M, M, M, F, F, F, F, F, I)
    sig("sgd_momentum_step_multi", I, I, P(M), P(M), P(M), c_float_p, c_float_p, c_float_p, c_float_p)
    R = P(rnd_struct)
    sig("init_random", I, R, I)
    sig("fill_with_rand", I, R, M)
    sig("fill_with_randn", I, R, M)
    sig("sample_bernoulli", I, R, M, M)
    sig("dropout", I, R, M, F, F, F)
    sig("relu_dropout", I, R, M, F, F)
    return lib


lib = _load()

_ERRORS = {  # reference src/util.cc:226-246 GetStringError
    -1: "Incompatible matrix dimensions.", -2: "CUBLAS error.", -3: "CUDA error: ", -4: "Operation not supported on views.",
    -5: "Operation not supported on transposed matrices.", -6: "Generic error.",
    -7: "Incompatible transposedness.", -8: "Matrix is not in device memory.", -9: "Operation not supported."}


def profile_enable(on=True):
    lib.convnet_hip_profile_enable(1 if on else 0)


def profile_report():
    """[{kernel, op, launches, ms, flops, bytes, executed}] for launches since the last report (flops = algorithmic work,
    executed = MFMA work issued, which is larger for dgrad gathers that run border taps on the zero page)."""
    buf = ctypes.create_string_buffer(1 << 16)
    n = lib.convnet_hip_profile_report(buf, len(buf))
    rows = []
    for line in buf.value.decode().splitlines() if n else []:
        k, op, cnt, ms, fl, by, ex = line.split("|")
        rows.append({"kernel": k, "op": op, "launches": int(cnt), "ms": float(ms), "flops": float(fl), "bytes": float(by), "executed": float(ex)})
    return rows


def GetStringError(err_code):
    msg = _ERRORS.get(err_code, "Unknown error")
    if err_code == -3:
        msg += lib.get_last_cuda_error().decode()
    return msg


def probe_matrix_pipe(random_operands=True, seconds=0.05):
    """{bf16_tflops, tflops_eq, ghz_counter, ghz_issue}: the sustained rate of a pure v_mfma_f32_32x32x16_bf16 stream on this chip
    (csrc/probe.hip: one wave per SIMD, register operands, nothing else issued) — the power-limited ceiling of the bf16-split kernels."""
    out = (ctypes.c_double * 4)()
    rc = lib.convnet_hip_probe_matrix_pipe(1 if random_operands else 0, float(seconds), out)
    if rc != 0:
        raise RuntimeError(f"convnet_hip_probe_matrix_pipe failed: {rc}")
    return {"bf16_tflops": out[0], "tflops_eq": out[1], "ghz_counter": out[2], "ghz_issue": out[3]}
