"""CPU: libconvnet_hip.so loads and exports every symbol include/convnet_hip.h declares, and the
ctypes struct mirrors have the reference's layout.  No compute calls (no GPU here)."""
import ctypes

from convnet_amd import _lib


def test_every_declared_symbol_is_exported():
    names = _lib.declared_symbols()
    assert len(names) >= 75, names
    missing = [n for n in names if not hasattr(_lib.lib, n)]
    assert not missing, missing


def test_struct_layouts_match_reference_abi():
    # struct cudamat: 2 pointers + 6 ints + 64-bit slot = 48 bytes (cudamat/cudamat.cuh:28-37)
    assert ctypes.sizeof(_lib.cudamat) == 48
    assert _lib.cudamat.size.offset == 24 and _lib.cudamat.is_trans.offset == 32 and _lib.cudamat.tex_obj.offset == 40
    assert ctypes.sizeof(_lib.ConvDesc) == 64 and ctypes.sizeof(_lib.Shape4D) == 16
    assert _lib.ConvDesc.padding_y.offset == 32 and _lib.ConvDesc.num_groups.offset == 60


def test_version_and_error_strings():
    assert b"gfx950" in _lib.lib.convnet_hip_version()
    assert "dimensions" in _lib.GetStringError(-1)


def test_host_only_view_ops():
    # reshape / get_slice are pure host bookkeeping (cudamat.cu:587-626): callable without a GPU
    m = _lib.cudamat()
    m.size[0], m.size[1], m.on_device, m.data_device = 6, 4, 1, 4096
    assert _lib.lib.reshape(ctypes.byref(m), -1, 8) == 0 and (m.size[0], m.size[1]) == (3, 8)
    assert _lib.lib.reshape(ctypes.byref(m), 5, -1) == -1
    s = _lib.cudamat()
    assert _lib.lib.get_slice(ctypes.byref(m), ctypes.byref(s), 2, 5) == 0
    assert (s.size[0], s.size[1], s.owns_data) == (3, 3, 0) and s.data_device == 4096 + 2 * 3 * 4
    assert _lib.lib.get_slice(ctypes.byref(m), ctypes.byref(s), 5, 9) == -1


def test_library_default_matrix_path_is_ieee_fp32():
    """The C ABI's default is path 0 (v_mfma_f32_32x32x2_f32: exact fp32 products, inf / NaN as cublasSgemm gives them) — the
    bf16-split products are something a host selects (convnet_hip_set_matrix_path(1) or CONVNET_GG_SPLIT=1), as Matrix.SetupCUDADevice
    does for the Python trainer, the tests and bench.py.  No device call is made here."""
    import os
    import subprocess
    import sys
    code = "from convnet_amd import _lib; print(_lib.lib.convnet_hip_get_matrix_path(), _lib.lib.convnet_hip_get_patch_mode())"
    env = {k: v for k, v in os.environ.items() if k not in ("CONVNET_GG_SPLIT", "CONVNET_GG_PATCH")}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=root, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.split() == ["0", "3"], out.stdout   # (the patch mode only matters on path 1: gpw_kernel where its launch policy applies)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=root, env=dict(env, CONVNET_GG_SPLIT="1"), timeout=300)
    assert out.stdout.split()[0] == "1", out.stdout
