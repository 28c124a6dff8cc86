"""The library's GEMM-shaped kernels executed FUNCTIONALLY on the CPU: the real kernel sources (convnet_amd/csrc/gather_gemm.hip,
patch_gemm.hip, wgrad_wide.hip) compiled as host C++ against tests/emu/hip/hip_runtime.h (every thread of a block a fiber; wave
collectives, the MFMAs in the register layouts the kernels assume, LDS-DMA as an immediate copy) and run on small problems against a
double-precision reference (tests/emu/emu_main.cc).
* The default kernels, through the C ABI (convUp / convDown / convOutpBias / dot: ggp_kernel incl. its generic-k mode and the
  stride-class table, gg_kernel, wg_kernel in both tile sizes with the bias row, the slab reduces).  They are green on hardware: here
  they are the calibration of the harness, and a functional check of the product kernels that needs no GPU.
* gpw_kernel and wgw_kernel (and their variants) — written after the last hardware run of their round — through their own launchers.
* The HBM-bound kernels of the path through the C ABI against the CPU oracle (oracle/liboracle.so): max pooling and its undo (both
  undo kernels), cross-map response normalisation and its undo, the fused SGD step.
No GPU; not a product path.  What this cannot see: timing, late-landing loads (tests/test_patch_wide_cpu.py / test_wgrad_wide_cpu.py model
those), the M0 range above 84 KB, instruction hazards.  CONVNET_EMU_ALL=1 runs every case (~5 minutes) instead of a subset (~1.5)."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _clang():
    for c in ("/opt/rocm/lib/llvm/bin/clang++", shutil.which("clang++")):
        if c and os.path.exists(c):
            return c
    return None


SOURCES = ["gather_gemm.hip", "patch_gemm.hip", "wgrad_wide.hip", "fewc_conv.hip", "pool_norm.hip", "elementwise.hip"]


@pytest.fixture(scope="module")
def emu_binary(tmp_path_factory):
    cc = _clang()
    if not cc:
        pytest.skip("no clang++ (the kernels use clang's vector extensions and __bf16)")
    oracle_dir = os.path.join(ROOT, "oracle")
    if not os.path.exists(os.path.join(oracle_dir, "liboracle.so")):
        pytest.skip("oracle/liboracle.so is not built (python -c 'import __graft_entry__ as g; g.build()')")
    d = tmp_path_factory.mktemp("emu")
    flags = ["-std=c++17", "-O1", "-x", "c++", "-I", os.path.join(HERE, "emu"), "-I", os.path.join(ROOT, "convnet_amd", "csrc"), "-Wno-everything"]
    jobs = [(os.path.join(ROOT, "convnet_amd", "csrc", f), str(d / (f + ".o"))) for f in SOURCES] + [(os.path.join(HERE, "emu", "emu_main.cc"), str(d / "emu_main.o"))]
    procs = [subprocess.Popen([cc, *flags, "-c", src, "-o", obj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for src, obj in jobs]
    for p, (src, _) in zip(procs, jobs):
        out, _ = p.communicate()
        assert p.returncode == 0, src + "\n" + out
    exe = d / "emu_main"
    subprocess.run([cc, *[o for _, o in jobs], "-o", str(exe), "-L", oracle_dir, "-loracle", "-Wl,-rpath," + oracle_dir], check=True)
    return str(exe)


def test_wide_kernels_run_correctly_in_emulation(emu_binary):
    which = "all" if os.environ.get("CONVNET_EMU_ALL") else "quick"
    r = subprocess.run([emu_binary, which], capture_output=True, text=True, timeout=1200)
    lines = r.stdout.strip().splitlines()
    print(r.stdout)
    assert r.returncode == 0 and lines and lines[-1] == "ALL PASSED", r.stdout + r.stderr
    # the calibration cases and both new kernels really ran
    assert any(l.startswith("PASS abi convUp") for l in lines) and any(l.startswith("PASS abi convOutpBias") for l in lines)
    assert any(l.startswith("PASS gpp(raw)") for l in lines) and any(l.startswith("PASS gpw fprop") for l in lines)
    assert any(l.startswith("PASS gpw dgrad") for l in lines) and any(l.startswith("PASS wgw wgrad") for l in lines)
    # gpv_kernel: conv2's forward form and its four stride classes (the latter through convDown)
    assert any(l.startswith("PASS gpv fprop") for l in lines) and any(l.startswith("PASS abi convDown") and "k5 s2" in l and "gpw_kernel(dgrad)" in l for l in lines)


def test_wide_wgrad_kernel_single_block_epilogue_in_emulation(emu_binary):
    """one block per tile (forced: the launch policy splits the reduction at emulation sizes): scaleTargets, scaleOutput and the bias
    row in the kernel's own epilogue, through the 16-byte write-out and through the direct one (F % 4 != 0)"""
    r = subprocess.run([emu_binary, "wgwfin"], capture_output=True, text=True, timeout=1200)
    lines = r.stdout.strip().splitlines()
    assert r.returncode == 0 and lines[-1] == "ALL PASSED" and sum("splits=1 " in l and l.startswith("PASS wgw") for l in lines) == 2, r.stdout + r.stderr


def test_wide_patch_kernel_tail_split_in_emulation(emu_binary):
    """11 tiles on an 8-slot "chip": the last round's three tiles are cut into three K-ranges (the kernel's tail-split branch, its raw
    partial tiles, gpw_tail_fix_kernel) — forced by the harness, the cost model never picks it at emulation sizes"""
    r = subprocess.run([emu_binary, "gpwtail"], capture_output=True, text=True, timeout=1200)
    lines = r.stdout.strip().splitlines()
    assert r.returncode == 0 and lines[-1] == "ALL PASSED" and "tail_splits=3" in lines[0], r.stdout + r.stderr


def test_group_patch_kernel_tail_split_in_emulation(emu_binary):
    """gpv_kernel (5 x 5 stride 2: superchunks of three and two chunks): 13 tiles on an 8-slot "chip", the last round's five tiles cut
    into three K-ranges whose borders fall inside a tap row's groups"""
    r = subprocess.run([emu_binary, "gpvtail"], capture_output=True, text=True, timeout=1200)
    lines = r.stdout.strip().splitlines()
    assert r.returncode == 0 and lines[-1] == "ALL PASSED" and "tail_splits=3" in lines[0], r.stdout + r.stderr
