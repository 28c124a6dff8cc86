"""Builds convnet_amd/lib/libconvnet_hip.so (the C-ABI library declared in include/convnet_hip.h)
from convnet_amd/csrc/*.hip with hipcc for gfx950.  hipcc cross-compiles without a GPU; the .so
stays in-tree (git-ignored) so it travels to the GPU box with the snapshot."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "lib", "libconvnet_hip.so")
SOURCES = ["state.hip", "gather_gemm.hip", "patch_gemm.hip", "wgrad_wide.hip", "fewc_conv.hip", "pool_norm.hip", "elementwise.hip", "input_staging.hip", "comm.hip", "rccl_abi_check.hip", "probe.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-comment", "-Wno-inline-asm"]
# Per-file additions.  patch_gemm.hip: the SLP vectoriser packs the split's fp32 subtractions into v_pk_add_f32 (+ the v_mov / s_nop
# that feed them) — fewer instructions on paper, but beside MFMAs a packed fp32 op costs more issue time than the two plain ones
# (MI355X_MICROARCH.md), and gpw_kernel's one wave per SIMD has nobody to hide it: 177 instead of 191 VALU + 10 s_nop per chunk.
FILE_FLAGS = {"patch_gemm.hip": ["-fno-slp-vectorize"], "wgrad_wide.hip": ["-fno-slp-vectorize"], "fewc_conv.hip": ["-fno-slp-vectorize"]}
# CONVNET_BUILD_NOSLP=1: the same for gather_gemm.hip (ggp_kernel's consumers and wg_kernel run the same split) — an A/B for the next
# round with hardware, not the product build: the default kernels were measured and parity-tested as compiled without it.
if os.environ.get("CONVNET_BUILD_NOSLP"):
    FILE_FLAGS["gather_gemm.hip"] = ["-fno-slp-vectorize"]
# CONVNET_BUILD_DIAG=1: compile the experiment knobs of csrc/common.h (CHIP_DIAG_KNOB) into the library.  Never set for the product.
if os.environ.get("CONVNET_BUILD_DIAG"):
    FLAGS.append("-DCONVNET_DIAG")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(os.path.join(HERE, "lib"), exist_ok=True)
    objdir = os.path.join(HERE, "lib", "obj")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(SRC, "common.h"), os.path.join(SRC, "gather_gemm.h"), os.path.join(SRC, "wgrad_wide_schedule.h"),
               os.path.join(HERE, "..", "include", "convnet_hip.h")]
    objs, procs = [], []
    for s in SOURCES:
        src = os.path.join(SRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            cmd = ["hipcc", *FLAGS, *FILE_FLAGS.get(s, []), "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            print(f"--- {s} failed ---\n{out}", file=sys.stderr)
        elif verbose and out.strip():
            print(out)
    if failed:
        raise RuntimeError("hipcc failed")
    if force or procs or _stale(OUT, objs):
        cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *objs, "-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            print(r.stdout, r.stderr, file=sys.stderr)
            raise RuntimeError("link failed")
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
