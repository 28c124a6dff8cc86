"""Times the REFERENCE'S OWN training loop on this library: the reference's unmodified ConvNet::TrainOneBatch (src/convnet.cc)
over its unmodified Layer / Edge / SGDOptimizer / Matrix (src/*.cc), linked to convnet_amd/lib/libconvnet_hip.so
(oracle/_ref/libref_host_hip.so, `make -C oracle host`; test infrastructure, see oracle/seam/seam_host.cc) — i.e. what a
maintainer of the reference gets by swapping libcudamat/libcudamat_conv for this library and changing nothing else: the
unfused cudamat call sequence, one performance-metric read-back per step.  `bench.py` is the product number (this repo's host,
fused entries, no per-step sync); this is the drop-in number beside it.

    python tools/ref_host_bench.py [--model alexnet|alexnet_nin|vgg16] [--batch 256] [--steps 20] [--warmup 5] [--cpu]

--cpu runs the same loop on the reference's CPU path (libref_host_cpu.so) with a small step count."""
import argparse
import ctypes
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="alexnet")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--cpu", action="store_true")
    ap.add_argument("--dp", action="store_true", help="the data-parallel subclass of the reference host (oracle/seam/seam_host.cc SeamDPNet) as a 1-rank world: every gradient bucket posted through convnet_hip_comm_*")
    a = ap.parse_args()

    import ref_host
    from convnet_amd import models
    if a.cpu:
        so = ref_host.CPU_SO
    else:
        import torch
        assert torch.cuda.is_available()
        from convnet_amd import _lib
        ctypes.CDLL(_lib.LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        so = ref_host.HIP_SO
    host = ref_host.RefHost(so)
    host.lib.seam_host_bench.restype = ctypes.c_double
    host.lib.seam_host_bench.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    text = getattr(models, a.model)()
    loss = ctypes.c_float()
    with tempfile.TemporaryDirectory() as tmp:
        m, d = ref_host.write_configs(tmp, text, a.batch, 2, 11, a.model)
        # the reference prints its layer table on stdout: keep stdout for the JSON line only
        out_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if a.dp:
                host.lib.seam_host_bench_dp.restype = ctypes.c_double
                host.lib.seam_host_bench_dp.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_long, ctypes.c_void_p]
                ms = host.lib.seam_host_bench_dp(m.encode(), d.encode(), a.warmup, a.steps, 8 << 20, ctypes.byref(loss))
                assert ms > 0, f"seam_host_bench_dp failed: {ms}"
            else:
                ms = host.lib.seam_host_bench(m.encode(), d.encode(), a.warmup, a.steps, ctypes.byref(loss))
        finally:
            sys.stdout.flush()
            os.dup2(out_fd, 1)
    line = {"metric": "images_per_sec", "host": "reference src/*.cc unmodified (oracle/_ref/libref_host_%s.so)" % ("cpu" if a.cpu else "hip"),
            "value": a.batch * 1e3 / ms, "unit": "images/s", "ms_per_step": ms, "steps": a.steps, "warmup": a.warmup,
            "config": {"workload": f"{a.model} bs={a.batch} fp32, ConvNet::TrainOneBatch", "batch": a.batch}, "last_loss": loss.value}
    print(json.dumps(line))


if __name__ == "__main__":
    main()
