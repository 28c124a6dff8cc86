"""Generates tests/golden/ref_host_alexnet224.npz: the REFERENCE'S OWN HOST on the reference's CPU path
(oracle/_ref/libref_host_cpu.so = src/convnet.cc, layer.cc, every *_edge.cc, loss_functions.cc, CPUMatrix.cc, eigenmat, compiled
unmodified; `make -C oracle host`, needs /root/reference) running ONE Fprop(train)/ComputeDeriv/Bprop of the full 224x224
AlexNet (models.alexnet(dropprob=0) == examples/imagenet/CLS_net_20140621074703.pbtxt without dropout: the CPU and GPU RNGs
cannot be matched) at N = 4 on hash-generated parameters (ref_host.golden_params) and the data shim's hash batch 0.

The 62 M-float gradient is not committed; per edge the file holds 4096 samples at seeded indices (`idx_<edge>`, `g_<edge>`),
the slice's L2 norm (`norm_<edge>`) and mean |g| (`absmean_<edge>`); `loss` is the CE loss the host read at those parameters.  Checked on the GPU by
tests/test_full_geometry_gpu.py::test_reference_cpu_host_gradient_on_the_full_alexnet_golden.

Run from the repo root:  python tests/golden/make_ref_host_alexnet_golden.py   (about a minute on 8 cores)"""
import os
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import ref_host  # noqa: E402
from convnet_amd import models  # noqa: E402

BATCH, SEED, SAMPLES = 4, 11, 4096

if __name__ == "__main__":
    host = ref_host.RefHost(ref_host.CPU_SO)
    text = models.alexnet(dropprob=0.0)
    with tempfile.TemporaryDirectory() as tmp:
        m, d = ref_host.write_configs(tmp, text, BATCH, 1, SEED)
        layers, edges, total = host.describe(m, d)
        slices, end = ref_host.slices_from_describe(layers, edges)
        assert end == total, (end, total)
        p0 = ref_host.golden_params(total, SEED, slices)
        t0 = time.time()
        g0 = host.gradient(m, d, p0)
        print(f"reference CPU host forward+backward: {time.time() - t0:.1f} s")
        _, _, loss = host.train(m, d, 1, p0)   # Layer::GetLoss on the state of the step's own Fprop, i.e. the loss AT p0
    out = {"cfg": np.array([BATCH, SEED], np.int32), "total": np.int64(total), "loss": np.float64(loss[0])}
    print("loss at p0:", loss[0])
    rng = np.random.default_rng(1234)
    named = [(f"{s}__{t}", n) for s, t, n in edges if n]
    for (name, n), (off, n2, _) in zip(named, slices):
        assert n == n2
        idx = np.sort(rng.choice(n, size=min(SAMPLES, n), replace=False)).astype(np.int64)
        g = g0[off:off + n]
        out[f"idx_{name}"] = idx
        out[f"g_{name}"] = g[idx]
        out[f"norm_{name}"] = np.float64(np.linalg.norm(g.astype(np.float64)))
        out[f"absmean_{name}"] = np.float64(np.abs(g).mean())
        print(f"{name:24s} {n:10d} floats  |g| = {out[f'norm_{name}']:.6g}  mean|g| = {out[f'absmean_{name}']:.3g}")
    path = os.path.join(HERE, "ref_host_alexnet224.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")
