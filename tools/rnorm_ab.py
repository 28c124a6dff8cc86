#!/usr/bin/env python
"""Response norm forward / undo on a set of shapes, results saved to --out (npz).  Run twice — CONVNET_RNORM_FAST=1 (the pipelined kernels,
default) and =0 (one tile per block) — and compare with --cmp a.npz b.npz: the two forms must agree bit for bit (same lanes, same channel
segments, same arithmetic order)."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = [(256, 96, 55, 24, False), (256, 256, 13, 64, False), (32, 100, 9, 24, False), (16, 250, 7, 64, False), (8, 64, 6, 64, False), (256, 96, 55, 5, False), (6, 40, 7, 5, False), (32, 64, 10, 3, False), (12, 500, 5, 5, False),
          (64, 96, 9, 5, True), (20, 130, 6, 4, False), (4, 700, 3, 5, False), (128, 192, 12, 5, False), (3, 17, 5, 9, False)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out")
    ap.add_argument("--cmp", nargs=2)
    args = ap.parse_args()
    if args.cmp:
        a, b = np.load(args.cmp[0]), np.load(args.cmp[1])
        bad = [k for k in a.files if not np.array_equal(a[k], b[k])]
        for k in a.files:
            print(k, "identical" if k not in bad else f"DIFFERENT max|d| {np.abs(a[k] - b[k]).max():.3e}")
        sys.exit(1 if bad else 0)
    import torch
    from convnet_amd.matrix import Matrix
    Matrix.SetupCUDADevice(0)
    out = {}
    for i, (N, C, H, sz, blocked) in enumerate(SHAPES):
        g = torch.Generator(device="cuda").manual_seed(100 + i)

        def mat():
            m = Matrix()
            m.AllocateGPUMemory(N, H * H * C)
            m._t.copy_(torch.randn(m._t.shape, generator=g, device="cuda"))
            m.SetShape4D(N, H, H, C)
            return m
        x, y, dy, dx = mat(), mat(), mat(), mat()
        Matrix.ConvResponseNormCrossMap(x, y, C, sz, 1e-4, 0.75, blocked)
        Matrix.ConvResponseNormCrossMapUndo(dy, x, y, dx, C, sz, 1e-4, 0.75, blocked)
        torch.cuda.synchronize()
        out[f"fwd{i}_N{N}C{C}H{H}s{sz}b{int(blocked)}"] = y.ToNumpy()
        out[f"undo{i}_N{N}C{C}H{H}s{sz}b{int(blocked)}"] = dx.ToNumpy()
    np.savez(args.out, **out)


if __name__ == "__main__":
    main()
