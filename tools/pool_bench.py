#!/usr/bin/env python
"""HBM-bound kernels (pool fwd/undo, response norm fwd/undo) at AlexNet shapes: us/launch and effective GB/s of
algorithmic traffic.  Usage: python tools/pool_bench.py [--n 256] [--reps 5]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from convnet_amd import _lib  # noqa: E402
from convnet_amd.matrix import Matrix, make_conv_desc  # noqa: E402

POOLS = {"pool1": (96, 110, 3, 2, 1), "pool2": (256, 26, 3, 2, 1), "pool5": (256, 11, 3, 2, 1), "p2x2": (96, 110, 2, 2, 0)}
NORMS = {"rnorm1": (96, 55, 24), "rnorm2": (256, 13, 64)}


def mat(N, C, H):
    m = Matrix()
    m.AllocateGPUMemory(N, H * H * C)
    m._t.normal_(0, 1)
    m.SetShape4D(N, H, H, C)
    return m


def timed(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    for _ in range(reps):
        fn()
    _lib.profile_enable(False)
    return _lib.profile_report()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    Matrix.SetupCUDADevice(0)
    N = args.n
    print(f"{'case':8s} {'kernel':34s} {'us':>8s} {'GB/s':>8s}")
    for name, (C, H, K, s, p) in POOLS.items():
        M = (H + 2 * p - K) // s + 1
        d = make_conv_desc(C, C, K, K, s, s, p, p)
        x, dx, y, dy = mat(N, C, H), mat(N, C, H), mat(N, C, M), mat(N, C, M)
        big, small = 4.0 * N * C * H * H, 4.0 * N * C * M * M
        mk = Matrix()
        mk.AllocateGPUMemory(N, (M * M * C + 1) // 2)
        masked = K == 3 and s == 2
        for tag, fn, nbytes in (("fwd", lambda: Matrix.ConvMaxPool(x, y, d), big + small),
                                *(((("fwd+mask", lambda: Matrix.ConvMaxPoolMask(x, y, mk, d), big + 1.5 * small),
                                    ("undo(mask)", lambda: Matrix.ConvMaxPoolUndoMask(dy, mk, dx, d, 0, False), big + 1.5 * small),
                                    ("undo(mask)+relu", lambda: Matrix.ConvMaxPoolUndoMask(dy, mk, dx, d, 0, True), big + 1.5 * small))) if masked else ()),
                                ("undo", lambda: Matrix.ConvMaxPoolUndo(x, dy, y, dx, d, 0), 2 * big + 2 * small),
                                ("undo+relu", lambda: Matrix.ConvMaxPoolUndoRelu(x, dy, y, dx, d, 0), 2 * big + 2 * small)):
            for r in timed(fn, args.reps):
                us = 1e3 * r["ms"] / r["launches"]
                print(f"{name:8s} {tag + ':' + r['kernel'][:26]:34s} {us:8.1f} {nbytes / us * 1e-3:8.0f}")
    for name, (C, H, sz) in NORMS.items():
        x, y, dy, dx = mat(N, C, H), mat(N, C, H), mat(N, C, H), mat(N, C, H)
        b = 4.0 * N * C * H * H
        for tag, fn, nbytes in (("fwd", lambda: Matrix.ConvResponseNormCrossMap(x, y, C, sz, 1e-4, 0.75, False), 2 * b),
                                ("undo", lambda: Matrix.ConvResponseNormCrossMapUndo(dy, x, y, dx, C, sz, 1e-4, 0.75, False), 3 * b)):
            for r in timed(fn, args.reps):
                us = 1e3 * r["ms"] / r["launches"]
                print(f"{name:8s} {tag + ':' + r['kernel'][:26]:34s} {us:8.1f} {nbytes / us * 1e-3:8.0f}")


if __name__ == "__main__":
    main()
