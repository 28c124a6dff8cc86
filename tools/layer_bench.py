#!/usr/bin/env python
"""Per-layer kernel timing (AlexNet-class shapes, N=256): fprop / dgrad / wgrad TFLOP/s per conv and FC
edge, measured with the library's HIP-event profiler.  Usage: python tools/layer_bench.py [--n 256] [--reps 5]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from convnet_amd import _lib  # noqa: E402
from convnet_amd.matrix import Matrix, make_conv_desc  # noqa: E402

CONVS = {"conv1": (3, 224, 96, 7, 2, 1), "conv2": (96, 55, 256, 5, 2, 0), "conv3": (256, 13, 384, 3, 1, 1),
         "conv4": (384, 13, 384, 3, 1, 1), "conv5": (384, 13, 256, 3, 1, 0)}
FCS = {"fc6": (9216, 4096), "fc7": (4096, 4096), "fc8": (4096, 1000)}


def mat(rows, cols, shape4=None, rng=None):
    m = Matrix()
    m.AllocateGPUMemory(rows, cols)
    m._t.normal_(0, 1)
    if shape4:
        m.SetShape4D(*shape4)
    return m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only", default="")
    ap.add_argument("--vgg", action="store_true", help="VGG-16 conv shapes (3x3 s1 p1) instead of AlexNet's")
    args = ap.parse_args()
    if args.vgg:
        CONVS.clear()
        FCS.clear()
        for i, (C, H, F) in enumerate([(3, 224, 64), (64, 224, 64), (64, 112, 128), (128, 112, 128), (128, 56, 256), (256, 56, 256),
                                       (256, 28, 512), (512, 28, 512), (512, 14, 512)]):
            CONVS[f"v{i}_{C}x{H}"] = (C, H, F, 3, 1, 1)
    Matrix.SetupCUDADevice(0)
    N = args.n
    rows = []
    for name, (C, H, F, K, s, p) in CONVS.items():
        if args.only and args.only not in name:
            continue
        M = (H + 2 * p - K) // s + 1
        d = make_conv_desc(C, F, K, K, s, s, p, p)
        x = mat(N, H * H * C, (N, H, H, C))
        w = mat(F, K * K * C, (F, K, K, C))
        y = mat(N, M * M * F, (N, M, M, F))
        dx = mat(N, H * H * C, (N, H, H, C))
        dw = mat(F, K * K * C, (F, K, K, C))
        for _ in range(2):
            Matrix.ConvUp(x, w, y, d, 0)
            Matrix.ConvDown(y, w, dx, d, 0)
            Matrix.ConvOutp(x, y, dw, d, 0, 0, 0, 1.0)
        torch.cuda.synchronize()
        _lib.profile_enable(True)
        for _ in range(args.reps):
            Matrix.ConvUp(x, w, y, d, 0)
            Matrix.ConvDown(y, w, dx, d, 0)
            Matrix.ConvOutp(x, y, dw, d, 0, 0, 0, 1.0)
        _lib.profile_enable(False)
        for r in _lib.profile_report():
            rows.append((name, r))
    for name, (D, F) in FCS.items():
        if args.only and args.only not in name:
            continue
        x, w, y, dx, dw = mat(N, D), mat(F, D), mat(N, F), mat(N, D), mat(F, D)
        for _ in range(2):
            Matrix.Dot(x, w, y, 0, 1, False, True)
            Matrix.Dot(y, w, dx, 0, 1)
            Matrix.Dot(y, x, dw, 0, 1.0 / N, True, False)
        torch.cuda.synchronize()
        _lib.profile_enable(True)
        for _ in range(args.reps):
            Matrix.Dot(x, w, y, 0, 1, False, True)
            Matrix.Dot(y, w, dx, 0, 1)
            Matrix.Dot(y, x, dw, 0, 1.0 / N, True, False)
        _lib.profile_enable(False)
        for r in _lib.profile_report():
            rows.append((name, r))
    print(f"{'layer':6s} {'op':12s} {'kernel':26s} {'us/launch':>10s} {'TFLOP/s':>8s}")
    for name, r in rows:
        us = 1e3 * r["ms"] / r["launches"]
        tf = r["flops"] / (r["ms"] * 1e-3) / 1e12 if r["flops"] else 0.0
        print(f"{name:6s} {r['op']:12s} {r['kernel']:26s} {us:10.1f} {tf:8.1f}")


if __name__ == "__main__":
    main()
