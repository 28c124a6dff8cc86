#!/usr/bin/env python
"""The chip's sustained rate on the instruction the default GEMM kernels execute, with nothing else in the way
(convnet_hip_probe_matrix_pipe, csrc/probe.hip: one wave per SIMD, sixteen accumulators, register operands, six products per block in
split_mac's order — no memory traffic, no split arithmetic, no barriers), on the h / m / l planes of N(0,1) values and on zeros, for
several durations.  Prints executed bf16 TFLOP/s, the same in algorithmic fp32 units (/ 6: what bench.py's `roofline.power_ceiling`
carries), and the effective clock.  Under `rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES` the same run gives
the clock as GRBM_GUI_ACTIVE / 8 XCDs / kernel duration (tools/profile_round.sh does that).
Usage: python tools/power_ceiling.py [--seconds 0.05 0.5]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from convnet_amd import _lib  # noqa: E402
from convnet_amd.matrix import Matrix  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, nargs="+", default=[0.01, 0.05, 0.5])
    args = ap.parse_args()
    Matrix.SetupCUDADevice(0)
    print("v_mfma_f32_32x32x16_bf16 stream, 1 wave / SIMD, 16 accumulators, register operands; nominal 2500 TFLOP/s bf16 = 416.7 TFLOP/s-eq at 2.4 GHz")
    print(f"{'operands':22s} {'seconds':>8s} {'bf16 TFLOP/s':>13s} {'TFLOP/s-eq':>11s} {'frac of 416.7':>14s} {'GHz (counter)':>14s} {'GHz (issue)':>12s}")
    for sec in args.seconds:
        for rnd, name in ((True, "N(0,1) h/m/l planes"), (False, "zeros")):
            r = _lib.probe_matrix_pipe(rnd, sec)
            print(f"{name:22s} {sec:8.3f} {r['bf16_tflops']:13.1f} {r['tflops_eq']:11.1f} {r['tflops_eq'] / (2500.0 / 6):14.3f} {r['ghz_counter']:14.3f} {r['ghz_issue']:12.3f}")


if __name__ == "__main__":
    main()
