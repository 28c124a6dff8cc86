#!/usr/bin/env python
"""bench.py — images/sec of one full training step (fprop + bprop + wgrad + SGD [+ gradient
exchange]) of the AlexNet-class model (BASELINE.json metric) on N MI355X GPUs of one node.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU, each rank a full replica on its own synthetic batch, gradients averaged over RCCL
(the reference's train_convnet_data_parallel semantics, src/convnet.cc:407-450).  With N > 1 the headline
run is BASELINE's configuration 4 — AlexNet at a GLOBAL batch of 256 split over the ranks (256/N each:
`scaling` "strong", `config4_value` = `value`), with the weak run (--batch images on EVERY rank) nested
as `weak` and `exchange_ms_exposed` = the step with the exchange minus the same step without it;
--weak swaps the roles, --global-batch G sets the global batch.  Rank 0 prints ONE JSON line.  `value` =
images processed by all ranks / max-over-ranks wall time of K steps, bracketed by barrier +
torch.cuda.synchronize().  The weight gradients run on a second HIP stream by default (--no-overlap-wgrad: one
stream); `one_stream_ms_per_step` / `two_stream_ms_per_step` time the same K steps both ways after the headline
region, without kernel timers, so that what the second stream is worth on the box is on every line.

Extra objects on the line:
  roofline      dominant MFMA kernel: algorithmic flops / HIP-event time per launch, against the peak of the matrix pipe the
                kernel EXECUTES on.  The kernel's duration is taken with the launch alone on the chip — four fully timed one-stream
                steps of this process right after the timed region (the clock that agrees with rocprofv3's serialised kernel trace,
                `rocprof_avg_launch_ms`); the event spans of the same launches INSIDE the timed region, where the second stream's
                weight-gradient kernel shares the chip, travel as `timed_region` with their own `frac`.  With the default
                matrix path the fp32 products are formed on the bf16 pipe from exact three-way operand splits —
                six v_mfma_f32_32x32x16_bf16 per 32x32x16 block (include/convnet_hip.h) — so one algorithmic flop
                costs six bf16 flops and the ceiling in algorithmic units is 2500 / 6 = 416.7 TFLOP/s; `frac` =
                achieved / 416.7 (the same number as executed bf16 flops / 2500).  With --matrix-path fp32 the peak is
                the fp32 matrix instruction's 157.3 TFLOP/s.  No fraction on the line can exceed 1; the ratio to the
                fp32-instruction peak travels only as the labelled extra `vs_fp32_instruction_peak`.  Also
                `model_frac` = whole-step algorithmic flops / step time / the same peak, and per-family rows under
                `families` (from four extra one-stream steps with every launch timed: additive, they sum to at most the
                one-stream step).  `power_ceiling`: what THIS chip sustains on a pure stream of the executed instruction
                with N(0,1) operand planes, measured in the run by the library's probe (csrc/probe.hip) — the part clocks
                to its power budget, ~0.74 of the nominal peak; `frac` stays against the nominal 416.7,
                `frac_of_power_ceiling` travels beside it.
  fp32_mfma_path  the same step with every GEMM kernel on v_mfma_f32_32x32x2_f32 instead (--matrix-path fp32),
                timed in this process right after the main run (rank 0, N=1 only), with its OWN `roofline` object
                (dominant kernel, `frac` against the fp32 instruction's 157.3 TFLOP/s, `all_mfma_kernels`): the numbers
                to read if the bf16-split products are not accepted as fp32 arithmetic.
  cpu_baseline  the reference's own CPU path (oracle/_ref, eigenmat+CPUMatrix compiled unmodified)
                — or the C port if that build is absent — running the same model's training step
                on a bounded sample (N=6 images, 1 step, ~13 s), rank 0, N=1 only.
  ref_host      the reference's unmodified C++ host loop on this library (same model, batch, step count),
                timed after the product run in a child process; rank 0, N=1 only (--no-ref-host skips it).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # before the HIP runtime starts: see convnet_amd/__init__.py (streams share hardware queues)

PEAK_FP32_MATRIX_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD @ 2.4 GHz
PEAK_BF16_MATRIX_TFLOPS = 2500.0  # dense v_mfma_f32_32x32x16_bf16: 1024 FLOP/clk/SIMD
PEAK_HBM_GBS = 8000.0
SPLIT_PRODUCTS = 6                # bf16 MFMAs per fp32 product block on the default matrix path


def _same_kernel(bench_name, prof_name):
    """bench.py names a kernel family "ggp_kernel<2,2,2,128,split>" / "gg_kernel<1,4,3,64,rc>" / "wg_kernel<2,2,5,3,x16,split>"; rocprofv3
    prints the full template argument list ("ggp_kernel<2, 2, 2, 128, true>").  True when both name the same instantiation."""
    def parse(n):
        n = n.replace(" ", "")
        if "<" not in n or ">" not in n[n.index("<"):]:
            return n.split("::")[-1], []
        return n[:n.index("<")].split("::")[-1], n[n.index("<") + 1:n.rindex(">")].split(",")
    wb, wa = parse(bench_name)
    # the wide kernels: "gpw_kernel<128x512,split,raw>" is chip::gpw_kernel (no template; its tail-fix kernel is "gpw_tail_fix_kernel");
    # "wgw_kernel<256x192,split>" / "<256x256,split>" are wgw_kernel<3> / <4>; "gpp_kernel<2,2,2,128,split,raw|planes>" is <2,2,2,128,BRAW>
    flat = prof_name.replace(" ", "")
    if wb == "gpw_kernel":
        return "gpw_kernel(" in flat or flat.endswith("gpw_kernel") or "gpw_kernel<" in flat
    if wb == "gpv_kernel":   # "gpv_kernel<128x512,split,raw>" / "<96x512,...>" are chip::gpv_kernel<4> / <3>
        return ("gpv_kernel<4>" in flat and "128x512" in wa) or ("gpv_kernel<3>" in flat and "96x512" in wa)
    if wb == "gfc_kernel":   # "gfc_kernel<96x128,split>" is chip::gfc_kernel(chip::gfc::Params)
        return "gfc_kernel<" in flat or "gfc_kernel(" in flat or flat.endswith("gfc_kernel")   # gfc_kernel<RELU>
    if wb == "wgw_kernel":
        return ("wgw_kernel<3>" in flat and "256x192" in wa) or ("wgw_kernel<4>" in flat and "256x256" in wa)
    if wb == "gpp_kernel":
        return ("gpp_kernel<2,2,2,128,true>" in flat and "raw" in wa) or ("gpp_kernel<2,2,2,128,false>" in flat and "planes" in wa)
    fb, fa = parse(prof_name)
    nums = [a for a in wa if a.isdigit()]
    if wb != fb or fa[:len(nums)] != nums:
        return False
    rest = fa[len(nums):]
    split = "split" in wa
    if wb == "ggp_kernel":       # <WR, WC, MT, CW, SPLIT, APRE>
        return (rest[:1] == ["true"]) == split and ((len(rest) > 1 and rest[1] == "true") == ("pre" in wa))
    if wb == "wg_kernel":        # <WM, WN, MT, NTL, VEC, TS, SPLIT>
        return len(rest) >= 2 and (rest[1] == "16") == ("x16" in wa) and ((len(rest) > 2 and rest[2] == "true") == split)
    if wb == "gg_kernel":        # <WR, WC, MT, CW, A_KCONTIG, VEC, O3, SPLIT>
        return len(rest) >= 1 and (rest[0] == "true") == ("kc" in wa) and ((len(rest) > 3 and rest[3] == "true") == split)
    return not rest


def pmc_traffic(kernel, args):
    """`roofline.traffic`: fabric-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this
    same workload (FETCH_SIZE and WRITE_SIZE in separate passes, gfx950 half-count corrected: tools/pmc_traffic.py).  PMC
    collection needs rocprofv3 around the process, so the figure is read from profiles/, newest round first; null when the
    workload is not the profiled one."""
    import glob
    if args.model != "alexnet" or args.batch != 256 or args.unfused:
        return {"traffic": None}
    for path in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_pmc_traffic_bench.json")), reverse=True):
        try:
            with open(path) as f:
                kernels = json.load(f)["kernels"]
        except (OSError, ValueError, KeyError):
            continue
        for name, rec in kernels.items():
            if _same_kernel(kernel, name):
                return {"traffic": rec["traffic_bytes"], "traffic_unit": "bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, fabric side)",
                        "traffic_source": "NOT measured in this run (PMC needs rocprofv3 around the process): read from the committed "
                                          "passes of this same command, " + os.path.relpath(path, os.path.dirname(os.path.abspath(__file__)))}
    return {"traffic": None}


def live_pmc_traffic(kernel, args):
    """`roofline.traffic` measured in THIS run: the same workload, 3 steps, twice more in child processes under
    `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `... --pmc WRITE_SIZE` (separate passes: together the two counters need 5 of the
    4 TCC slots; --pmc is combined with --kernel-trace only, as MI355X_MICROARCH.md's HBM section prescribes), corrected like
    tools/pmc_traffic.py (KiB; FETCH_SIZE doubled on gfx950).  After the timed region; rank 0, one GPU only.  None when rocprofv3 is
    missing or a pass fails — the caller then falls back to the committed passes of the same command."""
    import csv
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None or os.environ.get("ROCP_TOOL_LIBRARIES") or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None   # no profiler, or this process is itself running under one
    tmp = tempfile.mkdtemp(prefix="bench_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    child = [sys.executable, os.path.abspath(__file__), "--steps", "3", "--warmup", "1", "--batch", str(args.batch), "--model", args.model,
             "--matrix-path", args.matrix_path, "--no-cpu-baseline", "--no-ref-host", "--no-other-path", "--no-kernel-timers", "--no-live-traffic"]
    child += ["--no-power-probe", "--no-two-stream-leg"] + (["--unfused"] if args.unfused else []) + (["--no-overlap-wgrad"] if args.no_overlap_wgrad else []) + \
             (["--side-stream-update"] if args.side_stream_update and not args.no_side_stream_update else [])
    sums = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", counter, "-d", out, "-o", "p", "--output-format", "csv", "--"] + child,
                               cwd="/tmp", env=env, capture_output=True, text=True, timeout=150)
            paths = [os.path.join(d, f) for d, _, fs in os.walk(out) for f in fs if f.endswith("counter_collection.csv")]
            if r.returncode != 0 or not paths:
                return None
            vals = []
            with open(paths[0]) as f:
                for row in csv.DictReader(f):
                    if row["Counter_Name"] == counter and _same_kernel(kernel, row["Kernel_Name"].split("(")[0].replace("void ", "")):
                        vals.append(float(row["Counter_Value"]))
            if not vals:
                return None
            sums[counter] = sum(vals) / len(vals)
            # the same pass's kernel trace: rocprofv3's own duration of every launch of the family (dispatches are serialised under
            # counter collection: each launch alone on the chip) — the second clock beside the HIP-event spans of the timed region
            tr = [os.path.join(d, f) for d, _, fs in os.walk(out) for f in fs if f.endswith("kernel_trace.csv")]
            if tr and "durs" not in sums:
                durs = []
                try:
                    with open(tr[0]) as f:
                        for row in csv.DictReader(f):
                            if _same_kernel(kernel, row["Kernel_Name"].split("(")[0].replace("void ", "")):
                                durs.append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
                except (OSError, KeyError, ValueError):
                    durs = []   # (the traffic figure does not depend on it)
                if durs:
                    sums["durs"] = sum(durs) / len(durs) * 1e-6   # ns -> ms
    except (OSError, subprocess.SubprocessError, KeyError, ValueError):
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    rd, wr = 2 * 1024 * sums["FETCH_SIZE"], 1024 * sums["WRITE_SIZE"]
    return {"traffic": round(rd + wr), "traffic_unit": "bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, fabric side: Infinity-Cache hits included)",
            "traffic_read": round(rd), "traffic_write": round(wr),
            **({"rocprof_avg_launch_ms": round(sums["durs"], 4)} if "durs" in sums else {}),
            "traffic_source": "measured in this run: two rocprofv3 --kernel-trace --pmc passes (FETCH_SIZE, WRITE_SIZE separately) of the same "
                              "workload, 3 steps each, after the timed region"}


def kernel_peak(name):
    """Peak of the pipe a GEMM kernel family executes on, in ALGORITHMIC fp32 TFLOP/s: the bf16-split builds (",split" in the name)
    issue SPLIT_PRODUCTS bf16 MFMA flops per algorithmic flop on the 2.5 PFLOP/s dense bf16 pipe; the others use the fp32 instruction."""
    return PEAK_BF16_MATRIX_TFLOPS / SPLIT_PRODUCTS if ",split" in name else PEAK_FP32_MATRIX_TFLOPS


def family_table(rows, steps):
    """{family: launches / ms per step, rate and fraction of ITS roofline} from kernel-timer rows of `steps` fully timed steps"""
    fam = {}
    for r in rows:
        f = fam.setdefault(r["kernel"], {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0, "executed": 0.0})
        for k in ("launches", "ms", "flops", "bytes", "executed"):
            f[k] += r[k]
    return {k: {"launches_per_step": v["launches"] / steps, "ms_per_step": round(v["ms"] / steps, 4),
                **({"tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                    "executed_tflops": round(v["executed"] / (v["ms"] * 1e-3) / 1e12, 2),
                    "frac": round(v["flops"] / (v["ms"] * 1e-3) / 1e12 / kernel_peak(k), 4)} if v["flops"] > 0 else
                   {"gbs": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1),
                    "frac": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)})}
            for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"]) if v["ms"] > 0}


def path_roofline(rows, steps):
    """A roofline object from `steps` fully timed one-stream steps (every launch alone on the chip): the MFMA family with the most time,
    its algorithmic rate against the peak of the pipe it executes on, every MFMA family priced the same way, the per-family table."""
    fam = {}
    for r in rows:
        if r["flops"] > 0:
            f = fam.setdefault(r["kernel"], {"launches": 0, "ms": 0.0, "flops": 0.0})
            for k in ("launches", "ms", "flops"):
                f[k] += r[k]
    if not fam:
        return None
    dom_name = max(fam, key=lambda k: fam[k]["ms"])
    dom = fam[dom_name]
    achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
    all_ms = sum(v["ms"] for v in fam.values())
    all_ideal_ms = sum(v["flops"] / (kernel_peak(k) * 1e12) * 1e3 for k, v in fam.items())
    return {"bound": "mfma", "kernel": dom_name, "achieved": round(achieved, 2), "peak": round(kernel_peak(dom_name), 2), "unit": "TFLOP/s",
            "frac": round(achieved / kernel_peak(dom_name), 4), "flops_per_launch": dom["flops"] / dom["launches"],
            "avg_launch_ms": round(dom["ms"] / dom["launches"], 4), "launches": dom["launches"], "sampled_steps": steps,
            "clock": "HIP events, one stream: every launch alone on the chip",
            "all_mfma_kernels": {"achieved": round(sum(v["flops"] for v in fam.values()) / (all_ms * 1e-3) / 1e12, 2),
                                 "frac": round(all_ideal_ms / all_ms, 4), "ms_per_step": round(all_ms / steps, 3)},
            "families": family_table(rows, steps)}


def one_stream_fields(rows, kernel):
    """`roofline.one_stream`: the dominant kernel's rate in the extra one-stream steps (no co-running kernel)."""
    rows = [r for r in (rows or []) if r["kernel"] == kernel]
    ms, flops, n = sum(r["ms"] for r in rows), sum(r["flops"] for r in rows), sum(r["launches"] for r in rows)
    if not rows or ms <= 0:
        return {}
    tf = flops / (ms * 1e-3) / 1e12
    return {"one_stream": {"achieved": round(tf, 2), "frac": round(tf / kernel_peak(kernel), 4), "avg_launch_ms": round(ms / n, 4), "launches": n,
                           "note": "the same kernel in 4 extra steps with everything on one stream (no co-running kernel), after the timed region"}}


def cpu_baseline(sample_n=6):   # ~13 s of CPU work on the 256-core GPU box (N=2 took 4.1-4.7 s; the cost is linear in N)
    """Time the reference CPU path on the AlexNet-class training step at a reduced batch.
    Test infrastructure used strictly as a *baseline*, never as the thing measured above."""
    import numpy as np
    import oracle
    from oracle import Geom
    impl = oracle.ref if oracle.ref is not None else oracle.port
    N = sample_n
    rng = np.random.default_rng(0)

    def rnd(*s):
        return rng.standard_normal(s).astype(np.float32)

    convs = [Geom(N, 3, 224, 224, 96, 7, 7, 2, 2, 1, 1), Geom(N, 96, 55, 55, 256, 5, 5, 2, 2, 0, 0),
             Geom(N, 256, 13, 13, 384, 3, 3, 1, 1, 1, 1), Geom(N, 384, 13, 13, 384, 3, 3, 1, 1, 1, 1),
             Geom(N, 384, 13, 13, 256, 3, 3, 1, 1, 0, 0)]
    pools = {0: Geom(N, 96, 110, 110, 96, 3, 3, 2, 2, 1, 1), 1: Geom(N, 256, 26, 26, 256, 3, 3, 2, 2, 1, 1),
             4: Geom(N, 256, 11, 11, 256, 3, 3, 2, 2, 1, 1)}
    rn = {0: 24, 1: 64}
    fcs = [(9216, 4096), (4096, 4096), (4096, 1000)]
    W = [rnd(*g.filt_shape()) * 0.01 for g in convs]
    Wf = [rnd(d, f) * 0.01 for d, f in fcs]
    x = rnd(*convs[0].in_shape())
    t0 = time.perf_counter()
    acts, cache = [x], []
    h = x
    for i, g in enumerate(convs):   # forward
        y = impl.lower_bound(impl.conv_up(g, h, W[i]), 0.0)
        rec = {"in": h, "y": y}
        h = y
        if i in pools:
            p = impl.max_pool(pools[i], h)
            rec["pool_in"], rec["pool_out"] = h, p
            h = p
        if i in rn:
            r = impl.rnorm(h, rn[i], 0.0005, 0.75)
            rec["rn_in"] = h
            h = r
        cache.append(rec)
    h = np.ascontiguousarray(h.reshape(-1, N))
    fc_in = []
    for (d, f), w in zip(fcs, Wf):
        fc_in.append(h)
        h = impl.dot(h, w, np.zeros((f, N), np.float32), 0.0, 1.0, False, True)
        if f != 1000:
            h = impl.lower_bound(h, 0.0)
    p = impl.softmax_row_major(h)
    dy = impl.softmax_grad_row_major(p, np.zeros(N, np.float32))
    for (d, f), w, hin in reversed(list(zip(fcs, Wf, fc_in))):   # backward
        impl.dot(dy, hin, np.zeros((d, f), np.float32), 0.0, 1.0 / N, True, False)
        dy = impl.dot(dy, w, np.zeros((d, N), np.float32), 0.0, 1.0)
    dy = np.ascontiguousarray(dy.reshape(convs[4].C if False else 256, 6, 6, N))
    for i in reversed(range(len(convs))):
        g, rec = convs[i], cache[i]
        if i in rn:
            dy = impl.rnorm_undo(dy, rec["rn_in"], rn[i], 0.0005, 0.75)
        if i in pools:
            dy = impl.max_pool_undo(pools[i], rec["pool_in"], dy, rec["pool_out"])
        dy = impl.relu_deriv(dy, rec["y"])
        impl.conv_outp(g, rec["in"], dy, None, 0.0, 1.0 / N)
        if i > 0:
            dy = impl.conv_down(g, dy, W[i])
    dt = time.perf_counter() - t0
    cores = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
    return {"value": round(N / dt, 4), "unit": "images/sec", "cores": cores, "kind": impl.kind,
            "sample": f"AlexNet-class training step (conv/fc fprop+dgrad+wgrad, max-pool, response-norm, softmax) at N={N}, 1 step, "
                      f"{dt:.1f}s; the reference's conv is a single-threaded naive sgemm per output location "
                      f"(eigenmat.cc:2284-2298), only its pooling/softmax loops use OpenMP"}


def ref_host_leg(args, dp=False, defer=True):
    """The north-star driver on the same clock: the reference's UNMODIFIED C++ host (src/convnet.cc ConvNet::TrainOneBatch over its own
    Layer / Edge / SGDOptimizer / Matrix, compiled from /root/reference by oracle/Makefile into oracle/_ref/libref_host_hip.so) linked
    to this library, stepping the same model and batch.  Reported beside the product number, never part of it: it runs after the
    timed region, in a child process (its own HIP context), through tools/ref_host_bench.py."""
    import subprocess
    so = os.path.join(ROOT, "oracle", "_ref", "libref_host_hip.so")
    if not os.path.exists(so):
        return {"value": None, "note": "oracle/_ref/libref_host_hip.so not built on this box (needs /root/reference at build time)"}
    cmd = [sys.executable, os.path.join(ROOT, "tools", "ref_host_bench.py"), "--model", args.model, "--batch", str(args.batch),
           "--steps", str(args.steps), "--warmup", str(args.warmup)] + (["--dp"] if dp else [])
    try:
        # the C++ host opts into the bf16-split products the way INTEGRATION.md §2 says (one environment variable or one call): the
        # library's own default at the ABI is the IEEE fp32 matrix instruction
        # ... and, the same way, into deferred epilogues (convnet_hip_set_deferred_epilogues / CONVNET_DEFER_EPILOGUES): the unfused call
        # sequence of src/conv_edge.cc + src/layer.cc then runs as fused launches, bit for bit the eager results
        env = dict(os.environ, CONVNET_GG_SPLIT="1" if args.matrix_path == "split" else "0", CONVNET_DEFER_EPILOGUES="1" if defer else "0")
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        j = json.loads(line)
        return {"value": round(j["value"], 2), "unit": "images/sec", "ms_per_step": round(j["ms_per_step"], 3), "steps": j["steps"],
                "warmup": j["warmup"], "matrix_path": args.matrix_path, "deferred_epilogues": bool(defer),
                "host": "reference src/*.cc unmodified -> reference Matrix (src/matrix.cc) -> this library; "
                                               "unfused cudamat call sequence, one metric read-back per step",
                "last_loss": j.get("last_loss")}
    except Exception as e:   # noqa: BLE001 — a reported extra, never a reason to lose the bench line
        return {"value": None, "note": f"failed: {e!r}"}


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): become the launcher — N ranks of this same
    command under torch.distributed.run on 127.0.0.1, one per GPU; rank 0 of the children prints the ONE JSON line on our stdout."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < args.gpus and not (args.share_device and have >= 1):
        sys.stderr.write(f"bench.py: --gpus {args.gpus} needs {args.gpus} visible GPUs (one process per GPU), this box has {have}; "
                         f"nothing was run\n")
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this host driver (RCCL needs it)
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU per step")
    ap.add_argument("--model", default="alexnet", choices=["alexnet", "alexnet_nin", "mnist_conv", "lenet5", "vgg"])
    ap.add_argument("--side-stream-update", action="store_true",
                    help="every edge's optimizer step on the second stream as soon as its gradient is final, instead of on the main stream "
                         "behind the backward pass (bit-identical either way).  Was the default through round 4 (11.23 -> 11.08 ms then); "
                         "with round 5's kernels, which own their CUs, it costs 0.2 ms (9.46 vs 9.65 ms, profiles/r05_stream_configs.txt): off")
    ap.add_argument("--no-side-stream-update", action="store_true", help="(the default now; accepted for older command lines)")
    ap.add_argument("--no-overlap-wgrad", action="store_true",
                    help="every edge's weight gradient on the main stream instead of on a second HIP stream beside the rest of the backward pass "
                         "(bit-identical either way).  Round 6, same-call legs without kernel timers on two gpurun calls: two streams 9.44-9.51 ms, one "
                         "stream 9.58 (profiles/r06_stream_legs.txt) — the second stream stays the default, and BOTH legs travel on every line "
                         "(`one_stream_ms_per_step`, `two_stream_ms_per_step`)")
    ap.add_argument("--overlap-wgrad", action="store_true", help="(the default; accepted for symmetry)")
    ap.add_argument("--no-two-stream-leg", action="store_true", help="skip the extra untimed-by-events runs of the same steps on one / two streams")
    ap.add_argument("--weak", action="store_true",
                    help="with --gpus N > 1: --batch images on EVERY rank as the headline run (weak scaling).  Default for N > 1 is "
                         "BASELINE's configuration 4 — a GLOBAL batch of 256 split over the ranks — with the weak run nested as `weak`")
    ap.add_argument("--timer-every", type=int, default=20,
                    help="steps between kernel-timer (HIP event) sampled steps of the timed region (a sampled step costs ~0.6 ms of event packets: every "
                         "20th step = 0.03 ms per step; the fully timed one-stream steps behind the region carry the per-family table)")
    ap.add_argument("--no-kernel-timers", action="store_true", help="diagnostic: no per-launch HIP events (roofline fields empty)")
    ap.add_argument("--staged-input", action="store_true", help="GPU-resident 256x256 chunk + crop/flip/transpose staging per batch instead of pre-staged batches")
    ap.add_argument("--unfused", action="store_true", help="issue the reference's unfused Matrix-call sequence")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true")
    ap.add_argument("--bucket-mb", type=float, default=8.0)
    ap.add_argument("--force-exchange", action="store_true", help="run the RCCL exchange path even with 1 rank (self-test)")
    ap.add_argument("--transport", default="torch", choices=["torch", "abi"],
                    help="gradient exchange through torch.distributed collectives (default) or through the library's own "
                         "convnet_hip_comm_* entries (the path a C/C++ host drives)")
    ap.add_argument("--share-device", action="store_true",
                    help="TEST flag: all ranks of --gpus N on GPU 0, the gradient exchange over gloo instead of RCCL (RCCL refuses two ranks on "
                         "one device) — the self-launch, the N-rank control flow, the strong leg and the line format on a 1-GPU box; the "
                         "numbers it prints are not a measurement of anything")
    ap.add_argument("--strong-selftest", action="store_true",
                    help="run the `strong` leg (normally only with more than one rank) on one rank too; needs --force-exchange")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="STRONG scaling (SURVEY 8(d) config 4): this many images per step in total, split evenly over the ranks "
                         "(--batch is ignored); default 0 = weak scaling, --batch images on every rank")
    ap.add_argument("--matrix-path", default="split", choices=["split", "fp32"],
                    help="how the GEMM kernels form fp32 products: exact three-way bf16 splits on the bf16 matrix pipe (default) or "
                         "the fp32 matrix instruction (convnet_hip_set_matrix_path)")
    ap.add_argument("--no-other-path", action="store_true", help="skip the second timing with the other matrix path (rank 0, 1 GPU only)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="skip the two rocprofv3 PMC child passes that measure `roofline.traffic` (rank 0, 1 GPU only; ~25 s); the figure "
                         "is then read from the committed passes under profiles/")
    ap.add_argument("--no-power-probe", action="store_true", help="skip `roofline.power_ceiling` (two 0.05 s runs of the library's matrix-pipe probe)")
    ap.add_argument("--no-ref-host", action="store_true",
                    help="skip the `ref_host` leg (the reference's own unmodified C++ ConvNet::TrainOneBatch loop linked to this library, "
                         "tools/ref_host_bench.py, timed in a child process after the product run; rank 0, 1 GPU only)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    # RCCL prints a version banner on the C-level stdout at communicator creation; keep the real stdout
    # for the single JSON line and send everything else (python and C) to stderr.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    result_line = ""

    import torch
    import torch.distributed as dist
    from convnet_amd import _lib, models
    from convnet_amd.convnet import ConvNet
    from convnet_amd.datahandler import SyntheticDataHandler
    from convnet_amd.matrix import Matrix

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.share_device else int(os.environ.get("LOCAL_RANK", "0"))
    if args.share_device:
        args.transport = "torch"   # (over gloo, below)
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; nothing was run\n")
        sys.exit(2)
    # N > 1 without an explicit choice: SURVEY 8(d) config 4 (AlexNet at a GLOBAL batch of 256, src/convnet.cc:429-431) is the headline
    if world > 1 and args.global_batch == 0 and not args.weak and args.model == "alexnet" and 256 % world == 0 and not args.staged_input:
        args.global_batch = 256
    strong = args.global_batch > 0
    if strong:
        assert args.global_batch % world == 0, f"--global-batch {args.global_batch} does not divide over {world} ranks"
        args.batch = args.global_batch // world
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback for the product path)"
    Matrix.SetupCUDADevice(local_rank)
    _lib.lib.convnet_hip_set_matrix_path(1 if args.matrix_path == "split" else 0)
    exchange = None
    if world > 1 or args.force_exchange:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # --transport abi: the library owns the ONE RCCL communicator of the process (what a C/C++ host has); torch.distributed only
        # carries the 128-byte id, rendezvous and the timing reductions, over gloo.  (Round 3 created torch's NCCL group here too: two
        # RCCL communicators in one process cost 20 % of every kernel, profiles/r03_bench_dp1_abi.json.)
        if args.transport == "abi" or args.share_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        from convnet_amd.data_parallel import GradientExchange
        exchange = GradientExchange(bucket_bytes=int(args.bucket_mb * (1 << 20)), overlap=not args.no_overlap, transport=args.transport)

    text = getattr(models, args.model)()
    net = ConvNet(text, fused=not args.unfused, process_id=rank, num_processes=world, exchange=exchange,
                  overlap_update=args.side_stream_update and not args.no_side_stream_update, overlap_wgrad=not args.no_overlap_wgrad)
    net.SetBatchsize(args.batch)
    if args.staged_input:
        # the reference's real input path: a GPU-resident chunk of 256x256 images, per-batch random 224 crop + flip +
        # transpose to CHWN by extract_patches, mean/std normalised at load, columns shuffled at chunk wrap
        import numpy as np
        from convnet_amd.datahandler import ChunkDataHandler
        inp = [l for l in net.layers_ if l.IsInput()][0]
        crop, colors = inp.GetSizeY(), inp.GetNumChannels()
        rng = np.random.default_rng(1000 + rank)
        chunk = 3 * args.batch
        data = ChunkDataHandler(rng.integers(0, 256, (chunk, colors * (crop + 32) ** 2), dtype=np.uint8),
                                rng.integers(0, net.layers_[-1].GetNumChannels(), chunk), args.batch, crop + 32, crop, colors,
                                mean=np.float32(120.0), std=np.float32(60.0), seed=rank)
    else:
        data = SyntheticDataHandler(net, args.batch, seed=1000 + rank, num_batches=2)
    net.SetupDataset(data)
    net.AllocateMemory(False)
    fwd_macs, train_macs = models.count_macs(net)
    step_flops = 2.0 * train_macs * args.batch

    def sync_all():
        # device first: with --transport abi the library's own communicator must be idle before a torch collective runs
        # (two communicators in one process: data_parallel.GradientExchange._drain_library_comm)
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        net.TrainOneBatch()
    sync_all()
    # Per-launch HIP events (the roofline leg) bracket every kernel of every `timer_every`-th timed step: live
    # inside the timed region, but sampled, because two event packets per launch cost ~3 % of the step when
    # every step carries them (18.9 vs 18.3 ms measured).
    timer_every = max(1, args.timer_every)
    timed_steps = 0
    t0 = time.perf_counter()
    for i in range(args.steps):
        on = (not args.no_kernel_timers) and i % timer_every == 0
        _lib.profile_enable(on)
        timed_steps += int(on)
        net.TrainOneBatch()
    dt_enqueue = time.perf_counter() - t0   # host time to enqueue the K steps (launches are asynchronous)
    sync_all()
    dt = time.perf_counter() - t0
    _lib.profile_enable(False)
    prof = _lib.profile_report()

    def max_over_ranks(x):
        if not dist.is_initialized():
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def timed_steps_of(n, steps, warm):
        for _ in range(warm):
            n.TrainOneBatch()
        sync_all()
        t1 = time.perf_counter()
        for _ in range(steps):
            n.TrainOneBatch()
        sync_all()
        return max_over_ranks(time.perf_counter() - t1)

    dt = max_over_ranks(dt)

    # Four more steps on ONE stream with every launch timed: each kernel alone on the chip, so the per-family times are additive (the
    # `families` table) and the dominant kernel's rate is undisturbed.  When the timed region itself ran on one stream (the default since
    # round 6) this only adds fully sampled steps; with --overlap-wgrad a launch's event span in the timed region includes what its
    # neighbour on the other stream took.  Every rank runs them (collectives inside).
    prof_one_stream, dt_one_stream, dt_two_stream = None, None, None
    side_update = args.side_stream_update and not args.no_side_stream_update
    two_main = (not args.no_overlap_wgrad) or side_update
    if not args.no_kernel_timers:
        keep = (net.overlap_wgrad_, net.overlap_update_)
        net.overlap_wgrad_, net.overlap_update_ = False, False
        net.TrainOneBatch()
        sync_all()
        _lib.profile_enable(True)
        for _ in range(4):
            net.TrainOneBatch()
        sync_all()
        _lib.profile_enable(False)
        prof_one_stream = _lib.profile_report()
        # the same K steps on one stream by the wall clock, WITHOUT kernel timers (the sampled event pairs of the timed region cost
        # ~0.07 ms per step): `one_stream_ms_per_step` and `two_stream_ms_per_step` are measured alike, whichever is the headline
        if not args.no_two_stream_leg:
            dt_one_stream = timed_steps_of(net, args.steps, 1)
        net.overlap_wgrad_, net.overlap_update_ = keep
    if not args.no_two_stream_leg:
        # ... and the same K steps with the weight gradients on a second stream: what the overlap is worth on THIS box
        keep = net.overlap_wgrad_
        net.overlap_wgrad_ = True
        dt_two_stream = timed_steps_of(net, args.steps, 2)
        net.overlap_wgrad_ = keep

    other = None
    if world == 1 and not args.no_other_path:
        # the same step on the other matrix path, same process, same net: 3 warm-up steps, then the same number of timed steps
        other_name = "fp32" if args.matrix_path == "split" else "split"
        _lib.lib.convnet_hip_set_matrix_path(1 if other_name == "split" else 0)
        for _ in range(3):
            net.TrainOneBatch()
        sync_all()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            net.TrainOneBatch()
        sync_all()
        dt_other = time.perf_counter() - t1
        prof_other = None
        if not args.no_kernel_timers:   # four more steps on one stream, every launch timed: the other path's own roofline object
            keep = (net.overlap_wgrad_, net.overlap_update_)
            net.overlap_wgrad_, net.overlap_update_ = False, False
            net.TrainOneBatch()
            sync_all()
            _lib.profile_enable(True)
            for _ in range(4):
                net.TrainOneBatch()
            sync_all()
            _lib.profile_enable(False)
            prof_other = _lib.profile_report()
            net.overlap_wgrad_, net.overlap_update_ = keep
        _lib.lib.convnet_hip_set_matrix_path(1 if args.matrix_path == "split" else 0)
        other = {"matrix_path": other_name, "value": round(args.batch * args.steps / dt_other, 2), "unit": "images/sec",
                 "ms_per_step": round(1e3 * dt_other / args.steps, 3), "steps": args.steps,
                 "model_frac": round(step_flops / (dt_other / args.steps) / 1e12 /
                                     (PEAK_BF16_MATRIX_TFLOPS / SPLIT_PRODUCTS if other_name == "split" else PEAK_FP32_MATRIX_TFLOPS), 4),
                 **({"roofline": path_roofline(prof_other, 4)} if prof_other else {})}

    # The other scaling mode beside the headline, same process, plus what the exchange leaves exposed.  Headline for N > 1 ranks is
    # SURVEY 8(d) config 4 — a GLOBAL batch of 256 split over the ranks (src/convnet.cc:429-431 semantics) — unless --weak; the weak run
    # (--batch images on every rank, global 256 x N) is then nested as `weak`, and vice versa (`strong` + `config4_value`).
    # ranks that actually took part in the gradient exchange over RCCL: the NCCL process group's size, or for the C-ABI transport the
    # library's own communicator (convnet_hip_comm_size: what ncclCommCount returned); 0 = no RCCL exchange (one GPU, or --share-device,
    # where gloo carries it)
    def rccl_count():
        if exchange is None or args.share_device:
            return 0
        return _lib.lib.convnet_hip_comm_size() if args.transport == "abi" else (dist.get_world_size() if dist.get_backend() == "nccl" else 0)
    rccl_ranks = rccl_count()
    strong_obj, weak_obj, compute_only_ms = None, None, None
    if (world > 1 or (args.strong_selftest and exchange is not None)) and args.model == "alexnet" and 256 % world == 0 and not args.staged_input:
        from convnet_amd.data_parallel import GradientExchange
        exchange.Close()

        def leg(batch, with_exchange):
            ex2 = GradientExchange(bucket_bytes=int(args.bucket_mb * (1 << 20)), overlap=not args.no_overlap, transport=args.transport) if with_exchange else None
            n2 = ConvNet(text, fused=not args.unfused, process_id=rank, num_processes=world, exchange=ex2,
                         overlap_update=args.side_stream_update and not args.no_side_stream_update, overlap_wgrad=not args.no_overlap_wgrad)
            n2.SetBatchsize(batch)
            n2.SetupDataset(SyntheticDataHandler(n2, batch, seed=2000 + rank, num_batches=2))
            n2.AllocateMemory(False)
            t = timed_steps_of(n2, args.steps, 3)
            ranks = rccl_count() if with_exchange else 0
            if ex2 is not None:
                ex2.Close()
            del n2
            return t, ranks
        if strong:
            t_co, _ = leg(args.batch, False)                      # the headline configuration without the exchange
            compute_only_ms = 1e3 * t_co / args.steps
            t_w, ranks_w = leg(256, True)                          # weak: 256 images on every rank
            weak_obj = {"scaling": "weak", "global_batch": 256 * world, "batch_per_gpu": 256, "n_gpus": world, "steps": args.steps,
                        "value": round(256 * world * args.steps / t_w, 2), "unit": "images/sec", "ms_per_step": round(1e3 * t_w / args.steps, 3),
                        "rccl_ranks": ranks_w}
        else:
            sb = 256 // world
            t_co, _ = leg(sb, False)
            t_ex, strong_ranks = leg(sb, True)
            strong_obj = {"scaling": "strong", "global_batch": 256, "batch_per_gpu": sb, "n_gpus": world, "steps": args.steps,
                          "value": round(256 * args.steps / t_ex, 2), "unit": "images/sec",
                          "ms_per_step": round(1e3 * t_ex / args.steps, 3),
                          "compute_only_ms_per_step": round(1e3 * t_co / args.steps, 3),
                          "exchange_exposed_ms": round(1e3 * (t_ex - t_co) / args.steps, 3),
                          "rccl_ranks": strong_ranks}

    # every replica must hold the same parameters after the same steps (the exchange's result is identical on all ranks and the
    # optimizer is deterministic): checked on the weak run's net, reported on the line
    replicas_identical = None
    if dist.is_initialized() and world > 1:
        import hashlib
        torch.cuda.synchronize()
        digest = hashlib.sha1(net.parameters_.ToNumpy().tobytes()).hexdigest()
        digests = [None] * world
        dist.all_gather_object(digests, digest)
        replicas_identical = all(d == digests[0] for d in digests)

    if rank == 0:
        images = args.batch * world * args.steps
        value = images / dt
        ms_per_step = 1e3 * dt / args.steps
        fam = {}
        for r in prof:
            f = fam.setdefault(r["kernel"], {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0, "executed": 0.0})
            for k in ("launches", "ms", "flops", "bytes", "executed"):
                f[k] += r[k]
        mfma = {k: v for k, v in fam.items() if v["flops"] > 0}
        # What this chip sustains on the instruction the split kernels execute, nothing else in the way (csrc/probe.hip): the part clocks to
        # its power budget, so the ceiling on real operands is below the nominal 416.7 — `frac` stays against the nominal peak, this
        # travels beside it.  After every timed leg; rank 0.
        power_ceiling = None
        if args.matrix_path == "split" and not args.no_power_probe:
            try:
                pr, pz = _lib.probe_matrix_pipe(True, 0.05), _lib.probe_matrix_pipe(False, 0.05)
                power_ceiling = {"tflops": round(pr["tflops_eq"], 1), "unit": "TFLOP/s (algorithmic fp32: executed bf16 / 6)",
                                 "frac_of_peak": round(pr["tflops_eq"] / (PEAK_BF16_MATRIX_TFLOPS / SPLIT_PRODUCTS), 4),
                                 "ghz": round(pr["ghz_issue"], 3), "ghz_shader_counter": round(pr["ghz_counter"], 3),
                                 "zero_operands_tflops": round(pz["tflops_eq"], 1), "zero_operands_ghz": round(pz["ghz_issue"], 3),
                                 "probe": "convnet_hip_probe_matrix_pipe (csrc/probe.hip): a pure v_mfma_f32_32x32x16_bf16 stream, one wave per SIMD, "
                                          "16 accumulators, register operands = h/m/l planes of N(0,1) values, no memory traffic, no split "
                                          "arithmetic, 0.05 s, measured in this run; zeros for comparison (the chip clocks to its power budget)"}
            except Exception as e:  # noqa: BLE001 — a reported extra
                power_ceiling = {"tflops": None, "note": f"probe failed: {e!r}"}
        roofline = None
        if mfma:
            # The dominant kernel is the MFMA family with the most time on the chip TO ITSELF: ranked by the one-stream steps (every
            # launch alone) where they were run.  `achieved` / `frac` / `avg_launch_ms` are the kernel's DURATION: algorithmic flops over its
            # HIP-event spans in the four fully timed one-stream steps of this process (the clock that agrees with rocprofv3's kernel trace,
            # which serialises launches: `rocprof_avg_launch_ms`).  In the timed region itself the weight gradients run on a second stream
            # beside the backward pass's other GEMMs, so a launch's event span THERE includes the time its blocks wait for CUs the other
            # stream's kernel holds — a span, not a duration; it travels as `timed_region` (with its own `frac`) and is all there is when
            # the one-stream steps were not run (--no-kernel-timers off but world > 1 etc.).
            alone_ms = {}
            for r in (prof_one_stream or []):
                if r["flops"] > 0:
                    alone_ms[r["kernel"]] = alone_ms.get(r["kernel"], 0.0) + r["ms"]
            top = max(alone_ms, key=alone_ms.get) if alone_ms else None
            dom_name = top if top in mfma else max(mfma, key=lambda k: mfma[k]["ms"])
            region = mfma[dom_name]                      # spans inside the timed region (sampled steps)
            alone = {"launches": 0, "ms": 0.0, "flops": 0.0, "executed": 0.0}
            for r in (prof_one_stream or []):
                if r["kernel"] == dom_name:
                    for k in alone:
                        alone[k] += r[k]
            dom = alone if alone["ms"] > 0 else region   # the kernel alone on the chip, where those steps were run
            dom_steps = 4 if dom is alone else timed_steps
            achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
            executed = dom["executed"] / (dom["ms"] * 1e-3) / 1e12
            region_tf = region["flops"] / (region["ms"] * 1e-3) / 1e12
            mfma_rows = {}
            for r in (prof_one_stream or prof):
                if r["flops"] > 0:
                    f = mfma_rows.setdefault(r["kernel"], {"ms": 0.0, "flops": 0.0})
                    f["ms"] += r["ms"]
                    f["flops"] += r["flops"]
            all_flops = sum(v["flops"] for v in mfma_rows.values())
            all_ms = sum(v["ms"] for v in mfma_rows.values())
            all_steps = 4 if prof_one_stream else timed_steps
            traffic_fields = None
            if world == 1 and not args.no_live_traffic:
                traffic_fields = live_pmc_traffic(dom_name, args)
            if traffic_fields is None:
                traffic_fields = pmc_traffic(dom_name, args)
            split_dom = ",split" in dom_name
            peak = kernel_peak(dom_name)
            step_peak = PEAK_BF16_MATRIX_TFLOPS / SPLIT_PRODUCTS if args.matrix_path == "split" else PEAK_FP32_MATRIX_TFLOPS
            # every MFMA family priced on its own pipe: sum of (family time at its peak) / sum of measured family time
            all_ideal_ms = sum(v["flops"] / (kernel_peak(k) * 1e12) * 1e3 for k, v in mfma_rows.items())
            roofline = {
                "bound": "mfma", "kernel": dom_name, "achieved": round(achieved, 2), "peak": round(peak, 2),
                "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                "peak_note": ("algorithmic fp32 TFLOP/s the executed pipe can deliver: 2500 dense bf16 / 6 bf16 MFMA flops per fp32 product "
                              "(v_mfma_f32_32x32x16_bf16, exact three-way operand splits)") if split_dom else
                             "v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD",
                # `achieved` counts ALGORITHMIC flops (2*N*My*Mx*F*C*Ky*Kx for every conv direction, 2*M*N*K for FC); `executed`
                # also counts the MFMA work a dgrad gather spends on border taps that read the zero page
                "executed": round(executed, 2), "executed_frac": round(executed / peak, 4),
                "dominant_by": "one-stream time per step (each launch alone on the chip)" if alone_ms else "time in the timed region",
                "vs_fp32_instruction_peak": round(achieved / PEAK_FP32_MATRIX_TFLOPS, 4),
                **traffic_fields,
                # two clocks for the kernel's duration: `avg_launch_ms` = HIP events with the launch alone on the chip (one-stream steps of
                # this process); `rocprof_avg_launch_ms` = rocprofv3's kernel duration in the PMC child pass (launches serialised); either
                # gives a fraction from `flops_per_launch` and `peak`.  `timed_region`: the event SPAN of the same launches inside the timed
                # region, where the second stream's kernel shares the chip
                **({"rocprof_frac": round(dom["flops"] / dom["launches"] / (traffic_fields["rocprof_avg_launch_ms"] * 1e-3) / 1e12 / peak, 4)}
                   if traffic_fields.get("rocprof_avg_launch_ms") else {}),
                "clocks": ("avg_launch_ms: HIP events, " + ("4 fully timed one-stream steps after the timed region (every launch alone on the chip)"
                                                             if dom is alone else "sampled steps of the timed region") +
                           "; rocprof_avg_launch_ms: rocprofv3 kernel trace, launches serialised; timed_region.avg_span_ms: HIP events in the timed region"),
                "flops_per_launch": dom["flops"] / dom["launches"], "avg_launch_ms": round(dom["ms"] / dom["launches"], 4),
                "launches": dom["launches"], "sampled_steps": dom_steps,
                "timed_region": {"avg_span_ms": round(region["ms"] / region["launches"], 4), "launches": region["launches"], "sampled_steps": timed_steps,
                                 "achieved": round(region_tf, 2), "frac": round(region_tf / peak, 4),
                                 "streams": 2 if two_main else 1,
                                 "note": "event spans of the same launches INSIDE the timed region" +
                                         ("; with the weight gradients on a second stream a span includes waiting for CUs the other stream's kernel holds"
                                          if two_main else "")},
                **({"power_ceiling": power_ceiling,
                    "frac_of_power_ceiling": round(achieved / power_ceiling["tflops"], 4)} if power_ceiling and power_ceiling.get("tflops") and split_dom else {}),
                "all_mfma_kernels": {"achieved": round(all_flops / (all_ms * 1e-3) / 1e12, 2),
                                     "frac": round(all_ideal_ms / all_ms, 4),
                                     "ms_per_step": round(all_ms / all_steps, 3)},
                "model_frac": round(step_flops / (ms_per_step * 1e-3) / 1e12 / step_peak, 4),
                "model_vs_fp32_instruction_peak": round(step_flops / (ms_per_step * 1e-3) / 1e12 / PEAK_FP32_MATRIX_TFLOPS, 4),
                **({"pipe": {"instruction": "v_mfma_f32_32x32x16_bf16", "executed_per_algorithmic_flop": SPLIT_PRODUCTS,
                             "achieved": round(SPLIT_PRODUCTS * (executed if executed > 0 else achieved), 1), "peak": PEAK_BF16_MATRIX_TFLOPS,
                             "unit": "TFLOP/s (bf16, executed)",
                             "frac": round(SPLIT_PRODUCTS * (executed if executed > 0 else achieved) / PEAK_BF16_MATRIX_TFLOPS, 4)}}
                   if split_dom else {}),
                **one_stream_fields(prof_one_stream, dom_name),   # (the same figures under their name of rounds 2-5)
                # per family, from the four fully timed ONE-STREAM steps (every launch alone on the chip): the ms_per_step column is
                # additive and sums to at most the one-stream step; whatever is not event-timed (copies, host gaps) is the remainder
                "families": family_table(prof_one_stream, 4) if prof_one_stream else family_table(prof, max(1, timed_steps)),
                "families_source": "4 extra one-stream steps, every launch timed" if prof_one_stream else "timed region (sampled steps)",
                "families_ms_per_step_sum": round(sum(r["ms"] for r in (prof_one_stream or prof)) / (4 if prof_one_stream else max(1, timed_steps)), 3),
                "ops": {f'{r["kernel"]}|{r["op"]}': round(r["ms"] / timed_steps, 4) for r in sorted(prof, key=lambda r: -r["ms"])},
            }
        out = {
            "metric": "images/sec (fprop+bprop+wgrad) AlexNet 224x224 bs=256" if args.model == "alexnet" else f"images/sec {args.model}",
            "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "one_stream_ms_per_step": round(1e3 * dt_one_stream / args.steps, 3) if dt_one_stream else None,
            "stream_legs_note": "one_stream / two_stream_ms_per_step: the same K steps timed after the headline region without kernel timers (weight gradients on "
                                "the main stream / on a second HIP stream); the headline region carries sampled HIP-event pairs (~0.07 ms per step)",
            "two_stream_ms_per_step": round(1e3 * dt_two_stream / args.steps, 3) if dt_two_stream else None,
            "host_enqueue_ms_per_step": round(1e3 * dt_enqueue / args.steps, 3), "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "rccl_ranks": rccl_ranks,
            "arithmetic": ("fp32 operands, fp32 accumulation, fp32 results; GEMM products formed on the bf16 matrix pipe from exact three-way "
                           "operand splits, 6 of 9 cross terms (dropped terms <= 2^-23 of a product; measured max error 4.08 x 2^-24 of sum|ab| at "
                           "K=3456 vs 4.55 x 2^-24 for the fp32 matrix instruction, tools/split_gemm.hip); same parity tests and tolerances as "
                           "the fp32-MFMA path, which `fp32_mfma_path` times beside it") if args.matrix_path == "split" else
                          "fp32 operands, products and accumulation (v_mfma_f32_32x32x2_f32)",
            "config": {"workload": f"{args.model} (convnet_amd.models.{args.model}" + (": the reference's AlexNet-class ILSVRC pbtxt) " if args.model == "alexnet" else ") ") +
                                   f"training step, 224x224x3 synthetic " + ("uint8-valued 256x256 chunk, random crop+flip staged on the GPU each step, " if args.staged_input else "N(0,1) images, ") + f"{args.batch} images per GPU, "
                                   f"SGD+momentum+L2, dropout on, {'fused' if not args.unfused else 'unfused'} ABI calls",
                       "batch_per_gpu": args.batch, "global_batch": args.batch * world,
                       "parallelism": f"dp{world}" + ("" if world == 1 else (" rccl-allreduce " + ("overlapped" if not args.no_overlap else "serial") + (" (C-ABI entries)" if args.transport == "abi" else "")))
                                      + (f" strong (global batch {args.global_batch} = {args.batch}/GPU)" if strong else f" weak ({args.batch}/GPU)"),
                       "streams": ("weight gradients" if not args.no_overlap_wgrad else "") + (" + optimizer steps" if (args.side_stream_update and not args.no_side_stream_update) else "") +
                                  (" on a second HIP stream beside the backward pass" if (not args.no_overlap_wgrad or (args.side_stream_update and not args.no_side_stream_update)) else "one stream"),
                       "params": net.NumParameters(), "train_gflop_per_image": round(2e-9 * train_macs, 4)},
            "roofline": roofline,
        }
        if strong_obj is not None:
            out["strong"] = strong_obj
            # SURVEY 8(d) config 4 — AlexNet at a GLOBAL batch of 256 on N GPUs — at top level beside the weak `value` (global 256 x N)
            out["config4_value"] = strong_obj["value"]
        if strong and args.model == "alexnet" and args.global_batch == 256:
            out["config4_value"] = out["value"]   # the headline IS config 4
        if compute_only_ms is not None:
            out["compute_only_ms_per_step"] = round(compute_only_ms, 3)
            out["exchange_ms_exposed"] = round(ms_per_step - compute_only_ms, 3)   # step with the exchange - the same step without it
        if weak_obj is not None:
            out["weak"] = weak_obj
        if replicas_identical is not None:
            out["replicas_identical"] = replicas_identical
        if args.share_device:
            out["share_device"] = "TEST run: every rank on GPU 0, exchange over gloo — not a measurement"
        if other is not None:
            out["fp32_mfma_path" if other["matrix_path"] == "fp32" else "split_path"] = other
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline()
            except Exception as e:  # the baseline is a report, never a reason to lose the bench line
                out["cpu_baseline"] = {"value": None, "unit": "images/sec", "cores": 0, "kind": "port", "sample": f"failed: {e!r}"}
        if world == 1 and not args.no_ref_host and args.model in ("alexnet", "alexnet_nin") and not args.staged_input:
            out["ref_host"] = ref_host_leg(args)
            out["ref_host_eager"] = ref_host_leg(args, defer=False)   # the same host with every unfused call launched as it is made
            # ... and the same host with the reference's data-parallel step re-expressed on the library's exchange entries
            # (SeamDPNet), a world of one rank: what the exchange plumbing costs a C++ host
            out["ref_host_dp"] = ref_host_leg(args, dp=True)
        result_line = json.dumps(out)
    if exchange is not None:
        exchange.Close()
    if dist.is_initialized():
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (result_line + "\n").encode())   # the ONE line on the real stdout


if __name__ == "__main__":
    main()
