#!/usr/bin/env python
"""HBM-side traffic per kernel launch from two rocprofv3 PMC passes (MI355X_MICROARCH.md, "HBM" and "rocprofv3 PMC slots":
FETCH_SIZE takes 3 of the 4 TCC slots and WRITE_SIZE 2, so they need SEPARATE passes; both are in KiB; on gfx950 FETCH_SIZE
reports exactly half of the bytes of wide coalesced reads, so it is doubled — calibrated here on kernels whose traffic is known:
rnorm_fwd (reads its input once: 2 x FETCH == WRITE == 290 400 KiB) and the fused SGD pass (reads g,w,h of fc6: 2 x 216.0 MiB ==
3 x 144 MiB).  Infinity-Cache hits are counted as fetches, so this is fabric traffic, an upper bound on HBM traffic.)

    cd /tmp && export TMPDIR=/tmp
    for c in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --kernel-trace --pmc $c -d out_$c -o p --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timers
    done
    python tools/pmc_traffic.py out_FETCH_SIZE/p_counter_collection.csv out_WRITE_SIZE/p_counter_collection.csv > profiles/rNN_pmc_traffic_bench.json
"""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter and "chip::" in r["Kernel_Name"]:
                name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("chip::", "")
                agg[name].append(float(r["Counter_Value"]))
    return agg


def main():
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"command": "bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timers (AlexNet bs=256, 4 steps incl. warm-up)",
           "unit": "bytes per launch (mean over the launches of the run)",
           "correction": "FETCH_SIZE KiB x 1024 x 2 (gfx950 half-count of wide reads), WRITE_SIZE KiB x 1024; fabric-side (Infinity-Cache hits included)",
           "kernels": {}}
    for k in sorted(fetch, key=lambda k: -sum(fetch[k])):
        f, w = fetch[k], write.get(k, [0.0])
        rd, wr = 2 * 1024 * sum(f) / len(f), 1024 * sum(w) / len(w)
        out["kernels"][k] = {"launches": len(f), "read_bytes": round(rd), "write_bytes": round(wr), "traffic_bytes": round(rd + wr),
                             "fetch_size_kib_raw": round(sum(f) / len(f), 1), "write_size_kib_raw": round(sum(w) / len(w), 1)}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
