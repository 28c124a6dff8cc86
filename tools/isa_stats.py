#!/usr/bin/env python
"""Static numbers of the main loops of gpw_kernel / wgw_kernel / gfc_kernel (the hand-scheduled one-block-per-CU kernels): compiles
the translation units for gfx950 with the library's flags (-save-temps), finds each kernel's inner loop in the assembly and counts
what one chunk issues — MFMAs, everything else per MFMA gap, branches, scratch accesses, LDS-DMA loads, LDS reads — plus the register
and scratch figures of the kernel descriptor.  No GPU.  Usage: python tools/isa_stats.py > profiles/rNN_isa_wide_kernels.txt"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from convnet_amd import build as B  # noqa: E402

KERNELS = {"patch_gemm.hip": [("gpw_kernel (128 x 512 patch tile)", "_ZN4chip10gpw_kernelENS_8GGParamsENS_12GGClassTableE", 96)],
           "wgrad_wide.hip": [("wgw_kernel<3> (256 x 192)", "_ZN4chip10wgw_kernelILi3EEEvNS_8WGParamsE", 144),
                              ("wgw_kernel<4> (256 x 256)", "_ZN4chip10wgw_kernelILi4EEEvNS_8WGParamsE", 192)],
           "fewc_conv.hip": [("gfc_kernel<true> (conv1 fprop, fused ReLU)", "_ZN4chip10gfc_kernelILb1EEEvNS_3gfc6ParamsE", 18)]}


def instr(lines):
    for l in lines:
        t = l.strip()
        if not t or t.startswith(";") or t.endswith(":") or t.startswith("."):
            continue
        yield t.split()[0]


def main():
    with tempfile.TemporaryDirectory() as d:
        for src, kernels in KERNELS.items():
            cmd = ["hipcc", *B.FLAGS, *B.FILE_FLAGS.get(src, []), "-c", os.path.join(B.SRC, src), "-o", os.path.join(d, "o.o"), "-save-temps=obj"]
            subprocess.run(cmd, check=True, cwd=d, capture_output=True)
            asm = open(os.path.join(d, src.replace(".hip", "") + "-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
            L = asm.split("\n")
            for name, sym, per_chunk in kernels:
                a = next(i for i, l in enumerate(L) if l.startswith(sym + ":"))
                e = next(i for i in range(a, len(L)) if L[i].strip().startswith("s_endpgm"))
                body = L[a:e]
                # the hot loop: the back-branch whose span holds the most MFMAs
                labels = {l.split(":")[0]: i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
                best, loose = None, None
                for i, l in enumerate(body):
                    m = re.match(r"\s*s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.match(r"\s*s_branch\s+(\.LBB\d+_\d+)", l)
                    if m and m.group(1) in labels and labels[m.group(1)] < i:
                        span = body[labels[m.group(1)]:i]
                        n = sum(1 for x in span if x.strip().startswith("v_mfma"))
                        if any(re.match(r"\s*s_(c)?branch", x) for x in span):
                            if n > 0 and (loose is None or n > loose[2]):
                                loose = (labels[m.group(1)], i, n)   # a loop with branches inside (gfc_kernel: two tiles per turn, the last-tile checks)
                            continue   # not the innermost loop
                        if best is None or n > best[2]:
                            best = (labels[m.group(1)], i, n)
                if best is None or best[2] == 0:
                    best = loose
                lo, hi, nm = best
                ops = list(instr(body[lo:hi + 1]))
                cnt = collections.Counter(ops)
                gaps, cur = [], 0
                for o in ops:
                    if o.startswith("v_mfma"):
                        gaps.append(cur)
                        cur = 0
                    else:
                        cur += 1
                gaps[0] += cur
                chunks = nm / per_chunk
                est = sum(max(32, 8 + 4 * g) for g in gaps) / len(gaps)
                meta = asm[asm.index("    .name:           " + sym):]
                meta = meta[:meta.index("  - .a") if "  - .a" in meta[10:] else len(meta)]
                get = lambda k: re.search(k + r":\s+(\d+)", meta).group(1)  # noqa: E731
                print(f"{name}")
                print(f"  descriptor: vgpr_count {get('.vgpr_count')}  sgpr_count {get('.sgpr_count')}  vgpr_spill {get('.vgpr_spill_count')}  "
                      f"sgpr_spill {get('.sgpr_spill_count')}  scratch bytes {get('.private_segment_fixed_size')}")
                print(f"  main loop: {nm} MFMAs = {chunks:g} chunk(s) per iteration; per chunk: {per_chunk} MFMAs, "
                      f"{(len(ops) - nm) / chunks:.0f} other instructions ({(len(ops) - nm) / nm:.2f} per MFMA gap), "
                      f"{sum(v for k, v in cnt.items() if k.startswith('global_load_lds')) / chunks:g} LDS-DMA loads, "
                      f"{sum(v for k, v in cnt.items() if k.startswith('ds_read')) / chunks:g} ds_read, "
                      f"{sum(v for k, v in cnt.items() if k.startswith('s_barrier')) / chunks:g} barrier")
                print(f"  in the loop: {sum(v for k, v in cnt.items() if k.startswith('s_cbranch')) - 1} branches beside the loop's own, "
                      f"{sum(v for k, v in cnt.items() if k.startswith('scratch_'))} scratch accesses, "
                      f"{sum(v for k, v in cnt.items() if k in ('v_pk_add_f32', 'v_pk_mul_f32', 'v_pk_fma_f32'))} packed fp32 VALU, {cnt.get('s_nop', 0)} s_nop")
                print(f"  largest filler run between two MFMAs: {max(gaps)}; issue-slot estimate max(32, 8 + 4 x fillers) per gap: {est:.1f} cycles per MFMA "
                      f"({32 / est:.2f} of the pipe)")
                top = ", ".join(f"{k} {v / chunks:.0f}" for k, v in cnt.most_common(9) if not k.startswith("v_mfma"))
                print(f"  per chunk: {top}")
                print()


if __name__ == "__main__":
    main()
