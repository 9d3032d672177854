#!/usr/bin/env python3
"""Inter-kernel gaps of a rocprofv3 --kernel-trace CSV (one queue): for every consecutive pair of kernels the idle
time between the end of one and the start of the next, summarised per (previous -> next) kernel pair, plus the kernel
time and the gap time per CG iteration.  Used to see what a HIP-graph replay of the CG iteration pays between nodes
compared with eager launches of the same kernels.
usage: python benchmarks/trace_gaps.py TRACE.csv [LABEL]"""
import collections
import csv
import sys


def short(n):
    n = n.replace("hpcla::", "")
    return n.split("(")[0][:56]


def main():
    path = sys.argv[1]
    label = sys.argv[2] if len(sys.argv) > 2 else path
    rows = [r for r in csv.DictReader(open(path, newline="")) if "hpcla::" in r["Kernel_Name"] or "Memcpy" in r["Kernel_Name"]
            or "copy" in r["Kernel_Name"].lower()]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the steady part: the last 60 % of the SpMV launches
    spmv = [i for i, r in enumerate(rows) if "spmv_row" in r["Kernel_Name"]]
    if len(spmv) < 10:
        raise SystemExit("too few SpMV launches in the trace")
    first = spmv[int(len(spmv) * 0.4)]
    last = spmv[-1]
    seg = rows[first:last]
    pair_gap = collections.defaultdict(list)
    kern = collections.defaultdict(list)
    for a, b in zip(seg, seg[1:]):
        gap = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
        pair_gap[(short(a["Kernel_Name"]), short(b["Kernel_Name"]))].append(gap)
    for r in seg:
        kern[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    iters = sum(1 for r in seg if "spmv_row" in r["Kernel_Name"])
    span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3
    ktot = sum(sum(v) for v in kern.values())
    print(f"# {label}: {iters} iterations, {span / iters:.1f} us per iteration on the device timeline; "
          f"kernels {ktot / iters:.1f} us, gaps {(span - ktot) / iters:.1f} us per iteration")
    print("# kernel                                                     launches  mean us")
    for k, v in sorted(kern.items(), key=lambda kv: -sum(kv[1])):
        print(f"{k:<60} {len(v):>8} {sum(v) / len(v):>8.2f}")
    print("# gap between  previous -> next                                                             count  mean us   max us")
    for (a, b), v in sorted(pair_gap.items(), key=lambda kv: -sum(kv[1])):
        print(f"{a:<44} -> {b:<44} {len(v):>6} {sum(v) / len(v):>8.2f} {max(v):>8.2f}")


if __name__ == "__main__":
    main()
