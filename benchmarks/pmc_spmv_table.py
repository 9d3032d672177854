#!/usr/bin/env python3
"""Side-by-side counter table of the SpMV contexts of benchmarks/pmc_spmv_cases.py.

usage: python benchmarks/pmc_spmv_table.py TAG [OUT]
reads  gpurun_out/TAG_pmcs_manifest.json and every gpurun_out/TAG_pmcs_*/**/*counter_collection.csv (one rocprofv3 --pmc
pass each), splits the dispatches of spmv_rowblock_quad_kernel by the manifest (cases in launch order, the first launch
of every case dropped), and prints mean counter value per launch and per case, plus ratios against the 2-D case
normalised per stored entry where that makes sense."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "spmv_row"            # spmv_rowgather_kernel (round 4 default) or spmv_rowblock_quad_kernel (HPCLA_SPMV_KERNEL=quad)


def main(tag, out=None):
    man = json.load(open(os.path.join(ROOT, "gpurun_out", f"{tag}_pmcs_manifest.json")))
    cases = man["cases"]
    table = collections.OrderedDict()          # counter -> {case: mean}
    dur = collections.defaultdict(list)        # case -> kernel durations under counter collection (ns)
    for path in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"{tag}_pmcs_*", "**", "*counter_collection.csv"), recursive=True)):
        rows = collections.defaultdict(dict)   # dispatch id -> {counter: value}
        stamp = {}
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                if KERNEL in r["Kernel_Name"]:
                    d = int(r["Dispatch_Id"])
                    rows[d][r["Counter_Name"]] = rows[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                    stamp[d] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        ids = sorted(rows)
        want = sum(c["launches"] for c in cases)
        if len(ids) != want:
            print(f"# {os.path.relpath(path, ROOT)}: {len(ids)} dispatches of the kernel, manifest says {want} -- skipped")
            continue
        pos = 0
        for c in cases:
            mine = ids[pos + 1:pos + c["launches"]]           # first launch of the case dropped
            pos += c["launches"]
            for name in rows[mine[0]]:
                table.setdefault(name, {})[c["case"]] = sum(rows[d][name] for d in mine) / len(mine)
            dur[c["case"]] += [stamp[d] for d in mine]
    names = [c["case"] for c in cases]
    nnz = {c["case"]: c["nnz"] for c in cases}
    lines = []
    lines.append(f"# counters per launch of hpcla::{man.get('kernel', 'spmv_rowblock_quad_kernel')}<int, false, false(, false)>, mean over launches 2..n of each case; block order: "
                 f"2-D groups of {man['order2d']}, 3-D groups of {man['order3d']}")
    lines.append("# cases: " + "; ".join(f"{c['case']}: {c['rows']} rows, {c['nnz']} entries, {c['algorithmic_bytes']} algorithmic bytes" for c in cases))
    lines.append("# event_ms (no profiler, same process): " + json.dumps({k: round(v, 4) for k, v in man["event_ms"].items()}))
    lines.append("# kernel duration UNDER counter collection (us, mean over all passes): " +
                 ", ".join(f"{n} {sum(dur[n]) / max(len(dur[n]), 1) / 1e3:.1f}" for n in names))
    w = max(len(k) for k in table) + 2
    lines.append("counter".ljust(w) + "".join(n.rjust(16) for n in names) + "   per-entry ratio vs 2d: " + " ".join(n for n in names[1:]))
    for k, v in table.items():
        row = k.ljust(w) + "".join((f"{v.get(n, float('nan')):16.4g}") for n in names)
        base = v.get("2d")
        if base:
            row += "   " + " ".join(f"{(v.get(n, float('nan')) / nnz[n]) / (base / nnz['2d']):7.3f}" for n in names[1:])
        lines.append(row)
    text = "\n".join(lines)
    print(text)
    if out:
        open(out, "w").write(text + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
