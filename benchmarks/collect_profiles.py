#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of `run_gpu_checks.sh TAG prof pmc_rd pmc_wr` (under gpurun_out/) into the
tracked evidence under profiles/: kernel stats + trace CSV, the two PMC CSVs and
profiles/traffic_latest.json (HBM bytes per launch of the dominant kernel, corrected as
/opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE counts 128-byte requests as
64 bytes -> doubled; WRITE_SIZE is exact; both are reported in KiB).

usage: python benchmarks/collect_profiles.py TAG [ROUND]      (ROUND defaults to r02)
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "spmv_rowblock_quad_kernel<int, false, false>"     # <index type, SPLIT, WAIT>: the single-GPU headline kernel
B_ALG = 1_341_980_676          # config 2, Int32


def find(tag, step, suffix):
    hits = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"{tag}_{step}", "**", f"*{suffix}"), recursive=True))
    if not hits:
        raise SystemExit(f"no *{suffix} under gpurun_out/{tag}_{step}")
    return hits[-1]


def counter_mean(path, name):
    vals = []
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if KERNEL in row["Kernel_Name"] and row["Counter_Name"] == name:
                vals.append(float(row["Counter_Value"]))
    if not vals:
        raise SystemExit(f"{name}: no rows for {KERNEL} in {path}")
    return sum(vals) / len(vals), len(vals)


def main():
    tag = sys.argv[1]
    rnd = sys.argv[2] if len(sys.argv) > 2 else "r02"
    prof = os.path.join(ROOT, "profiles")
    shutil.copy(find(tag, "prof", "kernel_stats.csv"), os.path.join(prof, f"{rnd}_bench_kernel_stats.csv"))
    # the trace is large: keep the SpMV launches only
    src = find(tag, "prof", "kernel_trace.csv")
    with open(src, newline="") as f, open(os.path.join(prof, f"{rnd}_bench_kernel_trace.csv"), "w", newline="") as g:
        r = csv.reader(f)
        w = csv.writer(g)
        head = next(r)
        w.writerow(head)
        kcol = head.index("Kernel_Name")
        for row in r:
            if "hpcla::" in row[kcol]:
                w.writerow(row)
    rd = find(tag, "pmc_rd", "counter_collection.csv")
    wr = find(tag, "pmc_wr", "counter_collection.csv")
    shutil.copy(rd, os.path.join(prof, f"{rnd}_bench_pmc_FETCH_SIZE.csv"))
    shutil.copy(wr, os.path.join(prof, f"{rnd}_bench_pmc_WRITE_SIZE.csv"))
    fetch_kb, nf = counter_mean(rd, "FETCH_SIZE")
    write_kb, nw = counter_mean(wr, "WRITE_SIZE")
    hbm = int(round(2 * fetch_kb * 1024 + write_kb * 1024))
    out = {
        "kernel": "hpcla::" + KERNEL,
        "source": f"profiles/{rnd}_bench_pmc_FETCH_SIZE.csv + profiles/{rnd}_bench_pmc_WRITE_SIZE.csv (separate rocprofv3 "
                  f"--pmc passes over `python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-strong --no-extras --no-packed`; {nf} / {nw} launches)",
        "FETCH_SIZE_KB_mean": round(fetch_kb, 1), "WRITE_SIZE_KB_mean": round(write_kb, 1),
        "correction": "gfx950: FETCH_SIZE counts 128-B requests at 64 B -> doubled (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact",
        "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": B_ALG,
        "ratio_to_algorithmic": round(hbm / B_ALG, 4),
        "calibration_note": "k_copy (known 1 207 762 944 B read with 16-B loads) read FETCH_SIZE 612 339 KB in the r01 "
                            "calibration pass (profiles/r01_calib_pmc_*.csv) -> factor 1.926 rather than 2",
    }
    with open(os.path.join(prof, "traffic_latest.json"), "w") as f:
        json.dump(out, f, indent=1)
    with open(os.path.join(prof, f"{rnd}_bench_kernel_stats.csv")) as f:
        for line in f.readlines()[:3]:
            print(line.rstrip()[:200])
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
