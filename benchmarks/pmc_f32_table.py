#!/usr/bin/env python3
"""HBM bytes per launch of the Float32 kernels (csrc/f32.hip) from the two rocprofv3 --pmc passes of
`./run_gpu_checks.sh TAG pmc_f32` (FETCH_SIZE, WRITE_SIZE: separate runs; FETCH_SIZE doubled for gfx950 as
/opt/skills/guides/MI355X_MICROARCH.md prescribes, both in KiB), next to the algorithmic bytes of SURVEY 8d restated for
4-byte values.  Writes the table to stdout and the SpMV figure into profiles/traffic_latest.json
(workloads.poisson2d_spmv_float32) so that bench.py's `other_configs.float32` record carries it.

usage: python benchmarks/pmc_f32_table.py TAG [ROUND]"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = 4096


def mean_of(path, counter, kernel, skip=1):
    vals = []
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter and kernel in row["Kernel_Name"]:
                vals.append(float(row["Counter_Value"]))
    vals = vals[skip:]
    if not vals:
        raise SystemExit(f"{counter}: no rows for {kernel} in {path}")
    return sum(vals) / len(vals), len(vals)


def newest(tag, step, counter):
    hits = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"{tag}_pmc_f32{step}_{counter}", "**", "*counter_collection.csv"),
                            recursive=True), key=os.path.getmtime)
    if not hits:
        raise SystemExit(f"no counter_collection.csv under gpurun_out/{tag}_pmc_f32{step}_{counter}")
    return hits[-1]


def main():
    tag = sys.argv[1]
    rnd = sys.argv[2] if len(sys.argv) > 2 else "r05"
    prof = os.path.join(ROOT, "profiles")
    n = N * N
    nnz = 5 * n - 4 * N
    ns, nnzs = N * (N // 2), 5 * N * (N // 2) - 2 * N - 2 * (N // 2)
    # kernel names as rocprofv3 prints them: the lanes = rows kernels are instances of the shared template
    # rowgather_kernel<T, I, SPLIT, KC, URX> (csrc/rowgather_t.h) since round 4's refactor
    rg = lambda kc: "rowgather_kernel<float, int, false, %d, 0>" % kc
    cases = [("spmv", rg(1), "SpMV, 5-point 4096^2, Float32 / Int32", nnz * 8 + (n + 1) * 4 + 8 * n),
             ("spmm", "rowmajor_f32_kernel<int, false, 4, 4>", "SpMM x 16 row-major, 5-point 4096 x 2048", nnzs * 8 + (ns + 1) * 4 + 128 * ns),
             ("spmm", rg(16), "SpMM x 16 column-major, 5-point 4096 x 2048", nnzs * 8 + (ns + 1) * 4 + 128 * ns)]
    out = {}
    print(f"# HBM traffic per launch of the Float32 kernels, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), tag {tag}")
    print("# FETCH_SIZE doubled (gfx950 counts 128-byte requests as 64), WRITE_SIZE exact, both KiB")
    for step, kernel, label, alg in cases:
        rd, wr = newest(tag, step, "FETCH_SIZE"), newest(tag, step, "WRITE_SIZE")
        shutil.copy(rd, os.path.join(prof, f"{rnd}_pmc_f32{step}_FETCH_SIZE.csv"))
        shutil.copy(wr, os.path.join(prof, f"{rnd}_pmc_f32{step}_WRITE_SIZE.csv"))
        f, nf = mean_of(rd, "FETCH_SIZE", kernel)
        w, nw = mean_of(wr, "WRITE_SIZE", kernel)
        b = int(round(2 * f * 1024 + w * 1024))
        print(f"{label:52s} {kernel:42s} traffic {b:>13,d} B  algorithmic {alg:>13,d} B  ratio {b / alg:.3f}  ({nf} / {nw} launches)")
        out[label] = {"kernel": kernel, "hbm_bytes": b, "algorithmic_bytes": alg, "ratio_to_algorithmic": round(b / alg, 4),
                      "FETCH_SIZE_KB_mean": round(f, 1), "WRITE_SIZE_KB_mean": round(w, 1)}
    tl = os.path.join(prof, "traffic_latest.json")
    doc = json.load(open(tl))
    first = out[cases[0][2]]
    doc.setdefault("workloads", {})["poisson2d_spmv_float32"] = dict(
        first, what="Float32 SpMV on the headline matrix (csrc/f32.hip)",
        correction="gfx950: FETCH_SIZE counts 128-B requests at 64 B -> doubled (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact",
        source=f"profiles/{rnd}_pmc_f32spmv_FETCH_SIZE.csv + profiles/{rnd}_pmc_f32spmv_WRITE_SIZE.csv (separate rocprofv3 --pmc passes over "
               f"`python3 benchmarks/bench_f32.py --only spmv2d --no-f64 --settle-ms 0 --reps 20`)")
    json.dump(doc, open(tl, "w"), indent=1)
    print(f"# stored workloads.poisson2d_spmv_float32 in profiles/traffic_latest.json")


if __name__ == "__main__":
    main()
