#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of `./run_gpu_checks.sh TAG pmc_all` (under gpurun_out/) into the tracked evidence under
profiles/: kernel stats + trace CSV of the headline command, the PMC CSVs, and profiles/traffic_latest.json -- HBM bytes
per launch of the headline kernel and, under "workloads", per launch / per CG iteration of every sub-record of the bench
line -- corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE counts 128-byte requests
as 64 bytes -> doubled; WRITE_SIZE is exact; both are reported in KiB; separate --pmc passes.

usage: python benchmarks/collect_profiles.py TAG [ROUND]      (ROUND defaults to r05)
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "spmv_rowgather_kernel<int, false, false, false>"     # <index type, SPLIT, WAIT, LONGR>: the single-GPU headline kernel (row gather; round 5 added LONGR)
B_ALG = 1_341_980_676          # config 2, Int32
ORDER_NOTE = ("block order fixed for the profiled runs (HPCLA_BLOCK_ORDER=32, 64 for the 3-D slab: what the plans' measurement picks on "
              "these matrices) so that no launch of the plan-time measurement sits in the per-kernel means")
CORRECTION = "gfx950: FETCH_SIZE counts 128-B requests at 64 B -> doubled (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact"


def find(tag, step, suffix):
    hits = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"{tag}_{step}", "**", f"*{suffix}"), recursive=True),
                  key=os.path.getmtime)                    # the NEWEST: a reused tag may have left older files behind
    if not hits:
        raise SystemExit(f"no *{suffix} under gpurun_out/{tag}_{step}")
    return hits[-1]


def per_kernel(path, counter):
    """kernel name -> list of counter values (one per launch)"""
    acc = collections.defaultdict(list)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter:
                acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return acc


def counter_mean(path, name, kernel=KERNEL, skip=0):
    vals = [v for k, vs in per_kernel(path, name).items() if kernel in k for v in vs][skip:]
    if not vals:
        raise SystemExit(f"{name}: no rows for {kernel} in {path}")
    return sum(vals) / len(vals), len(vals)


def hbm(fetch_kb, write_kb):
    return int(round(2 * fetch_kb * 1024 + write_kb * 1024))


def workload_single_kernel(tag, step, kernel, rnd, prof, label, alg, cmd, skip=1):
    rd = find(tag, f"pmc_{step}_FETCH_SIZE", "counter_collection.csv")
    wr = find(tag, f"pmc_{step}_WRITE_SIZE", "counter_collection.csv")
    shutil.copy(rd, os.path.join(prof, f"{rnd}_pmc_{step}_FETCH_SIZE.csv"))
    shutil.copy(wr, os.path.join(prof, f"{rnd}_pmc_{step}_WRITE_SIZE.csv"))
    f, nf = counter_mean(rd, "FETCH_SIZE", kernel, skip)
    w, nw = counter_mean(wr, "WRITE_SIZE", kernel, skip)
    b = hbm(f, w)
    return {"what": label, "kernel": kernel, "hbm_bytes": b, "FETCH_SIZE_KB_mean": round(f, 1), "WRITE_SIZE_KB_mean": round(w, 1),
            "algorithmic_bytes": alg, "ratio_to_algorithmic": round(b / alg, 4) if alg else None, "correction": CORRECTION,
            "source": f"profiles/{rnd}_pmc_{step}_FETCH_SIZE.csv + profiles/{rnd}_pmc_{step}_WRITE_SIZE.csv (separate rocprofv3 --pmc "
                      f"passes over `{cmd}`; {nf} / {nw} launches; block order fixed, HPCLA_BLOCK_ORDER=32)"}


def workload_cg(tag, rnd, prof, alg_textbook, alg_moved):
    """HBM bytes per CG ITERATION: every kernel of the iteration (SpMV + p.Ap epilogue, its partial reduction, the
    residual update with its two reduction stages, the direction update), per-launch means x launches per iteration."""
    rd = find(tag, "pmc_cg_FETCH_SIZE", "counter_collection.csv")
    wr = find(tag, "pmc_cg_WRITE_SIZE", "counter_collection.csv")
    shutil.copy(rd, os.path.join(prof, f"{rnd}_pmc_cg_FETCH_SIZE.csv"))
    shutil.copy(wr, os.path.join(prof, f"{rnd}_pmc_cg_WRITE_SIZE.csv"))
    names = ("spmv_rowgather_kernel", "cg_direction_kernel", "cg_residual_kernel", "reduce_stage1", "reduce_stage2")
    fk, wk = per_kernel(rd, "FETCH_SIZE"), per_kernel(wr, "WRITE_SIZE")
    n_iter_f = sum(len(v) for k, v in fk.items() if "spmv_rowgather_kernel" in k)
    n_iter_w = sum(len(v) for k, v in wk.items() if "spmv_rowgather_kernel" in k)
    ftot = sum(sum(v) for k, v in fk.items() if any(n in k for n in names))
    wtot = sum(sum(v) for k, v in wk.items() if any(n in k for n in names))
    per = {}
    for n in names:
        f = [x for k, v in fk.items() if n in k for x in v]
        w = [x for k, v in wk.items() if n in k for x in v]
        per[n] = {"launches_per_iteration": round(len(f) / max(n_iter_f, 1), 2),
                  "hbm_bytes_per_launch": hbm(sum(f) / max(len(f), 1), sum(w) / max(len(w), 1))}
    b = hbm(ftot / n_iter_f, wtot / n_iter_w)
    return {"what": "CG iteration, 3-D 7-pt 512x512x64 slab (config 4's per-GPU share), fused form",
            "hbm_bytes": b, "per_kernel": per, "algorithmic_bytes_textbook": alg_textbook, "moved_bytes_fused_form": alg_moved,
            "ratio_to_textbook": round(b / alg_textbook, 4), "ratio_to_moved": round(b / alg_moved, 4), "correction": CORRECTION,
            "source": f"profiles/{rnd}_pmc_cg_FETCH_SIZE.csv + profiles/{rnd}_pmc_cg_WRITE_SIZE.csv (separate rocprofv3 --pmc passes over "
                      f"`python3 bench.py --workload poisson3d_cg --steps 10 --warmup 5`; {n_iter_f} / {n_iter_w} iterations, all of "
                      "them counted: setup launches of the same kernels included; block order fixed to XCD groups of 64, HPCLA_BLOCK_ORDER=64)"}


def main():
    tag = sys.argv[1]
    rnd = sys.argv[2] if len(sys.argv) > 2 else "r05"
    prof = os.path.join(ROOT, "profiles")
    shutil.copy(find(tag, "prof", "kernel_stats.csv"), os.path.join(prof, f"{rnd}_bench_kernel_stats.csv"))
    # the trace is large: keep the library's launches only
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
    rd = find(tag, "pmc_head_FETCH_SIZE", "counter_collection.csv")
    wr = find(tag, "pmc_head_WRITE_SIZE", "counter_collection.csv")
    shutil.copy(rd, os.path.join(prof, f"{rnd}_bench_pmc_FETCH_SIZE.csv"))
    shutil.copy(wr, os.path.join(prof, f"{rnd}_bench_pmc_WRITE_SIZE.csv"))
    fetch_kb, nf = counter_mean(rd, "FETCH_SIZE")
    write_kb, nw = counter_mean(wr, "WRITE_SIZE")
    out = {
        "kernel": "hpcla::" + KERNEL,
        "source": f"profiles/{rnd}_bench_pmc_FETCH_SIZE.csv + profiles/{rnd}_bench_pmc_WRITE_SIZE.csv (separate rocprofv3 "
                  f"--pmc passes over `python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-strong --no-extras --no-packed`; {nf} / {nw} launches)",
        "FETCH_SIZE_KB_mean": round(fetch_kb, 1), "WRITE_SIZE_KB_mean": round(write_kb, 1),
        "correction": CORRECTION, "block_order": ORDER_NOTE,
        "hbm_bytes_per_launch": hbm(fetch_kb, write_kb), "algorithmic_bytes_per_launch": B_ALG,
        "ratio_to_algorithmic": round(hbm(fetch_kb, write_kb) / B_ALG, 4),
        "calibration_note": "k_copy (known 1 207 762 944 B read with 16-B loads) read FETCH_SIZE 612 339 KB in the r01 "
                            "calibration pass (profiles/r01_calib_pmc_*.csv) -> factor 1.926 rather than 2",
    }
    # the same command under the other block orders the plan's measurement may choose (the traffic differs: neighbouring
    # XCD groups share x lines, which FETCH_SIZE counts once per L2 although the Infinity Cache serves the repeats)
    by_order = {"32": {"hbm_bytes_per_launch": out["hbm_bytes_per_launch"], "ratio_to_algorithmic": out["ratio_to_algorithmic"]}}
    for key, step in (("1", "headnat"), ("8", "head8"), ("64", "head64")):
        try:
            rdo = find(tag, f"pmc_{step}_FETCH_SIZE", "counter_collection.csv")
            wro = find(tag, f"pmc_{step}_WRITE_SIZE", "counter_collection.csv")
        except SystemExit:
            continue
        shutil.copy(rdo, os.path.join(prof, f"{rnd}_bench_pmc_order{key}_FETCH_SIZE.csv"))
        shutil.copy(wro, os.path.join(prof, f"{rnd}_bench_pmc_order{key}_WRITE_SIZE.csv"))
        fo, _ = counter_mean(rdo, "FETCH_SIZE")
        wo, _ = counter_mean(wro, "WRITE_SIZE")
        by_order[key] = {"hbm_bytes_per_launch": hbm(fo, wo), "ratio_to_algorithmic": round(hbm(fo, wo) / B_ALG, 4),
                         "source": f"profiles/{rnd}_bench_pmc_order{key}_FETCH_SIZE.csv + ..._WRITE_SIZE.csv (HPCLA_BLOCK_ORDER={'natural' if key == '1' else key})"}
    out["by_block_order"] = by_order
    wl = {}
    wl["poisson2d_spmv_int64"] = workload_single_kernel(
        tag, "i64", "spmv_rowgather_kernel<long, false, false, false>", rnd, prof, "headline matrix, Int64 indices STREAMED (HPCLA_NARROW_INDICES=0)", 1_744_568_328,
        "HPCLA_NARROW_INDICES=0 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-strong --no-extras --no-packed --index i64")
    n3, nnz3 = 16_777_216, 116_785_152
    b_spmv3 = 12 * nnz3 + 4 * (n3 + 1) + 8 * n3 + 8 * n3
    wl["poisson3d_cg_iteration"] = workload_cg(tag, rnd, prof, b_spmv3 + 96 * n3, b_spmv3 + 64 * n3)
    wl["poisson2d_spmm"] = workload_single_kernel(
        tag, "spmm2d", "spmm_rowblock_runs_kernel", rnd, prof, "5-point matrix 4096x2048 rows x 16 columns (run-tile kernel)", 2_684_207_108,
        "python3 bench.py --workload poisson2d_spmm --steps 5 --warmup 5")
    wl["sprand_spmm_b2e24"] = workload_single_kernel(
        tag, "sprand8", "spmm_rowblock_vec_kernel", rnd, prof, "sprand 2 097 152 rows x 29.8, B = 2^24 rows x 16 (config 5's gather set)", 3_122_571_200,
        "HPCLA_SPMM_COLS_MULT=8 python3 bench.py --workload sprand_spmm --steps 5 --warmup 5")
    wl["sprand_spmm_mall_sized"] = workload_single_kernel(
        tag, "sprand1", "spmm_rowblock_vec_kernel", rnd, prof, "sprand 2 097 152 rows x 29.8, B = 2 097 152 rows x 16 (Infinity-Cache-sized)", 1_295_259_584,
        "HPCLA_SPMM_COLS_MULT=1 python3 bench.py --workload sprand_spmm --steps 5 --warmup 5")
    # round 5: the unstructured matrix times ONE vector (bench.py's `spmv_same_matrix` records); absent passes leave the keys out
    for key, step, label, alg, cmd in (
            ("sprand_spmv_b2e24", "sprandv8", "sprand 2 097 152 rows x 29.8 times a vector of 2^24 entries (x = 134 MB)", 906_150_080,
             "HPCLA_SPRAND_SPMV=1 HPCLA_SPMM_COLS_MULT=8 python3 bench.py --workload sprand_spmm --steps 5 --warmup 5"),
            ("sprand_spmv_mall_sized", "sprandv1", "sprand 2 097 152 rows x 29.8 times a vector of 2^21 entries (x = 17 MB)", 791_943_104,
             "HPCLA_SPRAND_SPMV=1 HPCLA_SPMM_COLS_MULT=1 python3 bench.py --workload sprand_spmm --steps 5 --warmup 5")):
        try:
            wl[key] = workload_single_kernel(tag, step, KERNEL, rnd, prof, label, alg, cmd)
        except SystemExit as exc:
            print(f"(no passes for {key}: {exc})")
    out["workloads"] = wl
    with open(os.path.join(prof, "traffic_latest.json"), "w") as f:
        json.dump(out, f, indent=1)
    with open(os.path.join(prof, f"{rnd}_bench_kernel_stats.csv")) as f:
        for line in f.readlines()[:3]:
            print(line.rstrip()[:200])
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
