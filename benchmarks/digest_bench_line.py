#!/usr/bin/env python3
"""Print a short digest of one bench.py JSON line (file argument): headline, every sub-record's time / roofline
fraction / traffic, the strong-scaling record and the budget -- what a gpurun call's tail should show."""
import json
import sys


def main(path):
    try:
        r = json.loads(open(path).read())
    except Exception as exc:
        print(f"digest: {path}: no JSON line ({type(exc).__name__}: {exc})")
        return 1
    rf = r.get("roofline", {})
    print("headline", r.get("value"), r.get("unit"), "ms/step", r.get("ms_per_step"), "frac", rf.get("frac"),
          "launch_ms", rf.get("launch_ms_timed_region"), "order", rf.get("block_order_group"), "traffic", rf.get("traffic"),
          "n_gpus", r.get("n_gpus"), "verified", r.get("verified_vs_closed_form"))
    for k, v in (r.get("other_configs") or {}).items():
        if not isinstance(v, dict):
            continue
        vr = v.get("roofline", {}) or {}
        print(" ", k, "ms", v.get("ms_per_step", v.get("wall_ms_per_iter")), "dev", v.get("device_ms_per_step", v.get("device_ms_per_iter")),
              "frac", vr.get("frac"), "traffic", vr.get("traffic"), {kk: v[kk] for kk in ("narrowed", "skipped", "error", "block_order_group") if kk in v})
    ss = r.get("strong_scaling")
    if isinstance(ss, dict):
        print("  strong_scaling", {k: ss.get(k) for k in ("ms_per_step", "gflops", "speedup_vs_n1", "n1_ms_per_step_rank0_alone", "skipped", "error") if k in ss})
    for k in ("strong_scaling_4096", "strong_scaling_speedup_vs_n1", "strong_scaling_4096_speedup_vs_n1", "halo_mode", "peer_windows", "budget"):
        if k in r:
            print(" ", k, r[k])
    cb = r.get("cpu_baseline")
    if cb:
        print("  cpu_baseline", {k: cb.get(k) for k in ("value", "unit", "cores", "kind")})
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
