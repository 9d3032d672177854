#!/usr/bin/env python3
"""Excerpt of a rocprofv3 --kernel-trace CSV of benchmarks/bench_halo_overhead.py: for each halo mode, a few
consecutive steps as (kernel, queue, start, end, grid), times relative to the excerpt's first kernel.
usage: python benchmarks/trace_excerpt.py TRACE.csv > profiles/rNN_halo_step_trace_excerpt.txt"""
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1], newline="")))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    name = lambda r: r["Kernel_Name"]
    fused = [i for i, r in enumerate(rows) if ("spmv_rowgather_kernel<int, true, true" in name(r) or "spmv_rowblock_quad_kernel<int, true, true>" in name(r))]
    rccl = [i for i, r in enumerate(rows) if "rcclGenericKernel" in name(r) or "ncclDevKernel" in name(r)]
    out = []

    def dump(title, first, count):
        t0 = int(rows[first]["Start_Timestamp"])
        out.append(f"# {title}")
        for r in rows[first:first + count]:
            grid = r.get("Grid_Size_X") or r.get("Grid_Size") or "?"
            out.append(f"{name(r)[:64]:<64} q={r.get('Queue_Id', '?'):<3} start {(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us"
                       f"  end {(int(r['End_Timestamp']) - t0) / 1e3:9.1f} us  grid {grid}")

    if fused:
        mid = fused[len(fused) // 2]
        dump("push transport: ONE launch per step -- leading workgroups push, boundary workgroups wait in-kernel", mid, 6)
    main_q = rows[fused[0]]["Queue_Id"] if fused else None
    serial = [i for i in rccl if rows[i]["Queue_Id"] == main_q]          # exchange on the caller's stream
    overlap = [i for i in rccl if rows[i]["Queue_Id"] != main_q]         # exchange on the plan's side stream
    if serial:
        dump("RCCL serial ordering: (pack,) send/recv kernel, then ONE launch over all row blocks, same stream",
             serial[len(serial) // 2], 8)
    # full steps only (the harness also times the exchange alone): an exchange kernel that runs NEXT TO row blocks
    def beside_rows(i):
        a, b = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
        return any("spmv_row" in name(r) and int(r["Start_Timestamp"]) < b and int(r["End_Timestamp"]) > a
                   for r in rows[max(i - 4, 0):i + 5])
    overlap = [i for i in overlap if beside_rows(i)]
    if overlap:
        dump("RCCL overlap ordering: side-stream (pack,) send/recv next to the interior row blocks, boundary blocks behind it",
             overlap[len(overlap) // 3], 10)
    print("\n".join(out))


if __name__ == "__main__":
    main()
