#!/bin/bash
# r02f: SpMM XCD-grouped block order + PMC counters of the new kernel
set -o pipefail
mkdir -p gpurun_out
step() { local t=$1 log=$2; shift 2
  timeout -k 10 "$t" "$@" > "$log" 2>&1; local rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out -- stopping: $*"; tail -20 "$log"; exit $rc; fi
  return $rc; }
OUT=gpurun_out/r02f_spmm_xcd.log; : > $OUT
for g in 0 16 64 128 256 512 1024 2048; do
  export HPCLA_SPMM_XCD_GROUP=$g
  step 300 gpurun_out/r02f_tmp.log python bench.py --workload poisson2d_spmm --steps 30 --warmup 3 || { tail -5 gpurun_out/r02f_tmp.log; exit 1; }
  echo "XCD_GROUP=$g poisson2d_spmm: $(tail -1 gpurun_out/r02f_tmp.log | python -c 'import sys,json; r=json.loads(sys.stdin.read()); print(r["ms_per_step"], "ms", r["value"], "GFLOP/s frac", r["roofline"]["frac"])')" | tee -a $OUT
done
for g in 0 256; do
  export HPCLA_SPMM_XCD_GROUP=$g
  HPCLA_SPMM_COLS_MULT=8 step 300 gpurun_out/r02f_tmp.log python bench.py --workload sprand_spmm --steps 20 --warmup 3 || { tail -5 gpurun_out/r02f_tmp.log; exit 1; }
  echo "XCD_GROUP=$g sprand_spmm x8:  $(tail -1 gpurun_out/r02f_tmp.log | python -c 'import sys,json; r=json.loads(sys.stdin.read()); print(r["ms_per_step"], "ms frac", r["roofline"]["frac"], "gather GB/s", r["roofline"]["gather_gbs"])')" | tee -a $OUT
done
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for g in 0 512; do
  export HPCLA_SPMM_XCD_GROUP=$g
  step 600 gpurun_out/r02f_pmc_sq_g$g.log rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES --output-format csv -d gpurun_out/r02f_pmc_sq_g$g -- python3 bench.py --workload poisson2d_spmm --steps 5 --warmup 2
  step 600 gpurun_out/r02f_pmc_tcc_g$g.log rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d gpurun_out/r02f_pmc_tcc_g$g -- python3 bench.py --workload poisson2d_spmm --steps 5 --warmup 2
  step 600 gpurun_out/r02f_pmc_fs_g$g.log rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r02f_pmc_fs_g$g -- python3 bench.py --workload poisson2d_spmm --steps 5 --warmup 2
  step 600 gpurun_out/r02f_pmc_ws_g$g.log rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r02f_pmc_ws_g$g -- python3 bench.py --workload poisson2d_spmm --steps 5 --warmup 2
done
python3 - <<'PY'
import csv, glob, collections
for g in (0, 512):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/r02f_pmc_*_g{g}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "spmm_rowblock_vec" in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    print(f"XCD_GROUP={g}: " + "  ".join(f"{k}={sum(v)/len(v):.4g}(n={len(v)})" for k, v in sorted(acc.items())))
PY
