#!/bin/bash
# PMC passes (FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 runs, no tracing flags beside --pmc) for the headline and
# for every sub-record of the bench line, plus the kernel-trace stats of the headline command.  Collected into
# profiles/ by benchmarks/collect_profiles.py.   usage: ./benchmarks/run_r03_pmc.sh TAG
set -o pipefail
TAG=${1:-r03p}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
step() { local t=$1 log=$2; shift 2
  timeout -k 10 "$t" "$@" > "$log" 2>&1; local rc=$?
  echo "[$(date +%T)] rc=$rc :: $*" | tee -a gpurun_out/${TAG}_steps.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out -- stopping: $*"; tail -20 "$log"; exit $rc; fi
  return $rc; }
HEAD="--steps 10 --warmup 2 --no-cpu-baseline --no-strong --no-extras --no-packed"
# The plan's block-order measurement (64 launches of the SAME kernel under four orders) would sit in every per-kernel mean:
# the profiled runs take the order the plans choose on these matrices (groups of 32 row blocks in 2-D, 64 on the 3-D slab;
# profiles/r03_spmv_xcd_group_order.log) as a fixed setting instead, so every counted launch is a launch of the step.
export HPCLA_BLOCK_ORDER=32
step 300 gpurun_out/${TAG}_prof.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_prof -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-strong --no-extras --no-packed
for c in FETCH_SIZE WRITE_SIZE; do
  step 300 gpurun_out/${TAG}_pmc_head_$c.log rocprofv3 --pmc $c --output-format csv -d gpurun_out/${TAG}_pmc_head_$c -- python3 bench.py $HEAD
  HPCLA_BLOCK_ORDER=natural step 300 gpurun_out/${TAG}_pmc_headnat_$c.log rocprofv3 --pmc $c --output-format csv -d gpurun_out/${TAG}_pmc_headnat_$c -- python3 bench.py $HEAD
  HPCLA_BLOCK_ORDER=8 step 300 gpurun_out/${TAG}_pmc_head8_$c.log rocprofv3 --pmc $c --output-format csv -d gpurun_out/${TAG}_pmc_head8_$c -- python3 bench.py $HEAD
  HPCLA_BLOCK_ORDER=64 step 300 gpurun_out/${TAG}_pmc_head64_$c.log rocprofv3 --pmc $c --output-format csv -d gpurun_out/${TAG}_pmc_head64_$c -- python3 bench.py $HEAD
  step 300 gpurun_out/${TAG}_pmc_i64_$c.log rocprofv3 --pmc $c --output-format csv -d gpurun_out/${TAG}_pmc_i64_$c -- python3 bench.py $HEAD --index i64
  export HPCLA_BLOCK_ORDER=64
  step 300 gpurun_out/${TAG}_pmc_cg_$c.log rocprofv3 --pmc $c --output-format csv -d gpurun_out/${TAG}_pmc_cg_$c -- python3 bench.py --workload poisson3d_cg --steps 10 --warmup 5
  export HPCLA_BLOCK_ORDER=32
  step 300 gpurun_out/${TAG}_pmc_spmm2d_$c.log rocprofv3 --pmc $c --output-format csv -d gpurun_out/${TAG}_pmc_spmm2d_$c -- python3 bench.py --workload poisson2d_spmm --steps 5 --warmup 5
  export HPCLA_SPMM_COLS_MULT=8
  step 300 gpurun_out/${TAG}_pmc_sprand8_$c.log rocprofv3 --pmc $c --output-format csv -d gpurun_out/${TAG}_pmc_sprand8_$c -- python3 bench.py --workload sprand_spmm --steps 5 --warmup 5
  export HPCLA_SPMM_COLS_MULT=1
  step 300 gpurun_out/${TAG}_pmc_sprand1_$c.log rocprofv3 --pmc $c --output-format csv -d gpurun_out/${TAG}_pmc_sprand1_$c -- python3 bench.py --workload sprand_spmm --steps 5 --warmup 5
  unset HPCLA_SPMM_COLS_MULT
done
# keep what is merged back small: only the counter CSVs and the stats
find gpurun_out/${TAG}_p* -type f ! -name '*counter_collection.csv' ! -name '*kernel_stats.csv' ! -name '*kernel_trace.csv' -delete
ls gpurun_out | grep ${TAG} | head -40
