#!/bin/bash
# r03e: production SpMM after the k = 16 specialisation vs the harness copy; SQ / TCC counters of the SpMM kernel, its
# ablations and the candidates that lost; then the PMC passes of every bench sub-record (run_r03_pmc.sh)
set -o pipefail
mkdir -p gpurun_out
step() { local t=$1 log=$2; shift 2
  timeout -k 10 "$t" "$@" > "$log" 2>&1; local rc=$?
  echo "[$(date +%T)] rc=$rc :: $*" | tee -a gpurun_out/r03e_steps.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out -- stopping: $*"; tail -20 "$log"; exit $rc; fi
  return $rc; }
step 400 gpurun_out/r03e_tune_spmm.log python benchmarks/tune_spmm.py --variants 100,0,9,17 --rounds 9
tail -7 gpurun_out/r03e_tune_spmm.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
V="100,0,1,2,4,13,17,19,20:4,9"
step 400 gpurun_out/r03e_pmc_sq.log rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES --output-format csv -d gpurun_out/r03e_pmc_sq -- python3 benchmarks/tune_spmm.py --variants $V --rounds 1 --reps 2
step 400 gpurun_out/r03e_pmc_tcc.log rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d gpurun_out/r03e_pmc_tcc -- python3 benchmarks/tune_spmm.py --variants $V --rounds 1 --reps 2
step 400 gpurun_out/r03e_pmc_ta.log rocprofv3 --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum --output-format csv -d gpurun_out/r03e_pmc_ta -- python3 benchmarks/tune_spmm.py --variants $V --rounds 1 --reps 2
step 400 gpurun_out/r03e_pmc_tcp.log rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d gpurun_out/r03e_pmc_tcp -- python3 benchmarks/tune_spmm.py --variants $V --rounds 1 --reps 2
step 400 gpurun_out/r03e_pmc_tlb.log rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum --output-format csv -d gpurun_out/r03e_pmc_tlb -- python3 benchmarks/tune_spmm.py --variants $V --rounds 1 --reps 2
python3 - <<'PY'
import csv, glob, collections
out = open("gpurun_out/r03e_spmm_counters.txt", "w")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("sq", "tcc", "ta", "tcp", "tlb"):
    for f in glob.glob(f"gpurun_out/r03e_pmc_{d}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "spmm" in k:
                acc[k[:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    out.write(k + "\n")
    for c, v in sorted(acc[k].items()):
        out.write(f"    {c:<28} launches {len(v):>3}  mean {sum(v)/len(v):.5g}\n")
out.close()
print(open("gpurun_out/r03e_spmm_counters.txt").read()[:3000])
PY
rm -rf gpurun_out/r03e_pmc_sq gpurun_out/r03e_pmc_tcc gpurun_out/r03e_pmc_ta gpurun_out/r03e_pmc_tcp gpurun_out/r03e_pmc_tlb
./benchmarks/run_r03_pmc.sh r03p
