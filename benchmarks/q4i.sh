set -o pipefail
mkdir -p gpurun_out
for i in 1 2; do
for r in 1 0; do
HPCLA_SPMM_RUNS=$r timeout -k 10 200 python bench.py --workload poisson2d_spmm --steps 50 --warmup 5 > gpurun_out/q4i_spmm2d_r${r}_$i.log 2>&1; echo "runs=$r rc=$?"; tail -1 gpurun_out/q4i_spmm2d_r${r}_$i.log | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['device_ms_per_step'], r['roofline'].get('run_tiles'))"
done; done
timeout -k 10 300 python benchmarks/tune_spmm.py --variants 100,102,103 --rounds 5 --reps 50 > gpurun_out/q4i_spmm_runs.log 2>&1; echo "rc=$?"; tail -7 gpurun_out/q4i_spmm_runs.log | head -6
