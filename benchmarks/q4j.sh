set -o pipefail
mkdir -p gpurun_out
for w in 5 100; do
for r in 1 0; do
HPCLA_SPMM_RUNS=$r timeout -k 10 200 python bench.py --workload poisson2d_spmm --steps 50 --warmup $w > gpurun_out/q4j_spmm2d_r${r}_w$w.log 2>&1; echo "runs=$r warmup=$w rc=$?"; tail -1 gpurun_out/q4j_spmm2d_r${r}_w$w.log | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['device_ms_per_step'])"
done; done
