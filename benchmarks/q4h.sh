set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python benchmarks/tune_spmm.py --variants 100,30,102,103,104 --rounds 5 --reps 10 > gpurun_out/q4h_spmm_runs.log 2>&1; echo "rc=$?"; tail -9 gpurun_out/q4h_spmm_runs.log | head -8
