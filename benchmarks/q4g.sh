set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python benchmarks/tune_spmm.py --variants 100,30,102 --rounds 5 --reps 10 > gpurun_out/q4g_spmm_runs.log 2>&1; echo "rc=$?"; tail -7 gpurun_out/q4g_spmm_runs.log | head -6
timeout -k 10 300 python benchmarks/tune_spmv.py --dim 3 --size 512 --nz 64 --variants 100,104,105,106,98,123 --rounds 5 --reps 10 > gpurun_out/q4g_rg3d.log 2>&1; echo "rc=$?"; tail -10 gpurun_out/q4g_rg3d.log | head -9
timeout -k 10 300 python benchmarks/tune_spmv.py --variants 100,104,105,106,93,112 --rounds 5 --reps 10 > gpurun_out/q4g_rg2d.log 2>&1; echo "rc=$?"; tail -10 gpurun_out/q4g_rg2d.log | head -9
timeout -k 10 300 python benchmarks/tune_spmv.py --size 8192 --variants 100,105,93 --rounds 4 --reps 6 > gpurun_out/q4g_rg8k.log 2>&1; echo "rc=$?"; tail -7 gpurun_out/q4g_rg8k.log | head -6
bash run_gpu_checks.sh q4g pytest driverbench
