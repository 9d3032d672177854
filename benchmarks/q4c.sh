set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python benchmarks/tune_spmv.py --dim 3 --size 512 --nz 64 --variants 100,16,26,90,91,92,93,94,95,96,97,98,99 --rounds 5 --reps 10 > gpurun_out/q4c_rowg3d.log 2>&1; echo "rc=$?"; tail -18 gpurun_out/q4c_rowg3d.log | head -16
timeout -k 10 300 python benchmarks/tune_spmv.py --variants 100,16,25,90,91,92,93,94,95,96,97,98,99 --rounds 5 --reps 10 > gpurun_out/q4c_rowg2d.log 2>&1; echo "rc=$?"; tail -18 gpurun_out/q4c_rowg2d.log | head -16
timeout -k 10 300 python benchmarks/tune_spmm.py --variants 100,0,30,31,32,33 --rounds 5 --reps 10 > gpurun_out/q4c_spmm_runs.log 2>&1; echo "rc=$?"; tail -12 gpurun_out/q4c_spmm_runs.log | head -11
