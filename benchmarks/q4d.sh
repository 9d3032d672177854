set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python benchmarks/tune_spmv.py --dim 3 --size 512 --nz 64 --variants 100,26,93,97,98,110,111,112,113,114,115,116,117,95,96 --rounds 5 --reps 10 > gpurun_out/q4d_rowg3d.log 2>&1; echo "rc=$?"; tail -19 gpurun_out/q4d_rowg3d.log | head -18
timeout -k 10 300 python benchmarks/tune_spmv.py --variants 100,25,93,97,98,110,111,112,113,114,115,116,117,95,96 --rounds 5 --reps 10 > gpurun_out/q4d_rowg2d.log 2>&1; echo "rc=$?"; tail -19 gpurun_out/q4d_rowg2d.log | head -18
timeout -k 10 300 python benchmarks/tune_spmv.py --size 8192 --variants 100,26,93,97,98,111,112,113,96 --rounds 4 --reps 6 > gpurun_out/q4d_rowg8k.log 2>&1; echo "rc=$?"; tail -13 gpurun_out/q4d_rowg8k.log | head -12
