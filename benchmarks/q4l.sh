set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python benchmarks/tune_spmv.py --dim 3 --size 512 --nz 64 --variants 100,105,123,130,132,133,134,136 --rounds 5 --reps 10 > gpurun_out/q4l_rg3d.log 2>&1; echo "rc=$?"; tail -12 gpurun_out/q4l_rg3d.log | head -11
timeout -k 10 300 python benchmarks/tune_spmv.py --variants 100,105,122,131,132,133,135,137 --rounds 5 --reps 10 > gpurun_out/q4l_rg2d.log 2>&1; echo "rc=$?"; tail -12 gpurun_out/q4l_rg2d.log | head -11
