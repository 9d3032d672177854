set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python benchmarks/tune_spmv.py --dim 3 --size 512 --nz 64 --variants 100,93,98,112,120,121,122,123,124,125,126 --rounds 5 --reps 10 > gpurun_out/q4e_rowg3d.log 2>&1; echo "rc=$?"; tail -15 gpurun_out/q4e_rowg3d.log | head -14
timeout -k 10 300 python benchmarks/tune_spmv.py --variants 100,93,98,112,120,121,122,123,124,125,126 --rounds 5 --reps 10 > gpurun_out/q4e_rowg2d.log 2>&1; echo "rc=$?"; tail -15 gpurun_out/q4e_rowg2d.log | head -14
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "run_tiles or block_order or narrowing or int64" > gpurun_out/q4e_pytest.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/q4e_pytest.log
timeout -k 10 300 python bench.py --workload poisson2d_spmm --steps 30 --warmup 5 > gpurun_out/q4e_spmm2d.log 2>&1; echo "rc=$?"; tail -1 gpurun_out/q4e_spmm2d.log | cut -c1-900
timeout -k 10 300 HPCLA_SPMM_COLS_MULT=8 python bench.py --workload sprand_spmm --steps 20 --warmup 5 > gpurun_out/q4e_sprand.log 2>&1; echo "rc=$?"; tail -1 gpurun_out/q4e_sprand.log | cut -c1-900
