set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_multirank.py -m gpu -q -x -k "bench_line" > gpurun_out/r06n_pytest.log 2>&1; rc=$?
tail -3 gpurun_out/r06n_pytest.log
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
./run_gpu_checks.sh r06n driverbench
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06n_bench.json'))
print(json.dumps(d['other_configs']['poisson2d_spmm'].get('odd_k'),indent=1)[:900])
print(d['configs_digest'])
PY
