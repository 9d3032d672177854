#!/bin/bash
# r03a: first GPU call of round 3 -- the GPU suite on the new code, the bench line, CG in its three host forms with
# kernel traces of the eager and the graph-replayed loop, and the SpMM ablation / candidate harness.
set -o pipefail
mkdir -p gpurun_out
step() { local t=$1 log=$2; shift 2
  timeout -k 10 "$t" "$@" > "$log" 2>&1; local rc=$?
  echo "[$(date +%T)] rc=$rc :: $*" | tee -a gpurun_out/r03a_steps.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out -- stopping: $*"; tail -20 "$log"; exit $rc; fi
  return $rc; }
step 1000 gpurun_out/r03a_pytest.log python -m pytest tests -m gpu -q -x
tail -5 gpurun_out/r03a_pytest.log
step 500 gpurun_out/r03a_bench.log python bench.py --steps 20 --warmup 5 || tail -20 gpurun_out/r03a_bench.log
tail -c 3000 gpurun_out/r03a_bench.log | grep -v "^{" | tail -30
for form in native python graph; do
  case $form in native) E="";; python) E="HPCLA_CG_PYTHON_LOOP=1";; graph) E="HPCLA_CG_GRAPH=1";; esac
  env $E timeout -k 10 300 python bench.py --workload poisson3d_cg --steps 100 --warmup 8 > gpurun_out/r03a_cg_$form.log 2>&1
  echo "cg $form: $(tail -1 gpurun_out/r03a_cg_$form.log | python -c 'import sys,json; r=json.loads(sys.stdin.read()); print("wall", r["wall_ms_per_iter"], "device", r["device_ms_per_iter"], "host enqueue", r["host_enqueue_ms_per_iter"], "frac", r["roofline"]["frac"])' 2>&1 | tail -1)" | tee -a gpurun_out/r03a_cg_forms.log
done
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
step 400 gpurun_out/r03a_trace_eager.log rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r03a_trace_eager -- python3 bench.py --workload poisson3d_cg --steps 40 --warmup 8
export HPCLA_CG_GRAPH=1
step 400 gpurun_out/r03a_trace_graph.log rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r03a_trace_graph -- python3 bench.py --workload poisson3d_cg --steps 40 --warmup 8
unset HPCLA_CG_GRAPH
for f in eager graph; do
  t=$(find gpurun_out/r03a_trace_$f -name '*kernel_trace.csv' | head -1)
  [ -n "$t" ] && python benchmarks/trace_gaps.py "$t" "CG 512x512x64, $f" > gpurun_out/r03a_cg_gaps_$f.txt 2>&1
  rm -rf gpurun_out/r03a_trace_$f
done
cat gpurun_out/r03a_cg_gaps_*.txt | head -60
step 600 gpurun_out/r03a_tune_spmm.log python benchmarks/tune_spmm.py
tail -22 gpurun_out/r03a_tune_spmm.log
step 600 gpurun_out/r03a_tune_spmm_sprand.log python benchmarks/tune_spmm.py --workload sprand --variants 100,0,1,2,4,7:2048,8:2048,5 --rounds 5 --reps 5
tail -14 gpurun_out/r03a_tune_spmm_sprand.log
