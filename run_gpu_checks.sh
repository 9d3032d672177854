#!/bin/bash
# THE runner for gpurun calls (round 4: the 60-odd one-off benchmarks/run_r02*.sh / run_r03*.sh launch scripts were folded
# into the steps below or deleted; benchmarks/EXPERIMENTS.md keeps the table of what each measured, its log under profiles/
# and the commit that holds the variant).
# usage: ./run_gpu_checks.sh TAG [steps...]   steps: pytest smoke bench driverbench torchrun2 validate pmc_all cgtrace
#        rehearse2 rehearse4 rehearse6 arrow tune ... (see the case list)
# Stops at the first step that is killed by its timeout (never start a GPU step after a hang).
set -o pipefail
mkdir -p gpurun_out
TAG=${1:-r01}; shift
STEPS=${@:-pytest bench}
run() {  # run <timeout_s> <logfile> <cmd...>
  local t=$1 log=$2; shift 2
  timeout -k 10 "$t" "$@" > "$log" 2>&1
  local rc=$?
  echo "[$(date +%T)] rc=$rc :: $*" | tee -a gpurun_out/${TAG}_steps.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step timed out/killed -- stopping"; exit $rc; fi
  return 0
}
for st in $STEPS; do
  case $st in
    pytest) run 900 gpurun_out/${TAG}_pytest.log python -m pytest tests -m gpu -q -x; tail -5 gpurun_out/${TAG}_pytest.log;;
    smoke)  run 300 gpurun_out/${TAG}_smoke.log python -c "import __graft_entry__ as g; g.smoke()"; tail -1 gpurun_out/${TAG}_smoke.log;;
    driverbench)   # the bench line at the DRIVER's flags, with a short digest of every record
      run 560 gpurun_out/${TAG}_bench.log python bench.py --gpus 1 --steps 20 --warmup 5
      grep "^{" gpurun_out/${TAG}_bench.log | tail -1 > gpurun_out/${TAG}_bench.json
      python3 benchmarks/digest_bench_line.py gpurun_out/${TAG}_bench.json;;
    torchrun2)     # the N > 1 line under the driver's launcher, two ranks sharing the one GPU (rehearsal: timings mean nothing)
      HPCLA_ALLOW_SHARED_GPU=1 run 600 gpurun_out/${TAG}_torchrun2.log python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 ${REHEARSE_ARGS:---no-extras}
      grep "^{" gpurun_out/${TAG}_torchrun2.log | tail -1 > gpurun_out/${TAG}_torchrun2.json
      python3 benchmarks/digest_bench_line.py gpurun_out/${TAG}_torchrun2.json;;
    rehearse2|rehearse4|rehearse5|rehearse6)  # bench.py starting its own ranks on the shared GPU (6 = the box's limit on GPU processes)
      n=${st#rehearse}
      HPCLA_ALLOW_SHARED_GPU=1 run 900 gpurun_out/${TAG}_reh$n.log python bench.py --gpus $n ${REHEARSE_FLAGS:---steps 5 --warmup 2} ${REHEARSE_ARGS}
      grep "^{" gpurun_out/${TAG}_reh$n.log | tail -1 > gpurun_out/${TAG}_reh$n.json
      python3 benchmarks/digest_bench_line.py gpurun_out/${TAG}_reh$n.json;;
    validate) bash "$0" "$TAG" pytest smoke driverbench torchrun2;;
    arrow) run 600 gpurun_out/${TAG}_arrow.log python benchmarks/bench_arrow.py; tail -1 gpurun_out/${TAG}_arrow.log;;
    pmc_all)       # kernel stats + FETCH_SIZE / WRITE_SIZE passes (separate runs) of the headline and of every sub-record;
                   # then: python benchmarks/collect_profiles.py TAG rNN   (-> profiles/, traffic_latest.json)
      cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
      HEADARGS="--steps 10 --warmup 2 --no-cpu-baseline --no-strong --no-extras --no-packed"
      # the plan's block-order measurement (64 launches of the SAME kernel under four orders) would sit in every per-kernel
      # mean: the profiled runs take a fixed order instead, so every counted launch is a launch of the step
      export HPCLA_BLOCK_ORDER=${PMC_ORDER_2D:-32}
      export HPCLA_SPMM_BLOCK_ORDER=natural HPCLA_BENCH_SETTLE_MS=0     # likewise: no SpMM order measurement, no settle-time warm-up under the profiler
      run 300 gpurun_out/${TAG}_prof.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_prof -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-strong --no-extras --no-packed
      for c in FETCH_SIZE WRITE_SIZE; do
        run 300 gpurun_out/${TAG}_pmc_head_$c.log rocprofv3 --pmc $c --output-format csv -d gpurun_out/${TAG}_pmc_head_$c -- python3 bench.py $HEADARGS
        for o in natural 8 64; do
          HPCLA_BLOCK_ORDER=$o run 300 gpurun_out/${TAG}_pmc_head${o/natural/nat}_$c.log rocprofv3 --pmc $c --output-format csv -d gpurun_out/${TAG}_pmc_head${o/natural/nat}_$c -- python3 bench.py $HEADARGS
        done
        HPCLA_NARROW_INDICES=0 run 300 gpurun_out/${TAG}_pmc_i64_$c.log rocprofv3 --pmc $c --output-format csv -d gpurun_out/${TAG}_pmc_i64_$c -- python3 bench.py $HEADARGS --index i64
        HPCLA_BLOCK_ORDER=${PMC_ORDER_3D:-64} run 300 gpurun_out/${TAG}_pmc_cg_$c.log rocprofv3 --pmc $c --output-format csv -d gpurun_out/${TAG}_pmc_cg_$c -- python3 bench.py --workload poisson3d_cg --steps 10 --warmup 5
        run 300 gpurun_out/${TAG}_pmc_spmm2d_$c.log rocprofv3 --pmc $c --output-format csv -d gpurun_out/${TAG}_pmc_spmm2d_$c -- python3 bench.py --workload poisson2d_spmm --steps 5 --warmup 5
        HPCLA_SPMM_COLS_MULT=8 run 300 gpurun_out/${TAG}_pmc_sprand8_$c.log rocprofv3 --pmc $c --output-format csv -d gpurun_out/${TAG}_pmc_sprand8_$c -- python3 bench.py --workload sprand_spmm --steps 5 --warmup 5
        HPCLA_SPMM_COLS_MULT=1 run 300 gpurun_out/${TAG}_pmc_sprand1_$c.log rocprofv3 --pmc $c --output-format csv -d gpurun_out/${TAG}_pmc_sprand1_$c -- python3 bench.py --workload sprand_spmm --steps 5 --warmup 5
        HPCLA_SPRAND_SPMV=1 HPCLA_SPMM_COLS_MULT=8 run 300 gpurun_out/${TAG}_pmc_sprandv8_$c.log rocprofv3 --pmc $c --output-format csv -d gpurun_out/${TAG}_pmc_sprandv8_$c -- python3 bench.py --workload sprand_spmm --steps 5 --warmup 5
        HPCLA_SPRAND_SPMV=1 HPCLA_SPMM_COLS_MULT=1 run 300 gpurun_out/${TAG}_pmc_sprandv1_$c.log rocprofv3 --pmc $c --output-format csv -d gpurun_out/${TAG}_pmc_sprandv1_$c -- python3 bench.py --workload sprand_spmm --steps 5 --warmup 5
      done
      unset HPCLA_BLOCK_ORDER HPCLA_SPMM_BLOCK_ORDER HPCLA_BENCH_SETTLE_MS
      # keep what is merged back small: only the counter CSVs and the stats
      find gpurun_out/${TAG}_p* -type f ! -name '*counter_collection.csv' ! -name '*kernel_stats.csv' ! -name '*kernel_trace.csv' ! -name '*.log' -delete;;
    pmc_f32)       # HBM traffic of the Float32 kernels (csrc/f32.hip): separate FETCH_SIZE / WRITE_SIZE passes over the raw-ABI
                   # microbenchmark; then: python benchmarks/pmc_f32_table.py TAG rNN
      cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
      for c in FETCH_SIZE WRITE_SIZE; do
        run 300 gpurun_out/${TAG}_pmc_f32spmv_$c.log rocprofv3 --pmc $c --output-format csv -d gpurun_out/${TAG}_pmc_f32spmv_$c -- python3 benchmarks/bench_f32.py --only spmv2d --no-f64 --settle-ms 0 --reps 20
        run 300 gpurun_out/${TAG}_pmc_f32spmm_$c.log rocprofv3 --pmc $c --output-format csv -d gpurun_out/${TAG}_pmc_f32spmm_$c -- python3 benchmarks/bench_f32.py --only spmm --no-f64 --settle-ms 0 --reps 40
      done
      find gpurun_out/${TAG}_pmc_f32* -type f ! -name '*counter_collection.csv' ! -name '*.log' -delete;;
    pmc_spmv_study)  # round 4: SQ / TCP / TA / UTCL1 counters of the SpMV kernel in five contexts (2-D, 2-D + dot, 3-D, 3-D + dot,
                   # inside CG), one pass per small counter set (a pass that asks for more than a block's slots aborts the
                   # profiler: gpurun_out/r03e_pmc_ta.log); then: python benchmarks/pmc_spmv_table.py TAG profiles/rNN_...txt
      cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
      run 300 gpurun_out/${TAG}_pmcs_plain.log python3 benchmarks/pmc_spmv_cases.py --reps ${PMC_REPS:-6} --manifest gpurun_out/${TAG}_pmcs_manifest.json
      i=0
      while read -r set; do
        [ -z "$set" ] && continue
        i=$((i+1))
        run 150 gpurun_out/${TAG}_pmcs_$i.log rocprofv3 --pmc $set --output-format csv -d gpurun_out/${TAG}_pmcs_$i -- python3 benchmarks/pmc_spmv_cases.py --reps ${PMC_REPS:-6}
      done <<'SETS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_VMEM
SQ_LEVEL_WAVES SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVES SQ_INSTS_SMEM SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM_WR
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum
TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum
TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum
TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum
TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_SERIALIZATION_STALL_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
TCC_TAG_STALL_sum TCC_EA0_RDREQ_32B_sum
GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum
SETS
      find gpurun_out/${TAG}_pmcs_* -type f ! -name '*counter_collection.csv' ! -name '*.log' ! -name '*.json' -delete 2>/dev/null
      python3 benchmarks/pmc_spmv_table.py ${TAG} gpurun_out/${TAG}_pmcs_table.txt | tail -60;;
    cgtrace)       # kernel timeline of the CG iteration (fixed block order: see pmc_all)
      cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
      rm -rf gpurun_out/${TAG}_cgtrace
      HPCLA_BLOCK_ORDER=${PMC_ORDER_3D:-64} run 300 gpurun_out/${TAG}_cgtrace.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_cgtrace -- python3 bench.py --workload poisson3d_cg --steps 40 --warmup 8
      t=$(find gpurun_out/${TAG}_cgtrace -name '*kernel_trace.csv' | head -1)
      python benchmarks/trace_gaps.py "$t" "CG 512x512x64, one hpcla_cg_iterations call per timed window (${TAG})" > gpurun_out/${TAG}_cg_gaps.txt 2>&1
      st=$(find gpurun_out/${TAG}_cgtrace -name '*kernel_stats.csv' | head -1); [ -n "$st" ] && cp "$st" gpurun_out/${TAG}_cg_kernel_stats.csv
      grep "^{" gpurun_out/${TAG}_cgtrace.log | tail -1 > gpurun_out/${TAG}_cg_under_trace.json
      rm -rf gpurun_out/${TAG}_cgtrace; head -20 gpurun_out/${TAG}_cg_gaps.txt;;
    bench)  run 600 gpurun_out/${TAG}_bench.log python bench.py --steps 100 --warmup 10; tail -3 gpurun_out/${TAG}_bench.log;;
    tune)   run 600 gpurun_out/${TAG}_tune.log python benchmarks/tune_spmv.py ${TUNE_ARGS}; tail -25 gpurun_out/${TAG}_tune.log;;
    tune8k) run 900 gpurun_out/${TAG}_tune8k.log python benchmarks/tune_spmv.py --size 8192 --variants 16,100,101,102,20 --rounds 5 --reps 10; tail -8 gpurun_out/${TAG}_tune8k.log;;
    tune3dslab) run 900 gpurun_out/${TAG}_tune3dslab.log python benchmarks/tune_spmv.py --dim 3 --size 512 --nz 64 --variants ${TUNE3D_VARIANTS:-16,10,12,19,15,100} --rounds 5 --reps 10; tail -9 gpurun_out/${TAG}_tune3dslab.log;;
    tune3d) run 600 gpurun_out/${TAG}_tune3d.log python benchmarks/tune_spmv.py --dim 3 --size 256 ${TUNE_ARGS}; tail -25 gpurun_out/${TAG}_tune3d.log;;
    prof)
      cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
      run 600 gpurun_out/${TAG}_prof.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_prof -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-strong --no-extras --no-packed
      tail -3 gpurun_out/${TAG}_prof.log;;
    spgemm) run 600 gpurun_out/${TAG}_spgemm.log python benchmarks/bench_spgemm.py; tail -5 gpurun_out/${TAG}_spgemm.log;;
    halo)   run 600 gpurun_out/${TAG}_halo.log python benchmarks/bench_halo_overhead.py; tail -5 gpurun_out/${TAG}_halo.log;;
    halo_modes) for mode in serial overlap; do HPCLA_HALO_MODE=$mode run 600 gpurun_out/${TAG}_halo_${mode}.log python benchmarks/bench_halo_overhead.py; echo "$mode: $(grep -E 'plain split|halo \+ interior' gpurun_out/${TAG}_halo_${mode}.log | sed 's/host enqueue.*//' | tr '\n' ' ')"; done;;
    halo_ch) for ch in 4 16; do NCCL_MIN_P2P_NCHANNELS=$ch run 600 gpurun_out/${TAG}_halo_ch$ch.log python benchmarks/bench_halo_overhead.py; grep -E 'plain split|halo \+ interior|exchange only' gpurun_out/${TAG}_halo_ch$ch.log; done;;
    halotrace)
      run 600 gpurun_out/${TAG}_halotrace.log rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${TAG}_halotrace -- python3 benchmarks/bench_halo_overhead.py
      tail -2 gpurun_out/${TAG}_halotrace.log;;
    vecops) run 600 gpurun_out/${TAG}_vecops.log python benchmarks/bench_vecops.py; tail -8 gpurun_out/${TAG}_vecops.log;;
    cg)     run 900 gpurun_out/${TAG}_cg.log python bench.py --workload poisson3d_cg --steps 100 --warmup 10; tail -2 gpurun_out/${TAG}_cg.log;;
    cggraph) for g in 0 1; do HPCLA_CG_GRAPH=$g run 600 gpurun_out/${TAG}_cggraph$g.log python bench.py --workload poisson3d_cg --size 48 --steps 400 --warmup 10; tail -1 gpurun_out/${TAG}_cggraph$g.log; done;;
    i64) run 600 gpurun_out/${TAG}_bench_i64.log python bench.py --index i64 --no-cpu-baseline --no-packed; tail -1 gpurun_out/${TAG}_bench_i64.log;;
    cgsmall) run 900 gpurun_out/${TAG}_cgsmall.log python bench.py --workload poisson3d_cg --size 256 --steps 100 --warmup 10; tail -2 gpurun_out/${TAG}_cgsmall.log;;
    spmm2d) run 900 gpurun_out/${TAG}_spmm2d.log python bench.py --workload poisson2d_spmm --steps 30 --warmup 3; tail -1 gpurun_out/${TAG}_spmm2d.log;;
    rank4) run 600 gpurun_out/${TAG}_rank4.log python benchmarks/bench_rank4.py; tail -14 gpurun_out/${TAG}_rank4.log;;
    sprand_spmv) for m in 1 8; do HPCLA_SPRAND_SPMV=1 HPCLA_SPMM_COLS_MULT=$m run 900 gpurun_out/${TAG}_sprand_spmv$m.log python bench.py --workload sprand_spmm --steps 30 --warmup 3; tail -1 gpurun_out/${TAG}_sprand_spmv$m.log; done;;
    single) run 900 gpurun_out/${TAG}_single.log python benchmarks/bench_single_rank.py; tail -40 gpurun_out/${TAG}_single.log;;
    spmm8)  HPCLA_SPMM_COLS_MULT=8 run 900 gpurun_out/${TAG}_spmm8.log python bench.py --workload sprand_spmm --steps 20 --warmup 3; tail -2 gpurun_out/${TAG}_spmm8.log;;
    spmm)   run 900 gpurun_out/${TAG}_spmm.log python bench.py --workload sprand_spmm --steps 20 --warmup 3; tail -2 gpurun_out/${TAG}_spmm.log;;
    pmc_sq)
      run 600 gpurun_out/${TAG}_pmc_sq.log rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/${TAG}_pmc_sq -- python3 benchmarks/tune_spmv.py --variants 100,5,20,0 --rounds 1 --reps 2
      tail -2 gpurun_out/${TAG}_pmc_sq.log;;
    pmc_spmm)
      run 600 gpurun_out/${TAG}_pmc_spmm_sq.log rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES --output-format csv -d gpurun_out/${TAG}_pmc_spmm_sq -- python3 bench.py --workload poisson2d_spmm --steps 5 --warmup 2
      run 600 gpurun_out/${TAG}_pmc_spmm_tcc.log rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d gpurun_out/${TAG}_pmc_spmm_tcc -- python3 bench.py --workload poisson2d_spmm --steps 5 --warmup 2
      run 600 gpurun_out/${TAG}_pmc_spmm_fs.log rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${TAG}_pmc_spmm_fs -- python3 bench.py --workload poisson2d_spmm --steps 5 --warmup 2
      tail -1 gpurun_out/${TAG}_pmc_spmm_fs.log;;
    profcg)
      run 600 gpurun_out/${TAG}_profcg.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_profcg -- python3 bench.py --workload poisson3d_cg --steps 50 --warmup 8
      tail -2 gpurun_out/${TAG}_profcg.log;;
    pmc_tcc)
      run 600 gpurun_out/${TAG}_pmc_tcc.log rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d gpurun_out/${TAG}_pmc_tcc -- python3 benchmarks/tune_spmv.py --variants 100,5,20,0 --rounds 1 --reps 2
      tail -2 gpurun_out/${TAG}_pmc_tcc.log;;
    pmc_sq2)
      run 600 gpurun_out/${TAG}_pmc_sq2.log rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${TAG}_pmc_sq2 -- python3 benchmarks/tune_spmv.py --variants 100,5,20,0 --rounds 1 --reps 2
      tail -2 gpurun_out/${TAG}_pmc_sq2.log;;
    pmc_tune_rd)
      run 600 gpurun_out/${TAG}_pmc_tune_rd.log rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${TAG}_pmc_tune_rd -- python3 benchmarks/tune_spmv.py --variants 20,100,101 --rounds 1 --reps 3
      tail -2 gpurun_out/${TAG}_pmc_tune_rd.log;;
    pmc_tune_wr)
      run 600 gpurun_out/${TAG}_pmc_tune_wr.log rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${TAG}_pmc_tune_wr -- python3 benchmarks/tune_spmv.py --variants 20,100,101 --rounds 1 --reps 3
      tail -2 gpurun_out/${TAG}_pmc_tune_wr.log;;
    pmc_rd)
      run 600 gpurun_out/${TAG}_pmc_rd.log rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${TAG}_pmc_rd -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-strong --no-extras --no-packed
      tail -2 gpurun_out/${TAG}_pmc_rd.log;;
    pmc_wr)
      run 600 gpurun_out/${TAG}_pmc_wr.log rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${TAG}_pmc_wr -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-strong --no-extras --no-packed
      tail -2 gpurun_out/${TAG}_pmc_wr.log;;
  esac
done
