#!/bin/bash
# r03: memory-side read latency (TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ) and translation-cache activity of the headline kernel
# under the natural and the grouped block orders (separate rocprofv3 --pmc passes, counters of TCC / GRBM only)
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
HEAD="--steps 10 --warmup 2 --no-cpu-baseline --no-strong --no-extras --no-packed"
for ord in natural 8 32 64; do
  for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum" "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE" "TCC_TAG_STALL_sum TCC_REQ_sum" "TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $set | tr ' ' '+')
    HPCLA_BLOCK_ORDER=$ord timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d gpurun_out/r03lat_${ord}_$tag -- python3 bench.py $HEAD > gpurun_out/r03lat_${ord}_$tag.log 2>&1
    echo "order=$ord set=$tag rc=$?"
  done
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/r03lat_*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if "spmv_rowblock_quad_kernel<int, false, false>" in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        print(d.split("/")[1], {k: round(sum(v[-10:]) / len(v[-10:]), 1) for k, v in acc.items()}, "launches", {k: len(v) for k, v in acc.items()})
PY
find gpurun_out/r03lat_* -type f ! -name '*counter_collection.csv' ! -name '*.log' -delete
true
