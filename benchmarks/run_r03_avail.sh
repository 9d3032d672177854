#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 120 rocprofv3 --list-avail > gpurun_out/r03_avail.txt 2>&1; echo "rc=$?"
grep -i -c "TCC_" gpurun_out/r03_avail.txt
grep -i -o "TCC_EA0_[A-Z0-9_]*\|TCC_[A-Z0-9_]*DRAM[A-Z0-9_]*\|TCC_[A-Z0-9_]*MALL[A-Z0-9_]*" gpurun_out/r03_avail.txt | sort -u | head -80
true
