#!/bin/bash
# profiling passes, then the validation of the tree (GPU suite, smoke, bench at the driver's flags, 2-rank torchrun)
./benchmarks/run_r03_pmc.sh ${1:-r03p3} && ./benchmarks/run_r03_validate.sh
