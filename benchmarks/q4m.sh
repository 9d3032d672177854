set -o pipefail
bash run_gpu_checks.sh q4m pmc_spmv_study cgtrace
