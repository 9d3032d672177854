set -o pipefail
mkdir -p gpurun_out
REHEARSE_ARGS="" ./run_gpu_checks.sh r06p torchrun2
REHEARSE_ARGS="" ./run_gpu_checks.sh r06p rehearse4
