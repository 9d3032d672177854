set -o pipefail
mkdir -p gpurun_out
./run_gpu_checks.sh r06m pytest smoke driverbench
