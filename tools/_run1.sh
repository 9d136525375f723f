set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1150 python -X faulthandler -m pytest tests -x -q -m gpu > gpurun_out/t48_gpu_suite.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/t48_gpu_suite.log
