set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/t14_gpu_tests.log 2>&1; echo "rc=$?" >> gpurun_out/t14_gpu_tests.log
tail -22 gpurun_out/t14_gpu_tests.log
