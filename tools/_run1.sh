set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_lowprec.py tests/test_gpu_kernels.py tests/test_gpu_trainer_trace.py tests/test_gpu_flow.py tests/test_gpu_unet.py -x -q > gpurun_out/t34_tests.log 2>&1; echo "rc=$?" >> gpurun_out/t34_tests.log; tail -8 gpurun_out/t34_tests.log
