set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_lowprec.py tests/test_gpu_conv_f9h.py tests/test_gpu_family_b.py tests/test_gpu_fullsize.py tests/test_gpu_flow.py tests/test_gpu_unet.py -x -q > gpurun_out/t16_tests.log 2>&1; echo "rc=$?" >> gpurun_out/t16_tests.log
tail -5 gpurun_out/t16_tests.log
