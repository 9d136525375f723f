set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_lowprec.py tests/test_gpu_kernels.py tests/test_gpu_trainer_trace.py tests/test_gpu_flow.py tests/test_gpu_unet.py tests/test_gpu_fullsize.py -x -q > gpurun_out/t35_tests.log 2>&1; echo "rc=$?" >> gpurun_out/t35_tests.log; tail -8 gpurun_out/t35_tests.log
for i in 1 2; do
DIQT_NO_TRAIN_HALF=1 timeout -k 10 300 python tools/train_bf16_only.py 24 2>&1 | grep micro-step | sed 's/^/fp32 between blocks: /'
timeout -k 10 300 python tools/train_bf16_only.py 24 2>&1 | grep micro-step | sed 's/^/16-bit between blocks: /'
done
timeout -k 10 300 python tools/fa_autocast_prof.py 16 2>&1 | grep "autocast"
