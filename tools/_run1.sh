set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_lowprec.py -x -q -k "bf16_training_block" > gpurun_out/t26_tests.log 2>&1; echo "rc=$?" >> gpurun_out/t26_tests.log
tail -3 gpurun_out/t26_tests.log
for i in 1 2; do
DIQT_NO_TRAIN_FUSE=1 timeout -k 10 300 python tools/train_bf16_only.py 24 2>&1 | grep micro-step | sed 's/^/two nodes: /'
timeout -k 10 300 python tools/train_bf16_only.py 24 2>&1 | grep micro-step | sed 's/^/one node:  /'
done
