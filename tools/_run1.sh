set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_train_graph.py -x -q -m gpu > gpurun_out/t39_test.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/t39_test.log
timeout -k 10 300 python tools/train_bf16_only.py 32 > gpurun_out/t39_tb.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/t39_tb.log
DIQT_TRAIN_GRAPH=0 timeout -k 10 300 python tools/train_bf16_only.py 32 > gpurun_out/t39_tb0.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/t39_tb0.log
