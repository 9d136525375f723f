set -o pipefail
mkdir -p gpurun_out
for i in 1 2; do
timeout -k 10 300 python tools/train_bf16_only.py 32 > gpurun_out/t50_a$i.log 2>&1; echo "half rc=$?"; tail -1 gpurun_out/t50_a$i.log
DIQT_NO_DACT_HALF=1 timeout -k 10 300 python tools/train_bf16_only.py 32 > gpurun_out/t50_b$i.log 2>&1; echo "fp32 dact rc=$?"; tail -1 gpurun_out/t50_b$i.log
done
