set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_unet.py -x -q > gpurun_out/t30_tests.log 2>&1; echo "rc=$?" >> gpurun_out/t30_tests.log; tail -3 gpurun_out/t30_tests.log
timeout -k 10 600 python bench.py --config C4 > gpurun_out/t30_c4.json 2> gpurun_out/t30_c4.err; python - <<'P'
import json
d=json.loads(open('gpurun_out/t30_c4.json').read().strip().splitlines()[-1]); print('C4', d['value'], d['ms_per_step'], d['roofline']['frac'], d['autocast_fp16'])
P
