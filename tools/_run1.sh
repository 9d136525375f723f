set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout -k 10 900 python -X faulthandler -m pytest tests/test_gpu_train_graph.py tests/test_gpu_lowprec.py tests/test_gpu_rccl.py tests/test_gpu_ddp_trace.py tests/test_gpu_trainer_trace.py -q -m gpu -x > gpurun_out/t55_test.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/t55_test.log | cut -c1-300
for i in 1 2; do timeout -k 10 300 python tools/train_bf16_only.py 32 > gpurun_out/t55_tb$i.log 2>&1; tail -1 gpurun_out/t55_tb$i.log; done
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ptb -o tb -- python3 $R/tools/train_bf16_only.py 16 > $R/gpurun_out/t55_tb.log 2>&1; cd $R; grep "micro-step" gpurun_out/t55_tb.log
python - <<'P'
import csv,glob,shutil
f=sorted(glob.glob('/tmp/ptb/**/*kernel_stats.csv',recursive=True))[-1]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows); calls=sum(int(r['Calls']) for r in rows)
print(f"kernel time {tot/1e6/24:.2f} ms per micro-step over 24 steps, {calls/24:.0f} launches per step")
for r in rows:
    if 'pack' in r['Name']: print(f"{float(r['Percentage']):6.2f}%  {int(r['Calls'])/24:7.1f} x {float(r['AverageNs'])/1e3:9.2f} us  {r['Name'][:105]}")
shutil.copy(f,'gpurun_out/r04_train_bf16_kernel_stats.csv')
P
