set -o pipefail
mkdir -p gpurun_out
bash tools/c5_prof.sh r04b > gpurun_out/t33.log 2>&1; tail -1 gpurun_out/t33.log
python - <<'P'
import csv
rows=list(csv.DictReader(open('gpurun_out/c5prof_r04b/c5s2_fp16_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('kernel ms per eval', tot/1e6/8)
for r in rows:
    if any(k in r['Name'] for k in ('concat','colreduce','gn_stats_final','se_pool','gn_act_fwd')): print(f"{float(r['Percentage']):6.2f}%  {int(r['Calls'])/8:6.1f} x {float(r['AverageNs'])/1e3:9.2f} us  {r['Name'][:90]}")
P
