set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in occ1 occ2; do
  if [ $v = occ2 ]; then export DIQT_LIB=$R/gpurun_libocc2.so; else unset DIQT_LIB; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa_$v -o a -- python3 $R/tools/unet3d_train_bench.py 64 32 8 > $R/gpurun_out/t24_$v.log 2>&1
  grep -i "ms" $R/gpurun_out/t24_$v.log | tail -2
  python3 - $v <<'P'
import csv,glob,sys
f=sorted(glob.glob(f'/tmp/pa_{sys.argv[1]}/**/*kernel_stats.csv',recursive=True))[-1]
for r in csv.DictReader(open(f)):
    if 'mqa_' in r['Name']: print(sys.argv[1], f"{int(r['Calls']):5d} x {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:70]}")
P
done
