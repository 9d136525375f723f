set -e
cd /tmp && export TMPDIR=/tmp
for n in 64 256; do
  export DIQT_RED_NBLK=$n
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gnp$n -o gn -- python3 $GRAFT_REPO_ROOT/tools/gn_bench.py 50 8 32 64 > /dev/null 2>&1
  echo "== RED_NBLK cap $n"
  python3 - <<PY
import csv,glob
f=glob.glob('/tmp/gnp$n/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'diqt' in r['Name']: print(r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
done
