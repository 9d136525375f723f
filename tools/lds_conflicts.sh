# LDS bank-conflict cycles per kernel for a workload.   bash tools/lds_conflicts.sh <tag> <python script + args...>
set -e
R=$GRAFT_REPO_ROOT
tag=$1; shift
OUT=$R/gpurun_out/ldsc_$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --output-format csv -d /tmp/ldsc_$tag -o p -- python3 "$@" > $OUT/run.log 2>&1 || true
f=$(find /tmp/ldsc_$tag -name "*counter_collection.csv" | head -1)
python3 - "$f" > $OUT/summary.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'][:70]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'SQ_WAVE_CYCLES': n[k] += 1
rows = sorted(acc.items(), key=lambda kv: -kv[1]['SQ_LDS_BANK_CONFLICT'])
print(f"{'kernel':70s} {'calls':>6s} {'conflict cyc':>13s} {'idx active':>12s} {'ratio':>6s} {'conflict/wave_cyc*4':>10s}")
for k, v in rows[:25]:
    print(f"{k:70s} {n[k]:6d} {v['SQ_LDS_BANK_CONFLICT']:13.4g} {v['SQ_LDS_IDX_ACTIVE']:12.4g} {v['SQ_LDS_BANK_CONFLICT'] / max(v['SQ_LDS_IDX_ACTIVE'], 1):6.2f} {v['SQ_LDS_BANK_CONFLICT'] / max(4 * v['SQ_WAVE_CYCLES'], 1):10.4f}")
PY
cat $OUT/summary.txt
