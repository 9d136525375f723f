set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/tattn_${1:-x}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/tattn_bench.py > $OUT/bench.log 2>&1
python3 $R/tools/tattn_bench.py 128 64 1024 8 20 >> $OUT/bench.log 2>&1
python3 $R/tools/tattn_bench.py 64 32 1024 8 20 >> $OUT/bench.log 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_$i -o p -- python3 $R/tools/tattn_bench.py 64 64 4096 8 3 > $OUT/pmc_$i.log 2>&1 || true
  f=$(find /tmp/pmc_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" >> $OUT/pmc.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if 'temporal_attn' in r['Kernel_Name']:
        a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
for k, (v, n) in acc.items():
    print(f"{k}: {v / n:.4g} per launch ({n} rows)")
PY
done
cat $OUT/bench.log | grep temporal; cat $OUT/pmc.txt
