set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/wgradh_${1:-x}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/wpmc_$i -o p -- python3 $R/tools/wgradh_bench.py 8 32 64 64 > $OUT/pmc_$i.log 2>&1 || true
  f=$(find /tmp/wpmc_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" >> $OUT/pmc.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if 'conv_wgrad_h' in r['Kernel_Name']:
        a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
for k, (v, n) in acc.items():
    print(f"{k}: {v / n:.4g} per launch ({n} rows)")
PY
  echo "set $i" >> $OUT/progress.txt
done
cat $OUT/pmc.txt
