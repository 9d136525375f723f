# L2 counters of the joint space-time attention (mqa_flash_fwd_h_kernel) in one C5 stage-2 eval
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/attnmid_${1:-x}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export NO_LAYER_ATTNS=1 AUTOCAST=fp16
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/apmc_$i -o p -- python3 $R/tools/unet3d_bench.py 64 64 8 > $OUT/pmc_$i.log 2>&1 || true
  f=$(find /tmp/apmc_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" >> $OUT/pmc.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if 'mqa_flash_fwd_h' in r['Kernel_Name']:
        a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
for k, (v, n) in acc.items():
    print(f"{k}: {v / n:.4g} per launch ({n} rows)")
PY
  echo "set $i done" >> $OUT/progress.txt
done
cat $OUT/pmc.txt
