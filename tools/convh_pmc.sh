# PMC counters of the low-precision conv kernel on one shape.  (No TA_* counters: that set hung the profiler on this pool.)   bash tools/convh_pmc.sh <tag> <KSHAPE> [env...]
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/convh_${1:-x}
mkdir -p $OUT
export KSHAPE=${2:-1,3,3} SHAPE=64,64,64 WARM=3
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/cpmc_$i -o p -- python3 $R/tools/conv_bench.py fwdh 3 8 64 64 64 > $OUT/pmc_$i.log 2>&1 || true
  f=$(find /tmp/cpmc_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" >> $OUT/pmc.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if 'conv_fwd_h' in r['Kernel_Name']:
        a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
for k, (v, n) in acc.items():
    print(f"{k}: {v / n:.4g} per launch ({n} rows)")
PY
done
cat $OUT/pmc.txt
