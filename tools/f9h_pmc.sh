# PMC counters of conv_f9h_kernel on one shape.   bash tools/f9h_pmc.sh <tag> B D H W Cin Cout kd kh kw xh yh
set -e
R=$GRAFT_REPO_ROOT
TAG=$1; shift
OUT=$R/gpurun_out/f9h_$TAG
mkdir -p $OUT
: > $OUT/pmc.txt
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/f9pmc_$i -o p -- python3 $R/tools/convh_io_bench.py "$@" > $OUT/pmc_$i.log 2>&1 || true
  f=$(find /tmp/f9pmc_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" >> $OUT/pmc.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if 'conv_f9h' in r['Kernel_Name'] or 'conv_fwd_h' in r['Kernel_Name']:
        a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
for k, (v, n) in acc.items():
    print(f"{k}: {v / n:.5g} per launch ({n} rows)")
PY
done
cat $OUT/pmc.txt
