cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pq; ONLY=3 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES --output-format csv -d /tmp/pq -- python3 $GRAFT_REPO_ROOT/tools/attn_bench.py > /dev/null 2>&1
python3 - <<PY
import csv,glob
from collections import defaultdict
f=glob.glob("/tmp/pq/**/*counter_collection.csv",recursive=True)[0]
tot=defaultdict(lambda: defaultdict(float)); cnt=defaultdict(lambda: defaultdict(int))
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"].split("(")[0][-40:]
    tot[n][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[n][r["Counter_Name"]]+=1
for n in tot:
    if "mqa" not in n: continue
    v={k:tot[n][k]/cnt[n][k] for k in tot[n]}
    print(n, {k:int(x) for k,x in v.items()})
    print("  mfma busy/(32*sq busy)", round(v["SQ_VALU_MFMA_BUSY_CYCLES"]/(32*v["SQ_BUSY_CYCLES"]),3), " wait_any/wave_cycles", round(v["SQ_WAIT_ANY"]/v["SQ_WAVE_CYCLES"],3), " wait_inst/wave_cycles", round(v["SQ_WAIT_INST_ANY"]/v["SQ_WAVE_CYCLES"],3), " waves", int(v["SQ_WAVES"]), " wave_cycles per wave", int(4*v["SQ_WAVE_CYCLES"]/v["SQ_WAVES"]))
PY
