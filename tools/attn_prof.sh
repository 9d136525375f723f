cd /tmp && export TMPDIR=/tmp
for o in 3 5 6; do
rm -rf /tmp/pa; ONLY=$o BWD=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa -o a -- python3 $GRAFT_REPO_ROOT/tools/attn_bench.py > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("/tmp/pa/**/*kernel_stats.csv",recursive=True)[0]
print("shape $o:", "; ".join(r["Name"].split("(")[0][-28:] + " " + str(round(float(r["AverageNs"])/1e3,1)) for r in list(csv.DictReader(open(f)))[:4]))
PY
done
