cd /tmp && export TMPDIR=/tmp
for a in "262144 64 512" "262144 64 128" "524288 64 512"; do
for v in 0 1; do
  export DIQT_NO_PW64=$v
  rm -rf /tmp/pp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o p -- python3 $GRAFT_REPO_ROOT/tools/pwf_bench.py $a > /dev/null 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("/tmp/pp/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:1]:
    print("$a NO_PW64=$v", r["Name"][:50], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us")
PY
done; done
