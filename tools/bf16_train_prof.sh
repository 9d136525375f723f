set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/bf16prof_${1:-x}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_bf -o h -- python3 $R/tools/host_bound.py > $OUT/h.log 2>&1
cp $(find /tmp/p_bf -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
