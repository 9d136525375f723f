set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/fbprof_${1:-x}
mkdir -p $OUT
cd $R

cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_e -o e -- python3 $R/tools/unet3d_bench.py 64 32 8 > $OUT/eval.log 2>&1
cp $(find /tmp/p_e -name "*kernel_stats.csv" | head -1) $OUT/eval_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_t -o t -- python3 $R/tools/unet3d_train_bench.py 64 32 8 > $OUT/train.log 2>&1
cp $(find /tmp/p_t -name "*kernel_stats.csv" | head -1) $OUT/train_kernel_stats.csv
grep "ms" $OUT/eval.log $OUT/train.log | grep Unet3D
