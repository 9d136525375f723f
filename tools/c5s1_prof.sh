# kernel breakdown of one C5 stage-1 eval (Unet3D dim 64, 32 frames x 32 x 32, batch 8) under autocast fp16
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/c5s1_${1:-x}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export NO_LAYER_ATTNS=1
AUTOCAST=fp16 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_s1 -o h -- python3 $R/tools/unet3d_bench.py 64 32 8 > $OUT/h.log 2>&1
cp $(find /tmp/p_s1 -name "*kernel_stats.csv" | head -1) $OUT/c5s1_fp16_kernel_stats.csv
grep "ms" $OUT/h.log | grep Unet3D
