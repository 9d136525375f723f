# kernel breakdown of one C5 stage-2 eval (Unet3D dim 64, 64 frames x 64 x 64, batch 8) under autocast fp16
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/c5prof_${1:-x}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export NO_LAYER_ATTNS=1
AUTOCAST=fp16 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_h -o h -- python3 $R/tools/unet3d_bench.py 64 64 8 > $OUT/h.log 2>&1
cp $(find /tmp/p_h -name "*kernel_stats.csv" | head -1) $OUT/c5s2_fp16_kernel_stats.csv
grep "ms" $OUT/h.log | grep Unet3D
