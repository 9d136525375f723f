#!/bin/bash
# Re-collects the judged profiles of a round on the GPU box (one gpurun call):  bash tools/collect_profiles.sh <tag>
# Outputs under gpurun_out/prof_<tag>/ ; copy the summaries into profiles/ afterwards (gpurun only merges gpurun_out/ back).
set -e
TAG=${1:-r02b}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-extras --no-kernel-timer"
stats() {  # name, command...
  local n=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$n -o $n -- "$@" > $OUT/$n.log 2>&1
  cp $(find /tmp/p_$n -name "*kernel_stats.csv" | head -1) $OUT/${n}_kernel_stats.csv
  echo "done $n"
}
stats sample python3 $R/bench.py --mode sample --steps 20 --warmup 3 $B
stats train python3 $R/bench.py --mode train --steps 12 --warmup 4 $B
stats unet3d_eval python3 $R/tools/unet3d_bench.py 64 32 8
stats unet3d_train python3 $R/tools/unet3d_train_bench.py 64 32 8
# one C5 stage-2 U-Net eval under autocast fp16 (Unet3D dim 64, 64 frames x 64 x 64, batch 8): the cascade's dominant loop body
NO_LAYER_ATTNS=1 AUTOCAST=fp16 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c5s2 -o c5s2 -- python3 $R/tools/unet3d_bench.py 64 64 8 > $OUT/c5s2_fp16.log 2>&1
cp $(find /tmp/p_c5s2 -name "*kernel_stats.csv" | head -1) $OUT/c5s2_fp16_kernel_stats.csv
echo "done c5s2"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_fetch -- python3 $R/bench.py --mode both --steps 4 --warmup 4 $B > $OUT/pmc_fetch.log 2>&1
echo "done fetch"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_write -- python3 $R/bench.py --mode both --steps 4 --warmup 4 $B > $OUT/pmc_write.log 2>&1
echo "done write"
python3 $R/tools/pmc_traffic.py /tmp/pmc_fetch /tmp/pmc_write $OUT/pmc_hbm_traffic.json "bench.py --mode both --steps 4 --warmup 4 $B"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/pmc_sq -- python3 $R/tools/conv_bench.py both 10 > $OUT/pmc_sq.log 2>&1
cp $(find /tmp/pmc_sq -name "*counter_collection.csv" | head -1) $OUT/pmc_sq_counter_collection.csv
echo "done sq"
# ---- round 4 ----
# C4 eval kernel stats; HBM traffic (two passes each) of the C4 eval, of one C5 stage-2 eval under autocast and of the Family-B eval
stats c4_eval python3 $R/bench.py --config C4 --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timer
stats sample_autocast python3 $R/tools/autocast_bench.py
stats train_bf16 python3 $R/tools/train_bf16_only.py 16
traffic() {  # tag, command...
  local t=$1; shift
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmcf_$t -- "$@" > $OUT/pmc_fetch_$t.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmcw_$t -- "$@" > $OUT/pmc_write_$t.log 2>&1
  python3 $R/tools/pmc_traffic.py /tmp/pmcf_$t /tmp/pmcw_$t $OUT/pmc_hbm_traffic_$t.json "$*"
  echo "done traffic $t"
}
traffic C4 python3 $R/bench.py --config C4 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer
NO_LAYER_ATTNS=1 AUTOCAST=fp16 traffic C5 python3 $R/tools/unet3d_bench.py 64 64 8
traffic unet3d python3 $R/tools/unet3d_bench.py 64 32 8
# matrix-pipe busy fraction of whole workloads (SQ counters, own passes) and the conv_f9h_kernel counter set
sq() {  # tag, command...
  local t=$1; shift
  # (the profiler itself can die on a long workload with this counter set -- it segfaulted on the Family-B training bench: 2445 dispatches
  # per step -- so a failed pass is reported and skipped, not fatal)
  if rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/sq_$t -- "$@" > $OUT/sq_$t.log 2>&1; then
    cp $(find /tmp/sq_$t -name "*counter_collection.csv" | head -1) $OUT/sq_${t}_counter_collection.csv
    echo "done sq $t"
  else
    echo "sq $t: profiler failed (rc $?), skipped"
  fi
}
sq sample python3 $R/bench.py --mode sample --steps 6 --warmup 2 $B
sq train python3 $R/bench.py --mode train --steps 8 --warmup 4 $B
sq unet3d_eval python3 $R/tools/unet3d_bench.py 64 32 8
sq f9h_333 python3 $R/tools/convh_io_bench.py 8 32 32 32 64 64 3 3 3 1 1
sq f9h_133 python3 $R/tools/convh_io_bench.py 8 64 64 64 64 64 1 3 3 1 1
python3 $R/tools/pmc_sq_workloads.py $OUT $OUT/pmc_sq_workloads.json
