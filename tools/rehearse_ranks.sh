#!/bin/bash
# N ranks on ONE card over gloo (bench.py --rehearse): the multi-rank plumbing of the trainer on device tensors, with the reducer's
# collective trace and a watchdog long enough to tell "slow" from "hung".  Usage: tools/rehearse_ranks.sh N [batch]  (on the GPU box)
N=${1:-4}; B=${2:-2}
mkdir -p gpurun_out/rehearse
export DIQT_DDP_TRACE=1 DIQT_BENCH_WATCHDOG=${WD:-500}
exec timeout -k 10 ${LIMIT:-900} python bench.py --gpus $N --rehearse --mode train --steps 4 --warmup 4 --batch $B --no-kernel-timer \
  > gpurun_out/rehearse/n${N}.json 2> gpurun_out/rehearse/n${N}.err
