set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
run() { BENCH_PROF=1 timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $R/gpurun_out/t21_b.json 2> $R/gpurun_out/t21_b.err; grep "\[prof\]" $R/gpurun_out/t21_b.err; python - "$1" $R <<'P'
import json,sys
d=json.loads(open(sys.argv[2]+'/gpurun_out/t21_b.json').read().strip().splitlines()[-1])
print(sys.argv[1], d.get('value'), {k:(v.get('ms_per_step') if isinstance(v,dict) else None) for k,v in d.items() if k in ('train','train_bf16','autocast_fp16')})
P
}
cd $R; BENCH_DBG=direct run "direct"
cd $R; run "unet_eval"
