set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python bench.py > gpurun_out/r04_bench_C2.json 2> gpurun_out/t53_bench.err; echo "bench rc=$?"
timeout -k 10 600 python bench.py --config C4 > gpurun_out/r04_bench_C4.json 2> gpurun_out/t53_c4.err; echo "c4 rc=$?"
timeout -k 10 900 python bench.py --config C5 > gpurun_out/r04_bench_C5.json 2> gpurun_out/t53_c5.err; echo "c5 rc=$?"
python - <<'P'
import json
def last(f):
    for line in open(f):
        if line.startswith('{'): d=json.loads(line)
    return d
d=last('gpurun_out/r04_bench_C2.json')
print("C2", d['value'], d['ms_per_step'], "roofline", d['roofline']['frac'], d['roofline'].get('achieved'))
print(" train", d['train']['ms_per_step'], json.dumps(d['train'].get('step_graphs')))
print(" train_bf16", d['train_bf16']['ms_per_step'], " autocast", d['autocast_fp16']['ms_per_step'])
u=d['unet3d_edm']; print(" u3", u['eval_ms'],u['fwd_bwd_ms'],u['fwd_bwd_bf16_ms'])
print(" c4", d['c4']['eval_ms'], d['c4']['autocast_fp16_eval_ms'], " c5_short", d['c5_short']['cascade_ms'])
print(" cpu", json.dumps(d['cpu_baseline'])[:300])
d=last('gpurun_out/r04_bench_C4.json'); print("C4", d['value'], d['ms_per_step'], json.dumps(d.get('autocast_fp16'))[:200])
d=last('gpurun_out/r04_bench_C5.json'); print("C5", d['value'], d['ms_per_step'], d.get('whole_step_tflops'))
P
