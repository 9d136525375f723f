set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python bench.py > gpurun_out/t43_bench.json 2> gpurun_out/t43_bench.err; echo "bench rc=$?"
python - <<'P'
import json
for line in open('gpurun_out/t43_bench.json'):
    if line.startswith('{'):
        d=json.loads(line)
        print(d['value'], d['ms_per_step'])
        print(json.dumps(d.get('train_bf16'))[:120])
        t=d.get('train'); print(t['ms_per_step'], json.dumps(t.get('step_graphs')))
        print({k:(v.get('ms_per_step') if isinstance(v,dict) else None) for k,v in d.items() if 'autocast' in k})
        u=d['unet3d_edm']; print(u['eval_ms'],u['fwd_bwd_ms'],u['fwd_bwd_bf16_ms'])
P
