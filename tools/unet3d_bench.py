"""Family-B (pseudo-3D Unet3D) eval timing + kernel breakdown.   python tools/unet3d_bench.py [dim] [size] [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusioniqt_amd import ops, _lib
from diffusioniqt_amd.imagen_video import Unet3D
_lib.load()
dim, S, B = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 32, 8)
dev = torch.device("cuda:0")
torch.manual_seed(0)
unet = Unet3D(dim=dim, dim_mults=(1, 2, 4), channels=1, cond_on_text=False, text_embed_dim=None, lowres_cond=True,
              layer_attns=False if os.environ.get('NO_LAYER_ATTNS') == '1' else (False, False, True), layer_cross_attns=False, attend_at_middle=True, num_resnet_blocks=2,
              attn_pool_text=False).to(dev).eval()
x = torch.randn(B, 1, S, S, S, device=dev)
lr = torch.randn(B, 1, S, S, S, device=dev)
t = torch.randn(B, device=dev) * 0.5
lt = torch.full((B,), 0.2, device=dev)


AC = os.environ.get("AUTOCAST")       # fp16 | bf16: run under torch.autocast


def step():
    with torch.no_grad():
        if AC:
            with torch.autocast('cuda', dtype=torch.float16 if AC == 'fp16' else torch.bfloat16):
                unet(x, t, lowres_cond_img=lr, lowres_noise_times=lt)
        else:
            unet(x, t, lowres_cond_img=lr, lowres_noise_times=lt)


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 5
for _ in range(N):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N
print(f"Unet3D dim={dim} {S}^3 B={B}: {dt * 1e3:.2f} ms/eval  ({B / dt:.1f} patch-evals/s); params {sum(p.numel() for p in unet.parameters()) / 1e6:.1f} M")

if os.environ.get("BGEMM_SHAPES") == "1":
    import collections
    from diffusioniqt_amd import _lib as L
    real_call = L.call
    stats = collections.OrderedDict()

    def spy(name, *a):
        if name == "diqt_bgemm":
            g, M, N, K, tA, tB = a[3:9]
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); real_call(name, *a); e.record()
            stats.setdefault((g, M, N, K, tA, tB, a[12], a[13], a[14]), []).append((s, e))
            return
        return real_call(name, *a)
    L.call = spy
    ops._lib.call = spy
    step()
    torch.cuda.synchronize()
    tot = 0
    print("batch      M      N      K  tA tB  lda ldb ldc   calls   us/call   TFLOP/s")
    for k, evs in sorted(stats.items(), key=lambda kv: -sum(s.elapsed_time(e) for s, e in kv[1])):
        ms = sum(s.elapsed_time(e) for s, e in evs)
        tot += ms
        g, M, N, K = k[:4]
        print(f"{g:5d} {M:6d} {N:6d} {K:6d}  {k[4]}  {k[5]}  {k[6]:4d} {k[7]:4d} {k[8]:4d}  {len(evs):5d} {1e3 * ms / len(evs):9.1f} {2.0 * g * M * N * K * len(evs) / ms / 1e9:9.1f}")
    print("bgemm total ms per eval:", tot)

if os.environ.get("CONV_SHAPES") == "1":
    ops.TIMER.enabled = True
    ops.TIMER.reset()
    for _ in range(3):
        step()
    rows = sorted(ops.TIMER.by_shape().items(), key=lambda kv: -kv[1][0])
    tot = sum(v[0] for _, v in rows)
    print(f"conv launches {tot / 3:.3f} ms per eval")
    for (tag, sh), (ms, fl, nn) in rows[:16]:
        Bq, Dd, H, W, Ci, Co, kd, kh, kw = sh
        print(f"{tag:18s} {f'{Bq}x{Dd}x{H}x{W}':>16s} {f'{Ci}->{Co}':>10s} {f'{kd}{kh}{kw}':>5s} {nn // 3:4d} {1e3 * ms / nn:9.1f} us {fl / ms / 1e9:7.1f} TF {100 * ms / tot:5.1f}%")
