"""C2 sampler step and Family-B eval under torch.autocast (fp16 / bf16 MFMA conv kernel) vs fp32.   python tools/autocast_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import unet_kwargs
from diffusioniqt_amd import ops, _lib
from diffusioniqt_amd.imagen_pytorch3D import SRUnet256
from diffusioniqt_amd.imagen_video import Unet3D
_lib.load()
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, S = 8, 32
ua = SRUnet256(**unet_kwargs(S)).to(dev).eval()
ub = Unet3D(dim=64, dim_mults=(1, 2, 4), channels=1, cond_on_text=False, text_embed_dim=None, lowres_cond=True,
            layer_attns=(False, False, True), layer_cross_attns=False, attend_at_middle=True, num_resnet_blocks=2,
            attn_pool_text=False).to(dev).eval()
for p_ in ub.final_conv.parameters():           # zero-initialised in the reference: give the comparison something to compare
    torch.nn.init.normal_(p_, std=0.05)
x = torch.randn(B, 1, S, S, S, device=dev); lr = torch.randn(B, 1, S, S, S, device=dev)
t = torch.rand(B, device=dev); lt = torch.full((B,), 0.2, device=dev)
fa = lambda: ua(x, None, t, lowres_cond_img=lr)
fb = lambda: ub(x, t, lowres_cond_img=lr, lowres_noise_times=lt)


def timeit(fn, n=10, warm=4):
    with torch.no_grad():
        for _ in range(warm):
            y = fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            y = fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, y


for name, fn in (("Family A C2 eval", fa), ("Family B Unet3D eval", fb)):
    ms32, y32 = timeit(fn)
    print(f"{name}: fp32 {ms32:.2f} ms")
    for dt in (torch.float16, torch.bfloat16):
        with torch.autocast('cuda', dtype=dt):
            ms, y = timeit(fn)
        print(f"{name}: autocast {dt} {ms:.2f} ms  ({ms32 / ms:.2f}x)  rel-L2 vs fp32 {((y - y32).norm() / y32.norm()).item():.2e}")
if os.environ.get("CONV_SHAPES") == "1":
    ops.TIMER.enabled = True
    for name, fn in (("A", fa), ("B", fb)):
        ops.TIMER.reset()
        with torch.autocast('cuda', dtype=torch.float16), torch.no_grad():
            for _ in range(3):
                fn()
        rows = sorted(ops.TIMER.by_shape().items(), key=lambda kv: -kv[1][0])
        tot = sum(v[0] for _, v in rows)
        print(f"{name}: conv launches {tot / 3:.3f} ms per eval")
        for (tag, sh), (ms, fl, nn) in rows[:14]:
            Bq, Dd, H, W, Ci, Co, kd, kh, kw = sh
            print(f"{tag:18s} {f'{Bq}x{Dd}x{H}x{W}':>16s} {f'{Ci}->{Co}':>10s} {f'{kd}{kh}{kw}':>5s} {nn // 3:4d} {1e3 * ms / nn:9.1f} us {fl / ms / 1e9:7.1f} TF {100 * ms / tot:5.1f}%")
