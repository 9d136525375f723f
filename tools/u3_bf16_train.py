"""Unet3D (bench configuration) forward + backward under bf16 autocast: ms per micro-step and the kernels' shares.
   python tools/u3_bf16_train.py            (DIQT_NO_WGRADH=1 for the fp32 weight-gradient kernels)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from diffusioniqt_amd.imagen_video import Unet3D
from diffusioniqt_amd import ops

dev = torch.device("cuda", 0)
torch.manual_seed(43)
B, S = 8, 32
u3 = Unet3D(**bench.unet3d_kwargs()).to(dev).train()
hr, lr = torch.randn(B, 1, S, S, S, device=dev), torch.randn(B, 1, S, S, S, device=dev)
tb, ltb = torch.randn(B, device=dev) * 0.5, torch.full((B,), 0.2, device=dev)


def step(lp):
    u3.zero_grad(set_to_none=True)
    if lp:
        with torch.autocast('cuda', dtype=torch.bfloat16):
            y = u3(hr, tb, lowres_cond_img=lr, lowres_noise_times=ltb)
    else:
        y = u3(hr, tb, lowres_cond_img=lr, lowres_noise_times=ltb)
    y.float().square().mean().backward()


for lp in (0, 1):
    for _ in range(3):
        step(lp)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        step(lp)
    e.record(); torch.cuda.synchronize()
    print(f"Unet3D fwd+bwd {'bf16 autocast' if lp else 'fp32'}: {s.elapsed_time(e) / 5:.2f} ms", flush=True)
    ops.TIMER.reset(); ops.TIMER.enabled = True
    step(lp)
    torch.cuda.synchronize()
    ops.TIMER.enabled = False
    summ = ops.TIMER.summary()                      # {tag: (ms, flops, launches)}
    tot = sum(v[0] for v in summ.values())
    for name, (ms, fl, n) in sorted(summ.items(), key=lambda kv: -kv[1][0])[:14]:
        print(f"   {name:42s} {n:4d} launches {ms:8.3f} ms  {100 * ms / tot:5.1f} %")
