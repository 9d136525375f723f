"""Per-shape time of the conv launches in one C2 sampler step and one training micro-step (HIP-event timed).
   python tools/shape_profile.py [sample|train] [a|b]      (b: Family B Unet3D dim 64)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import unet_kwargs
from diffusioniqt_amd import ops, _lib
from diffusioniqt_amd.imagen_pytorch3D import SRUnet256

mode = sys.argv[1] if len(sys.argv) > 1 else "sample"
fam = sys.argv[2] if len(sys.argv) > 2 else "a"
_lib.load()
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, S = 8, 32
if fam == "b":
    from bench import unet3d_kwargs
    from diffusioniqt_amd.imagen_video import Unet3D
    unet = Unet3D(**unet3d_kwargs()).to(dev)
else:
    unet = SRUnet256(**unet_kwargs(S)).to(dev)
x = torch.randn(B, 1, S, S, S, device=dev)
lr = torch.randn(B, 1, S, S, S, device=dev)
t = torch.rand(B, device=dev)


lt = torch.full((B,), 0.2, device=dev)


def fwd():
    if fam == "b":
        return unet(x, t * 0.5, lowres_cond_img=lr, lowres_noise_times=lt)
    return unet(x, None, t, lowres_cond_img=lr)


def step():
    if mode == "sample":
        with torch.no_grad():
            fwd()
    else:
        unet.zero_grad(set_to_none=True)
        fwd().square().mean().backward()


unet.train(mode == "train")
for _ in range(3):
    step()
torch.cuda.synchronize()
ops.TIMER.enabled = True
ops.TIMER.reset()
N = 5
for _ in range(N):
    step()
rows = sorted(ops.TIMER.by_shape().items(), key=lambda kv: -kv[1][0])
tot = sum(v[0] for _, v in rows)
print(f"{mode}: conv launches {tot / N:.3f} ms per step")
print(f"{'kernel':24s} {'B,D,H,W':>14s} {'Cin->Cout':>10s} {'k':>5s} {'n/step':>6s} {'us/launch':>10s} {'TFLOP/s':>8s} {'share':>6s}")
for (tag, sh), (ms, fl, n) in rows:
    Bq, D, H, W, Ci, Co, kd, kh, kw = sh
    print(f"{tag:24s} {f'{Bq}x{D}x{H}x{W}':>14s} {f'{Ci}->{Co}':>10s} {f'{kd}{kh}{kw}':>5s} {n // N:6d} {1e3 * ms / n:10.1f} "
          f"{fl / ms / 1e9:8.1f} {100 * ms / tot:5.1f}%")
