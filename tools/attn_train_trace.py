"""Per-call time and shape of the attention entry points during one Unet3D (bench configuration) training micro-step.
   python tools/attn_train_trace.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from diffusioniqt_amd import ops, _lib
from diffusioniqt_amd.imagen_video import Unet3D
_lib.load()
dev = torch.device("cuda:0")
torch.manual_seed(43)
B, S = 8, 32
u3 = Unet3D(**bench.unet3d_kwargs()).to(dev).train()
hr, lr = torch.randn(B, 1, S, S, S, device=dev), torch.randn(B, 1, S, S, S, device=dev)
tb, ltb = torch.randn(B, device=dev) * 0.5, torch.full((B,), 0.2, device=dev)


def step():
    u3.zero_grad(set_to_none=True)
    u3(hr, tb, lowres_cond_img=lr, lowres_noise_times=ltb).square().mean().backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
real = _lib.call
log = []


def spy(name, *a):
    if "attention" in name or "softmax" in name:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); r = real(name, *a); e.record()
        log.append((name, [v for v in a if isinstance(v, (int, float))], s, e))
        return r
    return real(name, *a)


_lib.call = spy
step()
torch.cuda.synchronize()
tot = 0.0
for name, ints, s, e in log:
    ms = s.elapsed_time(e); tot += ms
    print(f"{name:36s} {ms * 1e3:8.1f} us  {ints}")
print(f"total {tot:.2f} ms")
