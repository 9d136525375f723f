"""hipGraph capture of one C2 sampler step (U-Net eval + posterior step) through torch.cuda.CUDAGraph: does the ctypes launch path
capture, and what does replay save?   python tools/graph_probe.py [fp16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import unet_kwargs
from diffusioniqt_amd import ops, _lib
from diffusioniqt_amd.imagen_pytorch3D import SRUnet256
_lib.load()
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, S = 8, 32
lp = sys.argv[1] if len(sys.argv) > 1 else "off"
unet = SRUnet256(**unet_kwargs(S)).to(dev).eval()
img = torch.randn(B, 1, S, S, S, device=dev)
lr = torch.randn(B, 1, S, S, S, device=dev)
cond = torch.rand(B, device=dev)
coef = torch.rand(3, B, device=dev)
out = torch.empty_like(img)


def step():
    with torch.no_grad(), ops.low_precision(lp):
        pred = unet(img, None, cond, lowres_cond_img=lr)
        noise = torch.randn_like(pred)
        nxt, _ = ops.ddpm_step(img, pred, noise, coef[0], coef[1], coef[2], -0.72, 0.0, 0)
        out.copy_(nxt)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


eager = timeit(step)
ref = out.clone()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
graphed = timeit(g.replay)
print(f"{lp}: eager {eager:.3f} ms/step, graph replay {graphed:.3f} ms/step")
