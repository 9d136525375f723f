"""Family A (C2 U-Net) evals under torch.autocast(fp16) only -- for a rocprofv3 kernel trace of the autocast sampler's U-Net.   python tools/fa_autocast_prof.py [evals]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import unet_kwargs
from diffusioniqt_amd import _lib
from diffusioniqt_amd.imagen_pytorch3D import SRUnet256
_lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, S = 8, 32
ua = SRUnet256(**unet_kwargs(S)).to(dev).eval()
x = torch.randn(B, 1, S, S, S, device=dev); lr = torch.randn(B, 1, S, S, S, device=dev); t = torch.rand(B, device=dev)
with torch.no_grad(), torch.autocast('cuda', dtype=torch.float16):
    for _ in range(4):
        ua(x, None, t, lowres_cond_img=lr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        ua(x, None, t, lowres_cond_img=lr)
    torch.cuda.synchronize()
print(f"C2 U-Net eval under autocast fp16: {(time.perf_counter() - t0) / n * 1e3:.3f} ms ({n} evals + 4 warm-up evals in the trace)")
