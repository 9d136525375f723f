"""Host issue time vs GPU span of a step: which workloads are bound by Python launch overhead?   python tools/host_bound.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import unet_kwargs, unet3d_kwargs
from diffusioniqt_amd import ops, _lib
from diffusioniqt_amd.imagen_pytorch3D import SRUnet256
from diffusioniqt_amd.imagen_video import Unet3D
_lib.load()
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, S = 8, 32


def measure(name, fn, n=6):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    host, gpu = [], []
    for _ in range(n):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); t0 = time.perf_counter()
        fn()
        h = (time.perf_counter() - t0) * 1e3
        e1.record(); e1.synchronize()
        host.append(h); gpu.append(e0.elapsed_time(e1))
    host.sort(); gpu.sort()
    print(f"{name:44s} host issue {host[len(host) // 2]:7.2f} ms   GPU span {gpu[len(gpu) // 2]:7.2f} ms   -> {'HOST-bound' if host[len(host) // 2] > 0.85 * gpu[len(gpu) // 2] else 'GPU-bound'}", flush=True)


x = torch.randn(B, 1, S, S, S, device=dev); lr = torch.randn(B, 1, S, S, S, device=dev)
t = torch.rand(B, device=dev); lt = torch.full((B,), 0.2, device=dev)
ua = SRUnet256(**unet_kwargs(S)).to(dev)
ub = Unet3D(**unet3d_kwargs()).to(dev)
for p in ub.final_conv.parameters():
    torch.nn.init.normal_(p, std=0.05)


def fa_eval():
    with torch.no_grad():
        ua(x, None, t, lowres_cond_img=lr)


def fa_train():
    ua.zero_grad(set_to_none=True)
    ua(x, None, t, lowres_cond_img=lr).square().mean().backward()


def fb_eval():
    with torch.no_grad():
        ub(x, t, lowres_cond_img=lr, lowres_noise_times=lt)


def fb_train():
    ub.zero_grad(set_to_none=True)
    ub(x, t, lowres_cond_img=lr, lowres_noise_times=lt).square().mean().backward()


def fa_train_bf16():
    ua.zero_grad(set_to_none=True)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        y = ua(x, None, t, lowres_cond_img=lr)
    y.float().square().mean().backward()


def fb_eval_fp16():
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.float16):
        ub(x, t, lowres_cond_img=lr, lowres_noise_times=lt)


ua.eval(); ub.eval()
measure("Family A eval (C2 U-Net, fp32)", fa_eval)
measure("Family B eval (Unet3D, fp32)", fb_eval)
measure("Family B eval (Unet3D, autocast fp16)", fb_eval_fp16)
ua.train(); ub.train()
measure("Family A fwd+bwd (fp32)", fa_train)
measure("Family A fwd+bwd (autocast bf16)", fa_train_bf16)
measure("Family B fwd+bwd (fp32)", fb_train)
