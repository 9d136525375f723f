"""HBM-bound operator timings at the C2 32^3 level shapes (HIP events).  python tools/ew_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusioniqt_amd import ops, _lib
_lib.load()
dev = torch.device("cuda:0")


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for (B, S, C) in [(8, 32, 64), (8, 16, 128), (8, 8, 256)]:
    x = torch.randn(B, S, S, S, C, device=dev)
    r = torch.randn_like(x)
    gamma, beta = torch.randn(C, device=dev), torch.randn(C, device=dev)
    ss = torch.randn(B, 2 * C, device=dev)
    gate = torch.rand(B, C, device=dev)
    mb = x.numel() * 4 / 1e6
    with torch.no_grad():
        t = timeit(lambda: ops.groupnorm_act(x, gamma, beta, ss, 8, ops.ACT_MISH))
        print(f"[{B}x{S}^3x{C}] groupnorm+ss+mish fwd (stats+apply): {t:7.1f} us   ({3 * mb / t * 1e-3:.2f} TB/s of 3x{mb:.0f} MB)")
        t = timeit(lambda: ops.gate_residual(x, gate, r))
        print(f"[{B}x{S}^3x{C}] gate*h+res:                          {t:7.1f} us   ({3 * mb / t * 1e-3:.2f} TB/s)")
        t = timeit(lambda: ops.mish(x))
        print(f"[{B}x{S}^3x{C}] mish:                                {t:7.1f} us   ({2 * mb / t * 1e-3:.2f} TB/s)")
    xg = x.clone().requires_grad_()
    y = ops.groupnorm_act(xg, gamma.requires_grad_(), beta.requires_grad_(), ss, 8, ops.ACT_MISH)
    dy = torch.randn_like(y)
    t = timeit(lambda: torch.autograd.grad(y, xg, dy, retain_graph=True))
    print(f"[{B}x{S}^3x{C}] groupnorm+ss+mish bwd:                   {t:7.1f} us   ({5 * mb / t * 1e-3:.2f} TB/s of 5x)")
