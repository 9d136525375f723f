#!/usr/bin/env python3
"""Micro-benchmark of the fused GroupNorm + scale/shift + Mish kernels (forward and backward) on the shapes of the C2 training
step.  Usage: python tools/gn_bench.py [iters] [B S C]      (HBM-bound: the figures of merit are bytes moved per second)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusioniqt_amd import ops, _lib

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
B, S, C = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (8, 32, 64)
_lib.load()
dev = "cuda"
x = torch.randn(B, S, S, S, C, device=dev, requires_grad=True)
gamma = torch.ones(C, device=dev, requires_grad=True)
beta = torch.zeros(C, device=dev, requires_grad=True)
ss = (torch.randn(B, 2 * C, device=dev) * 0.1).requires_grad_()
dy = torch.randn(B, S, S, S, C, device=dev)
nbytes = x.numel() * 4


def timeit(fn, n):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


with torch.no_grad():
    ms = timeit(lambda: ops.groupnorm_act(x, gamma, beta, ss), iters)
print(f"gn+mish fwd (stats + apply)  B={B} {S}^3 C={C}: {ms*1e3:.1f} us  ({3 * nbytes / ms / 1e9:.2f} TB/s over 2 reads + 1 write)")
y = ops.groupnorm_act(x, gamma, beta, ss)


def bw():
    x.grad = gamma.grad = beta.grad = ss.grad = None
    y.backward(dy, retain_graph=True)


ms = timeit(bw, iters)
print(f"gn+mish bwd (reduce + dx)    B={B} {S}^3 C={C}: {ms*1e3:.1f} us  ({5 * nbytes / ms / 1e9:.2f} TB/s over 4 reads + 1 write)")
