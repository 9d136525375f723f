#!/usr/bin/env python3
"""Micro-benchmark of the MFMA conv kernels on the dominant C2 shape (B=8, 32^3, 64->64, 3x3x3) — for
rocprofv3 --pmc passes and A/B timing.  Usage: python tools/conv_bench.py [fwd|bwdw|both] [iters] [B S Cin Cout]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusioniqt_amd import ops, _lib

mode = sys.argv[1] if len(sys.argv) > 1 else "both"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
B, S, Cin, Cout = (int(v) for v in sys.argv[3:7]) if len(sys.argv) > 6 else (8, 32, 64, 64)
_lib.load()
dev = "cuda"
x = torch.randn(B, S, S, S, Cin, device=dev)
w = (torch.randn(Cout, Cin, 3, 3, 3, device=dev) * 0.02).requires_grad_()
bias = torch.zeros(Cout, device=dev, requires_grad=True)
dy = torch.randn(B, S, S, S, Cout, device=dev)
flops = 2.0 * B * S ** 3 * Cin * Cout * 27


def timeit(fn, n):
    for _ in range(max(3, int(os.environ.get("WARM", "300")))):     # DVFS: clocks need ~100 ms of load to settle
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


if mode in ("fwd", "both"):
    with torch.no_grad():
        ms = timeit(lambda: ops.conv3d(x, w, bias, (1, 1, 1)), iters)
    print(f"conv fwd  B={B} {S}^3 {Cin}->{Cout}: {ms*1e3:.1f} us  {flops/ms/1e9:.1f} TFLOP/s")
if mode in ("bwdw", "both"):
    xr = x.clone()
    y = ops.conv3d(xr, w, bias, (1, 1, 1))

    def bw():
        w.grad = None
        bias.grad = None
        y.backward(dy, retain_graph=True, inputs=[w, bias])
    ms = timeit(bw, iters)
    print(f"conv bwd-weight(+bias): {ms*1e3:.1f} us  {flops/ms/1e9:.1f} TFLOP/s")
