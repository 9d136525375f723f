"""1x1x1 conv / Linear rows (conv1x1_fwd_kernel vs the generic kernel, DIQT_CONV_NO1X1=1).  python tools/pw_gemm_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusioniqt_amd import ops, _lib
_lib.load()
dev = "cuda"
for rows, Cin, Cout in ((262144, 64, 512), (262144, 512, 64), (262144, 64, 128), (16384, 256, 512), (16384, 512, 256), (262144, 128, 64),
                        (32768, 128, 1024)):
    x = torch.randn(rows, Cin, device=dev)
    w = torch.randn(Cout, Cin, device=dev) * 0.05
    b = torch.randn(Cout, device=dev)
    with torch.no_grad():
        for _ in range(20):
            y = ops.linear(x, w, b)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            y = ops.linear(x, w, b)
        e.record()
        torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    ref = x[:256].double() @ w.double().t() + b.double()
    err = (y[:256].double() - ref).abs().max().item() / ref.abs().max().item()
    print(f"rows {rows:7d} {Cin:4d}->{Cout:4d}: {ms * 1e3:8.1f} us  {2.0 * rows * Cin * Cout / ms / 1e9:6.1f} TFLOP/s  rel err {err:.1e}")
