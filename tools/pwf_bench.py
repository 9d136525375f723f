"""Pointwise conv / Linear forward timing.  python tools/pwf_bench.py rows Cin Cout   (DIQT_NO_PW64=1: the per-block kernel)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusioniqt_amd import ops, _lib
_lib.load()
rows, Cin, Cout = (int(v) for v in sys.argv[1:4])
x = torch.randn(1, 1, 1, rows, Cin, device="cuda")
w = torch.randn(Cout, Cin, 1, 1, 1, device="cuda") * 0.05
with torch.no_grad():
    for _ in range(300):
        ops.conv3d(x, w, None, (0, 0, 0))
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(100):
        ops.conv3d(x, w, None, (0, 0, 0))
    e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 100
print(f"pointwise fwd rows={rows} {Cin}->{Cout}: {ms*1e3:.1f} us  {2.0*rows*Cin*Cout/ms/1e9:.1f} TFLOP/s  {(rows*(Cin+Cout)*4)/ms/1e9:.2f} TB/s")
