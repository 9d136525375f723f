"""1x1x1 conv backward-weight timing (rocprofv3 --kernel-trace --stats friendly).  python tools/pw_bench.py B S Cin Cout"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusioniqt_amd import ops, _lib
_lib.load()
B, S, Cin, Cout = (int(v) for v in sys.argv[1:5])
x = torch.randn(B, S, S, S, Cin, device="cuda")
w = (torch.randn(Cout, Cin, 1, 1, 1, device="cuda") * 0.05).requires_grad_()
b = torch.zeros(Cout, device="cuda", requires_grad=True)
y = ops.conv3d(x, w, b, (0, 0, 0))
dy = torch.randn_like(y)
for _ in range(20):
    w.grad = None; b.grad = None
    y.backward(dy, retain_graph=True, inputs=[w, b])
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(50):
    w.grad = None; b.grad = None
    y.backward(dy, retain_graph=True, inputs=[w, b])
e.record(); torch.cuda.synchronize()
print(f"1x1x1 bwd-weight B={B} {S}^3 {Cin}->{Cout}: {s.elapsed_time(e) / 50 * 1e3:.1f} us per call")
