"""conv_wgrad_h_kernel vs conv_wgrad3_kernel on one shape.   python tools/wgradh_bench.py [B D S Cin Cout kd kh kw]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusioniqt_amd import _lib
_lib.load()
B, D, S, Cin, Cout, kd, kh, kw = (int(v) for v in sys.argv[1:9]) if len(sys.argv) > 8 else (8, 32, 32, 64, 64, 3, 3, 3)
FLAGS = int(sys.argv[9]) if len(sys.argv) > 9 else 1      # bit 0 bf16, bit 1: x holds 16-bit values, bit 2: dY holds 16-bit values
geo = (B, D, S, S, Cin, Cout, kd, kh, kw, kd // 2, kh // 2, kw // 2, 0, 0, 0)
x = torch.randn(B, D, S, S, Cin, device="cuda"); dy = torch.randn(B, D, S, S, Cout, device="cuda")
dw = torch.empty(Cout, Cin, kd, kh, kw, device="cuda"); db = torch.empty(Cout, device="cuda")
st = torch.cuda.current_stream().cuda_stream
fl = 2.0 * B * D * S * S * Cin * Cout * kd * kh * kw


def timeit(fn, n=30):
    for _ in range(100):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


nh = _lib.query("diqt_conv3d_bwd_weight_h_workspace_bytes", *geo)
wsh = torch.empty(max(nh, 4) // 4, device="cuda")
n3 = _lib.query("diqt_conv3d_bwd_weight_workspace_bytes", *geo)
ws3 = torch.empty(max(n3, 4) // 4, device="cuda")
hdt = torch.bfloat16 if FLAGS & 1 else torch.float16
xh = x.to(hdt) if FLAGS & 2 else x
dyh = dy.to(hdt) if FLAGS & 4 else dy
mh = timeit(lambda: _lib.call("diqt_conv3d_bwd_weight_h", xh, dyh, dw, db, wsh, nh, *geo, FLAGS, st))
m3 = timeit(lambda: _lib.call("diqt_conv3d_bwd_weight", x, dy, dw, db, ws3, n3, *geo, st))
print(f"wgrad {B}x{D}x{S}^2 {Cin}->{Cout} ({kd},{kh},{kw}) flags {FLAGS}: 16-bit {mh * 1e3:.1f} us ({fl / mh / 1e9:.0f} TF/s)   fp32 {m3 * 1e3:.1f} us ({fl / m3 / 1e9:.0f} TF/s)  (each incl. the slab sum)")
