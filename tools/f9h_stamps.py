"""In-kernel cycle stamps of conv_f9h_kernel on one shape (DIQT_F9H_DBG=1).   python tools/f9h_stamps.py B D H W Cin Cout kd kh kw xh yh"""
import ctypes, os, sys
os.environ.setdefault("DIQT_F9H_DBG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from diffusioniqt_amd import _lib
lib = _lib.load()
a = [int(v) for v in sys.argv[1:12]]
B, D, H, W, Cin, Cout, kd, kh, kw, xh, yh = a
pad = (kd // 2, kh // 2, kw // 2)
geo = (B, D, H, W, Cin, Cout, kd, kh, kw, *pad, 0, 0, 0)
dev = "cuda"
st = torch.cuda.current_stream().cuda_stream
w = torch.randn(Cout, Cin, kd, kh, kw, device=dev) / (Cin * kd * kh * kw) ** 0.5
n = _lib.query("diqt_conv_packed_h_elems", Cout, Cin, kd, kh, kw)
packed = torch.empty(n, dtype=torch.int16, device=dev)
_lib.call("diqt_conv_pack_weight_h", w, packed, Cout, Cin, kd, kh, kw, 0, 0, st)
x = torch.randn(B, D, H, W, Cin, device=dev).half()
y = torch.empty(B, D, H, W, Cout, device=dev, dtype=torch.float16 if yh else torch.float32)
bias = torch.randn(Cout, device=dev)
for _ in range(20):
    _lib.call("diqt_conv3d_fwd_h_io", x, packed, bias, None, y, *geo, 0, 1, xh, yh, None, st)
torch.cuda.synchronize()
buf = np.zeros((1024, 32), dtype=np.uint64)
fn = ctypes.CDLL(_lib.LIB_PATH).diqt_debug_f9h_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_uint]
nw = fn(buf.ctypes.data, 1024)
s = buf[:nw].astype(np.int64)
start, end = s[:, 30], s[:, 31]
body = s[:, :28]
rt0, rt1 = s[:, 28], s[:, 29]
nst = int((body[0] > 0).sum())
rel = body[:, :nst] - body[:, 0:1]
print(f"{nw} waves, {nst} stamps per wave; median cycles since the first recorded stamp, and the median step from the previous stamp:")
med = np.median(rel, axis=0)
step = np.median(np.diff(rel, axis=1), axis=0)
for i in range(nst):
    print(f"  stamp {i:2d}: {med[i]:9.0f}   (+{step[i - 1] if i else 0:8.0f})")
life = end - start
clk = (end - start) / np.maximum(rt1 - rt0, 1) * 100.0
print(f"in-kernel clock (cycles per 100 MHz tick): median {np.median(clk):.0f} MHz, min {clk.min():.0f}, max {clk.max():.0f}; kernel span by the real-time counter: {(rt1.max() - rt0.min()) / 100.0:.1f} us, start spread {(rt0.max() - rt0.min()) / 100.0:.1f} us, end spread {(rt1.max() - rt1.min()) / 100.0:.1f} us")
print(f"wave lifetime: median {np.median(life):.0f}, min {life.min()}, max {life.max()} cycles")
print(f"kernel: first start -> last end {end.max() - start.min()} cycles; start times spread {start.max() - start.min()}, end times spread {end.max() - end.min()}")
wg = start.reshape(-1, 4)[:, 0]
order = np.argsort(wg)
print("workgroup start offsets (sorted, every 32nd):", (wg[order] - wg.min())[::32])
