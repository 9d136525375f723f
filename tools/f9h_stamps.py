"""In-kernel cycle stamps of conv_f9h_kernel on one shape (DIQT_F9H_DBG=1).   python tools/f9h_stamps.py B D H W Cin Cout kd kh kw xh yh"""
import ctypes, os, sys
os.environ["DIQT_F9H_DBG"] = "1"
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
t0 = s[:, 0:1]
rel = s - t0
valid = (s > 0)
nst = int(valid[0].sum())
print(f"{nw} waves, {nst} stamps per wave; median cycles since the wave's start, and the median step from the previous stamp:")
med = np.median(rel[:, :nst], axis=0)
step = np.median(np.diff(rel[:, :nst], axis=1), axis=0)
for i in range(nst):
    print(f"  stamp {i:2d}: {med[i]:9.0f}   (+{step[i - 1] if i else 0:8.0f})")
span = s[:, :nst].max() - s[:, 0].min()
print(f"first start -> last end over all waves: {span} cycles; per-wave lifetime median {np.median(rel[:, nst - 1]):.0f}, max {rel[:, nst - 1].max()}")
print("spread of wave start times: ", int(s[:, 0].max() - s[:, 0].min()))
