"""One shape through diqt_conv3d_fwd_h_io (16-bit tensors at either end).   python tools/convh_io_bench.py B D H W Cin Cout kd kh kw xh yh [res]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusioniqt_amd import _lib
_lib.load()
a = [int(v) for v in sys.argv[1:12]] if len(sys.argv) > 11 else [8, 64, 64, 64, 64, 64, 3, 1, 1, 1, 1]
B, D, H, W, Cin, Cout, kd, kh, kw, xh, yh = a
res = len(sys.argv) > 12 and sys.argv[12] == "1"
causal = kd == 3 and kh == 1
pad = (2, 0, 0) if causal else (kd // 2, kh // 2, kw // 2)
epad = (-2, 0, 0) if causal else (0, 0, 0)
geo = (B, D, H, W, Cin, Cout, kd, kh, kw, *pad, *epad)
dev = "cuda"
st = torch.cuda.current_stream().cuda_stream
w = torch.randn(Cout, Cin, kd, kh, kw, device=dev) / (Cin * kd * kh * kw) ** 0.5
n = _lib.query("diqt_conv_packed_h_elems", Cout, Cin, kd, kh, kw)
packed = torch.empty(n, dtype=torch.int16, device=dev)
_lib.call("diqt_conv_pack_weight_h", w, packed, Cout, Cin, kd, kh, kw, 0, 0, st)
x = torch.randn(B, D, H, W, Cin, device=dev)
if xh:
    x = x.half()
y = torch.empty(B, D, H, W, Cout, device=dev, dtype=torch.float16 if yh else torch.float32)
bias = torch.randn(Cout, device=dev)
r = torch.randn(B, D, H, W, Cout, device=dev) if res else None
assert _lib.query("diqt_conv3d_fwd_h_io16_supported", *geo, xh, yh)


def run():
    _lib.call("diqt_conv3d_fwd_h_io", x, packed, bias, r, y, *geo, 0, 1, xh, yh, None, st)


for _ in range(5):
    run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20):
    run()
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 20
fl = 2.0 * B * D * H * W * Cin * Cout * kd * kh * kw
by = x.numel() * x.element_size() + y.numel() * y.element_size() + (r.numel() * 4 if res else 0)
print(f"convh_io {geo[:9]} xh={xh} yh={yh} res={int(res)}: {ms * 1e3:.1f} us  {fl / ms / 1e9:.0f} TF/s  {by / ms / 1e9:.2f} TB/s")
