"""Per-shape time of the 16-bit conv entry points during one C5 stage-2 U-Net eval (Unet3D dim 64, 64^3, batch 8, autocast fp16).
   python tools/convh_shapes.py"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusioniqt_amd import ops, _lib
from diffusioniqt_amd.imagen_video import Unet3D
_lib.load()
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, S = 8, 64
unet = Unet3D(dim=64, dim_mults=(1, 2, 4), channels=1, cond_on_text=False, text_embed_dim=None, lowres_cond=True, layer_attns=False,
              layer_cross_attns=False, attend_at_middle=True, num_resnet_blocks=2, attn_pool_text=False).to(dev).eval()
x = torch.randn(B, 1, S, S, S, device=dev); lr = torch.randn(B, 1, S, S, S, device=dev)
t = torch.randn(B, device=dev) * 0.5; lt = torch.full((B,), 0.2, device=dev)


def step():
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.float16):
        unet(x, t, lowres_cond_img=lr, lowres_noise_times=lt)


for _ in range(2):
    step()
torch.cuda.synchronize()
real = _lib.call
log = []


def spy(name, *a):
    if name.startswith("diqt_conv3d_fwd"):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); r = real(name, *a); e.record()
        ints = tuple(v for v in a if isinstance(v, int) and not isinstance(v, bool))
        log.append((name, ints, s, e))
        return r
    return real(name, *a)


_lib.call = spy
step()
torch.cuda.synchronize()
agg = collections.OrderedDict()
for name, ints, s, e in log:
    k = (name, ints[:-1])
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += s.elapsed_time(e)
tot = sum(v[1] for v in agg.values())
print(f"conv launches {len(log)}, {tot:.2f} ms")
for (name, ints), (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    Bq, D, H, W, Cin, Cout, kd, kh, kw = ints[:9]
    fl = 2.0 * Bq * D * H * W * Cin * Cout * kd * kh * kw
    print(f"{name[12:]:22s} {str(ints[:9]):46s} rest {str(ints[9:]):34s} x{n:3d} {ms / n * 1e3:8.1f} us  {fl * n / ms / 1e9:7.0f} TF/s  {100 * ms / tot:5.1f} %")
