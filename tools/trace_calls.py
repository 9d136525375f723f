"""Which Python call sites launch a given C entry point during one Unet3D eval (C5 stage-2 shape).   python tools/trace_calls.py diqt_axpby3"""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusioniqt_amd import ops, _lib
from diffusioniqt_amd.imagen_video import Unet3D
_lib.load()
names = sys.argv[1:]
dev = torch.device("cuda:0")
unet = Unet3D(dim=64, dim_mults=(1, 2, 4), channels=1, cond_on_text=False, text_embed_dim=None, lowres_cond=True, layer_attns=False,
              layer_cross_attns=False, attend_at_middle=True, num_resnet_blocks=2, attn_pool_text=False).to(dev).eval()
B, S = 2, 32
x = torch.randn(B, 1, S, S, S, device=dev); lr = torch.randn(B, 1, S, S, S, device=dev)
t = torch.randn(B, device=dev) * 0.5; lt = torch.full((B,), 0.2, device=dev)
seen = collections.Counter()
real = _lib.call


def spy(name, *a):
    if name in names:
        st = traceback.extract_stack()[:-1]
        site = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}({f.name})" for f in st[-5:][::-1])
        shapes = [tuple(v.shape) for v in a if torch.is_tensor(v)][:2]
        seen[(name, site, str(shapes))] += 1
    return real(name, *a)


_lib.call = spy
with torch.no_grad(), torch.autocast('cuda', dtype=torch.float16):
    unet(x, t, lowres_cond_img=lr, lowres_noise_times=lt)
for (name, site, shapes), n in seen.most_common():
    print(n, name, shapes, site)
