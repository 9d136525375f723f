"""C4: Family-A SRUnet256 img 64, dim 128, mults (1,2,4), linear attention at every level + middle,
deep_feature, batch_sample factor 1, B = 1 volume.  python tools/c4_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusioniqt_amd import _lib
from diffusioniqt_amd.imagen_pytorch3D import SRUnet256
_lib.load()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
kw = dict(img_size=S, dim=128, init_dim=128, dim_mults=(1, 2, 4), channels=1, num_resnet_blocks=(2, 2, 2), init_conv_kernel_size=3,
          lowres_cond=True, init_cross_embed=False, att_type='linear', attn_dim_head=64, attend_at_middle=True,
          attend_at_enc=[True, True, True], attend_at_enc_depth=[1, 1, 1], attend_at_enc_heads=[8, 8, 8], memory_efficient=False,
          use_se_attn='True,', pixel_shuffle_upsample=True, boundary=False, batch_sample=True, batch_sample_factor=1, deep_feature=True)
torch.manual_seed(0)
unet = SRUnet256(**kw).cuda().eval()
x = torch.randn(1, 1, S, S, S, device="cuda"); lr = torch.randn_like(x); t = torch.rand(1, device="cuda")
def step():
    with torch.no_grad():
        unet(x, None, t, lowres_cond_img=lr)
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print(f"C4 eval {S}^3 dim 128: {dt * 1e3:.2f} ms  (6133 GFLOP at 64^3 -> {6133.0 * (S / 64) ** 3 / dt / 1e3:.1f} TFLOP/s); params {sum(p.numel() for p in unet.parameters()) / 1e6:.1f} M")
