"""Family-B (Unet3D) forward + backward of a bf16 training step, timed.   python tools/u3_train_bf16.py [dim] [size] [batch] [steps]
DIQT_NO_TRAIN_FUSE=1 runs the two-node Block (fp32 activation between GroupNorm-apply and the per-frame conv)."""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusioniqt_amd import ops, _lib
from diffusioniqt_amd.imagen_video import Unet3D
_lib.load()
dim, S, B, N = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (64, 32, 8, 8)
dev = torch.device("cuda:0")
torch.manual_seed(43)
u3 = Unet3D(dim=dim, dim_mults=(1, 2, 4), channels=1, cond_on_text=False, text_embed_dim=None, lowres_cond=True,
            layer_attns=(False, False, True), layer_cross_attns=False, attend_at_middle=True, num_resnet_blocks=2,
            attn_pool_text=False).to(dev).train()
for p in u3.final_conv.parameters():
    torch.nn.init.normal_(p, std=0.05)
x = torch.randn(B, 1, S, S, S, device=dev)
lr = torch.randn(B, 1, S, S, S, device=dev)
t = torch.randn(B, device=dev) * 0.5
lt = torch.full((B,), 0.2, device=dev)


def step():
    u3.zero_grad(set_to_none=True)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        y = u3(x, t, lowres_cond_img=lr, lowres_noise_times=lt)
    y.float().square().mean().backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
gc.collect(); gc.freeze()
with _lib.census() as c:
    step()
    torch.cuda.synchronize()
    print("launches per step:", c.total() if hasattr(c, "total") else "?", " f9h forwards:", c.count("conv3d_fwd_h(v9h)"))
t0 = time.perf_counter()
for _ in range(N):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N
print(f"Unet3D dim={dim} {S}^3 B={B} bf16 fwd+bwd: {dt * 1e3:.2f} ms/step")
