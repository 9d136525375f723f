"""End-to-end training sanity on the HIP path: a small SRUnet256 (dim 32, 16^3 patches) learns x_start prediction for a synthetic
LR -> HR relation (HR = LR + a fixed smooth field) through ImagenTrainer (gradient accumulation 2, fused Adam, EMA) for a few hundred
optimiser steps, in fp32 and with precision='bf16'.  Prints the loss trajectory; the loss must fall by a large factor and stay finite.
    python tools/train_sanity.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusioniqt_amd import _lib
from diffusioniqt_amd.imagen_pytorch3D import SRUnet256, Imagen, NullUnet
from diffusioniqt_amd.trainer import ImagenTrainer
_lib.load()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
S, B = 16, 8
dev = "cuda"
g = torch.Generator().manual_seed(0)
zz, yy, xx = torch.meshgrid(*[torch.linspace(-1, 1, S)] * 3, indexing="ij")
field = (0.5 * torch.sin(3 * xx) * torch.cos(2 * yy) + 0.3 * zz)[None, None]


def batch():
    lr = torch.randn(B, 1, S, S, S, generator=g) * 0.5
    return (lr + field).to(dev), lr.to(dev)


for mode in ("no", "bf16"):
    torch.manual_seed(1)
    unet = SRUnet256(img_size=S, dim=32, dim_mults=(1, 2, 4), channels=1, num_resnet_blocks=(2, 2, 2), init_conv_kernel_size=3,
                     lowres_cond=True, init_cross_embed=False, att_type='linear', attend_at_middle=False, attend_at_enc=[False] * 3,
                     attend_at_enc_depth=[1] * 3, attend_at_enc_heads=[8] * 3, init_dim=32, memory_efficient=False, use_se_attn='True,',
                     pixel_shuffle_upsample=True, boundary=False, batch_sample=False, batch_sample_factor=3, deep_feature=False)
    configs = {'Data': {'norm': 'z-score'}, 'Train': {'batch_sample': False, 'patch_size_sub': S, 'pred_obj': 'x_start'}, 'Eval': {'repeat': 1}}
    imagen = Imagen(unets=(NullUnet(), unet), configs=configs, min_bound=-10.0, image_sizes=(S, S), channels=1, pred_objectives='x_start',
                    timesteps=16, dynamic_thresholding=False, p2_loss_weight_gamma=0.0, cond_drop_prob=0.0).to(dev)
    ImagenTrainer.locked = False
    trainer = ImagenTrainer(configs=configs, imagen=imagen, gradient_accumulation_steps=2, lr=3e-4, verbose=False,
                            **({} if mode == "no" else {"precision": mode}))
    losses, t0 = [], time.perf_counter()
    for it in range(steps * 2):
        hr, lr = batch()
        out = trainer(hr, lowres_img=lr, unet_number=2, max_batch_size=B)
        trainer.update(unet_number=2)
        losses.append(float(out[0]) if isinstance(out, tuple) else float(out))
    torch.cuda.synchronize()
    k = max(1, len(losses) // 10)
    traj = [sum(losses[i:i + k]) / len(losses[i:i + k]) for i in range(0, len(losses), k)]
    ok = all(l == l and l < 1e6 for l in losses) and traj[-1] < 0.25 * traj[0]
    print(f"precision={mode}: {steps} optimiser steps in {time.perf_counter() - t0:.1f} s; mean loss per decile: " + " ".join(f"{v:.4f}" for v in traj)
          + ("  OK" if ok else "  FAILED"))
    assert ok
    # ... and the trained model super-resolves: 16-step ancestral sampling from fresh LR patches lands on LR + field
    hr, lr = batch()
    out = trainer.sample(batch_size=B, start_image_or_video=lr, start_at_unet_number=2, use_tqdm=False)[0]
    err = (out - hr).pow(2).mean().item()
    base = (lr - hr).pow(2).mean().item()                       # the error of returning the input unchanged
    print(f"  sampling: MSE(sample, HR) = {err:.5f} vs MSE(LR, HR) = {base:.5f}  ({base / err:.0f}x closer)")
    assert err < 0.1 * base
