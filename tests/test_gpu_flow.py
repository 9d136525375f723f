"""train.py / test_all.py flow on a real MI355X through the compat/ import names: construct as train.py:81-141 does,
train_step + update (train.py:159-162), valid_step (:168), save/load (:194, test_all.py:173), trainer.sample with the
test_all.py:234 kwargs — all on the HIP path, with the synthetic (hr, lr) dataset."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_train_and_inference_flow(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, 'compat'))
    from imagen_pytorch3D import NullUnet, Imagen, SRUnet256
    from trainer import ImagenTrainer
    from utils_mine import set_seed
    from data import SyntheticPatchDataset
    set_seed(42)
    S = 16
    configs = {'Data': {'norm': 'z-score', 'mean': 271.64814106698583, 'std': 377.117173547721},
               'Train': {'batch_sample': False, 'patch_size_sub': S, 'batch_sample_factor': 3, 'pred_obj': 'x_start',
                         'timesteps': 6, 'dynamic_threshold': False, 'batch_size': 4, 'lpips': False, 'medlpips': False,
                         'boundary': False},
               'Eval': {'repeat': 1, 'batch_size': 4}}
    min_bound = (0. - configs['Data']['mean']) / configs['Data']['std']
    unet2 = SRUnet256(img_size=S, dim=32, dim_mults=(1, 2, 4), channels=1, num_resnet_blocks=(2, 2, 2), init_conv_kernel_size=3,
                      lowres_cond=True, init_cross_embed=False, init_cross_embed_kernel_sizes=(3, 5, 7), att_type='linear',
                      attn_dim_head=64, attend_at_middle=False, attend_at_middle_depth=1, attend_at_middle_heads=8,
                      attend_at_enc=[False] * 3, attend_at_enc_depth=[1] * 3, attend_at_enc_heads=[8] * 3, att_drop=0.0,
                      att_forward_drop=0.0, att_forward_expansion=2, att_skip_scale=False, att_localvit=False, groups=1,
                      emb_size=256, init_dim=32, memory_efficient=False, use_se_attn='True,', pixel_shuffle_upsample=True,
                      boundary=False, batch_sample=False, batch_sample_factor=3, deep_feature=False)
    imagen = Imagen(configs=configs, unets=(NullUnet(), unet2), min_bound=min_bound, image_sizes=(S, S), channels=1,
                    pred_objectives='x_start', timesteps=6, dynamic_thresholding=False, p2_loss_weight_gamma=0.0,
                    auto_normalize_img=False, cond_drop_prob=0.0, lpips=False, medlpips=False, boundary=False).to('cuda')
    ImagenTrainer.locked = False
    trainer = ImagenTrainer(configs=configs, imagen=imagen, gradient_accumulation_steps=4, split_valid_from_train=False, verbose=False)
    trainer.add_train_dataset(SyntheticPatchDataset(n=16, size=S, seed=1), batch_size=4)
    trainer.add_valid_dataset(SyntheticPatchDataset(n=4, size=S, seed=2), batch_size=4)
    w0 = trainer.imagen.unets[1].final_conv.weight.detach().clone()
    losses = []
    for i in range(3):
        losses.append(trainer.train_step(unet_number=2, max_batch_size=4))
        trainer.update(unet_number=2)
    assert all(np.isfinite(losses)) and int(trainer.steps[1]) == 15      # 3 x (4 batches + the extra update)
    assert not torch.equal(trainer.imagen.unets[1].final_conv.weight.detach(), w0)     # Adam stepped (micro-steps 4, 8, 12)
    assert losses[-1] < losses[0] * 1.5
    vloss, preds, x_noisy, (hrs, lrs), ssim, psnr = trainer.valid_step(unet_number=2, max_batch_size=4)
    assert preds.shape == (4, 1, S, S, S) and np.isfinite(vloss) and np.isfinite(psnr)
    path = os.path.join(tmp_path, 'model', '3dimagen.pt')
    trainer.save(path)
    trainer.load(path)
    lr = torch.randn(2, 1, S, S, S)
    out = trainer.sample(batch_size=2, skip_steps=None, return_all_outputs=False, return_pil_images=False,
                         start_image_or_video=lr, start_at_unet_number=2)
    assert tuple(out[0].shape) == (2, 1, S, S, S) and torch.isfinite(out[0]).all() and float(out[0].min()) >= min_bound - 1e-6
    assert len(out[1]) == 7 and isinstance(out[1][0], np.ndarray)
    # EMA weights (initialised by copy at the first update) are what sample() used; non-EMA path also runs
    out2 = trainer.sample(batch_size=2, start_image_or_video=lr, start_at_unet_number=2, use_non_ema=True)
    assert torch.isfinite(out2[0]).all()


@pytest.mark.parametrize("precision", ["no", "bf16"])
def test_training_converges_on_a_synthetic_relation(precision):
    """End to end through ImagenTrainer on the HIP path (loss, backward, gradient accumulation 2, fused Adam, EMA): a dim-32 U-Net on
    16^3 patches learns HR = LR + a fixed smooth field; the x_start loss must fall at least 4x within 80 optimiser steps and stay finite,
    in fp32 and with the mixed-precision switch (tools/train_sanity.py runs the longer version: 0.218 -> 0.003 in 300 steps)."""
    from diffusioniqt_amd.imagen_pytorch3D import SRUnet256, Imagen, NullUnet
    from diffusioniqt_amd.trainer import ImagenTrainer
    S, B, steps = 16, 8, 80
    g = torch.Generator().manual_seed(0)
    zz, yy, xx = torch.meshgrid(*[torch.linspace(-1, 1, S)] * 3, indexing="ij")
    field = (0.5 * torch.sin(3 * xx) * torch.cos(2 * yy) + 0.3 * zz)[None, None]
    torch.manual_seed(1)
    unet = SRUnet256(img_size=S, dim=32, dim_mults=(1, 2, 4), channels=1, num_resnet_blocks=(2, 2, 2), init_conv_kernel_size=3,
                     lowres_cond=True, init_cross_embed=False, att_type='linear', attend_at_middle=False, attend_at_enc=[False] * 3,
                     attend_at_enc_depth=[1] * 3, attend_at_enc_heads=[8] * 3, init_dim=32, memory_efficient=False, use_se_attn='True,',
                     pixel_shuffle_upsample=True, boundary=False, batch_sample=False, batch_sample_factor=3, deep_feature=False)
    configs = {'Data': {'norm': 'z-score'}, 'Train': {'batch_sample': False, 'patch_size_sub': S, 'pred_obj': 'x_start'}, 'Eval': {'repeat': 1}}
    imagen = Imagen(unets=(NullUnet(), unet), configs=configs, min_bound=-10.0, image_sizes=(S, S), channels=1, pred_objectives='x_start',
                    timesteps=16, dynamic_thresholding=False, p2_loss_weight_gamma=0.0, cond_drop_prob=0.0).to('cuda')
    ImagenTrainer.locked = False
    trainer = ImagenTrainer(configs=configs, imagen=imagen, gradient_accumulation_steps=2, lr=3e-4, verbose=False,
                            **({} if precision == "no" else {"precision": precision}))
    losses = []
    for _ in range(steps * 2):
        lr = torch.randn(B, 1, S, S, S, generator=g) * 0.5
        out = trainer((lr + field).to('cuda'), lowres_img=lr.to('cuda'), unet_number=2, max_batch_size=B)
        trainer.update(unet_number=2)
        losses.append(float(out[0]) if isinstance(out, tuple) else float(out))
    assert all(np.isfinite(losses))
    first, last = np.mean(losses[:10]), np.mean(losses[-10:])
    assert last < 0.25 * first, (first, last)
