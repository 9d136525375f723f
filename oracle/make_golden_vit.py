"""TEST INFRASTRUCTURE (container-only): Family-A U-Net with att_type='vit' (ViT3D, imagen_pytorch3D.py:871-910) at every
level + middle, produced by running the REAL reference.   Run:  python oracle/make_golden_vit.py"""
import os
import sys
import json
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402
from make_golden import unet_kwargs_train_py, fill, save  # noqa: E402

if __name__ == "__main__":
    r3, _, _, _ = ref_shim.import_reference()
    g = torch.Generator().manual_seed(97)
    for tag, local in (('local', True), ('mlp', False)):
        S, dim = 16, 16
        kw = unet_kwargs_train_py(dim, S, att_type='vit', attend_at_middle=True, attend_at_enc=[True, True, True],
                                  attn_dim_head=8, attend_at_enc_heads=[2, 2, 2], attend_at_middle_heads=2, att_localvit=local,
                                  deep_feature=True, batch_sample=True, batch_sample_factor=1)
        unet = r3.SRUnet256(**kw)
        sd = fill(unet, seed=3)
        unet.eval()                    # dropouts are identities in eval mode
        x = torch.randn(1, 1, S, S, S, generator=g)
        lr = torch.randn(1, 1, S, S, S, generator=g)
        times = torch.rand(1, generator=g)
        log_snr = r3.alpha_cosine_log_snr(times)
        for p in unet.parameters():
            p.requires_grad_(True)
        y = unet(x, times, log_snr, lowres_cond_img=lr)
        (y ** 2).mean().backward()
        named = dict(unet.named_parameters())
        gk = ['downs.0.2.patch_embedding.positions', 'downs.0.2.transformer_encoder.layers.0.block.0.fn.1.qkv.weight',
              'downs.1.2.transformer_encoder.layers.0.block.0.fn.1.projection.bias',
              'mid_attn.transformer_encoder.layers.0.block.1.fn.0.weight', 'mid_attn.reconstruction.4.g',
              'downs.2.2.reconstruction.3.pointwise.weight', 'init_conv.weight']
        gk.append('downs.0.2.transformer_encoder.layers.0.block.1.fn.1.' + ('up_proj.1.weight' if local else 'net.3.weight'))
        grads = {('grad:' + k): named[k].grad for k in gk}
        save(f"unetA_attn_vit_{tag}", x=x, lowres=lr, times=times, log_snr=log_snr, y=y.detach(),
             keys=np.array(list(sd.keys())), shapes=np.array([json.dumps(list(v.shape)) for v in sd.values()]),
             kwargs=json.dumps(kw), **grads)
