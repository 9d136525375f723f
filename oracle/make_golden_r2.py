"""TEST INFRASTRUCTURE (container-only): round-2 fixtures from the REAL reference (needs /root/reference; CPU only).

  unetA_boundary_attn.npz — Family-A SRUnet256 with ``batch_sample=True, batch_sample_factor=3, boundary=True`` AND attention
      on the merged volume at encoder level 0 + the middle (imagen_pytorch3D.py:1610-1622, 1635-1641): 27 sub-volumes of 8^3
      are merged to 24^3 for every attention block and split back, convs pad each sub-volume with its neighbours' voxels
      (boundary_pad, :37-46).  Forward output + a few gradients.

Only numbers are written (inputs, outputs, gradients, parameter names/shapes) — never reference source.
Run:  python oracle/make_golden_r2.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402
from make_golden import unet_kwargs_train_py, fill, save  # noqa: E402


def gen_boundary_attention(r3):
    g = torch.Generator().manual_seed(2718)
    S, dim = 8, 16
    for kind in ('linear', 'softmax'):
        kw = unet_kwargs_train_py(dim, S * 3, dim_mults=(1, 2), num_resnet_blocks=(1, 1), att_type=kind, attn_dim_head=8,
                                  attend_at_enc=[True, False], attend_at_enc_depth=[1, 1], attend_at_enc_heads=[2, 2],
                                  attend_at_middle=True, attend_at_middle_heads=2, deep_feature=True, boundary=True,
                                  batch_sample=True, batch_sample_factor=3)
        unet = r3.SRUnet256(**kw)
        sd = fill(unet, seed=11)
        unet.eval()                                   # Dropout(0.05) in to_q/k/v is the identity in eval mode
        x = torch.randn(27, 1, S, S, S, generator=g)
        lr = torch.randn(27, 1, S, S, S, generator=g)
        times = torch.rand(1, generator=g).repeat(27)  # one t for all sub-volumes (imagen_pytorch3D.py:2424-2425)
        log_snr = r3.alpha_cosine_log_snr(times)
        for p in unet.parameters():
            p.requires_grad_(True)
        y = unet(x, times, log_snr, lowres_cond_img=lr)
        (y ** 2).mean().backward()
        named = dict(unet.named_parameters())
        gk = [k for k in ('init_conv.weight', 'downs.0.1.block1.project.weight', 'downs.0.2.layers.0.0.to_q.1.weight',
                          'downs.0.2.layers.0.0.to_out.0.weight', 'mid_attn.layers.0.1.1.weight', 'mid_block.block1.project.weight',
                          'downs.1.3.0.block2.project.weight', 'final_conv.weight') if k in named and named[k].grad is not None]
        assert len(gk) >= 6, gk
        grads = {('grad:' + k): named[k].grad for k in gk}
        save(f"unetA_boundary_attn_{kind}", x=x, lowres=lr, times=times, log_snr=log_snr, y=y.detach(),
             keys=np.array(list(sd.keys())), shapes=np.array([json.dumps(list(v.shape)) for v in sd.values()]),
             kwargs=json.dumps(kw), **grads)


if __name__ == "__main__":
    r3, rv, re_, rt = ref_shim.import_reference()
    torch.set_num_threads(8)
    gen_boundary_attention(r3)
