"""TEST INFRASTRUCTURE (container-only): EDM sampler with dynamic thresholding (elucidated_imagen.py:298-311), produced by
running the REAL reference.   Run:  python oracle/make_golden_b_next.py"""
import os
import sys
import json
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402
from iqt_oracle import hash_fill_state_dict  # noqa: E402
from make_golden import save  # noqa: E402
from make_golden_b import unet3d_kwargs  # noqa: E402

if __name__ == "__main__":
    r3, rv, re_, rt = ref_shim.import_reference()
    g = torch.Generator().manual_seed(777)
    S = 8
    kw = unet3d_kwargs()
    base = rv.Unet3D(**unet3d_kwargs(lowres_cond=False, dim_mults=(1, 2), layer_attns=False))
    sr = rv.Unet3D(**kw)
    elu = re_.ElucidatedImagen(unets=(base, sr), image_sizes=(S, S), channels=1, condition_on_text=False,
                               auto_normalize_img=False, cond_drop_prob=0.0, num_sample_steps=3, dynamic_thresholding=True,
                               dynamic_thresholding_percentile=0.9)
    unet = elu.unets[1]
    unet.load_state_dict(hash_fill_state_dict(unet.state_dict(), 11))
    B = 2
    lowres = torch.randn(B, 1, S, S, S, generator=g).clamp(-1, 1)
    lr_noise = torch.randn(B, 1, S, S, S, generator=g)
    init_noise = torch.randn(B, 1, S, S, S, generator=g)
    step_noise = [torch.randn(B, 1, S, S, S, generator=g) for _ in range(3)]
    queue = [lr_noise, init_noise] + step_noise
    o_randn, o_like = torch.randn, torch.randn_like
    torch.randn = lambda *a, **k: queue.pop(0).clone()
    torch.randn_like = lambda *a, **k: queue.pop(0).clone()
    try:
        img = elu.sample(batch_size=B, video_frames=S, start_image_or_video=lowres, start_at_unet_number=2, use_tqdm=False)
    finally:
        torch.randn, torch.randn_like = o_randn, o_like
    assert len(queue) == 0
    save("edm_sample_dyn", lowres=lowres, lowres_noise=lr_noise, init_noise=init_noise, step_noise=torch.stack(step_noise),
         img=img, percentile=0.9, lowres_noise_level=0.2)
