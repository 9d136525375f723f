"""TEST INFRASTRUCTURE (container-only): fixtures for BASELINE.json configs[0] (C1) and the cascade layout of configs[4]
(C5) from the REAL reference.  Run: python oracle/make_golden_c1c5.py   (needs /root/reference; CPU; ~1 min).  Numbers only.

* ``edm_c1``      : SURVEY.md §8 C1, exactly: ElucidatedImagen((Unet3D dim 32 mults (1,2), Unet3D dim 32 mults (1,2,4) lowres_cond,
                    attention at the last level + middle), image_sizes (16,16), 10 EDM steps), ``sample(batch_size=1,
                    video_frames=16, start_image_or_video=lr, start_at_unet_number=2)`` with injected noise.
* ``edm_cascade`` : the C5 layouts at 8^3 -> 16^3, dim 16, 3 steps per stage, temporal_downsample_factor (2,1):
                    (i) the full generative cascade (stage 1 un-conditioned at 8 frames x 8x8, stage 2 conditioned on its
                    nearest-upsampled output), (ii) ``start_at_unet_number=2`` from an 8^3 low-resolution volume.
"""
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


def run_sample(elu, queue, **kw):
    o_randn, o_like = torch.randn, torch.randn_like
    torch.randn = lambda *a, **k: queue.pop(0).clone()
    torch.randn_like = lambda *a, **k: queue.pop(0).clone()
    try:
        out = elu.sample(use_tqdm=False, **kw)
    finally:
        torch.randn, torch.randn_like = o_randn, o_like
    assert len(queue) == 0, len(queue)
    return out


if __name__ == "__main__":
    r3, rv, re_, rt = ref_shim.import_reference()
    torch.set_num_threads(8)
    g = torch.Generator().manual_seed(31)

    # ---------------- C1 exactly ----------------
    S, N = 16, 10
    kw_sr = dict(dim=32, dim_mults=(1, 2, 4), channels=1, cond_on_text=False, text_embed_dim=None, lowres_cond=True,
                 layer_attns=(False, False, True), layer_cross_attns=False, attend_at_middle=True, num_resnet_blocks=1,
                 attn_pool_text=False)
    kw_base = dict(kw_sr, dim_mults=(1, 2), lowres_cond=False, layer_attns=False)
    elu = re_.ElucidatedImagen(unets=(rv.Unet3D(**kw_base), rv.Unet3D(**kw_sr)), image_sizes=(S, S), channels=1,
                               condition_on_text=False, auto_normalize_img=False, cond_drop_prob=0.0, num_sample_steps=N)
    u = elu.unets[1]
    assert sum(p.numel() for p in u.parameters()) == 4712921          # SURVEY.md §8 C1 probe
    u.load_state_dict(hash_fill_state_dict(u.state_dict(), 21))
    lowres = torch.randn(1, 1, S, S, S, generator=g).clamp(-1, 1)
    draws = [torch.randn(1, 1, S, S, S, generator=g) for _ in range(2 + N)]     # lowres aug noise, init, one per step
    img = run_sample(elu, [d for d in draws], batch_size=1, video_frames=S, start_image_or_video=lowres, start_at_unet_number=2)
    sdk = u.state_dict()
    save("edm_c1", keys=np.array(list(sdk.keys())), shapes=np.array([json.dumps(list(v.shape)) for v in sdk.values()]),
         kwargs_sr=json.dumps(kw_sr), kwargs_base=json.dumps(kw_base), lowres=lowres, draws=torch.stack(draws),
         img=img, n_params=np.int64(4712921))

    # ---------------- C5 layouts, tiny ----------------
    S1, S2, N = 8, 16, 3
    kw2 = unet3d_kwargs()
    kw1 = unet3d_kwargs(lowres_cond=False, dim_mults=(1, 2), layer_attns=False)
    elu = re_.ElucidatedImagen(unets=(rv.Unet3D(**kw1), rv.Unet3D(**kw2)), image_sizes=(S1, S2), channels=1,
                               condition_on_text=False, auto_normalize_img=False, cond_drop_prob=0.0, num_sample_steps=N,
                               temporal_downsample_factor=(2, 1))
    elu.unets[0].load_state_dict(hash_fill_state_dict(elu.unets[0].state_dict(), 22))
    elu.unets[1].load_state_dict(hash_fill_state_dict(elu.unets[1].state_dict(), 23))
    # (i) stage 1: init + N step draws at [1,1,8,8,8]; stage 2: lowres aug noise + init + N step draws at [1,1,16,16,16]
    d1 = [torch.randn(1, 1, S1, S1, S1, generator=g) for _ in range(1 + N)]
    d2 = [torch.randn(1, 1, S2, S2, S2, generator=g) for _ in range(2 + N)]
    outs = run_sample(elu, d1 + d2, batch_size=1, video_frames=S2, return_all_unet_outputs=True)
    assert tuple(outs[0].shape) == (1, 1, S1, S1, S1) and tuple(outs[1].shape) == (1, 1, S2, S2, S2)
    # (ii) start at unet 2 from an 8^3 volume (resized by the reference to 16 frames x 16 x 16)
    lowres = torch.randn(1, 1, S1, S1, S1, generator=g).clamp(-1, 1)
    d3 = [torch.randn(1, 1, S2, S2, S2, generator=g) for _ in range(2 + N)]
    img2 = run_sample(elu, [d for d in d3], batch_size=1, video_frames=S2, start_image_or_video=lowres, start_at_unet_number=2)
    sd1, sd2 = elu.unets[0].state_dict(), elu.unets[1].state_dict()
    save("edm_cascade", keys1=np.array(list(sd1.keys())), shapes1=np.array([json.dumps(list(v.shape)) for v in sd1.values()]),
         keys2=np.array(list(sd2.keys())), shapes2=np.array([json.dumps(list(v.shape)) for v in sd2.values()]),
         kwargs1=json.dumps(kw1), kwargs2=json.dumps(kw2), draws1=torch.stack(d1), draws2=torch.stack(d2),
         stage1=outs[0], stage2=outs[1], lowres=lowres, draws3=torch.stack(d3), img_from2=img2)
