"""TEST INFRASTRUCTURE (container-only): Family-B fixtures (imagen_video.Unet3D + ElucidatedImagen) from the REAL
reference.  Run: python oracle/make_golden_b.py   (needs /root/reference; CPU; ~30 s).  Numbers only."""
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


def unet3d_kwargs(**over):
    """SURVEY.md §8 C1 canonical kwargs at dim 16."""
    kw = dict(dim=16, dim_mults=(1, 2, 4), channels=1, cond_on_text=False, text_embed_dim=None, lowres_cond=True,
              layer_attns=(False, False, True), layer_cross_attns=False, attend_at_middle=True, num_resnet_blocks=1,
              attn_pool_text=False, attn_heads=2, attn_dim_head=8)
    kw.update(over)
    return kw


GRAD_KEYS = ['init_conv.convs.0.weight', 'init_conv.convs.2.weight', 'init_temporal_peg.fn.1.weight',
             'init_temporal_attn.fn.fn.to_q.weight', 'init_temporal_attn.fn.fn.to_kv.weight',
             'init_temporal_attn.fn.fn.null_kv', 'init_temporal_attn.fn.fn.null_attn_bias',
             'init_temporal_attn.fn.fn.rel_pos_bias.mlp.0.0.weight', 'init_temporal_attn.fn.fn.rel_pos_bias.mlp.2.weight',
             'init_temporal_attn.fn.fn.to_out.1.g', 'to_time_tokens.0.weight', 'to_lowres_time_cond.0.weight',
             'norm_cond.weight', 'norm_cond.bias', 'downs.0.1.block1.project.spatial_conv.weight',
             'downs.0.1.block1.project.temporal_conv.weight', 'downs.0.2.0.gca.to_k.weight', 'downs.0.2.0.gca.net.2.weight',
             'downs.0.7.1.weight', 'downs.2.3.layers.0.0.fn.to_context.1.weight', 'downs.2.3.layers.0.0.fn.to_context.0.bias',
             'downs.2.3.layers.0.1.4.weight', 'downs.2.7.fns.0.weight', 'mid_block1.cross_attn.fn.to_kv.weight',
             'mid_block1.cross_attn.fn.null_kv', 'mid_attn.fn.fn.to_out.0.weight', 'ups.0.6.net.0.weight',
             'ups.2.1.0.res_conv.weight', 'final_res_block.gca.net.0.bias', 'final_conv.weight', 'final_conv.bias']


if __name__ == "__main__":
    r3, rv, re_, rt = ref_shim.import_reference()
    torch.set_num_threads(8)
    g = torch.Generator().manual_seed(2024)
    S = 8
    kw = unet3d_kwargs()
    base = rv.Unet3D(**unet3d_kwargs(lowres_cond=False, dim_mults=(1, 2), layer_attns=False))
    sr = rv.Unet3D(**kw)
    elu = re_.ElucidatedImagen(unets=(base, sr), image_sizes=(S, S), channels=1, condition_on_text=False,
                               auto_normalize_img=False, cond_drop_prob=0.0, num_sample_steps=3, dynamic_thresholding=False)
    unet = elu.unets[1]
    sd = hash_fill_state_dict(unet.state_dict(), 11)
    unet.load_state_dict(sd)
    meta = dict(keys=np.array(list(sd.keys())), shapes=np.array([json.dumps(list(v.shape)) for v in sd.values()]),
                kwargs=json.dumps(kw))

    # ---- Unet3D forward + grads ----
    x = torch.randn(2, 1, S, S, S, generator=g)
    lr = torch.randn(2, 1, S, S, S, generator=g)
    time = torch.randn(2, generator=g) * 0.5
    ltime = torch.rand(2, generator=g)
    unet.train()
    y = unet(x, time, lowres_cond_img=lr, lowres_noise_times=ltime)
    (y ** 2).mean().backward()
    named = dict(unet.named_parameters())
    grads = {('grad:' + k): named[k].grad for k in GRAD_KEYS}
    unused = [k for k, p in named.items() if p.grad is None]
    save("unet3d_tiny", x=x, lowres=lr, time=time, lowres_times=ltime, y=y.detach(), unused=np.array(unused), **meta, **grads)
    for p in unet.parameters():
        p.grad = None

    # ---- EDM sample, 3 steps, injected noise ----
    B = 1
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
    sig = elu.sample_schedule(3, 7, 0.002, 80)
    save("edm_sample", lowres=lowres, lowres_noise=lr_noise, init_noise=init_noise, step_noise=torch.stack(step_noise),
         img=img, sigmas=sig, sigmas32=elu.sample_schedule(32, 7, 0.002, 80), sigmas10=elu.sample_schedule(10, 7, 0.002, 80),
         lowres_noise_level=0.2)

    # ---- EDM training loss, injected randomness ----
    B = 2
    images = torch.randn(B, 1, S, S, S, generator=g).clamp(-1, 1)
    aug_t = torch.rand(1, generator=g)
    lr_noise = torch.randn(B, 1, S, S, S, generator=g)
    sig_n = torch.randn(B, generator=g)
    noise = torch.randn(B, 1, S, S, S, generator=g)
    elu.lowres_noise_schedule.sample_random_times = lambda b, device: aug_t.clone()
    queue = [lr_noise, sig_n, noise]
    torch.randn = lambda *a, **k: queue.pop(0).clone()
    torch.randn_like = lambda *a, **k: queue.pop(0).clone()
    try:
        unet.train()
        loss = elu(images, unet_number=2)
    finally:
        torch.randn, torch.randn_like = o_randn, o_like
    assert len(queue) == 0
    loss.backward()
    named = dict(elu.unets[1].named_parameters())
    gk = ['final_conv.weight', 'init_conv.convs.1.weight', 'mid_attn.fn.fn.to_q.weight']
    save("edm_loss", images=images, aug_time=aug_t, lowres_noise=lr_noise, sigma_randn=sig_n, noise=noise,
         loss=loss.detach(), **{('grad:' + k): named[k].grad for k in gk})
