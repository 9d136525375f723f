"""Oracle (oracle/iqt_oracle.py) pinned against fixtures produced by the real reference
(oracle/make_golden.py).  CPU only.  Tolerances: the oracle uses the same ATen kernels as the
reference did when the fixtures were made, so agreement is to fp32 round-off (1e-5 abs/rel)."""
import json

import numpy as np
import pytest
import torch

from oracle import iqt_oracle as O
from tests.conftest import load_golden

T = lambda a: torch.from_numpy(np.asarray(a))


def build_sd(g, seed=0):
    keys = [str(k) for k in g['keys']]
    shapes = [tuple(json.loads(str(s))) for s in g['shapes']]
    sd = {k: torch.zeros(s) for k, s in zip(keys, shapes)}
    return O.hash_fill_state_dict(sd, seed)


def cfg_of(g):
    kw = json.loads(str(g['kwargs']))
    return O.unet_config(**kw)


def test_schedules_and_posterior():
    g = load_golden('schedulesA')
    t = T(g['t'])
    assert torch.allclose(O.alpha_cosine_log_snr(t), T(g['cosine']), atol=1e-6, rtol=1e-6)
    assert torch.allclose(O.beta_linear_log_snr(t), T(g['linear']), atol=1e-6, rtol=1e-6)
    mean, var, logvar = O.q_posterior(T(g['post_xs']), T(g['post_xt']), T(g['post_t']), T(g['post_tn']))
    assert torch.allclose(mean, T(g['post_mean']), atol=1e-6)
    assert torch.allclose(var, T(g['post_var']), atol=1e-7)
    assert torch.allclose(logvar, T(g['post_logvar']), atol=1e-5)


def test_subvolume_split_merge_halo_bit_exact():
    g = load_golden('schedulesA')
    vol = T(g['vol'])
    sub = O.convert_volume_to_subvolume(vol, (27, 2, 4, 4, 4))
    assert torch.equal(sub, T(g['sub']))
    assert torch.equal(O.merge_sub_volumes(sub, (1, 2, 12, 12, 12)), T(g['merged']))
    assert torch.equal(O.boundary_pad(sub, 3), T(g['halo']))
    with pytest.raises(ValueError):
        O.merge_sub_volumes(sub[:5], (1, 2, 12, 12, 12))


def test_unet_forward_loss_and_grads():
    g = load_golden('unetA_tiny')
    sd = build_sd(g)
    cfg = cfg_of(g)
    with torch.no_grad():
        y = O.unet_forward(sd, cfg, T(g['x']), T(g['times']), T(g['log_snr']), lowres_cond_img=T(g['lowres']))
    assert torch.allclose(y, T(g['y']), atol=2e-5, rtol=1e-5)

    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    loss, pred, x_noisy, _ = O.p_losses(sdg, cfg, T(g['hr']), T(g['lowres']), T(g['times']), T(g['noise']),
                                        min_bound=float(g['min_bound']))
    assert abs(loss.item() - float(g['loss'])) < 1e-5 * max(1, abs(float(g['loss'])))
    assert torch.allclose(pred, T(g['pred']), atol=2e-5, rtol=1e-5)
    assert torch.allclose(x_noisy, T(g['x_noisy']), atol=1e-6)
    loss.backward()
    for k in g:
        if k.startswith('grad:'):
            ref = T(g[k])
            got = sdg[k[5:]].grad
            assert got is not None, k
            scale = ref.abs().max().item() + 1e-12
            assert (got - ref).abs().max().item() <= 2e-4 * scale + 1e-7, k
    # parameters the reference never touches (mid_block, norm_cond with deep_feature=False)
    unused = set(str(u) for u in g['unused'])
    assert any(u.startswith('mid_block') for u in unused) and 'norm_cond.weight' in unused
    for k in unused:
        assert sdg[k].grad is None, k


@pytest.mark.parametrize('kind', ['linear', 'softmax'])
def test_unet_attention_variants(kind):
    g = load_golden(f'unetA_attn_{kind}')
    sd = build_sd(g, seed=1)
    cfg = cfg_of(g)
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    y = O.unet_forward(sdg, cfg, T(g['x']), T(g['times']), T(g['log_snr']), lowres_cond_img=T(g['lowres']))
    assert torch.allclose(y, T(g['y']), atol=5e-5, rtol=1e-4)
    (y ** 2).mean().backward()
    for k in g:
        if k.startswith('grad:'):
            ref, got = T(g[k]), sdg[k[5:]].grad
            scale = ref.abs().max().item() + 1e-12
            assert (got - ref).abs().max().item() <= 5e-4 * scale + 1e-7, k


@pytest.mark.parametrize('kind', ['linear', 'softmax'])
def test_unet_boundary_with_merged_volume_attention_factor3(kind):
    """batch_sample_factor=3 + boundary=True + attention on the merged 24^3 volume at encoder level 0 and the middle
    (imagen_pytorch3D.py:1610-1622, 1635-1641; fixture: oracle/make_golden_r2.py from the real reference)."""
    g = load_golden(f'unetA_boundary_attn_{kind}')
    sd = build_sd(g, seed=11)
    cfg = cfg_of(g)
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    y = O.unet_forward(sdg, cfg, T(g['x']), T(g['times']), T(g['log_snr']), lowres_cond_img=T(g['lowres']))
    assert torch.allclose(y, T(g['y']), atol=5e-5, rtol=1e-4), (y - T(g['y'])).abs().max()
    (y ** 2).mean().backward()
    for k in g:
        if k.startswith('grad:'):
            ref, got = T(g[k]), sdg[k[5:]].grad
            scale = ref.abs().max().item() + 1e-12
            assert (got - ref).abs().max().item() <= 5e-4 * scale + 1e-7, k


@pytest.mark.parametrize('tag', ['local', 'mlp'])
def test_unet_vit3d_attention(tag):
    """att_type='vit' (ViT3D, imagen_pytorch3D.py:871-910), local (conv) and MLP feed-forward."""
    g = load_golden(f'unetA_attn_vit_{tag}')
    sd = build_sd(g, seed=3)
    cfg = cfg_of(g)
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    y = O.unet_forward(sdg, cfg, T(g['x']), T(g['times']), T(g['log_snr']), lowres_cond_img=T(g['lowres']))
    assert torch.allclose(y, T(g['y']), atol=5e-5, rtol=1e-4)
    (y ** 2).mean().backward()
    for k in g:
        if k.startswith('grad:'):
            # FeedForwardBlock registers its sub-modules twice (own name + `net`): the tensor in effect is the `net.*` entry
            name = k[5:].replace('.up_proj.1.', '.net.0.1.')
            ref, got = T(g[k]), sdg[name].grad
            scale = ref.abs().max().item() + 1e-12
            assert (got - ref).abs().max().item() <= 5e-4 * scale + 1e-7, k


def test_unet_memory_efficient_cross_embed_and_boundary():
    g = load_golden('unetA_memeff')
    sd, cfg = build_sd(g, seed=2), cfg_of(g)
    t = T(g['times'])
    with torch.no_grad():
        y = O.unet_forward(sd, cfg, T(g['x']), t, O.alpha_cosine_log_snr(t), lowres_cond_img=T(g['lowres']))
    assert torch.allclose(y, T(g['y']), atol=2e-5, rtol=1e-5)
    g = load_golden('unetA_boundary')
    sd, cfg = build_sd(g, seed=3), cfg_of(g)
    t = T(g['times'])
    with torch.no_grad():
        y = O.unet_forward(sd, cfg, T(g['x']), t, O.alpha_cosine_log_snr(t), lowres_cond_img=T(g['lowres']))
    assert torch.allclose(y, T(g['y']), atol=2e-5, rtol=1e-5)


def test_ddpm_trajectory():
    g = load_golden('ddpmA_traj')
    gu = load_golden('unetA_tiny')
    sd, cfg = build_sd(gu), cfg_of(gu)
    with torch.no_grad():
        img, noisy, x0 = O.p_sample_loop(sd, cfg, T(g['lowres']), T(g['init_noise']), list(T(g['step_noise'])),
                                         timesteps=int(g['T']), min_bound=float(g['min_bound']))
    assert len(noisy) == int(g['T']) + 1 and len(x0) == int(g['T']) + 1
    assert torch.allclose(img, T(g['img']), atol=1e-4, rtol=1e-4)
    # The fixture was made on CPU, where the reference's `img.cpu().numpy()` (imagen_pytorch3D.py:2148,2152)
    # ALIASES `img`, so its final in-place clamp (:2157) leaks into the last two list entries.  On the
    # reference's intended device (.cpu() copies) they are un-clamped — that is what oracle and product return.
    mb = float(g['min_bound'])
    ref_noisy = T(g['noisy'])
    assert torch.allclose(torch.stack(noisy[:-2]), ref_noisy[:-2], atol=1e-4, rtol=1e-4)
    assert torch.allclose(torch.stack(noisy[-2:]).clamp(min=mb), ref_noisy[-2:], atol=1e-4, rtol=1e-4)
    assert torch.allclose(torch.stack(x0), T(g['x0']), atol=1e-4, rtol=1e-4)


SAMPLER_OPTION_CASES = [
    # tag, objective, dynamic threshold, norm, T, extra kwargs
    ('noise_dyn', 'noise', True, 'z-score', 4, {}),
    ('v_static', 'v', False, 'z-score', 4, {}),
    ('x0_dyn_minmax', 'x_start', True, 'min-max', 4, {}),
    ('skip2', 'x_start', False, 'z-score', 6, {'skip_steps': 2}),
    ('inpaint', 'x_start', False, 'z-score', 3, {'inpaint_resample_times': 1}),
]


@pytest.mark.parametrize("tag,objective,dyn,norm,Tn,extra", SAMPLER_OPTION_CASES)
def test_sampler_options_oracle_vs_reference(tag, objective, dyn, norm, Tn, extra):
    """noise/v objectives, dynamic thresholding (p = 0.9), skip_steps, inpainting."""
    g = load_golden('ddpmA_options')
    gu = load_golden('unetA_tiny')
    sd, cfg = build_sd(gu), cfg_of(gu)
    draws = list(T(g[f'{tag}:draws']))
    kw = dict(extra)
    if tag == 'inpaint':
        kw.update(inpaint_images=T(g['inpaint:images']), inpaint_masks=T(g['inpaint:mask']))
    mb = float(g['min_bound'])
    with torch.no_grad():
        img, noisy, x0 = O.p_sample_loop(sd, cfg, T(g['lowres']), draws[0], draws[1:], timesteps=Tn, min_bound=mb, norm=norm,
                                         pred_objective=objective, dynamic_threshold=dyn, percentile=0.9, **kw)
    clampf = (lambda t: t.clamp(-1., 1.)) if norm == 'min-max' else (lambda t: t.clamp(min=mb))
    assert torch.allclose(img, T(g[f'{tag}:img']), atol=2e-4, rtol=2e-4)
    ref_noisy = T(g[f'{tag}:noisy'])
    assert len(noisy) == ref_noisy.shape[0]
    assert torch.allclose(torch.stack(noisy[:-2]), ref_noisy[:-2], atol=2e-4, rtol=2e-4)
    assert torch.allclose(clampf(torch.stack(noisy[-2:])), ref_noisy[-2:], atol=2e-4, rtol=2e-4)   # CPU numpy aliasing, see above
    assert torch.allclose(torch.stack(x0), T(g[f'{tag}:x0']), atol=2e-4, rtol=2e-4)


@pytest.mark.parametrize("objective", ['noise', 'v'])
def test_noise_and_v_objective_losses_oracle_vs_reference(objective):
    g = load_golden('ddpmA_options')
    gu = load_golden('unetA_tiny')
    sd, cfg = build_sd(gu), cfg_of(gu)
    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    loss, pred, _, _ = O.p_losses(sd, cfg, T(g['hr']), T(g['lowres']), T(g['times']), T(g['noise']),
                                  min_bound=float(g['min_bound']), pred_objective=objective)
    assert abs(loss.item() - float(g[f'loss_{objective}'])) <= 2e-5 * abs(float(g[f'loss_{objective}']))
    assert torch.allclose(pred, T(g[f'pred_{objective}']), atol=2e-4, rtol=2e-4)
    loss.backward()
    for k in ('final_conv.weight', 'init_conv.weight'):
        ref = T(g[f'grad_{objective}:{k}'])
        assert torch.allclose(sd[k].grad, ref, atol=1e-3 * ref.abs().max().item(), rtol=1e-3), k
