"""Family B on a real MI355X: new kernels vs plain PyTorch, product Unet3D / ElucidatedImagen vs the committed golden
vectors of the real reference.  fp32 tolerances as in tests/test_gpu_unet.py."""
import json
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import iqt_oracle as O
from oracle import iqt_oracle_b as OB
from tests.conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda"
T = lambda a: torch.from_numpy(np.asarray(a))


def close(got, ref, tol, what=""):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    assert got.shape == ref.shape, f"{what}: {tuple(got.shape)} vs {tuple(ref.shape)}"
    err, scale = (got - ref).abs().max().item(), ref.abs().max().item()
    assert err <= tol * scale + 1e-6, f"{what}: max err {err:.3e}, scale {scale:.3e}"


def cl(x):
    return x.permute(0, 2, 3, 4, 1).contiguous().to(DEV)


def cf(x):
    return x.detach().cpu().permute(0, 4, 1, 2, 3).contiguous()


def test_causal_temporal_convs():
    from diffusioniqt_amd import ops
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 12, 6, 4, 4, generator=g)
    w = torch.randn(12, 12, 3, 1, 1, generator=g) * 0.3
    b = torch.randn(12, generator=g)
    xr, wr, br = x.double().requires_grad_(), w.double().requires_grad_(), b.double().requires_grad_()
    yr = F.conv3d(F.pad(xr, (0, 0, 0, 0, 2, 0)), wr, br)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy.double())
    xd, wd, bd = cl(x).requires_grad_(), w.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
    y = ops.conv3d(xd, wd, bd, (2, 0, 0), extra_pad=(-2, 0, 0))
    close(cf(y), yr, 2e-5, "causal dense temporal conv")
    y.backward(cl(dy))
    close(cf(xd.grad), xr.grad, 2e-5, "dx")
    close(wd.grad, wr.grad, 5e-5, "dw")
    close(bd.grad, br.grad, 5e-5, "db")
    wdw = torch.randn(12, 1, 3, 1, 1, generator=g)
    xr2, wr2 = x.double().requires_grad_(), wdw.double().requires_grad_()
    yr2 = F.conv3d(F.pad(xr2, (0, 0, 0, 0, 2, 0)), wr2, None, groups=12)
    yr2.backward(dy.double())
    xd2, wd2 = cl(x).requires_grad_(), wdw.to(DEV).requires_grad_()
    y2 = ops.conv3d_direct(xd2, wd2, None, 1, (2, 0, 0), 12, extra_pad=(-2, 0, 0))
    close(cf(y2), yr2, 2e-5, "causal depthwise PEG")
    y2.backward(cl(dy))
    close(cf(xd2.grad), xr2.grad, 2e-5, "peg dx")
    close(wd2.grad, wr2.grad, 1e-4, "peg dw")


def test_layernorm_bias_shuffles_transpose_resize():
    from diffusioniqt_amd import ops
    g = torch.Generator().manual_seed(2)
    x = torch.randn(3, 5, 24, generator=g)
    w, b = torch.randn(24, generator=g), torch.randn(24, generator=g)
    xr, wr, br = x.double().requires_grad_(), w.double().requires_grad_(), b.double().requires_grad_()
    yr = F.layer_norm(xr, (24,), wr, br)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy.double())
    xd, wd, bd = x.to(DEV).requires_grad_(), w.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
    y = ops.chan_layernorm(xd, wd, 1e-5, bias=bd)
    close(y, yr, 2e-5, "ln")
    y.backward(dy.to(DEV))
    close(xd.grad, xr.grad, 1e-4, "ln dx"); close(wd.grad, wr.grad, 1e-4, "ln dw"); close(bd.grad, br.grad, 1e-4, "ln db")

    v = torch.randn(2, 3, 4, 6, 8, generator=g)              # b c f h w
    ref = v.reshape(2, 3, 4, 3, 2, 4, 2).permute(0, 1, 4, 6, 2, 3, 5).reshape(2, 12, 4, 3, 4)
    got = ops.space_to_depth_nd(cl(v), (1, 2, 2))
    assert torch.equal(cf(got), ref)
    assert torch.equal(cf(ops.depth_to_space_nd(got, (1, 2, 2))), v)
    u = torch.randn(2, 8, 3, 4, 5, generator=g)
    ps = F.pixel_shuffle(u.permute(0, 2, 1, 3, 4).reshape(6, 8, 4, 5), 2).reshape(2, 3, 2, 8, 10).permute(0, 2, 1, 3, 4)
    assert torch.equal(cf(ops.depth_to_space_nd(cl(u), (1, 2, 2))), ps)
    t = torch.randn(2, 3, 5, 7, generator=g)
    assert torch.equal(ops.transpose_mid(t.to(DEV)).cpu(), t.transpose(1, 2).contiguous())
    z = torch.randn(1, 2, 4, 4, 4, generator=g)
    assert torch.equal(cf(ops.nearest_resize(cl(z), (8, 8, 8))), F.interpolate(z, (8, 8, 8), mode='nearest'))
    assert torch.equal(cf(ops.nearest_resize(cl(z), (4, 6, 2))), F.interpolate(z, (4, 6, 2), mode='nearest'))


@pytest.mark.parametrize("causal,with_rel,E", [(True, True, 1), (False, False, 5), (False, False, 1)])
def test_attn_softmax(causal, with_rel, E):
    from diffusioniqt_amd import ops
    g = torch.Generator().manual_seed(3)
    G, n, h = 3, 7, 2
    M = E + n
    sim = torch.randn(G, n, h, M, generator=g)
    rel = torch.randn(2 * n - 1, h, generator=g) if with_rel else None
    nb = torch.randn(h, generator=g) if with_rel else None
    sr = sim.double().requires_grad_()
    relr = rel.double().requires_grad_() if with_rel else None
    nbr = nb.double().requires_grad_() if with_rel else None
    s2 = sr
    if with_rel:
        idx = torch.arange(n)[:, None] - torch.arange(n)[None, :] + n - 1
        bias = relr[idx]                                             # [i, j, h]
        full = torch.cat((torch.zeros(n, E - 1, h, dtype=torch.double), nbr[None, None, :].expand(n, 1, h), bias), dim=1)
        s2 = s2 + full.permute(0, 2, 1)[None]
    if causal:
        mask = torch.ones(n, M, dtype=torch.bool).triu(M - n + 1)
        s2 = s2.masked_fill(mask[None, :, None, :], -torch.finfo(torch.float64).max)
    pr = s2.softmax(dim=-1)
    dp = torch.randn(pr.shape, generator=g)
    pr.backward(dp.double())
    sd = sim.to(DEV).requires_grad_()
    reld = rel.to(DEV).requires_grad_() if with_rel else None
    nbd = nb.to(DEV).requires_grad_() if with_rel else None
    p = ops.attn_softmax(sd, reld, nbd, n, h, E, n, causal)
    close(p, pr, 2e-5, "attn softmax")
    p.backward(dp.to(DEV))
    close(sd.grad, sr.grad, 1e-4, "dsim")
    if with_rel:
        close(reld.grad, relr.grad, 1e-4, "drel"); close(nbd.grad, nbr.grad, 1e-4, "dnull")


def build(g, seed=11):
    from diffusioniqt_amd.imagen_video import Unet3D
    kw = json.loads(str(g['kwargs']))
    kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in kw.items()}
    unet = Unet3D(**kw)
    ref_keys = [str(k) for k in g['keys']]
    assert list(unet.state_dict().keys()) == ref_keys, set(ref_keys) ^ set(unet.state_dict().keys())
    shapes = [tuple(json.loads(str(s))) for s in g['shapes']]
    assert [tuple(v.shape) for v in unet.state_dict().values()] == shapes
    unet.load_state_dict(O.hash_fill_state_dict(unet.state_dict(), seed))
    return unet.to(DEV), kw


def test_unet3d_forward_and_grads_match_reference_golden():
    g = load_golden('unet3d_tiny')
    unet, kw = build(g)
    unet.train()
    y = unet(T(g['x']).to(DEV), T(g['time']).to(DEV), lowres_cond_img=T(g['lowres']).to(DEV), lowres_noise_times=T(g['lowres_times']).to(DEV))
    close(y, T(g['y']), 3e-4, "unet3d fwd")
    (y ** 2).mean().backward()
    named = dict(unet.named_parameters())
    for k in g:
        if k.startswith('grad:'):
            assert named[k[5:]].grad is not None, k
            close(named[k[5:]].grad, T(g[k]), 2e-3, k)
    unused = set(str(u) for u in g['unused'])
    for k, p in named.items():
        assert (p.grad is None) == (k in unused), k


def make_edm(unet_kw, steps, dynamic=False, percentile=0.95):
    from diffusioniqt_amd.imagen_video import Unet3D
    from diffusioniqt_amd.elucidated_imagen import ElucidatedImagen
    base = Unet3D(**{**unet_kw, 'lowres_cond': False, 'dim_mults': (1, 2), 'layer_attns': False})
    sr = Unet3D(**unet_kw)
    elu = ElucidatedImagen(unets=(base, sr), image_sizes=(8, 8), channels=1, condition_on_text=False, auto_normalize_img=False,
                           cond_drop_prob=0.0, num_sample_steps=steps, dynamic_thresholding=dynamic,
                           dynamic_thresholding_percentile=percentile)
    u = elu.unets[1]
    u.load_state_dict(O.hash_fill_state_dict(u.state_dict(), 11))
    return elu.to(DEV)


def test_edm_sample_and_loss_match_reference_golden():
    gu = load_golden('unet3d_tiny')
    kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in json.loads(str(gu['kwargs'])).items()}
    g = load_golden('edm_sample')
    elu = make_edm(kw, 3)
    noise = [T(g['lowres_noise']), T(g['init_noise'])] + list(T(g['step_noise']))
    img = elu.sample(batch_size=1, video_frames=8, start_image_or_video=T(g['lowres']).to(DEV), start_at_unet_number=2,
                     use_tqdm=False, noise=noise)
    ref = T(g['img'])
    err = (img.cpu() - ref).abs()
    assert err.max().item() <= 5e-3 and (err > 2e-4).float().mean().item() < 0.03, (err.max().item(), (err > 2e-4).float().mean().item())

    g = load_golden('edm_loss')
    hp = OB.EDM_DEFAULTS
    images = T(g['images'])
    sig = (hp['P_mean'] + hp['P_std'] * T(g['sigma_randn'])).exp()
    elu.unets[1].train()
    loss = elu(images.to(DEV), unet_number=2, noise=T(g['noise']), sigmas=sig, lowres_aug_times=T(g['aug_time']).repeat(2),
               lowres_noise=T(g['lowres_noise']))
    assert abs(loss.item() - float(g['loss'])) <= 1e-4 * abs(float(g['loss'])), (loss.item(), float(g['loss']))
    loss.backward()
    named = dict(elu.unets[1].named_parameters())
    for k in g:
        if k.startswith('grad:'):
            close(named[k[5:]].grad, T(g[k]), 3e-3, k)


def test_edm_sample_dynamic_thresholding_matches_reference_golden():
    gu = load_golden('unet3d_tiny')
    kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in json.loads(str(gu['kwargs'])).items()}
    g = load_golden('edm_sample_dyn')
    elu = make_edm(kw, 3, dynamic=True, percentile=float(g['percentile']))
    noise = [T(g['lowres_noise']), T(g['init_noise'])] + list(T(g['step_noise']))
    img = elu.sample(batch_size=2, video_frames=8, start_image_or_video=T(g['lowres']).to(DEV), start_at_unet_number=2,
                     use_tqdm=False, noise=noise)
    err = (img.cpu() - T(g['img'])).abs()
    assert err.max().item() <= 5e-3 and (err > 2e-4).float().mean().item() < 0.03, (err.max().item(), (err > 2e-4).float().mean().item())


def test_edm_drives_the_true_conv3d_unet_superset():
    """ElucidatedImagen + Family-A Unet: impossible in the reference, supported here; checked against the oracles."""
    from diffusioniqt_amd.imagen_pytorch3D import SRUnet256, NullUnet
    from diffusioniqt_amd.elucidated_imagen import ElucidatedImagen
    ga = load_golden('unetA_tiny')
    kwa = json.loads(str(ga['kwargs']))
    unet = SRUnet256(**kwa)
    elu = ElucidatedImagen(unets=(NullUnet(), unet), image_sizes=(8, 8), channels=1, condition_on_text=False,
                           auto_normalize_img=False, cond_drop_prob=0.0, num_sample_steps=3, dynamic_thresholding=False)
    u = elu.unets[1]
    sd = O.hash_fill_state_dict(u.state_dict(), 0)
    u.load_state_dict(sd)
    elu = elu.to(DEV)
    gen = torch.Generator().manual_seed(5)
    lowres = torch.randn(1, 1, 8, 8, 8, generator=gen).clamp(-1, 1)
    noise = [torch.randn(1, 1, 8, 8, 8, generator=gen) for _ in range(5)]
    img = elu.sample(batch_size=1, video_frames=8, start_image_or_video=lowres.to(DEV), start_at_unet_number=2, use_tqdm=False,
                     noise=[n.clone() for n in noise])
    cfg = O.unet_config(**kwa)
    lt = torch.full((1,), 0.2)
    lr_noisy = OB.lowres_q_sample(lowres, lt, noise[0])
    fn = lambda x, cn: O.unet_forward(sd, cfg, x, None, cn, lowres_cond_img=lr_noisy)
    with torch.no_grad():
        ref = OB.edm_sample(fn, (1, 1, 8, 8, 8), noise[1], noise[2:], dict(OB.EDM_DEFAULTS, num_sample_steps=3))
    err = (img.cpu() - ref).abs()
    assert err.max().item() <= 5e-3 and (err > 2e-4).float().mean().item() < 0.03, err.max().item()


def named_unet(g, keys, shapes, kwargs, seed):
    from diffusioniqt_amd.imagen_video import Unet3D
    kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in json.loads(str(g[kwargs])).items()}
    unet = Unet3D(**kw)
    assert list(unet.state_dict().keys()) == [str(k) for k in g[keys]]
    assert [tuple(v.shape) for v in unet.state_dict().values()] == [tuple(json.loads(str(s))) for s in g[shapes]]
    unet.load_state_dict(O.hash_fill_state_dict(unet.state_dict(), seed))
    return unet, kw


def sample_err(img, ref):
    err = (img.detach().cpu() - ref).abs()
    return err.max().item(), (err > 2e-4).float().mean().item()


def test_c1_exact_config_matches_reference_golden():
    """BASELINE.json configs[0] exactly: Unet3D dim 32 / 16^3 / 10 EDM steps (19 evals, dynamic thresholding
    at the reference default), same weights and noise draws as the reference run that made tests/golden/edm_c1.npz."""
    from diffusioniqt_amd.imagen_video import Unet3D
    from diffusioniqt_amd.elucidated_imagen import ElucidatedImagen
    g = load_golden('edm_c1')
    sr, kw = named_unet(g, 'keys', 'shapes', 'kwargs_sr', 21)
    assert sum(p.numel() for p in sr.parameters()) == int(g['n_params']) == 4712921
    base = Unet3D(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in json.loads(str(g['kwargs_base'])).items()})
    elu = ElucidatedImagen(unets=(base, sr), image_sizes=(16, 16), channels=1, condition_on_text=False, auto_normalize_img=False,
                           cond_drop_prob=0.0, num_sample_steps=10).to(DEV)
    img = elu.sample(batch_size=1, video_frames=16, start_image_or_video=T(g['lowres']).to(DEV), start_at_unet_number=2,
                     use_tqdm=False, noise=list(T(g['draws'])))
    assert tuple(img.shape) == (1, 1, 16, 16, 16)
    mx, frac = sample_err(img, T(g['img']))
    assert mx <= 5e-3 and frac < 0.03, (mx, frac)


def test_c5_cascade_layouts_match_reference_golden():
    """BASELINE.json configs[4] layouts at 8^3 -> 16^3: full two-stage cascade (temporal_downsample_factor (2,1), nearest
    resize between the stages) and start_at_unet_number=2 from an 8^3 volume; fixtures from the real reference."""
    from diffusioniqt_amd.elucidated_imagen import ElucidatedImagen
    g = load_golden('edm_cascade')
    u1, _ = named_unet(g, 'keys1', 'shapes1', 'kwargs1', 22)
    u2, _ = named_unet(g, 'keys2', 'shapes2', 'kwargs2', 23)
    elu = ElucidatedImagen(unets=(u1, u2), image_sizes=(8, 16), channels=1, condition_on_text=False, auto_normalize_img=False,
                           cond_drop_prob=0.0, num_sample_steps=3, temporal_downsample_factor=(2, 1)).to(DEV)
    outs = elu.sample(batch_size=1, video_frames=16, return_all_unet_outputs=True, use_tqdm=False,
                      noise=list(T(g['draws1'])) + list(T(g['draws2'])))
    assert tuple(outs[0].shape) == (1, 1, 8, 8, 8) and tuple(outs[1].shape) == (1, 1, 16, 16, 16)
    mx, frac = sample_err(outs[0], T(g['stage1']))
    assert mx <= 5e-3 and frac < 0.03, ("stage1", mx, frac)
    mx, frac = sample_err(outs[1], T(g['stage2']))
    assert mx <= 1e-2 and frac < 0.05, ("stage2", mx, frac)         # stage 2 is conditioned on stage 1's own round-off
    img = elu.sample(batch_size=1, video_frames=16, start_image_or_video=T(g['lowres']).to(DEV), start_at_unet_number=2,
                     use_tqdm=False, noise=list(T(g['draws3'])))
    mx, frac = sample_err(img, T(g['img_from2']))
    assert mx <= 5e-3 and frac < 0.03, ("from2", mx, frac)


@pytest.mark.parametrize("B,F,H,W,C,heads,dh,causal", [(2, 32, 16, 16, 64, 8, 64, True), (1, 7, 5, 3, 32, 2, 32, True),
                                                       (2, 40, 4, 4, 128, 4, 64, False)])
def test_temporal_attention_walks_the_frame_axis_in_place(B, F, H, W, C, heads, dh, causal):
    """Sampling path of EinopsToAndFrom('b c f h w', '(b h w) f c', Residual(Attention)) (imagen_video.py:1351-1354, 410-525): the
    attention kernel reads q / k|v and writes its output along the frame axis of the channels-last tensors (no transposed copies, the
    null key / value from its own pointer instead of a concatenated kv): bit-identical to the transposed path, which the golden
    fixtures pin against the reference."""
    from diffusioniqt_amd.imagen_video import Attention, Residual, TokensOverTime
    torch.manual_seed(F + C)
    attn = Attention(C, heads=heads, dim_head=dh, causal=causal, rel_pos_bias=True, rel_pos_bias_mlp_depth=2, init_zero=True)
    with torch.no_grad():
        attn.to_out[1].g.normal_()                 # init_zero would make the block the identity
    mod = TokensOverTime(Residual(attn)).to(DEV)
    x = torch.randn(B, F, H, W, C, device=DEV)
    with torch.no_grad():
        assert attn.frames_ok(B * H * W)
        y_new = mod(x)
        attn.frames_ok = lambda G: False
        y_old = mod(x)
    assert torch.isfinite(y_new).all() and (y_new - x).abs().max().item() > 1e-3
    assert torch.equal(y_new, y_old), f"max diff {(y_new - y_old).abs().max().item():.3e}"


# ------------------------------------------------------------------------------------------------------------------------------------
# constructor options beyond the IQT defaults (fixtures of the real reference: oracle/make_golden_opts.py)
# ------------------------------------------------------------------------------------------------------------------------------------
OPTION_CASES = ['memeff', 'tstride_a', 'tstride_b', 'cosine', 'selfcond', 'combine', 'initres', 'condimg']


def _kw(g):
    return {k: (tuple(v) if isinstance(v, list) else v) for k, v in json.loads(str(g['kwargs'])).items()}


@pytest.mark.parametrize('case', OPTION_CASES)
def test_unet3d_constructor_option_matches_reference_golden(case):
    """Unet3D(memory_efficient / temporal_strides / cosine_sim_attn / self_cond / combine_upsample_fmaps /
    init_conv_to_final_conv_residual / cond_images_channels) on the HIP path: state-dict keys and shapes in the reference's order,
    forward, (y^2).mean() gradients and the set of parameters without gradient (imagen_video.py:1176-1213, 1371-1492, 1585-1822)."""
    from diffusioniqt_amd.imagen_video import Unet3D
    g = load_golden(f'unet3d_opt_{case}')
    unet = Unet3D(**_kw(g))
    sd0 = unet.state_dict()
    assert list(sd0.keys()) == [str(k) for k in g['keys']], "state_dict keys / order differ from the reference"
    assert [list(v.shape) for v in sd0.values()] == [json.loads(str(s)) for s in g['shapes']]
    unet.load_state_dict(O.hash_fill_state_dict(sd0, 21))
    unet = unet.to(DEV).train()
    extra = {k: T(g[k]).to(DEV) for k in ('cond_images', 'self_cond') if k in g}
    args = (T(g['x']).to(DEV), T(g['time']).to(DEV))
    kw = dict(lowres_cond_img=T(g['lowres']).to(DEV), lowres_noise_times=T(g['lowres_times']).to(DEV))
    y = unet(*args, **kw, **extra)
    close(y, T(g['y']), 4e-4, f"unet3d {case} fwd")
    (y ** 2).mean().backward()
    named = dict(unet.named_parameters())
    n = 0
    for k in g:
        if k.startswith('grad:'):
            assert named[k[5:]].grad is not None, k
            close(named[k[5:]].grad, T(g[k]), 3e-3, k)
            n += 1
    assert n >= 10
    unused = set(str(u) for u in g['unused'])
    for k, p in named.items():
        assert (p.grad is None) == (k in unused), k
    with torch.no_grad():                                                         # the sampling path (fused kernels) agrees
        unet.eval()
        close(unet(*args, **kw, **extra), T(g['y']), 4e-4, f"unet3d {case} fwd (no grad)")
        if 'y_no_self_cond' in g:
            close(unet(*args, **kw), T(g['y_no_self_cond']), 4e-4, "self_cond input omitted: zeros")


def test_unet3d_options_the_reference_cannot_run_raise():
    from diffusioniqt_amd.imagen_video import Unet3D
    g = load_golden('unet3d_opt_memeff')
    broken = json.loads(str(g['unrunnable']))
    assert broken == {'nearest_upsample': 'RuntimeError', 'cross_embed_downsample': 'TypeError', 'use_linear_attn': 'EinopsError'}
    base = {k: v for k, v in _kw(g).items() if k != 'memory_efficient'}
    for over in (dict(pixel_shuffle_upsample=False), dict(cross_embed_downsample=True), dict(use_linear_attn=True)):
        with pytest.raises(NotImplementedError):
            Unet3D(**{**base, **over})


def test_edm_self_conditioning_sample_and_loss_match_reference_golden(monkeypatch):
    """ElucidatedImagen over a self-conditioning Unet3D: the x0 estimate is fed back at every evaluation of the Heun sampler
    (elucidated_imagen.py:483-524), and training runs a gradient-free estimate first on half of the steps (:847-860)."""
    from diffusioniqt_amd.imagen_video import Unet3D
    from diffusioniqt_amd import elucidated_imagen as EI
    g = load_golden('edm_selfcond')
    kw = _kw(g)
    base = Unet3D(**{**kw, 'lowres_cond': False, 'layer_attns': False, 'self_cond': False})
    sr = Unet3D(**kw)
    elu = EI.ElucidatedImagen(unets=(base, sr), image_sizes=(8, 8), channels=1, condition_on_text=False, auto_normalize_img=False,
                              cond_drop_prob=0.0, num_sample_steps=3, dynamic_thresholding=False)
    u = elu.unets[1]
    u.load_state_dict(O.hash_fill_state_dict(u.state_dict(), 23))
    elu = elu.to(DEV)
    noise = [T(g['lr_noise']), T(g['init_noise'])] + list(T(g['step_noise']))
    img = elu.sample(batch_size=1, video_frames=4, start_image_or_video=T(g['lowres']).to(DEV), start_at_unet_number=2, use_tqdm=False,
                     noise=noise)
    err = (img.cpu() - T(g['img'])).abs()
    assert err.max().item() <= 5e-3 and (err > 2e-4).float().mean().item() < 0.03, (err.max().item(), (err > 2e-4).float().mean().item())
    elu.unets[1].train()
    for tag, draw in (('on', 0.0), ('off', 0.9)):
        monkeypatch.setattr(EI, 'random', lambda d=draw: d)
        elu.unets[1].zero_grad(set_to_none=True)
        loss = elu(T(g['images']).to(DEV), unet_number=2, noise=T(g['loss_noise']), sigmas=T(g['sigmas']),
                   lowres_aug_times=T(g['aug_t']).repeat(2), lowres_noise=T(g['loss_lr_noise']))
        ref = float(g[f'loss_{tag}'])
        assert abs(loss.item() - ref) <= 2e-4 * abs(ref), (tag, loss.item(), ref)
        loss.backward()
        named = dict(elu.unets[1].named_parameters())
        for k in g:
            if k.startswith(f'grad_{tag}:'):
                close(named[k.split(':', 1)[1]].grad, T(g[k]), 3e-3, k)


def test_gate_residual_hands_groupnorm_statistics_to_the_next_block():
    """Sampling path: h * gate + res also emits per-workgroup column sums of its output (diqt_gate_residual_fwd_stats), which the next
    block's GroupNorm uses instead of a pass over the tensor."""
    from diffusioniqt_amd import ops
    g = torch.Generator().manual_seed(0)
    B, Fr, S, C = 2, 6, 12, 64
    h, res = torch.randn(B, Fr, S, S, C, generator=g).to(DEV), torch.randn(B, Fr, S, S, C, generator=g).to(DEV)
    gate = torch.rand(B, C, generator=g).to(DEV)
    gamma, beta = torch.randn(C, generator=g).to(DEV), torch.randn(C, generator=g).to(DEV)
    with torch.no_grad():
        y = ops.gate_residual(h, gate, res)
        assert getattr(y, "_diqt_stats", None) is not None
        assert torch.equal(y, h * gate.view(B, 1, 1, 1, C) + res)
        a = ops.groupnorm_act(y, gamma, beta, None, 8, ops.ACT_SILU, 1e-5)
        b = ops.groupnorm_act(y.clone(), gamma, beta, None, 8, ops.ACT_SILU, 1e-5)         # the clone carries no statistics
    assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item()


@pytest.mark.parametrize("lp", [None, torch.float16])
def test_sampling_replays_the_unet_as_a_graph_bit_identically(lp):
    """ElucidatedImagen.sample: after two eager U-Net evaluations the third and later ones of a stage replay a captured hipGraph
    (graphs.py).  Same kernels, same arguments: the sample must be BIT-IDENTICAL to the all-eager run; an optimizer step
    (parameter version change) retires the graph."""
    import contextlib
    from diffusioniqt_amd import graphs
    from diffusioniqt_amd.imagen_video import Unet3D
    from diffusioniqt_amd.elucidated_imagen import ElucidatedImagen
    torch.manual_seed(0)
    kw = dict(dim=16, dim_mults=(1, 2), channels=1, cond_on_text=False, text_embed_dim=None, layer_attns=False, layer_cross_attns=False,
              attend_at_middle=True, num_resnet_blocks=1, attn_pool_text=False, attn_heads=2, attn_dim_head=16)
    u1, u2 = Unet3D(**{**kw, 'lowres_cond': False}), Unet3D(**{**kw, 'lowres_cond': True})
    for u in (u1, u2):
        for p in u.final_conv.parameters():
            torch.nn.init.normal_(p, std=0.05)
    elu = ElucidatedImagen(unets=(u1, u2), image_sizes=(8, 16), channels=1, condition_on_text=False, auto_normalize_img=False,
                           num_sample_steps=4, temporal_downsample_factor=(2, 1)).to(DEV)
    B, Fr = 2, 8
    g = torch.Generator().manual_seed(1)
    noise = [[torch.randn(B, 1, Fr // 2, 8, 8, generator=g) for _ in range(5)],                       # init + 4 steps
             [torch.randn(B, 1, Fr, 16, 16, generator=g) for _ in range(6)]]                          # lowres noise + init + 4 steps
    ctx = (lambda: torch.autocast('cuda', dtype=lp)) if lp is not None else contextlib.nullcontext

    def run():
        with ctx():
            return elu.sample(batch_size=B, video_frames=Fr, use_tqdm=False, noise=[t for n in noise for t in n]).cpu()
    was, wasf = graphs.ENABLED, graphs.FORCE
    try:
        graphs.FORCE = True                               # capture even if this tiny net were not launch-bound
        graphs.ENABLED = False
        eager = run()
        graphs.ENABLED = True
        elu._graphs.clear(); elu._graphs.replays = 0
        got = run()
        n1 = elu._graphs.replays
        assert n1 == 2 * (7 - 2), n1                      # 7 evals per stage (4 Heun steps: 2 N - 1), the first two eager
        assert all(not e["failed"] for e in elu._graphs.entries.values())
        assert torch.equal(got, eager)
        again = run()                                     # the graphs are reused: all 14 evals replay
        assert elu._graphs.replays == n1 + 14 and torch.equal(again, eager)
        with torch.no_grad():
            u2.final_conv.weight.mul_(1.5)                # a parameter changes: its graph is retired, the result follows the new weights
        graphs.ENABLED = False
        eager2 = run()
        graphs.ENABLED = True
        got2 = run()
        assert torch.equal(got2, eager2) and not torch.equal(got2, eager)
    finally:
        graphs.ENABLED, graphs.FORCE = was, wasf


@pytest.mark.parametrize("B,n,C,scale", [(2, 5000, 64, 1.0), (1, 777, 128, 4.0), (3, 64, 256, 1.0), (2, 70000, 64, 20.0)])
def test_global_context_pooling_in_one_pass(B, n, C, scale):
    """diqt_softmax_pool: pooled = sum_n softmax_n(x . w) x in one pass (online soft-max; large logits exercise the rescaling), against
    float64; and GlobalContext on the sampling path against its three-kernel form (autograd path)."""
    from diffusioniqt_amd import ops
    from diffusioniqt_amd.imagen_video import GlobalContext
    g = torch.Generator().manual_seed(n)
    x = torch.randn(B, n, C, generator=g)
    w = torch.randn(C, generator=g) * scale / C ** 0.5
    with torch.no_grad():
        got = ops.softmax_pool_nograd(x.to(DEV), w.to(DEV))
    assert got is not None
    p = (x.double() @ w.double()).softmax(dim=-1)
    ref = torch.einsum('bn,bnc->bc', p, x.double())
    assert (got.cpu().double() - ref).abs().max().item() <= 2e-5 * max(ref.abs().max().item(), 1e-3)
    if C == 64 and n == 5000:
        gc = GlobalContext(dim_in=C, dim_out=C).to(DEV).eval()
        xv = x.reshape(B, 10, 20, 25, C).to(DEV)
        with torch.no_grad():
            fast = gc(xv)
        slow = gc(xv)                                    # autograd on: to_k conv -> softmax -> weighted_pool
        assert (fast - slow.detach()).abs().max().item() <= 1e-5
