"""End-to-end parity of the HIP path on a real MI355X: product modules (diffusioniqt_amd) vs the committed
golden vectors of the real reference (tests/golden) and vs the CPU oracle on fresh seeded inputs.

Tolerance: fp32 everywhere; a U-Net eval chains ~40 convs (K up to 3456) + GroupNorms, so we allow
max-abs error 2e-4 * max|ref| on outputs (rel-L2 is asserted <= 2e-5) and 1e-3 * max|ref| on gradients.
"""
import json

import numpy as np
import pytest
import torch

from oracle import iqt_oracle as O
from tests.conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda"
T = lambda a: torch.from_numpy(np.asarray(a))


def close(got, ref, tol, what=""):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    assert got.shape == ref.shape, f"{what}: {tuple(got.shape)} vs {tuple(ref.shape)}"
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= tol * scale + 1e-6, f"{what}: max err {err:.3e}, scale {scale:.3e}"
    return (got - ref).norm().item() / (ref.norm().item() + 1e-30)


def build(g, seed):
    from diffusioniqt_amd.imagen_pytorch3D import SRUnet256
    kw = json.loads(str(g['kwargs']))
    unet = SRUnet256(**kw)
    sd = O.hash_fill_state_dict(unet.state_dict(), seed)
    unet.load_state_dict(sd)
    return unet.to(DEV), sd, O.unet_config(**kw)


def test_unet_forward_loss_grads_match_reference_golden():
    from diffusioniqt_amd.imagen_pytorch3D import Imagen, NullUnet
    g = load_golden('unetA_tiny')
    unet, sd, cfg = build(g, 0)
    unet.eval()
    with torch.no_grad():
        y = unet(T(g['x']).to(DEV), T(g['times']).to(DEV), T(g['log_snr']).to(DEV), lowres_cond_img=T(g['lowres']).to(DEV))
    rel = close(y, T(g['y']), 2e-4, "unet fwd vs reference")
    assert rel <= 2e-5, rel

    configs = {'Data': {'norm': 'z-score'}, 'Train': {'batch_sample': False}}
    imagen = Imagen(unets=(NullUnet(), unet), configs=configs, min_bound=float(g['min_bound']), image_sizes=(8, 8),
                    channels=1, pred_objectives='x_start', timesteps=4, dynamic_thresholding=False,
                    p2_loss_weight_gamma=0.0, cond_drop_prob=0.0).to(DEV)
    times = T(g['times'])
    imagen.noise_schedulers[1].sample_random_times = lambda b, device: times.clone()
    u = imagen.unets[1]
    u.train()
    loss, pred, x_noisy, _ = imagen(T(g['hr']).to(DEV), lowres_img=T(g['lowres']).to(DEV), unet_number=2,
                                    noise=T(g['noise']).to(DEV))
    assert abs(loss.item() - float(g['loss'])) <= 2e-5 * abs(float(g['loss']))
    close(pred, T(g['pred']), 2e-4, "pred (clamped in place)")
    close(x_noisy, T(g['x_noisy']), 1e-6, "x_noisy")
    loss.backward()
    named = dict(u.named_parameters())
    for k in g:
        if k.startswith('grad:'):
            close(named[k[5:]].grad, T(g[k]), 1e-3, k)
    unused = set(str(s) for s in g['unused'])
    for k, p in named.items():
        assert (p.grad is None) == (k in unused), k        # mid_block / norm_cond never get a gradient


@pytest.mark.parametrize('kind', ['linear', 'softmax'])
def test_unet_attention_variants_match_reference_golden(kind):
    g = load_golden(f'unetA_attn_{kind}')
    unet, sd, cfg = build(g, 1)
    unet.eval()
    y = unet(T(g['x']).to(DEV), T(g['times']).to(DEV), T(g['log_snr']).to(DEV), lowres_cond_img=T(g['lowres']).to(DEV))
    close(y, T(g['y']), 5e-4, f"attn {kind} fwd")
    (y ** 2).mean().backward()
    named = dict(unet.named_parameters())
    for k in g:
        if k.startswith('grad:'):
            close(named[k[5:]].grad, T(g[k]), 2e-3, k)


@pytest.mark.parametrize('kind', ['linear', 'softmax'])
def test_unet_boundary_merged_volume_attention_factor3_matches_reference_golden(kind):
    """27 sub-volumes (factor 3) with neighbour-halo convs (boundary=True) and attention over the MERGED 24^3
    volume at encoder level 0 + the middle (/root/reference/imagen_pytorch3D.py:1610-1622, 1635-1641, 37-46) — forward and
    gradients of the HIP path against the fixture the real reference produced (oracle/make_golden_r2.py)."""
    g = load_golden(f'unetA_boundary_attn_{kind}')
    unet, sd, cfg = build(g, 11)
    assert list(unet.state_dict().keys()) == [str(k) for k in g['keys']]
    unet.eval()
    y = unet(T(g['x']).to(DEV), T(g['times']).to(DEV), T(g['log_snr']).to(DEV), lowres_cond_img=T(g['lowres']).to(DEV))
    rel = close(y, T(g['y']), 5e-4, f"boundary + merged attention ({kind}) fwd")
    assert rel <= 5e-5, rel
    with torch.no_grad():      # sampling path: convs read the neighbours' voxels in place (no merge / pad / split copies): same bits
        y_ng = unet(T(g['x']).to(DEV), T(g['times']).to(DEV), T(g['log_snr']).to(DEV), lowres_cond_img=T(g['lowres']).to(DEV))
    assert torch.equal(y_ng, y.detach()), "neighbour-halo convs differ from the boundary_pad copies"
    (y ** 2).mean().backward()
    named = dict(unet.named_parameters())
    n = 0
    for k in g:
        if k.startswith('grad:'):
            close(named[k[5:]].grad, T(g[k]), 2e-3, k)
            n += 1
    assert n >= 6


@pytest.mark.parametrize('tag', ['local', 'mlp'])
def test_unet_vit3d_attention_matches_reference_golden(tag):
    g = load_golden(f'unetA_attn_vit_{tag}')
    unet, sd, cfg = build(g, 3)
    assert list(unet.state_dict().keys()) == [str(k) for k in g['keys']], "ViT3D state_dict keys/order differ from the reference"
    unet.eval()
    y = unet(T(g['x']).to(DEV), T(g['times']).to(DEV), T(g['log_snr']).to(DEV), lowres_cond_img=T(g['lowres']).to(DEV))
    rel = close(y, T(g['y']), 3e-4, "vit unet fwd vs reference")
    assert rel <= 5e-5, rel
    (y ** 2).mean().backward()
    named = dict(unet.named_parameters())
    for k in g:
        if k.startswith('grad:'):
            close(named[k[5:]].grad, T(g[k]), 2e-3, k)


def test_unet_memory_efficient_cross_embed_and_boundary_match_reference_golden():
    for name, seed in (('unetA_memeff', 2), ('unetA_boundary', 3)):
        g = load_golden(name)
        unet, sd, cfg = build(g, seed)
        unet.eval()
        t = T(g['times'])
        with torch.no_grad():
            y = unet(T(g['x']).to(DEV), t.to(DEV), O.alpha_cosine_log_snr(t).to(DEV), lowres_cond_img=T(g['lowres']).to(DEV))
        close(y, T(g['y']), 2e-4, name)


def test_ddpm_trajectory_matches_reference_golden():
    from diffusioniqt_amd.imagen_pytorch3D import Imagen, NullUnet
    g = load_golden('ddpmA_traj')
    gu = load_golden('unetA_tiny')
    unet, sd, cfg = build(gu, 0)
    configs = {'Data': {'norm': 'z-score'}, 'Train': {'batch_sample': False}}
    mb = float(g['min_bound'])
    imagen = Imagen(unets=(NullUnet(), unet), configs=configs, min_bound=mb, image_sizes=(8, 8), channels=1,
                    pred_objectives='x_start', timesteps=int(g['T']), dynamic_thresholding=False,
                    p2_loss_weight_gamma=0.0, cond_drop_prob=0.0).to(DEV)
    noise = [T(g['init_noise'])] + list(T(g['step_noise']))
    img, noisy, x0 = imagen.sample(batch_size=2, start_image_or_video=T(g['lowres']).to(DEV), start_at_unet_number=2,
                                   use_tqdm=False, noise=noise)
    assert isinstance(noisy, list) and isinstance(noisy[0], np.ndarray) and len(noisy) == int(g['T']) + 1
    close(img, T(g['img']), 5e-4, "sample img")
    ref_noisy = T(g['noisy'])
    close(T(np.stack(noisy[:-2])), ref_noisy[:-2], 5e-4, "noisy list")
    # the CPU-made fixture has its last two entries clamped through numpy aliasing (see tests/test_oracle_golden.py)
    close(T(np.stack(noisy[-2:])).clamp(min=mb), ref_noisy[-2:], 5e-4, "noisy tail")
    close(T(np.stack(x0)), T(g['x0']), 5e-4, "x0 list")


def test_unet_config2_shape_vs_oracle_fresh_inputs():
    """The BASELINE config-2 network (dim 64, mults (1,2,4)) at 16^3, B=2: product (GPU) vs oracle (CPU)."""
    from diffusioniqt_amd.imagen_pytorch3D import SRUnet256
    kw = dict(img_size=16, dim=64, dim_mults=(1, 2, 4), channels=1, num_resnet_blocks=(2, 2, 2), init_conv_kernel_size=3,
              lowres_cond=True, init_cross_embed=False, att_type='linear', attn_dim_head=64, attend_at_middle=False,
              attend_at_enc=[False] * 3, attend_at_enc_depth=[1] * 3, attend_at_enc_heads=[8] * 3, init_dim=64,
              memory_efficient=False, use_se_attn='True,', pixel_shuffle_upsample=True, boundary=False,
              batch_sample=False, batch_sample_factor=3, deep_feature=False)
    unet = SRUnet256(**kw)
    sd = O.hash_fill_state_dict(unet.state_dict(), 5)
    unet.load_state_dict(sd)
    unet = unet.to(DEV).eval()
    assert sum(p.numel() for p in unet.parameters()) == 13606665
    gen = torch.Generator().manual_seed(42)
    x, lr = torch.randn(2, 1, 16, 16, 16, generator=gen), torch.randn(2, 1, 16, 16, 16, generator=gen)
    t = torch.rand(2, generator=gen)
    ls = O.alpha_cosine_log_snr(t)
    with torch.no_grad():
        y = unet(x.to(DEV), t.to(DEV), ls.to(DEV), lowres_cond_img=lr.to(DEV))
        yr = O.unet_forward(sd, O.unet_config(**kw), x, t, ls, lowres_cond_img=lr)
    rel = close(y, yr, 2e-4, "config-2 unet")
    assert rel <= 2e-5, rel


def test_batched_time_mlps_equal_the_per_block_launches():
    """K5: on the sampling path the ~20 per-block time MLPs (imagen_pytorch3D.py:586-589) run as ONE launch over concatenated
    weights; the U-Net output is bit-identical to the per-block path (taken when autograd records), also after a weight update."""
    g = load_golden('unetA_tiny')
    unet, sd, cfg = build(g, 0)
    unet.eval()
    args = (T(g['x']).to(DEV), T(g['times']).to(DEV), T(g['log_snr']).to(DEV))
    lr = T(g['lowres']).to(DEV)
    for rnd in range(2):
        with torch.no_grad():
            y_b = unet(*args, lowres_cond_img=lr)
        assert unet._time_mlps.w is not None and unet._time_mlps.w.shape[1] == 64      # time_cond_dim = 4 * dim
        y_p = unet(*args, lowres_cond_img=lr).detach()
        assert torch.equal(y_b, y_p), f"round {rnd}"
        with torch.no_grad():                      # an in-place weight change must invalidate the packed copy
            for m in unet.modules():
                if hasattr(m, 'time_mlp') and m.time_mlp is not None:
                    m.time_mlp[1].weight.mul_(1.5)


@pytest.mark.parametrize("B,kid", [(2, 4), (4, 4)])
def test_unet_config2_at_32cubed_on_the_headline_conv_kernel_vs_oracle(B, kid):
    """The exact BASELINE config-2 network at its real 32^3 patch size, whole-U-Net composition on the GPU vs ``oracle.unet_forward``
    on the host (same tolerances as the 16^3 case): B=2 fills one round of 256 workgroups of ``conv_fwd9_kernel``'s 256-voxel-tile variant,
    B=4 one round of its 512-voxel-tile variant, which carries the headline's 32^3-level convs; the persistent tile walks and the 8-wave
    ``conv_fwd8_kernel`` (ragged / neighbour-halo launches) are covered at the kernel level (tests/test_gpu_kernels.py)."""
    from bench import unet_kwargs
    from diffusioniqt_amd import _lib
    from diffusioniqt_amd.imagen_pytorch3D import SRUnet256
    assert _lib.query("diqt_conv3d_fwd_kernel_id", B, 32, 32, 32, 64, 64, 3, 3, 3, 1, 1, 1, 0, 0, 0) == kid, \
        "the 64->64 3x3x3 conv at 32^3 no longer dispatches to the kernel this case is meant to cover"
    kw = unet_kwargs(32)
    unet = SRUnet256(**kw)
    sd = O.hash_fill_state_dict(unet.state_dict(), 7)
    unet.load_state_dict(sd)
    unet = unet.to(DEV).eval()
    gen = torch.Generator().manual_seed(4242 + B)
    x, lr = torch.randn(B, 1, 32, 32, 32, generator=gen), torch.randn(B, 1, 32, 32, 32, generator=gen)
    t = torch.rand(B, generator=gen)
    ls = O.alpha_cosine_log_snr(t)
    with torch.no_grad():
        y = unet(x.to(DEV), t.to(DEV), ls.to(DEV), lowres_cond_img=lr.to(DEV))
        yr = O.unet_forward(sd, O.unet_config(**kw), x, t, ls, lowres_cond_img=lr)
    rel = close(y, yr, 2e-4, f"config-2 unet at 32^3, B={B}")
    assert rel <= 2e-5, rel


def test_batched_time_mlp_backward_and_residual_taps_equal_the_unfused_autograd_graph():
    """Training path: the ~20 time-MLP Linears run as one launch with ONE backward launch set (ops._BatchedLinearSmallFn: the GroupNorm
    backward writes d scale/shift into its block of a shared buffer), and a ResnetBlock's residual branch reads x through the GroupNorm's
    alias so that its gradient is added inside the dx kernel.  Against the same network with per-block time MLPs: identical loss bits,
    every gradient equal up to the summation order of the time-embedding gradient, the same set of parameters without gradient."""
    from diffusioniqt_amd.imagen_pytorch3D import BatchedTimeMLPs
    g = load_golden('unetA_tiny')
    unet, sd, cfg = build(g, 0)
    unet.train()
    args = (T(g['x']).to(DEV), T(g['times']).to(DEV), T(g['log_snr']).to(DEV))
    lr = T(g['lowres']).to(DEV)
    res = {}
    for mode in (True, False):
        BatchedTimeMLPs.train_batched = mode
        try:
            unet.zero_grad(set_to_none=True)
            y = unet(*args, lowres_cond_img=lr)
            (y ** 2).mean().backward()
        finally:
            BatchedTimeMLPs.train_batched = True
        res[mode] = (y.detach().clone(), {n: (p.grad.clone() if p.grad is not None else None) for n, p in unet.named_parameters()})
    assert torch.equal(res[True][0], res[False][0])
    for n, ga in res[True][1].items():
        gb = res[False][1][n]
        assert (ga is None) == (gb is None), n
        if ga is not None:
            close(ga, gb, 2e-5, f"grad {n}: batched vs per-block time MLPs")
    assert sum(v is None for v in res[True][1].values()) == len(g['unused'])


_C2_GRAD_ORACLE = {}


@pytest.mark.parametrize("fuse", [False, True], ids=["gn_reduce_own_pass(default)", "gn_reduce_in_conv_epilogue"])
def test_unet_config2_at_32cubed_whole_network_gradients_vs_oracle_in_both_groupnorm_backward_modes(fuse):
    """Whole-network backward of the exact BASELINE config-2 net at its real 32^3 size (B = 2: the 32^3-level backward-data convs run on
    ``conv_fwd9_kernel``, so the fused mode really takes ``diqt_conv3d_fwd_gnbwd``): loss and EVERY parameter gradient against autograd
    of the CPU oracle, once with the GroupNorm-backward reduction as its own pass (the default: what bench.py times) and once in the
    conv epilogue.  Reference: imagen_pytorch3D.py:535-614, 1554-1684."""
    from bench import unet_kwargs
    from diffusioniqt_amd import ops, _lib
    from diffusioniqt_amd.imagen_pytorch3D import SRUnet256
    B = 2
    kw = unet_kwargs(32)
    unet = SRUnet256(**kw)
    sd = O.hash_fill_state_dict(unet.state_dict(), 9)
    unet.load_state_dict(sd)
    unet = unet.to(DEV).train()
    gen = torch.Generator().manual_seed(777)
    x, lr = torch.randn(B, 1, 32, 32, 32, generator=gen), torch.randn(B, 1, 32, 32, 32, generator=gen)
    t = torch.rand(B, generator=gen)
    ls = O.alpha_cosine_log_snr(t)
    if not _C2_GRAD_ORACLE:
        sdg = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
        yr = O.unet_forward(sdg, O.unet_config(**kw), x, t, ls, lowres_cond_img=lr)
        (yr ** 2).mean().backward()
        _C2_GRAD_ORACLE.update(y=yr.detach(), grads={k: v.grad for k, v in sdg.items() if v.grad is not None})
    with ops.gnbwd_fuse(fuse):
        bp = (1, 1, 1)
        took = _lib.query("diqt_conv3d_fwd_gnbwd_blocks", B, 32, 32, 32, 64, 64, 3, 3, 3, *bp, 0, 0, 0) > 0
        assert took == fuse
        y = unet(x.to(DEV), t.to(DEV), ls.to(DEV), lowres_cond_img=lr.to(DEV))
        (y ** 2).mean().backward()
    rel = close(y, _C2_GRAD_ORACLE['y'], 2e-4, "config-2 unet at 32^3 (train mode)")
    assert rel <= 2e-5, rel
    named = dict(unet.named_parameters())
    ref = _C2_GRAD_ORACLE['grads']
    n = 0
    for k, p in named.items():
        if k in ref:
            assert p.grad is not None, k
            close(p.grad, ref[k], 1e-3, f"grad {k} (fuse={fuse})")
            n += 1
        else:
            assert p.grad is None, k
    assert n > 200


SAMPLER_OPTION_CASES = [
    ('noise_dyn', 'noise', True, 'z-score', 4, {}),
    ('v_static', 'v', False, 'z-score', 4, {}),
    ('x0_dyn_minmax', 'x_start', True, 'min-max', 4, {}),
    ('skip2', 'x_start', False, 'z-score', 6, {'skip_steps': 2}),
    ('inpaint', 'x_start', False, 'z-score', 3, {'inpaint_resample_times': 1}),
]


def _imagen_for(gu, objective, dyn, norm, Tn, mb):
    from diffusioniqt_amd.imagen_pytorch3D import Imagen, NullUnet
    unet, sd, cfg = build(gu, 0)
    configs = {'Data': {'norm': norm}, 'Train': {'batch_sample': False}}
    return Imagen(unets=(NullUnet(), unet), configs=configs, min_bound=mb, image_sizes=(8, 8), channels=1,
                  pred_objectives=objective, timesteps=Tn, dynamic_thresholding=dyn, dynamic_thresholding_percentile=0.9,
                  p2_loss_weight_gamma=0.0, cond_drop_prob=0.0).to(DEV)


@pytest.mark.parametrize("tag,objective,dyn,norm,Tn,extra", SAMPLER_OPTION_CASES)
def test_sampler_options_match_reference_golden(tag, objective, dyn, norm, Tn, extra):
    """SURVEY.md §8(f) item 4 on the HIP path: noise/v objectives, dynamic thresholding (radix-select quantile),
    skip_steps and inpainting against fixtures of the real reference (oracle/make_golden_next.py)."""
    g = load_golden('ddpmA_options')
    gu = load_golden('unetA_tiny')
    mb = float(g['min_bound'])
    imagen = _imagen_for(gu, objective, dyn, norm, Tn, mb)
    kw = dict(extra)
    if tag == 'inpaint':
        kw.update(inpaint_images=T(g['inpaint:images']).to(DEV), inpaint_masks=T(g['inpaint:mask']).to(DEV))
    img, noisy, x0 = imagen.sample(batch_size=2, start_image_or_video=T(g['lowres']).to(DEV), start_at_unet_number=2,
                                   use_tqdm=False, noise=list(T(g[f'{tag}:draws'])), **kw)
    clampf = (lambda t: t.clamp(-1., 1.)) if norm == 'min-max' else (lambda t: t.clamp(min=mb))
    # dynamic thresholding divides by the per-sample quantile: looser bound (the selected order statistics are exact,
    # the U-Net outputs they are taken from carry the usual fp32 conv error)
    tol = 2e-3 if dyn else 5e-4
    close(img, T(g[f'{tag}:img']), tol, f"{tag} img")
    ref_noisy = T(g[f'{tag}:noisy'])
    assert len(noisy) == ref_noisy.shape[0]
    close(T(np.stack(noisy[:-2])), ref_noisy[:-2], tol, f"{tag} noisy list")
    close(clampf(T(np.stack(noisy[-2:]))), ref_noisy[-2:], tol, f"{tag} noisy tail")
    close(T(np.stack(x0)), T(g[f'{tag}:x0']), tol, f"{tag} x0 list")


@pytest.mark.parametrize("objective", ['noise', 'v'])
def test_noise_and_v_objective_training_match_reference_golden(objective):
    g = load_golden('ddpmA_options')
    gu = load_golden('unetA_tiny')
    imagen = _imagen_for(gu, objective, False, 'z-score', 4, float(g['min_bound']))
    times = T(g['times'])
    imagen.noise_schedulers[1].sample_random_times = lambda b, device: times.clone()
    u = imagen.unets[1]
    u.train()
    loss, pred, _, _ = imagen(T(g['hr']).to(DEV), lowres_img=T(g['lowres']).to(DEV), unet_number=2, noise=T(g['noise']).to(DEV))
    ref = float(g[f'loss_{objective}'])
    assert abs(loss.item() - ref) <= 2e-5 * abs(ref)
    close(pred, T(g[f'pred_{objective}']), 2e-4, "pred")
    loss.backward()
    named = dict(u.named_parameters())
    for k in ('final_conv.weight', 'init_conv.weight'):
        close(named[k].grad, T(g[f'grad_{objective}:{k}']), 1e-3, f"grad {k}")


def test_abs_quantile_matches_torch_quantile():
    from diffusioniqt_amd import ops
    gen = torch.Generator().manual_seed(3)
    for shape, q in (((3, 1, 8, 8, 8), 0.9), ((2, 4097), 0.95), ((4, 1, 32, 32, 32), 0.95), ((2, 5), 0.5), ((1, 7), 1.0 - 1e-7)):
        x = torch.randn(*shape, generator=gen)
        x[0].view(-1)[:3] = 2.5                     # ties around the selected rank
        ref = torch.quantile(x.flatten(1).abs(), q, dim=-1)
        got = ops.abs_quantile(x.to(DEV), q).cpu()
        assert torch.allclose(got, ref, atol=0, rtol=1e-6), (shape, q, got, ref)


@pytest.mark.parametrize("loss_type", ["l1", "huber"])
def test_imagen_loss_types_match_reference_golden(loss_type):
    """Imagen(loss_type='l1' | 'huber') -> F.l1_loss / F.smooth_l1_loss (imagen_pytorch3D.py:1785-1790, 2370) through the fused
    clamp + loss kernel: loss, clamped prediction and gradients against the real reference."""
    from diffusioniqt_amd.imagen_pytorch3D import Imagen, NullUnet
    g = load_golden('imagenA_loss_types')
    gu = load_golden('unetA_tiny')
    unet, sd, cfg = build(gu, 0)
    configs = {'Data': {'norm': 'z-score'}, 'Train': {'batch_sample': False}}
    imagen = Imagen(unets=(NullUnet(), unet), configs=configs, min_bound=float(gu['min_bound']), image_sizes=(8, 8), channels=1,
                    pred_objectives='x_start', timesteps=4, dynamic_thresholding=False, p2_loss_weight_gamma=0.0, cond_drop_prob=0.0,
                    loss_type=loss_type).to(DEV)
    times = T(gu['times'])
    imagen.noise_schedulers[1].sample_random_times = lambda b, device: times.clone()
    u = imagen.unets[1].train()
    loss, pred, _, _ = imagen((T(gu['hr']) * float(g['hr_scale'])).to(DEV), lowres_img=T(gu['lowres']).to(DEV), unet_number=2,
                              noise=T(gu['noise']).to(DEV))
    ref = float(g[f'loss_{loss_type}'])
    assert abs(loss.item() - ref) <= 2e-5 * abs(ref), (loss.item(), ref)
    close(pred, T(g[f'pred_{loss_type}']), 2e-4, "pred")
    loss.backward()
    named = dict(u.named_parameters())
    n = 0
    for k in g:
        if k.startswith(f'grad_{loss_type}:'):
            close(named[k.split(':', 1)[1]].grad, T(g[k]), 1e-3, k)
            n += 1
    assert n == 3
    with pytest.raises(NotImplementedError):
        Imagen(unets=(NullUnet(), unet), configs=configs, min_bound=0.0, image_sizes=(8, 8), channels=1, loss_type='l3')
