"""Mixed-precision (autocast) forward path on a real MI355X: the fp16 / bf16 MFMA conv kernel (csrc/conv_half.hip).

* integer-valued operands are exact in fp16/bf16 and their products sum exactly in fp32, so the kernel must be BIT-EXACT against a
  float64 convolution rounded once to the operand type (autocast's output rounding) -- this pins the fragment layouts, tap offsets,
  chunking, padding and the ragged edges without any tolerance;
* random data: within one unit in the last place of the operand type of that same reference;
* whole U-Nets under ``torch.autocast``: the deviation from the reference's own autocast output (fixtures made by the real reference
  under CPU autocast, oracle/make_golden_autocast.py) is of the size of the reference's own fp16/bf16 round-off.
"""
import json

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import iqt_oracle as O
from tests.conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda"
T = lambda a: torch.from_numpy(np.asarray(a))
LP = {'fp16': torch.float16, 'bf16': torch.bfloat16}


def ref_conv(x, w, bias, pad, epad, dt, residual=None):
    """float64 conv of the operands rounded to `dt`, + bias, rounded once to `dt`, + residual in fp32 (channels-last in/out)."""
    xr = x.to(dt).double().permute(0, 4, 1, 2, 3)
    wr = w.to(dt).double()
    xp = F.pad(xr, (pad[2], pad[2] + epad[2], pad[1], pad[1] + epad[1], pad[0], pad[0] + epad[0]))
    y = F.conv3d(xp, wr)
    if bias is not None:
        y = y + bias.double().view(1, -1, 1, 1, 1)
    y = y.float().to(dt).float().permute(0, 2, 3, 4, 1).contiguous()
    return y if residual is None else y + residual


SHAPES = [  # B, D, H, W, Cin, Cout, k, pad, epad
    (2, 8, 8, 8, 32, 64, (3, 3, 3), (1, 1, 1), (0, 0, 0)),
    (1, 9, 10, 11, 36, 72, (3, 3, 3), (1, 1, 1), (0, 0, 0)),          # ragged tiles, ragged chunk, two Cout tiles
    (2, 6, 12, 12, 64, 32, (1, 3, 3), (0, 1, 1), (0, 0, 0)),          # per-frame spatial conv of the pseudo-3D block
    (2, 10, 6, 6, 32, 32, (3, 1, 1), (1, 0, 0), (0, 0, 0)),           # temporal conv
    (1, 10, 4, 4, 40, 24, (3, 1, 1), (2, 0, 0), (-2, 0, 0)),          # causal: both pads on the low side
    (1, 4, 8, 8, 96, 128, (1, 1, 1), (0, 0, 0), (0, 0, 0)),           # 1x1x1 / Linear rows
    (1, 5, 5, 5, 8, 8, (3, 3, 3), (0, 0, 0), (0, 0, 0)),              # unpadded (boundary_pad mode), tiny
    (1, 6, 6, 6, 128, 64, (3, 3, 3), (1, 1, 1), (0, 0, 0)),           # four K-chunks x three weight groups
    (1, 2, 9, 9, 16, 16, (1, 5, 5), (0, 2, 2), (0, 0, 0)),            # 25 taps: groups of 9, 9, 7
]


@pytest.mark.parametrize("mode", ["fp16", "bf16"])
@pytest.mark.parametrize("shape", SHAPES)
def test_conv_half_is_bit_exact_on_integer_data(shape, mode):
    from diffusioniqt_amd import ops, _lib
    B, D, H, W, Cin, Cout, k, pad, epad = shape
    assert _lib.query("diqt_conv3d_fwd_h_supported", B, D, H, W, Cin, Cout, *k, *pad, *epad) == 1
    g = torch.Generator().manual_seed(B * 1000 + Cin)
    x = torch.randint(-3, 4, (B, D, H, W, Cin), generator=g).float()
    w = torch.randint(-2, 3, (Cout, Cin, *k), generator=g).float()
    bias = torch.randint(-4, 5, (Cout,), generator=g).float()
    with ops.low_precision(mode), torch.no_grad():
        y = ops.conv3d(x.to(DEV), w.to(DEV), bias.to(DEV), pad, extra_pad=epad)
    ref = ref_conv(x, w, bias, pad, epad, LP[mode])
    assert y.shape == ref.shape
    assert torch.equal(y.cpu(), ref), (y.cpu() - ref).abs().max()
    # the persistent walk of the same kernel (production: launches of >= 512 units): 1 and 3 workgroups over all units, with a residual
    res = torch.randint(-5, 6, tuple(ref.shape), generator=g).float()
    for wgs in (1, 3):
        prev = _lib.query("diqt_set_convh_workgroups", wgs)
        try:
            with ops.low_precision(mode), torch.no_grad():
                yp = ops.conv3d(x.to(DEV), w.to(DEV), bias.to(DEV), pad, residual=res.to(DEV), extra_pad=epad)
        finally:
            _lib.query("diqt_set_convh_workgroups", prev)
        assert torch.equal(yp.cpu(), ref + res), (wgs, (yp.cpu() - ref - res).abs().max())


@pytest.mark.parametrize("mode", ["fp16", "bf16"])
def test_conv_half_random_data_residual_and_linear(mode):
    from diffusioniqt_amd import ops
    dt = LP[mode]
    ulp = 2.0 ** -10 if mode == 'fp16' else 2.0 ** -7
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, 8, 12, 12, 64, generator=g)
    w = torch.randn(64, 64, 3, 3, 3, generator=g) * 0.03
    bias = torch.randn(64, generator=g)
    res = torch.randn(2, 8, 12, 12, 64, generator=g)
    with ops.low_precision(mode), torch.no_grad():
        y = ops.conv3d(x.to(DEV), w.to(DEV), bias.to(DEV), (1, 1, 1), residual=res.to(DEV))
    ref = ref_conv(x, w, bias, (1, 1, 1), (0, 0, 0), dt, residual=res)
    err = (y.cpu() - ref).abs()
    assert (err <= ulp * (ref - res).abs() + 1e-6).all(), err.max()
    assert (err > 0).float().mean().item() < 0.02          # fp32 accumulation order only rarely flips the final rounding
    # Linear with > 64 rows goes through the same kernel
    xl = torch.randn(3, 100, 96, generator=g)
    wl = torch.randn(160, 96, generator=g) * 0.1
    bl = torch.randn(160, generator=g)
    with ops.low_precision(mode), torch.no_grad():
        yl = ops.linear(xl.to(DEV), wl.to(DEV), bl.to(DEV))
    refl = (xl.to(dt).double() @ wl.to(dt).double().t() + bl.double()).float().to(dt).float()
    assert ((yl.cpu() - refl).abs() <= ulp * refl.abs() + 1e-6).all()


def test_low_precision_follows_torch_autocast_and_off_switch():
    from diffusioniqt_amd import ops
    assert ops.lp_mode() is None
    with torch.autocast('cuda', dtype=torch.float16):
        assert ops.lp_mode() == 0
        with ops.low_precision('off'):
            assert ops.lp_mode() is None
    with torch.autocast('cuda', dtype=torch.bfloat16):
        assert ops.lp_mode() == 1
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 8, 8, 8, 32, generator=g).to(DEV)
    w = (torch.randn(32, 32, 3, 3, 3, generator=g) * 0.05).to(DEV)
    with torch.no_grad():
        y32 = ops.conv3d(x, w, None, (1, 1, 1))
        with torch.autocast('cuda', dtype=torch.float16):
            y16 = ops.conv3d(x, w, None, (1, 1, 1))
            with ops.low_precision('off'):
                yoff = ops.conv3d(x, w, None, (1, 1, 1))
    assert torch.equal(yoff, y32) and not torch.equal(y16, y32)
    assert y16.dtype == torch.float32 and (y16 - y32).abs().max().item() <= 4e-3 * y32.abs().max().item()


def test_backward_under_autocast_uses_fp32_gradients():
    """Forward on the fp16 kernel, backward-data / backward-weight on the exact fp32 kernels (more precise than the reference)."""
    from diffusioniqt_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 8, 8, 8, 32, generator=g).to(DEV).requires_grad_()
    w = (torch.randn(32, 32, 3, 3, 3, generator=g) * 0.05).to(DEV).requires_grad_()
    dy = torch.randn(1, 8, 8, 8, 32, generator=g).to(DEV)
    ops.conv3d(x, w, None, (1, 1, 1)).backward(dy)
    gx, gw = x.grad.clone(), w.grad.clone()
    x.grad = w.grad = None
    with torch.autocast('cuda', dtype=torch.float16):
        ops.conv3d(x, w, None, (1, 1, 1)).backward(dy)
    assert torch.equal(x.grad, gx) and torch.equal(w.grad, gw)


@pytest.mark.parametrize("shape", [(2, 8, 8, 8, 32, 64, (3, 3, 3), (1, 1, 1), (0, 0, 0)), (1, 9, 6, 7, 40, 24, (3, 1, 1), (2, 0, 0), (-2, 0, 0)),
                                   (1, 4, 8, 8, 96, 128, (1, 1, 1), (0, 0, 0), (0, 0, 0)), (2, 5, 12, 12, 64, 32, (1, 3, 3), (0, 1, 1), (0, 0, 0))])
def test_bf16_training_backward_data_on_the_bf16_kernel(shape):
    """precision='bf16': the forward's compute type is remembered and backward-data (dX = conv(dY, flipped W), mode-1 packing) runs on
    the bf16 kernel as well -- bit-exact on integer-valued data against float64 autograd rounded once to bf16; the weight gradient (on
    the bf16 MFMA too for 3x3x3 filters with Cin % 32 == 0, fp32 kernels otherwise) and the bias gradient are exact on such data."""
    from diffusioniqt_amd import ops
    B, D, H, W, Cin, Cout, k, pad, epad = shape
    g = torch.Generator().manual_seed(Cin + Cout)
    x = torch.randint(-3, 4, (B, D, H, W, Cin), generator=g).float()
    w = torch.randint(-2, 3, (Cout, Cin, *k), generator=g).float()
    xr = x.double().permute(0, 4, 1, 2, 3).requires_grad_()
    wr = w.double().requires_grad_()
    xp = F.pad(xr, (pad[2], pad[2] + epad[2], pad[1], pad[1] + epad[1], pad[0], pad[0] + epad[0]))
    yr = F.conv3d(xp, wr)
    dy = torch.randint(-2, 3, tuple(yr.shape), generator=g).double()
    yr.backward(dy)
    xd, wd = x.to(DEV).requires_grad_(), w.to(DEV).requires_grad_()
    with torch.autocast('cuda', dtype=torch.bfloat16):
        y = ops.conv3d(xd, wd, None, pad, extra_pad=epad)
    y.backward(dy.float().permute(0, 2, 3, 4, 1).contiguous().to(DEV))            # outside the autocast region, like loss.backward()
    dx_ref = xr.grad.float().to(torch.bfloat16).float().permute(0, 2, 3, 4, 1)
    assert torch.equal(xd.grad.cpu(), dx_ref), (xd.grad.cpu() - dx_ref).abs().max()
    assert torch.equal(wd.grad.cpu(), wr.grad.float())                             # exact: integer data (fp32 or bf16 operands)


def rel(a, b):
    return ((a - b).norm() / b.norm()).item()


@pytest.mark.parametrize("mode", ["fp16", "bf16"])
def test_unet3d_under_autocast_vs_reference_autocast(mode):
    from diffusioniqt_amd.imagen_video import Unet3D
    g = load_golden('unet3d_tiny')
    a = load_golden('autocast_fwd')
    kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in json.loads(str(g['kwargs'])).items()}
    unet = Unet3D(**kw)
    unet.load_state_dict(O.hash_fill_state_dict(unet.state_dict(), 11))
    unet = unet.to(DEV).eval()
    args = (T(g['x']).to(DEV), T(g['time']).to(DEV))
    kws = dict(lowres_cond_img=T(g['lowres']).to(DEV), lowres_noise_times=T(g['lowres_times']).to(DEV))
    with torch.no_grad():
        y32 = unet(*args, **kws).cpu()
        with torch.autocast('cuda', dtype=LP[mode]):
            ylp = unet(*args, **kws).cpu()
    ref32, reflp = T(a['B_y32']), T(a['B_y_' + mode])
    assert rel(y32, ref32) < 1e-4
    own, refs = rel(ylp, y32), rel(reflp, ref32)
    assert 0 < own <= 1.5 * refs, (own, refs)                       # the low-precision kernel ran, and is no noisier than the reference
    assert rel(ylp, reflp) <= 1.5 * refs, (rel(ylp, reflp), refs)   # and lands within the reference's own round-off of it


@pytest.mark.parametrize("mode", ["fp16", "bf16"])
def test_family_a_unet_under_autocast_vs_reference_autocast(mode):
    from diffusioniqt_amd.imagen_pytorch3D import SRUnet256
    g = load_golden('unetA_tiny')
    a = load_golden('autocast_fwd')
    unet = SRUnet256(**json.loads(str(g['kwargs'])))
    unet.load_state_dict(O.hash_fill_state_dict(unet.state_dict(), 0))
    unet = unet.to(DEV).eval()
    args = (T(g['x']).to(DEV), T(g['times']).to(DEV), T(g['log_snr']).to(DEV))
    with torch.no_grad():
        y32 = unet(*args, lowres_cond_img=T(g['lowres']).to(DEV)).cpu()
        with torch.autocast('cuda', dtype=LP[mode]):
            ylp = unet(*args, lowres_cond_img=T(g['lowres']).to(DEV)).cpu()
    ref32, reflp = T(a['A_y32']), T(a['A_y_' + mode])
    assert rel(y32, ref32) < 1e-4
    own, refs = rel(ylp, y32), rel(reflp, ref32)
    assert 0 < own <= 1.5 * refs, (own, refs)
    assert rel(ylp, reflp) <= 1.5 * refs, (rel(ylp, reflp), refs)


def test_trainer_mixed_precision_switch():
    """ImagenTrainer(fp16=True) / precision='bf16' (trainer.py:293-311): forward under autocast on the fp16 / bf16 kernel, fp32
    gradients, optimiser and master weights; the first micro-step's loss agrees with the fp32 trainer's to low-precision
    round-off and an optimiser step is taken after `gradient_accumulation_steps` micro-steps."""
    from diffusioniqt_amd import ops
    from diffusioniqt_amd.imagen_pytorch3D import SRUnet256, Imagen, NullUnet
    from diffusioniqt_amd.trainer import ImagenTrainer
    g = load_golden('unetA_tiny')
    kw = json.loads(str(g['kwargs']))
    configs = {'Data': {'norm': 'z-score'}, 'Train': {'batch_sample': False, 'patch_size_sub': 8, 'pred_obj': 'x_start'},
               'Eval': {'repeat': 1}}
    hr, lr, noise, times = T(g['hr']), T(g['lowres']), T(g['noise']), T(g['times'])
    losses = {}
    for mode in ('no', 'fp16', 'bf16'):
        unet = SRUnet256(**kw)
        unet.load_state_dict(O.hash_fill_state_dict(unet.state_dict(), 0))
        imagen = Imagen(unets=(NullUnet(), unet), configs=configs, min_bound=float(g['min_bound']), image_sizes=(8, 8), channels=1,
                        pred_objectives='x_start', timesteps=4, dynamic_thresholding=False, p2_loss_weight_gamma=0.0,
                        cond_drop_prob=0.0).to(DEV)
        imagen.noise_schedulers[1].sample_random_times = lambda b, device: times.clone()
        ImagenTrainer.locked = False
        trainer = ImagenTrainer(configs=configs, imagen=imagen, gradient_accumulation_steps=2, verbose=False,
                                **({'fp16': True} if mode == 'fp16' else {'precision': mode} if mode == 'bf16' else {}))
        assert trainer.mixed_precision == mode and trainer.cast_half_at_training == (mode == 'fp16')
        w0 = trainer.imagen.unets[1].final_conv.weight.detach().clone()
        seen = []
        real = ops._conv_fwd_half
        ops._conv_fwd_half = lambda *a, **k: (seen.append(a[-1]), real(*a, **k))[1]
        try:
            out = [trainer(hr, lowres_img=lr, unet_number=2, max_batch_size=2, noise=noise) for _ in range(2)]
        finally:
            ops._conv_fwd_half = real
        trainer.update(unet_number=2)
        losses[mode] = float(out[0][0]) if isinstance(out[0], tuple) else float(out[0])
        assert (len(seen) > 0) == (mode != 'no') and all(b == (1 if mode == 'bf16' else 0) for b in seen)
        w1 = trainer.imagen.unets[1].final_conv.weight.detach()
        assert torch.isfinite(w1).all() and not torch.equal(w1, w0)
    assert abs(losses['no'] - float(g['loss'])) <= 1e-4 * abs(float(g['loss']))
    assert abs(losses['fp16'] - losses['no']) <= 5e-3 * abs(losses['no']), losses
    assert abs(losses['bf16'] - losses['no']) <= 3e-2 * abs(losses['no']), losses


@pytest.mark.parametrize("mode", ["fp16", "bf16"])
@pytest.mark.parametrize("G,n,h,d,E,use_rel,causal", [(2, 40, 4, 64, 1, False, False), (3, 16, 8, 64, 1, True, True),
                                                       (1, 300, 2, 32, 3, False, False), (2, 33, 3, 32, 1, True, False),
                                                       (1, 1, 8, 64, 1, True, True), (1, 2048, 8, 64, 5, False, False),
                                                       (2, 70, 2, 64, 1, True, True)])
def test_fused_attention_low_precision(G, n, h, d, E, use_rel, causal, mode):
    """diqt_mqa_attention_fwd_h under ops.low_precision: float64 attention of the operands as the kernel rounds them (scaled q, k, v
    and the probabilities to 16 bit; scores, soft-max statistics and accumulation in fp32), result rounded once to the operand type."""
    from diffusioniqt_amd import ops
    dt = LP[mode]
    ulp = 2.0 ** -10 if mode == 'fp16' else 2.0 ** -7
    gen = torch.Generator().manual_seed(G * 100 + n)
    q = torch.randn(G, n, h * d, generator=gen)
    kv = torch.randn(G, E + n, 2 * d, generator=gen)
    rel = torch.randn(2 * n - 1, h, generator=gen) if use_rel else None
    nb = torch.randn(h, generator=gen) if use_rel else None
    scale = d ** -0.5
    qd = (q * scale).to(dt).double().reshape(G, n, h, d)
    k, v = kv.to(dt).double()[..., :d], kv.to(dt).double()[..., d:]
    sim = torch.einsum('gihd,gjd->gihj', qd, k)
    if use_rel:
        i = torch.arange(n)[:, None]; j = torch.arange(n)[None, :]
        sim[..., E:] += rel.double()[(i - j + n - 1)].permute(0, 2, 1)[None]
        sim[..., E - 1] += nb.double()[None, None, :]
    if causal:
        i = torch.arange(n)[:, None]; j = torch.arange(n)[None, :]
        sim[..., E:] = sim[..., E:].masked_fill((j > i)[None, :, None, :].expand(G, n, h, n), float('-inf'))
    p = sim.softmax(dim=-1)
    ref = torch.einsum('gihj,gjd->gihd', p, v).reshape(G, n, h * d)
    with ops.low_precision(mode), torch.no_grad():
        got = ops.mqa_attention_nograd(q.to(DEV), kv.to(DEV), rel.to(DEV) if use_rel else None, nb.to(DEV) if use_rel else None,
                                       n, h, d, E, n, causal, scale).cpu().double()
    # un-normalised probabilities are rounded to 16 bit before p v (relative error <= ulp/2 each) and the result once more
    err = (got - ref).abs().max().item()
    assert err <= 2.5 * ulp * ref.abs().max().item(), (err, ref.abs().max().item())
    with torch.no_grad():
        full = ops.mqa_attention_nograd(q.to(DEV), kv.to(DEV), rel.to(DEV) if use_rel else None, nb.to(DEV) if use_rel else None,
                                        n, h, d, E, n, causal, scale).cpu().double()
    assert not torch.equal(full, got)          # the low-precision kernel really ran


def _temporal_block_ref(x, attn, dt, causal):
    """float64 model of diqt_temporal_attention_h with the kernel's rounding points: LayerNorm output, the projections' weights,
    q / k / v, the un-normalised probabilities, the head outputs and the to_out product are rounded to the operand type."""
    r = lambda t: t.to(dt).double()
    B, Fr, P, C = x.shape
    h, d = attn.heads, attn.dim_head
    xd = x.double().permute(0, 2, 1, 3).reshape(B * P, Fr, C)                     # sequences (b, p) of F frames
    ln = lambda t, g: (t - t.mean(-1, keepdim=True)) * (t.var(-1, unbiased=False, keepdim=True) + 1e-5).rsqrt() * g.double()
    xn = r(ln(xd, attn.norm.g.detach().cpu()).float())
    wq = r((attn.to_q.weight.detach().cpu() * attn.scale))
    wkv, wo = r(attn.to_kv.weight.detach().cpu()), r(attn.to_out[0].weight.detach().cpu())
    q = r((xn @ wq.T).float()).reshape(-1, Fr, h, d)
    kv = r((xn @ wkv.T).float())
    k, v = kv[..., :d], kv[..., d:]
    nk, nv = r(attn.null_kv.detach().cpu())
    sim = torch.einsum('gihd,gjd->gihj', q, k)
    i = torch.arange(Fr)[:, None]; j = torch.arange(Fr)[None, :]
    with torch.no_grad():
        rel = attn.rel_pos_bias(Fr, torch.device(DEV, 0)).detach().cpu().double()
    sim = sim + rel[(i - j + Fr - 1)].permute(0, 2, 1)[None]
    if causal:
        sim = sim.masked_fill((j > i)[None, :, None, :], float('-inf'))
    snull = torch.einsum('gihd,d->gih', q, nk) + attn.null_attn_bias.detach().cpu().double()
    m = torch.maximum(sim.amax(-1), snull)
    p, pn = (sim - m[..., None]).exp(), (snull - m).exp()
    l = p.sum(-1) + pn
    o = (torch.einsum('gihj,gjd->gihd', r(p.float()), v) + r(pn.float())[..., None] * nv) / l[..., None]
    o = r(o.float()).reshape(-1, Fr, h * d)
    yo = r((o @ wo.T).float())
    out = ln(yo, attn.to_out[1].g.detach().cpu()) + xd
    return out.reshape(B, P, Fr, C).permute(0, 2, 1, 3), yo


@pytest.mark.parametrize("mode", ["fp16", "bf16"])
@pytest.mark.parametrize("B,Fr,P,C,h,causal", [(2, 64, 20, 64, 8, False), (1, 32, 37, 64, 8, True), (2, 64, 9, 128, 8, False),
                                               (1, 32, 300, 128, 4, False), (1, 64, 530, 64, 4, True), (1, 64, 11, 256, 8, False),
                                               (2, 32, 5, 256, 8, True)])
def test_temporal_attention_block_one_kernel(B, Fr, P, C, h, causal, mode):
    """diqt_temporal_attention_h = Residual(Attention) over the frame axis, in place on [B, F, H, W, C]: against the float64 model
    of its own rounding points, against the unfused autocast chain, and selected by TokensOverTime under autocast only."""
    from diffusioniqt_amd import ops
    from diffusioniqt_amd.imagen_video import Attention, Residual, TokensOverTime
    dt = LP[mode]
    ulp = 2.0 ** -10 if mode == 'fp16' else 2.0 ** -7
    torch.manual_seed(B * 1000 + P)
    attn = Attention(C, heads=h, dim_head=64, causal=causal, rel_pos_bias=True, init_zero=False)
    with torch.no_grad():
        attn.norm.g.uniform_(0.5, 1.5)
        attn.to_out[1].g.uniform_(0.5, 1.5)
    blk = TokensOverTime(Residual(attn)).to(DEV).eval()
    Hh = 1
    x = torch.randn(B, Fr, Hh, P, C)
    xg = x.to(DEV)
    with torch.no_grad():
        y32 = blk(xg).cpu()
        with torch.autocast('cuda', dtype=dt):
            assert attn.block_ok(B, Fr, P, C)
            got = blk(xg).cpu()
            attn.block_ok = lambda *a: False                     # the unfused autocast chain
            chain = blk(xg).cpu()
            del attn.block_ok
        assert not attn.block_ok(B, Fr, P, C)                    # fp32: never
    ref, yo = _temporal_block_ref(x.reshape(B, Fr, P, C), attn, dt, causal)
    ref = ref.reshape(B, Fr, Hh, P, C)
    # the to_out product carries ~1 ulp of relative error (rounded operands of fp32 sums); the LayerNorm behind it scales rows to unit variance
    err = (got.double() - ref).abs().max().item()
    assert err <= 6 * ulp * (ref - x.double()).abs().max().item(), (err, (ref - x.double()).abs().max().item())
    assert rel(got.double() - x.double(), ref - x.double()) <= 1.5 * ulp
    assert not torch.equal(got, y32) and not torch.equal(got, chain)
    assert rel(got - x, y32 - x) <= 6 * ulp and rel(got - x, chain - x) <= 6 * ulp


@pytest.mark.parametrize("mode", ["fp16", "bf16"])
@pytest.mark.parametrize("B,Fr,S,C1,C2,res", [(2, 10, 20, 32, 64, True), (1, 7, 33, 72, 40, False), (2, 12, 16, 64, 64, True)])
def test_conv_pair_with_a_16_bit_intermediate_is_bit_identical(B, Fr, S, C1, C2, res, mode):
    """The per-frame + temporal conv pair of a pseudo-3D block on the sampling path under autocast: the tensor between the two
    convs is stored in the operand type (diqt_conv3d_fwd_h_io) -- the same values the fp32 tensor holds, so the pair's output must
    be BIT-IDENTICAL to two separate launches with an fp32 tensor in between."""
    from diffusioniqt_amd import ops, _lib
    g = torch.Generator().manual_seed(B * 100 + C1)
    x = torch.randn(B, Fr, S, S, C1, generator=g).to(DEV)
    w1 = (torch.randn(C2, C1, 1, 3, 3, generator=g) * 0.05).to(DEV); b1 = torch.randn(C2, generator=g).to(DEV)
    w2 = (torch.randn(C2, C2, 3, 1, 1, generator=g) * 0.1).to(DEV); b2 = torch.randn(C2, generator=g).to(DEV)
    r = torch.randn(B, Fr, S, S, C2, generator=g).to(DEV) if res else None
    prev = _lib.query("diqt_set_convh_workgroups", 3)          # the persistent walk (production: >= 512 units) on a small shape
    try:
        with ops.low_precision(mode), torch.no_grad():
            y = ops.conv_pair_nograd_h(x, w1, b1, (0, 1, 1), w2, b2, (2, 0, 0), (-2, 0, 0), r, want_stats=True)
            assert y is not None and y.dtype == torch.float32
            st = y._diqt_stats              # per-(tile, wave) column sums of the stored values, for the next GroupNorm
            assert st.rows == Fr * S * S and st.partials.shape[0] == B and st.partials.shape[2:] == (2, C2)
            yd = y.double().reshape(B, -1, C2)
            assert torch.allclose(st.partials.double().sum(1)[:, 0], yd.sum(1), rtol=1e-5, atol=1e-3)
            assert torch.allclose(st.partials.double().sum(1)[:, 1], (yd * yd).sum(1), rtol=1e-5, atol=1e-3)
            mid = ops.conv3d(x, w1, b1, (0, 1, 1))
            ref = ops.conv3d(mid, w2, b2, (2, 0, 0), residual=r, extra_pad=(-2, 0, 0))
    finally:
        _lib.query("diqt_set_convh_workgroups", prev)
    assert torch.equal(y, ref), (y - ref).abs().max()
    with torch.no_grad():
        assert ops.conv_pair_nograd_h(x, w1, b1, (0, 1, 1), w2, b2, (2, 0, 0), (-2, 0, 0), r) is None      # fp32: never


@pytest.mark.parametrize("mode", ["fp16", "bf16"])
@pytest.mark.parametrize("with_ss", [False, True])
def test_block_with_16_bit_groupnorm_output_is_bit_identical(mode, with_ss):
    """GroupNorm-apply + SiLU stored in the operand type (diqt_gn_act_fwd_h) in front of the 16-bit conv pair: the conv would round the
    fp32 values to that type while staging them, so the block's output is bit-identical to groupnorm_act (fp32) + two convs."""
    from diffusioniqt_amd import ops, _lib
    B, Fr, S, C1, C2 = 2, 9, 24, 64, 32
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(B, Fr, S, S, C1, generator=g) * 2 + 0.5).to(DEV)
    gamma, beta = torch.randn(C1, generator=g).to(DEV), torch.randn(C1, generator=g).to(DEV)
    ss = (torch.randn(B, 2 * C1, generator=g) * 0.3).to(DEV) if with_ss else None
    w1 = (torch.randn(C2, C1, 1, 3, 3, generator=g) * 0.05).to(DEV); b1 = torch.randn(C2, generator=g).to(DEV)
    w2 = (torch.randn(C2, C2, 3, 1, 1, generator=g) * 0.1).to(DEV); b2 = torch.randn(C2, generator=g).to(DEV)
    prev = _lib.query("diqt_set_convh_workgroups", 2)
    try:
        with ops.low_precision(mode), torch.no_grad():
            y = ops.conv_pair_nograd_h(x, w1, b1, (0, 1, 1), w2, b2, (2, 0, 0), (-2, 0, 0), None, gn=(gamma, beta, ss, 8, ops.ACT_SILU, 1e-5))
            assert y is not None
            xn = ops.groupnorm_act(x, gamma, beta, ss, 8, ops.ACT_SILU, 1e-5)
            ref = ops.conv3d(ops.conv3d(xn, w1, b1, (0, 1, 1)), w2, b2, (2, 0, 0), extra_pad=(-2, 0, 0))
    finally:
        _lib.query("diqt_set_convh_workgroups", prev)
    assert torch.equal(y, ref), (y - ref).abs().max()


@pytest.mark.parametrize("mode", ["fp16", "bf16"])
def test_resnet_block_with_16_bit_tensors_between_its_blocks_is_bit_identical(mode):
    """Sampling under autocast: block1's output (temporal conv, no residual) only feeds block2's GroupNorm, so it is stored in the operand
    type together with its GroupNorm statistics (diqt_conv3d_fwd_h_io y_half + stats, diqt_gn_act_fwd_h x_half).  The values are the ones
    the fp32 tensor holds: the ResnetBlock's output must be bit-identical to the run with an fp32 tensor in between."""
    from diffusioniqt_amd import ops, _lib
    from diffusioniqt_amd.imagen_video import ResnetBlock
    torch.manual_seed(4)
    blk = ResnetBlock(32, 64, time_cond_dim=48, groups=8, use_gca=True).to(DEV).eval()
    x = torch.randn(2, 8, 16, 16, 32, device=DEV)
    t = torch.randn(2, 48, device=DEV)
    seen = []
    real_call, real_pair = _lib.call, ops.conv_pair_nograd_h
    spy = lambda name, *a: (seen.append((name, a)), real_call(name, *a))[1]
    prev = _lib.query("diqt_set_convh_workgroups", 2)
    try:
        _lib.call = spy
        with torch.no_grad(), torch.autocast('cuda', dtype=LP[mode]):
            y = blk(x, t)
            halves = [a for n, a in seen if n == "diqt_gn_act_fwd_h"]
            assert len(halves) == 2 and [a[-2] for a in halves] == [0, 1]          # block1 reads fp32, block2 reads block1's 16-bit output
            ops.conv_pair_nograd_h = lambda *a, **k: real_pair(*a, **{**k, 'out_half': False})
            ref = blk(x, t)
    finally:
        _lib.call, ops.conv_pair_nograd_h = real_call, real_pair
        _lib.query("diqt_set_convh_workgroups", prev)
    assert torch.equal(y, ref), (y - ref).abs().max()


@pytest.mark.parametrize("bf16", [1, 0])
@pytest.mark.parametrize("B,D,H,W,Cin,Cout,k,pad,epad", [
    (2, 8, 8, 8, 32, 64, (3, 3, 3), (1, 1, 1), (0, 0, 0)), (1, 9, 10, 19, 64, 72, (3, 3, 3), (1, 1, 1), (0, 0, 0)),
    (3, 6, 8, 32, 96, 32, (3, 3, 3), (1, 1, 1), (0, 0, 0)), (1, 10, 9, 18, 32, 40, (3, 3, 3), (0, 0, 0), (0, 0, 0)),
    (8, 32, 32, 32, 64, 64, (3, 3, 3), (1, 1, 1), (0, 0, 0)),
    # Family B's per-frame and temporal filters (imagen_video.py Conv3d: (1, k, k) then (k, 1, 1), the latter causal = one-sided padding)
    (2, 5, 9, 18, 64, 64, (1, 3, 3), (0, 1, 1), (0, 0, 0)), (1, 3, 32, 32, 32, 96, (1, 3, 3), (0, 1, 1), (0, 0, 0)),
    (2, 7, 6, 17, 64, 40, (3, 1, 1), (1, 0, 0), (0, 0, 0)), (1, 8, 16, 16, 96, 64, (3, 1, 1), (2, 0, 0), (-2, 0, 0)),
    (1, 32, 64, 64, 64, 64, (1, 3, 3), (0, 1, 1), (0, 0, 0)), (1, 32, 64, 64, 64, 64, (3, 1, 1), (2, 0, 0), (-2, 0, 0))])
def test_weight_gradient_on_the_16_bit_mfma(B, D, H, W, Cin, Cout, k, pad, epad, bf16):
    """diqt_conv3d_bwd_weight_h (conv_wgrad_h_kernel: row-major LDS images, transposing fragment reads): integer-valued operands are
    exact in bf16 / fp16 and their products sum exactly in fp32, so dW and dbias must be BIT-EXACT against float64 autograd -- this pins
    the fragment layouts, tap offsets, padding (low = pad, high = pad + epad), ragged tiles, the split-K slabs and the bias partials;
    random data: within the rounding of the operands."""
    from diffusioniqt_amd import ops, _lib
    geo = (B, D, H, W, Cin, Cout, *k, *pad, *epad)
    nbytes = _lib.query("diqt_conv3d_bwd_weight_h_workspace_bytes", *geo)
    assert nbytes > 0
    g = torch.Generator().manual_seed(B * 7 + Cin)
    big = B * D * H * W > 100000
    dy_shape = tuple([B] + [n + 2 * p + e - kk + 1 for n, p, e, kk in zip((D, H, W), pad, epad, k)] + [Cout])
    fpad = (pad[2], pad[2] + epad[2], pad[1], pad[1] + epad[1], pad[0], pad[0] + epad[0])
    for kind in (("int",) if big else ("int", "rand")):
        if kind == "int":
            x = torch.randint(-3, 4, (B, D, H, W, Cin), generator=g).float()
            dy = torch.randint(-2, 3, dy_shape, generator=g).float()
        else:
            x = torch.randn(B, D, H, W, Cin, generator=g)
            dy = torch.randn(dy_shape, generator=g)
        dt = torch.bfloat16 if bf16 else torch.float16
        xr = x.to(dt).double().permute(0, 4, 1, 2, 3)
        wr = torch.zeros(Cout, Cin, *k, dtype=torch.float64, requires_grad=True)
        if big:      # float64 conv of 2 M voxels on the CPU is slow: check through linearity instead -- a random probe of dW
            probe = torch.randint(-1, 2, (Cout, Cin, *k), generator=g).double()
            yp = F.conv3d(F.pad(xr, fpad), probe)
            want_probe = (yp * dy.to(dt).double().permute(0, 4, 1, 2, 3)).sum().item()
        else:
            yr = F.conv3d(F.pad(xr, fpad), wr)
            yr.backward(dy.to(dt).double().permute(0, 4, 1, 2, 3))
        dw = torch.empty(Cout, Cin, *k, device=DEV)
        db = torch.empty(Cout, device=DEV)
        ws = torch.empty(nbytes // 4, device=DEV)
        _lib.call("diqt_conv3d_bwd_weight_h", x.to(DEV), dy.to(DEV), dw, db, ws, nbytes, *geo, bf16, torch.cuda.current_stream().cuda_stream)
        db_ref = dy.double().sum(dim=(0, 1, 2, 3))
        if kind == "int":
            assert torch.equal(db.cpu().double(), db_ref)
            if big:
                assert (dw.cpu().double() * probe).sum().item() == want_probe
            else:
                assert torch.equal(dw.cpu().double(), wr.grad), (dw.cpu().double() - wr.grad).abs().max()
        else:
            assert (dw.cpu().double() - wr.grad).abs().max().item() <= 2e-5 * wr.grad.abs().max().item()
            assert (db.cpu().double() - db_ref).abs().max().item() <= 1e-5 * db_ref.abs().max().item()


@pytest.mark.parametrize("bf16", [0, 1])
@pytest.mark.parametrize("rows,Cin,Cout,res", [(2048 + 77, 64, 128, False), (4096, 256, 512, True), (2500, 96, 192, False),
                                               (3000, 128, 384, True), (2304, 512, 256, False), (2100, 128, 64, True), (4000, 64, 40, False)])
def test_pointwise_conv_with_many_output_channels_as_a_gemm(rows, Cin, Cout, res, bf16):
    """conv_pw_h_kernel (behind diqt_conv3d_fwd_h: 1x1x1, Cin % 32 == 0, Cout > 64, >= 2048 rows): 256 rows x 128 channels per
    workgroup, K-blocked.  Integer-valued operands are exact in fp16 / bf16 and sum exactly in fp32, so the result -- rounded once to the
    operand type, then the fp32 residual added -- must be BIT-EXACT against float64: pins the fragment layouts, the double-buffered
    staging, ragged row tiles, channel tiles past Cout / CoutPad and the XCD-aware tile mapping."""
    from diffusioniqt_amd import _lib
    g = torch.Generator().manual_seed(rows + Cin)
    dt = torch.bfloat16 if bf16 else torch.float16
    x = torch.randint(-3, 4, (1, 1, 1, rows, Cin), generator=g).float()
    w = torch.randint(-2, 3, (Cout, Cin, 1, 1, 1), generator=g).float()
    b = torch.randint(-4, 5, (Cout,), generator=g).float()
    r = torch.randn(1, 1, 1, rows, Cout, generator=g) if res else None
    st = torch.cuda.current_stream().cuda_stream
    n = _lib.query("diqt_conv_packed_h_elems", Cout, Cin, 1, 1, 1)
    packed = torch.empty(n, dtype=torch.int16, device=DEV)
    _lib.call("diqt_conv_pack_weight_h", w.to(DEV), packed, Cout, Cin, 1, 1, 1, 0, bf16, st)
    y = torch.full((1, 1, 1, rows, Cout), float('nan'), device=DEV)
    _lib.call("diqt_conv3d_fwd_h", x.to(DEV), packed, b.to(DEV), r.to(DEV) if res else None, y, 1, 1, 1, rows, Cin, Cout, 1, 1, 1,
              0, 0, 0, 0, 0, 0, bf16, 1, st)
    want = (x.double().reshape(rows, Cin) @ w.double().reshape(Cout, Cin).T + b.double()).to(dt).float()
    if res:
        want = want + r.reshape(rows, Cout)
    assert torch.equal(y.cpu().reshape(rows, Cout), want), (y.cpu().reshape(rows, Cout) - want).abs().max()


@pytest.mark.parametrize("bf16", [0, 1])
@pytest.mark.parametrize("B,D,H,W,Cin,Cout,pd,epd,res", [(2, 5, 16, 16, 64, 64, 2, -2, False), (1, 8, 16, 20, 96, 160, 1, 0, True),
                                                          (2, 3, 32, 32, 128, 128, 2, -2, True), (1, 7, 20, 19, 64, 32, 2, -2, False)])
def test_temporal_conv_as_a_gemm_over_shifted_rows(B, D, H, W, Cin, Cout, pd, epd, res, bf16):
    """The (3,1,1) temporal conv of a pseudo-3D block on conv_pw_h_kernel (16-bit x rows from the per-frame conv, fp32 y): K = taps x Cin,
    the rows of tap t read (t - pad) H W rows away, zero outside the volume (causal: pad (2, 0); symmetric: (1, 1)).  Integer-valued data:
    BIT-EXACT against float64 conv3d, rounded once to the operand type, fp32 residual added after."""
    from diffusioniqt_amd import _lib
    g = torch.Generator().manual_seed(B * 100 + D)
    dt = torch.bfloat16 if bf16 else torch.float16
    x = torch.randint(-3, 4, (B, D, H, W, Cin), generator=g).float()
    w = torch.randint(-2, 3, (Cout, Cin, 3, 1, 1), generator=g).float()
    b = torch.randint(-4, 5, (Cout,), generator=g).float()
    r = torch.randn(B, D, H, W, Cout, generator=g) if res else None
    st = torch.cuda.current_stream().cuda_stream
    geo = (B, D, H, W, Cin, Cout, 3, 1, 1, pd, 0, 0, epd, 0, 0)
    n = _lib.query("diqt_conv_packed_h_elems", Cout, Cin, 3, 1, 1)
    packed = torch.empty(n, dtype=torch.int16, device=DEV)
    _lib.call("diqt_conv_pack_weight_h", w.to(DEV), packed, Cout, Cin, 3, 1, 1, 0, bf16, st)
    y = torch.full((B, D, H, W, Cout), float('nan'), device=DEV)
    _lib.call("diqt_conv3d_fwd_h_io", x.to(DEV).to(dt), packed, b.to(DEV), r.to(DEV) if res else None, y, *geo, bf16, 1, 1, 0, None, st)
    xr = F.pad(x.double().permute(0, 4, 1, 2, 3), (0, 0, 0, 0, pd, pd + epd))
    want = (F.conv3d(xr, w.double()) + b.double().view(1, -1, 1, 1, 1)).permute(0, 2, 3, 4, 1).to(dt).float()
    if res:
        want = want + r
    assert torch.equal(y.cpu(), want), (y.cpu() - want).abs().max()


@pytest.mark.parametrize("with_ss,res,tap", [(True, False, True), (False, True, False)])
def test_bf16_training_block_as_one_node_with_a_bf16_activation_is_bit_identical(with_ss, res, tap):
    """ImagenTrainer(precision='bf16'): GroupNorm-apply + conv as ONE autograd node (ops._GnActConvHFn) whose intermediate exists only in
    bf16 -- written so by the GroupNorm-apply pass, read by conv_f9h_kernel in the forward and by the bf16 weight-gradient kernel in the
    backward -- against the two-node path (fp32 activation, rounded to bf16 while each conv kernel stages it): the same bits everywhere
    (output, dx, dW, db, GroupNorm parameter gradients, scale/shift gradient, the residual's gradient)."""
    from diffusioniqt_amd import ops, _lib
    B, S, C, Co, G = 2, 16, 64, 64, 8
    g = torch.Generator().manual_seed(17)
    x0 = torch.randn(B, S, S, S, C, generator=g)
    gamma0, beta0 = torch.randn(C, generator=g), torch.randn(C, generator=g)
    w0 = torch.randn(Co, C, 3, 3, 3, generator=g) / (27 * C) ** 0.5
    b0 = torch.randn(Co, generator=g)
    ss0 = torch.randn(B, 2 * C, generator=g) * 0.3 if with_ss else None
    r0 = torch.randn(B, S, S, S, Co, generator=g) if res else None
    dy = torch.randn(B, S, S, S, Co, generator=g).to(DEV)

    def run(fused):
        leaves = [t.clone().to(DEV).requires_grad_(True) if t is not None else None for t in (x0, gamma0, beta0, w0, b0, ss0, r0)]
        x, gamma, beta, w, b, ss, r = leaves
        xin = x * 1.0                                        # a non-leaf input, as inside the U-Net (the tap routes its second consumer)
        with ops.low_precision('bf16'), _lib.census() as c:
            if fused:
                out = ops.gn_conv3d_train_h(xin, gamma, beta, ss, G, ops.ACT_MISH, 1e-5, w, b, (1, 1, 1), r, want_stats=True, tap=tap)
                assert out is not None
            else:
                h = ops.groupnorm_act(xin, gamma, beta, ss, G, ops.ACT_MISH, 1e-5, tap=tap)
                h, alias = h if tap else (h, None)
                y = ops.conv3d(h, w, b, (1, 1, 1), r, want_stats=True)
                out = (y, alias) if tap else y
            y, alias = out if tap else (out, None)
            loss = (y * dy).sum() + ((alias * 0.5).sum() if tap else 0.0)
            loss.backward()
            torch.cuda.synchronize()
            nf9 = c.count("conv3d_fwd_h(v9h)")
        return y.detach(), [t.grad for t in leaves if t is not None], nf9, getattr(y, "_diqt_stats", None)

    ya, ga, na, sa = run(True)
    yb, gb, nb, sb = run(False)
    assert na == 1 and nb == 0                               # the fused node's forward ran on the LDS-DMA kernel
    assert torch.equal(ya, yb)
    for a, b_ in zip(ga, gb):
        assert a is not None and b_ is not None and torch.equal(a, b_), (a - b_).abs().max()
    # the fused node's conv also hands the next GroupNorm its column sums (the two-node path has none under autocast): against y itself
    assert sa is not None
    flat = ya.double().reshape(B, -1, Co)
    got = sa.partials.double().sum(1)
    assert torch.allclose(got[:, 0], flat.sum(1), rtol=1e-6, atol=1e-2) and torch.allclose(got[:, 1], (flat * flat).sum(1), rtol=1e-6, atol=1e-2)


def test_fp16_training_with_the_loss_scaler():
    """ImagenTrainer(fp16=True) = autocast + GradScaler in the reference (trainer.py:293-311, 364; accelerate's backward / optimizer wrappers):
    the loss is scaled by 2^16 before backward, the conv backward kernels run on the fp16 MFMA, the optimiser step un-scales inside the
    fused Adam pass; a non-finite gradient skips the step, zeroes the gradients and halves the scale; the checkpoint carries torch's
    GradScaler state keys.  The first Adam step moves every weight by ~lr whatever the gradient's magnitude, so the fp16 step lands on
    the fp32 trainer's weights to within a fraction of lr."""
    from diffusioniqt_amd import ops, _lib
    from diffusioniqt_amd.imagen_pytorch3D import SRUnet256, Imagen, NullUnet
    from diffusioniqt_amd.trainer import ImagenTrainer
    g = load_golden('unetA_tiny')
    kw = json.loads(str(g['kwargs']))
    configs = {'Data': {'norm': 'z-score'}, 'Train': {'batch_sample': False, 'patch_size_sub': 8, 'pred_obj': 'x_start'},
               'Eval': {'repeat': 1}}
    hr, lr, noise, times = T(g['hr']), T(g['lowres']), T(g['noise']), T(g['times'])

    def make(**tkw):
        unet = SRUnet256(**kw)
        unet.load_state_dict(O.hash_fill_state_dict(unet.state_dict(), 0))
        imagen = Imagen(unets=(NullUnet(), unet), configs=configs, min_bound=float(g['min_bound']), image_sizes=(8, 8), channels=1,
                        pred_objectives='x_start', timesteps=4, dynamic_thresholding=False, p2_loss_weight_gamma=0.0,
                        cond_drop_prob=0.0).to(DEV)
        imagen.noise_schedulers[1].sample_random_times = lambda b, device: times.clone()
        ImagenTrainer.locked = False
        return ImagenTrainer(configs=configs, imagen=imagen, gradient_accumulation_steps=2, verbose=False, use_ema=False, lr=1e-3, **tkw)

    step = lambda tr: tr(hr, lowres_img=lr, unet_number=2, max_batch_size=2, noise=noise)
    ref = make()
    for _ in range(2):
        step(ref)
    w_ref = ref.imagen.unets[1].final_conv.weight.detach().clone()

    tr = make(fp16=True)
    sc = tr.scaler1
    assert sc.enabled and list(sc.state_dict()) == ["scale", "growth_factor", "backoff_factor", "growth_interval", "_growth_tracker"]
    assert sc.get_scale() == 65536.0
    w0 = tr.imagen.unets[1].final_conv.weight.detach().clone()
    calls = []
    real = _lib.call
    _lib.call = lambda name, *a: (calls.append((name, a)), real(name, *a))[1]
    try:
        for _ in range(2):
            step(tr)
    finally:
        _lib.call = real
    wg = [a for n, a in calls if n == "diqt_conv3d_bwd_weight_h"]
    assert wg and all((a[-2] & 1) == 0 for a in wg)                     # the weight gradients ran with fp16 operands
    w1 = tr.imagen.unets[1].final_conv.weight.detach().clone()
    assert sc.state_dict()["_growth_tracker"] == 1 and sc.get_scale() == 65536.0 and tr.optim1.step_count == 1
    assert torch.isfinite(w1).all() and not torch.equal(w1, w0)
    assert (w1 - w_ref).abs().max().item() <= 0.25 * 1e-3, (w1 - w_ref).abs().max()     # a quarter of lr
    assert float(tr.steps[1]) == 2
    # ---- a non-finite gradient: the step is skipped, the gradients are dropped, the scale halves ----
    step(tr)                                                              # first micro-step of the next accumulation window
    tr._arena.grad[7] = float('inf')
    step(tr)
    w2 = tr.imagen.unets[1].final_conv.weight.detach()
    assert torch.equal(w2, w1) and tr.optim1.step_count == 1 and tr.optim1.step_was_skipped
    assert sc.get_scale() == 32768.0 and sc.state_dict()["_growth_tracker"] == 0
    assert float(tr._arena.grad.abs().max()) == 0.0 and float(tr.steps[1]) == 4
    # ... and training goes on
    for _ in range(2):
        step(tr)
    assert tr.optim1.step_count == 2 and not torch.equal(tr.imagen.unets[1].final_conv.weight.detach(), w1)
    # checkpoint: the scaler state travels under the reference's key
    import tempfile, os as _os
    with tempfile.TemporaryDirectory() as d:
        path = _os.path.join(d, "ck.pt")
        tr.save(path)
        ck = torch.load(path, map_location="cpu", weights_only=False)
        assert ck["scaler1"]["scale"] == 32768.0 and ck["scaler1"]["_growth_tracker"] == 1
        tr2 = make(fp16=True)
        tr2.load(path)
        assert tr2.scaler1.state_dict() == tr.scaler1.state_dict()


@pytest.mark.parametrize("mode,use_se", [("bf16", True), ("bf16", False), ("fp16", True)])
def test_low_precision_training_resnet_block_with_16_bit_tensor_and_gradient_between_its_blocks(mode, use_se):
    """A ResnetBlock of a low-precision training step (imagen_pytorch3D.py:568-614 under trainer.py:293-311): block1's output -- a conv
    result autocast rounds anyway -- and the gradient flowing back into it live in the operand type only (written by conv_f9h_kernel /
    the GroupNorm backward, read by the GroupNorm-apply pass, conv_f9h_kernel's backward-data and the weight gradient).
    The 16-bit tensors hold exactly what the conv kernels round the fp32 ones to, so against the same one-node Blocks with fp32 tensors
    between them (DIQT_NO_TRAIN_HALF) and against the two-node path everything agrees to the round-off of the GroupNorm statistics (the
    conv epilogue sums its columns in another order) -- except block1's conv bias gradient, which is now summed from the 16-bit gradient
    (what the reference's autocast backward sums, too) instead of from its fp32 precursor."""
    from diffusioniqt_amd import ops, _lib
    from diffusioniqt_amd.imagen_pytorch3D import ResnetBlock
    B, S, C = 2, 16, 64
    g = torch.Generator().manual_seed(23)
    x0 = torch.randn(B, S, S, S, C, generator=g)
    t0 = torch.randn(B, 32, generator=g)
    dy = torch.randn(B, S, S, S, C, generator=g).to(DEV)
    torch.manual_seed(5)
    blk = ResnetBlock(C, C, time_cond_dim=32, groups=8, use_se=use_se).to(DEV).train()
    for p_ in blk.parameters():
        torch.nn.init.normal_(p_, std=0.2 if p_.dim() == 1 else 0.05)

    def run(fused, half):
        blk.zero_grad(set_to_none=True)
        x = x0.clone().to(DEV).requires_grad_(True)
        t = t0.clone().to(DEV).requires_grad_(True)
        was = ops._NO_TRAIN_FUSE, ops._NO_TRAIN_HALF
        ops._NO_TRAIN_FUSE, ops._NO_TRAIN_HALF = not fused, not half
        ops.FP16_BACKWARD = mode == "fp16"                   # (what the trainer's loss scaler switches on)
        try:
            with ops.low_precision(mode), _lib.census() as c:
                y = blk(x * 1.0, t)
                (y * dy).sum().backward()
                torch.cuda.synchronize()
                n9 = c.count("conv3d_fwd_h(v9h)")
        finally:
            ops._NO_TRAIN_FUSE, ops._NO_TRAIN_HALF = was
            ops.FP16_BACKWARD = False
        out = {k: p_.grad.clone() for k, p_ in blk.named_parameters() if p_.grad is not None}
        out.update(y=y.detach(), gx=x.grad, gt=t.grad)
        return out, n9

    a, na = run(True, True)
    b, nb = run(True, False)
    c2, nc = run(False, False)
    # both forwards + block1's backward-data over the 16-bit gradient (+ block2's, whose output feeds only the SE gate, when there is one)
    assert (na, nb, nc) == (4 if use_se else 3, 2, 0)
    assert a.keys() == b.keys() == c2.keys() and len(a) >= 11
    # one rounding of a value to the operand type can flip when the GroupNorm statistics move in their last fp32 bits (another order of
    # the same column sums): a few 16-bit ulps of the tensor's scale; on this seed the bf16 runs agree bit for bit except that bias
    ulp = 2.0 ** -8 if mode == "bf16" else 2.0 ** -11
    for k in a:
        tol = 3e-2 if k in ("block1.project.bias", "block2.project.bias") else 2 * ulp
        assert (a[k] - b[k]).abs().max().item() <= tol * b[k].abs().max().item() + 1e-6, (k, (a[k] - b[k]).abs().max())
        assert (a[k] - c2[k]).abs().max().item() <= tol * c2[k].abs().max().item() + 1e-6, (k, (a[k] - c2[k]).abs().max())


@pytest.mark.parametrize("ignore_time", [False, True])
def test_family_b_block_of_a_bf16_training_step_as_one_node_is_bit_identical(ignore_time):
    """imagen_video.Block (GroupNorm -> SiLU -> per-frame conv -> temporal conv, imagen_video.py:671-697) in a bf16 training step: the
    GroupNorm-apply + per-frame conv pair runs as one autograd node with a bf16 activation between the two (Conv3d.train_h), against
    the two-node path (``ops._NO_TRAIN_FUSE``): same output, same gradients, bit for bit."""
    from diffusioniqt_amd import ops, _lib
    from diffusioniqt_amd.imagen_video import Block
    torch.manual_seed(5)
    B, Fr, S, C = 2, 8, 64, 64
    blk = Block(C, C).to(DEV)
    with torch.no_grad():
        blk.project.temporal_conv.weight.add_(torch.randn_like(blk.project.temporal_conv.weight) * 0.05)   # not the identity it starts as
    x0 = torch.randn(B, Fr, S, S, C, device=DEV)
    ss0 = torch.randn(B, 2 * C, device=DEV) * 0.3
    r0 = torch.randn(B, Fr, S, S, C, device=DEV)
    dy = torch.randn(B, Fr, S, S, C, device=DEV)

    def run(fused):
        blk.zero_grad(set_to_none=True)
        x, ss, r = (t.clone().requires_grad_(True) for t in (x0, ss0, r0))
        old = ops._NO_TRAIN_FUSE
        ops._NO_TRAIN_FUSE = not fused
        try:
            with ops.low_precision('bf16'), _lib.census() as c:
                y, alias = blk(x * 1.0, scale_shift=ss, ignore_time=ignore_time, residual=r, emit_stats=True, tap=True)
                ((y * dy).sum() + (alias * 0.5).sum()).backward()
                torch.cuda.synchronize()
                n9 = c.count("conv3d_fwd_h(v9h)")
        finally:
            ops._NO_TRAIN_FUSE = old
        grads = [x.grad, ss.grad, r.grad] + [p.grad for p in blk.parameters()]
        return y.detach(), grads, n9

    ya, ga, na = run(True)
    yb, gb, nb = run(False)
    assert na >= 1 and nb == 0
    assert torch.equal(ya, yb)
    for a, b_ in zip(ga, gb):
        if ignore_time and a is None and b_ is None:         # the temporal conv takes no part
            continue
        assert a is not None and b_ is not None and torch.equal(a, b_), (a - b_).abs().max()
