"""TEST INFRASTRUCTURE — CPU oracle for Family B of the DiffusionIQT hot path: the pseudo-3D ``imagen_video.Unet3D``
and the EDM wrapper ``elucidated_imagen.ElucidatedImagen`` (preconditioning, Karras schedule, stochastic Heun
sampler, weighted loss), restated in plain PyTorch fp32 over a reference-named ``state_dict``.

Scope: the text-free IQT instantiation (``cond_on_text=False``, ``attn_pool_text=False`` — SURVEY.md §8 C1/C5) with the
constructor options the reference can run (memory_efficient, temporal_strides, cosine_sim_attn, self_cond,
combine_upsample_fmaps, init_conv_to_final_conv_residual, cond_images_channels; fixtures ``unet3d_opt_*.npz``).  Only ``tests/``, ``smoke()`` and
``bench.py``'s cpu_baseline may import this file.  Pinned against fixtures made by the imported reference
(``oracle/make_golden_b.py`` -> ``tests/golden/unet3d_*.npz``, ``edm_*.npz``).  Citations: reference file:line.
"""
import math
from typing import Dict, Optional, Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def unet3d_config(**kw) -> dict:
    """Unet3D.__init__ defaults (imagen_video.py:1163-1215) for the knobs the IQT path touches."""
    cfg = dict(dim=64, num_resnet_blocks=1, cond_dim=None, num_time_tokens=2, learned_sinu_pos_emb_dim=16,
               dim_mults=(1, 2, 4, 8), channels=3, channels_out=None, attn_dim_head=64, attn_heads=8, ff_mult=2.,
               lowres_cond=False, layer_attns=False, layer_attns_depth=1, attend_at_middle=True,
               time_causal_attn=True, layer_cross_attns=True, init_dim=None, resnet_groups=8, init_conv_kernel_size=7,
               init_cross_embed=True, init_cross_embed_kernel_sizes=(3, 7, 15), use_global_context_attn=True,
               scale_skip_connection=True, final_resnet_block=True, final_conv_kernel_size=3, self_cond=False,
               cond_on_text=True, memory_efficient=False, temporal_strides=1, cosine_sim_attn=False,
               combine_upsample_fmaps=False, init_conv_to_final_conv_residual=False, cond_images_channels=0)
    cfg.update({k: v for k, v in kw.items() if k in cfg})
    n = len(cfg['dim_mults'])
    tup = lambda v: tuple(v) if isinstance(v, (list, tuple)) else (v,) * n
    for k in ('num_resnet_blocks', 'layer_attns', 'layer_attns_depth', 'layer_cross_attns', 'temporal_strides'):
        cfg[k] = tup(cfg[k])
    cfg['init_dim'] = cfg['init_dim'] if cfg['init_dim'] is not None else cfg['dim']
    cfg['cond_dim'] = cfg['cond_dim'] if cfg['cond_dim'] is not None else cfg['dim']
    cfg['channels_out'] = cfg['channels_out'] if cfg['channels_out'] is not None else cfg['channels']
    assert not cfg['cond_on_text'], 'the IQT path is text-free (cond_on_text=False)'
    return cfg


def _ln(x, g):                                  # LayerNorm, gain only — imagen_video.py:172-185
    var = torch.var(x, dim=-1, unbiased=False, keepdim=True)
    mean = torch.mean(x, dim=-1, keepdim=True)
    return (x - mean) * (var + 1e-5).rsqrt() * g


def _chan_ln(x, g):                             # ChanLayerNorm — imagen_video.py:187-200 (g is [1,C,1,1,1])
    var = torch.var(x, dim=1, unbiased=False, keepdim=True)
    mean = torch.mean(x, dim=1, keepdim=True)
    return (x - mean) * (var + 1e-5).rsqrt() * g


def _conv2d(sd, p, x, padding=0):               # Conv2d = nn.Conv3d with (1,k,k) — imagen_video.py:529-543
    w = sd[p + '.weight']
    return F.conv3d(x, w, sd.get(p + '.bias'), padding=(0, padding, padding))


def _pseudo_conv3d(sd, p, x, ignore_time=False):
    """Conv3d (pseudo) — imagen_video.py:352-406: per-frame 3x3 conv then CAUSAL temporal conv1d (left pad k-1)."""
    b, c, f, h, w = x.shape
    ws = sd[p + '.spatial_conv.weight']
    k = ws.shape[-1]
    y = F.conv2d(x.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w), ws, sd[p + '.spatial_conv.bias'], padding=k // 2)
    co = y.shape[1]
    y = y.reshape(b, f, co, h, w).permute(0, 2, 1, 3, 4)
    if ignore_time or (p + '.temporal_conv.weight') not in sd:
        return y
    t = y.permute(0, 3, 4, 1, 2).reshape(b * h * w, co, f)
    t = F.pad(t, (k - 1, 0))
    t = F.conv1d(t, sd[p + '.temporal_conv.weight'], sd[p + '.temporal_conv.bias'])
    return t.reshape(b, h, w, co, f).permute(0, 3, 4, 1, 2)


def _dyn_pos_bias(sd, p, n):
    """DynamicPositionBias — imagen_video.py:1119-1160 -> [heads, n, n]."""
    pos = torch.arange(-n + 1, n, dtype=torch.float32)[:, None]
    i = 0
    while (f'{p}.mlp.{i}.0.weight') in sd:
        pos = F.linear(pos, sd[f'{p}.mlp.{i}.0.weight'], sd[f'{p}.mlp.{i}.0.bias'])
        pos = F.silu(_ln(pos, sd[f'{p}.mlp.{i}.1.g']))
        i += 1
    pos = F.linear(pos, sd[f'{p}.mlp.{i}.weight'], sd[f'{p}.mlp.{i}.bias'])
    idx = torch.arange(n)[:, None] - torch.arange(n)[None, :] + (n - 1)
    return pos[idx].permute(2, 0, 1)


def _attention(sd, p, x, heads, causal=False, context=None, cosine=False):
    """Attention (multi-query, null kv, optional rel-pos bias + causal mask) — imagen_video.py:410-525.  x: [b, n, dim]."""
    b, n, _ = x.shape
    x = _ln(x, sd[p + '.norm.g'])
    q = F.linear(x, sd[p + '.to_q.weight'])
    k, v = F.linear(x, sd[p + '.to_kv.weight']).chunk(2, dim=-1)
    dh = k.shape[-1]
    scale = dh ** -0.5 if not cosine else 1.                                    # :427
    q = q.reshape(b, n, heads, dh).permute(0, 2, 1, 3) * scale
    nk, nv = sd[p + '.null_kv'][0], sd[p + '.null_kv'][1]
    k = torch.cat((nk.expand(b, 1, dh), k), dim=-2)
    v = torch.cat((nv.expand(b, 1, dh), v), dim=-2)
    if context is not None:                                                     # :477-481
        c = F.layer_norm(context, (context.shape[-1],), sd[p + '.to_context.0.weight'], sd[p + '.to_context.0.bias'])
        ck, cv = F.linear(c, sd[p + '.to_context.1.weight'], sd[p + '.to_context.1.bias']).chunk(2, dim=-1)
        k = torch.cat((ck, k), dim=-2)
        v = torch.cat((cv, v), dim=-2)
    if cosine:                                                                    # :484-490
        q, k = F.normalize(q, dim=-1), F.normalize(k, dim=-1)
    sim = torch.einsum('bhid,bjd->bhij', q, k) * (16 if cosine else 1)
    if (p + '.rel_pos_bias.mlp.0.0.weight') in sd:                               # :494-500
        bias = _dyn_pos_bias(sd, p + '.rel_pos_bias', n)
        null_bias = sd[p + '.null_attn_bias'][:, None, None].expand(heads, n, 1)
        sim = sim + torch.cat((null_bias, bias), dim=-1)
    if causal:                                                                    # :506-509
        i, j = sim.shape[-2:]
        sim = sim.masked_fill(torch.ones((i, j), dtype=torch.bool).triu(j - i + 1), -torch.finfo(sim.dtype).max)
    attn = sim.softmax(dim=-1)
    out = torch.einsum('bhij,bjd->bhid', attn, v).permute(0, 2, 1, 3).reshape(b, n, heads * dh)
    return _ln(F.linear(out, sd[p + '.to_out.0.weight']), sd[p + '.to_out.1.g'])


def _cross_attention(sd, p, x, context, heads, cosine=False):
    """CrossAttention — imagen_video.py:772-846 (per-head k/v from the context, shared null kv)."""
    b, n, _ = x.shape
    x = _ln(x, sd[p + '.norm.g'])
    q = F.linear(x, sd[p + '.to_q.weight'])
    k, v = F.linear(context, sd[p + '.to_kv.weight']).chunk(2, dim=-1)
    dh = sd[p + '.null_kv'].shape[-1]          # mid blocks are built with the CrossAttention defaults (8 x 64), :1446
    heads = q.shape[-1] // dh
    split = lambda t: t.reshape(b, -1, heads, dh).permute(0, 2, 1, 3)
    q, k, v = split(q), split(k), split(v)
    nk, nv = sd[p + '.null_kv'][0], sd[p + '.null_kv'][1]
    k = torch.cat((nk.expand(b, heads, 1, dh), k), dim=-2)
    v = torch.cat((nv.expand(b, heads, 1, dh), v), dim=-2)
    q = q * (dh ** -0.5 if not cosine else 1.)
    if cosine:                                                                    # :826-833
        q, k = F.normalize(q, dim=-1), F.normalize(k, dim=-1)
    attn = (torch.einsum('bhid,bhjd->bhij', q, k) * (16 if cosine else 1)).softmax(dim=-1)
    out = torch.einsum('bhij,bhjd->bhid', attn, v).permute(0, 2, 1, 3).reshape(b, n, heads * dh)
    return _ln(F.linear(out, sd[p + '.to_out.0.weight']), sd[p + '.to_out.1.g'])


def _tokens(x):                                  # 'b c f h w -> b (f h w) c'
    b, c = x.shape[:2]
    return x.reshape(b, c, -1).transpose(1, 2)


def _untokens(t, like):
    b, c, f, h, w = like.shape
    return t.transpose(1, 2).reshape(b, t.shape[-1], f, h, w)


def _temporal_peg(sd, p, x, causal=True):
    """Residual(Pad + depthwise Conv3d (3,1,1)) — imagen_video.py:1351-1352."""
    pad = (0, 0, 0, 0, 2, 0) if causal else (0, 0, 0, 0, 1, 1)
    return F.conv3d(F.pad(x, pad), sd[p + '.fn.1.weight'], sd[p + '.fn.1.bias'], groups=x.shape[1]) + x


def _temporal_attn(sd, p, x, heads, causal=True, cosine=False):
    """EinopsToAndFrom('b c f h w', '(b h w) f c', Residual(Attention(causal, rel_pos_bias))) — :1354."""
    b, c, f, h, w = x.shape
    t = x.permute(0, 3, 4, 2, 1).reshape(b * h * w, f, c)
    t = _attention(sd, p + '.fn.fn', t, heads, causal=causal, cosine=cosine) + t
    return t.reshape(b, h, w, f, c).permute(0, 4, 3, 1, 2)


def _block(sd, p, x, scale_shift=None, ignore_time=False):
    """Block — imagen_video.py:671-697: GN(8) -> scale/shift -> SiLU -> pseudo Conv3d."""
    x = F.group_norm(x, 8, sd[p + '.groupnorm.weight'], sd[p + '.groupnorm.bias'], eps=1e-5)
    if scale_shift is not None:
        x = x * (scale_shift[0] + 1) + scale_shift[1]
    return _pseudo_conv3d(sd, p + '.project', F.silu(x), ignore_time)


def _global_context(sd, p, x):
    """GlobalContext — imagen_video.py:957-982."""
    ctx = _conv2d(sd, p + '.to_k', x)
    b, c = x.shape[:2]
    out = torch.einsum('bin,bcn->bci', ctx.reshape(b, 1, -1).softmax(dim=-1), x.reshape(b, c, -1))[..., None, None]
    out = F.silu(_conv2d(sd, p + '.net.0', out))
    return torch.sigmoid(_conv2d(sd, p + '.net.2', out))


def _resnet_block(sd, p, x, t, cond, heads, ignore_time=False, cosine=False):
    """ResnetBlock — imagen_video.py:699-770.  ``cosine``: the block was built with the U-Net's attn_kwargs (level init blocks, :1411;
    the middle blocks are not, :1446-1450)."""
    scale_shift = None
    if (p + '.time_mlp.1.weight') in sd and t is not None:
        te = F.linear(F.silu(t), sd[p + '.time_mlp.1.weight'], sd[p + '.time_mlp.1.bias'])[:, :, None, None, None]
        scale_shift = te.chunk(2, dim=1)
    h = _block(sd, p + '.block1', x, ignore_time=ignore_time)
    if (p + '.cross_attn.fn.to_q.weight') in sd:
        assert cond is not None
        h = _untokens(_cross_attention(sd, p + '.cross_attn.fn', _tokens(h), cond, heads, cosine), h) + h
    h = _block(sd, p + '.block2', h, scale_shift=scale_shift, ignore_time=ignore_time)
    if (p + '.gca.to_k.weight') in sd:
        h = h * _global_context(sd, p + '.gca', h)
    res = _conv2d(sd, p + '.res_conv', x) if (p + '.res_conv.weight') in sd else x
    return h + res


def _transformer_block(sd, p, x, context, heads, depth, cosine=False):
    """TransformerBlock — imagen_video.py:1004-1029 (Attention with the conditioning tokens as extra keys + ChanFF)."""
    for i in range(depth):
        x = _untokens(_attention(sd, f'{p}.layers.{i}.0.fn', _tokens(x), heads, context=context, cosine=cosine), x) + x
        ff = f'{p}.layers.{i}.1'
        hdn = F.gelu(_conv2d(sd, ff + '.1', _chan_ln(x, sd[ff + '.0.g'])))
        x = _conv2d(sd, ff + '.4', _chan_ln(hdn, sd[ff + '.3.g'])) + x
    return x


def _learned_sinu(w, t):
    tt = t[:, None]
    fr = tt * w[None, :] * 2 * math.pi
    return torch.cat((tt, fr.sin(), fr.cos()), dim=-1)


def _downsample(sd, p, x):                                                     # Downsample :595-600
    b_, c_, f_, h_, w_ = x.shape
    y = x.reshape(b_, c_, f_, h_ // 2, 2, w_ // 2, 2).permute(0, 1, 4, 6, 2, 3, 5).reshape(b_, c_ * 4, f_, h_ // 2, w_ // 2)
    return _conv2d(sd, p + '.1', y)


def _pixel_shuffle_up(sd, p, x):                                                # PixelShuffleUpsample :564-593
    y = F.silu(_conv2d(sd, p + '.net.0', x))
    b_, c4, f_, h_, w_ = y.shape
    y = F.pixel_shuffle(y.permute(0, 2, 1, 3, 4).reshape(b_ * f_, c4, h_, w_), 2)
    return y.reshape(b_, f_, c4 // 4, h_ * 2, w_ * 2).permute(0, 2, 1, 3, 4)


def _resize_video(x, size):                                                     # resize_video_to :137-158 (frames kept)
    if x.shape[-1] == size:
        return x
    return F.interpolate(x, (x.shape[2], size, size), mode='nearest')


def unet3d_forward(sd: Dict[str, Tensor], cfg: dict, x: Tensor, time: Tensor, *, lowres_cond_img=None,
                   lowres_noise_times=None, ignore_time=False, self_cond=None, cond_images=None) -> Tensor:
    """Unet3D.forward — imagen_video.py:1585-1822 for the text-free configuration.  x: [b, c, f, h, w]."""
    assert x.ndim == 5
    heads = cfg['attn_heads']
    causal = cfg['time_causal_attn']
    cos = cfg['cosine_sim_attn']
    mem = cfg['memory_efficient']
    assert not (cfg['lowres_cond'] and lowres_cond_img is None), 'low resolution conditioning image must be present'
    assert not (cfg['lowres_cond'] and lowres_noise_times is None), 'low resolution conditioning noise time must be present'
    if cfg['self_cond']:                                                          # :1605-1609
        x = torch.cat((x, self_cond if self_cond is not None else torch.zeros_like(x)), dim=1)
    if lowres_cond_img is not None:
        x = torch.cat((x, lowres_cond_img), dim=1)
    assert (cfg['cond_images_channels'] > 0) == (cond_images is not None)
    if cond_images is not None:                                                   # :1621-1627
        x = torch.cat((_resize_video(cond_images, x.shape[-1]), x), dim=1)
    if cfg['init_cross_embed']:                                                   # CrossEmbedLayer :1058-1083, stride 1
        ks = sorted(cfg['init_cross_embed_kernel_sizes'])
        x = torch.cat([_conv2d(sd, f'init_conv.convs.{i}', x, padding=(k - 1) // 2) for i, k in enumerate(ks)], dim=1)
    else:
        x = _conv2d(sd, 'init_conv', x, padding=cfg['init_conv_kernel_size'] // 2)
    if not ignore_time:
        x = _temporal_peg(sd, 'init_temporal_peg', x, causal)
        x = _temporal_attn(sd, 'init_temporal_attn', x, heads, causal, cos)
    init_conv_residual = x.clone() if cfg['init_conv_to_final_conv_residual'] else None

    th = F.silu(F.linear(_learned_sinu(sd['to_time_hiddens.0.weights'], time), sd['to_time_hiddens.1.weight'], sd['to_time_hiddens.1.bias']))
    r = cfg['num_time_tokens']
    time_tokens = F.linear(th, sd['to_time_tokens.0.weight'], sd['to_time_tokens.0.bias']).reshape(x.shape[0], r, -1)
    t = F.linear(th, sd['to_time_cond.0.weight'], sd['to_time_cond.0.bias'])
    if cfg['lowres_cond']:                                                        # :1659-1665
        lh = F.silu(F.linear(_learned_sinu(sd['to_lowres_time_hiddens.0.weights'], lowres_noise_times),
                             sd['to_lowres_time_hiddens.1.weight'], sd['to_lowres_time_hiddens.1.bias']))
        ltok = F.linear(lh, sd['to_lowres_time_tokens.0.weight'], sd['to_lowres_time_tokens.0.bias']).reshape(x.shape[0], r, -1)
        t = t + F.linear(lh, sd['to_lowres_time_cond.0.weight'], sd['to_lowres_time_cond.0.bias'])
        time_tokens = torch.cat((time_tokens, ltok), dim=-2)
    c = F.layer_norm(time_tokens, (time_tokens.shape[-1],), sd['norm_cond.weight'], sd['norm_cond.bias'])   # :1732-1736

    if mem:                                                                       # :1744-1745
        x = _resnet_block(sd, 'init_resnet_block', x, t, None, heads, ignore_time, cos)

    n_levels = len(cfg['dim_mults'])
    hiddens = []
    for i in range(n_levels):
        p = f'downs.{i}'
        if mem:                                                                   # pre-downsample :1754-1755
            x = _downsample(sd, p + '.0', x)
        x = _resnet_block(sd, p + '.1', x, t, c, heads, ignore_time, cos)
        for j in range(cfg['num_resnet_blocks'][i]):
            x = _resnet_block(sd, f'{p}.2.{j}', x, t, None, heads, ignore_time)
            hiddens.append(x)
        if cfg['layer_attns'][i]:
            x = _transformer_block(sd, p + '.3', x, c, heads, cfg['layer_attns_depth'][i], cos)
        if not ignore_time:
            x = _temporal_peg(sd, p + '.4', x, causal)
            x = _temporal_attn(sd, p + '.5', x, heads, causal, cos)
        hiddens.append(x)
        ts = cfg['temporal_strides'][i]
        if ts > 1 and not ignore_time:                                            # TemporalDownsample :636-641
            b_, c_, f_, h_, w_ = x.shape
            y = x.reshape(b_, c_, f_ // ts, ts, h_, w_).permute(0, 1, 3, 2, 4, 5).reshape(b_, c_ * ts, f_ // ts, h_, w_)
            x = _conv2d(sd, p + '.6.1', y)
        if not mem:
            if i < n_levels - 1:
                x = _downsample(sd, p + '.7', x)
            else:                                                                 # Parallel(3x3, 1x1) :1429
                x = _conv2d(sd, p + '.7.fns.0', x, padding=1) + _conv2d(sd, p + '.7.fns.1', x)

    x = _resnet_block(sd, 'mid_block1', x, t, c, heads, ignore_time)
    if cfg['attend_at_middle']:
        tk = _tokens(x)
        x = _untokens(_attention(sd, 'mid_attn.fn.fn', tk, heads, cosine=cos) + tk, x)
    if not ignore_time:
        x = _temporal_peg(sd, 'mid_temporal_peg', x, causal)
        x = _temporal_attn(sd, 'mid_temporal_attn', x, heads, causal, cos)
    x = _resnet_block(sd, 'mid_block2', x, t, c, heads, ignore_time)

    skip = 1. if not cfg['scale_skip_connection'] else 2 ** -0.5
    up_hiddens = []
    for i in range(n_levels):
        p = f'ups.{i}'
        lvl = n_levels - 1 - i
        ts = cfg['temporal_strides'][lvl]
        if ts > 1 and not ignore_time:                                            # TemporalPixelShuffleUpsample :604-634
            b_, c_, f_, h_, w_ = x.shape
            v = x.permute(0, 3, 4, 1, 2).reshape(b_ * h_ * w_, c_, f_)
            v = F.silu(F.conv1d(v, sd[p + '.5.net.0.weight'], sd[p + '.5.net.0.bias']))
            co = v.shape[1] // ts
            v = v.reshape(b_ * h_ * w_, co, ts, f_).permute(0, 1, 3, 2).reshape(b_ * h_ * w_, co, f_ * ts)     # 'b (c r) n -> b c (n r)'
            x = v.reshape(b_, h_, w_, co, f_ * ts).permute(0, 3, 4, 1, 2)
        x = torch.cat((x, hiddens.pop() * skip), dim=1)
        x = _resnet_block(sd, p + '.0', x, t, c, heads, ignore_time, cos)
        for j in range(cfg['num_resnet_blocks'][lvl]):
            x = torch.cat((x, hiddens.pop() * skip), dim=1)
            x = _resnet_block(sd, f'{p}.1.{j}', x, t, None, heads, ignore_time)
        if cfg['layer_attns'][lvl]:
            x = _transformer_block(sd, p + '.2', x, c, heads, cfg['layer_attns_depth'][lvl], cos)
        if not ignore_time:
            x = _temporal_peg(sd, p + '.3', x, causal)
            x = _temporal_attn(sd, p + '.4', x, heads, causal, cos)
        up_hiddens.append(x)
        if i < n_levels - 1 or mem:
            x = _pixel_shuffle_up(sd, p + '.6', x)
    if cfg['combine_upsample_fmaps']:                                             # UpsampleCombiner :1085-1117
        outs = [_block(sd, f'upsample_combiner.fmap_convs.{k}', _resize_video(fm, x.shape[-1]), ignore_time=False)
                for k, fm in enumerate(up_hiddens)]
        x = torch.cat((x, *outs), dim=1)
    if init_conv_residual is not None:
        x = torch.cat((x, init_conv_residual), dim=1)
    if cfg['final_resnet_block']:
        x = _resnet_block(sd, 'final_res_block', x, t, None, heads, ignore_time)
    if lowres_cond_img is not None:
        x = torch.cat((x, lowres_cond_img), dim=1)
    return _conv2d(sd, 'final_conv', x, padding=cfg['final_conv_kernel_size'] // 2)


# ----------------------------------------------------------------------------------------------
# EDM (elucidated_imagen.py)
# ----------------------------------------------------------------------------------------------
EDM_DEFAULTS = dict(num_sample_steps=32, sigma_min=0.002, sigma_max=80, sigma_data=0.5, rho=7, P_mean=-1.2, P_std=1.2,
                    S_churn=80, S_tmin=0.05, S_tmax=50, S_noise=1.003)                     # :96-106


def c_skip(sd_, s): return (sd_ ** 2) / (s ** 2 + sd_ ** 2)                                # :314-315
def c_out(sd_, s): return s * sd_ * (sd_ ** 2 + s ** 2) ** -0.5                            # :317-318
def c_in(sd_, s): return 1 * (s ** 2 + sd_ ** 2) ** -0.5                                   # :320-321
def c_noise(s): return torch.log(s.clamp(min=1e-20)) * 0.25                                # :323-324, :71-72
def loss_weight(sd_, s): return (s ** 2 + sd_ ** 2) * (s * sd_) ** -2                      # :706-707


def sample_schedule(N, rho, sigma_min, sigma_max):                                          # :365-379
    inv_rho = 1 / rho
    steps = torch.arange(N, dtype=torch.float32)
    sigmas = (sigma_max ** inv_rho + steps / (N - 1) * (sigma_min ** inv_rho - sigma_max ** inv_rho)) ** rho
    return F.pad(sigmas, (0, 1), value=0.)


def gammas_of(sigmas, hp):                                                                  # :418-422
    return torch.where((sigmas >= hp['S_tmin']) & (sigmas <= hp['S_tmax']),
                       min(hp['S_churn'] / hp['num_sample_steps'], math.sqrt(2) - 1), 0.)


def beta_linear_log_snr(t):
    return -torch.log(torch.special.expm1(1e-4 + 10 * (t ** 2)))


def lowres_q_sample(x, t, noise):
    """lowres_noise_schedule.q_sample with the LINEAR schedule — elucidated_imagen.py:134,657,819."""
    ls = beta_linear_log_snr(t)
    a, s = torch.sqrt(torch.sigmoid(ls)), torch.sqrt(torch.sigmoid(-ls))
    sh = (-1,) + (1,) * (x.ndim - 1)
    return a.view(sh) * x + s.view(sh) * noise


def threshold_x_start(x_start: Tensor, dynamic: bool, percentile: float = 0.95) -> Tensor:
    """elucidated_imagen.py:298-311"""
    if not dynamic:
        return x_start.clamp(-1., 1.)
    s = torch.quantile(x_start.flatten(1).abs(), percentile, dim=-1).clamp(min=1.)
    s = s.view(-1, *((1,) * (x_start.ndim - 1)))
    return x_start.clamp(-s, s) / s


def preconditioned(unet_fn, noised, sigma: Tensor, sigma_data, clamp=False, dynamic=False, percentile=0.95, **unet_kw):
    """preconditioned_network_forward — :329-358 (``unet_kw``: e.g. self_cond, forwarded to the network)"""
    ps = sigma.view(-1, *((1,) * (noised.ndim - 1)))
    net = unet_fn(c_in(sigma_data, ps) * noised, c_noise(sigma), **unet_kw)
    out = c_skip(sigma_data, ps) * noised + c_out(sigma_data, ps) * net
    return threshold_x_start(out, dynamic, percentile) if clamp else out


def edm_sample(unet_fn, shape, init_noise: Tensor, step_noises: Sequence[Tensor], hp: dict, dynamic=False, percentile=0.95,
               self_cond=False):
    """one_unet_sample — elucidated_imagen.py:382-532 with injected noise (draw order :430, :476).
    ``unet_fn(x, c_noise)`` closes over the low-res conditioning; ``self_cond``: the last x0 estimate is fed back (:483, 505, 524)."""
    sigmas = sample_schedule(hp['num_sample_steps'], hp['rho'], hp['sigma_min'], hp['sigma_max'])
    gammas = gammas_of(sigmas, hp)
    images = sigmas[0] * init_noise
    b = shape[0]
    x_start = None
    sc = (lambda v: dict(self_cond=v)) if self_cond else (lambda v: {})
    for ind, (sigma, sigma_next, gamma) in enumerate(zip(sigmas[:-1].tolist(), sigmas[1:].tolist(), gammas[:-1].tolist())):
        eps = hp['S_noise'] * step_noises[ind]
        sigma_hat = sigma + gamma * sigma
        images_hat = images + math.sqrt(sigma_hat ** 2 - sigma ** 2) * eps
        out = preconditioned(unet_fn, images_hat, torch.full((b,), sigma_hat), hp['sigma_data'], clamp=True, dynamic=dynamic,
                             percentile=percentile, **sc(x_start))
        d = (images_hat - out) / sigma_hat
        images_next = images_hat + (sigma_next - sigma_hat) * d
        x_start = out
        if sigma_next != 0:
            out2 = preconditioned(unet_fn, images_next, torch.full((b,), sigma_next), hp['sigma_data'], clamp=True,
                                  dynamic=dynamic, percentile=percentile, **sc(out))
            d2 = (images_next - out2) / sigma_next
            images_next = images_hat + 0.5 * (sigma_next - sigma_hat) * (d + d2)
            x_start = out2
        images = images_next
    return images.clamp(-1., 1.)


def edm_loss(unet_fn, images: Tensor, sigmas: Tensor, noise: Tensor, sigma_data, self_cond_draw=False):
    """ElucidatedImagen.forward tail — :823-882: x + sigma*eps -> preconditioned net -> weighted MSE mean.
    ``self_cond_draw``: the 50 % branch of a self-conditioning U-Net (:847-860) — a gradient-free x0 estimate is fed back."""
    ps = sigmas.view(-1, *((1,) * (images.ndim - 1)))
    kw = {}
    if self_cond_draw:
        with torch.no_grad():
            kw['self_cond'] = preconditioned(unet_fn, images + ps * noise, sigmas, sigma_data).detach()
    den = preconditioned(unet_fn, images + ps * noise, sigmas, sigma_data, **kw)
    losses = F.mse_loss(den, images, reduction='none').flatten(1).mean(1)
    return (losses * loss_weight(sigma_data, sigmas)).mean()
