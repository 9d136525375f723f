"""MI355X-native counterpart of the reference's ``imagen_video.Unet3D`` (Family B, imagen_video.py:1162-1822):
the pseudo-3D U-Net (per-frame (1,k,k) convs + causal temporal convs, GroupNorm+SiLU, GlobalContext gating,
multi-query softmax attention over (f h w) tokens and causal temporal attention with a T5-style dynamic
position bias) for the text-free IQT configuration (``cond_on_text=False``).

Same constructor kwargs / ``forward`` signature / ``state_dict`` keys and shapes as the reference.  Activations
are channels-last ``[B, F, H, W, C]``; convolutions run on the MFMA implicit-GEMM kernel, attention as two
strided batched MFMA GEMMs around one fused bias/mask/soft-max kernel (multi-query: all heads of a sequence
share K/V, so QK^T is ONE GEMM with M = tokens x heads).
"""
import math
from functools import partial

import os

import torch
from torch import nn

from . import ops
from .ops import ACT_SILU, ACT_GELU, ACT_SIGMOID
from .imagen_pytorch3D import (exists, default, cast_tuple, Conv3d as _Conv3dND, Linear, Act, Identity,
                               LearnedSinusoidalPosEmb, to_channels_last, to_channels_first, print_once, TimeCond, BatchedTimeMLPs)


def SiLU():
    return Act(ACT_SILU)


def Conv2d(dim_in, dim_out, kernel, stride=1, padding=0, **kwargs):
    """imagen_video.py:529-543: a Conv3d with a (1,k,k) kernel."""
    kernel, stride, padding = cast_tuple(kernel, 2), cast_tuple(stride, 2), cast_tuple(padding, 2)
    assert tuple(stride) == (1, 1), 'strided pseudo-2D convs are outside the IQT path'
    return _Conv3dND(dim_in, dim_out, (1, *kernel), stride=(1, *stride), padding=(0, *padding), **kwargs)


class LayerNorm(nn.Module):
    """gain-only LayerNorm over the last axis (imagen_video.py:172-185)."""

    def __init__(self, dim, stable=False):
        super().__init__()
        assert not stable
        self.g = nn.Parameter(torch.ones(dim))

    def forward(self, x, residual=None, tap=False):
        return ops.chan_layernorm(x, self.g, 1e-5, residual=residual, tap=tap)


class ChanLayerNorm(nn.Module):
    """imagen_video.py:187-200; g is [1, C, 1, 1, 1]."""

    def __init__(self, dim, stable=False):
        super().__init__()
        assert not stable
        self.g = nn.Parameter(torch.ones(1, dim, 1, 1, 1))

    def forward(self, x, tap=False):
        return ops.chan_layernorm(x, self.g, 1e-5, tap=tap)


class AffineLayerNorm(nn.LayerNorm):
    """nn.LayerNorm (weight + bias) parameters, HIP kernel underneath (imagen_video.py:444, 1306)."""

    def forward(self, x):
        return ops.chan_layernorm(x, self.weight, self.eps, bias=self.bias)


class Residual(nn.Module):
    def __init__(self, fn):
        super().__init__()
        self.fn = fn

    def forward(self, x, **kwargs):
        if isinstance(self.fn, Attention):           # its to_out ends in a LayerNorm: `+ x` rides in that kernel
            return self.fn(x, residual=x, **kwargs)
        if isinstance(self.fn, nn.Sequential) and len(self.fn) == 2 and isinstance(self.fn[1], TemporalPEGConv):
            return self.fn[1](self.fn[0](x), residual=x)                          # temporal PEG: conv(pad(x)) + x in one kernel
        return ops.add(self.fn(x, **kwargs), x)


class Parallel(nn.Module):
    """sum of branches (imagen_video.py:217-224); two convs: the second accumulates in the first's epilogue."""

    def __init__(self, *fns):
        super().__init__()
        self.fns = nn.ModuleList(fns)

    def forward(self, x):
        out = self.fns[0](x)
        for fn in self.fns[1:]:
            out = fn(x, residual=out) if isinstance(fn, _Conv3dND) else ops.add(fn(x), out)
        return out


class TokensOverSpaceTime(nn.Module):
    """EinopsToAndFrom('b c f h w', 'b (f h w) c', fn): channels-last makes this a free view."""

    def __init__(self, fn):
        super().__init__()
        self.fn = fn

    def forward(self, x, residual=None, **kwargs):
        B, F, H, W, C = x.shape
        tokens = x.reshape(B, F * H * W, C)
        if residual is not None:                     # `fn(x) + residual`, added inside fn (Attention); `+ x` stays recognisable as such
            kwargs['residual'] = tokens if residual is x else residual.reshape(B, F * H * W, -1)
        return self.fn(tokens, **kwargs).reshape(B, F, H, W, -1)


class TokensOverTime(nn.Module):
    """EinopsToAndFrom('b c f h w', '(b h w) f c', fn): one mid-axis transpose each way."""

    def __init__(self, fn):
        super().__init__()
        self.fn = fn

    def forward(self, x, **kwargs):
        B, F, H, W, C = x.shape
        fn = self.fn.fn if isinstance(self.fn, Residual) else None
        if isinstance(fn, Attention) and not kwargs and fn.block_ok(B, F, H * W, C):
            # sampling under autocast: the whole block -- both LayerNorms, the projections, the attention, the residual -- in one kernel
            return fn.forward_block(x.reshape(B, F, H * W, C)).reshape(B, F, H, W, C)
        if isinstance(fn, Attention) and not kwargs and fn.frames_ok(B * H * W):
            # sampling: LayerNorm and the projections are per-row, and the attention kernel walks the frame axis in place
            return fn.forward_frames(x.reshape(B, F, H * W, C)).reshape(B, F, H, W, C)
        t = ops.transpose_mid(x.reshape(B, F, H * W, C)).reshape(B * H * W, F, C)
        t = self.fn(t, **kwargs)
        return ops.transpose_mid(t.reshape(B, H * W, F, C)).reshape(B, F, H, W, C)


class Conv3d(nn.Module):
    """Pseudo-3D conv (imagen_video.py:352-406): per-frame k x k conv, then a CAUSAL temporal conv (dirac init)."""

    def __init__(self, dim, dim_out=None, kernel_size=3, *, temporal_kernel_size=None, **kwargs):
        super().__init__()
        dim_out = default(dim_out, dim)
        temporal_kernel_size = default(temporal_kernel_size, kernel_size)
        self.spatial_conv = nn.Conv2d(dim, dim_out, kernel_size=kernel_size, padding=kernel_size // 2)
        self.temporal_conv = nn.Conv1d(dim_out, dim_out, kernel_size=temporal_kernel_size) if kernel_size > 1 else None
        self.kernel_size = kernel_size
        if exists(self.temporal_conv):
            nn.init.dirac_(self.temporal_conv.weight.data)
            nn.init.zeros_(self.temporal_conv.bias.data)

    def forward(self, x, ignore_time=False, residual=None, want_stats=False, gn=None, out_half=False):
        """``residual`` is added in the epilogue of the LAST conv of the pair (the caller's ``h + res``), whose per-tile column sums
        (``want_stats``) feed the consumer's GroupNorm.  ``gn`` = (GroupNorm module, scale_shift): x is the RAW GroupNorm input and the
        per-frame conv applies GroupNorm + SiLU while staging it (sampling path, ops.gn_conv3d); returns None when that is not taken."""
        sc = self.spatial_conv
        k = self.kernel_size
        last = ignore_time or not exists(self.temporal_conv)
        if gn is not None:
            norm, ss = gn
            if not last and ops.lp_mode() is not None:
                # sampling under autocast: GroupNorm-apply writes the operand type, and so does the per-frame conv (None: shape not taken)
                tc = self.temporal_conv
                kt = tc.weight.shape[-1]
                return ops.conv_pair_nograd_h(x, sc.weight.unsqueeze(2), sc.bias, (0, k // 2, k // 2), tc.weight.unsqueeze(-1).unsqueeze(-1),
                                              tc.bias, (kt - 1, 0, 0), (-(kt - 1), 0, 0), residual,
                                              gn=(norm.weight, norm.bias, ss, norm.num_groups, ACT_SILU, norm.eps), want_stats=want_stats,
                                              out_half=out_half)
            x = ops.gn_conv3d(x, norm.weight, norm.bias, ss, norm.num_groups, ACT_SILU, norm.eps, sc.weight.unsqueeze(2), sc.bias,
                              (0, k // 2, k // 2), residual if last else None, want_stats=want_stats and last)
            if x is None:
                return None
        else:
            if not last:        # sampling under autocast: the tensor between the two convs stays in the operand type (half the bytes)
                tc = self.temporal_conv
                kt = tc.weight.shape[-1]
                y = ops.conv_pair_nograd_h(x, sc.weight.unsqueeze(2), sc.bias, (0, k // 2, k // 2), tc.weight.unsqueeze(-1).unsqueeze(-1),
                                           tc.bias, (kt - 1, 0, 0), (-(kt - 1), 0, 0), residual, want_stats=want_stats)
                if y is not None:
                    return y
            x = ops.conv3d(x, sc.weight.unsqueeze(2), sc.bias, (0, k // 2, k // 2), residual=residual if last else None,
                           want_stats=want_stats and last)
        if last:
            return x
        tc = self.temporal_conv
        kt = tc.weight.shape[-1]
        w5 = tc.weight.unsqueeze(-1).unsqueeze(-1)                       # [Co, Co, kt, 1, 1]
        return ops.conv3d(x, w5, tc.bias, (kt - 1, 0, 0), residual=residual, extra_pad=(-(kt - 1), 0, 0),    # left pad k-1 only (:399-402)
                          want_stats=want_stats)


    def train_h(self, x, norm, ss, ignore_time=False, residual=None, want_stats=False, tap=False):
        """Training step under autocast: ``norm`` + SiLU + the per-frame conv as one autograd node whose activation between the two is
        kept in the operand type (ops.gn_conv3d_train_h), then the temporal conv.  None: shape not taken."""
        sc = self.spatial_conv
        k = self.kernel_size
        last = ignore_time or not exists(self.temporal_conv)
        out = ops.gn_conv3d_train_h(x, norm.weight, norm.bias, ss, norm.num_groups, ACT_SILU, norm.eps, sc.weight.unsqueeze(2), sc.bias,
                                    (0, k // 2, k // 2), residual if last else None, want_stats=want_stats and last, tap=tap)
        if out is None or last:
            return out
        y, alias = out if tap else (out, None)
        tc = self.temporal_conv
        kt = tc.weight.shape[-1]
        y = ops.conv3d(y, tc.weight.unsqueeze(-1).unsqueeze(-1), tc.bias, (kt - 1, 0, 0), residual=residual, extra_pad=(-(kt - 1), 0, 0),
                       want_stats=want_stats)
        return (y, alias) if tap else y


class DynamicPositionBias(nn.Module):
    """imagen_video.py:1119-1160 — returns the table [2n-1, heads]; the (i - j) gather is fused into the soft-max kernel."""

    def __init__(self, dim, *, heads, depth):
        super().__init__()
        self.mlp = nn.ModuleList([nn.Sequential(Linear(1, dim), LayerNorm(dim), SiLU())])
        for _ in range(max(depth - 1, 0)):
            self.mlp.append(nn.Sequential(Linear(dim, dim), LayerNorm(dim), SiLU()))
        self.mlp.append(Linear(dim, heads))

    def forward(self, n, device):
        # the table depends on the parameters and n only, not on the input: on the sampling path it is computed once and kept until a
        # parameter changes (the reference re-runs the MLP in every attention call of every U-Net evaluation: 7 small kernels each)
        cache = not torch.is_grad_enabled()
        if cache:
            # one table per (n, device) for as long as the parameters are unchanged: a captured hipGraph holds a table's address, so a
            # call with another frame count must not replace (and free) the one an earlier capture reads
            wkey = (ops._WEIGHT_EPOCH,) + tuple((p.data_ptr(), p._version) for p in self.parameters())
            tabs = getattr(self, '_tables', None)
            if tabs is None or tabs[0] != wkey:
                if tabs is not None:
                    ops.retire(*tabs[1].values())
                tabs = self._tables = (wkey, {})
            key = (n, str(device))
            hit = tabs[1].get(key)
            if hit is not None:
                return hit
        pos = torch.arange(-n + 1, n, device=device, dtype=torch.float32).unsqueeze(-1)
        for layer in self.mlp:
            pos = layer(pos)
        if cache:
            tabs[1][key] = pos
        return pos


_UNFUSED_ATTN = os.environ.get("DIQT_UNFUSED_ATTN") == "1"     # A/B switch: training attention through GEMM -> soft-max -> GEMM


class Attention(nn.Module):
    """Multi-query attention with a learned null key/value, optional conditioning tokens as extra keys, optional
    relative position bias + causal mask (imagen_video.py:410-525).  x: [G, n, dim]."""

    def __init__(self, dim, *, dim_head=64, heads=8, causal=False, context_dim=None, cosine_sim_attn=False,
                 rel_pos_bias=False, rel_pos_bias_mlp_depth=2, init_zero=False):
        super().__init__()
        # cosine-sim attention (:427-431, 484-490): unit scale, l2-normalised queries and keys, similarities times 16
        self.cosine_sim_attn = cosine_sim_attn
        self.scale = dim_head ** -0.5 if not cosine_sim_attn else 16.
        self.causal, self.heads, self.dim_head = causal, heads, dim_head
        self.rel_pos_bias = DynamicPositionBias(dim=dim, heads=heads, depth=rel_pos_bias_mlp_depth) if rel_pos_bias else None
        inner_dim = dim_head * heads
        self.norm = LayerNorm(dim)
        self.null_attn_bias = nn.Parameter(torch.randn(heads))
        self.null_kv = nn.Parameter(torch.randn(2, dim_head))
        self.to_q = Linear(dim, inner_dim, bias=False)
        self.to_kv = Linear(dim, dim_head * 2, bias=False)
        self.to_context = nn.Sequential(AffineLayerNorm(context_dim), Linear(context_dim, dim_head * 2)) if exists(context_dim) else None
        self.to_out = nn.Sequential(Linear(inner_dim, dim, bias=False), LayerNorm(dim))
        if init_zero:
            nn.init.zeros_(self.to_out[-1].g)

    def _out(self, out, residual):
        return self.to_out[1](self.to_out[0](out), residual=residual)              # Linear -> LayerNorm (+ residual)

    def frames_ok(self, G):
        return (not torch.is_grad_enabled() and ops.lp_mode() is None and self.dim_head in (32, 64) and G <= 65535
                and self.to_context is None and not self.cosine_sim_attn)

    def block_ok(self, B, F, P, C):
        return (not torch.is_grad_enabled() and ops.lp_mode() is not None and self.to_context is None and not self.cosine_sim_attn
                and ops.temporal_attention_h_ok(B, F, P, C, self.heads, self.dim_head))

    def forward_block(self, x):
        """``Residual(Attention)`` over the frame axis of x[B, F, P, C] as ONE kernel (sampling path under autocast)."""
        lp = ops.lp_mode()
        ws = (self.to_q.weight, self.to_kv.weight, self.to_out[0].weight)
        key = (lp, ops._WEIGHT_EPOCH) + tuple((w.data_ptr(), w._version) for w in ws)
        packs = self.__dict__.setdefault('_packed_h', {})                      # one pack per operand type (a graph may hold either)
        hit = packs.get(lp)
        if hit is None or hit[0] != key:
            if hit is not None:
                ops.retire(*hit[1])
            hit = packs[lp] = (key, ops.pack_temporal_attention_h(*ws, self.heads, self.dim_head, self.scale, lp))
        rel = null_bias = None
        if exists(self.rel_pos_bias):
            rel = self.rel_pos_bias(x.shape[1], x.device).contiguous()
            null_bias = self.null_attn_bias.contiguous()
        return ops.temporal_attention_h(x, self.norm.g, hit[1], self.to_out[1].g, self.null_kv.reshape(-1).contiguous(), rel, null_bias,
                                        self.heads, self.dim_head, self.causal, 1e-5, lp)

    def forward_frames(self, x):
        """``Residual(Attention)`` over the FRAME axis of x[B, F, P, C] without leaving that layout (sampling path): what
        EinopsToAndFrom('b c f h w', '(b h w) f c', ...) computes through two transposes (imagen_video.py:1351-1354)."""
        B, F, P, _ = x.shape
        h, d = self.heads, self.dim_head
        xn = self.norm(x)
        q = self.to_q(xn)                                                         # [B, F, P, h*d]
        kv = self.to_kv(xn)                                                       # [B, F, P, 2d]
        rel = null_bias = None
        if exists(self.rel_pos_bias):
            rel = self.rel_pos_bias(F, x.device).contiguous()
            null_bias = self.null_attn_bias.contiguous()
        out = ops.mqa_attention_frames_nograd(q.contiguous(), kv.contiguous(), self.null_kv.reshape(-1).contiguous(), rel, null_bias,
                                              h, d, self.causal, self.scale)
        return self._out(out, x)

    def forward(self, x, context=None, mask=None, attn_bias=None, residual=None):
        assert mask is None and attn_bias is None
        G, n, _ = x.shape
        h, d = self.heads, self.dim_head
        if residual is x:      # `attn(x) + x`: the skip reads x through the LayerNorm's alias (its gradient joins the LayerNorm backward)
            x, residual = self.norm(x, tap=True)
        else:
            x = self.norm(x)
        q = self.to_q(x)                                                          # [G, n, h*d]
        kv = self.to_kv(x)                                                        # [G, n, 2d]  (k | v), shared by all heads
        extra = self.null_kv.reshape(1, 2 * d).expand(G, 2 * d)                   # null key/value row
        E = 1
        if exists(context):
            assert exists(self.to_context)
            ckv = self.to_context(context)                                        # [G, nc, 2d]
            E += ckv.shape[1]
            extra = ops.concat_channels(ckv.reshape(G, -1), extra)                # [context..., null]   (:471-481)
        kv_ext = ops.concat_channels(extra, kv.reshape(G, n * 2 * d))             # [G, (E+n)*2d]
        M = E + n
        if self.cosine_sim_attn:                                                  # l2norm of q and of EVERY key (null, context, self)
            q = ops.l2norm_rows(q.reshape(G, n * h, d)).reshape(G, n, h * d)
            kn = ops.l2norm_rows(kv_ext.reshape(G * M, 2 * d), 0, d)              # the k half of the k|v rows
            _, vv = ops.split_channels(kv_ext.reshape(G * M, 2 * d), d)
            kv_ext = ops.concat_channels(kn, vv).reshape(G, M * 2 * d)
        rel = null_bias = None
        if exists(self.rel_pos_bias):
            rel = self.rel_pos_bias(n, x.device)                                  # [2n-1, h]
            null_bias = self.null_attn_bias                                        # only added together with the bias (:497-500)
        if not torch.is_grad_enabled() and d in (32, 64) and G <= 65535:
            # sampling path: fused QK^T -> bias/mask -> softmax -> PV, the [G, n*h, M] scores stay in registers
            out = ops.mqa_attention_nograd(q.contiguous(), kv_ext.reshape(G, M, 2 * d).contiguous(),
                                           rel.contiguous() if exists(rel) else None,
                                           null_bias.contiguous() if exists(null_bias) else None, n, h, d, E, n, self.causal, self.scale)
            return self._out(out, residual)
        if not _UNFUSED_ATTN and ops.mqa_attention_fused_ok(G, n, h, d, n, exists(rel)):
            # training path: the same fused kernel family with autograd (flash-style backward, no materialised scores)
            out = ops.mqa_attention(q, kv_ext.reshape(G, M, 2 * d), rel, null_bias, n, h, d, E, n, self.causal, self.scale)
            return self._out(out, residual)
        sim = ops.bmm_strided(q, kv_ext, (G, n * h, M, d, False, True, n * h * d, d, M * 2 * d, 2 * d, n * h * M, M,
                                          self.scale, (G, n, h, M), 0, 0))
        p = ops.attn_softmax(sim, rel, null_bias, n, h, E, n, self.causal)
        out = ops.bmm_strided(p, kv_ext, (G, n * h, d, M, False, False, n * h * M, M, M * 2 * d, 2 * d, n * h * d, d,
                                          1.0, (G, n, h * d), 0, d))
        return self._out(out, residual)


class CrossAttention(nn.Module):
    """Per-head keys/values from the conditioning tokens + shared null kv (imagen_video.py:772-846)."""

    def __init__(self, dim, *, context_dim=None, dim_head=64, heads=8, norm_context=False, cosine_sim_attn=False):
        super().__init__()
        assert not norm_context
        self.cosine_sim_attn = cosine_sim_attn                                    # (:784-786, 826-833)
        self.scale = dim_head ** -0.5 if not cosine_sim_attn else 16.
        self.heads, self.dim_head = heads, dim_head
        inner_dim = dim_head * heads
        context_dim = default(context_dim, dim)
        self.norm = LayerNorm(dim)
        self.norm_context = Identity()
        self.null_kv = nn.Parameter(torch.randn(2, dim_head))
        self.to_q = Linear(dim, inner_dim, bias=False)
        self.to_kv = Linear(context_dim, inner_dim * 2, bias=False)
        self.to_out = nn.Sequential(Linear(inner_dim, dim, bias=False), LayerNorm(dim))

    def forward(self, x, context, mask=None):
        assert mask is None
        B, n, _ = x.shape
        h, d = self.heads, self.dim_head
        m = context.shape[1]
        M = m + 1
        q = self.to_q(self.norm(x))                                               # [B, n, h*d]
        kv = self.to_kv(context)                                                  # [B, m, 2*h*d]  (k heads | v heads)
        null_row = torch.cat((self.null_kv[0].repeat(h), self.null_kv[1].repeat(h))).reshape(1, 2 * h * d).expand(B, -1)
        kv_ext = ops.concat_channels(null_row, kv.reshape(B, m * 2 * h * d)).reshape(B, M, 2 * h * d)   # null first (:817-820)
        if self.cosine_sim_attn:
            q = ops.l2norm_rows(q.reshape(B, n * h, d)).reshape(B, n, h * d)
            kk, vv = ops.split_channels(kv_ext.reshape(B * M, 2 * h * d), h * d)  # (k heads | v heads) per key
            kk = ops.l2norm_rows(kk.reshape(B * M * h, d)).reshape(B * M, h * d)
            kv_ext = ops.concat_channels(kk, vv).reshape(B, M, 2 * h * d)
        outs = []
        for b in range(B):
            sim = ops.bmm_strided(q[b], kv_ext[b], (h, n, M, d, False, True, d, h * d, d, 2 * h * d, M, h * M, self.scale,
                                                    (n, h, M), 0, 0))
            p = ops.attn_softmax(sim, None, None, n, h, 1, m, False)
            outs.append(ops.bmm_strided(p, kv_ext[b], (h, n, d, M, False, False, M, h * M, d, 2 * h * d, d, h * d, 1.0,
                                                       (n, h * d), 0, h * d)))
        out = outs[0].unsqueeze(0) if B == 1 else torch.stack(outs, dim=0)
        return self.to_out(out)


class Block(nn.Module):
    """GN(8) -> x*(scale+1)+shift -> SiLU (one kernel) -> pseudo Conv3d (imagen_video.py:671-697)."""

    def __init__(self, dim, dim_out, groups=8, norm=True):
        super().__init__()
        assert norm
        self.groupnorm = nn.GroupNorm(groups, dim)
        self.activation = SiLU()
        self.project = Conv3d(dim, dim_out, 3, padding=1)

    def forward(self, x, scale_shift=None, ignore_time=False, residual=None, emit_stats=False, tap=False, out_half=False):
        """``tap``: also returns an alias of x for its other consumer (the ResnetBlock's residual branch; ops.groupnorm_act).
        ``out_half`` (sampling under autocast): the output only feeds another Block's GroupNorm and may come back in the operand type."""
        gn = self.groupnorm
        if not torch.is_grad_enabled():
            # sampling: GroupNorm-apply + SiLU inside the per-frame conv's input staging (no elementwise pass); None: shape not taken
            y = self.project(x, ignore_time=ignore_time, residual=residual, want_stats=emit_stats, gn=(gn, scale_shift), out_half=out_half)
            if y is not None:
                return (y, x) if tap else y
            if x.dtype != torch.float32:                 # a 16-bit block output whose consumer does not take the 16-bit path after all
                x = x.float()
        elif ops.lp_mode() is not None:
            # low-precision training step: GroupNorm-apply -> operand type -> per-frame conv as one autograd node (ops._GnActConvHFn)
            out = self.project.train_h(x, gn, scale_shift, ignore_time=ignore_time, residual=residual, want_stats=emit_stats, tap=tap)
            if out is not None:
                return out
        if tap:
            x, alias = ops.groupnorm_act(x, gn.weight, gn.bias, scale_shift, gn.num_groups, ACT_SILU, gn.eps, tap=True)
            return self.project(x, ignore_time=ignore_time, residual=residual, want_stats=emit_stats), alias
        x = ops.groupnorm_act(x, gn.weight, gn.bias, scale_shift, gn.num_groups, ACT_SILU, gn.eps)
        return self.project(x, ignore_time=ignore_time, residual=residual, want_stats=emit_stats)


class GlobalContext(nn.Module):
    """imagen_video.py:957-982 — returns the per-(b, c) gate; ResnetBlock fuses gate * h + residual."""

    def __init__(self, *, dim_in, dim_out):
        super().__init__()
        self.to_k = Conv2d(dim_in, 1, 1)
        hidden_dim = max(3, dim_out // 2)
        self.net = nn.Sequential(Conv2d(dim_in, hidden_dim, 1), SiLU(), Conv2d(hidden_dim, dim_out, 1), Act(ACT_SIGMOID))

    def forward(self, x):
        B, C = x.shape[0], x.shape[-1]
        n = x.numel() // (B * C)
        pooled = ops.softmax_pool_nograd(x.reshape(B, n, C), self.to_k.weight.reshape(C))      # sampling: to_k, soft-max and pooling in ONE pass
        if pooled is not None:
            return self.net(pooled.reshape(B, 1, 1, 1, C)).reshape(B, -1)
        ctx = self.to_k(x).reshape(B, n)
        p = ops.softmax(ctx, dim=-1)                                              # over all (f h w) positions
        pooled = ops.weighted_pool(p, x.reshape(B, n, C))                         # [B, C] = softmax(ctx) . x
        return self.net(pooled.reshape(B, 1, 1, 1, C)).reshape(B, -1)


class ResnetBlock(nn.Module):
    """imagen_video.py:699-770."""

    def __init__(self, dim, dim_out, *, cond_dim=None, time_cond_dim=None, groups=8, linear_attn=False, use_gca=False,
                 squeeze_excite=False, **attn_kwargs):
        super().__init__()
        assert not linear_attn, 'linear cross attention is outside the IQT path'
        self.time_mlp = nn.Sequential(SiLU(), Linear(time_cond_dim, dim_out * 2)) if exists(time_cond_dim) else None
        self.cross_attn = TokensOverSpaceTime(CrossAttention(dim=dim_out, context_dim=cond_dim, **attn_kwargs)) if exists(cond_dim) else None
        self.block1 = Block(dim, dim_out, groups=groups)
        self.block2 = Block(dim_out, dim_out, groups=groups)
        self.gca = GlobalContext(dim_in=dim_out, dim_out=dim_out) if use_gca else None
        self.res_conv = Conv2d(dim, dim_out, 1) if dim != dim_out else Identity()

    def forward(self, x, time_emb=None, cond=None, ignore_time=False):
        scale_shift = None
        if exists(self.time_mlp) and exists(time_emb):                            # [B, 2C]: scale | shift
            if isinstance(time_emb, TimeCond):       # SiLU(t) shared by all blocks, the Linears batched on the sampling path
                pre = time_emb.batched.get(id(self.time_mlp[1])) if time_emb.batched is not None else None
                scale_shift = pre if pre is not None else self.time_mlp[1](time_emb.activated())
            else:
                scale_shift = self.time_mlp(time_emb)
        # block2's GroupNorm statistics come from the epilogue of block1's last conv (unless cross attention rewrites h in between)
        # (the residual branch reads x through block1's alias: its gradient is added inside the GroupNorm backward)
        h, x = self.block1(x, ignore_time=ignore_time, emit_stats=not exists(self.cross_attn), tap=True, out_half=not exists(self.cross_attn))
        if exists(self.cross_attn):
            assert exists(cond)
            h = ops.add(self.cross_attn(h, context=cond), h)
        res = self.res_conv(x)
        if exists(self.gca):
            h = self.block2(h, scale_shift=scale_shift, ignore_time=ignore_time)
            return ops.gate_residual(h, self.gca(h), res)                         # h * gca(h) + res_conv(x)
        return self.block2(h, scale_shift=scale_shift, ignore_time=ignore_time, residual=res)      # h + res in the conv epilogue


def ChanFeedForward(dim, mult=2):
    hidden_dim = int(dim * mult)
    return nn.Sequential(ChanLayerNorm(dim), Conv2d(dim, hidden_dim, 1, bias=False), Act(ACT_GELU), ChanLayerNorm(hidden_dim),
                         Conv2d(hidden_dim, dim, 1, bias=False))


class TransformerBlock(nn.Module):
    """imagen_video.py:1004-1029."""

    def __init__(self, dim, *, depth=1, heads=8, dim_head=32, ff_mult=2, context_dim=None, cosine_sim_attn=False):
        super().__init__()
        self.layers = nn.ModuleList([])
        for _ in range(depth):
            self.layers.append(nn.ModuleList([
                TokensOverSpaceTime(Attention(dim=dim, heads=heads, dim_head=dim_head, context_dim=context_dim,
                                              cosine_sim_attn=cosine_sim_attn)),
                ChanFeedForward(dim=dim, mult=ff_mult)]))

    def forward(self, x, context=None):
        for attn, ff in self.layers:
            x = attn(x, context=context, residual=x)                              # attn(x) + x in the attention's last LayerNorm
            h, x = ff[0](x, tap=True)                                             # ChanLayerNorm; the skip reads x through its alias
            for layer in ff[1:-1]:
                h = layer(h)
            x = ff[-1](h, residual=x)                                             # ff(x) + x in the last 1x1 conv's epilogue
        return x


class CrossEmbedLayer(nn.Module):
    """imagen_video.py:1058-1083 (stride 1 on the IQT path)."""

    def __init__(self, dim_in, kernel_sizes, dim_out=None, stride=2):
        super().__init__()
        assert stride == 1 and all((k % 2) == (stride % 2) for k in kernel_sizes)
        dim_out = default(dim_out, dim_in)
        kernel_sizes = sorted(kernel_sizes)
        n = len(kernel_sizes)
        dim_scales = [int(dim_out / (2 ** i)) for i in range(1, n)]
        dim_scales = [*dim_scales, dim_out - sum(dim_scales)]
        self.convs = nn.ModuleList([Conv2d(dim_in, ds, k, stride=stride, padding=(k - stride) // 2)
                                    for k, ds in zip(kernel_sizes, dim_scales)])

    def forward(self, x):
        out = None
        for conv in self.convs:
            y = conv(x)
            out = y if out is None else ops.concat_channels(out, y)
        return out


class SpaceToDepth2D(nn.Module):
    """Rearrange 'b c f (h p1) (w p2) -> b (c p1 p2) f h w' (imagen_video.py:598)."""

    def forward(self, x):
        return ops.space_to_depth_nd(x, (1, 2, 2))


def Downsample(dim, dim_out=None):
    return nn.Sequential(SpaceToDepth2D(), Conv2d(dim * 4, default(dim_out, dim), 1))


class PixelShuffleUpsample(nn.Module):
    """imagen_video.py:564-593: 1x1 conv (C -> 4C') -> SiLU -> per-frame PixelShuffle(2)."""

    def __init__(self, dim, dim_out=None):
        super().__init__()
        dim_out = default(dim_out, dim)
        conv = Conv2d(dim, dim_out * 4, 1)
        self.net = nn.Sequential(conv, SiLU())
        o, i, f, h, w = conv.weight.shape
        w0 = torch.empty(o // 4, i, f, h, w)
        nn.init.kaiming_uniform_(w0)
        conv.weight.data.copy_(w0.repeat_interleave(4, dim=0))
        nn.init.zeros_(conv.bias.data)

    def forward(self, x):
        return ops.depth_to_space_nd(self.net(x), (1, 2, 2))


class TemporalDownsample(nn.Sequential):
    """nn.Sequential(Rearrange('b c (f p) h w -> b (c p) f h w'), Conv2d(dim * p, dim_out, 1)) (imagen_video.py:636-641); the conv keeps
    index 1 for the state-dict key."""

    def __init__(self, dim, dim_out=None, stride=2):
        super().__init__(Identity(), Conv2d(dim * stride, default(dim_out, dim), 1))
        self.stride = stride

    def forward(self, x):
        return self[1](ops.space_to_depth_nd(x, (self.stride, 1, 1)))


class TemporalPixelShuffleUpsample(nn.Module):
    """Conv1d(dim, dim_out * r, 1) over the frame axis -> SiLU -> 'b (c r) n -> b c (n r)' (imagen_video.py:604-634)."""

    def __init__(self, dim, dim_out=None, stride=2):
        super().__init__()
        self.stride = stride
        dim_out = default(dim_out, dim)
        conv = nn.Conv1d(dim, dim_out * stride, 1)
        self.net = nn.Sequential(conv, SiLU())
        o, i, f = conv.weight.shape
        w0 = torch.empty(o // stride, i, f)
        nn.init.kaiming_uniform_(w0)
        conv.weight.data.copy_(w0.repeat_interleave(stride, dim=0))
        nn.init.zeros_(conv.bias.data)

    def forward(self, x):
        conv = self.net[0]
        y = self.net[1](ops.linear(x, conv.weight.squeeze(-1), conv.bias))        # per-position linear == the 1-wide conv1d
        return ops.depth_to_space_nd(y, (self.stride, 1, 1))


class UpsampleCombiner(nn.Module):
    """imagen_video.py:1085-1117: every up-path feature map, resized (nearest) to the output size and passed through a Block, is
    concatenated to the final feature map."""

    def __init__(self, dim, *, enabled=False, dim_ins=tuple(), dim_outs=tuple()):
        super().__init__()
        dim_outs = cast_tuple(dim_outs, len(dim_ins))
        assert len(dim_ins) == len(dim_outs)
        self.enabled = enabled
        if not enabled:
            self.dim_out = dim
            return
        self.fmap_convs = nn.ModuleList([Block(di, do) for di, do in zip(dim_ins, dim_outs)])
        self.dim_out = dim + (sum(dim_outs) if len(dim_outs) > 0 else 0)

    def forward(self, x, fmaps=None):
        fmaps = default(fmaps, tuple())
        if not self.enabled or len(fmaps) == 0 or len(self.fmap_convs) == 0:
            return x
        size = x.shape[3]
        for fmap, conv in zip(fmaps, self.fmap_convs):
            if fmap.shape[3] != size:                                             # resize_video_to: frames kept, H = W = size
                fmap = ops.nearest_resize(fmap, (fmap.shape[1], size, size))
            x = ops.concat_channels(x, conv(fmap))
        return x


class CausalPad(nn.Module):
    """Pad((0,0,0,0,2,0)) of the temporal PEG (imagen_video.py:1351) — folded into the conv's one-sided padding."""

    def forward(self, x):
        return x


class TemporalPEGConv(nn.Conv3d):
    """nn.Conv3d(dim, dim, (3,1,1), groups=dim) after a causal (2,0) / symmetric (1,1) frame pad."""

    def __init__(self, dim, causal):
        super().__init__(dim, dim, (3, 1, 1), groups=dim)
        self.causal = causal

    def forward(self, x, residual=None):
        if ops.dwconv_temporal_ok(x, self.weight, self.groups, self.stride):      # elementwise-class kernel, `+ residual` in the same pass
            return ops.dwconv_temporal(x, self.weight, self.bias, 2 if self.causal else 1, residual)
        pad, extra = ((2, 0, 0), (-2, 0, 0)) if self.causal else ((1, 0, 0), (0, 0, 0))
        y = ops.conv3d_direct(x, self.weight, self.bias, (1, 1, 1), pad, self.groups, extra_pad=extra)
        return y if residual is None else ops.add(y, residual)


class Unet3D(nn.Module):
    def __init__(
        self, *, dim, image_embed_dim=1024, text_embed_dim=768, num_resnet_blocks=1, cond_dim=None, num_image_tokens=4,
        num_time_tokens=2, learned_sinu_pos_emb_dim=16, out_dim=None, dim_mults=(1, 2, 4, 8), temporal_strides=1,
        cond_images_channels=0, channels=3, channels_out=None, attn_dim_head=64, attn_heads=8, ff_mult=2.,
        lowres_cond=False, layer_attns=False, layer_attns_depth=1, layer_attns_add_text_cond=True, attend_at_middle=True,
        time_rel_pos_bias_depth=2, time_causal_attn=True, layer_cross_attns=True, use_linear_attn=False,
        use_linear_cross_attn=False, cond_on_text=True, max_text_len=256, init_dim=None, resnet_groups=8,
        init_conv_kernel_size=7, init_cross_embed=True, init_cross_embed_kernel_sizes=(3, 7, 15),
        cross_embed_downsample=False, cross_embed_downsample_kernel_sizes=(2, 4), attn_pool_text=True,
        attn_pool_num_latents=32, dropout=0., memory_efficient=False, init_conv_to_final_conv_residual=False,
        use_global_context_attn=True, scale_skip_connection=True, final_resnet_block=True, final_conv_kernel_size=3,
        cosine_sim_attn=False, self_cond=False, combine_upsample_fmaps=False, pixel_shuffle_upsample=True,
    ):
        super().__init__()
        self._locals = {k: v for k, v in locals().items() if k not in ('self', '__class__')}
        assert attn_heads > 1, 'you need to have more than 1 attention head, ideally at least 4 or 8'
        if cond_on_text or attn_pool_text and cond_on_text:
            raise NotImplementedError('text conditioning (T5) is not part of the IQT hot path (SURVEY.md §2 #13): use cond_on_text=False')
        # options the REFERENCE itself cannot run (probed against /root/reference): the nearest Upsample doubles the frame
        # axis too and its output no longer concatenates with the skip tensors (imagen_video.py:556-562 -> RuntimeError in forward),
        # cross_embed_downsample passes kernel_sizes twice (:1381 -> TypeError in __init__), use_linear_attn's 2-D rearrange meets a 5-D
        # tensor (:888-955 -> EinopsError in forward).  use_linear_cross_attn works there and is not built.
        broken = dict(nearest_upsample=not pixel_shuffle_upsample, cross_embed_downsample=cross_embed_downsample, use_linear_attn=use_linear_attn)
        bad = [k for k, v in broken.items() if v]
        if bad:
            raise NotImplementedError(f'Unet3D options that raise in the reference as well: {bad}')
        if use_linear_cross_attn:
            raise NotImplementedError('use_linear_cross_attn is outside the IQT hot path (SURVEY.md §8)')
        if dim < 128:
            print_once('The base dimension of your u-net should ideally be no smaller than 128, as recommended by a '
                       'professional DDPM trainer https://nonint.com/2022/05/04/friends-dont-let-friends-train-small-diffusion-models/')
        self.self_cond = self_cond
        self.channels = channels
        self.channels_out = default(channels_out, channels)
        init_channels = channels * (1 + int(lowres_cond) + int(self_cond))
        init_dim = default(init_dim, dim)
        self.has_cond_image = cond_images_channels > 0
        self.cond_images_channels = cond_images_channels
        init_channels += cond_images_channels

        self.init_conv = CrossEmbedLayer(init_channels, dim_out=init_dim, kernel_sizes=init_cross_embed_kernel_sizes, stride=1) \
            if init_cross_embed else Conv2d(init_channels, init_dim, init_conv_kernel_size, padding=init_conv_kernel_size // 2)
        dims = [init_dim, *map(lambda m: dim * m, dim_mults)]
        in_out = list(zip(dims[:-1], dims[1:]))
        cond_dim = default(cond_dim, dim)
        time_cond_dim = dim * 4 * (2 if lowres_cond else 1)
        self.num_time_tokens = num_time_tokens

        self.to_time_hiddens = nn.Sequential(LearnedSinusoidalPosEmb(learned_sinu_pos_emb_dim),
                                             Linear(learned_sinu_pos_emb_dim + 1, time_cond_dim), SiLU())
        self.to_time_cond = nn.Sequential(Linear(time_cond_dim, time_cond_dim))
        self.to_time_tokens = nn.Sequential(Linear(time_cond_dim, cond_dim * num_time_tokens), Identity())
        self.lowres_cond = lowres_cond
        if lowres_cond:
            self.to_lowres_time_hiddens = nn.Sequential(LearnedSinusoidalPosEmb(learned_sinu_pos_emb_dim),
                                                        Linear(learned_sinu_pos_emb_dim + 1, time_cond_dim), SiLU())
            self.to_lowres_time_cond = nn.Sequential(Linear(time_cond_dim, time_cond_dim))
            self.to_lowres_time_tokens = nn.Sequential(Linear(time_cond_dim, cond_dim * num_time_tokens), Identity())
        self.norm_cond = AffineLayerNorm(cond_dim)
        self.text_to_cond = None
        self.cond_on_text = cond_on_text
        self.attn_pool = None
        self.max_text_len = max_text_len
        self.null_text_embed = nn.Parameter(torch.randn(1, max_text_len, cond_dim))      # kept for state-dict parity
        self.null_text_hidden = nn.Parameter(torch.randn(1, time_cond_dim))
        self.to_text_non_attn_cond = None

        attn_kwargs = dict(heads=attn_heads, dim_head=attn_dim_head, cosine_sim_attn=cosine_sim_attn)
        num_layers = len(in_out)

        def temporal_peg(d):
            return Residual(nn.Sequential(CausalPad(), TemporalPEGConv(d, time_causal_attn)))

        def temporal_attn(d):
            return TokensOverTime(Residual(Attention(d, **{**attn_kwargs, 'causal': time_causal_attn, 'init_zero': True,
                                                          'rel_pos_bias': True, 'rel_pos_bias_mlp_depth': time_rel_pos_bias_depth})))

        num_resnet_blocks = cast_tuple(num_resnet_blocks, num_layers)
        resnet_groups = cast_tuple(resnet_groups, num_layers)
        resnet_klass = partial(ResnetBlock, **attn_kwargs)
        layer_attns = cast_tuple(layer_attns, num_layers)
        layer_attns_depth = cast_tuple(layer_attns_depth, num_layers)
        layer_cross_attns = cast_tuple(layer_cross_attns, num_layers)
        assert all(len(t) == num_layers for t in (resnet_groups, layer_attns, layer_cross_attns))
        temporal_strides = cast_tuple(temporal_strides, num_layers)
        self.total_temporal_divisor = 1
        for ts in temporal_strides:
            self.total_temporal_divisor *= ts
        self.init_resnet_block = resnet_klass(init_dim, init_dim, time_cond_dim=time_cond_dim, groups=resnet_groups[0],
                                              use_gca=use_global_context_attn) if memory_efficient else None
        self.init_temporal_peg = temporal_peg(init_dim)
        self.init_temporal_attn = temporal_attn(init_dim)
        self.skip_connect_scale = 1. if not scale_skip_connection else (2 ** -0.5)

        self.downs = nn.ModuleList([])
        self.ups = nn.ModuleList([])
        skip_connect_dims = []
        params = list(zip(num_resnet_blocks, resnet_groups, layer_attns, layer_attns_depth, layer_cross_attns, temporal_strides))
        for ind, ((dim_in, dim_out), (n_blocks, groups, layer_attn, layer_attn_depth, layer_cross_attn, temporal_stride)) in enumerate(zip(in_out, params)):
            is_last = ind >= (num_layers - 1)
            layer_cond_dim = cond_dim if layer_cross_attn else None
            current_dim = dim_in
            pre_downsample = None
            if memory_efficient:                      # memory-efficient U-Net: down-sample in front of the level's blocks (:1400-1404)
                pre_downsample = Downsample(dim_in, dim_out)
                current_dim = dim_out
            skip_connect_dims.append(current_dim)
            post_downsample = None
            if not memory_efficient:
                post_downsample = Downsample(current_dim, dim_out) if not is_last else \
                    Parallel(Conv2d(dim_in, dim_out, 3, padding=1), Conv2d(dim_in, dim_out, 1))
            transformer = TransformerBlock(dim=current_dim, depth=layer_attn_depth, ff_mult=ff_mult, context_dim=cond_dim,
                                           **attn_kwargs) if layer_attn else Identity()
            self.downs.append(nn.ModuleList([
                pre_downsample,
                resnet_klass(current_dim, current_dim, cond_dim=layer_cond_dim, time_cond_dim=time_cond_dim, groups=groups),
                nn.ModuleList([ResnetBlock(current_dim, current_dim, time_cond_dim=time_cond_dim, groups=groups,
                                           use_gca=use_global_context_attn) for _ in range(n_blocks)]),
                transformer, temporal_peg(current_dim), temporal_attn(current_dim),
                TemporalDownsample(current_dim, stride=temporal_stride) if temporal_stride > 1 else None, post_downsample]))

        mid_dim = dims[-1]
        self.mid_block1 = ResnetBlock(mid_dim, mid_dim, cond_dim=cond_dim, time_cond_dim=time_cond_dim, groups=resnet_groups[-1])
        self.mid_attn = TokensOverSpaceTime(Residual(Attention(mid_dim, **attn_kwargs))) if attend_at_middle else None
        self.mid_temporal_peg = temporal_peg(mid_dim)
        self.mid_temporal_attn = temporal_attn(mid_dim)
        self.mid_block2 = ResnetBlock(mid_dim, mid_dim, cond_dim=cond_dim, time_cond_dim=time_cond_dim, groups=resnet_groups[-1])

        upsample_fmap_dims = []
        for ind, ((dim_in, dim_out), (n_blocks, groups, layer_attn, layer_attn_depth, layer_cross_attn, temporal_stride)) in enumerate(
                zip(reversed(in_out), reversed(params))):
            is_last = ind == (len(in_out) - 1)
            layer_cond_dim = cond_dim if layer_cross_attn else None
            skip_connect_dim = skip_connect_dims.pop()
            upsample_fmap_dims.append(dim_out)
            transformer = TransformerBlock(dim=dim_out, depth=layer_attn_depth, ff_mult=ff_mult, context_dim=cond_dim,
                                           **attn_kwargs) if layer_attn else Identity()
            self.ups.append(nn.ModuleList([
                resnet_klass(dim_out + skip_connect_dim, dim_out, cond_dim=layer_cond_dim, time_cond_dim=time_cond_dim, groups=groups),
                nn.ModuleList([ResnetBlock(dim_out + skip_connect_dim, dim_out, time_cond_dim=time_cond_dim, groups=groups,
                                           use_gca=use_global_context_attn) for _ in range(n_blocks)]),
                transformer, temporal_peg(dim_out), temporal_attn(dim_out),
                TemporalPixelShuffleUpsample(dim_out, stride=temporal_stride) if temporal_stride > 1 else None,
                PixelShuffleUpsample(dim_out, dim_in) if not is_last or memory_efficient else Identity()]))

        # feature maps of every up level joined to the output (:1471-1478); residual from the initial conv (:1480-1483)
        self.upsample_combiner = UpsampleCombiner(dim=dim, enabled=combine_upsample_fmaps, dim_ins=upsample_fmap_dims, dim_outs=dim)
        self.init_conv_to_final_conv_residual = init_conv_to_final_conv_residual
        final_conv_dim = self.upsample_combiner.dim_out + (dim if init_conv_to_final_conv_residual else 0)
        self.final_res_block = ResnetBlock(final_conv_dim, dim, time_cond_dim=time_cond_dim, groups=resnet_groups[0], use_gca=True) \
            if final_resnet_block else None
        final_conv_dim_in = (dim if final_resnet_block else final_conv_dim) + (channels if lowres_cond else 0)
        self.final_conv = Conv2d(final_conv_dim_in, self.channels_out, final_conv_kernel_size, padding=final_conv_kernel_size // 2)
        nn.init.zeros_(self.final_conv.weight)
        nn.init.zeros_(self.final_conv.bias)

    def cast_model_parameters(self, *, lowres_cond, text_embed_dim=None, channels, channels_out, cond_on_text=False):
        if lowres_cond == self.lowres_cond and channels == self.channels and cond_on_text == self.cond_on_text and \
                text_embed_dim == self._locals['text_embed_dim'] and channels_out == self.channels_out:
            return self
        return self.__class__(**{**self._locals, **dict(lowres_cond=lowres_cond, text_embed_dim=text_embed_dim,
                                                        channels=channels, channels_out=channels_out, cond_on_text=cond_on_text)})

    def to_config_and_state_dict(self):
        return self._locals, self.state_dict()

    @classmethod
    def from_config_and_state_dict(klass, config, state_dict):
        unet = klass(**config)
        unet.load_state_dict(state_dict)
        return unet

    def forward_with_cond_scale(self, *args, cond_scale=1., **kwargs):
        logits = self.forward(*args, **kwargs)
        if cond_scale == 1:
            return logits
        raise NotImplementedError('classifier-free guidance needs text conditioning, which the IQT path does not use')

    def _time_tokens(self, module_h, module_tok, module_cond, time):
        hid = module_h(time.float().contiguous())
        B = hid.shape[0]
        tokens = module_tok[0](hid)                                               # [B, r*cond_dim]  ('b (r d) -> b r d')
        return hid, tokens, module_cond(hid)

    def forward(self, x, time, *, lowres_cond_img=None, lowres_noise_times=None, text_embeds=None, text_mask=None,
                cond_images=None, self_cond=None, cond_drop_prob=0., ignore_time=False):
        """(x[B,C,F,H,W], time[B]) -> [B,C_out,F,H,W]  (imagen_video.py:1585-1822)."""
        assert x.ndim == 5, 'input to 3d unet must have 5 dimensions (batch, channels, time, height, width)'
        assert text_embeds is None
        frames = x.shape[2]
        assert ignore_time or frames % self.total_temporal_divisor == 0, \
            f'number of input frames {frames} must be divisible by {self.total_temporal_divisor}'
        assert not (self.lowres_cond and not exists(lowres_cond_img)), 'low resolution conditioning image must be present'
        assert not (self.lowres_cond and not exists(lowres_noise_times)), 'low resolution conditioning noise time must be present'
        x = to_channels_last(x.float())
        if self.self_cond:                                                        # (:1605-1609) the previous x0 estimate, or zeros
            sc = to_channels_last(self_cond.float()) if exists(self_cond) else torch.zeros_like(x)
            x = ops.concat_channels(x, sc)
        lowres_cl = None
        if exists(lowres_cond_img):
            lowres_cl = to_channels_last(lowres_cond_img.float())
            x = ops.concat_channels(x, lowres_cl)
        assert not (self.has_cond_image ^ exists(cond_images)), \
            'you either requested to condition on an image on the unet, but the conditioning image is not supplied, or vice versa'
        if exists(cond_images):                                                   # (:1621-1627) resized (nearest) and put IN FRONT
            assert cond_images.shape[1] == self.cond_images_channels, \
                'the number of channels on the conditioning image you are passing in does not match what you specified on initialiation of the unet'
            ci = to_channels_last(cond_images.float())
            if ci.shape[3] != x.shape[3]:
                ci = ops.nearest_resize(ci, (ci.shape[1], x.shape[3], x.shape[3]))
            x = ops.concat_channels(ci, x)
        x = self.init_conv(x)
        if not ignore_time:
            x = self.init_temporal_peg(x)
            x = self.init_temporal_attn(x)
        init_conv_residual = x if self.init_conv_to_final_conv_residual else None

        B = x.shape[0]
        _, tok, t = self._time_tokens(self.to_time_hiddens, self.to_time_tokens, self.to_time_cond, time)
        if self.lowres_cond:
            _, ltok, lowres_t = self._time_tokens(self.to_lowres_time_hiddens, self.to_lowres_time_tokens,
                                                  self.to_lowres_time_cond, lowres_noise_times)
            t = ops.add(t, lowres_t)
            tok = ops.concat_channels(tok, ltok)                                  # tokens concatenated along the sequence axis
        r = self.num_time_tokens * (2 if self.lowres_cond else 1)
        c = self.norm_cond(tok.reshape(B, r, -1))                                 # conditioning tokens  (:1732-1736)
        # every ResnetBlock's time_mlp starts with the same SiLU(t) (:716-719): evaluated once; on the sampling path the ~20 Linears run
        # as one launch over their concatenated weights
        t = TimeCond(t, ACT_SILU)
        if getattr(self, '_time_mlps', None) is None:
            self._time_mlps = BatchedTimeMLPs(m.time_mlp[1] for m in self.modules() if isinstance(m, ResnetBlock) and m.time_mlp is not None)
        t.batched = self._time_mlps(t)

        if exists(self.init_resnet_block):
            x = self.init_resnet_block(x, t, ignore_time=ignore_time)
        hiddens = []
        for pre_downsample, init_block, resnet_blocks, attn_block, temporal_peg, temporal_attn, temporal_downsample, post_downsample in self.downs:
            if exists(pre_downsample):
                x = pre_downsample(x)
            x = init_block(x, t, c, ignore_time=ignore_time)
            for resnet_block in resnet_blocks:
                x = resnet_block(x, t, ignore_time=ignore_time)
                hiddens.append(x)
            if not isinstance(attn_block, Identity):
                x = attn_block(x, c)
            if not ignore_time:
                x = temporal_peg(x)
                x = temporal_attn(x)
            hiddens.append(x)
            if exists(temporal_downsample) and not ignore_time:
                x = temporal_downsample(x)
            if exists(post_downsample):
                x = post_downsample(x)

        x = self.mid_block1(x, t, c, ignore_time=ignore_time)
        if exists(self.mid_attn):
            x = self.mid_attn(x)
        if not ignore_time:
            x = self.mid_temporal_peg(x)
            x = self.mid_temporal_attn(x)
        x = self.mid_block2(x, t, c, ignore_time=ignore_time)

        def add_skip_connection(x):
            return ops.concat_channels(x, hiddens.pop(), 1.0, self.skip_connect_scale, want_stats=True)   # cat(x, skip * scale) in one pass (+ the next GroupNorm's column sums)

        up_hiddens = []
        for init_block, resnet_blocks, attn_block, temporal_peg, temporal_attn, temporal_upsample, upsample in self.ups:
            if exists(temporal_upsample) and not ignore_time:
                x = temporal_upsample(x)
            x = add_skip_connection(x)
            x = init_block(x, t, c, ignore_time=ignore_time)
            for resnet_block in resnet_blocks:
                x = add_skip_connection(x)
                x = resnet_block(x, t, ignore_time=ignore_time)
            if not isinstance(attn_block, Identity):
                x = attn_block(x, c)
            if not ignore_time:
                x = temporal_peg(x)
                x = temporal_attn(x)
            up_hiddens.append(x)
            x = upsample(x)

        x = self.upsample_combiner(x, up_hiddens)
        if exists(init_conv_residual):
            x = ops.concat_channels(x, init_conv_residual)
        if exists(self.final_res_block):
            x = self.final_res_block(x, t, ignore_time=ignore_time)
        if exists(lowres_cl):
            x = ops.concat_channels(x, lowres_cl)
        return to_channels_first(self.final_conv(x))
