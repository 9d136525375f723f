"""MI355X-native counterpart of the reference's ``imagen_pytorch3D`` module (Family A):
true-Conv3d U-Net (``Unet``/``SRUnet256``/``BaseUnet64``/``NullUnet``) and the continuous-time DDPM
wrapper ``Imagen`` — same constructor kwargs, ``forward``/``sample`` signatures, return conventions
and ``state_dict`` keys/shapes (OIDHW conv weights) as the reference, so its ``train.py`` /
``test_all.py`` drive it unchanged.

Inside the U-Net activations are channels-last fp32 ``[B, D, H, W, C]`` and every operator is a HIP
kernel from ``csrc/`` (via ``ops``); public tensors stay ``[B, C, D, H, W]``.  Host-side scalars
(noise schedule, per-step posterior coefficients) are computed in Python/torch-CPU on ``[B]``-sized
vectors and uploaded once — they are control data, not the hot path.

Reference citations are ``file:line`` relative to the reference root.
"""
import math
from contextlib import contextmanager, nullcontext
from functools import partial
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import ops
from .ops import ACT_MISH, ACT_GELU

# ----------------------------------------------------------------------------------------------
# helpers (imagen_pytorch3D.py:48-128)
# ----------------------------------------------------------------------------------------------


def exists(val):
    return val is not None


def default(val, d):
    if exists(val):
        return val
    return d() if callable(d) else d


def cast_tuple(val, length=None):
    if isinstance(val, list):
        val = tuple(val)
    output = val if isinstance(val, tuple) else ((val,) * default(length, 1))
    if exists(length):
        assert len(output) == length
    return output


def pad_tuple_to_length(t, length, fillvalue=None):
    remain = length - len(t)
    return t if remain <= 0 else (*t, *((fillvalue,) * remain))


def identity(t, *args, **kwargs):
    return t


def maybe(fn):
    def inner(x):
        return x if not exists(x) else fn(x)
    return inner


def right_pad_dims_to(x, t):
    pad = x.ndim - t.ndim
    return t if pad <= 0 else t.view(*t.shape, *((1,) * pad))


def normalize_neg_one_to_one(img):
    return img * 2 - 1


def unnormalize_zero_to_one(img):
    return (img + 1) * 0.5


def eval_decorator(fn):
    def inner(model, *args, **kwargs):
        was_training = model.training
        model.eval()
        out = fn(model, *args, **kwargs)
        model.train(was_training)
        return out
    return inner


def to_channels_last(x):
    """[B,C,D,H,W] -> contiguous [B,D,H,W,C] (a free view when C == 1)."""
    B, C = x.shape[:2]
    if C == 1:
        return x.contiguous().view(B, *x.shape[2:], 1)
    return x.permute(0, 2, 3, 4, 1).contiguous()


def to_channels_first(x):
    B, C = x.shape[0], x.shape[-1]
    if C == 1:
        return x.contiguous().view(B, 1, *x.shape[1:-1])
    return x.permute(0, 4, 1, 2, 3).contiguous()


# ----------------------------------------------------------------------------------------------
# continuous-time Gaussian diffusion (imagen_pytorch3D.py:225-357) — host-side [B] scalars
# ----------------------------------------------------------------------------------------------
def log(t, eps=1e-12):
    return torch.log(t.clamp(min=eps))


def beta_linear_log_snr(t):
    return -torch.log(torch.special.expm1(1e-4 + 10 * (t ** 2)))


def alpha_cosine_log_snr(t, s: float = 0.008):
    return -log((torch.cos((t + s) / (1 + s) * math.pi * 0.5) ** -2) - 1, eps=1e-5)


def log_snr_to_alpha_sigma(log_snr):
    return torch.sqrt(torch.sigmoid(log_snr)), torch.sqrt(torch.sigmoid(-log_snr))


class GaussianDiffusionContinuousTimes(nn.Module):
    """Noise schedule + posterior coefficients.  ``q_sample`` / the sampler step run as HIP kernels
    (ops.q_sample, ops.ddpm_step); this class produces their per-batch coefficients on the host."""

    def __init__(self, *, noise_schedule, timesteps=1000):
        super().__init__()
        if noise_schedule == "linear":
            self.log_snr = beta_linear_log_snr
        elif noise_schedule == "cosine":
            self.log_snr = alpha_cosine_log_snr
        else:
            raise ValueError(f'invalid noise schedule {noise_schedule}')
        self.num_timesteps = timesteps

    def get_times(self, batch_size, noise_level, *, device):
        return torch.full((batch_size,), noise_level, device=device, dtype=torch.float32)

    def sample_random_times(self, batch_size, *, device):
        # drawn on the CPU generator like the reference (:253-256)
        return torch.zeros((batch_size,)).float().uniform_(0, 1).to(device)

    def get_condition(self, times):
        return maybe(self.log_snr)(times)

    def get_sampling_timesteps(self, batch, *, device):
        times = torch.linspace(1., 0., self.num_timesteps + 1, device=device)
        times = times[None, :].repeat(batch, 1)
        times = torch.stack((times[:, :-1], times[:, 1:]), dim=0)
        return times.unbind(dim=-1)

    def posterior_coefficients(self, t, t_next):
        """q_posterior (:290-309) folded into x_next = ca*x_t + cb*x0 + cn*noise for [B] CPU tensors."""
        log_snr, log_snr_next = self.log_snr(t), self.log_snr(t_next)
        alpha, _ = log_snr_to_alpha_sigma(log_snr)
        alpha_next, sigma_next = log_snr_to_alpha_sigma(log_snr_next)
        c = -torch.special.expm1(log_snr - log_snr_next)
        var = (sigma_next ** 2) * c
        logvar = log(var, eps=1e-20)
        nonzero = 1 - (t_next == 0).float()
        return alpha_next * (1 - c) / alpha, alpha_next * c, nonzero * (0.5 * logvar).exp()

    def q_posterior(self, x_start, x_t, t, *, t_next=None):
        t_next = default(t_next, lambda: (t - 1. / self.num_timesteps).clamp(min=0.))
        log_snr, log_snr_next = (right_pad_dims_to(x_t, self.log_snr(u)) for u in (t, t_next))
        alpha, sigma = log_snr_to_alpha_sigma(log_snr)
        alpha_next, sigma_next = log_snr_to_alpha_sigma(log_snr_next)
        c = -torch.special.expm1(log_snr - log_snr_next)
        mean = alpha_next * (x_t * (1 - c) / alpha + c * x_start)
        var = (sigma_next ** 2) * c
        return mean, var, log(var, eps=1e-20)

    def q_sample(self, x_start, t, noise=None):
        if isinstance(t, float):
            t = torch.full((x_start.shape[0],), t, dtype=x_start.dtype)
        noise = default(noise, lambda: torch.randn_like(x_start))
        log_snr = self.log_snr(t.detach().cpu().float())
        alpha, sigma = log_snr_to_alpha_sigma(log_snr)
        dev = x_start.device
        x_t = ops.q_sample(x_start.contiguous(), noise.contiguous(), alpha.to(dev), sigma.to(dev))
        return x_t, log_snr.to(dev), right_pad_dims_to(x_start, alpha.to(dev)), right_pad_dims_to(x_start, sigma.to(dev))


# ----------------------------------------------------------------------------------------------
# layers (channels-last, HIP kernels underneath; parameters keep the reference names/shapes)
# ----------------------------------------------------------------------------------------------
class Conv3d(nn.Conv3d):
    """nn.Conv3d parameters (OIDHW) + the MFMA implicit-GEMM kernel; grouped/strided -> direct kernel."""

    def forward(self, x, residual=None, want_stats=False):
        if self.groups != 1 or tuple(self.stride) != (1, 1, 1):
            y = ops.conv3d_direct(x, self.weight, self.bias, self.stride, self.padding, self.groups)
            return y if residual is None else ops.add(y, residual)
        return ops.conv3d(x, self.weight, self.bias, self.padding, residual, want_stats=want_stats)


class Linear(nn.Linear):
    def forward(self, x):
        return ops.linear(x, self.weight, self.bias)


class Act(nn.Module):
    def __init__(self, act):
        super().__init__()
        self.act = act

    def forward(self, x):
        return ops.activation(x, self.act)


def Mish():
    return Act(ops.ACT_MISH)


class Identity(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()

    def forward(self, x, *args, **kwargs):
        return x


class LayerNorm(nn.Module):
    """ChanLayerNorm — imagen_pytorch3D.py:361-382: gain-only LN over channels, g is [C,1,1,1]."""

    def __init__(self, feats, stable=False, dim=-1):
        super().__init__()
        assert not stable, 'stable LayerNorm is not on the hot path'
        self.g = nn.Parameter(torch.ones(feats, *((1,) * (-dim - 1))))

    def forward(self, x):
        return ops.chan_layernorm(x, self.g, 1e-5)


ChanLayerNorm = partial(LayerNorm, dim=-4)


class SpaceToDepth(nn.Module):
    """Rearrange 'b c (h s1) (w s2) (d s3) -> b (c s1 s2 s3) h w d' (:494); parameter-free."""

    def forward(self, x):
        return ops.space_to_depth(x)


class PixelShuffle3D(nn.Module):
    def __init__(self, scale):
        super().__init__()
        assert scale == 2
        self.scale = scale

    def forward(self, x):
        return ops.depth_to_space(x)


class PixelShuffleUpsample(nn.Module):
    """imagen_pytorch3D.py:459-487: 1x1 conv (C -> 8C') -> Mish -> depth-to-space."""

    def __init__(self, dim, dim_out=None):
        super().__init__()
        dim_out = default(dim_out, dim)
        conv = Conv3d(dim, dim_out * 8, 1)
        self.net = nn.Sequential(conv, Mish(), PixelShuffle3D(2))
        self.init_conv_(conv)

    def init_conv_(self, conv):
        o, i, h, w, d = conv.weight.shape
        conv_weight = torch.empty(o // 4, i, h, w, d)
        nn.init.kaiming_uniform_(conv_weight)
        conv_weight = conv_weight.repeat_interleave(4, dim=0)        # 'o ... -> (o 4) ...'
        conv.weight.data.copy_(conv_weight)
        nn.init.zeros_(conv.bias.data)

    def forward(self, x):
        return self.net(x)


def Downsample(dim, dim_out=None):
    dim_out = default(dim_out, dim)
    return nn.Sequential(SpaceToDepth(), Conv3d(dim * 8, dim_out, 1))


class LearnedSinusoidalPosEmb(nn.Module):
    def __init__(self, dim):
        super().__init__()
        assert (dim % 2) == 0
        self.weights = nn.Parameter(torch.randn(dim // 2))

    def forward(self, x):
        return ops.learned_sinusoidal(x, self.weights)


def boundary_pad(x, batch_sample_factor=3):
    """imagen_pytorch3D.py:37-46 on channels-last sub-volume batches: merge -> zero halo -> overlapping blocks."""
    A = x.shape[1]
    vol = ops.merge_volume(x, batch_sample_factor)
    return ops.split_volume(vol, batch_sample_factor, A, halo=1)


class Block(nn.Module):
    """GN(8) -> x*(scale+1)+shift -> Mish -> Conv3d 3^3 (imagen_pytorch3D.py:535-566); the first three are one kernel."""

    def __init__(self, dim, dim_out, groups=8, norm=True, boundary=False, factor=3):
        super().__init__()
        self.groupnorm = nn.GroupNorm(groups, dim) if norm else Identity()
        self.activation = Mish()
        self.boundary = boundary
        self.factor = factor
        self.project = Conv3d(dim, dim_out, 3) if boundary else Conv3d(dim, dim_out, 3, padding=1)

    def forward(self, x, scale_shift=None, residual=None, emit_stats=False, tap=False, out_half=False):
        """``emit_stats``: the conv epilogue also writes per-tile column sums of its output for the next GroupNorm / SE pool.
        ``tap``: also returns an alias of the input for its other consumer (ops.groupnorm_act): ``(y, x_alias)``.
        ``out_half`` (sampling under autocast): y only feeds the next Block's GroupNorm and may be stored in the operand type."""
        gn = self.groupnorm
        x_in = x
        if isinstance(gn, nn.GroupNorm) and not self.boundary and not torch.is_grad_enabled():
            # sampling: GroupNorm-apply + Mish inside the conv's input staging (one launch, no elementwise pass); None: shape not taken
            pr = self.project
            y = None
            if pr.groups == 1 and tuple(pr.stride) == (1, 1, 1):
                y = ops.gn_conv3d(x, gn.weight, gn.bias, scale_shift, gn.num_groups, ACT_MISH, gn.eps, pr.weight, pr.bias, pr.padding,
                                  residual, want_stats=emit_stats)
                if y is None:       # under autocast: GroupNorm-apply writes the operand type, the conv runs on the LDS-DMA 16-bit kernel
                    y = ops.gn_conv3d_h(x, gn.weight, gn.bias, scale_shift, gn.num_groups, ACT_MISH, gn.eps, pr.weight, pr.bias,
                                        pr.padding, residual, want_stats=emit_stats, out_half=out_half)
            if y is not None:
                return (y, x_in) if tap else y
        if isinstance(gn, nn.GroupNorm) and not self.boundary and torch.is_grad_enabled():
            # bf16 training: GroupNorm-apply + conv as one autograd node whose intermediate exists only in bf16 (None: not that mode / shape)
            pr = self.project
            if pr.groups == 1 and tuple(pr.stride) == (1, 1, 1):
                out = ops.gn_conv3d_train_h(x, gn.weight, gn.bias, scale_shift, gn.num_groups, ACT_MISH, gn.eps, pr.weight, pr.bias, pr.padding,
                                            residual, want_stats=emit_stats, tap=tap, out_half=out_half)
                if out is not None:
                    return out
        if isinstance(gn, nn.GroupNorm):
            x = ops.groupnorm_act(x, gn.weight, gn.bias, scale_shift, gn.num_groups, ACT_MISH, gn.eps, tap=tap)
            if tap:
                x, x_in = x
        else:
            assert scale_shift is None
            x = self.activation(x)
        if tap:
            return self._project(x, residual, emit_stats), x_in
        return self._project(x, residual, emit_stats)

    def _project(self, x, residual, emit_stats):
        if self.boundary:
            if not torch.is_grad_enabled():
                # sampling: the conv reads each sub-volume's halo straight from its neighbours (no merged / re-split copies)
                y = ops.conv3d_neighbours(x, self.project.weight, self.project.bias, self.factor, residual, emit_stats)
                if y is not None:
                    return y
            x = boundary_pad(x, self.factor)
        return self.project(x, residual=residual, want_stats=emit_stats)


class SE3D(nn.Module):
    """imagen_pytorch3D.py:617-632 (parameters only; ResnetBlock runs the fused pool->MLP->gate+residual op)."""

    def __init__(self, channel, reduction=16):
        super().__init__()
        self.fc = nn.Sequential(Linear(channel, channel // reduction, bias=False), Act(ops.ACT_RELU),
                                Linear(channel // reduction, channel, bias=False), Act(ops.ACT_SIGMOID))

    def forward(self, x, residual=None):
        return ops.se_gate_residual(x, self.fc[0].weight, self.fc[2].weight, residual)


class TimeCond:
    """The time embedding handed to every ResnetBlock of one U-Net evaluation.  Each block's time_mlp starts with the
    same Mish(t) (imagen_pytorch3D.py:575-578), so it is evaluated once here and shared; on the sampling path the Linear
    layers of ALL blocks then run as one launch over their concatenated weights (``batched``: id(linear) -> SSView)."""
    __slots__ = ("t", "_act", "batched", "act")

    def __init__(self, t, act=ops.ACT_MISH):
        self.t, self._act, self.batched, self.act = t, None, None, act      # act: Mish here, SiLU in the pseudo-3D U-Net

    def activated(self):
        if self._act is None:
            self._act = ops.activation(self.t, self.act)
        return self._act

    mish = activated


class BatchedTimeMLPs:
    """Concatenation of the time-MLP Linear layers of a U-Net's ResnetBlocks (reference: one ``time_mlp`` call per block,
    imagen_pytorch3D.py:586-589 -- ~20 launches of a [B, 256] x [256, 2C] product per eval): weights / biases are packed into one
    [sum 2C_i, 256] matrix, rebuilt whenever a parameter changed, and applied with one skinny-linear launch."""

    def __init__(self, linears):
        self.linears = list(linears)
        self.key, self.w, self.b, self.offs = None, None, None, None

    def __call__(self, tc):
        if not self.linears or tc.t.shape[0] > 64:
            return None
        key = tuple((l.weight.data_ptr(), l.weight._version, l.bias._version if l.bias is not None else -1) for l in self.linears) \
            + (ops._WEIGHT_EPOCH,)
        if key != self.key:
            ops.retire(self.w, self.b)
            with torch.no_grad():
                self.w = ops.born(torch.cat([l.weight for l in self.linears], dim=0).contiguous())
                self.b = ops.born(torch.cat([l.bias if l.bias is not None else torch.zeros(l.weight.shape[0], device=l.weight.device)
                                             for l in self.linears]).contiguous())
            self.offs, off = {}, 0
            for l in self.linears:
                self.offs[id(l)] = (off, l.weight.shape[0])
                off += l.weight.shape[0]
            self.key = key
        if torch.is_grad_enabled():
            # training: the same single launch with autograd (one backward launch set for all blocks, ops._BatchedLinearSmallFn)
            if not BatchedTimeMLPs.train_batched:
                return None
            outs = ops.batched_linear_small(tc.activated(), self.w, self.b, self.linears)
            return {id(l): o for l, o in zip(self.linears, outs)}
        out = ops.linear(tc.activated(), self.w, self.b)                              # ONE launch: [B, sum 2C_i]
        return {k: ops.SSView(out, off, n) for k, (off, n) in self.offs.items()}

    train_batched = True       # False: per-block time MLPs when autograd records (A/B and tests)


class ResnetBlock(nn.Module):
    """imagen_pytorch3D.py:568-614."""

    def __init__(self, dim, dim_out, time_cond_dim=None, groups=8, use_se=False, boundary=False, factor=3):
        super().__init__()
        self.time_mlp = None
        if exists(time_cond_dim):
            self.time_mlp = nn.Sequential(Mish(), Linear(time_cond_dim, dim_out * 2))
        self.block1 = Block(dim, dim_out, groups=groups, boundary=boundary, factor=factor)
        self.block2 = Block(dim_out, dim_out, groups=groups, boundary=boundary, factor=factor)
        self.se = SE3D(dim_out, reduction=16) if use_se else Identity()
        self.res_conv = Conv3d(dim, dim_out, 1) if dim != dim_out else Identity()

    def forward(self, x, time_emb=None):
        scale_shift = None
        if exists(self.time_mlp) and exists(time_emb):      # [B, 2C]: scale | shift
            if isinstance(time_emb, TimeCond):
                pre = time_emb.batched.get(id(self.time_mlp[1])) if time_emb.batched is not None else None
                scale_shift = pre if pre is not None else self.time_mlp[1](time_emb.mish())
            else:
                scale_shift = self.time_mlp(time_emb)
        # block2's GroupNorm statistics come from block1's conv epilogue; the residual branch reads x through block1's alias so that
        # its gradient is added inside the GroupNorm backward (ops.groupnorm_act, tap)
        # (sampling under autocast: h only feeds block2's GroupNorm -- 16-bit when block2's conv takes a 16-bit input)
        b2 = self.block2.project
        oh = (not self.block2.boundary and isinstance(self.block2.groupnorm, nn.GroupNorm) and b2.groups == 1 and tuple(b2.stride) == (1, 1, 1)
              and x.dim() == 5
              and (ops.conv_half_out_ok((*x.shape[:4], b2.weight.shape[1]), b2.weight, b2.padding)
                   or ops.train_half_out_ok((*x.shape[:4], b2.weight.shape[1]), b2.weight, b2.padding, self.block2.groupnorm.num_groups)))
        h, x = self.block1(x, emit_stats=True, tap=True, out_half=oh)
        res = self.res_conv(x)
        if isinstance(self.se, SE3D):
            # ... and the SE pooling from this one's.  Under autocast sampling / low-precision training block2's output only meets the SE
            # gate (statistics from the column sums, one elementwise pass): it may leave in the operand type (gn_conv3d_h / gn_conv3d_train_h
            # decide whether the shape allows it)
            h = self.block2(h, scale_shift=scale_shift, emit_stats=True, out_half=ops.lp_mode() is not None and not ops._NO_TRAIN_HALF)
            return self.se(h, residual=res)
        return self.block2(h, scale_shift=scale_shift, residual=res)   # residual add fused in the conv epilogue


class CrossEmbedLayer(nn.Module):
    """imagen_pytorch3D.py:661-686."""

    def __init__(self, dim_in, kernel_sizes, dim_out=None, stride=2):
        super().__init__()
        assert all([(t % 2) == (stride % 2) for t in kernel_sizes])
        dim_out = default(dim_out, dim_in)
        kernel_sizes = sorted(kernel_sizes)
        num_scales = len(kernel_sizes)
        dim_scales = [int(dim_out / (2 ** i)) for i in range(1, num_scales)]
        dim_scales = [*dim_scales, dim_out - sum(dim_scales)]
        self.convs = nn.ModuleList([Conv3d(dim_in, ds, k, stride=stride, padding=(k - stride) // 2)
                                    for k, ds in zip(kernel_sizes, dim_scales)])

    def forward(self, x):
        out = None
        for conv in self.convs:
            y = conv(x)
            out = y if out is None else ops.concat_channels(out, y)
        return out


class depthwise_separable_conv3d(nn.Module):
    """imagen_pytorch3D.py:858-869."""

    def __init__(self, input_dim, output_dim, kernel_size, stride, padding=0):
        super().__init__()
        self.depthwise = Conv3d(input_dim, input_dim, kernel_size=kernel_size, stride=stride, padding=padding,
                                groups=input_dim)
        self.pointwise = Conv3d(input_dim, output_dim, kernel_size=1)

    def forward(self, x):
        return self.pointwise(self.depthwise(x))


class Patchify(nn.Module):
    """imagen_pytorch3D.py:913-924."""

    def __init__(self, in_channels=3, patch_size=4, emb_size=128, img_size=224, reduction=False):
        super().__init__()
        self.patch_size = patch_size
        self.norm = ChanLayerNorm(in_channels)
        self.projection = depthwise_separable_conv3d(in_channels, emb_size, kernel_size=patch_size, stride=patch_size)

    def forward(self, x):
        return self.projection(self.norm(x))


class Upsample(nn.Module):
    """nn.Upsample(scale_factor, 'trilinear', align_corners=True) (:954) — parameter-free."""

    def __init__(self, scale_factor):
        super().__init__()
        self.scale_factor = scale_factor

    def forward(self, x):
        return ops.trilinear_upsample(x, self.scale_factor)


class _TokenAttention(nn.Module):
    """Shared body of LinearAttention (:926-1016) and SoftMaxAttention (:1018-1106).
    Heads are addressed by strides inside the channels-last [tokens, heads*dim_head] maps, so no
    transposing copies are made; the contractions run on the MFMA bgemm kernel."""
    linear = True

    def __init__(self, dim, dim_head=32, heads=8, dropout=0.05, context_dim=None, patch_size=2, img_size=48,
                 patch=False, groups=1, **kwargs):
        super().__init__()
        assert not exists(context_dim), 'context conditioning is not used by the IQT path'
        self.patch, self.patch_size = patch, patch_size
        self.scale = dim_head ** -0.5
        self.heads, self.dim_head = heads, dim_head
        inner_dim = dim_head * heads
        self.norm = ChanLayerNorm(dim)
        self.nonlin = Mish()
        if self.patch:
            self.patch_embed = Patchify(in_channels=dim, patch_size=patch_size, emb_size=dim, img_size=img_size)
            self.reconstruct = nn.Sequential(Upsample(patch_size),
                                             depthwise_separable_conv3d(dim, dim, kernel_size=3, stride=1, padding=1),
                                             ChanLayerNorm(dim))

        def proj():
            return nn.Sequential(nn.Dropout(dropout), Conv3d(dim, inner_dim, 1, bias=False),
                                 Conv3d(inner_dim, inner_dim, 3, bias=False, padding=1, groups=inner_dim))
        self.to_q, self.to_k, self.to_v = proj(), proj(), proj()
        self.to_context = None
        self.to_out = nn.Sequential(Conv3d(inner_dim, dim, 1, bias=False), ChanLayerNorm(dim))

    def forward(self, fmap, context=None):
        assert context is None
        if self.patch:
            fmap = self.patch_embed(fmap)
        b, X, Y, Z, _ = fmap.shape
        n, h, d = X * Y * Z, self.heads, self.dim_head
        fmap = self.norm(fmap)
        q, k, v = self.to_q(fmap), self.to_k(fmap), self.to_v(fmap)       # [b, X,Y,Z, h*d]
        outs = []
        for bi in range(b):                                                 # b == 1 on the IQT path (merged volume)
            qb, kb, vb = (t[bi].reshape(n, h * d) for t in (q, k, v))
            if self.linear:
                qs = ops.softmax(qb.reshape(n * h, d), dim=-1, scale=self.scale).reshape(n, h * d)   # :1003,1006
                ks = ops.softmax(kb, dim=0)                                                           # :1004
                # ctx[h] = ks[:,h,:]^T vb[:,h,:]  -> [h, d, d]                                         :1008
                ctx = ops.bmm_strided(ks, vb, (h, d, d, n, True, False, d, h * d, d, h * d, d * d, d, 1.0, (h, d, d)))
                # out[:,h,:] = qs[:,h,:] ctx[h]  -> written head-interleaved [n, h*d]                   :1009
                ob = ops.bmm_strided(qs, ctx, (h, n, d, d, False, False, d, h * d, d * d, d, d, h * d, 1.0, (n, h * d)))
            else:
                # energy[h] = q k^T * scale -> [h, n, n]                                                :1088
                en = ops.bmm_strided(qb, kb, (h, n, n, d, False, True, d, h * d, d, h * d, n * n, n, self.scale,
                                              (h, n, n)))
                att = ops.softmax(en, dim=-1)
                ob = ops.bmm_strided(att, vb, (h, n, d, n, False, False, n * n, n, d, h * d, d, h * d, 1.0, (n, h * d)))
            outs.append(ob.reshape(1, X, Y, Z, h * d))
        out = outs[0] if b == 1 else torch.cat(outs, dim=0)
        out = self.to_out(self.nonlin(out))
        if self.patch:
            out = self.reconstruct(out)
        return out


class LinearAttention(_TokenAttention):
    linear = True


class SoftMaxAttention(_TokenAttention):
    linear = False


# ---- ViT3D (imagen_pytorch3D.py:723-856, 871-910) -----------------------------------------------------------------
class TokenLayerNorm(nn.LayerNorm):
    """nn.LayerNorm(emb) over the last (channel) axis of [b, tokens..., emb]: the HIP channel-LayerNorm with a bias."""

    def forward(self, x):
        return ops.chan_layernorm(x, self.weight, self.eps, bias=self.bias)


class ResidualAdd(nn.Module):
    def __init__(self, fn):
        super().__init__()
        self.fn = fn

    def forward(self, x, **kwargs):
        return ops.add(self.fn(x, **kwargs), x)


class _ToVolume(nn.Module):
    """Rearrange 'b (h w d) c -> b c h w d' (:778): a free view in channels-last memory."""

    def __init__(self, n):
        super().__init__()
        self.n = n

    def forward(self, x):
        return x.reshape(x.shape[0], self.n, self.n, self.n, x.shape[-1])


class _ToTokens(nn.Module):
    """Rearrange 'b c h w d -> b (h w d) c' (:791): a free view in channels-last memory."""

    def forward(self, x):
        return x.reshape(x.shape[0], -1, x.shape[-1])


class FeedForwardBlock(nn.Sequential):
    """:774-809 — registers the sub-modules under both their own names and ``net`` like the reference (shared tensors,
    duplicated state_dict keys)."""

    def __init__(self, emb_size, expansion=4, drop_p=0., patch_num=4, local=False):
        super().__init__()
        if local:
            self.up_proj = nn.Sequential(_ToVolume(patch_num), Conv3d(emb_size, emb_size * expansion, kernel_size=1), Mish())
            self.depth_conv = nn.Sequential(depthwise_separable_conv3d(emb_size * expansion, emb_size * expansion,
                                                                       kernel_size=3, stride=1, padding=1), Mish())
            self.down_proj = nn.Sequential(Conv3d(emb_size * expansion, emb_size, kernel_size=1), nn.Dropout(drop_p), _ToTokens())
            self.net = nn.Sequential(self.up_proj, self.depth_conv, self.down_proj)
        else:
            self.net = nn.Sequential(Linear(emb_size, expansion * emb_size), Mish(), nn.Dropout(drop_p),
                                     Linear(expansion * emb_size, emb_size))

    def forward(self, x):
        return self.net(x)


class MultiHeadAttention(nn.Module):
    """:811-838.  qkv channels are laid out '(h d qkv)'; one transpose kernel de-interleaves them into [rows, 3, h*d], after
    which q, k, v and the heads are addressed by offsets / strides inside that tensor (no further copies)."""

    def __init__(self, emb_size=128, num_heads=8, dim_head=64, dropout=0):
        super().__init__()
        self.emb_size, self.dim_head, self.num_heads = emb_size, dim_head, num_heads
        self.inner_dim = dim_head * num_heads
        self.qkv = Linear(emb_size, self.inner_dim * 3)
        self.att_drop = nn.Dropout(dropout)
        self.projection = Linear(self.inner_dim, emb_size)
        self.scaling = dim_head ** -0.5

    def forward(self, x, mask=None):
        assert mask is None, "the reference's mask branch is a no-op (`mask_fill` result discarded, :829)"
        b, n, _ = x.shape
        h, d = self.num_heads, self.dim_head
        hd = h * d
        t = ops.transpose_mid(self.qkv(x).reshape(b * n, hd, 3, 1)).reshape(b * n, 3 * hd)     # rows: [q | k | v]
        outs = []
        for bi in range(b):
            base = bi * n * 3 * hd
            en = ops.bmm_strided(t, t, (h, n, n, d, False, True, d, 3 * hd, d, 3 * hd, n * n, n, self.scaling, (h, n, n),
                                        base, base + hd))
            att = self.att_drop(ops.softmax(en, dim=-1))
            outs.append(ops.bmm_strided(att, t, (h, n, d, n, False, False, n * n, n, d, 3 * hd, d, hd, 1.0, (1, n, hd),
                                                 0, base + 2 * hd)))
        out = outs[0] if b == 1 else torch.cat(outs, dim=0)
        return self.projection(out)


class TransformerEncoderBlock(nn.Module):
    def __init__(self, emb_size=256, num_heads=8, dim_head=64, drop_p=0., forward_expansion=4, forward_drop_p=0.,
                 patch_num=4, local=True):
        super().__init__()
        self.block = nn.Sequential(
            ResidualAdd(nn.Sequential(TokenLayerNorm(emb_size),
                                      MultiHeadAttention(emb_size, num_heads=num_heads, dropout=drop_p, dim_head=dim_head),
                                      nn.Dropout(drop_p))),
            ResidualAdd(nn.Sequential(TokenLayerNorm(emb_size),
                                      FeedForwardBlock(emb_size, expansion=forward_expansion, drop_p=forward_drop_p,
                                                       patch_num=patch_num, local=local),
                                      nn.Dropout(drop_p))))

    def forward(self, x):
        return self.block(x)


class TransformerEncoder(nn.Module):
    def __init__(self, depth=12, **kwargs):
        super().__init__()
        self.layers = nn.ModuleList([TransformerEncoderBlock(**kwargs) for _ in range(depth)])

    def forward(self, x):
        for layer in self.layers:
            x = layer(x)
        return x


class PatchEmbedding(nn.Module):
    """:841-856"""

    def __init__(self, in_channels=3, patch_size=4, emb_size=128, img_size=224, reduction=False):
        super().__init__()
        self.patch_size = patch_size
        self.projection = nn.Sequential(depthwise_separable_conv3d(in_channels, emb_size, kernel_size=patch_size,
                                                                   stride=patch_size), _ToTokens())
        self.positions = nn.Parameter(torch.randn((img_size // patch_size) ** 3, emb_size))

    def forward(self, x):
        x = self.projection(x)
        return ops.add(x, self.positions.unsqueeze(0).expand_as(x).contiguous())


class ViT3D(nn.Module):
    """:871-910 — operates on channels-last volumes [b, X, Y, Z, C] like the other attention blocks."""

    def __init__(self, in_channels=3, patch_size=16, num_heads=8, dim_head=64, img_size=224, depth=1, drop_p=0.1,
                 forward_drop_p=0.3, forward_expansion=2, reduction=False, local=True, groups=1, **kwargs):
        super().__init__()
        self.reduction = reduction
        self.emb_size = in_channels
        n = img_size // patch_size
        self.patch_embedding = PatchEmbedding(in_channels, patch_size, self.emb_size, img_size, reduction=reduction)
        self.transformer_encoder = TransformerEncoder(depth, emb_size=self.emb_size, num_heads=num_heads, dim_head=dim_head,
                                                      patch_num=n, drop_p=drop_p, forward_drop_p=forward_drop_p,
                                                      forward_expansion=forward_expansion, local=local, **kwargs)
        self.reconstruction = nn.Sequential(TokenLayerNorm(in_channels), _ToVolume(n), Upsample(patch_size),
                                            depthwise_separable_conv3d(in_channels, in_channels, kernel_size=3, stride=1,
                                                                       padding=1),
                                            ChanLayerNorm(in_channels))

    def forward(self, x):
        return self.reconstruction(self.transformer_encoder(self.patch_embedding(x)))


class _ChanFeedForward(nn.Sequential):
    """ChanFeedForward (:1108-1116): ChanLN -> 1x1 -> GELU -> ChanLN -> 1x1."""

    def __init__(self, dim, mult=2):
        hidden = int(dim * mult)
        super().__init__(ChanLayerNorm(dim), Conv3d(dim, hidden, 1, bias=False), Act(ACT_GELU), ChanLayerNorm(hidden),
                         Conv3d(hidden, dim, 1, bias=False))


def ChanFeedForward(dim, mult=2):
    return _ChanFeedForward(dim, mult)


class _AttentionTransformerBlock(nn.Module):
    attn_klass = LinearAttention

    def __init__(self, dim, *, depth=1, heads=8, dim_head=32, ff_mult=2, context_dim=None, patch_size=2, img_size=48,
                 patch=False, groups=1, **kwargs):
        super().__init__()
        self.layers = nn.ModuleList([])
        self.patch, self.img_size = patch, img_size
        for _ in range(depth):
            self.layers.append(nn.ModuleList([
                self.attn_klass(dim=dim, heads=heads, dim_head=dim_head, context_dim=context_dim, patch_size=patch_size,
                                img_size=img_size, patch=patch, groups=groups),
                ChanFeedForward(dim=dim, mult=ff_mult)]))

    def forward(self, x, context=None):
        for attn, ff in self.layers:
            x = ops.add(attn(x, context=context), x)
            x = ops.add(ff(x), x)
        return x


class LinearAttentionTransformerBlock(_AttentionTransformerBlock):
    attn_klass = LinearAttention


class SoftMaxAttentionTransformerBlock(_AttentionTransformerBlock):
    attn_klass = SoftMaxAttention


# ----------------------------------------------------------------------------------------------
# U-Net (imagen_pytorch3D.py:1188-1684)
# ----------------------------------------------------------------------------------------------
_print_once_done = set()


def print_once(msg):
    if msg not in _print_once_done:
        _print_once_done.add(msg)
        print(msg)


class Unet(nn.Module):
    def __init__(
        self, *, dim, img_size=96, num_resnet_blocks=1, cond_dim=None, learned_sinu_pos_emb_dim=16,
        dim_mults=(1, 2, 4, 8), cond_images_channels=0, channels=3, channels_out=None, attn_dim_head=64,
        attn_heads=8, ff_mult=2., lowres_cond=False, att_type='vit', attend_at_middle=True,
        attend_at_middle_depth=1, attend_at_middle_heads=8, attend_at_enc=True, attend_at_enc_depth=1,
        attend_at_enc_heads=8, att_drop=0.1, att_forward_drop=0.3, att_forward_expansion=2, att_skip_scale=False,
        att_localvit=True, groups=1, emb_size=768, init_dim=32, resnet_groups=8, init_conv_kernel_size=3,
        init_cross_embed=True, init_cross_embed_kernel_sizes=(3, 7, 15), cross_embed_downsample=False,
        cross_embed_downsample_kernel_sizes=(2, 4), memory_efficient=False, init_conv_to_final_conv_residual=False,
        use_se_attn=True, scale_skip_connection=False, final_resnet_block=True, final_conv_kernel_size=1,
        self_cond=False, combine_upsample_fmaps=False, pixel_shuffle_upsample=True, boundary=False,
        batch_sample=True, batch_sample_factor=3, deep_feature=True,
    ):
        super().__init__()
        self._locals = {k: v for k, v in locals().items() if k not in ('self', '__class__')}
        self.att_type, self.dim_head = att_type, attn_dim_head
        self.batch_sample, self.batch_sample_factor = batch_sample, batch_sample_factor
        self.img_size, self.boundary, self.num_groups, self.deep_feature = img_size, boundary, groups, deep_feature
        assert attn_heads > 1, 'you need to have more than 1 attention head, ideally at least 4 or 8'
        assert not init_conv_to_final_conv_residual and not cross_embed_downsample and pixel_shuffle_upsample, \
            'option outside the IQT hot path (SURVEY.md §8): not built'
        if dim < 128:
            print_once('The base dimension of your u-net should ideally be no smaller than 128, as recommended by a '
                       'professional DDPM trainer https://nonint.com/2022/05/04/friends-dont-let-friends-train-small-diffusion-models/')

        self.channels = channels
        self.channels_out = default(channels_out, channels)
        init_channels = channels * (1 + int(lowres_cond))
        init_dim = default(init_dim, dim)
        self.self_cond = self_cond
        if self_cond:
            init_channels += channels
        self.has_cond_image = cond_images_channels > 0
        self.cond_images_channels = cond_images_channels
        init_channels += cond_images_channels

        if init_cross_embed:
            self.init_conv = CrossEmbedLayer(init_channels, dim_out=init_dim, kernel_sizes=init_cross_embed_kernel_sizes,
                                             stride=1)
        elif boundary:
            self.init_conv = Conv3d(init_channels, init_dim, init_conv_kernel_size)
        else:
            self.init_conv = Conv3d(init_channels, init_dim, init_conv_kernel_size, padding=init_conv_kernel_size // 2)

        dims = [init_dim, *map(lambda m: dim * m, dim_mults)]
        in_out = list(zip(dims[:-1], dims[1:]))
        self.dims, self.in_out = dims, in_out

        cond_dim = default(cond_dim, dim)
        time_cond_dim = dim * 4
        self.to_time_hiddens = nn.Sequential(LearnedSinusoidalPosEmb(learned_sinu_pos_emb_dim),
                                             Linear(learned_sinu_pos_emb_dim + 1, time_cond_dim), Mish())
        self.to_time_cond = nn.Sequential(Linear(time_cond_dim, time_cond_dim))
        self.lowres_cond = lowres_cond
        self.norm_cond = nn.LayerNorm(cond_dim)          # constructed and never used, like the reference (:1322)
        self.text_to_cond = None

        num_layers = len(in_out)
        num_resnet_blocks = cast_tuple(num_resnet_blocks, num_layers)
        resnet_groups = cast_tuple(resnet_groups, num_layers)
        attend_at_enc = cast_tuple(attend_at_enc, num_layers)
        attend_at_enc_depth = cast_tuple(attend_at_enc_depth, num_layers)
        attend_at_enc_heads = cast_tuple(attend_at_enc_heads, num_layers)
        self.skip_connect_scale = 1. if not scale_skip_connection else (2 ** -0.5)

        def make_attn(d, depth, heads, patch_size, size):
            if att_type == 'linear':
                klass = LinearAttentionTransformerBlock
            elif att_type == 'softmax':
                klass = SoftMaxAttentionTransformerBlock
            else:                                                                      # (:1393-1395, 1429-1430)
                return ViT3D(in_channels=d, patch_size=patch_size, num_heads=heads, dim_head=attn_dim_head, img_size=size,
                             depth=depth, forward_drop_p=att_forward_drop, drop_p=att_drop,
                             forward_expansion=att_forward_expansion, reduction=False, local=att_localvit, groups=groups)
            return klass(dim=d, depth=depth, heads=heads, dim_head=attn_dim_head, ff_mult=att_forward_expansion,
                         patch_size=patch_size, img_size=size, patch=True, groups=groups)

        rb = partial(ResnetBlock, time_cond_dim=time_cond_dim, boundary=boundary, factor=batch_sample_factor)
        self.downs = nn.ModuleList([])
        self.ups = nn.ModuleList([])
        skip_connect_dims, img_sizes = [], []
        self.patch_size = 8
        cur_size = img_size
        for ind, ((dim_in, dim_out), n_blocks, groups_) in enumerate(zip(in_out, num_resnet_blocks, resnet_groups)):
            is_last = ind >= (num_layers - 1)
            current_dim = dim_in
            pre_downsample = None
            if memory_efficient:
                pre_downsample = Downsample(dim_in, dim_out)
                current_dim = dim_out
            img_sizes.append(cur_size)
            if ind != num_layers - 1:
                skip_connect_dims.append(current_dim)
            if not memory_efficient:
                post_downsample = Downsample(current_dim, dim_out) if not is_last else Conv3d(current_dim, dim_out, 1)
            else:
                post_downsample = Conv3d(dim_out, dim_out, 1)
            transformer_enc = make_attn(current_dim, attend_at_enc_depth[ind], attend_at_enc_heads[ind],
                                        self.patch_size, cur_size) if attend_at_enc[ind] else None
            self.downs.append(nn.ModuleList([
                pre_downsample,
                rb(current_dim, current_dim, groups=groups_, use_se=use_se_attn),
                transformer_enc,
                nn.ModuleList([rb(current_dim, current_dim, groups=groups_, use_se=use_se_attn) for _ in range(n_blocks)]),
                post_downsample]))
            cur_size = cur_size // 2
            if not is_last:
                self.patch_size = self.patch_size // 2

        mid_dim = dims[-1]
        if deep_feature:
            self.mid_attn = make_attn(mid_dim, attend_at_middle_depth, attend_at_middle_heads, self.patch_size,
                                      img_sizes[-1]) if attend_at_middle else None
        self.mid_block = rb(mid_dim, mid_dim, groups=resnet_groups[-1])      # built even when unused (:1431-1434)

        for ind, ((dim_out, dim_in), n_blocks, groups_) in enumerate(zip(reversed(in_out), reversed(num_resnet_blocks),
                                                                         reversed(resnet_groups))):
            if ind == 0:
                dim_in = mid_dim
            is_last = ind == (len(in_out) - 1)
            if not is_last:
                skip_connect_dim = skip_connect_dims.pop()
            self.ups.append(nn.ModuleList([
                PixelShuffleUpsample(dim_in, dim_out) if not is_last else None,
                rb(dim_out + skip_connect_dim, dim_out, groups=groups_, use_se=use_se_attn) if not is_last else
                rb(dim_in, dim_out, groups=groups_, use_se=use_se_attn),
                nn.ModuleList([rb(dim_out, dim_out, groups=groups_, use_se=use_se_attn) for _ in range(n_blocks)])]))

        final_conv_dim = dim_out
        self.final_res_block = rb(final_conv_dim, dim, groups=resnet_groups[0], use_se=use_se_attn) \
            if final_resnet_block else None
        final_conv_dim_in = dim if final_resnet_block else final_conv_dim
        self.final_conv = Conv3d(final_conv_dim_in, self.channels_out, final_conv_kernel_size,
                                 padding=final_conv_kernel_size // 2 if final_conv_kernel_size > 1 else 0)

    # cascading-DDPM re-init (:1482-1500); accepts the ElucidatedImagen spelling too (superset)
    def cast_model_parameters(self, *, lowres_cond, channels, channels_out, cond_on_text=None, text_embed_dim=None):
        if lowres_cond == self.lowres_cond and channels == self.channels and channels_out == self.channels_out:
            return self
        return self.__class__(**{**self._locals, **dict(lowres_cond=lowres_cond, channels=channels,
                                                        channels_out=channels_out)})

    def to_config_and_state_dict(self):
        return self._locals, self.state_dict()

    @classmethod
    def from_config_and_state_dict(klass, config, state_dict):
        unet = klass(**config)
        unet.load_state_dict(state_dict)
        return unet

    def persist_to_file(self, path):
        path = Path(path)
        path.parents[0].mkdir(exist_ok=True, parents=True)
        config, state_dict = self.to_config_and_state_dict()
        torch.save(dict(config=config, state_dict=state_dict), str(path))

    @classmethod
    def hydrate_from_file(klass, path):
        pkg = torch.load(str(Path(path)))
        assert 'config' in pkg and 'state_dict' in pkg
        return Unet.from_config_and_state_dict(pkg['config'], pkg['state_dict'])

    def forward_with_cond_scale(self, *args, cond_scale=1., **kwargs):
        logits = self.forward(*args, **kwargs)
        if cond_scale == 1:
            return logits
        null_logits = self.forward(*args, cond_drop_prob=1., **kwargs)
        return null_logits + (logits - null_logits) * cond_scale

    def _merged_attention(self, block, x):
        """merge the f^3 sub-volumes into one volume, attend, split back (:1610-1622)."""
        f = self.batch_sample_factor
        B, A = x.shape[0], x.shape[1]
        if B != f ** 3:
            raise ValueError("The batch size must be the product of split dimensions")     # utils_mine.py:57-59
        vol = ops.merge_volume(x, f) if f > 1 else x
        vol = block(vol)
        return ops.split_volume(vol, f, A) if f > 1 else vol

    def forward(self, x, time_steps=None, time=None, *, lowres_cond_img=None, cond_images=None, self_cond=None,
                cond_drop_prob=0.):
        """(x[B,C,S,S,S], time_steps (ignored, as in the reference), time = log-SNR [B]) -> [B,C_out,S,S,S].
        Also accepts the 2-positional EDM form ``forward(x, c_noise)``."""
        if time is None:
            time, time_steps = time_steps, None
        x = to_channels_last(x.float())
        if self.self_cond:
            sc = to_channels_last(self_cond) if exists(self_cond) else torch.zeros_like(x)
            x = ops.concat_channels(x, sc)
        assert not (self.lowres_cond and not exists(lowres_cond_img)), 'low resolution conditioning image must be present'
        if exists(lowres_cond_img):
            x = ops.concat_channels(x, to_channels_last(lowres_cond_img.float()))
        assert not (self.has_cond_image ^ exists(cond_images)), \
            'you either requested to condition on an image on the unet, but the conditioning image is not supplied, or vice versa'
        if exists(cond_images):
            assert cond_images.shape[1] == self.cond_images_channels
            x = ops.concat_channels(to_channels_last(cond_images.float()), x)

        if self.boundary and not torch.is_grad_enabled() and isinstance(self.init_conv, Conv3d) and \
                (y0 := ops.conv3d_neighbours(x, self.init_conv.weight, self.init_conv.bias, self.batch_sample_factor)) is not None:
            x = y0
        else:
            if self.boundary:
                x = boundary_pad(x)
            x = self.init_conv(x)

        t = TimeCond(self.to_time_cond(self.to_time_hiddens(time.float().contiguous())))
        if getattr(self, '_time_mlps', None) is None:
            self._time_mlps = BatchedTimeMLPs(m.time_mlp[1] for m in self.modules() if isinstance(m, ResnetBlock) and m.time_mlp is not None)
        t.batched = self._time_mlps(t)

        hiddens = []
        last = len(self.downs)
        for i, (pre_downsample, init_block, attn_block, resnet_blocks, post_downsample) in enumerate(self.downs):
            if exists(pre_downsample):
                x = pre_downsample(x)
            x = init_block(x, t)
            if exists(attn_block):
                x = ops.add(self._merged_attention(attn_block, x), x)
            for resnet_block in resnet_blocks:
                x = resnet_block(x, t)
            if i != last - 1:
                hiddens.append(x)
            if exists(post_downsample):
                x = post_downsample(x)

        if self.deep_feature:
            if exists(self.mid_attn):
                x = self._merged_attention(self.mid_attn, x)
            x = self.mid_block(x, t)

        for upsample, init_block, resnet_blocks in self.ups:
            if exists(upsample):
                x = upsample(x)
                x = ops.concat_channels(x, hiddens.pop(), 1.0, self.skip_connect_scale)      # cat(x, skip * scale) in one pass
            x = init_block(x, t)
            for resnet_block in resnet_blocks:
                x = resnet_block(x, t)

        if exists(self.final_res_block):
            x = self.final_res_block(x, t)
        return to_channels_first(self.final_conv(x))


class NullUnet(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        self.lowres_cond = False
        self.dummy_parameter = nn.Parameter(torch.tensor([0.]))

    def cast_model_parameters(self, *args, **kwargs):
        return self

    def forward(self, x, *args, **kwargs):
        return x


class BaseUnet64(Unet):
    def __init__(self, *args, **kwargs):
        default_kwargs = dict(dim=512, dim_mults=(1, 2, 3, 4), num_resnet_blocks=3, attn_heads=8, ff_mult=2.,
                              memory_efficient=False)
        super().__init__(*args, **{**default_kwargs, **kwargs})


class SRUnet256(Unet):
    def __init__(self, *args, **kwargs):
        default_kwargs = dict(dim=128, dim_mults=(1, 2, 4, 8), num_resnet_blocks=(2, 4, 8, 8), attn_heads=8, ff_mult=2.,
                              memory_efficient=True)
        super().__init__(*args, **{**default_kwargs, **kwargs})


# ----------------------------------------------------------------------------------------------
# Imagen: continuous-time DDPM wrapper (imagen_pytorch3D.py:1741-2442)
# ----------------------------------------------------------------------------------------------
class Imagen(nn.Module):
    def __init__(
        self, unets, configs, *, image_sizes, min_bound=0, channels=3, timesteps=1000, cond_drop_prob=0.1,
        loss_type='l2', noise_schedules='cosine', pred_objectives='noise', lowres_noise_schedule='linear',
        lowres_sample_noise_level=0.2, per_sample_random_aug_noise_level=False, auto_normalize_img=False,
        p2_loss_weight_gamma=0.5, p2_loss_weight_k=1, dynamic_thresholding=True, dynamic_thresholding_percentile=0.95,
        only_train_unet_number=None, temporal_downsample_factor=1, lpips=False, medlpips=False, boundary=False,
    ):
        super().__init__()
        self.configs = configs
        from .graphs import GraphCache
        self._graphs = GraphCache()           # hipGraph replay of launch-bound U-Net evaluations of the sampling loop (graphs.py)
        self.medlpips = medlpips
        self.boundary = boundary
        assert not lpips and not medlpips, 'perceptual losses are dead code in the reference (SURVEY.md §2 #10)'
        self.lpips = None
        if loss_type not in ops.LOSS_KINDS:                    # 'l1' F.l1_loss, 'l2' F.mse_loss, 'huber' F.smooth_l1_loss (:1785-1792)
            raise NotImplementedError()
        self.loss_type = loss_type
        self.min_bound = min_bound
        self.condition_on_text = False
        self.unconditional = True
        self.channels = channels

        unets = cast_tuple(unets)
        num_unets = len(unets)
        timesteps = cast_tuple(timesteps, num_unets)
        noise_schedules = cast_tuple(noise_schedules)
        noise_schedules = pad_tuple_to_length(noise_schedules, 2, 'cosine')
        noise_schedules = pad_tuple_to_length(noise_schedules, num_unets, 'linear')
        self.noise_schedulers = nn.ModuleList([GaussianDiffusionContinuousTimes(noise_schedule=s, timesteps=t)
                                               for t, s in zip(timesteps, noise_schedules)])
        self.lowres_noise_schedule = GaussianDiffusionContinuousTimes(noise_schedule=lowres_noise_schedule)
        self.pred_objectives = cast_tuple(pred_objectives, num_unets)

        self.unets = nn.ModuleList([])
        self.unet_being_trained_index = -1
        self.only_train_unet_number = only_train_unet_number
        for ind, one_unet in enumerate(unets):
            assert isinstance(one_unet, (Unet, NullUnet)) or hasattr(one_unet, 'cast_model_parameters')
            one_unet = one_unet.cast_model_parameters(lowres_cond=not ind == 0, channels=self.channels,
                                                      channels_out=self.channels)
            self.unets.append(one_unet)

        image_sizes = cast_tuple(image_sizes)
        self.image_sizes = image_sizes
        assert num_unets == len(image_sizes), \
            f'you did not supply the correct number of u-nets ({len(unets)}) for resolutions {image_sizes}'
        self.sample_channels = cast_tuple(self.channels, num_unets)
        temporal_downsample_factor = cast_tuple(temporal_downsample_factor, num_unets)
        self.temporal_downsample_factor = temporal_downsample_factor
        assert temporal_downsample_factor[-1] == 1, 'downsample factor of last stage must be 1'

        lowres_conditions = tuple(map(lambda t: t.lowres_cond, self.unets))
        assert lowres_conditions == (False, *((True,) * (num_unets - 1))), \
            'the first unet must be unconditioned (by low resolution image), and the rest of the unets must have `lowres_cond` set to True'

        self.lowres_sample_noise_level = lowres_sample_noise_level
        self.per_sample_random_aug_noise_level = per_sample_random_aug_noise_level
        self.cond_drop_prob = cond_drop_prob
        self.can_classifier_guidance = cond_drop_prob > 0.
        self.normalize_img = normalize_neg_one_to_one if auto_normalize_img else identity
        self.unnormalize_img = unnormalize_zero_to_one if auto_normalize_img else identity
        self.dynamic_thresholding = cast_tuple(dynamic_thresholding, num_unets)
        self.dynamic_thresholding_percentile = dynamic_thresholding_percentile
        self.p2_loss_weight_k = p2_loss_weight_k
        self.p2_loss_weight_gamma = cast_tuple(p2_loss_weight_gamma, num_unets)
        assert all([(g <= 2) for g in self.p2_loss_weight_gamma]), 'in paper, they noticed any gamma greater than 2 is harmful'
        self.register_buffer('_temp', torch.tensor([0.]), persistent=False)
        self.to(next(self.unets.parameters()).device)

    @property
    def device(self):
        return self._temp.device

    def get_unet(self, unet_number):
        assert 0 < unet_number <= len(self.unets)
        index = unet_number - 1
        if isinstance(self.unets, nn.ModuleList):
            unets_list = [unet for unet in self.unets]
            delattr(self, 'unets')
            self.unets = unets_list
        if index != self.unet_being_trained_index:
            for unet_index, unet in enumerate(self.unets):
                unet.to(self.device if unet_index == index else 'cpu')
        self.unet_being_trained_index = index
        return self.unets[index]

    def reset_unets_all_one_device(self, device=None):
        device = default(device, self.device)
        self.unets = nn.ModuleList([*self.unets])
        self.unets.to(device)
        self.unet_being_trained_index = -1

    @contextmanager
    def one_unet_in_gpu(self, unet_number=None, unet=None):
        assert exists(unet_number) ^ exists(unet)
        if exists(unet_number):
            unet = self.unets[unet_number - 1]
        devices = [next(u.parameters()).device for u in self.unets]
        self.unets.cpu()
        unet.to(self.device)
        yield
        for u, d in zip(self.unets, devices):
            u.to(d)

    def state_dict(self, *args, **kwargs):
        self.reset_unets_all_one_device()
        return super().state_dict(*args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self.reset_unets_all_one_device()
        return super().load_state_dict(*args, **kwargs)

    # ---- sampling -------------------------------------------------------------------------------
    def _clamp_cfg(self):
        if self.configs['Data']['norm'] == 'min-max':
            return -1., 1., 1          # clamp_(-1, 1)            (:2023-2024)
        return float(self.min_bound), 0., 0   # clamp_(min = min_bound)  (:2025-2026)

    def unet_eval(self, unet, img, cond, *, lowres_cond_img=None, cond_images=None, cond_scale=1., self_cond=None):
        """One U-Net evaluation of the sampling loop (``unet.forward_with_cond_scale``, imagen_pytorch3D.py:2029-2031 of the reference).  When it
        is launch-bound -- under autocast the C2 eval is ~3.2 ms of kernels behind ~3.3 ms of Python-issued launches -- the third call with the
        same shapes / weights / precision is captured into a hipGraph and replayed from then on (bit-identical: the same launches); GPU-bound
        evaluations (fp32) stay eager (graphs.py decides from the timed second call)."""
        def fwd(x, c, **kw):
            return unet.forward_with_cond_scale(x, None, c, **kw)
        kw = dict(cond_images=cond_images, cond_scale=cond_scale, lowres_cond_img=lowres_cond_img, self_cond=self_cond)
        return self._graphs.run(unet, fwd, (img, cond), kw)

    @torch.no_grad()
    def p_sample_loop(self, unet, shape, *, noise_scheduler, lowres_cond_img=None, cond_images=None, inpaint_images=None,
                      inpaint_masks=None, inpaint_resample_times=5, init_images=None, skip_steps=None, cond_scale=1,
                      pred_objective='noise', dynamic_threshold=True, use_tqdm=True, noise=None):
        """Ancestral sampler (:2059-2160).  Per step: one U-Net eval (HIP) + ONE fused posterior-step kernel;
        the per-step coefficients for all steps are computed on the host up front.  ``noise`` (optional) is a list
        [init, step_0, ...] of injected tensors with the reference's draw order (:2080, :2051)."""
        if pred_objective not in ('noise', 'x_start', 'v'):
            raise ValueError(f'unknown objective {pred_objective}')
        device = self.device
        batch = shape[0]
        noise = list(noise) if exists(noise) else None
        draw = (lambda: noise.pop(0).to(device).contiguous()) if exists(noise) else \
            (lambda: torch.randn(shape, device=device))
        img = draw()
        if exists(init_images):
            img = ops.add(img, init_images.to(device).float())

        has_inpainting = exists(inpaint_images) and exists(inpaint_masks)                  # (:2090-2091)
        resample_times = inpaint_resample_times if has_inpainting else 1
        if has_inpainting:
            inpaint_images = inpaint_images.to(device).float().expand(shape).contiguous()
            mask_f = inpaint_masks.to(device).bool().expand(shape).float().contiguous()

        timesteps = list(noise_scheduler.get_sampling_timesteps(batch, device='cpu'))
        skip_steps = default(skip_steps, 0)
        if skip_steps > 1:
            timesteps = timesteps[::skip_steps] + [timesteps[-1]]           # (:2105-2107)

        # host: coefficients of every step, uploaded once  [T, 3, B] ; log-SNR conditioning [T, B]
        coefs = torch.stack([torch.stack(noise_scheduler.posterior_coefficients(t, tn)) for t, tn in timesteps])
        conds = torch.stack([noise_scheduler.get_condition(t) for t, _ in timesteps])
        # x0 from a noise / v prediction (:343-357) and the inpainting re-noise coefficients (:324-341), per step [T, 2, B]
        al, sg = log_snr_to_alpha_sigma(conds)
        al_n, sg_n = log_snr_to_alpha_sigma(torch.stack([noise_scheduler.get_condition(tn) for _, tn in timesteps]))
        if pred_objective == 'noise':
            x0c = torch.stack((1. / al.clamp(min=1e-8), -sg / al.clamp(min=1e-8)), dim=1)
        else:
            x0c = torch.stack((al, -sg), dim=1)
        last = torch.stack([(tn == 0) for _, tn in timesteps])                                        # [T, B]
        renoise = torch.stack((torch.where(last, torch.ones_like(al), al / al_n),
                               torch.where(last, torch.zeros_like(al), (sg * al_n - sg_n * al) / al_n)), dim=1)
        coefs, conds, x0c, renoise, qs = (t.to(device) for t in (coefs, conds, x0c, renoise, torch.stack((al, sg), dim=1)))
        lo, hi, mode = self._clamp_cfg()
        lowres = lowres_cond_img.to(device).float().contiguous() if exists(lowres_cond_img) else None
        inf = float('inf')

        noisy_dev, x0_dev = [], []
        x_start = None
        for i in range(len(timesteps)):
            all_last = bool(last[i].all())
            for r in reversed(range(resample_times)):
                if has_inpainting:                                                         # (:2119-2123)
                    noised = ops.q_sample(inpaint_images, draw(), qs[i, 0], qs[i, 1])
                    img = ops.mask_blend(img, noised, mask_f)
                pred = self.unet_eval(unet, img, conds[i], cond_images=cond_images, cond_scale=cond_scale,
                                      lowres_cond_img=lowres, self_cond=x_start if unet.self_cond else None)
                pred = pred.contiguous()
                if pred_objective != 'x_start':                                            # (:1996-2003)
                    pred = ops.axpby3(img, pred, None, x0c[i, 0], x0c[i, 1], None, 0.0, 0.0, 0)
                if dynamic_threshold:                                                      # (:2006-2021)
                    s = ops.abs_quantile(pred, self.dynamic_thresholding_percentile)
                    s.clamp_(min=1. if self.configs['Data']['norm'] == 'min-max' else float(self.min_bound))
                    pred = ops.dynamic_threshold(pred, s)
                    img, x_start = ops.ddpm_step(img, pred, draw(), coefs[i, 0], coefs[i, 1], coefs[i, 2], -inf, inf, 1)
                else:
                    img, x_start = ops.ddpm_step(img, pred, draw(), coefs[i, 0], coefs[i, 1], coefs[i, 2], lo, hi, mode)
                if has_inpainting and not (r == 0 or all_last):                            # (:2139-2146)
                    img = ops.axpby3(img, draw(), None, renoise[i, 0], renoise[i, 1], None, 0.0, 0.0, 0)
            noisy_dev.append(img)
            x0_dev.append(x_start)
        noisy_dev.append(img)
        x0_dev.append(x_start)
        # one D2H at the end instead of two per step (:2148-2149); same returned values
        noisy_pred_img = [t.cpu().numpy() for t in noisy_dev]
        pred_img = [t.cpu().numpy() for t in x0_dev]
        one = torch.ones(batch, device=device)
        img = ops.axpby3(img, None, None, one, None, None, lo, hi, 1 if mode == 0 else 2)   # final clamp (:2154-2157)
        return self.unnormalize_img(img), noisy_pred_img, pred_img

    @torch.no_grad()
    @eval_decorator
    def sample(self, text_masks=None, text_embeds=None, video_frames=None, cond_images=None, inpaint_images=None,
               inpaint_masks=None, inpaint_resample_times=5, init_images=None, skip_steps=None, batch_size=1,
               cond_scale=1., lowres_sample_noise_level=None, start_at_unet_number=1, start_image_or_video=None,
               stop_at_unet_number=None, return_all_outputs=False, return_all_unet_outputs=None, return_pil_images=False,
               device=None, use_tqdm=True, noise=None):
        """imagen_pytorch3D.py:2165-2274 -> (img, [noisy per step], [x0 per step]).  Accepts both spellings
        ``return_all_outputs`` / ``return_all_unet_outputs`` (test.py:182 uses the latter)."""
        if exists(return_all_unet_outputs):
            return_all_outputs = return_all_unet_outputs
        device = default(device, self.device)
        self.reset_unets_all_one_device(device=device)
        num_unets = len(self.unets)
        cond_scale = cast_tuple(cond_scale, num_unets)
        init_images = list(cast_tuple(init_images, num_unets))
        skip_steps = cast_tuple(skip_steps, num_unets)
        if start_at_unet_number > 1:
            assert start_at_unet_number <= num_unets, 'must start a unet that is less than the total number of unets'
            assert not exists(stop_at_unet_number) or start_at_unet_number <= stop_at_unet_number
            assert exists(start_image_or_video), 'starting image or video must be supplied if only doing upscaling'
            img = start_image_or_video
        outputs = []
        lst_pred_noisy = lst_pred = None
        for unet_number, unet, image_size, noise_scheduler, pred_objective, dynamic_threshold, unet_cond_scale, \
                unet_init_images, unet_skip_steps in zip(range(1, num_unets + 1), self.unets, self.image_sizes,
                                                         self.noise_schedulers, self.pred_objectives,
                                                         self.dynamic_thresholding, cond_scale, init_images, skip_steps):
            if unet_number < start_at_unet_number:
                continue
            assert not isinstance(unet, NullUnet), 'one cannot sample from null / placeholder unets'
            lowres_cond_img = img if unet.lowres_cond else None
            shape = (batch_size, self.channels, image_size, image_size, image_size)
            img, lst_pred_noisy, lst_pred = self.p_sample_loop(
                unet, shape, cond_images=cond_images, inpaint_images=inpaint_images, inpaint_masks=inpaint_masks,
                inpaint_resample_times=inpaint_resample_times, init_images=unet_init_images, skip_steps=unet_skip_steps,
                cond_scale=unet_cond_scale, lowres_cond_img=lowres_cond_img, noise_scheduler=noise_scheduler,
                pred_objective=pred_objective, dynamic_threshold=dynamic_threshold, use_tqdm=use_tqdm, noise=noise)
            outputs.append(img)
            if exists(stop_at_unet_number) and stop_at_unet_number == unet_number:
                break
        output_index = -1 if not return_all_outputs else slice(None)
        return outputs[output_index], lst_pred_noisy, lst_pred

    # ---- training ---------------------------------------------------------------------------------
    def p_losses(self, unet, x_start, times, *, noise_scheduler, lowres_cond_img=None, cond_images=None, noise=None,
                 pred_objective='noise', p2_loss_weight_gamma=0., deferred=False, **kwargs):
        """imagen_pytorch3D.py:2277-2387 -> (loss, pred, x_noisy, lowres).

        ``deferred=True`` returns ``(core, tensors)`` instead: everything that touches the host (the noise schedule is evaluated on the
        host like in the reference's trace, its values uploaded) has happened, ``core(*tensors)`` is device work only -- q_sample, the
        U-Net, the loss -- with every device tensor it reads passed in, so the trainer can capture it (and the backward behind it) into a
        hipGraph and replay it on other tensors of the same shapes (trainer._TrainGraphs)."""
        if pred_objective not in ('noise', 'x_start', 'v'):
            raise ValueError(f'unknown objective {pred_objective}')
        device = x_start.device
        x_start = self.normalize_img(x_start).float().contiguous()
        lowres_cond_img = maybe(self.normalize_img)(lowres_cond_img)
        noise = default(noise, lambda: torch.randn_like(x_start)).to(device).contiguous()
        times_cpu = times.detach().cpu().float()
        log_snr = noise_scheduler.log_snr(times_cpu)
        alpha, sigma = log_snr_to_alpha_sigma(log_snr)
        inner = unet.module if hasattr(unet, 'module') else unet
        assert not inner.self_cond, 'self-conditioning is not used by the IQT path'
        weight = None
        if p2_loss_weight_gamma > 0:                                                       # (:2368-2370)
            weight = ((self.p2_loss_weight_k + log_snr.exp()) ** -p2_loss_weight_gamma).to(device)
        min_bound, loss_type, drop = float(self.min_bound), self.loss_type, self.cond_drop_prob

        def core(x_start, noise, alpha, sigma, neg_sigma, noise_cond, lowres_cond_img, cond_images, weight):
            x_noisy = ops.q_sample(x_start, noise, alpha, sigma)                           # (:311-322)
            pred = unet.forward(x_noisy, times, noise_cond, lowres_cond_img=lowres_cond_img, cond_images=cond_images, cond_drop_prob=drop)
            if pred_objective == 'x_start':
                # in-place clamp_(min_bound) + MSE mean in one kernel; returns the clamped pred like the reference (:2361-2364)
                loss, pred = ops.mse_clamp(pred, x_start, lo=min_bound, do_clamp=True, weight=weight, kind=loss_type)
            else:
                if pred_objective == 'noise':                                              # (:2344-2345)
                    target = noise
                else:                                                                      # v = alpha*eps - sigma*x0 (:2348-2352)
                    target = ops.axpby3(noise, x_start, None, alpha, neg_sigma, None, 0.0, 0.0, 0)
                loss, pred = ops.mse_clamp(pred, target, lo=0.0, do_clamp=False, weight=weight, kind=loss_type)
            return loss, pred, x_noisy, lowres_cond_img

        tensors = (x_start, noise, alpha.to(device), sigma.to(device), (-sigma).to(device) if pred_objective == 'v' else None,
                   log_snr.to(device), lowres_cond_img, cond_images, weight)
        # everything else `core` depends on by value: part of the key of a captured micro-step
        core.static_key = (pred_objective, loss_type, min_bound, drop, bool(inner.training), id(inner))
        return (core, tensors) if deferred else core(*tensors)

    def forward(self, images, lowres_img=None, unet=None, text_embeds=None, text_masks=None, unet_number=None,
                cond_images=None, **kwargs):
        """imagen_pytorch3D.py:2390-2442."""
        assert images.shape[-1] == images.shape[-2], \
            f'the images you pass in must be a square, but received dimensions of {images.shape[2]}, {images.shape[-1]}'
        assert not (len(self.unets) > 1 and not exists(unet_number)), \
            f'you must specify which unet you want trained, from a range of 1 to {len(self.unets)}, if you are training cascading DDPM (multiple unets)'
        unet_number = default(unet_number, 1)
        assert not exists(self.only_train_unet_number) or self.only_train_unet_number == unet_number, \
            f'you can only train on unet #{self.only_train_unet_number}'
        assert images.dtype == torch.float, f'images tensor needs to be floats but {images.dtype} dtype found instead'
        unet_index = unet_number - 1
        unet = default(unet, lambda: self.get_unet(unet_number))
        inner = unet.module if hasattr(unet, 'module') else unet
        assert not isinstance(inner, NullUnet), 'null unet cannot and should not be trained'
        noise_scheduler = self.noise_schedulers[unet_index]
        target_image_size = self.image_sizes[unet_index]
        b, c, h, w = images.shape[0], images.shape[1], images.shape[-3], images.shape[-2]
        assert c == self.channels
        assert h >= target_image_size and w >= target_image_size
        if self.configs['Train']['batch_sample']:
            times = noise_scheduler.sample_random_times(1, device='cpu').repeat(b)          # one t for all sub-volumes
        else:
            times = noise_scheduler.sample_random_times(b, device='cpu')
        assert lowres_img is not None, 'lowres image must be provided'
        self.lowres_cond_img = lowres_img
        return self.p_losses(unet, images, times, cond_images=cond_images, noise_scheduler=noise_scheduler,
                             lowres_cond_img=lowres_img, pred_objective=self.pred_objectives[unet_index],
                             p2_loss_weight_gamma=self.p2_loss_weight_gamma[unet_index], **kwargs)
