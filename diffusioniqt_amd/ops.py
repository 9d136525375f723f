"""Autograd operators of the hot path, each a thin ``torch.autograd.Function`` over libdiqt_hip.so.

PyTorch supplies device memory, streams and the autograd tape; all arithmetic runs in the HIP kernels
of ``csrc/``.  Activations are fp32 channels-last: ``x[B, D, H, W, C]`` (or ``[rows, C]``).
There is no CPU fallback: a non-CUDA tensor raises.
"""
import contextlib
import os

import torch
from torch.autograd import Function

from . import _lib

ACT_NONE, ACT_MISH, ACT_SILU, ACT_GELU, ACT_RELU, ACT_SIGMOID = range(6)

_WEIGHT_EPOCH = 0     # bumped by the fused optimiser (it writes parameters through raw pointers)


def bump_weight_epoch():
    global _WEIGHT_EPOCH
    _WEIGHT_EPOCH += 1
    if _LIVE_GRAPHS > 0:          # every captured graph froze the old weights' packed copies: stale from here on (graphs.drop_all)
        from . import graphs
        graphs.drop_all()


# A captured hipGraph (graphs.py) holds raw device addresses of everything its launches read: besides the tensors of its private pool
# also what the host side had cached BEFORE the capture (packed weights, position-bias tables, the scratch arena).  When such a cache
# replaces its tensor while a graph is alive the old one is parked here instead of being freed, until the last graph is gone.
_GRAPH_PINS = []
_LIVE_GRAPHS = 0


_CAPTURE_CLOCK = 0      # number of graph captures begun so far


def capture_begins():
    """graphs.py, right before a capture: tensors the host caches created up to here may be addressed by the new graph."""
    global _CAPTURE_CLOCK
    _CAPTURE_CLOCK += 1


def born(t):
    """Stamps a tensor a host-side cache is about to keep with the capture clock (see ``retire``)."""
    if t is not None:
        t._diqt_born = _CAPTURE_CLOCK
    return t


def retire(*tensors):
    """A host-side cache is dropping these device tensors: keep them alive while any captured graph may still address them.  A tensor
    created after the newest capture began (``born``) is addressed by no graph -- or lives in that graph's own pool -- and is not pinned:
    with a captured TRAINING step alive for the whole run, every validation / sampling pass in between re-packs the weights, and those
    copies must not pile up."""
    if _LIVE_GRAPHS > 0:
        _GRAPH_PINS.extend(t for t in tensors if t is not None and getattr(t, "_diqt_born", -1) < _CAPTURE_CLOCK)


def graphs_alive(delta):
    """graphs.py: ``delta`` graphs were captured (+) or dropped (-)."""
    global _LIVE_GRAPHS
    _LIVE_GRAPHS = max(_LIVE_GRAPHS + delta, 0)
    if _LIVE_GRAPHS == 0:
        _GRAPH_PINS.clear()


def set_gnbwd_fuse(on: bool) -> bool:
    """Switches the GroupNorm-backward reduction between its own pass (False, the default) and the epilogue of the conv's
    backward-data launch (True; ``diqt_conv3d_fwd_gnbwd``).  Returns the previous setting.  Both are product paths and the parity
    suite runs whole-network gradients in each."""
    return bool(_lib.query("diqt_set_gnbwd_fuse", int(bool(on))))


@contextlib.contextmanager
def gnbwd_fuse(on: bool):
    prev = set_gnbwd_fuse(on)
    try:
        yield
    finally:
        set_gnbwd_fuse(prev)


# --------------------------------------------------------------------------------------------
# mixed precision: the reference's fp16 switch is torch.autocast (ImagenTrainer(fp16=True) -> Accelerator(mixed_precision='fp16'),
# trainer.py:293-311; `torch.autocast` around ElucidatedImagen.sample).  Under autocast -- or inside
# ``low_precision('fp16' | 'bf16')`` -- conv3d / linear FORWARDS run on the fp16 / bf16 MFMA kernel (fp32 accumulate, result
# rounded once to the operand type like an fp16 output tensor); everything autocast keeps in fp32 (GroupNorm, soft-max, losses,
# the sampler arithmetic) is untouched, and so are attention products and backward passes (fp32 here: more precise than the
# reference, never less).  Activations stay fp32 in HBM; the cast happens inside the kernel while the halo tile is staged.
# --------------------------------------------------------------------------------------------
_LP_FORCED = None      # None: follow torch.autocast; 'off' | 'fp16' | 'bf16'


class low_precision:
    """Context manager forcing the conv/linear compute type: 'fp16', 'bf16' or 'off' (fp32 even under torch.autocast)."""

    def __init__(self, mode):
        assert mode in ('off', 'fp16', 'bf16'), mode
        self.mode = mode

    def __enter__(self):
        global _LP_FORCED
        self.prev, _LP_FORCED = _LP_FORCED, self.mode
        return self

    def __exit__(self, *exc):
        global _LP_FORCED
        _LP_FORCED = self.prev
        return False


def lp_mode():
    """None (fp32), 0 (fp16 operands) or 1 (bf16 operands) for the conv/linear forward launched now."""
    if _LP_FORCED is not None:
        return {'off': None, 'fp16': 0, 'bf16': 1}[_LP_FORCED]
    if torch.is_autocast_enabled('cuda'):
        dt = torch.get_autocast_dtype('cuda')
        return 0 if dt == torch.float16 else (1 if dt == torch.bfloat16 else None)
    return None


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("diffusioniqt_amd ops run only on MI355X (HIP) tensors; got a CPU tensor. "
                               "There is no CPU fallback — use oracle/ for CPU checks.")
        if t.dtype != torch.float32:
            raise RuntimeError(f"diffusioniqt_amd ops are fp32; got {t.dtype}")
        if not t.is_contiguous():
            raise RuntimeError("diffusioniqt_amd ops need contiguous tensors")


class KernelTimer:
    """Optional HIP-event timing of the dominant kernel (conv3d fwd/bwd-data launches) for bench.py's roofline:
    events are recorded on the stream the kernel is launched on (torch's current stream)."""
    def __init__(self):
        self.records = []      # (start_event, end_event, flops, tag, shape)
        self.enabled = False

    def reset(self):
        self.records = []

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for s, e, flops, tag, _ in self.records:
            ms, fl, n = out.get(tag, (0.0, 0.0, 0))
            out[tag] = (ms + s.elapsed_time(e), fl + flops, n + 1)
        return out

    def by_shape(self):
        """{(tag, shape): (ms, flops, launches)} -- tools/shape_profile.py"""
        torch.cuda.synchronize()
        out = {}
        for s, e, flops, tag, shape in self.records:
            ms, fl, n = out.get((tag, shape), (0.0, 0.0, 0))
            out[(tag, shape)] = (ms + s.elapsed_time(e), fl + flops, n + 1)
        return out


TIMER = KernelTimer()

_WS = {}


def _workspace(nbytes, device):
    """Grow-only scratch arena per device; kernels using it are ordered by the stream."""
    key = device.index
    ws = _WS.get(key)
    if ws is None or ws.numel() < nbytes:
        retire(ws)                                        # a captured graph may hold the old arena's address
        ws = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _WS[key] = ws
    return ws


def _reduce_ws(B, C, device):
    n = _lib.query("diqt_reduce_workspace_bytes", B, C)
    return _workspace(n, device), n


# --------------------------------------------------------------------------------------------
# convolution / linear
# --------------------------------------------------------------------------------------------
def _packed(weight5, mode):
    """Packed copy of an OIDHW weight for the MFMA kernel, cached on the tensor until it changes."""
    owner = weight5._base if weight5._base is not None else weight5   # views (nn.Linear) share the cache
    cache = getattr(owner, "_diqt_pack", None)
    if cache is None:
        cache = {}
        try:
            owner._diqt_pack = cache
        except Exception:
            pass
    key = (mode, weight5.data_ptr(), weight5._version, _WEIGHT_EPOCH)
    hit = cache.get(mode)
    if hit is not None and hit[0] == key:
        return hit[1]
    Cout, Cin, kd, kh, kw = weight5.shape
    eff = (Cout, Cin) if mode == 0 else (Cin, Cout)
    n = _lib.query("diqt_conv_packed_elems", eff[0], eff[1], kd, kh, kw)
    packed = torch.empty(n, dtype=torch.float32, device=weight5.device)
    _lib.call("diqt_conv_pack_weight", weight5.detach(), packed, Cout, Cin, kd, kh, kw, mode, _stream())
    if hit is not None:
        retire(hit[1])
    cache[mode] = (key, born(packed))
    return packed


def _packed_h(weight5, bf16, mode=0):
    """16-bit packed copy of an OIDHW weight for the fp16 / bf16 MFMA kernel, cached like ``_packed`` (mode 1: backward-data)."""
    owner = weight5._base if weight5._base is not None else weight5
    cache = getattr(owner, "_diqt_pack", None)
    if cache is None:
        cache = {}
        try:
            owner._diqt_pack = cache
        except Exception:
            pass
    slot = ('h', bf16, mode)
    key = (slot, weight5.data_ptr(), weight5._version, _WEIGHT_EPOCH)
    hit = cache.get(slot)
    if hit is not None and hit[0] == key:
        return hit[1]
    Cout, Cin, kd, kh, kw = weight5.shape
    eff = (Cout, Cin) if mode == 0 else (Cin, Cout)
    n = _lib.query("diqt_conv_packed_h_elems", eff[0], eff[1], kd, kh, kw)
    packed = torch.empty(n, dtype=torch.int16, device=weight5.device)
    _lib.call("diqt_conv_pack_weight_h", weight5.detach(), packed, Cout, Cin, kd, kh, kw, mode, bf16, _stream())
    if hit is not None:
        retire(hit[1])
    cache[slot] = (key, born(packed), weight5.detach())   # (a graph-free alias of) the weight: repack_cached_h re-derives the copy from it
    return packed


def repack_cached_h(module, bf16):
    """Re-derives every cached 16-bit packed copy (``_packed_h``) of ``module``'s weights for operand type ``bf16`` in ONE launch per 64
    weights (``diqt_conv_pack_weight_h_multi``) into fresh tensors keyed on the current weight epoch.  graphs.TrainStepGraphs calls it as the
    first thing inside the capture of a training micro-step: a replay then refreshes all packed weights with one or two launches instead
    of one per conv and direction (86 for the C2 U-Net).  Returns the number of copies re-derived."""
    rows = []
    for p in module.parameters():
        cache = getattr(p, "_diqt_pack", None)
        if not cache or not p.is_cuda:
            continue
        for slot, ent in list(cache.items()):
            if not (isinstance(slot, tuple) and slot[0] == 'h' and slot[1] == bf16 and len(ent) == 3):
                continue
            w5 = ent[2]
            if w5.data_ptr() != ent[0][1] or not w5.is_contiguous() or w5.dtype != torch.float32:
                continue                                  # the parameter moved: the lazy path repacks it
            mode = slot[2]
            packed = torch.empty(ent[1].numel(), dtype=torch.int16, device=w5.device)
            Cout, Cin, kd, kh, kw = w5.shape
            rows.append((w5.data_ptr(), packed.data_ptr(), Cout, Cin, kd, kh, kw, mode))
            retire(ent[1])
            cache[slot] = ((slot, w5.data_ptr(), w5._version, _WEIGHT_EPOCH), born(packed), w5)
    if rows:
        _lib.call("diqt_conv_pack_weight_h_multi", torch.tensor(rows, dtype=torch.int64), len(rows), int(bf16), _stream())
    return len(rows)


def _conv_fwd_half(x5, weight, bias, residual, pad, epad, bf16, mode=0, x_half=False, y_half=False, stats_out=None):
    """fp16 / bf16-operand forward (mode 0) or backward-data (mode 1: x5 is dY, pad / epad already transformed) through
    diqt_conv3d_fwd_h, or None when the low-precision kernel does not take this shape.  x_half / y_half: the tensor at that end holds
    16-bit values of the operand type (diqt_conv3d_fwd_h_io; the caller has asked conv_half_io16_ok)."""
    B, D, H, W, Cin = x5.shape
    kd, kh, kw = weight.shape[2:]
    Cout = weight.shape[0] if mode == 0 else weight.shape[1]
    geo = (B, D, H, W, Cin, Cout, kd, kh, kw, *pad, *epad)
    hdt = torch.bfloat16 if bf16 else torch.float16
    ex, ey = (2 if x_half else 4), (2 if y_half else 4)
    Bc = B                      # batch entries per launch: the kernel addresses a tensor through one buffer descriptor (< 1 GiB of fp32)
    io16 = x_half or y_half
    while not io16 and not _lib.query("diqt_conv3d_fwd_h_supported", Bc, *geo[1:]):
        if Bc % 2 or max(x5[:Bc].numel(), Bc * D * H * W * Cout) * 4 < (1 << 30):
            return None
        Bc //= 2
    assert not io16 or (Bc == B and x5.dtype == (hdt if x_half else torch.float32))
    Do, Ho, Wo = D + 2 * pad[0] + epad[0] - kd + 1, H + 2 * pad[1] + epad[1] - kh + 1, W + 2 * pad[2] + epad[2] - kw + 1
    y = torch.empty((B, Do, Ho, Wo, Cout), dtype=hdt if y_half else torch.float32, device=x5.device)
    if TIMER.enabled:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
    packed = _packed_h(weight, bf16, mode)
    if io16:
        stats = None
        if stats_out is not None:
            nblk = _lib.query("diqt_conv3d_fwd_h_stats_blocks", *geo, int(x_half), int(y_half))
            if nblk > 0:                     # per-(tile, wave) column sums of y for the consumer's GroupNorm
                stats = torch.empty((B, nblk, 2, Cout), dtype=torch.float32, device=x5.device)
                stats_out.append(ColStats(stats, nblk, Do * Ho * Wo))
        _lib.call("diqt_conv3d_fwd_h_io", x5, packed, bias, residual, y, *geo, bf16, 1, int(x_half), int(y_half), stats, _stream())
    else:
        for b0 in range(0, B, Bc):
            _lib.call("diqt_conv3d_fwd_h", x5[b0:b0 + Bc], packed, bias, residual[b0:b0 + Bc] if residual is not None else None,
                      y[b0:b0 + Bc], Bc, *geo[1:], bf16, 1, _stream())
    if TIMER.enabled:
        e.record()
        # rocprofv3's name of the kernel that ran: launches of >= 2 x 256 units take the persistent form
        last = _lib.last_launch()
        tag = ("conv_f9h_kernel" if "v9h" in last else "conv_pw_h_kernel" if "gemm" in last
               else "conv_fwd_hp_kernel" if "persistent" in last else "conv_fwd_h_kernel")
        TIMER.records.append((s, e, 2.0 * B * Do * Ho * Wo * Cout * Cin * kd * kh * kw, tag, (B, D, H, W, Cin, Cout, kd, kh, kw)))
    return y


def _groupnorm_act_h(x, gamma, beta, ss, groups, act, eps, lp):
    """act(GN(x) * (scale + 1) + shift) stored in the operand type of the 16-bit conv that consumes it (sampling path under autocast).
    x: fp32, or the 16-bit output of a conv of this path (conv_pair_nograd_h(out_half=True): it carries its GroupNorm statistics)."""
    B, C = x.shape[0], x.shape[-1]
    rows = x.numel() // (B * C)
    mean = torch.empty(B * groups, dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    s = _stream()
    pre = getattr(x, "_diqt_stats", None)
    x_half = x.dtype != torch.float32
    if pre is not None and pre.rows == rows and pre.partials.shape == (B, pre.nblk, 2, C):
        _lib.call("diqt_groupnorm_stats_from_partials", pre.partials, mean, rstd, B, pre.nblk, rows, C, groups, float(eps), s)
    elif x_half:
        raise RuntimeError("a 16-bit block output must carry the GroupNorm statistics of the conv that wrote it")
    else:
        ws, n = _reduce_ws(B, C, x.device)
        _lib.call("diqt_groupnorm_stats", x, mean, rstd, ws, n, B, rows, C, groups, float(eps), s)
    scale = shift = None
    cs = 0
    if isinstance(ss, SSView):
        assert ss.width == 2 * C and ss.base.shape[0] == B
        scale, cs = ss.base.data_ptr() + 4 * ss.off, ss.base.shape[1]
        shift = scale + 4 * C
    elif ss is not None:
        assert ss.shape == (B, 2 * C) and ss.stride(1) == 1 and ss.is_cuda and ss.dtype == torch.float32
        scale, shift, cs = ss.data_ptr(), ss.data_ptr() + 4 * C, ss.stride(0)
    y = torch.empty(x.shape, dtype=torch.bfloat16 if lp == 1 else torch.float16, device=x.device)
    _lib.call("diqt_gn_act_fwd_h", x, mean, rstd, gamma, beta, scale, shift, cs, y, B, rows, C, groups, act, lp, int(x_half), s)
    return y


def gn_conv3d_h(x, gamma, beta, scale_shift, groups, act, eps, weight, bias, padding, residual=None, want_stats=False, out_half=False):
    """Sampling path under autocast: ``conv3d(act(GN(x) * (scale + 1) + shift))`` with the GroupNorm-apply pass writing the conv's input in
    the operand type and the conv on the LDS-DMA 16-bit kernel (``conv_f9h_kernel``: the 3x3x3 Blocks of Family A, imagen_pytorch3D.py:535-566).
    x: fp32, or the 16-bit output of such a conv (``out_half``: it carries its GroupNorm statistics).  ``out_half``: y only feeds the next
    Block's GroupNorm (ResnetBlock: block1 -> block2) -- stored in the operand type, the values autocast's conv output tensor holds anyway.
    None when not under autocast or the shape is not taken by a 16-bit-input kernel."""
    lp = lp_mode()
    if lp is None or torch.is_grad_enabled() or x.dim() != 5:
        return None
    B, D, H, W, C = x.shape
    Cout, Cin, kd, kh, kw = weight.shape
    padding = tuple(int(p) for p in ((padding,) * 3 if isinstance(padding, int) else padding))
    x_is_half = x.dtype != torch.float32
    yh = bool(out_half and want_stats and residual is None)
    geo = (B, D, H, W, C, Cout, kd, kh, kw, *padding, 0, 0, 0)
    ok = Cin == C and C % groups == 0 and C % 4 == 0 and _lib.query("diqt_conv3d_fwd_h_io16_supported", *geo, 1, int(yh))
    if ok and yh:
        ok = _lib.query("diqt_conv3d_fwd_h_stats_blocks", *geo, 1, 1) > 0
    if not ok:
        if x_is_half:
            raise RuntimeError("gn_conv3d_h: a 16-bit block output reached a conv that does not take 16-bit input")
        return None
    _chk(None if x_is_half else x, gamma, beta, weight, bias, residual, scale_shift.base if isinstance(scale_shift, SSView) else None)
    xin = _groupnorm_act_h(x, gamma, beta, scale_shift, groups, act, eps, lp)
    holder = [] if want_stats else None
    y = _conv_fwd_half(xin, weight, bias, residual, padding, (0, 0, 0), lp, x_half=True, y_half=yh, stats_out=holder)
    if holder:
        y._diqt_stats = holder[0]
    assert not yh or holder
    return y


def conv_half_out_ok(shape5, weight, padding):
    """May a Block hand its conv output on in the operand type?  (the NEXT Block's conv has to take a 16-bit input of that shape)"""
    if lp_mode() is None or torch.is_grad_enabled():
        return False
    B, D, H, W, C = shape5
    Cout, Cin, kd, kh, kw = weight.shape
    padding = tuple(int(p) for p in ((padding,) * 3 if isinstance(padding, int) else padding))
    return bool(Cin == C and _lib.query("diqt_conv3d_fwd_h_io16_supported", B, D, H, W, C, Cout, kd, kh, kw, *padding, 0, 0, 0, 1, 0))


def conv_pair_nograd_h(x, w1, b1, pad1, w2, b2, pad2, epad2, residual=None, gn=None, want_stats=False, out_half=False):
    """Two convs in a row on the sampling path under autocast -- the per-frame and the temporal conv of a pseudo-3D block -- with
    the tensor between them in the operand type (it holds exactly the values the fp32 tensor would: the first conv's result is rounded
    to that type either way).  ``gn`` = (gamma, beta, scale_shift, groups, act, eps): x is the raw input of the block's GroupNorm, whose
    apply pass then writes the first conv's input in the operand type as well.  None when not under autocast or a shape is not taken
    by the persistent 16-bit kernel."""
    lp = lp_mode()
    if lp is None or torch.is_grad_enabled():
        return None
    x_is_half = x.dtype != torch.float32                  # the 16-bit output of the previous block's pair (out_half)
    assert not x_is_half or gn is not None
    _chk(None if x_is_half else x, w1, b1, w2, b2, residual)
    B, D, H, W, Cin = x.shape
    Cm, C2 = w1.shape[0], w2.shape[0]
    k1, k2 = tuple(w1.shape[2:]), tuple(w2.shape[2:])
    D1, H1, W1 = D + 2 * pad1[0] - k1[0] + 1, H + 2 * pad1[1] - k1[1] + 1, W + 2 * pad1[2] - k1[2] + 1
    xh = gn is not None
    if xh and (Cin % 4 != 0 or Cin % gn[3] != 0):
        return None
    if not (_lib.query("diqt_conv3d_fwd_h_io16_supported", B, D, H, W, Cin, Cm, *k1, *pad1, 0, 0, 0, int(xh), 1)
            and _lib.query("diqt_conv3d_fwd_h_io16_supported", B, D1, H1, W1, Cm, C2, *k2, *pad2, *epad2, 1, 0)):
        return None
    xin = x
    if xh:
        gamma, beta, ss, groups, act, eps = gn
        _chk(gamma, beta, ss.base if isinstance(ss, SSView) else None)
        xin = _groupnorm_act_h(x, gamma, beta, ss, groups, act, eps, lp)
    mid = _conv_fwd_half(xin, w1, b1, None, pad1, (0, 0, 0), lp, x_half=xh, y_half=True)
    # out_half: the pair's output only feeds the next block's GroupNorm (ResnetBlock: block1 -> block2) -- it is stored in the operand type
    # too (no residual, so it holds fp16-exact values) PROVIDED the statistics that GroupNorm needs come along
    out_half = bool(out_half and want_stats and residual is None
                    and _lib.query("diqt_conv3d_fwd_h_io16_supported", B, D1, H1, W1, Cm, C2, *k2, *pad2, *epad2, 1, 1)
                    and _lib.query("diqt_conv3d_fwd_h_stats_blocks", B, D1, H1, W1, Cm, C2, *k2, *pad2, *epad2, 1, 1) > 0)
    holder = [] if want_stats else None
    y = _conv_fwd_half(mid, w2, b2, residual, pad2, epad2, lp, x_half=True, y_half=out_half, stats_out=holder)
    if holder:
        y._diqt_stats = holder[0]           # consumed by the next GroupNorm on this exact tensor
    assert not out_half or holder
    return y


def _conv_fwd_smallcout(x5, weight, bias, residual, pad, epad):
    """Forward of a conv with one or two output channels through diqt_conv3d_fwd_smallcout (exact fp32), or None."""
    B, D, H, W, Cin = x5.shape
    Cout, _, kd, kh, kw = weight.shape
    geo = (B, D, H, W, Cin, Cout, kd, kh, kw, *pad, *epad)
    if not _lib.query("diqt_conv3d_fwd_smallcout_supported", *geo):
        return None
    Do, Ho, Wo = D + 2 * pad[0] + epad[0] - kd + 1, H + 2 * pad[1] + epad[1] - kh + 1, W + 2 * pad[2] + epad[2] - kw + 1
    y = torch.empty((B, Do, Ho, Wo, Cout), dtype=torch.float32, device=x5.device)
    if TIMER.enabled:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
    _lib.call("diqt_conv3d_fwd_smallcout", x5, weight.detach().contiguous(), bias, residual, y, *geo, _stream())
    if TIMER.enabled:
        e.record()
        TIMER.records.append((s, e, 2.0 * B * Do * Ho * Wo * Cout * Cin * kd * kh * kw, "conv_smallcout_kernel",
                              (B, D, H, W, Cin, Cout, kd, kh, kw)))
    return y


class ColStats:
    """Per-tile column sums (sum, sum of squares) of a conv output, written by the conv epilogue: [B, nblk, 2, C].
    Attached to the output tensor as ``_diqt_stats`` for the consumer's GroupNorm statistics / SE pooling."""
    __slots__ = ("partials", "nblk", "rows")

    def __init__(self, partials, nblk, rows):
        self.partials, self.nblk, self.rows = partials, nblk, rows


def _conv_fwd_raw(x5, packed, bias, residual, Cout, k, pad, epad=(0, 0, 0), stats_out=None):
    B, D, H, W, Cin = x5.shape
    kd, kh, kw = k
    pd, ph, pw = pad
    epd, eph, epw = epad
    Do, Ho, Wo = D + 2 * pd + epd - kd + 1, H + 2 * ph + eph - kh + 1, W + 2 * pw + epw - kw + 1
    y = torch.empty((B, Do, Ho, Wo, Cout), dtype=torch.float32, device=x5.device)
    if TIMER.enabled:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
    n = _lib.query("diqt_conv3d_fwd_workspace_bytes", B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph, epw)
    ws = _workspace(n, x5.device) if n else None
    stats = None
    if stats_out is not None:
        nblk = _lib.query("diqt_conv3d_fwd_stats_blocks", B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph, epw)
        if nblk > 0:
            stats = torch.empty((B, nblk, 2, Cout), dtype=torch.float32, device=x5.device)
            stats_out.append(ColStats(stats, nblk, Do * Ho * Wo))
    _lib.call("diqt_conv3d_fwd_ex", x5, packed, bias, residual, y, stats, ws, n, B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw,
              epd, eph, epw, _stream())
    if TIMER.enabled:
        e.record()
        # the tag names the kernel the C side dispatches to, so that bench.py's per-kernel numbers line up with rocprofv3's
        taps = kd * kh * kw
        kid = _lib.query("diqt_conv3d_fwd_kernel_id", B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph, epw)
        # (a split-K launch -- n > 0 -- is conv_fwd9_kernel or conv_fwd_kernel writing slabs; its interval includes the slab sum)
        tag = ("conv_fwd_kernel", "conv_fwd_smallcin_kernel", "conv1x1_fwd_kernel", "conv_fwd8_kernel", "conv_fwd9_kernel")[kid if kid > 0 and (n == 0 or kid == 4) else 0]
        TIMER.records.append((s, e, 2.0 * B * Do * Ho * Wo * Cout * Cin * taps, tag, (B, D, H, W, Cin, Cout, kd, kh, kw)))
    return y


class GnCtx:
    """What the backward-data pass of a conv needs to know about the fused GroupNorm + activation that produced its input."""
    __slots__ = ("x", "mean", "rstd", "gamma", "beta", "ss", "groups", "act")

    def __init__(self, x, mean, rstd, gamma, beta, ss, groups, act):
        self.x, self.mean, self.rstd, self.gamma, self.beta, self.ss, self.groups, self.act = x, mean, rstd, gamma, beta, ss, groups, act


def _conv_bwd_data_gn(dy, packed, Cout, k, pad, epad, gn):
    """dX of a conv whose input was act(GN(gn.x)): the conv_fwd9_kernel launch that computes dX also reduces the GroupNorm backward's
    per-channel sums in its epilogue; they travel to _GnActFn.backward on the gradient tensor (``_diqt_gnbwd``).  None: not this shape."""
    B, D, H, W, Cin = dy.shape
    kd, kh, kw = k
    geo = (B, D, H, W, Cin, Cout, kd, kh, kw, *pad, *epad)
    if gn.act not in (ACT_MISH, ACT_SILU) or tuple(gn.x.shape[-1:]) != (Cout,):
        return None
    nblk = _lib.query("diqt_conv3d_fwd_gnbwd_blocks", *geo)
    Do, Ho, Wo = D + 2 * pad[0] + epad[0] - kd + 1, H + 2 * pad[1] + epad[1] - kh + 1, W + 2 * pad[2] + epad[2] - kw + 1
    if nblk <= 0 or tuple(gn.x.shape) != (B, Do, Ho, Wo, Cout) or Cout % gn.groups != 0:
        return None
    dx = torch.empty((B, Do, Ho, Wo, Cout), dtype=torch.float32, device=dy.device)
    partials = torch.empty((B, nblk, 2, Cout), dtype=torch.float32, device=dy.device)
    scale = shift = None
    cs = 0
    if gn.ss is not None:
        scale, shift, cs = gn.ss.data_ptr(), gn.ss.data_ptr() + 4 * Cout, gn.ss.stride(0)
    if TIMER.enabled:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
    _lib.call("diqt_conv3d_fwd_gnbwd", dy, packed, dx, partials, gn.x, gn.mean, gn.rstd, gn.gamma, gn.beta, scale, shift, cs, gn.groups,
              gn.act, *geo, _stream())
    if TIMER.enabled:
        e.record()
        TIMER.records.append((s, e, 2.0 * B * Do * Ho * Wo * Cout * Cin * kd * kh * kw, "conv_fwd9_kernel", geo[:9]))
    dx._diqt_gnbwd = (partials, nblk, dx._version, dx.data_ptr())
    return dx


class _Conv3dFn(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, pad, residual, epad=(0, 0, 0), stats_out=None, gnctx=None):
        _chk(x, weight, bias, residual)
        ctx.gnctx = gnctx           # x = act(GN(.)): the backward-data pass can do the GroupNorm backward's reduction in its epilogue
        Cout, Cin, kd, kh, kw = weight.shape
        assert x.dim() == 5 and x.shape[-1] == Cin, f"conv3d: x {tuple(x.shape)} vs weight {tuple(weight.shape)}"
        lp = lp_mode()
        y = None
        if Cout <= 2 and stats_out is None:
            y = _conv_fwd_smallcout(x, weight, bias, residual, pad, epad)      # dim -> image channel: one pass over x on the vector ALU
            if y is not None:
                lp = None
        if y is None and lp is not None:
            y = _conv_fwd_half(x, weight, bias, residual, pad, epad, lp)
        if y is None:
            y = _conv_fwd_raw(x, _packed(weight, 0), bias, residual, Cout, (kd, kh, kw), pad, epad, stats_out)
        ctx.save_for_backward(x, weight)
        ctx.pad = pad
        ctx.epad = epad
        ctx.lp = lp                 # backward runs outside the autocast region: remember the forward's compute type
        ctx.has_bias = bias is not None
        ctx.has_res = residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        Cout, Cin, kd, kh, kw = weight.shape
        pd, ph, pw = ctx.pad
        epd, eph, epw = ctx.epad
        B, D, H, W, _ = x.shape
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            # dX = conv(dY, flipped W): low pad' = k-1-pad_lo, high pad' = k-1-pad_hi  ->  epad' = -epad
            bpad, bepad = (kd - 1 - pd, kh - 1 - ph, kw - 1 - pw), (-epd, -eph, -epw)
            # bf16 training: backward-data on the bf16 MFMA kernel too (the reference's autocast backward runs in the forward's
            # type).  fp16 gradients would need the GradScaler the reference pairs with fp16; they stay on the fp32 kernel.
            lpb = _lp_backward(ctx.lp)
            dx = _conv_fwd_half(dy, weight, None, None, bpad, bepad, lpb, mode=1) if lpb is not None else None
            gn = ctx.gnctx
            if dx is None and gn is not None:
                dx = _conv_bwd_data_gn(dy, _packed(weight, 1), Cin, (kd, kh, kw), bpad, bepad, gn)
            if dx is None:
                dx = _conv_fwd_raw(dy, _packed(weight, 1), None, None, Cin, (kd, kh, kw), bpad, bepad)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw = torch.empty_like(weight)
            db = torch.empty(Cout, dtype=torch.float32, device=x.device) if ctx.has_bias else None
            lpb = _lp_backward(ctx.lp)
            nh = _lib.query("diqt_conv3d_bwd_weight_h_workspace_bytes", B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph, epw) \
                if lpb is not None else 0
            if nh:
                # bf16 training: the weight gradient on the bf16 MFMA too (x and dY rounded to bf16 while staged, fp32 accumulation; the
                # reference's autocast backward runs in the forward's type); other shapes stay on the fp32 kernel
                ws = _workspace(nh, x.device)
                if TIMER.enabled:
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record()
                _lib.call("diqt_conv3d_bwd_weight_h", x, dy, dw, db, ws, nh, B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph, epw, lpb,
                          _stream())
                if TIMER.enabled:
                    e.record()
                    Do, Ho, Wo = dy.shape[1:4]
                    TIMER.records.append((s, e, 2.0 * B * Do * Ho * Wo * Cout * Cin * kd * kh * kw, "conv_wgrad_h_kernel",
                                          (B, D, H, W, Cin, Cout, kd, kh, kw)))
                return dx, dw, db, None, (dy if ctx.has_res else None), None, None, None
            n = _lib.query("diqt_conv3d_bwd_weight_workspace_bytes", B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph, epw)
            ws = _workspace(n, x.device)
            if TIMER.enabled:
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
            _lib.call("diqt_conv3d_bwd_weight", x, dy, dw, db, ws, n, B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw,
                      epd, eph, epw, _stream())
            if TIMER.enabled:
                e.record()
                Do, Ho, Wo = dy.shape[1:4]
                # tagged with rocprofv3's name of the kernel diqt_conv3d_bwd_weight dispatches to; the interval also holds the
                # fixed-order split-K slab sum that finishes the gradient
                kid = _lib.query("diqt_conv3d_bwd_weight_kernel_id", B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph, epw)
                tag = ("conv_bwd_weight_gemm", "conv_bwd_weight_kernel", "conv_bwd_weight2_kernel", "conv_wgrad3_kernel")[max(kid, 0)]
                TIMER.records.append((s, e, 2.0 * B * Do * Ho * Wo * Cout * Cin * kd * kh * kw, tag, (B, D, H, W, Cin, Cout, kd, kh, kw)))
        return dx, dw, db, None, (dy if ctx.has_res else None), None, None, None


def conv3d(x, weight, bias=None, padding=(0, 0, 0), residual=None, extra_pad=(0, 0, 0), want_stats=False):
    """Stride-1 conv on channels-last x[B,D,H,W,Cin] with an OIDHW weight (MFMA implicit GEMM).
    Filters whose halo tile cannot fit the 160 KiB LDS (e.g. 15^3 cross-embed taps) take the direct kernel."""
    if isinstance(padding, int):
        padding = (padding,) * 3
    padding = tuple(int(p) for p in padding)
    extra_pad = tuple(int(p) for p in extra_pad)
    _, D, H, W, _ = x.shape
    kd, kh, kw = weight.shape[2:]
    if _lib.query("diqt_conv3d_lds_bytes", D, H, W, kd, kh, kw, *padding, *extra_pad) > 160 * 1024:
        y = _ConvDirectFn.apply(x, weight, bias, (1, 1, 1), padding, 1, extra_pad)
        return y if residual is None else add(y, residual)
    gnctx = getattr(x, "_diqt_gnctx", None) if torch.is_grad_enabled() else None
    if not want_stats:
        return _Conv3dFn.apply(x, weight, bias, padding, residual, extra_pad, None, gnctx)
    holder = []
    y = _Conv3dFn.apply(x, weight, bias, padding, residual, extra_pad, holder, gnctx)
    if holder:
        y._diqt_stats = holder[0]           # consumed by groupnorm_act / se_gate_residual on this exact tensor
    return y


def conv3d_neighbours(x, weight, bias, f, residual=None, want_stats=False):
    """Inference-only 'same' conv over the f^3 sub-volume batch x[f^3, A, A, A, Cin] of one merged volume whose halo voxels come
    from the NEIGHBOUR sub-volumes (zero only outside the merged volume) -- the reference's ``boundary_pad`` + unpadded Conv3d
    (imagen_pytorch3D.py:37-46, 550-566) without the merge / pad / split copies.  No autograd: training keeps the copy path."""
    _chk(x, weight, bias, residual)
    assert not torch.is_grad_enabled() or not (x.requires_grad or weight.requires_grad), 'conv3d_neighbours is the sampling path'
    B, A, A2, A3, Cin = x.shape
    Cout, Cin2, k, k2, k3 = weight.shape
    assert B == f ** 3 and A == A2 == A3 and Cin == Cin2 and k == k2 == k3 and k % 2 == 1, (tuple(x.shape), tuple(weight.shape), f)
    if lp_mode() is not None:          # mixed precision: the 16-bit kernel has no neighbour tables -- materialise the copies
        return None
    p = k // 2
    geo = (B, A, A, A, Cin, Cout, k, k, k, p, p, p, 0, 0, 0)
    y = torch.empty((B, A, A, A, Cout), dtype=torch.float32, device=x.device)
    n = _lib.query("diqt_conv3d_fwd_workspace_bytes", *geo)
    ws = _workspace(n, x.device) if n else None
    stats = None
    if want_stats:
        nblk = _lib.query("diqt_conv3d_fwd_neighbours_stats_blocks", f, A, Cin, Cout, k)
        if nblk > 0:
            stats = torch.empty((B, nblk, 2, Cout), dtype=torch.float32, device=x.device)
            y._diqt_stats = ColStats(stats, nblk, A * A * A)
    _lib.call("diqt_conv3d_fwd_neighbours", x, _packed(weight, 0), bias, residual, y, stats, ws, n, f, A, Cin, Cout, k, _stream())
    return y


class _LinearSmallFn(Function):
    """nn.Linear over <= 64 rows (time-conditioning MLPs): one wave per output column, no MFMA pipeline fill."""
    @staticmethod
    def forward(ctx, x2, weight, bias):
        _chk(x2, weight, bias)
        M, K = x2.shape
        N = weight.shape[0]
        y = torch.empty((M, N), dtype=torch.float32, device=x2.device)
        _lib.call("diqt_linear_small_fwd", x2, weight, bias, y, M, K, N, _stream())
        ctx.save_for_backward(x2, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x2, weight = ctx.saved_tensors
        dy = dy.contiguous()
        M, K = x2.shape
        N = weight.shape[0]
        need_dx, need_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
        dx = torch.empty_like(x2) if need_dx else None
        dw = torch.empty_like(weight) if need_dw else None
        db = torch.empty(N, dtype=torch.float32, device=x2.device) if (need_dw and ctx.has_bias) else None
        n = _lib.query("diqt_linear_small_workspace_bytes", M, K, N)
        ws = _workspace(n, x2.device)
        _lib.call("diqt_linear_small_bwd", x2, weight, dy, dx, dw, db, ws, n, M, K, N, _stream())
        return dx, dw, db


class _BatchedLinearSmallFn(Function):
    """y_i = a W_i^T + b_i for n Linear layers that share the input a[M <= 64, K] (the time MLPs of a U-Net's ResnetBlocks,
    imagen_pytorch3D.py:586-589 / imagen_video.py:716-719) as ONE launch over the concatenated weights, with autograd: the outputs are
    column blocks of one [M, sum N_i] buffer, their gradients are written by the consumers (GroupNorm backward) into the same blocks of
    one shared gradient buffer, and backward is one dx / dW / db launch set instead of three per layer plus a sum of n gradients of a."""
    @staticmethod
    def forward(ctx, a, wcat, bcat, gradbuf, sizes, *params):
        _chk(a, wcat, bcat)
        M, K = a.shape
        N = wcat.shape[0]
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
        _lib.call("diqt_linear_small_fwd", a, wcat, bcat, out, M, K, N, _stream())
        ctx.save_for_backward(a, wcat)
        ctx.gradbuf, ctx.sizes, ctx.has_bias = gradbuf, sizes, [b is not None for b in params[1::2]]
        ctx.set_materialize_grads(False)
        offs, o = [], 0
        for n in sizes:
            offs.append(o)
            o += n
        ctx.offs = offs
        return tuple(out[:, o:o + n] for o, n in zip(offs, sizes))

    @staticmethod
    def backward(ctx, *grads):
        a, wcat = ctx.saved_tensors
        M, K = a.shape
        N = wcat.shape[0]
        buf = ctx.gradbuf
        used = [g is not None for g in grads]     # a layer whose output nobody consumed (mid_block) keeps grad None, like unbatched
        for g, o, n in zip(grads, ctx.offs, ctx.sizes):
            dst = buf[:, o:o + n]
            if g is None:
                dst.zero_()
            elif g.data_ptr() != dst.data_ptr() or g.stride() != dst.stride():
                dst.copy_(g)                    # a consumer that did not write in place (not the fused GroupNorm backward)
        dx = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(wcat)
        db = torch.empty(N, dtype=torch.float32, device=a.device)
        n = _lib.query("diqt_linear_small_workspace_bytes", M, K, N)
        ws = _workspace(n, a.device)
        _lib.call("diqt_linear_small_bwd", a, wcat, buf, dx, dw, db, ws, n, M, K, N, _stream())
        pg = []
        for o, nn_, hb, u in zip(ctx.offs, ctx.sizes, ctx.has_bias, used):
            pg.append(dw[o:o + nn_] if u else None)
            pg.append(db[o:o + nn_] if (hb and u) else None)
        return (dx, None, None, None, None, *pg)


def batched_linear_small(a, wcat, bcat, linears):
    """One launch for ``[l(a) for l in linears]`` (wcat / bcat: their concatenated weights / biases, rebuilt by the caller when a
    parameter changes).  Returns the outputs as [M, N_i] column blocks; each carries ``_diqt_gradview``, its block of the shared
    gradient buffer (ops.groupnorm_act's backward writes d scale/shift there)."""
    sizes = tuple(l.weight.shape[0] for l in linears)
    gradbuf = torch.empty((a.shape[0], wcat.shape[0]), dtype=torch.float32, device=a.device)
    params = []
    for l in linears:
        params += [l.weight, l.bias]
    outs = _BatchedLinearSmallFn.apply(a.contiguous(), wcat, bcat, gradbuf, sizes, *params)
    o = 0
    for t, n in zip(outs, sizes):
        t._diqt_gradview = gradbuf[:, o:o + n]
        o += n
    return outs


def linear(x, weight, bias=None):
    """x[..., Cin] @ weight[Cout, Cin]^T + bias: skinny kernel for <= 64 rows, otherwise the MFMA kernel (1x1x1 conv)."""
    Cout, Cin = weight.shape
    lead = x.shape[:-1]
    rows = x.numel() // Cin
    if 0 < rows <= 64:
        return _LinearSmallFn.apply(x.reshape(rows, Cin).contiguous(), weight.contiguous(), bias).reshape(*lead, Cout)
    y = _Conv3dFn.apply(x.reshape(1, 1, 1, -1, Cin), weight.view(Cout, Cin, 1, 1, 1), bias, (0, 0, 0), None, (0, 0, 0))
    return y.reshape(*lead, Cout)


class _ConvDirectFn(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, groups, epad=(0, 0, 0)):
        _chk(x, weight, bias)
        B, D, H, W, Cin = x.shape
        Cout, _, kd, kh, kw = weight.shape
        sd, sh, sw = stride
        pd, ph, pw = pad
        epd, eph, epw = epad
        Do, Ho, Wo = (D + 2 * pd + epd - kd) // sd + 1, (H + 2 * ph + eph - kh) // sh + 1, (W + 2 * pw + epw - kw) // sw + 1
        y = torch.empty((B, Do, Ho, Wo, Cout), dtype=torch.float32, device=x.device)
        ctx.geom = (B, D, H, W, Cin, Cout, groups, kd, kh, kw, sd, sh, sw, pd, ph, pw, epd, eph, epw)
        _lib.call("diqt_conv3d_direct_fwd", x, weight, bias, y, *ctx.geom, _stream())
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _lib.call("diqt_conv3d_direct_bwd_data", dy, weight, dx, *ctx.geom, _stream())
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(weight)
            db = torch.empty(weight.shape[0], dtype=torch.float32, device=x.device) if ctx.has_bias else None
            nws = _lib.query("diqt_conv3d_direct_bwd_weight_workspace_bytes", *ctx.geom)
            ws = _workspace(nws, x.device) if nws else None
            _lib.call("diqt_conv3d_direct_bwd_weight_ws", x, dy, dw, db, ws, nws, *ctx.geom, _stream())
        return dx, dw, db, None, None, None, None


def conv3d_direct(x, weight, bias=None, stride=(1, 1, 1), padding=(0, 0, 0), groups=1, extra_pad=(0, 0, 0)):
    t3 = lambda v: (v,) * 3 if isinstance(v, int) else tuple(v)
    return _ConvDirectFn.apply(x, weight, bias, t3(stride), t3(padding), int(groups), t3(extra_pad))


class _DwConvTemporalFn(Function):
    """Depthwise (3,1,1) temporal conv + bias (+ residual) on channels-last x[B,F,H,W,C] (the TemporalPEG of the pseudo-3D U-Net)."""
    @staticmethod
    def forward(ctx, x, weight, bias, left, res, res_is_x=False):
        """res_is_x: the residual IS the input (`conv(x) + x`, the TemporalPEG): one autograd input, and the backward-data launch adds
        dy to its own result -- otherwise autograd sums the two gradients of x in a separate pass."""
        if res_is_x:
            res = x
        _chk(x, weight, bias, res)
        B, F, H, W, C = x.shape
        kt = weight.shape[2]
        w2 = weight.reshape(C, kt).contiguous()
        y = torch.empty_like(x)
        _lib.call("diqt_dwconv_temporal_fwd", x, w2, bias, res, y, B, F, H * W, C, kt, int(left), 0, _stream())
        ctx.save_for_backward(x, w2)
        ctx.cfg = (B, F, H * W, C, kt, int(left), bias is not None, res is not None and not res_is_x, tuple(weight.shape))
        ctx.res_is_x = res_is_x
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w2 = ctx.saved_tensors
        B, F, P, C, kt, left, has_bias, has_res, wshape = ctx.cfg
        dy = dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _lib.call("diqt_dwconv_temporal_fwd", dy, w2, None, dy if ctx.res_is_x else None, dx, B, F, P, C, kt, kt - 1 - left, 1, _stream())
        if ctx.needs_input_grad[1]:
            n = _lib.query("diqt_dwconv_temporal_bwd_weight_workspace_bytes", B, F, P, C, kt)
            ws = _workspace(n, x.device)
            dwb = torch.empty((kt + 1, C), dtype=torch.float32, device=x.device)
            _lib.call("diqt_dwconv_temporal_bwd_weight", x, dy, dwb, ws, n, B, F, P, C, kt, left, _stream())
            dw = dwb[:kt].t().reshape(wshape)
            db = dwb[kt].clone() if has_bias else None
        return dx, dw, db, None, (dy if has_res else None), None


def dwconv_temporal_ok(x, weight, groups, stride=(1, 1, 1)):
    C = x.shape[-1]
    return (x.dim() == 5 and tuple(weight.shape) == (C, 1, 3, 1, 1) and groups == C and tuple(stride) == (1, 1, 1)
            and C % 4 == 0 and 256 % (C // 4) == 0)


def dwconv_temporal(x, weight, bias, left, residual=None):
    """nn.Conv3d(C, C, (3,1,1), groups=C) on frames padded with ``left`` zero frames in front (2: causal, 1: symmetric), + residual."""
    if residual is x:
        return _DwConvTemporalFn.apply(x.contiguous(), weight, bias, int(left), None, True)
    return _DwConvTemporalFn.apply(x.contiguous(), weight, bias, int(left), residual.contiguous() if residual is not None else None)


# --------------------------------------------------------------------------------------------
# GroupNorm + scale/shift + activation
# --------------------------------------------------------------------------------------------
class SSView:
    """Columns [off, off + width) of a batched time-MLP output ``base`` [B, sum of widths]: the (scale | shift) embedding of one
    ResnetBlock when all blocks' time MLPs ran as ONE launch (sampling path)."""
    __slots__ = ("base", "off", "width")

    def __init__(self, base, off, width):
        self.base, self.off, self.width = base, off, width


class _GnActFn(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, ss, groups, act, eps, pre=None, gn_out=None, tap=False, ss_grad=None):
        """tap=True: also returns x itself as a second output.  The caller routes x's OTHER consumer (the residual branch of a
        ResnetBlock) through that alias, so autograd sees one consumer of x and hands the branch's gradient to backward(), where it is
        added inside the dx kernel instead of by a separate elementwise pass."""
        _chk(x, gamma, beta, ss.base if isinstance(ss, SSView) else None)
        if ss is not None and not isinstance(ss, SSView):          # [B, 2C] rows, possibly a column block of a wider buffer
            assert ss.is_cuda and ss.dtype == torch.float32 and ss.dim() == 2 and ss.stride(1) == 1, "scale/shift rows must be fp32 HIP tensors"
        ctx.tap = tap
        ctx.set_materialize_grads(False)
        B, C = x.shape[0], x.shape[-1]
        rows = x.numel() // (B * C)
        mean = torch.empty(B * groups, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        s = _stream()
        if pre is not None and pre.rows == rows and pre.partials.shape == (B, pre.nblk, 2, C):
            # the producing conv's epilogue already summed x and x^2 per tile: no pass over x for the statistics
            _lib.call("diqt_groupnorm_stats_from_partials", pre.partials, mean, rstd, B, pre.nblk, rows, C, groups, float(eps), s)
        else:
            ws, n = _reduce_ws(B, C, x.device)
            _lib.call("diqt_groupnorm_stats", x, mean, rstd, ws, n, B, rows, C, groups, float(eps), s)
        y = torch.empty_like(x)
        scale = shift = None
        cs = 0
        if isinstance(ss, SSView):
            # one row block of a batched time-MLP output [B, sum 2C_i] (sampling path, no autograd): columns off .. off + 2C
            assert ss.width == 2 * C and ss.base.shape[0] == B and not torch.is_grad_enabled()
            scale, cs = ss.base.data_ptr() + 4 * ss.off, ss.base.shape[1]
            shift = scale + 4 * C
            _lib.call("diqt_gn_act_fwd", x, mean, rstd, gamma, beta, scale, shift, cs, y, B, rows, C, groups, act, s)
            return y
        if ss is not None:
            assert ss.shape == (B, 2 * C) and ss.stride(1) == 1, f"scale/shift embedding must be [B, 2C] rows, got {tuple(ss.shape)}"
            scale, shift, cs = ss.data_ptr(), ss.data_ptr() + 4 * C, ss.stride(0)
        _lib.call("diqt_gn_act_fwd", x, mean, rstd, gamma, beta, scale, shift, cs, y, B, rows, C, groups, act, s)
        # a column block of the batched time-MLP output (batched_time_mlps): its gradient is written straight into the block's place in
        # the shared gradient buffer
        ctx.ss_grad = ss_grad
        ctx.save_for_backward(x, gamma, beta, ss, mean, rstd)
        ctx.cfg = (B, rows, C, groups, act)
        if gn_out is not None and x.dim() == 5:
            gn_out.append(GnCtx(x, mean, rstd, gamma, beta, ss, groups, act))
        return (y, x) if tap else y

    @staticmethod
    def backward(ctx, dy, dtap=None):
        x, gamma, beta, ss, mean, rstd = ctx.saved_tensors
        B, rows, C, groups, act = ctx.cfg
        if dy is None:                      # only the alias was used downstream
            return dtap, None, None, None, None, None, None, None, None, None, None
        if dtap is not None:
            dtap = dtap.contiguous()
        # partial sums from the producer of dy -- valid only for the very tensor they were computed for (autograd adds other
        # branches' gradients in place: the version counter tells)
        pre = getattr(dy, "_diqt_gnbwd", None)
        if pre is not None and (pre[2] != dy._version or pre[3] != dy.data_ptr() or not dy.is_contiguous()):
            pre = None
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dgamma = torch.empty_like(gamma) if gamma is not None else None
        dbeta = torch.empty_like(beta) if beta is not None else None
        scale = shift = dscale = dshift = None
        dss = None
        cs = 0
        if ss is not None:
            cs = ss.stride(0)
            dss = ctx.ss_grad if ctx.ss_grad is not None else torch.empty_strided(ss.shape, ss.stride(), dtype=ss.dtype, device=ss.device)
            assert dss.stride() == ss.stride()
            scale, shift = ss.data_ptr(), ss.data_ptr() + 4 * C
            dscale, dshift = dss.data_ptr(), dss.data_ptr() + 4 * C
        ws, n = _reduce_ws(B, C, x.device)
        # pre: the conv behind this GroupNorm reduced (sum dz, sum dz xhat) in the epilogue of its backward-data pass; dtap: the
        # gradient of x's other consumer, added in the dx pass
        part, nblk = (pre[0], pre[1]) if pre is not None and pre[0].shape == (B, pre[1], 2, C) else (None, 0)
        _lib.call("diqt_gn_act_bwd_ex", x, dy, part, nblk, dtap, mean, rstd, gamma, beta, scale, shift, cs, dx, dgamma, dbeta,
                  dscale, dshift, ws, n, B, rows, C, groups, act, _stream())
        return dx, dgamma, dbeta, dss, None, None, None, None, None, None, None


# fp16 training with a loss scaler (ImagenTrainer(fp16=True): the reference pairs fp16 autocast with torch's GradScaler, trainer.py:293-311,
# 364): the trainer sets this while its scaler is active, and the backward kernels of fp16-forward convs then run on the fp16 MFMA too
# (dY carries the loss scale, which is what keeps fp16 gradients out of the denormals).  Without a scaler fp16 backward stays in fp32.
FP16_BACKWARD = False


def _lp_backward(lp):
    """The 16-bit type of a conv's backward kernels: bf16 always follows the forward; fp16 only under a loss scaler; else None (fp32)."""
    return lp if lp == 1 or (lp == 0 and FP16_BACKWARD) else None


_NO_TRAIN_FUSE = os.environ.get("DIQT_NO_TRAIN_FUSE") == "1"      # A/B switch: bf16 training Blocks as two autograd nodes
_NO_TRAIN_HALF = os.environ.get("DIQT_NO_TRAIN_HALF") == "1"      # A/B switch: fp32 tensor (and gradient) between block1 and block2
_NO_DACT_HALF = os.environ.get("DIQT_NO_DACT_HALF") == "1"      # A/B: the backward-data output in front of a GroupNorm backward stays fp32


class _GnActConvHFn(Function):
    """Training under ``ImagenTrainer(precision='bf16')`` -- or ``fp16=True`` with its loss scaler -- (trainer.py:293-311):
    ``conv3d(act(GN(x) * (scale + 1) + shift))`` as ONE autograd node whose intermediate -- the conv's input -- exists only in the 16-bit
    operand type: the GroupNorm-apply pass writes it so (the bits autocast's cast of the fp32 tensor produces), the forward conv reads it
    through ``conv_f9h_kernel`` (LDS-DMA, 16-bit x) and the weight gradient reads the same tensor (``diqt_conv3d_bwd_weight_h``, bit 1).
    ``out_half``: the conv's OUTPUT is an autograd tensor of that type too (a ResnetBlock's block1 -> block2: a conv result autocast rounds
    anyway); the node that consumes it (x of a 16-bit type, statistics from the column sums) hands back a 16-bit gradient
    (``diqt_gn_act_bwd_h``), which this node's backward-data runs through ``conv_f9h_kernel`` and its weight gradient reads as it is (bit 2)
    -- the same bits the fp32 tensors round to inside those kernels.  Bit-identical to the two-node path (``_GnActFn`` + ``_Conv3dFn``)."""
    @staticmethod
    def forward(ctx, x, gamma, beta, ss, groups, act, eps, pre, weight, bias, pad, residual, stats_out, tap, ss_grad, out_half):
        x_half = x.dtype != torch.float32
        _chk(None if x_half else x, gamma, beta, weight, bias, residual)
        if ss is not None:
            assert ss.is_cuda and ss.dtype == torch.float32 and ss.dim() == 2 and ss.stride(1) == 1, "scale/shift rows must be fp32 HIP tensors"
        assert not (x_half and tap), "the alias output routes an fp32 input's second consumer"
        ctx.tap = tap
        ctx.set_materialize_grads(False)
        B, C = x.shape[0], x.shape[-1]
        rows = x.numel() // (B * C)
        mean = torch.empty(B * groups, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        s = _stream()
        if pre is not None and pre.rows == rows and pre.partials.shape == (B, pre.nblk, 2, C):
            _lib.call("diqt_groupnorm_stats_from_partials", pre.partials, mean, rstd, B, pre.nblk, rows, C, groups, float(eps), s)
        else:
            assert not x_half, "a 16-bit block output must carry the GroupNorm statistics of the conv that wrote it"
            ws, n = _reduce_ws(B, C, x.device)
            _lib.call("diqt_groupnorm_stats", x, mean, rstd, ws, n, B, rows, C, groups, float(eps), s)
        scale = shift = None
        cs = 0
        if ss is not None:
            assert ss.shape == (B, 2 * C), f"scale/shift embedding must be [B, 2C] rows, got {tuple(ss.shape)}"
            scale, shift, cs = ss.data_ptr(), ss.data_ptr() + 4 * C, ss.stride(0)
        lp = lp_mode()                                       # 1: bf16, 0: fp16 (under a loss scaler)
        hdt = torch.bfloat16 if lp == 1 else torch.float16
        assert not x_half or x.dtype == hdt
        y16 = torch.empty(x.shape, dtype=hdt, device=x.device)
        _lib.call("diqt_gn_act_fwd_h", x, mean, rstd, gamma, beta, scale, shift, cs, y16, B, rows, C, groups, act, lp, int(x_half), s)
        y = _conv_fwd_half(y16, weight, bias, residual, pad, (0, 0, 0), lp, x_half=True, y_half=bool(out_half), stats_out=stats_out)
        ctx.lp = lp
        ctx.ss_grad = ss_grad
        ctx.save_for_backward(x, gamma, beta, ss, mean, rstd, y16, weight)
        ctx.cfg = (B, rows, C, groups, act, pad, bias is not None, residual is not None, x_half, bool(out_half))
        return (y, x) if tap else y

    @staticmethod
    def backward(ctx, dy, dtap=None):
        x, gamma, beta, ss, mean, rstd, y16, weight = ctx.saved_tensors
        B, rows, C, groups, act, pad, has_bias, has_res, x_half, y_half = ctx.cfg
        none = (None,) * 16
        if dy is None:                      # only the alias was used downstream
            return (dtap,) + none[1:]
        dy = dy.contiguous()
        assert (dy.dtype != torch.float32) == y_half
        Cout, Cin, kd, kh, kw = weight.shape
        D, H, W = x.shape[1:4]
        pd, ph, pw = pad
        bpad = (kd - 1 - pd, kh - 1 - ph, kw - 1 - pw)
        # ---- conv: dX on the 16-bit MFMA kernel (flipped weights; conv_f9h_kernel when dY is 16-bit), dW / db with the 16-bit activation ----
        # the gradient w.r.t. the activated tensor only feeds the GroupNorm backward's two passes: in the operand type when the 16-bit
        # kernel writes it (what autocast's conv backward returns anyway) -- half the bytes written here and read twice there
        dact_half = bool(y_half and not _NO_TRAIN_HALF and not _NO_DACT_HALF and Cin % 8 == 0
                         and _lib.query("diqt_conv3d_fwd_h_io16_supported", B, D, H, W, Cout, Cin, kd, kh, kw, *bpad, 0, 0, 0, 1, 1))
        dact = _conv_fwd_half(dy, weight, None, None, bpad, (0, 0, 0), ctx.lp, mode=1, x_half=y_half, y_half=dact_half)
        if dact is None:
            assert not y_half
            dact = _conv_fwd_raw(dy, _packed(weight, 1), None, None, Cin, (kd, kh, kw), bpad, (0, 0, 0))
        dw = torch.empty_like(weight)
        db = torch.empty(Cout, dtype=torch.float32, device=x.device) if has_bias else None
        geo = (B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, 0, 0, 0)
        nh = _lib.query("diqt_conv3d_bwd_weight_h_workspace_bytes", *geo)
        assert nh > 0
        if TIMER.enabled:
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
        _lib.call("diqt_conv3d_bwd_weight_h", y16, dy, dw, db, _workspace(nh, x.device), nh, *geo, ctx.lp | 2 | (4 if y_half else 0),
                  _stream())                                # bit 1: x is 16-bit, bit 2: dY is 16-bit
        if TIMER.enabled:
            t1.record()
            Do, Ho, Wo = dy.shape[1:4]
            TIMER.records.append((t0, t1, 2.0 * B * Do * Ho * Wo * Cout * Cin * kd * kh * kw, "conv_wgrad_h_kernel", geo[:9]))
        # ---- GroupNorm + activation backward (as _GnActFn.backward; a 16-bit x gets a 16-bit gradient) ----
        if dtap is not None:
            dtap = dtap.contiguous()
        dx = torch.empty_like(x)
        dgamma = torch.empty_like(gamma) if gamma is not None else None
        dbeta = torch.empty_like(beta) if beta is not None else None
        scale = shift = dscale = dshift = None
        dss = None
        cs = 0
        if ss is not None:
            cs = ss.stride(0)
            dss = ctx.ss_grad if ctx.ss_grad is not None else torch.empty_strided(ss.shape, ss.stride(), dtype=ss.dtype, device=ss.device)
            assert dss.stride() == ss.stride()
            scale, shift = ss.data_ptr(), ss.data_ptr() + 4 * C
            dscale, dshift = dss.data_ptr(), dss.data_ptr() + 4 * C
        ws, n = _reduce_ws(B, C, x.device)
        if x_half or dact_half:
            ty = 2 if ctx.lp == 1 else 1
            xt = ty if x_half else 0
            _lib.call("diqt_gn_act_bwd_h", x, dact, None, 0, dtap, mean, rstd, gamma, beta, scale, shift, cs, dx, dgamma, dbeta,
                      dscale, dshift, ws, n, B, rows, C, groups, act, xt, xt, ty if dact_half else 0, _stream())
        else:
            _lib.call("diqt_gn_act_bwd_ex", x, dact, None, 0, dtap, mean, rstd, gamma, beta, scale, shift, cs, dx, dgamma, dbeta,
                      dscale, dshift, ws, n, B, rows, C, groups, act, _stream())
        return dx, dgamma, dbeta, dss, None, None, None, None, dw, db, None, (dy if has_res else None), None, None, None, None


def _train_h_geo_ok(B, D, H, W, Cin, Cout, k, padding, groups):
    """Is a 'same' conv of this shape taken by the one-node low-precision training Block (16-bit-x forward kernel + 16-bit weight gradient)?"""
    kd, kh, kw = k
    geo = (B, D, H, W, Cin, Cout, kd, kh, kw, *padding, 0, 0, 0)
    if Cin % groups != 0 or Cin % 8 != 0 or not _lib.query("diqt_conv3d_fwd_h_io16_supported", *geo, 1, 0):
        return False
    if _lib.query("diqt_conv3d_bwd_weight_h_workspace_bytes", *geo) == 0:
        return False
    return D + 2 * padding[0] - kd + 1 == D and H + 2 * padding[1] - kh + 1 == H and W + 2 * padding[2] - kw + 1 == W


def train_half_out_ok(shape5, w_next, padding_next, groups_next):
    """May a Block of a low-precision TRAINING step hand its conv output on in the operand type?  (the next Block has to take it: see
    ``gn_conv3d_train_h``)"""
    if _lp_backward(lp_mode()) is None or not torch.is_grad_enabled() or _NO_TRAIN_FUSE or _NO_TRAIN_HALF:
        return False
    B, D, H, W, C = shape5
    Cout, Cin, kd, kh, kw = w_next.shape
    padding_next = tuple(int(p) for p in ((padding_next,) * 3 if isinstance(padding_next, int) else padding_next))
    return Cin == C and _train_h_geo_ok(B, D, H, W, C, Cout, (kd, kh, kw), padding_next, groups_next)


def gn_conv3d_train_h(x, gamma, beta, scale_shift, groups, act, eps, weight, bias, padding, residual=None, want_stats=False, tap=False,
                      out_half=False):
    """``Block.forward`` of a low-precision training step as one autograd node (``_GnActConvHFn``); ``tap`` as in ``groupnorm_act``;
    ``out_half``: the output may leave in the operand type (it only feeds the next Block's GroupNorm).  None when not in such a step or
    the shape is not taken by the 16-bit-input kernels (the caller then runs ``groupnorm_act`` + ``conv3d``)."""
    x_half = x.dtype != torch.float32
    if _lp_backward(lp_mode()) is None or not torch.is_grad_enabled() or x.dim() != 5 or isinstance(scale_shift, SSView) or _NO_TRAIN_FUSE:
        assert not x_half
        return None
    B, D, H, W, C = x.shape
    Cout, Cin, kd, kh, kw = weight.shape
    padding = tuple(int(p) for p in ((padding,) * 3 if isinstance(padding, int) else padding))
    if Cin != C or not _train_h_geo_ok(B, D, H, W, C, Cout, (kd, kh, kw), padding, groups):
        if x_half:
            raise RuntimeError("gn_conv3d_train_h: a 16-bit block output reached a conv that does not take 16-bit input")
        return None
    geo = (B, D, H, W, C, Cout, kd, kh, kw, *padding, 0, 0, 0)
    bpad = (kd - 1 - padding[0], kh - 1 - padding[1], kw - 1 - padding[2])
    # a 16-bit output: statistics rows for its consumer, and this node's backward-data (a conv Cout -> Cin over the 16-bit gradient)
    yh = bool(out_half and want_stats and residual is None
              and _lib.query("diqt_conv3d_fwd_h_io16_supported", *geo, 1, 1)
              and _lib.query("diqt_conv3d_fwd_h_stats_blocks", *geo, 1, 1) > 0
              and _lib.query("diqt_conv3d_fwd_h_io16_supported", B, D, H, W, Cout, C, kd, kh, kw, *bpad, 0, 0, 0, 1, 0)
              and Cout % 8 == 0)
    use_tap = tap and x.requires_grad
    ss_grad = getattr(scale_shift, "_diqt_gradview", None)
    if scale_shift is not None and ss_grad is None and not scale_shift.is_contiguous():
        scale_shift = scale_shift.contiguous()
    holder = [] if want_stats else None
    out = _GnActConvHFn.apply(x, gamma, beta, scale_shift, groups, act, eps, getattr(x, "_diqt_stats", None), weight, bias, padding, residual,
                              holder, use_tap, ss_grad, yh)
    y, alias = out if use_tap else (out, x)
    if holder:
        y._diqt_stats = holder[0]
    assert not yh or holder
    return (y, alias) if tap else y


def groupnorm_act(x, gamma, beta, scale_shift=None, groups=8, act=ACT_MISH, eps=1e-5, tap=False):
    """act(GN(x) * (scale+1) + shift) with scale_shift = one [B, 2C] embedding (scale first).
    ``tap=True`` returns ``(y, x_alias)``: route x's other consumer through ``x_alias`` (see _GnActFn.forward)."""
    if not torch.is_grad_enabled() or isinstance(scale_shift, SSView):
        y = _GnActFn.apply(x, gamma, beta, scale_shift, groups, act, eps, getattr(x, "_diqt_stats", None))
        return (y, x) if tap else y
    holder = []
    use_tap = tap and x.requires_grad
    ss_grad = getattr(scale_shift, "_diqt_gradview", None)     # a block of the batched time-MLP output: its place in the shared gradient buffer
    if scale_shift is not None and ss_grad is None and not scale_shift.is_contiguous():
        scale_shift = scale_shift.contiguous()
    out = _GnActFn.apply(x, gamma, beta, scale_shift, groups, act, eps, getattr(x, "_diqt_stats", None), holder, use_tap, ss_grad)
    y, alias = out if use_tap else (out, x)
    if holder:
        y._diqt_gnctx = holder[0]           # read by conv3d on this exact tensor (Block: GN -> act -> conv)
    return (y, alias) if tap else y


def gn_conv3d(x, gamma, beta, scale_shift, groups, act, eps, weight, bias, padding, residual=None, extra_pad=(0, 0, 0), want_stats=False):
    """Sampling path (no autograd, fp32): conv3d(act(GN(x) * (scale + 1) + shift)) as ONE conv launch -- the GroupNorm statistics
    (from the producer's column sums when x carries them) are folded into per-(batch, channel) coefficients by one tiny launch and
    ``conv_fwd9_kernel`` applies them to its input tiles while staging them (``diqt_conv3d_fwd_gn``).  Returns None when the shape is
    not taken (the caller then runs ``groupnorm_act`` + ``conv3d``)."""
    if torch.is_grad_enabled() or lp_mode() is not None or x.dim() != 5:
        return None
    B, D, H, W, C = x.shape
    Cout, Cin, kd, kh, kw = weight.shape
    padding = tuple(int(p) for p in ((padding,) * 3 if isinstance(padding, int) else padding))
    extra_pad = tuple(int(p) for p in extra_pad)
    geo = (B, D, H, W, C, Cout, kd, kh, kw, *padding, *extra_pad)
    if Cin != C or C % groups != 0 or not _lib.query("diqt_conv3d_fwd_gn_supported", *geo, act):
        return None
    _chk(x, gamma, beta, weight, bias, residual, scale_shift.base if isinstance(scale_shift, SSView) else None)
    rows = D * H * W
    dev = x.device
    s = _stream()
    mean = torch.empty(B * groups, dtype=torch.float32, device=dev)
    rstd = torch.empty_like(mean)
    coef = torch.empty(2 * B * C, dtype=torch.float32, device=dev)
    scale = shift = None
    cs = 0
    if isinstance(scale_shift, SSView):
        assert scale_shift.width == 2 * C and scale_shift.base.shape[0] == B
        scale, cs = scale_shift.base.data_ptr() + 4 * scale_shift.off, scale_shift.base.shape[1]
        shift = scale + 4 * C
    elif scale_shift is not None:
        assert scale_shift.shape == (B, 2 * C) and scale_shift.stride(1) == 1 and scale_shift.is_cuda and scale_shift.dtype == torch.float32
        scale, shift, cs = scale_shift.data_ptr(), scale_shift.data_ptr() + 4 * C, scale_shift.stride(0)
    pre = getattr(x, "_diqt_stats", None)
    if pre is not None and pre.rows == rows and pre.partials.shape == (B, pre.nblk, 2, C):
        _lib.call("diqt_gn_coef_from_partials", pre.partials, pre.nblk, rows, gamma, beta, scale, shift, cs, mean, rstd, coef, B, C, groups,
                  float(eps), s)
    else:
        ws, n = _reduce_ws(B, C, dev)
        _lib.call("diqt_groupnorm_stats_coef", x, gamma, beta, scale, shift, cs, mean, rstd, coef, ws, n, B, rows, C, groups, float(eps), s)
    Do, Ho, Wo = D + 2 * padding[0] + extra_pad[0] - kd + 1, H + 2 * padding[1] + extra_pad[1] - kh + 1, W + 2 * padding[2] + extra_pad[2] - kw + 1
    y = torch.empty((B, Do, Ho, Wo, Cout), dtype=torch.float32, device=dev)
    if TIMER.enabled:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    n = _lib.query("diqt_conv3d_fwd_workspace_bytes", *geo)
    ws = _workspace(n, dev) if n else None
    stats = None
    if want_stats:
        nblk = _lib.query("diqt_conv3d_fwd_stats_blocks", *geo)
        if nblk > 0:
            stats = torch.empty((B, nblk, 2, Cout), dtype=torch.float32, device=dev)
            y._diqt_stats = ColStats(stats, nblk, Do * Ho * Wo)
    _lib.call("diqt_conv3d_fwd_gn", x, _packed(weight, 0), bias, residual, y, stats, ws, n, coef, act, *geo, s)
    if TIMER.enabled:
        ev1.record()
        TIMER.records.append((ev0, ev1, 2.0 * B * Do * Ho * Wo * Cout * C * kd * kh * kw, "conv_fwd9_kernel", geo[:9]))
    return y


class _ActFn(Function):
    @staticmethod
    def forward(ctx, x, act):
        _chk(x)
        y = torch.empty_like(x)
        _lib.call("diqt_act_fwd", x, y, x.numel(), act, _stream())
        ctx.save_for_backward(x)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        _lib.call("diqt_act_bwd", x, dy.contiguous(), dx, x.numel(), ctx.act, _stream())
        return dx, None


def activation(x, act):
    return _ActFn.apply(x.contiguous(), act)


def mish(x):
    return activation(x, ACT_MISH)


class _ChanLayerNormFn(Function):
    @staticmethod
    def forward(ctx, x, g, b, eps, res=None, tap=False):
        """tap: as _GnActFn.forward -- x comes back as a second output for its other consumer (the `+ x` of the block)."""
        _chk(x, g, b, res)
        ctx.tap = tap
        ctx.set_materialize_grads(False)
        C = x.shape[-1]
        rows = x.numel() // C
        y = torch.empty_like(x)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        _lib.call("diqt_chan_layernorm_fwd_res", x, g, b, res, y, mean, rstd, rows, C, float(eps), _stream())
        ctx.save_for_backward(x, g, mean, rstd)
        ctx.has_bias = b is not None
        ctx.has_res = res is not None
        return (y, x) if tap else y

    @staticmethod
    def backward(ctx, dy, dtap=None):
        x, g, mean, rstd = ctx.saved_tensors
        if dy is None:
            return dtap, None, None, None, None, None
        C = x.shape[-1]
        rows = x.numel() // C
        dx = torch.empty_like(x)
        dg = torch.empty_like(g)
        db = torch.empty_like(g) if ctx.has_bias else None
        ws, n = _reduce_ws(1, C, x.device)
        dy = dy.contiguous()
        if dtap is not None:
            dtap = dtap.contiguous()
        _lib.call("diqt_chan_layernorm_bwd_ex", x, dy, dtap, g, mean, rstd, dx, dg, db, ws, n, rows, C, _stream())
        return dx, dg, db, None, (dy if ctx.has_res else None), None


def chan_layernorm(x, g, eps=1e-5, bias=None, residual=None, tap=False):
    """LayerNorm over the channel (last) axis; g (and the optional bias) are any tensors with C elements.  ``residual`` (same shape as
    x) is added to the result in the same pass.  ``tap=True`` returns ``(y, x_alias)`` (see _GnActFn.forward)."""
    if residual is not None:
        assert residual.shape == x.shape, (tuple(residual.shape), tuple(x.shape))
        residual = residual.contiguous()
    xc = x.contiguous()
    use_tap = tap and torch.is_grad_enabled() and xc.requires_grad
    out = _ChanLayerNormFn.apply(xc, g.reshape(-1), bias.reshape(-1) if bias is not None else None, eps, residual, use_tap)
    y, alias = out if use_tap else (out, x)
    y = y.view(x.shape)
    return (y, alias.view(x.shape)) if tap else y


# --------------------------------------------------------------------------------------------
# learned sinusoidal embedding
# --------------------------------------------------------------------------------------------
class _LearnedSinuFn(Function):
    @staticmethod
    def forward(ctx, t, w):
        _chk(t, w)
        B, half = t.shape[0], w.shape[0]
        out = torch.empty((B, 2 * half + 1), dtype=torch.float32, device=t.device)
        _lib.call("diqt_learned_sinu_fwd", t, w, out, B, half, _stream())
        ctx.save_for_backward(t, w)
        return out

    @staticmethod
    def backward(ctx, dout):
        t, w = ctx.saved_tensors
        dw = torch.empty_like(w)
        _lib.call("diqt_learned_sinu_bwd", t, w, dout.contiguous(), dw, t.shape[0], w.shape[0], _stream())
        return None, dw


def learned_sinusoidal(t, w):
    return _LearnedSinuFn.apply(t.contiguous(), w)


# --------------------------------------------------------------------------------------------
# squeeze-excite gate + residual (one autograd node for pool -> MLP -> h*gate + res)
# --------------------------------------------------------------------------------------------
class _SEResidualFn(Function):
    @staticmethod
    def forward(ctx, h, w1, w2, res, pre=None, stats_out=None):
        h_half = h.dtype != torch.float32      # block2's conv output in the operand type (autocast sampling / low-precision training)
        hty = 0 if not h_half else (2 if h.dtype == torch.bfloat16 else 1)
        _chk(None if h_half else h, w1, w2, res)
        B, C = h.shape[0], h.shape[-1]
        rows = h.numel() // (B * C)
        Cr = w1.shape[0]
        dev = h.device
        s = _stream()
        pooled = torch.empty((B, C), dtype=torch.float32, device=dev)
        hidden = torch.empty((B, Cr), dtype=torch.float32, device=dev)
        gate = torch.empty((B, C), dtype=torch.float32, device=dev)
        fused = 256 % min(C, 256) == 0 and (C <= 256 or C % 256 == 0)
        if fused and pre is not None and pre.rows == rows and pre.partials.shape == (B, pre.nblk, 2, C):
            # channel means from the producing conv's per-tile column sums + the two-layer gate MLP in ONE launch
            _lib.call("diqt_se_pool_mlp_fwd", pre.partials, pre.nblk, rows, pooled, w1, w2, hidden, gate, B, C, Cr, s)
        else:
            if pre is not None and pre.rows == rows and pre.partials.shape == (B, pre.nblk, 2, C):
                _lib.call("diqt_channel_mean_from_partials", pre.partials, pooled, B, pre.nblk, rows, C, s)
            else:
                assert not h_half, "a 16-bit conv output must carry its column sums (the SE squeeze reads them)"
                ws, n = _reduce_ws(B, C, dev)
                _lib.call("diqt_channel_mean", h, pooled, ws, n, B, rows, C, s)
            _lib.call("diqt_se_pool_mlp_fwd", None, 0, 0, pooled, w1, w2, hidden, gate, B, C, Cr, s)
        y = torch.empty(h.shape, dtype=torch.float32, device=dev)
        nblk = _lib.query("diqt_gate_residual_stats_blocks", rows, C) if stats_out is not None else 0
        if nblk > 0:
            # the block's output feeds the next block's first GroupNorm: its per-workgroup column sums ride along (no statistics pass)
            stats = torch.empty((B, nblk, 2, C), dtype=torch.float32, device=dev)
            if h_half:
                _lib.call("diqt_gate_residual_fwd_stats_h", h, gate, res, y, stats, B, rows, C, hty, s)
            else:
                _lib.call("diqt_gate_residual_fwd_stats", h, gate, res, y, stats, B, rows, C, s)
            stats_out.append(ColStats(stats, nblk, rows))
        elif h_half:
            _lib.call("diqt_gate_residual_fwd_h", h, gate, res, None, 0.0, y, B, rows, C, hty, 0, s)
        else:
            _lib.call("diqt_gate_residual_fwd", h, gate, res, None, 0.0, y, B, rows, C, s)
        ctx.save_for_backward(h, w1, w2, pooled, hidden, gate)
        ctx.has_res = res is not None
        ctx.hty = hty
        return y

    @staticmethod
    def backward(ctx, dy):
        h, w1, w2, pooled, hidden, gate = ctx.saved_tensors
        dy = dy.contiguous()
        B, C = h.shape[0], h.shape[-1]
        rows = h.numel() // (B * C)
        Cr = w1.shape[0]
        dev = h.device
        s = _stream()
        dgate = torch.empty_like(gate)
        ws, n = _reduce_ws(B, C, dev)
        if ctx.hty:
            _lib.call("diqt_gate_residual_bwd_h", h, dy, dgate, ws, n, B, rows, C, ctx.hty, s)
        else:
            _lib.call("diqt_gate_residual_bwd", h, dy, dgate, ws, n, B, rows, C, s)
        dpooled = torch.empty_like(pooled)
        dw1, dw2 = torch.empty_like(w1), torch.empty_like(w2)
        scratch = torch.empty(B * (C + Cr), dtype=torch.float32, device=dev)
        _lib.call("diqt_se_mlp_bwd", pooled, w1, w2, hidden, gate, dgate, dpooled, dw1, dw2, scratch, B, C, Cr, s)
        dh = torch.empty_like(h)                       # (a 16-bit h gets a 16-bit gradient: only 16-bit-operand conv kernels read it)
        if ctx.hty:
            _lib.call("diqt_gate_residual_fwd_h", dy, gate, None, dpooled, 1.0 / rows, dh, B, rows, C, 0, ctx.hty, s)
        else:
            _lib.call("diqt_gate_residual_fwd", dy, gate, None, dpooled, 1.0 / rows, dh, B, rows, C, s)
        return dh, dw1, dw2, (dy if ctx.has_res else None), None, None


def se_gate_residual(h, w1, w2, res=None):
    """h * sigmoid(relu(mean(h) w1^T) w2^T) + res."""
    holder = []
    y = _SEResidualFn.apply(h, w1, w2, res, getattr(h, "_diqt_stats", None), holder)
    if holder:
        y._diqt_stats = holder[0]           # consumed by groupnorm_act on this exact tensor
    return y


class _AddFn(Function):
    """a + b through the gate kernel with unit gates is wasteful; use axpby3 with unit coefficients."""
    @staticmethod
    def forward(ctx, a, b):
        _chk(a, b)
        out = torch.empty_like(a)
        one = _ones(1, a.device)
        _lib.call("diqt_axpby3", a.view(1, -1), b.view(1, -1), None, one, one, None, 0.0, 0.0, 0, out, 1, a.numel(),
                  _stream())
        return out

    @staticmethod
    def backward(ctx, d):
        return d, d


_ONES = {}


def _ones(n, device):
    key = (n, device.index)
    t = _ONES.get(key)
    if t is None:
        t = torch.ones(n, dtype=torch.float32, device=device)
        _ONES[key] = t
    return t


def add(a, b):
    return _AddFn.apply(a.contiguous(), b.contiguous())


# --------------------------------------------------------------------------------------------
# data movement
# --------------------------------------------------------------------------------------------
class _SpaceToDepthFn(Function):
    @staticmethod
    def forward(ctx, x):
        _chk(x)
        B, D2, H2, W2, C = x.shape
        y = torch.empty((B, D2 // 2, H2 // 2, W2 // 2, C * 8), dtype=torch.float32, device=x.device)
        _lib.call("diqt_space_to_depth2", x, y, B, D2 // 2, H2 // 2, W2 // 2, C, _stream())
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        B, D, H, W, C8 = dy.shape
        dx = torch.empty((B, 2 * D, 2 * H, 2 * W, C8 // 8), dtype=torch.float32, device=dy.device)
        _lib.call("diqt_depth_to_space2", dy, dx, B, D, H, W, C8 // 8, _stream())
        return dx


class _DepthToSpaceFn(Function):
    @staticmethod
    def forward(ctx, x):
        _chk(x)
        B, D, H, W, C8 = x.shape
        y = torch.empty((B, 2 * D, 2 * H, 2 * W, C8 // 8), dtype=torch.float32, device=x.device)
        _lib.call("diqt_depth_to_space2", x, y, B, D, H, W, C8 // 8, _stream())
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        B, D2, H2, W2, C = dy.shape
        dx = torch.empty((B, D2 // 2, H2 // 2, W2 // 2, C * 8), dtype=torch.float32, device=dy.device)
        _lib.call("diqt_space_to_depth2", dy, dx, B, D2 // 2, H2 // 2, W2 // 2, C, _stream())
        return dx


def space_to_depth(x):
    return _SpaceToDepthFn.apply(x)


def depth_to_space(x):
    return _DepthToSpaceFn.apply(x)


class _ConcatFn(Function):
    @staticmethod
    def forward(ctx, a, b, fa, fb):
        _chk(a, b)
        Ca, Cb = a.shape[-1], b.shape[-1]
        rows = a.numel() // Ca
        y = torch.empty((*a.shape[:-1], Ca + Cb), dtype=torch.float32, device=a.device)
        if fa == 1.0 and fb == 1.0 and (Ca % 4 or Cb % 4):
            _lib.call("diqt_concat_channels", a, Ca, b, Cb, y, rows, _stream())
        else:
            _lib.call("diqt_concat_channels_scaled", a, Ca, b, Cb, fa, fb, y, rows, _stream())
        ctx.shapes = (a.shape, b.shape)
        ctx.f = (fa, fb)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        sa, sb = ctx.shapes
        da = torch.empty(sa, dtype=torch.float32, device=dy.device) if ctx.needs_input_grad[0] else None
        db = torch.empty(sb, dtype=torch.float32, device=dy.device) if ctx.needs_input_grad[1] else None
        if da is not None or db is not None:
            _lib.call("diqt_split_channels_scaled", dy, da, sa[-1], db, sb[-1], ctx.f[0], ctx.f[1], dy.numel() // dy.shape[-1], _stream())
        return da, db, None, None


def concat_channels(a, b, scale_a=1.0, scale_b=1.0, want_stats=False):
    """cat(scale_a * a, scale_b * b) over the last axis in one pass (the scaled skip connections).  ``want_stats`` (sampling path): the
    pass also writes the column sums of its output for the GroupNorm that consumes it (``_diqt_stats``), when the shape is taken."""
    if want_stats and not torch.is_grad_enabled() and a.dim() >= 3 and a.dtype == torch.float32 and b.dtype == torch.float32:
        a, b = a.contiguous(), b.contiguous()
        Ca, Cb, B = a.shape[-1], b.shape[-1], a.shape[0]
        rows = a.numel() // (B * Ca)
        nblk = _lib.query("diqt_concat_channels_stats_blocks", Ca, Cb, rows)
        if nblk > 0 and b.shape[:-1] == a.shape[:-1]:
            _chk(a, b)
            y = torch.empty((*a.shape[:-1], Ca + Cb), dtype=torch.float32, device=a.device)
            stats = torch.empty((B, nblk, 2, Ca + Cb), dtype=torch.float32, device=a.device)
            _lib.call("diqt_concat_channels_stats", a, Ca, b, Cb, float(scale_a), float(scale_b), y, B, rows, stats, _stream())
            y._diqt_stats = ColStats(stats, nblk, rows)
            return y
    return _ConcatFn.apply(a.contiguous(), b.contiguous(), float(scale_a), float(scale_b))


class _SplitFn(Function):
    @staticmethod
    def forward(ctx, x, ca):
        _chk(x)
        C = x.shape[-1]
        rows = x.numel() // C
        a = torch.empty((*x.shape[:-1], ca), dtype=torch.float32, device=x.device)
        b = torch.empty((*x.shape[:-1], C - ca), dtype=torch.float32, device=x.device)
        _lib.call("diqt_split_channels", x, a, ca, b, C - ca, rows, _stream())
        ctx.set_materialize_grads(False)
        ctx.cfg = (ca, C - ca, a.shape, b.shape)
        return a, b

    @staticmethod
    def backward(ctx, da, db):
        ca, cb, sa, sb = ctx.cfg
        ref = da if da is not None else db
        if ref is None:
            return None, None
        da = da.contiguous() if da is not None else torch.zeros(sa, dtype=torch.float32, device=ref.device)
        db = db.contiguous() if db is not None else torch.zeros(sb, dtype=torch.float32, device=ref.device)
        dx = torch.empty((*sa[:-1], ca + cb), dtype=torch.float32, device=ref.device)
        _lib.call("diqt_concat_channels", da, ca, db, cb, dx, da.numel() // ca, _stream())
        return dx, None


def split_channels(x, ca):
    """(x[..., :ca], x[..., ca:]) as two contiguous tensors (the inverse of ``concat_channels``)."""
    return _SplitFn.apply(x.contiguous(), int(ca))


class _SubvolumeFn(Function):
    """merged volume [1,S,S,S,C] -> sub-volume batch [f^3,A',A',A',C] (halo>0: zero-padded overlap)."""
    @staticmethod
    def forward(ctx, vol, f, A, halo):
        _chk(vol)
        C = vol.shape[-1]
        Ap = A + 2 * halo
        sub = torch.empty((f ** 3, Ap, Ap, Ap, C), dtype=torch.float32, device=vol.device)
        _lib.call("diqt_subvolume_gather", vol, sub, f, A, C, halo, _stream())
        ctx.cfg = (f, A, C, halo, vol.shape)
        return sub

    @staticmethod
    def backward(ctx, dsub):
        f, A, C, halo, shape = ctx.cfg
        dvol = torch.zeros(shape, dtype=torch.float32, device=dsub.device)
        _lib.call("diqt_subvolume_scatter", dsub.contiguous(), dvol, f, A, C, halo, 1 if halo > 0 else 0, _stream())
        return dvol, None, None, None


class _MergeFn(Function):
    @staticmethod
    def forward(ctx, sub, f):
        _chk(sub)
        n, A, _, _, C = sub.shape
        vol = torch.empty((1, f * A, f * A, f * A, C), dtype=torch.float32, device=sub.device)
        _lib.call("diqt_subvolume_scatter", sub, vol, f, A, C, 0, 0, _stream())
        ctx.cfg = (f, A, C)
        return vol

    @staticmethod
    def backward(ctx, dvol):
        f, A, C = ctx.cfg
        dsub = torch.empty((f ** 3, A, A, A, C), dtype=torch.float32, device=dvol.device)
        _lib.call("diqt_subvolume_gather", dvol.contiguous(), dsub, f, A, C, 0, _stream())
        return dsub, None


class _TrilinearUpFn(Function):
    @staticmethod
    def forward(ctx, x, scale):
        _chk(x)
        B, D, H, W, C = x.shape
        y = torch.empty((B, D * scale, H * scale, W * scale, C), dtype=torch.float32, device=x.device)
        _lib.call("diqt_trilinear_up_fwd", x, y, B, D, H, W, C, scale, _stream())
        ctx.cfg = (B, D, H, W, C, scale)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, D, H, W, C, scale = ctx.cfg
        dx = torch.zeros((B, D, H, W, C), dtype=torch.float32, device=dy.device)
        _lib.call("diqt_trilinear_up_bwd", dy.contiguous(), dx, B, D, H, W, C, scale, _stream())
        return dx, None


def trilinear_upsample(x, scale):
    """nn.Upsample(scale_factor=scale, mode='trilinear', align_corners=True) on channels-last x."""
    return _TrilinearUpFn.apply(x.contiguous(), int(scale))


def split_volume(vol, f, A, halo=0):
    return _SubvolumeFn.apply(vol.contiguous(), f, A, halo)


def merge_volume(sub, f):
    return _MergeFn.apply(sub.contiguous(), f)


# --------------------------------------------------------------------------------------------
# attention pieces
# --------------------------------------------------------------------------------------------
class _SoftmaxFn(Function):
    @staticmethod
    def forward(ctx, x, outer, n, inner, scale):
        _chk(x)
        y = torch.empty_like(x)
        _lib.call("diqt_softmax_fwd", x, y, outer, n, inner, float(scale), _stream())
        ctx.save_for_backward(y)
        ctx.cfg = (outer, n, inner, float(scale))
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dx = torch.empty_like(y)
        _lib.call("diqt_softmax_bwd", y, dy.contiguous(), dx, *ctx.cfg, _stream())
        return dx, None, None, None, None


def softmax(x, dim, scale=1.0):
    """scale * softmax(x, dim) for a contiguous x."""
    x = x.contiguous()
    dim = dim % x.dim()
    outer = 1
    for s in x.shape[:dim]:
        outer *= s
    inner = 1
    for s in x.shape[dim + 1:]:
        inner *= s
    return _SoftmaxFn.apply(x, outer, x.shape[dim], inner, scale)


def _bgemm_strided(A, Bm, C, g, M, N, K, tA, tB, sA, lda, sB, ldb, sC, ldc, alpha, oA=0, oB=0, oC=0):
    nws = _lib.query("diqt_bgemm_workspace_bytes", g, M, N, K)
    ws = _workspace(nws, A.device) if nws else None
    _lib.call("diqt_bgemm_ws", A.data_ptr() + 4 * oA, Bm.data_ptr() + 4 * oB, C.data_ptr() + 4 * oC, ws, nws, g, M, N, K, int(tA),
              int(tB), sA, sB, sC, lda, ldb, ldc, float(alpha), 0.0, _stream())


class _BmmStridedFn(Function):
    """C[g] = alpha * op(A[g]) op(B[g]) with explicit batch strides / leading dimensions / element offsets, so attention
    heads (and the k|v halves of a fused projection) are addressed inside channels-last tensors without copies.
    spec = (g, M, N, K, tA, tB, sA, lda, sB, ldb, sC, ldc, alpha, out_shape[, offA, offB])."""
    @staticmethod
    def forward(ctx, A, Bm, spec):
        _chk(A, Bm)
        g, M, N, K, tA, tB, sA, lda, sB, ldb, sC, ldc, alpha, out_shape = spec[:14]
        oA, oB = (spec[14], spec[15]) if len(spec) > 14 else (0, 0)
        C = torch.empty(out_shape, dtype=torch.float32, device=A.device)
        _bgemm_strided(A, Bm, C, g, M, N, K, tA, tB, sA, lda, sB, ldb, sC, ldc, alpha, oA, oB)
        ctx.save_for_backward(A, Bm)
        ctx.spec = spec
        return C

    @staticmethod
    def backward(ctx, dC):
        A, Bm = ctx.saved_tensors
        spec = ctx.spec
        g, M, N, K, tA, tB, sA, lda, sB, ldb, sC, ldc, alpha, _ = spec[:14]
        oA, oB = (spec[14], spec[15]) if len(spec) > 14 else (0, 0)
        dC = dC.contiguous()
        dA = dB = None
        # operands addressed through offsets / interleaved strides cover only part of their tensor: zero-fill then
        partial_A = oA != 0 or len(spec) > 14
        if ctx.needs_input_grad[0]:
            dA = torch.zeros_like(A) if partial_A else torch.empty_like(A)
            if not tA:   # dA[M,K] = a * dC[M,N] op(B)^T
                _bgemm_strided(dC, Bm, dA, g, M, K, N, False, not tB, sC, ldc, sB, ldb, sA, lda, alpha, 0, oB, oA)
            else:        # A stored [K,M]: dA[K,M] = a * op(B)[K,N] dC^T[N,M]
                _bgemm_strided(Bm, dC, dA, g, K, M, N, tB, True, sB, ldb, sC, ldc, sA, lda, alpha, oB, 0, oA)
        if ctx.needs_input_grad[1]:
            dB = torch.zeros_like(Bm) if partial_A else torch.empty_like(Bm)
            if not tB:   # dB[K,N] = a * op(A)^T[K,M] dC[M,N]
                _bgemm_strided(A, dC, dB, g, K, N, M, not tA, False, sA, lda, sC, ldc, sB, ldb, alpha, oA, 0, oB)
            else:        # B stored [N,K]: dB[N,K] = a * dC^T[N,M] op(A)[M,K]
                _bgemm_strided(dC, A, dB, g, N, K, M, True, tA, sC, ldc, sA, lda, sB, ldb, alpha, 0, oA, oB)
        return dA, dB, None


def bmm_strided(A, Bm, spec):
    return _BmmStridedFn.apply(A.contiguous(), Bm.contiguous(), spec)


def _bgemm_raw(A, Bm, tA, tB, alpha=1.0):
    """A:[g,M,K] (or [g,K,M] if tA), B:[g,K,N] (or [g,N,K] if tB) -> [g,M,N]; contiguous inputs."""
    g = A.shape[0]
    M, K = (A.shape[2], A.shape[1]) if tA else (A.shape[1], A.shape[2])
    N = Bm.shape[1] if tB else Bm.shape[2]
    C = torch.empty((g, M, N), dtype=torch.float32, device=A.device)
    nws = _lib.query("diqt_bgemm_workspace_bytes", g, M, N, K)
    ws = _workspace(nws, A.device) if nws else None
    _lib.call("diqt_bgemm_ws", A, Bm, C, ws, nws, g, M, N, K, int(tA), int(tB), A.shape[1] * A.shape[2], Bm.shape[1] * Bm.shape[2],
              M * N, A.shape[2], Bm.shape[2], N, float(alpha), 0.0, _stream())
    return C


class _BmmFn(Function):
    @staticmethod
    def forward(ctx, A, Bm, tA, tB, alpha):
        _chk(A, Bm)
        ctx.save_for_backward(A, Bm)
        ctx.cfg = (tA, tB, alpha)
        return _bgemm_raw(A, Bm, tA, tB, alpha)

    @staticmethod
    def backward(ctx, dC):
        A, Bm = ctx.saved_tensors
        tA, tB, alpha = ctx.cfg
        dC = dC.contiguous()
        dA = dB = None
        # C = a * op(A) op(B)
        if ctx.needs_input_grad[0]:
            if not tA:   # dA = a * dC op(B)^T
                dA = _bgemm_raw(dC, Bm, False, not tB, alpha)
            else:        # dA^T = a*dC op(B)^T -> dA = a * op(B) dC^T
                dA = _bgemm_raw(Bm, dC, tB, True, alpha)
        if ctx.needs_input_grad[1]:
            if not tB:   # dB = a * op(A)^T dC
                dB = _bgemm_raw(A, dC, not tA, False, alpha)
            else:        # dB = a * dC^T op(A)
                dB = _bgemm_raw(dC, A, True, tA, alpha)
        return dA, dB, None, None, None


class _WeightedPoolFn(Function):
    """out[b, c] = sum_n w[b, n] x[b, n, c] (GlobalContext pooling): one column-reduction pass over x instead of a
    1-row GEMM with K = n; the gradients are two thin GEMMs (dw = x dout, dx = w (x) dout)."""
    @staticmethod
    def forward(ctx, w, x):
        _chk(w, x)
        B, n, C = x.shape
        out = torch.empty((B, C), dtype=torch.float32, device=x.device)
        ws, nws = _reduce_ws(B, C, x.device)
        _lib.call("diqt_weighted_colsum", x, w, out, ws, nws, B, n, C, _stream())
        ctx.save_for_backward(w, x)
        return out

    @staticmethod
    def backward(ctx, dout):
        w, x = ctx.saved_tensors
        B, n, C = x.shape
        dout = dout.contiguous()
        dw = dx = None
        if ctx.needs_input_grad[0]:
            dw = _bgemm_raw(x, dout.reshape(B, C, 1), False, False).reshape(B, n)
        if ctx.needs_input_grad[1]:
            dx = _bgemm_raw(w.reshape(B, n, 1), dout.reshape(B, 1, C), False, False)
        return dw, dx


def softmax_pool_nograd(x, w):
    """GlobalContext pooling in one pass (sampling path): x[B, n, C], w[C] -> pooled[B, C] = sum_n softmax_n(x . w)[n] x[n]; None when
    the shape is not taken or autograd is recording."""
    if torch.is_grad_enabled():
        return None
    B, n, C = x.shape
    if not _lib.query("diqt_softmax_pool_supported", B, n, C):
        return None
    x, w = x.contiguous(), w.contiguous()
    _chk(x, w)
    pooled = torch.empty((B, C), dtype=torch.float32, device=x.device)
    ws, nb = _reduce_ws(B, C, x.device)
    _lib.call("diqt_softmax_pool", x, w, pooled, ws, nb, B, n, C, _stream())
    return pooled


def weighted_pool(w, x):
    """w: [B, n] weights, x: [B, n, C] -> [B, C]"""
    return _WeightedPoolFn.apply(w.contiguous(), x.contiguous())


def bmm(A, Bm, transA=False, transB=False, alpha=1.0):
    return _BmmFn.apply(A.contiguous(), Bm.contiguous(), bool(transA), bool(transB), float(alpha))


class _ShuffleNdFn(Function):
    @staticmethod
    def forward(ctx, x, factors, to_space):
        _chk(x)
        sd, sh, sw = factors
        S = sd * sh * sw
        if to_space:
            B, D, H, W, CS = x.shape
            C = CS // S
            y = torch.empty((B, D * sd, H * sh, W * sw, C), dtype=torch.float32, device=x.device)
            _lib.call("diqt_depth_to_space_nd", x, y, B, D, H, W, C, sd, sh, sw, _stream())
        else:
            B, D2, H2, W2, C = x.shape
            D, H, W = D2 // sd, H2 // sh, W2 // sw
            y = torch.empty((B, D, H, W, C * S), dtype=torch.float32, device=x.device)
            _lib.call("diqt_space_to_depth_nd", x, y, B, D, H, W, C, sd, sh, sw, _stream())
        ctx.cfg = (factors, to_space)
        return y

    @staticmethod
    def backward(ctx, dy):
        factors, to_space = ctx.cfg
        return _ShuffleNdFn.apply(dy.contiguous(), factors, not to_space), None, None


def space_to_depth_nd(x, factors):
    """'b (d sd) (h sh) (w sw) c -> b d h w (c sd sh sw)' for factors in {1,2}."""
    return _ShuffleNdFn.apply(x.contiguous(), tuple(factors), False)


def depth_to_space_nd(x, factors):
    return _ShuffleNdFn.apply(x.contiguous(), tuple(factors), True)


class _TransposeMidFn(Function):
    @staticmethod
    def forward(ctx, x):
        _chk(x)
        A, M, N, C = x.shape
        y = torch.empty((A, N, M, C), dtype=torch.float32, device=x.device)
        _lib.call("diqt_transpose_mid", x, y, A, M, N, C, _stream())
        return y

    @staticmethod
    def backward(ctx, dy):
        return _TransposeMidFn.apply(dy.contiguous())


def transpose_mid(x):
    """[A, M, N, C] -> [A, N, M, C]."""
    return _TransposeMidFn.apply(x.contiguous())


class _NearestResizeFn(Function):
    @staticmethod
    def forward(ctx, x, size):
        _chk(x)
        B, D, H, W, C = x.shape
        Do, Ho, Wo = size
        y = torch.empty((B, Do, Ho, Wo, C), dtype=torch.float32, device=x.device)
        _lib.call("diqt_nearest_resize", x, y, B, D, H, W, C, Do, Ho, Wo, _stream())
        ctx.cfg = (B, D, H, W, C, Do, Ho, Wo)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, D, H, W, C, Do, Ho, Wo = ctx.cfg
        dx = torch.empty((B, D, H, W, C), dtype=torch.float32, device=dy.device)
        _lib.call("diqt_nearest_resize_bwd", dy.contiguous(), dx, B, D, H, W, C, Do, Ho, Wo, _stream())       # whole-number up-scaling only
        return dx, None


def nearest_resize(x, size):
    """F.interpolate(mode='nearest') of a channels-last volume to (Do, Ho, Wo).  Differentiable for whole-number up-scaling factors
    (the feature maps of UpsampleCombiner); other sizes are data preparation only."""
    return _NearestResizeFn.apply(x.contiguous(), tuple(int(v) for v in size))


class _L2NormRowsFn(Function):
    """F.normalize(dim=-1) of the d-wide rows of x[..., stride] taken at column offset ``off`` (rows ``stride`` floats apart)."""
    @staticmethod
    def forward(ctx, x, off, d):
        _chk(x)
        stride = x.shape[-1]
        rows = x.numel() // stride
        y = torch.empty(x.shape[:-1] + (d,), dtype=torch.float32, device=x.device)
        inv = torch.empty(rows, dtype=torch.float32, device=x.device)
        _lib.call("diqt_l2norm_rows_fwd", x.data_ptr() + 4 * off, y, inv, rows, d, stride, d, _stream())
        ctx.save_for_backward(y, inv)
        ctx.cfg = (off, d, stride, rows, tuple(x.shape))
        return y

    @staticmethod
    def backward(ctx, dy):
        y, inv = ctx.saved_tensors
        off, d, stride, rows, shape = ctx.cfg
        dx = torch.zeros(shape, dtype=torch.float32, device=dy.device) if d != stride else torch.empty(shape, dtype=torch.float32, device=dy.device)
        _lib.call("diqt_l2norm_rows_bwd", y, dy.contiguous(), inv, dx.data_ptr() + 4 * off, rows, d, d, stride, _stream())
        return dx, None, None


def l2norm_rows(x, off=0, d=None):
    """l2-normalised copy of columns [off, off + d) of every row of x[..., C] (cosine-sim attention: q heads, the k half of k|v rows)."""
    d = x.shape[-1] if d is None else d
    return _L2NormRowsFn.apply(x.contiguous(), int(off), int(d))


class _AttnSoftmaxFn(Function):
    @staticmethod
    def forward(ctx, sim, rel, null_bias, n, h, n_extra, n_self, causal):
        _chk(sim, rel, null_bias)
        G = sim.numel() // (n * h * (n_extra + n_self))
        p = torch.empty_like(sim)
        _lib.call("diqt_attn_softmax_fwd", sim, rel, null_bias, p, G, n, h, n_extra, n_self, int(causal), _stream())
        ctx.save_for_backward(p)
        ctx.cfg = (G, n, h, n_extra, n_self, int(causal), rel is not None, null_bias is not None)
        ctx.shapes = (rel.shape if rel is not None else None, null_bias.shape if null_bias is not None else None)
        return p

    @staticmethod
    def backward(ctx, dp):
        (p,) = ctx.saved_tensors
        G, n, h, n_extra, n_self, causal, has_rel, has_null = ctx.cfg
        dsim = torch.empty_like(p)
        drel = torch.zeros(ctx.shapes[0], dtype=torch.float32, device=p.device) if has_rel else None
        dnull = torch.zeros(ctx.shapes[1], dtype=torch.float32, device=p.device) if has_null else None
        nws = _lib.query("diqt_attn_softmax_bwd_workspace_bytes", G, n, h, n_extra, n_self)
        ws = _workspace(nws, p.device) if nws else None
        _lib.call("diqt_attn_softmax_bwd_ws", p, dp.contiguous(), dsim, drel, dnull, ws, nws, G, n, h, n_extra, n_self, causal,
                  _stream())
        return dsim, drel, dnull, None, None, None, None, None


def mqa_attention_nograd(q, kv_ext, rel, null_bias, n, h, d, n_extra, n_self, causal, scale):
    """Fused multi-query attention forward (no autograd: sampling path).  q: [G, n, h*d]; kv_ext: [G, n_extra + n_self, 2d]."""
    _chk(q, kv_ext, rel, null_bias)
    G = q.shape[0]
    out = torch.empty((G, n, h * d), dtype=torch.float32, device=q.device)
    lp = lp_mode()
    if lp is not None:          # autocast: q k^T and p v on the fp16 / bf16 MFMA, soft-max statistics in fp32
        kv_h = torch.empty(kv_ext.shape, dtype=torch.int16, device=q.device)
        _lib.call("diqt_cast_to_h", kv_ext, kv_h, kv_ext.numel(), lp, _stream())
        _lib.call("diqt_mqa_attention_fwd_h", q, kv_h, rel, null_bias, out, G, n, h, d, n_extra, n_self, int(causal), float(scale),
                  lp, 1, _stream())
        return out
    _lib.call("diqt_mqa_attention_fwd", q, kv_ext, rel, null_bias, out, G, n, h, d, n_extra, n_self, int(causal), float(scale),
              _stream())
    return out


def mqa_attention_frames_nograd(q, kv, null_kv, rel, null_bias, h, d, causal, scale):
    """The temporal attention of the pseudo-3D U-Net in place (sampling path, fp32): q[B, F, P, h*d], kv[B, F, P, 2d] channels-last,
    sequences = (b, pixel), tokens = frames, ``null_kv``[2d] the extra key / value.  Returns out[B, F, P, h*d] -- no transposes, no
    concatenated copy of kv."""
    _chk(q, kv, null_kv, rel, null_bias)
    B, F, P, _ = q.shape
    out = torch.empty_like(q)
    _lib.call("diqt_mqa_attention_fwd_frames", q, kv, null_kv, rel, null_bias, out, B, F, P, h, d, int(causal), float(scale), _stream())
    return out


def temporal_attention_h_ok(B, F, P, C, h, d):
    return bool(_lib.query("diqt_temporal_attention_h_supported", B, F, P, C, h, d))


def pack_temporal_attention_h(to_q_w, to_kv_w, to_out_w, h, d, scale, lp):
    """16-bit operand copies of an Attention block's projections for diqt_temporal_attention_h: scale * to_q.weight [h d, C],
    to_kv.weight [2 d, C], and to_out.weight [C, h d] as [h][C][d] with a head's channels in the MFMA accumulator order."""
    dt = torch.bfloat16 if lp == 1 else torch.float16
    C = to_out_w.shape[0]
    wq = (to_q_w.detach().float() * scale).to(dt).contiguous()
    wkv = to_kv_w.detach().to(dt).contiguous()
    i = torch.arange(d, device=to_out_w.device)
    pos = 16 * (i >> 4) + 8 * ((i >> 2) & 1) + 4 * ((i >> 3) & 1) + (i & 3)
    wo = torch.empty((h, C, d), dtype=dt, device=to_out_w.device)
    wo[:, :, pos] = to_out_w.detach().reshape(C, h, d).permute(1, 0, 2).to(dt)
    return wq.view(torch.int16), wkv.view(torch.int16), wo.view(torch.int16)


def temporal_attention_h(x, norm_g, packed, out_g, null_kv, rel, null_bias, h, d, causal, eps, lp):
    """y = LayerNorm(Attention(LayerNorm(x)) W_o) + x over the frame axis of x[B, F, P, C], one kernel (sampling path, autocast)."""
    _chk(x, norm_g, out_g, null_kv, rel, null_bias)
    B, F, P, C = x.shape
    y = torch.empty_like(x)
    _lib.call("diqt_temporal_attention_h", x, norm_g, packed[0], packed[1], packed[2], out_g, null_kv, rel, null_bias, y, B, F, P, C,
              h, d, int(causal), float(eps), lp, 1, _stream())
    return y


class _MqaAttentionFn(Function):
    """Fused multi-query attention with autograd (training path): scores and probabilities never reach HBM, the backward
    recomputes them from the row log-sum-exp (diqt_mqa_attention_fwd_lse / diqt_mqa_attention_bwd)."""
    @staticmethod
    def forward(ctx, q, kv_ext, rel, null_bias, n, h, d, n_extra, n_self, causal, scale):
        _chk(q, kv_ext, rel, null_bias)
        G = q.shape[0]
        out = torch.empty((G, n, h * d), dtype=torch.float32, device=q.device)
        lse = torch.empty((G, n * h), dtype=torch.float32, device=q.device)
        _lib.call("diqt_mqa_attention_fwd_lse", q, kv_ext, rel, null_bias, out, lse, G, n, h, d, n_extra, n_self, int(causal),
                  float(scale), _stream())
        ctx.save_for_backward(q, kv_ext, rel, null_bias, out, lse)
        ctx.cfg = (G, n, h, d, n_extra, n_self, int(causal), float(scale))
        return out

    @staticmethod
    def backward(ctx, dout):
        q, kv_ext, rel, null_bias, out, lse = ctx.saved_tensors
        G, n, h, d, n_extra, n_self, causal, scale = ctx.cfg
        dout = dout.contiguous()
        dq = torch.empty_like(q)
        dkv = torch.empty_like(kv_ext)
        drel = torch.empty_like(rel) if rel is not None else None
        dnull = torch.empty_like(null_bias) if null_bias is not None else None
        nws = _lib.query("diqt_mqa_attention_bwd_workspace_bytes", G, n, h, d, n_extra, n_self, int(rel is not None))
        ws = _workspace(nws, q.device)
        _lib.call("diqt_mqa_attention_bwd", q, kv_ext, rel, null_bias, out, dout, lse, dq, dkv, drel, dnull, ws, nws, G, n, h, d,
                  n_extra, n_self, causal, scale, _stream())
        return dq, dkv, drel, dnull, None, None, None, None, None, None, None


def mqa_attention(q, kv_ext, rel, null_bias, n, h, d, n_extra, n_self, causal, scale):
    """Fused multi-query attention WITH autograd.  q: [G, n, h*d]; kv_ext: [G, n_extra + n_self, 2d]; rel: [2n-1, h] or None."""
    return _MqaAttentionFn.apply(q.contiguous(), kv_ext.contiguous(), rel.contiguous() if rel is not None else None,
                                 null_bias.contiguous() if null_bias is not None else None, n, h, d, n_extra, n_self, bool(causal),
                                 float(scale))


def mqa_attention_fused_ok(G, n, h, d, n_self, has_rel):
    """Shapes the fused training kernels take (otherwise: GEMM -> attn_softmax -> GEMM)."""
    return d in (32, 64) and G <= 65535 and (not has_rel or 4 * (2 * n_self - 1) * h * 4 <= 24 * 1024)


def attn_softmax(sim, rel, null_bias, n, h, n_extra, n_self, causal):
    """softmax over keys of sim[G, n, h, n_extra + n_self] with T5-style relative bias table rel[2n-1, h] on the self
    keys, null_bias[h] on the null key (last extra key) and an optional causal mask (imagen_video.py:490-518)."""
    return _AttnSoftmaxFn.apply(sim.contiguous(), rel, null_bias, n, h, n_extra, n_self, bool(causal))


class _GateResidualFn(Function):
    """y = h * gate[b, c] + res  (GlobalContext gating + residual, imagen_video.py:768-770)."""
    @staticmethod
    def forward(ctx, h, gate, res):
        _chk(h, gate, res)
        B, C = h.shape[0], h.shape[-1]
        rows = h.numel() // (B * C)
        y = torch.empty_like(h)
        _lib.call("diqt_gate_residual_fwd", h, gate, res, None, 0.0, y, B, rows, C, _stream())
        ctx.save_for_backward(h, gate)
        ctx.has_res = res is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        h, gate = ctx.saved_tensors
        dy = dy.contiguous()
        B, C = h.shape[0], h.shape[-1]
        rows = h.numel() // (B * C)
        s = _stream()
        dgate = torch.empty_like(gate)
        ws, n = _reduce_ws(B, C, h.device)
        _lib.call("diqt_gate_residual_bwd", h, dy, dgate, ws, n, B, rows, C, s)
        dh = torch.empty_like(h)
        _lib.call("diqt_gate_residual_fwd", dy, gate, None, None, 0.0, dh, B, rows, C, s)
        return dh, dgate, (dy if ctx.has_res else None)


def gate_residual(h, gate, res=None):
    if not torch.is_grad_enabled() and res is not None:
        # sampling: a ResnetBlock's output usually feeds the next block's first GroupNorm -- its per-workgroup column sums ride along
        # (groupnorm_act / gn_conv3d read them from ``_diqt_stats``: no statistics pass over the tensor)
        h, gate = h.contiguous(), gate.contiguous()
        _chk(h, gate, res)
        B, C = h.shape[0], h.shape[-1]
        rows = h.numel() // (B * C)
        nblk = _lib.query("diqt_gate_residual_stats_blocks", rows, C)
        if nblk > 0:
            y = torch.empty_like(h)
            stats = torch.empty((B, nblk, 2, C), dtype=torch.float32, device=h.device)
            _lib.call("diqt_gate_residual_fwd_stats", h, gate, res, y, stats, B, rows, C, _stream())
            y._diqt_stats = ColStats(stats, nblk, rows)
            return y
    return _GateResidualFn.apply(h.contiguous(), gate.contiguous(), res)


class _ScaleFn(Function):
    @staticmethod
    def forward(ctx, x, alpha):
        _chk(x)
        out = torch.empty_like(x)
        c0 = torch.full((1,), float(alpha), dtype=torch.float32, device=x.device)
        _lib.call("diqt_axpby3", x, None, None, c0, None, None, 0.0, 0.0, 0, out, 1, x.numel(), _stream())
        ctx.alpha = float(alpha)
        return out

    @staticmethod
    def backward(ctx, d):
        return _ScaleFn.apply(d.contiguous(), ctx.alpha), None


def scale(x, alpha):
    return x if alpha == 1.0 else _ScaleFn.apply(x.contiguous(), alpha)


# --------------------------------------------------------------------------------------------
# diffusion math
# --------------------------------------------------------------------------------------------
def q_sample(x0, noise, alpha, sigma):
    """alpha[b]*x0 + sigma[b]*noise (no autograd: inputs are data)."""
    _chk(x0, noise, alpha, sigma)
    B = x0.shape[0]
    out = torch.empty_like(x0)
    _lib.call("diqt_q_sample", x0, noise, alpha, sigma, out, B, x0.numel() // B, _stream())
    return out


def ddpm_step(x_t, pred, noise, ca, cb, cn, lo, hi, clamp_mode):
    _chk(x_t, pred, noise, ca, cb, cn)
    B = x_t.shape[0]
    x_next, x0 = torch.empty_like(x_t), torch.empty_like(x_t)
    _lib.call("diqt_ddpm_step", x_t, pred, noise, ca, cb, cn, float(lo), float(hi), int(clamp_mode), x_next, x0, B,
              x_t.numel() // B, _stream())
    return x_next, x0


def abs_quantile(x, q):
    """torch.quantile(x.flatten(1).abs(), q, dim=-1) (linear interpolation) by radix select: [B] thresholds."""
    import numpy as np
    _chk(x)
    B = x.shape[0]
    per = x.numel() // B
    rank = np.float32(q) * np.float32(per - 1)             # the fp32 rank arithmetic of aten's quantile_compute
    k_lo = int(np.floor(rank))
    weight = float(np.float32(rank - np.float32(k_lo)))
    out = torch.empty(B, dtype=torch.float32, device=x.device)
    _lib.call("diqt_abs_quantile", x, out, B, per, k_lo, weight, _stream())
    return out


def dynamic_threshold(x0, s):
    """clamp(x0, -s[b], s[b]) / s[b]"""
    _chk(x0, s)
    B = x0.shape[0]
    out = torch.empty_like(x0)
    _lib.call("diqt_dynamic_threshold", x0, s, out, B, x0.numel() // B, _stream())
    return out


def mask_blend(x, y, mask):
    """mask ? y : x (mask: 0/1 float tensor of x's shape)"""
    _chk(x, y, mask)
    assert x.shape == y.shape == mask.shape
    out = torch.empty_like(x)
    _lib.call("diqt_mask_blend", x, y, mask, out, x.numel(), _stream())
    return out


def axpby3(a, b, c, c0, c1, c2, lo=0.0, hi=0.0, clamp_mode=0):
    _chk(a, b, c, c0, c1, c2)
    B = a.shape[0]
    out = torch.empty_like(a)
    _lib.call("diqt_axpby3", a, b, c, c0, c1, c2, float(lo), float(hi), int(clamp_mode), out, B, a.numel() // B,
              _stream())
    return out


class _MseClampFn(Function):
    @staticmethod
    def forward(ctx, pred, target, weight, lo, do_clamp, kind=0):
        _chk(pred, target, weight)
        B = pred.shape[0]
        per = pred.numel() // B
        partials = torch.empty(1024, dtype=torch.float32, device=pred.device)
        loss = torch.empty((), dtype=torch.float32, device=pred.device)
        clamped = torch.empty_like(pred)
        _lib.call("diqt_loss_clamp_fwd", pred, clamped, target, weight, float(lo), int(do_clamp), int(kind), partials, loss, B, per,
                  _stream())
        ctx.mark_non_differentiable(clamped)
        ctx.save_for_backward(clamped, target, weight)
        ctx.cfg = (float(lo), int(do_clamp), B, per, int(kind))
        return loss, clamped

    @staticmethod
    def backward(ctx, dloss, _dclamped):
        clamped, target, weight = ctx.saved_tensors
        lo, do_clamp, B, per, kind = ctx.cfg
        dpred = torch.empty_like(clamped)
        _lib.call("diqt_loss_clamp_bwd", clamped, target, weight, lo, do_clamp, kind, 1.0, dpred, B, per, _stream())
        return _ScaleByScalar.apply_raw(dpred, dloss), None, None, None, None, None


class _ScaleByScalar:
    @staticmethod
    def apply_raw(t, s):
        """t * s for a 0-dim device tensor s (chunk_frac scaling of the loss) — one axpby3 launch."""
        out = torch.empty_like(t)
        _lib.call("diqt_axpby3", t.view(1, -1), None, None, s.reshape(1).contiguous(), None, None, 0.0, 0.0, 0, out, 1,
                  t.numel(), _stream())
        return out


LOSS_KINDS = {'l2': 0, 'l1': 1, 'huber': 2}      # F.mse_loss / F.l1_loss / F.smooth_l1_loss (imagen_pytorch3D.py:1785-1790)


def mse_clamp(pred, target, lo=0.0, do_clamp=False, weight=None, kind='l2'):
    """mean(loss(clamp_min(pred, lo) - target) * weight[b]) with loss = square ('l2'), absolute value ('l1') or Huber with beta 1
    ('huber'); returns (loss, clamped_pred).  The reference clamps ``pred`` in place and returns it (imagen_pytorch3D.py:2361-2364);
    here the clamped values come back as a separate (non-differentiable) tensor — same numbers, no aliasing of an autograd view."""
    return _MseClampFn.apply(pred.contiguous(), target.contiguous(), weight, lo, do_clamp, LOSS_KINDS[kind])


# --------------------------------------------------------------------------------------------
# optimiser
# --------------------------------------------------------------------------------------------
def adam_step(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step, zero_grad=True, grad_scale=None):
    """Fused Adam over a flat arena.  ``grad_scale``: device scalar every gradient is multiplied with first (the clip coefficient of
    ``grad_norm_clip``)."""
    _chk(param, grad, exp_avg, exp_avg_sq)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    if grad_scale is None:
        _lib.call("diqt_adam_step", param, grad, exp_avg, exp_avg_sq, param.numel(), float(lr), float(beta1), float(beta2),
                  float(eps), float(weight_decay), float(bc1), float(bc2), int(zero_grad), _stream())
    else:
        _chk(grad_scale)
        _lib.call("diqt_adam_step_scaled", param, grad, exp_avg, exp_avg_sq, param.numel(), float(lr), float(beta1), float(beta2),
                  float(eps), float(weight_decay), float(bc1), float(bc2), int(zero_grad), grad_scale, _stream())
    bump_weight_epoch()


def grad_norm_clip(grad_flat, max_norm):
    """torch.nn.utils.clip_grad_norm_ over a flat gradient arena WITHOUT touching it: returns a device tensor [total L2 norm,
    min(1, max_norm / (norm + 1e-6))]; the coefficient goes to ``adam_step(grad_scale=out[1:])``, which consumes the gradients."""
    _chk(grad_flat)
    nb = _lib.query("diqt_grad_norm_workspace_bytes")
    ws = _workspace(nb, grad_flat.device)
    out = torch.empty(2, dtype=torch.float32, device=grad_flat.device)
    _lib.call("diqt_grad_norm_clip", grad_flat, grad_flat.numel(), float(max_norm), ws, out, _stream())
    return out


def multi_accumulate(dst_flat, srcs, offsets):
    """dst_flat[off : off + t.numel()] += t for every (t, off): one launch for a whole micro-step's parameter gradients."""
    _chk(dst_flat, *srcs)
    if not srcs:
        return
    rows = []
    for t, off in zip(srcs, offsets):
        assert t.is_contiguous() and t.dtype == torch.float32
        rows.append((t.data_ptr(), int(off), t.numel()))
    # the table stays on the host and travels in the kernel arguments: no upload, and a captured micro-step replays the launch as it is
    table = torch.tensor(rows, dtype=torch.int64)
    _lib.call("diqt_multi_accumulate_host", dst_flat, table, len(rows), 16, _stream())   # same stream as the producers/allocator


def ema_lerp(ema, param, one_minus_decay):
    _chk(ema, param)
    _lib.call("diqt_ema_lerp", ema, param, ema.numel(), float(one_minus_decay), _stream())
    bump_weight_epoch()


# --------------------------------------------------------------------------------------------
# whole-volume inference (test_all.py:182-300)
# --------------------------------------------------------------------------------------------
def _chk_int(*ts):
    for t in ts:
        if not (t.is_cuda and t.dtype == torch.int32 and t.is_contiguous()):
            raise RuntimeError("index tables must be contiguous int32 HIP tensors")


def patch_gather(vol, idx, P, mean, std, want_patches=True, want_nonzero=False):
    """(patches [n,1,P,P,P] normalised, nonzero [n] int32) of the sliding-window origins ``idx`` [n,3] in a raw [D,H,W] volume."""
    _chk(vol)
    _chk_int(idx)
    D, H, W = vol.shape
    n = idx.shape[0]
    out = torch.empty((n, 1, P, P, P), dtype=torch.float32, device=vol.device) if want_patches else None
    nz = torch.empty(n, dtype=torch.int32, device=vol.device) if want_nonzero else None
    for lo in range(0, n, 65535):
        hi = min(n, lo + 65535)
        _lib.call("diqt_patch_gather", vol, idx[lo:hi], out[lo:hi] if out is not None else None,
                  nz[lo:hi] if nz is not None else None, hi - lo, D, H, W, P, float(mean), float(std), _stream())
    return out, nz


def patch_scatter(patches, idx, margins, pred, P):
    _chk(patches, pred)
    _chk_int(idx, margins)
    D, H, W = pred.shape
    _lib.call("diqt_patch_scatter", patches, idx, margins, pred, idx.shape[0], D, H, W, P, _stream())


def background_reset(pred, vol, mean, std, min_val):
    _chk(pred, vol)
    _lib.call("diqt_background_reset", pred, vol, pred.numel(), float(mean), float(std), float(min_val), _stream())


def min_value(x):
    _chk(x)
    ws = torch.empty(1024, dtype=torch.float32, device=x.device)
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    _lib.call("diqt_min_value", x, x.numel(), ws, out, _stream())
    return out


def patch_pair_crop(lr_vols, hr_vols, sel, P, mode, mean, std):
    """data.py:119-132: crop + normalise ``sel[n] = (volume, i0, j0, k0)`` patch pairs out of the HBM-resident [V,D,H,W]
    volume stacks in one launch.  Returns (lr [n,P,P,P], hr [n,P,P,P])."""
    _chk(lr_vols, hr_vols)
    V, D, H, W = lr_vols.shape
    n = sel.shape[0]
    lr = torch.empty(n, P, P, P, device=lr_vols.device, dtype=torch.float32)
    hr = torch.empty_like(lr)
    nws = _lib.query("diqt_patch_pair_crop_workspace_bytes", n, P) if mode == 1 else 0
    ws = _workspace(nws, lr_vols.device) if nws else None
    _lib.call("diqt_patch_pair_crop", lr_vols, hr_vols, sel, lr, hr, ws, nws, n, V, D, H, W, P, int(mode), float(mean), float(std),
              _stream())
    return lr, hr


def minmax(x):
    """Device [2] tensor {min, max} of x (no host sync)."""
    _chk(x)
    out = torch.empty(2, device=x.device, dtype=torch.float32)
    _lib.call("diqt_minmax", x, x.numel(), _workspace(8192, x.device), out, _stream())
    return out


def psnr(pred, target, stats=None, data_range=1.0):
    """Device [2] tensor {mse, PSNR}; ``stats`` = device [4] {pred min, max, target min, max} for min-max normalisation."""
    _chk(pred, target)
    out = torch.empty(2, device=pred.device, dtype=torch.float32)
    _lib.call("diqt_psnr", pred, target, pred.numel(), stats, float(data_range), _workspace(8192, pred.device), out, _stream())
    return out


def ssim3d(pred, target, taps, stats=None, data_range=1.0, k1=0.01, k2=0.03):
    """pred/target: contiguous [N, D, H, W]; taps: host float32 numpy array (odd length <= 11).  Device [1] tensor."""
    import ctypes
    _chk(pred, target)
    N, D, H, W = pred.shape
    K = int(taps.shape[0])
    nws = _lib.query("diqt_ssim3d_workspace_bytes", N, D, H, W, K)
    if nws == 0:
        raise RuntimeError(f"ssim3d: volume {D}x{H}x{W} smaller than the {K}-tap filter")
    out = torch.empty(1, device=pred.device, dtype=torch.float32)
    _lib.call("diqt_ssim3d", pred, target, N, D, H, W, taps.ctypes.data_as(ctypes.c_void_p), K, stats, float(data_range), float(k1),
              float(k2), _workspace(nws, pred.device), nws, out, _stream())
    return out
