"""Production-size parity of Family B and of the 16-bit path against the CPU oracle.

The reference goldens pin ``Unet3D`` at dim 16 / 32 on 8^3 / 16^3 volumes, where the production kernels do not engage (the persistent
16-bit conv needs >= 512 units, the pointwise GEMM >= 2048 rows, the one-kernel temporal attention C in {64, 128, 256} and 32 / 64
frames, the flash attention its long-sequence builds).  Here the networks ``bench.py`` times run at their real sizes against
``oracle.iqt_oracle_b`` evaluated on the host in the same process (seconds), and a launch census (``_lib.census``) asserts that the
kernels under test were really dispatched (round 4: ``conv3d_fwd_h(v9h)`` = conv_f9h_kernel carries the per-frame convs).  Reference: imagen_video.py:1585-1822 (Unet3D.forward), elucidated_imagen.py:329-358.

Tolerances: fp32 as for Family A (tests/test_gpu_unet.py: max 2e-4 of max|ref|, rel-L2 <= 2e-5; gradients 1e-3 of max|ref|);
autocast: rel-L2 to the fp32 oracle <= 1.5 x the round-off of the ORACLE ITSELF under ``torch.autocast('cpu')`` in that type -- the
rule tests/test_gpu_lowprec.py uses against the reference's own autocast fixtures."""
import pytest
import torch

from oracle import iqt_oracle as O
from oracle import iqt_oracle_b as OB

pytestmark = pytest.mark.gpu
DEV = "cuda"
LP = {"fp16": torch.float16, "bf16": torch.bfloat16}


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm()).item()


def close(got, ref, tol, what=""):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    assert got.shape == ref.shape, f"{what}: {tuple(got.shape)} vs {tuple(ref.shape)}"
    err, scale = (got - ref).abs().max().item(), ref.abs().max().item()
    assert err <= tol * scale + 1e-6, f"{what}: max err {err:.3e}, scale {scale:.3e}"
    return rel(got, ref)


def _net(kw, seed):
    from diffusioniqt_amd.imagen_video import Unet3D
    unet = Unet3D(**kw)
    sd = O.hash_fill_state_dict(unet.state_dict(), seed)
    unet.load_state_dict(sd)
    return unet, sd, OB.unet3d_config(**kw)


def _inputs(B, F, S, seed):
    g = torch.Generator().manual_seed(seed)
    x, lr = torch.randn(B, 1, F, S, S, generator=g), torch.randn(B, 1, F, S, S, generator=g)
    t, lt = torch.rand(B, generator=g) * 2 - 1, torch.rand(B, generator=g)      # c_noise-like times, lowres augmentation times
    return x, lr, t, lt


_ORACLE = {}


def _oracle(tag, sd, cfg, x, t, lr, lt, mode, nb=None):
    """fp32 oracle output, or the oracle under CPU autocast in ``mode`` (cached per (net, mode): three parametrisations share them).
    ``nb``: only the first nb batch entries (torch's CPU fp16 matmuls are ~10x slower than its bf16 ones: the oracle's own fp16 round-off,
    which only sets the tolerance, is measured on one entry)."""
    key = (tag, mode, nb)
    if key not in _ORACLE:
        if nb is not None:
            x, t, lr, lt = x[:nb], t[:nb], lr[:nb], lt[:nb]
        with torch.no_grad():
            if mode == "fp32":
                y = OB.unet3d_forward(sd, cfg, x, t, lowres_cond_img=lr, lowres_noise_times=lt)
            else:
                with torch.autocast('cpu', dtype=LP[mode]):
                    y = OB.unet3d_forward(sd, cfg, x, t, lowres_cond_img=lr, lowres_noise_times=lt)
        _ORACLE[key] = y.float()
    return _ORACLE[key]


def _own_roundoff(tag, sd, cfg, x, t, lr, lt, mode, y32):
    """rel-L2 of the oracle under CPU autocast to the fp32 oracle (fp16: on the first batch entry)."""
    nb = 1 if mode == "fp16" else None
    ya = _oracle(tag, sd, cfg, x, t, lr, lt, mode, nb)
    return rel(ya, y32[:ya.shape[0]])


@pytest.mark.parametrize("mode", ["fp32", "fp16", "bf16"])
def test_unet3d_dim64_at_32cubed_vs_oracle(mode):
    """(a) the network ``BENCH.unet3d_edm`` times -- Unet3D dim 64, mults (1,2,4), attention at the last level + middle -- at 32^3, B = 4
    (512 tiles of 256 voxels at the 64-channel level: the size from which the persistent 16-bit conv walk engages)."""
    from bench import unet3d_kwargs
    from diffusioniqt_amd import _lib
    kw = unet3d_kwargs()
    unet, sd, cfg = _net(kw, 11)
    unet = unet.to(DEV).eval()
    x, lr, t, lt = _inputs(4, 32, 32, 2024)
    y32 = _oracle("u3", sd, cfg, x, t, lr, lt, "fp32")
    args = (x.to(DEV), t.to(DEV))
    kws = dict(lowres_cond_img=lr.to(DEV), lowres_noise_times=lt.to(DEV))
    with torch.no_grad(), _lib.census() as c:
        if mode == "fp32":
            y = unet(*args, **kws)
        else:
            with torch.autocast('cuda', dtype=LP[mode]):
                y = unet(*args, **kws)
    if mode == "fp32":
        r = close(y, y32, 2e-4, "Unet3D dim 64 @ 32^3")
        assert r <= 2e-5, r
        assert c.count("conv3d_fwd(v9)") > 0 and c.count("mqa_attention_fwd") > 0 and c.count("conv3d_fwd(1x1x1") > 0
    else:
        own_ref = _own_roundoff("u3", sd, cfg, x, t, lr, lt, mode, y32)           # the oracle's own round-off in this type
        got = rel(y, y32)
        assert 0 < got <= 1.5 * own_ref, (got, own_ref)
        # the production 16-bit kernels really ran: the persistent conv walk, the pointwise / temporal GEMM, the one-kernel temporal
        # attention block and the 16-bit GroupNorm-apply
        for tag in ("conv3d_fwd_h(v9h)", "conv3d_fwd_h(persistent)", "conv3d_fwd_h(gemm)", "temporal_attention_h", "gn_act_fwd_h", "mqa_attention_fwd_h"):
            assert c.count(tag) > 0, f"{tag} was not dispatched"


@pytest.mark.parametrize("mode", ["fp16", "bf16", "fp32"])
def test_c5_stage2_eval_at_64_frames_vs_oracle(mode):
    """(b) one evaluation of the C5 cascade's second stage (Unet3D dim 64, lowres-conditioned, no layer attention, attention in the
    middle over 64 x 8 x 8 = 4096 tokens) at 64 frames x 32 x 32, B = 2 -- the 16-bit path composed at a production size."""
    from bench import unet3d_kwargs
    from diffusioniqt_amd import _lib
    kw = unet3d_kwargs(layer_attns=False)
    unet, sd, cfg = _net(kw, 5)
    unet = unet.to(DEV).eval()
    x, lr, t, lt = _inputs(2, 64, 32, 77)
    y32 = _oracle("c5s2", sd, cfg, x, t, lr, lt, "fp32")
    args = (x.to(DEV), t.to(DEV))
    kws = dict(lowres_cond_img=lr.to(DEV), lowres_noise_times=lt.to(DEV))
    with torch.no_grad(), _lib.census() as c:
        if mode == "fp32":
            y = unet(*args, **kws)
        else:
            with torch.autocast('cuda', dtype=LP[mode]):
                y = unet(*args, **kws)
    if mode == "fp32":
        r = close(y, y32, 2e-4, "C5 stage 2 @ 64 x 32 x 32")
        assert r <= 2e-5, r
    else:
        own_ref = _own_roundoff("c5s2", sd, cfg, x, t, lr, lt, mode, y32)
        got = rel(y, y32)
        assert 0 < got <= 1.5 * own_ref, (got, own_ref)
        for tag in ("conv3d_fwd_h(v9h)", "conv3d_fwd_h(persistent)", "conv3d_fwd_h(gemm)", "temporal_attention_h", "mqa_attention_fwd_h"):
            assert c.count(tag) > 0, f"{tag} was not dispatched"


def test_c5_stage2_batch_of_8_through_graph_replay_equals_single_samples():
    """(b) batch invariance at the size bench.py runs (B = 8) THROUGH the hipGraph replay path: the preconditioned evaluation of a batch
    of 8 -- eager, then captured and replayed -- against the same 8 samples evaluated one at a time.  Replay vs eager of the same
    batch: bit-identical.  Batch of 8 vs 8 x batch of 1: tile -> workgroup assignment and split-K decisions differ, so equal to the
    16-bit round-off (rel-L2 <= 3e-3, the level of one fp16 rounding of the activations; no sample may deviate more than the others)."""
    from bench import unet3d_kwargs
    from diffusioniqt_amd import graphs
    from diffusioniqt_amd.imagen_video import Unet3D
    from diffusioniqt_amd.elucidated_imagen import ElucidatedImagen
    kw = unet3d_kwargs(layer_attns=False)
    u1, u2 = Unet3D(**{**kw, 'lowres_cond': False}), Unet3D(**kw)
    elu = ElucidatedImagen(unets=(u1, u2), image_sizes=(16, 32), channels=1, condition_on_text=False, auto_normalize_img=False,
                           cond_drop_prob=0.0, num_sample_steps=4, temporal_downsample_factor=(2, 1))
    u = elu.unets[1]
    u.load_state_dict(O.hash_fill_state_dict(u.state_dict(), 5))
    elu = elu.to(DEV).eval()
    u = elu.unets[1]
    B = 8
    x, lr, _, lt = _inputs(B, 64, 32, 99)
    x, lr, lt = x.to(DEV), lr.to(DEV), lt.to(DEV)
    sig = 1.7
    was, wasf = graphs.ENABLED, graphs.FORCE
    try:
        graphs.ENABLED, graphs.FORCE = True, True
        elu._graphs.clear()
        elu._graphs.replays = 0
        outs = []
        with torch.no_grad(), torch.autocast('cuda', dtype=torch.float16):
            for _ in range(4):                       # two eager calls, the capture + first replay, one more replay
                outs.append(elu.preconditioned_network_forward(u.forward, x, sig, sigma_data=0.5, lowres_cond_img=lr,
                                                               lowres_noise_times=lt).clone())
            assert elu._graphs.replays == 2
            assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]) and torch.equal(outs[0], outs[3])
            graphs.ENABLED = False
            singles = [elu.preconditioned_network_forward(u.forward, x[i:i + 1].contiguous(), sig, sigma_data=0.5,
                                                          lowres_cond_img=lr[i:i + 1].contiguous(), lowres_noise_times=lt[i:i + 1].contiguous())
                       for i in range(B)]
    finally:
        graphs.ENABLED, graphs.FORCE = was, wasf
        elu._graphs.clear()
    errs = [rel(outs[0][i:i + 1], singles[i]) for i in range(B)]
    assert max(errs) <= 3e-3, errs
    assert torch.isfinite(outs[0]).all()


def test_unet3d_dim64_at_32cubed_whole_network_gradients_vs_oracle():
    """(c) fp32 whole-network backward of the same Unet3D at 32^3 (B = 1): loss and EVERY parameter gradient against autograd of the CPU
    oracle -- the one-pass temporal-attention backward (mqa_seq_bwd), the flash dQ / dK|dV kernels of the joint attentions and the
    (1,3,3) / (3,1,1) weight-gradient kernels composed, not only alone."""
    from bench import unet3d_kwargs
    from diffusioniqt_amd import _lib
    kw = unet3d_kwargs()
    unet, sd, cfg = _net(kw, 13)
    unet = unet.to(DEV).train()
    x, lr, t, lt = _inputs(1, 32, 32, 4711)
    sdg = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    yr = OB.unet3d_forward(sdg, cfg, x, t, lowres_cond_img=lr, lowres_noise_times=lt)
    (yr ** 2).mean().backward()
    ref = {k: v.grad for k, v in sdg.items() if v.grad is not None}
    with _lib.census() as c:
        y = unet(x.to(DEV), t.to(DEV), lowres_cond_img=lr.to(DEV), lowres_noise_times=lt.to(DEV))
        (y ** 2).mean().backward()
    r = close(y, yr, 2e-4, "Unet3D dim 64 @ 32^3 (train mode)")
    assert r <= 2e-5, r
    for tag in ("mqa_attention_bwd(seq)", "mqa_attention_bwd(dq)", "mqa_attention_bwd(dkv)", "conv3d_bwd_weight(v3)"):
        assert c.count(tag) > 0, f"{tag} was not dispatched"
    n = 0
    for k, p in unet.named_parameters():
        if k in ref:
            assert p.grad is not None, k
            close(p.grad, ref[k], 1e-3, f"grad {k}")
            n += 1
        else:
            assert p.grad is None, k
    assert n > 300, n
