"""Per-kernel parity on a real MI355X: every HIP operator (through the C ABI via diffusioniqt_amd.ops)
against plain PyTorch fp32/fp64 on the CPU with the same seeded inputs.

Tolerance (fp32 path, SURVEY.md §8d / BASELINE.md §4): the MFMA f32 conv is a k-ordered fmaf chain, so
|err| <= ~1e-6 * sum|a*b|; tests use max-abs error <= 2e-5 * max|ref| (+1e-6) unless noted.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "needs an MI355X"
    from diffusioniqt_amd import ops as _ops, _lib
    _lib.load()
    return _ops


DEV = "cuda"


def close(got, ref, tol=2e-5, what=""):
    got = got.detach().double().cpu()
    ref = ref.detach().double().cpu()
    assert got.shape == ref.shape, f"{what}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    scale = ref.abs().max().item()
    err = (got - ref).abs().max().item()
    assert err <= tol * scale + 1e-6, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


def cl(x):      # NCDHW -> NDHWC on device
    return x.permute(0, 2, 3, 4, 1).contiguous().to(DEV)


def cf(x):      # NDHWC device -> NCDHW cpu
    return x.detach().cpu().permute(0, 4, 1, 2, 3).contiguous()


CONV_CASES = [
    # B, (D,H,W), Cin, Cout, k, pad
    (2, (16, 16, 16), 64, 64, (3, 3, 3), (1, 1, 1)),
    (1, (8, 8, 8), 128, 128, (3, 3, 3), (1, 1, 1)),
    (2, (8, 8, 8), 2, 16, (3, 3, 3), (1, 1, 1)),          # init conv: Cin=2 (scalar staging path)
    (1, (8, 8, 8), 192, 128, (3, 3, 3), (1, 1, 1)),        # concat skip: 3 K-chunks, 2 N-tiles
    (3, (10, 10, 10), 16, 16, (3, 3, 3), (0, 0, 0)),       # un-padded 'boundary' conv 10^3 -> 8^3
    (1, (7, 9, 10), 20, 24, (3, 3, 3), (1, 1, 1)),         # ragged extents, Cin % 32 != 0, Cout % 32 != 0
    (2, (4, 4, 4), 512, 64, (1, 1, 1), (0, 0, 0)),         # 1x1x1 after space-to-depth
    (1, (16, 16, 16), 64, 1, (1, 1, 1), (0, 0, 0)),        # final conv Cout = 1
    (2, (16, 16, 16), 64, 512, (1, 1, 1), (0, 0, 0)),      # attention to_q-shaped pointwise conv: weight gradient on conv_wgrad3_kernel<1,1,1> (8 co blocks x 32 row slices)
    (1, (8, 8, 8), 256, 72, (1, 1, 1), (0, 0, 0)),         # ... 4 ci blocks, ragged co block
    (3, (4, 4, 12), 48, 136, (1, 1, 1), (0, 0, 0)),        # ... 9 row tiles over 256 workgroups' worth of slices, ragged ci / co blocks
    (2, (6, 8, 8), 32, 32, (1, 3, 3), (0, 1, 1)),          # pseudo-3D spatial conv
    (2, (6, 8, 8), 8, 40, (1, 7, 7), (0, 3, 3)),           # cross-embed (1,7,7)
    (2, (5, 1, 1), 33, 7, (3, 1, 1), (1, 0, 0)),           # temporal conv, odd channels
    (2, (6, 16, 16), 2, 16, (1, 15, 15), (0, 7, 7)),       # Family-B cross-embed (1,15,15) on 2 channels: tap-packed K (15 chunks)
    (2, (9, 10, 12), 1, 20, (3, 3, 3), (1, 1, 1)),         # Cin = 1, ragged extents (tap-packed, 1 chunk)
    (1, (8, 8, 8), 3, 70, (3, 3, 3), (1, 1, 1)),           # Cin = 3 padded to 4, two N tiles
    (2, (8, 8, 8), 24, 2, (3, 3, 3), (1, 1, 1)),           # Cout = 2: the backward-data pass is a tap-packed conv
    (1, (6, 8, 8), 4, 8, (1, 7, 7), (0, 3, 3)),            # Cin = 4 (1,7,7)
    (2, (16, 16, 16), 2, 24, (3, 3, 3), (1, 1, 1)),        # Cin = 2 with >= 4096 voxels: dW through im2col + split-K GEMM (K' = 54)
    (1, (4, 32, 32), 3, 20, (1, 7, 7), (0, 3, 3)),         # ... Cin = 3 padded to 4, K' = 196 (128-row tiles)
    (2, (16, 16, 16), 128, 64, (1, 1, 1), (0, 0, 0)),      # 1x1x1 with many voxels: dW through the split-K GEMM (x on the M side)
    (1, (16, 16, 16), 64, 192, (1, 1, 1), (0, 0, 0)),      # ... dY on the M side
    (2, (8, 16, 16), 72, 40, (1, 1, 1), (0, 0, 0)),        # ... channel counts that are not multiples of the tile
    # weight gradient, version 3 (conv_wgrad3_kernel): split-K ranges of SEVERAL 64-voxel tiles per workgroup, i.e. the in-loop
    # LDS-DMA of the next tile into the other buffer, tile-walk wrap-around over w / h / d / batch, the shared middle tap
    (2, (32, 32, 32), 32, 32, (3, 3, 3), (1, 1, 1)),       # 1024 tiles on 256 workgroups: 4 tiles each
    (2, (16, 16, 16), 96, 40, (3, 3, 3), (1, 1, 1)),       # 3 input-channel blocks, ragged Cout, 2 tiles each
    (1, (9, 13, 22), 32, 32, (3, 3, 3), (1, 1, 1)),        # ragged extents in every axis (tiles 2x4x8 hang over)
    (2, (18, 18, 18), 16, 16, (3, 3, 3), (0, 0, 0)),       # un-padded conv: no halo outside the volume
    (4, (8, 32, 32), 64, 64, (1, 3, 3), (0, 1, 1)),        # (1,3,3): co-half x ci-half waves, 1x8x8 tiles, 2 tiles each
    (3, (5, 20, 12), 24, 72, (1, 3, 3), (0, 1, 1)),        # ... ragged channels (Cin < 32: one half-empty ci half) and extents
    (4, (32, 16, 16), 64, 48, (3, 1, 1), (1, 0, 0)),       # (3,1,1) temporal conv: 8x2x4 tiles
    (2, (12, 6, 10), 36, 20, (3, 1, 1), (1, 0, 0)),        # ... ragged
]


@pytest.mark.parametrize("B,sp,Cin,Cout,k,pad", CONV_CASES)
def test_conv3d_fwd_bwd(ops, B, sp, Cin, Cout, k, pad):
    g = torch.Generator().manual_seed(hash((B, sp, Cin, Cout, k)) % 2 ** 31)
    x = torch.randn(B, Cin, *sp, generator=g)
    w = torch.randn(Cout, Cin, *k, generator=g) / math.sqrt(Cin * k[0] * k[1] * k[2])
    b = torch.randn(Cout, generator=g) * 0.1
    xr, wr, br = x.double().requires_grad_(), w.double().requires_grad_(), b.double().requires_grad_()
    yr = F.conv3d(xr, wr, br, padding=pad)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy.double())

    xd, wd, bd = cl(x).requires_grad_(), w.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
    y = ops.conv3d(xd, wd, bd, pad)
    close(cf(y), yr, what="conv fwd")
    y.backward(cl(dy))
    close(cf(xd.grad), xr.grad, what="conv dx")
    close(wd.grad, wr.grad, tol=5e-5, what="conv dw")
    close(bd.grad, br.grad, tol=5e-5, what="conv db")


def test_conv3d_residual_epilogue_and_linear(ops):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 16, 8, 8, 8, generator=g)
    w = torch.randn(16, 16, 3, 3, 3, generator=g) * 0.05
    r = torch.randn(2, 16, 8, 8, 8, generator=g)
    y = ops.conv3d(cl(x), w.to(DEV), None, (1, 1, 1), residual=cl(r))
    close(cf(y), F.conv3d(x, w, padding=1) + r, what="conv+residual")
    xl = torch.randn(3, 17, generator=g)
    wl = torch.randn(64, 17, generator=g)
    bl = torch.randn(64, generator=g)
    xld, wld, bld = xl.to(DEV).requires_grad_(), wl.to(DEV).requires_grad_(), bl.to(DEV).requires_grad_()
    yl = ops.linear(xld, wld, bld)
    close(yl, F.linear(xl, wl, bl), what="linear")
    yl.sum().backward()
    close(wld.grad, xl.sum(0)[None, :].expand(64, 17), what="linear dw")
    close(xld.grad, wl.sum(0)[None, :].expand(3, 17), what="linear dx")


@pytest.mark.parametrize("M,K,N,bias", [(8, 256, 128, True), (1, 17, 256, True), (64, 100, 33, False), (65, 64, 48, True),
                                         (16, 1024, 256, True)])
def test_linear_skinny_and_mfma_paths(ops, M, K, N, bias):
    """nn.Linear call sites (time MLPs: <= 64 rows take diqt_linear_small_*, more rows the MFMA kernel)."""
    g = torch.Generator().manual_seed(M * 1000 + K)
    x = torch.randn(M, K, generator=g).double()
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).double()
    b = torch.randn(N, generator=g).double() if bias else None
    up = torch.randn(M, N, generator=g).double()
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    br = b.clone().requires_grad_() if bias else None
    F.linear(xr, wr, br).backward(up)
    xd, wd = x.float().to(DEV).requires_grad_(), w.float().to(DEV).requires_grad_()
    bd = b.float().to(DEV).requires_grad_() if bias else None
    y = ops.linear(xd, wd, bd)
    close(y, F.linear(x, w, b), what="linear fwd")
    y.backward(up.float().to(DEV))
    close(xd.grad, xr.grad, what="linear dx")
    close(wd.grad, wr.grad, what="linear dw")
    if bias:
        close(bd.grad, br.grad, what="linear db")


@pytest.mark.parametrize("groups,stride,k,pad,Cin,Cout", [(16, 1, 3, 1, 16, 16), (8, 4, 4, 0, 8, 8), (1, 2, 2, 0, 4, 6),
                                                            (12, 1, (3, 1, 1), (1, 0, 0), 12, 12)])
def test_conv3d_direct(ops, groups, stride, k, pad, Cin, Cout):
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, Cin, 8, 8, 8, generator=g)
    kk = (k,) * 3 if isinstance(k, int) else k
    w = torch.randn(Cout, Cin // groups, *kk, generator=g) * 0.2
    b = torch.randn(Cout, generator=g)
    xr, wr, br = x.double().requires_grad_(), w.double().requires_grad_(), b.double().requires_grad_()
    yr = F.conv3d(xr, wr, br, stride=stride, padding=pad, groups=groups)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy.double())
    xd, wd, bd = cl(x).requires_grad_(), w.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
    y = ops.conv3d_direct(xd, wd, bd, stride, pad, groups)
    close(cf(y), yr, what="direct fwd")
    y.backward(cl(dy))
    close(cf(xd.grad), xr.grad, what="direct dx")
    close(wd.grad, wr.grad, tol=1e-4, what="direct dw")
    close(bd.grad, br.grad, tol=1e-4, what="direct db")


@pytest.mark.parametrize("C,G,with_ss,act", [(64, 8, True, "mish"), (192, 8, False, "mish"), (16, 8, True, "silu"), (6, 3, True, "mish")])
def test_groupnorm_scale_shift_act(ops, C, G, with_ss, act):
    g = torch.Generator().manual_seed(3)
    B, S = 3, 6
    x = torch.randn(B, C, S, S, S, generator=g) * 2 + 0.5
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
    ss = torch.randn(B, 2 * C, generator=g) * 0.5 if with_ss else None
    fn = F.mish if act == "mish" else F.silu
    xr, gr, br = x.double().requires_grad_(), gamma.double().requires_grad_(), beta.double().requires_grad_()
    ssr = ss.double().requires_grad_() if with_ss else None
    h = F.group_norm(xr, G, gr, br, eps=1e-5)
    if with_ss:
        sc, sh = ssr[:, :C, None, None, None], ssr[:, C:, None, None, None]
        h = h * (sc + 1) + sh
    yr = fn(h)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy.double())
    xd, gd, bd = cl(x).requires_grad_(), gamma.to(DEV).requires_grad_(), beta.to(DEV).requires_grad_()
    ssd = ss.to(DEV).requires_grad_() if with_ss else None
    y = ops.groupnorm_act(xd, gd, bd, ssd, G, ops.ACT_MISH if act == "mish" else ops.ACT_SILU, 1e-5)
    close(cf(y), yr, what="gn fwd")
    y.backward(cl(dy))
    close(cf(xd.grad), xr.grad, tol=1e-4, what="gn dx")
    close(gd.grad, gr.grad, tol=1e-4, what="gn dgamma")
    close(bd.grad, br.grad, tol=1e-4, what="gn dbeta")
    if with_ss:
        close(ssd.grad, ssr.grad, tol=1e-4, what="gn dss")


@pytest.mark.parametrize("name,fn", [("mish", F.mish), ("silu", F.silu), ("gelu", F.gelu), ("relu", F.relu), ("sigmoid", torch.sigmoid)])
def test_activations(ops, name, fn):
    x = torch.linspace(-30, 30, 4099)
    xr = x.double().requires_grad_()
    yr = fn(xr)
    yr.sum().backward()
    xd = x.to(DEV).requires_grad_()
    y = ops.activation(xd, getattr(ops, "ACT_" + name.upper()))
    close(y, yr, tol=2e-6, what=name)
    y.sum().backward()
    close(xd.grad, xr.grad, tol=1e-5, what=name + " grad")


def test_se_gate_residual(ops):
    g = torch.Generator().manual_seed(8)
    B, C, S = 2, 32, 6
    h, res = torch.randn(B, C, S, S, S, generator=g), torch.randn(B, C, S, S, S, generator=g)
    w1, w2 = torch.randn(C // 16, C, generator=g) * 0.3, torch.randn(C, C // 16, generator=g) * 0.3
    hr, rr, w1r, w2r = (t.double().requires_grad_() for t in (h, res, w1, w2))
    yv = torch.sigmoid(F.linear(F.relu(F.linear(hr.mean(dim=(2, 3, 4)), w1r)), w2r))
    yr = hr * yv[:, :, None, None, None] + rr
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy.double())
    hd, rd, w1d, w2d = cl(h).requires_grad_(), cl(res).requires_grad_(), w1.to(DEV).requires_grad_(), w2.to(DEV).requires_grad_()
    y = ops.se_gate_residual(hd, w1d, w2d, rd)
    close(cf(y), yr, what="se fwd")
    y.backward(cl(dy))
    close(cf(hd.grad), hr.grad, tol=1e-4, what="se dh")
    close(cf(rd.grad), rr.grad, what="se dres")
    close(w1d.grad, w1r.grad, tol=1e-4, what="se dw1")
    close(w2d.grad, w2r.grad, tol=1e-4, what="se dw2")


def test_chan_layernorm_and_learned_sinu(ops):
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 24, 4, 4, 4, generator=g)
    gn = torch.randn(24, 1, 1, 1, generator=g)
    xr, gr = x.double().requires_grad_(), gn.double().requires_grad_()
    var = torch.var(xr, dim=1, unbiased=False, keepdim=True)
    yr = (xr - xr.mean(dim=1, keepdim=True)) * (var + 1e-5).rsqrt() * gr
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy.double())
    xd, gd = cl(x).requires_grad_(), gn.to(DEV).requires_grad_()
    y = ops.chan_layernorm(xd, gd)
    close(cf(y), yr, what="chanln")
    y.backward(cl(dy))
    close(cf(xd.grad), xr.grad, tol=1e-4, what="chanln dx")
    close(gd.grad, gr.grad, tol=1e-4, what="chanln dg")

    t, w = torch.randn(5, generator=g) * 3, torch.randn(8, generator=g)
    tr, wr = t.double(), w.double().requires_grad_()
    fr = tr[:, None] * wr[None, :] * 2 * math.pi
    er = torch.cat((tr[:, None], fr.sin(), fr.cos()), dim=-1)
    de = torch.randn(er.shape, generator=g)
    er.backward(de.double())
    wd = w.to(DEV).requires_grad_()
    e = ops.learned_sinusoidal(t.to(DEV), wd)
    close(e, er, tol=2e-5, what="sinu")
    e.backward(de.to(DEV))
    close(wd.grad, wr.grad, tol=1e-4, what="sinu dw")


def test_shuffles_concat_subvolumes_bit_exact(ops):
    from oracle import iqt_oracle as O
    g = torch.Generator().manual_seed(10)
    x = torch.randn(2, 3, 8, 6, 4, generator=g)
    y = ops.space_to_depth(cl(x))
    assert torch.equal(cf(y), O._space_to_depth(x))
    z = torch.randn(2, 16, 4, 3, 2, generator=g)
    assert torch.equal(cf(ops.depth_to_space(cl(z))), O._pixel_shuffle3d(z))
    assert torch.equal(cf(ops.depth_to_space(y)), x)                      # exact inverse
    a, b = torch.randn(2, 5, 3, 3, 3, generator=g), torch.randn(2, 7, 3, 3, 3, generator=g)
    assert torch.equal(cf(ops.concat_channels(cl(a), cl(b))), torch.cat((a, b), 1))
    vol = torch.arange(2 * 12 ** 3, dtype=torch.float32).reshape(1, 2, 12, 12, 12)
    sub = ops.split_volume(cl(vol), 3, 4)
    assert torch.equal(cf(sub), O.convert_volume_to_subvolume(vol, (27, 2, 4, 4, 4)))
    assert torch.equal(cf(ops.merge_volume(sub, 3)), vol)
    halo = ops.split_volume(ops.merge_volume(sub, 3), 3, 4, halo=1)
    assert torch.equal(cf(halo), O.boundary_pad(cf(sub), 3))
    # gradients of the data-movement ops (adjoints)
    v = cl(vol).requires_grad_()
    hs = ops.split_volume(v, 3, 4, halo=1)
    wgt = torch.randn(hs.shape, generator=g).to(DEV)
    (hs * wgt).sum().backward()
    vr = vol.clone().requires_grad_()
    (O.boundary_pad(O.convert_volume_to_subvolume(vr, (27, 2, 4, 4, 4)), 3) * cf(wgt)).sum().backward()
    close(cf(v.grad), vr.grad, what="halo adjoint")


def test_trilinear_softmax_bmm(ops):
    g = torch.Generator().manual_seed(12)
    x = torch.randn(1, 6, 2, 3, 2, generator=g)
    xr = x.double().requires_grad_()
    yr = F.interpolate(xr, scale_factor=4, mode='trilinear', align_corners=True)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy.double())
    xd = cl(x).requires_grad_()
    y = ops.trilinear_upsample(xd, 4)
    close(cf(y), yr, what="trilinear")
    y.backward(cl(dy))
    close(cf(xd.grad), xr.grad, tol=1e-4, what="trilinear adjoint")

    s = torch.randn(7, 33, 5, generator=g)
    for dim in (-1, 1, 0):
        sr = s.double().requires_grad_()
        pr = sr.softmax(dim=dim) * 0.25
        dp = torch.randn(pr.shape, generator=g)
        pr.backward(dp.double())
        sd = s.to(DEV).requires_grad_()
        p = ops.softmax(sd, dim, scale=0.25)
        close(p, pr, what=f"softmax dim {dim}")
        p.backward(dp.to(DEV))
        close(sd.grad, sr.grad, tol=1e-4, what=f"softmax bwd dim {dim}")

    A, Bm = torch.randn(3, 70, 45, generator=g), torch.randn(3, 45, 66, generator=g)
    for tA in (False, True):
        for tB in (False, True):
            Ar, Br = A.double().requires_grad_(), Bm.double().requires_grad_()
            Cr = 0.5 * torch.bmm(Ar.transpose(1, 2) if False else Ar, Br)
            Ain = A.transpose(1, 2).contiguous() if tA else A
            Bin = Bm.transpose(1, 2).contiguous() if tB else Bm
            Ad, Bd = Ain.to(DEV).requires_grad_(), Bin.to(DEV).requires_grad_()
            C = ops.bmm(Ad, Bd, tA, tB, 0.5)
            close(C, Cr, what=f"bmm {tA}{tB}")
            dC = torch.randn(Cr.shape, generator=g)
            Cr.backward(dC.double())
            C.backward(dC.to(DEV))
            close(Ad.grad, Ar.grad.transpose(1, 2) if tA else Ar.grad, tol=1e-4, what=f"bmm dA {tA}{tB}")
            close(Bd.grad, Br.grad.transpose(1, 2) if tB else Br.grad, tol=1e-4, what=f"bmm dB {tA}{tB}")


def test_diffusion_steps_loss_adam(ops):
    from oracle import iqt_oracle as O
    g = torch.Generator().manual_seed(13)
    B, n = 3, 4 * 5 * 6
    x0, noise, pred = (torch.randn(B, 1, 4, 5, 6, generator=g) for _ in range(3))
    t, tn = torch.tensor([0.9, 0.5, 0.1]), torch.tensor([0.8, 0.25, 0.0])
    xt, _, alpha, sigma = O.q_sample(x0, t, noise)
    close(ops.q_sample(x0.to(DEV), noise.to(DEV), alpha.flatten().to(DEV), sigma.flatten().to(DEV)), xt, what="q_sample")
    from diffusioniqt_amd.imagen_pytorch3D import GaussianDiffusionContinuousTimes
    sched = GaussianDiffusionContinuousTimes(noise_schedule='cosine', timesteps=10)
    ca, cb, cn = sched.posterior_coefficients(t, tn)
    lo = -0.72
    x_next, x0c = ops.ddpm_step(xt.to(DEV), pred.to(DEV), noise.to(DEV), ca.to(DEV), cb.to(DEV), cn.to(DEV), lo, 0.0, 0)
    xs = pred.clamp(min=lo)
    mean, _, logvar = O.q_posterior(xs, xt, t, tn)
    nz = (1 - (tn == 0).float()).reshape(B, 1, 1, 1, 1)
    close(x0c, xs, tol=1e-7, what="ddpm x0")
    close(x_next, mean + nz * (0.5 * logvar).exp() * noise, what="ddpm step")

    p = torch.randn(B, 1, 4, 5, 6, generator=g)
    pr = p.clone().requires_grad_()
    lr_ = ((pr.clamp(min=lo) - x0) ** 2).flatten(1).mean(1).mean()
    lr_.backward()
    pd = p.to(DEV).requires_grad_()
    loss, pc = ops.mse_clamp(pd, x0.to(DEV), lo=lo, do_clamp=True)
    close(loss, lr_, what="mse loss")
    close(pc, p.clamp(min=lo), tol=1e-7, what="clamped pred")
    (loss * 0.5).backward()
    close(pd.grad, 0.5 * pr.grad, what="mse grad")

    w, gr = torch.randn(1000, generator=g), torch.randn(1000, generator=g)
    wr = w.clone().requires_grad_()
    opt = torch.optim.Adam([wr], lr=1e-2, betas=(0.9, 0.99), eps=1e-8)
    wd, gd = w.to(DEV), gr.to(DEV)
    m, v = torch.zeros_like(wd), torch.zeros_like(wd)
    for step in range(1, 4):
        wr.grad = gr.clone() * step
        opt.step()
        gd2 = gd * step
        ops.adam_step(wd, gd2, m, v, 1e-2, 0.9, 0.99, 1e-8, 0.0, step, zero_grad=True)
        assert float(gd2.abs().max()) == 0.0
    close(wd, wr, tol=1e-5, what="adam")
    e = torch.randn(1000, generator=g)
    ed = e.to(DEV)
    ops.ema_lerp(ed, wd, 0.25)
    close(ed, e + (wr.detach() - e) * 0.25, tol=1e-5, what="ema")


@pytest.mark.parametrize("g_,M,N,K", [(2, 300, 70, 45), (3, 129, 33, 64), (1, 1024, 2053, 64), (2, 513, 64, 2049), (4, 256, 64, 33),
                                      (2, 5, 64, 2048), (1, 1, 64, 5000), (3, 33, 64, 256), (2, 200, 64, 4100), (2, 64, 70, 1300)])
def test_bgemm_large_tiles_all_layouts(ops, g_, M, N, K):
    """128x64-tile batched GEMM (attention QK^T / PV shapes, odd extents, 16-byte aligned and unaligned leading dims)."""
    gen = torch.Generator().manual_seed(M + N + K)
    A, Bm = torch.randn(g_, M, K, generator=gen), torch.randn(g_, K, N, generator=gen)
    ref = torch.bmm(A.double(), Bm.double())
    for tA in (False, True):
        for tB in (False, True):
            Ain = (A.transpose(1, 2).contiguous() if tA else A).to(DEV).requires_grad_()
            Bin = (Bm.transpose(1, 2).contiguous() if tB else Bm).to(DEV).requires_grad_()
            C = ops.bmm(Ain, Bin, tA, tB, 1.0)
            close(C, ref, tol=3e-5, what=f"bgemm {tA}{tB} {M}x{N}x{K}")
    dC = torch.randn(g_, M, N, generator=gen)
    Ar, Br = A.double().requires_grad_(), Bm.double().requires_grad_()
    torch.bmm(Ar, Br).backward(dC.double())
    Ad, Bd = A.to(DEV).requires_grad_(), Bm.to(DEV).requires_grad_()
    ops.bmm(Ad, Bd, False, False, 1.0).backward(dC.to(DEV))
    close(Ad.grad, Ar.grad, tol=1e-4, what="bgemm dA")
    close(Bd.grad, Br.grad, tol=1e-4, what="bgemm dB")


@pytest.mark.parametrize("B,Cin,Cout,expect", [(2, 40, 128, 3), (4, 24, 128, 3), (8, 64, 64, 3), (1, 32, 64, 0)])
def test_conv3d_on_the_8_wave_kernel_ragged_tiles_and_persistent_walk(ops, B, Cin, Cout, expect):
    """conv_fwd8_kernel (256-voxel tiles, 512 threads): ragged tile edges in all three axes (15 x 30 x 31), a ragged K-chunk, two Cout
    blocks, exactly one round of workgroups (B = 2: 256) and the persistent tile walk (B = 4, 8: 512 tiles on 256 workgroups), with
    bias, residual and the epilogue's per-tile column sums -- against float64 on the CPU.  The last case stays on conv_fwd_kernel."""
    from diffusioniqt_amd import _lib
    D, H, W = 15, 30, 31
    assert _lib.query("diqt_conv3d_fwd_kernel_id", B, D, H, W, Cin, Cout, 3, 3, 3, 1, 1, 1, 0, 0, 0) == expect
    gen = torch.Generator().manual_seed(B * 7 + Cin)
    x = torch.randn(B, Cin, D, H, W, generator=gen)
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=gen) * 0.05
    bias = torch.randn(Cout, generator=gen)
    res = torch.randn(B, Cout, D, H, W, generator=gen)
    ref = F.conv3d(x.double(), w.double(), bias.double(), padding=1) + res.double()
    with torch.no_grad():
        y = ops.conv3d(cl(x), w.to(DEV), bias.to(DEV), (1, 1, 1), residual=cl(res), want_stats=True)
    close(cf(y), ref, what="conv on the 8-wave kernel")
    st = getattr(y, "_diqt_stats", None)
    assert st is not None
    sums = st.partials.double().sum(dim=1).cpu()                                       # [B, 2, Cout]
    close(sums[:, 0], ref.sum(dim=(2, 3, 4)), tol=1e-5, what="epilogue column sums")
    close(sums[:, 1], (ref ** 2).sum(dim=(2, 3, 4)), tol=1e-5, what="epilogue column sums of squares")


def test_gate_residual_emits_groupnorm_statistics(ops):
    """se_gate_residual attaches per-workgroup column sums of its OUTPUT; the next GroupNorm finalises its statistics from them
    (no pass over the tensor) and must produce what it produces from a statistics pass over the same tensor."""
    from diffusioniqt_amd import _lib
    gen = torch.Generator().manual_seed(31)
    for B, S, C in ((2, 8, 64), (3, 6, 128), (1, 5, 32), (2, 4, 48)):              # 48 does not divide 1024: no statistics
        h = torch.randn(B, S, S, S, C, generator=gen).to(DEV)
        res = torch.randn(B, S, S, S, C, generator=gen).to(DEV)
        w1 = (torch.randn(max(C // 16, 1), C, generator=gen) * 0.2).to(DEV)
        w2 = (torch.randn(C, max(C // 16, 1), generator=gen) * 0.2).to(DEV)
        gamma, beta = torch.randn(C, generator=gen).to(DEV), torch.randn(C, generator=gen).to(DEV)
        with torch.no_grad():
            y = ops.se_gate_residual(h, w1, w2, res)
            st = getattr(y, "_diqt_stats", None)
            assert (st is not None) == (_lib.query("diqt_gate_residual_stats_blocks", S ** 3, C) > 0) == (1024 % C == 0)
            ref = h * torch.sigmoid(torch.relu(h.mean(dim=(1, 2, 3)) @ w1.t()) @ w2.t())[:, None, None, None, :] + res
            close(y, ref, tol=3e-5, what="gate residual")
            if st is None:
                continue
            sums = st.partials.double().sum(dim=1)                                      # [B, 2, C]
            close(sums[:, 0], y.double().sum(dim=(1, 2, 3)), tol=1e-5, what="column sums")
            close(sums[:, 1], (y.double() ** 2).sum(dim=(1, 2, 3)), tol=1e-5, what="column sums of squares")
            a = ops.groupnorm_act(y, gamma, beta, None, 8, ops.ACT_MISH)               # statistics from the partials
            yc = y.clone()                                                              # same values, no partials attached
            bfull = ops.groupnorm_act(yc, gamma, beta, None, 8, ops.ACT_MISH)
            close(a, bfull, tol=2e-5, what="GroupNorm from gate_residual partials")


def test_softmax_over_tokens_with_few_columns(ops):
    """LinearAttention's k.softmax(dim=-2) on [b*h, n, d] (imagen_pytorch3D.py:926-1016): the workgroup-per-outer kernel (inner divides
    1024, n >= 64) and the one-thread-per-column kernel next to it."""
    gen = torch.Generator().manual_seed(21)
    for outer, n, inner in ((8, 512, 64), (3, 100, 32), (2, 64, 1024), (2, 70, 48)):
        x = torch.randn(outer, n, inner, generator=gen) * 3
        got = ops.softmax(x.to(DEV), dim=1, scale=0.125)
        close(got, 0.125 * torch.softmax(x.double(), dim=1), tol=3e-5, what=f"softmax over n {outer}x{n}x{inner}")


def test_softmax_few_long_rows(ops):
    """GlobalContext soft-max over all positions: 8 rows of 32768 (one workgroup per row)."""
    gen = torch.Generator().manual_seed(5)
    for rows, n in ((8, 32768), (3, 5000), (1, 4096)):
        x = torch.randn(rows, n, generator=gen) * 3
        xd = x.to(DEV).requires_grad_()
        got = ops.softmax(xd, dim=-1, scale=0.5)
        xr = x.double().requires_grad_()
        want = 0.5 * torch.softmax(xr, dim=-1)
        close(got, want, tol=3e-5, what=f"softmax {rows}x{n}")
        dy = torch.randn(rows, n, generator=gen)
        want.backward(dy.double())
        got.backward(dy.to(DEV))
        close(xd.grad, xr.grad, tol=1e-4, what=f"softmax bwd {rows}x{n} (one workgroup per row)")


def test_weighted_pool_global_context(ops):
    """GlobalContext pooling (imagen_video.py:975-979): out[b,c] = sum_n softmax(ctx)[b,n] x[b,n,c]."""
    gen = torch.Generator().manual_seed(21)
    for (B, n, C) in ((2, 4096, 64), (3, 777, 20), (1, 32768, 128)):
        w = torch.softmax(torch.randn(B, n, generator=gen), dim=-1)
        x = torch.randn(B, n, C, generator=gen)
        up = torch.randn(B, C, generator=gen)
        wr, xr = w.double().requires_grad_(), x.double().requires_grad_()
        torch.einsum('bn,bnc->bc', wr, xr).backward(up.double())
        wd, xd = w.to(DEV).requires_grad_(), x.to(DEV).requires_grad_()
        out = ops.weighted_pool(wd, xd)
        close(out, torch.einsum('bn,bnc->bc', w.double(), x.double()), tol=3e-5, what="weighted pool")
        out.backward(up.to(DEV))
        close(wd.grad, wr.grad, tol=1e-4, what="weighted pool dw")
        close(xd.grad, xr.grad, tol=1e-4, what="weighted pool dx")


@pytest.mark.parametrize("G,n,h,d,E,use_rel,causal", [(2, 40, 4, 64, 1, False, False), (3, 16, 8, 64, 1, True, True),
                                                       (1, 300, 2, 32, 3, False, False), (2, 33, 3, 32, 1, True, False),
                                                       (1, 1, 8, 64, 1, True, True), (1, 2048, 8, 64, 5, False, False)])
def test_fused_mqa_attention_forward(ops, G, n, h, d, E, use_rel, causal):
    """diqt_mqa_attention_fwd == softmax(scale q k^T + rel bias / null bias, causal mask) v (imagen_video.py:483-520) in fp64."""
    gen = torch.Generator().manual_seed(G * 100 + n)
    q = torch.randn(G, n, h * d, generator=gen)
    kv = torch.randn(G, E + n, 2 * d, generator=gen)
    rel = torch.randn(2 * n - 1, h, generator=gen) if use_rel else None
    nb = torch.randn(h, generator=gen) if use_rel else None
    scale = d ** -0.5
    qd = q.double().reshape(G, n, h, d)
    k, v = kv.double()[..., :d], kv.double()[..., d:]
    sim = torch.einsum('gihd,gjd->gihj', qd, k) * scale
    if use_rel:
        i = torch.arange(n)[:, None]; j = torch.arange(n)[None, :]
        sim[..., E:] += rel.double()[(i - j + n - 1)].permute(0, 2, 1)[None]          # [n, h, n] indexed (i, hh, j)
        sim[..., E - 1] += nb.double()[None, None, :]
    if causal:
        i = torch.arange(n)[:, None]; j = torch.arange(n)[None, :]
        mask = (j > i)[None, :, None, :].expand(G, n, h, n)
        sim[..., E:] = sim[..., E:].masked_fill(mask, float('-inf'))
    ref = torch.einsum('gihj,gjd->gihd', sim.softmax(dim=-1), v).reshape(G, n, h * d)
    got = ops.mqa_attention_nograd(q.to(DEV), kv.to(DEV), rel.to(DEV) if use_rel else None, nb.to(DEV) if use_rel else None,
                                   n, h, d, E, n, causal, scale)
    close(got, ref, tol=3e-5, what="fused MQA attention")


def _mqa_ref(q, kv, rel, nb, n, h, d, E, causal, scale):
    """float64 restatement of Attention.forward's products (imagen_video.py:483-520) on leaf tensors that require grad"""
    G = q.shape[0]
    qd = q.reshape(G, n, h, d)
    k, v = kv[..., :d], kv[..., d:]
    sim = torch.einsum('gihd,gjd->gihj', qd, k) * scale
    if rel is not None:
        i = torch.arange(n)[:, None]; j = torch.arange(n)[None, :]
        bias = rel[(i - j + n - 1)].permute(0, 2, 1)[None]                                  # [1, n, h, n] indexed (i, hh, j)
        sim = torch.cat((sim[..., :E - 1], sim[..., E - 1:E] + nb[None, None, :, None], sim[..., E:] + bias), dim=-1)
    if causal:
        i = torch.arange(n)[:, None]; j = torch.arange(n)[None, :]
        mask = torch.cat((torch.zeros(n, E, dtype=torch.bool), j > i), dim=1)[None, :, None, :]
        sim = sim.masked_fill(mask, float('-inf'))
    return torch.einsum('gihj,gjd->gihd', sim.softmax(dim=-1), v).reshape(G, n, h * d)


@pytest.mark.parametrize("G,n,h,d,E,use_rel,causal", [
    (3, 32, 8, 64, 1, True, True),       # temporal attention of Unet3D: 32 frames + null key, causal, relative bias: key tile + VALU null key
    (2, 64, 8, 64, 1, False, False),     # 8x8 spatial attention: 2 key tiles, the query tiles split over wave pairs
    (2, 40, 4, 64, 1, False, False),     # ragged key / query tiles
    (2, 33, 3, 32, 1, True, False),      # dim_head 32, heads that do not divide the wave, relative bias without mask
    (1, 150, 2, 32, 3, False, False),    # context tokens in front of the null key (E = 3): all keys through the tile loop; > 128 keys
    (2, 16, 8, 64, 2, True, True),       # E = 2 with bias and mask
    (2, 170, 4, 64, 5, True, True),      # E = 5 in front of 6 key tiles: every extra key is VALU work of one of the first five tiles (null bias on the last)
    (1, 70, 8, 64, 3, False, True),      # ... E = 3 = the number of key tiles, causal
    (300, 8, 2, 32, 1, True, True),      # more batch entries than resident workgroups: the persistent walk of the dQ kernel
    (2051, 12, 2, 32, 1, True, True),    # thousands of short sequences: the dK/dV kernel gives every WAVE a sequence (ragged last workgroup)
    (2048, 20, 4, 64, 1, False, False),  # ... dim_head 64, no bias / mask
    (2100, 32, 8, 64, 1, True, True),    # the temporal attention at its real shape: the one-pass short-sequence kernel (mqa_seq_bwd_kernel)
    (2300, 7, 2, 32, 1, True, False),    # ... ragged query tile (14 rows), bias without mask, dim_head 32
    (515, 32, 8, 64, 1, True, True),     # ... the 8 x 8 level of Unet3D (512 sequences): fewer workgroups than CUs, ragged last workgroup
    (1, 1, 8, 64, 1, True, True)])
def test_fused_mqa_attention_backward(ops, G, n, h, d, E, use_rel, causal):
    """diqt_mqa_attention_fwd_lse / diqt_mqa_attention_bwd (flash-style: no stored scores) against float64 autograd of the
    reference formula: out, dq, dkv (incl. the null / context key rows), d rel-bias table, d null bias; run twice: bit-identical."""
    gen = torch.Generator().manual_seed(G * 131 + n)
    q = torch.randn(G, n, h * d, generator=gen)
    kv = torch.randn(G, E + n, 2 * d, generator=gen)
    rel = torch.randn(2 * n - 1, h, generator=gen) if use_rel else None
    nb = torch.randn(h, generator=gen) if use_rel else None
    up = torch.randn(G, n, h * d, generator=gen)
    scale = d ** -0.5
    leaf = lambda t: t.double().requires_grad_() if t is not None else None
    qr, kvr, relr, nbr = leaf(q), leaf(kv), leaf(rel), leaf(nb)
    ref = _mqa_ref(qr, kvr, relr, nbr, n, h, d, E, causal, scale)
    ref.backward(up.double())
    runs = []
    for _ in range(2):
        dl = lambda t: t.to(DEV).requires_grad_() if t is not None else None
        qd, kvd, reld, nbd = dl(q), dl(kv), dl(rel), dl(nb)
        out = ops.mqa_attention(qd, kvd, reld, nbd, n, h, d, E, n, causal, scale)
        out.backward(up.to(DEV))
        runs.append((out, qd.grad, kvd.grad, reld.grad if use_rel else None, nbd.grad if use_rel else None))
    out, dq, dkv, drel, dnb = runs[0]
    close(out, ref, tol=3e-5, what="fused MQA attention (training forward)")
    close(dq, qr.grad, tol=1e-4, what="dq")
    close(dkv, kvr.grad, tol=1e-4, what="dkv")
    if use_rel:
        close(drel, relr.grad, tol=1e-4, what="d rel-bias table")
        close(dnb, nbr.grad, tol=1e-4, what="d null bias")
    for a, b in zip(runs[0], runs[1]):
        assert a is None or torch.equal(a, b), "fused attention backward is not run-to-run deterministic"


@pytest.mark.parametrize("f,A,Cin,Cout,k", [(3, 4, 16, 16, 3),      # 27 x 4^3 (the reference fixture's size): 4-wave kernel, split-K levels
                                            (3, 8, 64, 64, 3),      # 27 x 8^3
                                            (3, 16, 2, 16, 3),      # init conv: 2 input channels (tap-packed kernel)
                                            (2, 16, 20, 24, 3),     # ragged channels, factor 2
                                            (3, 32, 64, 64, 3),     # 27 x 32^3 (one 96^3 block of eval_config.yaml): 8-wave kernel, persistent walk
                                            (3, 8, 32, 32, 5)])     # 5^3 filter: a 2-voxel halo from the neighbours
def test_conv3d_neighbour_halo_equals_boundary_pad_copies(ops, f, A, Cin, Cout, k):
    """diqt_conv3d_fwd_neighbours (halo voxels read in place from the neighbouring sub-volumes) is bit-identical to
    the reference's formulation -- merge_sub_volumes -> zero pad -> overlapping split (boundary_pad, imagen_pytorch3D.py:37-46) ->
    unpadded conv -- incl. bias, residual epilogue and the per-tile output statistics."""
    gen = torch.Generator().manual_seed(f * 1000 + A)
    B, p = f ** 3, k // 2
    x = torch.randn(B, A, A, A, Cin, generator=gen).to(DEV)
    w = (torch.randn(Cout, Cin, k, k, k, generator=gen) / math.sqrt(Cin * k ** 3)).to(DEV)
    b = torch.randn(Cout, generator=gen).to(DEV)
    r = torch.randn(B, A, A, A, Cout, generator=gen).to(DEV)
    with torch.no_grad():
        padded = ops.split_volume(ops.merge_volume(x, f), f, A, halo=p)                 # [f^3, A + 2p, ...]: the copies
        from diffusioniqt_amd import _lib
        ref = ops.conv3d(padded, w, b, (0, 0, 0), residual=r)
        got = ops.conv3d_neighbours(x, w, b, f, residual=r)
        # same kernel on both sides -> same bits; the copies of a 96^3 block are big enough for conv_fwd9_kernel (another summation
        # order over K), which has no neighbour addressing: there the two agree to fp32 round-off
        same_kernel = _lib.query("diqt_conv3d_fwd_kernel_id", B, A + 2 * p, A + 2 * p, A + 2 * p, Cin, Cout, k, k, k, 0, 0, 0, 0, 0, 0) == \
            _lib.query("diqt_conv3d_fwd_kernel_id", B, A, A, A, Cin, Cout, k, k, k, p, p, p, 0, 0, 0) != 4
        got2 = ops.conv3d_neighbours(x, w, b, f, want_stats=True)
        ref2 = ops.conv3d(padded, w, b, (0, 0, 0))
        if same_kernel:
            assert torch.equal(got, ref), f"max diff {(got - ref).abs().max().item():.3e}"
            assert torch.equal(got2, ref2)
        else:
            close(got, ref, tol=2e-6, what="neighbour conv vs copies (different kernels)")
            close(got2, ref2, tol=2e-6, what="neighbour conv vs copies (different kernels)")
        st = getattr(got2, "_diqt_stats", None)
        if st is not None:                                                                # the consumer's GroupNorm statistics ride along
            sums = st.partials[:, :, 0, :].double().sum(1)
            close(sums, got2.double().sum(dim=(1, 2, 3)), tol=1e-5, what="per-tile column sums")
    # against a float64 conv of the merged, zero-padded volume
    vol = ops.merge_volume(x, f)[0].permute(3, 0, 1, 2).double().cpu()[None]
    full = F.conv3d(vol, w.double().cpu(), b.double().cpu(), padding=p)
    want = ops.split_volume(full[0].permute(1, 2, 3, 0)[None].float().contiguous().to(DEV), f, A, 0)
    close(ops.conv3d_neighbours(x, w, b, f), want, what="neighbour conv vs float64 conv of the merged volume")


@pytest.mark.parametrize("B,sp,Cin,Cout,k,pad,res", [
    (4, (32, 32, 32), 64, 64, (3, 3, 3), 1, False),      # exactly one round of 256 workgroups (no tile walk)
    (8, (32, 32, 32), 64, 64, (3, 3, 3), 1, True),       # the headline's dominant launch: persistent walk, 2 tiles per workgroup, 4 chunks; residual
    (8, (32, 32, 30), 32, 128, (3, 3, 3), 1, False),     # ragged W tiles, two 64-channel blocks (the workgroup keeps its block over the walk)
    (8, (34, 34, 34), 16, 48, (3, 3, 3), 0, True),       # un-padded ('boundary' copies): no halo outside the volume; one chunk; ragged Cout
    (12, (31, 32, 32), 48, 64, (3, 3, 3), 1, False),     # 3 chunks, ragged D, 3 tiles per workgroup
    (8, (16, 16, 16), 128, 128, (3, 3, 3), 1, True),     # the 16^3 level: 256-voxel tiles (4 x 8 x 8), one round
    (16, (15, 16, 16), 32, 128, (3, 3, 3), 1, False),    # 256-voxel tiles: ragged D, walk of 2 tiles
    (8, (32, 32, 32), 64, 64, (1, 3, 3), 1, True),       # Family B full resolution: (1,3,3) filter, 1 x 16 x 32 tiles, walk of 2
    (8, (32, 30, 32), 32, 64, (1, 3, 3), 1, False),      # ... ragged H
    (8, (32, 34, 34), 16, 48, (1, 3, 3), 0, False),      # ... un-padded, ragged Cout
    (8, (32, 16, 16), 128, 128, (1, 3, 3), 1, True),     # Family B second level: 2 x 16 x 16 tiles
    (8, (32, 8, 8), 256, 256, (1, 3, 3), 1, False),      # Family B third level: 4 x 8 x 8 tiles of 256 voxels, four 64-channel blocks
    (8, (8, 8, 8), 256, 256, (3, 3, 3), 1, True),        # the 8^3 level: 64 tiles x co blocks -> split-K over 4 x 4 chunks, slabs + reduce
    (8, (8, 8, 8), 384, 256, (3, 3, 3), 1, False),       # ... 24 chunks in 4 shares (the decoder's concatenated input)
    (2, (16, 16, 16), 64, 128, (3, 3, 3), 1, True),      # ... one chunk per share
    (4, (16, 8, 8), 256, 224, (1, 3, 3), 1, False),      # (1,3,3) split-K, ragged Cout
    (8, (32, 32, 32), 64, 64, (3, 1, 1), 1, True),       # Family B temporal conv: 3 taps = 2 steps per chunk, 8 x 8 x 8 tiles, walk of 2
    (8, (32, 16, 16), 128, 128, (3, 1, 1), 1, False),    # ... second level, one round
    (8, (30, 8, 8), 256, 256, (3, 1, 1), 1, False)])     # ... third level: 4 x 8 x 8 tiles, ragged D
def test_conv3d_on_the_one_wave_per_simd_kernel(ops, B, sp, Cin, Cout, k, pad, res):
    """conv_fwd9_kernel (512- / 256-voxel tiles, LDS-DMA double-buffered 16-channel halo chunks, weight ring) against a float64 conv:
    output, residual epilogue, per-tile column sums, run-to-run determinism."""
    from diffusioniqt_amd import _lib
    D, H, W = sp
    pads = tuple(pad * (kk // 2) for kk in k)
    assert _lib.query("diqt_conv3d_fwd_kernel_id", B, D, H, W, Cin, Cout, *k, *pads, 0, 0, 0) == 4, "not routed to conv_fwd9_kernel"
    g = torch.Generator().manual_seed(B * 1000 + Cin)
    x = torch.randn(B, Cin, D, H, W, generator=g)
    w = torch.randn(Cout, Cin, *k, generator=g) / math.sqrt(Cin * k[0] * k[1] * k[2])
    b = torch.randn(Cout, generator=g) * 0.1
    Do, Ho, Wo = (n + 2 * p - kk + 1 for n, p, kk in zip(sp, pads, k))
    r = torch.randn(B, Cout, Do, Ho, Wo, generator=g) if res else None
    xd, wd, bd = cl(x), w.to(DEV), b.to(DEV)
    with torch.no_grad():
        y = ops.conv3d(xd, wd, bd, pads, residual=cl(r) if res else None, want_stats=True)
    # float64 reference on the host for the first / last batch entry and the first / last 8 output channels (the full conv in
    # float64 would take the CPU minutes); the column sums below tie the rest of the tensor to these
    bs, cs = [0, B - 1], list(range(8)) + list(range(Cout - 8, Cout))
    ref = F.conv3d(x[bs].double(), w[cs].double(), b[cs].double(), padding=pads)
    if res:
        ref = ref + r[bs][:, cs].double()
    close(cf(y)[bs][:, cs], ref, what="conv_fwd9 output")
    st = getattr(y, "_diqt_stats", None)
    split = _lib.query("diqt_conv3d_fwd_workspace_bytes", B, D, H, W, Cin, Cout, *k, *pads, 0, 0, 0) > 0
    assert (st is None) == split, "statistics come from un-split launches only"
    if st is not None:
        assert st.rows == y.shape[1] * y.shape[2] * y.shape[3]
        close(st.partials[:, :, 0, :].double().sum(1), y.double().sum(dim=(1, 2, 3)), tol=1e-5, what="per-tile column sums")
        close(st.partials[:, :, 1, :].double().sum(1), (y.double() ** 2).sum(dim=(1, 2, 3)), tol=1e-5, what="per-tile sums of squares")
    with torch.no_grad():
        y2 = ops.conv3d(xd, wd, bd, pads, residual=cl(r) if res else None)
    assert torch.equal(y, y2), "conv_fwd9 is not run-to-run deterministic"
    # the middle batch entries and channels: a batch-rotated input must give the batch-rotated output, bit for bit (tiles of different
    # workgroups / walk positions compute the same voxels), which ties every batch entry to the two checked in float64 ...
    with torch.no_grad():
        yr = ops.conv3d(torch.roll(xd, 1, 0).contiguous(), wd, bd, pads, residual=torch.roll(cl(r), 1, 0).contiguous() if res else None)
    assert torch.equal(torch.roll(y, 1, 0), yr), "conv_fwd9: a tile's result depends on which workgroup computed it"
    # ... and an output-channel rotation of the filter ties every channel to the 16 checked
    with torch.no_grad():
        yc = ops.conv3d(xd, torch.roll(wd, 8, 0).contiguous(), torch.roll(bd, 8, 0).contiguous(), pads,
                        residual=torch.roll(cl(r), 8, -1).contiguous() if res else None)
    assert torch.equal(torch.roll(y, 8, -1), yc), "conv_fwd9: a channel's result depends on its position in the 64-channel block"


@pytest.mark.parametrize("B,sp,C", [(8, (32, 32, 32), 64), (8, (32, 8, 8), 256)])
def test_causal_temporal_conv_on_the_one_wave_per_simd_kernel(ops, B, sp, C):
    """The pseudo-3D blocks' temporal conv is CAUSAL: (3,1,1) taps over frames f-2..f (left pad 2, no right pad; imagen_video.py:399-402),
    expressed as padding 2 with extra_pad -2.  conv_fwd9_kernel's (3,1,1) variants against a float64 conv of the left-padded input,
    forward and (through autograd: the anti-causal flipped conv on the same kernel) the input gradient."""
    from diffusioniqt_amd import _lib
    D, H, W = sp
    assert _lib.query("diqt_conv3d_fwd_kernel_id", B, D, H, W, C, C, 3, 1, 1, 2, 0, 0, -2, 0, 0) == 4, "not routed to conv_fwd9_kernel"
    g = torch.Generator().manual_seed(C)
    x = torch.randn(B, C, D, H, W, generator=g)
    w = torch.randn(C, C, 3, 1, 1, generator=g) / math.sqrt(3 * C)
    b = torch.randn(C, generator=g) * 0.1
    dy = torch.randn(B, C, D, H, W, generator=g)
    bs, cs = [0, B - 1], list(range(8)) + list(range(C - 8, C))
    xr = x[bs].double().requires_grad_()
    ref = F.conv3d(F.pad(xr, (0, 0, 0, 0, 2, 0)), w.double(), b.double())
    ref.backward(dy[bs].double())
    xd = cl(x).requires_grad_()
    y = ops.conv3d(xd, w.to(DEV), b.to(DEV), (2, 0, 0), extra_pad=(-2, 0, 0))
    close(cf(y)[bs], ref, what="causal temporal conv")
    y.backward(cl(dy))
    close(cf(xd.grad)[bs], xr.grad, what="causal temporal conv: input gradient")


@pytest.mark.parametrize("B,F_,H,W,C,causal,res", [(2, 32, 8, 8, 64, True, True), (3, 11, 5, 4, 128, False, False), (1, 7, 3, 3, 256, True, True),
                                                   (2, 16, 6, 6, 32, False, True)])
def test_depthwise_temporal_conv_matches_grouped_conv3d(ops, B, F_, H, W, C, causal, res):
    """The TemporalPEG conv -- nn.Conv3d(C, C, (3,1,1), groups=C) after a causal (2,0) / symmetric (1,1) frame pad, + residual
    (imagen_video.py:1340-1362) -- on the elementwise kernels: forward, and the gradients w.r.t. x, weight, bias and residual."""
    g = torch.Generator().manual_seed(C + F_)
    x = torch.randn(B, C, F_, H, W, generator=g)
    w = torch.randn(C, 1, 3, 1, 1, generator=g) * 0.5
    b = torch.randn(C, generator=g)
    r = torch.randn(B, C, F_, H, W, generator=g) if res else None
    dy = torch.randn(B, C, F_, H, W, generator=g)
    xr, wr, br = x.double().requires_grad_(), w.double().requires_grad_(), b.double().requires_grad_()
    rr = r.double().requires_grad_() if res else None
    pad = (0, 0, 0, 0, 2, 0) if causal else (0, 0, 0, 0, 1, 1)
    ref = F.conv3d(F.pad(xr, pad), wr, br, groups=C)
    if res:
        ref = ref + rr
    ref.backward(dy.double())
    xd, wd, bd = cl(x).requires_grad_(), w.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
    rd = cl(r).requires_grad_() if res else None
    assert ops.dwconv_temporal_ok(xd, wd, C)
    y = ops.dwconv_temporal(xd, wd, bd, 2 if causal else 1, rd)
    close(cf(y), ref, what="depthwise temporal conv")
    y.backward(cl(dy))
    close(cf(xd.grad), xr.grad, what="d x")
    close(wd.grad, wr.grad, tol=1e-4, what="d weight")
    close(bd.grad, br.grad, tol=1e-4, what="d bias")
    if res:
        close(cf(rd.grad), rr.grad, what="d residual")


def test_pointwise_weight_gradient_routes_to_the_dma_kernel(ops):
    """1x1x1 convs and Linear layers (= one long row axis) whose row count is a multiple of 64: the weight gradient runs on
    conv_wgrad3_kernel's pointwise variant (kernel id 3), other row counts stay on the batched-GEMM path (0); Linear gradients vs float64."""
    from diffusioniqt_amd import _lib
    assert _lib.query("diqt_conv3d_bwd_weight_kernel_id", 8, 32, 32, 32, 64, 512, 1, 1, 1, 0, 0, 0, 0, 0, 0) == 3
    assert _lib.query("diqt_conv3d_bwd_weight_kernel_id", 1, 1, 1, 262144, 512, 64, 1, 1, 1, 0, 0, 0, 0, 0, 0) == 3
    assert _lib.query("diqt_conv3d_bwd_weight_kernel_id", 1, 1, 1, 4100, 128, 64, 1, 1, 1, 0, 0, 0, 0, 0, 0) == 0
    g = torch.Generator().manual_seed(77)
    for rows, Cin, Cout in ((8192, 64, 128), (4100, 128, 64)):
        x = torch.randn(4, rows // 4, Cin, generator=g)
        w = torch.randn(Cout, Cin, generator=g) / math.sqrt(Cin)
        b = torch.randn(Cout, generator=g)
        dy = torch.randn(4, rows // 4, Cout, generator=g)
        xr, wr, br = x.double().requires_grad_(), w.double().requires_grad_(), b.double().requires_grad_()
        F.linear(xr, wr, br).backward(dy.double())
        xd, wd, bd = x.to(DEV).requires_grad_(), w.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
        ops.linear(xd, wd, bd).backward(dy.to(DEV))
        close(xd.grad, xr.grad, what=f"linear dx rows={rows}")
        close(wd.grad, wr.grad, tol=1e-4, what=f"linear dw rows={rows}")
        close(bd.grad, br.grad, tol=1e-4, what=f"linear db rows={rows}")


@pytest.mark.parametrize("B,sp,Cin,Cout,k,act,with_ss", [
    (4, (32, 32, 32), 16, 64, (3, 3, 3), "mish", True),       # backward-data 64 -> 16 on 512-voxel tiles, one round
    (8, (16, 16, 16), 128, 64, (3, 3, 3), "mish", False),     # ... 256-voxel tiles (backward-data 64 -> 128: two channel blocks), no scale / shift
    (8, (32, 8, 8), 256, 64, (1, 3, 3), "silu", True)])       # the pseudo-3D block: SiLU, (1,3,3) filter, four 64-channel blocks
def test_block_backward_with_groupnorm_reduction_in_the_conv_epilogue(ops, B, sp, Cin, Cout, k, act, with_ss):
    """Block = GroupNorm -> (scale + 1) x + shift -> Mish / SiLU -> conv (imagen_pytorch3D.py:535-566, imagen_video.py:671-697).  The
    backward-data pass of the conv reduces the GroupNorm backward's per-channel sums in its epilogue (diqt_conv3d_fwd_gnbwd) and
    diqt_gn_act_bwd_from_partials finishes without a reduction pass: every gradient of the chain against float64 autograd."""
    from diffusioniqt_amd import _lib
    D, H, W = sp
    pads = tuple(kk // 2 for kk in k)
    bp = tuple(kk - 1 - p for kk, p in zip(k, pads))
    assert _lib.query("diqt_conv3d_fwd_gnbwd_blocks", B, D, H, W, Cout, Cin, *k, *bp, 0, 0, 0) == 0, "the fusion is off by default"
    with ops.gnbwd_fuse(True):
        assert _lib.query("diqt_conv3d_fwd_gnbwd_blocks", B, D, H, W, Cout, Cin, *k, *bp, 0, 0, 0) > 0, "backward-data pass not on conv_fwd9_kernel"
        _block_backward_case(ops, B, sp, Cin, Cout, k, act, with_ss)
    assert not _lib.query("diqt_get_gnbwd_fuse")
    _block_backward_case(ops, B, sp, Cin, Cout, k, act, with_ss)            # and the default: reduction in its own pass


def _block_backward_case(ops, B, sp, Cin, Cout, k, act, with_ss):
    D, H, W = sp
    pads = tuple(kk // 2 for kk in k)
    g = torch.Generator().manual_seed(Cin * 7 + Cout)
    x = torch.randn(B, Cin, D, H, W, generator=g) * 1.5 + 0.3
    gamma, beta = torch.randn(Cin, generator=g), torch.randn(Cin, generator=g) * 0.3
    ss = torch.randn(B, 2 * Cin, generator=g) * 0.3 if with_ss else None
    w = torch.randn(Cout, Cin, *k, generator=g) / math.sqrt(Cin * k[0] * k[1] * k[2])
    bias = torch.randn(Cout, generator=g) * 0.1
    dy = torch.randn(B, Cout, D, H, W, generator=g)
    # float64 reference
    xr, gr, br, wr, cr = (t.double().requires_grad_() for t in (x, gamma, beta, w, bias))
    sr = ss.double().requires_grad_() if with_ss else None
    h = F.group_norm(xr, 8, gr, br, eps=1e-5)
    if with_ss:
        h = h * (sr[:, :Cin, None, None, None] + 1) + sr[:, Cin:, None, None, None]
    h = F.mish(h) if act == "mish" else F.silu(h)
    F.conv3d(h, wr, cr, padding=pads).backward(dy.double())
    # device
    xd = cl(x).requires_grad_()
    gd, bd, wd, cd = (t.to(DEV).requires_grad_() for t in (gamma, beta, w, bias))
    sd = ss.to(DEV).requires_grad_() if with_ss else None
    a = ops.groupnorm_act(xd, gd, bd, sd, 8, ops.ACT_MISH if act == "mish" else ops.ACT_SILU)
    assert getattr(a, "_diqt_gnctx", None) is not None, "GroupNorm context not published to the conv"
    y = ops.conv3d(a, wd, cd, pads)
    y.backward(cl(dy))
    close(cf(xd.grad), xr.grad, tol=5e-5, what="d x through conv + GroupNorm")
    close(gd.grad, gr.grad, tol=1e-4, what="d gamma")
    close(bd.grad, br.grad, tol=1e-4, what="d beta")
    close(wd.grad, wr.grad, tol=1e-4, what="d weight")
    if with_ss:
        close(sd.grad, sr.grad, tol=1e-4, what="d scale / shift")
    # two consumers of the activated tensor: autograd sums their gradients in place, the partial sums of one branch must not be used
    xd2 = cl(x).requires_grad_()
    a2 = ops.groupnorm_act(xd2, gd.detach(), bd.detach(), sd.detach() if with_ss else None, 8, ops.ACT_MISH if act == "mish" else ops.ACT_SILU)
    y2 = ops.conv3d(a2, wd.detach(), cd.detach(), pads)
    (y2.float() * cl(dy)).sum().add((a2 * 0.5).sum()).backward()
    xr2 = x.double().requires_grad_()
    h2 = F.group_norm(xr2, 8, gamma.double(), beta.double(), eps=1e-5)
    if with_ss:
        h2 = h2 * (ss.double()[:, :Cin, None, None, None] + 1) + ss.double()[:, Cin:, None, None, None]
    h2 = F.mish(h2) if act == "mish" else F.silu(h2)
    ((F.conv3d(h2, w.double(), bias.double(), padding=pads) * dy.double()).sum() + (h2 * 0.5).sum()).backward()
    close(cf(xd2.grad), xr2.grad, tol=5e-5, what="d x with a second consumer of the activated tensor")


@pytest.mark.parametrize("rows_shape,Cout,res", [((2, 16, 16, 16), 512, True), ((1, 17, 17, 17), 264, False), ((1, 1, 1, 4100), 256, True)])
def test_pointwise_conv_with_64_input_channels_on_the_resident_kernel(ops, rows_shape, Cout, res):
    """conv1x1_k64_kernel (x tile resident, every 64-channel block of the output from one workgroup, persistent row walk): the
    attention to_q / feed-forward shapes 64 -> 512 / 128, a ragged channel block, a ragged last row tile, bias and residual; float64."""
    B, D, H, W = rows_shape
    g = torch.Generator().manual_seed(Cout)
    x = torch.randn(B, 64, D, H, W, generator=g)
    w = torch.randn(Cout, 64, 1, 1, 1, generator=g) / 8.0
    b = torch.randn(Cout, generator=g)
    r = torch.randn(B, Cout, D, H, W, generator=g) if res else None
    ref = F.conv3d(x.double(), w.double(), b.double())
    if res:
        ref = ref + r.double()
    with torch.no_grad():
        y = ops.conv3d(cl(x), w.to(DEV), b.to(DEV), (0, 0, 0), residual=cl(r) if res else None)
        y2 = ops.conv3d(cl(x), w.to(DEV), b.to(DEV), (0, 0, 0), residual=cl(r) if res else None)
    close(cf(y), ref, what="pointwise conv, K = 64")
    assert torch.equal(y, y2)


def test_multi_accumulate_matches_per_tensor_adds(ops):
    """Gradient accumulation into the flat arena: one launch == the per-parameter `grad += new` adds (bit-exact)."""
    g = torch.Generator().manual_seed(11)
    sizes = [1, 3, 64, 1000, 27 * 64 * 64, 17, 4096 + 2]
    offs, off = [], 0
    for n in sizes:
        offs.append(off)
        off += (n + 3) // 4 * 4
    flat = torch.randn(off, generator=g)
    srcs = [torch.randn(n, generator=g) for n in sizes]
    ref = flat.clone()
    for t, o in zip(srcs, offs):
        ref[o:o + t.numel()] += t
    dflat = flat.to(DEV)
    ops.multi_accumulate(dflat, [t.to(DEV) for t in srcs], offs)
    assert torch.equal(dflat.cpu(), ref)


def test_ops_refuse_cpu_tensors(ops):
    with pytest.raises(RuntimeError):
        ops.mish(torch.randn(4))


@pytest.mark.parametrize("B,sp,Cin,Cout,k,act,with_ss,with_res", [
    (8, (32, 32, 32), 64, 64, (3, 3, 3), "mish", True, True),     # the headline shape: 512-voxel tiles, persistent walk (2 tiles per workgroup)
    (2, (32, 32, 32), 32, 64, (3, 3, 3), "mish", False, False),   # 256-voxel tiles, one round, two 16-channel chunks
    (8, (16, 16, 16), 128, 128, (3, 3, 3), "mish", True, False),  # 16^3 level: two 64-channel output blocks per tile
    (8, (8, 8, 8), 256, 256, (3, 3, 3), "mish", True, True),      # 8^3 level: split-K slabs + reduce (bias / residual in the reduce)
    (8, (32, 32, 32), 64, 64, (1, 3, 3), "silu", True, True),     # pseudo-3D per-frame conv, 1x16x32 tiles
    (8, (32, 16, 16), 128, 128, (1, 3, 3), "silu", False, True),  # 2x16x16 tiles
    (8, (32, 8, 8), 256, 64, (1, 3, 3), "silu", True, False)])    # 4x8x8 tiles
def test_groupnorm_apply_inside_the_conv_staging_matches_the_two_kernel_path(ops, B, sp, Cin, Cout, k, act, with_ss, with_res):
    """Sampling path: Block.forward (GroupNorm -> (scale + 1) x + shift -> Mish / SiLU -> conv, imagen_pytorch3D.py:546-566,
    imagen_video.py:680-697) as one conv launch whose input tiles are rewritten act(A x + Bc) in the LDS (``ops.gn_conv3d`` ->
    ``diqt_conv3d_fwd_gn``), against float64 and against the two-kernel path (``groupnorm_act`` + ``conv3d``); statistics of the
    producer (``_diqt_stats``) and of the output ride along in both."""
    D, H, W = sp
    pads = tuple(kk // 2 for kk in k)
    A = ops.ACT_MISH if act == "mish" else ops.ACT_SILU
    g = torch.Generator().manual_seed(Cin + 3 * Cout + k[0])
    x = torch.randn(B, Cin, D, H, W, generator=g) * 1.7 + 0.4
    x[:, :, :, :2] += 25.0 if act == "mish" else 0.0        # a few activations beyond Mish's x > 20 branch
    gamma, beta = torch.randn(Cin, generator=g), torch.randn(Cin, generator=g) * 0.3
    ss = torch.randn(B, 2 * Cin, generator=g) * 0.3 if with_ss else None
    w = torch.randn(Cout, Cin, *k, generator=g) / math.sqrt(Cin * k[0] * k[1] * k[2])
    bias = torch.randn(Cout, generator=g) * 0.1
    res = torch.randn(B, Cout, D, H, W, generator=g) if with_res else None
    h = F.group_norm(x.double(), 8, gamma.double(), beta.double(), eps=1e-5)
    if with_ss:
        h = h * (ss.double()[:, :Cin, None, None, None] + 1) + ss.double()[:, Cin:, None, None, None]
    h = F.mish(h) if act == "mish" else F.silu(h)
    ref = F.conv3d(h, w.double(), bias.double(), padding=pads)
    if with_res:
        ref = ref + res.double()
    xd, gd, bd, wd, cd = cl(x), gamma.to(DEV), beta.to(DEV), w.to(DEV), bias.to(DEV)
    sd = ss.to(DEV) if with_ss else None
    rd = cl(res) if with_res else None
    with torch.no_grad():
        y = ops.gn_conv3d(xd, gd, bd, sd, 8, A, 1e-5, wd, cd, pads, rd, want_stats=True)
        assert y is not None, "shape not taken by the GroupNorm-apply instantiation of conv_fwd9_kernel"
        y2 = ops.conv3d(ops.groupnorm_act(xd, gd, bd, sd, 8, A), wd, cd, pads, rd, want_stats=True)
    close(cf(y), ref, tol=3e-5, what="fused GroupNorm + act + conv vs float64")
    close(cf(y), cf(y2).double(), tol=1e-5, what="fused vs two-kernel path")
    st, st2 = getattr(y, "_diqt_stats", None), getattr(y2, "_diqt_stats", None)
    assert (st is None) == (st2 is None)
    if st is not None:
        close(st.partials, st2.partials.double(), tol=1e-5, what="output column sums")
    with torch.no_grad():
        y3 = ops.gn_conv3d(xd, gd, bd, sd, 8, A, 1e-5, wd, cd, pads, rd)
    assert torch.equal(y3, y), "the result must not depend on whether output statistics are requested"


@pytest.mark.parametrize("B,sp,Cin,Cout,k,pad,epad,res", [
    (2, (6, 20, 20), 65, 1, (1, 3, 3), (0, 1, 1), (0, 0, 0), False),      # the pseudo-3D final conv: 65 channels (rows not 16-byte aligned)
    (1, (5, 9, 33), 64, 1, (1, 1, 1), (0, 0, 0), (0, 0, 0), True),        # GlobalContext.to_k, ragged tiles, residual
    (2, (9, 10, 12), 64, 1, (3, 3, 3), (1, 1, 1), (0, 0, 0), False),      # the 3-D final conv: 27 taps, channel chunks
    (1, (12, 6, 6), 40, 1, (3, 1, 1), (2, 0, 0), (-2, 0, 0), True),       # causal temporal: both pads on the low side
    (1, (4, 18, 17), 128, 2, (1, 3, 3), (0, 1, 1), (0, 0, 0), False),     # two output channels, three channel chunks
    (1, (3, 40, 40), 200, 2, (1, 1, 1), (0, 0, 0), (0, 0, 0), True),
    (1, (7, 7, 7), 16, 1, (3, 3, 3), (0, 0, 0), (0, 0, 0), False),        # un-padded
])
def test_conv3d_with_one_or_two_output_channels_on_the_vector_alu(ops, B, sp, Cin, Cout, k, pad, epad, res):
    """diqt_conv3d_fwd_smallcout (forward only; the backward stays on the generic kernels) against a float64 convolution, and that
    ops.conv3d routes there with gradients intact."""
    from diffusioniqt_amd import _lib
    assert _lib.query("diqt_conv3d_fwd_smallcout_supported", B, *sp, Cin, Cout, *k, *pad, *epad) == 1
    g = torch.Generator().manual_seed(Cin * 7 + Cout)
    x = torch.randn(B, Cin, *sp, generator=g)
    w = torch.randn(Cout, Cin, *k, generator=g) / math.sqrt(Cin * k[0] * k[1] * k[2])
    b = torch.randn(Cout, generator=g) * 0.1
    xr, wr, br = x.double().requires_grad_(), w.double().requires_grad_(), b.double().requires_grad_()
    xp = F.pad(xr, (pad[2], pad[2] + epad[2], pad[1], pad[1] + epad[1], pad[0], pad[0] + epad[0]))
    yr = F.conv3d(xp, wr, br)
    r = torch.randn(yr.shape, generator=g) if res else None
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy.double())
    seen = []
    real = _lib.call
    _lib.call = lambda name, *a: (seen.append(name), real(name, *a))[1]
    try:
        xd, wd, bd = cl(x).requires_grad_(), w.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
        y = ops.conv3d(xd, wd, bd, pad, residual=cl(r) if res else None, extra_pad=epad)
    finally:
        _lib.call = real
    assert "diqt_conv3d_fwd_smallcout" in seen and "diqt_conv3d_fwd" not in seen
    close(cf(y), yr + (r.double() if res else 0), tol=2e-6, what="small-Cout conv fwd")
    y.backward(cl(dy))
    close(cf(xd.grad), xr.grad, what="small-Cout conv dx")
    close(wd.grad, wr.grad, tol=5e-5, what="small-Cout conv dw")
    close(bd.grad, br.grad, tol=5e-5, what="small-Cout conv db")


@pytest.mark.parametrize("shape,Ca,Cb,fa,fb", [((2, 3, 5, 7), 64, 32, 1.0, 2 ** -0.5), ((3, 11), 6, 9, 0.5, 1.5), ((1, 4, 4, 4), 4, 4, 1.0, 1.0),
                                              ((2, 5), 1, 1, 1.0, 1.0)])
def test_scaled_concat_and_its_adjoint_bit_exact(ops, shape, Ca, Cb, fa, fb):
    """cat(fa * a, fb * b) in one pass (the scaled skip connections) and its backward: one fp32 multiply per element, so bit-exact."""
    g = torch.Generator().manual_seed(Ca * 10 + Cb)
    a, b = torch.randn(*shape, Ca, generator=g), torch.randn(*shape, Cb, generator=g)
    dy = torch.randn(*shape, Ca + Cb, generator=g)
    ad, bd = a.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
    y = ops.concat_channels(ad, bd, fa, fb)
    assert torch.equal(y.cpu(), torch.cat((a * fa, b * fb), -1))
    y.backward(dy.to(DEV))
    assert torch.equal(ad.grad.cpu(), dy[..., :Ca] * fa) and torch.equal(bd.grad.cpu(), dy[..., Ca:] * fb)


@pytest.mark.parametrize("B,rows,Ca,Cb", [(2, 1000, 64, 64), (1, 777, 128, 256), (3, 256, 4, 8), (2, 4096, 256, 128)])
def test_concat_with_column_sums_for_the_next_groupnorm(B, rows, Ca, Cb):
    """cat(x, skip * 2^-0.5) of the up path (imagen_video.py:1743, imagen_pytorch3D.py:1631) in one pass that also writes the column sums
    of its output: y bit-identical to the plain concat, the sums equal to a float64 reduction of y, and the GroupNorm that follows takes
    them (``_diqt_stats``) instead of a statistics pass."""
    from diffusioniqt_amd import ops, _lib
    g = torch.Generator().manual_seed(B * 7 + Ca)
    a = torch.randn(B, rows, 1, 1, Ca, generator=g).to(DEV)
    b = torch.randn(B, rows, 1, 1, Cb, generator=g).to(DEV)
    sb = 2 ** -0.5
    with torch.no_grad():
        y0 = ops.concat_channels(a, b, 1.0, sb)
        with _lib.census() as c:
            y = ops.concat_channels(a, b, 1.0, sb, want_stats=True)
            assert c.count("concat_channels_stats") == 1
    assert torch.equal(y, y0)
    st = y._diqt_stats
    assert st.rows == rows and st.partials.shape == (B, st.nblk, 2, Ca + Cb)
    got = st.partials.double().sum(1)
    flat = y.double().reshape(B, rows, Ca + Cb)
    assert torch.allclose(got[:, 0], flat.sum(1), rtol=1e-6, atol=1e-3) and torch.allclose(got[:, 1], (flat * flat).sum(1), rtol=1e-6, atol=1e-3)
    # the consumer: GroupNorm statistics from the partials == from a pass over the tensor
    gamma, beta = torch.ones(Ca + Cb, device=DEV), torch.zeros(Ca + Cb, device=DEV)
    with torch.no_grad(), _lib.census() as c:
        z1 = ops.groupnorm_act(y, gamma, beta, None, 4, ops.ACT_SILU, 1e-5)
        assert c.count("groupnorm_stats_from_partials") == 1 and c.count("groupnorm_stats/reduce") == 0
    with torch.no_grad():
        z0 = ops.groupnorm_act(y0, gamma, beta, None, 4, ops.ACT_SILU, 1e-5)
    assert (z1 - z0).abs().max().item() <= 2e-5 * z0.abs().max().item()


def test_multi_accumulate_with_the_table_in_the_kernel_arguments(ops):
    """diqt_multi_accumulate_host (the gradient hand-over of a captured training micro-step: rows travel in the kernel arguments, 120 per
    launch) against diqt_multi_accumulate with the table in device memory and against plain adds -- bit-exact, also past one launch's rows."""
    from diffusioniqt_amd import _lib
    torch.manual_seed(9)
    sizes = [int(v) for v in torch.randint(1, 700, (300,))] + [70001, 4, 3]
    offs, off = [], 0
    for n in sizes:
        offs.append(off)
        off += (n + 3) // 4 * 4
    d0 = torch.randn(off)
    srcs = [torch.randn(n) for n in sizes]
    want = d0.clone()
    for t, o in zip(srcs, offs):
        want[o:o + t.numel()] += t
    dev = [t.to(DEV) for t in srcs]
    rows = torch.tensor([(t.data_ptr(), o, t.numel()) for t, o in zip(dev, offs)], dtype=torch.int64)
    st = torch.cuda.current_stream().cuda_stream
    a = d0.to(DEV)
    _lib.call("diqt_multi_accumulate_host", a, rows, len(sizes), 16, st)
    b = d0.to(DEV)
    _lib.call("diqt_multi_accumulate", b, rows.to(DEV), len(sizes), 16, st)
    torch.cuda.synchronize()
    assert torch.equal(a.cpu(), want) and torch.equal(b.cpu(), want)


@pytest.mark.parametrize("bf16", [0, 1])
def test_many_weights_packed_by_one_launch_equal_the_single_packs(bf16):
    """diqt_conv_pack_weight_h_multi (a captured training micro-step re-derives every 16-bit packed weight with it) == one
    diqt_conv_pack_weight_h per weight, bit for bit: forward and backward-data layouts, ragged channel counts, more than 64 weights."""
    from diffusioniqt_amd import _lib
    torch.manual_seed(4)
    shapes = [(64, 64, 3, 3, 3), (128, 64, 3, 3, 3), (40, 24, 1, 3, 3), (64, 128, 3, 1, 1), (256, 256, 1, 1, 1), (8, 96, 3, 3, 3)] * 12
    st = torch.cuda.current_stream().cuda_stream
    rows, singles, multis, ws = [], [], [], []
    for k, shp in enumerate(shapes):
        mode = k % 2
        w = torch.randn(*shp, device=DEV)
        Cout, Cin, kd, kh, kw = shp
        eff = (Cout, Cin) if mode == 0 else (Cin, Cout)
        n = _lib.query("diqt_conv_packed_h_elems", eff[0], eff[1], kd, kh, kw)
        s = torch.zeros(n, dtype=torch.int16, device=DEV)
        m = torch.zeros(n, dtype=torch.int16, device=DEV)
        _lib.call("diqt_conv_pack_weight_h", w, s, Cout, Cin, kd, kh, kw, mode, bf16, st)
        rows.append((w.data_ptr(), m.data_ptr(), Cout, Cin, kd, kh, kw, mode))
        singles.append(s); multis.append(m); ws.append(w)
    _lib.call("diqt_conv_pack_weight_h_multi", torch.tensor(rows, dtype=torch.int64), len(rows), bf16, st)
    torch.cuda.synchronize()
    assert len(rows) > 64
    for s, m in zip(singles, multis):
        assert torch.equal(s, m)
