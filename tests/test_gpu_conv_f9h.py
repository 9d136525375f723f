"""conv_f9h_kernel (round 4): the 16-bit forward conv with LDS-DMA halo images, a swizzled conflict-free LDS layout, register-streamed
weight fragments and the transposed product -- behind ``diqt_conv3d_fwd_h_io`` with ``x_half = 1`` for 3x3x3 and (1,3,3) filters.

Bit-exact against a float64 convolution of the operands, rounded once to the operand type, on integer-valued data (every product and
partial sum is exact in fp32, so the kernel's K order does not matter); statistics rows against the stored tensor's column sums.
Reference semantics: ATen's autocast conv3d (inputs cast to the low-precision type, the result rounded to it once) --
/root/reference/imagen_pytorch3D.py:535-566 (Block.project), /root/reference/imagen_video.py:352-381 (per-frame Conv2d)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
LP = {0: torch.float16, 1: torch.bfloat16}


def ref_conv(x, w, bias, pad, epad, dt):
    xr = x.to(dt).double().permute(0, 4, 1, 2, 3)
    xp = F.pad(xr, (pad[2], pad[2] + epad[2], pad[1], pad[1] + epad[1], pad[0], pad[0] + epad[0]))
    y = F.conv3d(xp, w.to(dt).double()) + bias.double().view(1, -1, 1, 1, 1)
    return y.float().to(dt).float().permute(0, 2, 3, 4, 1).contiguous()


SHAPES = [  # B, D, H, W, Cin, Cout, k, pad, epad
    (2, 8, 8, 8, 32, 64, (3, 3, 3), (1, 1, 1), (0, 0, 0)),            # one 512-voxel tile per batch entry, one chunk
    (1, 16, 16, 16, 64, 64, (3, 3, 3), (1, 1, 1), (0, 0, 0)),         # 8 tiles, two chunks (the image ring turns)
    (1, 9, 10, 11, 96, 72, (3, 3, 3), (1, 1, 1), (0, 0, 0)),          # ragged tiles, three chunks, two channel blocks (one ragged)
    (1, 7, 9, 9, 32, 8, (3, 3, 3), (0, 0, 0), (0, 0, 0)),             # unpadded (boundary mode), 8 output channels
    (2, 3, 16, 32, 64, 64, (1, 3, 3), (0, 1, 1), (0, 0, 0)),          # per-frame conv: 1 x 16 x 32 tiles, three images
    (1, 5, 20, 19, 128, 40, (1, 3, 3), (0, 1, 1), (0, 0, 0)),         # ragged frames, four chunks
    (1, 2, 12, 12, 32, 32, (1, 3, 3), (0, 2, 2), (0, -2, -2)),        # both pads on the low side
]


@pytest.mark.parametrize("bf16", [0, 1])
@pytest.mark.parametrize("y_half,res,stats", [(1, False, True), (0, True, True), (0, False, False), (1, False, False)])
@pytest.mark.parametrize("shape", SHAPES)
def test_conv_f9h_is_bit_exact_on_integer_data(shape, y_half, res, stats, bf16):
    from diffusioniqt_amd import _lib
    _lib.load()
    B, D, H, W, Cin, Cout, k, pad, epad = shape
    geo = (B, D, H, W, Cin, Cout, *k, *pad, *epad)
    dt = LP[bf16]
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(B * 1000 + Cin + Cout)
    x = torch.randint(-3, 4, (B, D, H, W, Cin), generator=g).float()
    w = torch.randint(-2, 3, (Cout, Cin, *k), generator=g).float()
    bias = torch.randint(-4, 5, (Cout,), generator=g).float()
    ref = ref_conv(x, w, bias, pad, epad, dt)
    r = torch.randint(-5, 6, tuple(ref.shape), generator=g).float() if res else None
    prev = _lib.query("diqt_set_conv_f9h_mode", 2)
    try:
        assert _lib.query("diqt_conv3d_fwd_h_io16_supported", *geo, 1, y_half) == 1
        n = _lib.query("diqt_conv_packed_h_elems", Cout, Cin, *k)
        packed = torch.empty(n, dtype=torch.int16, device=DEV)
        _lib.call("diqt_conv_pack_weight_h", w.to(DEV), packed, Cout, Cin, *k, 0, bf16, st)
        y = torch.full(ref.shape, 7.0, dtype=dt if y_half else torch.float32, device=DEV)
        nblk = _lib.query("diqt_conv3d_fwd_h_stats_blocks", *geo, 1, y_half)
        assert nblk > 0
        sbuf = torch.full((B, nblk, 2, Cout), -1.0, device=DEV) if stats else None
        with _lib.census() as c:
            _lib.call("diqt_conv3d_fwd_h_io", x.to(DEV).to(dt), packed, bias.to(DEV), r.to(DEV) if res else None, y, *geo, bf16, 1, 1, y_half, sbuf, st)
            torch.cuda.synchronize()
            assert c.count("conv3d_fwd_h(v9h)") == 1
    finally:
        _lib.query("diqt_set_conv_f9h_mode", prev)
    want = ref + r if res else ref
    got = y.float().cpu()
    assert torch.equal(got, want), (got - want).abs().max()
    if stats:
        s = sbuf.double().sum(1).cpu()                     # [B][2][Cout]
        flat = want.double().reshape(B, -1, Cout)
        assert torch.equal(s[:, 0], flat.sum(1)), (s[:, 0] - flat.sum(1)).abs().max()
        assert torch.equal(s[:, 1], (flat * flat).sum(1))


@pytest.mark.parametrize("shape", [(6, 32, 32, 64, 32, 64, (3, 3, 3), (1, 1, 1)),      # 768 tiles of 512 voxels: 3 per workgroup, two images
                                   (5, 64, 32, 32, 64, 64, (1, 3, 3), (0, 1, 1))])      # 640 tiles of 1 x 16 x 32: 2-3 per workgroup, three images
def test_conv_f9h_persistent_walk_over_many_tiles(shape):
    """More tiles than workgroups (the production case): the halo fetch runs one / two units ahead of the MFMAs across tile boundaries."""
    from diffusioniqt_amd import _lib
    _lib.load()
    B, D, H, W, Cin, Cout, k, pad = shape
    geo = (B, D, H, W, Cin, Cout, *k, *pad, 0, 0, 0)
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(77)
    x = torch.randint(-3, 4, (B, D, H, W, Cin), generator=g).float()
    w = torch.randint(-2, 3, (Cout, Cin, *k), generator=g).float()
    bias = torch.randint(-4, 5, (Cout,), generator=g).float()
    assert _lib.query("diqt_conv3d_fwd_h_io16_supported", *geo, 1, 1) == 1
    n = _lib.query("diqt_conv_packed_h_elems", Cout, Cin, *k)
    packed = torch.empty(n, dtype=torch.int16, device=DEV)
    _lib.call("diqt_conv_pack_weight_h", w.to(DEV), packed, Cout, Cin, *k, 0, 0, st)
    y = torch.empty(B, D, H, W, Cout, dtype=torch.float16, device=DEV)
    nblk = _lib.query("diqt_conv3d_fwd_h_stats_blocks", *geo, 1, 1)
    sbuf = torch.empty(B, nblk, 2, Cout, device=DEV)
    with _lib.census() as c:
        _lib.call("diqt_conv3d_fwd_h_io", x.to(DEV).half(), packed, bias.to(DEV), None, y, *geo, 0, 1, 1, 1, sbuf, st)
        torch.cuda.synchronize()
        assert c.count("conv3d_fwd_h(v9h)") == 1
    # reference on the device in fp32 (exact on these integers), rounded once
    xr = x.to(DEV).permute(0, 4, 1, 2, 3)
    ref = F.conv3d(xr, w.to(DEV), bias.to(DEV), padding=pad).half().float().permute(0, 2, 3, 4, 1)
    assert torch.equal(y.float(), ref), (y.float() - ref).abs().max()
    assert torch.equal(sbuf.double().sum(1)[:, 0], ref.double().reshape(B, -1, Cout).sum(1))


@pytest.mark.parametrize("bf16", [0, 1])
def test_conv_f9h_random_data_within_one_ulp_of_the_operand_type(bf16):
    """Production-size launch (the dominant conv of the C2 U-Net under autocast: 64 -> 64, 3x3x3, 8 x 32^3 -- the default mode takes it)
    on random data against the float64 convolution of the rounded operands."""
    from diffusioniqt_amd import _lib
    _lib.load()
    B, S, C = 8, 32, 64
    geo = (B, S, S, S, C, C, 3, 3, 3, 1, 1, 1, 0, 0, 0)
    dt = LP[bf16]
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, S, S, S, C, generator=g)
    w = torch.randn(C, C, 3, 3, 3, generator=g) / (27 * C) ** 0.5
    bias = torch.randn(C, generator=g)
    assert _lib.query("diqt_conv3d_fwd_h_io16_supported", *geo, 1, 0) == 1
    n = _lib.query("diqt_conv_packed_h_elems", C, C, 3, 3, 3)
    packed = torch.empty(n, dtype=torch.int16, device=DEV)
    _lib.call("diqt_conv_pack_weight_h", w.to(DEV), packed, C, C, 3, 3, 3, 0, bf16, st)
    y = torch.empty(B, S, S, S, C, device=DEV)
    with _lib.census() as c:
        _lib.call("diqt_conv3d_fwd_h_io", x.to(DEV).to(dt), packed, bias.to(DEV), None, y, *geo, bf16, 1, 1, 0, None, st)
        torch.cuda.synchronize()
        assert c.count("conv3d_fwd_h(v9h)") == 1
    # float64 reference of two batch entries on the device-independent host path
    xr = x[:2].to(dt).double().permute(0, 4, 1, 2, 3)
    ref = F.conv3d(xr, w.to(dt).double(), bias.double(), padding=1).permute(0, 2, 3, 4, 1)
    got = y[:2].double().cpu()
    ulp = 2.0 ** (-10 if not bf16 else -7)
    err = (got - ref).abs()
    assert (err <= ulp * ref.abs().clamp_min(2.0 ** -8) * 0.51 + 1e-6).all(), err.max()      # one rounding of an fp32-accumulated sum


@pytest.mark.parametrize("bf16", [0, 1])
def test_conv_f9h_equals_the_persistent_16_bit_kernel_bit_for_bit_on_random_data(bf16):
    """Same K order (chunk -> tap -> k-half), fp32 bias behind the sum, one rounding: conv_f9h_kernel and conv_fwd_hp_kernel agree in every
    bit on random data too, so switching kernels by shape (the planner) never changes a result."""
    from diffusioniqt_amd import _lib
    _lib.load()
    B, D, H, W, Cin, Cout = 2, 6, 24, 40, 96, 72
    geo = (B, D, H, W, Cin, Cout, 1, 3, 3, 0, 1, 1, 0, 0, 0)
    dt = LP[bf16]
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, D, H, W, Cin, generator=g).to(DEV).to(dt)
    w = (torch.randn(Cout, Cin, 1, 3, 3, generator=g) / (9 * Cin) ** 0.5).to(DEV)
    bias = torch.randn(Cout, generator=g).to(DEV)
    n = _lib.query("diqt_conv_packed_h_elems", Cout, Cin, 1, 3, 3)
    packed = torch.empty(n, dtype=torch.int16, device=DEV)
    _lib.call("diqt_conv_pack_weight_h", w, packed, Cout, Cin, 1, 3, 3, 0, bf16, st)
    outs = []
    for mode, tag in ((2, "conv3d_fwd_h(v9h)"), (0, "conv3d_fwd_h(persistent)")):
        prev, prevw = _lib.query("diqt_set_conv_f9h_mode", mode), _lib.query("diqt_set_convh_workgroups", 3)
        try:
            y = torch.empty(B, D, H, W, Cout, dtype=dt, device=DEV)
            with _lib.census() as c:
                _lib.call("diqt_conv3d_fwd_h_io", x, packed, bias, None, y, *geo, bf16, 1, 1, 1, None, st)
                torch.cuda.synchronize()
                assert c.count(tag) == 1, tag
            outs.append(y)
        finally:
            _lib.query("diqt_set_conv_f9h_mode", prev); _lib.query("diqt_set_convh_workgroups", prevw)
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
