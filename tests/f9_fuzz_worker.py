"""Worker of tests/test_gpu_conv_fuzz.py: random small shapes through conv_fwd9_kernel (DIQT_CONV_F9=2 lifts its tile-count rule, so
ragged extents, single tiles and every variant's edge handling run) against float64 convs on the host."""
import math
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from diffusioniqt_amd import ops, _lib

_lib.load()
dev = "cuda"
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 36
bad = 0
for case in range(n_cases):
    k = rnd.choice([(3, 3, 3), (1, 3, 3), (3, 1, 1)])
    B = rnd.randint(1, 3)
    D, H, W = (rnd.randint(1, 19) for _ in range(3))
    Cin = 16 * rnd.randint(1, 5)
    Cout = rnd.choice([8, 16, 40, 64, 72, 130])
    causal = k == (3, 1, 1) and rnd.random() < 0.5
    if causal:
        pads, epad = (2, 0, 0), (-2, 0, 0)
    else:
        p = rnd.choice([0, 1])
        pads, epad = tuple(p * (kk // 2) for kk in k), (0, 0, 0)
    Do, Ho, Wo = (n + 2 * p + e - kk + 1 for n, p, e, kk in zip((D, H, W), pads, epad, k))
    if min(Do, Ho, Wo) < 1:
        continue
    kid = _lib.query("diqt_conv3d_fwd_kernel_id", B, D, H, W, Cin, Cout, *k, *pads, *epad)
    g = torch.Generator().manual_seed(case)
    x = torch.randn(B, Cin, D, H, W, generator=g)
    w = torch.randn(Cout, Cin, *k, generator=g) / math.sqrt(Cin * k[0] * k[1] * k[2])
    b = torch.randn(Cout, generator=g)
    use_res = rnd.random() < 0.5
    r = torch.randn(B, Cout, Do, Ho, Wo, generator=g) if use_res else None
    xp = F.pad(x.double(), (0, 0, 0, 0, 2, 0)) if causal else x.double()
    ref = F.conv3d(xp, w.double(), b.double(), padding=(0, 0, 0) if causal else pads)
    if use_res:
        ref = ref + r.double()
    with torch.no_grad():
        y = ops.conv3d(x.permute(0, 2, 3, 4, 1).contiguous().to(dev), w.to(dev), b.to(dev), pads,
                       residual=r.permute(0, 2, 3, 4, 1).contiguous().to(dev) if use_res else None, extra_pad=epad, want_stats=True)
    got = y.cpu().permute(0, 4, 1, 2, 3).double()
    err = (got - ref).abs().max().item() / max(ref.abs().max().item(), 1e-6)
    st = getattr(y, "_diqt_stats", None)
    serr = 0.0
    if st is not None:
        serr = ((st.partials[:, :, 0, :].double().sum(1).cpu() - ref.sum(dim=(2, 3, 4))).abs().max().item()
                / max(ref.sum(dim=(2, 3, 4)).abs().max().item(), 1e-6))
    ok = err < 3e-5 and serr < 1e-4
    bad += 0 if ok else 1
    print(f"case {case:2d} kid={kid} k={k} B={B} {D}x{H}x{W} {Cin}->{Cout} pad={pads} epad={epad} res={use_res} err={err:.2e} stats={serr:.2e} {'ok' if ok else 'FAIL'}")
print("FUZZ_OK" if bad == 0 else f"FUZZ_FAILED {bad}")
