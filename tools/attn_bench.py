"""Fused multi-query attention forward (diqt_mqa_attention_fwd) on the Unet3D shapes of C5 (64^3, B = 2) and C2-B (32^3, B = 8).
   python tools/attn_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusioniqt_amd import ops, _lib
_lib.load()
dev = "cuda"
torch.manual_seed(0)
SHAPES = [  # name, G, n, h, d, E, rel, causal
    ("mid joint 64*16*16 tokens, B=2", 2, 16384, 8, 64, 1, False, False),
    ("mid joint 32*8*8 tokens, B=8", 8, 2048, 8, 64, 1, False, False),
    ("temporal full-res 64 frames, B=2", 8192, 64, 8, 64, 1, True, True),
    ("temporal full-res 32 frames, B=8", 8192, 32, 8, 64, 1, True, True),
    ("spatial 16x16 per frame, B=2 x 64", 128, 256, 8, 64, 1, False, False),
    ("temporal 32 frames, no bias, causal", 8192, 32, 8, 64, 1, False, True),          # 5, 6: what the bias table / the mask cost
    ("temporal 32 frames, no bias, no mask", 8192, 32, 8, 64, 1, False, False),
]
if os.environ.get('ONLY'):
    SHAPES = [SHAPES[int(os.environ['ONLY'])]]
for name, G, n, h, d, E, use_rel, causal in SHAPES:
    q = torch.randn(G, n, h * d, device=dev)
    kv = torch.randn(G, E + n, 2 * d, device=dev)
    rel = torch.randn(2 * n - 1, h, device=dev) if use_rel else None
    nb = torch.randn(h, device=dev) if use_rel else None
    lpm = os.environ.get("LP", "off")
    def fn():
        with ops.low_precision(lpm):
            return ops.mqa_attention_nograd(q, kv, rel, nb, n, h, d, E, n, causal, d ** -0.5)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 5
    s.record()
    for _ in range(it):
        fn()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / it
    fl = 4.0 * G * n * h * (E + n) * d * (0.5 if causal else 1.0)
    print(f"{name:38s} G={G:5d} n={n:6d}: {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s (algorithmic{', causal half' if causal else ''})")
    if os.environ.get("BWD") == "1" and ops.mqa_attention_fused_ok(G, n, h, d, n, use_rel):
        # training path: forward with the row log-sum-exp + flash-style backward (dQ kernel, dK/dV kernel, bias-gradient reduce)
        qg, kvg = q.clone().requires_grad_(), kv.clone().requires_grad_()
        relg = rel.clone().requires_grad_() if use_rel else None
        nbg = nb.clone().requires_grad_() if use_rel else None
        out = ops.mqa_attention(qg, kvg, relg, nbg, n, h, d, E, n, causal, d ** -0.5)
        dout = torch.randn_like(out)
        def bw():
            qg.grad = kvg.grad = None
            out.backward(dout, retain_graph=True)
        for _ in range(3):
            bw()
        torch.cuda.synchronize()
        s.record()
        for _ in range(it):
            bw()
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / it
        print(f"{'':38s} backward: {ms:8.3f} ms  {2.5 * fl / ms / 1e9:7.1f} TFLOP/s (5 products)")
