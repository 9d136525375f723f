"""The one-kernel temporal attention block (diqt_temporal_attention_h) alone.   python tools/tattn_bench.py [C] [F] [P] [B] [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusioniqt_amd import ops, _lib
from diffusioniqt_amd.imagen_video import Attention, Residual, TokensOverTime
_lib.load()
C, Fr, P, B, iters = (int(v) for v in sys.argv[1:6]) if len(sys.argv) > 5 else (64, 64, 4096, 8, 20)
dev = torch.device("cuda:0")
torch.manual_seed(0)
attn = Attention(C, heads=8, dim_head=64, causal=False, rel_pos_bias=True, init_zero=False)
blk = TokensOverTime(Residual(attn)).to(dev).eval()
x = torch.randn(B, Fr, 1, P, C, device=dev)
LP = os.environ.get("LP", "fp16")


def run(n):
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.float16 if LP == 'fp16' else torch.bfloat16):
        for _ in range(n):
            blk(x)


run(5)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); run(iters); e.record()
torch.cuda.synchronize()
ms = s.elapsed_time(e) / iters
G = B * P
fl = G * (2 * Fr * C * (8 * 64 + 128) + 8 * 4 * Fr * Fr * 64 + 2 * Fr * 8 * 64 * C)
print(f"temporal attention block C={C} F={Fr} sequences={G}: {ms * 1e3:.1f} us  {fl / ms / 1e9:.1f} TFLOP/s  "
      f"{2 * x.numel() * 4 / ms / 1e6:.0f} GB/s of x + y;  {ms * 1e-3 * 2.4e9 * 256 / G:.0f} cycles per sequence and CU at 2.4 GHz")
