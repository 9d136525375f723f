"""cProfile of the host side of bf16 training micro-steps (where do the ~21 us per launch go?).   python tools/train_bf16_cprofile.py"""
import cProfile, os, pstats, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0], "4"]
import runpy
import torch
ns = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "train_bf16_only.py"), run_name="prep")
step = ns["step"]
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(8):
    step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
