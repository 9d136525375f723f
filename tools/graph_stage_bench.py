"""C5 stage timing with and without hipGraph replay of the U-Net evals.   python tools/graph_stage_bench.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import unet3d_kwargs
from diffusioniqt_amd import graphs, _lib
from diffusioniqt_amd.imagen_video import Unet3D
from diffusioniqt_amd.elucidated_imagen import ElucidatedImagen
_lib.load()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
torch.manual_seed(5)
kw = unet3d_kwargs(layer_attns=False)
u1, u2 = Unet3D(**{**kw, 'lowres_cond': False}), Unet3D(**kw)
elu = ElucidatedImagen(unets=(u1, u2), image_sizes=(32, 64), channels=1, condition_on_text=False, auto_normalize_img=False,
                       num_sample_steps=steps, temporal_downsample_factor=(2, 1)).to(dev)
lr = torch.randn(8, 1, 32, 32, 32, device=dev).clamp(-1, 1)


def run(stage):
    with torch.autocast('cuda', dtype=torch.float16):
        if stage == 1:
            return elu.sample(batch_size=8, video_frames=64, use_tqdm=False, stop_at_unet_number=1)
        return elu.sample(batch_size=8, video_frames=64, use_tqdm=False, start_at_unet_number=2, start_image_or_video=lr)


for stage in (1, 2):
    for on in (False, True, False, True):
        graphs.ENABLED = on
        run(stage)                                    # warm (and capture)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(stage)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"stage {stage} graphs={'on ' if on else 'off'}: {dt * 1e3:8.1f} ms for {2 * steps - 1} evals = {dt * 1e3 / (2 * steps - 1):6.2f} ms per eval", flush=True)
