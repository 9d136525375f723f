"""Whole-volume super-resolution end to end (the reference's test_all.py loop: sliding 32^3 windows over a 256^3 low-resolution volume,
5 % non-zero rejection, 32-step ancestral sampling with the C2 U-Net, stitching, background reset) on one MI355X, fp32 and under
torch.autocast(float16).  Synthetic "head": an ellipsoid of smooth texture in a zero background, random-init weights.
    python tools/volume_bench.py [N=256] [timesteps=32] [batch=32]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import unet_kwargs
from diffusioniqt_amd import _lib
from diffusioniqt_amd.imagen_pytorch3D import SRUnet256, Imagen, NullUnet
from diffusioniqt_amd.trainer import ImagenTrainer
from diffusioniqt_amd.inference import VolumeInference
_lib.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
Tn = int(sys.argv[2]) if len(sys.argv) > 2 else 32
BS = int(sys.argv[3]) if len(sys.argv) > 3 else 32
dev = torch.device("cuda:0")
torch.manual_seed(0)
S = 32
configs = {'Data': {'norm': 'z-score', 'mean': 271.64814106698583, 'std': 377.117173547721},
           'Train': {'batch_sample': False, 'patch_size_sub': S, 'batch_sample_factor': 3, 'pred_obj': 'x_start'},
           'Eval': {'repeat': 1, 'overlap': 32, 'batch_size': BS}}
mb = (0. - configs['Data']['mean']) / configs['Data']['std']
imagen = Imagen(unets=(NullUnet(), SRUnet256(**unet_kwargs(S))), configs=configs, min_bound=mb, image_sizes=(S, S), channels=1,
                pred_objectives='x_start', timesteps=Tn, dynamic_thresholding=False, p2_loss_weight_gamma=0.0, cond_drop_prob=0.0).to(dev)
ImagenTrainer.locked = False
trainer = ImagenTrainer(configs=configs, imagen=imagen, use_ema=False, verbose=False)
ax = torch.linspace(-1, 1, N, device=dev)
zz, yy, xx = torch.meshgrid(ax, ax, ax, indexing="ij")
head = ((zz / 0.8) ** 2 + (yy / 0.7) ** 2 + (xx / 0.6) ** 2) < 1
vol = torch.where(head, 600 + 300 * torch.sin(9 * xx) * torch.cos(7 * yy) + 200 * zz, torch.zeros_like(xx)).float()
nkept = [0]


def sample_fn(x):
    nkept[0] += x.shape[0]
    return trainer.sample(batch_size=x.shape[0], start_image_or_video=x, start_at_unet_number=2, use_tqdm=False)[0]


infer = VolumeInference(configs, sample_fn)
for name, ctx in (("fp32", None), ("autocast fp16", torch.autocast('cuda', dtype=torch.float16))):
    nkept[0] = 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if ctx is None:
        pred = infer(vol)
    else:
        with ctx:
            pred = infer(vol)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    assert pred.shape == vol.shape and torch.isfinite(pred).all()
    print(f"{name}: {N}^3 volume, {nkept[0]} of {(N // S) ** 3} patches kept, {Tn}-step sampling, batch {BS}: {dt:.2f} s "
          f"({nkept[0] * Tn / dt:.0f} patch-steps/s)")
