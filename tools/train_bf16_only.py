"""ImagenTrainer(precision='bf16') micro-steps of the C2 U-Net only (kernel-time accounting under rocprofv3).   python tools/train_bf16_only.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import unet_kwargs
from diffusioniqt_amd.imagen_pytorch3D import SRUnet256, Imagen, NullUnet
from diffusioniqt_amd.trainer import ImagenTrainer
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
B, S = 8, 32
torch.manual_seed(42)
unet = SRUnet256(**unet_kwargs(S))
configs = {'Data': {'norm': 'z-score', 'mean': 271.64814106698583, 'std': 377.117173547721},
           'Train': {'batch_sample': False, 'patch_size_sub': S, 'batch_sample_factor': 3, 'pred_obj': 'x_start'},
           'Eval': {'repeat': 1, 'overlap': S, 'batch_size': 4 * B}}
imagen = Imagen(unets=(NullUnet(), unet), configs=configs, min_bound=-0.7203, image_sizes=(S, S), channels=1, pred_objectives='x_start',
                timesteps=32, dynamic_thresholding=False, p2_loss_weight_gamma=0.0, cond_drop_prob=0.0).to(dev)
trainer = ImagenTrainer(configs=configs, imagen=imagen, gradient_accumulation_steps=4, verbose=False)
trainer.prepare_for = 2
trainer.validate_and_set_unet_being_trained(2)
unet = imagen.unets[1]
trainer.mixed_precision = 'bf16'
g = torch.Generator().manual_seed(42)
hr = torch.randn(B, 1, S, S, S, generator=g).to(dev); lr = torch.randn(B, 1, S, S, S, generator=g).to(dev)
unet.train()
step = lambda: trainer.forward(hr, lowres_img=lr, unet_number=2, max_batch_size=B)
for _ in range(8):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
print(f"bf16 micro-step: {(time.perf_counter() - t0) / steps * 1e3:.2f} ms wall over {steps} steps (+8 warm-up steps in the trace)")
