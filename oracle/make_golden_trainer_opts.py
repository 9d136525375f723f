"""TEST INFRASTRUCTURE (container-only): traces of the REAL reference trainer with gradient clipping and the cosine LR schedule on.

Run:  python oracle/make_golden_trainer_opts.py        (needs /root/reference; CPU only; ~20 s)

``ImagenTrainer(max_grad_norm=..., cosine_decay_max_steps=...)`` (/root/reference/trainer.py:350-382 construction, :1054 the clip,
:1063-1067 the scheduler step) on the tiny Family-A trainer of tests/golden (`unetA_tiny`), gradient_accumulation_steps = 2, eight
micro-steps = four Adam steps.  Recorded per micro-step: loss, ``steps``, the optimiser's learning rate AFTER the call, and
``final_conv.weight``.  Both options are torch / accelerate code (``clip_grad_norm_``, ``CosineAnnealingLR`` with the reference's
``eta_min = lr[1] * 0.001``), present in this image, so these traces PIN them.  ``warmup_steps`` goes through ``pytorch_warmup`` 0.1.1,
which is absent (oracle/ref_shim.py stubs it): no fixture -- the product restates its published LinearWarmup / dampening() and says
"parity unpinned" there.  Writes tests/golden/trainerA_trace_opts.npz (numbers only)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402
from make_golden import base_configs, unet_kwargs_train_py, fill, save, MIN_BOUND  # noqa: E402

CASES = {
    'clip': dict(max_grad_norm=0.02),
    'cosine': dict(cosine_decay_max_steps=3),
    'clip_cosine': dict(max_grad_norm=0.02, cosine_decay_max_steps=3, lr=3e-4),
}


def run(r3, rt, kw, data):
    S, dim = 8, 16
    torch.manual_seed(0)
    np.random.seed(0)
    unet = r3.SRUnet256(**unet_kwargs_train_py(dim, S))
    fill(unet)
    cfgs = base_configs()
    imagen = r3.Imagen(unets=(r3.NullUnet(), unet), configs=cfgs, min_bound=MIN_BOUND, image_sizes=(S, S), channels=1,
                       pred_objectives='x_start', timesteps=4, dynamic_thresholding=False, p2_loss_weight_gamma=0.0,
                       auto_normalize_img=False, cond_drop_prob=0.0)
    rt.ImagenTrainer.locked = False
    trainer = rt.ImagenTrainer(configs=cfgs, imagen=imagen, gradient_accumulation_steps=2, split_valid_from_train=False,
                               verbose=False, **kw)
    hr, lr, times, noise = data
    trainer.training = True
    losses, steps, lrs, ws, norms = [], [], [], [], []
    for i in range(hr.shape[0]):
        trainer.imagen.noise_schedulers[1].sample_random_times = (lambda b, device, i=i: times[i].clone())
        loss, *_ = trainer.forward(hr[i], lowres_img=lr[i], unet_number=2, max_batch_size=2, noise=noise[i])
        u = trainer.imagen.unets[1]
        losses.append(float(loss))
        steps.append(int(trainer.steps[1].item()))
        lrs.append(float(trainer.optim1.param_groups[0]['lr']))
        ws.append(u.final_conv.weight.detach().clone().flatten())
    return dict(losses=np.array(losses), steps=np.array(steps), lrs=np.array(lrs, dtype=np.float64), w=torch.stack(ws))


if __name__ == "__main__":
    r3, rv, re_, rt = ref_shim.import_reference()
    torch.set_num_threads(8)
    S, n_micro = 8, 8
    g = torch.Generator().manual_seed(17)
    data = (torch.randn(n_micro, 2, 1, S, S, S, generator=g), torch.randn(n_micro, 2, 1, S, S, S, generator=g),
            torch.rand(n_micro, 2, generator=g), torch.randn(n_micro, 2, 1, S, S, S, generator=g))
    out = dict(hr=data[0], lowres=data[1], times=data[2], noise=data[3], min_bound=MIN_BOUND)
    for tag, kw in CASES.items():
        res = run(r3, rt, kw, data)
        print(tag, 'lrs', res['lrs'], 'losses', res['losses'])
        for k, v in res.items():
            out[f'{tag}:{k}'] = v
        out[f'{tag}:kw'] = np.array(str(sorted(kw.items())))
    save("trainerA_trace_opts", **out)
