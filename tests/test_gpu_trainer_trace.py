"""T1/T2 on the HIP path: the trace recorded from the REAL reference trainer (tests/golden/trainerA_trace.npz, made by
oracle/make_golden.py::gen_trainer_trace — losses, ``steps``, weight-changed flags and ``final_conv.weight`` after each of
6 micro-steps at gradient_accumulation_steps=4) replayed through the product's ImagenTrainer + HIP SRUnet256 on an MI355X:
chunk_frac scaling, loss / accumulation division, Adam bias correction, EMA and ``steps`` cadence in the composition of the
real kernels (/root/reference/trainer.py:1099-1128, 1038-1081).  tests/test_host_trainer.py replays the same trace with CPU
doubles; this is the one that exercises the fused Adam / multi-accumulate / mse_clamp / q_sample kernels."""
import json

import numpy as np
import pytest
import torch

from oracle import iqt_oracle as O
from tests.conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda"
T = lambda a: torch.from_numpy(np.asarray(a))


def make_gpu_trainer(**kw):
    from diffusioniqt_amd.imagen_pytorch3D import Imagen, NullUnet, SRUnet256
    from diffusioniqt_amd.trainer import ImagenTrainer
    gu = load_golden('unetA_tiny')
    unet = SRUnet256(**json.loads(str(gu['kwargs'])))
    assert list(unet.state_dict().keys()) == [str(k) for k in gu['keys']]
    unet.load_state_dict(O.hash_fill_state_dict(unet.state_dict(), 0))
    configs = {'Data': {'norm': 'z-score'}, 'Train': {'batch_sample': False, 'patch_size_sub': 8, 'pred_obj': 'x_start'},
               'Eval': {'repeat': 1}}
    imagen = Imagen(unets=(NullUnet(), unet), configs=configs, min_bound=float(gu['min_bound']), image_sizes=(8, 8),
                    channels=1, pred_objectives='x_start', timesteps=4, dynamic_thresholding=False,
                    p2_loss_weight_gamma=0.0, cond_drop_prob=0.0).to(DEV)
    ImagenTrainer.locked = False
    trainer = ImagenTrainer(configs=configs, imagen=imagen, gradient_accumulation_steps=4, verbose=False, **kw)
    return trainer, imagen.unets[1]


def test_trainer_trace_losses_cadence_and_weights_match_reference_on_hip():
    g = load_golden('trainerA_trace')
    trainer, unet = make_gpu_trainer()
    trainer.training = True
    unet.train()
    w = unet.final_conv.weight
    w_prev = w.detach().clone()
    for i in range(g['hr'].shape[0]):
        times = T(g['times'][i])
        trainer.imagen.noise_schedulers[1].sample_random_times = lambda b, device, t=times: t.clone().to(device)
        loss, pred, x_noisy, _ = trainer.forward(T(g['hr'][i]), lowres_img=T(g['lowres'][i]), unet_number=2, max_batch_size=2,
                                                 noise=T(g['noise'][i]))
        steps_ref, changed_ref = int(g['trace'][i][0]), bool(g['trace'][i][1])
        assert int(trainer.steps[1].item()) == steps_ref, f'micro-step {i}: steps'
        w_now = w.detach().clone()
        assert (not torch.equal(w_now, w_prev)) == changed_ref, f'micro-step {i}: Adam cadence differs from the reference'
        w_prev = w_now
        ref_loss = float(g['losses'][i])
        assert abs(loss - ref_loss) <= 2e-5 * abs(ref_loss), (i, loss, ref_loss)
        ref_w = T(g['final_conv_w'][i])
        # Adam's first steps move every weight by ~lr whatever the gradient's size: atol = 2 % of one lr-sized step
        assert torch.allclose(w_now.flatten().cpu(), ref_w, atol=2e-6, rtol=1e-4), \
            f'weights after micro-step {i}: max diff {(w_now.flatten().cpu() - ref_w).abs().max().item():.3e}'
    # the EMA copy tracked the online weights (warm-up: plain copies every 10th call, trainer.py:1060-1061)
    ema = trainer.ema_unets[1]
    assert int(ema.step.item()) == g['hr'].shape[0] and not bool(ema.initted.item())


def test_trainer_trace_chunked_batches_scale_losses_by_chunk_fraction_on_hip():
    """max_batch_size=1 splits each batch of 2 into two chunks: each chunk's loss is scaled by 1/2 (trainer.py:1115-1117) and
    every chunk is a micro-step of its own — so after 2 batches (4 chunks) Adam has stepped once, on gradients that equal the
    unchunked run's up to the fp32 sum order."""
    g = load_golden('trainerA_trace')
    trainer, unet = make_gpu_trainer()
    trainer.training = True
    unet.train()
    total = []
    for i in range(2):
        times = T(g['times'][i])
        calls = {'n': 0}

        def srt(b, device, t=times, c=calls):
            out = t[c['n']:c['n'] + b].clone().to(device)
            c['n'] += b
            return out
        trainer.imagen.noise_schedulers[1].sample_random_times = srt
        loss, *_ = trainer.forward(T(g['hr'][i]), lowres_img=T(g['lowres'][i]), unet_number=2, max_batch_size=1, noise=T(g['noise'][i]))
        total.append(loss)
    assert int(trainer.steps[1].item()) == 4 and trainer.optim1.step_count == 1
    for i in range(2):      # sum over chunks of (chunk mean * 1/2) == the batch mean the reference trace recorded
        assert abs(total[i] - float(g['losses'][i])) <= 2e-5 * abs(float(g['losses'][i])), (i, total[i], g['losses'][i])
