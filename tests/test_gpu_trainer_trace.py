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
    kw.setdefault('gradient_accumulation_steps', 4)
    trainer = ImagenTrainer(configs=configs, imagen=imagen, verbose=False, **kw)
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


# ------------------------------------------------------------------------------------------------------------------------------------
# T4: ImagenTrainer.sample + use_ema_unets (/root/reference/trainer.py:982-1005, 1083-1097)
# ------------------------------------------------------------------------------------------------------------------------------------
def close(got, ref, tol, what=""):
    got, ref = torch.as_tensor(got).detach().double().cpu(), torch.as_tensor(ref).detach().double().cpu()
    assert got.shape == ref.shape, f"{what}: {tuple(got.shape)} vs {tuple(ref.shape)}"
    err, scale = (got - ref).abs().max().item(), ref.abs().max().item()
    assert err <= tol * scale + 1e-6, f"{what}: max err {err:.3e}, scale {scale:.3e}"


def test_trainer_sample_uses_the_ema_unet_and_matches_the_reference_trajectory_on_hip():
    """``trainer.sample`` swaps the EMA U-Nets in (trainer.py:1083-1097 -> 982-1005).  A fresh trainer's EMA model is a copy of the
    online weights, so sampling through the trainer with injected noise must reproduce the trajectory the REAL reference recorded
    (tests/golden/ddpmA_traj.npz: final image, per-step noisy / x0 lists) -- through the EMA module, not the online one."""
    g = load_golden('ddpmA_traj')
    trainer, unet = make_gpu_trainer()
    mb = float(g['min_bound'])
    ema_model = trainer.ema_unets[1].ema_model
    calls = {'ema': 0, 'online': 0}

    def counted(module, tag):                     # the sampler calls forward_with_cond_scale (imagen_pytorch3D.py:1990), not __call__
        inner = module.forward_with_cond_scale

        def fn(*a, **k):
            calls[tag] += 1
            return inner(*a, **k)
        module.forward_with_cond_scale = fn
    counted(ema_model, 'ema')
    counted(unet, 'online')
    noise = [T(g['init_noise'])] + list(T(g['step_noise']))
    img, noisy, x0 = trainer.sample(batch_size=2, start_image_or_video=T(g['lowres']), start_at_unet_number=2, noise=noise)
    assert calls == {'ema': int(g['T']), 'online': 0}, calls
    assert trainer.imagen.unets[1] is unet, "the trainable U-Net was not restored after the EMA swap"
    close(img, T(g['img']), 5e-4, "trainer.sample img")
    ref_noisy = T(g['noisy'])
    close(np.stack(noisy[:-2]), ref_noisy[:-2], 5e-4, "noisy list")
    close(T(np.stack(noisy[-2:])).clamp(min=mb), ref_noisy[-2:], 5e-4, "noisy tail (see test_gpu_unet.py: numpy aliasing of the CPU fixture)")
    close(np.stack(x0), T(g['x0']), 5e-4, "x0 list")


def test_trainer_sample_ema_vs_online_weights_after_training_on_hip():
    """After optimiser steps with the EMA past its warm-up (ema_update_after_step=0, ema_update_every=1) the EMA weights differ
    from the online ones: ``trainer.sample`` (EMA) and ``trainer.sample(use_non_ema=True)`` must equal the oracle's ancestral loop
    run with the EMA and with the online state dict respectively, and the EMA weights themselves must follow the lerp
    ema += (1 - decay) (online - ema) with ema_pytorch's decay schedule (restated, parity unpinned: trainer.py:362, 1060-1061)."""
    g = load_golden('trainerA_trace')
    gt = load_golden('ddpmA_traj')
    gu = load_golden('unetA_tiny')
    trainer, unet = make_gpu_trainer(ema_update_after_step=0, ema_update_every=1)
    cfg = O.unet_config(**json.loads(str(gu['kwargs'])))
    trainer.training = True
    unet.train()
    ema = trainer.ema_unets[1]
    w_on, w_ema = unet.final_conv.weight, ema.ema_model.final_conv.weight
    expect = w_ema.detach().clone().double()
    for i in range(5):                                           # Adam steps on the 4th micro-step; EMA lerps on every one
        times = T(g['times'][i])
        trainer.imagen.noise_schedulers[1].sample_random_times = lambda b, device, t=times: t.clone().to(device)
        step_before = int(ema.step.item())
        trainer.forward(T(g['hr'][i]), lowres_img=T(g['lowres'][i]), unet_number=2, max_batch_size=2, noise=T(g['noise'][i]))
        on = w_on.detach().double()
        if step_before <= 0 or i == 1:                           # warm-up copy (step 0) and the first post-warm-up update (copy, then lerp: no-op)
            expect = on.clone()
        else:
            epoch = max(step_before + 1 - 0 - 1, 0.)
            decay = 0. if epoch <= 0 else min(max(1 - (1 + epoch) ** (-2 / 3), 0.), 0.9999)
            expect = expect + (1. - decay) * (on - expect)
        close(w_ema, expect, 1e-6, f"EMA final_conv.weight after micro-step {i}")
    assert bool(ema.initted.item()) and not torch.equal(w_on, w_ema), "EMA and online weights should differ after the Adam step"
    trainer.training = False
    unet.eval()
    mb, Tn = float(gt['min_bound']), int(gt['T'])
    noise = [T(gt['init_noise'])] + list(T(gt['step_noise']))
    lowres = T(gt['lowres'])
    out = {}
    for tag, kw in (('ema', {}), ('online', {'use_non_ema': True})):
        img, noisy, x0 = trainer.sample(batch_size=2, start_image_or_video=lowres, start_at_unet_number=2, noise=[n.clone() for n in noise], **kw)
        sd = {k: v.detach().cpu() for k, v in (ema.ema_model if tag == 'ema' else unet).state_dict().items()}
        ref_img, ref_noisy, ref_x0 = O.p_sample_loop(sd, cfg, lowres, noise[0], noise[1:], timesteps=Tn, min_bound=mb)
        close(img, ref_img, 5e-4, f"trainer.sample ({tag}) vs oracle with the {tag} weights")
        close(np.stack(x0), torch.stack(ref_x0), 5e-4, f"x0 list ({tag})")
        out[tag] = img.cpu()
    gap = (out['ema'] - out['online']).abs().max().item()
    assert gap > 1e-4, f"EMA and online sampling gave the same image (gap {gap:.2e}): the swap is not observable"


# ------------------------------------------------------------------------------------------------------------------------------------
# T6 / (f)4: checkpoint interop against the manifest of a file the REAL reference wrote (oracle/make_golden_ckpt.py)
# ------------------------------------------------------------------------------------------------------------------------------------
def test_checkpoint_after_training_matches_the_reference_manifest_on_hip(tmp_path):
    """Five micro-steps (one fused-Adam step) through the HIP path, then ``trainer.save``: the file has the reference file's key order,
    ``model`` / ``ema`` entries, ``steps``, scaler dicts and -- from the flat-arena optimiser -- torch.optim.Adam's state layout:
    state only for the 251 parameters that received gradients (none for mid_block / norm_cond), ``step`` = 1 as a 0-dim float32,
    moments in the parameter's shape (/root/reference/trainer.py:813-878)."""
    import os
    from tests.test_host_trainer import load_manifest, entries
    man = load_manifest()
    g = load_golden('trainerA_trace')
    trainer, unet = make_gpu_trainer()
    trainer.training = True
    unet.train()
    for i in range(man['n_micro']):
        trainer.forward(T(g['hr'][i]), lowres_img=T(g['lowres'][i]), unet_number=2, max_batch_size=2, noise=T(g['noise'][i]))
    path = os.path.join(tmp_path, '3dimagen.pt')
    trainer.save(path)
    obj = torch.load(path, map_location='cpu', weights_only=False)
    assert list(obj.keys()) == man['top_keys']
    assert entries(obj['model']) == man['model'] and entries(obj['ema']) == man['ema']
    assert obj['steps'].tolist() == man['steps'] and str(obj['version']) == man['version']
    for k, ref in man['optim'].items():
        got = obj[k]
        assert [{**gr, 'betas': list(gr['betas'])} for gr in got['param_groups']] == ref['param_groups']
        assert sorted(got['state']) == sorted(int(i) for i in ref['state']), k
        for i, st in got['state'].items():
            assert entries(st) == ref['state'][str(i)]['entries'], (k, i)
            assert float(st['step']) == ref['state'][str(i)]['step']
    assert all(obj[k] == ref for k, ref in man['scaler'].items())
    m = obj['optim1']['state'][0]
    assert m['exp_avg'].abs().max() > 0 and (m['exp_avg_sq'] >= 0).all()


def test_reference_layout_checkpoint_resumes_on_hip(tmp_path):
    """A file with the reference's structure (seeded values at the manifest's shapes) is loaded, the Adam moments land at the right
    arena offsets (zeros where the file has no state), ``steps`` / the bias-correction step resume, and training continues."""
    import os
    from tests.test_host_trainer import load_manifest, checkpoint_from_manifest
    man = load_manifest()
    obj = checkpoint_from_manifest(man)
    path = os.path.join(tmp_path, 'ref_layout.pt')
    torch.save(obj, path)
    trainer, unet = make_gpu_trainer()
    trainer.load(path)
    assert trainer.steps.tolist() == man['steps']
    trainer.validate_and_set_unet_being_trained(2)
    opt, arena = trainer.optim1, trainer._arena
    assert opt.step_count == 1
    names = man['optim1_param_names']
    for i in (0, 5, 7, 100, 240, len(names) - 1):
        o, n = arena.offsets[i], arena.params[i].numel()
        st = obj['optim1']['state'].get(i)
        for buf, key in ((opt.exp_avg, 'exp_avg'), (opt.exp_avg_sq, 'exp_avg_sq')):
            want = st[key].reshape(-1) if st is not None else torch.zeros(n)
            assert torch.equal(buf[o:o + n].cpu(), want), (names[i], key)
        assert torch.equal(arena.params[i].detach().cpu(), obj['model'][f'unets.1.{names[i]}']), names[i]
    assert torch.equal(trainer.ema_unets[1].ema_model.final_conv.weight.cpu(), obj['ema']['1.ema_model.final_conv.weight'])
    g = load_golden('trainerA_trace')
    trainer.training = True
    unet.train()
    for i in range(3):                       # the accumulation phase is not part of a checkpoint (nor of the reference's): it restarts
        loss, *_ = trainer.forward(T(g['hr'][i]), lowres_img=T(g['lowres'][i]), unet_number=2, max_batch_size=2, noise=T(g['noise'][i]))
        assert np.isfinite(loss)
    assert trainer.steps.tolist() == [0, man['steps'][1] + 3]
    assert torch.isfinite(unet.final_conv.weight).all()


# ------------------------------------------------------------------------------------------------------------------------------------
# constructor options of T2: max_grad_norm / cosine_decay_max_steps (/root/reference/trainer.py:350-382, 1054, 1063-1069)
# ------------------------------------------------------------------------------------------------------------------------------------
OPT_CASES = {'clip': dict(max_grad_norm=0.02), 'cosine': dict(cosine_decay_max_steps=3),
             'clip_cosine': dict(max_grad_norm=0.02, cosine_decay_max_steps=3, lr=3e-4)}


@pytest.mark.parametrize("tag", list(OPT_CASES))
def test_trainer_grad_clip_and_cosine_schedule_match_reference_trace_on_hip(tag):
    """The traces of the REAL reference trainer with gradient clipping / the cosine schedule on (tests/golden/trainerA_trace_opts.npz),
    replayed through the HIP path: the norm reduction over the flat gradient arena (diqt_grad_norm_clip), the coefficient applied
    inside the fused Adam (diqt_adam_step_scaled), torch's CosineAnnealingLR on the lr carrier."""
    g = load_golden('trainerA_trace_opts')
    trainer, unet = make_gpu_trainer(gradient_accumulation_steps=2, **OPT_CASES[tag])
    trainer.training = True
    unet.train()
    w = unet.final_conv.weight
    for i in range(g['hr'].shape[0]):
        times = T(g['times'][i])
        trainer.imagen.noise_schedulers[1].sample_random_times = lambda b, device, t=times: t.clone().to(device)
        loss, *_ = trainer.forward(T(g['hr'][i]), lowres_img=T(g['lowres'][i]), unet_number=2, max_batch_size=2, noise=T(g['noise'][i]))
        assert int(trainer.steps[1].item()) == int(g[f'{tag}:steps'][i])
        ref_lr = float(g[f'{tag}:lrs'][i])
        assert abs(trainer.get_lr(2) - ref_lr) <= 1e-12 + 1e-9 * ref_lr, (i, trainer.get_lr(2), ref_lr)
        ref_l = float(g[f'{tag}:losses'][i])
        assert abs(loss - ref_l) <= 5e-5 * abs(ref_l), (i, loss, ref_l)
        wv, ref_w = w.detach().flatten().cpu(), T(g[f'{tag}:w'][i])
        assert torch.allclose(wv, ref_w, atol=5e-6, rtol=1e-3), f'{tag}: weights after micro-step {i}: {(wv - ref_w).abs().max().item():.3e}'


def test_grad_norm_clip_kernel_equals_torch_clip_grad_norm():
    from diffusioniqt_amd import ops
    g = torch.Generator().manual_seed(3)
    flat = (torch.randn(1_300_037, generator=g) * 0.01).to(DEV)
    for max_norm in (0.5, 50.0):
        out = ops.grad_norm_clip(flat, max_norm).cpu()
        p = torch.nn.Parameter(torch.zeros_like(flat))
        p.grad = flat.clone()
        total = torch.nn.utils.clip_grad_norm_([p], max_norm).cpu()
        assert abs(out[0] - total) <= 2e-6 * total, (out, total)
        want = min(1.0, max_norm / (float(total) + 1e-6))
        assert abs(float(out[1]) - want) <= 2e-6, (out, want)
