"""The captured training micro-step (graphs.TrainStepGraphs; the loop of /root/reference/trainer.py:1099-1128): forward + loss + backward +
gradient hand-over replayed as one hipGraph must be the eager micro-step bit for bit -- same losses, same weights after every Adam step --
including across optimiser steps (the packed 16-bit weight copies are re-derived inside the graph) and with an eager evaluation in between."""
import json

import numpy as np
import pytest
import torch

from oracle import iqt_oracle as O
from tests.conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda"
T = lambda a: torch.from_numpy(np.asarray(a))


def run_trainer(graph_mode, precision, n_steps, max_grad_norm=None, sample_between=False, batch=2, fixture='unetA_tiny', imagen_kw=None,
                trainer_kw=None):
    from diffusioniqt_amd import graphs, ops
    from diffusioniqt_amd.imagen_pytorch3D import Imagen, NullUnet, SRUnet256
    from diffusioniqt_amd.trainer import ImagenTrainer
    gu = load_golden(fixture)
    unet = SRUnet256(**json.loads(str(gu['kwargs'])))
    unet.load_state_dict(O.hash_fill_state_dict(unet.state_dict(), 0))
    configs = {'Data': {'norm': 'z-score'}, 'Train': {'batch_sample': False, 'patch_size_sub': 8, 'pred_obj': 'x_start'},
               'Eval': {'repeat': 1}}
    xs = gu['x'].shape if 'x' in gu else gu['hr'].shape           # the fixture's input: batch and patch size its U-Net was built for
    S = int(xs[-1])
    if batch == 2 and int(xs[0]) != 2:
        batch = int(xs[0])
    configs = {'Data': {'norm': 'z-score'}, 'Train': {'batch_sample': False, 'patch_size_sub': S, 'pred_obj': 'x_start'}, 'Eval': {'repeat': 1}}
    imagen = Imagen(unets=(NullUnet(), unet), configs=configs, min_bound=float(gu['min_bound']) if 'min_bound' in gu else -1.0, image_sizes=(S, S), channels=1,
                    **{**dict(pred_objectives='x_start', timesteps=4, dynamic_thresholding=False, p2_loss_weight_gamma=0.0, cond_drop_prob=0.0),
                       **(imagen_kw or {})}).to(DEV)
    ImagenTrainer.locked = False
    kw = {'fp16': True} if precision == 'fp16' else {'precision': precision} if precision else {}
    trainer = ImagenTrainer(configs=configs, imagen=imagen, verbose=False, gradient_accumulation_steps=2, max_grad_norm=max_grad_norm, **kw,
                            **(trainer_kw or {}))
    old = graphs.TRAIN_ENABLED, graphs.TRAIN_FORCE
    graphs.TRAIN_ENABLED, graphs.TRAIN_FORCE = graph_mode != 0, graph_mode == 2
    try:
        torch.manual_seed(11)
        g = torch.Generator().manual_seed(3)
        losses, preds, samples = [], [], []
        for i in range(n_steps):
            hr, lr = torch.randn(batch, 1, S, S, S, generator=g), torch.randn(batch, 1, S, S, S, generator=g)
            loss, pred, x_noisy, _ = trainer.forward(hr, lowres_img=lr, unet_number=2, max_batch_size=max(2, batch if batch > 4 else 2))     # batch 4: two chunks per call
            losses.append(loss)
            preds.append(pred.detach().clone())
            if sample_between and i == n_steps // 2:
                # sampling between two replays: the EMA swap, eager (or separately captured) U-Net evaluations that re-pack weights on
                # their own, and the trainable U-Net put back -- the next replay must find everything it addresses untouched
                gs = torch.Generator().manual_seed(21)
                noise = [torch.randn(2, 1, S, S, S, generator=gs) for _ in range(5)]
                with ops.low_precision(precision if precision in ('fp16', 'bf16') else 'off'):
                    img = trainer.sample(batch_size=2, start_image_or_video=lr[:2], start_at_unet_number=2, noise=noise)[0]
                samples.append(img.detach().clone())
        replays = trainer._train_graphs.replays
        errors = [e.get("error") for e in trainer._train_graphs.entries.values() if e.get("error")]
        trainer._train_graphs.clear()
    finally:
        graphs.TRAIN_ENABLED, graphs.TRAIN_FORCE = old
    return losses, preds + samples, [p.detach().clone() for p in imagen.unets[1].parameters()], replays, errors


@pytest.mark.parametrize("precision,mgn,between", [('bf16', None, False), (None, None, True), ('bf16', 0.5, True), ('fp16', None, False)])
def test_captured_micro_step_is_the_eager_micro_step_bit_for_bit(precision, mgn, between):
    n = 10
    la, pa, wa, ra, ea = run_trainer(2, precision, n, mgn, between)
    lb, pb, wb, rb, eb = run_trainer(0, precision, n, mgn, between)
    assert not ea, ea
    assert ra == n - 3 and rb == 0                       # three eager warm-up micro-steps, then the capture and replays
    assert la == lb, (la, lb)
    for a, b in zip(pa, pb):
        assert torch.equal(a, b)
    moved = 0
    for a, b in zip(wa, wb):
        assert torch.equal(a, b), (a - b).abs().max()
        moved += 1
    assert moved > 10 and all(np.isfinite(la))


def test_gpu_bound_steps_stay_eager_and_the_switch_turns_capture_off():
    from diffusioniqt_amd import graphs
    assert graphs.TRAIN_ENABLED                          # the default
    _, _, _, r0, _ = run_trainer(0, 'bf16', 6)
    assert r0 == 0


def test_two_chunks_per_forward_call_replay_the_same_graph():
    """``max_batch_size`` splits a batch of 4 into two micro-steps per ``forward`` call (trainer.py:1104-1123: chunk fraction 0.5 in the
    loss): both chunks go through the one captured graph, losses and weights as in the eager run."""
    n = 6
    la, pa, wa, ra, ea = run_trainer(2, 'bf16', n, batch=4)
    lb, pb, wb, rb, eb = run_trainer(0, 'bf16', n, batch=4)
    assert not ea and ra == 2 * n - 3 and rb == 0
    assert la == lb
    assert all(torch.equal(a, b) for a, b in zip(pa, pb)) and all(torch.equal(a, b) for a, b in zip(wa, wb))


def test_ragged_last_chunk_gets_its_own_graph():
    """A batch of 3 with ``max_batch_size = 2`` splits into chunks of 2 and 1 (chunk fractions 2/3 and 1/3, trainer.py:1104-1123): two
    keys, two captured graphs, results as in the eager run."""
    n = 6
    la, pa, wa, ra, ea = run_trainer(2, 'bf16', n, batch=3)
    lb, pb, wb, rb, eb = run_trainer(0, 'bf16', n, batch=3)
    assert not ea and ra == 2 * (n - 3) and rb == 0
    assert la == lb
    assert all(torch.equal(a, b) for a, b in zip(pa, pb)) and all(torch.equal(a, b) for a, b in zip(wa, wb))


@pytest.mark.parametrize("fixture", ['unetA_attn_linear', 'unetA_attn_softmax', 'unetA_attn_vit_local', 'unetA_attn_vit_mlp', 'unetA_memeff',
                                     'unetA_boundary_attn_linear', 'unetA_boundary'])
@pytest.mark.parametrize("precision", [None, 'bf16'])
def test_unets_with_attention_layers_capture_or_fall_back_cleanly(fixture, precision):
    """The other U-Net variants of the reference (linear / soft-max / ViT3D attention, memory-efficient layout, boundary padding;
    imagen_pytorch3D.py:1090-1640) through the captured micro-step: whatever the capture makes of them -- a graph, or a refusal that
    leaves the step eager -- the training run is the eager one.  The depthwise-conv weight gradient of the attention layers and the
    boundary-pad adjoint accumulate with atomics (two EAGER runs of these nets already differ in the last bits after the first Adam
    step), so the comparison is bit-exact only for the variant without them (memory-efficient layout)."""
    n = 7
    la, pa, wa, ra, ea = run_trainer(1, precision, n, fixture=fixture)
    lb, pb, wb, rb, eb = run_trainer(0, precision, n, fixture=fixture)
    assert rb == 0
    assert ra == n - 3 or ea, "neither captured nor a recorded refusal"
    if fixture == 'unetA_memeff':
        assert la == lb and all(torch.equal(a, b) for a, b in zip(wa, wb)), ea
    else:
        assert np.allclose(la, lb, rtol=2e-3, atol=0), (la, lb, ea)
        assert all(torch.isfinite(a).all() for a in wa)
        assert max(float((a - b).abs().max()) for a, b in zip(wa, wb)) <= 8e-4      # a few Adam steps of lr = 1e-4 apart at most


@pytest.mark.parametrize("imagen_kw,trainer_kw", [
    (dict(pred_objectives='noise'), None),
    (dict(pred_objectives='v', loss_type='l1'), None),
    (dict(pred_objectives='x_start', loss_type='huber', p2_loss_weight_gamma=0.5), None),
    (None, dict(cosine_decay_max_steps=6, warmup_steps=3)),
])
def test_objectives_loss_types_and_lr_schedules_through_the_captured_step(imagen_kw, trainer_kw):
    """The noise / v objectives, L1 / Huber losses, the p2 loss weight (its per-sample weights are an input of the captured step) and the
    learning-rate schedules (host-side: the rate is an argument of the fused Adam launch outside the graph) -- imagen_pytorch3D.py:2277-2387,
    trainer.py:350-382: graph == eager, bit for bit."""
    n = 8
    la, pa, wa, ra, ea = run_trainer(2, 'bf16', n, imagen_kw=imagen_kw, trainer_kw=trainer_kw)
    lb, pb, wb, rb, eb = run_trainer(0, 'bf16', n, imagen_kw=imagen_kw, trainer_kw=trainer_kw)
    assert not ea and ra == n - 3 and rb == 0
    assert la == lb, (la, lb)
    assert all(torch.equal(a, b) for a, b in zip(pa, pb)) and all(torch.equal(a, b) for a, b in zip(wa, wb))
