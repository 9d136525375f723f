"""Host logic of ImagenTrainer / Imagen (no GPU): chunking, accumulation cadence, loss scaling, EMA schedule and
checkpoint layout, pinned against the trace recorded from the real reference trainer
(tests/golden/trainerA_trace.npz, made by oracle/make_golden.py with accelerate 1.14: parity of the cadence with the
reference's pinned accelerate 0.16 is unpinned — SURVEY.md Appendix A).  Device ops are replaced by the plain-torch
doubles of tests/cpu_doubles.py and the U-Net by the CPU oracle."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import iqt_oracle as O
from tests.conftest import load_golden
from tests.cpu_doubles import cpu_op_doubles, OracleUnet

T = lambda a: torch.from_numpy(np.asarray(a))


def make_trainer(tmp_path=None, **kw):
    from diffusioniqt_amd.imagen_pytorch3D import Imagen, NullUnet
    from diffusioniqt_amd.trainer import ImagenTrainer
    gu = load_golden('unetA_tiny')
    keys = [str(k) for k in gu['keys']]
    shapes = [tuple(json.loads(str(s))) for s in gu['shapes']]
    sd = O.hash_fill_state_dict({k: torch.zeros(s) for k, s in zip(keys, shapes)}, 0)
    cfg = O.unet_config(**json.loads(str(gu['kwargs'])))
    unet = OracleUnet(sd, cfg)
    configs = {'Data': {'norm': 'z-score'}, 'Train': {'batch_sample': False, 'patch_size_sub': 8, 'pred_obj': 'x_start'},
               'Eval': {'repeat': 1}}
    imagen = Imagen(unets=(NullUnet(), unet), configs=configs, min_bound=float(gu['min_bound']), image_sizes=(8, 8),
                    channels=1, pred_objectives='x_start', timesteps=4, dynamic_thresholding=False,
                    p2_loss_weight_gamma=0.0, cond_drop_prob=0.0)
    ImagenTrainer.locked = False
    kw.setdefault('gradient_accumulation_steps', 4)
    trainer = ImagenTrainer(configs=configs, imagen=imagen, verbose=False, **kw)
    return trainer, unet


def test_trainer_cadence_losses_and_weights_match_reference_trace():
    g = load_golden('trainerA_trace')
    with cpu_op_doubles():
        trainer, unet = make_trainer()
        trainer.training = True
        idx = unet.names.index('final_conv.weight')
        w_prev = unet.plist[idx].detach().clone()
        for i in range(g['hr'].shape[0]):
            times = T(g['times'][i])
            trainer.imagen.noise_schedulers[1].sample_random_times = lambda b, device, t=times: t.clone()
            loss, pred, x_noisy, _ = trainer.forward(T(g['hr'][i]), lowres_img=T(g['lowres'][i]), unet_number=2,
                                                     max_batch_size=2, noise=T(g['noise'][i]))
            w = unet.plist[idx].detach().clone()
            steps_ref, changed_ref = int(g['trace'][i][0]), bool(g['trace'][i][1])
            assert int(trainer.steps[1].item()) == steps_ref
            assert (not torch.equal(w, w_prev)) == changed_ref, f'micro-step {i}: Adam cadence differs from the reference'
            w_prev = w
            assert abs(loss - float(g['losses'][i])) <= 2e-5 * abs(float(g['losses'][i])), (i, loss, g['losses'][i])
            ref_w = T(g['final_conv_w'][i])
            assert torch.allclose(w.flatten(), ref_w, atol=2e-6, rtol=1e-4), f'weights after micro-step {i}'


def test_checkpoint_layout_and_roundtrip(tmp_path):
    with cpu_op_doubles():
        trainer, unet = make_trainer()
        trainer.training = True
        g = load_golden('trainerA_trace')
        for i in range(4):
            trainer.forward(T(g['hr'][i]), lowres_img=T(g['lowres'][i]), unet_number=2, max_batch_size=2, noise=T(g['noise'][i]))
        path = os.path.join(tmp_path, 'ck', '3dimagen.pt')
        trainer.save(path)
        obj = torch.load(path, map_location='cpu', weights_only=False)
        assert {'model', 'version', 'steps', 'optim0', 'optim1', 'scaler0', 'scaler1', 'ema'} <= set(obj)
        assert all(k.startswith('unets.') for k in obj['model'])
        assert any(k.startswith('1.ema_model.') for k in obj['ema'])
        assert set(obj['optim1']) == {'state', 'param_groups'} and obj['optim1']['param_groups'][0]['betas'] == (0.9, 0.99)
        st = obj['optim1']['state'][0]
        assert set(st) == {'step', 'exp_avg', 'exp_avg_sq'} and float(st['step']) == 1.0
        w_saved = obj['model']['unets.1.plist.0'].clone()
        # perturb, reload, compare
        with torch.no_grad():
            unet.plist[0].add_(1.0)
        trainer.load(path)
        assert torch.equal(unet.plist[0].detach(), w_saved)
        assert int(trainer.steps[1]) == 4


def test_train_step_is_one_pass_over_the_loader_and_sample_returns_three_tuple():
    from diffusioniqt_amd.data import SyntheticPatchDataset
    with cpu_op_doubles():
        trainer, unet = make_trainer()
        ds = SyntheticPatchDataset(n=6, size=8, seed=1)
        trainer.add_train_dataset(ds, batch_size=2)
        trainer.add_valid_dataset(SyntheticPatchDataset(n=2, size=8, seed=2), batch_size=2)
        loss = trainer.train_step(unet_number=2, max_batch_size=2)
        assert isinstance(loss, float) and int(trainer.steps[1]) == 3          # 3 batches -> 3 micro-steps
        trainer.update(unet_number=2)                                           # train.py:162's extra update
        assert int(trainer.steps[1]) == 4
        out = trainer.valid_step(unet_number=2, max_batch_size=2)
        assert len(out) == 6 and out[1].shape == (2, 1, 8, 8, 8) and isinstance(out[3], list)
        lr = torch.randn(2, 1, 8, 8, 8)
        res = trainer.sample(batch_size=2, skip_steps=None, return_all_outputs=False, return_pil_images=False,
                             start_image_or_video=lr, start_at_unet_number=2)
        assert len(res) == 3 and tuple(res[0].shape) == (2, 1, 8, 8, 8) and len(res[1]) == 5 and isinstance(res[1][0], np.ndarray)
        res2 = trainer.sample(batch_size=2, return_all_unet_outputs=True, start_image_or_video=lr, start_at_unet_number=2,
                              use_non_ema=True)
        assert isinstance(res2[0], list)


def test_ema_schedule_matches_published_defaults():
    from diffusioniqt_amd.trainer import EMA
    m = torch.nn.Linear(2, 2)
    e = EMA(m)
    decays = []
    for s in (0, 100, 101, 102, 111, 1000, 10 ** 7):
        e.step.fill_(s)
        decays.append(e.get_current_decay())
    assert decays[:3] == [0.0, 0.0, 0.0]
    assert abs(decays[3] - (1 - 2 ** (-2 / 3))) < 1e-9 and abs(decays[4] - (1 - 11 ** (-2 / 3))) < 1e-9
    assert decays[-1] == 0.9999


# ------------------------------------------------------------------------------------------------------------------------------------
# checkpoint interop against what the REAL reference's trainer.save wrote (tests/golden/ckpt_manifest.npz, oracle/make_golden_ckpt.py;
# /root/reference/trainer.py:813-945)
# ------------------------------------------------------------------------------------------------------------------------------------
def load_manifest():
    return json.loads(str(load_golden('ckpt_manifest')['manifest']))


def entries(sd):
    return [[k, list(v.shape), str(v.dtype).replace('torch.', '')] for k, v in sd.items()]


def checkpoint_from_manifest(man, seed=3):
    """A checkpoint dictionary with the reference file's exact structure and seeded values (the fixture holds no payloads)."""
    g = torch.Generator().manual_seed(seed)

    def tensor(shape, dtype):
        if dtype == 'bool':
            return torch.zeros(shape, dtype=torch.bool)
        if dtype.startswith('int'):
            return torch.full(shape, 5, dtype=getattr(torch, dtype))
        return torch.randn(shape, generator=g, dtype=getattr(torch, dtype)) * 0.1
    obj = {}
    for k in man['top_keys']:
        if k == 'model':
            obj[k] = {name: tensor(shape, dt) for name, shape, dt in man[k]}
        elif k == 'ema':
            # ``{i}.online_model.X`` IS ``unets.{i}.X`` (the EMA wrapper holds the trained module): one tensor in a real file
            obj[k] = {}
            for name, shape, dt in man[k]:
                i, kind, rest = (name.split('.', 2) + [''])[:3]
                obj[k][name] = obj['model'][f'unets.{i}.{rest}'].clone() if kind == 'online_model' else tensor(shape, dt)
        elif k == 'version':
            obj[k] = man['version']
        elif k == 'steps':
            obj[k] = torch.tensor(man['steps'])
        elif k.startswith('scaler'):
            obj[k] = dict(man['scaler'][k])
        elif k.startswith('optim'):
            o = man['optim'][k]
            groups = [{**gr, 'betas': tuple(gr['betas'])} for gr in o['param_groups']]
            state = {}
            for i, st in o['state'].items():
                d = {name: (tensor(shape, dt).abs() if name == 'exp_avg_sq' else tensor(shape, dt)) for name, shape, dt in st['entries']}
                d['step'] = torch.tensor(st['step'])
                state[int(i)] = d
            obj[k] = dict(state=state, param_groups=groups)
    return obj


def make_real_module_trainer(**kw):
    """ImagenTrainer over the product's real SRUnet256 (constructing and (de)serialising need no GPU; a forward would)."""
    from diffusioniqt_amd.imagen_pytorch3D import Imagen, NullUnet, SRUnet256
    from diffusioniqt_amd.trainer import ImagenTrainer
    gu = load_golden('unetA_tiny')
    unet = SRUnet256(**json.loads(str(gu['kwargs'])))
    configs = {'Data': {'norm': 'z-score'}, 'Train': {'batch_sample': False, 'patch_size_sub': 8, 'pred_obj': 'x_start'}, 'Eval': {'repeat': 1}}
    imagen = Imagen(unets=(NullUnet(), unet), configs=configs, min_bound=float(gu['min_bound']), image_sizes=(8, 8), channels=1,
                    pred_objectives='x_start', timesteps=4, dynamic_thresholding=False, p2_loss_weight_gamma=0.0, cond_drop_prob=0.0)
    ImagenTrainer.locked = False
    return ImagenTrainer(configs=configs, imagen=imagen, gradient_accumulation_steps=4, verbose=False, **kw)


def test_saved_checkpoint_has_the_reference_manifest_before_any_step(tmp_path):
    """Key order of the file, every ``model`` / ``ema`` entry (name, shape, dtype, order), scaler dicts, Adam hyper-parameters and
    parameter index lists equal what the reference's own ``trainer.save`` produced for the same network."""
    man = load_manifest()
    trainer = make_real_module_trainer()
    path = os.path.join(tmp_path, '3dimagen.pt')
    trainer.save(path)
    obj = torch.load(path, map_location='cpu', weights_only=False)
    assert list(obj.keys()) == man['top_keys']
    assert entries(obj['model']) == man['model']
    assert entries(obj['ema']) == man['ema']
    assert str(obj['version']) == man['version']
    assert obj['steps'].dtype == torch.int64 and obj['steps'].tolist() == [0, 0]
    for k, ref in man['scaler'].items():
        assert obj[k] == ref
    for k, ref in man['optim'].items():
        assert list(obj[k].keys()) == ref['keys']
        got = [{**g, 'betas': list(g['betas'])} for g in obj[k]['param_groups']]
        assert got == ref['param_groups'], (k, got, ref['param_groups'])
        assert obj[k]['state'] == {}


def test_load_accepts_a_checkpoint_with_the_reference_layout(tmp_path):
    """A file with the reference's structure (rebuilt from the manifest with seeded values) loads with ``strict=True``: weights, EMA
    weights and buffers, ``steps``; the Adam state waits for the arena (its landing is checked on the GPU, tests/test_gpu_flow.py)."""
    man = load_manifest()
    obj = checkpoint_from_manifest(man)
    path = os.path.join(tmp_path, 'ref_layout.pt')
    torch.save(obj, path)
    trainer = make_real_module_trainer()
    trainer.load(path)
    sd = trainer.imagen.state_dict()
    for name, _, _ in man['model']:
        assert torch.equal(sd[name].cpu(), obj['model'][name]), name
    esd = trainer.ema_unets.state_dict()
    for name, _, _ in man['ema']:
        assert torch.equal(esd[name].cpu(), obj['ema'][name]), name
    assert trainer.steps.tolist() == man['steps']
    pend = trainer.optim1._pending_state
    assert pend is not None and sorted(pend['state']) == sorted(int(i) for i in man['optim']['optim1']['state'])
    # parameters without Adam state in the reference file are exactly the ones that never receive a gradient
    names = man['optim1_param_names']
    assert names == [n for n, _ in trainer.imagen.unets[1].named_parameters()]
    stateless = {names[i] for i in range(len(names)) if str(i) not in man['optim']['optim1']['state']}
    assert stateless and all(n.startswith(('mid_block.', 'norm_cond.')) for n in stateless), stateless


OPT_CASES = {'clip': dict(max_grad_norm=0.02), 'cosine': dict(cosine_decay_max_steps=3),
             'clip_cosine': dict(max_grad_norm=0.02, cosine_decay_max_steps=3, lr=3e-4)}


@pytest.mark.parametrize("tag", list(OPT_CASES))
def test_trainer_grad_clip_and_cosine_schedule_match_reference_trace(tag):
    """ImagenTrainer(max_grad_norm=, cosine_decay_max_steps=) against traces of the REAL reference trainer with those options on
    (oracle/make_golden_trainer_opts.py -> trainerA_trace_opts.npz; /root/reference/trainer.py:350-382, 1054, 1063-1069): the learning
    rate after every micro-step (the scheduler moves with the Adam step, every 2nd micro-step here), losses and final_conv.weight."""
    g = load_golden('trainerA_trace_opts')
    with cpu_op_doubles():
        trainer, unet = make_trainer(gradient_accumulation_steps=2, **OPT_CASES[tag])
        trainer.training = True
        idx = unet.names.index('final_conv.weight')
        for i in range(g['hr'].shape[0]):
            times = T(g['times'][i])
            trainer.imagen.noise_schedulers[1].sample_random_times = lambda b, device, t=times: t.clone()
            loss, *_ = trainer.forward(T(g['hr'][i]), lowres_img=T(g['lowres'][i]), unet_number=2, max_batch_size=2, noise=T(g['noise'][i]))
            assert int(trainer.steps[1].item()) == int(g[f'{tag}:steps'][i])
            assert abs(trainer.get_lr(2) - float(g[f'{tag}:lrs'][i])) <= 1e-12 + 1e-9 * float(g[f'{tag}:lrs'][i]), (i, trainer.get_lr(2))
            ref_l = float(g[f'{tag}:losses'][i])
            assert abs(loss - ref_l) <= 5e-5 * abs(ref_l), (i, loss, ref_l)
            w = unet.plist[idx].detach().flatten()
            assert torch.allclose(w, T(g[f'{tag}:w'][i]), atol=5e-6, rtol=1e-3), (i, float((w - T(g[f'{tag}:w'][i])).abs().max()))


def test_linear_warmup_restatement_and_checkpoint_keys(tmp_path):
    """warmup_steps (pytorch_warmup 0.1.1 LinearWarmup + dampening(), restated: parity unpinned -- the package is absent): the rate is
    damped by min(1, (k + 1) / period) with k advancing on EVERY update call, around a cosine schedule that moves with the Adam step;
    scheduler / warmup states are saved under the reference's keys (trainer.py:851-855) and restored."""
    import math
    with cpu_op_doubles():
        trainer, unet = make_trainer(gradient_accumulation_steps=2, warmup_steps=5, cosine_decay_max_steps=4, lr=2e-4)
        trainer.training = True
        g = load_golden('trainerA_trace_opts')
        assert abs(trainer.get_lr(2) - 2e-4 / 5) < 1e-15                                    # damped at construction (step 0)
        eta_min = 2e-4 * 0.001
        for i in range(6):
            times = T(g['times'][i])
            trainer.imagen.noise_schedulers[1].sample_random_times = lambda b, device, t=times: t.clone()
            trainer.forward(T(g['hr'][i]), lowres_img=T(g['lowres'][i]), unet_number=2, max_batch_size=2, noise=T(g['noise'][i]))
            adam_steps = (i + 1) // 2
            undamped = eta_min + (2e-4 - eta_min) * (1 + math.cos(math.pi * adam_steps / 4)) / 2
            want = undamped * min(1.0, (i + 2) / 5)
            assert abs(trainer.get_lr(2) - want) <= 1e-9 * want, (i, trainer.get_lr(2), want)
        path = str(tmp_path / 'ck.pt')
        trainer.save(path)
        ck = torch.load(path, weights_only=False)
        keys = list(ck.keys())
        assert keys.index('scheduler1') < keys.index('warmup1') < keys.index('scaler1') < keys.index('optim1')
        assert ck['warmup1']['last_step'] == 6 and ck['scheduler1']['last_epoch'] == 3
        lr_now = trainer.get_lr(2)
        trainer.warmup1.last_step, trainer.optim1.param_groups[0]['lr'] = 0, 1.0
        trainer.load(path)
        assert trainer.warmup1.last_step == 6 and trainer.scheduler1.last_epoch == 3 and trainer.get_lr(2) == lr_now


def test_grad_scaler_follows_torch_gradscaler_update_rule_and_state_keys():
    """ImagenTrainer(fp16=True) pairs autocast with a loss scaler (reference trainer.py:311, 364: torch's GradScaler).  The product's scaler
    for the flat-arena optimiser restates its update rule: same scale / growth tracker after the same sequence of finite / non-finite steps,
    same ``state_dict`` keys (checkpoint interop: the reference saves ``scaler{i}.state_dict()``, trainer.py:857)."""
    import torch
    from diffusioniqt_amd.trainer import _GradScaler, _NullScaler
    ref = torch.amp.GradScaler('cpu', init_scale=2.0 ** 16, growth_interval=3)
    mine = _GradScaler(init_scale=2.0 ** 16, growth_interval=3)
    p = torch.nn.Parameter(torch.ones(2))
    opt = torch.optim.SGD([p], lr=0.0)
    for found_inf in (False, False, True, False, False, False, False, True, True, False, False, False):
        p.grad = torch.full((2,), float('inf') if found_inf else 1.0)
        ref.scale(torch.zeros(()))                         # (lazy init of the reference's scale tensor)
        ref.step(opt)
        ref.update()
        mine.update(found_inf)
        want = ref.state_dict()
        got = mine.state_dict()
        assert list(got) == list(want)
        assert got["scale"] == want["scale"] and got["_growth_tracker"] == want["_growth_tracker"], (got, want)
    other = _GradScaler()
    other.load_state_dict(mine.state_dict())
    assert other.state_dict() == mine.state_dict()
    assert _NullScaler().state_dict() == {} and _NullScaler().get_scale() == 1.0
