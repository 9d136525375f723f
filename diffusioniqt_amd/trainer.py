"""MI355X-native counterpart of the reference's ``trainer.ImagenTrainer`` (trainer.py:236-1128).

Same constructor surface, ``add_*_dataset`` / ``train_step`` / ``valid_step`` / ``update`` / ``sample`` /
``save`` / ``load`` methods, return conventions and checkpoint dictionary keys, without ``accelerate``:
data parallelism is ``distributed.BucketedGradReducer`` (RCCL over xGMI), the optimiser is ONE fused
Adam(+zero_grad) kernel over a flat arena and the EMA is one lerp kernel.

Semantics kept on purpose (SURVEY.md Appendix A "Trainer facts", pinned by tests/golden/trainerA_trace.npz):
``train_step`` is one pass over the dataloader; inside ``forward`` every chunk calls ``update()``; Adam only
steps on every ``gradient_accumulation_steps``-th micro-step of a global counter (gradients keep accumulating
in between), while the EMA update and ``steps`` advance on every micro-step; ``sample`` uses the EMA weights.
"""
import copy
import os
from contextlib import contextmanager, nullcontext
from functools import partial, wraps
from math import ceil
from collections.abc import Iterable

import numpy as np
import torch
from torch import nn
from torch.optim.lr_scheduler import CosineAnnealingLR, LambdaLR
from torch.utils.data import DataLoader, random_split

from . import ops, graphs
from . import distributed as D
from .imagen_pytorch3D import Imagen, NullUnet
from .utils_mine import convertVolume2subVolume, merge_sub_volumes
from .metrics import SSIM, PSNR

device = torch.device('cuda' if torch.cuda.is_available() else 'cpu')
# the checkpoint dictionary follows the layout of the reference at its version.py:1 ('1.20.1'); that string goes into the
# ``version`` entry so that either trainer loads the other's files without the version notice (trainer.py:833, 896-897)
CHECKPOINT_VERSION = '1.20.1'


def exists(val):
    return val is not None


def default(val, d):
    if exists(val):
        return val
    return d() if callable(d) else d


def cast_tuple(val, length=1):
    if isinstance(val, list):
        val = tuple(val)
    return val if isinstance(val, tuple) else ((val,) * length)


def cycle(dl):
    while True:
        for data in dl:
            yield data


def num_to_groups(num, divisor):
    groups, remainder = num // divisor, num % divisor
    arr = [divisor] * groups
    if remainder > 0:
        arr.append(remainder)
    return arr


def groupby_prefix_and_trim(prefix, d):
    with_prefix = {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}
    without = {k: v for k, v in d.items() if not k.startswith(prefix)}
    return with_prefix, without


def eval_decorator(fn):
    def inner(model, *args, **kwargs):
        was_training = model.training
        model.eval()
        out = fn(model, *args, **kwargs)
        model.train(was_training)
        return out
    return inner


def cast_torch_tensor(fn, cast_fp16=False):
    """trainer.py:123-147: numpy -> torch, move tensors to the trainer's device."""
    @wraps(fn)
    def inner(model, *args, **kwargs):
        dev = kwargs.pop('_device', model.device)
        cast_device = kwargs.pop('_cast_device', True)
        keys = list(kwargs.keys())
        all_args = (*args, *kwargs.values())
        split = len(all_args) - len(keys)
        all_args = tuple(torch.from_numpy(t) if isinstance(t, np.ndarray) else t for t in all_args)
        if cast_device:
            all_args = tuple(t.to(dev) if isinstance(t, torch.Tensor) else t for t in all_args)
        if cast_fp16 and getattr(model, 'cast_half_at_training', False):
            # the reference hands .half() tensors to the model (:138-139); activations stay fp32 in HBM here, so the inputs are
            # rounded through fp16 instead -- the same values
            all_args = tuple(t.half().float() if isinstance(t, torch.Tensor) and t.is_floating_point() else t for t in all_args)
        args, kw_values = all_args[:split], all_args[split:]
        return fn(model, *args, **dict(zip(keys, kw_values)))
    return inner


def _cut(value, size, n_chunks):
    """One share per gradient-accumulation chunk: tensors and sequences are cut along the batch, anything else is repeated."""
    if isinstance(value, torch.Tensor):
        return value.split(size, dim=0)
    if isinstance(value, (list, tuple)):
        return [value[s:s + size] for s in range(0, len(value), size)]
    return [value] * n_chunks


def split_args_and_kwargs(*args, split_size=None, **kwargs):
    """Gradient-accumulation chunks of a call (the reference's generator of the same name, trainer.py:176-197): yields
    ``(share of the batch, (args, kwargs))`` per chunk of at most ``split_size`` samples; the batch size is that of the first
    tensor argument."""
    values = list(args) + list(kwargs.values())
    batch = next((len(v) for v in values if isinstance(v, torch.Tensor)), None)
    assert batch is not None, 'split_args_and_kwargs needs at least one tensor argument'
    size = batch if split_size is None else split_size
    starts = range(0, batch, size)
    columns = [_cut(v, size, len(starts)) for v in values]
    for c, start in enumerate(starts):
        piece = [col[c] for col in columns]
        yield min(size, batch - start) / batch, (tuple(piece[:len(args)]), dict(zip(kwargs, piece[len(args):])))


def imagen_sample_in_chunks(fn):
    @wraps(fn)
    def inner(self, *args, max_batch_size=None, **kwargs):
        if not exists(max_batch_size):
            return fn(self, *args, **kwargs)
        if self.imagen.unconditional:
            batch_sizes = num_to_groups(kwargs.get('batch_size'), max_batch_size)
            outputs = [fn(self, *args, **{**kwargs, 'batch_size': b}) for b in batch_sizes]
        else:
            outputs = [fn(self, *ca, **ck) for _, (ca, ck) in split_args_and_kwargs(*args, split_size=max_batch_size, **kwargs)]
        if isinstance(outputs[0], torch.Tensor):
            return torch.cat(outputs, dim=0)
        return list(map(lambda t: torch.cat(t, dim=0), list(zip(*outputs))))
    return inner


def restore_parts(state_dict_target, state_dict_from):
    """Partial load (trainer.py:222-233): every entry whose name AND shape match is copied, mismatching shapes are reported."""
    for name, src in state_dict_from.items():
        dst = state_dict_target.get(name)
        if dst is None:
            continue
        if tuple(dst.shape) == tuple(src.shape):
            dst.copy_(src)
        else:
            print(f'skipped {name}: checkpoint {tuple(src.shape)} vs model {tuple(dst.shape)}')
    return state_dict_target


# ----------------------------------------------------------------------------------------------
# fused Adam over a flat arena (torch.optim.Adam semantics, trainer.py:352-359)
# ----------------------------------------------------------------------------------------------
class FusedAdam:
    """Adam(lr, betas, eps, weight_decay=0) whose state lives in flat fp32 arenas.  The arena is created lazily
    (after the model sits on its device); until then / on CPU the per-parameter state mirrors torch's so that
    ``state_dict()`` is interchangeable with ``torch.optim.Adam``'s."""

    def __init__(self, params, lr=1e-4, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.0):
        self.params = [p for p in params]
        self.param_groups = [dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=False,
                                  maximize=False, foreach=None, capturable=False, differentiable=False, fused=None,
                                  decoupled_weight_decay=False, params=list(range(len(self.params))))]
        self.arena = None
        self.exp_avg = self.exp_avg_sq = None
        self.step_count = 0
        self._pending_state = None
        self.stateful = set()      # parameter indices with Adam state: torch creates it lazily, for parameters that had a gradient

    def attach(self, arena: D.FlatArena):
        self.arena = arena
        self.exp_avg = torch.zeros_like(arena.flat)
        self.exp_avg_sq = torch.zeros_like(arena.flat)
        if self._pending_state is not None:
            self._load_into_arena(self._pending_state)
            self._pending_state = None

    step_was_skipped = False

    def step(self, max_grad_norm=None, scaler=None):
        """One Adam step over the arena, fused with zero_grad.  ``max_grad_norm``: the reference's clip in front of the step
        (trainer.py:1054, torch's clip_grad_norm_): one norm reduction over the flat gradients, the coefficient is applied inside the
        Adam pass (the gradients are consumed and zeroed there, so scaling them in place would be a wasted pass)."""
        assert self.arena is not None, 'optimizer used before the trainer prepared the unet'
        g = self.param_groups[0]
        self.step_was_skipped = False
        if scaler is not None and scaler.enabled:
            # GradScaler.step: the gradients carry the loss scale.  One norm reduction over the arena serves the finite check (read back:
            # the one host sync of an fp16 optimiser step), the un-scaling and the clip -- all applied as ONE coefficient inside the Adam pass
            import math
            out = ops.grad_norm_clip(self.arena.grad, 1e30)
            norm = float(out[0].item())
            if not math.isfinite(norm):
                self.arena.grad.zero_()                    # optimizer.zero_grad() still runs (trainer.py:1057)
                scaler.update(True)
                self.step_was_skipped = True
                return
            inv = 1.0 / scaler.get_scale()
            c = inv if max_grad_norm is None else inv * min(1.0, max_grad_norm / (norm * inv + 1e-6))
            self.last_grad_norm = out
            self.step_count += 1
            self.stateful |= self.arena.touched
            ops.adam_step(self.arena.flat, self.arena.grad, self.exp_avg, self.exp_avg_sq, g['lr'], g['betas'][0], g['betas'][1], g['eps'],
                          g['weight_decay'], self.step_count, zero_grad=True,
                          grad_scale=torch.tensor([c], dtype=torch.float32, device=self.arena.grad.device))
            scaler.update(False)
            return
        self.step_count += 1
        self.stateful |= self.arena.touched
        coef = None
        if max_grad_norm is not None:
            self.last_grad_norm = ops.grad_norm_clip(self.arena.grad, max_grad_norm)     # [norm, coefficient] on the device
            coef = self.last_grad_norm[1:]
        ops.adam_step(self.arena.flat, self.arena.grad, self.exp_avg, self.exp_avg_sq, g['lr'], g['betas'][0],
                      g['betas'][1], g['eps'], g['weight_decay'], self.step_count, zero_grad=True, grad_scale=coef)

    def clip_accumulated(self, max_grad_norm):
        """The reference clips in EVERY ``update`` call (trainer.py:1054 has no sync check): on an accumulation micro-step that rescales
        the gradients accumulated so far, in place, before the next micro-step adds to them (pinned by trainerA_trace_opts.npz)."""
        self.last_grad_norm = ops.grad_norm_clip(self.arena.grad, max_grad_norm)
        g = self.arena.grad
        g.copy_(ops.axpby3(g.view(1, -1), None, None, self.last_grad_norm[1:], None, None).view(-1))

    def zero_grad(self, set_to_none=False):
        if self.arena is not None:
            self.arena.grad.zero_()

    def state_dict(self):
        state = {}
        if self.arena is not None and self.step_count > 0:
            # like torch.optim.Adam, parameters that never had a gradient (mid_block / norm_cond of the C2 net) carry no state --
            # pinned by the reference-made manifest tests/golden/ckpt_manifest.npz; gradients that bypassed the arena's
            # collect() leave no record, then every parameter is listed
            listed = sorted(self.stateful) if self.stateful else range(len(self.arena.params))
            for i in listed:
                p, o = self.arena.params[i], self.arena.offsets[i]
                n = p.numel()
                state[i] = dict(step=torch.tensor(float(self.step_count)),
                                exp_avg=self.exp_avg[o:o + n].view_as(p).clone(),
                                exp_avg_sq=self.exp_avg_sq[o:o + n].view_as(p).clone())
        groups = [{k: v for k, v in g.items()} for g in self.param_groups]
        return dict(state=state, param_groups=groups)

    def _load_into_arena(self, sd):
        for i, st in sd.get('state', {}).items():
            i = int(i)
            p, o = self.arena.params[i], self.arena.offsets[i]
            n = p.numel()
            self.exp_avg[o:o + n].copy_(st['exp_avg'].reshape(-1))
            self.exp_avg_sq[o:o + n].copy_(st['exp_avg_sq'].reshape(-1))
            self.step_count = max(self.step_count, int(float(st['step'])))
            self.stateful.add(i)

    def load_state_dict(self, sd):
        for k in ('lr', 'betas', 'eps', 'weight_decay'):
            if sd.get('param_groups'):
                self.param_groups[0][k] = sd['param_groups'][0].get(k, self.param_groups[0][k])
        if self.arena is None:
            self._pending_state = sd
        else:
            self._load_into_arena(sd)


class _LrCarrier(torch.optim.SGD):
    """A real ``torch.optim.Optimizer`` whose only job is to hold the learning rate: torch's own ``CosineAnnealingLR`` / ``LambdaLR``
    (what the reference constructs, trainer.py:368-375) run on it unchanged, and ``ImagenTrainer.update`` copies its lr into the fused
    Adam's param group.  It owns one dummy parameter and never steps it."""

    def __init__(self, lr):
        super().__init__([torch.nn.Parameter(torch.zeros(1))], lr=lr)


class _LinearWarmup:
    """``pytorch_warmup.LinearWarmup`` 0.1.1 (trainer.py:372, 1065) restated from its published algorithm -- the package is absent from
    this image and from the reference tree, so this part is **parity unpinned**: the learning rate is damped by
    ``omega(step) = min(1, (step + 1) / warmup_period)``; construction damps once (step 0); ``dampening()`` restores the undamped rates,
    lets the wrapped scheduler act, remembers the new rates and damps them for the next step."""

    def __init__(self, optimizer, warmup_period):
        assert isinstance(warmup_period, int) and warmup_period > 0
        self.optimizer, self.warmup_period = optimizer, warmup_period
        self.last_step = -1
        self.lrs = [g['lr'] for g in optimizer.param_groups]
        self.dampen()

    def warmup_factor(self, step):
        return min(1.0, (step + 1) / self.warmup_period)

    def dampen(self, step=None):
        step = self.last_step + 1 if step is None else step
        self.last_step = step
        for g in self.optimizer.param_groups:
            g['lr'] *= self.warmup_factor(step)

    @contextmanager
    def dampening(self):
        for g, lr in zip(self.optimizer.param_groups, self.lrs):
            g['lr'] = lr
        yield
        self.lrs = [g['lr'] for g in self.optimizer.param_groups]
        self.dampen()

    def state_dict(self):
        return {k: v for k, v in self.__dict__.items() if k != 'optimizer'}

    def load_state_dict(self, sd):
        self.__dict__.update(sd)


class _GradScaler:
    """``torch.cuda.amp.GradScaler`` for the flat-arena optimiser -- what the reference pairs with ``fp16=True`` (trainer.py:309-311, 364:
    ``GradScaler(enabled=fp16)``, set on the accelerator per U-Net): the loss is multiplied by ``scale`` before ``backward``, the optimiser
    step un-scales inside the fused Adam pass (one coefficient, together with the clip), a non-finite gradient norm skips the step and
    halves the scale, ``growth_interval`` good steps in a row double it.  Same state keys and update rule as torch's (pinned against
    ``torch.amp.GradScaler`` on the host, tests/test_host_trainer.py); the finite check is the arena's norm reduction, read back once per
    optimiser step."""

    def __init__(self, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
        self._scale, self.growth_factor, self.backoff_factor, self.growth_interval = float(init_scale), growth_factor, backoff_factor, growth_interval
        self._growth_tracker = 0

    enabled = True

    def get_scale(self):
        return self._scale

    def update(self, found_inf):
        if found_inf:
            self._scale *= self.backoff_factor
            self._growth_tracker = 0
        else:
            self._growth_tracker += 1
            if self._growth_tracker == self.growth_interval:
                self._scale *= self.growth_factor
                self._growth_tracker = 0

    def state_dict(self):
        return {"scale": self._scale, "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor,
                "growth_interval": self.growth_interval, "_growth_tracker": self._growth_tracker}

    def load_state_dict(self, sd):
        if not sd:
            return
        self._scale, self.growth_factor, self.backoff_factor = float(sd["scale"]), sd["growth_factor"], sd["backoff_factor"]
        self.growth_interval, self._growth_tracker = sd["growth_interval"], int(sd["_growth_tracker"])


class _NullScaler:
    """fp32 path: the reference's GradScaler(enabled=False) (trainer.py:364) — empty state."""

    enabled = False

    def get_scale(self):
        return 1.0

    def state_dict(self):
        return {}

    def load_state_dict(self, sd):
        pass


class EMA(nn.Module):
    """ema_pytorch==0.1.4 semantics restated from its published algorithm (the package is absent here: parity
    unpinned): ``update_every=10``, copy until ``update_after_step=100``, then
    decay = clamp(1 - (1 + epoch/inv_gamma)^-power, min_value, beta) with beta 0.9999, inv_gamma 1, power 2/3.
    The lerp is one HIP kernel over the flat arenas when the model is on the GPU."""

    def __init__(self, model, beta=0.9999, update_after_step=100, update_every=10, inv_gamma=1.0, power=2 / 3,
                 min_value=0.0, **_):
        super().__init__()
        self.online_model = model
        self.ema_model = copy.deepcopy(model)
        self.ema_model.requires_grad_(False)
        self.beta, self.update_after_step, self.update_every = beta, update_after_step, update_every
        self.inv_gamma, self.power, self.min_value = inv_gamma, power, min_value
        self.register_buffer('initted', torch.tensor([False]))
        self.register_buffer('step', torch.tensor([0]))
        self._arena = None

    def restore_ema_model_device(self):
        self.ema_model.to(self.initted.device)

    def get_current_decay(self):
        epoch = max(self.step.item() - self.update_after_step - 1, 0.)
        value = 1 - (1 + epoch / self.inv_gamma) ** -self.power
        return 0. if epoch <= 0 else min(max(value, self.min_value), self.beta)

    def _flat(self):
        """(ema_flat, online_flat): both models as flat arenas on the online model's device."""
        on = getattr(self.online_model, '_diqt_arena', None)
        if on is None or not on.intact():
            on = D.FlatArena(list(self.online_model.parameters()), with_grad=False)
            self.online_model._diqt_arena = on
        if self._arena is None or not self._arena.intact() or self._arena.flat.device != on.flat.device:
            self.ema_model.to(on.flat.device)
            self._arena = D.FlatArena(list(self.ema_model.parameters()), with_grad=False)
        return self._arena.flat, on.flat

    def update(self):
        step = self.step.item()
        self.step += 1
        if (step % self.update_every) != 0:
            return
        ema_flat, online_flat = self._flat()
        # ema_pytorch 0.1.4 control flow: warm-up copies leave ``initted`` untouched; the first update past the warm-up
        # copies the online weights, sets ``initted`` and then lerps (a no-op on identical weights), later ones only lerp
        if step <= self.update_after_step:
            ema_flat.copy_(online_flat)                      # plain copy (no arithmetic)
            ops.bump_weight_epoch()
            return
        if not self.initted.item():
            ema_flat.copy_(online_flat)
            ops.bump_weight_epoch()
            self.initted.data.copy_(torch.tensor([True]))
        ops.ema_lerp(ema_flat, online_flat, 1. - self.get_current_decay())     # HIP kernel; raises on CPU tensors

    def forward(self, *args, **kwargs):
        return self.ema_model(*args, **kwargs)


class _ReplicaUnet(nn.Module):
    """What ``accelerator.prepare(unet)`` returned in the reference (a DDP wrapper exposing ``.module``)."""

    def __init__(self, module, reducer):
        super().__init__()
        self.module = module
        self.reducer = reducer

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)


# ----------------------------------------------------------------------------------------------
class ImagenTrainer(nn.Module):
    locked = False

    def __init__(self, configs, imagen=None, imagen_checkpoint_path=None, use_ema=True, lr=1e-4, eps=1e-8, beta1=0.9,
                 beta2=0.99, max_grad_norm=None, group_wd_params=True, warmup_steps=None, cosine_decay_max_steps=None,
                 only_train_unet_number=None, fp16=False, precision=None, split_batches=True,
                 dl_tuple_output_keywords_names=('images', 'lowres_img', 'text_embeds', 'text_masks', 'cond_images'),
                 verbose=True, split_valid_fraction=0.025, split_valid_from_train=False, split_random_seed=42,
                 checkpoint_path=None, checkpoint_every=None, checkpoint_fs=None, fs_kwargs: dict = None,
                 max_checkpoints_keep=20, gradient_accumulation_steps=4, **kwargs):
        super().__init__()
        assert not ImagenTrainer.locked, 'ImagenTrainer can only be initialized once per process - for the sake of distributed training, you will now have to create a separate script to train each unet (or a script that accepts unet number as an argument)'
        assert exists(imagen) ^ exists(imagen_checkpoint_path), 'either imagen instance is passed into the trainer, or a checkpoint path that contains the imagen config'
        assert exists(imagen), 'imagen_checkpoint_path (CLI configs) is dead code in the reference and not built'
        # mixed precision (trainer.py:293-311: Accelerator(mixed_precision=...)): the model forward runs under torch.autocast, where
        # conv3d / linear forwards take the fp16 / bf16 MFMA kernel (ops.lp_mode); master weights, accumulated gradients, loss and the
        # optimiser stay fp32.  bf16: the conv backward kernels run in bf16 too.  `fp16=True`: the reference's GradScaler(enabled=fp16)
        # (trainer.py:311, 364) is real here as well (_GradScaler: scaled loss, fp16 backward kernels, un-scale + finite check + skip inside
        # the optimiser step); precision='fp16' WITHOUT the fp16 switch has no scaler in the reference either: its backward stays fp32 here
        assert not (fp16 and exists(precision)), 'either set fp16 = True or forward the precision ("fp16", "bf16") to Accelerator'
        self.mixed_precision = default(precision, 'fp16' if fp16 else 'no')
        assert self.mixed_precision in ('no', 'fp16', 'bf16'), self.mixed_precision
        self.cast_half_at_training = self.mixed_precision == 'fp16'
        self.configs = configs
        ema_kwargs, kwargs = groupby_prefix_and_trim('ema_', kwargs)
        _, kwargs = groupby_prefix_and_trim('accelerate_', kwargs)
        from .elucidated_imagen import ElucidatedImagen
        assert isinstance(imagen, (Imagen, ElucidatedImagen))
        self.is_elucidated = isinstance(imagen, ElucidatedImagen)

        # one process per GPU; RANK/LOCAL_RANK/WORLD_SIZE from the launcher (torchrun)
        self.world_size, self.rank, self._device = D.init_from_env()
        self.split_batches = split_batches
        self.gradient_accumulation_steps = gradient_accumulation_steps
        self._micro_step = 0
        self._train_graphs = graphs.TrainStepGraphs()          # captured micro-steps (launch-bound ones only)
        ImagenTrainer.locked = self.is_distributed

        self.imagen = imagen
        self.num_unets = len(self.imagen.unets)
        self.use_ema = use_ema and self.is_main
        self.ema_unets = nn.ModuleList([])
        self.train_dl_iter = self.train_dl = self.valid_dl_iter = self.valid_dl = None
        self.dl_tuple_output_keywords_names = dl_tuple_output_keywords_names
        self.split_valid_from_train = split_valid_from_train
        assert 0 <= split_valid_fraction <= 1, 'split valid fraction must be between 0 and 1'
        self.split_valid_fraction, self.split_random_seed = split_valid_fraction, split_random_seed

        lr, eps, warmup_steps, cosine_decay_max_steps = (cast_tuple(v, self.num_unets) for v in (lr, eps, warmup_steps, cosine_decay_max_steps))
        for ind, (unet, unet_lr, unet_eps, unet_warmup, unet_cosine) in enumerate(zip(self.imagen.unets, lr, eps, warmup_steps,
                                                                                       cosine_decay_max_steps)):
            optimizer = FusedAdam(unet.parameters(), lr=unet_lr, eps=unet_eps, betas=(beta1, beta2), **kwargs)
            setattr(self, f'optim{ind}', optimizer)
            if self.use_ema:
                self.ema_unets.append(EMA(unet, **ema_kwargs))
            # LR schedules (trainer.py:366-375): torch's own schedulers on an lr carrier; `eta_min = lr[1] * 0.001` is the reference's
            # expression (the SECOND U-Net's rate, whichever U-Net the scheduler belongs to)
            scheduler = warmup_scheduler = None
            if exists(unet_cosine) or exists(unet_warmup):
                optimizer.lr_carrier = _LrCarrier(unet_lr)
            if exists(unet_cosine):
                scheduler = CosineAnnealingLR(optimizer.lr_carrier, T_max=unet_cosine, eta_min=lr[1] * 0.001)
            if exists(unet_warmup):
                warmup_scheduler = _LinearWarmup(optimizer.lr_carrier, warmup_period=unet_warmup)
                if not exists(scheduler):
                    scheduler = LambdaLR(optimizer.lr_carrier, lr_lambda=lambda step: 1.0)
                optimizer.param_groups[0]['lr'] = optimizer.lr_carrier.param_groups[0]['lr']     # damped from the first step on
            # trainer.py:311, 364: GradScaler(enabled = fp16) -- only the `fp16=True` switch turns it on (precision='fp16' alone does not)
            setattr(self, f'scaler{ind}', _GradScaler() if fp16 else _NullScaler())
            setattr(self, f'scheduler{ind}', scheduler)
            setattr(self, f'warmup{ind}', warmup_scheduler)
        self.max_grad_norm = max_grad_norm
        self.register_buffer('steps', torch.tensor([0] * self.num_unets))
        self.verbose = verbose
        self.imagen.to(self.device)
        self.to(self.device)

        assert not (exists(checkpoint_path) ^ exists(checkpoint_every))
        self.checkpoint_path, self.checkpoint_every, self.max_checkpoints_keep = checkpoint_path, checkpoint_every, max_checkpoints_keep
        self.can_checkpoint = self.is_main
        if exists(checkpoint_path) and self.can_checkpoint:
            os.makedirs(checkpoint_path, exist_ok=True)
            self.load_from_checkpoint_folder()
        self.only_train_unet_number = only_train_unet_number
        self.prepared = False
        self.valid_images_save = False

    # ---- properties ---------------------------------------------------------------------------------
    @property
    def device(self):
        return self._device

    @property
    def is_distributed(self):
        return self.world_size > 1

    @property
    def is_main(self):
        return self.rank == 0

    @property
    def is_local_main(self):
        return D.env_world()[2] == 0

    @property
    def unwrapped_unet(self):
        return self.unet_being_trained.module

    def print(self, msg):
        if self.is_main and self.verbose:
            print(msg)

    def get_lr(self, unet_number):
        self.validate_unet_number(unet_number)
        return getattr(self, f'optim{unet_number - 1}').param_groups[0]['lr']

    # ---- preparing the unet being trained (the reference's accelerator.prepare / DDP wrap) ----------
    def prepare(self):
        assert not self.prepared, 'The trainer is allready prepared'
        self.validate_and_set_unet_being_trained(self.only_train_unet_number)
        self.prepared = True

    def validate_and_set_unet_being_trained(self, unet_number=None):
        if exists(unet_number):
            self.validate_unet_number(unet_number)
        assert not exists(self.only_train_unet_number) or self.only_train_unet_number == unet_number, \
            'you cannot only train on one unet at a time. you will need to save the trainer into a checkpoint, and resume training on a new unet'
        self.only_train_unet_number = unet_number
        self.imagen.only_train_unet_number = unet_number
        if not exists(unet_number):
            return
        self.wrap_unet(unet_number)

    def wrap_unet(self, unet_number):
        if hasattr(self, 'one_unet_wrapped'):
            return
        unet = self.imagen.get_unet(unet_number)
        optimizer = getattr(self, f'optim{unet_number - 1}')
        arena = D.FlatArena(list(unet.parameters()))
        unet._diqt_arena = arena
        optimizer.attach(arena)
        D.broadcast_arena(arena)                         # rank-0 weights to every replica (trainer.py:487)
        if self.is_distributed:
            # only rank 0 reads the checkpoint folder (can_checkpoint): a resumed run must hand its Adam moments, bias-correction
            # step, ``steps`` and the accumulation phase to every replica too, or they step differently from the first update on
            # and hit the checkpoint barrier on different micro-steps
            D.broadcast_tensors([optimizer.exp_avg, optimizer.exp_avg_sq, self.steps])
            optimizer.step_count, self._micro_step = D.broadcast_ints([optimizer.step_count, self._micro_step], arena.flat.device)
        reducer = D.BucketedGradReducer(arena) if self.is_distributed else None
        self._arena = arena
        self.unet_being_trained = _ReplicaUnet(unet, reducer)
        self.one_unet_wrapped = True

    def validate_unet_number(self, unet_number=None):
        if self.num_unets == 1:
            unet_number = default(unet_number, 1)
        assert 0 < unet_number <= self.num_unets, f'unet number should be in between 1 and {self.num_unets}'
        return unet_number

    def num_steps_taken(self, unet_number=None):
        if self.num_unets == 1:
            unet_number = default(unet_number, 1)
        return self.steps[unet_number - 1].item()

    def print_untrained_unets(self):
        flag = False
        for ind, (steps, unet) in enumerate(zip(self.steps.tolist(), self.imagen.unets)):
            if steps > 0 or isinstance(unet, NullUnet):
                continue
            self.print(f'unet {ind + 1} has not been trained')
            flag = True
        if flag:
            self.print('when sampling, you can pass stop_at_unet_number to stop early in the cascade, so it does not try to generate with untrained unets')

    # ---- data -----------------------------------------------------------------------------------------
    def add_train_dataloader(self, dl=None):
        if not exists(dl):
            return
        assert not exists(self.train_dl), 'training dataloader was already added'
        assert not self.prepared, 'You need to add the dataset before preperation'
        self.train_dl = dl

    def add_valid_dataloader(self, dl):
        if not exists(dl):
            return
        assert not exists(self.valid_dl), 'validation dataloader was already added'
        assert not self.prepared, 'You need to add the dataset before preperation'
        self.valid_dl = dl

    def add_train_dataset(self, ds=None, *, batch_size, **dl_kwargs):
        if not exists(ds):
            return
        assert not exists(self.train_dl), 'training dataloader was already added'
        valid_ds = None
        if self.split_valid_from_train:
            train_size = int((1 - self.split_valid_fraction) * len(ds))
            valid_size = len(ds) - train_size
            ds, valid_ds = random_split(ds, [train_size, valid_size], generator=torch.Generator().manual_seed(self.split_random_seed))
            self.print(f'training with dataset of {len(ds)} samples and validating with randomly splitted {len(valid_ds)} samples')
        self.add_train_dataloader(DataLoader(ds, batch_size=batch_size, **dl_kwargs))
        if self.split_valid_from_train:
            self.add_valid_dataset(valid_ds, batch_size=batch_size, **dl_kwargs)

    def add_valid_dataset(self, ds, *, batch_size, **dl_kwargs):
        if not exists(ds):
            return
        assert not exists(self.valid_dl), 'validation dataloader was already added'
        self.valid_batch_size = batch_size
        self.add_valid_dataloader(DataLoader(ds, batch_size=self.valid_batch_size, **dl_kwargs))

    def create_train_iter(self):
        assert exists(self.train_dl), 'training dataloader has not been registered with the trainer yet'
        if not exists(self.train_dl_iter):
            self.train_dl_iter = cycle(self.train_dl)

    def create_valid_iter(self):
        assert exists(self.valid_dl), 'validation dataloader has not been registered with the trainer yet'
        if not exists(self.valid_dl_iter):
            self.valid_dl_iter = cycle(self.valid_dl)

    # ---- steps ----------------------------------------------------------------------------------------
    def train_step(self, unet_number=None, **kwargs):
        self.training = True
        if not self.prepared:
            self.prepare()
        self.create_train_iter()
        return self.step_with_dl_iter(self.train_dl, unet_number=unet_number, **kwargs)

    @torch.no_grad()
    @eval_decorator
    def valid_step(self, unet_number=None, **kwargs):
        self.training = False
        if not self.prepared:
            self.prepare()
        self.create_valid_iter()
        context = self.use_ema_unets if kwargs.pop('use_ema_unets', False) else nullcontext
        with context():
            np.random.seed(42)
            torch.manual_seed(42)
            loss, preds, condi1, condi2, hrs, ssim, psnr = self.step_with_dl_iter(self.valid_dl, unet_number=unet_number, **kwargs)
        return loss, preds, condi1, [hrs, condi2], ssim, psnr

    def step_with_dl_iter(self, dl_iter, **kwargs):
        """trainer.py:705-765 — one pass (``Eval.repeat`` passes when validating) over the whole dataloader."""
        self.total_loss = 0.
        if self.training:
            self.repeat = 1
        else:
            self.repeat = self.configs['Eval']['repeat']
            preds, condi1, condi2, hrs, ssims, psnrs = [], [], [], [], [], []
        full_bs = getattr(dl_iter, 'batch_size', None)
        for r in range(self.repeat):
            first = None
            for i, data in enumerate(dl_iter):
                hr_data, lr_data = data[0], data[1]
                if self.split_batches and self.is_distributed and not self.configs['Train']['batch_sample']:
                    if first is None:
                        first = (hr_data, lr_data)           # completes a short last batch (drop_last=False), see D.shard_batch
                    hr_data = D.shard_batch(hr_data, self.world_size, self.rank, full=full_bs, initial=first[0])
                    lr_data = D.shard_batch(lr_data, self.world_size, self.rank, full=full_bs, initial=first[1])
                if self.configs['Train']['batch_sample']:
                    new_batch = (hr_data.shape[-1] // self.configs['Train']['patch_size_sub']) ** 3
                    c, h = hr_data.shape[1], self.configs['Train']['patch_size_sub']
                    hr_data = convertVolume2subVolume(hr_data, target_shape=(new_batch, c, h, h, h))
                    lr_data = convertVolume2subVolume(lr_data, target_shape=(new_batch, c, h, h, h))
                model_input = dict(list(zip(self.dl_tuple_output_keywords_names, (hr_data, lr_data))))
                loss, pred, x_noisy, lowres_cond_img_noisy = self.forward(**{**kwargs, **model_input})
                if not self.training:
                    if self.configs['Train']['batch_sample']:
                        pred_merge, hr_merge = merge_sub_volumes(pred), merge_sub_volumes(hr_data)
                    else:
                        pred_merge, hr_merge = pred, hr_data
                    if self.configs['Train']['pred_obj'] == 'x_start':
                        ssims.append(np.asarray(SSIM(pred_merge, hr_merge).cpu()))         # on the device; scalars come back
                        psnrs.append(np.asarray(PSNR(pred_merge, hr_merge).cpu()))
                    if r < 2:
                        preds.append(pred.cpu().numpy())
                        condi1.append(x_noisy.cpu().numpy())
                        condi2.append(lowres_cond_img_noisy.cpu().numpy())
                        hrs.append(hr_data.cpu().numpy())
                self.total_loss += loss
        loss = self.total_loss / (len(dl_iter) * self.repeat)
        if self.training:
            return loss
        return (loss, np.concatenate(preds), np.concatenate(condi1), np.concatenate(condi2), np.concatenate(hrs),
                np.mean(np.array(ssims)), np.mean(np.array(psnrs)))

    # ---- checkpoints (trainer.py:769-945) ---------------------------------------------------------------
    @property
    def all_checkpoints_sorted(self):
        import glob
        cks = glob.glob(os.path.join(self.checkpoint_path, '*.pt'))
        return sorted(cks, key=lambda x: int(str(x).split('.')[-2]), reverse=True)

    def load_from_checkpoint_folder(self, last_total_steps=-1):
        if last_total_steps != -1:
            self.load(os.path.join(self.checkpoint_path, f'checkpoint.{last_total_steps}.pt'))
            return
        cks = self.all_checkpoints_sorted
        if len(cks) == 0:
            self.print(f'no checkpoints found to load from at {self.checkpoint_path}')
            return
        self.load(cks[0])

    def save_to_checkpoint_folder(self):
        D.barrier()
        if not self.can_checkpoint:
            return
        total_steps = int(self.steps.sum().item())
        self.save(os.path.join(self.checkpoint_path, f'checkpoint.{total_steps}.pt'))
        if self.max_checkpoints_keep <= 0:
            return
        for ck in self.all_checkpoints_sorted[self.max_checkpoints_keep:]:
            os.remove(ck)

    def save(self, path, overwrite=True, without_optim_and_sched=False, **kwargs):
        D.barrier()
        if not self.can_checkpoint:
            return
        assert not (os.path.exists(path) and not overwrite)
        self.reset_ema_unets_all_one_device()
        save_obj = dict(model=self.imagen.state_dict(), version=CHECKPOINT_VERSION, steps=self.steps.cpu(), **kwargs)
        for ind in (range(0, self.num_unets) if not without_optim_and_sched else tuple()):
            scheduler, warmup_scheduler = getattr(self, f'scheduler{ind}'), getattr(self, f'warmup{ind}')
            if exists(scheduler):                                                    # key order of trainer.py:851-857
                save_obj = {**save_obj, f'scheduler{ind}': scheduler.state_dict()}
            if exists(warmup_scheduler):
                save_obj = {**save_obj, f'warmup{ind}': warmup_scheduler.state_dict()}
            save_obj = {**save_obj, f'scaler{ind}': getattr(self, f'scaler{ind}').state_dict(),
                        f'optim{ind}': getattr(self, f'optim{ind}').state_dict()}
        if self.use_ema:
            save_obj = {**save_obj, 'ema': self.ema_unets.state_dict()}
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, 'wb') as f:
            torch.save(save_obj, f)
        self._rebind_after_device_moves()
        self.print(f'checkpoint saved to {path}')

    def load(self, path, only_model=False, strict=True, noop_if_not_exist=False):
        if noop_if_not_exist and not os.path.exists(path):
            self.print(f'trainer checkpoint not found at {str(path)}')
            return
        assert os.path.exists(path), f'{path} does not exist'
        self.reset_ema_unets_all_one_device()
        with open(path, 'rb') as f:
            loaded_obj = torch.load(f, map_location='cpu', weights_only=False)
        if str(loaded_obj.get('version')) != CHECKPOINT_VERSION:
            self.print(f'checkpoint written at version {loaded_obj.get("version")}; this trainer reads and writes the {CHECKPOINT_VERSION} layout')
        try:
            self.imagen.load_state_dict(loaded_obj['model'], strict=strict)
        except RuntimeError:
            print("Failed loading state dict. Trying partial load")
            self.imagen.load_state_dict(restore_parts(self.imagen.state_dict(), loaded_obj['model']))
        ops.bump_weight_epoch()
        self._rebind_after_device_moves()
        if only_model:
            return loaded_obj
        self.steps.copy_(loaded_obj['steps'])
        for ind in range(0, self.num_unets):
            scheduler, warmup_scheduler = getattr(self, f'scheduler{ind}'), getattr(self, f'warmup{ind}')
            if exists(scheduler) and f'scheduler{ind}' in loaded_obj:                # trainer.py:922-926
                scheduler.load_state_dict(loaded_obj[f'scheduler{ind}'])
            if exists(warmup_scheduler) and f'warmup{ind}' in loaded_obj:
                warmup_scheduler.load_state_dict(loaded_obj[f'warmup{ind}'])
            try:
                getattr(self, f'optim{ind}').load_state_dict(loaded_obj[f'optim{ind}'])
                getattr(self, f'scaler{ind}').load_state_dict(loaded_obj[f'scaler{ind}'])
                if exists(scheduler):                # the optimiser's param group carries the rate in effect (torch restores it there)
                    getattr(self, f'optim{ind}').lr_carrier.param_groups[0]['lr'] = getattr(self, f'optim{ind}').param_groups[0]['lr']
            except Exception:
                self.print('could not load optimizer and scaler, possibly because you have turned on mixed precision training since the last run. resuming with new optimizer and scalers')
        if self.use_ema:
            assert 'ema' in loaded_obj
            try:
                self.ema_unets.load_state_dict(loaded_obj['ema'], strict=strict)
            except RuntimeError:
                print("Failed loading state dict. Trying partial load")
                self.ema_unets.load_state_dict(restore_parts(self.ema_unets.state_dict(), loaded_obj['ema']))
        self.print(f'checkpoint loaded from {path}')
        return loaded_obj

    def _rebind_after_device_moves(self):
        """``state_dict()``/``load`` shuffle unets across devices like the reference; a broken arena is rebuilt."""
        arena = getattr(self, '_arena', None)
        if arena is not None and not arena.intact():
            raise RuntimeError('the trained unet was moved off its flat arena; re-create the trainer')
        if arena is not None:
            arena.reinstall_grads()

    # ---- EMA unets (trainer.py:949-1005) -----------------------------------------------------------------
    # One process drives one GPU and every U-Net of the cascade lives there, so the reference's shuffling of EMA copies between
    # the device and the host has no counterpart: ``ema_unets`` stays one ModuleList (its state_dict keys ``{i}.ema_model.*`` /
    # ``{i}.online_model.*`` / ``{i}.initted`` / ``{i}.step`` are the checkpoint contract, tests/golden/ckpt_manifest.npz).
    @property
    def unets(self):
        return nn.ModuleList([ema.ema_model for ema in self.ema_unets])

    def get_ema_unet(self, unet_number=None):
        if not self.use_ema:
            return None
        return self.ema_unets[self.validate_unet_number(unet_number) - 1]

    def reset_ema_unets_all_one_device(self, device=None):
        """Kept for callers of the reference API; a no-op unless something moved an EMA copy off the trainer's device."""
        if not self.use_ema:
            return
        target = torch.device(default(device, self.device))
        for ema in self.ema_unets:
            if any(p.device != target for p in ema.ema_model.parameters()):
                ema.ema_model.to(target)

    @torch.no_grad()
    @contextmanager
    def use_ema_unets(self):
        """Inside the context ``self.imagen`` samples with the EMA weights (trainer.py:982-1005)."""
        if not self.use_ema:
            yield
            return
        self.reset_ema_unets_all_one_device()
        self.imagen.reset_unets_all_one_device()
        online = self.imagen.unets
        swapped = self.unets
        swapped.eval()
        self.imagen.unets = swapped
        try:
            yield
        finally:
            self.imagen.unets = online

    def state_dict(self, *args, **kwargs):
        self.reset_ema_unets_all_one_device()
        return super().state_dict(*args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self.reset_ema_unets_all_one_device()
        return super().load_state_dict(*args, **kwargs)

    # ---- optimisation -----------------------------------------------------------------------------------
    def _is_sync_step(self):
        return (self._micro_step % self.gradient_accumulation_steps) == 0

    def update(self, unet_number=None):
        """trainer.py:1038-1081.  The optimiser really steps only when the micro-step counter hit the accumulation
        boundary (``_sync_now``); EMA and ``steps`` advance on every call."""
        unet_number = self.validate_unet_number(unet_number)
        self.validate_and_set_unet_being_trained(unet_number)
        index = unet_number - 1
        optimizer = getattr(self, f'optim{index}')
        scheduler, warmup_scheduler = getattr(self, f'scheduler{index}'), getattr(self, f'warmup{index}')
        stepped = getattr(self, '_sync_now', False)
        scaler = getattr(self, f'scaler{index}')
        if stepped:
            optimizer.step(max_grad_norm=self.max_grad_norm, scaler=scaler)      # (unscale, clip ->) fused Adam + zero_grad (trainer.py:1054-1057)
            self._sync_now = False
            stepped = not optimizer.step_was_skipped        # a skipped step (non-finite fp16 gradients) does not move the LR schedule either
        elif exists(self.max_grad_norm) and self.training and not scaler.enabled:
            # (with a loss scaler the reference's clip on accumulation micro-steps would un-scale twice -- torch raises; clipped at the step only)
            optimizer.clip_accumulated(self.max_grad_norm)
        if self.use_ema:
            self.ema_unets[index].update()
        # trainer.py:1063-1069: the warm-up's dampening() context runs on EVERY update call; the scheduler inside it is the one
        # accelerator.prepare() wrapped (trainer.py:492), which steps only when the gradients were synchronised -- i.e. with the Adam
        # step (pinned by tests/golden/trainerA_trace_opts.npz: the rate moves every gradient_accumulation_steps-th micro-step)
        if exists(scheduler):
            with (warmup_scheduler.dampening() if exists(warmup_scheduler) else nullcontext()):
                if stepped:
                    optimizer.lr_carrier.step()          # no gradients: a no-op that keeps torch's step-order bookkeeping quiet
                    scheduler.step()
            optimizer.param_groups[0]['lr'] = optimizer.lr_carrier.param_groups[0]['lr']
        self.steps[index] += 1
        if exists(self.checkpoint_path) and int(self.steps.sum().item()) % self.checkpoint_every == 0:
            self.save_to_checkpoint_folder()

    @torch.no_grad()
    @cast_torch_tensor
    @imagen_sample_in_chunks
    def sample(self, *args, **kwargs):
        context = nullcontext if kwargs.pop('use_non_ema', False) else self.use_ema_unets
        self.print_untrained_unets()
        kwargs['use_tqdm'] = False
        with context():
            output = self.imagen.sample(*args, device=self.device, **kwargs)
        return output

    def _autocast(self):
        if self.mixed_precision == 'no' or not torch.cuda.is_available():
            return nullcontext()
        return torch.autocast('cuda', dtype=torch.float16 if self.mixed_precision == 'fp16' else torch.bfloat16)

    @partial(cast_torch_tensor, cast_fp16=True)
    def forward(self, *args, unet_number=None, max_batch_size=None, **kwargs):
        """trainer.py:1099-1128 -> (total_loss, pred, x_noisy, lowres)."""
        unet_number = self.validate_unet_number(unet_number)
        self.validate_and_set_unet_being_trained(unet_number)
        self.max_batch_size = max_batch_size
        assert not exists(self.only_train_unet_number) or self.only_train_unet_number == unet_number, \
            f'you can only train unet #{self.only_train_unet_number}'
        if not (self.training and self.device.type == 'cuda' and getattr(self, '_arena', None) is not None):
            return self._forward_chunks(args, kwargs, unet_number, max_batch_size, None)
        # Training steps run on the trainer's own stream, fenced against the caller's on both sides.  autograd's AccumulateGrad nodes
        # remember the stream they were created on and run THERE; a node that outlives its iteration (anything holding a tensor of an
        # old graph does that) would pull the null stream into the capture of a micro-step (graphs.TrainStepGraphs captures on this very
        # stream) -- an un-joined fork that ends the capture in the runtime, not in an exception.  So the nodes are created once, on this
        # stream, and kept (as DDP keeps them).
        ts = self._train_stream = getattr(self, '_train_stream', None) or torch.cuda.Stream(device=self.device)
        cur = torch.cuda.current_stream(self.device)
        ts.wait_stream(cur)
        try:
            with torch.cuda.stream(ts):
                if getattr(self, '_acc_nodes_of', None) is not self._arena:
                    with torch.enable_grad():
                        self._acc_nodes = [p.view_as(p).grad_fn.next_functions[0][0] for p in self._arena.params if p.requires_grad]
                    self._acc_nodes_of = self._arena
                out = self._forward_chunks(args, kwargs, unet_number, max_batch_size, ts)
        finally:
            cur.wait_stream(ts)
        for t in out[1:]:
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(cur)                    # allocated on the training stream, read by the caller on its own
        return out

    def _forward_chunks(self, args, kwargs, unet_number, max_batch_size, stream):
        total_loss = 0.
        reducer = self.unet_being_trained.reducer
        arena = getattr(self, '_arena', None)
        scaler = getattr(self, f'scaler{unet_number - 1}')
        for chunk_size_frac, (chunked_args, chunked_kwargs) in split_args_and_kwargs(*args, split_size=max_batch_size, **kwargs):
            self._micro_step += 1
            sync = self._is_sync_step()
            # the loss scale of this micro-step (accelerator.backward: scaler.scale(loss).backward(); accumulate: / accumulation steps)
            back_scale = (scaler.get_scale() if scaler.enabled else 1.0) / self.gradient_accumulation_steps

            def device_work(core, tensors):
                """Forward, loss, backward, gradients handed to the arena: device launches only (captured as one hipGraph when the step
                is launch-bound, graphs.TrainStepGraphs)."""
                with self._autocast():
                    out = core(*tensors)
                # Imagen.forward returns (loss, pred, x_noisy, lowres); ElucidatedImagen.forward a scalar loss (the reference
                # trainer crashes on the latter, trainer.py:1119 — driving EDM is a superset feature here)
                loss, pred, x_noisy, lowres = out if isinstance(out, tuple) else (out, None, None, None)
                loss = loss * chunk_size_frac
                if self.training:
                    if arena is not None:
                        arena.begin_backward()
                    if scaler.enabled:                  # fp16 kernels in the backward too
                        ops.FP16_BACKWARD = True
                    try:
                        (loss * back_scale).backward()
                    finally:
                        ops.FP16_BACKWARD = False
                # handed back without their graph (the reference's are attached to a graph backward() has already freed): a caller
                # keeping `pred` would otherwise keep every node of this micro-step -- and its AccumulateGrad nodes -- alive
                det = lambda t: t.detach() if torch.is_tensor(t) else t
                return loss.detach(), det(pred), det(x_noisy), det(lowres)

            if exists(reducer) and self.training:
                reducer.prepare_backward(sync=sync)     # like accelerate.accumulate: DDP syncs on the boundary
            # a captured micro-step holds no collective: with a live reducer only the accumulation steps in between go through the graph
            graphable = (self.training and arena is not None and not self.is_elucidated and self.device.type == 'cuda'
                         and not (exists(reducer) and reducer.active and sync))
            if graphable:
                core, tensors = self.imagen(*chunked_args, unet=self.unet_being_trained, unet_number=unet_number, deferred=True,
                                            **chunked_kwargs)

                def step(*ts):
                    out = device_work(core, ts)
                    arena.collect()
                    return out
                key = (unet_number, self.mixed_precision, chunk_size_frac, back_scale, arena.grad.data_ptr(), arena.flat.data_ptr(),
                       sum(1 for p in arena.params if p.requires_grad), core.static_key)
                loss, pred, x_noisy, lowres_cond_img_noisy = self._train_graphs.run(
                    key, step, tensors, arena.reinstall_grads, stream,
                    prepare=(lambda: ops.repack_cached_h(self.unet_being_trained.module, 1 if self.mixed_precision == 'bf16' else 0))
                    if self.mixed_precision != 'no' else None)
            else:
                loss, pred, x_noisy, lowres_cond_img_noisy = device_work(
                    lambda: self.imagen(*chunked_args, unet=self.unet_being_trained, unet_number=unet_number, **chunked_kwargs), ())
            if self.training:
                if exists(reducer):
                    reducer.finalize_backward()
                if arena is not None and not graphable:
                    arena.collect()             # whatever no bucket launch has collected yet (all of it on non-sync steps)
                self._sync_now = sync
                self.update(unet_number=unet_number)
            total_loss += loss.item()
        return total_loss, pred, x_noisy, lowres_cond_img_noisy
