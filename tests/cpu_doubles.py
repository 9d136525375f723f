"""TEST DOUBLES (tests only): plain-torch stand-ins for a few ``diffusioniqt_amd.ops`` entry points so that the
HOST logic of the trainer / DDPM wrapper / reducer (chunking, accumulation cadence, EMA schedule, checkpoint
layout, gloo sharding) can be exercised by the ``-m "not gpu"`` suite, where no MI355X exists.  The product
never imports this file; on a GPU box the real HIP ops run and are checked by the ``-m gpu`` tests."""
import contextlib

import torch
from torch import nn

from oracle import iqt_oracle as O


def _q_sample(x0, noise, alpha, sigma):
    sh = (-1,) + (1,) * (x0.dim() - 1)
    return alpha.view(sh) * x0 + sigma.view(sh) * noise


def _mse_clamp(pred, target, lo=0.0, do_clamp=False, weight=None, kind='l2'):
    p = pred.clamp(min=lo) if do_clamp else pred
    fn = {'l2': torch.nn.functional.mse_loss, 'l1': torch.nn.functional.l1_loss, 'huber': torch.nn.functional.smooth_l1_loss}[kind]
    losses = fn(p, target, reduction='none').flatten(1).mean(1)
    if weight is not None:
        losses = losses * weight
    return losses.mean(), p.detach()


def _grad_norm_clip(grad_flat, max_norm):
    norm = grad_flat.double().pow(2).sum().sqrt().float()
    return torch.stack([norm, (max_norm / (norm + 1e-6)).clamp(max=1.0)])


def _adam_step(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step, zero_grad=True, grad_scale=None):
    if grad_scale is not None:
        grad.mul_(grad_scale)
    g = grad + weight_decay * param if weight_decay else grad
    exp_avg.lerp_(g, 1 - beta1)
    exp_avg_sq.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1, bc2 = 1 - beta1 ** step, 1 - beta2 ** step
    param.addcdiv_(exp_avg, (exp_avg_sq.sqrt() / bc2 ** 0.5).add_(eps), value=-lr / bc1)
    if zero_grad:
        grad.zero_()


def _ema_lerp(ema, param, w):
    ema.lerp_(param, w)


def _ddpm_step(x_t, pred, noise, ca, cb, cn, lo, hi, clamp_mode):
    sh = (-1,) + (1,) * (x_t.dim() - 1)
    x0 = pred.clamp(min=lo) if clamp_mode == 0 else pred.clamp(lo, hi)
    return ca.view(sh) * x_t + cb.view(sh) * x0 + cn.view(sh) * noise, x0


def _axpby3(a, b, c, c0, c1, c2, lo=0.0, hi=0.0, clamp_mode=0):
    sh = (-1,) + (1,) * (a.dim() - 1)
    v = c0.view(sh) * a
    if b is not None:
        v = v + c1.view(sh) * b
    if c is not None:
        v = v + c2.view(sh) * c
    return v.clamp(min=lo) if clamp_mode == 1 else (v.clamp(lo, hi) if clamp_mode == 2 else v)


@contextlib.contextmanager
def cpu_op_doubles():
    from diffusioniqt_amd import ops
    names = dict(q_sample=_q_sample, mse_clamp=_mse_clamp, adam_step=_adam_step, ema_lerp=_ema_lerp,
                 ddpm_step=_ddpm_step, axpby3=_axpby3, grad_norm_clip=_grad_norm_clip)
    saved = {k: getattr(ops, k) for k in names}
    for k, v in names.items():
        setattr(ops, k, v)
    from diffusioniqt_amd import trainer as T                       # valid_step's device metrics -> the CPU restatements
    from oracle import iqt_data_oracle as DO
    saved_metrics = (T.SSIM, T.PSNR)
    T.SSIM, T.PSNR = (lambda p, t: DO.ssim(p, t)), (lambda p, t: DO.psnr(p, t))
    try:
        yield
    finally:
        for k, v in saved.items():
            setattr(ops, k, v)
        T.SSIM, T.PSNR = saved_metrics


class OracleUnet(nn.Module):
    """A CPU 'unet' for host-logic tests: parameters named like the reference, forward = the CPU oracle."""
    lowres_cond = True
    self_cond = False

    def __init__(self, state_dict, cfg):
        super().__init__()
        self.cfg = cfg
        self.names = list(state_dict.keys())
        self.plist = nn.ParameterList([nn.Parameter(v.clone()) for v in state_dict.values()])

    def cast_model_parameters(self, **kw):
        return self

    def named_sd(self):
        return dict(zip(self.names, self.plist))

    def forward(self, x, time_steps=None, time=None, *, lowres_cond_img=None, cond_images=None, cond_drop_prob=0.,
                self_cond=None):
        return O.unet_forward(self.named_sd(), self.cfg, x, time_steps, time, lowres_cond_img=lowres_cond_img)

    def forward_with_cond_scale(self, *a, cond_scale=1., **k):
        return self.forward(*a, **k)
