"""MI355X-native counterpart of the reference's ``elucidated_imagen.ElucidatedImagen`` (EDM / Karras et al.):
preconditioning (elucidated_imagen.py:314-358), sigma schedule (:365-379), stochastic Heun sampler (:382-532),
cascade ``sample`` (:536-702) and the weighted training loss (:706-882), text-free IQT configuration.

Every per-voxel operation is a HIP kernel (``ops.axpby3`` folds the scalar coefficients of a sampler sub-step into
ONE pass; the training loss is one fused weighted-MSE kernel).  The sigma / gamma / c_* scalars are host Python
floats computed once before the loop (the reference pulls them from the device with ``.item()`` every step, :471).

Superset of the reference: it also drives the true-Conv3d Family-A ``Unet`` (``forward(x, c_noise)``), which the
reference cannot.
"""
from collections import namedtuple
from contextlib import contextmanager
from functools import partial
from math import sqrt
from random import random

import torch
from torch import nn

from . import ops
from .graphs import GraphCache
from .imagen_pytorch3D import (GaussianDiffusionContinuousTimes, Unet, NullUnet, exists, default, cast_tuple, identity, maybe,
                               normalize_neg_one_to_one, unnormalize_zero_to_one, eval_decorator, to_channels_last,
                               to_channels_first, log_snr_to_alpha_sigma)
from .imagen_video import Unet3D

Hparams_fields = ['num_sample_steps', 'sigma_min', 'sigma_max', 'sigma_data', 'rho', 'P_mean', 'P_std', 'S_churn', 'S_tmin',
                  'S_tmax', 'S_noise']
Hparams = namedtuple('Hparams', Hparams_fields)


def calc_all_frame_dims(downsample_factors, frames):
    if not exists(frames):
        return (tuple(),) * len(downsample_factors)
    out = []
    for divisor in downsample_factors:
        assert frames % divisor == 0
        out.append((frames // divisor,))
    return out


class ElucidatedImagen(nn.Module):
    def __init__(
        self, unets, *, image_sizes, text_encoder_name=None, text_embed_dim=None, channels=3, cond_drop_prob=0.1,
        random_crop_sizes=None, temporal_downsample_factor=1, lowres_sample_noise_level=0.2,
        per_sample_random_aug_noise_level=False, condition_on_text=True, auto_normalize_img=True, dynamic_thresholding=True,
        dynamic_thresholding_percentile=0.95, only_train_unet_number=None, lowres_noise_schedule='linear',
        num_sample_steps=32, sigma_min=0.002, sigma_max=80, sigma_data=0.5, rho=7, P_mean=-1.2, P_std=1.2, S_churn=80,
        S_tmin=0.05, S_tmax=50, S_noise=1.003,
    ):
        super().__init__()
        if condition_on_text:
            raise NotImplementedError('text conditioning (T5) is not part of the IQT hot path: pass condition_on_text=False')
        self.only_train_unet_number = only_train_unet_number
        self.condition_on_text = False
        self.unconditional = True
        self.channels = channels
        unets = cast_tuple(unets)
        num_unets = len(unets)
        self.random_crop_sizes = cast_tuple(random_crop_sizes, num_unets)
        assert all(r is None for r in self.random_crop_sizes), 'random crops (kornia) are outside the IQT path'
        self.lowres_noise_schedule = GaussianDiffusionContinuousTimes(noise_schedule=lowres_noise_schedule)
        self.text_embed_dim = None

        self.unets = nn.ModuleList([])
        self.unet_being_trained_index = -1
        for ind, one_unet in enumerate(unets):
            assert isinstance(one_unet, (Unet, Unet3D, NullUnet))
            one_unet = one_unet.cast_model_parameters(lowres_cond=not ind == 0, cond_on_text=False, text_embed_dim=None,
                                                      channels=self.channels, channels_out=self.channels)
            self.unets.append(one_unet)
        self.is_video = any(isinstance(u, Unet3D) for u in self.unets) or any(isinstance(u, Unet) for u in self.unets)
        self.image_sizes = cast_tuple(image_sizes)
        assert num_unets == len(self.image_sizes), \
            f'you did not supply the correct number of u-nets ({len(self.unets)}) for resolutions {self.image_sizes}'
        self.sample_channels = cast_tuple(self.channels, num_unets)
        lowres_conditions = tuple(map(lambda t: t.lowres_cond, self.unets))
        assert lowres_conditions == (False, *((True,) * (num_unets - 1))), \
            'the first unet must be unconditioned (by low resolution image), and the rest of the unets must have `lowres_cond` set to True'
        self.lowres_sample_noise_level = lowres_sample_noise_level
        self.per_sample_random_aug_noise_level = per_sample_random_aug_noise_level
        self.cond_drop_prob = cond_drop_prob
        self.can_classifier_guidance = cond_drop_prob > 0.
        self.normalize_img = normalize_neg_one_to_one if auto_normalize_img else identity
        self.unnormalize_img = unnormalize_zero_to_one if auto_normalize_img else identity
        self.input_image_range = (0. if auto_normalize_img else -1., 1.)
        self.dynamic_thresholding = cast_tuple(dynamic_thresholding, num_unets)
        self.dynamic_thresholding_percentile = dynamic_thresholding_percentile
        temporal_downsample_factor = cast_tuple(temporal_downsample_factor, num_unets)
        self.temporal_downsample_factor = temporal_downsample_factor
        assert temporal_downsample_factor[-1] == 1, 'downsample factor of last stage must be 1'
        assert all(l >= r for l, r in zip((1, *temporal_downsample_factor[:-1]), temporal_downsample_factor[1:])), \
            'temporal downssample factor must be in order of descending'
        hparams = [num_sample_steps, sigma_min, sigma_max, sigma_data, rho, P_mean, P_std, S_churn, S_tmin, S_tmax, S_noise]
        hparams = [cast_tuple(hp, num_unets) for hp in hparams]
        self.hparams = [Hparams(*unet_hp) for unet_hp in zip(*hparams)]
        self._graphs = GraphCache()           # hipGraph replay of the U-Net evaluations of a sampling loop (graphs.py)
        self.register_buffer('_temp', torch.tensor([0.]), persistent=False)
        self.to(next(self.unets.parameters()).device)

    @property
    def device(self):
        return self._temp.device

    def get_unet(self, unet_number):
        assert 0 < unet_number <= len(self.unets)
        index = unet_number - 1
        if isinstance(self.unets, nn.ModuleList):
            unets_list = [unet for unet in self.unets]
            delattr(self, 'unets')
            self.unets = unets_list
        if index != self.unet_being_trained_index:
            for unet_index, unet in enumerate(self.unets):
                unet.to(self.device if unet_index == index else 'cpu')
        self.unet_being_trained_index = index
        return self.unets[index]

    def reset_unets_all_one_device(self, device=None):
        device = default(device, self.device)
        self.unets = nn.ModuleList([*self.unets])
        self.unets.to(device)
        self.unet_being_trained_index = -1

    def state_dict(self, *args, **kwargs):
        self.reset_unets_all_one_device()
        return super().state_dict(*args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self.reset_unets_all_one_device()
        return super().load_state_dict(*args, **kwargs)

    # ---- EDM scalars (host floats / [B] tensors) -----------------------------------------------------
    def c_skip(self, sigma_data, sigma):
        return (sigma_data ** 2) / (sigma ** 2 + sigma_data ** 2)

    def c_out(self, sigma_data, sigma):
        return sigma * sigma_data * (sigma_data ** 2 + sigma ** 2) ** -0.5

    def c_in(self, sigma_data, sigma):
        return 1 * (sigma ** 2 + sigma_data ** 2) ** -0.5

    def c_noise(self, sigma):
        return torch.log(sigma.clamp(min=1e-20)) * 0.25

    def loss_weight(self, sigma_data, sigma):
        return (sigma ** 2 + sigma_data ** 2) * (sigma * sigma_data) ** -2

    def noise_distribution(self, P_mean, P_std, batch_size):
        return (P_mean + P_std * torch.randn((batch_size,))).exp()          # host: one scalar per sample

    def sample_schedule(self, num_sample_steps, rho, sigma_min, sigma_max):
        N, inv_rho = num_sample_steps, 1 / rho
        steps = torch.arange(num_sample_steps, dtype=torch.float32)
        sigmas = (sigma_max ** inv_rho + steps / (N - 1) * (sigma_min ** inv_rho - sigma_max ** inv_rho)) ** rho
        return torch.nn.functional.pad(sigmas, (0, 1), value=0.)

    def threshold_x_start(self, x_start, dynamic_threshold=True):
        """elucidated_imagen.py:298-311: clamp(-1, 1), or per-sample s = max(quantile(|x0|, p), 1) then clamp(-s, s) / s."""
        x_start = x_start.contiguous()
        if dynamic_threshold:
            s = ops.abs_quantile(x_start, self.dynamic_thresholding_percentile)
            s.clamp_(min=1.)
            return ops.dynamic_threshold(x_start, s)
        one = torch.ones(x_start.shape[0], device=x_start.device)
        return ops.axpby3(x_start, None, None, one, None, None, -1., 1., 2)

    def _unet_kwargs(self, unet, lowres_cond_img, lowres_noise_times):
        inner = unet.module if hasattr(unet, 'module') else unet
        kw = dict(lowres_cond_img=lowres_cond_img)
        if isinstance(inner, Unet3D):
            kw['lowres_noise_times'] = lowres_noise_times
        return kw

    def preconditioned_network_forward(self, unet_forward, noised_images, sigma, *, sigma_data, clamp=False,
                                       dynamic_threshold=True, _replay=True, **kwargs):
        """:329-358 with per-batch sigma as a host float or a CPU/GPU [B] tensor.  ``_replay=False``: never through the hipGraph cache
        (the no-grad self-conditioning pre-pass of a TRAINING step: its weights change every optimiser step, a capture would be retired at once)."""
        B = noised_images.shape[0]
        dev = noised_images.device
        sig = torch.full((B,), float(sigma)) if isinstance(sigma, float) else sigma.detach().float().cpu()
        cin, cskip, cout = (f(sigma_data, sig).to(dev) for f in (self.c_in, self.c_skip, self.c_out))
        x_in = ops.axpby3(noised_images.contiguous(), None, None, cin, None, None)
        owner = getattr(unet_forward, '__self__', None)
        if _replay and owner is not None and not torch.is_grad_enabled():
            # sampling: after two eager calls the U-Net evaluation replays as a hipGraph (the small stages of a cascade are launch-bound)
            net_out = self._graphs.run(owner, unet_forward, (x_in, self.c_noise(sig).to(dev)), kwargs)
        else:
            net_out = unet_forward(x_in, self.c_noise(sig).to(dev), **kwargs)
        if clamp and dynamic_threshold:
            out = ops.axpby3(noised_images.contiguous(), net_out.contiguous(), None, cskip, cout, None, 0., 0., 0)
            return self.threshold_x_start(out, True)
        return ops.axpby3(noised_images.contiguous(), net_out.contiguous(), None, cskip, cout, None, -1., 1., 2 if clamp else 0)

    @torch.no_grad()
    def one_unet_sample(self, unet, shape, *, unet_number, clamp=True, dynamic_threshold=True, cond_scale=1., use_tqdm=True,
                        inpaint_images=None, inpaint_masks=None, inpaint_resample_times=5, init_images=None,
                        skip_steps=None, sigma_min=None, sigma_max=None, noise=None, **kwargs):
        """Stochastic Heun sampler (:382-532).  ``noise``: optional injected list [init, step_0, ...]."""
        assert not exists(inpaint_images) and not exists(inpaint_masks), 'inpainting: SURVEY.md §8(f) next'
        hp = self.hparams[unet_number - 1]
        sigma_min, sigma_max = default(sigma_min, hp.sigma_min), default(sigma_max, hp.sigma_max)
        sigmas = self.sample_schedule(hp.num_sample_steps, hp.rho, sigma_min, sigma_max)
        gammas = torch.where((sigmas >= hp.S_tmin) & (sigmas <= hp.S_tmax), min(hp.S_churn / hp.num_sample_steps, sqrt(2) - 1), 0.)
        sched = list(zip(sigmas[:-1].tolist(), sigmas[1:].tolist(), gammas[:-1].tolist()))[default(skip_steps, 0):]
        dev = self.device
        B = shape[0]
        noise = list(noise) if exists(noise) else None
        draw = (lambda: noise.pop(0).to(dev).float().contiguous()) if exists(noise) else (lambda: torch.randn(shape, device=dev))
        vec = lambda v: torch.full((B,), float(v), device=dev)
        images = ops.axpby3(draw(), None, None, vec(sigmas[0].item()), None, None)
        if exists(init_images):
            images = ops.add(images, init_images.to(dev).float())
        fwd = partial(self.preconditioned_network_forward, unet.forward_with_cond_scale, sigma_data=hp.sigma_data, clamp=clamp,
                      dynamic_threshold=dynamic_threshold, cond_scale=cond_scale, **kwargs)
        self_cond_on = bool(getattr(unet, 'self_cond', False))                   # self-conditioning (:483, 505, 524): the last x0 estimate
        sc = (lambda x0: dict(self_cond=x0)) if self_cond_on else (lambda x0: {})
        x_start = None
        for sigma, sigma_next, gamma in sched:
            eps = draw()
            sigma_hat = sigma + gamma * sigma
            images_hat = ops.axpby3(images, eps, None, vec(1.), vec(hp.S_noise * sqrt(max(sigma_hat ** 2 - sigma ** 2, 0.))), None)
            out = fwd(images_hat, float(sigma_hat), **sc(x_start))
            r = (sigma_next - sigma_hat) / sigma_hat
            # x_next = x_hat + (s_next - s_hat) * (x_hat - D)/s_hat
            images_next = ops.axpby3(images_hat, out, None, vec(1. + r), vec(-r), None)
            x_start = out
            if sigma_next != 0:                                                   # 2nd-order correction (:502-516)
                out2 = fwd(images_next, float(sigma_next), **sc(out))
                r2 = 0.5 * (sigma_next - sigma_hat) / sigma_next
                tmp = ops.axpby3(images_hat, out, images_next, vec(1. + 0.5 * r), vec(-0.5 * r), vec(r2))
                images_next = ops.axpby3(tmp, out2, None, vec(1.), vec(-r2), None)
                x_start = out2
            images = images_next
        images = ops.axpby3(images, None, None, vec(1.), None, None, -1., 1., 2)    # clamp(-1, 1)   (:527)
        return self.unnormalize_img(images)

    def _resize(self, x, size, frames=None):
        """resize_video_to (imagen_video.py:137-158): nearest, no-op when the spatial size already matches."""
        if x.shape[-1] == size:
            return x
        f = default(frames, x.shape[2])
        return to_channels_first(ops.nearest_resize(to_channels_last(x.float().to(self.device)), (f, size, size)))

    def _noise_lowres(self, lowres, times_cpu, noise):
        log_snr = self.lowres_noise_schedule.log_snr(times_cpu)
        alpha, sigma = log_snr_to_alpha_sigma(log_snr)
        dev = lowres.device
        return ops.q_sample(lowres.contiguous(), noise.contiguous(), alpha.to(dev), sigma.to(dev))

    @torch.no_grad()
    @eval_decorator
    def sample(self, texts=None, text_masks=None, text_embeds=None, cond_images=None, inpaint_images=None, inpaint_masks=None,
               inpaint_resample_times=5, init_images=None, skip_steps=None, sigma_min=None, sigma_max=None, video_frames=None,
               batch_size=1, cond_scale=1., lowres_sample_noise_level=None, start_at_unet_number=1, start_image_or_video=None,
               stop_at_unet_number=None, return_all_unet_outputs=False, return_pil_images=False, use_tqdm=True, device=None,
               noise=None):
        """:536-702.  ``noise``: optional injected list [lowres_noise, init, step_0, ...] per sampled unet (tests)."""
        assert texts is None and text_embeds is None and not return_pil_images
        device = default(device, self.device)
        self.reset_unets_all_one_device(device=device)
        lowres_sample_noise_level = default(lowres_sample_noise_level, self.lowres_sample_noise_level)
        num_unets = len(self.unets)
        cond_scale = cast_tuple(cond_scale, num_unets)
        assert exists(video_frames), 'video_frames (the depth of the 3-D patch) must be passed in on sample time'
        all_frame_dims = calc_all_frame_dims(self.temporal_downsample_factor, video_frames)
        init_images = [maybe(self.normalize_img)(i) for i in cast_tuple(init_images, num_unets)]
        skip_steps, sigma_min, sigma_max = (cast_tuple(v, num_unets) for v in (skip_steps, sigma_min, sigma_max))
        noise = list(noise) if exists(noise) else None
        if start_at_unet_number > 1:
            assert start_at_unet_number <= num_unets, 'must start a unet that is less than the total number of unets'
            assert not exists(stop_at_unet_number) or start_at_unet_number <= stop_at_unet_number
            assert exists(start_image_or_video), 'starting image or video must be supplied if only doing upscaling'
            img = self._resize(start_image_or_video.to(device), self.image_sizes[start_at_unet_number - 2])
        outputs = []
        for unet_number, unet, image_size, frame_dims, dynamic_threshold, unet_cond_scale, unet_init, unet_skip, smin, smax in zip(
                range(1, num_unets + 1), self.unets, self.image_sizes, all_frame_dims, self.dynamic_thresholding, cond_scale,
                init_images, skip_steps, sigma_min, sigma_max):
            if unet_number < start_at_unet_number:
                continue
            assert not isinstance(unet, NullUnet), 'cannot sample from null unet'
            lowres_cond_img = lowres_noise_times = None
            if unet.lowres_cond:
                t_cpu = torch.full((batch_size,), float(lowres_sample_noise_level))
                lowres_noise_times = t_cpu.to(device)                              # the RAW time at sampling (:652, 680)
                lowres_cond_img = self.normalize_img(self._resize(img, image_size, frame_dims[0])).float().to(device)
                ln = noise.pop(0).to(device).float() if exists(noise) else torch.randn_like(lowres_cond_img)
                lowres_cond_img = self._noise_lowres(lowres_cond_img, t_cpu, ln)
            if exists(unet_init):
                unet_init = self._resize(unet_init, image_size, frame_dims[0])
            shape = (batch_size, self.channels, *frame_dims, image_size, image_size)
            n_draws = len(list(zip(range(self.hparams[unet_number - 1].num_sample_steps)))) - default(unet_skip, 0) + 1
            unet_noise = [noise.pop(0) for _ in range(n_draws)] if exists(noise) else None
            img = self.one_unet_sample(unet, shape, unet_number=unet_number, init_images=unet_init, skip_steps=unet_skip,
                                       sigma_min=smin, sigma_max=smax, cond_scale=unet_cond_scale, dynamic_threshold=dynamic_threshold,
                                       use_tqdm=use_tqdm, noise=unet_noise,
                                       **({'cond_images': cond_images.to(device).float()} if exists(cond_images) else {}),
                                       **self._unet_kwargs(unet, lowres_cond_img, lowres_noise_times))
            outputs.append(img)
            if exists(stop_at_unet_number) and stop_at_unet_number == unet_number:
                break
        return outputs[-1] if not return_all_unet_outputs else outputs

    # ---- training ------------------------------------------------------------------------------------
    def forward(self, images, unet=None, texts=None, text_embeds=None, text_masks=None, unet_number=None, cond_images=None,
                noise=None, sigmas=None, lowres_aug_times=None, lowres_noise=None, lowres_img=None, **kwargs):
        """:712-882 -> scalar loss.  ``noise`` / ``sigmas`` / ``lowres_aug_times`` / ``lowres_noise`` are optional injection
        hooks for parity tests (the reference draws them internally)."""
        assert texts is None and text_embeds is None
        assert images.shape[-1] == images.shape[-2], \
            f'the images you pass in must be a square, but received dimensions of {images.shape[2]}, {images.shape[-1]}'
        assert not (len(self.unets) > 1 and not exists(unet_number)), \
            f'you must specify which unet you want trained, from a range of 1 to {len(self.unets)}, if you are training cascading DDPM (multiple unets)'
        unet_number = default(unet_number, 1)
        assert not exists(self.only_train_unet_number) or self.only_train_unet_number == unet_number
        assert images.dtype == torch.float, f'images tensor needs to be floats but {images.dtype} dtype found instead'
        unet_index = unet_number - 1
        unet = default(unet, lambda: self.get_unet(unet_number))
        inner = unet.module if hasattr(unet, 'module') else unet
        assert not isinstance(inner, NullUnet), 'null unet cannot and should not be trained'
        target_image_size = self.image_sizes[unet_index]
        prev_image_size = self.image_sizes[unet_index - 1] if unet_index > 0 else None
        hp = self.hparams[unet_index]
        B, c, frames, h, w = images.shape
        device = images.device
        assert c == self.channels and h >= target_image_size and w >= target_image_size
        all_frame_dims = tuple(fd[0] for fd in calc_all_frame_dims(self.temporal_downsample_factor, frames))
        target_frames = all_frame_dims[unet_index]
        prev_frames = all_frame_dims[unet_index - 1] if unet_index > 0 else None

        lowres_cond_img = None
        if exists(prev_image_size):
            # the reference derives the conditioning by down/up-sampling `images` (:781-782); IQT supplies a real
            # low-quality patch through `lowres_img` (superset: ImagenTrainer's dataloader keyword)
            lowres_cond_img = self._resize(lowres_img.to(device), target_image_size, target_frames) if exists(lowres_img) else \
                self._resize(self._resize(images, prev_image_size, prev_frames), target_image_size, target_frames)
            if not exists(lowres_aug_times):
                if self.per_sample_random_aug_noise_level:
                    lowres_aug_times = self.lowres_noise_schedule.sample_random_times(B, device='cpu')
                else:
                    lowres_aug_times = self.lowres_noise_schedule.sample_random_times(1, device='cpu').repeat(B)
            lowres_aug_times = lowres_aug_times.detach().cpu().float()
        images = self.normalize_img(self._resize(images, target_image_size, target_frames)).float().contiguous()
        lowres_noise_cond = None
        if exists(lowres_cond_img):
            lowres_cond_img = self.normalize_img(lowres_cond_img).float().contiguous()
            ln = default(lowres_noise, lambda: torch.randn_like(lowres_cond_img)).to(device)
            lowres_cond_img = self._noise_lowres(lowres_cond_img, lowres_aug_times, ln)
            lowres_noise_cond = self.lowres_noise_schedule.get_condition(lowres_aug_times).to(device)   # log-SNR at training (:838)

        sig = default(sigmas, lambda: self.noise_distribution(hp.P_mean, hp.P_std, B)).detach().float().cpu()
        noise = default(noise, lambda: torch.randn_like(images)).to(device).contiguous()
        one = torch.ones(B, device=device)
        noised = ops.axpby3(images, noise, None, one, sig.to(device), None)              # alphas are 1 in EDM (:829)

        # denoised = c_skip*x + c_out*F(c_in*x, c_noise); weighted MSE folded into ONE kernel on the raw network output:
        #   w*(c_skip x + c_out F - y)^2 = (w c_out^2) * (F - (y - c_skip x)/c_out)^2
        cin, cskip, cout = (f(hp.sigma_data, sig) for f in (self.c_in, self.c_skip, self.c_out))
        x_in = ops.axpby3(noised, None, None, cin.to(device), None, None)
        ukw = {**self._unet_kwargs(unet, lowres_cond_img, lowres_noise_cond), **kwargs}
        if exists(cond_images):                                                   # image conditioning of the U-Net (:718, 844)
            ukw['cond_images'] = cond_images.to(device).float()
        # self-conditioning (:847-860): half of the steps first estimate x0 without gradients and feed it back.  The reference's gate
        # `self_cond = unet.module.self_cond if DDP else unet` (:849) is always truthy for an un-wrapped U-Net, so it consumes one
        # `random()` per call whether or not the U-Net self-conditions (a U-Net without the option ignores the estimate): the draw is
        # mirrored so Python's RNG stream matches, the wasted pre-pass is not
        draw = random() if (not hasattr(unet, 'module') or getattr(inner, 'self_cond', False)) else 1.
        if getattr(inner, 'self_cond', False) and draw < 0.5:
            with torch.no_grad():
                pred_x0 = self.preconditioned_network_forward(unet.forward, noised, sig, sigma_data=hp.sigma_data, _replay=False,
                                                              **ukw).detach()
            ukw['self_cond'] = pred_x0
        net_out = unet.forward(x_in, self.c_noise(sig).to(device), **ukw)
        target = ops.axpby3(images, noised, None, (1. / cout).to(device), (-cskip / cout).to(device), None)
        weight = (self.loss_weight(hp.sigma_data, sig) * cout ** 2).to(device)
        loss, _ = ops.mse_clamp(net_out, target, do_clamp=False, weight=weight)
        return loss
