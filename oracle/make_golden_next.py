"""TEST INFRASTRUCTURE (container-only): fixtures for the sampler / loss options of SURVEY.md §8(f) item 4, produced by
running the REAL reference:  noise- and v-objectives, dynamic thresholding, skip_steps, the inpainting resample loop.

Run:  python oracle/make_golden_next.py        (needs /root/reference; CPU only)
Only numbers are written (inputs, injected noise in the reference's draw order, outputs)."""
import os
import sys
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402
from make_golden import MIN_BOUND, base_configs, unet_kwargs_train_py, fill, save  # noqa: E402


def run_sample(imagen, queue, **kw):
    orig_randn, orig_randn_like = torch.randn, torch.randn_like
    torch.randn = lambda *a, **k: queue.pop(0).clone()
    torch.randn_like = lambda *a, **k: queue.pop(0).clone()
    try:
        out = imagen.sample(use_tqdm=False, **kw)
    finally:
        torch.randn, torch.randn_like = orig_randn, orig_randn_like
    assert len(queue) == 0, f"{len(queue)} injected draws unused"
    return out


def main():
    r3, _, _, _ = ref_shim.import_reference()
    g = torch.Generator().manual_seed(4321)
    B, S, dim = 2, 8, 16
    kw = unet_kwargs_train_py(dim, S)
    lr = torch.randn(B, 1, S, S, S, generator=g)
    hr = torch.randn(B, 1, S, S, S, generator=g)
    times = torch.rand(B, generator=g)
    noise = torch.randn(B, 1, S, S, S, generator=g)
    out = dict(lowres=lr, hr=hr, times=times, noise=noise, min_bound=MIN_BOUND)

    def make(objective, dyn, norm='z-score', T=4):
        unet = r3.SRUnet256(**kw)
        fill(unet)
        im = r3.Imagen(unets=(r3.NullUnet(), unet), configs=base_configs(norm=norm), min_bound=MIN_BOUND, image_sizes=(S, S),
                       channels=1, pred_objectives=objective, timesteps=T, dynamic_thresholding=dyn,
                       dynamic_thresholding_percentile=0.9, p2_loss_weight_gamma=0.0, auto_normalize_img=False,
                       cond_drop_prob=0.0, lpips=False, medlpips=False, boundary=False)
        return im

    # ---- training losses with the noise / v objectives ----
    for obj in ('noise', 'v'):
        im = make(obj, False)
        im.noise_schedulers[1].sample_random_times = lambda b, device: times.clone()
        im.unets[1].train()
        loss, pred, x_noisy, _ = im(hr, lowres_img=lr, unet_number=2, noise=noise)
        loss.backward()
        named = dict(im.unets[1].named_parameters())
        out[f'loss_{obj}'] = loss.detach()
        out[f'pred_{obj}'] = pred.detach()
        out[f'grad_{obj}:final_conv.weight'] = named['final_conv.weight'].grad
        out[f'grad_{obj}:init_conv.weight'] = named['init_conv.weight'].grad

    # ---- sampler variants ----
    def traj(tag, objective, dyn, norm='z-score', T=4, n_draws=None, **skw):
        im = make(objective, dyn, norm=norm, T=T)
        im.unets[1].eval()
        draws = [torch.randn(B, 1, S, S, S, generator=g) for _ in range(n_draws)]
        img, noisy, x0 = run_sample(im, list(draws), batch_size=B, start_image_or_video=lr, start_at_unet_number=2, **skw)
        out[f'{tag}:draws'] = torch.stack(draws)
        out[f'{tag}:img'] = img
        out[f'{tag}:noisy'] = np.stack(noisy)
        out[f'{tag}:x0'] = np.stack(x0)

    traj('noise_dyn', 'noise', True, n_draws=1 + 4)                       # noise objective + dynamic threshold (p = 0.9)
    traj('v_static', 'v', False, n_draws=1 + 4)                           # v objective, static clamp_(min_bound)
    traj('x0_dyn_minmax', 'x_start', True, norm='min-max', n_draws=1 + 4)  # s.clamp_(min=1) branch
    traj('skip2', 'x_start', False, T=6, n_draws=1 + 4, skip_steps=2)     # 6 steps -> [0, 2, 4] + [5]
    # inpainting: the reference's re-noise branch calls a method that does not exist (`self.right_pad_dims_to_datatype`,
    # imagen_pytorch3D.py:2143), so it only runs with inpaint_resample_times = 1: per step (q_sample, p_sample) draws
    mask = torch.rand(B, 1, S, S, S, generator=g) > 0.5
    inp = torch.randn(B, 1, S, S, S, generator=g)
    out['inpaint:mask'] = mask
    out['inpaint:images'] = inp
    T = 3
    traj('inpaint', 'x_start', False, T=T, n_draws=1 + 2 * T, inpaint_images=inp, inpaint_masks=mask,
         inpaint_resample_times=1)
    save("ddpmA_options", **out)


if __name__ == "__main__":
    main()
