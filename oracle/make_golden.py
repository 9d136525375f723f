"""TEST INFRASTRUCTURE (container-only): generate tests/golden/*.npz by running the REAL reference.

Run:  python oracle/make_golden.py            (needs /root/reference; CPU only; ~1 min)

Only numbers (inputs, injected noise, outputs, a few gradients) are written — never reference
source.  The weights are not stored: both sides rebuild them with ``hash_fill_state_dict`` from
the parameter names, and each fixture records the key list + shapes so the name contract
 is pinned too.
"""
import os
import sys
import json
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402
from iqt_oracle import hash_fill_state_dict  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
os.makedirs(OUT, exist_ok=True)

MIN_BOUND = (0.0 - 271.64814106698583) / 377.117173547721        # train.py:72 with config.yaml:12-13


def base_configs(batch_sample=False, norm='z-score'):
    return {'Data': {'norm': norm, 'mean': 271.64814106698583, 'std': 377.117173547721},
            'Train': {'batch_sample': batch_sample, 'patch_size_sub': 8, 'batch_sample_factor': 3,
                      'pred_obj': 'x_start', 'timesteps': 4, 'dynamic_threshold': False},
            'Eval': {'repeat': 1, 'batch_size': 27}}


def unet_kwargs_train_py(dim, img_size, **over):
    """train.py:83-116 with config.yaml values (attention off, deep_feature False, use_se='True,')."""
    kw = dict(img_size=img_size, dim=dim, dim_mults=(1, 2, 4), channels=1, num_resnet_blocks=(2, 2, 2),
              init_conv_kernel_size=3, lowres_cond=True, init_cross_embed=False,
              init_cross_embed_kernel_sizes=(3, 5, 7), att_type='linear', attn_dim_head=64,
              attend_at_middle=False, attend_at_middle_depth=1, attend_at_middle_heads=8,
              attend_at_enc=[False, False, False], attend_at_enc_depth=[1, 1, 1], attend_at_enc_heads=[8, 8, 8],
              att_drop=0.0, att_forward_drop=0.0, att_forward_expansion=2, att_skip_scale=False,
              att_localvit=False, groups=1, emb_size=256, init_dim=dim, memory_efficient=False,
              use_se_attn='True,', pixel_shuffle_upsample=True, boundary=False, batch_sample=False,
              batch_sample_factor=3, deep_feature=False)
    kw.update(over)
    return kw


def fill(module, seed=0):
    sd = hash_fill_state_dict(module.state_dict(), seed)
    module.load_state_dict(sd)
    return sd


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
                                 for k, v in arrays.items()})
    print("wrote", path, f"{os.path.getsize(path) / 1024:.1f} KiB")


GRAD_KEYS = ['init_conv.weight', 'init_conv.bias', 'to_time_hiddens.0.weights', 'to_time_cond.0.weight',
             'downs.0.1.block1.project.weight', 'downs.0.1.block2.groupnorm.weight',
             'downs.0.1.time_mlp.1.weight', 'downs.0.1.se.fc.0.weight', 'downs.0.4.1.weight',
             'downs.2.3.1.block2.project.weight', 'downs.2.4.weight', 'ups.0.0.net.0.weight',
             'ups.0.1.res_conv.weight', 'ups.1.1.block1.project.weight', 'final_res_block.block2.project.bias',
             'final_conv.weight', 'final_conv.bias']


def gen_unet_family_a(r3):
    g = torch.Generator().manual_seed(1234)
    B, S, dim = 2, 8, 16
    kw = unet_kwargs_train_py(dim, S)
    unet = r3.SRUnet256(**kw)
    sd = fill(unet)
    x = torch.randn(B, 1, S, S, S, generator=g)
    lr = torch.randn(B, 1, S, S, S, generator=g)
    times = torch.rand(B, generator=g)
    log_snr = r3.alpha_cosine_log_snr(times)
    unet.eval()
    with torch.no_grad():
        y = unet(x, times, log_snr, lowres_cond_img=lr)

    # Imagen.forward (training loss) with injected times and noise
    imagen = r3.Imagen(unets=(r3.NullUnet(), unet), configs=base_configs(), min_bound=MIN_BOUND,
                       image_sizes=(S, S), channels=1, pred_objectives='x_start', timesteps=4,
                       dynamic_thresholding=False, p2_loss_weight_gamma=0.0, auto_normalize_img=False,
                       cond_drop_prob=0.0, lpips=False, medlpips=False, boundary=False)
    unet2 = imagen.unets[1]
    hr = torch.randn(B, 1, S, S, S, generator=g)
    noise = torch.randn(B, 1, S, S, S, generator=g)
    # make a visible share of predictions fall under min_bound so the clamp's zero-gradient matters
    imagen.noise_schedulers[1].sample_random_times = lambda b, device: times.clone()
    unet2.train()
    loss, pred, x_noisy, lowres = imagen(hr, lowres_img=lr, unet_number=2, noise=noise)
    loss.backward()
    named = dict(unet2.named_parameters())
    grads = {('grad:' + k): named[k].grad for k in GRAD_KEYS}
    unused = [k for k, p in named.items() if p.grad is None]
    save("unetA_tiny", x=x, lowres=lr, times=times, log_snr=log_snr, y=y, hr=hr, noise=noise,
         loss=loss.detach(), pred=pred.detach(), x_noisy=x_noisy.detach(), min_bound=MIN_BOUND,
         keys=np.array(list(sd.keys())), shapes=np.array([json.dumps(list(v.shape)) for v in sd.values()]),
         unused=np.array(unused), kwargs=json.dumps({k: v for k, v in kw.items()}), **grads)

    # 4-step DDPM trajectory with injected noise (draw order: randn(shape) then one randn_like per step)
    T = 4
    init_noise = torch.randn(B, 1, S, S, S, generator=g)
    step_noise = [torch.randn(B, 1, S, S, S, generator=g) for _ in range(T)]
    queue = [init_noise] + step_noise
    orig_randn, orig_randn_like = torch.randn, torch.randn_like
    torch.randn = lambda *a, **k: queue.pop(0).clone()
    torch.randn_like = lambda *a, **k: queue.pop(0).clone()
    try:
        img, noisy_l, x0_l = imagen.sample(batch_size=B, start_image_or_video=lr, start_at_unet_number=2,
                                           use_tqdm=False)
    finally:
        torch.randn, torch.randn_like = orig_randn, orig_randn_like
    assert len(queue) == 0
    save("ddpmA_traj", lowres=lr, init_noise=init_noise, step_noise=torch.stack(step_noise), img=img,
         noisy=np.stack(noisy_l), x0=np.stack(x0_l), T=T, min_bound=MIN_BOUND)


def gen_unet_attention(r3):
    """C4-shaped tiny net: linear attention at every level + middle, deep_feature, one volume per batch."""
    g = torch.Generator().manual_seed(4321)
    for kind in ('linear', 'softmax'):
        S, dim = 16, 16
        kw = unet_kwargs_train_py(dim, S, att_type=kind, attend_at_middle=True, attend_at_enc=[True, True, True],
                                  attn_dim_head=8, attend_at_enc_heads=[2, 2, 2], attend_at_middle_heads=2,
                                  deep_feature=True, batch_sample=True, batch_sample_factor=1)
        unet = r3.SRUnet256(**kw)
        sd = fill(unet, seed=1)
        unet.eval()      # Dropout(0.05) inside to_q/k/v is the identity in eval mode
        x = torch.randn(1, 1, S, S, S, generator=g)
        lr = torch.randn(1, 1, S, S, S, generator=g)
        times = torch.rand(1, generator=g)
        log_snr = r3.alpha_cosine_log_snr(times)
        for p in unet.parameters():
            p.requires_grad_(True)
        y = unet(x, times, log_snr, lowres_cond_img=lr)
        (y ** 2).mean().backward()
        named = dict(unet.named_parameters())
        gk = ['downs.0.2.layers.0.0.to_q.1.weight', 'downs.0.2.layers.0.0.patch_embed.projection.depthwise.weight',
              'downs.1.2.layers.0.0.reconstruct.1.pointwise.weight', 'mid_attn.layers.0.1.1.weight',
              'mid_attn.layers.0.0.to_out.1.g', 'init_conv.weight']
        grads = {('grad:' + k): named[k].grad for k in gk}
        save(f"unetA_attn_{kind}", x=x, lowres=lr, times=times, log_snr=log_snr, y=y.detach(),
             keys=np.array(list(sd.keys())), shapes=np.array([json.dumps(list(v.shape)) for v in sd.values()]),
             kwargs=json.dumps(kw), **grads)


def gen_variants(r3):
    """memory_efficient=True (SRUnet256 default) + init_cross_embed + boundary mode (27 sub-volumes)."""
    g = torch.Generator().manual_seed(99)
    S, dim = 8, 16
    kw = unet_kwargs_train_py(dim, S, memory_efficient=True, init_cross_embed=True,
                              init_cross_embed_kernel_sizes=(3, 5, 7), deep_feature=True)
    unet = r3.SRUnet256(**kw)
    sd = fill(unet, seed=2)
    unet.eval()
    x = torch.randn(2, 1, S, S, S, generator=g)
    lr = torch.randn(2, 1, S, S, S, generator=g)
    times = torch.rand(2, generator=g)
    with torch.no_grad():
        y = unet(x, times, r3.alpha_cosine_log_snr(times), lowres_cond_img=lr)
    save("unetA_memeff", x=x, lowres=lr, times=times, y=y, keys=np.array(list(sd.keys())),
         shapes=np.array([json.dumps(list(v.shape)) for v in sd.values()]), kwargs=json.dumps(kw))

    S = 4
    kw = unet_kwargs_train_py(dim, S * 3, dim_mults=(1, 2), num_resnet_blocks=(1, 1), attend_at_enc=[False, False],
                              attend_at_enc_depth=[1, 1], attend_at_enc_heads=[8, 8], boundary=True,
                              batch_sample=True, batch_sample_factor=3)
    unet = r3.SRUnet256(**kw)
    sd = fill(unet, seed=3)
    unet.eval()
    x = torch.randn(27, 1, S, S, S, generator=g)
    lr = torch.randn(27, 1, S, S, S, generator=g)
    times = torch.rand(1, generator=g).repeat(27)
    with torch.no_grad():
        y = unet(x, times, r3.alpha_cosine_log_snr(times), lowres_cond_img=lr)
    save("unetA_boundary", x=x, lowres=lr, times=times, y=y, keys=np.array(list(sd.keys())),
         shapes=np.array([json.dumps(list(v.shape)) for v in sd.values()]), kwargs=json.dumps(kw))


def gen_schedules_and_subvolumes(r3):
    from utils_mine import convertVolume2subVolume, merge_sub_volumes
    t = torch.linspace(0, 1, 33)
    sched = r3.GaussianDiffusionContinuousTimes(noise_schedule='cosine', timesteps=32)
    xs = torch.linspace(-2, 2, 24).reshape(2, 1, 3, 2, 2)
    xt = torch.linspace(1.5, -1.0, 24).reshape(2, 1, 3, 2, 2)
    tt, tn = torch.tensor([0.75, 0.25]), torch.tensor([0.5, 0.0])
    mean, var, logvar = sched.q_posterior(xs, xt, tt, t_next=tn)
    vol = torch.arange(2 * 12 ** 3, dtype=torch.float32).reshape(1, 2, 12, 12, 12)
    sub = convertVolume2subVolume(vol, target_shape=(27, 2, 4, 4, 4))
    merged = merge_sub_volumes(sub, original_shape=(1, 2, 12, 12, 12))
    halo = r3.boundary_pad(sub, batch_sample_factor=3)
    save("schedulesA", t=t, cosine=r3.alpha_cosine_log_snr(t), linear=r3.beta_linear_log_snr(t),
         post_xs=xs, post_xt=xt, post_t=tt, post_tn=tn, post_mean=mean, post_var=var, post_logvar=logvar,
         vol=vol, sub=sub, merged=merged, halo=halo)


def gen_trainer_trace(r3, rt):
    """ImagenTrainer semantics trace (SURVEY.md Appendix A 'Trainer facts'): steps / Adam cadence / losses."""
    torch.manual_seed(0)
    np.random.seed(0)
    S, dim = 8, 16
    unet = r3.SRUnet256(**unet_kwargs_train_py(dim, S))
    fill(unet)
    cfgs = base_configs()
    imagen = r3.Imagen(unets=(r3.NullUnet(), unet), configs=cfgs, min_bound=MIN_BOUND, image_sizes=(S, S),
                       channels=1, pred_objectives='x_start', timesteps=4, dynamic_thresholding=False,
                       p2_loss_weight_gamma=0.0, auto_normalize_img=False, cond_drop_prob=0.0)
    trainer = rt.ImagenTrainer(configs=cfgs, imagen=imagen, gradient_accumulation_steps=4,
                               split_valid_from_train=False, verbose=False)
    g = torch.Generator().manual_seed(7)
    n_micro = 6
    hr = torch.randn(n_micro, 2, 1, S, S, S, generator=g)
    lr = torch.randn(n_micro, 2, 1, S, S, S, generator=g)
    times = torch.rand(n_micro, 2, generator=g)
    noise = torch.randn(n_micro, 2, 1, S, S, S, generator=g)
    u = trainer.imagen.unets[1]
    w0 = u.final_conv.weight.detach().clone()
    trace, losses, wsnap = [], [], []
    trainer.training = True
    for i in range(n_micro):
        trainer.imagen.noise_schedulers[1].sample_random_times = (lambda b, device, i=i: times[i].clone())
        loss, pred, x_noisy, _ = trainer.forward(hr[i], lowres_img=lr[i], unet_number=2, max_batch_size=2,
                                                 noise=noise[i])
        uu = trainer.imagen.unets[1]
        changed = not torch.equal(uu.final_conv.weight.detach(), w0)
        w0 = uu.final_conv.weight.detach().clone()
        trace.append((int(trainer.steps[1].item()), int(changed)))
        losses.append(loss)
        wsnap.append(uu.final_conv.weight.detach().clone().flatten())
    save("trainerA_trace", hr=hr, lowres=lr, times=times, noise=noise, trace=np.array(trace),
         losses=np.array(losses), final_conv_w=torch.stack(wsnap), min_bound=MIN_BOUND)


if __name__ == "__main__":
    r3, rv, re_, rt = ref_shim.import_reference()
    torch.set_num_threads(8)
    gen_schedules_and_subvolumes(r3)
    gen_unet_family_a(r3)
    gen_unet_attention(r3)
    gen_variants(r3)
    gen_trainer_trace(r3, rt)
