#!/usr/bin/env python3
"""bench.py — headline benchmark of the DiffusionIQT hot path on MI355X.

Workload (BASELINE.json configs[1], SURVEY.md §8 "C2"): Family-A SRUnet256(dim=64, dim_mults=(1,2,4),
2 resnet blocks/level, SE, no attention, deep_feature=False — the train.py kwargs) on 32^3 1-channel
patches, batch 8 per GPU, fp32.

  --mode sample (default): one "step" = one DDPM ancestral sampler step over the batch = one U-Net eval
                           (186.06 GFLOP/patch) + one fused posterior-step kernel.  value = patches denoised / s.
  --mode train           : one "step" = one ImagenTrainer micro-step (Imagen.forward + backward + grad
                           all-reduce on sync steps + fused Adam every 4th + EMA), 558 GFLOP/patch.
Default mode 'both' times sample steps as the headline value and reports the train rate beside it.

N > 1 (launched with torch.distributed.run, one rank per GPU): every rank holds its own batch of 8 patches
(weak scaling); sampling shards patches with no data-path collective, training all-reduces gradients over RCCL.

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (conv_fwd_kernel: MFMA f32 implicit-GEMM
conv), timed live with HIP events on the launch stream inside the timed region; `cpu_baseline` is the CPU oracle
(oracle/iqt_oracle.py, a port of the reference algorithm — the reference itself cannot travel to the GPU box)
timed on the host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

GFLOP_PER_PATCH_EVAL = 186.06      # SURVEY.md §8d (FlopCounterMode on the reference, 2*MAC)
PEAK_F32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak


def unet_kwargs(size):
    return dict(img_size=size, dim=64, dim_mults=(1, 2, 4), channels=1, num_resnet_blocks=(2, 2, 2),
                init_conv_kernel_size=3, lowres_cond=True, init_cross_embed=False, att_type='linear', attn_dim_head=64,
                attend_at_middle=False, attend_at_enc=[False] * 3, attend_at_enc_depth=[1] * 3,
                attend_at_enc_heads=[8] * 3, init_dim=64, memory_efficient=False, use_se_attn='True,',
                pixel_shuffle_upsample=True, boundary=False, batch_sample=False, batch_sample_factor=3, deep_feature=False)


def pmc_traffic(batch, size):
    """HBM bytes per average conv_fwd launch of one C2 sampler step, from the committed rocprofv3 --pmc passes
    (FETCH_SIZE and WRITE_SIZE collected separately; FETCH_SIZE doubled for gfx950, MI355X_MICROARCH.md).  The counters
    cannot be read from inside this process, so this is the profiled value of the same command, or None off-config."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_hbm_traffic.json")
    if batch != 8 or size != 32 or not os.path.exists(path):
        return None
    try:
        with open(path) as f:
            k = json.load(f)["kernels"]
        return next(v["hbm_bytes_per_launch"] for name, v in k.items() if "conv_fwd8_kernel" in name)
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", choices=["sample", "train", "both"], default="both")
    ap.add_argument("--batch", type=int, default=8, help="patches per GPU")
    ap.add_argument("--size", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--no-family-b", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline loops only (no EDM / autocast / Family-B lines): the command profiled under profiles/")
    args = ap.parse_args()

    from diffusioniqt_amd import distributed as D, ops, _lib
    from diffusioniqt_amd.imagen_pytorch3D import SRUnet256, Imagen, NullUnet
    from diffusioniqt_amd.trainer import ImagenTrainer
    import torch.distributed as dist

    world, rank, device = D.init_from_env()
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    assert world == args.gpus or world == 1, f"launched with WORLD_SIZE={world} but --gpus {args.gpus}"
    _lib.load()
    B, S, K, W = args.batch, args.size, args.steps, args.warmup

    torch.manual_seed(42)                                   # train.py:29 set_seed(42): reference init under seed 42
    unet = SRUnet256(**unet_kwargs(S))
    configs = {'Data': {'norm': 'z-score'}, 'Train': {'batch_sample': False, 'patch_size_sub': S, 'pred_obj': 'x_start'},
               'Eval': {'repeat': 1}}
    min_bound = (0. - 271.64814106698583) / 377.117173547721
    imagen = Imagen(unets=(NullUnet(), unet), configs=configs, min_bound=min_bound, image_sizes=(S, S), channels=1,
                    pred_objectives='x_start', timesteps=max(K + W, 2), dynamic_thresholding=False,
                    p2_loss_weight_gamma=0.0, cond_drop_prob=0.0).to(device)
    trainer = ImagenTrainer(configs=configs, imagen=imagen, gradient_accumulation_steps=4, verbose=False)
    trainer.prepare_for = 2
    trainer.validate_and_set_unet_being_trained(2)
    unet = imagen.unets[1]

    g = torch.Generator().manual_seed(42 + rank)            # data.py:259-261 style synthetic z-scored patches
    hr = torch.randn(B, 1, S, S, S, generator=g).to(device)
    lr = torch.randn(B, 1, S, S, S, generator=g).to(device)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, n_warm, n_steps):
        for _ in range(n_warm):
            fn()
        sync_all()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            fn()
        sync_all()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    # ---------------- sampling: K ancestral steps, inputs resident in HBM ----------------
    result = {}
    sched = imagen.noise_schedulers[1]
    sched.num_timesteps = K + W
    ts = list(sched.get_sampling_timesteps(B, device='cpu'))
    coefs = torch.stack([torch.stack(sched.posterior_coefficients(t, tn)) for t, tn in ts]).to(device)
    conds = torch.stack([sched.get_condition(t) for t, _ in ts]).to(device)
    state = {"img": torch.randn(B, 1, S, S, S, device=device), "i": 0}
    unet.eval()

    def sample_step():
        i = state["i"]
        with torch.no_grad():
            pred = unet(state["img"], None, conds[i], lowres_cond_img=lr)
            noise = torch.randn_like(pred)
            state["img"], _ = ops.ddpm_step(state["img"], pred, noise, coefs[i, 0], coefs[i, 1], coefs[i, 2],
                                            min_bound, 0.0, 0)
        state["i"] = i + 1

    roof = None
    if args.mode in ("sample", "both"):
        for _ in range(W):
            sample_step()
        sync_all()
        ops.TIMER.enabled = not args.no_kernel_timer
        ops.TIMER.reset()
        t0 = time.perf_counter()
        for _ in range(K):
            sample_step()
        sync_all()
        dt = time.perf_counter() - t0
        ops.TIMER.enabled = False
        if world > 1:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        result["sample"] = dict(ms_per_step=1e3 * dt / K, patches_per_s=world * B * K / dt)
        summ = ops.TIMER.summary()
        dom = max((k for k in ("conv_fwd8_kernel", "conv_fwd_kernel") if k in summ), key=lambda k: summ[k][0], default=None)
        if dom is not None:
            ms, flops, n = summ[dom]
            roof = dict(bound="mfma", kernel=dom, achieved=round(flops / (ms * 1e-3) / 1e12, 2),
                        peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s", frac=round(flops / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                        traffic=pmc_traffic(B, S), launches=n, avg_launch_ms=round(ms / n, 4),
                        share_of_step=round(ms / (1e3 * dt), 3),
                        flops_per_launch=round(flops / n / 1e9, 3), flops_unit="GFLOP (algorithmic, 2*MAC) per average launch")
            # every MFMA conv launch of the step together (conv_fwd8 + conv_fwd + conv1x1 + small-Cin), for comparison across rounds
            ams = sum(v[0] for k, v in summ.items() if k.startswith("conv"))
            afl = sum(v[1] for k, v in summ.items() if k.startswith("conv"))
            roof["all_conv_kernels"] = dict(achieved=round(afl / (ams * 1e-3) / 1e12, 2), frac=round(afl / (ams * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                                            share_of_step=round(ams / (1e3 * dt), 3))
        assert torch.isfinite(state["img"]).all()

    # ---------------- the same sampler step under torch.autocast(fp16): conv / linear forwards on the fp16 MFMA kernel (fp32
    #                  accumulate), everything else fp32 -- the reference's mixed-precision switch.  Reported
    #                  beside the fp32 headline, never as `value` ----
    if args.mode in ("sample", "both") and not args.no_extras:
        state["i"] = 0

        def sample_step_fp16():
            with torch.autocast('cuda', dtype=torch.float16):
                sample_step()

        ka = max(4, K // 2)
        dt = timed(sample_step_fp16, min(W, 2), min(ka, K + W - min(W, 2)))
        ka = min(ka, K + W - min(W, 2))
        result["autocast_fp16"] = dict(ms_per_step=1e3 * dt / ka, patches_per_s=world * B * ka / dt, steps=ka)

    # ---------------- EDM (Karras) stochastic Heun sampler on the same U-Net: one call of ElucidatedImagen.sample with
    #                  n = max(4, K // 2) steps = 2n - 1 U-Net evals (elucidated_imagen.py:382-532); reported beside the headline ----
    if args.mode in ("sample", "both") and not args.no_extras:
        from diffusioniqt_amd.elucidated_imagen import ElucidatedImagen
        n_edm = max(4, K // 2)
        elu = ElucidatedImagen(unets=(NullUnet(), unet), image_sizes=(S, S), channels=1, condition_on_text=False,
                               auto_normalize_img=False, cond_drop_prob=0.0, num_sample_steps=n_edm,
                               dynamic_thresholding=False).to(device)
        unet.eval()

        def edm_sample():
            elu.sample(batch_size=B, video_frames=S, start_image_or_video=lr.clamp(-1, 1), start_at_unet_number=2, use_tqdm=False)

        dt = timed(edm_sample, 1, 1)
        result["edm"] = dict(heun_steps=n_edm, unet_evals=2 * n_edm - 1, ms_per_heun_step=1e3 * dt / n_edm,
                             heun_steps_per_s=n_edm / dt, patch_steps_per_s=world * B * n_edm / dt,
                             patch_evals_per_s=world * B * (2 * n_edm - 1) / dt)

    # ---------------- Family B: pseudo-3D Unet3D, dim 64, mults (1,2,4), mid + last-level attention, 32^3,
    #                  same batch: one eval (sampling path: fused MQA attention) and one fwd+bwd; reported beside the headline ----
    if args.mode == "both" and not args.no_family_b and not args.no_extras:
        from diffusioniqt_amd.imagen_video import Unet3D
        torch.manual_seed(43)
        u3 = Unet3D(dim=64, dim_mults=(1, 2, 4), channels=1, cond_on_text=False, text_embed_dim=None, lowres_cond=True,
                    layer_attns=(False, False, True), layer_cross_attns=False, attend_at_middle=True, num_resnet_blocks=2,
                    attn_pool_text=False).to(device)
        tb = torch.randn(B, device=device) * 0.5
        ltb = torch.full((B,), 0.2, device=device)

        def u3_eval():
            with torch.no_grad():
                u3(hr, tb, lowres_cond_img=lr, lowres_noise_times=ltb)

        def u3_train():
            u3.zero_grad(set_to_none=True)
            u3(hr, tb, lowres_cond_img=lr, lowres_noise_times=ltb).square().mean().backward()

        u3.eval()
        dte = timed(u3_eval, 2, 4) / 4
        u3.train()
        dtt = timed(u3_train, 1, 2) / 2
        result["unet3d"] = dict(eval_ms=1e3 * dte, eval_patches_per_s=world * B / dte, fwd_bwd_ms=1e3 * dtt,
                                fwd_bwd_patches_per_s=world * B / dtt)
        del u3

    # ---------------- training: K micro-steps through ImagenTrainer.forward ----------------
    if args.mode in ("train", "both"):
        trainer.training = True
        unet.train()

        def train_step():
            trainer.forward(hr, lowres_img=lr, unet_number=2, max_batch_size=B)

        kt = K if args.mode == "train" else max(4, K // 2)
        ops.TIMER.reset()
        dt = timed(train_step, W, kt)
        result["train"] = dict(ms_per_step=1e3 * dt / kt, steps_per_s=kt / dt, patches_per_s=world * B * kt / dt,
                               steps=kt)
        if not args.no_kernel_timer:
            # roofline of the training step's own dominant kernel (weight gradients), HIP events around every launch of a few
            # extra micro-steps outside the timed region (the events would perturb it)
            ops.TIMER.reset()
            ops.TIMER.enabled = True
            for _ in range(4):
                train_step()
            sync_all()
            ops.TIMER.enabled = False
            summ_t = ops.TIMER.summary()
            if "conv_bwd_weight_kernel" in summ_t:
                ms, flops, n = summ_t["conv_bwd_weight_kernel"]
                result["train"]["bwd_weight_tflops"] = flops / (ms * 1e-3) / 1e12
                result["train"]["bwd_weight_frac_of_f32_mfma_peak"] = flops / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS
                result["train"]["bwd_weight_avg_launch_ms"] = ms / n
        if not args.no_extras:
            # the same micro-steps with ImagenTrainer's mixed-precision switch (precision='bf16', trainer.py:293-311): forward and
            # backward-data on the bf16 MFMA kernel, weight gradients / optimiser / master weights fp32.  Beside the fp32 line.
            trainer.mixed_precision = 'bf16'
            dtb = timed(train_step, 2, kt)
            trainer.mixed_precision = 'no'
            result["train_bf16"] = dict(ms_per_step=1e3 * dtb / kt, patches_per_s=world * B * kt / dtb, steps=kt)

    # ---------------- CPU baseline: the oracle on the host cores, bounded sample ----------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import iqt_oracle as O
        ncores = os.cpu_count() or 1
        sd = {k: v.detach().cpu() for k, v in unet.state_dict().items()}
        cfg = O.unet_config(**unet_kwargs(S))
        nb = 2
        x_c, lr_c = torch.randn(nb, 1, S, S, S), torch.randn(nb, 1, S, S, S)
        t_c = torch.full((nb,), 0.5)
        with torch.no_grad():
            # torch-CPU convs do not scale to every hardware thread of the host: probe a few thread counts on one
            # eval each and keep the fastest (the count actually used is reported as `cores`)
            best = (1e30, 1)
            for nt in sorted({min(ncores, c) for c in (8, 16, 32, 64, 128)}):
                torch.set_num_threads(nt)
                O.unet_forward(sd, cfg, x_c[:1], t_c[:1], O.alpha_cosine_log_snr(t_c[:1]), lowres_cond_img=lr_c[:1])
                tp = time.perf_counter()
                O.unet_forward(sd, cfg, x_c[:1], t_c[:1], O.alpha_cosine_log_snr(t_c[:1]), lowres_cond_img=lr_c[:1])
                tp = time.perf_counter() - tp
                if tp < best[0]:
                    best = (tp, nt)
                if tp > 8.0:
                    break
            torch.set_num_threads(best[1])
            O.unet_forward(sd, cfg, x_c, t_c, O.alpha_cosine_log_snr(t_c), lowres_cond_img=lr_c)      # warm-up
            n_it, t0 = 0, time.perf_counter()
            while (time.perf_counter() - t0 < 10.0 or n_it < 2) and time.perf_counter() - t0 < 40.0:
                pred = O.unet_forward(sd, cfg, x_c, t_c, O.alpha_cosine_log_snr(t_c), lowres_cond_img=lr_c)
                mean, _, logvar = O.q_posterior(pred.clamp(min=min_bound), x_c, t_c, t_c - 0.01)
                x_c = mean + (0.5 * logvar).exp() * torch.randn_like(x_c)
                n_it += 1
            dtc = time.perf_counter() - t0
        cpu = dict(value=round(nb * n_it / dtc, 3), unit="patches/s", cores=torch.get_num_threads(), kind="port",
                   sample=f"{n_it} DDPM sampler steps (U-Net eval + posterior step) of {nb} 32^3 patches, oracle/iqt_oracle.py "
                          f"on torch-CPU fp32, anomaly detection off")

    if rank == 0:
        primary = "sample" if "sample" in result else "train"
        out = {
            "metric": ("3D patches/sec (32³, 1ch) — sample steps/sec x batch (one U-Net eval per patch and step: DDPM ancestral; EDM Heun in `edm`, "
                       "train-steps/sec in `train`)") if primary == "sample" else "3D patches/sec (32³, 1ch) — train micro-steps",
            "value": round(result[primary]["patches_per_s"], 2), "unit": "patches/s",
            "n_gpus": world, "steps": K if primary == "sample" else result["train"]["steps"], "warmup": W,
            "ms_per_step": round(result[primary]["ms_per_step"], 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"C2: SRUnet256 dim=64 mults=(1,2,4) 32^3 1ch, batch {B}/GPU, DDPM ancestral sampler step "
                                   f"(1 U-Net eval = {GFLOP_PER_PATCH_EVAL} GFLOP/patch) + ImagenTrainer micro-step",
                       "global_batch": world * B, "patch": f"{S}^3", "parallelism": f"dp{world}"},
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        if "train" in result:
            out["train"] = {k: round(v, 3) for k, v in result["train"].items()}
            out["train"]["note"] = "ImagenTrainer.forward micro-step: fwd+bwd (558 GFLOP/patch), grad all-reduce + fused Adam every 4th, EMA"
        if "train_bf16" in result:
            out["train_bf16"] = {k: round(v, 3) for k, v in result["train_bf16"].items()}
            out["train_bf16"]["note"] = ("same micro-steps with ImagenTrainer(precision='bf16'): forward + backward-data on the bf16 MFMA kernel, "
                                         "weight gradients / Adam / master weights fp32; reduced precision, NOT the headline")
        if "sample" in result:
            out["sample_steps_per_s"] = round(1e3 / result["sample"]["ms_per_step"], 3)
        if "unet3d" in result:
            out["unet3d"] = {k: round(v, 3) for k, v in result["unet3d"].items()}
            out["unet3d"]["note"] = "Family B: Unet3D dim 64, mults (1,2,4), 2 resnet blocks, attention at the last level + middle, 32^3 (190 GFLOP/patch/eval)"
        if "autocast_fp16" in result:
            out["autocast_fp16"] = {k: round(v, 3) for k, v in result["autocast_fp16"].items()}
            out["autocast_fp16"]["note"] = ("same DDPM sampler step under torch.autocast(float16): conv/linear forwards on v_mfma_f32_32x32x16_f16 "
                                            "(fp32 accumulate), all else fp32; reduced precision, NOT the headline")
        if "edm" in result:
            out["edm"] = {k: round(v, 3) for k, v in result["edm"].items()}
            out["edm"]["note"] = "ElucidatedImagen.sample (stochastic Heun, 2 U-Net evals per step except the last) driving the same C2 U-Net"
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
