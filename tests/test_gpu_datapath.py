"""MI355X checks of the device-side training data path and validation metrics, through the C ABI:
the HBM-resident patch sampler against the fixtures the reference's own supervisedIQT produced (bit-exact), batched crops
against the oracle, PSNR / SSIM kernels against the oracle's torchmetrics-0.9.0 restatement."""
import hashlib

import numpy as np
import pytest
import torch

from oracle import iqt_data_oracle as DO
from oracle.iqt_infer_oracle import synthetic_volume
from tests.conftest import load_golden
from tests.test_data_metrics import CASES, cfg_of, sha

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def vols():
    return [(synthetic_volume(256, seed=v), synthetic_volume(256, seed=100 + v)) for v in range(2)]


def make_dataset(cfg, vols, train):
    from diffusioniqt_amd.data import supervisedIQT
    names = [f'/data/s{v}/lr_norm.nii.gz' for v in range(2)]
    volumes = {}
    for v, (lr, hr) in enumerate(vols):
        volumes[names[v]] = lr
        volumes[names[v].replace('lr_norm', cfg['Data']['groundtruth_fname'])] = hr
    return supervisedIQT(cfg, names, [n.replace('lr_norm', cfg['Data']['groundtruth_fname']) for n in names], train=train,
                         volumes=volumes)


@pytest.mark.parametrize('tag', list(CASES))
def test_sampler_reproduces_the_reference_items_bit_exact(tag, vols):
    g = load_golden('train_patches')
    cfg = cfg_of(g, tag)
    ds = make_dataset(cfg, vols, bool(g[f'{tag}:train']))
    np.random.seed(int(g[f'{tag}:seed']))
    for n, idx in enumerate(g[f'{tag}:order']):
        hr, lr = ds[int(idx)]
        assert hr.is_cuda and tuple(hr.shape) == (1,) + (ds.patch_size,) * 3
        hr, lr = hr.cpu().numpy(), lr.cpu().numpy()
        assert np.array_equal(hr[0, 1::5, 2::5, 3::5], g[f'{tag}:{n}:hr:sub'])
        assert np.array_equal(sha(hr), g[f'{tag}:{n}:hr:sha256']) and np.array_equal(sha(lr), g[f'{tag}:{n}:lr:sha256'])
    assert np.random.randint(0, 1 << 30) == int(g[f'{tag}:next_randint'])


@pytest.mark.parametrize('norm', ['z-score', 'min-max'])
def test_loader_batches_equal_the_oracle_items_in_dataloader_order(norm, vols):
    from diffusioniqt_amd.data import DevicePatchLoader
    g = load_golden('train_patches')
    cfg = cfg_of(g, 'train32')
    cfg['Data']['norm'] = norm
    ds = make_dataset(cfg, vols, True)
    np.random.seed(3)
    torch.manual_seed(9)
    batches = list(DevicePatchLoader(ds, batch_size=2, shuffle=True))
    assert len(batches) == 1 and tuple(batches[0][0].shape) == (2, 1, 32, 32, 32)
    torch.manual_seed(9)
    order = next(iter(torch.utils.data.DataLoader(range(2), batch_size=2, shuffle=True))).tolist()
    np.random.seed(3)
    for r, idx in enumerate(order):
        hr, lr, _, _ = DO.supervised_iqt_getitem(vols[idx][0], vols[idx][1], cfg, train=True)
        assert np.array_equal(batches[0][0][r].cpu().numpy(), hr) and np.array_equal(batches[0][1][r].cpu().numpy(), lr)


@pytest.mark.parametrize('shape', [(2, 1, 32, 32, 32), (1, 2, 24, 19, 13), (1, 1, 96, 96, 96), (3, 1, 11, 11, 11)])
def test_psnr_ssim_kernels_match_the_oracle(shape):
    from diffusioniqt_amd.metrics import PSNR, SSIM
    gen = torch.Generator().manual_seed(sum(shape))
    t = torch.randn(*shape, generator=gen)
    p = t + 0.2 * torch.randn(*shape, generator=gen)
    got_p, got_s = PSNR(p.cuda(), t.cuda()), SSIM(p.cuda(), t.cuda())
    assert got_p.is_cuda and got_p.ndim == 0
    assert abs(float(got_p) - float(DO.psnr(p, t))) < 1e-4 * abs(float(DO.psnr(p, t)))
    assert abs(float(got_s) - float(DO.ssim(p, t))) < 1e-5
    got_r = SSIM(p.cuda(), t.cuda(), data_range=3.0)
    assert abs(float(got_r) - float(DO.ssim(p, t, data_range=3.0))) < 1e-5
    cpu_in = SSIM(p, t)                                           # CPU tensors in -> computed on the device, CPU scalar out
    assert not cpu_in.is_cuda and float(cpu_in) == float(got_s)


def test_ssim_properties_and_small_volume():
    from diffusioniqt_amd.metrics import PSNR, SSIM
    x = torch.randn(2, 1, 32, 32, 32, device='cuda')
    assert abs(float(SSIM(x, x)) - 1.0) < 1e-6                    # identical volumes
    assert float(SSIM(x, x * 3 + 1)) > 0.999999                   # min-max normalisation removes affine intensity maps
    assert float(SSIM(x, -x)) < 0.1
    assert float(PSNR(x, x + 0.0)) == float('inf')
    assert torch.isnan(SSIM(x[..., :8, :8, :8].contiguous(), x[..., :8, :8, :8].contiguous()))   # nothing survives the crop
    a, b = float(SSIM(x, x.flip(2))), float(SSIM(x, x.flip(2)))
    assert a == b                                                 # fixed-order reductions


def test_minmax_and_crop_argument_checks():
    from diffusioniqt_amd import ops
    x = torch.randn(100003, device='cuda')
    mm = ops.minmax(x).cpu()
    assert float(mm[0]) == float(x.min()) and float(mm[1]) == float(x.max())
    v = torch.zeros(1, 8, 8, 8, device='cuda')
    with pytest.raises(RuntimeError, match='bad shape'):
        ops.patch_pair_crop(v, v, torch.zeros(1, 4, dtype=torch.int32, device='cuda'), 9, 0, 0.0, 1.0)
    with pytest.raises(RuntimeError, match='mode'):
        ops.patch_pair_crop(v, v, torch.zeros(1, 4, dtype=torch.int32, device='cuda'), 4, 2, 0.0, 1.0)


def test_valid_step_scores_on_the_device():
    """trainer.valid_step (trainer.py:685-765) end to end on 16^3 patches: SSIM / PSNR come from the device kernels."""
    from diffusioniqt_amd.imagen_pytorch3D import Imagen, NullUnet, SRUnet256
    from diffusioniqt_amd.trainer import ImagenTrainer
    from diffusioniqt_amd.data import SyntheticPatchDataset
    S = 16
    configs = {'Data': {'norm': 'z-score'}, 'Train': {'batch_sample': False, 'patch_size_sub': 16, 'pred_obj': 'x_start',
                                                      'batch_sample_factor': 3}, 'Eval': {'repeat': 1}}
    configs['Data'].update(mean=271.64814106698583, std=377.117173547721)
    configs['Train'].update(timesteps=4, dynamic_threshold=False, batch_size=2, lpips=False, medlpips=False, boundary=False)
    configs['Eval']['batch_size'] = 2
    torch.manual_seed(0)
    unet = SRUnet256(img_size=S, dim=32, dim_mults=(1, 2, 4), channels=1, num_resnet_blocks=(2, 2, 2), init_conv_kernel_size=3,
                     lowres_cond=True, init_cross_embed=False, att_type='linear', attn_dim_head=64, attend_at_middle=False,
                     attend_at_enc=[False] * 3, attend_at_enc_depth=[1] * 3, attend_at_enc_heads=[8] * 3, init_dim=32,
                     memory_efficient=False, use_se_attn='True,', pixel_shuffle_upsample=True, boundary=False,
                     batch_sample=False, batch_sample_factor=3, deep_feature=False)
    imagen = Imagen(configs=configs, unets=(NullUnet(), unet), min_bound=-0.72, image_sizes=(S, S), channels=1,
                    pred_objectives='x_start', timesteps=4, dynamic_thresholding=False, p2_loss_weight_gamma=0.0,
                    auto_normalize_img=False, cond_drop_prob=0.0, lpips=False, medlpips=False, boundary=False).to('cuda')
    ImagenTrainer.locked = False
    trainer = ImagenTrainer(configs=configs, imagen=imagen, split_valid_from_train=False, verbose=False)
    trainer.add_valid_dataset(SyntheticPatchDataset(n=4, size=16, seed=2), batch_size=2)
    loss, preds, x_noisy, (hrs, lowres), ssim, psnr = trainer.valid_step(unet_number=2, max_batch_size=2)
    assert preds.shape == (4, 1, 16, 16, 16) and np.isfinite(ssim) and np.isfinite(psnr)
    want_s = np.mean([float(DO.ssim(preds[i:i + 2], hrs[i:i + 2])) for i in (0, 2)])
    want_p = np.mean([float(DO.psnr(preds[i:i + 2], hrs[i:i + 2])) for i in (0, 2)])
    assert abs(ssim - want_s) < 1e-5 and abs(psnr - want_p) < 1e-3
