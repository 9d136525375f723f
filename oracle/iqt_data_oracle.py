"""TEST INFRASTRUCTURE ONLY (never imported by the product): CPU restatement of the reference's training data path and
validation metrics.

* ``supervised_iqt_getitem`` — data.py:88-137 (supervisedIQT.__getitem__): random crop of a 256^3 volume pair, non-zero
  rejection with re-draw, z-score / min-max normalisation.  PINNED by tests/golden/train_patches.npz, which
  oracle/make_golden_data.py produced by executing the reference's own class.
* ``psnr`` / ``ssim`` — metrics.py:19-31 on top of torchmetrics==0.9.0 (requirements.txt:201).  torchmetrics is absent from
  /root/reference and from this image: PARITY UNPINNED for that part — the functions restate its published 0.9.0 algorithm
  (functional/image/psnr.py ``_psnr_compute`` and functional/image/ssim.py ``_ssim_compute`` for 5-D input: Gaussian window
  sized from sigma, reflect padding, depthwise conv3d of {p, t, p*p, t*t, p*t}, crop of the padded border, mean) literally,
  with torch CPU fp32 ops.
"""
import numpy as np
import torch
import torch.nn.functional as F


def supervised_iqt_getitem(lr_vol, hr_vol, cfg, train=True, rng=np.random):
    """-> (hr [1,P,P,P], lr [1,P,P,P], origin, n_draws); ``rng`` must offer ``randint`` (np.random or a RandomState)."""
    tr = cfg['Train']
    P = tr['patch_size_sub'] * tr['batch_sample_factor'] if tr['batch_sample'] else tr['patch_size_sub']   # data.py:59-62
    ratio = 0.2 if train else 0.8                                                                          # data.py:64-67
    low, high = 0, 256                                                                                     # data.py:104
    assert lr_vol.shape == (256, 256, 256) and hr_vol.shape == (256, 256, 256)
    draws = 0
    while True:                                                                                            # the recursion of :122
        o = rng.randint(low=0, high=(high - low) - P, size=3)                                              # data.py:112
        draws += 1
        lr = lr_vol[o[0]:o[0] + P, o[1]:o[1] + P, o[2]:o[2] + P]
        if np.count_nonzero(lr) / (P * P * P) >= ratio:                                                    # data.py:116-122
            break
    hr = hr_vol[o[0]:o[0] + P, o[1]:o[1] + P, o[2]:o[2] + P]

    def normalize(img):                                                                                    # data.py:82-86
        img = torch.from_numpy(np.ascontiguousarray(img, dtype=np.float32))
        if cfg['Data']['norm'] == 'min-max':
            return 2 * (((img - img.min()) / (img.max() - img.min())) - 0.5)
        return (img - cfg['Data']['mean']) / cfg['Data']['std']
    return normalize(hr).unsqueeze(0).numpy(), normalize(lr).unsqueeze(0).numpy(), tuple(int(v) for v in o), draws


def _minmax(x):
    return (x - x.min()) / (x.max() - x.min())


def psnr(pred, target):
    """metrics.py:19-23 -> torchmetrics 0.9.0 peak_signal_noise_ratio(data_range=1.0, base=10, elementwise_mean)."""
    pred, target = _minmax(torch.as_tensor(pred).float()), _minmax(torch.as_tensor(target).float())
    diff = pred - target
    sum_sq, n = torch.sum(diff * diff), target.numel()
    base_e = 2 * torch.log(torch.tensor(1.0)) - torch.log(sum_sq / n)
    return base_e * (10 / torch.log(torch.tensor(10.0)))


def _gaussian(k, sigma):
    dist = torch.arange(start=(1 - k) / 2, end=(1 + k) / 2, step=1, dtype=torch.float32)
    g = torch.exp(-torch.pow(dist / sigma, 2) / 2)
    return (g / g.sum()).unsqueeze(0)                       # [1, k]


def ssim(pred, target, data_range=None, sigma=1.5, k1=0.01, k2=0.03):
    """metrics.py:25-31 (kernel_size=3 only sizes the UNIFORM window of torchmetrics 0.9.0 and is ignored for the default
    Gaussian one) on [B,C,D,H,W]."""
    pred, target = torch.as_tensor(pred).float(), torch.as_tensor(target).float()
    if data_range is None:
        pred, target, data_range = _minmax(pred), _minmax(target), 1.0
    assert pred.ndim == 5
    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    C = pred.size(1)
    K = int(3.5 * sigma + 0.5) * 2 + 1
    pad = (K - 1) // 2
    pred = F.pad(pred, (pad,) * 6, mode='reflect')
    target = F.pad(target, (pad,) * 6, mode='reflect')
    g = _gaussian(K, sigma)
    kxy = torch.matmul(g.t(), g)                                                   # [K, K]
    kernel = (kxy.unsqueeze(-1).repeat(1, 1, K) * g.expand(K, K, K)).expand(C, 1, K, K, K)
    inp = torch.cat((pred, target, pred * pred, target * target, pred * target))
    out = F.conv3d(inp, kernel, groups=C).split(pred.shape[0])
    mp2, mt2, mpt = out[0].pow(2), out[1].pow(2), out[0] * out[1]
    sp, st, spt = out[2] - mp2, out[3] - mt2, out[4] - mpt
    upper, lower = 2 * spt + c2, sp + st + c2
    full = ((2 * mpt + c1) * upper) / ((mp2 + mt2 + c1) * lower)
    return full[..., pad:-pad, pad:-pad, pad:-pad].mean()
