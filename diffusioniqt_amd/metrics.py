"""Validation metrics of ``valid_step`` on the device (reference metrics.py:6-35).

The reference min-max normalises both tensors and calls torchmetrics 0.9.0 (requirements.txt:201), which is not vendored:
``peak_signal_noise_ratio(data_range=1.0)`` and ``StructuralSimilarityIndexMeasure(kernel_size=3, data_range=1.0)`` on 5-D
tensors — i.e. the 3-D SSIM with torchmetrics' default GAUSSIAN window (sigma 1.5 -> 11 taps per axis; ``kernel_size`` only
sizes the uniform window and is ignored for the Gaussian one), reflect padding and a crop of the padded border.  Here the
normalisation, the five filtered moments, the SSIM map and its mean are two kernel launches on volumes that never leave HBM
(``diqt_minmax`` + ``diqt_ssim3d`` / ``diqt_psnr``, csrc/datapath.hip).  Parity: torchmetrics is absent from the reference
tree and from this image, so these follow its published 0.9.0 algorithm (restated in oracle/iqt_data_oracle.py) — "parity
unpinned" for the third-party part, pinned for the reference's own normalisation and call pattern.
"""
import numpy as np
import torch

from . import ops


def count_parameters(model):
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


def _dev(t):
    t = t.detach()
    if not t.is_cuda:
        if not torch.cuda.is_available():
            raise RuntimeError("diffusioniqt_amd.metrics runs on the MI355X only (no CPU fallback); use oracle/ for CPU checks")
        t = t.cuda()
    return t.float().contiguous()


def _stats(p, t):
    return torch.cat((ops.minmax(p), ops.minmax(t)))


def gaussian_taps(sigma=1.5):
    """torchmetrics 0.9.0 ``_gaussian``: size int(3.5 sigma + 0.5) * 2 + 1, fp32 arithmetic (host; 11 numbers)."""
    k = int(3.5 * sigma + 0.5) * 2 + 1
    dist = torch.arange(start=(1 - k) / 2, end=(1 + k) / 2, step=1, dtype=torch.float32)
    g = torch.exp(-torch.pow(dist / sigma, 2) / 2)
    return np.ascontiguousarray((g / g.sum()).numpy())


def psnr_impl(pred, target):
    """metrics.py:9-15 (unused by the trainer): 20 log10(max(pred, target) / sqrt(mse))."""
    p, t = _dev(pred), _dev(target)
    mse = ops.psnr(p, t)[0]
    if mse == 0:
        return float('inf')
    peak = torch.maximum(ops.minmax(p)[1], ops.minmax(t)[1])
    return (20 * torch.log10(peak / torch.sqrt(mse))).to(pred.device)


def PSNR(pred, target):
    p, t = _dev(pred), _dev(target)
    return ops.psnr(p, t, _stats(p, t), 1.0)[1].to(pred.device)


def SSIM(pred, target, kernel_size=3, data_range=None):
    p, t = _dev(pred), _dev(target)
    if p.ndim != 5:
        raise NotImplementedError("SSIM: the reference's hot path only scores 5-D [B,C,D,H,W] volumes")
    p, t = p.reshape(-1, *p.shape[2:]), t.reshape(-1, *t.shape[2:])
    taps = gaussian_taps(1.5)
    if min(p.shape[1:]) < taps.shape[0]:
        # torchmetrics crops (K-1)/2 from every face of the SSIM map: nothing is left, and the mean of nothing is NaN
        return torch.full((), float('nan'), device=pred.device)
    if data_range is None:
        out = ops.ssim3d(p, t, taps, _stats(p, t), 1.0)
    else:
        out = ops.ssim3d(p, t, taps, None, float(data_range))
    return out[0].to(pred.device)


def MSSIM(pred, target):
    raise NotImplementedError("MSSIM (torchmetrics MultiScaleStructuralSimilarityIndexMeasure) is imported but never called by "
                              "the reference's trainer (trainer.py:42) — outside the hot path")
