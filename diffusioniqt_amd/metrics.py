"""Eval-only host metrics for ``valid_step`` (reference metrics.py:19-34 wraps torchmetrics, which is outside the
hot path — SURVEY.md §2 #9): PSNR on min-max-normalised volumes as the reference computes it, and a plain
global SSIM stand-in (the reference's windowed torchmetrics SSIM is out of scope)."""
import torch


def _minmax(t):
    return (t - t.min()) / (t.max() - t.min())


def PSNR(pred, target):
    pred, target = _minmax(pred.float()), _minmax(target.float())
    return 10 * torch.log10(1.0 / torch.mean((pred - target) ** 2))


def SSIM(pred, target, kernel_size=3, data_range=None):
    x, y = _minmax(pred.float()), _minmax(target.float())
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    mx, my = x.mean(), y.mean()
    vx, vy = x.var(unbiased=False), y.var(unbiased=False)
    cov = ((x - mx) * (y - my)).mean()
    return ((2 * mx * my + c1) * (2 * cov + c2)) / ((mx ** 2 + my ** 2 + c1) * (vx + vy + c2))


def MSSIM(pred, target):
    return SSIM(pred, target)
