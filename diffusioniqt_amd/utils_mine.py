"""Counterpart of the reference's ``utils_mine`` (utils_mine.py:8-67): ``set_seed`` and the sub-volume
split / merge used by the trainer and inside ``Unet.forward``.  Tensors are public-layout ``[B, C, X, Y, Z]``;
the movement itself is the HIP gather/scatter kernel (sub-volume n = b2 + f*b3 + f*f*b4, first axis fastest)."""
import random

import numpy as np
import torch

from . import ops
from .imagen_pytorch3D import to_channels_last, to_channels_first


def set_seed(seed):
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)


def count_parameters(model):
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


def _dev(t):
    return t if t.is_cuda else t.cuda()


def convertVolume2subVolume(image, target_shape=(27, 1, 32, 32, 32)):
    """utils_mine.py:25-42."""
    if len(image.shape) != 5 or len(target_shape) != 5:
        raise ValueError("Both input and target shapes must have 5 dimensions")
    _, C1, W, H, D = image.shape
    B, C2, A, _, _ = target_shape
    assert C1 == C2, 'channels are not same'
    assert image.shape[0] == 1 and W == H == D, 'one cubic volume per call (as every reference call site)'
    f = int(W // A)
    if B != f * int(H // A) * int(D // A):
        raise ValueError("The target batch size must be the product of split dimensions")
    was_cuda = image.is_cuda
    sub = to_channels_first(ops.split_volume(to_channels_last(_dev(image).float()), f, A))
    return sub if was_cuda else sub.cpu()


def merge_sub_volumes(sub_volumes, original_shape=(1, 1, 96, 96, 96)):
    """utils_mine.py:44-67."""
    if len(sub_volumes.shape) != 5 or len(original_shape) != 5:
        raise ValueError("Both input and target shapes must have 5 dimensions")
    B, C1, A, _, _ = sub_volumes.shape
    _, C2, W, H, D = original_shape
    assert C1 == C2, 'channels are not same'
    f = int(W // A)
    if B != f * int(H // A) * int(D // A):
        print(B, f, int(H // A), int(D // A))
        raise ValueError("The batch size must be the product of split dimensions")
    was_cuda = sub_volumes.is_cuda
    vol = to_channels_first(ops.merge_volume(to_channels_last(_dev(sub_volumes).float()), f))
    return vol if was_cuda else vol.cpu()
