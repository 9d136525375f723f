"""Synthetic stand-in for the reference's patch datasets (data.py:50-262; NIfTI I/O is out of scope,
SURVEY.md §2 #7): same ``(hr, lr)`` tuple contract as ``supervisedIQT`` / ``IQTDataset(fake=True)``."""
import torch
from torch.utils.data import Dataset


def cycle(dl):
    while True:
        for data in dl:
            yield data


class SyntheticPatchDataset(Dataset):
    """``n`` pairs of z-scored-like N(0,1) fp32 patches [1,S,S,S] from a fixed seed (data.py:259-261)."""

    def __init__(self, n=8, size=32, seed=42, channels=1):
        g = torch.Generator().manual_seed(seed)
        self.hr = torch.randn(n, channels, size, size, size, generator=g)
        self.lr = torch.randn(n, channels, size, size, size, generator=g)

    def __len__(self):
        return self.hr.shape[0]

    def __getitem__(self, idx):
        return self.hr[idx], self.lr[idx]


IQTDatasetFake = SyntheticPatchDataset
