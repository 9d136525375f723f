"""Drop-in import shim for the reference's ``data`` module: ``supervisedIQT`` is the HBM-resident patch sampler
(diffusioniqt_amd/data.py); ``IQTDataset(fake=True)`` keeps the synthetic (hr, lr) contract; ``supervisedIQT_INF`` is the sliding-window patch
Dataset of data.py:139-202 (same item contract, so test_all.py:189-190 iterates it unchanged); the device-resident form of the
same loop is ``diffusioniqt_amd.inference.VolumeInference``."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusioniqt_amd.data import (cycle, my_collate, SyntheticPatchDataset, supervisedIQT, supervisedIQT_INF,  # noqa: E402,F401
                                   DevicePatchLoader)


class IQTDataset(SyntheticPatchDataset):
    """data.py:206-262 with fake=True (the only backend that needs no files)."""

    def __init__(self, hr_files, lr_files, fake=True):
        assert fake, 'npy-slice loading (2-D legacy path) is outside the hot path; use fake=True or supervisedIQT'
        super().__init__(n=max(len(hr_files), 1), size=32)
