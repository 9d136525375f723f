"""Drop-in import shim for the reference's ``data`` module.  NIfTI I/O (supervisedIQT / supervisedIQT_INF, data.py:50-202)
is host-side disk work outside the hot path: the synthetic patch datasets keep the same (hr, lr) tuple
contract; the NIfTI classes raise with an explanation."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusioniqt_amd.data import cycle, SyntheticPatchDataset  # noqa: E402
from torch.utils.data.dataloader import default_collate  # noqa: E402


class IQTDataset(SyntheticPatchDataset):
    """data.py:206-262 with fake=True (the only backend that needs no files)."""

    def __init__(self, hr_files, lr_files, fake=True):
        assert fake, 'npy-file loading is outside the hot path; use fake=True or your own Dataset returning (hr, lr)'
        super().__init__(n=max(len(hr_files), 1), size=32)


def my_collate(batch):
    batch = [b for b in batch if b is not None]
    return None if batch == [] else default_collate(batch)


def _nifti(*a, **k):
    raise NotImplementedError('supervisedIQT / supervisedIQT_INF read HCP NIfTI volumes with nibabel: host I/O outside the '
                              'MI355X hot path (SURVEY.md §2 #7). Supply any torch Dataset yielding (hr, lr) [1,S,S,S] pairs.')


supervisedIQT = supervisedIQT_INF = _nifti
