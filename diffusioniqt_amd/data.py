"""Training data path on the device (reference data.py:50-137 ``supervisedIQT``).

The reference decodes two 64 MB NIfTI volumes from disk for EVERY patch, crops on the host, counts non-zeros, possibly
re-draws (recursing into another decode), normalises and ships 2 x 128 KB to the GPU — per sample.  Here a volume pair is
decoded once (host, nibabel / .npy), parked in HBM (288 GB holds ~2000 pairs) and every later ``__getitem__`` / batch is

  host : draw the crop origin with the reference's RNG call (``np.random.randint(0, 256 - P, 3)``), evaluate the non-zero
         count of the candidate crop in O(1) from a summed-area table of the low-res volume, re-draw while it is below the
         ratio (0.2 train / 0.8 validation) — the same accept/reject sequence and RNG consumption as the reference,
  GPU  : ONE ``diqt_patch_pair_crop`` launch for the whole batch (crop + z-score / min-max normalisation).

``DevicePatchLoader`` walks the dataset in ``torch.utils.data.DataLoader(shuffle=...)`` order (same draws from torch's
default generator), so a run is sample-for-sample the reference's ``num_workers=0`` run.
"""
import numpy as np
import torch
from torch.utils.data import Dataset
from torch.utils.data.dataloader import default_collate

from . import ops


def cycle(dl):
    while True:
        for data in dl:
            yield data


def my_collate(batch):
    """data.py:42-48."""
    batch = [b for b in batch if b is not None]
    return None if batch == [] else default_collate(batch)


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


def load_volume(path):
    """Host decode of one volume to float32 (data.py:95-101).  ``.npy`` needs nothing; NIfTI needs nibabel."""
    path = str(path)
    if path.endswith('.npy'):
        return np.load(path).astype(np.float32)
    try:
        import nibabel as nib
    except ImportError as e:
        raise RuntimeError(f"reading {path!r} needs nibabel (the reference's NIfTI decoder, requirements.txt:104); pass "
                           "pre-decoded arrays through `volumes=` or .npy files instead") from e
    return nib.load(path).get_fdata().astype(np.float32)


def nonzero_sat(vol):
    """int32 summed-area table S[i,j,k] = #non-zeros in vol[:i,:j,:k]  ([n+1]^3, zero-padded front faces)."""
    nz = (np.asarray(vol) != 0)
    s = np.zeros(tuple(d + 1 for d in nz.shape), dtype=np.int32)
    s[1:, 1:, 1:] = nz.cumsum(0, dtype=np.int32).cumsum(1, dtype=np.int32).cumsum(2, dtype=np.int32)
    return s


def sat_count(s, i, j, k, P):
    """np.count_nonzero(vol[i:i+P, j:j+P, k:k+P]) from the table."""
    a, b, c = i + P, j + P, k + P
    return int(s[a, b, c]) - int(s[i, b, c]) - int(s[a, j, c]) - int(s[a, b, k]) \
        + int(s[i, j, c]) + int(s[i, b, k]) + int(s[a, j, k]) - int(s[i, j, k])


class supervisedIQT(Dataset):
    """Mirror of data.py:50-137 with the volumes resident in HBM.

    Same constructor (``config, lr_files, hr_files, train=True``) and the same ``(hr, lr)`` item contract, except that the
    items are device tensors.  ``volumes`` optionally maps a file name to an already decoded float32 array (the low-res name
    and the derived high-res name, data.py:91), which bypasses the file system."""

    volume_size = 256           # data.py:106-107 asserts 256^3; the crop origin is drawn in [0, 256 - patch)

    def __init__(self, config, lr_files, hr_files, train=True, device=None, volumes=None):
        self.config = config
        self.lr_files, self.hr_files = lr_files, hr_files
        self.mean_lr, self.std_lr = config['Data']['mean'], config['Data']['std']
        tr = config['Train']
        self.patch_size = tr['patch_size_sub'] * tr['batch_sample_factor'] if tr['batch_sample'] else tr['patch_size_sub']
        self.train = train
        self.ratio = 0.2 if train else 0.8
        self.files_lr, self.files_hr = list(lr_files), list(hr_files)[:len(lr_files)]
        self.device = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
        self._volumes = volumes or {}
        n, s = len(self.files_lr), self.volume_size
        self._lr = self._hr = None          # [V, s, s, s] device stacks, allocated on first use
        self._sat = [None] * n
        self.total_voxel = self.patch_size ** 3
        self.draws = 0                      # crop origins drawn so far (accepted + rejected)

    def __len__(self):
        return len(self.files_lr)

    # ---- cache -------------------------------------------------------------------------------------------------------
    def _decode(self, name):
        return np.asarray(self._volumes[name], dtype=np.float32) if name in self._volumes else load_volume(name)

    def _upload(self, idx, lr, hr):
        s = self.volume_size
        if self._lr is None:
            self._lr = torch.empty(len(self), s, s, s, device=self.device, dtype=torch.float32)
            self._hr = torch.empty_like(self._lr)
        self._lr[idx].copy_(torch.from_numpy(np.ascontiguousarray(lr)))
        self._hr[idx].copy_(torch.from_numpy(np.ascontiguousarray(hr)))

    def _ensure(self, idx):
        if self._sat[idx] is not None:
            return
        s = self.volume_size
        lr_name = self.files_lr[idx]
        hr_name = lr_name.replace('lr_norm', self.config['Data']['groundtruth_fname'])          # data.py:91
        lr, hr = self._decode(lr_name), self._decode(hr_name)
        assert lr.shape == (s, s, s), f'lr must be {s} {s} {s} but got {lr.shape}'
        assert hr.shape == (s, s, s), f'hr must be {s} {s} {s} but got {hr.shape}'
        self._upload(idx, lr, hr)
        self._sat[idx] = nonzero_sat(lr)

    def preload(self):
        for i in range(len(self)):
            self._ensure(i)
        return self

    # ---- host side: the reference's accept / reject sequence ------------------------------------------------------------
    def draw_origin(self, idx):
        """data.py:112-122: ``np.random.randint(0, 256 - P, 3)`` until the low-res crop holds >= ratio non-zero voxels."""
        self._ensure(idx)
        P, sat = self.patch_size, self._sat[idx]
        while True:
            o = np.random.randint(low=0, high=self.volume_size - P, size=3)
            self.draws += 1
            if sat_count(sat, int(o[0]), int(o[1]), int(o[2]), P) / self.total_voxel >= self.ratio:
                return int(o[0]), int(o[1]), int(o[2])

    # ---- device side ---------------------------------------------------------------------------------------------------
    def get_batch(self, indices):
        """(hr [B,1,P,P,P], lr [B,1,P,P,P]) for the dataset indices, drawn in order, cropped by one launch."""
        sel = np.empty((len(indices), 4), dtype=np.int32)
        for r, idx in enumerate(indices):
            sel[r, 0] = idx
            sel[r, 1:] = self.draw_origin(int(idx))
        mode = 1 if self.config['Data']['norm'] == 'min-max' else 0
        lr, hr = ops.patch_pair_crop(self._lr, self._hr, torch.from_numpy(sel).to(self.device), self.patch_size, mode,
                                     self.mean_lr, self.std_lr)
        return hr.unsqueeze(1), lr.unsqueeze(1)

    def __getitem__(self, idx):
        hr, lr = self.get_batch([idx])
        return hr[0], lr[0]


class supervisedIQT_INF(Dataset):
    """Mirror of data.py:139-202 — the sliding-window patch Dataset the reference's test scripts iterate
    (``DataLoader(supervisedIQT_INF(configs, lr_file), batch_size, shuffle=False, collate_fn=my_collate)``, test_all.py:189-190).

    Same constructor ``(config, lr_file)``, candidate order (i outermost, stride ``Eval.overlap``, data.py:157-160), 5 %
    non-zero rejection (``None`` items, dropped by ``my_collate``) and ``[lr[1,P,P,P], tensor([i,j,k])]`` item contract; items
    are host tensors like the reference's.  ``lr_file`` is a NIfTI / ``.npy`` path or an already decoded array (cropped to
    ``[0:256]^3`` like data.py:155).  ``VolumeInference`` is the device-resident pipeline over the same candidates; this class
    exists so the scripts' own loops run unchanged."""

    def __init__(self, config, lr_file):
        self.lr_file, self.config = lr_file, config
        self.mean_lr, self.std_lr = config['Data']['mean'], config['Data']['std']
        tr = config['Train']
        self.patch_size = tr['patch_size_sub'] * tr['batch_sample_factor'] if tr['batch_sample'] else tr['patch_size_sub']
        self.overlap = config['Eval']['overlap']
        self.ratio = 0.05
        self.total_voxel = self.patch_size ** 3
        vol = load_volume(lr_file) if isinstance(lr_file, (str, bytes)) or hasattr(lr_file, '__fspath__') \
            else (lr_file.detach().cpu().numpy() if torch.is_tensor(lr_file) else np.asarray(lr_file))
        self.lr_data = vol[0:256, 0:256, 0:256]
        P, n = self.patch_size, self.lr_data.shape
        self.lr_idx = [[i, j, k] for i in range(0, n[0] - P + 1, self.overlap) for j in range(0, n[1] - P + 1, self.overlap)
                       for k in range(0, n[2] - P + 1, self.overlap)]

    def __len__(self):
        return len(self.lr_idx)

    def normalize(self, img):
        return (torch.as_tensor(img, dtype=torch.float32) - self.mean_lr) / self.std_lr

    def __getitem__(self, idx):
        i, j, k = self.lr_idx[idx]
        P = self.patch_size
        self.lr = torch.tensor(np.asarray(self.lr_data[i:i + P, j:j + P, k:k + P]).astype(np.float32))
        self.img_shape = self.lr.shape
        if np.count_nonzero(self.lr) / self.total_voxel < self.ratio:
            return None
        self.lr = self.normalize(self.lr)
        return [torch.unsqueeze(self.lr, 0), torch.tensor(self.lr_idx[idx])]


class DevicePatchLoader:
    """``DataLoader(dataset, batch_size, shuffle, drop_last)`` (train.py:56,67) over a ``supervisedIQT``: same index order —
    including the two draws a DataLoader epoch takes from torch's default generator — one crop launch per batch."""

    def __init__(self, dataset, batch_size=1, shuffle=False, drop_last=False):
        self.dataset, self.batch_size, self.shuffle, self.drop_last = dataset, int(batch_size), shuffle, drop_last

    def __len__(self):
        n = len(self.dataset)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def _order(self):
        n = len(self.dataset)
        torch.empty((), dtype=torch.int64).random_()                       # _BaseDataLoaderIter's base seed
        if not self.shuffle:
            return list(range(n))
        seed = int(torch.empty((), dtype=torch.int64).random_().item())   # RandomSampler.__iter__
        g = torch.Generator()
        g.manual_seed(seed)
        return torch.randperm(n, generator=g).tolist()

    def __iter__(self):
        order = self._order()
        for b0 in range(0, len(order), self.batch_size):
            batch = order[b0:b0 + self.batch_size]
            if len(batch) < self.batch_size and self.drop_last:
                return
            yield self.dataset.get_batch(batch)
