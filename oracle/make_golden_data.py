"""TEST INFRASTRUCTURE (container-only): training-patch fixtures produced by EXECUTING the reference's
own dataset class (data.py:50-137 supervisedIQT: random crop, non-zero rejection + re-draw, normalisation) with nibabel
replaced by a loader of closed-form synthetic volumes (``iqt_infer_oracle.synthetic_volume``, shared with tests/).

Run:  python oracle/make_golden_data.py        (needs /root/reference)
The fixture holds, per drawn sample, a checksum, a strided sub-sample and the sum of each patch; the reference text is read
from /root/reference at run time and never written anywhere."""
import hashlib
import importlib.util
import os
import sys
import types
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402
from make_golden import save  # noqa: E402
from iqt_infer_oracle import synthetic_volume  # noqa: E402

MEAN, STD = 271.64814106698583, 377.117173547721
GT = 'hr_norm'


def volumes(n):
    """file name -> raw volume.  Low-res: the hash ball; high-res: another hash stream over the same support."""
    out = {}
    for v in range(n):
        out[f'/data/s{v}/lr_norm.nii.gz'] = synthetic_volume(256, seed=v)
        out[f'/data/s{v}/{GT}.nii.gz'] = synthetic_volume(256, seed=100 + v)
    return out


def reference_dataset():
    ref_shim.import_reference()
    for name in ('datasets', 'datasets.utils', 'datasets.utils.file_utils'):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules['datasets.utils.file_utils'].get_datasets_user_agent = lambda: 'none'
    spec = importlib.util.spec_from_file_location('ref_data_real', '/root/reference/data.py')
    data = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(data)
    return data, sys.modules['nibabel']


def case(tag, cfg, train, seed, order, vols, data, out):
    lr_files = sorted(k for k in vols if 'lr_norm' in k)
    ds = data.supervisedIQT(cfg, lr_files, [f.replace('lr_norm', GT) for f in lr_files], train=train)
    np.random.seed(seed)
    sums = []
    for n, idx in enumerate(order):
        hr, lr = ds[idx]
        for nm, t in (('hr', hr), ('lr', lr)):
            a = np.ascontiguousarray(t.numpy())
            assert a.dtype == np.float32 and a.shape[0] == 1
            out[f'{tag}:{n}:{nm}:sha256'] = np.frombuffer(hashlib.sha256(a.tobytes()).digest(), dtype=np.uint8)
            out[f'{tag}:{n}:{nm}:sub'] = a[0, 1::5, 2::5, 3::5].copy()
            sums.append(float(a.astype(np.float64).sum()))
    out[f'{tag}:order'] = np.asarray(order, dtype=np.int64)
    out[f'{tag}:seed'] = seed
    out[f'{tag}:train'] = int(train)
    out[f'{tag}:sums'] = np.asarray(sums)
    out[f'{tag}:next_randint'] = np.random.randint(0, 1 << 30)          # pins how much of the RNG stream was consumed
    print(tag, 'patch', tuple(t.shape), 'sums', sums[:2])


if __name__ == "__main__":
    data, nib = reference_dataset()
    vols = volumes(2)

    class _Img:
        affine = np.eye(4)

        def __init__(self, v):
            self.v = v

        def get_fdata(self):
            return self.v.astype(np.float64)
    nib.load = lambda path: _Img(vols[path])
    out = dict(mean=MEAN, std=STD, groundtruth_fname=GT, n_volumes=2)
    base = {'Data': {'mean': MEAN, 'std': STD, 'norm': 'z-score', 'groundtruth_fname': GT}}
    t32 = {'batch_sample': False, 'patch_size_sub': 32, 'batch_sample_factor': 3}
    t96 = {'batch_sample': True, 'patch_size_sub': 32, 'batch_sample_factor': 3}
    case('train32', {**base, 'Train': t32}, True, 7, [0, 1, 1, 0, 1, 0], vols, data, out)
    case('valid32', {**base, 'Train': t32}, False, 42, [0, 1, 0], vols, data, out)         # ratio 0.8: many re-draws
    case('train96', {**base, 'Train': t96}, True, 3, [1, 0], vols, data, out)
    mm = {'Data': {**base['Data'], 'norm': 'min-max'}, 'Train': t32}
    case('minmax32', mm, True, 11, [0, 1, 0], vols, data, out)
    save('train_patches', **out)
