"""TEST INFRASTRUCTURE (container-only): whole-volume inference fixtures produced by EXECUTING
the reference's own script loop (test_all.py:182-300: sliding-window dataset, 5 % non-zero rejection, batch collation,
sub-volume split/merge, overlap-crop stitching, background reset) against its own dataset class (data.py:138-202), with
nibabel and the trainer replaced by deterministic stand-ins.

Run:  python oracle/make_golden_infer.py        (needs /root/reference)
The reference text is read from /root/reference at run time and never written anywhere; the fixture holds only the
parameters, a checksum of the stitched volume and a strided sub-sample of it.  The synthetic input volume is a closed-form
function of the voxel coordinates (``iqt_infer_oracle.synthetic_volume``, shared with tests/)."""
import hashlib
import importlib.util
import os
import sys
import time
import types
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402
from make_golden import save  # noqa: E402

N = 256     # the script hard-codes 256^3 volumes (test_all.py:201-206)


from iqt_infer_oracle import synthetic_volume, fake_sampler  # noqa: E402  (shared with the tests)


def run_reference_loop(cfg, vol):
    src = open('/root/reference/test_all.py').read().split('\n')
    seg = '\n'.join(src[181:300])            # `for lrfile, hrfile in zip(...)` ... background reset (1-based lines 182-300)
    ref_shim.import_reference()              # installs the package stubs (nibabel, torchvision, ...) and sys.path
    for name in ('datasets', 'datasets.utils', 'datasets.utils.file_utils'):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules['datasets.utils.file_utils'].get_datasets_user_agent = lambda: 'none'
    nib = sys.modules['nibabel']

    class _Img:
        affine = np.eye(4)

        def get_fdata(self):
            return vol.astype(np.float64)
    nib.load = lambda path: _Img()
    spec = importlib.util.spec_from_file_location('ref_data_real', '/root/reference/data.py')
    data = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(data)
    um = sys.modules['utils_mine']

    class _Trainer:
        def sample(self, batch_size, start_image_or_video, **kw):
            assert batch_size == start_image_or_video.shape[0]
            return fake_sampler(start_image_or_video), None, None
    g = dict(configs=cfg, lrfiles=['a/b/c/lr'], hrfiles=['a/b/c/hr'], nib=nib, np=np, torch=torch, time=time,
             DataLoader=torch.utils.data.DataLoader, supervisedIQT_INF=data.supervisedIQT_INF, my_collate=data.my_collate,
             convertVolume2subVolume=um.convertVolume2subVolume, merge_sub_volumes=um.merge_sub_volumes, trainer=_Trainer(),
             device='cpu', cube=lambda v: v, print=lambda *a, **k: None)
    exec(compile(seg, 'reference:test_all.py[182:300]', 'exec'), g)
    return g['pred_ary'].numpy()


def fixture(tag, cfg, out):
    vol = synthetic_volume()
    pred = run_reference_loop(cfg, vol)
    out[f'{tag}:sha256'] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(pred).tobytes()).digest(), dtype=np.uint8)
    out[f'{tag}:sub'] = pred[3::16, 5::16, 7::16].copy()
    out[f'{tag}:sum'] = np.float64(pred.astype(np.float64).sum())
    print(tag, pred.shape, float(pred.mean()))


if __name__ == "__main__":
    mean, std = 271.64814106698583, 377.117173547721
    out = dict(mean=mean, std=std)
    base = {'Data': {'mean': mean, 'std': std, 'norm': 'z-score'}}
    # (a) eval_config.yaml: 32^3 patches, stride 32, batches of 27
    fixture('plain32', {**base, 'Train': {'batch_sample': False, 'boundary': False, 'patch_size_sub': 32, 'batch_sample_factor': 3},
                        'Eval': {'batch_size': 27, 'overlap': 32}}, out)
    # (b) batch_sample: 96^3 blocks at stride 32 -> 27 x 32^3 sub-volumes, overlap-crop stitching with op = 16
    fixture('block96', {**base, 'Train': {'batch_sample': True, 'boundary': False, 'patch_size_sub': 32, 'batch_sample_factor': 3},
                        'Eval': {'batch_size': 27, 'overlap': 32}}, out)
    # (c) batch_sample with stride 64 (wider interior crop) and boundary flag set
    fixture('block96_s64', {**base, 'Train': {'batch_sample': True, 'boundary': True, 'patch_size_sub': 32, 'batch_sample_factor': 3},
                            'Eval': {'batch_size': 27, 'overlap': 64}}, out)
    save("volume_inference", **out)
