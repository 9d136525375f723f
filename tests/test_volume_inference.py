"""Whole-volume inference: the oracle restatement and the device pipeline against fixtures made by
executing the reference's own script loop (oracle/make_golden_infer.py).  Pure index / data-movement work: bit-exact (SHA-256
of the stitched 256^3 volume)."""
import hashlib

import numpy as np
import pytest
import torch

from oracle import iqt_oracle as O
from oracle.iqt_infer_oracle import volume_inference, synthetic_volume, fake_sampler
from tests.conftest import load_golden

CASES = {
    'plain32': dict(Train={'batch_sample': False, 'boundary': False, 'patch_size_sub': 32, 'batch_sample_factor': 3},
                    Eval={'batch_size': 27, 'overlap': 32}),
    'block96': dict(Train={'batch_sample': True, 'boundary': False, 'patch_size_sub': 32, 'batch_sample_factor': 3},
                    Eval={'batch_size': 27, 'overlap': 32}),
    'block96_s64': dict(Train={'batch_sample': True, 'boundary': True, 'patch_size_sub': 32, 'batch_sample_factor': 3},
                        Eval={'batch_size': 27, 'overlap': 64}),
}


def _cfg(g, tag):
    return {'Data': {'mean': float(g['mean']), 'std': float(g['std']), 'norm': 'z-score'}, **CASES[tag]}


def _check(pred, g, tag):
    assert pred.dtype == np.float32 and pred.shape == (256, 256, 256)
    assert np.array_equal(pred[3::16, 5::16, 7::16], g[f'{tag}:sub']), f'{tag}: sub-sampled voxels differ'
    sha = np.frombuffer(hashlib.sha256(np.ascontiguousarray(pred).tobytes()).digest(), dtype=np.uint8)
    assert (sha == g[f'{tag}:sha256']).all(), f'{tag}: stitched volume is not bit-identical to the reference loop'


@pytest.mark.parametrize("tag", list(CASES))
def test_oracle_matches_reference_loop(tag):
    g = load_golden('volume_inference')
    conv = lambda x: O.convert_volume_to_subvolume(torch.from_numpy(x), (27, 1, 32, 32, 32)).numpy()
    mer = lambda y: O.merge_sub_volumes(torch.from_numpy(y), (1, 1, 96, 96, 96)).numpy()
    _check(volume_inference(synthetic_volume(), _cfg(g, tag), fake_sampler, conv, mer), g, tag)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", list(CASES))
def test_device_pipeline_matches_reference_loop(tag):
    from diffusioniqt_amd.inference import VolumeInference
    g = load_golden('volume_inference')
    vol = torch.from_numpy(synthetic_volume()).cuda()
    pred = VolumeInference(_cfg(g, tag), sample_fn=lambda x: x * 0.5 + 0.25)(vol)
    _check(pred.cpu().numpy(), g, tag)


@pytest.mark.gpu
def test_device_pipeline_shards_over_ranks():
    """Two ranks take alternate kept patches; a voxel is owned by whichever rank wrote it last in candidate order -- for the
    non-overlapping tiling the union of the two shards equals the single-GPU result."""
    from diffusioniqt_amd.inference import VolumeInference
    g = load_golden('volume_inference')
    vol = torch.from_numpy(synthetic_volume()).cuda()
    inf = VolumeInference(_cfg(g, 'plain32'), sample_fn=lambda x: x * 0.5 + 0.25)
    a, b = inf(vol, patch_slice=(0, 2)), inf(vol, patch_slice=(1, 2))
    fill = float((np.float32(0.) - np.float32(g['mean'])) / np.float32(g['std']))
    merged = torch.where(a != fill, a, b)
    _check(merged.cpu().numpy(), g, 'plain32')


def test_supervisedIQT_INF_dataset_drives_the_script_loop():
    """compat/data.supervisedIQT_INF (data.py:139-202) iterated the way test_all.py:189-263 does — DataLoader(shuffle=False,
    collate_fn=my_collate), whole-patch placement — reproduces the reference loop's stitched volume bit for bit."""
    import os
    import sys
    from torch.utils.data import DataLoader
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'compat'))
    import data as compat_data                                                 # what `from data import ...` resolves to
    g = load_golden('volume_inference')
    cfg = _cfg(g, 'plain32')
    vol = synthetic_volume()
    ds = compat_data.supervisedIQT_INF(cfg, vol)
    assert len(ds) == 512 and ds.lr_idx[1] == [0, 0, 32] and ds.lr_idx[8] == [0, 32, 0]
    assert ds[0] is None                                                       # an empty corner: rejected (< 5 % non-zero)
    mean32, std32 = np.float32(cfg['Data']['mean']), np.float32(cfg['Data']['std'])
    pred = np.full(vol.shape, (np.float32(0.) - mean32) / std32, dtype=np.float32)
    kept = 0
    for batch in DataLoader(ds, batch_size=cfg['Eval']['batch_size'], shuffle=False, collate_fn=compat_data.my_collate):
        if batch is None:
            continue
        x, idx = batch
        assert x.shape[1:] == (1, 32, 32, 32) and idx.shape[1] == 3
        y = fake_sampler(x.numpy())
        for q, (i, j, k) in enumerate(idx.tolist()):
            pred[i:i + 32, j:j + 32, k:k + 32] = y[q, 0]
        kept += x.shape[0]
    lowres = ((vol - mean32) / std32).astype(np.float32)
    pred[lowres == lowres.min()] = lowres.min()
    assert 0 < kept < 512
    _check(pred, g, 'plain32')


@pytest.mark.gpu
def test_device_pipeline_overlapping_windows_are_deterministic():
    """Eval.overlap (the stride) below half a patch: cropped interiors of neighbouring windows overlap; the device pipeline
    must reproduce the serial candidate-order loop ("later overwrites", test_all.py:267-296 crop rule) exactly."""
    from diffusioniqt_amd.inference import VolumeInference, sliding_window_origins, crop_margins
    rng = np.random.default_rng(0)
    n, P, stride = 64, 32, 8
    vol = rng.integers(1, 1000, size=(n, n, n)).astype(np.float32)
    cfg = {'Data': {'mean': 300.0, 'std': 200.0, 'norm': 'z-score'},
           'Train': {'batch_sample': False, 'boundary': False, 'patch_size_sub': P, 'batch_sample_factor': 3},
           'Eval': {'batch_size': 16, 'overlap': stride}}
    calls = {'n': 0}

    def sampler(x):                     # differs per call, so which window wrote a voxel last is visible in the result
        calls['n'] += 1
        return x * 0.5 + 0.125 * torch.arange(x.shape[0], device=x.device, dtype=x.dtype).view(-1, 1, 1, 1, 1) + calls['n']
    got = VolumeInference(cfg, sample_fn=sampler)(torch.from_numpy(vol).cuda()).cpu().numpy()
    mean32, std32 = np.float32(300.0), np.float32(200.0)
    org = sliding_window_origins((n, n, n), P, stride)
    mg = crop_margins(org, (n, n, n), P, stride)
    ref = np.full((n, n, n), (np.float32(0.) - mean32) / std32, dtype=np.float32)
    for b0 in range(0, org.shape[0], 16):
        for q, ((i, j, k), m) in enumerate(zip(org[b0:b0 + 16], mg[b0:b0 + 16])):
            x = ((vol[i:i + P, j:j + P, k:k + P] - mean32) / std32).astype(np.float32)
            y = (x * np.float32(0.5) + np.float32(0.125) * np.float32(q) + np.float32(b0 // 16 + 1)).astype(np.float32)
            ref[i + m[0]:i + P - m[1], j + m[2]:j + P - m[3], k + m[4]:k + P - m[5]] = y[m[0]:P - m[1], m[2]:P - m[3], m[4]:P - m[5]]
    lowres = ((vol - mean32) / std32).astype(np.float32)
    ref[lowres == lowres.min()] = lowres.min()
    assert np.array_equal(got, ref)
