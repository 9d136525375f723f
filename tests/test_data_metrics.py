"""CPU checks of the training data path: the oracle restatement of supervisedIQT.__getitem__ against
the fixtures the reference's own class produced (tests/golden/train_patches.npz), the product's HOST logic (summed-area-table
rejection, DataLoader index order) against the oracle / torch's DataLoader, and the metric restatements against an
independent float64 formulation.  The device kernels are covered by tests/test_gpu_datapath.py."""
import hashlib

import numpy as np
import pytest
import torch

from oracle import iqt_data_oracle as DO
from oracle.iqt_infer_oracle import synthetic_volume
from tests.conftest import load_golden

CASES = {
    'train32': dict(norm='z-score', batch_sample=False),
    'valid32': dict(norm='z-score', batch_sample=False),
    'train96': dict(norm='z-score', batch_sample=True),
    'minmax32': dict(norm='min-max', batch_sample=False),
}


@pytest.fixture(scope='module')
def vols():
    return [(synthetic_volume(256, seed=v), synthetic_volume(256, seed=100 + v)) for v in range(2)]


def cfg_of(g, tag):
    c = CASES[tag]
    return {'Data': {'mean': float(g['mean']), 'std': float(g['std']), 'norm': c['norm'], 'groundtruth_fname': str(g['groundtruth_fname'])},
            'Train': {'batch_sample': c['batch_sample'], 'patch_size_sub': 32, 'batch_sample_factor': 3}}


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)


@pytest.mark.parametrize('tag', list(CASES))
def test_oracle_getitem_matches_reference_fixture_bit_exact(tag, vols):
    g = load_golden('train_patches')
    cfg = cfg_of(g, tag)
    np.random.seed(int(g[f'{tag}:seed']))
    for n, idx in enumerate(g[f'{tag}:order']):
        lr_vol, hr_vol = vols[int(idx)]
        hr, lr, origin, draws = DO.supervised_iqt_getitem(lr_vol, hr_vol, cfg, train=bool(g[f'{tag}:train']))
        assert np.array_equal(sha(hr), g[f'{tag}:{n}:hr:sha256']) and np.array_equal(sha(lr), g[f'{tag}:{n}:lr:sha256'])
        assert np.array_equal(hr[0, 1::5, 2::5, 3::5], g[f'{tag}:{n}:hr:sub'])
    assert np.random.randint(0, 1 << 30) == int(g[f'{tag}:next_randint'])      # same RNG consumption as the reference


@pytest.mark.parametrize('tag', ['train32', 'valid32', 'train96'])
def test_product_host_rejection_draws_the_oracles_origins(tag, vols):
    from diffusioniqt_amd.data import supervisedIQT

    class HostOnly(supervisedIQT):          # the HBM upload is exercised on the GPU box
        def _upload(self, idx, lr, hr):
            pass
    g = load_golden('train_patches')
    cfg = cfg_of(g, tag)
    train = bool(g[f'{tag}:train'])
    names = [f'/data/s{v}/lr_norm.nii.gz' for v in range(2)]
    volumes = {}
    for v, (lr, hr) in enumerate(vols):
        volumes[names[v]] = lr
        volumes[names[v].replace('lr_norm', cfg['Data']['groundtruth_fname'])] = hr
    ds = HostOnly(cfg, names, names, train=train, device='cpu', volumes=volumes)
    order = [int(i) for i in g[f'{tag}:order']]
    np.random.seed(int(g[f'{tag}:seed']))
    got = [ds.draw_origin(i) for i in order]
    assert np.random.randint(0, 1 << 30) == int(g[f'{tag}:next_randint'])
    np.random.seed(int(g[f'{tag}:seed']))
    want, draws = [], 0
    for i in order:
        _, _, o, d = DO.supervised_iqt_getitem(vols[i][0], vols[i][1], cfg, train=train)
        want.append(o)
        draws += d
    assert got == want and ds.draws == draws
    if tag == 'valid32':
        assert draws > len(order)           # the 0.8 ratio really rejected some crops


def test_sat_count_equals_count_nonzero():
    from diffusioniqt_amd.data import nonzero_sat, sat_count
    rng = np.random.RandomState(0)
    v = (rng.rand(20, 17, 23) > 0.6) * rng.rand(20, 17, 23).astype(np.float32)
    s = nonzero_sat(v)
    for _ in range(50):
        P = int(rng.randint(1, 8))
        i, j, k = (int(rng.randint(0, d - P + 1)) for d in v.shape)
        assert sat_count(s, i, j, k, P) == np.count_nonzero(v[i:i + P, j:j + P, k:k + P])


@pytest.mark.parametrize('shuffle,drop_last,n,bs', [(True, False, 7, 3), (False, False, 5, 2), (True, True, 8, 3)])
def test_device_patch_loader_walks_the_dataset_in_dataloader_order(shuffle, drop_last, n, bs):
    from diffusioniqt_amd.data import DevicePatchLoader

    class Idx(torch.utils.data.Dataset):
        def __len__(self):
            return n

        def __getitem__(self, i):
            return i

        def get_batch(self, idx):
            return list(idx)
    torch.manual_seed(5)
    dl = torch.utils.data.DataLoader(Idx(), batch_size=bs, shuffle=shuffle, drop_last=drop_last)
    want = [[b.tolist() for b in dl] for _ in range(2)]            # two epochs
    after_ref = torch.rand(1)
    torch.manual_seed(5)
    mine = DevicePatchLoader(Idx(), batch_size=bs, shuffle=shuffle, drop_last=drop_last)
    got = [list(mine) for _ in range(2)]
    assert got == want and len(mine) == len(dl)
    assert torch.equal(torch.rand(1), after_ref)                    # same consumption of torch's default generator


def test_metric_restatements_against_float64_formulation():
    import scipy.ndimage as ndi
    torch.manual_seed(0)
    p = torch.randn(2, 1, 24, 24, 24)
    t = p + 0.1 * torch.randn_like(p)
    pn, tn = (x.double().numpy() for x in (p, t))
    pn, tn = (pn - pn.min()) / (pn.max() - pn.min()), (tn - tn.min()) / (tn.max() - tn.min())
    k = np.arange(-5, 6)
    gk = np.exp(-(k / 1.5) ** 2 / 2)
    gk /= gk.sum()

    def filt(x):
        for ax in (2, 3, 4):
            x = ndi.correlate1d(x, gk, axis=ax, mode='reflect')
        return x[..., 5:-5, 5:-5, 5:-5]
    mp, mt = filt(pn), filt(tn)
    sp, st, spt = filt(pn * pn) - mp * mp, filt(tn * tn) - mt * mt, filt(pn * tn) - mp * mt
    want = (((2 * mp * mt + 1e-4) * (2 * spt + 9e-4)) / ((mp * mp + mt * mt + 1e-4) * (sp + st + 9e-4))).mean()
    assert abs(float(DO.ssim(p, t)) - want) < 2e-6
    assert abs(float(DO.psnr(p, t)) - 10 * np.log10(1.0 / np.mean((pn - tn) ** 2))) < 1e-4


def test_gaussian_taps_match_the_oracle_window():
    from diffusioniqt_amd.metrics import gaussian_taps
    assert np.array_equal(gaussian_taps(1.5), DO._gaussian(11, 1.5)[0].numpy())
