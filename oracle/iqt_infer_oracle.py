"""TEST INFRASTRUCTURE ONLY (never imported by the product): numpy restatement of the reference's whole-volume inference loop,
test_all.py:182-300 with supervisedIQT_INF (data.py:138-202).  Pinned by tests/golden/volume_inference.npz, which
oracle/make_golden_infer.py produced by executing the reference's own loop."""
import numpy as np


def synthetic_volume(n=256, seed=0):
    """Raw-intensity 'brain': a ball of hash-valued intensities (200..1199) in a zero background, plus a zero cavity, so
    the 5 % non-zero rejection (data.py:187-191) and the background reset (test_all.py:300) both fire."""
    i, j, k = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing='ij')
    h = (i * 73856093) ^ (j * 19349663) ^ (k * 83492791) ^ (seed * 2654435761)
    h = (h ^ (h >> 13)) * 1274126177
    h = h ^ (h >> 16)
    val = 200.0 + (h % 1000).astype(np.float32)
    c = n / 2 - 0.5
    r2 = (i - c) ** 2 + (j - c * 0.9) ** 2 + (k - c * 1.1) ** 2
    ball = r2 < (0.40 * n) ** 2
    cavity = ((i - 0.6 * n) ** 2 + (j - 0.5 * n) ** 2 + (k - 0.45 * n) ** 2) < (0.08 * n) ** 2
    return np.where(ball & ~cavity, val, 0.0).astype(np.float32)


def fake_sampler(x):
    """Stand-in for trainer.sample(...)[0]: an exactly representable affine map of the conditioning patch."""
    return x * 0.5 + 0.25


def volume_inference(vol_raw: np.ndarray, cfg: dict, sampler, convert_fn=None, merge_fn=None) -> np.ndarray:
    """vol_raw: float32 [N,N,N] raw intensities.  sampler(patches [B,1,S,S,S] float32 ndarray) -> same shape."""
    mean32, std32 = np.float32(cfg['Data']['mean']), np.float32(cfg['Data']['std'])
    tr = cfg['Train']
    sub = tr['patch_size_sub']
    block = bool(tr['batch_sample'])
    P = sub * tr['batch_sample_factor'] if block else sub                     # data.py:148-151
    stride = cfg['Eval']['overlap']
    op = stride // 2                                                          # test_all.py:219
    n = vol_raw.shape
    lowres = ((vol_raw - mean32) / std32).astype(np.float32)                  # test_all.py:214
    min_val = lowres.min()
    pred = np.full(n, (np.float32(0.) - mean32) / std32, dtype=np.float32)    # test_all.py:210-211
    total = P * P * P
    cands = [(i, j, k) for i in range(0, n[0] - P + 1, stride) for j in range(0, n[1] - P + 1, stride)
             for k in range(0, n[2] - P + 1, stride)]                         # data.py:157-160
    bs = 1 if block else cfg['Eval']['batch_size']                            # test_all.py:184-187
    for b0 in range(0, len(cands), bs):                                       # DataLoader(shuffle=False) + my_collate
        items = []
        for (i, j, k) in cands[b0:b0 + bs]:
            raw = vol_raw[i:i + P, j:j + P, k:k + P].astype(np.float32)
            if np.count_nonzero(raw) / total < 0.05:                          # data.py:187-191
                continue
            items.append((((raw - mean32) / std32).astype(np.float32), (i, j, k)))
        if not items:
            continue
        x = np.stack([it[0] for it in items])[:, None]
        idx = [it[1] for it in items]
        if block:
            y = merge_fn(sampler(convert_fn(x)))                              # test_all.py:229-234, 265-266
        else:
            y = sampler(x)
        if not block and not tr.get('boundary', False):
            assert stride >= P, "the script's overlap < patch branch of this mode raises (test_all.py:241)"
            for q, (i, j, k) in enumerate(idx):                               # test_all.py:262-263
                pred[i:i + P, j:j + P, k:k + P] = y[q, 0]
        else:
            (i, j, k) = idx[0]
            if stride < P:                                                    # test_all.py:267-296
                lo = [0 if o == 0 else op for o in (i, j, k)]
                hi = [0 if (n[a] == o + P or n[a] - P <= o) else op for a, o in enumerate((i, j, k))]
                pred[i + lo[0]:i + P - hi[0], j + lo[1]:j + P - hi[1], k + lo[2]:k + P - hi[2]] = \
                    y[0, 0][lo[0]:P - hi[0], lo[1]:P - hi[1], lo[2]:P - hi[2]]
            else:
                pred[i:i + P, j:j + P, k:k + P] = y[0, 0]
    pred[lowres == min_val] = min_val                                         # test_all.py:300
    return pred
