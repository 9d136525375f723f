"""Whole-volume inference — the sliding-window loop of the reference's test scripts (test_all.py:182-300, test.py) with its
dataset (`supervisedIQT_INF`, data.py:138-202), kept on the device: the raw low-res volume stays resident in HBM, patches are
gathered / normalised / rejected by one kernel, sampled in batches, stitched by a scatter kernel that applies the script's
overlap-crop rules, and the background is reset in place.  One job per volume; several volumes or patch ranges shard over
GPUs with no collective.

Semantics kept from the scripts:
* candidate origins: ``range(0, N - P + 1, overlap)`` per axis, i outermost (data.py:157-160; ``Eval.overlap`` is the STRIDE);
* a candidate is dropped when fewer than 5 % of its RAW voxels are non-zero (data.py:187-191);
* patches are z-scored with ``Data.mean/std`` (data.py:165-169); the output volume starts at ``(0 - mean) / std``;
* ``Train.batch_sample``: the window is ``patch_size_sub * batch_sample_factor`` (96), each block is split into 27 sub-volumes
  for the sampler and merged back (utils_mine.py:25-67); blocks overlap, and each writes its interior ``[op : P - op]``
  (``op = overlap // 2``) except on faces that touch the volume boundary (test_all.py:267-296);
* without ``batch_sample`` and with ``overlap >= patch`` patches are placed whole (test_all.py:262-263).  The script's
  overlap < patch branch of that mode raises on its first interior patch (a 3-element tensor in a boolean ``or``,
  test_all.py:241) — here it applies the same per-face crop rule as the block mode instead;
* voxels whose normalised low-res value equals the volume minimum are reset to it (test_all.py:300).
"""
import numpy as np
import torch

from . import ops
from .utils_mine import convertVolume2subVolume, merge_sub_volumes


def sliding_window_origins(shape, patch, stride):
    """data.py:157-160 -> int32 [n, 3]"""
    rng = [range(0, s - patch + 1, stride) for s in shape]
    return np.array([[i, j, k] for i in rng[0] for j in rng[1] for k in rng[2]], dtype=np.int32).reshape(-1, 3)


def crop_margins(origins, n, patch, overlap):
    """Per-patch {lo, hi} crop per axis (test_all.py:267-296): overlap//2 inside, 0 on faces at the volume boundary."""
    op = overlap // 2
    m = np.zeros((origins.shape[0], 6), dtype=np.int32)
    if overlap >= patch:
        return m
    for a in range(3):
        o = origins[:, a]
        m[:, 2 * a] = np.where(o == 0, 0, op)
        m[:, 2 * a + 1] = np.where((o + patch == n[a]) | (n[a] - patch <= o), 0, op)
    return m


class VolumeInference:
    def __init__(self, configs, sample_fn, nonzero_ratio=0.05):
        """``sample_fn(lr_patches [B,1,S,S,S]) -> hr_patches`` — e.g. ``lambda x: trainer.sample(batch_size=x.shape[0],
        start_image_or_video=x, start_at_unet_number=2)[0]`` (test_all.py:234)."""
        self.cfg = configs
        self.sample_fn = sample_fn
        self.ratio = nonzero_ratio
        tr = configs['Train']
        self.sub = int(tr['patch_size_sub'])
        self.block_mode = bool(tr.get('batch_sample', False))
        self.factor = int(tr.get('batch_sample_factor', 3))
        self.patch = self.sub * self.factor if self.block_mode else self.sub
        self.overlap = int(configs['Eval']['overlap'])
        self.batch = int(configs['Eval'].get('batch_size', 27))
        self.mean, self.std = float(configs['Data']['mean']), float(configs['Data']['std'])

    @torch.no_grad()
    def __call__(self, lowres_raw, patch_slice=None):
        """lowres_raw: fp32 [D,H,W] raw intensities on the GPU.  Returns the stitched, z-scored prediction [D,H,W].
        ``patch_slice`` (rank, world) restricts the work to every world-th kept patch (multi-GPU sharding; merge the shards with
        the returned mask-free volumes by taking, per voxel, the value of the rank that owns it — see ``shard_volumes``)."""
        vol = lowres_raw.float().contiguous()
        dev = vol.device
        shape = tuple(vol.shape)
        P = self.patch
        origins = sliding_window_origins(shape, P, self.overlap)
        idx_all = torch.from_numpy(origins).to(dev)
        _, nz = ops.patch_gather(vol, idx_all, P, self.mean, self.std, want_patches=False, want_nonzero=True)
        keep = (nz.cpu().numpy().astype(np.float64) / float(P ** 3)) >= self.ratio          # data.py:187-191
        kept = origins[keep]
        if patch_slice is not None:
            rank, world = patch_slice
            kept = kept[rank::world]
        margins = crop_margins(kept, shape, P, self.overlap)
        mean32, std32 = np.float32(self.mean), np.float32(self.std)
        pred = torch.full(shape, float((np.float32(0.) - mean32) / std32), dtype=torch.float32, device=dev)
        # overlapping blocks are written one launch at a time in candidate order (later blocks overwrite, as in the script);
        # non-overlapping patches go out in one launch per batch
        per_call = 1 if self.block_mode else self.batch
        # cropped interiors (width P - 2*(overlap//2), stride overlap) of neighbouring windows still overlap when the stride is
        # below half a patch: one scatter launch per patch, in candidate order, keeps "later overwrites" deterministic
        serial_scatter = (not self.block_mode) and self.overlap < P and P - 2 * (self.overlap // 2) > self.overlap
        for lo in range(0, kept.shape[0], per_call):
            o = np.ascontiguousarray(kept[lo:lo + per_call])
            idx = torch.from_numpy(o).to(dev)
            x, _ = ops.patch_gather(vol, idx, P, self.mean, self.std)
            if self.block_mode:                                                           # test_all.py:229-231, 265-266
                sub = convertVolume2subVolume(x, target_shape=(self.factor ** 3, 1, self.sub, self.sub, self.sub))
                y = self.sample_fn(sub)
                y = merge_sub_volumes(y.float(), original_shape=(1, 1, P, P, P))
            else:
                y = self.sample_fn(x)
            y = y.float().contiguous()
            mg = torch.from_numpy(np.ascontiguousarray(margins[lo:lo + per_call])).to(dev)
            if serial_scatter:
                for j in range(o.shape[0]):
                    ops.patch_scatter(y[j:j + 1], idx[j:j + 1], mg[j:j + 1], pred, P)
            else:
                ops.patch_scatter(y, idx, mg, pred, P)
        min_raw = float(ops.min_value(vol).item())
        min_val = (np.float32(min_raw) - mean32) / std32                                  # monotone map: min of the normalised volume
        ops.background_reset(pred, vol, self.mean, self.std, float(min_val))             # test_all.py:300
        return pred
