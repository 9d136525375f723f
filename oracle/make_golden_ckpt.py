"""TEST INFRASTRUCTURE (container-only): the layout of a checkpoint the REAL reference trainer writes.

Run:  python oracle/make_golden_ckpt.py        (needs /root/reference; CPU only; seconds)

A tiny Family-A trainer (the `unetA_tiny` network of tests/golden) takes five micro-steps through the real
``ImagenTrainer`` (one Adam step on the 4th) and calls the reference's own ``ImagenTrainer.save``
(/root/reference/trainer.py:813-878).  What ``torch.load`` finds in that file is recorded as a MANIFEST — names, shapes,
dtypes and the small scalar entries; no tensor payloads, no source — in tests/golden/ckpt_manifest.npz:

  top_keys                       keys of the checkpoint dict, in the order the reference wrote them
  model / ema                    [name, shape, dtype] per state-dict entry, in order
  optim{i}                       param_groups (all hyper-parameters, `params` index lists) and per state index the entry names,
                                 shapes, dtypes and the `step` value
  scaler{i}                      the GradScaler state dict (empty with fp16 off, trainer.py:364)
  steps / version                values

The `ema` module tree comes from ``ema_pytorch`` (absent here; oracle/ref_shim.py's stand-in follows its published
attribute names: online_model / ema_model / initted / step) wrapped by the reference's own ``nn.ModuleList``
(trainer.py:347-362): that part of the manifest is the reference's key PREFIXES over a restated package — parity unpinned for
the package, like the EMA decay.
"""
import json
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402
from make_golden import base_configs, unet_kwargs_train_py, fill, MIN_BOUND, OUT  # noqa: E402


def entries(sd):
    return [[k, list(v.shape), str(v.dtype).replace('torch.', '')] if isinstance(v, torch.Tensor) else [k, None, type(v).__name__]
            for k, v in sd.items()]


def jsonable(v):
    if isinstance(v, torch.Tensor):
        return v.tolist()
    if isinstance(v, (list, tuple)):
        return [jsonable(x) for x in v]
    if isinstance(v, dict):
        return {str(k): jsonable(x) for k, x in v.items()}
    return v


def main():
    r3, rv, re_, rt = ref_shim.import_reference()
    torch.manual_seed(0)
    np.random.seed(0)
    S, dim = 8, 16
    unet = r3.SRUnet256(**unet_kwargs_train_py(dim, S))
    fill(unet)
    cfgs = base_configs()
    imagen = r3.Imagen(unets=(r3.NullUnet(), unet), configs=cfgs, min_bound=MIN_BOUND, image_sizes=(S, S), channels=1,
                       pred_objectives='x_start', timesteps=4, dynamic_thresholding=False, p2_loss_weight_gamma=0.0,
                       auto_normalize_img=False, cond_drop_prob=0.0)
    trainer = rt.ImagenTrainer(configs=cfgs, imagen=imagen, gradient_accumulation_steps=4, split_valid_from_train=False, verbose=False)
    g = torch.Generator().manual_seed(7)
    trainer.training = True
    n_micro = 5
    for i in range(n_micro):
        hr, lr = torch.randn(2, 1, S, S, S, generator=g), torch.randn(2, 1, S, S, S, generator=g)
        trainer.forward(hr, lowres_img=lr, unet_number=2, max_batch_size=2)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, '3dimagen.pt')
        trainer.save(path)
        obj = torch.load(path, map_location='cpu', weights_only=False)

    man = dict(top_keys=list(obj.keys()), model=entries(obj['model']), ema=entries(obj['ema']),
               steps=jsonable(obj['steps']), version=str(obj['version']), n_micro=n_micro, optim={}, scaler={})
    for k in obj:
        if k.startswith('optim'):
            o = obj[k]
            man['optim'][k] = dict(keys=list(o.keys()), param_groups=jsonable(o['param_groups']),
                                   state={str(i): dict(entries=entries(st), step=float(st['step'])) for i, st in o['state'].items()})
        if k.startswith('scaler'):
            man['scaler'][k] = jsonable(obj[k])
    # which parameter INDEX of optim1 is which named parameter of unets.1 (torch.optim numbers parameters in .parameters() order)
    man['optim1_param_names'] = [n for n, _ in trainer.imagen.unets[1].named_parameters()]
    path = os.path.join(OUT, 'ckpt_manifest.npz')
    np.savez_compressed(path, manifest=np.array(json.dumps(man)))
    print('wrote', path, f'{os.path.getsize(path) / 1024:.1f} KiB;', 'top keys:', man['top_keys'])
    print('optim1: state for', len(man['optim']['optim1']['state']), 'of', len(man['optim1_param_names']), 'parameters; optim0:',
          len(man['optim']['optim0']['state']), '; scaler1:', man['scaler'].get('scaler1'))


if __name__ == '__main__':
    main()
