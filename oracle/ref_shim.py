"""TEST INFRASTRUCTURE (container-only): import the real reference from /root/reference.

The reference (edshkim98/DiffusionIQT) needs nine third-party packages that are not in
this image plus its network-bound local `t5` module.  This shim registers minimal
stand-ins in ``sys.modules`` *before* the reference is imported, so that the reference's
own arithmetic (all of which is torch) runs unmodified on CPU.  It is used only by
``oracle/make_golden*.py`` to produce the fixtures under ``tests/golden/`` and by
``-m "not gpu"`` tests that are skipped when /root/reference is absent.  Nothing here
is imported by the product (``diffusioniqt_amd``), ``bench.py`` or any ``-m gpu`` test.

Recipe follows SURVEY.md Appendix B.
"""
import sys
import types
import copy

REFERENCE_DIR = "/root/reference"


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    """Register stand-ins and put the reference on sys.path.  Idempotent."""
    if getattr(install, "_done", False):
        return
    import torch
    from torch import nn
    from einops import rearrange, repeat
    import typing

    # einops_exts 0.0.4 published semantics: map an einops op over a tuple of tensors.
    def rearrange_many(tensors, pattern, **kw):
        return tuple(rearrange(t, pattern, **kw) for t in tensors)

    def repeat_many(tensors, pattern, **kw):
        return tuple(repeat(t, pattern, **kw) for t in tensors)

    def check_shape(t, pattern, **kw):
        return rearrange(t, f"{pattern} -> {pattern}", **kw)

    class EinopsToAndFrom(nn.Module):
        def __init__(self, from_einops, to_einops, fn):
            super().__init__()
            self.from_einops, self.to_einops, self.fn = from_einops, to_einops, fn

        def forward(self, x, **kwargs):
            shape = x.shape
            names = self.from_einops.split(" ")
            recon = dict(zip(names, shape))
            x = rearrange(x, f"{self.from_einops} -> {self.to_einops}")
            x = self.fn(x, **kwargs)
            x = rearrange(x, f"{self.to_einops} -> {self.from_einops}", **recon)
            return x

    ee = _mod("einops_exts", rearrange_many=rearrange_many, repeat_many=repeat_many,
              check_shape=check_shape)
    eet = _mod("einops_exts.torch", EinopsToAndFrom=EinopsToAndFrom)
    ee.torch = eet

    bt = _mod("beartype", beartype=lambda f: f)
    btt = _mod("beartype.typing", List=typing.List, Union=typing.Union, Tuple=typing.Tuple,
               Optional=typing.Optional, Callable=typing.Callable, Iterable=typing.Iterable,
               Dict=typing.Dict)
    bt.typing = btt
    _mod("beartype.door", is_bearable=lambda *a, **k: True)
    _mod("beartype.vale", Is=type("Is", (), {"__class_getitem__": classmethod(lambda c, x: x)}))

    k = _mod("kornia")
    k.augmentation = _mod("kornia.augmentation", RandomCrop=object)

    tv = _mod("torchvision")
    tvt = _mod("torchvision.transforms", ToPILImage=object, Compose=object, Resize=object,
               Lambda=object, CenterCrop=object, ToTensor=object, RandomHorizontalFlip=object)
    tvt.functional = _mod("torchvision.transforms.functional")
    tv.transforms = tvt
    tv.utils = _mod("torchvision.utils")

    tm = _mod("torchmetrics", StructuralSimilarityIndexMeasure=object,
              MultiScaleStructuralSimilarityIndexMeasure=object)
    tmi = _mod("torchmetrics.image")
    tmi.lpip = _mod("torchmetrics.image.lpip", LearnedPerceptualImagePatchSimilarity=object)
    tm.image = tmi
    tm.functional = _mod("torchmetrics.functional", peak_signal_noise_ratio=None)

    mn = _mod("MedicalNet")
    mn.model = _mod("MedicalNet.model", generate_model=None)
    mn.setting = _mod("MedicalNet.setting", parse_opts=None)

    _mod("t5", get_encoded_dim=lambda name: 768, t5_encode_text=lambda *a, **k: None,
         DEFAULT_T5_NAME="google/t5-v1_1-base")

    # ema_pytorch==0.1.4 published algorithm (parity unpinned: package absent): warm-up
    # decay 1-(1+step/inv_gamma)^-power clamped to [min_value,beta], copy before
    # update_after_step, update every `update_every` calls.
    class EMA(nn.Module):
        def __init__(self, model, beta=0.9999, update_after_step=100, update_every=10,
                     inv_gamma=1.0, power=2 / 3, min_value=0.0, **_):
            super().__init__()
            self.online_model = model
            self.ema_model = copy.deepcopy(model)
            self.ema_model.requires_grad_(False)
            self.beta, self.update_after_step, self.update_every = beta, update_after_step, update_every
            self.inv_gamma, self.power, self.min_value = inv_gamma, power, min_value
            self.register_buffer("initted", torch.tensor([False]))
            self.register_buffer("step", torch.tensor([0]))

        def restore_ema_model_device(self):
            self.ema_model.to(self.initted.device)

        def get_current_decay(self):
            epoch = max(self.step.item() - self.update_after_step - 1, 0.0)
            value = 1 - (1 + epoch / self.inv_gamma) ** -self.power
            return 0.0 if epoch <= 0 else min(max(value, self.min_value), self.beta)

        def update(self):
            step = self.step.item()
            self.step += 1
            if (step % self.update_every) != 0:
                return
            if step <= self.update_after_step or not self.initted.item():
                for pe, po in zip(self.ema_model.parameters(), self.online_model.parameters()):
                    pe.data.copy_(po.data)
                self.initted.data.copy_(torch.tensor([True]))
                if step <= self.update_after_step:
                    return
            d = self.get_current_decay()
            for pe, po in zip(self.ema_model.parameters(), self.online_model.parameters()):
                pe.data.lerp_(po.data, 1 - d)

        def forward(self, *a, **k):
            return self.ema_model(*a, **k)

    _mod("ema_pytorch", EMA=EMA)
    _mod("pytorch_warmup", LinearWarmup=object)
    _mod("nibabel")

    # the trainer's two local imports that drag in torchmetrics / nibabel / torchvision
    def cycle(dl):
        while True:
            for d in dl:
                yield d

    def _psnr(pred, target):
        pred = (pred - pred.min()) / (pred.max() - pred.min())
        target = (target - target.min()) / (target.max() - target.min())
        return 10 * torch.log10(1.0 / torch.mean((pred - target) ** 2))

    _mod("metrics", SSIM=lambda p, t, **k: torch.tensor(0.0), MSSIM=lambda p, t: torch.tensor(0.0),
         PSNR=_psnr)
    _mod("data", cycle=cycle)

    if REFERENCE_DIR not in sys.path:
        sys.path.insert(0, REFERENCE_DIR)
    install._done = True


def import_reference():
    """Returns the reference modules (imagen_pytorch3D, imagen_video, elucidated_imagen, trainer)."""
    install()
    import torch
    import imagen_pytorch3D as ref3d  # noqa: E402  (reference module, /root/reference)
    import imagen_video as refvid
    import elucidated_imagen as refedm
    import trainer as reftrainer
    # the reference switches anomaly detection on at import; keep the fixtures cheap
    torch.autograd.set_detect_anomaly(False)
    reftrainer.ImagenTrainer.set_accelerator_scaler = lambda self, n: None
    return ref3d, refvid, refedm, reftrainer
