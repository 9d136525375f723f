"""TEST INFRASTRUCTURE (container-only): the reference's mixed-precision forward (torch.autocast on CPU, fp16 and bf16) on the
inputs and weights of the committed ``unet3d_tiny`` / ``unetA_tiny`` fixtures.  Run: python oracle/make_golden_autocast.py
(needs /root/reference; CPU; ~20 s).  Numbers only: y32 (fp32 eval), y_fp16, y_bf16 per family.

The reference has no fp16 switch of its own for sampling: C5's "fp16" is ``torch.autocast`` around ``sample`` and
``ImagenTrainer(fp16=True)`` hands Family-A training to accelerate's autocast (trainer.py:293-311)."""
import os
import sys
import json
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402
from iqt_oracle import hash_fill_state_dict  # noqa: E402
from make_golden import save  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")
T = lambda a: torch.from_numpy(np.asarray(a))


def three(fn):
    with torch.no_grad():
        out = {'y32': fn()}
        for name, dt in (('y_fp16', torch.float16), ('y_bf16', torch.bfloat16)):
            with torch.autocast('cpu', dtype=dt):
                out[name] = fn().float()
    return out


if __name__ == "__main__":
    r3, rv, re_, rt = ref_shim.import_reference()
    torch.set_num_threads(8)

    gb = dict(np.load(os.path.join(GOLDEN, "unet3d_tiny.npz")))
    kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in json.loads(str(gb['kwargs'])).items()}
    u3 = rv.Unet3D(**kw).eval()
    u3.load_state_dict(hash_fill_state_dict(u3.state_dict(), 11))
    outs_b = three(lambda: u3(T(gb['x']), T(gb['time']), lowres_cond_img=T(gb['lowres']), lowres_noise_times=T(gb['lowres_times'])))

    ga = dict(np.load(os.path.join(GOLDEN, "unetA_tiny.npz")))
    kwa = json.loads(str(ga['kwargs']))
    ua = r3.SRUnet256(**kwa).eval()
    ua.load_state_dict(hash_fill_state_dict(ua.state_dict(), 0))
    outs_a = three(lambda: ua(T(ga['x']), T(ga['times']), T(ga['log_snr']), lowres_cond_img=T(ga['lowres'])))
    assert torch.allclose(outs_a['y32'], T(ga['y']), atol=1e-5), "fp32 eval must reproduce the committed fixture"

    for fam, o in (('B', outs_b), ('A', outs_a)):
        for k in ('y_fp16', 'y_bf16'):
            print(fam, k, 'rel-L2 vs fp32:', ((o[k] - o['y32']).norm() / o['y32'].norm()).item())
    save("autocast_fwd", **{f"B_{k}": v for k, v in outs_b.items()}, **{f"A_{k}": v for k, v in outs_a.items()})
