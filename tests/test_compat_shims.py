"""The reference scripts' imports (train.py:11-13,22; test_all.py:21-23,36-37) resolve through compat/ to the
MI355X-native implementation (import-level check; the flow itself runs in tests/test_gpu_flow.py)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_script_imports_resolve():
    code = ("import sys; sys.path.insert(0, %r);"
            "from imagen_pytorch3D import Unet, NullUnet, Imagen, SRUnet256, BaseUnet64, alpha_cosine_log_snr;"
            "from trainer import ImagenTrainer; from imagen_video import Unet3D; from elucidated_imagen import ElucidatedImagen;"
            "from utils_mine import set_seed, convertVolume2subVolume, merge_sub_volumes; from data import IQTDataset, cycle, my_collate;"
            "from metrics import PSNR, SSIM; import diffusioniqt_amd.imagen_pytorch3D as m; assert SRUnet256 is m.SRUnet256; print('ok')"
            ) % os.path.join(ROOT, 'compat')
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and 'ok' in out.stdout, out.stderr[-2000:]
