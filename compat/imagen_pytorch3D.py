"""Drop-in import shim: put this directory on sys.path (or copy it next to the reference's train.py / test_all.py) and
`from imagen_pytorch3D import ...` resolves to the MI355X-native implementation in diffusioniqt_amd.imagen_pytorch3D."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusioniqt_amd.imagen_pytorch3D import *  # noqa: F401,F403,E402
from diffusioniqt_amd import imagen_pytorch3D as _impl  # noqa: E402

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith('__')})
