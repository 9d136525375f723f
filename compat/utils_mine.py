"""Drop-in import shim: put this directory on sys.path (or copy it next to the reference's train.py / test_all.py) and
`from utils_mine import ...` resolves to the MI355X-native implementation in diffusioniqt_amd.utils_mine."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusioniqt_amd.utils_mine import *  # noqa: F401,F403,E402
from diffusioniqt_amd import utils_mine as _impl  # noqa: E402

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith('__')})
