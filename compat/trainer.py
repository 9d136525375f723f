"""Drop-in import shim: put this directory on sys.path (or copy it next to the reference's train.py / test_all.py) and
`from trainer import ...` resolves to the MI355X-native implementation in diffusioniqt_amd.trainer."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusioniqt_amd.trainer import *  # noqa: F401,F403,E402
from diffusioniqt_amd import trainer as _impl  # noqa: E402

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith('__')})
