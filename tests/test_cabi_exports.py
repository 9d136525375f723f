"""The C-ABI library loads and exports every symbol include/diqt.h declares (no compute calls: no GPU here),
the ctypes table in diffusioniqt_amd/_lib.py covers exactly that set, and the product refuses CPU tensors."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "diqt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(diqt_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    from diffusioniqt_amd import _lib
    syms = declared_symbols()
    assert len(syms) >= 45
    assert os.path.exists(_lib.LIB_PATH), "build with __graft_entry__.build()"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/diqt.h but not exported"
    assert sorted(_lib.PROTOTYPES) == syms, set(_lib.PROTOTYPES) ^ set(syms)
    assert _lib.load().diqt_version() >= 100


def test_shape_queries_without_a_gpu():
    from diffusioniqt_amd import _lib
    _lib.load()
    assert _lib.query("diqt_conv_packed_elems", 64, 64, 3, 3, 3) == 2 * 27 * 64 * 32
    assert _lib.query("diqt_conv_packed_elems", 1, 2, 3, 3, 3) == 1 * 27 * 64 * 32
    assert _lib.query("diqt_conv3d_lds_bytes", 32, 32, 32, 3, 3, 3, 1, 1, 1, 0, 0, 0) < 80 * 1024      # two workgroups per CU
    assert _lib.query("diqt_conv3d_lds_bytes", 16, 16, 16, 15, 15, 15, 7, 7, 7, 0, 0, 0) > 160 * 1024  # -> direct kernel
    assert _lib.query("diqt_conv3d_lds_bytes", 0, 4, 4, 3, 3, 3, 1, 1, 1, 0, 0, 0) < 0
    assert _lib.query("diqt_reduce_workspace_bytes", 8, 64) > 0
    assert _lib.query("diqt_conv3d_bwd_weight_workspace_bytes", 8, 32, 32, 32, 64, 64, 3, 3, 3, 1, 1, 1, 0, 0, 0) > 0
    # mixed-precision conv: 16-bit packed size and the shapes it takes (Cin % 4, >= 8 channels for multi-tap filters, < 1 GiB, LDS)
    assert _lib.query("diqt_conv_packed_h_elems", 64, 64, 3, 3, 3) == 2 * 27 * 64 * 32
    geo = (8, 32, 32, 32, 64, 64, 3, 3, 3, 1, 1, 1, 0, 0, 0)
    assert _lib.query("diqt_conv3d_fwd_h_supported", *geo) == 1
    assert _lib.query("diqt_conv3d_fwd_h_supported", 8, 32, 32, 32, 65, 1, 1, 3, 3, 0, 1, 1, 0, 0, 0) == 0     # Cin % 4
    assert _lib.query("diqt_conv3d_fwd_h_supported", 8, 32, 32, 32, 2, 64, 3, 3, 3, 1, 1, 1, 0, 0, 0) == 0      # tap-packed fp32 kernel
    assert _lib.query("diqt_conv3d_fwd_h_supported", 64, 64, 64, 64, 64, 64, 3, 3, 3, 1, 1, 1, 0, 0, 0) == 0    # 4 GiB tensors
    assert _lib.query("diqt_conv3d_fwd_h_supported", 1, 16, 16, 16, 32, 32, 15, 15, 15, 7, 7, 7, 0, 0, 0) == 0  # halo beyond the LDS


def test_bad_arguments_return_error_codes_not_crashes():
    from diffusioniqt_amd import _lib
    lib = _lib.load()
    assert lib.diqt_act_fwd(None, None, 16, 1, None) == -2                      # DIQT_E_ALIGN (null pointer)
    assert b"null pointer" in lib.diqt_last_error()
    assert lib.diqt_groupnorm_stats(None, None, None, None, 0, 1, 1, 6, 4, 1e-5, None) == -2
    with pytest.raises(RuntimeError):
        _lib.call("diqt_act_fwd", None, None, 16, 1, None)
    assert lib.diqt_conv3d_fwd_h(None, None, None, None, None, *([1] * 15), 0, 1, None) == -2
    assert lib.diqt_conv_pack_weight_h(None, None, 8, 8, 3, 3, 3, 0, 0, None) == -2
    assert lib.diqt_mqa_attention_fwd_h(None, None, None, None, None, 1, 1, 1, 64, 1, 1, 0, 1.0, 0, 1, None) == -2
    assert lib.diqt_cast_to_h(None, None, 16, 0, None) == -2


def test_product_has_no_cpu_fallback():
    from diffusioniqt_amd import ops
    from diffusioniqt_amd.imagen_pytorch3D import SRUnet256
    with pytest.raises(RuntimeError, match="no CPU fallback|HIP"):
        ops.conv3d(torch.randn(1, 4, 4, 4, 8), torch.randn(8, 8, 3, 3, 3), None, 1)
    unet = SRUnet256(dim=16, img_size=8, dim_mults=(1, 2), num_resnet_blocks=(1, 1), channels=1, lowres_cond=True,
                     init_cross_embed=False, init_dim=16, memory_efficient=False, attend_at_enc=[False, False],
                     attend_at_middle=False, deep_feature=False)
    with pytest.raises(RuntimeError):
        unet(torch.randn(1, 1, 8, 8, 8), None, torch.zeros(1), lowres_cond_img=torch.randn(1, 1, 8, 8, 8))


def test_state_dict_contract_matches_reference_keys():
    import json
    import numpy as np
    from diffusioniqt_amd.imagen_pytorch3D import SRUnet256
    from diffusioniqt_amd.imagen_video import Unet3D
    for name, klass in (('unetA_tiny', SRUnet256), ('unetA_attn_linear', SRUnet256), ('unetA_memeff', SRUnet256),
                        ('unetA_boundary', SRUnet256), ('unet3d_tiny', Unet3D)):
        g = dict(np.load(os.path.join(ROOT, 'tests', 'golden', name + '.npz')))
        kw = {k: (tuple(v) if isinstance(v, list) and klass is Unet3D else v) for k, v in json.loads(str(g['kwargs'])).items()}
        sd = klass(**kw).state_dict()
        assert list(sd.keys()) == [str(k) for k in g['keys']], name
        assert [list(v.shape) for v in sd.values()] == [json.loads(str(s)) for s in g['shapes']], name
