"""ctypes loader for libdiqt_hip.so (the C ABI declared in include/diqt.h).

The product path has no CPU fallback: if the library is missing or a call fails, this module
raises.  ``call(name, *args)`` converts torch tensors to device pointers, appends nothing on its
own (the caller passes the stream explicitly) and raises ``RuntimeError(diqt_last_error())`` on a
non-zero return code.
"""
import ctypes
import os

# torch must load ITS bundled HIP runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7) before this
# library is dlopen'ed, so that both share one runtime; the other order gives two runtimes in one process
# ("no ROCm-capable device is detected" on the second).
import torch  # noqa: F401
from ctypes import c_int, c_float, c_size_t, c_void_p, c_longlong, c_char_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DIQT_LIB") or os.path.join(_HERE, "csrc", "libdiqt_hip.so")      # DIQT_LIB: another build of the same library (A/B timing)

P, I, F, Z, L = c_void_p, c_int, c_float, c_size_t, c_longlong

# name -> (restype, argtypes)   — must list every function declared in include/diqt.h
PROTOTYPES = {
    "diqt_version": (I, []),
    "diqt_last_error": (c_char_p, []),
    "diqt_census_enable": (I, [I]),
    "diqt_census_count": (L, [c_char_p]),
    "diqt_get_last_launch": (c_char_p, []),
    "diqt_conv_packed_elems": (Z, [I, I, I, I, I]),
    "diqt_conv_pack_weight": (I, [P, P, I, I, I, I, I, I, P]),
    "diqt_conv3d_fwd": (I, [P, P, P, P, P] + [I] * 15 + [P]),
    "diqt_conv3d_fwd_workspace_bytes": (Z, [I] * 15),
    "diqt_conv3d_fwd_ws": (I, [P, P, P, P, P, P, Z] + [I] * 15 + [P]),
    "diqt_conv3d_lds_bytes": (L, [I] * 12),
    "diqt_trilinear_up_fwd": (I, [P, P, I, I, I, I, I, I, P]),
    "diqt_trilinear_up_bwd": (I, [P, P, I, I, I, I, I, I, P]),
    "diqt_conv3d_bwd_weight_workspace_bytes": (Z, [I] * 15),
    "diqt_conv3d_bwd_weight": (I, [P, P, P, P, P, Z] + [I] * 15 + [P]),
    "diqt_conv3d_direct_fwd": (I, [P, P, P, P] + [I] * 19 + [P]),
    "diqt_conv3d_direct_bwd_data": (I, [P, P, P] + [I] * 19 + [P]),
    "diqt_conv3d_direct_bwd_weight": (I, [P, P, P, P] + [I] * 19 + [P]),
    "diqt_conv3d_direct_bwd_weight_workspace_bytes": (Z, [I] * 19),
    "diqt_conv3d_direct_bwd_weight_ws": (I, [P, P, P, P, P, Z] + [I] * 19 + [P]),
    "diqt_reduce_workspace_bytes": (Z, [I, I]),
    "diqt_groupnorm_stats": (I, [P, P, P, P, Z, I, I, I, I, F, P]),
    "diqt_gn_act_fwd": (I, [P, P, P, P, P, P, P, I, P, I, I, I, I, I, P]),
    "diqt_gn_act_fwd_h": (I, [P, P, P, P, P, P, P, I, P, I, I, I, I, I, I, I, P]),
    "diqt_gn_act_bwd": (I, [P, P, P, P, P, P, P, P, I, P, P, P, P, P, P, Z, I, I, I, I, I, P]),
    "diqt_gn_coef_from_partials": (I, [P, I, I, P, P, P, P, I, P, P, P, I, I, I, F, P]),
    "diqt_gn_coef": (I, [P, P, P, P, P, P, I, P, I, I, I, P]),
    "diqt_conv3d_fwd_gn_supported": (I, [I] * 16),
    "diqt_conv3d_fwd_gn": (I, [P, P, P, P, P, P, P, Z, P, I] + [I] * 15 + [P]),
    "diqt_gn_act_bwd_ex": (I, [P, P, P, I, P, P, P, P, P, P, P, I, P, P, P, P, P, P, Z, I, I, I, I, I, P]),
    "diqt_gn_act_bwd_h": (I, [P, P, P, I, P, P, P, P, P, P, P, I, P, P, P, P, P, P, Z, I, I, I, I, I, I, I, I, P]),
    "diqt_chan_layernorm_bwd_ex": (I, [P, P, P, P, P, P, P, P, P, P, Z, I, I, P]),
    "diqt_gn_act_bwd_from_partials": (I, [P, P, P, I, P, P, P, P, P, P, I, P, P, P, P, P, P, Z, I, I, I, I, I, P]),
    "diqt_conv3d_fwd_gnbwd_blocks": (I, [I] * 15),
    "diqt_get_gnbwd_fuse": (I, []),
    "diqt_set_gnbwd_fuse": (I, [I]),
    "diqt_conv3d_fwd_gnbwd": (I, [P, P, P, P, P, P, P, P, P, P, P, I, I, I] + [I] * 15 + [P]),
    "diqt_chan_layernorm_fwd": (I, [P, P, P, P, P, P, I, I, F, P]),
    "diqt_chan_layernorm_fwd_res": (I, [P, P, P, P, P, P, P, I, I, F, P]),
    "diqt_dwconv_temporal_fwd": (I, [P, P, P, P, P, I, I, I, I, I, I, I, P]),
    "diqt_dwconv_temporal_bwd_weight_workspace_bytes": (Z, [I, I, I, I, I]),
    "diqt_dwconv_temporal_bwd_weight": (I, [P, P, P, P, Z, I, I, I, I, I, I, P]),
    "diqt_chan_layernorm_bwd": (I, [P, P, P, P, P, P, P, P, P, Z, I, I, P]),
    "diqt_act_fwd": (I, [P, P, Z, I, P]),
    "diqt_act_bwd": (I, [P, P, P, Z, I, P]),
    "diqt_learned_sinu_fwd": (I, [P, P, P, I, I, P]),
    "diqt_learned_sinu_bwd": (I, [P, P, P, P, I, I, P]),
    "diqt_channel_mean": (I, [P, P, P, Z, I, I, I, P]),
    "diqt_gate_residual_fwd": (I, [P, P, P, P, F, P, I, I, I, P]),
    "diqt_gate_residual_stats_blocks": (I, [I, I]),
    "diqt_gate_residual_fwd_stats": (I, [P, P, P, P, P, I, I, I, P]),
    "diqt_gate_residual_bwd": (I, [P, P, P, P, Z, I, I, I, P]),
    "diqt_gate_residual_fwd_h": (I, [P, P, P, P, F, P, I, I, I, I, I, P]),
    "diqt_gate_residual_fwd_stats_h": (I, [P, P, P, P, P, I, I, I, I, P]),
    "diqt_gate_residual_bwd_h": (I, [P, P, P, P, Z, I, I, I, I, P]),
    "diqt_se_mlp_fwd": (I, [P, P, P, P, P, I, I, I, P]),
    "diqt_se_pool_mlp_fwd": (I, [P, I, I, P, P, P, P, P, I, I, I, P]),
    "diqt_se_mlp_bwd": (I, [P, P, P, P, P, P, P, P, P, P, I, I, I, P]),
    "diqt_add_channel_broadcast": (I, [P, P, F, I, I, I, P]),
    "diqt_space_to_depth2": (I, [P, P, I, I, I, I, I, P]),
    "diqt_depth_to_space2": (I, [P, P, I, I, I, I, I, P]),
    "diqt_concat_channels": (I, [P, I, P, I, P, Z, P]),
    "diqt_split_channels": (I, [P, P, I, P, I, Z, P]),
    "diqt_concat_channels_scaled": (I, [P, I, P, I, F, F, P, Z, P]),
    "diqt_concat_channels_stats_blocks": (I, [I, I, I]),
    "diqt_concat_channels_stats": (I, [P, I, P, I, F, F, P, I, I, P, P]),
    "diqt_split_channels_scaled": (I, [P, P, I, P, I, F, F, Z, P]),
    "diqt_subvolume_gather": (I, [P, P, I, I, I, I, P]),
    "diqt_subvolume_scatter": (I, [P, P, I, I, I, I, I, P]),
    "diqt_q_sample": (I, [P, P, P, P, P, I, Z, P]),
    "diqt_ddpm_step": (I, [P, P, P, P, P, P, F, F, I, P, P, I, Z, P]),
    "diqt_axpby3": (I, [P, P, P, P, P, P, F, F, I, P, I, Z, P]),
    "diqt_loss_clamp_fwd": (I, [P, P, P, P, F, I, I, P, P, I, Z, P]),
    "diqt_loss_clamp_bwd": (I, [P, P, P, F, I, I, F, P, I, Z, P]),
    "diqt_nearest_resize_bwd": (I, [P, P] + [I] * 8 + [P]),
    "diqt_l2norm_rows_fwd": (I, [P, P, P, Z, I, I, I, P]),
    "diqt_l2norm_rows_bwd": (I, [P, P, P, P, Z, I, I, I, P]),
    "diqt_groupnorm_stats_coef": (I, [P, P, P, P, P, I, P, P, P, P, Z, I, I, I, I, F, P]),
    "diqt_mse_clamp_fwd": (I, [P, P, P, P, F, I, P, P, I, Z, P]),
    "diqt_mse_clamp_bwd": (I, [P, P, P, F, I, F, P, I, Z, P]),
    "diqt_adam_step": (I, [P, P, P, P, Z, F, F, F, F, F, F, F, I, P]),
    "diqt_grad_norm_workspace_bytes": (Z, []),
    "diqt_grad_norm_clip": (I, [P, Z, F, P, P, P]),
    "diqt_adam_step_scaled": (I, [P, P, P, P, Z] + [F] * 7 + [I, P, P]),
    "diqt_ema_lerp": (I, [P, P, Z, F, P]),
    "diqt_softmax_fwd": (I, [P, P, Z, I, I, F, P]),
    "diqt_softmax_bwd": (I, [P, P, P, Z, I, I, F, P]),
    "diqt_space_to_depth_nd": (I, [P, P] + [I] * 8 + [P]),
    "diqt_depth_to_space_nd": (I, [P, P] + [I] * 8 + [P]),
    "diqt_transpose_mid": (I, [P, P, I, I, I, I, P]),
    "diqt_nearest_resize": (I, [P, P] + [I] * 8 + [P]),
    "diqt_attn_softmax_fwd": (I, [P, P, P, P, I, I, I, I, I, I, P]),
    "diqt_attn_softmax_bwd": (I, [P, P, P, P, P, I, I, I, I, I, I, P]),
    "diqt_bgemm": (I, [P, P, P, I, I, I, I, I, I, L, L, L, I, I, I, F, F, P]),
    "diqt_bgemm_workspace_bytes": (Z, [I, I, I, I]),
    "diqt_bgemm_ws": (I, [P, P, P, P, Z, I, I, I, I, I, I, L, L, L, I, I, I, F, F, P]),
    "diqt_multi_accumulate": (I, [P, P, I, I, P]),
    "diqt_multi_accumulate_host": (I, [P, P, I, I, P]),
    "diqt_conv3d_fwd_stats_blocks": (I, [I] * 15),
    "diqt_conv3d_fwd_kernel_id": (I, [I] * 15),
    "diqt_conv3d_bwd_weight_kernel_id": (I, [I] * 15),
    "diqt_conv3d_fwd_ex": (I, [P, P, P, P, P, P, P, Z] + [I] * 15 + [P]),
    "diqt_conv3d_fwd_neighbours_stats_blocks": (I, [I, I, I, I, I]),
    "diqt_conv3d_fwd_neighbours": (I, [P, P, P, P, P, P, P, Z, I, I, I, I, I, P]),
    "diqt_groupnorm_stats_from_partials": (I, [P, P, P, I, I, I, I, I, F, P]),
    "diqt_channel_mean_from_partials": (I, [P, P, I, I, I, I, P]),
    "diqt_attn_softmax_bwd_workspace_bytes": (Z, [I, I, I, I, I]),
    "diqt_attn_softmax_bwd_ws": (I, [P, P, P, P, P, P, Z, I, I, I, I, I, I, P]),
    "diqt_mqa_attention_fwd": (I, [P, P, P, P, P, I, I, I, I, I, I, I, F, P]),
    "diqt_mqa_attention_fwd_frames": (I, [P, P, P, P, P, P, I, I, I, I, I, I, F, P]),
    "diqt_set_convh_workgroups": (I, [I]),
    "diqt_set_conv_f9h_mode": (I, [I]),
    "diqt_conv3d_bwd_weight_h_workspace_bytes": (Z, [I] * 15),
    "diqt_conv3d_bwd_weight_h": (I, [P, P, P, P, P, Z] + [I] * 15 + [I, P]),
    "diqt_conv3d_fwd_h_io16_supported": (I, [I] * 17),
    "diqt_conv3d_fwd_h_stats_blocks": (I, [I] * 17),
    "diqt_conv3d_fwd_h_io": (I, [P, P, P, P, P] + [I] * 15 + [I, I, I, I, P, P]),
    "diqt_conv3d_fwd_smallcout_supported": (I, [I] * 15),
    "diqt_conv3d_fwd_smallcout": (I, [P, P, P, P, P] + [I] * 15 + [P]),
    "diqt_temporal_attention_h_supported": (I, [I, I, I, I, I, I]),
    "diqt_temporal_attention_h": (I, [P] * 10 + [I, I, I, I, I, I, I, F, I, I, P]),
    "diqt_mqa_attention_fwd_lse": (I, [P, P, P, P, P, P, I, I, I, I, I, I, I, F, P]),
    "diqt_mqa_attention_bwd_workspace_bytes": (Z, [I, I, I, I, I, I, I]),
    "diqt_mqa_attention_bwd": (I, [P] * 11 + [P, Z, I, I, I, I, I, I, I, F, P]),
    "diqt_weighted_colsum": (I, [P, P, P, P, Z, I, I, I, P]),
    "diqt_softmax_pool_supported": (I, [I, I, I]),
    "diqt_softmax_pool": (I, [P, P, P, P, Z, I, I, I, P]),
    "diqt_patch_gather": (I, [P, P, P, P, I, I, I, I, I, F, F, P]),
    "diqt_patch_scatter": (I, [P, P, P, P, I, I, I, I, I, P]),
    "diqt_background_reset": (I, [P, P, Z, F, F, F, P]),
    "diqt_min_value": (I, [P, Z, P, P, P]),
    "diqt_patch_pair_crop_workspace_bytes": (Z, [I, I]),
    "diqt_patch_pair_crop": (I, [P, P, P, P, P, P, Z, I, I, I, I, I, I, I, F, F, P]),
    "diqt_minmax": (I, [P, Z, P, P, P]),
    "diqt_psnr": (I, [P, P, Z, P, F, P, P, P]),
    "diqt_ssim3d_workspace_bytes": (Z, [I, I, I, I, I]),
    "diqt_ssim3d": (I, [P, P, I, I, I, I, P, I, P, F, F, F, P, Z, P, P]),
    "diqt_abs_quantile": (I, [P, P, I, Z, ctypes.c_uint, F, P]),
    "diqt_dynamic_threshold": (I, [P, P, P, I, Z, P]),
    "diqt_mask_blend": (I, [P, P, P, P, Z, P]),
    "diqt_linear_small_workspace_bytes": (Z, [I, I, I]),
    "diqt_linear_small_fwd": (I, [P, P, P, P, I, I, I, P]),
    "diqt_linear_small_bwd": (I, [P, P, P, P, P, P, P, Z, I, I, I, P]),
    "diqt_mqa_attention_fwd_h": (I, [P, P, P, P, P, I, I, I, I, I, I, I, F, I, I, P]),
    "diqt_cast_to_h": (I, [P, P, Z, I, P]),
    "diqt_conv_packed_h_elems": (Z, [I, I, I, I, I]),
    "diqt_conv_pack_weight_h": (I, [P, P, I, I, I, I, I, I, I, P]),
    "diqt_conv_pack_weight_h_multi": (I, [P, I, I, P]),
    "diqt_conv3d_fwd_h_supported": (I, [I] * 15),
    "diqt_conv3d_fwd_h": (I, [P, P, P, P, P] + [I] * 17 + [P]),
}

_lib = None


def load():
    """Loads libdiqt_hip.so once; raises if it has not been built (``__graft_entry__.build()``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"diffusioniqt_amd: {LIB_PATH} is missing — build it with `python -c 'import __graft_entry__ as g; "
            f"g.build()'` or `bash diffusioniqt_amd/csrc/build.sh`. There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)      # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    return load().diqt_last_error().decode("utf-8", "replace")


def _ptr(a):
    if a is None:
        return None
    if hasattr(a, "data_ptr"):
        return a.data_ptr()
    return a


_PLAIN = (int, float, type(None), bool, bytes)
_FN = {}


def call(name, *args):
    """Calls a status-returning entry point; tensors become raw pointers.  (The marshalling is on the hot path of the host-bound steps:
    ~730 launches per bf16 training micro-step, a dozen arguments each -- plain numbers are passed through by type, no attribute probing.)"""
    fn = _FN.get(name)
    if fn is None:
        fn = _FN[name] = getattr(load(), name)
    rc = fn(*[a if type(a) in _PLAIN else _ptr(a) for a in args])
    if rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {last_error()}")


_memo = {}
SWITCH_EPOCH = 0      # counts diqt_set_* calls: a captured hipGraph froze the kernels the switches selected (graphs.py keys on it)


def query(name, *args):
    """Calls a value-returning entry point.  The shape queries (``*_supported``, ``*_bytes``, ``*_blocks``, ``*_elems``, ``*_kernel_id``) are
    pure functions of their integer arguments and of the library's run-time switches, so their results are memoised -- a conv launch asks
    three or four of them, and launch-bound steps (744 launches per bf16 training micro-step) pay for every ctypes round trip.  Any
    ``diqt_set_*`` call (a switch changes) empties the memo; ``diqt_get_*`` is never cached."""
    if name.startswith("diqt_set_"):
        global SWITCH_EPOCH
        SWITCH_EPOCH += 1
        _memo.clear()
        return getattr(load(), name)(*args)
    if name.startswith("diqt_get_"):
        return getattr(load(), name)(*args)
    key = (name, args)
    try:
        return _memo[key]
    except KeyError:
        v = _memo[key] = getattr(load(), name)(*args)
        return v
    except TypeError:                                    # an unhashable argument: not a shape query
        return getattr(load(), name)(*args)


class census:
    """``with _lib.census() as c: ...; c.count("conv3d_fwd_h(persistent)")`` -- launches of the library inside the block, by tag
    substring (diqt_census_*): the parity suite asserts which kernels a network really dispatched."""

    def __enter__(self):
        self._was = load().diqt_census_enable(1)
        self._frozen = None
        return self

    def __exit__(self, *exc):
        lib = load()
        self._frozen = {}
        lib.diqt_census_enable(0)
        self._lib = lib
        return False

    def count(self, substr=None):
        # counters stay readable after the block (enable(0) stops counting, it does not clear)
        return int(load().diqt_census_count(substr.encode() if substr is not None else None))


def last_launch():
    return load().diqt_get_last_launch().decode()
