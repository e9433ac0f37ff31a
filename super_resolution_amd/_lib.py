"""ctypes binding of libsr355.so (C ABI declared in include/sr355.h).

The library is built in-tree by ``super_resolution_amd/csrc/Makefile``.  There is no
fallback: if the shared object is missing, loading raises and every op fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# SR355_LIB names another build of the same ABI (A/B measurements of kernel changes)
LIB_PATH = os.environ.get("SR355_LIB") or os.path.join(_HERE, "libsr355.so")
CSRC_DIR = os.path.join(_HERE, "csrc")

DTYPE_BF16 = 0
DTYPE_F32 = 1
ABI_VERSION = 18
SPLIT_NONE, SPLIT_WEIGHT, SPLIT_INPUT = 0, 1, 2
STATIC_SCRATCH_BYTES = 16 << 20   # SR355_STATIC_SCRATCH_BYTES
PACK_F16 = 3     # sr355_pack_weight: one weight image stored as fp16
TRUNK_RRDB = 2   # sr355_generator_cfg.trunk_fp32: fp32 carriers for the RRDB / global skips only (inference)
HR_F16 = 2       # sr355_generator_cfg.hr_split: HR-side convs in fp16 (inference)

# every symbol include/sr355.h declares
EXPORTS = (
    "sr355_abi_version", "sr355_last_error", "sr355_unit_channels",
    "sr355_nchw_to_blocked", "sr355_blocked_to_nchw",
    "sr355_packed_weight_bytes", "sr355_pack_weight",
    "sr355_conv3x3_fwd", "sr355_dense_block_fwd",
    "sr355_pack_weight_transposed", "sr355_pack_weights", "sr355_pack_weights_transposed", "sr355_pool2x2_sum", "sr355_wgrad_workspace_bytes", "sr355_conv3x3_wgrad",
    "sr355_conv3x3_wgrad_static", "sr355_wgrad_f32_workspace_bytes", "sr355_conv3x3_wgrad_f32",
    "sr355_generator_num_convs", "sr355_generator_packed_bytes", "sr355_generator_pack",
    "sr355_generator_workspace_bytes", "sr355_generator_forward", "sr355_generator_forward_u8",
    "sr355_chain_timing", "sr355_chain_last_ms", "sr355_chain_status", "sr355_chain_reenable",
    "sr355_dense_block_wgrad_workspace_bytes", "sr355_dense_block_wgrad", "sr355_dense_blocks_wgrad",
    "sr355_generator_saved_bytes", "sr355_generator_train_workspace_bytes", "sr355_generator_forward_train",
    "sr355_generator_packed_bwd_bytes", "sr355_generator_pack_bwd", "sr355_generator_bwd_workspace_bytes",
    "sr355_generator_grads_floats", "sr355_generator_backward",
    "sr355_generator_backward_phases", "sr355_generator_backward_range",
    "sr355_adam_step",
    "sr355_bn_workspace_bytes", "sr355_bn_lrelu_fwd", "sr355_bn_lrelu_bwd", "sr355_bn_lrelu_fwd_groups",
    "sr355_bn_lrelu_bwd_groups",
    "sr355_pixel_shuffle2", "sr355_maxpool2x2_fwd", "sr355_maxpool2x2_bwd", "sr355_loss_workspace_bytes", "sr355_loss_fwd",
)


class ConvArgs(C.Structure):
    """Mirror of ``sr355_conv_args``."""
    _fields_ = [
        ("dtype", C.c_int), ("N", C.c_int), ("H", C.c_int), ("W", C.c_int), ("upsample", C.c_int),
        ("in_dev", C.c_void_p), ("in_bytes", C.c_size_t),
        ("in_units", C.c_int), ("cin_units", C.c_int),
        ("w_packed_dev", C.c_void_p), ("w_packed_bytes", C.c_size_t), ("weight_split", C.c_int),
        ("bias_packed_dev", C.c_void_p),
        ("cout", C.c_int), ("lrelu_slope", C.c_float),
        ("mask_dev", C.c_void_p), ("mask_bytes", C.c_size_t),
        ("mask_units", C.c_int), ("mask_unit_off", C.c_int), ("mask_slope", C.c_float),
        ("res1_dev", C.c_void_p), ("res1_weight", C.c_float), ("res1_scale", C.c_float),
        ("res2_dev", C.c_void_p), ("res2_weight", C.c_float), ("res2_scale", C.c_float),
        ("res_bytes", C.c_size_t),
        ("out_act_dev", C.c_void_p), ("out_act_bytes", C.c_size_t),
        ("out_units", C.c_int), ("out_unit_off", C.c_int),
        ("out_f32_dev", C.c_void_p),
        ("out_nchw_dev", C.c_void_p), ("out_nchw_bytes", C.c_size_t),
        ("f16_in", C.c_int), ("f16_out", C.c_int),
        ("res_act_dev", C.c_void_p), ("res_act_bytes", C.c_size_t), ("res_act_units", C.c_int),
        ("stride", C.c_int), ("in_H", C.c_int), ("in_W", C.c_int), ("zero_insert", C.c_int),
    ]


class DenseBlockArgs(C.Structure):
    """Mirror of ``sr355_dense_block_args``."""
    _fields_ = [
        ("N", C.c_int), ("H", C.c_int), ("W", C.c_int),
        ("buf_dev", C.c_void_p), ("buf_bytes", C.c_size_t), ("store_growth", C.c_int),
        ("w_packed_dev", C.c_void_p * 5), ("w_packed_bytes", C.c_size_t * 5),
        ("bias_packed_dev", C.c_void_p * 5),
        ("lrelu_slope", C.c_float),
        ("mask_dev", C.c_void_p), ("mask_bytes", C.c_size_t),
        ("mask_units", C.c_int), ("mask_unit_off", C.c_int * 4), ("mask_slope", C.c_float),
        ("res1_dev", C.c_void_p), ("res1_weight", C.c_float), ("res1_scale", C.c_float),
        ("res2_dev", C.c_void_p), ("res2_weight", C.c_float), ("res2_scale", C.c_float),
        ("res_bytes", C.c_size_t),
        ("res_act_dev", C.c_void_p), ("res_act_bytes", C.c_size_t), ("res_act_units", C.c_int),
        ("out_act_dev", C.c_void_p), ("out_act_bytes", C.c_size_t),
        ("out_units", C.c_int), ("out_unit_off", C.c_int),
        ("out_f32_dev", C.c_void_p),
    ]


class PackJob(C.Structure):
    """Mirror of ``sr355_pack_job``."""
    _fields_ = [("w_dev", C.c_void_p), ("bias_dev", C.c_void_p), ("wp_dev", C.c_void_p), ("wp_bytes", C.c_size_t),
                ("bp_dev", C.c_void_p), ("cout", C.c_int), ("cin", C.c_int), ("split", C.c_int)]


class PackTJob(C.Structure):
    """Mirror of ``sr355_pack_t_job``."""
    _fields_ = [("w_dev", C.c_void_p), ("wp_dev", C.c_void_p), ("wp_bytes", C.c_size_t), ("cout", C.c_int),
                ("cin", C.c_int), ("ci_lo", C.c_int), ("n_out", C.c_int), ("packed_units", C.c_int),
                ("unit_off", C.c_int), ("scale", C.c_float)]


class WgradArgs(C.Structure):
    """Mirror of ``sr355_wgrad_args``."""
    _fields_ = [
        ("N", C.c_int), ("H", C.c_int), ("W", C.c_int), ("upsample", C.c_int),
        ("in_dev", C.c_void_p), ("in_bytes", C.c_size_t), ("in_units", C.c_int), ("cin", C.c_int),
        ("dy_dev", C.c_void_p), ("dy_bytes", C.c_size_t), ("dy_units", C.c_int), ("dy_unit_off", C.c_int),
        ("cout", C.c_int), ("scale", C.c_float), ("accumulate", C.c_int),
        ("dw_dev", C.c_void_p), ("dw_bytes", C.c_size_t), ("db_dev", C.c_void_p),
        ("workspace_dev", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("dy_stride2", C.c_int),
    ]


class DenseWgradArgs(C.Structure):
    """Mirror of ``sr355_dense_wgrad_args``."""
    _fields_ = [
        ("N", C.c_int), ("H", C.c_int), ("W", C.c_int),
        ("act_dev", C.c_void_p), ("act_bytes", C.c_size_t),
        ("stack_dev", C.c_void_p), ("stack_bytes", C.c_size_t),
        ("units", C.c_int), ("num_features", C.c_int), ("growth_channels", C.c_int),
        ("growth_scale", C.c_float), ("conv5_scale", C.c_float),
        ("dw_dev", C.c_void_p * 5), ("db_dev", C.c_void_p * 5),
        ("workspace_dev", C.c_void_p), ("workspace_bytes", C.c_size_t),
    ]


class GeneratorCfg(C.Structure):
    """Mirror of ``sr355_generator_cfg``."""
    _fields_ = [
        ("in_channels", C.c_int), ("out_channels", C.c_int), ("num_features", C.c_int),
        ("num_blocks", C.c_int), ("growth_channels", C.c_int), ("dtype", C.c_int),
        ("trunk_fp32", C.c_int), ("hr_split", C.c_int),
    ]


class Sr355Error(RuntimeError):
    pass


def build(verbose: bool = False) -> str:
    """Compile libsr355.so for gfx950 with hipcc (works without a GPU)."""
    cmd = ["make", "-C", CSRC_DIR, "-j8"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode:
        print(res.stdout)
    if res.returncode:
        raise Sr355Error(f"building libsr355.so failed (exit {res.returncode})")
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Sr355Error(
            f"{LIB_PATH} is missing: build it with `make -C {CSRC_DIR}` (or "
            "`python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, sz, i, f = C.c_void_p, C.c_size_t, C.c_int, C.c_float
    L.sr355_abi_version.restype = i
    L.sr355_last_error.restype = C.c_char_p
    L.sr355_unit_channels.argtypes = [i]
    L.sr355_nchw_to_blocked.argtypes = [vp, i, i, i, i, i, vp, i, i, sz, vp]
    L.sr355_blocked_to_nchw.argtypes = [vp, i, i, i, i, i, i, i, vp, sz, vp]
    L.sr355_packed_weight_bytes.argtypes = [i, i, i, i]
    L.sr355_packed_weight_bytes.restype = sz
    L.sr355_pack_weight.argtypes = [vp, vp, i, i, i, i, vp, sz, vp, vp]
    L.sr355_conv3x3_fwd.argtypes = [C.POINTER(ConvArgs), vp]
    L.sr355_dense_block_fwd.argtypes = [C.POINTER(DenseBlockArgs), vp]
    L.sr355_pack_weight_transposed.argtypes = [vp, i, i, i, i, i, f, vp, i, i, sz, vp]
    L.sr355_pack_weights.argtypes = [C.POINTER(PackJob), i, i, vp]
    L.sr355_pack_weights_transposed.argtypes = [C.POINTER(PackTJob), i, i, vp]
    L.sr355_pool2x2_sum.argtypes = [vp, sz, i, i, i, i, i, vp, sz, i, i, f, vp, sz, vp, sz, i, i, vp]
    L.sr355_wgrad_workspace_bytes.argtypes = [i, i, i, i, i]
    L.sr355_wgrad_workspace_bytes.restype = sz
    L.sr355_conv3x3_wgrad.argtypes = [C.POINTER(WgradArgs), vp]
    L.sr355_conv3x3_wgrad_static.argtypes = [C.POINTER(WgradArgs), vp]
    L.sr355_dense_block_wgrad_workspace_bytes.argtypes = [i, i, i, i, i]
    L.sr355_dense_block_wgrad_workspace_bytes.restype = sz
    L.sr355_dense_block_wgrad.argtypes = [C.POINTER(DenseWgradArgs), vp]
    L.sr355_dense_blocks_wgrad.argtypes = [C.POINTER(DenseWgradArgs), i, vp]
    L.sr355_wgrad_f32_workspace_bytes.argtypes = [i, i, i, i, i]
    L.sr355_wgrad_f32_workspace_bytes.restype = sz
    L.sr355_conv3x3_wgrad_f32.argtypes = [C.POINTER(WgradArgs), vp]
    L.sr355_generator_num_convs.argtypes = [C.POINTER(GeneratorCfg)]
    L.sr355_generator_packed_bytes.argtypes = [C.POINTER(GeneratorCfg)]
    L.sr355_generator_packed_bytes.restype = sz
    L.sr355_generator_pack.argtypes = [C.POINTER(GeneratorCfg), C.POINTER(vp), C.POINTER(vp), i, vp, sz, vp]
    L.sr355_generator_workspace_bytes.argtypes = [C.POINTER(GeneratorCfg), i, i, i]
    L.sr355_generator_workspace_bytes.restype = sz
    L.sr355_generator_forward.argtypes = [C.POINTER(GeneratorCfg), vp, sz, vp, vp, i, i, i, vp, sz, vp]
    L.sr355_adam_step.argtypes = [vp, vp, vp, vp, sz, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, i, vp]
    L.sr355_generator_forward_u8.argtypes = [C.POINTER(GeneratorCfg), vp, sz, vp, vp, i, i, i, vp, sz, vp]
    L.sr355_chain_timing.argtypes = [i]
    L.sr355_chain_timing.restype = None
    L.sr355_chain_last_ms.argtypes = []
    L.sr355_chain_last_ms.restype = C.c_float
    L.sr355_chain_status.argtypes = [i]
    L.sr355_chain_reenable.argtypes = []
    L.sr355_chain_reenable.restype = None
    cfgp = C.POINTER(GeneratorCfg)
    for name in ("sr355_generator_saved_bytes", "sr355_generator_train_workspace_bytes",
                 "sr355_generator_bwd_workspace_bytes"):
        getattr(L, name).argtypes = [cfgp, i, i, i]
        getattr(L, name).restype = sz
    for name in ("sr355_generator_packed_bwd_bytes", "sr355_generator_grads_floats"):
        getattr(L, name).argtypes = [cfgp]
        getattr(L, name).restype = sz
    L.sr355_generator_forward_train.argtypes = [cfgp, vp, sz, vp, vp, i, i, i, vp, sz, vp, sz, vp]
    L.sr355_generator_pack_bwd.argtypes = [cfgp, C.POINTER(vp), i, vp, sz, vp]
    L.sr355_generator_backward.argtypes = [cfgp, vp, sz, vp, sz, vp, vp, sz, vp, i, i, i, vp, sz, vp]
    L.sr355_generator_backward_phases.argtypes = [cfgp]
    L.sr355_generator_backward_range.argtypes = [cfgp, vp, sz, vp, sz, vp, vp, sz, vp, i, i, i, vp, sz, i, i, vp]
    L.sr355_bn_workspace_bytes.argtypes = [i, i, i, i]
    L.sr355_bn_workspace_bytes.restype = sz
    L.sr355_bn_lrelu_fwd.argtypes = [vp, sz, i, i, i, i, vp, vp, f, f, i, f, vp, vp, vp, vp, vp, vp, sz, i, i, vp, sz, vp]
    L.sr355_bn_lrelu_bwd.argtypes = [vp, sz, vp, sz, i, i, i, i, i, i, vp, vp, vp, vp, f, i, vp, sz, i, i, vp, vp, vp, sz, vp]
    L.sr355_bn_lrelu_fwd_groups.argtypes = [vp, sz, i, i, i, i, vp, vp, f, f, i, f, vp, vp, vp, vp, vp, vp, sz, i, i, vp, sz, i, vp]
    L.sr355_bn_lrelu_bwd_groups.argtypes = [vp, sz, vp, sz, i, i, i, i, i, i, vp, vp, vp, vp, f, i, vp, sz, i, i, vp, vp, vp, sz,
                                            i, vp]
    L.sr355_maxpool2x2_fwd.argtypes = [vp, sz, i, i, i, i, vp, sz, vp]
    L.sr355_pixel_shuffle2.argtypes = [vp, sz, i, i, i, i, i, i, vp, sz, vp]
    L.sr355_maxpool2x2_bwd.argtypes = [vp, sz, vp, sz, i, i, i, i, vp, sz, vp]
    L.sr355_loss_workspace_bytes.argtypes = [sz]
    L.sr355_loss_workspace_bytes.restype = sz
    L.sr355_loss_fwd.argtypes = [vp, vp, f, sz, i, vp, vp, vp, sz, vp]
    if L.sr355_abi_version() != ABI_VERSION:
        raise Sr355Error(f"libsr355.so ABI {L.sr355_abi_version()} != expected {ABI_VERSION}; rebuild")
    _lib = L
    return L


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().sr355_last_error().decode("utf-8", "replace")
        raise Sr355Error(f"{what} failed ({rc}): {msg}")
