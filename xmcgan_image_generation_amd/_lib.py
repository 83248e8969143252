"""ctypes binding of ``libxmcgan_hip.so`` (the C ABI declared in ``include/xmcgan_hip.h``).

The product path has NO fallback: if the shared library is missing or a symbol is absent,
loading fails loudly (``build()`` in ``__graft_entry__.py`` / ``make -C csrc`` produce it).
"""
from __future__ import annotations

import ctypes as C
import os

# torch ships its own HIP runtime (torch/lib/libamdhip64.so).  It must be the FIRST HIP runtime
# mapped into the process, otherwise libxmcgan_hip.so binds to the system one and launches kernels
# on a runtime that shares no device context with torch's allocator ("hipErrorNoDevice").
import torch  # noqa: F401  (ordering matters)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libxmcgan_hip.so")
PROBE_LIB_PATH = os.path.join(_HERE, "libxmc_probe.so")

XMC_F32, XMC_BF16 = 0, 1
ABI_VERSION = 22


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("n", "hi", "wi", "cin", "cout", "ks", "ups", "relu_in", "res_ups", "out_f32", "dtype")] + \
               [("alpha", C.c_float), ("res_scale", C.c_float), ("w_packed", C.c_int32), ("pool_out", C.c_int32),
                ("relu_out", C.c_int32), ("mask_after_res", C.c_int32), ("valid_h", C.c_int32), ("valid_w", C.c_int32),
                ("alpha_dev", C.c_void_p)]


class WgradDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("n", "hi", "wi", "cin", "cout", "ks", "x_ups", "x_relu", "dy_ups", "dtype", "variant")] + \
               [("alpha", C.c_float)]


class SnEntry(C.Structure):          # mirrors xmc_sn_entry
    _fields_ = [("w_off", C.c_int64), ("rows", C.c_int32), ("cols", C.c_int32), ("u_axis", C.c_int32),
                ("u_off", C.c_int32), ("v_off", C.c_int32), ("blk_a", C.c_int32), ("blk_b", C.c_int32),
                ("taps", C.c_int32), ("is_conv", C.c_int32), ("wf_off", C.c_int64), ("wd_off", C.c_int64),
                ("blk_p", C.c_int32), ("packed", C.c_int32)]


class WprepEntry(C.Structure):       # mirrors xmc_wprep_entry
    _fields_ = [("w_off", C.c_int64), ("wf_off", C.c_int64), ("wd_off", C.c_int64), ("pf_off", C.c_int64), ("pd_off", C.c_int64),
                ("part_off", C.c_int64), ("cout", C.c_int32), ("cin", C.c_int32), ("taps", C.c_int32), ("blk0", C.c_int32),
                ("flags", C.c_int32), ("u_off", C.c_int32), ("v_off", C.c_int32), ("blk_c", C.c_int32)]


_P, _I, _L, _F = C.c_void_p, C.c_int32, C.c_int64, C.c_float

# name -> argtypes (every entry point returns int); must list every symbol of include/xmcgan_hip.h
SIGNATURES = {
    "xmc_abi_version": [],
    "xmc_create": [_I, C.POINTER(_P)],
    "xmc_destroy": [_P],
    "xmc_handle_device": [_P],
    "xmc_set_tuning": [C.c_char_p, _I],
    "xmc_get_tuning": [C.c_char_p, C.POINTER(_I)],
    "xmc_conv2d_nhwc": [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P],
    "xmc_conv2d_wgrad": [C.POINTER(WgradDesc), _P, _P, _P, _P, _P],
    "xmc_conv2d_workspace_bytes": [C.POINTER(ConvDesc)],
    "xmc_conv2d_phase_supported": [C.POINTER(ConvDesc)],
    "xmc_conv2d_wgrad_workspace_bytes": [C.POINTER(WgradDesc)],
    "xmc_conv2d_wgrad_ws": [C.POINTER(WgradDesc), _P, _P, _P, _P, _P, _L, _P],
    "xmc_reduce_mid_ws_floats": [_L, _L, _L],
    "xmc_reduce_mid_ws": [_P, _P, _P, _L, _L, _L, _I, _I, _F, _I, _P],
    "xmc_conv2d_nhwc_ws": [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P],
    "xmc_conv2d_nhwc_bits": [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "xmc_prep_conv_weight": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "xmc_pack_conv_weight": [_P, _P, _I, _I, _I, _P],
    "xmc_gemm_ws_floats": [_I, _I, _I, _I, _I],
    "xmc_gemm_f32": [_P, _P, _P, _I, _I, _I, _L, _L, _L, _L, _L, _L, _L, _L, _F, _P, _F, _I, _P, _P],
    "xmc_gemm_f32_bf16mfma": [_P, _P, _P, _I, _I, _I, _L, _L, _L, _L, _L, _L, _L, _L, _F, _P, _F, _I, _P, _P],
    "xmc_reduce_mid": [_P, _P, _L, _L, _L, _I, _I, _F, _I, _P],
    "xmc_bn_stats": [_P, _P, _L, _I, _I, _P],
    "xmc_bn_finalize": [_P, _P, _P, _P, _P, _L, _I, _F, _F, _I, _P],
    "xmc_bn_from_running": [_P, _P, _P, _P, _I, _F, _P],
    "xmc_cbn_act_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "xmc_cbn_act_bwd_cells": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "xmc_cbn_bwd_sums_ws_floats": [_L, _I],
    "xmc_cbn_bwd_sums": [_P, _P, _P, _P, _P, _L, _I, _I, _I, _P],
    "xmc_bn_stats_ws_floats": [_L, _I],
    "xmc_bn_batch_stats": [_P, _P, _P, _P, _P, _P, _L, _I, _I, _F, _F, _I, _P],
    "xmc_cbn_act_bwd_dx": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "xmc_pool2": [_P, _P, _P, _I, _I, _I, _I, _F, _I, _P],
    "xmc_pool2_relu": [_P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P],
    "xmc_expand_taps": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "xmc_bcast_relu_bwd": [_P, _P, _P, _L, _L, _L, _I, _P],
    "xmc_tanh_out_fwd": [_P, _P, _L, _I, _P],
    "xmc_tanh_out_bwd": [_P, _P, _P, _L, _I, _P],
    "xmc_cast": [_P, _I, _P, _I, _L, _P],
    "xmc_add": [_P, _P, _P, _L, _I, _P],
    "xmc_attn_g_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P],
    "xmc_attn_g_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P],
    "xmc_l2norm_rows_fwd": [_P, _P, _P, _L, _I, _I, _P],
    "xmc_l2norm_rows_bwd": [_P, _P, _P, _P, _L, _I, _I, _P],
    "xmc_wl_softmax": [_P, _P, _P, _P, _I, _I, _I, _F, _P],
    "xmc_wl_qdot": [_P, _P, _P, _I, _I, _I, _P],
    "xmc_wl_rows": [_P, _P, _P, _P, _P, _I, _I, _F, _F, _P],
    "xmc_wl_bwd_cols": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _F, _P],
    "xmc_wl_fused_supported": [_I, _I, _I, _I],
    "xmc_wl_fused_ldp": [_I, _I],
    "xmc_wl_prep_regions": [_P, _P, _P, _P, _I, _I, _I, _P],
    "xmc_wl_prep_words": [_P, _P, _P, _I, _I, _I, _P],
    "xmc_wl_tn_gemm": [_P, _L, _I, _P, _L, _I, _I, _P, _L, _I, _P, _L, _I, _I, _P, _L, _I, _I, _F, _I, _I, _I, _P],
    "xmc_wl_cols_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P],
    "xmc_wl_cols_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _P],
    "xmc_l2norm_rows_bwd_bf16y": [_P, _P, _P, _P, _L, _I, _I, _P],
    "xmc_xent_sym": [_P, _I, _F, _P, _P, _P, _P],
    "xmc_loss_assemble": [_P, _P, _P, _P],
    "xmc_cl_logits": [_P, _P, _P, _P, _P, _I, _I, _F, _P],
    "xmc_cl_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _F, _I, _I, _P],
    "xmc_hinge": [_P, _I, _P, _P, _P, _P, _P],
    "xmc_proj_head_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "xmc_proj_head_bwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "xmc_spectral_power_iter": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P],
    "xmc_spectral_grad_fix": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "xmc_sn_batched_power_iter": [_P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P],
    "xmc_conv2d_pw_dual": [C.POINTER(ConvDesc), _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P],
    "xmc_sn_batched_prep": [_P, _I, _P, _P, _P, _P, _I, _I, _P],
    "xmc_sn_batched_grad_fix": [_P, _I, _P, _P, _P, _P, _P, _P, _I, _P],
    "xmc_adam_ema": [_P, _P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _F, _F, _F, _P],
    "xmc_adam_ema_dev": [_P, _P, _P, _P, _P, _L, _F, C.c_double, C.c_double, _F, _P, _F, _F, _P],
    "xmc_resize_bilinear": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "xmc_stem_im2col": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "xmc_stem_conv7x7s2": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "xmc_stem_conv7x7s2_dgrad": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "xmc_maxpool3x3s2": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "xmc_maxpool3x3s2_bwd": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "xmc_zero_margin": [_P, _I, _I, _I, _I, _I, _I, _I, _P],
    "xmc_subsample2": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "xmc_add_relu": [_P, _P, _P, _L, _I, _P],
    "xmc_relu_bwd": [_P, _P, _P, _P, _L, _I, _P],
    "xmc_phase_conv_weight": [_P, _P, _P, _P, _I, _I, _I, _P],
    "xmc_cbn_act_fwd_mx8": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "xmc_mx8_quantize": [_P, _P, _L, _I, _I, _P],
    "xmc_mx8_pack_conv_weight": [_P, _P, _P, _I, _I, _I, _P],
    "xmc_conv2d_mx8_workspace_bytes": [C.POINTER(ConvDesc)],
    "xmc_conv2d_mx8": [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P],
    "xmc_conv2d_mx8_bits": [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P],
    "xmc_mx8_probe": [_P, _P, _P, _P, _P, _P],
    "xmc_attn_g_mfma_supported": [_I, _I, _I, _I],
    "xmc_attn_g_fwd_mfma": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P],
    "xmc_attn_g_bwd_mfma": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P],
    "xmc_attn_g_fwd_mfma_ld": [_P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _F, _P],
    "xmc_attn_g_bwd_mfma_ld": [_P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P],
    "xmc_wprep_batched": [_P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P],
    "xmc_sn_power_iter_fused": [_P, _I, _I, _I, _P, _I, _I, _I, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _F, _P],
    "xmc_sn_batched_dot": [_P, _I, _P, _P, _P, _P, _P, _I, _P],
    "xmc_adam_wprep_tiles": [_P, _I, _I, _P, _P, _P, _P, _P, _F, C.c_double, C.c_double, _F, _P, _F, _F, _I, _P, _P, _P, _P, _P, _P, _P, _P,
                             _P, _P],
    "xmc_adam_ema_dev_sn": [_P, _P, _P, _P, _P, _L, _F, C.c_double, C.c_double, _F, _P, _F, _F, _I, _P, _P, _I, _P, _P, _P, _P, _P],
}

# diagnostic probes: include/xmc_probe.h, libxmc_probe.so (csrc_probe/) -- outside the product ABI
PROBE_SIGNATURES = {
    "xmc_probe_layouts": [_P, _P],
    "xmc_mfma_rate_probe": [_I, _I, _I, _P, _P],
    "xmc_load_path_probe": [_I, _I, _I, _P, _L, _P, _P],
    "xmc_delay": [_I, _P],
    "xmc_pk_add_cross_probe": [_I, _I, _I, _P, _P],
    "xmc_class_neighbour": [_I, _I, _I, _P, _L, _P, _P],
}

_INT64_RETURNS = ("xmc_conv2d_mx8_workspace_bytes", "xmc_conv2d_workspace_bytes", "xmc_bn_stats_ws_floats", "xmc_cbn_bwd_sums_ws_floats",
                  "xmc_conv2d_wgrad_workspace_bytes", "xmc_reduce_mid_ws_floats", "xmc_gemm_ws_floats")
_lib = None


class XmcError(RuntimeError):
    pass


def load():
    """Load the shared library once; raise (never fall back) when it cannot be used."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise XmcError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C xmcgan_image_generation_amd/csrc`.  There is no CPU / PyTorch fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise XmcError(f"{LIB_PATH} does not export {name}; rebuild the library") from e
        fn.argtypes = argtypes
        fn.restype = C.c_int64 if name in _INT64_RETURNS else C.c_int
    if lib.xmc_abi_version() != ABI_VERSION:
        raise XmcError("libxmcgan_hip.so ABI version mismatch; rebuild the library")
    _lib = lib
    return lib


_probe = None


def load_probe():
    """The diagnostic probes (tests, bench.py's instrumented step, tools/): a separate library, never needed by a training step"""
    global _probe
    if _probe is not None:
        return _probe
    if not os.path.exists(PROBE_LIB_PATH):
        raise XmcError(f"{PROBE_LIB_PATH} not found: build it with `make -C xmcgan_image_generation_amd/csrc_probe`")
    lib = C.CDLL(PROBE_LIB_PATH)
    for name, argtypes in PROBE_SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    _probe = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        if rc <= -1000:
            raise XmcError(f"{what}: HIP error {-rc - 1000}")
        raise XmcError(f"{what}: invalid argument (rc={rc})")
