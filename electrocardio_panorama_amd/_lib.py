"""ctypes binding of libnefnet_hip.so (declared in include/nefnet_hip.h).

There is no CPU fallback: if the shared library is missing or fails to load, importing any compute
entry point raises.  Build it with `python -m electrocardio_panorama_amd.csrc.build`.
"""
import ctypes as C
import os

import torch  # noqa: F401  (first: the library must bind to the HIP runtime PyTorch-ROCm has already loaded)
from . import _env

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = _env.get("NEF_LIB") or os.path.join(_HERE, "csrc", "libnefnet_hip.so")   # NEF_LIB: A/B builds

NEF_OK = 0
OPT_H2_FORM, OPT_H2P_WGS = 1, 2      # nef_set_option keys (include/nefnet_hip.h)
_ERR = {-1: "NEF_E_SHAPE", -2: "NEF_E_NULL", -3: "NEF_E_WORKSPACE", -4: "NEF_E_UNSUPPORTED"}

p = C.c_void_p
i32 = C.c_int
i64 = C.c_int64
f32 = C.c_float
sz = C.c_size_t


class ConvArgs(C.Structure):
    """Mirror of `nef_conv_args` (include/nefnet_hip.h)."""
    _fields_ = [
        ("x", p), ("wp", p), ("y", p), ("bias", p), ("in_scale", p), ("res", p), ("gate", p), ("mask", p),
        ("x_bs", i64), ("x_gs", i64), ("y_bs", i64), ("y_gs", i64), ("sc_bs", i64), ("sc_gs", i64),
        ("res_bs", i64), ("res_gs", i64), ("gate_bs", i64), ("gate_gs", i64),
        ("B", i32), ("T", i32), ("G", i32), ("Cin_g", i32), ("Cout_g", i32), ("K", i32),
        ("relu", i32), ("gate_scale", f32), ("drop_scale", f32), ("drop_p", f32), ("rng_seed", C.c_uint64),
        ("pro_a", p), ("pro_b", p), ("pro_mode", i32), ("pro_Bp", i32), ("rng_seed_dev", p), ("wino", i32),
        ("stats", p),
        ("bnb_x", p), ("bnb_mean", p), ("bnb_invstd", p), ("bnb_a", p), ("bnb_b", p), ("bnb_slots", p), ("bnb_Bp", i32),
        ("bnb_up", i32), ("x_scale", f32), ("reserved0", i32), ("x_amax", p), ("x_amax_next", p), ("x_clamped", p),
        ("res_scale", p), ("rs_bs", i64), ("rs_gs", i64), ("gate_rowscale", p), ("gr_bs", i64), ("gr_gs", i64),
        ("stats_mode", i32), ("reserved1", i32),
    ]


class PackDesc(C.Structure):
    """Mirror of `nef_pack_desc` (include/nefnet_hip.h)."""
    _fields_ = [("w", p), ("wp", p), ("G", i32), ("Cog", i32), ("Cig", i32), ("K", i32), ("transpose_flip", i32),
                ("wino", i32), ("src_mode", i32), ("src_Cr", i32)]


# name -> (restype, argtypes); every symbol include/nefnet_hip.h declares
SIGNATURES = {
    "nef_abi_version": (i32, []),
    "nef_set_option": (i32, [i32, i32]),
    "nef_get_option": (i32, [i32]),
    "nef_stem_fwd": (i32, [p, p, p, i32, i32, i32, p]),
    "nef_stem_bwd_ws_bytes": (sz, [i32]),
    "nef_stem_bwd_weight": (i32, [p, p, p, p, p, sz, i32, i32, i32, p]),
    "nef_pack_weight": (i32, [p, p, i32, i32, i32, i32, i32, p]),
    "nef_pack_weight_wino": (i32, [p, p, i32, i32, i32, i32, i32, p]),
    "nef_pack_weight_wino4": (i32, [p, p, i32, i32, i32, i32, i32, p]),
    "nef_pack_weights": (i32, [C.POINTER(PackDesc), i32, p]),
    "nef_pack_weight_h2": (i32, [p, p, i32, i32, i32, i32, i32, p]),
    "nef_pack_weight_h2_bytes": (sz, [i32, i32, i32, i32, i32]),
    "nef_conv_fwd": (i32, [C.POINTER(ConvArgs), p]),
    "nef_conv_args_bytes": (sz, []),
    "nef_conv_bwd_weight_ws_bytes": (sz, [i32, i32, i32, i32, i32, i32]),
    "nef_conv_bwd_weight": (i32, [p, i64, i64, p, i64, i64, p, i64, i64, p, p, sz, i32, i32, i32, i32, i32, i32, p]),
    "nef_conv_bwd_weight_pro": (i32, [p, i64, i64, p, p, i32, i32, p, i64, i64, p, p, sz, i32, i32, i32, i32, i32, i32, p]),
    "nef_conv_bwd_weight_wino4": (i32, [p, i64, i64, p, i64, i64, p, p, i32, i32, p, i64, i64, p, p, sz, i32, i32, i32, i32, i32,
                                        i32, p]),
    "nef_conv_bwd_weight_h2_ws_bytes": (sz, [i32, i32, i32, i32, i32, i32]),
    "nef_conv_bwd_weight_h2": (i32, [p, i64, i64, p, i64, i64, p, p, i32, i32, p, i64, i64, p, p, sz, i32, i32, i32, i32, i32, i32,
                               f32, f32, p, p, p, p, p, p]),
    "nef_chan_sum_ws_bytes": (sz, [i32]),
    "nef_chan_sum": (i32, [p, p, p, sz, i32, i32, i32, p]),
    "nef_convt2_fwd": (i32, [p, p, p, p, i32, i32, i32, i32, i32, p]),
    "nef_convt2_bwd_data": (i32, [p, p, p, i32, i32, i32, i32, i32, p]),
    "nef_group_transpose": (i32, [p, p, i32, i32, i32, p]),
    "nef_convt2_interleave": (i32, [p, p, p, i32, i32, i32, p]),
    "nef_convt2_deinterleave": (i32, [p, p, i32, i32, i32, p]),
    "nef_convt2_bwd_weight_ws_bytes": (sz, [i32, i32, i32]),
    "nef_convt2_bwd_weight": (i32, [p, p, p, p, p, sz, i32, i32, i32, i32, i32, p]),
    "nef_theta_mlp_fwd": (i32, [p, p, p, p, i32, i32, p]),
    "nef_theta_mlp_bwd": (i32, [p, p, p, p, i32, i32, p]),
    "nef_theta_encode": (i32, [p, p, i32, p]),
    "nef_chscale_fwd": (i32, [p, p, i64, p, i32, i32, i32, p]),
    "nef_chscale_bwd": (i32, [p, p, p, i64, p, p, i32, i32, i32, i32, p]),
    "nef_gate": (i32, [p, p, p, f32, i64, p]),
    "nef_add": (i32, [p, p, p, i64, p]),
    "nef_roi_align_fwd": (i32, [p, p, p, i32, i32, i32, i32, i32, p]),
    "nef_roi_align_bwd": (i32, [p, p, p, i32, i32, i32, i32, i32, p]),
    "nef_window_crop": (i32, [p, i64, i64, p, i32, i32, i32, i32, i32, i32, p]),
    "nef_window_scatter": (i32, [p, p, i64, i64, i32, i32, i32, i32, i32, i32, p]),
    "nef_roi_unpool_fwd": (i32, [p, p, p, p, i32, i32, i32, p]),
    "nef_roi_unpool_bwd": (i32, [p, p, p, i32, i32, i32, p]),
    "nef_roi_segment_table": (i32, [p, p, p, i32, p]),
    "nef_lead_mean": (i32, [p, p, p, i32, i32, i32, p]),
    "nef_mix_fwd": (i32, [p, p, p, p, p, i32, i32, i32, i32, i32, p, p]),
    "nef_mix_bwd": (i32, [p, p, p, p, p, p, p, p, i32, i32, i32, i32, i32, p, i32, p]),
    "nef_mix_fwd_shared": (i32, [p, p, p, p, p, i32, i32, i32, i32, i32, p, p]),
    "nef_lead_mean_mix_shared": (i32, [p, p, p, p, p, i32, i32, i32, i32, i32, p, p]),
    "nef_mix_bwd_shared_up": (i32, [p, p, p, p, p, p, p, p, i32, i32, i32, i32, i32, p, i32, p]),
    "nef_mix_bwd_shared": (i32, [p, p, p, p, p, p, p, p, i32, i32, i32, i32, i32, p, i32, p]),
    "nef_pass_combine_fwd": (i32, [p, p, p, i32, i32, i32, p]),
    "nef_pass_combine_bwd": (i32, [p, p, i32, i32, i32, p]),
    "nef_pass_combine_stats_ws_bytes": (sz, [i32, i32]),
    "nef_pass_combine_fwd_stats": (i32, [p, p, p, p, p, p, p, p, p, p, p, p, sz, i32, i32, i32, f32, f32, p, p]),
    "nef_mix_bwd_up": (i32, [p, p, p, p, p, p, p, p, i32, i32, i32, i32, i32, p, i32, p]),
    "nef_upsample2_fwd": (i32, [p, p, i64, i32, p]),
    "nef_upsample2_bwd": (i32, [p, p, i64, i32, p]),
    "nef_upsample2_aff_fwd": (i32, [p, p, p, p, i32, i32, i32, i32, p]),
    "nef_bn_ws_bytes": (sz, [i32, i32]),
    "nef_bn_train_stats": (i32, [p, p, p, p, p, p, p, p, p, p, sz, i32, i32, i32, i32, f32, f32, p, p]),
    "nef_conv_stats_slots": (i32, [i32, i32]),
    "nef_bn_stats_from_slots": (i32, [p, i32, p, p, p, p, p, p, p, p, p, sz, i32, i32, i32, i32, f32, f32, p, p]),
    "nef_bn_eval_affine": (i32, [p, p, p, p, p, p, i32, f32, p]),
    "nef_fold_bn": (i32, [p, p, p, p, p, p, i32, i32, p]),
    "nef_affine_relu_fwd": (i32, [p, p, p, p, i32, i32, i32, i32, p]),
    "nef_bn_bwd_ws_bytes": (sz, [i32, i32, i32]),
    "nef_bn_relu_bwd": (i32, [p, p, p, p, p, p, p, p, p, p, p, p, sz, i32, i32, i32, i32, p, i32, p]),
    "nef_bn_relu_bwd_phase_major": (i32, [p, p, p, p, p, p, p, p, p, p, p, p, sz, i32, i32, i32, i32, p, i32, p]),
    "nef_bn_relu_bwd_up": (i32, [p, p, p, p, p, p, p, p, p, p, p, sz, i32, i32, i32, i32, p, i32, p]),
    "nef_bn_relu_bwd_combine3": (i32, [p, p, p, p, p, p, p, p, p, p, p, sz, i32, i32, i32, p, i32, p]),
    "nef_bn_relu_bwd_combine3_phase_major": (i32, [p, p, p, p, p, p, p, p, p, p, p, sz, i32, i32, i32, p, i32, p]),
    "nef_bn_bwd_outconv_ws_bytes": (sz, [i32, i32, i32, i32]),
    "nef_bn_relu_bwd_outconv": (i32, [p, p, p, p, p, p, p, p, p, p, p, p, p, sz, i32, i32, i32, i32, p]),
    "nef_outconv_fwd": (i32, [p, p, p, p, i32, i32, i32, p]),
    "nef_outconv_fwd_pro": (i32, [p, p, p, i32, p, p, p, i32, i32, i32, p]),
    "nef_outconv_bwd_weight_pro": (i32, [p, p, p, p, p, i32, p, p, p, sz, i32, i32, i32, p]),
    "nef_outconv_bwd_data": (i32, [p, p, p, p, i32, i32, i32, p]),
    "nef_outconv_bwd_weight_ws_bytes": (sz, [i32]),
    "nef_outconv_bwd_weight": (i32, [p, p, p, p, p, p, sz, i32, i32, i32, p]),
    "nef_loss_ws_bytes": (sz, []),
    "nef_loss_fwd": (i32, [p, p, p, p, p, p, sz, i64, f32, f32, f32, i32, i32, p]),
    "nef_loss_bwd": (i32, [p, p, p, p, p, p, p, p, i64, f32, f32, f32, i32, i32, p]),
    "nef_sgd_momentum": (i32, [p, p, p, i64, f32, f32, f32, i32, p, p, p, p]),
    "nef_h2_taint": (i32, [p, p, p, p]),
    "nef_amax_roll": (i32, [p, p, i32, f32, f32, i32, p]),
    "nef_step_words": (i32, [p, p, i32, i32, i64, p]),
    "nef_flatten": (i32, [C.POINTER(p), C.POINTER(i64), i32, p, p]),
    "nef_regroup_halves": (i32, [p, p, i32, i32, i32, i32, p]),
    "nef_slots_to_rows": (i32, [p, i32, p, i32, i32, p]),
    "nef_poly_weights": (i32, [p, p, i32, i32, i32, p]),
    "nef_poly_fwd_edge": (i32, [p, p, p, i32, i32, i32, i32, i32, p, p, i32, p, i32, p, p]),
    "nef_poly_wgrad_fold_ws_bytes": (sz, [i32, i32, i32, i32]),
    "nef_poly_wgrad_fold": (i32, [p, p, p, p, p, sz, i32, i32, i32, i32, i32, p]),
    "nef_poly_bwd_edge": (i32, [p, p, p, i32, i32, i32, i32, i32, p, p, p, p, p, i32, p, i32, i32, p]),
    "nef_view_metrics": (i32, [p, p, p, p, p, i32, i32, i32, p]),
    "nef_pano_h_from_f32": (i32, [p, p, i32, i32, i32, p]),
    "nef_pano_h_pack_weight": (i32, [p, p, i32, i32, p]),
    "nef_pano_h_conv": (i32, [p, p, p, p, p, i32, i32, i32, i32, i32, i32, i32, i64, i64, p]),
    "nef_pano_h_conv_pair": (i32, [p, p, p, p, p, p, p, i32, i32, i32, i32, i64, i64, p]),
    "nef_pano_h_conv_outconv": (i32, [p, p, p, p, p, p, i32, i32, i32, i64, i64, p]),
    "nef_pano_h_conv_tail": (i32, [p, p, p, p, p, p, p, p, i32, i32, i32, i64, i64, p]),
    "nef_pano_h_outconv": (i32, [p, p, p, p, i32, i32, i32, i64, i64, p]),
}

_lib = None


class NefLibraryError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle with typed entry points."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NefLibraryError(
            f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). "
            "Build it with `python -m electrocardio_panorama_amd.csrc.build`.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)        # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    lib.nef_debug_spin_us.restype, lib.nef_debug_spin_us.argtypes = i32, [f32, i32, p]      # diagnostics, not in the header
    if lib.nef_conv_args_bytes() != C.sizeof(ConvArgs):      # a stale .so next to a newer binding (or the reverse)
        raise NefLibraryError(f"{LIB_PATH}: nef_conv_args is {lib.nef_conv_args_bytes()} bytes, the binding mirrors "
                              f"{C.sizeof(ConvArgs)}; rebuild with `python -m electrocardio_panorama_amd.csrc.build`")
    _lib = lib
    return lib


def check(rc, what=""):
    if rc == NEF_OK:
        return
    if rc < 0:
        raise NefLibraryError(f"{what}: {_ERR.get(rc, rc)}")
    raise NefLibraryError(f"{what}: hipError_t {rc}")
