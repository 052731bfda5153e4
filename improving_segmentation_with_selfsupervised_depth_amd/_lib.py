"""ctypes binding of libsegsde_hip.so (C ABI declared in include/segsde_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` (hipcc, gfx950).  There is NO fallback: if
the shared object is missing, or a tensor that is not on the GPU reaches an op, the call raises.
"""
import ctypes
import os
from ctypes import c_int, c_long, c_float, c_void_p, c_size_t, c_uint64, c_int64, POINTER

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SEGSDE_LIB") or os.path.join(_HERE, "libsegsde_hip.so")   # override: kernel experiments
ABI_VERSION = 13

_LIB = None
# Set only by the test-suite when it injects the host-interpreted build of the same kernel sources
# (see tests/emu.py); the product never sets it.
HOST_POINTERS_OK = False


class PackJob(ctypes.Structure):
    """mirror of ``segsde_pack_job`` (include/segsde_hip.h)"""
    _fields_ = [("w", c_void_p), ("fwd", c_void_p), ("dgrad", c_void_p)] + [(n, c_int) for n in (
        "O", "I", "KH", "KW", "block0", "reserved")]


class WinoJob(ctypes.Structure):
    """mirror of ``segsde_wino_job`` (include/segsde_hip.h)"""
    _fields_ = [("w", c_void_p), ("u_fwd", c_void_p), ("u_dgrad", c_void_p)] + [(n, c_int) for n in ("O", "I", "block0", "reserved")]


class ConvDesc(ctypes.Structure):
    """mirror of ``segsde_conv_desc`` (include/segsde_hip.h)"""
    _fields_ = [(n, c_int) for n in (
        "B", "H", "W", "C0", "C1", "ld0", "ld1", "up0", "Ho", "Wo", "Cout", "ldy", "ldy2", "nsplit",
        "KH", "KW", "stride", "dil", "pad", "pad_mode", "in_div", "act", "sum2x2", "accumulate", "compute")]


P = c_void_p
_SIGS = {
    "segsde_abi_version": (c_int, []),
    "segsde_conv2d_forward": (c_int, [POINTER(ConvDesc), P, P, P, P, P, P, P]),
    "segsde_conv2d_stats_rows": (c_long, [POINTER(ConvDesc)]),
    "segsde_conv2d_forward_stats": (c_int, [POINTER(ConvDesc), P, P, P, P, P, P, P, P]),
    "segsde_conv2d_dgrad_actgrad": (c_int, [POINTER(ConvDesc), P, P, P, P, P, P, P, P, c_int, c_int, P]),
    "segsde_bn_stats_from_partials_workspace": (c_size_t, [c_int]),
    "segsde_bn_stats_from_partials": (c_int, [P, c_long, c_long, c_int, P, P, P, P, c_float, c_float, P, P, c_size_t, P]),
    "segsde_conv2d_wgrad_workspace": (c_size_t, [POINTER(ConvDesc)]),
    "segsde_conv2d_wgrad": (c_int, [POINTER(ConvDesc), P, P, P, c_int, P, P, c_size_t, P]),
    "segsde_conv2d_winograd_workspace": (c_size_t, [POINTER(ConvDesc)]),
    "segsde_conv2d_winograd_stats_rows": (ctypes.c_long, [POINTER(ConvDesc)]),
    "segsde_winograd_pack": (c_int, [P, c_int, c_int, P, P, P]),
    "segsde_winograd_pack_multi": (c_int, [P, c_int, c_int, P]),
    "segsde_conv2d_winograd": (c_int, [POINTER(ConvDesc), P, P, P, P, P, P, P, P, c_size_t, P]),
    "segsde_conv2d_wgrad_winograd_workspace": (c_size_t, [POINTER(ConvDesc)]),
    "segsde_conv2d_wgrad_winograd": (c_int, [POINTER(ConvDesc), P, P, P, c_int, P, P, P, c_size_t, P]),
    "segsde_winograd_fused_ok": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "segsde_winograd_fused_stats_rows": (ctypes.c_long, [c_int, c_int, c_int]),
    "segsde_winograd_fused_pack": (c_int, [P, c_int, c_int, c_int, P, P]),
    "segsde_conv2d_winograd_fused": (c_int, [P, c_int, c_int, c_int, c_int, c_int, c_int, P, c_int, P, c_int, P, c_int, c_int, P, P]),
    "segsde_conv2d_winograd_fused2": (c_int, [P, c_int, c_int, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, P, c_int, P, c_int, P,
                                              c_int, P, P]),
    "segsde_conv2d_winograd_fused_dgrad": (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, c_int, c_int, P, c_int, c_int, P, c_int, c_int, P]),
    "segsde_reflect_adjoint_borders": (c_int, [P, P, P, P, P, c_int, c_int, P]),
    "segsde_reflect_adjoint_borders_ok": (c_int, [P, c_int]),
    "segsde_reflect_adjoint_borders2": (c_int, [P, c_int, P, c_int, P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "segsde_conv2d_wgrad_winograd_fused_workspace": (c_size_t, [P]),
    "segsde_conv2d_wgrad_winograd_fused": (c_int, [P, P, P, P, c_int, P, P, c_size_t, P]),
    "segsde_upfold_pack": (c_int, [P, c_int, c_int, c_int, P, P, P]),
    "segsde_conv2d_forward_upfold": (c_int, [POINTER(ConvDesc), P, P, P, P, P, P, P]),
    "segsde_conv2d_dgrad_upfold": (c_int, [POINTER(ConvDesc), P, c_int, P, P, P, P, P, c_int, P, c_int, c_int, P]),
    "segsde_conv2d_wgrad_upfold_workspace": (c_size_t, [POINTER(ConvDesc)]),
    "segsde_conv2d_wgrad_upfold": (c_int, [POINTER(ConvDesc), P, P, P, c_int, P, P, c_size_t, P]),
    "segsde_stem_pack": (c_int, [P, c_int, c_int, c_int, P, P]),
    "segsde_stem7x7_stats_rows": (c_long, [c_int, c_int, c_int, c_int, c_int]),
    "segsde_stem7x7_forward": (c_int, [P, c_int, c_int, c_int, c_int, P, c_int, P, P, P]),
    "segsde_stem7x7_wgrad_workspace": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "segsde_stem7x7_wgrad": (c_int, [P, c_int, c_int, c_int, c_int, P, c_int, c_int, c_int, P, P, c_size_t, P]),
    "segsde_nchw_to_nhwc_bordered": (c_int, [P, c_int, c_int, c_int, c_int, c_float, c_float, P, c_int, c_int, c_int, c_int, c_int, P]),
    "segsde_pack_weight": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "segsde_pack_weight_both": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P]),
    "segsde_pack_weight_both_multi": (c_int, [P, c_int, c_int, P]),
    "segsde_reflect_dgrad_fix": (c_int, [P, c_int, P, P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "segsde_bn_stats_workspace": (c_size_t, [c_long, c_int]),
    "segsde_bn_stats": (c_int, [P, c_int, c_long, c_int, P, P, P, P, c_float, c_float, P, P, c_size_t, P]),
    "segsde_bn_eval_stats": (c_int, [P, P, c_int, c_float, P, P, P]),
    "segsde_bn_apply": (c_int, [P, c_int, c_long, c_int, P, P, P, P, P, c_int, P, c_int, c_int, c_float, c_uint64, P]),
    "segsde_bn_backward_workspace": (c_size_t, [c_long, c_int]),
    "segsde_bn_backward": (c_int, [P, c_int, P, c_int, P, c_int, c_long, c_int, P, P, P, P, c_int, c_float, c_uint64, c_int,
                                   P, P, P, c_int, P, c_int, P, c_size_t, P]),
    "segsde_colsum_workspace": (c_size_t, [c_long, c_int]),
    "segsde_act_backward": (c_int, [P, c_int, P, c_int, c_long, c_int, c_int, P, c_int, P, P, c_size_t, P]),
    "segsde_colsum": (c_int, [P, c_int, c_long, c_int, P, P, c_size_t, P]),
    "segsde_maxpool3x3s2_forward": (c_int, [P, c_int, c_int, c_int, c_int, P, P, P]),
    "segsde_maxpool3x3s2_backward": (c_int, [P, P, c_int, c_int, c_int, c_int, P, c_int, P]),
    "segsde_upsample2x_backward": (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, c_int, P]),
    "segsde_upsample2x_forward": (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, c_int, P]),
    "segsde_resize_bilinear_forward": (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, c_int, c_int, c_int, c_int, P]),
    "segsde_resize_bilinear_backward": (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, c_int, c_int, c_int, c_int, P]),
    "segsde_global_avgpool_workspace": (c_size_t, [c_int, c_long, c_int]),
    "segsde_global_avgpool_forward": (c_int, [P, c_int, c_int, c_long, c_int, P, P, c_size_t, P]),
    "segsde_global_avgpool_backward": (c_int, [P, c_int, c_long, c_int, P, c_int, P]),
    "segsde_gate_forward": (c_int, [P, P, c_long, P, P]),
    "segsde_gate_backward": (c_int, [P, P, P, c_long, P, P, P]),
    "segsde_axpby": (c_int, [c_long, c_float, P, c_float, P, P, P]),
    "segsde_axpby_dev": (c_int, [c_long, P, P, P, P, P, P]),
    "segsde_scale_channels": (c_int, [P, c_int, c_int, c_long, c_int, P, P, c_int, P]),
    "segsde_dropout": (c_int, [P, c_int, c_long, c_int, c_float, c_uint64, P, c_int, P]),
    "segsde_copy_channels": (c_int, [P, c_int, P, c_int, c_long, c_int, P]),
    "segsde_nchw_to_nhwc": (c_int, [P, c_int, c_int, c_int, c_int, c_float, c_float, P, c_int, P]),
    "segsde_nhwc_to_nchw": (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, P]),
    "segsde_pose_matrix_forward": (c_int, [P, P, c_int, c_int, c_int, P, P]),
    "segsde_pose_matrix_backward": (c_int, [P, P, P, c_int, c_int, c_int, P, P, P]),
    "segsde_warp_forward": (c_int, [P, c_int, c_int, P, P, P, P, c_int, c_int, c_int, c_float, c_float, P, P, P, P]),
    "segsde_warp_backward_workspace": (c_size_t, [c_int, c_int, c_int]),
    "segsde_warp_backward": (c_int, [P, P, c_int, c_int, P, P, P, P, c_int, c_int, c_int, c_float, c_float, P, P, P,
                                     c_size_t, P]),
    "segsde_reprojection_error_forward": (c_int, [P, P, c_int, c_int, c_int, c_int, P, c_long, P]),
    "segsde_reprojection_error_backward_workspace": (c_size_t, [c_int, c_int, c_int]),
    "segsde_reprojection_error_backward": (c_int, [P, P, P, c_long, c_int, c_int, c_int, c_int, P, P, c_size_t, P]),
    "segsde_photometric_workspace": (c_size_t, [c_int, c_int, c_int]),
    "segsde_photometric_identity": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P, P]),
    "segsde_photometric_forward": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, P, P, c_size_t, P]),
    "segsde_photometric_backward": (c_int, [P, P, P, P, c_int, P, c_int, c_int, P, P, P, P, P, P, c_int, c_int, c_int,
                                            c_float, c_float, c_int, c_int, c_float, P, P, P, P, P, c_size_t, P]),
    "segsde_automask_workspace": (c_size_t, [c_int, c_int, c_int]),
    "segsde_automask_min_forward": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, P, P, c_size_t, P]),
    "segsde_automask_min_backward": (c_int, [P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, P, P]),
    "segsde_smoothness_workspace": (c_size_t, [c_int, c_int, c_int]),
    "segsde_smoothness_forward": (c_int, [P, P, c_int, c_int, c_int, P, P, P, c_size_t, P]),
    "segsde_smoothness_backward": (c_int, [P, P, P, c_int, c_int, c_int, c_float, P, P, c_size_t, P]),
    "segsde_smooth_loss_workspace": (c_size_t, [c_int, c_int, c_int]),
    "segsde_smooth_loss_forward": (c_int, [P, P, c_int, c_int, c_int, P, P, c_size_t, P]),
    "segsde_smooth_loss_backward": (c_int, [P, P, c_int, c_int, c_int, c_float, P, P, c_size_t, P]),
    "segsde_ssim_map_forward": (c_int, [P, P, c_int, c_int, c_int, c_int, P, P]),
    "segsde_ssim_map_backward": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P, P, P]),
    "segsde_backproject_depth": (c_int, [P, P, c_int, c_int, c_int, P, P]),
    "segsde_project3d": (c_int, [P, P, P, c_int, c_int, c_int, c_float, P, P]),
    "segsde_backproject_depth_backward": (c_int, [P, P, c_int, c_int, c_int, P, P]),
    "segsde_project3d_backward_workspace": (c_size_t, [c_int, c_int, c_int]),
    "segsde_project3d_backward": (c_int, [P, P, P, P, c_int, c_int, c_int, c_float, P, P, P, c_size_t, P]),
    "segsde_cross_entropy_workspace": (c_size_t, [c_long]),
    "segsde_cross_entropy_forward": (c_int, [P, c_int, c_long, c_int, P, c_int64, P, P, P, P, c_size_t, P]),
    "segsde_cross_entropy_backward": (c_int, [P, c_int, c_long, c_int, P, c_int64, P, P, P, P, c_int, P]),
    "segsde_mix": (c_int, [P, c_int, c_int, P, c_int, c_int, c_int, c_int, c_long, c_long, c_long, c_long, P, P]),
    "segsde_mix_labels": (c_int, [P, P, c_int, c_int, c_int, P, P]),
    "segsde_depthcomp_mask": (c_int, [P, c_int, c_long, c_float, c_float, P, P, P]),
    "segsde_depth_threshold_mask": (c_int, [P, c_long, c_float, c_float, c_int, P, P]),
    "segsde_class_mask": (c_int, [P, c_long, P, c_int, P, P]),
    "segsde_confusion_update": (c_int, [P, c_long, c_long, c_long, P, P, c_int, c_long, c_int, P, P]),
    "segsde_multi_tensor_lerp": (c_int, [P, c_int, c_float, c_float, P]),
    "segsde_pseudo_label": (c_int, [P, c_int, c_int, c_long, c_float, c_int64, P, P, P, P, P]),
    "segsde_color_jitter": (c_int, [P, c_int, c_long, P, POINTER(c_int), P, P]),
    "segsde_gaussian_blur": (c_int, [P, c_int, c_int, c_int, P, c_int, P, c_int, P, P, P]),
    "segsde_softmax_nhwc_to_nchw": (c_int, [P, c_int, c_int, c_long, c_int, P, P]),
    "segsde_onehot_select": (c_int, [P, P, c_int, P, c_int, c_int, c_long, P]),
    "segsde_minmax_normalize_workspace": (c_size_t, [c_int, c_long]),
    "segsde_minmax_normalize": (c_int, [P, c_int, c_long, P, P, P, P, c_size_t, P]),
    "segsde_disp_to_depth": (c_int, [P, c_int, c_int, c_int, c_int, c_int, c_float, c_float, P, P]),
}
EXPORTS = sorted(_SIGS)


def bind(cdll):
    """Attach argtypes/restype for every symbol of include/segsde_hip.h (raises if one is missing)."""
    for name, (res, args) in _SIGS.items():
        fn = getattr(cdll, name)
        fn.restype = res
        fn.argtypes = args
    v = cdll.segsde_abi_version()
    if v != ABI_VERSION:
        raise RuntimeError("libsegsde_hip ABI %d != binding ABI %d" % (v, ABI_VERSION))
    return cdll


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "%s not found: the HIP extension is required (no CPU fallback). Build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'`." % LIB_PATH)
        _LIB = bind(ctypes.CDLL(LIB_PATH))
    return _LIB


ERRORS = {-1: "null pointer", -2: "bad shape / descriptor", -3: "workspace too small", -4: "unsupported"}


def check(code, what):
    if code != 0:
        raise RuntimeError("%s failed: %s" % (what, ERRORS.get(code, "hipError %d" % code)))
