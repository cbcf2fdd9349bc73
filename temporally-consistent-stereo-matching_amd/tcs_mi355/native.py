"""ctypes binding of libtcs_mi355.so (include/tcs_mi355.h).

PyTorch is only plumbing here: it owns device memory and the HIP stream; every kernel on the hot
path is in the shared library.  There is NO fallback: if the library is missing, or a tensor is not
a contiguous float32 HIP tensor, the call raises.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import torch

from . import build as _build

_LIB = None
_LOCK = threading.Lock()

c_fp = C.c_void_p      # device float*
c_int = C.c_int
c_f = C.c_float
c_sz = C.c_size_t


class ConvDesc(C.Structure):
    """struct tcs_conv_desc (include/tcs_mi355.h)."""
    _fields_ = [
        ("src", c_fp * 4), ("src_ch", c_int * 4), ("n_src", c_int),
        ("weight", c_fp), ("bias", c_fp),
        ("B", c_int), ("H", c_int), ("W", c_int),
        ("Cin", c_int), ("Cout", c_int), ("ksize", c_int),
        ("epilogue", c_int), ("act", c_int), ("post_scale", c_f),
        ("addend", c_fp), ("addend2", c_fp), ("h", c_fp), ("z", c_fp),
        ("blend_keep_z", c_int),
        ("out", c_fp), ("out_ctot", c_int), ("out_coff", c_int), ("out2", c_fp),
        ("stride", c_int), ("math", c_int), ("weight_unscale", c_f),
        ("out16", c_fp), ("out16_groups", c_int), ("out16_group_offset", c_int),
        ("in_transform", c_int), ("src_batch2", c_fp), ("batch_split", c_int),
    ]


class ConvS16Desc(C.Structure):
    """struct tcs_conv_s16_desc (include/tcs_mi355.h)."""
    _fields_ = [
        ("src", c_fp * 4), ("src_ch", c_int * 4), ("src_groups", c_int * 4), ("n_src", c_int),
        ("weight", c_fp), ("bias", c_fp),
        ("B", c_int), ("H", c_int), ("W", c_int),
        ("Cin", c_int), ("Cout", c_int), ("ksize", c_int), ("stride", c_int),
        ("epilogue", c_int), ("act", c_int), ("post_scale", c_f), ("weight_unscale", c_f),
        ("addend", c_fp), ("addend2", c_fp), ("addend16", c_fp), ("addend16_groups", c_int), ("h", c_fp), ("h_groups", c_int), ("z", c_fp),
        ("blend_keep_z", c_int),
        ("out16", c_fp), ("out16_groups", c_int), ("out16_group_offset", c_int),
        ("out32", c_fp), ("out_ctot", c_int), ("out_coff", c_int),
        ("tile_cfg", c_int), ("addend_ctot", c_int),
        ("blend_cand", c_fp), ("blend_cand_ctot", c_int), ("blend_disp", c_fp), ("blend_refined", c_fp), ("blend_delta", c_fp),
        ("blend_coords1", c_fp), ("blend_flow_x", c_fp), ("blend_flow16", c_fp), ("blend_flow16_groups", c_int), ("blend_flow16_channel", c_int),
        ("out16b", c_fp), ("out16b_groups", c_int), ("out16_split", c_int),
        ("in_stats", c_fp), ("in_eps", c_f),
        ("tap_weights", c_fp), ("tap_out", c_fp), ("tap_nout", c_int), ("tap_tiles", c_int), ("tap_unscale", c_f),
        ("blend_warm_pyr", c_fp * 4), ("blend_warm_radius", c_int),
    ]


# name -> (restype, argtypes); must list every symbol declared in include/tcs_mi355.h
SIGNATURES = {
    "tcs_abi_version": (c_int, []),
    "tcs_error_string": (C.c_char_p, [c_int]),
    "tcs_corr_level_bytes": (c_sz, [c_int, c_int, c_int, c_int]),
    "tcs_corr_build_workspace_bytes": (c_sz, [c_int, c_int, c_int]),
    "tcs_corr_build": (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp,
                               c_fp, c_fp, c_fp, c_fp, c_fp]),
    "tcs_corr_ws_level0": (c_fp, [c_fp]),
    "tcs_corr_lookup": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp, c_fp, c_fp]),
    "tcs_corr_lookup_blocks": (c_int, [c_int, c_int, c_int]),
    "tcs_pose_prepare": (c_int, [c_fp, c_fp, c_fp, c_f, c_int, c_fp, c_fp, c_fp, c_fp, c_fp]),
    "tcs_warp_workspace_bytes": (c_sz, [c_int, c_int, c_int, c_int]),
    "tcs_warp_forward": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp, c_fp, c_fp, c_fp, c_fp,
                                 c_fp, c_fp]),
    "tcs_warp_geometry": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp]),
    "tcs_softsplat_sum": (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp, c_fp]),
    "tcs_backward_grid": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_fp, c_fp]),
    "tcs_bilinear_sample": (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_int, c_int, c_int, c_fp, c_fp]),
    "tcs_grid_halve": (c_int, [c_fp, c_int, c_int, c_int, c_fp, c_fp]),
    "tcs_flow_step": (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_fp, c_fp]),
    "tcs_flow_step_grads": (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_f, c_fp, c_fp, c_fp, c_fp]),
    "tcs_disp_gradient_xy": (c_int, [c_fp, c_int, c_int, c_int, c_f, c_fp, c_fp]),
    "tcs_grad_candidates": (c_int, [c_fp, c_int, c_int, c_int, c_fp, c_fp]),
    "tcs_propagate_disparity": (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_fp, c_fp]),
    "tcs_softmax_blend": (c_int, [c_fp, c_fp, c_int, c_fp, c_int, c_int, c_int, c_fp, c_fp, c_fp, c_fp, c_fp, C.c_longlong, c_fp]),
    "tcs_convex_upsample": (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp, c_fp, c_fp]),
    "tcs_avgpool3s2": (c_int, [c_fp, c_int, c_int, c_int, c_int, c_fp, c_fp]),
    "tcs_resize_bilinear": (c_int, [c_fp, c_int, c_int, c_int, c_int, c_int, c_int, c_fp, c_fp]),
    "tcs_conv_packed_floats": (c_sz, [c_int, c_int, c_int]),
    "tcs_pack_conv_weight": (c_int, [c_fp, c_int, c_int, c_int, c_fp, c_fp]),
    "tcs_conv_packed_floats_f16x3": (c_sz, [c_int, c_int, c_int]),
    "tcs_pack_conv_weight_f16x3": (c_int, [c_fp, c_int, c_int, c_int, c_int, c_fp, c_fp]),
    "tcs_deconv_packed_floats_f16x3": (c_sz, [c_int, c_int]),
    "tcs_pack_deconv4x4s2_f16x3": (c_int, [c_fp, c_int, c_int, c_int, c_fp, c_fp, c_fp]),
    "tcs_instance_norm": (c_int, [c_fp, c_int, c_int, c_int, c_int, c_f, c_int, c_fp, c_fp, c_fp]),
    "tcs_conv3x3_cout1": (c_int, [c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp, c_fp]),
    "tcs_conv2d": (c_int, [C.POINTER(ConvDesc), c_fp]),
    "tcs_conv2d_group": (c_int, [C.POINTER(C.POINTER(ConvDesc)), c_int, c_fp]),
    "tcs_s16_bytes": (c_sz, [c_int, c_int, c_int, c_int]),
    "tcs_s16_flags": (c_int, [C.POINTER(C.c_uint)]),
    "tcs_s16_flags_detail": (c_int, [C.POINTER(C.c_uint)]),
    "tcs_s16_from_f32": (c_int, [c_fp, c_int, c_int, c_int, c_int, c_fp, c_int, c_int, c_fp]),
    "tcs_s16_to_f32": (c_int, [c_fp, c_int, c_int, c_int, c_int, c_int, c_int, c_fp, c_fp]),
    "tcs_conv2d_s16": (c_int, [C.POINTER(ConvS16Desc), c_fp]),
    "tcs_conv2d_s16_group": (c_int, [C.POINTER(C.POINTER(ConvS16Desc)), c_int, c_fp]),
    "tcs_conv2d_s16_group_fused": (c_int, [C.POINTER(C.POINTER(ConvS16Desc)), c_int]),
    "tcs_avgpool3s2_s16": (c_int, [c_fp, c_int, c_int, c_int, c_int, c_fp, c_int, c_fp]),
    "tcs_resize_bilinear_s16": (c_int, [c_fp, c_int, c_int, c_int, c_int, c_int, c_int, c_fp, c_int, c_fp]),
    "tcs_instance_norm_s16_workspace_bytes": (c_sz, [c_int, c_int, c_int, c_int]),
    "tcs_instance_norm_s16": (c_int, [c_fp, c_int, c_int, c_int, c_int, c_f, c_int, c_fp, c_int, c_fp, c_int, c_fp, c_fp]),
    "tcs_deconv_in_stats_bytes": (c_sz, [c_int, c_int, c_int, c_int]),
    "tcs_instance_norm_apply_s16": (c_int, [c_fp, c_int, c_int, c_int, c_int, c_int, c_fp, c_int, c_fp, c_int, c_fp, c_int, c_f, c_fp]),
    "tcs_tap_weights_floats": (c_sz, [c_int, c_int]),
    "tcs_pack_tap_weights": (c_int, [c_fp, c_int, c_int, c_int, c_fp, c_fp]),
    "tcs_taps_sum": (c_int, [c_fp, c_int, c_int, c_fp, c_fp, c_f, c_int, c_int, c_int, c_fp, c_fp]),
    "tcs_flow_taps_step_grads": (c_int, [c_fp, c_fp, c_int, c_fp, c_int, c_int, c_int, c_f, c_fp, c_fp, c_fp, c_fp, c_fp]),
    "tcs_taps_propagate_s16": (c_int, [c_fp, c_int, c_fp, c_fp, c_f, c_fp, c_int, c_int, c_int, c_fp, c_fp, c_fp, c_int, c_fp]),
    "tcs_propagate_disparity_s16": (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_fp, c_fp, c_int, c_fp]),
    "tcs_s16_set_channel": (c_int, [c_fp, c_int, c_int, c_int, c_fp, c_int, c_int, c_fp]),
    "tcs_weight_frags_bytes": (c_sz, [c_int, c_int]),
    "tcs_pack_weight_frags": (c_int, [c_fp, c_int, c_int, c_int, c_int, c_fp, c_fp]),
    "tcs_hidden_update_s16": (c_int, [c_fp, c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_f, c_fp, c_fp, c_f, c_fp, c_fp, c_f, c_int, c_int, c_int, c_fp]),
    "tcs_softmax_blend_s16": (c_int, [c_fp, c_fp, c_int, c_fp, c_int, c_int, c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_fp]),
}


class NativeLibraryMissing(RuntimeError):
    pass


def lib_path() -> str:
    return os.environ.get("TCS_MI355_LIB", _build.LIB_PATH)


def lib():
    """The loaded library.  Raises NativeLibraryMissing if it has not been built
    (`python __graft_entry__.py` or `python -m tcs_mi355.build`)."""
    global _LIB
    if _LIB is None:
        with _LOCK:
            if _LIB is None:
                path = lib_path()
                if not os.path.exists(path):
                    raise NativeLibraryMissing(
                        f"{path} not found. The TC-Stereo hot path has no CPU or PyTorch fallback: build the HIP "
                        f"library first (python -c 'import __graft_entry__ as g; g.build()').")
                handle = C.CDLL(path)
                for name, (res, args) in SIGNATURES.items():
                    fn = getattr(handle, name)       # AttributeError if the .so lacks a declared symbol
                    fn.restype = res
                    fn.argtypes = args
                _LIB = handle
    return _LIB


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().tcs_error_string(rc).decode()
        raise RuntimeError(f"{what} failed: {msg} ({rc})")


def ptr(t: torch.Tensor | None, name: str = "tensor", dtype=torch.float32):
    """Device pointer of a contiguous float32 (or `dtype`) HIP tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"{name}: tcs_mi355 kernels need a HIP device tensor, got device={t.device} "
                           f"(there is no CPU path; use the oracle only for checking)")
    if t.dtype != dtype:
        raise ValueError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous tensor")
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream
