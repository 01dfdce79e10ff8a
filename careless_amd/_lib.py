"""ctypes binding of libcareless_hip.so (the C-ABI declared in include/careless_hip.h).

The library is the only compute path of this package: if it is missing or cannot be loaded the import of any
compute entry point raises -- there is no CPU or PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CARELESS_HIP_LIB") or os.path.join(_HERE, "lib", "libcareless_hip.so")   # env override: A/B builds

CL_MLP_TILE = 128
CL_HIST_STRIDE = 8
CL_SC_NLL, CL_SC_KL, CL_SC_GNORM2, CL_SC_GNORM2_SANE, CL_SC_COUNT = 0, 1, 2, 3, 4
CL_LIK_NORMAL, CL_LIK_STUDENTT = 0, 1
CL_BIJ_EXP, CL_BIJ_SOFTPLUS = 0, 1
CL_PRIOR_WILSON, CL_PRIOR_DOUBLE_WILSON = 0, 1
CL_LAUE_LIK_MAX_BLOCKS = 2048
CL_EV11_WAVES = 8               # wave slots per workgroup in ev11_part (include/careless_hip.h)

_vp = C.c_void_p


class TnArgs(C.Structure):
    """mirror of `cl_tn_args` (include/careless_hip.h)"""
    _fields_ = [
        ("q_loc_raw", _vp), ("q_scale_raw", _vp), ("low", _vp), ("centric", _vp), ("es", _vp),
        ("R", C.c_int), ("S", C.c_int),
        ("high", C.c_float), ("eps", C.c_float),
        ("w_kl", C.c_float), ("kl_grad_mult", C.c_float),
        ("kl_begin", C.c_int), ("kl_end", C.c_int),
        ("r_begin", C.c_int), ("r_end", C.c_int),
        ("u_f", _vp),
        ("seed", C.c_ulonglong), ("step", C.c_uint),
        ("z_f", _vp), ("dz_f", _vp), ("d_loc_raw", _vp), ("d_scale_raw", _vp),
        ("scalars", _vp), ("kl_part", _vp), ("kl_part_dw", _vp),
        ("zero_ptr", _vp), ("zero_n", C.c_longlong), ("zero_dzf", _vp),
        ("red_partials", _vp), ("red_nparts", C.c_int), ("red_P", C.c_int), ("red_out", _vp),
        ("stop_flag", _vp),
        ("prior_kind", C.c_int),
        ("parent_ids", _vp), ("root", _vp), ("dw_r", _vp), ("dz_f_out", _vp),
        ("dw_r_raw", _vp), ("asu_ids", _vp), ("d_dw_r_raw", _vp), ("n_asu", C.c_int),
        ("dw_child_seg", _vp), ("dw_child_ids", _vp),
    ]


class MlpArgs(C.Structure):
    """mirror of `cl_mlp_args` (include/careless_hip.h)"""
    _fields_ = [
        ("refl_id", _vp), ("image_id", _vp), ("meta_t", _vp), ("iobs", _vp), ("sig", _vp),
        ("n_obs", C.c_int), ("n_pad", C.c_int),
        ("obs_offset", C.c_longlong),
        ("mlp", _vp),
        ("d", C.c_int), ("w", C.c_int), ("L", C.c_int),
        ("leak", C.c_float),
        ("img", _vp),
        ("use_img", C.c_int),
        ("z_f", _vp),
        ("R", C.c_int), ("S", C.c_int),
        ("lik_kind", C.c_int), ("dof", C.c_float), ("lik_const", C.c_float),
        ("bij_kind", C.c_int), ("eps", C.c_float), ("shift", C.c_float),
        ("w_ll", C.c_float),
        ("eta", _vp),
        ("seed", C.c_ulonglong), ("step", C.c_uint),
        ("dz_f", _vp), ("d_img", _vp), ("partials", _vp), ("scalars", _vp),
        ("ipred_out", _vp), ("loc_out", _vp), ("sig_out", _vp), ("dO_ext", _vp),
        ("stop_flag", _vp),
        ("ev11", _vp), ("d_ev11", _vp),
        ("imgl", _vp), ("d_imgl", _vp), ("n_imgl", C.c_int), ("n_images", C.c_int), ("tile_img", _vp), ("row_map", _vp),
        ("gmeta", _vp), ("tile_gmax", _vp), ("noise_row", _vp),
        ("act_out", _vp), ("dH_ext", _vp), ("dX_out", _vp),
        ("dzf_obs", _vp), ("dimg_obs", _vp), ("nll_part", _vp), ("det_slot", _vp), ("dZ0_out", _vp), ("ev11_part", _vp),
    ]


class DetArgs(C.Structure):
    """mirror of `cl_det_args` (include/careless_hip.h)"""
    _fields_ = [
        ("dzf_obs", _vp), ("perm_refl", _vp), ("seg_refl", _vp), ("R", C.c_int), ("S", C.c_int), ("dz_f", _vp),
        ("dimg_obs", _vp), ("perm_img", _vp), ("seg_img", _vp), ("n_images", C.c_int), ("d_img", _vp),
        ("nll_part", _vp), ("nparts", C.c_int), ("scalars", _vp),
        ("stop_flag", _vp),
        ("ev11_part", _vp), ("n_ev11", C.c_int), ("d_ev11", _vp),
    ]


class LaueArgs(C.Structure):
    """mirror of `cl_laue_args` (include/careless_hip.h)"""
    _fields_ = [
        ("refl_id", _vp), ("image_id", _vp), ("harmonic_id", _vp), ("loc", _vp), ("sigma", _vp), ("iobs", _vp), ("sig", _vp),
        ("n_obs", C.c_int),
        ("obs_offset", C.c_longlong),
        ("img", _vp), ("use_img", C.c_int),
        ("z_f", _vp), ("R", C.c_int), ("S", C.c_int),
        ("lik_kind", C.c_int), ("dof", C.c_float), ("lik_const", C.c_float),
        ("shift", C.c_float), ("w_ll", C.c_float),
        ("eta", _vp),
        ("seed", C.c_ulonglong), ("step", C.c_uint),
        ("iconv", _vp), ("dz_f", _vp), ("d_img", _vp), ("dO", _vp), ("scalars", _vp), ("ipred_out", _vp), ("stop_flag", _vp),
        ("ev11", _vp), ("d_ev11", _vp), ("row_index", _vp), ("nll_part", _vp),
        ("dzf_obs", _vp), ("dimg_obs", _vp), ("det_slot", _vp), ("ev11_part", _vp),
    ]


class FrozenArgs(C.Structure):
    """mirror of `cl_frozen_args` (include/careless_hip.h)"""
    _fields_ = [
        ("refl_id", _vp), ("loc", _vp), ("sigma", _vp), ("aim", _vp), ("iobs", _vp), ("sig", _vp), ("key", _vp),
        ("obs_offset", C.c_longlong), ("n", C.c_longlong),
        ("R", C.c_int), ("S", C.c_int),
        ("z_f", _vp), ("dz_f", _vp),
        ("accumulate", C.c_int),
        ("lik_kind", C.c_int), ("dof", C.c_float), ("lik_const", C.c_float),
        ("shift", C.c_float), ("w_ll", C.c_float),
        ("eta", _vp),
        ("seed", C.c_ulonglong), ("step", C.c_uint),
        ("scalars", _vp), ("ipred_out", _vp), ("stop_flag", _vp),
        ("ev11", _vp), ("d_ev11", _vp),
        ("edge_rid", _vp), ("edge_val", _vp),
        ("nll_part", _vp), ("ev11_part", _vp),
        ("gmeta", _vp), ("gbuf", _vp), ("src", _vp),
    ]


class AdamArgs(C.Structure):
    """mirror of `cl_adam_args` (include/careless_hip.h)"""
    _fields_ = [
        ("p", _vp), ("g", _vp), ("m", _vp), ("v", _vp),
        ("n", C.c_int),
        ("alpha", C.c_float),
        ("beta1", C.c_float), ("beta2", C.c_float), ("adam_eps", C.c_float),
        ("clipnorm", C.c_float), ("clipvalue", C.c_float), ("global_clipnorm", C.c_float),
        ("seg_off", _vp),
        ("nseg", C.c_int),
        ("seg_sq", _vp), ("frozen", _vp), ("scalars", _vp), ("stop_flag", _vp), ("norm_out", _vp),
        ("n_ranges", C.c_int), ("range_begin", C.c_int * 3), ("range_end", C.c_int * 3), ("norm_skip_ranges", C.c_int),
        ("norm_extra", _vp), ("norm_part", _vp),
    ]


EXPORTS = {
    # name: (restype, argtypes)
    "cl_version": (C.c_char_p, []),
    "cl_abi_sizes": (None, [C.POINTER(C.c_size_t)]),
    "cl_mlp_default_grid": (C.c_int, []),
    "cl_mlp_param_count": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "cl_mlp_meta_rows": (C.c_int, [C.c_int]),
    "cl_mlp_max_layers": (C.c_int, [C.c_int]),
    "cl_mlp_max_layers_imgl": (C.c_int, [C.c_int]),
    "cl_tn_forward": (C.c_int, [C.POINTER(TnArgs), _vp]),
    "cl_tn_backward": (C.c_int, [C.POINTER(TnArgs), _vp]),
    "cl_dw_prior_forward": (C.c_int, [C.POINTER(TnArgs), _vp]),
    "cl_elbo_mono_fwd_bwd": (C.c_int, [C.POINTER(MlpArgs), C.c_int, _vp]),
    "cl_mlp_forward": (C.c_int, [C.POINTER(MlpArgs), C.c_int, _vp]),
    "cl_mlp_backward_ext": (C.c_int, [C.POINTER(MlpArgs), C.c_int, _vp]),
    "cl_mlp_kernel_name": (C.c_int, [C.POINTER(MlpArgs), C.c_int, C.c_char_p, C.c_size_t]),
    "cl_wide_ld": (C.c_int, [C.c_int]),
    "cl_wide_dense_forward": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_longlong, C.c_int, C.c_int, C.c_float, C.c_int, _vp, C.c_int, _vp, _vp]),
    "cl_wide_dense_forward_head": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_longlong, C.c_int, C.c_int, C.c_float, _vp, C.c_int, _vp, C.c_int, C.c_float, _vp, _vp, _vp, _vp, _vp]),
    "cl_wide_dense_forward_head_lik": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_longlong, C.c_int, C.c_int, C.c_float, _vp, C.c_int, _vp, C.c_int, C.c_float, _vp, _vp, _vp,
                                                 C.POINTER(LaueArgs), _vp, _vp]),
    "cl_wide_head_bwd_supported": (C.c_int, [C.c_int, C.c_int]),
    "cl_wide_dense_wgrad_head": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, C.c_float, _vp, C.c_int, C.c_longlong, C.c_int, C.c_int, _vp, _vp, C.c_int, _vp, _vp]),
    "cl_wide_dense_dgrad_head": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, C.c_longlong, C.c_int, C.c_int, _vp, C.c_int, C.c_float, _vp, C.c_int, _vp, _vp]),
    "cl_wide_pre_supported": (C.c_int, [C.c_int, C.c_int]),
    "cl_wide_dense2_forward": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, C.c_longlong, C.c_int, C.c_int, C.c_float, _vp, C.c_int, _vp, C.c_int, C.c_float,
                                         _vp, _vp, _vp, _vp]),
    "cl_wide_dense_dgrad_pre": (C.c_int, [_vp, C.c_int, _vp, C.c_longlong, C.c_int, C.c_int, _vp, C.c_int, C.c_int, _vp, _vp, C.c_float, _vp, C.c_int, _vp, _vp]),
    "cl_wide_dgrad_wgrad0_parts": (C.c_int, [C.c_longlong]),
    "cl_wide_dense_dgrad_pre_wgrad0": (C.c_int, [_vp, C.c_int, _vp, C.c_longlong, C.c_int, C.c_int, _vp, C.c_int, C.c_int, _vp, _vp, C.c_float, _vp, _vp, _vp]),
    "cl_wide_dense_wgrad_pre": (C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_int, _vp, _vp, C.c_float, C.c_longlong, C.c_int, C.c_int, _vp, C.c_int, _vp, _vp]),
    "cl_wide_dense_dgrad": (C.c_int, [_vp, C.c_int, _vp, C.c_longlong, C.c_int, C.c_int, _vp, C.c_int, C.c_float, _vp, C.c_int, _vp, _vp]),
    "cl_wide_wgrad_splits": (C.c_int, [C.c_longlong]),
    "cl_wide_dense_wgrad": (C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_longlong, C.c_int, C.c_int, _vp, C.c_int, _vp, _vp]),
    "cl_wide_image_forward": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, C.c_int, C.c_longlong, C.c_int, C.c_float, _vp, C.c_int, _vp, _vp]),
    "cl_wide_image_dgrad": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_int, C.c_longlong, C.c_int, _vp, C.c_int, C.c_float, _vp, C.c_int, _vp, _vp]),
    "cl_wide_image_forward_tiles": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_float, _vp, C.c_int, _vp, _vp]),
    "cl_wide_image_dgrad_tiles": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, C.c_int, C.c_int, _vp, C.c_int, C.c_float, _vp, C.c_int, _vp, _vp]),
    "cl_wide_image_wgrad": (C.c_int, [_vp, C.c_int, _vp, C.c_int, _vp, C.c_int, C.c_longlong, C.c_int, _vp, _vp, _vp, _vp]),
    "cl_wide_head_forward": (C.c_int, [_vp, C.c_int, _vp, C.c_longlong, C.c_int, C.c_int, C.c_float, _vp, _vp, _vp, _vp]),
    "cl_wide_head_blocks": (C.c_int, [C.c_longlong]),
    "cl_wide_head_backward": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_longlong, C.c_int, C.c_int, C.c_float, C.c_float, _vp, C.c_int, _vp, C.c_int, _vp, _vp]),
    "cl_det_reduce": (C.c_int, [C.POINTER(DetArgs), _vp]),
    "cl_laue_predict": (C.c_int, [C.POINTER(LaueArgs), _vp]),
    "cl_laue_likelihood": (C.c_int, [C.POINTER(LaueArgs), _vp]),
    "cl_laue_backward": (C.c_int, [C.POINTER(LaueArgs), _vp]),
    "cl_slot_rows": (C.c_int, [C.POINTER(LaueArgs), _vp]),
    "cl_frozen_rows": (C.c_int, [C.POINTER(FrozenArgs), _vp]),
    "cl_frozen_edge_floats": (C.c_int, [C.c_longlong, C.c_int]),
    "cl_frozen_grid": (C.c_int, [C.c_longlong]),
    "cl_frozen_args_size": (C.c_size_t, []),
    "cl_reduce_partials": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, _vp]),
    "cl_grad_sqnorm": (C.c_int, [_vp, C.c_int, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp]),
    "cl_adam_step": (C.c_int, [C.POINTER(AdamArgs), _vp]),
    "cl_owner_qnorm": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp]),
    "cl_step_finalize": (C.c_int, [_vp, C.c_float, _vp, C.c_int, _vp, _vp, C.c_int, _vp]),
    "cl_adam_grid": (C.c_int, [C.POINTER(AdamArgs)]),
    "cl_peel_supported": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "cl_peel_parts": (C.c_int, [C.c_longlong]),
    "cl_peel_forward": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, C.c_int, _vp, _vp]),
    "cl_peel_backward": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, C.c_int, _vp, _vp]),
    "cl_chain_dx": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp]),
    "cl_tn_moments": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_double, C.c_double, C.c_float, _vp, _vp, _vp, _vp]),
    "cl_host_asu_map": (C.c_int, [_vp, C.c_longlong, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, C.c_int]),
    "cl_host_dense_ids": (C.c_int, [_vp, C.c_longlong, C.c_int64, C.c_int64, _vp, C.POINTER(C.c_longlong), C.c_int]),
    "cl_host_crystfel_count": (C.c_longlong, [_vp, C.c_longlong, C.POINTER(C.c_longlong), C.c_int]),
    "cl_host_crystfel_parse": (C.c_int, [_vp, C.c_longlong, C.c_longlong, _vp, C.c_int]),
    "cl_predict_moments": (C.c_int, [_vp, _vp, _vp, C.c_longlong, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp]),
    "cl_debug_noise": (C.c_int, [C.c_ulonglong, C.c_uint, C.c_int, C.c_longlong, C.c_longlong, C.c_int, _vp, _vp]),
}

_lib: Optional[C.CDLL] = None


class CarelessHipError(RuntimeError):
    pass


def get_lib() -> C.CDLL:
    """Load libcareless_hip.so (once).  Raises CarelessHipError when the HIP extension is not built/loadable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CarelessHipError(
            f"{LIB_PATH} is missing: build it with `python -m careless_amd.build` (needs hipcc, targets gfx950). "
            "careless_amd has no CPU fallback.")
    # torch ships its own libamdhip64; import it first so this library binds to the SAME HIP runtime (two runtimes in
    # one process cannot share streams or device memory: launches then fail with hipErrorNoDevice)
    import torch  # noqa: F401
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover - depends on the machine
        raise CarelessHipError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in EXPORTS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise CarelessHipError(f"{LIB_PATH} does not export {name}; rebuild it") from e
        fn.restype = res
        fn.argtypes = args
    sizes = (C.c_size_t * 5)()
    lib.cl_abi_sizes(sizes)
    mine = (C.sizeof(TnArgs), C.sizeof(MlpArgs), C.sizeof(AdamArgs), C.sizeof(LaueArgs), C.sizeof(DetArgs))
    if tuple(sizes) != mine:
        raise CarelessHipError(f"ABI mismatch between careless_amd/_lib.py {mine} and the library {tuple(sizes)}")
    if int(lib.cl_frozen_args_size()) != C.sizeof(FrozenArgs):
        raise CarelessHipError(f"ABI mismatch: cl_frozen_args is {int(lib.cl_frozen_args_size())} bytes in the library, {C.sizeof(FrozenArgs)} in careless_amd/_lib.py")
    _lib = lib
    return lib


def check(code: int, what: str) -> None:
    """Turn a C-ABI return code into an exception."""
    if code == 0:
        return
    if code == -2 and what.startswith("cl_wide"):
        raise NotImplementedError(f"{what}: this layer shape / buffer layout is outside the kernel's envelope (row pitch = cl_wide_ld(width), "
                                  "16-byte aligned buffers; the fused forms take hidden widths 65 .. 128)")
    if code == -2:
        raise NotImplementedError(
            f"{what}: scaler geometry not supported by the fused gfx950 kernel "
            "(needs mlp_width <= 64, metadata width <= 64 and mlp_layers <= 20 / 10 / 5 for width <= 16 / 32 / 64)")
    if code == -4:
        raise ValueError(f"{what}: one launch addresses < 4 GiB of metadata, z_f and per-(row, sample) arrays (32-bit lane offsets); the "
                         "engine cuts plain-layout shards into several launches (engine.ObsChunks) -- a packed layout (Laue, per-image "
                         "layers) or R x S amplitudes past the bound need more ranks")
    if code < 0:
        raise ValueError(f"{what}: invalid argument (code {code})")
    raise CarelessHipError(f"{what}: HIP error {code}")


def ptr(t) -> Optional[int]:
    """device pointer of a torch tensor (None -> NULL)"""
    if t is None:
        return None
    return t.data_ptr()
