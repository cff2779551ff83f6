"""ctypes binding of libmrgs.so (the C ABI declared in include/mrgs.h).

The product path has NO fallback: if the HIP library is missing or fails to load, importing the rasterizer
raises.  torch is imported first so that libmrgs.so binds to the HIP runtime torch already loaded
(same SONAME libamdhip64.so.7) and therefore shares its devices, streams and allocations.
"""
import ctypes
import os
import subprocess

import torch  # noqa: F401  (must precede the CDLL below, see module docstring)

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
# MRGS_LIB: developer override to load another build of the same library (kernel experiments); never a fallback
LIB_PATH = os.environ.get("MRGS_LIB") or os.path.join(_CSRC, "libmrgs.so")

c_int32, c_int64, c_float, c_void_p, c_size_t = ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t


class _Sized(ctypes.Structure):
    """Structs whose first field is `struct_size` (include/mrgs.h): filled in here, so positional arguments start at the second field."""

    def __init__(self, *args, **kw):
        super().__init__(ctypes.sizeof(type(self)), *args, **kw)


class MrgsRasterConfig(_Sized):
    _fields_ = [("struct_size", ctypes.c_uint32), ("P", c_int32), ("S", c_int32), ("D", c_int32), ("M", c_int32), ("H", c_int32), ("W", c_int32),
                ("tanfovx", c_float), ("tanfovy", c_float), ("scale_modifier", c_float), ("prefiltered", c_int32),
                ("debug", c_int32)]


class MrgsRasterInputs(_Sized):
    _fields_ = [("struct_size", ctypes.c_uint64)] + [(n, c_void_p) for n in ("bg", "means3D", "shs", "colors_precomp", "features", "opacities", "scales",
                                        "rotations", "transMat_precomp", "viewmatrix", "projmatrix", "campos", "work_hint", "shs_rest",
                                        "bwd_grad_ws")] + [("hint_flags", ctypes.c_uint32), ("features_live", ctypes.c_uint32)]


MRGS_HINT_REUSE_ORDER = 1
MRGS_HINT_VISIBLE_BYTES = 2


class MrgsRasterGrads(_Sized):
    _fields_ = [("struct_size", ctypes.c_uint64)] + [(n, c_void_p) for n in ("dL_dmeans2D", "dL_dcolors", "dL_dfeatures", "dL_dopacity", "dL_dmeans3D",
                                        "dL_dtransMat", "dL_dsh", "dL_dscales", "dL_drotations", "dL_dsh_rest",
                                        "glue_params", "glue_grads")]     # ABI 10: the glue epilogue (pointers to MrgsSurfelParams / MrgsSurfelGrads)


MRGS_MAX_MIPS = 8


class MrgsEnvMips(ctypes.Structure):
    _fields_ = [("n_levels", c_int32), ("res", c_int32 * MRGS_MAX_MIPS), ("tex", c_void_p * MRGS_MAX_MIPS),
                ("grad", c_void_p * MRGS_MAX_MIPS), ("grad_copies", c_int32 * MRGS_MAX_MIPS), ("min_roughness", c_float), ("max_roughness", c_float)]


class MrgsStridedMap(ctypes.Structure):
    _fields_ = [("ptr", c_void_p), ("stride_h", c_int64), ("stride_w", c_int64), ("stride_c", c_int64)]


class MrgsShadeFrame(ctypes.Structure):
    _fields_ = [("H", c_int32), ("W", c_int32), ("Kinv", c_float * 9), ("R", c_void_p), ("T", c_void_p),
                ("albedo", MrgsStridedMap), ("normal", MrgsStridedMap), ("alpha", MrgsStridedMap), ("refl", MrgsStridedMap),
                ("roughness", MrgsStridedMap), ("lut", c_void_p), ("lut_res", c_int32)]


class MrgsRasterTicket(ctypes.Structure):
    _fields_ = [("device", c_int32), ("slot", c_int32), ("seq", ctypes.c_uint64), ("capacity_pairs", c_int64)]


class MrgsSpmvDesc(ctypes.Structure):
    _fields_ = [("nrows", c_int32), ("lanes_per_row", c_int32), ("col_bytes", c_int32), ("val_bytes", c_int32), ("row_ptr", c_void_p),
                ("col", c_void_p), ("val", c_void_p), ("row_scale", c_void_p), ("x", c_void_p), ("y", c_void_p),
                ("image_rows", c_void_p), ("pre_scale", c_void_p), ("tile_ptr", c_void_p), ("panel_ptr", c_void_p), ("panel_src", c_void_p),
                ("res", c_int32), ("n_tiles", c_int32), ("max_panel", c_int32), ("reserved", c_int32)]


class MrgsSurfelParams(ctypes.Structure):
    _fields_ = [("P", c_int32)] + [(n, c_void_p) for n in ("xyz", "scaling_raw", "rotation_raw", "opacity_raw", "refl_raw", "rough_raw",
                                                             "ori_color_raw", "indirect_dc", "indirect_rest", "campos", "viewmatrix")]


class MrgsSurfelGrads(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in ("d_xyz", "d_scaling", "d_rotation", "d_opacity", "d_refl", "d_rough", "d_ori_color",
                                        "d_indirect_dc", "d_indirect_rest")]


class MrgsMapsFrame(ctypes.Structure):
    _fields_ = [("H", c_int32), ("W", c_int32), ("view_rot", c_float * 9), ("ray_matrix", c_float * 9), ("ray_origin", c_float * 3),
                ("depth_ratio", c_float), ("pgsr_fx", c_float), ("pgsr_fy", c_float), ("rend_distance", c_void_p), ("g_rend_distance", c_void_p)]


class MrgsLossConfig(ctypes.Structure):
    _fields_ = [("H", c_int32), ("W", c_int32), ("C", c_int32), ("lambda_dssim", c_float), ("lambda_normal", c_float),
                ("lambda_dist", c_float)]


class MrgsAdamTensor(ctypes.Structure):
    _fields_ = [("param", c_void_p), ("grad", c_void_p), ("exp_avg", c_void_p), ("exp_avg_sq", c_void_p), ("numel", c_int64),
                ("lr", c_float), ("step", c_int32)]


class MrgsCompactTensor(ctypes.Structure):
    _fields_ = [("src", c_void_p), ("dst", c_void_p), ("row_floats", c_int32)]


class MrgsKernelTimes(ctypes.Structure):
    _fields_ = [(n, c_float) for n in ("preprocess_ms", "sort_ms", "duplicate_ms", "render_fwd_ms", "render_bwd_ms",
                                       "preprocess_bwd_ms")]


# every symbol include/mrgs.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "mrgs_geom_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "mrgs_img_bytes": (c_size_t, [c_int32, c_int32]),
    "mrgs_binning_bytes": (c_size_t, [c_int64]),
    "mrgs_work_hint_bytes": (c_size_t, [c_int32, c_int32]),
    "mrgs_grad_bytes": (c_size_t, [c_int32, c_int32]),
    "mrgs_rasterize_forward_geom": (ctypes.c_int, [ctypes.POINTER(MrgsRasterConfig), ctypes.POINTER(MrgsRasterInputs), c_void_p,
                                                   c_size_t, c_void_p, ctypes.POINTER(c_int64), c_void_p]),
    "mrgs_rasterize_forward_render": (ctypes.c_int, [ctypes.POINTER(MrgsRasterConfig), ctypes.POINTER(MrgsRasterInputs), c_void_p,
                                                     c_void_p, c_size_t, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mrgs_rasterize_forward": (ctypes.c_int, [ctypes.POINTER(MrgsRasterConfig), ctypes.POINTER(MrgsRasterInputs), c_void_p, c_size_t,
                                              c_void_p, c_size_t, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                              ctypes.POINTER(c_int64), c_void_p]),
    "mrgs_rasterize_forward_begin": (ctypes.c_int, [ctypes.POINTER(MrgsRasterConfig), ctypes.POINTER(MrgsRasterInputs), c_void_p, c_size_t,
                                                    c_void_p, c_size_t, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                    ctypes.POINTER(MrgsRasterTicket), c_void_p]),
    "mrgs_rasterize_forward_finish": (ctypes.c_int, [ctypes.POINTER(MrgsRasterTicket), ctypes.POINTER(c_int64)]),
    "mrgs_rasterize_backward": (ctypes.c_int, [ctypes.POINTER(MrgsRasterConfig), ctypes.POINTER(MrgsRasterInputs), c_void_p, c_void_p,
                                               c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                               ctypes.POINTER(MrgsRasterGrads), c_void_p]),
    "mrgs_rasterize_backward_blend": (ctypes.c_int, [ctypes.POINTER(MrgsRasterConfig), ctypes.POINTER(MrgsRasterInputs), c_void_p, c_void_p,
                                                     c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mrgs_rasterize_backward_finish": (ctypes.c_int, [ctypes.POINTER(MrgsRasterConfig), ctypes.POINTER(MrgsRasterInputs), c_void_p, c_void_p,
                                                      c_void_p, ctypes.POINTER(MrgsRasterGrads), c_void_p]),
    "mrgs_surfel_features_forward": (ctypes.c_int, [ctypes.POINTER(MrgsSurfelParams), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mrgs_surfel_features_backward": (ctypes.c_int, [ctypes.POINTER(MrgsSurfelParams), c_void_p, c_void_p, c_void_p, c_void_p,
                                                     ctypes.POINTER(MrgsSurfelGrads), c_void_p, c_void_p]),
    "mrgs_surfel_maps_forward": (ctypes.c_int, [ctypes.POINTER(MrgsMapsFrame), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                c_void_p, c_void_p]),
    "mrgs_surfel_maps_backward": (ctypes.c_int, [ctypes.POINTER(MrgsMapsFrame), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                 c_void_p, c_void_p, c_void_p, c_void_p]),
    "mrgs_surfel_composite_forward": (ctypes.c_int, [c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                     c_void_p, c_void_p]),
    "mrgs_surfel_composite_backward": (ctypes.c_int, [c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "mrgs_cubemap_filter_count": (ctypes.c_int, [c_int32, c_int32, c_float, c_float, c_void_p, c_void_p, c_void_p]),
    "mrgs_cubemap_filter_fill": (ctypes.c_int, [c_int32, c_int32, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mrgs_csr_spmv3": (ctypes.c_int, [c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_void_p]),
    "mrgs_csr_spmv3_batched": (ctypes.c_int, [ctypes.POINTER(MrgsSpmvDesc), c_int32, c_void_p]),
    "mrgs_cube_symmetry_rows": (ctypes.c_int, [c_int32, c_void_p]),
    "mrgs_cubemap_mip_chain_forward": (ctypes.c_int, [c_int32, c_int32, c_void_p, ctypes.POINTER(c_void_p), c_void_p]),
    "mrgs_cubemap_mip_chain_backward": (ctypes.c_int, [c_int32, c_int32, ctypes.POINTER(c_void_p), c_void_p]),
    "mrgs_cubemap_mip_forward": (ctypes.c_int, [c_int32, c_void_p, c_void_p, c_void_p]),
    "mrgs_cubemap_mip_backward": (ctypes.c_int, [c_int32, c_void_p, c_void_p, c_void_p]),
    "mrgs_surfel_feature_grads": (ctypes.c_int, [c_int32, c_int32] + [c_void_p] * 11),
    "mrgs_loss_ws_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "mrgs_loss_forward": (ctypes.c_int, [ctypes.POINTER(MrgsLossConfig), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "mrgs_loss_backward": (ctypes.c_int, [ctypes.POINTER(MrgsLossConfig), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mrgs_bvh_bytes": (c_size_t, [c_int64]),
    "mrgs_bvh_build": (ctypes.c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_size_t]),
    "mrgs_bvh_trace": (ctypes.c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mrgs_bvh_visibility": (ctypes.c_int, [c_void_p, c_int64, c_int32, c_int32, ctypes.POINTER(c_float), c_void_p, c_void_p,
                                           ctypes.POINTER(MrgsStridedMap), ctypes.POINTER(MrgsStridedMap), c_void_p, c_void_p, c_void_p]),
    "mrgs_adam_step": (ctypes.c_int, [ctypes.POINTER(MrgsAdamTensor), c_int32, ctypes.c_double, ctypes.c_double, ctypes.c_double, c_void_p]),
    "mrgs_indirect_blend_forward": (ctypes.c_int, [c_int32, c_int32, c_void_p, c_void_p, ctypes.POINTER(MrgsStridedMap),
                                                   ctypes.POINTER(MrgsStridedMap), c_void_p, c_void_p, c_void_p, c_void_p]),
    "mrgs_indirect_blend_backward": (ctypes.c_int, [c_int32, c_int32, c_void_p, c_void_p, ctypes.POINTER(MrgsStridedMap),
                                                    ctypes.POINTER(MrgsStridedMap), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                    c_void_p, c_void_p]),
    "mrgs_cubemap_encode_forward": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int64, c_int32, c_int32, c_void_p]),
    "mrgs_cubemap_encode_backward": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int64, c_int32,
                                                    c_int32, c_void_p]),
    "mrgs_surfel_bvh_bytes": (c_size_t, [c_int64]),
    "mrgs_surfel_bvh_ws_bytes": (c_size_t, [c_int64]),
    "mrgs_surfel_trace_state_floats": (c_size_t, [c_int64, c_int32]),
    "mrgs_surfel_trace_state_floats_norecord": (c_size_t, [c_int64, c_int32]),
    "mrgs_surfel_trace_state_layout": (ctypes.c_int, [c_int64, c_int32, ctypes.POINTER(c_size_t)]),
    "mrgs_mirror_rays_forward": (ctypes.c_int, [c_int32, c_int32, ctypes.POINTER(c_float), c_void_p, c_void_p, ctypes.POINTER(MrgsStridedMap), c_void_p,
                                                c_void_p, c_void_p, c_void_p]),
    "mrgs_mirror_rays_backward": (ctypes.c_int, [c_int32, c_int32, ctypes.POINTER(c_float), c_void_p, c_void_p, ctypes.POINTER(MrgsStridedMap), c_void_p,
                                                 c_void_p, c_void_p, c_void_p, c_void_p]),
    "mrgs_mirror_rays_blended_forward": (ctypes.c_int, [c_int32, c_int32, ctypes.POINTER(c_float), c_void_p, c_void_p, ctypes.POINTER(MrgsStridedMap),
                                                        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mrgs_mirror_rays_blended_backward": (ctypes.c_int, [c_int32, c_int32, ctypes.POINTER(c_float), c_void_p, c_void_p, ctypes.POINTER(MrgsStridedMap),
                                                         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mrgs_traced_blend_forward": (ctypes.c_int, [c_int32, c_int32, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p]),
    "mrgs_traced_blend_backward": (ctypes.c_int, [c_int32, c_int32, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64] + [c_void_p] * 5),
    "mrgs_surfel_trace_prep_raw_forward": (ctypes.c_int, [c_int64] + [c_void_p] * 6 + [c_int32] + [c_void_p] * 2 + [c_float] + [c_void_p] * 4),
    "mrgs_surfel_trace_prep_raw_backward": (ctypes.c_int, [c_int64] + [c_void_p] * 6 + [c_int32, c_void_p, c_float] + [c_void_p] * 10),
    "mrgs_surfel_trace_prep_forward": (ctypes.c_int, [c_int64] + [c_void_p] * 5 + [c_int32, c_int32] + [c_void_p] * 3 + [c_float] + [c_void_p] * 4),
    "mrgs_surfel_trace_prep_backward": (ctypes.c_int, [c_int64] + [c_void_p] * 4 + [c_int32, c_int32, c_void_p, c_float] + [c_void_p] * 10),
    "mrgs_surfel_bvh_build": (ctypes.c_int, [c_void_p, c_int64, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p]),
    "mrgs_surfel_trace_forward": (ctypes.c_int, [c_void_p, c_int64, c_int64, c_int32] + [c_void_p] * 4 + [ctypes.POINTER(c_float)] + [c_void_p] * 8 + [c_size_t, c_void_p]),
    "mrgs_surfel_trace_backward": (ctypes.c_int, [c_void_p, c_int64, c_int64, c_int32] + [c_void_p] * 4 + [ctypes.POINTER(c_float)] + [c_void_p] * 6 + [c_size_t] + [c_void_p] * 11),
    "mrgs_sh_grad_expand": (ctypes.c_int, [c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "mrgs_sh_grad_expand_surfel": (ctypes.c_int, [c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                                  c_void_p, c_void_p]),
    "mrgs_sh_grad_expand_surfel_rows": (ctypes.c_int, [c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p,
                                                       c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mrgs_side_stream_fork": (ctypes.c_int, [c_void_p, ctypes.POINTER(c_void_p)]),
    "mrgs_side_stream_join": (ctypes.c_int, [c_void_p]),
    "mrgs_side_stream_fork_at_blend": (ctypes.c_int, [c_void_p, ctypes.POINTER(c_void_p)]),
    "mrgs_side_stream_arm_blend_mark": (ctypes.c_int, [c_void_p]),
    "mrgs_compact_ws_bytes": (c_size_t, [c_int64]),
    "mrgs_compact_count": (ctypes.c_int, [c_int64, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "mrgs_compact_rows": (ctypes.c_int, [c_int64, c_void_p, c_void_p, ctypes.POINTER(MrgsCompactTensor), c_int32, c_void_p]),
    "mrgs_mark_visible": (ctypes.c_int, [c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mrgs_envmap_lookup_forward": (ctypes.c_int, [ctypes.POINTER(MrgsEnvMips), c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mrgs_envmap_lookup_backward": (ctypes.c_int, [ctypes.POINTER(MrgsEnvMips), c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                   c_void_p]),
    "mrgs_shade_specular_forward": (ctypes.c_int, [ctypes.POINTER(MrgsEnvMips), ctypes.POINTER(MrgsShadeFrame), c_void_p, c_void_p,
                                                   c_void_p, c_void_p]),
    "mrgs_shade_specular_forward_composite": (ctypes.c_int, [ctypes.POINTER(MrgsEnvMips), ctypes.POINTER(MrgsShadeFrame), c_void_p, c_void_p, c_int32,
                                                             c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "mrgs_shade_specular_backward": (ctypes.c_int, [ctypes.POINTER(MrgsEnvMips), ctypes.POINTER(MrgsShadeFrame), c_void_p, c_void_p,
                                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mrgs_shade_specular_backward_features": (ctypes.c_int, [ctypes.POINTER(MrgsEnvMips), ctypes.POINTER(MrgsShadeFrame), c_void_p, c_void_p,
                                                             c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mrgs_surfel_shade_composite_backward": (ctypes.c_int, [ctypes.POINTER(MrgsEnvMips), ctypes.POINTER(MrgsShadeFrame), c_int32] + [c_void_p] * 13),
    "mrgs_debug_export": (ctypes.c_int, [ctypes.POINTER(MrgsRasterConfig), c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p,
                                         c_void_p]),
    "mrgs_set_profiling": (ctypes.c_int, [c_int32]),
    "mrgs_get_kernel_times": (ctypes.c_int, [ctypes.POINTER(MrgsKernelTimes)]),
    "mrgs_strerror": (ctypes.c_char_p, [ctypes.c_int]),
    "mrgs_last_hip_error": (ctypes.c_char_p, []),
    "mrgs_version": (ctypes.c_char_p, []),
    "mrgs_abi_version": (c_int32, []),
}
MRGS_ABI_VERSION = 10   # the revision of include/mrgs.h these ctypes declarations were written against

_lib = None


def build(force=False):
    """Compile libmrgs.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", _CSRC, "-s", "clean"])
    subprocess.check_call(["make", "-C", _CSRC, "-s", "-j8"])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: the HIP extension has not been built (run `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C materialrefgs_amd/csrc`). There is no CPU fallback for the rasterizer.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)   # AttributeError here = the library does not export what mrgs.h declares
            fn.restype = res
            fn.argtypes = args
        if L.mrgs_abi_version() != MRGS_ABI_VERSION:
            raise ImportError(f"{LIB_PATH} implements ABI revision {L.mrgs_abi_version()} of include/mrgs.h, this binding was written "
                              f"against revision {MRGS_ABI_VERSION}: rebuild the library (make -C materialrefgs_amd/csrc)")
        _lib = L
    return _lib


MRGS_E_WORKSPACE = 5


# ---- launch plumbing (host time per call matters: a full render is ~30 native calls a view) -----------------------------------------
class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()
_current_device = torch._C._cuda_getDevice                 # int
_raw_stream = torch._C._cuda_getCurrentRawStream           # (device index) -> the hipStream_t of torch's current stream, as an int


def guard(dev):
    """`with guard(dev):` -- torch.cuda.device(dev) only when `dev` is not already the current device (the usual case: one process, one
    GPU), otherwise nothing: the device context costs ~4 us a time, a view enters it ~25 times."""
    idx = dev.index
    return _NO_GUARD if (idx is None or idx == _current_device()) else torch.cuda.device(dev)


def stream_ptr(dev):
    """torch's current stream on `dev` as the integer handle the C ABI's `void* stream` arguments take (ctypes converts)."""
    idx = dev.index
    return _raw_stream(_current_device() if idx is None else idx)


def f32c(t):
    """detached float32 contiguous view of `t` (itself when it already is one: the common case costs two attribute reads)."""
    if t.dtype is torch.float32 and t.is_contiguous():
        return t.detach() if t.requires_grad else t
    return t.detach().to(torch.float32).contiguous()


def check(rc):
    if rc != 0:
        L = lib()
        msg = L.mrgs_strerror(rc).decode()
        if rc == 4:
            msg += ": " + L.mrgs_last_hip_error().decode()
        raise RuntimeError(f"libmrgs: {msg}")
