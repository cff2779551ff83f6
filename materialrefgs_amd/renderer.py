"""Host-side mirror of the reference's render functions for the hot path (appendix C).

  compute_2dgs_normal_and_regularizations <-> gaussian_renderer/__init__.py:42-90 ("2dgs" flavour: 7-channel allmap), with
                                              depths_to_points / depth_to_normal of utils/point_utils.py:9-37 inside the kernel
  render_initial                          <-> gaussian_renderer/__init__.py:94-220
  render_surfel                           <-> gaussian_renderer/__init__.py:225-483 (SH-indirect branch, opt.indirect = False)

Same arguments (`viewpoint_camera, pc, pipe, bg_color, scaling_modifier, override_color, srgb, opt, ...`) and the same output
dictionary keys.  `pc` is anything exposing the GaussianModel getters the functions read (scene/gaussian_model.py:236-347);
`SurfelModel` below is a minimal container with the reference's activations for tests and benchmarks -- the optimizer,
densification and I/O of GaussianModel are out of scope.
Rasterization, shading and the per-gaussian glue (activations, normals, mirror direction, indirect SH, feature assembly:
`surfel_features`) run in libmrgs.so; the per-map glue stays in torch as in the reference.
"""
import ctypes
import math
import os
from types import SimpleNamespace

import torch

from . import _lib
from ._lib import MrgsMapsFrame, MrgsSurfelGrads, MrgsSurfelParams

from .gs_utils import build_scaling_rotation, eval_sh, flip_align_view, linear_to_srgb, safe_normalize
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer, deferred_raster_count
from .shading import (EnvLight, get_full_color_volume, get_full_color_volume_indirect, get_specular_color_surfel,
                      shade_and_composite_surfel)


class SurfelModel:
    """The subset of scene/gaussian_model.py:GaussianModel that render_* reads: raw parameters + activated getters."""

    def __init__(self, xyz, scaling, rotation, opacity, features_dc, features_rest, refl_strength=None, roughness=None,
                 ori_color=None, indirect_dc=None, indirect_rest=None, envmap=None, active_sh_degree=3, max_sh_degree=3):
        P, dev = xyz.shape[0], xyz.device
        z = lambda *s: torch.zeros(*s, device=dev)
        self._xyz, self._scaling, self._rotation, self._opacity = xyz, scaling, rotation, opacity
        self._features_dc, self._features_rest = features_dc, features_rest
        self._refl_strength = refl_strength if refl_strength is not None else z(P, 1)
        self._roughness = roughness if roughness is not None else z(P, 1)
        self._ori_color = ori_color if ori_color is not None else z(P, 3)
        self._indirect_dc = indirect_dc if indirect_dc is not None else z(P, 1, 3)
        self._indirect_rest = indirect_rest if indirect_rest is not None else z(P, 15, 3)
        self._metalness = z(P, 1)       # GaussianModel._metalness: the per-gaussian blend weight render_surfel2 rasterizes (get_specular)
        self._indirect_asg = z(P, 32, 5)                                      # scene/gaussian_model.py:173 (read with pipe.use_asg only)
        self._asg_axes = None
        self.env_map = envmap
        self.env_map_2 = None           # the second environment map render_volume shades with (GaussianModel.get_envmap_2)
        self.active_sh_degree, self.max_sh_degree = active_sh_degree, max_sh_degree
        self.ray_tracer = None

    def parameters(self):
        return [self._xyz, self._scaling, self._rotation, self._opacity, self._features_dc, self._features_rest, self._refl_strength,
                self._roughness, self._ori_color, self._indirect_dc, self._indirect_rest]

    # activations: gaussian_model.py:56-78
    get_xyz = property(lambda s: s._xyz)
    get_scaling = property(lambda s: torch.exp(s._scaling))
    get_rotation = property(lambda s: torch.nn.functional.normalize(s._rotation))
    get_opacity = property(lambda s: torch.sigmoid(s._opacity))
    get_refl = property(lambda s: torch.sigmoid(s._refl_strength))
    get_rough = property(lambda s: torch.sigmoid(s._roughness))
    get_ori_color = property(lambda s: torch.sigmoid(s._ori_color))
    get_specular = property(lambda s: torch.sigmoid(s._metalness))          # gaussian_model.py:309-311
    get_features = property(lambda s: torch.cat((s._features_dc, s._features_rest), dim=1))
    get_indirect = property(lambda s: torch.cat((s._indirect_dc, s._indirect_rest), dim=1))
    get_asg = property(lambda s: s._indirect_asg)                            # gaussian_model.py:305-307

    @property
    def asg_param(self):                                                     # gaussian_model.py:77: init_predefined_omega(4, 8)
        if self._asg_axes is None or self._asg_axes[0].device != self._xyz.device:
            from .gs_utils import predefined_asg_axes
            self._asg_axes = predefined_asg_axes(4, 8, self._xyz.device)
        return self._asg_axes

    get_envmap = property(lambda s: s.env_map)
    get_envmap_2 = property(lambda s: s.env_map_2 if s.env_map_2 is not None else s.env_map)

    def get_covariance(self, scaling_modifier=1):
        """gaussian_model.py:48-54, 346-347: the splat-to-world transform, transposed -- rows s_u r_u, s_v r_v, r_w, mean."""
        s3 = torch.cat([self.get_scaling * scaling_modifier, torch.ones_like(self.get_scaling[:, :1])], dim=-1)
        trans = torch.zeros((self._xyz.shape[0], 4, 4), dtype=torch.float32, device=self._xyz.device)
        trans[:, :3, :3] = build_scaling_rotation(s3, self._rotation).permute(0, 2, 1)
        trans[:, 3, :3] = self.get_xyz
        trans[:, 3, 3] = 1
        return trans

    def get_normal(self, scaling_modifier, dir_pp_normalized):
        """gaussian_model.py:269-285 (return_delta=False): third column of R(q), flipped to face the viewer."""
        s3 = torch.cat([self.get_scaling * scaling_modifier, torch.ones_like(self.get_scaling[:, :1])], dim=-1)
        RS = build_scaling_rotation(s3, self._rotation)          # [P,3,3]; columns = scaled axes
        normals_raw = RS[:, :, 2]
        normals_raw, _ = flip_align_view(normals_raw, dir_pp_normalized)
        return safe_normalize(normals_raw)


def _p(t):
    return None if t is None else t.data_ptr()      # (an int: ctypes converts it for the `void*` parameters and struct fields)


_c = _lib.f32c


_AFTER_FEATURES_HOOK = [None]


def set_after_features_hook(fn):
    """fn(dL_dindirect_dc [P,1,3]) is called at the end of the per-gaussian glue's backward (the last kernel of a render_surfel backward):
    the second rank-one factor of a view-parallel step's SH exchange is final there (dist.SurfelGradReducer.begin_early_ind; the first one
    leaves the rasterizer earlier, rasterizer.set_after_blend_hook).  None removes the hook."""
    _AFTER_FEATURES_HOOK[0] = fn


class _GlueLink:
    """What ties a _SurfelFeatures node to the rasterizer node that consumes its outputs (the glue epilogue, MrgsRasterGrads::glue_params):
    `raw` = the node's nine raw parameter tensors; the rasterizer's backward leaves their gradients in `results`, the node's own backward
    hands them on instead of launching its kernel."""
    __slots__ = ("raw", "results", "viewmatrix")

    def __init__(self, raw, viewmatrix=None):
        self.raw, self.results, self.viewmatrix = raw, None, viewmatrix       # viewmatrix: the "pgsr" rows' (plane distance in channel 8)


class _LastLink(__import__("threading").local):      # the link of the most recent _SurfelFeatures.forward on THIS thread (render_surfel picks it up)
    def __init__(self):
        self.link = None

    def __getitem__(self, i):
        return self.link

    def __setitem__(self, i, v):
        self.link = v


_LAST_LINK = _LastLink()
_FUSE_GLUE = os.environ.get("MRGS_NO_GLUE_EPILOGUE", "0") != "1"


class _SurfelFeatures(torch.autograd.Function):
    """mrgs_surfel_features_forward/backward (include/mrgs.h): raw GaussianModel parameters -> (opacity, scales, rotations,
    features[P,8]) in one kernel each way (checker: oracle/glue_oracle.py, the reference's own chain of torch ops)."""

    @staticmethod
    def forward(ctx, xyz, scaling, rotation, opacity, refl, rough, ori_color, ind_dc, ind_rest, campos, pass_xyz=False, viewmatrix=None,
                indirect_live=True):
        ctx.set_materialize_grads(False)   # unused outputs arrive as None in backward, not as zero-filled tensors
        ctx.indirect_live = bool(indirect_live)
        if not xyz.is_cuda:
            raise RuntimeError("surfel_features needs CUDA(HIP) tensors: the per-gaussian glue runs in libmrgs.so, there is no CPU path")
        ts = [_c(t) for t in (xyz, scaling, rotation, opacity, refl, rough, ori_color, ind_dc, ind_rest, campos)]
        # viewmatrix ("pgsr"): feature rows of 12 floats, channel 8 = the plane distance of get_distance, 9..11 zero (MrgsSurfelParams)
        vm = None if viewmatrix is None else _c(viewmatrix)
        P, dev = ts[0].shape[0], ts[0].device
        L = _lib.lib()
        prm = MrgsSurfelParams(P, *[_p(t) for t in ts], _p(vm))
        o = dict(dtype=torch.float32, device=dev)
        op, sc, rot = torch.empty((P, 1), **o), torch.empty((P, 2), **o), torch.empty((P, 4), **o)
        feat = torch.empty((P, 8 if vm is None else 12), **o)
        with _lib.guard(dev):
            st = _lib.stream_ptr(dev)
            _lib.check(L.mrgs_surfel_features_forward(ctypes.byref(prm), _p(op), _p(sc), _p(rot), _p(feat), st))
        ctx.save_for_backward(*ts, *(() if vm is None else (vm,)))
        ctx.link = _LAST_LINK[0] = _GlueLink(ts[:9], vm)
        if pass_xyz:
            # the centres as a fifth output (the input itself): whoever consumes THEM -- the rasterizer -- sends its gradient through this
            # node, whose backward kernel adds it to its own: one sum inside a kernel instead of autograd's accumulation kernel
            return op, sc, rot, feat, xyz
        return op, sc, rot, feat

    @staticmethod
    def backward(ctx, g_op, g_sc, g_rot, g_feat, g_xyz=None):
        *ts, = ctx.saved_tensors
        vm = ts.pop() if len(ts) == 11 else None
        P, dev = ts[0].shape[0], ts[0].device
        fused, ctx.link.results = ctx.link.results, None
        if fused is not None and all(g is None for g in (g_op, g_sc, g_rot, g_feat, g_xyz)):
            # the rasterizer's per-gaussian backward has applied this node's backward already (the glue epilogue): nothing to launch
            if _AFTER_FEATURES_HOOK[0] is not None:
                _AFTER_FEATURES_HOOK[0](None)              # (only without a reader of the indirect radiance: its factor is structurally zero)
            return (*fused, None, None, None, None)
        L = _lib.lib()
        prm = MrgsSurfelParams(P, *[_p(t) for t in ts], _p(vm))
        outs = [torch.empty_like(t) for t in ts[:9]]
        grads = MrgsSurfelGrads(*[_p(t) for t in outs])
        gs = [None if g is None else _c(g) for g in (g_op, g_sc, g_rot, g_feat, g_xyz)]
        with _lib.guard(dev):
            st = _lib.stream_ptr(dev)
            _lib.check(L.mrgs_surfel_features_backward(ctypes.byref(prm), _p(gs[0]), _p(gs[1]), _p(gs[2]), _p(gs[3]), ctypes.byref(grads),
                                                       _p(gs[4]), st))
        if fused is not None:          # somebody else read this node's outputs as well: their share through the kernel, plus the epilogue's
            outs = [a + b for a, b in zip(outs, fused)]
        if _AFTER_FEATURES_HOOK[0] is not None:
            # (indirect_live False: the caller's step does not look at the blended indirect radiance -- its gradient is zero by the structure of
            #  the step, on every rank: the hook is told so instead of being handed a tensor of zeros to gather)
            _AFTER_FEATURES_HOOK[0](outs[7] if ctx.indirect_live else None)
        return (*outs, None, None, None, None)


def surfel_features(pc, camera_center, pass_xyz=False, viewmatrix=None, indirect_live=True):
    """(opacity[P,1], scales[P,2], rotations[P,4], features[P,8]) for `render_surfel` from the raw parameters of `pc`; with `pass_xyz`
    also the centres [P,3] as an output of the same node (hand THOSE to the rasterizer: its dL/dmeans3D is then summed with this node's
    own gradient of the centres inside the backward kernel).  `viewmatrix` (the camera's world_view_transform; "pgsr" flavour): features
    [P,12] with the plane distance of get_distance in channel 8 and zeros behind it."""
    return _SurfelFeatures.apply(pc._xyz, pc._scaling, pc._rotation, pc._opacity, pc._refl_strength, pc._roughness, pc._ori_color,
                                 pc._indirect_dc, pc._indirect_rest, camera_center, bool(pass_xyz), viewmatrix, bool(indirect_live))


_MAPS_FRAME_CACHE = {}


def _maps_frame(view, depth_ratio):
    """Camera constants of depths_to_points (utils/point_utils.py:9-24) for the fused kernels.  Built once per camera on the
    host (float64) from the camera's matrices and cached: the reference rebuilds them with ~10 small GPU kernels per view."""
    wvt, fpt = view.world_view_transform, view.full_proj_transform
    key = (wvt.data_ptr(), fpt.data_ptr(), wvt._version, fpt._version, int(view.image_width), int(view.image_height))
    ent = _MAPS_FRAME_CACHE.get(key)
    # An entry keeps its two matrices alive, so their addresses cannot be handed to other tensors while it exists; a hit must be
    # the very same tensor objects' storage at the same version (in-place writes bump _version).
    if ent is not None and not (ent[3].data_ptr() == wvt.data_ptr() and ent[4].data_ptr() == fpt.data_ptr()):
        ent = None
    if ent is None:
        import numpy as np
        wv = wvt.detach().cpu().double().numpy()
        fp = fpt.detach().cpu().double().numpy()
        W, H = int(view.image_width), int(view.image_height)
        c2w = np.linalg.inv(wv.T)
        ndc2pix = np.array([[W / 2, 0, 0, W / 2], [0, H / 2, 0, H / 2], [0, 0, 0, 1]], dtype=np.float64).T
        intrins = ((c2w.T @ fp) @ ndc2pix)[:3, :3].T
        M = c2w[:3, :3] @ np.linalg.inv(intrins)
        ent = (wv[:3, :3].reshape(-1).tolist(), M.reshape(-1).tolist(), c2w[:3, 3].tolist(), wvt, fpt)
        if len(_MAPS_FRAME_CACHE) > 4096:
            _MAPS_FRAME_CACHE.clear()
        _MAPS_FRAME_CACHE[key] = ent
    fr = MrgsMapsFrame()
    fr.H, fr.W = int(view.image_height), int(view.image_width)
    for i in range(9):
        fr.view_rot[i] = ent[0][i]
        fr.ray_matrix[i] = ent[1][i]
    for i in range(3):
        fr.ray_origin[i] = ent[2][i]
    fr.depth_ratio = float(depth_ratio)
    return fr


class _SurfelMaps(torch.autograd.Function):
    """mrgs_surfel_maps_forward/backward: allmap -> (rend_normal, surf_depth, surf_normal, normal_map, rend_alpha, rend_dist);
    the last two are the reference's plain slices of allmap, routed through here so that their gradients enter the one
    backward kernel instead of two extra [7,H,W] accumulation kernels."""

    @staticmethod
    def forward(ctx, allmap, fr, want_surf_normal, want_normal_map, twin_alpha=False, rend_distance=None):
        ctx.set_materialize_grads(False)   # unused outputs arrive as None in backward, not as zero-filled tensors
        if not allmap.is_cuda:
            raise RuntimeError("the fused map kernels need CUDA(HIP) tensors: there is no CPU path")
        allmap = _c(allmap)
        H, W, dev = fr.H, fr.W, allmap.device
        # "pgsr": the blended plane distance [1,H,W]; surf_depth is then the flavour's unbiased depth, formed inside the kernels (fr.pgsr_fx/fy)
        rd = None if rend_distance is None else _c(rend_distance)
        fr.rend_distance, fr.g_rend_distance = _p(rd), None
        o = dict(dtype=torch.float32, device=dev)
        rn, sd = torch.empty((3, H, W), **o), torch.empty((1, H, W), **o)
        sn = torch.empty((3, H, W), **o) if want_surf_normal else None
        nm = torch.empty((H, W, 3), **o) if want_normal_map else None
        # the plain slices allmap[1:2] / allmap[6:7] of the reference, written by the same kernel; with `twin_alpha` a second copy of the
        # alpha map for a second consumer (render_surfel's shading), whose gradient then comes back on a pointer of its own instead of
        # through autograd's accumulation kernel
        ra_rd = torch.empty((3 if twin_alpha else 2, 1, H, W), **o)
        with _lib.guard(dev):
            st = _lib.stream_ptr(dev)
            _lib.check(_lib.lib().mrgs_surfel_maps_forward(ctypes.byref(fr), _p(allmap), _p(rn), _p(sd), _p(sn), _p(nm), _p(ra_rd[0]), _p(ra_rd[1]),
                                                           _p(ra_rd[2]) if twin_alpha else None, st))
        ctx.save_for_backward(allmap, *(() if rd is None else (rd,)))
        ctx.fr = fr
        outs = (rn, sd, sn if sn is not None else rn.new_empty(0), nm if nm is not None else rn.new_empty(0))
        ctx.mark_non_differentiable(*[t for t in outs[2:] if t.numel() == 0])
        if twin_alpha:
            return (*outs, ra_rd[0], ra_rd[1], ra_rd[2])
        return (*outs, ra_rd[0], ra_rd[1])

    @staticmethod
    def backward(ctx, g_rn, g_sd, g_sn, g_nm, g_ra, g_rd, g_ra2=None):
        allmap, *rest = ctx.saved_tensors
        dev = allmap.device
        g = [None if (t is None or t.numel() == 0) else _c(t) for t in (g_rn, g_sd, g_sn, g_nm, g_ra, g_rd, g_ra2)]
        g_allmap = torch.empty_like(allmap)
        g_dist = torch.empty_like(rest[0]) if rest else None          # gradient of the plane-distance map ("pgsr")
        fr = ctx.fr
        fr.rend_distance, fr.g_rend_distance = (_p(rest[0]), _p(g_dist)) if rest else (None, None)
        with _lib.guard(dev):
            st = _lib.stream_ptr(dev)
            _lib.check(_lib.lib().mrgs_surfel_maps_backward(ctypes.byref(fr), _p(allmap), _p(g[0]), _p(g[1]), _p(g[2]), _p(g[3]), _p(g[4]), _p(g[5]),
                                                            _p(g[6]), _p(g_allmap), st))
        return g_allmap, None, None, None, None, g_dist


class _SurfelComposite(torch.autograd.Function):
    """mrgs_surfel_composite_forward/backward: (base, refl, specular, alpha, bg) -> (render, diffuse)."""

    @staticmethod
    def forward(ctx, base, refl, spec, alpha, bg, srgb):
        ctx.set_materialize_grads(False)   # unused outputs arrive as None in backward, not as zero-filled tensors
        base, refl, spec, alpha, bg = _c(base), _c(refl), _c(spec), _c(alpha), _c(bg)
        H, W, dev = base.shape[1], base.shape[2], base.device
        render, diffuse = torch.empty_like(base), torch.empty_like(base)
        with _lib.guard(dev):
            st = _lib.stream_ptr(dev)
            _lib.check(_lib.lib().mrgs_surfel_composite_forward(H, W, int(bool(srgb)), _p(base), _p(refl), _p(spec), _p(alpha), _p(bg),
                                                                _p(render), _p(diffuse), st))
        ctx.save_for_backward(base, refl, spec, bg)
        ctx.srgb = int(bool(srgb))
        return render, diffuse

    @staticmethod
    def backward(ctx, g_render, g_diffuse):
        base, refl, spec, bg = ctx.saved_tensors
        H, W, dev = base.shape[1], base.shape[2], base.device
        g_render = None if g_render is None else _c(g_render)
        g_diffuse = None if g_diffuse is None else _c(g_diffuse)
        g_base, g_spec = torch.empty_like(base), torch.empty_like(spec)
        g_refl, g_alpha = torch.empty_like(refl), torch.empty_like(refl)
        with _lib.guard(dev):
            st = _lib.stream_ptr(dev)
            _lib.check(_lib.lib().mrgs_surfel_composite_backward(H, W, ctx.srgb, _p(base), _p(refl), _p(spec), _p(bg), _p(g_render), _p(g_diffuse),
                                                                 _p(g_base), _p(g_refl), _p(g_spec), _p(g_alpha), None, 0, st))
        return g_base, g_refl, g_spec, g_alpha, None, None


def pgsr_unbiased_depth(allmap, rend_distance, viewpoint_camera):
    """The eighth all-map channel of the "pgsr" flavour (`allmap[7:8]`, gaussian_renderer/__init__.py:64-69 with the shipped
    arguments/config.py FLAG): the depth at which the pixel's ray meets the blended PLANE, from the blended plane distance and the blended
    view-space normal -- distance / -(normal . ray), ray = K^-1 (x, y, 1).  PARITY UNPINNED: the reference takes the channel from
    `diff_surfel_rasterization2`, which is not in its tree; this is the published PGSR definition (Chen et al. 2024,
    "unbiased depth": D / (N . K^-1 p~) with the sign of a camera-facing normal), applied to what this rasterizer blends: `rend_distance`
    = sum w |n . c| (get_distance, :30-40) and allmap[2:5] = sum w n, both un-normalised sums, so the accumulated alpha cancels.  The
    ray goes through pixel (x, y) of the rasterizer's own image plane (principal point (W - 1) / 2, the ndc2pix of forward.cu:114-118),
    so the depth is consistent with where the surfels were splatted.  Empty pixels give 0 / 0; callers apply nan_to_num as the
    reference does.  The product path evaluates this inside mrgs_surfel_maps_forward / _backward (MrgsMapsFrame::rend_distance); this
    function is the torch statement the kernels are tested against (tests/test_shading.py)."""
    H, W = int(viewpoint_camera.image_height), int(viewpoint_camera.image_width)
    fx = W / (2.0 * math.tan(viewpoint_camera.FoVx * 0.5))
    fy = H / (2.0 * math.tan(viewpoint_camera.FoVy * 0.5))
    dev, dt = allmap.device, allmap.dtype
    rx = (torch.arange(W, device=dev, dtype=dt) - 0.5 * (W - 1)) / fx
    ry = (torch.arange(H, device=dev, dtype=dt) - 0.5 * (H - 1)) / fy
    n_dot_ray = allmap[2] * rx[None, :] + allmap[3] * ry[:, None] + allmap[4]
    return rend_distance / (-n_dot_ray).unsqueeze(0)


def compute_2dgs_normal_and_regularizations(allmap, viewpoint_camera, pipe, return_depth_normal=True, return_normal_map=False,
                                            rend_distance=None, twin_alpha=False):
    """gaussian_renderer/__init__.py:42-90, one HIP kernel each way (`mrgs_surfel_maps_*`).  With `return_normal_map` the
    dictionary also holds render_surfel's `normal_map` [H,W,3] = render_normal / max(alpha, 1e-6) (:419-421).
    `rend_distance` (the "pgsr" flavour's blended plane distance): surf_depth is then the flavour's unbiased depth
    (`nan_to_num(allmap[7])`, :64-69; pgsr_unbiased_depth) instead of the expected / median mix, and surf_normal its finite differences."""
    fr = _maps_frame(viewpoint_camera, pipe.depth_ratio)
    if rend_distance is not None:
        # the unbiased depth is formed inside the kernels from the plane-distance map (round 5; pgsr_unbiased_depth is the torch statement
        # of the same expression, kept as the kernels' checker): no eight torch kernels and no [7,H,W] concatenation in front of them
        H, W = int(viewpoint_camera.image_height), int(viewpoint_camera.image_width)
        fr.pgsr_fx = W / (2.0 * math.tan(viewpoint_camera.FoVx * 0.5))
        fr.pgsr_fy = H / (2.0 * math.tan(viewpoint_camera.FoVy * 0.5))
    rn, sd, sn, nm, ra, rd, *twin = _SurfelMaps.apply(allmap, fr, bool(return_depth_normal), bool(return_normal_map), bool(twin_alpha), rend_distance)
    out = {"render_alpha": ra, "render_normal": rn, "render_depth_median": None, "render_depth_expected": None,
           "render_dist": rd, "surf_depth": sd, "surf_normal": sn if return_depth_normal else None}
    if return_normal_map:
        out["normal_map"] = nm
    if twin_alpha:
        out["render_alpha_twin"] = twin[0]      # the same map for a second consumer (see _SurfelMaps)
    return out


_BLACK = {}


def _black_like(bg_color):
    """A zero background of bg_color's shape / device / dtype, shared between renders (read-only: nothing writes a background): the
    reference fills a fresh one per render."""
    key = (bg_color.device, bg_color.dtype, tuple(bg_color.shape))
    t = _BLACK.get(key)
    if t is None:
        if len(_BLACK) > 16:
            _BLACK.clear()
        t = _BLACK[key] = torch.zeros_like(bg_color)
    return t


def _visibility(rasterizer, radii):
    """radii > 0: the mask the rasterizer's forward wrote next to the radii (GaussianRasterizer.visible), else the torch comparison."""
    vis = getattr(rasterizer, "visible", None)
    return vis if vis is not None and vis.shape == radii.shape else radii > 0


def _raster_settings(viewpoint_camera, pc, pipe, bg_color, scaling_modifier):
    return GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5),
        bg=_black_like(bg_color),                           # the rasterizer always composites over black (__init__.py:247)
        scale_modifier=scaling_modifier, viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform, sh_degree=pc.active_sh_degree, campos=viewpoint_camera.camera_center,
        prefiltered=False, debug=getattr(pipe, "debug", False))


class _SplitChannels(torch.autograd.Function):
    """feature maps [S,H,W] -> (maps[:k], maps[k:k+1]) as VIEWS, with ONE gradient assembly in the backward: the two slices of the
    "pgsr" flavour's rasterized channels (the eight material maps and the plane distance, gaussian_renderer/__init__.py:372-378, 411-413)
    would otherwise each pad their gradient to [S,H,W] with a fill and a copy and meet in an accumulation kernel."""

    @staticmethod
    def forward(ctx, maps, k, tail_unread=False):
        # tail_unread: the maps behind k + 1 are padding whose gradient the rasterizer's backward does not read (features_live)
        ctx.set_materialize_grads(False)
        ctx.k, ctx.shape, ctx.tail_unread = k, tuple(maps.shape), bool(tail_unread)
        return maps[:k], maps[k:k + 1]

    @staticmethod
    def backward(ctx, g_head, g_one):
        if g_head is None and g_one is None:
            return None, None, None
        k, shape = ctx.k, ctx.shape
        ref = g_head if g_head is not None else g_one
        hb = getattr(g_head, "_base", None) if g_head is not None else None
        from . import shading as _sh
        if hb is not None and tuple(hb.shape) == shape and g_head.storage_offset() == 0 and hb.is_contiguous() and hb.dtype == ref.dtype and \
                _sh.take_owned_stack(hb):
            # the producer laid its maps out as the head of a stack it allocated FOR this node (shading._SurfelShade.backward marks it):
            # nothing to copy.  A base of the right shape that is not so marked (e.g. the gradient of torch.cat((the eight maps, other
            # maps)): its rows behind the head belong to another branch) is left alone and copied from.
            g = hb
        else:
            g = torch.empty(shape, dtype=ref.dtype, device=ref.device)
            (g[:k].copy_(g_head) if g_head is not None else g[:k].zero_())
        (g[k:k + 1].copy_(g_one) if g_one is not None else g[k:k + 1].zero_())
        if shape[0] > k + 1 and not ctx.tail_unread:
            g[k + 1:].zero_()
        return g, None, None


def cov3D_precomp_of(pc, viewpoint_camera, scaling_modifier=1.0):
    """pipe.compute_cov3D_python (gaussian_renderer/__init__.py:136-147, 276-287, 572-583; optix_utils.py:138-150): the splat-to-pixel
    matrices built in torch from `pc.get_covariance` and handed to the rasterizer as `cov3D_precomp` [P,9] instead of scales /
    rotations -- the reference's own formula, term for term.  (Its normal-consistency caveat holds here too: with precomputed matrices
    the rasterizer's normal is the third column of the matrix it is given, :137.)  Off by default (arguments/__init__.py); torch ops,
    not fused."""
    splat2world = pc.get_covariance(scaling_modifier)
    W, H = int(viewpoint_camera.image_width), int(viewpoint_camera.image_height)
    near, far = float(viewpoint_camera.znear), float(viewpoint_camera.zfar)
    fpt = viewpoint_camera.full_proj_transform
    ndc2pix = torch.tensor([[W / 2, 0, 0, (W - 1) / 2], [0, H / 2, 0, (H - 1) / 2], [0, 0, far - near, near], [0, 0, 0, 1]],
                           dtype=torch.float32, device=fpt.device).T
    world2pix = fpt.float() @ ndc2pix
    return (splat2world[:, [0, 1, 3]] @ world2pix[:, [0, 1, 3]]).permute(0, 2, 1).reshape(-1, 9)      # column major, as glm wants it


_ZERO_POINTS = {}


def _screenspace_points(pc):
    """screenspace_points of the reference (gaussian_renderer/__init__.py:229-233): a zero tensor whose .grad receives the
    2D-mean gradients.  A leaf here (the reference adds 0 and calls retain_grad(): same .grad, one kernel more)."""
    xyz = pc.get_xyz
    key = (xyz.device, xyz.shape[0], xyz.dtype)
    z = _ZERO_POINTS.get(key)
    if z is None:
        if len(_ZERO_POINTS) > 8:
            _ZERO_POINTS.clear()
        z = _ZERO_POINTS[key] = torch.zeros_like(xyz)
    # a new leaf over the shared zeros (nothing reads or writes its values: the rasterizer only routes a gradient to it): no fill per view
    return z.detach().requires_grad_(True)


def get_distance(scaling_modifier, means3D, viewpoint_camera, pc):
    """gaussian_renderer/envgs_renderer.py:30-38: |facing normal . centre| in the camera frame, [P,1] (the plane distance the "pgsr"
    flavour rasterizes as its last feature channel)."""
    Wv = viewpoint_camera.world_view_transform
    d = pc.get_xyz - viewpoint_camera.camera_center
    normal_cam = pc.get_normal(scaling_modifier, d / d.norm(dim=1, keepdim=True)) @ Wv[:3, :3]
    centre_cam = means3D @ Wv[:3, :3] + Wv[3, :3]
    return (normal_cam * centre_cam).sum(-1).abs().unsqueeze(-1)


def _asg_indirect_of(pc, viewpoint_camera, scaling_modifier):
    """pipe.use_asg (gaussian_renderer/__init__.py:312-336, 604-627; off by default, arguments/__init__.py:101): the indirect radiance of
    every gaussian from its anisotropic-spherical-gaussian lobes along the mirror direction, in the frame of its facing normal.  Torch
    ops on [P,32,.] tensors, as in the reference (an off-default branch: not fused)."""
    from .gs_utils import asg_indirect
    dir_pp = pc.get_xyz - viewpoint_camera.camera_center
    dir_pp_normalized = dir_pp / dir_pp.norm(dim=1, keepdim=True)
    normals = pc.get_normal(scaling_modifier, dir_pp_normalized)
    w_o = -dir_pp_normalized
    reflection = 2 * torch.sum(normals * w_o, dim=1, keepdim=True) * normals - w_o
    return asg_indirect(pc.get_asg, pc.asg_param, normals, reflection)


@deferred_raster_count
def render_initial(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None, srgb=False, opt=None, flag="2dgs"):
    """gaussian_renderer/__init__.py:94-220: diffuse-only surfel rendering (S = 0 in the 2dgs flavour).  flag "pgsr"
    (arguments/config.py:1): the plane distance of get_distance rides as the one feature channel and comes back as "rend_distance"
    (:170-176, 215-218) -- blended by the vendored rasterizer's rule -- and surf_depth / surf_normal come from the flavour's unbiased
    depth (pgsr_unbiased_depth; parity unpinned)."""
    means2D = _screenspace_points(pc)
    dist_feature = get_distance(scaling_modifier, pc.get_xyz, viewpoint_camera, pc) if flag != "2dgs" else None
    rasterizer = GaussianRasterizer(raster_settings=_raster_settings(viewpoint_camera, pc, pipe, bg_color, scaling_modifier))
    shs, colors_precomp = ((pc._features_dc, pc._features_rest), None) if override_color is None else (None, override_color)
    if getattr(pipe, "compute_cov3D_python", False):       # :136-150
        scales, rotations, cov3D_precomp = None, None, cov3D_precomp_of(pc, viewpoint_camera, scaling_modifier)
    else:
        scales, rotations, cov3D_precomp = pc.get_scaling, pc.get_rotation, None
    contrib, rendered_image, rendered_features, radii, allmap = rasterizer(
        means3D=pc.get_xyz, means2D=means2D, shs=shs, colors_precomp=colors_precomp, features=dist_feature, opacities=pc.get_opacity,
        scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp)
    reg = compute_2dgs_normal_and_regularizations(allmap, viewpoint_camera, pipe,
                                                  rend_distance=rendered_features[0:1] if flag != "2dgs" else None)
    final_image = rendered_image
    if srgb:
        final_image = linear_to_srgb(final_image)
    final_image = final_image + bg_color[:, None, None] * (1 - reg["render_alpha"])
    out = {"render": final_image, "viewspace_points": means2D, "visibility_filter": _visibility(rasterizer, radii), "radii": radii,
           "rend_alpha": reg["render_alpha"], "rend_normal": reg["render_normal"], "rend_dist": reg["render_dist"],
           "surf_depth": reg["surf_depth"], "surf_normal": reg["surf_normal"]}
    if flag != "2dgs":
        out["rend_distance"] = rendered_features[0:1]
    return out


@deferred_raster_count
def render_surfel(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None, srgb=False, opt=None,
                  wo_render_img=False, normal_img_map=None, flag="2dgs"):
    """gaussian_renderer/__init__.py:225-483: per-gaussian material channels (S = 8: refl 1, roughness 1, albedo 3, indirect 3)
    blended by the rasterizer, then deferred split-sum shading."""
    if opt is None:
        opt = SimpleNamespace(indirect=False)
    means2D = _screenspace_points(pc)
    rasterizer = GaussianRasterizer(raster_settings=_raster_settings(viewpoint_camera, pc, pipe, bg_color, scaling_modifier))
    means3D = pc.get_xyz
    shs, colors_precomp = ((pc._features_dc, pc._features_rest), None) if override_color is None else (None, override_color)

    # activations, facing normal, mirror direction, indirect radiance along it and the feature concat (__init__.py:338-355):
    # one HIP kernel each way
    # (the centres come back as an output of the same node: the rasterizer's dL/dmeans3D joins the node's own in its backward kernel)
    # "pgsr": + the plane distance as a ninth channel, back as "rend_distance" (:348-355, 411-413, 478-480) -- from the same kernel, in
    # rows padded to twelve floats (the blend kernels move feature rows in 16-byte pieces); get_distance (torch) with the ASG lobes only
    use_asg = bool(getattr(pipe, "use_asg", False))
    fused_distance = flag != "2dgs" and not use_asg
    # (indirect_live: the blended indirect radiance reaches an output only under opt.indirect -- :423-430, 472-473 --; without it the SH
    #  indirect term's gradient is zero by construction, which a view-parallel step's exchange is told: dist.SurfelGradReducer.begin_early_ind)
    opacities, scales, rotations, features, means3D = surfel_features(pc, viewpoint_camera.camera_center, pass_xyz=True,
                                                                      viewmatrix=viewpoint_camera.world_view_transform if fused_distance else None,
                                                                      indirect_live=bool(getattr(opt, "indirect", False)) and not use_asg)
    if use_asg:                 # the lobes instead of the SH indirect term in channels 5..7 (:312-336)
        features = torch.cat((features[:, :5], _asg_indirect_of(pc, viewpoint_camera, scaling_modifier)), dim=-1)
        if flag != "2dgs":
            features = torch.cat((features, get_distance(scaling_modifier, means3D, viewpoint_camera, pc)), dim=-1)

    cov3D_precomp = None
    if getattr(pipe, "compute_cov3D_python", False):       # :276-290 (the fused node's scales / rotations stay unused: their gradients are None)
        scales, rotations, cov3D_precomp = None, None, cov3D_precomp_of(pc, viewpoint_camera, scaling_modifier)
    padded = fused_distance and features.shape[1] == 12
    if padded:      # nine channels in rows of twelve floats: the blend kernels leave the padding out of their arithmetic (features_live)
        rasterizer.features_live = 9
    # The glue epilogue (MrgsRasterGrads::glue_params): nothing reads the blended indirect radiance and the rows are the fused node's own ->
    # the rasterizer's per-gaussian backward continues through the activations' backward in the same kernel, the node above launches none.
    indirect_live = bool(getattr(opt, "indirect", False)) and not use_asg
    # ("pgsr": with the padded rows of the fused node only -- nine channels in twelve floats -- ; the plane distance's gradient is part of the epilogue)
    if (_FUSE_GLUE and not indirect_live and not use_asg and (padded or not fused_distance) and cov3D_precomp is None and means3D.is_cuda
            and torch.is_grad_enabled()):
        rasterizer.glue = _LAST_LINK[0]
    _LAST_LINK[0] = None
    contrib, rendered_image, rendered_features, radii, allmap = rasterizer(
        means3D=means3D, means2D=means2D, shs=shs, colors_precomp=colors_precomp, features=features, opacities=opacities,
        scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp)
    rend_distance = None
    if flag != "2dgs":          # the eight material maps and the plane distance out of the 9 (12: padded rows) rasterized channels
        rendered_features, rend_distance = _SplitChannels.apply(rendered_features, 8, padded)
    elif rendered_features.shape[0] != 8:        # (a slice of the full range is still an autograd node: a zero fill and a copy of 8 maps)
        rendered_features = rendered_features[:8]

    base_color = rendered_image
    refl_strength, roughness_map = rendered_features[:1], rendered_features[1:2]
    albedo, indirect_light = rendered_features[2:5], rendered_features[5:8]

    reg = compute_2dgs_normal_and_regularizations(allmap, viewpoint_camera, pipe, return_depth_normal=(not wo_render_img),
                                                  return_normal_map=(not wo_render_img), rend_distance=rend_distance,
                                                  twin_alpha=(not wo_render_img))
    render_alpha, render_normal = reg["render_alpha"], reg["render_normal"]
    geo = {"viewspace_points": means2D, "visibility_filter": _visibility(rasterizer, radii), "radii": radii, "rend_alpha": render_alpha,
           "rend_normal": render_normal, "rend_dist": reg["render_dist"], "surf_depth": reg["surf_depth"], "surf_normal": reg["surf_normal"]}
    if rend_distance is not None:
        geo["rend_distance"] = rend_distance
    if wo_render_img:
        return {"refl_strength_map": refl_strength, "base_color_map": albedo, "roughness_map": roughness_map, **geo}

    # get_specular_color_surfel (utils/refl_utils.py:364-419) + (1 - refl) * base + specular, optional sRGB, background
    # (__init__.py:436-445) as one autograd node on the whole material map
    # opt.indirect (:421-431, INDIRECT_TYPE other than "raytracing_residual"): where the mirror ray hits the mesh of `pc.ray_tracer`
    # the blended indirect radiance replaces the environment; same node, plus the visibility and blend kernels
    indirect = bool(getattr(opt, "indirect", False))
    tracer = getattr(pc, "ray_tracer", None) if indirect else None
    final_image, diffuse_map, specular, extra_dict = shade_and_composite_surfel(
        pc.get_envmap, base_color, rendered_features, viewpoint_camera.HWK, viewpoint_camera.R, viewpoint_camera.T, reg["normal_map"],
        reg["render_alpha_twin"], bg_color, srgb, ray_tracer=tracer, surf_depth=reg["surf_depth"])   # (the shading's own copy of the alpha map)
    if srgb:
        albedo = linear_to_srgb(albedo)
        specular = linear_to_srgb(specular)
    out = {"render": final_image, "refl_strength_map": refl_strength, "diffuse_map": diffuse_map,
           "diffuse_map_ori": base_color, "specular_map": specular, "base_color_map": albedo, "roughness_map": roughness_map, **geo}
    if indirect:
        if tracer is None:      # :379-406 without a tracer: everything is visible
            extra_dict = {**extra_dict, "visibility": torch.ones_like(render_alpha), "indirect_light": indirect_light,
                          "indirect_color": torch.zeros_like(base_color)}
        background = bg_color[:, None, None] * (1 - render_alpha)
        out.update(extra_dict)
        out["indirect_color"] = diffuse_map + extra_dict["indirect_color"] + background          # :446-449
    return out


@deferred_raster_count
def render_volume(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None, srgb=False, opt=None, flag="2dgs"):
    """gaussian_renderer/__init__.py:521-749: every gaussian is shaded on its own (per-gaussian normal, mirror direction, split-sum
    weight, environment lookups: utils/refl_utils.py:426-484) and the rasterizer blends the shaded colour (`colors_precomp =
    specular + diffuse`) plus S = 11 material channels (roughness, refl, diffuse 3, specular 3, ori_color 3; 18 with opt.indirect:
    + visibility, indirect 3, direct_light 3).  SH-indirect branch (`pipe.use_asg` False) by default.  Environment lookups and the rasterizer run in
    libmrgs.so; the per-gaussian elementwise glue is torch, as in the reference.  `pipe.compute_cov3D_python`: cov3D_precomp_of."""
    if opt is None:
        opt = SimpleNamespace(indirect=False)
    means2D = _screenspace_points(pc)
    rasterizer = GaussianRasterizer(raster_settings=_raster_settings(viewpoint_camera, pc, pipe, bg_color, scaling_modifier))
    means3D, opacity = pc.get_xyz, pc.get_opacity
    refl, ori_color, roughness = pc.get_refl, pc.get_ori_color, pc.get_rough
    if getattr(pipe, "compute_cov3D_python", False):       # :572-586
        scales, rotations, cov3D_precomp = None, None, cov3D_precomp_of(pc, viewpoint_camera, scaling_modifier)
    else:
        scales, rotations, cov3D_precomp = pc.get_scaling, pc.get_rotation, None
    dir_pp = means3D - viewpoint_camera.camera_center
    dir_pp_normalized = dir_pp / dir_pp.norm(dim=1, keepdim=True)
    normals = pc.get_normal(scaling_modifier, dir_pp_normalized)
    w_o = -dir_pp_normalized
    reflection = 2 * torch.sum(normals * w_o, dim=1, keepdim=True) * normals - w_o
    if getattr(pipe, "use_asg", False):                                      # :604-627
        from .gs_utils import asg_indirect
        indirect = asg_indirect(pc.get_asg, pc.asg_param, normals, reflection)
    else:
        shs_indirect = pc.get_indirect.transpose(1, 2).reshape(-1, 3, (pc.max_sh_degree + 1) ** 2)
        indirect = torch.clamp_min(eval_sh(3, shs_indirect, reflection), 0.0)
    indirect_on = bool(getattr(opt, "indirect", False))
    if indirect_on:
        diffuse, specular, extra = get_full_color_volume_indirect(pc.get_envmap_2, means3D, ori_color, viewpoint_camera.HWK, viewpoint_camera.R,
                                                                  viewpoint_camera.T, normals.contiguous(), opacity, refl_strength=refl,
                                                                  roughness=roughness, pc=pc, indirect_light=indirect)
        features = torch.cat((roughness, refl, diffuse, specular, ori_color, extra["visibility"], indirect, extra["direct_light"]), dim=-1)
    else:
        diffuse, specular = get_full_color_volume(pc.get_envmap_2, means3D, ori_color, viewpoint_camera.HWK, viewpoint_camera.R,
                                                  viewpoint_camera.T, normals.contiguous(), opacity, refl_strength=refl, roughness=roughness)
        features = torch.cat((roughness, refl, diffuse, specular, ori_color), dim=-1)
    colors_precomp = specular + diffuse
    if flag != "2dgs":          # "pgsr": + the plane distance as the last channel, back as "rend_distance" (:657-661, 744-746)
        features = torch.cat((features, get_distance(scaling_modifier, means3D, viewpoint_camera, pc)), dim=-1)
    contrib, rendered_image, rendered_features, radii, allmap = rasterizer(
        means3D=means3D, means2D=means2D, shs=None, colors_precomp=colors_precomp, features=features, opacities=opacity, scales=scales,
        rotations=rotations, cov3D_precomp=cov3D_precomp)
    full_color = rendered_image
    render_roughness, render_refl_strength = rendered_features[:1], rendered_features[1:2]
    render_diffuse_color, render_specular_color = rendered_features[2:5], rendered_features[5:8]
    render_ori_color = rendered_features[8:11]
    reg = compute_2dgs_normal_and_regularizations(allmap, viewpoint_camera, pipe, rend_distance=rendered_features[-1:] if flag != "2dgs" else None)
    render_alpha = reg["render_alpha"]
    if srgb:
        render_diffuse_color, render_specular_color = linear_to_srgb(render_diffuse_color), linear_to_srgb(render_specular_color)
        full_color = linear_to_srgb(full_color)
    final_image = full_color + bg_color[:, None, None] * (1 - render_alpha)
    out = {"render": final_image, "refl_strength_map": render_refl_strength, "diffuse_map": render_diffuse_color,
           "specular_map": render_specular_color, "base_color_map": render_ori_color, "roughness_map": render_roughness,
           "viewspace_points": means2D, "visibility_filter": _visibility(rasterizer, radii), "radii": radii, "rend_alpha": render_alpha,
           "rend_normal": reg["render_normal"], "rend_dist": reg["render_dist"], "surf_depth": reg["surf_depth"], "surf_normal": reg["surf_normal"]}
    if indirect_on:
        out.update({"visibility": rendered_features[11:12], "indirect_light": rendered_features[12:15], "direct_light": rendered_features[15:18]})
    if flag != "2dgs":
        out["rend_distance"] = rendered_features[-1:]
    return out


# ---- traced indirect light (second half) ---------------------------------------------------------------
def _pixel_rays_unnormalized(HWK, R, T, device, dtype=torch.float32):
    """sample_camera_rays_unnormalize (utils/refl_utils.py:75-93): per pixel the world-space vector from the camera centre to the point
    at view depth 1 behind the pixel (x, y integer pixel coordinates), and the camera centre.  R is Camera.R (stored transposed)."""
    H, W, K = HWK
    Kinv = torch.linalg.inv(torch.as_tensor(K, dtype=dtype, device=device))
    ys, xs = torch.meshgrid(torch.arange(H, device=device, dtype=dtype), torch.arange(W, device=device, dtype=dtype), indexing="ij")
    pix_cam = torch.stack([xs, ys, torch.ones_like(xs)], dim=-1) @ Kinv.T
    R = torch.as_tensor(R, dtype=dtype, device=device)
    T = torch.as_tensor(T, dtype=dtype, device=device)
    Rw = R.T                                               # the world-to-camera rotation
    rays_o = (-Rw.T @ T.unsqueeze(-1)).flatten()
    rays_d = (pix_cam - T[None, None]).reshape(-1, 3) @ Rw - rays_o[None]
    return rays_d.reshape(H, W, 3), rays_o


def mirror_rays_torch(viewpoint_camera, normal_map, surf_depth):
    """The ray set of render_indirect / render_surfel_with_envgs (gaussian_renderer/__init__.py:496-505, envgs_renderer.py:717-724)
    with torch ops (any device): the checker of mrgs_mirror_rays_*."""
    H, W, _ = viewpoint_camera.HWK
    rays_cam, rays_o = _pixel_rays_unnormalized(viewpoint_camera.HWK, viewpoint_camera.R, viewpoint_camera.T, surf_depth.device, normal_map.dtype)
    hit = rays_o + surf_depth.reshape(H, W, 1) * rays_cam
    w_o = safe_normalize(-rays_cam)
    n = normal_map.reshape(H, W, 3)
    refl = safe_normalize(2 * n * torch.sum(w_o * n, dim=-1, keepdim=True) - w_o)          # reflection(), utils/refl_utils.py:95-98
    return hit + 1e-3 * refl, refl


class _MirrorRays(torch.autograd.Function):
    """mrgs_mirror_rays_forward / _backward: one launch each way."""

    @staticmethod
    def forward(ctx, normal_map, surf_depth, Kinv, R, T):
        L = _lib.lib()
        H, W = normal_map.shape[0], normal_map.shape[1]
        dev = normal_map.device
        nm = normal_map.detach()
        sd = surf_depth.detach().reshape(H, W).contiguous().float()
        ray_o = torch.empty(H, W, 3, dtype=torch.float32, device=dev)
        ray_d = torch.empty(H, W, 3, dtype=torch.float32, device=dev)
        kinv = (ctypes.c_float * 9)(*Kinv)
        m = _lib.MrgsStridedMap(nm.data_ptr(), nm.stride(0), nm.stride(1), nm.stride(2))
        with _lib.guard(dev):
            _lib.check(L.mrgs_mirror_rays_forward(H, W, kinv, _p(R), _p(T), ctypes.byref(m), _p(sd), _p(ray_o), _p(ray_d),
                                                  _lib.stream_ptr(dev)))
        ctx.save_for_backward(nm, R, T)
        ctx.Kinv, ctx.depth_shape = Kinv, surf_depth.shape
        return ray_o, ray_d

    @staticmethod
    def backward(ctx, g_o, g_d):
        nm, R, T = ctx.saved_tensors
        L = _lib.lib()
        H, W = nm.shape[0], nm.shape[1]
        dev = nm.device
        z = lambda g: torch.zeros(H, W, 3, dtype=torch.float32, device=dev) if g is None else g.contiguous().float()
        g_o, g_d = z(g_o), z(g_d)
        g_n = torch.empty(H, W, 3, dtype=torch.float32, device=dev)
        g_sd = torch.empty(H, W, dtype=torch.float32, device=dev)
        kinv = (ctypes.c_float * 9)(*ctx.Kinv)
        m = _lib.MrgsStridedMap(nm.data_ptr(), nm.stride(0), nm.stride(1), nm.stride(2))
        with _lib.guard(dev):
            _lib.check(L.mrgs_mirror_rays_backward(H, W, kinv, _p(R), _p(T), ctypes.byref(m), _p(g_o), _p(g_d), _p(g_n), _p(g_sd),
                                                   _lib.stream_ptr(dev)))
        return g_n, g_sd.reshape(ctx.depth_shape), None, None, None


def _mirror_rays(viewpoint_camera, normal_map, surf_depth):
    """Mirror rays of every pixel of a rendered view: from the surface point along the mirror direction of the view ray about
    `normal_map`, origin moved 1e-3 along it (one HIP launch; `mirror_rays_torch` states the same with torch ops)."""
    import numpy as np
    H, W, K = viewpoint_camera.HWK
    dev = surf_depth.device
    Kinv = tuple(np.linalg.inv(np.asarray(K, dtype=np.float32)).astype(np.float32).reshape(-1).tolist())
    R = torch.as_tensor(viewpoint_camera.R, dtype=torch.float32, device=dev).contiguous()
    T = torch.as_tensor(viewpoint_camera.T, dtype=torch.float32, device=dev).contiguous()
    return _MirrorRays.apply(normal_map.reshape(H, W, 3).float(), surf_depth, Kinv, R, T)


class _MirrorRaysBlended(torch.autograd.Function):
    """mrgs_mirror_rays_blended_forward / _backward: the mirror rays straight from the rendered maps -- reflecting normal =
    safe_normalize(rend_normal / clamp_min(rend_alpha, 1e-6)) (gaussian_renderer/__init__.py:493-495) built inside the ray kernel."""

    @staticmethod
    def forward(ctx, rend_normal, rend_alpha, surf_depth, Kinv, R, T):
        L = _lib.lib()
        H, W = rend_normal.shape[1], rend_normal.shape[2]
        dev = rend_normal.device
        rn = rend_normal.detach().float()
        al = rend_alpha.detach().reshape(H, W).contiguous().float()
        sd = surf_depth.detach().reshape(H, W).contiguous().float()
        ray_o = torch.empty(H, W, 3, dtype=torch.float32, device=dev)
        ray_d = torch.empty(H, W, 3, dtype=torch.float32, device=dev)
        kinv = (ctypes.c_float * 9)(*Kinv)
        m = _lib.MrgsStridedMap(rn.data_ptr(), rn.stride(1), rn.stride(2), rn.stride(0))        # [3,H,W] read as [H,W,3]
        with _lib.guard(dev):
            _lib.check(L.mrgs_mirror_rays_blended_forward(H, W, kinv, _p(R), _p(T), ctypes.byref(m), _p(al), _p(sd), _p(ray_o), _p(ray_d),
                                                          _lib.stream_ptr(dev)))
        ctx.save_for_backward(rn, al, R, T)
        ctx.Kinv, ctx.shapes = Kinv, (rend_alpha.shape, surf_depth.shape)
        return ray_o, ray_d

    @staticmethod
    def backward(ctx, g_o, g_d):
        rn, al, R, T = ctx.saved_tensors
        L = _lib.lib()
        H, W = rn.shape[1], rn.shape[2]
        dev = rn.device
        z = lambda g: torch.zeros(H, W, 3, dtype=torch.float32, device=dev) if g is None else g.contiguous().float()
        g_o, g_d = z(g_o), z(g_d)
        g_n = torch.empty(H, W, 3, dtype=torch.float32, device=dev)
        g_al = torch.empty(H, W, dtype=torch.float32, device=dev)
        g_sd = torch.empty(H, W, dtype=torch.float32, device=dev)
        kinv = (ctypes.c_float * 9)(*ctx.Kinv)
        m = _lib.MrgsStridedMap(rn.data_ptr(), rn.stride(1), rn.stride(2), rn.stride(0))
        with _lib.guard(dev):
            _lib.check(L.mrgs_mirror_rays_blended_backward(H, W, kinv, _p(R), _p(T), ctypes.byref(m), _p(al), _p(g_o), _p(g_d), _p(g_n), _p(g_al),
                                                           _p(g_sd), _lib.stream_ptr(dev)))
        return g_n.permute(2, 0, 1), g_al.reshape(ctx.shapes[0]), g_sd.reshape(ctx.shapes[1]), None, None, None


_CAM_CONSTS = {}


def _fingerprint(x):
    """What a cached constant was built from: the bytes of a host array (a few dozen), address and version counter of a tensor."""
    if torch.is_tensor(x):
        return (x.data_ptr(), x._version, tuple(x.shape))
    import numpy as np
    return np.asarray(x).tobytes()


def _camera_consts(viewpoint_camera, dev):
    """K^-1 (host tuple), Camera.R and Camera.T as device tensors, built once per camera object AND pose: an entry is only reused while
    R, T and the intrinsics are what it was built from (a pose refinement or a resolution change on the same object rebuilds it)."""
    import numpy as np
    H, W, K = viewpoint_camera.HWK
    key = id(viewpoint_camera)
    stamp = (_fingerprint(viewpoint_camera.R), _fingerprint(viewpoint_camera.T), _fingerprint(K), int(H), int(W))
    ent = _CAM_CONSTS.get(key)
    if ent is None or ent[0] is not viewpoint_camera or ent[4] != dev or ent[5] != stamp:
        Kinv = tuple(np.linalg.inv(np.asarray(K, dtype=np.float32)).astype(np.float32).reshape(-1).tolist())
        R = torch.as_tensor(viewpoint_camera.R, dtype=torch.float32, device=dev).contiguous()
        T = torch.as_tensor(viewpoint_camera.T, dtype=torch.float32, device=dev).contiguous()
        if len(_CAM_CONSTS) > 4096:
            _CAM_CONSTS.clear()
        ent = _CAM_CONSTS[key] = (viewpoint_camera, Kinv, R, T, dev, stamp)      # holds the camera: its id stays its own
    return ent[1], ent[2], ent[3]


def _mirror_rays_blended(viewpoint_camera, rend_normal, rend_alpha, surf_depth):
    """Mirror rays of every pixel from render_surfel's maps: rend_normal [3,H,W], rend_alpha [1,H,W], surf_depth [1,H,W]."""
    Kinv, R, T = _camera_consts(viewpoint_camera, surf_depth.device)
    return _MirrorRaysBlended.apply(rend_normal, rend_alpha, surf_depth, Kinv, R, T)


class _TracedBlend(torch.autograd.Function):
    """final * (1 - specular) + specular * traced (gaussian_renderer/__init__.py:517), one kernel each way; `traced` [3,H,W] and `specular`
    [1,H,W] are read through their strides (the tracer's [H,W,C] tensors seen channel-first) and their gradients written the same way."""

    @staticmethod
    def forward(ctx, a, b, s):
        a = a.detach().contiguous().float()
        b, s = b.detach().float(), s.detach().float()
        H, W = a.shape[1], a.shape[2]
        if not (b.stride(1) == W * b.stride(2) and s.stride(1) == W * s.stride(2)):       # rows must follow each other: one pixel stride
            b, s = b.contiguous(), s.contiguous()
        out = torch.empty_like(a)
        with _lib.guard(a.device):
            _lib.check(_lib.lib().mrgs_traced_blend_forward(H, W, _p(a), _p(b), b.stride(0), b.stride(2), _p(s), s.stride(2), _p(out),
                                                            _lib.stream_ptr(a.device)))
        ctx.save_for_backward(a, b, s)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b, s = ctx.saved_tensors
        g = g.contiguous().float()
        H, W = a.shape[1], a.shape[2]
        ga = torch.empty_like(a)
        gb = torch.empty_strided(b.shape, b.stride(), dtype=torch.float32, device=a.device)      # g_b in b's own layout
        gs = torch.empty((1, H, W), dtype=torch.float32, device=a.device)
        with _lib.guard(a.device):
            _lib.check(_lib.lib().mrgs_traced_blend_backward(H, W, _p(a), _p(b), b.stride(0), b.stride(2), _p(s), s.stride(2), _p(g), _p(ga), _p(gb), _p(gs),
                                                             _lib.stream_ptr(a.device)))
        return ga, gb, gs


def render_indirect(indirect_renderer, viewpoint_camera, pc, pipe, bg_color, normal_map=None, surf_depth=None):
    """gaussian_renderer/envgs_renderer.py:716-731: the surfel set `pc` seen along the mirror rays of a rendered view."""
    ray_o, ray_d = _mirror_rays(viewpoint_camera, normal_map, surf_depth)
    return indirect_renderer.render_gaussians(viewpoint_camera, ray_o=ray_o, ray_d=ray_d, pcd=pc, pipe=pipe, bg_color=bg_color,
                                              start_from_first=True)


@deferred_raster_count
def render_surfel_with_envgs(indirect_renderer, viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None, srgb=False,
                             opt=None, wo_render_img=False, normal_img_map=None):
    """gaussian_renderer/__init__.py:486-520: render_surfel, then the same surfels traced along every pixel's mirror ray
    (`indirect_renderer`: HardwareRendering, here materialrefgs_amd.surfel_tracing) and blended in with the traced `specular`
    channel as weight; the tracer's dictionary is returned under "indirect_out"."""
    # wo_render_img / normal_img_map are handed on as the reference does (:487-489); with wo_render_img the dictionary has no "render"
    # and the lines below fail with the reference's own KeyError
    results = render_surfel(viewpoint_camera, pc, pipe, bg_color, scaling_modifier, override_color, srgb, opt, wo_render_img, normal_img_map)
    final_image = results["render"]
    # normal_map = safe_normalize(rend_normal / clamp_min(alpha, 1e-6)) (:493-495) is built inside the ray kernel
    ray_o, ray_d = _mirror_rays_blended(viewpoint_camera, results["rend_normal"], results["rend_alpha"], results["surf_depth"])
    traced = indirect_renderer(viewpoint_camera, ray_o=ray_o, ray_d=ray_d, pcd=pc, pipe=pipe, bg_color=bg_color, start_from_first=False)
    specular = traced["specular"]
    results["render"] = _TracedBlend.apply(final_image, traced["render"], specular) if final_image.is_cuda else \
        final_image * (1 - specular) + specular * traced["render"]
    results["indirect_out"] = traced
    return results


@deferred_raster_count
def render_surfel_with_envgs_sep(indirect_renderer, env, viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None, srgb=False,
                                 opt=None, wo_render_img=False, normal_img_map=None):
    """gaussian_renderer/envgs_renderer.py:771-807: as render_surfel_with_envgs, but the mirror rays see the separate ENVIRONMENT surfel
    set `env`, and the blend weight is render_surfel's per-pixel "specular_weight" ([H,W,3], returned with opt.indirect)."""
    results = render_surfel(viewpoint_camera, pc, pipe, bg_color, scaling_modifier, override_color, srgb, opt, wo_render_img, normal_img_map)
    weight = results["specular_weight"].permute(2, 0, 1)
    ray_o, ray_d = _mirror_rays_blended(viewpoint_camera, results["rend_normal"], results["rend_alpha"], results["surf_depth"])
    traced = indirect_renderer.render_gaussians(viewpoint_camera, ray_o=ray_o, ray_d=ray_d, pcd=env, pipe=pipe, bg_color=bg_color, start_from_first=True)
    traced["specular"] = weight
    results["render"] = results["render"] * (1 - weight) + weight * traced["render"]
    results["indirect_out"] = traced
    return results


@deferred_raster_count
def render_surfel2(indirect_renderer, env, viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None, srgb=False,
                   opt=None, wo_render_img=False, normal_img_map=None, flag="pgsr"):
    """gaussian_renderer/envgs_renderer.py:461-715, the last training stage (train_refnerf.py:1501-1504): render_surfel's material
    channels plus (flag "pgsr", arguments/config.py:1) the per-gaussian blend weight and plane distance (S = 10), the mirror rays of the
    view traced through the ENVIRONMENT surfel set `env` (render_indirect, :659), and that traced light standing in for the blended
    indirect radiance where the mesh occludes the environment (get_specular_color_surfel4, utils/refl_utils.py:302-362, with its
    `use_indirect_light_residual = False`).  `pipe.use_asg` (off by default, arguments/__init__.py:101): as in render_surfel.
    The reference binds `diff_surfel_rasterization2` here; its blending of feature channels is the vendored rasterizer's, which is what
    runs."""
    if opt is None:
        opt = SimpleNamespace(indirect=False)
    means2D = _screenspace_points(pc)
    settings = _raster_settings(viewpoint_camera, pc, pipe, _black_like(bg_color), scaling_modifier)     # bg = 0 as at :483
    rasterizer = GaussianRasterizer(raster_settings=settings)
    means3D = pc.get_xyz
    shs, colors_precomp = ((pc._features_dc, pc._features_rest), None) if override_color is None else (None, override_color)
    opacities, scales, rotations, features = surfel_features(pc, viewpoint_camera.camera_center)        # refl, rough, albedo 3, indirect 3
    if getattr(pipe, "use_asg", False):
        features = torch.cat((features[:, :5], _asg_indirect_of(pc, viewpoint_camera, scaling_modifier)), dim=-1)
    if flag != "2dgs":
        features = torch.cat((features, pc.get_specular, get_distance(scaling_modifier, means3D, viewpoint_camera, pc)), dim=-1)
    contrib, rendered_image, rendered_features, radii, allmap = rasterizer(
        means3D=means3D, means2D=means2D, shs=shs, colors_precomp=colors_precomp, features=features, opacities=opacities,
        scales=scales, rotations=rotations, cov3D_precomp=None)
    base_color = rendered_image
    refl_strength, roughness_map, albedo = rendered_features[:1], rendered_features[1:2], rendered_features[2:5]
    blend_weight = rendered_features[8:9]
    reg = compute_2dgs_normal_and_regularizations(allmap, viewpoint_camera, pipe, return_depth_normal=(not wo_render_img),
                                                  return_normal_map=True, rend_distance=rendered_features[-1:] if flag != "2dgs" else None)
    render_alpha = reg["render_alpha"]
    geo = {"viewspace_points": means2D, "visibility_filter": _visibility(rasterizer, radii), "radii": radii, "rend_alpha": render_alpha,
           "rend_normal": reg["render_normal"], "rend_dist": reg["render_dist"], "surf_depth": reg["surf_depth"], "surf_normal": reg["surf_normal"],
           "blend_weight": blend_weight}
    if flag != "2dgs":
        geo["rend_distance"] = rendered_features[-1:]
    if wo_render_img:
        return {"refl_strength_map": refl_strength, "base_color_map": albedo, "roughness_map": roughness_map, **geo}

    normal_map = reg["normal_map"]                                              # render_normal / max(alpha, 1e-6), not normalised (:655-657)
    indirect_results = render_indirect(indirect_renderer, viewpoint_camera, env, pipe, bg_color, normal_map, reg["surf_depth"])
    hwc = lambda m: m.permute(1, 2, 0)
    kw = dict(refl_strength=hwc(refl_strength), roughness=hwc(roughness_map), pc=pc, surf_depth=reg["surf_depth"])
    if getattr(opt, "indirect", False):
        kw["indirect_light"] = hwc(indirect_results["render"])
    specular, extra_dict = get_specular_color_surfel(pc.get_envmap, hwc(albedo), viewpoint_camera.HWK, viewpoint_camera.R, viewpoint_camera.T,
                                                     normal_map, hwc(render_alpha), **kw)
    final_image = (1 - refl_strength) * base_color + specular
    if srgb:
        final_image, albedo, specular = linear_to_srgb(final_image), linear_to_srgb(albedo), linear_to_srgb(specular)
    final_image = final_image + bg_color[:, None, None] * (1 - render_alpha)
    out = {"render": final_image, "refl_strength_map": refl_strength, "diffuse_map": (1 - refl_strength) * base_color,
           "diffuse_map_ori": base_color, "specular_map": specular, "base_color_map": albedo, "roughness_map": roughness_map, **geo,
           "indirect_out": indirect_results}
    if getattr(opt, "indirect", False):
        out.update(extra_dict)
    return out
