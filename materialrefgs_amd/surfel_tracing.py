"""Surfel ray tracing: the Python surface of the reference's `diff_surfel_tracing` extension and of `HardwareRendering`.

Mirrors (names, arguments, outputs)
  * `from diff_surfel_tracing import SurfelTracer, SurfelTracingSettings`          gaussian_renderer/optix_utils.py:7
      SurfelTracer().build_acceleration_structure(v, f, rebuild=True)              :76
      SurfelTracer()(ray_o, ray_d, v, means3D=..., grads3D=..., shs=..., colors_precomp=..., others_precomp=..., opacities=...,
                     scales=..., rotations=..., cov3D_precomp=..., tracer_settings=..., start_from_first=...)       :185-197
        -> rgb, dpt, acc, norm, dist, aux, mid, wet
  * `HardwareRendering` (optix_utils.py:14-271): get_disks, build_bvh, render_gaussians -> the same output dictionary.
The extension itself is not in the reference tree (OptiX, un-vendored): what it computes is defined in csrc/mrgs_surfel_trace.hip and
checked against oracle/surfel_trace_oracle.py; parity with the OptiX binary is unpinned.  The hierarchy is built on the
GPU by libmrgs (no host round trip), the tracing and its backward are HIP kernels; there is no CPU path.
"""
import ctypes
import math
from typing import NamedTuple

import torch
from torch import nn

from . import _lib
from .gs_utils import build_rotation, eval_sh

MID_CHANNELS = 16    # optix_utils.py:28-35: ray_o 3, ray_d 3, dpt 1, acc 1, norm 3, aux 2, rgb 3 per tracing depth


class SurfelTracingSettings(NamedTuple):
    """Fields as constructed at optix_utils.py:101-116."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool
    max_trace_depth: int = 0
    specular_threshold: float = 0.0


def _ptr(t):
    return None if t is None else t.data_ptr()      # (an int: ctypes converts it for the `void*` parameters and struct fields)


def _stream(dev):
    return _lib.stream_ptr(dev)


def _need_gpu(t, what):
    if not t.is_cuda:
        raise RuntimeError(f"{what} must be a CUDA(HIP) tensor: the surfel tracer has no CPU path")


class _Trace(torch.autograd.Function):
    """mrgs_surfel_trace_forward / _backward.  geom [P,16] = (mean, r_u / s_u, r_v / s_v, normal, opacity, -), attr [P,8] = (rgb, others, -)."""

    @staticmethod
    def forward(ctx, ray_o, ray_d, geom, attr, blob, bg3, ray_width):
        ctx.set_materialize_grads(False)
        L = _lib.lib()
        dev = ray_o.device
        n_rays, P = ray_o.shape[0], geom.shape[0]
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        rgb, norm, aux, dpt, acc, dist = new(n_rays, 3), new(n_rays, 3), new(n_rays, 2), new(n_rays), new(n_rays), new(n_rays)
        # the replay record (4 KB per wavefront and pass: ~250 MB for an 800x800 view) only when a backward can follow
        record = any(ctx.needs_input_grad)
        state_floats = (L.mrgs_surfel_trace_state_floats if record else L.mrgs_surfel_trace_state_floats_norecord)(n_rays, ray_width)
        wet, state = new(P), new(max(int(state_floats), 1))
        bg = (ctypes.c_float * 3)(*bg3)
        with _lib.guard(dev):
            _lib.check(L.mrgs_surfel_trace_forward(_ptr(blob), P, n_rays, ray_width, _ptr(ray_o), _ptr(ray_d), _ptr(geom), _ptr(attr), bg, _ptr(rgb),
                                                   _ptr(dpt), _ptr(acc), _ptr(norm), _ptr(dist), _ptr(aux), _ptr(wet), _ptr(state), state.numel(),
                                                   _stream(dev)))
        ctx.save_for_backward(ray_o, ray_d, geom, attr, blob, rgb, dpt, acc, norm, aux, state)
        ctx.bg3, ctx.ray_width = bg3, ray_width
        ctx.mark_non_differentiable(wet, state)
        return rgb, dpt, acc, norm, dist, aux, wet, state

    @staticmethod
    def backward(ctx, g_rgb, g_dpt, g_acc, g_norm, g_dist, g_aux, _g_wet, _g_state):
        ray_o, ray_d, geom, attr, blob, rgb, dpt, acc, norm, aux, state = ctx.saved_tensors
        L = _lib.lib()
        dev = ray_o.device
        n_rays, P = ray_o.shape[0], geom.shape[0]
        # an output nobody took a gradient of arrives as None and goes down as a null pointer (= zeros): no zero-filled maps
        c = lambda g: None if g is None else g.contiguous().float()
        g_rgb, g_dpt, g_acc, g_norm, g_aux, g_dist = c(g_rgb), c(g_dpt), c(g_acc), c(g_norm), c(g_aux), c(g_dist)
        g_geom, g_attr = torch.empty_like(geom), torch.empty_like(attr)
        g_o, g_d = torch.empty_like(ray_o), torch.empty_like(ray_d)
        bg = (ctypes.c_float * 3)(*ctx.bg3)
        with _lib.guard(dev):
            _lib.check(L.mrgs_surfel_trace_backward(_ptr(blob), P, n_rays, ctx.ray_width, _ptr(ray_o), _ptr(ray_d), _ptr(geom), _ptr(attr), bg, _ptr(rgb),
                                                    _ptr(dpt), _ptr(acc), _ptr(norm), _ptr(aux), _ptr(state), state.numel(), _ptr(g_rgb), _ptr(g_dpt),
                                                    _ptr(g_acc), _ptr(g_norm), _ptr(g_dist), _ptr(g_aux), _ptr(g_geom), _ptr(g_attr),
                                                    _ptr(g_o), _ptr(g_d), _stream(dev)))
        return g_o, g_d, g_geom, g_attr, None, None, None


class _Prep(torch.autograd.Function):
    """mrgs_surfel_trace_prep_forward / _backward: (means, scales, rotations, opacities, shs | colours, others) -> geom [P,16], attr [P,8]
    and get_disks' quad corners [4 P, 3] in one launch; exactly one of shs / colors is a tensor."""

    @staticmethod
    def forward(ctx, means, scales, rotations, opacities, shs, colors, others, campos, sh_degree, scale_modifier):
        L = _lib.lib()
        dev = means.device
        P = means.shape[0]
        f = lambda t: None if t is None else t.detach().contiguous().float()
        means, scales, rotations, opacities, shs, colors, others, campos = map(f, (means, scales, rotations, opacities, shs, colors, others, campos))
        geom = torch.empty(P, 16, dtype=torch.float32, device=dev)
        attr = torch.empty(P, 8, dtype=torch.float32, device=dev)
        quads = torch.empty(4 * P, 3, dtype=torch.float32, device=dev)
        M = 0 if shs is None else shs.shape[1]
        with _lib.guard(dev):
            _lib.check(L.mrgs_surfel_trace_prep_forward(P, _ptr(means), _ptr(scales), _ptr(rotations), _ptr(opacities), _ptr(shs), M, int(sh_degree),
                                                        _ptr(colors), _ptr(others), _ptr(campos), float(scale_modifier), _ptr(geom), _ptr(attr),
                                                        _ptr(quads), _stream(dev)))
        ctx.save_for_backward(means, scales, rotations, shs, campos)
        ctx.cfg = (M, int(sh_degree), float(scale_modifier), colors is not None, others is not None)
        ctx.mark_non_differentiable(quads)
        return geom, attr, quads

    @staticmethod
    def backward(ctx, g_geom, g_attr, _g_quads):
        means, scales, rotations, shs, campos = ctx.saved_tensors
        M, degree, modifier, has_colors, has_others = ctx.cfg
        L = _lib.lib()
        dev = means.device
        P = means.shape[0]
        z = lambda g, shape: torch.zeros(shape, dtype=torch.float32, device=dev) if g is None else g.contiguous().float()
        g_geom, g_attr = z(g_geom, (P, 16)), z(g_attr, (P, 8))
        new = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=dev)
        g_means, g_scales, g_rot, g_op = new(P, 3), new(P, 2), new(P, 4), new(P, 1)
        g_shs = new(P, M, 3) if shs is not None else None
        g_colors = new(P, 3) if has_colors else None
        g_others = new(P, 2) if has_others else None
        with _lib.guard(dev):
            _lib.check(L.mrgs_surfel_trace_prep_backward(P, _ptr(means), _ptr(scales), _ptr(rotations), _ptr(shs), M, degree, _ptr(campos), modifier,
                                                         _ptr(g_geom), _ptr(g_attr), _ptr(g_means), _ptr(g_scales), _ptr(g_rot), _ptr(g_op),
                                                         _ptr(g_shs), _ptr(g_colors), _ptr(g_others), _stream(dev)))
        return g_means, g_scales, g_rot, g_op, g_shs, g_colors, g_others, None, None, None


class _PrepRaw(torch.autograd.Function):
    """mrgs_surfel_trace_prep_raw_forward / _backward: the model's OWN tensors (raw scaling / rotation / opacity, the colour SH split into
    _features_dc and _features_rest) -> geom [P,16], attr [P,8], quad corners, GaussianModel's activations applied inside
    (scene/gaussian_model.py:56-78, 236-259) -- no getter kernels, no cat of the SH tensors, gradients in the model's layout."""

    @staticmethod
    def forward(ctx, xyz, scaling, rotation, opacity, f_dc, f_rest, others, grads3D, campos, sh_degree, scale_modifier):
        # grads3D: the reference's zero tensor added to the means (optix_utils.py:131-135) so that it receives dL/dmeans3D; here an input
        # whose value is never read and whose gradient is the means' gradient
        L = _lib.lib()
        dev = xyz.device
        P = xyz.shape[0]
        f = lambda t: None if t is None else t.detach().contiguous().float()
        xyz, scaling, rotation, opacity, f_dc, f_rest, others, campos = map(f, (xyz, scaling, rotation, opacity, f_dc, f_rest, others, campos))
        geom = torch.empty(P, 16, dtype=torch.float32, device=dev)
        attr = torch.empty(P, 8, dtype=torch.float32, device=dev)
        quads = torch.empty(4 * P, 3, dtype=torch.float32, device=dev)
        with _lib.guard(dev):
            _lib.check(L.mrgs_surfel_trace_prep_raw_forward(P, _ptr(xyz), _ptr(scaling), _ptr(rotation), _ptr(opacity), _ptr(f_dc), _ptr(f_rest),
                                                            int(sh_degree), _ptr(others), _ptr(campos), float(scale_modifier), _ptr(geom), _ptr(attr),
                                                            _ptr(quads), _stream(dev)))
        ctx.save_for_backward(xyz, scaling, rotation, opacity, f_dc, f_rest, campos)
        ctx.cfg = (int(sh_degree), float(scale_modifier), others is not None)
        ctx.mark_non_differentiable(quads)
        return geom, attr, quads

    @staticmethod
    def backward(ctx, g_geom, g_attr, _g_quads):
        xyz, scaling, rotation, opacity, f_dc, f_rest, campos = ctx.saved_tensors
        degree, modifier, has_others = ctx.cfg
        L = _lib.lib()
        dev = xyz.device
        P = xyz.shape[0]
        z = lambda g, shape: torch.zeros(shape, dtype=torch.float32, device=dev) if g is None else g.contiguous().float()
        g_geom, g_attr = z(g_geom, (P, 16)), z(g_attr, (P, 8))
        outs = [torch.empty_like(t) for t in (xyz, scaling, rotation, opacity, f_dc, f_rest)]
        g_others = torch.empty(P, 2, dtype=torch.float32, device=dev) if has_others else None
        with _lib.guard(dev):
            _lib.check(L.mrgs_surfel_trace_prep_raw_backward(P, _ptr(xyz), _ptr(scaling), _ptr(rotation), _ptr(opacity), _ptr(f_dc), _ptr(f_rest), degree,
                                                             _ptr(campos), modifier, _ptr(g_geom), _ptr(g_attr), *[_ptr(t) for t in outs], _ptr(g_others),
                                                             _stream(dev)))
        return (*outs, g_others if ctx.needs_input_grad[6] else None, outs[0] if ctx.needs_input_grad[7] else None, None, None, None)


def surfel_records(means3D, scales, rotations, opacities, colors, others, scale_modifier=1.0):
    """The per-surfel records of the C ABI stated with torch ops (the checker of mrgs_surfel_trace_prep_*; any device)."""
    R = build_rotation(rotations)
    s = scales * scale_modifier
    a = R[:, :, 0] / s[:, 0:1]
    b = R[:, :, 1] / s[:, 1:2]
    pad3 = torch.zeros_like(means3D)
    geom = torch.cat([means3D, a, b, R[:, :, 2], opacities.reshape(-1, 1), pad3], dim=1).contiguous()
    attr = torch.cat([colors, others, pad3], dim=1).contiguous()
    return geom, attr


class SurfelTracer(nn.Module):
    """Counterpart of diff_surfel_tracing.SurfelTracer (optix_utils.py:21, 76, 185)."""

    def __init__(self):
        super().__init__()
        self._blob = None
        self._blob_saved = False         # a trace that a backward may follow holds self._blob: the next build / trace must not write into it
        self._ws = None
        self._n = 0
        self._bg_host = {}               # settings.bg -> its three floats on the host (read back once per tensor, not once per trace)
        self.want_mid = True             # the per-depth record `mid` (41 MB for an 800x800 view): HardwareRendering reads it only for max_trace_depth > 0
        self.build_on_trace = False      # HardwareRendering sets it: build from the corners the record kernel writes (= get_disks')

    def build_acceleration_structure(self, vertices, faces=None, rebuild=True):
        """vertices [4 P, 3]: four corners per surfel in get_disks' order; faces are implied by that order (two triangles per quad,
        optix_utils.py:58-60) and only checked for their count."""
        _need_gpu(vertices, "vertices")
        if vertices.dim() != 2 or vertices.shape[1] != 3 or vertices.shape[0] % 4 != 0:
            raise RuntimeError("vertices must have dimensions (4 * num_surfels, 3)")
        P = vertices.shape[0] // 4
        if faces is not None and faces.shape[0] != 2 * P:
            raise RuntimeError("faces must hold two triangles per surfel")
        L = _lib.lib()
        dev = vertices.device
        v = vertices.detach().contiguous().float()
        with _lib.guard(dev):
            # A blob that an earlier trace saved for its backward is left to that backward (its lists and its replay record were made for
            # that hierarchy and for the records that trace wrote into it): this build gets a fresh one.
            if self._blob is None or self._n != P or self._blob.device != dev or self._blob_saved:
                self._blob = torch.empty(L.mrgs_surfel_bvh_bytes(P), dtype=torch.uint8, device=dev)
                self._blob_saved = False
            if self._ws is None or self._n != P or self._ws.device != dev:
                self._ws = torch.empty(L.mrgs_surfel_bvh_ws_bytes(P), dtype=torch.uint8, device=dev)
            self._n = P
            if P == 0:
                return self
            _lib.check(L.mrgs_surfel_bvh_build(_ptr(v), P, _ptr(self._blob), self._blob.numel(), _ptr(self._ws), self._ws.numel(), _stream(dev)))
        return self

    def _background(self, bg):
        """The three background floats on the host (the C ABI takes them by value).  Read back once per tensor (address + version): a
        `.tolist()` per trace is a device-to-host synchronisation in the middle of the stream-ordered pipeline."""
        if not bg.is_cuda:
            return tuple(float(x) for x in bg.detach().reshape(-1)[:3].tolist())
        key = _bg_key(bg)
        ent = self._bg_host.get(key)
        if ent is None or ent[1] is not bg:
            if len(self._bg_host) > 64:
                self._bg_host.clear()
            ent = self._bg_host[key] = (tuple(float(x) for x in bg.detach().reshape(-1)[:3].tolist()), bg)   # holds bg: the address stays its own
        return ent[0]

    def forward(self, ray_o, ray_d, v=None, means3D=None, grads3D=None, shs=None, colors_precomp=None, others_precomp=None, opacities=None,
                scales=None, rotations=None, cov3D_precomp=None, tracer_settings=None, start_from_first=True, records=None):
        """records (extension): (geom, attr, quads) already built by _PrepRaw from the model's raw tensors -- the getters' arguments
        (scales, shs, ...) are not read then."""
        ts = tracer_settings
        if self._blob is None and not self.build_on_trace and means3D is not None and means3D.shape[0] > 0:
            raise RuntimeError("build_acceleration_structure has not been called")
        if records is None and (cov3D_precomp is not None or scales is None or rotations is None):
            raise NotImplementedError("the tracer intersects surfels from scales / rotations; cov3D_precomp is not supported")
        if ts.max_trace_depth != 0:
            raise NotImplementedError("max_trace_depth > 0 (bounces inside the tracer) is not built; every caller of the reference uses 0")
        if records is not None:
            _need_gpu(means3D, "means3D")
            return self._trace(ray_o, ray_d, means3D.shape[0], records, ts)
        if (shs is None) == (colors_precomp is None):
            raise RuntimeError("Please provide exactly one of either SHs or precomputed colors!")
        _need_gpu(means3D, "means3D")
        P = means3D.shape[0]
        means = means3D if grads3D is None else means3D + grads3D          # the densification proxy receives d/d means3D
        # computeColorFromSH of the rasterizer family (forward.cu:20-81, direction from the settings' camera position), the splat frame
        # and get_disks' corners: one launch (mrgs_surfel_trace_prep_forward)
        shs_pm3 = None if shs is None else (shs if shs.shape[-1] == 3 else shs.transpose(1, 2))
        if P == 0:       # an empty model (e.g. everything pruned): background everywhere, nothing to build or to differentiate
            recs = (means3D.new_zeros((0, 16), dtype=torch.float32), means3D.new_zeros((0, 8), dtype=torch.float32), None)
        else:
            recs = _Prep.apply(means, scales, rotations, opacities.reshape(P, 1), shs_pm3, colors_precomp, others_precomp,
                               ts.campos.reshape(3), ts.sh_degree, float(ts.scale_modifier))
        return self._trace(ray_o, ray_d, P, recs, ts)

    def _trace(self, ray_o, ray_d, P, records, ts):
        geom, attr, _quads = records
        shape = ray_o.shape[:-1]
        if P == 0:
            self._n, self.build_on_trace = 0, False
        if self.build_on_trace:
            self.build_acceleration_structure(_quads, None)
            self.build_on_trace = False
        if P != self._n:
            raise RuntimeError("the acceleration structure was built for a different number of surfels")
        o = ray_o.reshape(-1, 3).contiguous().float()
        d = ray_d.reshape(-1, 3).contiguous().float()
        bg3 = self._background(ts.bg)
        will_save = torch.is_grad_enabled() and any(t.requires_grad for t in (o, d, geom, attr))
        if self._blob_saved and P > 0:
            # a second trace on one build (eval mode keeps the hierarchy): the trace writes the surfel records into the blob, which an
            # earlier trace's backward still reads -- work on a copy of the hierarchy
            self._blob = self._blob.clone()
            self._blob_saved = False
        rgb, dpt, acc, norm, dist, aux, wet, state = _Trace.apply(o, d, geom, attr, self._blob, bg3, int(ray_o.shape[-2]) if ray_o.dim() == 3 else 0)
        self._blob_saved = will_save
        self.last_state = state[:4 * o.shape[0]].reshape(-1, 4)      # diagnostics: sum w t^2, final T, hits blended, passes (negative: in a packet)
        self.last_state_all, self.last_ray_width = state, (int(ray_o.shape[-2]) if ray_o.dim() == 3 else 0)   # diagnostics (record_summary)
        r = lambda x, c: x.reshape(*shape, c)
        rgb, dpt, acc, norm, dist, aux = r(rgb, 3), r(dpt, 1), r(acc, 1), r(norm, 3), r(dist, 1), r(aux, 2)
        # stage 0 of the per-depth record (optix_utils.py:28-35); deeper stages do not exist at max_trace_depth = 0
        mid = torch.cat([ray_o.reshape(*shape, 3).float(), ray_d.reshape(*shape, 3).float(), dpt, acc, norm, aux, rgb], dim=-1).detach() \
            if self.want_mid else rgb.new_empty((*shape, 0))
        return rgb, dpt, acc, norm, dist, aux, mid, wet.reshape(P, 1)


def record_summary(tracer):
    """Diagnostics of the last trace of `tracer` (a SurfelTracer): rays traced one per wavefront, packets listed for the second launch,
    chunks of the replay record taken from the shared pool, and whether the record is usable (False: the backward walks again)."""
    st = tracer.last_state_all
    n_rays = tracer.last_state.shape[0]
    off = (ctypes.c_size_t * 5)()
    _lib.check(_lib.lib().mrgs_surfel_trace_state_layout(n_rays, tracer.last_ray_width, off))
    words = st.view(torch.int32)
    have_hdr = st.numel() >= off[3]
    pick = lambda i: int(words[i].item())
    pick8 = lambda i: int(words[i:i + 512:64].sum().item())       # both lists are kept in eight parts (one per XCD region of the launch): eight counts, 64 words apart
    return {"rays": n_rays, "lone_rays": pick8(off[0]), "listed_packets": pick8(off[1]), "pool_chunks": pick(off[2]) if have_hdr else None,
            "record_usable": have_hdr and st.numel() >= off[4] and pick(off[2] + 1) == 0}


_ZERO_LEAVES, _OTHERS = {}, {}


def _zero_leaf(like):
    """A leaf of zeros shaped like `like` whose .grad receives a gradient: a fresh tensor object over one shared, never written block of
    zeros (no fill kernel per view)."""
    key = (like.device, tuple(like.shape), like.dtype)
    z = _ZERO_LEAVES.get(key)
    if z is None:
        if len(_ZERO_LEAVES) > 8:
            _ZERO_LEAVES.clear()
        z = _ZERO_LEAVES[key] = torch.zeros_like(like, requires_grad=False)
    return z.detach().requires_grad_(True)


def _const_others(P, device):
    key = (device, P)
    t = _OTHERS.get(key)
    if t is None:
        if len(_OTHERS) > 8:
            _OTHERS.clear()
        t = _OTHERS[key] = torch.full((P, 2), 0.01, device=device)
    return t


def _raw_model(pcd, pipe, override_color):
    """(xyz, scaling, rotation, opacity, features_dc, features_rest) when `pcd` stores GaussianModel's raw tensors under their usual names
    with the usual activations and the colour comes from its SH coefficients; None -> the getters are used."""
    if override_color is not None or getattr(pipe, "convert_SHs_python", False):
        return None
    names = ("_xyz", "_scaling", "_rotation", "_opacity", "_features_dc", "_features_rest")
    if not all(hasattr(pcd, n) for n in names):
        return None
    ts = tuple(getattr(pcd, n) for n in names)
    P = ts[0].shape[0]
    if not all(torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in ts):
        return None
    if tuple(ts[1].shape) != (P, 2) or tuple(ts[2].shape) != (P, 4) or ts[3].numel() != P or tuple(ts[4].shape) != (P, 1, 3) or tuple(ts[5].shape) != (P, 15, 3):
        return None
    # a model with other activations than exp / sigmoid / normalize declares them (GaussianModel.setup_functions, scene/gaussian_model.py:56-78)
    for attr, fn in (("scaling_activation", torch.exp), ("opacity_activation", torch.sigmoid), ("rotation_activation", torch.nn.functional.normalize)):
        if getattr(pcd, attr, fn) is not fn:
            return None
    return ts


def _bg_key(bg):
    return (bg.data_ptr(), bg._version, bg.device)


_ZERO_MAPS = {}


def _zero_map(like):
    """A shared all-zero map of `like`'s shape (read-only by convention: it stands for a per-view torch.zeros_like whose value never changes)."""
    key = (str(like.device), tuple(like.shape), like.dtype)
    z = _ZERO_MAPS.get(key)
    if z is None:
        if len(_ZERO_MAPS) >= 8:
            _ZERO_MAPS.pop(next(iter(_ZERO_MAPS)))
        z = _ZERO_MAPS[key] = torch.zeros(like.shape, dtype=like.dtype, device=like.device)
    return z


def _depth_to_normal(view, depth):
    """depth_to_normal(view, depth) of utils/point_utils.py:26-37 for a depth map [H,W] through the maps kernel of render_surfel
    (`mrgs_surfel_maps_*`): an all-map whose expected depth is `depth` and whose alpha is 1 makes its surf_normal exactly that."""
    from types import SimpleNamespace
    from .renderer import compute_2dgs_normal_and_regularizations
    H, W = depth.shape
    allmap = torch.zeros(7, H, W, dtype=torch.float32, device=depth.device)
    allmap[0], allmap[1] = depth, 1.0
    return compute_2dgs_normal_and_regularizations(allmap, view, SimpleNamespace(depth_ratio=0.0))["surf_normal"].permute(1, 2, 0)


class HardwareRendering(nn.Module):
    """Counterpart of gaussian_renderer/optix_utils.py:14-271 on the HIP tracer."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.tracer = SurfelTracer()
        self.tracer.want_mid = False       # only read for max_trace_depth > 0 (:244-268), which is not built
        self.has_bvh = False
        self.mid_channel = MID_CHANNELS
        self.rayo_off, self.rayd_off, self.dpt_off, self.acc_off, self.norm_off, self.aux_off, self.rgb_off = 0, 3, 6, 7, 8, 11, 13

    def get_disks(self, pcd):
        """Corners (-3,3), (-3,-3), (3,3), (3,-3) of every surfel in its tangent frame and the two triangles (0,1,2), (1,2,3) (:36-66)."""
        T = pcd.get_covariance()                       # [P,4,4], rows: s_u r_u, s_v r_v, r_w, mean (gaussian_model.py:48-54)
        su, sv, m = T[:, 0, :3], T[:, 1, :3], T[:, 3, :3]
        corners = torch.tensor([[-3.0, 3.0], [-3.0, -3.0], [3.0, 3.0], [3.0, -3.0]], device=T.device)
        v = (m[:, None, :] + corners[None, :, 0:1] * su[:, None, :] + corners[None, :, 1:2] * sv[:, None, :]).reshape(-1, 3)
        idx = torch.arange(v.shape[0], device=T.device, dtype=torch.int32).reshape(-1, 4)
        f = torch.stack([idx[:, :3], idx[:, 1:]], dim=1).reshape(-1, 3)
        return v.contiguous(), f.contiguous()

    def build_bvh(self, pcd, rebuild=1):
        if rebuild or self.training or not self.has_bvh:
            v, f = self.get_disks(pcd)
            self.tracer.build_acceleration_structure(v.detach(), f, rebuild=True)
            self.has_bvh = not self.training
            return v, f
        return None, None

    def render_gaussians(self, camera, ray_o, ray_d, pcd, pipe, bg_color, max_trace_depth=0, specular_threshold=0.0, start_from_first=True,
                         scaling_modifier=1, override_color=None):
        settings = SurfelTracingSettings(
            image_height=int(camera.image_height), image_width=int(camera.image_width), tanfovx=math.tan(camera.FoVx * 0.5),
            tanfovy=math.tan(camera.FoVy * 0.5), bg=bg_color, scale_modifier=scaling_modifier,
            viewmatrix=camera.world_view_transform.contiguous(), projmatrix=camera.full_proj_transform.contiguous(),
            sh_degree=pcd.active_sh_degree, campos=camera.camera_center.contiguous(), prefiltered=False, debug=False,
            max_trace_depth=max_trace_depth, specular_threshold=specular_threshold)
        # build_bvh (:68-82) without the torch detour through get_disks: the tracer builds from the corners its record kernel writes
        if self.training or not self.has_bvh:
            self.tracer.build_on_trace = True
            self.has_bvh = not self.training
        v = None
        if getattr(pipe, "compute_cov3D_python", False):
            raise NotImplementedError("compute_cov3D_python: the tracer takes scales / rotations")
        raw = _raw_model(pcd, pipe, override_color)
        if raw is not None:
            # The model's own tensors go into the record kernel as they are (activations, SH colour and get_disks' corners inside; gradients
            # come back in the model's layout): the reference's getters (:128-170) are ~12 torch kernels forward and ~25 backward per view.
            means3D = raw[0]
            P = means3D.shape[0]
            grads3D = _zero_leaf(means3D)                                                 # receives dL/dmeans3D, as the reference's (:131-135)
            others = _const_others(P, means3D.device)                                     # the reference's placeholder (:173-177)
            recs = (means3D.new_zeros((0, 16)), means3D.new_zeros((0, 8)), None) if P == 0 else \
                _PrepRaw.apply(*raw, others, grads3D, settings.campos.reshape(3), settings.sh_degree, float(settings.scale_modifier))
            rgb, dpt, acc, norm, dist, aux, mid, wet = self.tracer(ray_o.contiguous(), ray_d.contiguous(), v, means3D=means3D, tracer_settings=settings,
                                                                   start_from_first=start_from_first, records=recs)
        else:
            means3D, opacities = pcd.get_xyz.contiguous(), pcd.get_opacity.contiguous()
            grads3D = torch.zeros_like(means3D, requires_grad=True) + 0
            try:
                grads3D.retain_grad()
            except RuntimeError:
                pass
            scales, rotations = pcd.get_scaling.contiguous(), pcd.get_rotation.contiguous()
            shs = colors_precomp = None
            if override_color is None or (getattr(pcd, "render_reflection", False) and getattr(pcd, "feature_splatting", False)):
                if getattr(pipe, "convert_SHs_python", False):
                    shs_view = pcd.get_features.transpose(1, 2).view(-1, 3, (pcd.max_sh_degree + 1) ** 2)
                    dirs = pcd.get_xyz - camera.camera_center.reshape(1, 3)
                    colors_precomp = torch.clamp_min(eval_sh(pcd.active_sh_degree, shs_view, dirs / dirs.norm(dim=1, keepdim=True)) + 0.5, 0.0)
                else:
                    shs = pcd.get_features.contiguous()
            else:
                colors_precomp = override_color.contiguous()
            others = torch.full((means3D.shape[0], 2), 0.01, device=means3D.device)          # the reference's placeholder (:173-177)
            rgb, dpt, acc, norm, dist, aux, mid, wet = self.tracer(
                ray_o.contiguous(), ray_d.contiguous(), v, means3D=means3D, grads3D=grads3D, shs=shs, colors_precomp=colors_precomp,
                others_precomp=others, opacities=opacities, scales=scales, rotations=rotations, cov3D_precomp=None, tracer_settings=settings,
                start_from_first=start_from_first)
        with torch.no_grad():
            visibility_filter = wet[..., 0] > 0.0
            if start_from_first:                                                             # + what "projects into the image" (:203-211)
                # The reference's expression, kept literally because `visibility_filter` is defined by it: Camera.R is the rotation as
                # STORED (transposed, scene/cameras.py:30) and T [3] broadcasts against R m [P,3,1] along the LAST axis, so column 0 of
                # the sum -- the one `[..., 0]` keeps -- is R m + T[0] (1, 1, 1).  Not the camera projection; a fixture produced by the
                # reference's own code pins it (tests/test_reference_render.py).
                _, _, K = camera.HWK
                K = torch.as_tensor(K, dtype=torch.float32, device=means3D.device)
                R = torch.as_tensor(camera.R, dtype=torch.float32, device=means3D.device)
                T = torch.as_tensor(camera.T, dtype=torch.float32, device=means3D.device)
                uvd = (K @ (R @ means3D.detach()[..., None] + T))[..., 0]
                uvd[..., :2] = uvd[..., :2] / uvd[..., 2:]
                vis = (uvd[..., 2] >= 0.2) & (uvd[..., 0] >= 0.0) & (uvd[..., 0] <= camera.image_width) & (uvd[..., 1] >= 0.0) & \
                      (uvd[..., 1] <= camera.image_height)
                visibility_filter = visibility_filter | vis
        chw = lambda x: x.permute(2, 0, 1)
        out = {"viewspace_points": grads3D, "visibility_filter": visibility_filter, "weight_accumulate": wet,      # (a fresh tensor already)
               "render": chw(rgb), "rend_alpha": chw(acc), "rend_normal": chw(norm), "rend_dist": chw(dist), "surf_depth": chw(dpt)}
        if start_from_first:
            out["surf_normal"] = chw(_depth_to_normal(camera, dpt[..., 0]) * acc.detach())
        else:
            out["surf_normal"] = _zero_map(chw(norm))        # (the reference's zeros_like(:231); a constant, not filled per view)
        out["specular"] = chw(aux[..., :1])
        out["roughness"] = chw(aux[..., 1:2])
        return out

    def forward(self, *args, **kwargs):
        return self.render_gaussians(*args, **kwargs)
