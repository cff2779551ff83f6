"""Drop-in replacement of the reference's `diff_surfel_rasterization` Python module on MI355X.

Same public surface as /root/reference/submodules/diff-surfel-rasterization/diff_surfel_rasterization/__init__.py:
`GaussianRasterizationSettings` (:167-179), `GaussianRasterizer` (:181-235, incl. `markVisible` :186-195),
`rasterize_gaussians` (:20-45) and the autograd function `_RasterizeGaussians` (:47-165) -- same argument
names and order, same return tuple `(contrib i32[1,H,W], color f32[3,H,W], feature f32[S,H,W], radii i32[P],
allmap f32[7,H,W])`, same gradient routing.  The native side is libmrgs.so (include/mrgs.h) instead of the
pybind `_C` module; there is no CPU or PyTorch fallback.
"""
import ctypes
import functools
import threading
import time
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _lib
from ._lib import MrgsRasterConfig, MrgsRasterGrads, MrgsRasterInputs, MrgsRasterTicket


_PAIR_GUESS = {}   # (device index, P, H, W) -> pair capacity to try first (previous count of that configuration + 25 %)
_PAIR_GUESS_MAX = 64
# Per-camera work hints (MrgsRasterInputs::work_hint): the forward orders its blend waves by what each 8x8 block cost the last time the
# same camera was rendered.  Keyed by the camera's matrices (the tensors a training loop keeps per camera), bounded, and purely a
# scheduling aid: results do not depend on it.  An entry holds its two matrices, so their storage cannot be handed to another camera's
# tensors while the entry exists (the address is the key), and a hit must be the same storage at the same version.
_WORK_HINTS = {}
_NO_HINT = bool(int(__import__("os").environ.get("MRGS_NO_WORK_HINT", "0")))   # developer switch for A/B timing
_NO_PREPARE = bool(int(__import__("os").environ.get("MRGS_NO_PREPARE_BWD", "0")))   # developer switch: the backward orders / clears by itself
_NO_DEFER = bool(int(__import__("os").environ.get("MRGS_NO_DEFER", "0")))   # developer switch: render functions wait for the pair count inside the rasterizer call
_WORK_HINTS_MAX = 2048                    # cameras ...
_WORK_HINTS_MAX_BYTES = 256 << 20         # ... and device bytes the cache may pin (a buffer is ~100 KB at 800x800, ~340 KB at 1600x1600), least recently used first out
_WORK_HINTS_BYTES = [0]


def _hint_key(raster_settings, device):
    """A camera's identity for the hint cache: the device, the image size and the addresses of its two matrices.  NOT the surfel count:
    what a hint holds is image-space -- the work each 8x8 block cost, the queues its (tile, quadrant) items were dealt into -- and a
    densification or pruning step (every 100 iterations until iteration 25 000-30 000 of the reference's 50 000,
    arguments/__init__.py:159-162) changes that picture by a few per cent, not into another one: the next visit orders its waves by the
    work the camera measured before the step (round 5 keyed by P as well, and every visit of the densifying phase was a first visit)."""
    vm, pm = raster_settings.viewmatrix, raster_settings.projmatrix
    return (device.index, int(raster_settings.image_height), int(raster_settings.image_width), vm.data_ptr(), pm.data_ptr())


class _Hint:
    """One camera's entry: the hint buffer, the two matrices (kept alive: their addresses are the key) with their versions, the number
    of renders whose measured work the buffer has received, and which surfel set the queues in the buffer were last DEALT for -- the
    condition of MRGS_HINT_REUSE_ORDER (a reused deal balances the scene it was made for; after a change of the set the next visit
    orders anew, from the measured work that is there)."""
    __slots__ = ("buf", "vm", "pm", "versions", "visits", "dealt_for")

    def __init__(self, buf, vm, pm):
        self.buf, self.vm, self.pm, self.versions, self.visits, self.dealt_for = buf, vm, pm, (vm._version, pm._version), 0, None


_GENERATION = [0]       # bumped by note_surfel_set_changed(): a surfel set of the same size as before is still another set


def _hint_entry(raster_settings, device, P=None):
    """The entry of this camera, or None when it was never rendered or when its matrices were written in place since."""
    key = _hint_key(raster_settings, device)
    ent = _WORK_HINTS.get(key)
    if ent is None:
        return None
    vm, pm = raster_settings.viewmatrix, raster_settings.projmatrix
    if ent.versions != (vm._version, pm._version):
        return None
    _WORK_HINTS[key] = _WORK_HINTS.pop(key)      # most recently used last (dicts keep insertion order)
    return ent


def _hint_is_warm(raster_settings, device, P=None):
    """True when this camera was rendered before, i.e. its hint holds measured work (of whichever surfel set).  Only then may the forward
    set up the backward's queues (they are a copy of its own): built from the cull counts alone they balance the backward a third worse
    than the backward's own ordering by what the forward waves walked."""
    ent = _hint_entry(raster_settings, device)
    return ent is not None and ent.visits > 0


_REORDER_EVERY = 16     # visits of a camera between two orderings of its blend waves (MRGS_HINT_REUSE_ORDER in between)
_NO_REUSE = bool(int(__import__("os").environ.get("MRGS_NO_REUSE_ORDER", "0")))   # developer switch for A/B timing


def _hint_flags(raster_settings, device, P, forward=False):
    """MRGS_HINT_REUSE_ORDER for this forward: the camera's hint buffer holds the queues an ordering launch dealt from measured work for
    THIS surfel set, and the forward deals its waves the same way again instead of ordering them anew -- the ordering is 15 us of a
    0.5 ms view and changes little from one visit to the next.  Every _REORDER_EVERY-th visit orders again, and so does the first visit
    after the surfel set changed (another P, or note_surfel_set_changed()).  forward=True: the call that decides for a forward -- it
    records that the ordering launch of this forward deals the queues for (P, generation); the backward's struct carries no such flag."""
    if _NO_HINT or _NO_REUSE or not forward:
        return 0
    ent = _hint_entry(raster_settings, device)
    if ent is None:
        return 0
    now = (int(P), _GENERATION[0])
    if ent.visits >= 2 and ent.visits % _REORDER_EVERY != 0 and ent.dealt_for == now:
        return _lib.MRGS_HINT_REUSE_ORDER
    if ent.visits >= 1:
        ent.dealt_for = now           # this forward orders from measured work: the deal in the buffer is this set's from here on
    return 0


def _work_hint(raster_settings, device, P=None, count_visit=False):
    if _NO_HINT:
        return None
    ent = _hint_entry(raster_settings, device)
    if ent is None:
        n = _lib.lib().mrgs_work_hint_bytes(int(raster_settings.image_height), int(raster_settings.image_width)) // 4
        key = _hint_key(raster_settings, device)
        old = _WORK_HINTS.pop(key, None)              # (stale: its matrices were written in place)
        if old is not None:
            _WORK_HINTS_BYTES[0] -= old.buf.numel() * 4
        while _WORK_HINTS and (len(_WORK_HINTS) >= _WORK_HINTS_MAX or _WORK_HINTS_BYTES[0] + 4 * n > _WORK_HINTS_MAX_BYTES):
            _WORK_HINTS_BYTES[0] -= _WORK_HINTS.pop(next(iter(_WORK_HINTS))).buf.numel() * 4      # (a render in flight keeps its buffer alive through its ctx)
        ent = _Hint(torch.zeros(max(int(n), 1), dtype=torch.int32, device=device), raster_settings.viewmatrix, raster_settings.projmatrix)
        _WORK_HINTS[key] = ent
        _WORK_HINTS_BYTES[0] += ent.buf.numel() * 4
    if count_visit:
        ent.visits += 1
    return ent.buf


def reset_work_hints():
    """Forget every camera's measured work (the next render of each camera is a first visit again); bench.py uses it to time first
    visits.  A training loop does NOT need to call this after densification / pruning: see note_surfel_set_changed()."""
    _WORK_HINTS.clear()
    _WORK_HINTS_BYTES[0] = 0
    _CAM_COPIES.clear()


def note_surfel_set_changed():
    """The surfel set was replaced by another one of the SAME size (a prune and a clone that cancel, a re-initialisation in place): the
    measured work of every camera stays -- it is image-space and still the best estimate there is -- but the next visit of each camera
    orders its blend waves anew instead of reusing the deal made for the old set.  A change of the surfel COUNT is seen by the
    rasterizer itself and needs no call."""
    _GENERATION[0] += 1


_ZERO_CONTRIB = {}


def _zero_contrib(dev, H, W):
    """`out_contrib` of the reference is allocated zero-filled and never written by any kernel (rasterize_points.cu:89); one read-only
    zero tensor per (device, size) stands in for it instead of a 4 H W byte fill per render."""
    key = (dev.index, H, W)
    t = _ZERO_CONTRIB.get(key)
    if t is None:
        if len(_ZERO_CONTRIB) > 16:
            _ZERO_CONTRIB.clear()
        t = _ZERO_CONTRIB[key] = torch.zeros((1, H, W), dtype=torch.int32, device=dev)
    return t


_AFTER_BLEND_HOOK = [None]


def set_after_blend_hook(fn):
    """fn(dL_dRGB_masked [P,3]) is called in the middle of every rasterizer backward, after the blend backward has been queued and
    before the per-gaussian backward: the clamp-masked colour gradient of every surfel (= dL/dsh[:, 0, :] / SH_C0 for SH colours,
    dL/dcolors_precomp otherwise) is final at that point of the stream.  A view-parallel training step registers
    dist.FactoredGradReducer.begin_early here so that its all-gather runs under the per-gaussian backward.  None removes the hook."""
    _AFTER_BLEND_HOOK[0] = fn


LAST_NUM_RENDERED = 0   # diagnostics: num_rendered of the most recent forward (bench.py reads it for the roofline figure)
COUNT_WAIT_SECONDS = 0.0   # diagnostics: host time spent waiting (on the GPU) for pair counts; bench.py takes it out of its host-issue figure


def cpu_deep_copy_tuple(input_tuple):
    """Host copies of the tensors of an argument tuple (for the snapshot_*.dump files of debug mode); other items pass through."""
    return tuple(x.detach().cpu().clone() if torch.is_tensor(x) else x for x in input_tuple)


def _ptr(t):
    return None if t is None else (t.data_ptr() or None)      # (an int, NULL for an empty tensor: ctypes converts it for the `void*` parameters and struct fields)


def _f32c(t):
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


_CAM_COPIES = {}     # (data_ptr, strides, dtype) of a camera matrix as the caller holds it -> (that tensor, its version, contiguous fp32 copy)
_CAM_COPIES_MAX = 4096


def _camera_f32c(t):
    """Contiguous fp32 form of a camera tensor, THE SAME tensor for every render of the same camera.  The reference's Camera builds
    `world_view_transform = torch.tensor(...).transpose(0, 1).cuda()` (scene/cameras.py:77): a transposed view, not contiguous, so a
    plain .contiguous() would hand the rasterizer a fresh copy (a fresh address) on every render -- and the per-camera work hints,
    which are keyed by the matrices' addresses, would never see the same camera twice.  The copy is cached per source tensor (address,
    strides, dtype; the entry keeps the source alive, so the address cannot be reused) and redone when the source was written in place."""
    if t.dtype == torch.float32 and t.is_contiguous():
        return t
    key = (t.data_ptr(), tuple(t.stride()), t.dtype, t.device.index)
    ent = _CAM_COPIES.get(key)
    if ent is not None and ent[1] == t._version and ent[0].shape == t.shape:
        return ent[2]
    if ent is None and len(_CAM_COPIES) >= _CAM_COPIES_MAX:
        _CAM_COPIES.pop(next(iter(_CAM_COPIES)))
    c = _f32c(t)
    _CAM_COPIES[key] = (t, t._version, c)
    return c


def _stream(device):
    return _lib.stream_ptr(device)


_RESOLVE = object()
# MrgsRasterInputs::features_live of the render being issued on this thread (GaussianRasterizer.features_live: the settings tuple keeps
# the reference's fields); the autograd node notes it for its backward
_LIVE = __import__("threading").local()


def _make_cfg_inputs(raster_settings, means3D, sh, colors_precomp, features, opacities, scales, rotations, cov3Ds_precomp, sh_rest=None,
                     bwd_grad_ws=None, work_hint=_RESOLVE, extra_flags=0):
    """work_hint: the camera's hint buffer (or None) -- the BACKWARD passes the very tensor its forward used (the queue state, the tickets
    and the forward's item assignment live in it: a buffer resolved anew at backward time could be a fresh, zeroed one after
    reset_work_hints(), an eviction or an in-place pose change, and a prepared backward would then pull item 0 in every wave); the
    forward leaves it out and gets the cache's.  Returns (cfg, inp, hint tensor)."""
    P = means3D.shape[0]
    is_forward = work_hint is _RESOLVE
    if work_hint is _RESOLVE:
        work_hint = _work_hint(raster_settings, means3D.device) if means3D.is_cuda else None
    S = features.shape[1] if features.dim() == 2 else 0
    M = sh.shape[1] if sh.numel() != 0 else 0
    if sh_rest is not None:       # split layout: sh = DC [P,1,3], sh_rest = [P,M-1,3]
        M = 1 + sh_rest.shape[1]
    cfg = MrgsRasterConfig(P, S, int(raster_settings.sh_degree), M, int(raster_settings.image_height),
                           int(raster_settings.image_width), float(raster_settings.tanfovx), float(raster_settings.tanfovy),
                           float(raster_settings.scale_modifier), int(bool(raster_settings.prefiltered)),
                           int(bool(raster_settings.debug)))
    inp = MrgsRasterInputs(_ptr(raster_settings.bg), _ptr(means3D), _ptr(sh), _ptr(colors_precomp), _ptr(features),
                           _ptr(opacities), _ptr(scales), _ptr(rotations), _ptr(cov3Ds_precomp),
                           _ptr(raster_settings.viewmatrix), _ptr(raster_settings.projmatrix), _ptr(raster_settings.campos),
                           _ptr(work_hint), _ptr(sh_rest),
                           _ptr(bwd_grad_ws), (_hint_flags(raster_settings, means3D.device, P, forward=is_forward) if means3D.is_cuda else 0) | extra_flags,
                           int(getattr(_LIVE, "n", 0) or 0))
    return cfg, inp, work_hint


class RasterWorkspaceOverflow(RuntimeError):
    """The pair count of a begun render did not fit the workspace carved from the guess: its outputs are undefined."""

    def __init__(self, num_rendered):
        super().__init__(f"libmrgs: {num_rendered} (tile, surfel) pairs did not fit the binning workspace sized from the previous view")
        self.num_rendered = num_rendered


_DEFER = threading.local()
_TICKET_RING = 16          # MRGS_TICKET_RING of csrc/mrgs_api.hip (include/mrgs.h: mrgs_rasterize_forward_begin)


def _note_count(guess_key, num_rendered, hint_settings, dev):
    """Bookkeeping once a view's pair count is known: the next view's workspace guess, this camera's work hint."""
    if guess_key not in _PAIR_GUESS and len(_PAIR_GUESS) >= _PAIR_GUESS_MAX:
        _PAIR_GUESS.pop(next(iter(_PAIR_GUESS)))
    _PAIR_GUESS[guess_key] = max(int(num_rendered * 1.25) + 65536, 1)
    if hint_settings is not None:
        _work_hint(hint_settings, dev, count_visit=True)   # this camera's hint now holds measured work


class _PendingCount:
    """The pair count of a render begun with mrgs_rasterize_forward_begin.  finish() waits for it (once), does the bookkeeping and
    raises RasterWorkspaceOverflow when the count did not fit; `ctx` (the autograd node of the render) gets its num_rendered then."""

    def __init__(self, ticket, guess_key, hint_settings, dev):
        self.ticket, self.guess_key, self.hint_settings, self.dev = ticket, guess_key, hint_settings, dev
        self.value, self.overflow, self.ctx = None, False, None

    def finish(self):
        global LAST_NUM_RENDERED, COUNT_WAIT_SECONDS
        if self.value is None:
            R = ctypes.c_int64(0)
            t0 = time.perf_counter()
            rc = _lib.lib().mrgs_rasterize_forward_finish(ctypes.byref(self.ticket), ctypes.byref(R))
            COUNT_WAIT_SECONDS += time.perf_counter() - t0
            if rc != _lib.MRGS_E_WORKSPACE:
                _lib.check(rc)
            self.value, self.overflow = int(R.value), rc == _lib.MRGS_E_WORKSPACE
            _note_count(self.guess_key, self.value, None if self.overflow else self.hint_settings, self.dev)
            LAST_NUM_RENDERED = self.value
            if self.ctx is not None:
                self.ctx.num_rendered = self.value
        if self.overflow:
            raise RasterWorkspaceOverflow(self.value)
        return self.value


class deferred_count:
    """with deferred_count() as box: ... -- rasterizer calls inside only BEGIN their render (nothing on the host waits for the tile scan);
    the caller queues whatever follows and calls box.finish() afterwards, which waits for the counts and raises RasterWorkspaceOverflow
    if one did not fit (everything computed from that render is then undefined: redo the view outside the context).  Nested contexts
    share the outermost box.  The first view of a (device, P, H, W) has no guess yet and runs the two-phase path at once."""

    def __init__(self):
        self.pending, self.outer = [], None

    def __enter__(self):
        self.outer = getattr(_DEFER, "box", None)
        if self.outer is None:
            _DEFER.box = self
            return self
        return self.outer

    def __exit__(self, *exc):
        if self.outer is None:
            _DEFER.box = None
        return False

    def finish(self):
        if self.outer is not None:            # an inner context: the outermost one finishes
            return
        pending, self.pending = self.pending, []
        first = None
        for p in pending:                     # every count is collected (bookkeeping) before the first overflow is reported
            try:
                p.finish()
            except RasterWorkspaceOverflow as ex:
                first = first or ex
        if first is not None:
            raise first


def deferred_raster_count(render_fn):
    """Decorator for the render functions: rasterizer counts are collected after the whole view has been queued; a view whose pair
    count outgrew the guess is rendered again (synchronously sized)."""
    if _NO_DEFER:
        return render_fn

    @functools.wraps(render_fn)
    def wrapper(*args, **kw):
        dc = deferred_count()
        with dc:
            out = render_fn(*args, **kw)
        if dc.outer is not None:              # called from inside another render function: that one collects the counts
            return out
        try:
            dc.finish()
        except RasterWorkspaceOverflow:
            return render_fn(*args, **kw)     # the guess is refreshed: the one-call path fits now (or the exact two-phase path runs)
        return out
    return wrapper


def _rasterize_forward_native(raster_settings, means3D, sh, colors_precomp, features, opacities, scales, rotations, cov3Ds_precomp,
                              sh_rest=None, prepare_backward=False):
    """Counterpart of `_C.rasterize_gaussians` (rasterize_points.cu:41-144).  prepare_backward: also allocate the backward's gradient-row
    workspace and let the forward clear it and set up the backward's work queues (MrgsRasterInputs::bwd_grad_ws); it is appended to
    the returned tuple."""
    if means3D.dim() != 2 or means3D.shape[1] != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    if not means3D.is_cuda:
        raise RuntimeError("means3D must be a CUDA tensor")   # CHECK_INPUT, rasterize_points.cu:29-31
    L = _lib.lib()
    dev = means3D.device
    H, W = int(raster_settings.image_height), int(raster_settings.image_width)
    S_ = features.shape[1] if features.dim() == 2 else 0
    grad_ws = None
    if prepare_backward and means3D.shape[0] > 0:
        with _lib.guard(dev):
            grad_ws = torch.empty((L.mrgs_grad_bytes(means3D.shape[0], S_),), dtype=torch.uint8, device=dev)
    cfg, inp, hint = _make_cfg_inputs(raster_settings, means3D, sh, colors_precomp, features, opacities, scales, rotations, cov3Ds_precomp,
                                      sh_rest, grad_ws, extra_flags=_lib.MRGS_HINT_VISIBLE_BYTES)
    P, S = cfg.P, cfg.S
    with _lib.guard(dev):
        st = _stream(dev)
        contrib = _zero_contrib(dev, H, W)   # allocated, never written
        color = torch.empty((3, H, W), dtype=torch.float32, device=dev)
        feature = torch.empty((S, H, W), dtype=torch.float32, device=dev)
        others = torch.empty((7, H, W), dtype=torch.float32, device=dev)
        # radii [P] int32 and, right behind them, a byte per gaussian that takes radii > 0 (MRGS_HINT_VISIBLE_BYTES): one allocation
        rbuf = torch.empty((5 * P,), dtype=torch.uint8, device=dev)
        radii = rbuf[:4 * P].view(torch.int32)
        _LIVE.visible = rbuf[4 * P:].view(torch.bool)
        geom = torch.empty((L.mrgs_geom_bytes(P, H, W),), dtype=torch.uint8, device=dev)
        img = torch.empty((L.mrgs_img_bytes(H, W),), dtype=torch.uint8, device=dev)
        R = ctypes.c_int64(0)
        global LAST_NUM_RENDERED
        guess_key = (dev.index, P, H, W)
        guess = _PAIR_GUESS.get(guess_key)
        if guess is None and P > 0:
            # another surfel count at this image size (the step after a densification / pruning): the most recent count, scaled by the
            # count ratio + 30 % -- a guess like any other (an overflow is reported and the view redone, exactly sized)
            for (d_, p_, h_, w_), g_ in reversed(list(_PAIR_GUESS.items())):
                if (d_, h_, w_) == (dev.index, H, W) and p_ > 0:
                    guess = int(g_ * max(P / p_, 1.0) * 1.3)
                    break
        pairs = None            # pair count the binning workspace is carved for (what the backward must be given)
        box = getattr(_DEFER, "box", None)
        if guess is not None and P > 0:
            # one call, no host round trip in the middle of the GPU work: the workspace is sized from the previous call's
            # count and the kernels take the actual count from device memory (include/mrgs.h, mrgs_rasterize_forward[_begin])
            pairs = guess
            binning = torch.empty((L.mrgs_binning_bytes(pairs),), dtype=torch.uint8, device=dev)
            ticket = MrgsRasterTicket()
            _lib.check(L.mrgs_rasterize_forward_begin(ctypes.byref(cfg), ctypes.byref(inp), _ptr(geom), geom.numel(), _ptr(binning), binning.numel(),
                                                      pairs, _ptr(img), _ptr(radii), _ptr(color), _ptr(feature), _ptr(others), ctypes.byref(ticket), st))
            pending = _PendingCount(ticket, guess_key, raster_settings if hint is not None else None, dev)
            if box is not None:
                # a renderer queues its own kernels behind the rasterizer first and asks for the count at its end (deferred_count)
                # (the library keeps _TICKET_RING landing slots per thread and device: a box that begins more renders than that collects
                # its oldest counts now -- a wait, but no ticket ever goes stale; an overflow among them stays on record for box.finish())
                waiting = [q for q in box.pending if q.value is None]
                for q in waiting[:max(0, len(waiting) - (_TICKET_RING - 3))]:
                    try:
                        q.finish()
                    except RasterWorkspaceOverflow:
                        pass
                box.pending.append(pending)
                return (pending, pairs), contrib, color, feature, others, radii, geom, binning, img, grad_ws, hint
            try:
                num_rendered = pending.finish()
            except RasterWorkspaceOverflow as ex:
                num_rendered, pairs = ex.num_rendered, None    # the guess was too small: phase 2 is redone below on an exactly sized workspace
        else:
            _lib.check(L.mrgs_rasterize_forward_geom(ctypes.byref(cfg), ctypes.byref(inp), _ptr(geom), geom.numel(), _ptr(radii),
                                                     ctypes.byref(R), st))
            num_rendered = int(R.value)
        LAST_NUM_RENDERED = num_rendered
        if pairs is None:
            pairs = num_rendered
            binning = torch.empty((L.mrgs_binning_bytes(pairs),), dtype=torch.uint8, device=dev)
            _lib.check(L.mrgs_rasterize_forward_render(ctypes.byref(cfg), ctypes.byref(inp), _ptr(geom), _ptr(binning), binning.numel(),
                                                       _ptr(img), pairs, _ptr(color), _ptr(feature), _ptr(others), st))
            if P > 0:
                _note_count(guess_key, num_rendered, raster_settings if hint is not None else None, dev)
    return (num_rendered, pairs), contrib, color, feature, others, radii, geom, binning, img, grad_ws, hint


def _rasterize_backward_native(raster_settings, means3D, radii, colors_precomp, features, scales, rotations, cov3Ds_precomp,
                               grad_out_color, grad_out_feature, grad_out_others, sh, opacities, geom, num_rendered, binning, img,
                               sh_rest=None, prepared_grad_ws=None, work_hint=_RESOLVE, glue=None):
    """Counterpart of `_C.rasterize_gaussians_backward` (rasterize_points.cu:146-252).  prepared_grad_ws: the workspace the forward of
    this render was given (cleared, queues set up) -- valid for one backward.  work_hint: the hint buffer that forward used (its ctx
    keeps it); a prepared backward without it takes the non-prepared path (it orders and clears by itself).
    glue (MrgsRasterGrads::glue_params, the glue epilogue): an object with `.raw` = the nine raw GaussianModel tensors of
    renderer._SurfelFeatures (xyz, scaling, rotation, opacity, refl, rough, ori_color, indirect_dc, indirect_rest) whose outputs this render
    was fed; their gradients are left in `glue.results` and the five gradients they replace come back as None."""
    L = _lib.lib()
    dev = means3D.device
    if prepared_grad_ws is not None and (work_hint is _RESOLVE or work_hint is None):
        prepared_grad_ws = None            # the forward's queues cannot be named: never guess them
    cfg, inp, _ = _make_cfg_inputs(raster_settings, means3D, sh, colors_precomp, features, opacities, scales, rotations, cov3Ds_precomp,
                                   sh_rest, prepared_grad_ws, work_hint)
    P, S, M = cfg.P, cfg.S, cfg.M
    with _lib.guard(dev):
        st = _stream(dev)
        opts = dict(dtype=torch.float32, device=dev)
        fused = glue is not None and P > 0
        keep = (lambda name, shape: None) if fused else (lambda name, shape: torch.empty(shape, **opts))    # what the glue epilogue replaces
        g = {"dL_dmeans2D": torch.empty((P, 3), **opts),
             "dL_dcolors": None if fused and colors_precomp.numel() == 0 else torch.empty((P, 3), **opts),
             "dL_dfeatures": keep("dL_dfeatures", (P, S)), "dL_dopacity": keep("dL_dopacity", (P, 1)),
             "dL_dmeans3D": keep("dL_dmeans3D", (P, 3)),
             "dL_dtransMat": None if fused and cov3Ds_precomp.numel() == 0 else torch.empty((P, 9), **opts),
             "dL_dsh": torch.empty((P, 1 if sh_rest is not None else M, 3), **opts), "dL_dscales": keep("dL_dscales", (P, 2)),
             "dL_drotations": keep("dL_drotations", (P, 4)),
             "dL_dsh_rest": torch.empty((P, M - 1, 3), **opts) if sh_rest is not None else None}
        glue_prm = glue_out = raw_grads = None
        if fused:
            from ._lib import MrgsSurfelGrads, MrgsSurfelParams
            raw_grads = [torch.empty_like(t_) for t_ in glue.raw]
            glue_prm = MrgsSurfelParams(P, *[_ptr(t_) for t_ in glue.raw], None, _ptr(getattr(glue, "viewmatrix", None)))   # (campos: the rasterizer's)
            glue_out = MrgsSurfelGrads(*[_ptr(t_) for t_ in raw_grads])
        grads = MrgsRasterGrads(*[_ptr(g[name]) for name, _ in MrgsRasterGrads._fields_[1:11]],
                                ctypes.addressof(glue_prm) if fused else None, ctypes.addressof(glue_out) if fused else None)
        grad_ws = prepared_grad_ws if prepared_grad_ws is not None else torch.empty((L.mrgs_grad_bytes(P, S),), dtype=torch.uint8, device=dev)
        hook = _AFTER_BLEND_HOOK[0]
        if hook is None or P == 0:
            _lib.check(L.mrgs_rasterize_backward(ctypes.byref(cfg), ctypes.byref(inp), _ptr(radii), _ptr(geom), _ptr(binning), _ptr(img),
                                                 num_rendered, _ptr(grad_out_color), _ptr(grad_out_feature), _ptr(grad_out_others),
                                                 _ptr(grad_ws), ctypes.byref(grads), st))
        else:
            # two halves: the colour gradients are final after the blend backward; whatever the hook queues on another stream (a
            # view-parallel step: the all-gather of this factor, dist.FactoredGradReducer.begin_early) overlaps the per-gaussian backward
            drgb = torch.empty((P, 3), **opts)
            _lib.check(L.mrgs_rasterize_backward_blend(ctypes.byref(cfg), ctypes.byref(inp), _ptr(radii), _ptr(geom), _ptr(binning), _ptr(img),
                                                       num_rendered, _ptr(grad_out_color), _ptr(grad_out_feature), _ptr(grad_out_others),
                                                       _ptr(grad_ws), _ptr(drgb), st))
            hook(drgb)
            _lib.check(L.mrgs_rasterize_backward_finish(ctypes.byref(cfg), ctypes.byref(inp), _ptr(radii), _ptr(geom), _ptr(grad_ws),
                                                        ctypes.byref(grads), st))
    if fused:
        glue.results = raw_grads
    return (g["dL_dmeans2D"], g["dL_dcolors"], g["dL_dfeatures"], g["dL_dopacity"], g["dL_dmeans3D"], g["dL_dtransMat"],
            g["dL_dsh"], g["dL_dscales"], g["dL_drotations"], g["dL_dsh_rest"])


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, features, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings, sh_rest=None):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, features, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings, sh_rest)


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, features, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings, sh_rest=None):
        ctx.set_materialize_grads(False)   # output gradients nobody supplied arrive as None (handled in backward), not as zero-filled maps
        # sh_rest: split SH layout (extension over the reference's signature): sh = _features_dc [P,1,3], sh_rest = _features_rest
        # [P,M-1,3] -- the model's own tensors, no torch.cat per render and no slicing of the gradient in its backward
        means3D, sh, colors_precomp, features = _f32c(means3D), _f32c(sh), _f32c(colors_precomp), _f32c(features)
        sh_rest = None if sh_rest is None else _f32c(sh_rest)
        opacities, scales, rotations, cov3Ds_precomp = _f32c(opacities), _f32c(scales), _f32c(rotations), _f32c(cov3Ds_precomp)
        rs = raster_settings._replace(bg=_f32c(raster_settings.bg), viewmatrix=_camera_f32c(raster_settings.viewmatrix),
                                      projmatrix=_camera_f32c(raster_settings.projmatrix), campos=_camera_f32c(raster_settings.campos))
        args = (rs, means3D, sh, colors_precomp, features, opacities, scales, rotations, cov3Ds_precomp, sh_rest,
                any(ctx.needs_input_grad) and not _NO_PREPARE and means3D.is_cuda and _hint_is_warm(rs, means3D.device))
        if raster_settings.debug:
            cpu_args = cpu_deep_copy_tuple(args[1:])   # copy them before they can be corrupted
            try:
                out = _rasterize_forward_native(*args)
            except Exception as ex:
                torch.save(cpu_args, "snapshot_fw.dump")
                print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                raise ex
        else:
            out = _rasterize_forward_native(*args)
        (num_rendered, binning_pairs), contrib, color, feature, depth, radii, geomBuffer, binningBuffer, imgBuffer, grad_ws, hint = out
        ctx.features_live = int(getattr(_LIVE, "n", 0) or 0)
        ctx.glue = getattr(_LIVE, "glue", None)          # the glue epilogue of the backward (GaussianRasterizer.glue, set by render_surfel)
        ctx.prepared_grad_ws = grad_ws       # cleared by the forward, queues of the backward set up: good for ONE backward
        ctx.work_hint = hint                 # the buffer those queues live in: the backward is handed this very tensor (never a re-resolved one)
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        if isinstance(num_rendered, _PendingCount):      # begun inside deferred_count(): the count arrives with the box's finish()
            ctx.num_rendered, num_rendered.ctx = None, ctx
        ctx.binning_pairs = binning_pairs   # pair count binningBuffer is carved for (>= num_rendered)
        ctx.save_for_backward(colors_precomp, features, means3D, scales, rotations, cov3Ds_precomp, radii, sh, opacities,
                              geomBuffer, binningBuffer, imgBuffer, contrib, sh_rest)
        ctx.mark_non_differentiable(contrib, radii)
        return contrib, color, feature, radii, depth

    @staticmethod
    def backward(ctx, grad_out_contrib, grad_out_color, grad_out_feature, grad_radii, grad_depth):
        num_rendered = ctx.binning_pairs
        rs = ctx.raster_settings
        (colors_precomp, features, means3D, scales, rotations, cov3Ds_precomp, radii, sh, opacities, geomBuffer, binningBuffer,
         imgBuffer, contrib, sh_rest) = ctx.saved_tensors
        H, W = int(rs.image_height), int(rs.image_width)
        S = features.shape[1] if features.dim() == 2 else 0
        dev = means3D.device
        if grad_out_color is None:
            grad_out_color = torch.zeros((3, H, W), dtype=torch.float32, device=dev)
        if grad_out_feature is None:
            grad_out_feature = torch.zeros((S, H, W), dtype=torch.float32, device=dev)
        if grad_depth is None:
            grad_depth = torch.zeros((7, H, W), dtype=torch.float32, device=dev)
        prepared, ctx.prepared_grad_ws = ctx.prepared_grad_ws, None
        args = (rs, means3D, radii, colors_precomp, features, scales, rotations, cov3Ds_precomp, _f32c(grad_out_color),
                _f32c(grad_out_feature), _f32c(grad_depth), sh, opacities, geomBuffer, num_rendered, binningBuffer, imgBuffer, sh_rest, prepared,
                ctx.work_hint, getattr(ctx, "glue", None))
        _LIVE.n = getattr(ctx, "features_live", 0)       # (the autograd engine's thread: the forward's hint again)
        try:
            if rs.debug:
                cpu_args = cpu_deep_copy_tuple(args[1:-1])
                try:
                    out = _rasterize_backward_native(*args)
                except Exception as ex:
                    torch.save(cpu_args, "snapshot_bw.dump")
                    print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
                    raise ex
            else:
                out = _rasterize_backward_native(*args)
        finally:
            _LIVE.n = 0
        (grad_means2D, grad_colors_precomp, grad_features, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales,
         grad_rotations, grad_sh_rest) = out
        # empty inputs (the `torch.Tensor([])` placeholders) get empty gradients of matching shape
        if sh.numel() == 0:
            grad_sh = None
        if colors_precomp.numel() == 0:
            grad_colors_precomp = None
        if scales.numel() == 0:
            grad_scales = None
            grad_rotations = None
        if cov3Ds_precomp.numel() == 0:
            grad_cov3Ds_precomp = None
        return (grad_means3D, grad_means2D, grad_sh, grad_colors_precomp, grad_features, grad_opacities, grad_scales,
                grad_rotations, grad_cov3Ds_precomp, None, grad_sh_rest)


class GaussianRasterizationSettings(NamedTuple):
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


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        """Boolean mask of the points that pass the frustum test (`_C.mark_visible`, rasterize_points.cu:254-273)."""
        with torch.no_grad():
            rs = self.raster_settings
            positions = _f32c(positions)
            P = positions.shape[0]
            present = torch.zeros((P,), dtype=torch.uint8, device=positions.device)
            with _lib.guard(positions.device):
                _lib.check(_lib.lib().mrgs_mark_visible(P, _ptr(positions), _ptr(_f32c(rs.viewmatrix)), _ptr(_f32c(rs.projmatrix)),
                                                        _ptr(present), _stream(positions.device)))
            visible = present.bool()
        return visible

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, features=None, scales=None, rotations=None,
                cov3D_precomp=None):
        raster_settings = self.raster_settings

        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')

        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')

        empty = torch.empty((0,), dtype=torch.float32, device=means3D.device)
        # extension: shs may be the pair (features_dc [P,1,3], features_rest [P,M-1,3]) -- GaussianModel's own two tensors instead of
        # their concatenation (get_features, scene/gaussian_model.py:256-259)
        shs_rest = None
        if isinstance(shs, (tuple, list)):
            shs, shs_rest = shs
            if shs.dim() != 3 or shs.shape[1] != 1 or shs_rest.dim() != 3 or shs_rest.shape[1] < 1 or shs_rest.shape[1] > 15:
                raise Exception('split SHs must be (dc [P,1,3], rest [P,1..15,3])')
        if shs is None:
            shs = empty
        if colors_precomp is None:
            colors_precomp = empty
        if features is None:
            features = torch.empty_like(means3D[..., :0])
        if scales is None:
            scales = empty
        if rotations is None:
            rotations = empty
        if cov3D_precomp is None:
            cov3D_precomp = empty

        _LIVE.n = int(getattr(self, "features_live", 0) or 0)      # (extension: feature channels n .. S - 1 are zero padding of the rows)
        _LIVE.glue = getattr(self, "glue", None)                   # (extension: the glue epilogue of the backward, _rasterize_backward_native)
        _LIVE.visible = None
        try:
            out = rasterize_gaussians(means3D, means2D, shs, colors_precomp, features, opacities, scales, rotations, cov3D_precomp,
                                      raster_settings, shs_rest)
            # extension: radii > 0 of this render as a bool tensor, written by the forward itself (every render function of the reference
            # returns it as "visibility_filter"; `radii > 0` is a torch kernel per view otherwise)
            self.visible = getattr(_LIVE, "visible", None)
            return out
        finally:
            _LIVE.n = 0
            _LIVE.glue = None
            _LIVE.visible = None
