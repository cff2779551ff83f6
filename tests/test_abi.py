"""The C-ABI library loads on a machine without a GPU and exports every symbol include/mrgs.h declares
(no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "mrgs.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mrgs_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from materialrefgs_amd import _lib
    L = _lib.lib()
    names = declared_symbols()
    assert len(names) >= 14
    for n in names:
        assert hasattr(L, n), f"libmrgs.so does not export {n}"
        assert n in _lib.SYMBOLS, f"_lib.py has no prototype for {n}"
    assert set(_lib.SYMBOLS) == set(names)


def declared_prototypes():
    """name -> number of parameters, from the prototypes of include/mrgs.h (comments removed; `(void)` = 0)."""
    src = open(os.path.join(ROOT, "include", "mrgs.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(mrgs_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        params = m.group(2).strip()
        out[m.group(1)] = 0 if params in ("", "void") else params.count(",") + 1
    return out


def test_ctypes_prototypes_have_the_headers_parameter_counts():
    """Every prototype of materialrefgs_amd/_lib.py takes as many arguments as the declaration in include/mrgs.h it binds (a binding that is
    one pointer short or long fails on the CPU, not in the first backward on a GPU box)."""
    from materialrefgs_amd import _lib
    protos = declared_prototypes()
    assert set(protos) == set(_lib.SYMBOLS)
    for name, (_res, argtypes) in _lib.SYMBOLS.items():
        assert len(argtypes) == protos[name], (name, len(argtypes), protos[name])


def test_size_queries_and_error_strings():
    from materialrefgs_amd import _lib
    L = _lib.lib()
    assert L.mrgs_geom_bytes(1000, 128, 128) > 1000 * 80
    assert L.mrgs_geom_bytes(0, 16, 16) > 0
    assert L.mrgs_img_bytes(800, 800) >= 800 * 800 * 20
    assert L.mrgs_binning_bytes(10 ** 6) >= 16 * 10 ** 6
    assert L.mrgs_grad_bytes(1000, 8) >= 1000 * 26 * 4
    assert L.mrgs_strerror(0) == b"ok"
    assert b"feature" in L.mrgs_strerror(2)
    assert L.mrgs_version().startswith(b"mrgs")


def test_argument_validation_without_gpu():
    """Contract violations are reported as status codes before any HIP call is made."""
    from materialrefgs_amd import _lib
    from materialrefgs_amd._lib import MrgsRasterConfig, MrgsRasterInputs
    L = _lib.lib()
    cfg = MrgsRasterConfig(10, 25, 3, 16, 64, 64, 0.3, 0.3, 1.0, 0, 0)   # S = 25 > MAX_FEATURES
    inp = MrgsRasterInputs()
    R = ctypes.c_int64(-1)
    assert L.mrgs_rasterize_forward_geom(ctypes.byref(cfg), ctypes.byref(inp), None, 0, None, ctypes.byref(R), None) == 2
    cfg.S = 0
    assert L.mrgs_rasterize_forward_geom(ctypes.byref(cfg), ctypes.byref(inp), None, 0, None, ctypes.byref(R), None) == 1
    cfg.P = 0   # empty scene: nothing to launch, succeeds with R = 0 (rasterize_points.cu:106)
    assert L.mrgs_rasterize_forward_geom(ctypes.byref(cfg), ctypes.byref(inp), None, 0, None, ctypes.byref(R), None) == 0
    assert R.value == 0


def test_blocked_spmv_argument_checks_without_gpu():
    """val_bytes 8 (rows as blocks of four columns) needs 16-bit block indices, 64 lanes per row, a row count that is a multiple of 4
    and aligned x / val: everything else is MRGS_E_BAD_ARG before anything is launched."""
    from materialrefgs_amd import _lib
    L = _lib.lib()
    p = ctypes.c_void_p(0x1000)      # never dereferenced: the calls below are all refused
    ok_args = dict(nrows=64, col_bytes=2, lanes=64, x=0x1000, val=0x1000)
    def call(**kw):
        a = dict(ok_args, **kw)
        return L.mrgs_csr_spmv3(a["nrows"], p, p, a["col_bytes"], ctypes.c_void_p(a["val"]), 8, p, ctypes.c_void_p(a["x"]), p, a["lanes"], None)
    assert call(col_bytes=4) == 1
    assert call(lanes=4) == 1
    assert call(nrows=66) == 1
    assert call(x=0x1004) == 1
    assert call(val=0x1004) == 1
    assert L.mrgs_csr_spmv3(64, p, p, 2, p, 8, None, p, p, 64, None) == 1      # fixed-point weights need the row scales


def test_struct_size_guard():
    """A caller built against another revision of include/mrgs.h -- e.g. the 14-pointer MrgsRasterInputs of the header before
    `bwd_grad_ws` was appended, which the forward ACTS on -- is refused with MRGS_E_BAD_ARG before anything is read (no GPU needed:
    the check precedes every HIP call).  sizeof of the ctypes declarations is pinned to the header's layout."""
    from materialrefgs_amd import _lib
    from materialrefgs_amd._lib import MrgsRasterConfig, MrgsRasterGrads, MrgsRasterInputs
    L = _lib.lib()
    assert L.mrgs_abi_version() == _lib.MRGS_ABI_VERSION == 10
    assert ctypes.sizeof(MrgsRasterConfig) == 4 + 11 * 4          # struct_size + 6 ints + 3 floats + 2 ints
    assert ctypes.sizeof(MrgsRasterInputs) == 8 + 15 * 8 + 8      # struct_size + 12 pointers + work_hint, shs_rest, bwd_grad_ws + hint_flags, reserved
    assert ctypes.sizeof(MrgsRasterGrads) == 8 + 12 * 8          # struct_size + 10 gradient pointers + glue_params, glue_grads (ABI 10)
    hdr = open(os.path.join(ROOT, "include", "mrgs.h")).read()
    assert "#define MRGS_ABI_VERSION 10" in hdr

    class OldInputs(ctypes.Structure):                           # the struct as INTEGRATION.md printed it in round 2: 14 pointers, no size
        _fields_ = [(n, ctypes.c_void_p) for n in ("bg", "means3D", "shs", "colors_precomp", "features", "opacities", "scales", "rotations",
                                                   "transMat_precomp", "viewmatrix", "projmatrix", "campos", "work_hint", "shs_rest")]

    class ShortInputs(ctypes.Structure):                         # right first field, one trailing pointer short
        _fields_ = [("struct_size", ctypes.c_uint64)] + OldInputs._fields_

    cfg = MrgsRasterConfig(10, 0, 3, 16, 64, 64, 0.3, 0.3, 1.0, 0, 0)
    R = ctypes.c_int64(-1)
    fwd = L.mrgs_rasterize_forward_geom
    keep = fwd.argtypes
    fwd.argtypes = None                                          # let the foreign structs through the ctypes type check
    try:
        junk = ctypes.c_void_p(0xdead0000)
        old = OldInputs(*([junk] * 14))
        assert fwd(ctypes.byref(cfg), ctypes.byref(old), None, ctypes.c_size_t(0), None, ctypes.byref(R), None) == 1
        short = ShortInputs(ctypes.sizeof(ShortInputs), *([junk] * 14))
        assert fwd(ctypes.byref(cfg), ctypes.byref(short), None, ctypes.c_size_t(0), None, ctypes.byref(R), None) == 1
        good = MrgsRasterInputs()
        bad_cfg = MrgsRasterConfig(0, 0, 3, 16, 64, 64, 0.3, 0.3, 1.0, 0, 0)
        bad_cfg.struct_size = 44                                 # the round-2 MrgsRasterConfig had no size field
        assert fwd(ctypes.byref(bad_cfg), ctypes.byref(good), None, ctypes.c_size_t(0), None, ctypes.byref(R), None) == 1
        bad_cfg.struct_size = ctypes.sizeof(MrgsRasterConfig)    # P = 0 with the right sizes: accepted, nothing to launch
        assert fwd(ctypes.byref(bad_cfg), ctypes.byref(good), None, ctypes.c_size_t(0), None, ctypes.byref(R), None) == 0
    finally:
        fwd.argtypes = keep
    g = MrgsRasterGrads()
    g.struct_size = 80
    assert L.mrgs_rasterize_backward(ctypes.byref(bad_cfg), ctypes.byref(good), None, None, None, None, 0, None, None, None, None,
                                     ctypes.byref(g), None) == 1


def test_python_wrapper_validation():
    """Same exceptions as the reference wrapper (diff_surfel_rasterization/__init__.py:201-205)."""
    import torch
    from materialrefgs_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    rs = GaussianRasterizationSettings(16, 16, 0.3, 0.3, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), 0, torch.zeros(3), False, False)
    rast = GaussianRasterizer(rs)
    m = torch.zeros(4, 3)
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        rast(means3D=m, means2D=m, opacities=torch.zeros(4, 1), scales=torch.ones(4, 2), rotations=torch.ones(4, 4))
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        rast(means3D=m, means2D=m, opacities=torch.zeros(4, 1), shs=torch.zeros(4, 16, 3), colors_precomp=torch.zeros(4, 3),
             scales=torch.ones(4, 2), rotations=torch.ones(4, 4))
    with pytest.raises(Exception, match="scale/rotation pair or precomputed"):
        rast(means3D=m, means2D=m, opacities=torch.zeros(4, 1), shs=torch.zeros(4, 16, 3))
    with pytest.raises(RuntimeError, match="CUDA tensor"):   # CHECK_INPUT: CPU tensors are rejected, no CPU fallback
        rast(means3D=m, means2D=m, opacities=torch.zeros(4, 1), shs=torch.zeros(4, 16, 3), scales=torch.ones(4, 2), rotations=torch.ones(4, 4))


def test_camera_matrix_copies_are_cached_per_source_tensor():
    """rasterizer._camera_f32c: a non-contiguous camera matrix (the reference's transposed views, scene/cameras.py:77) maps to ONE
    contiguous copy for as long as it is not written in place -- the address the work hints are keyed by."""
    import torch
    from materialrefgs_amd import rasterizer as rz
    rz.reset_work_hints()
    m = torch.arange(16, dtype=torch.float32).reshape(4, 4)
    assert rz._camera_f32c(m) is m                                  # already contiguous fp32: itself
    t = m.transpose(0, 1)
    c1, c2 = rz._camera_f32c(t), rz._camera_f32c(m.transpose(0, 1))      # a new view object of the same storage and strides
    assert c1 is c2 and c1.is_contiguous() and torch.equal(c1, t)
    m.add_(1.0)                                                     # written in place: copied again
    c3 = rz._camera_f32c(t)
    assert c3 is not c1 and torch.equal(c3, t)
    d = rz._camera_f32c(t.double())
    assert d.dtype == torch.float32 and torch.equal(d, t)


def test_diff_surfel_rasterization_shim_resolves_to_the_hip_rasterizer():
    """The reference's render functions import `diff_surfel_rasterization`; the shim package at the repo root must hand them the
    classes of materialrefgs_amd.rasterizer (whose native side is libmrgs.so -- there is no other implementation behind them)."""
    import diff_surfel_rasterization as dsr
    from materialrefgs_amd import rasterizer
    assert dsr.GaussianRasterizer is rasterizer.GaussianRasterizer
    assert dsr.GaussianRasterizationSettings is rasterizer.GaussianRasterizationSettings
    assert dsr.GaussianRasterizationSettings._fields == ("image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier",
                                                         "viewmatrix", "projmatrix", "sh_degree", "campos", "prefiltered", "debug")
    import torch
    rs = dsr.GaussianRasterizationSettings(16, 16, 1.0, 1.0, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), 0, torch.zeros(3), False, False)
    rast = dsr.GaussianRasterizer(raster_settings=rs)
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        rast(means3D=torch.zeros(1, 3), means2D=torch.zeros(1, 3), opacities=torch.zeros(1, 1))
    # CPU tensors are refused by the native front-end (CHECK_INPUT of rasterize_points.cu:29-31), never rendered by something else
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        rast(means3D=torch.zeros(1, 3), means2D=torch.zeros(1, 3), opacities=torch.zeros(1, 1), shs=torch.zeros(1, 16, 3),
             scales=torch.ones(1, 2), rotations=torch.tensor([[1.0, 0, 0, 0]]))


@pytest.mark.gpu
def test_shim_renders_through_libmrgs(gpu_device):
    import math
    import torch
    import diff_surfel_rasterization as dsr
    import kat
    cam = kat.frontal_camera(96, 128)
    sc = kat.one_surfel_scene(**kat.FRONTAL).to(gpu_device)
    rs = dsr.GaussianRasterizationSettings(
        image_height=96, image_width=128, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
        bg=torch.zeros(3, device=gpu_device), scale_modifier=1.0, viewmatrix=cam.world_view_transform.to(gpu_device),
        projmatrix=cam.full_proj_transform.to(gpu_device), sh_degree=0, campos=cam.camera_center.to(gpu_device), prefiltered=False, debug=False)
    contrib, color, feature, radii, allmap = dsr.GaussianRasterizer(rs)(
        means3D=sc.means3D, means2D=torch.zeros_like(sc.means3D), opacities=sc.opacities, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
    cf = kat.closed_form(cam, **kat.FRONTAL)
    inside = cf["rho3d"] <= 8.5
    assert abs(allmap[1].cpu().numpy()[inside] - cf["alpha"][inside]).max() < 3e-5
    assert contrib.shape == (1, 96, 128) and contrib.dtype == torch.int32 and radii.dtype == torch.int32


def test_raster_ticket_guards():
    """mrgs_rasterize_forward_finish without a GPU: the ticket of an empty model reports 0 pairs, a ticket this thread never began is refused."""
    from materialrefgs_amd import _lib
    from materialrefgs_amd._lib import MrgsRasterTicket
    L = _lib.lib()
    R = ctypes.c_int64(7)
    t = MrgsRasterTicket(-1, -1, 0, 0)
    assert L.mrgs_rasterize_forward_finish(ctypes.byref(t), ctypes.byref(R)) == 0 and R.value == 0
    for bad in (MrgsRasterTicket(0, 3, 12345, 10), MrgsRasterTicket(0, 99, 1, 10), MrgsRasterTicket(1000, 0, 1, 10)):
        assert L.mrgs_rasterize_forward_finish(ctypes.byref(bad), ctypes.byref(R)) == 1        # MRGS_E_BAD_ARG
    assert L.mrgs_rasterize_forward_finish(None, ctypes.byref(R)) == 1
    assert ctypes.sizeof(MrgsRasterTicket) == 24


def test_camera_constants_follow_an_in_place_pose_change():
    """renderer._camera_consts caches K^-1 / R / T per camera object; a pose refined in place (same object, same arrays) or new
    intrinsics must not be served from the cache."""
    from types import SimpleNamespace
    import numpy as np
    import torch
    from materialrefgs_amd import renderer
    K = np.array([[500.0, 0, 32], [0, 500.0, 24], [0, 0, 1]], dtype=np.float32)
    cam = SimpleNamespace(R=np.eye(3, dtype=np.float32), T=np.array([0.0, 0.0, 3.0], dtype=np.float32), HWK=(48, 64, K))
    dev = torch.device("cpu")
    k0, r0, t0 = renderer._camera_consts(cam, dev)
    k1, r1, t1 = renderer._camera_consts(cam, dev)
    assert r1 is r0 and t1 is t0 and k1 == k0                      # unchanged camera: the cached tensors
    cam.T[2] = 5.0                                                 # in place
    _, _, t2 = renderer._camera_consts(cam, dev)
    assert float(t2[2]) == 5.0
    cam.R[:] = np.array([[0, 1, 0], [-1, 0, 0], [0, 0, 1]], dtype=np.float32)
    _, r3, _ = renderer._camera_consts(cam, dev)
    assert float(r3[0, 1]) == 1.0 and float(r3[0, 0]) == 0.0
    K[0, 0] = 250.0
    k4, _, _ = renderer._camera_consts(cam, dev)
    assert abs(k4[0] - 1.0 / 250.0) < 1e-9


def test_glue_epilogue_arguments_are_checked_before_any_launch():
    """MrgsRasterGrads::glue_params / glue_grads (ABI 10): one pointer without the other or a missing tensor is MRGS_E_BAD_ARG, a render the
    epilogue does not serve -- row shapes other than eight channels or the "pgsr" nine-in-twelve with its viewmatrix -- MRGS_E_UNSUPPORTED, both before anything is
    queued (no GPU needed: the calls below carry no `radii`, which is the next check and fails them all the same).  With the epilogue the five
    tensors it replaces may be NULL."""
    from materialrefgs_amd import _lib
    from materialrefgs_amd._lib import MrgsRasterConfig, MrgsRasterGrads, MrgsRasterInputs, MrgsSurfelGrads, MrgsSurfelParams
    L = _lib.lib()
    BAD_ARG, UNSUPPORTED = 1, 6
    P = 4
    buf = (ctypes.c_float * 4096)()
    ptr = ctypes.addressof(buf)
    cfg = MrgsRasterConfig(P, 8, 3, 16, 32, 32, 0.5, 0.5, 1.0, 0, 0)
    cfg12 = MrgsRasterConfig(P, 12, 3, 16, 32, 32, 0.5, 0.5, 1.0, 0, 0)
    inp = MrgsRasterInputs(ptr, ptr, ptr, None, ptr, ptr, ptr, ptr, None, ptr, ptr, ptr, None, None, None, 0, 0)
    prm, prm_vm = MrgsSurfelParams(P, *([ptr] * 10), None), MrgsSurfelParams(P, *([ptr] * 11))
    out = MrgsSurfelGrads(*([ptr] * 9))
    out_short = MrgsSurfelGrads(*([ptr] * 8), None)

    inp9 = MrgsRasterInputs(ptr, ptr, ptr, None, ptr, ptr, ptr, ptr, None, ptr, ptr, ptr, None, None, None, 0, 9)

    def finish(grads, cfg_=cfg, inp_=inp):
        return L.mrgs_rasterize_backward_finish(ctypes.byref(cfg_), ctypes.byref(inp_), None, ptr, ptr, ctypes.byref(grads), None)
    full = [ptr] * 9 + [None]
    lean = [ptr, None, None, None, None, None, ptr, None, None, None]                                 # only dL_dmeans2D and dL_dsh
    a = ctypes.addressof
    assert finish(MrgsRasterGrads(*full, None, None)) == BAD_ARG                                      # (well formed: stops at `radii`)
    assert finish(MrgsRasterGrads(*lean, a(prm), a(out))) == BAD_ARG                                  # (well formed with the epilogue: stops at `radii`)
    assert finish(MrgsRasterGrads(*lean, None, None)) == BAD_ARG                                      # tensors missing without it
    assert finish(MrgsRasterGrads(*full, a(prm), None)) == BAD_ARG                                    # one pointer without the other
    assert finish(MrgsRasterGrads(*lean, a(prm), a(out_short))) == BAD_ARG                            # a raw gradient tensor missing
    assert finish(MrgsRasterGrads(*lean, a(prm), a(out)), cfg12) == UNSUPPORTED                       # twelve channels without the "pgsr" viewmatrix
    assert finish(MrgsRasterGrads(*lean, a(prm_vm), a(out))) == UNSUPPORTED                           # the viewmatrix with rows of eight
    assert finish(MrgsRasterGrads(*lean, a(prm_vm), a(out)), cfg12) == UNSUPPORTED                    # twelve channels, all of them live
    assert finish(MrgsRasterGrads(*lean, a(prm_vm), a(out)), cfg12, inp9) == BAD_ARG                  # "pgsr" rows (features_live = 9): well formed, stops at `radii`
