"""ctypes front-end of oracle/mrgs_oracle.c -- TEST INFRASTRUCTURE ONLY (parity unpinned, see the C header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# arithmetic modes of mrgs_oracle.c (see its header / oracle/Makefile):
#   fused  fp32, blend multiply-adds fused in the HIP kernels' pattern -- the bit-level checker of the parity tests
#   nocut  fused without the alpha < 1/255 cut-off (smooth forward for the finite-difference tests)
#   lit32  fp32, the reference's expression trees as written, nothing fused ("literal reading")
#   f64    the same literal trees in double on the same fp32 inputs ("true value of the reference's formulas")
VARIANTS = {"fused": "libmrgs_oracle.so", "nocut": "libmrgs_oracle_nocut.so", "lit32": "libmrgs_oracle_lit32.so",
            "f64": "libmrgs_oracle_f64.so"}
DEFAULT_VARIANT = "nocut" if os.environ.get("MRGS_ORACLE_NOCUT") == "1" else "fused"
_LIB_PATH = os.path.join(_HERE, VARIANTS[DEFAULT_VARIANT])
_libs = {}

FIELDS = {  # name -> (index in mrgs_oracle_field, dtype, shape builder)
    "depths": (0, np.float32, lambda c: (c.P,)),
    "radii": (1, np.int32, lambda c: (c.P,)),
    "means2D": (2, np.float32, lambda c: (c.P, 2)),
    "transMat": (3, np.float32, lambda c: (c.P, 9)),
    "normal_opacity": (4, np.float32, lambda c: (c.P, 4)),
    "rgb": (5, np.float32, lambda c: (c.P, 3)),
    "tiles_touched": (6, np.uint32, lambda c: (c.P,)),
    "clamped": (7, np.uint8, lambda c: (c.P, 3)),
    "point_offsets": (8, np.uint32, lambda c: (c.P,)),
    "keys": (9, np.uint64, lambda c: (c.R,)),
    "point_list": (10, np.uint32, lambda c: (c.R,)),
    "ranges": (11, np.uint32, lambda c: (c.tiles, 2)),
    "final_T": (12, np.float32, lambda c: (3, c.H, c.W)),
    "n_contrib": (13, np.uint32, lambda c: (2, c.H, c.W)),
    "color": (14, np.float32, lambda c: (3, c.H, c.W)),
    "feature": (15, np.float32, lambda c: (c.S, c.H, c.W)),
    "others": (16, np.float32, lambda c: (7, c.H, c.W)),
}


def build(force=False):
    src = os.path.getmtime(os.path.join(_HERE, "mrgs_oracle.c"))
    paths = [os.path.join(_HERE, f) for f in VARIANTS.values()]
    if force or any(not os.path.exists(q) or os.path.getmtime(q) < src for q in paths):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


def lib(variant=None):
    variant = variant or DEFAULT_VARIANT
    if variant not in _libs:
        build()
        L = ctypes.CDLL(os.path.join(_HERE, VARIANTS[variant]))
        fp = ctypes.c_void_p
        L.mrgs_oracle_forward.restype = ctypes.c_void_p
        L.mrgs_oracle_forward.argtypes = [ctypes.c_int] * 6 + [fp] * 7 + [ctypes.c_float] + [fp] * 5 + [ctypes.c_float] * 2
        L.mrgs_oracle_free.argtypes = [ctypes.c_void_p]
        L.mrgs_oracle_num_rendered.restype = ctypes.c_int64
        L.mrgs_oracle_num_rendered.argtypes = [ctypes.c_void_p]
        L.mrgs_oracle_field.restype = ctypes.c_void_p
        L.mrgs_oracle_field.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.mrgs_oracle_backward.restype = ctypes.c_int
        L.mrgs_oracle_backward.argtypes = [ctypes.c_void_p] + [fp] * 13
        L.mrgs_oracle_preprocess_backward_only.argtypes = [ctypes.c_void_p] + [fp] * 8
        L.mrgs_oracle_mark_visible.argtypes = [ctypes.c_int, fp, fp, fp, fp]
        L.mrgs_oracle_num_threads.restype = ctypes.c_int
        L.mrgs_oracle_real_bytes.restype = ctypes.c_int
        L.real = np.float64 if L.mrgs_oracle_real_bytes() == 8 else np.float32
        _libs[variant] = L
    return _libs[variant]


def _real(a, shape=None, dtype=np.float32):
    """The inputs are the fp32 tensors the reference would be given; the f64 build receives the same values widened."""
    if a is None:
        return None
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    a = np.ascontiguousarray(np.ascontiguousarray(a, dtype=np.float32), dtype=dtype)
    if shape is not None:
        a = a.reshape(shape)
    return a


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class OracleRender:
    """One forward pass of the CPU oracle; keeps the context alive for backward()."""

    def __init__(self, *, means3D, opacities, H, W, tanfovx, tanfovy, viewmatrix, projmatrix, campos, bg=None,
                 shs=None, colors_precomp=None, features=None, scales=None, rotations=None, transMat_precomp=None,
                 scale_modifier=1.0, sh_degree=0, variant=None):
        L = self._L = lib(variant)
        self.variant = variant or DEFAULT_VARIANT
        rt = self.rt = L.real
        _f32 = lambda a, shape=None: _real(a, shape, rt)   # noqa: E731 (every float array crosses the ABI in the build's type)
        self.means3D = _f32(means3D).reshape(-1, 3)
        self.P = self.means3D.shape[0]
        self.H, self.W = int(H), int(W)
        self.tiles = ((W + 15) // 16) * ((H + 15) // 16)
        self.shs = _f32(shs)
        self.M = 0 if self.shs is None or self.shs.size == 0 else self.shs.reshape(self.P, -1, 3).shape[1]
        if self.shs is not None and self.shs.size == 0:
            self.shs = None
        self.colors_precomp = _f32(colors_precomp)
        if self.colors_precomp is not None and self.colors_precomp.size == 0:
            self.colors_precomp = None
        assert (self.shs is None) != (self.colors_precomp is None) or self.P == 0
        self.features = _f32(features)
        self.S = 0 if self.features is None else self.features.reshape(self.P, -1).shape[1] if self.P else 0
        if self.features is None:
            self.features = np.zeros((self.P, 0), rt)
        self.opacities = _f32(opacities).reshape(-1)
        self.scales = _f32(scales)
        self.rotations = _f32(rotations)
        self.transMat_precomp = _f32(transMat_precomp)
        if self.scales is not None and self.scales.size == 0:
            self.scales = None
            self.rotations = None
        if self.transMat_precomp is not None and self.transMat_precomp.size == 0:
            self.transMat_precomp = None
        self.view = _f32(viewmatrix).reshape(16)
        self.proj = _f32(projmatrix).reshape(16)
        self.campos = _f32(campos).reshape(3)
        self.bg = np.zeros(3, rt) if bg is None else _f32(bg).reshape(3)
        self.D = int(sh_degree)
        self._ctx = L.mrgs_oracle_forward(
            self.P, self.S, self.D, self.M, self.H, self.W, _ptr(self.bg), _ptr(self.means3D), _ptr(self.shs),
            _ptr(self.colors_precomp), _ptr(self.features), _ptr(self.opacities), _ptr(self.scales),
            ctypes.c_float(scale_modifier), _ptr(self.rotations), _ptr(self.transMat_precomp), _ptr(self.view),
            _ptr(self.proj), _ptr(self.campos), ctypes.c_float(tanfovx), ctypes.c_float(tanfovy))
        if not self._ctx:
            raise RuntimeError("mrgs_oracle_forward failed")
        self.R = int(L.mrgs_oracle_num_rendered(self._ctx))

    def field(self, name):
        idx, dtype, shp = FIELDS[name]
        if dtype == np.float32:
            dtype = self.rt
        shape = shp(self)
        n = int(np.prod(shape))
        if n == 0:
            return np.zeros(shape, dtype)
        p = self._L.mrgs_oracle_field(self._ctx, idx)
        arr = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape=(n * np.dtype(dtype).itemsize,))
        return arr.view(dtype).reshape(shape).copy()

    def __getattr__(self, name):
        if name in FIELDS:
            return self.field(name)
        raise AttributeError(name)

    def backward(self, dL_dcolor, dL_dfeature, dL_dothers):
        P, S, M = self.P, self.S, self.M
        rt = self.rt
        _f32 = lambda a, shape=None: _real(a, shape, rt)   # noqa: E731
        g_c = _f32(dL_dcolor, (3, self.H, self.W))
        g_f = _f32(dL_dfeature, (S, self.H, self.W)) if S else np.zeros((0, self.H, self.W), rt)
        g_o = _f32(dL_dothers, (7, self.H, self.W))
        out = {
            "means2D": np.zeros((P, 3), rt), "normal": np.zeros((P, 3), rt),
            "opacity": np.zeros((P, 1), rt), "colors": np.zeros((P, 3), rt),
            "features": np.zeros((P, S), rt), "means3D": np.zeros((P, 3), rt),
            "transMat": np.zeros((P, 9), rt), "sh": np.zeros((P, M, 3), rt),
            "scales": np.zeros((P, 2), rt), "rotations": np.zeros((P, 4), rt),
        }
        order = ["means2D", "normal", "opacity", "colors", "features", "means3D", "transMat", "sh", "scales", "rotations"]
        rc = self._L.mrgs_oracle_backward(self._ctx, _ptr(g_c), _ptr(g_f), _ptr(g_o), *[_ptr(out[k]) for k in order])
        if rc != 0:
            raise RuntimeError("mrgs_oracle_backward failed")
        return out

    def preprocess_backward_only(self, dL_dtransMat, dL_dnormal, dL_dmean2D, dL_dcolors):
        """Test hook: the per-gaussian backward alone, on caller-supplied upstream gradients."""
        P, M = self.P, self.M
        rt = self.rt
        _f32 = lambda a, shape=None: _real(a, shape, rt)   # noqa: E731
        dT = _f32(dL_dtransMat, (P, 9)).copy()
        dn = _f32(dL_dnormal, (P, 3))
        dm2 = _f32(dL_dmean2D, (P, 3)).copy()
        dc = _f32(dL_dcolors, (P, 3))
        out = {"sh": np.zeros((P, M, 3), rt), "means3D": np.zeros((P, 3), rt),
               "scales": np.zeros((P, 2), rt), "rotations": np.zeros((P, 4), rt)}
        self._L.mrgs_oracle_preprocess_backward_only(self._ctx, _ptr(dT), _ptr(dn), _ptr(dm2), _ptr(dc), _ptr(out["sh"]),
                                                   _ptr(out["means3D"]), _ptr(out["scales"]), _ptr(out["rotations"]))
        out["transMat"] = dT
        out["means2D"] = dm2
        return out

    def close(self):
        if getattr(self, "_ctx", None):
            self._L.mrgs_oracle_free(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def render_scene(scene, cam, *, S=None, sh_degree=3, scale_modifier=1.0, colors_precomp=None, bg=None, variant=None):
    """Convenience: run the oracle on a materialrefgs_amd.synthetic.Scene + camera.MiniCam."""
    import math
    return OracleRender(
        means3D=scene.means3D, opacities=scene.opacities, H=cam.image_height, W=cam.image_width,
        tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), viewmatrix=cam.world_view_transform,
        projmatrix=cam.full_proj_transform, campos=cam.camera_center, bg=bg,
        shs=None if colors_precomp is not None else scene.shs, colors_precomp=colors_precomp,
        features=scene.features, scales=scene.scales, rotations=scene.rotations, scale_modifier=scale_modifier,
        sh_degree=sh_degree, variant=variant)


def mark_visible(means3D, viewmatrix, projmatrix):
    m = _real(means3D).reshape(-1, 3)
    out = np.zeros(m.shape[0], np.uint8)
    lib("fused").mrgs_oracle_mark_visible(m.shape[0], _ptr(m), _ptr(_real(viewmatrix).reshape(16)), _ptr(_real(projmatrix).reshape(16)), _ptr(out))
    return out.astype(bool)


def num_threads():
    return int(lib().mrgs_oracle_num_threads())
