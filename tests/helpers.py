"""Shared helpers of the parity tests: run the HIP rasterizer through its public (reference-shaped) API."""
import ctypes
import math

import numpy as np
import torch

EXPORT = {"depths": (0, torch.float32, lambda P, R, T, H, W: (P,)),
          "means2D": (1, torch.float32, lambda P, R, T, H, W: (P, 2)),
          "transMat": (2, torch.float32, lambda P, R, T, H, W: (P, 9)),
          "normal_opacity": (3, torch.float32, lambda P, R, T, H, W: (P, 4)),
          "rgb": (4, torch.float32, lambda P, R, T, H, W: (P, 3)),
          "tiles_touched": (5, torch.int32, lambda P, R, T, H, W: (P,)),
          "clamped": (6, torch.uint8, lambda P, R, T, H, W: (P, 3)),
          "point_list": (7, torch.int32, lambda P, R, T, H, W: (R,)),
          "ranges": (8, torch.int32, lambda P, R, T, H, W: (T, 2)),
          "final_T": (9, torch.float32, lambda P, R, T, H, W: (3, H, W)),
          "n_contrib": (10, torch.int32, lambda P, R, T, H, W: (2, H, W)),
          "order": (11, torch.int32, lambda P, R, T, H, W: (P,)),
          "redo_list": (12, torch.int32, lambda P, R, T, H, W: (2 + H * W,)),
          "qmask": (13, torch.uint8, lambda P, R, T, H, W: (R,)),
          "cull": (14, torch.float32, lambda P, R, T, H, W: (P, 12))}


def raster_settings(cam, device, sh_degree=3, scale_modifier=1.0, bg=None, debug=False):
    from materialrefgs_amd.rasterizer import GaussianRasterizationSettings
    bg = torch.zeros(3, device=device) if bg is None else bg.to(device)
    return GaussianRasterizationSettings(
        image_height=cam.image_height, image_width=cam.image_width, tanfovx=math.tan(cam.FoVx * 0.5),
        tanfovy=math.tan(cam.FoVy * 0.5), bg=bg, scale_modifier=scale_modifier,
        viewmatrix=cam.world_view_transform.to(device), projmatrix=cam.full_proj_transform.to(device), sh_degree=sh_degree,
        campos=cam.camera_center.to(device), prefiltered=False, debug=debug)


class HipRender:
    """Forward (+ optional backward) through materialrefgs_amd.rasterizer, keeping handles for introspection."""

    def __init__(self, scene, cam, device, sh_degree=3, scale_modifier=1.0, colors_precomp=None, bg=None, use_features=True, rs=None, features_live=0):
        from materialrefgs_amd.rasterizer import GaussianRasterizer
        self.dev = device
        sc = scene.to(device)
        self.leaves = {}
        def leaf(name, t):
            t = t.clone().requires_grad_(True)
            self.leaves[name] = t
            return t
        self.rs = rs if rs is not None else raster_settings(cam, device, sh_degree, scale_modifier, bg)   # rs: the settings (camera tensors) of an earlier render
        means3D = leaf("means3D", sc.means3D)
        means2D = leaf("means2D", torch.zeros_like(sc.means3D))
        opac = leaf("opacity", sc.opacities)
        scales = leaf("scales", sc.scales)
        rots = leaf("rotations", sc.rotations)
        feats = leaf("features", sc.features) if (use_features and sc.features.shape[1] > 0) else None
        kw = {}
        if colors_precomp is None:
            kw["shs"] = leaf("sh", sc.shs)
        else:
            kw["colors_precomp"] = leaf("colors", colors_precomp.to(device))
        rast = GaussianRasterizer(self.rs)
        rast.features_live = features_live         # extension: feature channels from here on are zero padding of the rows
        self.contrib, self.color, self.feature, self.radii, self.others = rast(
            means3D=means3D, means2D=means2D, opacities=opac, features=feats, scales=scales, rotations=rots, **kw)
        self.fn = self.color.grad_fn
        self.P = sc.means3D.shape[0]
        self.H, self.W = cam.image_height, cam.image_width
        self.S = 0 if feats is None else feats.shape[1]

    @property
    def num_rendered(self):
        return self.fn.num_rendered

    def export(self, name):
        from materialrefgs_amd import _lib
        from materialrefgs_amd._lib import MrgsRasterConfig
        which, dtype, shp = EXPORT[name]
        saved = self.fn.saved_tensors
        geom, binning, img = saved[9], saved[10], saved[11]
        R = self.num_rendered
        pairs = self.fn.binning_pairs      # what the binning workspace is carved for (>= R when the forward ran on a guess)
        T = ((self.W + 15) // 16) * ((self.H + 15) // 16)
        shape = shp(self.P, pairs if name in ("point_list", "qmask") else R, T, self.H, self.W)
        out = torch.zeros(shape, dtype=dtype, device=self.dev)
        cfg = MrgsRasterConfig(self.P, self.S, 0, 0, self.H, self.W, 0.0, 0.0, 1.0, 0, 0)
        p = lambda t: ctypes.c_void_p(t.data_ptr()) if t.numel() else None
        if out.numel():
            _lib.check(_lib.lib().mrgs_debug_export(ctypes.byref(cfg), p(geom), p(binning), p(img), pairs, which, p(out),
                                                    ctypes.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)))
        torch.cuda.synchronize(self.dev)
        out = out.cpu().numpy()
        return out[:R] if name in ("point_list", "qmask") else out

    def backward(self, g_color, g_feat, g_others):
        outs, grads = [self.color, self.others], [g_color.to(self.dev), g_others.to(self.dev)]
        if self.S > 0:
            outs.append(self.feature)
            grads.append(g_feat.to(self.dev))
        torch.autograd.backward(outs, grads)
        torch.cuda.synchronize(self.dev)
        return {k: v.grad.detach().cpu().numpy() for k, v in self.leaves.items() if v.grad is not None}


def rel_err(a, b):
    """max |a-b| / max |b| (the tensor-level relative error of BASELINE.json's 'grad max-rel-err')."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    denom = np.abs(b).max()
    if denom == 0:
        return float(np.abs(a).max())
    return float(np.abs(a - b).max() / denom)
