"""Known-answer scenes for the surfel rasterizer, derived BY HAND from the reference's formulas -- independent of
oracle/mrgs_oracle.c and of the HIP kernels, evaluated here in float64 numpy.

Set-up: a camera at the origin looking down +z with R = I, T = 0 (view space == world space), ONE surfel.

Derivation (file:line relative to submodules/diff-surfel-rasterization/cuda_rasterizer/):
  * a world point X = (X, Y, Z) lands on pixel  px = fx X / Z + (W - 1) / 2,  py = fy Y / Z + (H - 1) / 2  with
    fx = W / (2 tan(FoVx / 2)): full_proj maps X to ndc = X / (Z tan) (utils/graphics_utils.py:51-71), and ndc2pix of
    forward.cu:114-118 is  pix = ndc * W / 2 + (W - 1) / 2.  Pixel coordinates are the INTEGER indices (forward.cu:371-372).
  * the surfel is the plane  X(u, v) = p0 + u sx tu + v sy tv  with (tu, tv, n) the columns of R(q), q = (w, x, y, z)
    (forward.cu:77-103, auxiliary.h:220-242).  The ray of pixel (px, py) has direction d = ((px - (W-1)/2) / fx, (py - (H-1)/2) / fy, 1);
    it meets the plane at t = (n . p0) / (n . d); the hit point t d has view depth t and local coordinates
    u = tu . (t d - p0) / sx, v = tv . (t d - p0) / sy.  forward.cu:371-382 computes exactly this hit by intersecting the two
    planes k, l (s = (u, v), depth = s . Tw.xy + Tw.z = clip w = view z).
  * G = exp(-(u^2 + v^2) / 2) unless the screen-space low-pass 2 |mean2D - pix|^2 is smaller (forward.cu:376-381);
    alpha = min(0.99, o G), dropped below 1/255 (forward.cu:396-398); one surfel => T = 1:
        color = alpha c, others = [alpha depth, alpha, alpha n_view, depth (median, T = 1 > 0.5), 0 (distortion of one layer)]
    with n_view = n flipped to face the camera (cos = -(p0 . n), forward.cu:224-229).
  * the bounding box (forward.cu:129-159, cutoff 3) of a FRONTO-PARALLEL surfel is 3 sx fx / z0 by 3 sy fy / z0 pixels around the
    projected centre, radius = ceil(max) (forward.cu:245).
"""
import math

import numpy as np
import torch

from materialrefgs_amd.camera import make_camera
from materialrefgs_amd.synthetic import Scene

C0 = 0.28209479177387814


def frontal_camera(H, W, fovx=0.9, fovy=0.7):
    return make_camera(np.eye(3), np.zeros(3), fovx, fovy, H, W)


def quat_to_axes(q):
    """Columns of R(q) for q = (w, x, y, z), normalised (the textbook formula; the oracle's copy is pinned to the reference's
    build_rotation by tests/test_oracle.py and the golden vectors)."""
    w, x, y, z = np.asarray(q, np.float64) / np.linalg.norm(q)
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                  [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    return R[:, 0], R[:, 1], R[:, 2]


def one_surfel_scene(p0, scale, quat, opacity, rgb, S=0, feature=None):
    """Scene with a single surfel whose SH is the constant colour `rgb` (degree 0: rgb = C0 sh + 0.5, forward.cu:22-73)."""
    f32 = lambda a: torch.tensor(np.asarray(a, np.float32))
    shs = np.zeros((1, 16, 3), np.float32)
    shs[0, 0] = (np.asarray(rgb, np.float64) - 0.5) / C0
    feats = np.zeros((1, S), np.float32) if feature is None else np.asarray(feature, np.float32).reshape(1, S)
    q = np.asarray(quat, np.float64)
    return Scene(f32([p0]), f32([scale]), f32([q / np.linalg.norm(q)]), f32([[opacity]]), torch.tensor(shs), torch.tensor(feats))


def closed_form(cam, p0, scale, quat, opacity, rgb, feature=None):
    """Per-pixel expected maps of the single-surfel scene (float64).  Returns dict with alpha, depth, color[3], normal[3],
    u, v, rho2d_margin (how far the low-pass branch is from winning), centre (projected centre pixel)."""
    H, W = cam.image_height, cam.image_width
    fx = W / (2 * math.tan(cam.FoVx / 2))
    fy = H / (2 * math.tan(cam.FoVy / 2))
    p0 = np.asarray(p0, np.float64)
    tu, tv, n = quat_to_axes(quat)
    px, py = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    d = np.stack([(px - (W - 1) / 2) / fx, (py - (H - 1) / 2) / fy, np.ones_like(px)], -1)
    t = (n @ p0) / (d @ n)
    hit = t[..., None] * d - p0
    u = hit @ tu / scale[0]
    v = hit @ tv / scale[1]
    rho3d = u * u + v * v
    centre = np.array([fx * p0[0] / p0[2] + (W - 1) / 2, fy * p0[1] / p0[2] + (H - 1) / 2])
    rho2d = 2 * ((px - centre[0]) ** 2 + (py - centre[1]) ** 2)
    G = np.exp(-0.5 * np.minimum(rho3d, rho2d))
    alpha = np.minimum(0.99, opacity * G)
    alpha = np.where(alpha < 1.0 / 255.0, 0.0, alpha)
    depth = np.where(rho3d <= rho2d, t, p0[2])
    alpha = np.where(depth < 0.2, 0.0, alpha)
    n_view = n * (1.0 if -(p0 @ n) > 0 else -1.0)
    out = {"alpha": alpha, "depth": depth, "u": u, "v": v, "centre": centre, "n_view": n_view, "rho3d": rho3d, "rho2d": rho2d,
           "color": alpha[None] * np.asarray(rgb, np.float64)[:, None, None],
           "normal": alpha[None] * n_view[:, None, None]}
    if feature is not None:
        out["feature"] = alpha[None] * np.asarray(feature, np.float64)[:, None, None]
    return out


def check_against_closed_form(color, others, feature, cf, inside, tol):
    """`inside`: pixels guaranteed to lie in the surfel's tile rectangle (|u|, |v| within the 3-sigma box for a fronto-parallel surfel)."""
    m = inside
    a = cf["alpha"]
    assert m.sum() > 50 and (a[m] > 0).sum() > 20
    scale = max(a[m].max(), 1e-12)
    np.testing.assert_allclose(others[1][m], a[m], atol=tol * scale, rtol=0)
    np.testing.assert_allclose(others[0][m], (a * cf["depth"])[m], atol=tol * scale * cf["depth"][m].max(), rtol=0)
    for ch in range(3):
        np.testing.assert_allclose(color[ch][m], cf["color"][ch][m], atol=tol * scale, rtol=0)
        np.testing.assert_allclose(others[2 + ch][m], cf["normal"][ch][m], atol=tol * scale, rtol=0)
    hit = m & (a > 0)
    np.testing.assert_allclose(others[5][hit], cf["depth"][hit], rtol=tol, atol=0)          # median depth = the layer's depth
    np.testing.assert_allclose(others[6][m], 0.0, atol=tol)                                    # one layer: no distortion
    # expected depth of gaussian_renderer/__init__.py:57: allmap[0] / allmap[1] == view-space z of the hit
    np.testing.assert_allclose(others[0][hit] / others[1][hit], cf["depth"][hit], rtol=10 * tol, atol=0)
    if feature is not None and "feature" in cf:
        for ch in range(cf["feature"].shape[0]):
            np.testing.assert_allclose(feature[ch][m], cf["feature"][ch][m], atol=tol * scale, rtol=0)


FRONTAL = dict(p0=(0.31, -0.17, 2.6), scale=(0.088, 0.052), quat=(1.0, 0.0, 0.0, 0.0), opacity=0.83, rgb=(0.9, 0.4, 0.15))
SATURATED = dict(p0=(-0.2, 0.1, 1.9), scale=(0.06, 0.09), quat=(1.0, 0.0, 0.0, 0.0), opacity=1.0, rgb=(0.2, 0.7, 0.6))   # 0.99 clamp
TILTED = dict(p0=(0.12, 0.21, 3.1), scale=(0.16, 0.11), quat=(0.92, 0.25, -0.28, 0.1), opacity=0.6, rgb=(0.3, 0.8, 0.5))
SPUN = dict(p0=(-0.3, -0.1, 2.2), scale=(0.12, 0.05), quat=(0.8, 0.0, 0.0, 0.6), opacity=0.7, rgb=(0.5, 0.5, 0.9))       # in-plane spin only
