"""Camera matrices in the reference's conventions.

Mirrors /root/reference/utils/graphics_utils.py:37-71 (getWorld2View2, getProjectionMatrix) and
/root/reference/scene/cameras.py:65-86 (Camera: world_view_transform = W2V^T, full_proj_transform =
world_view_transform @ projection^T, camera_center = inverse(world_view_transform)[3, :3]).  The
rasterizer consumes these row-vector-convention tensors as-is (auxiliary.h:80-99 reads them column-major).
"""
import math
from typing import NamedTuple

import numpy as np
import torch


def get_world2view2(R, t, translate=(0.0, 0.0, 0.0), scale=1.0):
    """World-to-view matrix of a camera given as (camera-to-world rotation R, world-to-camera translation t), with the optional
    recentring / rescaling of the camera position the dataset readers use (same arithmetic as graphics_utils.py:37-49, so that the
    float32 result is the reference's: two 4x4 inversions in float64, then the cast)."""
    w2c = np.eye(4)
    w2c[:3, :3] = np.asarray(R).T
    w2c[:3, 3] = t
    c2w = np.linalg.inv(w2c)
    c2w[:3, 3] = (c2w[:3, 3] + np.asarray(translate, dtype=np.float64)) * scale
    return np.linalg.inv(c2w).astype(np.float32)


def get_projection_matrix(znear, zfar, fovX, fovY):
    """OpenGL-style perspective matrix of a symmetric frustum with z pointing forward (graphics_utils.py:51-71): depth is mapped to
    [0, 1] by rows 2 and 3, x / y by the frustum half-extents at the near plane."""
    half_w, half_h = math.tan(0.5 * fovX) * znear, math.tan(0.5 * fovY) * znear
    x0, x1, y0, y1 = -half_w, half_w, -half_h, half_h                    # left, right, bottom, top
    rows = [[2.0 * znear / (x1 - x0), 0.0, (x1 + x0) / (x1 - x0), 0.0],
            [0.0, 2.0 * znear / (y1 - y0), (y1 + y0) / (y1 - y0), 0.0],
            [0.0, 0.0, zfar / (zfar - znear), -(zfar * znear) / (zfar - znear)],
            [0.0, 0.0, 1.0, 0.0]]
    return torch.tensor(rows, dtype=torch.float32)


def fov2focal(fov, pixels):
    """Focal length in pixels of a pinhole with field of view `fov` across `pixels` (graphics_utils.py:73-74)."""
    return 0.5 * pixels / math.tan(0.5 * fov)


class MiniCam(NamedTuple):
    """The subset of scene/cameras.py:Camera that the render functions read."""
    image_height: int
    image_width: int
    FoVx: float
    FoVy: float
    znear: float
    zfar: float
    world_view_transform: torch.Tensor   # [4,4], W2V transposed
    full_proj_transform: torch.Tensor    # [4,4]
    camera_center: torch.Tensor          # [3]
    R: torch.Tensor                      # [3,3] c2w rotation (as stored by Camera.R)
    T: torch.Tensor                      # [3]  w2c translation

    @property
    def HWK(self):
        """(H, W, K) as the dataset readers hand it to Camera (scene/cameras.py:43-45): pinhole intrinsics with the
        principal point at the image centre (Cx = 0.5 W, Cy = 0.5 H, cameras.py:63-66)."""
        fx, fy = fov2focal(self.FoVx, self.image_width), fov2focal(self.FoVy, self.image_height)
        K = np.array([[fx, 0.0, 0.5 * self.image_width], [0.0, fy, 0.5 * self.image_height], [0.0, 0.0, 1.0]], dtype=np.float32)
        return (self.image_height, self.image_width, K)

    def to(self, device):
        return self._replace(world_view_transform=self.world_view_transform.to(device),
                             full_proj_transform=self.full_proj_transform.to(device),
                             camera_center=self.camera_center.to(device), R=self.R.to(device), T=self.T.to(device))


def make_camera(R, T, FoVx, FoVy, height, width, znear=0.01, zfar=100.0):
    """scene/cameras.py:70-86 without the image plumbing."""
    wvt = torch.tensor(get_world2view2(np.asarray(R, dtype=np.float64), np.asarray(T, dtype=np.float64))).transpose(0, 1).contiguous()
    proj = get_projection_matrix(znear=znear, zfar=zfar, fovX=FoVx, fovY=FoVy).transpose(0, 1)
    full = (wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0).contiguous()
    center = wvt.inverse()[3, :3].contiguous()
    return MiniCam(int(height), int(width), float(FoVx), float(FoVy), znear, zfar, wvt, full, center,
                   torch.tensor(np.asarray(R), dtype=torch.float32), torch.tensor(np.asarray(T), dtype=torch.float32))


def look_at_camera(azimuth_deg, elevation_deg, distance, FoV, height, width, target=(0.0, 0.0, 0.0)):
    """NeRF-synthetic style orbit camera: +z forward, +x right, +y down in view space."""
    az, el = math.radians(azimuth_deg), math.radians(elevation_deg)
    tgt = np.asarray(target, dtype=np.float64)
    eye = tgt + distance * np.array([math.cos(el) * math.cos(az), math.cos(el) * math.sin(az), math.sin(el)])
    fwd = tgt - eye
    fwd /= np.linalg.norm(fwd)
    up = np.array([0.0, 0.0, 1.0])
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    R_c2w = np.stack([right, down, fwd], axis=1)   # columns = camera axes in world space
    T = -R_c2w.T @ eye                             # world-to-camera translation
    return make_camera(R_c2w, T, FoV, FoV, height, width)
