"""Synthetic "shell" scene of SURVEY.md section 8d / BASELINE.md section 3 (no datasets on the GPU box).

P surfels with centres on a radius-1 sphere shell, disk normal = radial direction with a random in-plane
spin, log-normal scales sized so that the mean projected 3-sigma radius is `radius_px` pixels at
`image_size`^2 (=> about 6 tiles per surfel at 7 px / 800^2), opacity = sigmoid(N(1.5, 1)), degree-3 SH
colours, S sigmoid feature channels.  Tensors are laid out exactly as scene/gaussian_model.py getters hand
them to the rasterizer (xyz[P,3], scaling[P,2], rotation[P,4] wxyz, opacity[P,1], features[P,16,3]).
"""
import math
from typing import NamedTuple, Optional

import numpy as np
import torch

from .camera import fov2focal, look_at_camera

C0 = 0.28209479177387814
FOV = 0.6911
CAM_DISTANCE = 4.03


class Scene(NamedTuple):
    means3D: torch.Tensor      # [P,3]
    scales: torch.Tensor       # [P,2]  (activated, i.e. exp(_scaling))
    rotations: torch.Tensor    # [P,4]  (normalised, w x y z)
    opacities: torch.Tensor    # [P,1]  (activated)
    shs: torch.Tensor          # [P,16,3]
    features: torch.Tensor     # [P,S]

    def to(self, device):
        return Scene(*[t.to(device) for t in self])


def _quat_from_z_to(n, spin, rng):
    """Unit quaternion (w,x,y,z) rotating +z onto n, composed with a spin about z."""
    P = n.shape[0]
    z = np.array([0.0, 0.0, 1.0])
    axis = np.cross(np.broadcast_to(z, n.shape), n)
    s = np.linalg.norm(axis, axis=1, keepdims=True)
    c = n[:, 2:3]
    axis = np.where(s > 1e-8, axis / np.maximum(s, 1e-8), np.array([[1.0, 0.0, 0.0]]))
    ang = np.arctan2(s, c)
    q1 = np.concatenate([np.cos(ang / 2), axis * np.sin(ang / 2)], axis=1)
    q2 = np.concatenate([np.cos(spin / 2)[:, None], np.zeros((P, 2)), np.sin(spin / 2)[:, None]], axis=1)
    w1, x1, y1, z1 = q1.T
    w2, x2, y2, z2 = q2.T
    q = np.stack([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                  w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2], axis=1)
    return q / np.linalg.norm(q, axis=1, keepdims=True)


def make_shell_scene(P: int, S: int = 0, seed: int = 0, radius_px: float = 7.0, image_size: int = 800,
                     sh_degree_filled: int = 3, device: Optional[str] = None, scale_sigma: float = 0.35) -> Scene:
    rng = np.random.default_rng(seed)
    d = rng.normal(size=(P, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    r = rng.uniform(0.95, 1.05, size=(P, 1))
    xyz = d * r
    q = _quat_from_z_to(d, rng.uniform(0, 2 * math.pi, size=P), rng)
    focal = fov2focal(FOV, image_size)
    s_bar = radius_px * CAM_DISTANCE / (3.0 * focal)
    scales = np.exp(rng.normal(math.log(s_bar), scale_sigma, size=(P, 2)))   # scale_sigma: spread of the log-normal splat sizes
    opac = 1.0 / (1.0 + np.exp(-rng.normal(1.5, 1.0, size=(P, 1))))
    shs = np.zeros((P, 16, 3))
    shs[:, 0, :] = (rng.uniform(0, 1, size=(P, 3)) - 0.5) / C0
    ncoef = (sh_degree_filled + 1) ** 2
    shs[:, 1:ncoef, :] = rng.normal(0, 0.05, size=(P, ncoef - 1, 3))
    feats = 1.0 / (1.0 + np.exp(-rng.normal(0, 1, size=(P, S))))
    f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    sc = Scene(f32(xyz), f32(scales), f32(q), f32(opac), f32(shs), f32(feats))
    return sc.to(device) if device is not None else sc


def orbit_camera(view: int, height: int, width: int, n_views: int = 8, elevation: float = 30.0):
    """View `view` of the 8-azimuth orbit."""
    return look_at_camera(360.0 * (view % n_views) / n_views + 17.0, elevation, CAM_DISTANCE, FOV, height, width)


def upstream_grads(S: int, H: int, W: int, device=None):
    """Fixed upstream gradients of the benchmark: dL/dcolor = 1, dL/dfeature = 0.5, dL/dothers = 0.1 with the
    median-depth channel zeroed."""
    g_color = torch.ones(3, H, W, device=device)
    g_feat = torch.full((S, H, W), 0.5, device=device)
    g_others = torch.full((7, H, W), 0.1, device=device)
    g_others[5] = 0.0
    return g_color, g_feat, g_others


def sphere_mesh(n_lat: int, n_lon: int, radius: float = 1.0, bumps: float = 0.0, seed: int = 0, centre=(0.0, 0.0, 0.0)):
    """Outward-facing triangulated sphere (open at the poles), optional radial noise: (vertices f32 [V,3], triangles i32 [2 n_lat n_lon, 3])."""
    rng = np.random.default_rng(seed)
    th = np.linspace(0.02, np.pi - 0.02, n_lat + 1)
    ph = np.linspace(0, 2 * np.pi, n_lon, endpoint=False)
    T, Ph = np.meshgrid(th, ph, indexing="ij")
    r = radius * (1 + bumps * rng.standard_normal(T.shape))
    v = np.stack([r * np.sin(T) * np.cos(Ph), r * np.sin(T) * np.sin(Ph), r * np.cos(T)], -1).reshape(-1, 3)
    v = (v + np.asarray(centre)[None]).astype(np.float32)
    i, j = np.meshgrid(np.arange(n_lat), np.arange(n_lon), indexing="ij")
    a, b = i * n_lon + j, i * n_lon + (j + 1) % n_lon
    c, d = (i + 1) * n_lon + j, (i + 1) * n_lon + (j + 1) % n_lon
    tri = np.stack([np.stack([a, c, b], -1), np.stack([b, c, d], -1)], axis=2).reshape(-1, 3)
    return v, tri.astype(np.int32)


def make_occluder_mesh(n_triangles: int = 1_000_000, seed: int = 0):
    """Mesh for the visibility rays of the shell scene (BASELINE config 4: ~1 M triangles): a sphere just inside the surfel shell
    (60 % of the triangles) and six satellite spheres that block part of the mirror directions."""
    n_main = int(0.6 * n_triangles)
    lat = max(3, int(round(math.sqrt(n_main / 4.0))))
    vs, ts = [], []
    v, t = sphere_mesh(lat, 2 * lat, 0.93, 0.005, seed)
    vs.append(v); ts.append(t)
    lat_s = max(3, int(round(math.sqrt((n_triangles - len(t)) / 6.0 / 4.0))))
    off = len(v)
    for k, c in enumerate([(2.2, 0, 0), (-2.2, 0, 0), (0, 2.2, 0), (0, -2.2, 0), (0, 0, 2.2), (0, 0, -2.2)]):
        v, t = sphere_mesh(lat_s, 2 * lat_s, 0.8, 0.005, seed + 1 + k, centre=c)
        vs.append(v); ts.append(t + off)
        off += len(v)
    return np.concatenate(vs), np.concatenate(ts)


def make_surfel_model(P: int, image_size: int, device, seed: int = 0, radius_px: float = 7.0, env_res: int = 128, env_min: int = 16):
    """The shell scene as a renderer.SurfelModel with random material parameters and a trainable EnvLight -- what bench.py's surfel
    workloads (C3full, C3train, C3trace, C4*) and the full-size tests render.  Returns (pc, env, leaves): `leaves` are the raw parameter
    tensors + env.base, all requiring gradients."""
    from .renderer import SurfelModel
    from .shading import EnvLight
    scene = make_shell_scene(P, S=0, seed=seed, radius_px=radius_px, image_size=image_size).to(device)
    gen = torch.Generator().manual_seed(seed)
    rnd = lambda *sh: torch.randn(*sh, generator=gen).to(device)
    env = EnvLight(device=device, min_res=env_min, max_res=env_res, trainable=True)
    with torch.no_grad():
        env.base.copy_(rnd(6, env_res, env_res, 3))
    inv_sig = lambda x: torch.log(x / (1 - x))
    pc = SurfelModel(scene.means3D.clone(), torch.log(scene.scales), scene.rotations.clone(), inv_sig(scene.opacities.clamp(1e-4, 1 - 1e-4)),
                     scene.shs[:, :1].clone(), scene.shs[:, 1:].clone(), refl_strength=rnd(P, 1), roughness=rnd(P, 1), ori_color=rnd(P, 3),
                     indirect_dc=rnd(P, 1, 3) * 0.1, indirect_rest=rnd(P, 15, 3) * 0.01, envmap=env)
    leaves = pc.parameters() + [env.base]
    for t in leaves:
        t.requires_grad_(True)
    return pc, env, leaves
