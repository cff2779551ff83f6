"""Plain-PyTorch fp32 CPU restatement of the shading half of the hot path -- TEST INFRASTRUCTURE ONLY.

Restates, with torch.autograd supplying the reference gradients:
  * EnvLight.get_mip / EnvLight.__call__            (scene/light.py:88-129)
  * sample_camera_rays, reflection, get_specular_color_surfel without visibility tracing
                                                     (utils/refl_utils.py:54-73,95-98,364-419)
PARITY UNPINNED for the texture lookups: the reference performs them with nvdiffrast's `dr.texture`, which is not vendored
(requirements.txt:57 pins a local path) and cannot be imported here, and the reference has no test for this path.  The
sampling rules are therefore restated from nvdiffrast's documented behaviour (texel centres at (i+0.5)/res, bilinear,
`clamp` boundary for the 2D LUT, seamless cube edges, trilinear between integer mip levels at LOD = mip_level_bias); the
cube face/orientation convention IS pinned in-tree by cube_to_dir (scene/light_utils.py:24-31) and checked by the
known-answer test "a lookup at cube_to_dir(texel centre) returns that texel" (tests/test_shading.py).
Everything else (ray set-up, mirror direction, Fresnel-style weight, compositing) is elementwise arithmetic restated literally.
"""
import numpy as np
import torch

EPS_EDGE = 1.0 / 4096.0


def cube_to_dir(s, x, y):
    """scene/light_utils.py:24-31."""
    one = torch.ones_like(x)
    if s == 0: return torch.stack((one, -y, -x), -1)
    if s == 1: return torch.stack((-one, -y, x), -1)
    if s == 2: return torch.stack((x, one, y), -1)
    if s == 3: return torch.stack((x, -one, -y), -1)
    if s == 4: return torch.stack((x, -y, one), -1)
    return torch.stack((-x, -y, -one), -1)


def dir_to_face_uv(d):
    """Inverse of cube_to_dir: direction [N,3] -> face [N] (long), u, v in [-1,1] (differentiable w.r.t. d)."""
    ax, ay, az = d[:, 0].abs(), d[:, 1].abs(), d[:, 2].abs()
    is_x = (ax >= ay) & (ax >= az)
    is_y = (~is_x) & (ay >= az)
    is_z = ~(is_x | is_y)
    pos = torch.where(is_x, d[:, 0] >= 0, torch.where(is_y, d[:, 1] >= 0, d[:, 2] >= 0))
    face = torch.where(is_x, 0, torch.where(is_y, 2, 4)) + (~pos).long()
    ma = torch.where(is_x, ax, torch.where(is_y, ay, az))
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    sgn = torch.where(pos, 1.0, -1.0)
    u = torch.where(is_x, -sgn * z, torch.where(is_y, x, sgn * x)) / ma
    v = torch.where(is_x, -y, torch.where(is_y, sgn * z, -y)) / ma
    return face, u, v


def _wrap(face, x, y, res):
    """Integer tap (face, x, y) possibly off the face -> linear texel index on the face across the edge."""
    inb = (x >= 0) & (x < res) & (y >= 0) & (y < res)
    u = (x.float() + 0.5) / res * 2 - 1
    v = (y.float() + 0.5) / res * 2 - 1
    u = torch.where(x < 0, torch.full_like(u, -1 - EPS_EDGE), torch.where(x >= res, torch.full_like(u, 1 + EPS_EDGE), u))
    v = torch.where(y < 0, torch.full_like(v, -1 - EPS_EDGE), torch.where(y >= res, torch.full_like(v, 1 + EPS_EDGE), v))
    d = torch.zeros(u.shape[0], 3)
    for s in range(6):
        m = face == s
        if m.any():
            d[m] = cube_to_dir(s, u[m], v[m])
    f2, u2, v2 = dir_to_face_uv(d)
    xi = torch.clamp(torch.floor((u2 * 0.5 + 0.5) * res).long(), 0, res - 1)
    yi = torch.clamp(torch.floor((v2 * 0.5 + 0.5) * res).long(), 0, res - 1)
    wrapped = (f2 * res + yi) * res + xi
    direct = (face * res + y.clamp(0, res - 1)) * res + x.clamp(0, res - 1)
    return torch.where(inb, direct, wrapped)


def cube_fetch(tex, dirs):
    """Seamless bilinear fetch of a [6,res,res,C] cubemap at directions [N,3] -> [N,C]."""
    res = tex.shape[1]
    face, u, v = dir_to_face_uv(dirs)
    fx = (u * 0.5 + 0.5) * res - 0.5
    fy = (v * 0.5 + 0.5) * res - 0.5
    x0f, y0f = torch.floor(fx).detach(), torch.floor(fy).detach()
    wx, wy = fx - x0f, fy - y0f
    x0, y0 = x0f.long(), y0f.long()
    w = [(1 - wx) * (1 - wy), wx * (1 - wy), (1 - wx) * wy, wx * wy]
    offs = [(0, 0), (1, 0), (0, 1), (1, 1)]
    ox = [x0 < 0, x0 + 1 >= res, x0 < 0, x0 + 1 >= res]
    oy = [y0 < 0, y0 < 0, y0 + 1 >= res, y0 + 1 >= res]
    corner = [ox[k] & oy[k] for k in range(4)]
    w = [torch.where(corner[k], torch.zeros_like(w[k]), w[k]) for k in range(4)]
    any_corner = corner[0] | corner[1] | corner[2] | corner[3]
    norm = torch.where(any_corner, 1.0 / (w[0] + w[1] + w[2] + w[3]).detach(), torch.ones_like(wx))   # renormalisation held constant
    flat = tex.reshape(-1, tex.shape[-1])
    out = 0
    for k in range(4):
        idx = _wrap(face, x0 + offs[k][0], y0 + offs[k][1], res)
        idx = torch.where(corner[k], torch.zeros_like(idx), idx)
        out = out + (w[k] * norm).unsqueeze(-1) * flat[idx]
    return out


def get_mip(roughness, n_levels, min_roughness=0.08, max_roughness=0.5):
    """scene/light.py:88-96 (n_levels = len(self.specular))."""
    return torch.where(
        roughness < max_roughness,
        (torch.clamp(roughness, min_roughness, max_roughness) - min_roughness) / (max_roughness - min_roughness) * (n_levels - 2),
        (torch.clamp(roughness, max_roughness, 1.0) - max_roughness) / (1.0 - max_roughness) + n_levels - 2)


def env_lookup(mips, dirs, roughness=None, min_roughness=0.08, max_roughness=0.5):
    """EnvLight.__call__ (scene/light.py:99-129): mips = list of [6,res,res,3] (pre-sigmoid), dirs [N,3], roughness [N] or None."""
    if roughness is None:
        return torch.sigmoid(cube_fetch(mips[0], dirs))
    n = len(mips)
    level = get_mip(roughness, n, min_roughness, max_roughness)
    lc = torch.clamp(level, 0, n - 1)
    l0 = torch.clamp(torch.floor(lc).detach().long(), max=n - 1)
    l1 = torch.clamp(l0 + 1, max=n - 1)
    f = lc - l0.float()
    samples = torch.stack([cube_fetch(m, dirs) for m in mips], 0)   # [n,N,3]
    ar = torch.arange(dirs.shape[0])
    val = (1 - f).unsqueeze(-1) * samples[l0, ar] + f.unsqueeze(-1) * samples[l1, ar]
    return torch.sigmoid(val)


def lut_fetch(lut, uv):
    """dr.texture(FG_LUT, uv, filter_mode='linear', boundary_mode='clamp'): lut [R,R,2], uv [N,2] in [0,1] (u -> width)."""
    R = lut.shape[0]
    fx, fy = uv[:, 0] * R - 0.5, uv[:, 1] * R - 0.5
    x0f, y0f = torch.floor(fx).detach(), torch.floor(fy).detach()
    wx, wy = (fx - x0f).unsqueeze(-1), (fy - y0f).unsqueeze(-1)
    x0, x1 = x0f.long().clamp(0, R - 1), (x0f.long() + 1).clamp(0, R - 1)
    y0, y1 = y0f.long().clamp(0, R - 1), (y0f.long() + 1).clamp(0, R - 1)
    return (1 - wy) * ((1 - wx) * lut[y0, x0] + wx * lut[y0, x1]) + wy * ((1 - wx) * lut[y1, x0] + wx * lut[y1, x1])


def sample_camera_rays(H, W, K, R, T):
    """utils/refl_utils.py:54-73 (R is Camera.R, stored transposed; pixel centres at integer coordinates)."""
    R = R.T
    i, j = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing="xy")
    xy1 = np.stack([i, j, np.ones_like(i)], axis=2)
    pixel_camera = torch.tensor(np.dot(xy1, np.linalg.inv(K.astype(np.float32)).T))
    rays_o = (-R.T @ T.unsqueeze(-1)).flatten()
    pixel_world = (pixel_camera - T[None, None]).reshape(-1, 3) @ R
    rays_d = pixel_world - rays_o[None]
    rays_d = rays_d / torch.norm(rays_d, dim=1, keepdim=True)
    return rays_d.reshape(H, W, 3), rays_o


def specular_color_surfel(mips, lut, albedo, H, W, K, R, T, normal_map, render_alpha, refl_strength, roughness,
                          min_roughness=0.08, max_roughness=0.5):
    """get_specular_color_surfel (utils/refl_utils.py:364-419) with pc.ray_tracer = None.
    albedo/normal_map [H,W,3], render_alpha/refl_strength/roughness [H,W,1] -> specular [3,H,W], direct_light [3,H,W],
    specular_weight [H,W,3]."""
    rays_cam, _ = sample_camera_rays(H, W, K, R, T)
    w_o = -rays_cam
    NdotV = torch.sum(w_o * normal_map, dim=-1, keepdim=True)
    rays_refl = 2 * normal_map * NdotV - w_o
    rays_refl = rays_refl / torch.clamp(torch.linalg.norm(rays_refl, dim=-1, keepdim=True), min=1e-20)
    fg_uv = torch.cat([NdotV, roughness], -1).clamp(0, 1)
    fg = lut_fetch(lut, fg_uv.reshape(-1, 2)).reshape(H, W, 2)
    direct_light = env_lookup(mips, rays_refl.reshape(-1, 3), roughness.reshape(-1), min_roughness, max_roughness).reshape(H, W, 3)
    specular_weight = (0.04 * (1 - refl_strength) + albedo * refl_strength) * fg[..., 0:1] + fg[..., 1:2]
    specular = direct_light * render_alpha * specular_weight
    return specular.permute(2, 0, 1), direct_light.permute(2, 0, 1), specular_weight
