"""Pure-PyTorch CPU forward of the surfel rasterizer -- TEST INFRASTRUCTURE / CPU BASELINE ONLY.

BASELINE.json configs[0] ("1k synthetic gaussians, 128x128, forward-only alpha-blend via pure-PyTorch loop on CPU") and the
`north_star`'s "pure-PyTorch CPU alpha-blend baseline timed on the same box's host cores": the same algorithm as
oracle/mrgs_oracle.c (which see for the file:line citations into cuda_rasterizer/forward.cu and rasterizer_impl.cu), written with
torch tensor ops only -- no C, no custom kernels:

  preprocess   vectorised over the P surfels (forward.cu:163-266)
  binning      (tile, depth bits, emission order) keys, torch sort, tile ranges (rasterizer_impl.cu:72-140,283-324)
  blend        front to back; the k-th list entry of EVERY tile is processed in one step, vectorised over tiles x 256 pixels
               (forward.cu:272-463); the loop runs max-list-length times

Only tests/ and bench.py's cpu_baseline leg may import it.  It is checked against the C oracle in tests/test_torch_blend.py.
"""
import math

import torch

NEAR_N, FAR_N = 0.2, 100.0
SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658, 1.445305721320277,
         -0.5900435899266435)


def _quat_to_rot(q):
    q = q / q.norm(dim=1, keepdim=True)
    w, x, y, z = q.unbind(1)
    return torch.stack([torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], 1),
                        torch.stack([2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)], 1),
                        torch.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], 1)], 1)   # [P,3(row),3(col)]


def _sh_to_rgb(deg, shs, dirs):
    x, y, z = dirs.unbind(1)
    x, y, z = x[:, None], y[:, None], z[:, None]
    r = SH_C0 * shs[:, 0]
    if deg > 0:
        r = r - SH_C1 * y * shs[:, 1] + SH_C1 * z * shs[:, 2] - SH_C1 * x * shs[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        r = r + SH_C2[0] * xy * shs[:, 4] + SH_C2[1] * yz * shs[:, 5] + SH_C2[2] * (2 * zz - xx - yy) * shs[:, 6] + \
            SH_C2[3] * xz * shs[:, 7] + SH_C2[4] * (xx - yy) * shs[:, 8]
    if deg > 2:
        r = r + SH_C3[0] * y * (3 * xx - yy) * shs[:, 9] + SH_C3[1] * xy * z * shs[:, 10] + SH_C3[2] * y * (4 * zz - xx - yy) * shs[:, 11] + \
            SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * shs[:, 12] + SH_C3[4] * x * (4 * zz - xx - yy) * shs[:, 13] + \
            SH_C3[5] * z * (xx - yy) * shs[:, 14] + SH_C3[6] * x * (xx - 3 * yy) * shs[:, 15]
    return torch.clamp_min(r + 0.5, 0.0)


def preprocess(means3D, scales, rotations, opacities, shs, sh_degree, view, proj, campos, H, W, scale_modifier=1.0):
    """Per-surfel state: dict(T [P,3,3] rows Tu,Tv,Tw; mean2D [P,2]; depth [P]; normal [P,3]; rgb [P,3]; radius [P] int;
    rect [P,4] = xmin,ymin,xmax,ymax in tiles; visible [P] bool)."""
    P = means3D.shape[0]
    ones = torch.ones(P, 1, dtype=means3D.dtype)
    pv = torch.cat([means3D, ones], 1) @ view                                   # row-vector convention (auxiliary.h:80-99)
    R = _quat_to_rot(rotations)
    L0, L1, n_world = R[:, :, 0] * (scale_modifier * scales[:, :1]), R[:, :, 1] * (scale_modifier * scales[:, 1:2]), R[:, :, 2]
    zero = torch.zeros(P, 1, dtype=means3D.dtype)
    rows = torch.stack([torch.cat([L0, zero], 1), torch.cat([L1, zero], 1), torch.cat([means3D, ones], 1)], 1)   # [P,3,4] = splat2world^T
    clip = rows @ proj                                                           # [P,3,4]
    ndc2pix = torch.tensor([[W / 2, 0, 0], [0, H / 2, 0], [0, 0, 0], [(W - 1) / 2, (H - 1) / 2, 1]], dtype=means3D.dtype)
    T = (clip @ ndc2pix).transpose(1, 2)                                        # rows Tu, Tv, Tw; columns = splat axes u, v, centre
    normal = n_world @ view[:3, :3]
    cosv = -(pv[:, :3] * normal).sum(1)
    normal = normal * torch.where(cosv > 0, 1.0, -1.0)[:, None]
    t = torch.tensor([9.0, 9.0, -1.0], dtype=means3D.dtype)
    Tu, Tv, Tw = T[:, 0], T[:, 1], T[:, 2]
    dist = (Tw * Tw * t).sum(1)
    f = t[None] / dist[:, None]
    cx, cy = (f * Tu * Tw).sum(1), (f * Tv * Tw).sum(1)
    ex = torch.sqrt(torch.clamp_min(cx * cx - (f * Tu * Tu).sum(1), 1e-4))
    ey = torch.sqrt(torch.clamp_min(cy * cy - (f * Tv * Tv).sum(1), 1e-4))
    radius = torch.ceil(torch.maximum(ex, ey))
    gx, gy = (W + 15) // 16, (H + 15) // 16
    trunc = lambda v: torch.trunc(v).to(torch.int64)
    xmin = trunc((cx - radius) / 16).clamp(0, gx); ymin = trunc((cy - radius) / 16).clamp(0, gy)
    xmax = trunc((cx + radius + 15) / 16).clamp(0, gx); ymax = trunc((cy + radius + 15) / 16).clamp(0, gy)
    visible = (pv[:, 2] > 0.2) & (cosv != 0) & (dist != 0) & ((xmax - xmin) * (ymax - ymin) > 0)
    d = means3D - campos
    rgb = _sh_to_rgb(sh_degree, shs, d / d.norm(dim=1, keepdim=True))
    return {"T": T, "mean2D": torch.stack([cx, cy], 1), "depth": pv[:, 2], "normal": normal, "rgb": rgb,
            "radius": torch.where(visible, radius, torch.zeros_like(radius)).to(torch.int32),
            "rect": torch.stack([xmin, ymin, xmax, ymax], 1), "visible": visible}


def binning(st, H, W):
    """(point_list [R], ranges [tiles,2]) in the reference's order: tile-major, then depth bits, then emission order."""
    gx, gy = (W + 15) // 16, (H + 15) // 16
    vis = st["visible"].nonzero()[:, 0]
    rect = st["rect"][vis]
    w, h = rect[:, 2] - rect[:, 0], rect[:, 3] - rect[:, 1]
    n = w * h
    R = int(n.sum())
    gid = torch.repeat_interleave(vis, n)
    start = torch.cumsum(n, 0) - n
    k = torch.arange(R) - torch.repeat_interleave(start, n)
    wr = torch.repeat_interleave(w, n)
    tx = torch.repeat_interleave(rect[:, 0], n) + k % wr
    ty = torch.repeat_interleave(rect[:, 1], n) + k // wr
    tile = ty * gx + tx
    dbits = st["depth"].to(torch.float32).view(torch.int32).to(torch.int64)[gid]      # positive floats: bit order == value order
    key = (tile << 32) | dbits
    order = torch.sort(key, stable=True).indices
    point_list, tile_sorted = gid[order], tile[order]
    counts = torch.bincount(tile_sorted, minlength=gx * gy)
    ends = torch.cumsum(counts, 0)
    ranges = torch.stack([ends - counts, ends], 1)
    return point_list, ranges


def blend(st, opacities, features, point_list, ranges, H, W, bg=None):
    """color [3,H,W], feature [S,H,W], others [7,H,W], n_contrib [2,H,W] (front-to-back, all tiles in lock step)."""
    dt = st["T"].dtype
    gx, gy = (W + 15) // 16, (H + 15) // 16
    NT = gx * gy
    S = features.shape[1]
    tid = torch.arange(NT)
    ly, lx = torch.meshgrid(torch.arange(16), torch.arange(16), indexing="ij")
    px = ((tid % gx)[:, None] * 16 + lx.reshape(1, -1)).to(dt)                    # [NT,256]
    py = ((tid // gx)[:, None] * 16 + ly.reshape(1, -1)).to(dt)
    Tm = torch.ones(NT, 256, dtype=dt)
    C = torch.zeros(NT, 256, 3, dtype=dt); F = torch.zeros(NT, 256, S, dtype=dt); N = torch.zeros(NT, 256, 3, dtype=dt)
    D = torch.zeros(NT, 256, dtype=dt); M1 = torch.zeros_like(D); M2 = torch.zeros_like(D); dist = torch.zeros_like(D)
    med_d = torch.zeros_like(D)
    last = torch.zeros(NT, 256, dtype=torch.int64); med = torch.zeros_like(last)
    done = torch.zeros(NT, 256, dtype=torch.bool)
    length = ranges[:, 1] - ranges[:, 0]
    mscale = FAR_N / (FAR_N - NEAR_N)
    for k in range(int(length.max()) if NT else 0):
        act = length > k
        if not bool((act[:, None] & ~done).any()):
            break
        g = point_list[(ranges[:, 0] + k).clamp_max(max(point_list.shape[0] - 1, 0))]   # [NT]
        T9 = st["T"][g]                                                              # [NT,3,3]
        Tu, Tv, Tw = T9[:, 0, None], T9[:, 1, None], T9[:, 2, None]                  # [NT,1,3]
        kk = px[..., None] * Tw - Tu
        ll = py[..., None] * Tw - Tv
        p = torch.cross(kk, ll, dim=-1)
        ok = act[:, None] & ~done & (p[..., 2] != 0)
        pz = torch.where(p[..., 2] != 0, p[..., 2], torch.ones_like(p[..., 2]))
        sx, sy = p[..., 0] / pz, p[..., 1] / pz
        rho3d = sx * sx + sy * sy
        m2 = st["mean2D"][g]
        dx, dy = m2[:, None, 0] - px, m2[:, None, 1] - py
        rho2d = 2.0 * (dx * dx + dy * dy)
        depth = torch.where(rho3d <= rho2d, sx * Tw[..., 0] + sy * Tw[..., 1] + Tw[..., 2], Tw[..., 2].expand_as(sx))
        power = -0.5 * torch.minimum(rho3d, rho2d)
        alpha = torch.clamp_max(opacities[g][:, None] * torch.exp(power), 0.99)
        ok = ok & (depth >= NEAR_N) & (power <= 0) & (alpha >= 1.0 / 255.0)
        test_T = Tm * (1 - alpha)
        stop = ok & (test_T < 0.0001)
        done = done | stop
        ok = ok & ~stop
        w = torch.where(ok, alpha * Tm, torch.zeros_like(alpha))
        A = 1 - Tm
        m = mscale * (1 - NEAR_N / torch.where(ok, depth, torch.ones_like(depth)))
        dist = dist + (m * m * A + M2 - 2 * m * M1) * w
        D = D + depth * w
        M1 = M1 + m * w
        M2 = M2 + m * m * w
        upd = ok & (Tm > 0.5)
        med_d = torch.where(upd, depth, med_d)
        med = torch.where(upd, torch.full_like(med, k + 1), med)
        N = N + st["normal"][g][:, None, :] * w[..., None]
        C = C + st["rgb"][g][:, None, :] * w[..., None]
        if S:
            F = F + features[g][:, None, :] * w[..., None]
        Tm = torch.where(ok, test_T, Tm)
        last = torch.where(ok, torch.full_like(last, k + 1), last)

    def to_image(v):                                                                 # [NT,256,...] -> [..., H, W]
        v = v.reshape(gy, gx, 16, 16, -1).permute(4, 0, 2, 1, 3).reshape(-1, gy * 16, gx * 16)
        return v[:, :H, :W]
    bgv = torch.zeros(3, dtype=dt) if bg is None else bg.to(dt)
    color = to_image(C + Tm[..., None] * bgv)
    others = torch.cat([to_image(D), to_image(1 - Tm), to_image(N), to_image(med_d), to_image(dist)], 0)
    return color, to_image(F) if S else torch.zeros(0, H, W, dtype=dt), others, torch.cat([to_image(last), to_image(med)], 0)


def render(scene, cam, sh_degree=3, dtype=torch.float32):
    """Forward render of a materialrefgs_amd.synthetic.Scene: (color, feature, others, n_contrib, num_rendered)."""
    c = lambda t: t.to(dtype)
    H, W = cam.image_height, cam.image_width
    st = preprocess(c(scene.means3D), c(scene.scales), c(scene.rotations), c(scene.opacities), c(scene.shs), sh_degree,
                    c(cam.world_view_transform), c(cam.full_proj_transform), c(cam.camera_center), H, W)
    point_list, ranges = binning(st, H, W)
    color, feature, others, n_contrib = blend(st, c(scene.opacities)[:, 0], c(scene.features), point_list, ranges, H, W)
    return color, feature, others, n_contrib, int(point_list.shape[0])


if __name__ == "__main__":
    # bench.py's cpu_baseline_torch leg runs this file as a child process under a timeout: `python -m oracle.torch_blend <threads>`
    import json
    import os
    import sys
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from materialrefgs_amd.synthetic import make_shell_scene, orbit_camera
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    torch.set_num_threads(threads)
    runs = {}
    for name, (P_, HW_) in {"C1 (P=1000, 128x128)": (1000, 128), "C2/10 (P=30000, 256x256)": (30000, 256)}.items():
        sc_ = make_shell_scene(P_, S=0, seed=0, radius_px=7.0 * HW_ / 800.0 if P_ > 1000 else 7.0, image_size=HW_)
        t = time.perf_counter()
        r_ = render(sc_, orbit_camera(0, HW_, HW_))
        runs[name] = {"seconds": round(time.perf_counter() - t, 3), "num_rendered": r_[-1]}
    print(json.dumps({"threads": torch.get_num_threads(), "runs": runs}))
