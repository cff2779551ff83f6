"""What the deferred shading backward's texel scatter looks like for a rendered view of the bench scene (developer diagnostics, GPU):
mip level histogram of the pixels and, per wavefront footprint (64 x 1 pixels) and per 64 x 12 tile, the number of DISTINCT texels
the bilinear taps touch on every level.  Usage: python tools/shade_stats.py [P] [size]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import json
import numpy as np
import torch
from types import SimpleNamespace


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 800
    dev = torch.device("cuda:0")
    from materialrefgs_amd import shading
    from materialrefgs_amd.renderer import render_surfel
    from materialrefgs_amd.synthetic import make_surfel_model, orbit_camera
    pc, env, leaves = make_surfel_model(P, size, dev, seed=0)
    cams = [orbit_camera(0, size, size).to(dev)]
    cap = {}
    orig = shading._SurfelShade.forward

    def spy(ctx, base_color, features, normal_map, render_alpha, bg, srgb, lut, R, T, Kinv, lo, hi, vis, *mips):
        cap.update(normal=normal_map.detach().clone(), rough=features[1].detach().clone(), alpha=render_alpha.detach().clone(), R=R.clone(),
                   T=T.clone(), Kinv=Kinv, lo=lo, hi=hi, res=[m.shape[1] for m in mips])
        return orig(ctx, base_color, features, normal_map, render_alpha, bg, srgb, lut, R, T, Kinv, lo, hi, vis, *mips)
    shading._SurfelShade.forward = staticmethod(spy)
    pipe = SimpleNamespace(depth_ratio=0.0, debug=False, compute_cov3D_python=False, convert_SHs_python=False)
    env.build_mips()
    render_surfel(cams[0], pc, pipe, torch.zeros(3, device=dev), srgb=False, opt=SimpleNamespace(indirect=False))
    n, r, a = cap["normal"].reshape(size, size, 3), cap["rough"].reshape(size, size), cap["alpha"].reshape(size, size)
    Kinv = torch.tensor(cap["Kinv"], device=dev).reshape(3, 3)
    ys, xs = torch.meshgrid(torch.arange(size, device=dev, dtype=torch.float32), torch.arange(size, device=dev, dtype=torch.float32), indexing="ij")
    pc_ = torch.stack([xs, ys, torch.ones_like(xs)], -1) @ Kinv.T
    R, T = cap["R"].float(), cap["T"].float()
    pw = (pc_ - T) @ R.T
    ro = -(R @ T)
    rd = torch.nn.functional.normalize(pw - ro, dim=-1)
    wo = -rd
    ndv = (wo * n).sum(-1, keepdim=True)
    rn = torch.nn.functional.normalize(2 * n * ndv - wo, dim=-1)
    nl = len(cap["res"])
    lo, hi = cap["lo"], cap["hi"]
    n2 = nl - 2
    lev = torch.where(r < hi, (r.clamp(lo, hi) - lo) / (hi - lo) * n2, (r.clamp(hi, 1.0) - hi) / (1 - hi) + n2).clamp(0, nl - 1)
    l0 = lev.floor().long().clamp(max=nl - 1)
    out = {"levels_res": cap["res"], "alpha_gt0_frac": float((a > 0).float().mean()), "rough_mean": float(r.mean()),
           "level_floor_hist": torch.bincount(l0.flatten(), minlength=nl).tolist()}
    # face / texel of the (x0, y0) tap on every level
    ax = rn.abs()
    major = ax.argmax(-1)
    sgn = torch.gather(rn, -1, major[..., None])[..., 0] >= 0
    face = major * 2 + (~sgn).long()
    x, y, z = rn[..., 0], rn[..., 1], rn[..., 2]
    ma = ax.max(-1).values
    u = torch.where(major == 0, torch.where(sgn, -z, z), torch.where(major == 1, x, torch.where(sgn, x, -x))) / ma
    v = torch.where(major == 1, torch.where(sgn, z, -z), -y) / ma
    for li, res in enumerate(cap["res"]):
        tx = ((u * 0.5 + 0.5) * res - 0.5).floor().long().clamp(0, res - 1)
        ty = ((v * 0.5 + 0.5) * res - 0.5).floor().long().clamp(0, res - 1)
        key = (face * res + ty) * res + tx
        use = ((l0 == li) | (l0 + 1 == li))            # pixels whose trilinear pair includes this level
        key = torch.where(use, key, torch.full_like(key, -1))
        W64 = size // 64 * 64
        rows = key[:, :W64].reshape(size, W64 // 64, 64)
        srt = rows.sort(-1).values
        distinct = (srt[..., 1:] != srt[..., :-1]).sum(-1) + 1 - (srt[..., 0] == -1).long()     # -1 = unused lanes
        users = (rows >= 0).sum(-1)
        H12 = size // 12 * 12
        tiles = key[:H12, :W64].reshape(H12 // 12, 12, W64 // 64, 64).permute(0, 2, 1, 3).reshape(-1, 768)
        st = tiles.sort(-1).values
        dt = (st[..., 1:] != st[..., :-1]).sum(-1) + 1 - (st[..., 0] == -1).long()
        ut = (tiles >= 0).sum(-1)
        out[f"level{li}_res{res}"] = {"pixels_using": int(use.sum()), "wave_users_mean": float(users.float().mean()),
                                       "wave_distinct_x0y0_mean": float(distinct.float().mean()),
                                       "wave_distinct_p90": float(distinct.float().quantile(0.9)),
                                       "tile_users_mean": float(ut.float().mean()), "tile_distinct_mean": float(dt.float().mean()),
                                       "tile_distinct_p90": float(dt.float().quantile(0.9))}
    print(json.dumps(out, indent=1))


main()
