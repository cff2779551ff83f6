"""Developer diagnostic (GPU box): for cases of the soak sequence (tools/stress_parity.py numbering) find the pixels where the HIP
forward and the oracle disagree and print, for each, the list entries whose decisions sit near a threshold.

    python tools/diag_pixel.py <n_cases> <seed> <i,j,k>
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import HipRender  # noqa: E402
from materialrefgs_amd.synthetic import make_shell_scene, orbit_camera  # noqa: E402
from oracle import raster_oracle as ro  # noqa: E402

f32 = np.float32


def fma(a, b, c):
    return f32(np.float64(a) * np.float64(b) + np.float64(c))


def pixel_trace(orc, px, py):
    """The oracle's loop for one pixel, float32 with fused multiply-adds emulated through float64."""
    W = orc.W
    tile = (py // 16) * ((W + 15) // 16) + px // 16
    r0, r1 = orc.ranges[tile]
    Ts, xy, no = orc.transMat, orc.means2D, orc.normal_opacity
    pl = orc.point_list
    T = f32(1.0)
    rows = []
    fx, fy = f32(px), f32(py)
    for n, i in enumerate(range(r0, r1)):
        g = pl[i]
        t = Ts[g]
        Tu, Tv, Tw = t[0:3], t[3:6], t[6:9]
        k = [fma(fx, Tw[c], -Tu[c]) for c in range(3)]
        l = [fma(fy, Tw[c], -Tv[c]) for c in range(3)]
        p = [fma(k[1], l[2], -(k[2] * l[1])), fma(k[2], l[0], -(k[0] * l[2])), fma(k[0], l[1], -(k[1] * l[0]))]
        if p[2] == 0:
            continue
        inv = f32(1.0) / p[2]
        sx, sy = f32(p[0] * inv), f32(p[1] * inv)
        rho3 = fma(sx, sx, f32(sy * sy))
        dx, dy = f32(xy[g][0] - fx), f32(xy[g][1] - fy)
        rho2 = f32(2.0) * fma(dx, dx, f32(dy * dy))
        rho = min(rho3, rho2)
        depth = fma(sx, Tw[0], fma(sy, Tw[1], Tw[2])) if rho3 <= rho2 else Tw[2]
        power = f32(-0.5) * rho
        G = f32(np.exp(np.float64(power)))
        alpha = min(f32(0.99), f32(no[g][3] * G))
        hit = depth >= f32(0.2) and power <= 0 and alpha >= f32(1.0 / 255.0)
        test_T = f32(T * f32(f32(1.0) - alpha))
        rows.append((n + 1, g, float(alpha), float(depth), float(rho3), float(rho2), float(T), float(test_T), hit))
        if not hit:
            continue
        if test_T < f32(0.0001):
            break
        T = test_T
    return rows


def main():
    n, seed = int(sys.argv[1]), int(sys.argv[2])
    only = set(int(v) for v in sys.argv[3].split(","))
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    for i in range(n):
        P = int(rng.choice([1, 7, 63, 64, 65, 500, 3000, 12000, 40000]))
        S = int(rng.choice([0, 1, 3, 4, 8, 11, 12, 24]))
        H, W = int(rng.integers(17, 420)), int(rng.integers(17, 420))
        deg = int(rng.integers(0, 4))
        rpx = float(rng.choice([1.5, 4.0, 7.0, 15.0, 40.0]))
        view = int(rng.integers(0, 8))
        scene_seed = int(rng.integers(1 << 30))
        if i not in only:
            continue
        scene = make_shell_scene(P, S=S, seed=scene_seed, radius_px=rpx, image_size=max(H, W))
        cam = orbit_camera(view, H, W)
        orc = ro.render_scene(scene, cam, sh_degree=deg)
        hr = HipRender(scene, cam, dev, sh_degree=deg)
        nc_h, nc_o = hr.export("n_contrib").astype(np.uint32), orc.n_contrib
        col_h, col_o = hr.color.detach().cpu().numpy(), orc.color
        dcol = np.abs(col_h - col_o).max(0) / max(np.abs(col_o).max(), 1e-30)
        bad = np.argwhere((dcol > 2e-5) | (nc_h != nc_o).any(0))
        print(f"[{i}] P={P} {H}x{W}: {len(bad)} differing pixels, max|color| {np.abs(col_o).max():.3f}")
        for (py, px) in bad[:6]:
            print(f"  pixel ({px},{py}): colour diff {dcol[py, px]:.3e}, n_contrib hip {nc_h[:, py, px]} oracle {nc_o[:, py, px]}")
            rows = pixel_trace(orc, int(px), int(py))
            a0 = 1.0 / 255.0
            d = col_h[:, py, px] - col_o[:, py, px]
            print(f"    hip - oracle colour: {d}, final_T hip {hr.export('final_T')[0, py, px]:.9g} oracle {orc.final_T[0, py, px]:.9g}")
            # the same pixel in the oracle's other two readings of the reference's arithmetic (tests/test_truth_leg.py): the literal fp32 one
            # (no fused multiply-adds) and float64
            for v in ("lit32", "f64"):
                o2 = ro.render_scene(scene, cam, sh_degree=deg, variant=v)
                print(f"    oracle variant {v}: final_T {o2.final_T[0, py, px]:.9g} n_contrib {o2.n_contrib[:, py, px]} colour {o2.color[:, py, px]}")
                o2.close()
            print(f"    oracle (default) colour {col_o[:, py, px]} hip colour {col_h[:, py, px]}")
            rgb = orc.rgb
            for (c, g, alpha, depth, r3, r2, T, tT, hit) in rows:
                w = alpha * T
                contrib = w * rgb[g]
                if np.abs(np.abs(contrib) - np.abs(d)).max() < 0.2 * np.abs(d).max() + 1e-7:
                    print(f"    candidate entry {c} surfel {g}: alpha {alpha:.9g} (x255 = {alpha * 255:.7f}) depth {depth:.6g} rho3 {r3:.6g} rho2 {r2:.6g} T {T:.9g} hit {hit} w*rgb {contrib}")
            for (c, g, alpha, depth, r3, r2, T, tT, hit) in rows:
                tags = []
                if abs(alpha - a0) < 1e-4 * a0:
                    tags.append(f"alpha/a0-1={alpha / a0 - 1:+.2e}")
                if abs(depth - 0.2) < 1e-3:
                    tags.append(f"depth-0.2={depth - 0.2:+.2e}")
                if abs(r3 - r2) < 1e-4 * max(r2, 1e-30) and alpha > a0 * 0.5:
                    tags.append(f"rho3-rho2={r3 - r2:+.2e}")
                if hit and abs(tT - 1e-4) < 1e-3 * 1e-4:
                    tags.append(f"testT/1e-4-1={tT / 1e-4 - 1:+.2e}")
                if hit and abs(T - 0.5) < 1e-4:
                    tags.append(f"T-0.5={T - 0.5:+.2e}")
                if tags:
                    print(f"    entry {c} surfel {g} alpha {alpha:.9g} depth {depth:.6g} rho3 {r3:.6g} rho2 {r2:.6g} T {T:.9g} hit {hit}: " + ", ".join(tags))
        orc.close()


if __name__ == "__main__":
    main()
