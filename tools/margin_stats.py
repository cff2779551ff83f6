"""How far the fast transmittance is from the exact one, and whether the marked-pixel mechanism leaves any decision to chance
(developer tool for the GPU box; mrgs_blend_math.h "Exact decisions").

    python tools/margin_stats.py dump <out.npz> [n_scenes] [seed]     # render the soak's scenes with the library MRGS_LIB names
    python tools/margin_stats.py cmp <fast.npz> <exact.npz>           # fast = the shipped build, exact = built with -DMRGS_FWD_REDO_ALL

`cmp` prints, over all pixels of all scenes: pixels whose contributor counters differ (must be 0), the distribution of
|T_fast - T_exact| / T_exact, split by the pixel's final T (the 0.5 and 1e-4 tests are taken at T ~ 0.5 and T ~ 1e-4 ... 1e-2).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def dump(out, n, seed):
    from helpers import HipRender
    from materialrefgs_amd.synthetic import make_shell_scene, orbit_camera
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    res = {}
    for i in range(n):
        P = int(rng.choice([500, 3000, 12000, 40000]))
        S = int(rng.choice([0, 8]))
        H, W = int(rng.integers(100, 420)), int(rng.integers(100, 420))
        rpx = float(rng.choice([4.0, 7.0, 15.0, 40.0]))
        view = int(rng.integers(0, 8))
        scene = make_shell_scene(P, S=S, seed=int(rng.integers(1 << 30)), radius_px=rpx, image_size=max(H, W))
        hr = HipRender(scene, orbit_camera(view, H, W), dev)
        res[f"T{i}"] = hr.export("final_T")[0]
        res[f"n{i}"] = hr.export("n_contrib")
        res[f"c{i}"] = hr.color.detach().cpu().numpy()
    np.savez_compressed(out, **res)
    print("wrote", out, n, "scenes")


def cmp(fa, fb):
    a, b = np.load(fa), np.load(fb)
    n = len([k for k in a.files if k.startswith("T")])
    rel, Tall, bad = [], [], 0
    for i in range(n):
        Ta, Tb = a[f"T{i}"].astype(np.float64).ravel(), b[f"T{i}"].astype(np.float64).ravel()
        na, nb = a[f"n{i}"].reshape(2, -1), b[f"n{i}"].reshape(2, -1)
        same = (na == nb).all(0)
        bad += int((~same).sum())
        m = same & (Tb < 1.0)
        rel.append(np.abs(Ta[m] - Tb[m]) / Tb[m])
        Tall.append(Tb[m])
        dc = np.abs(a[f"c{i}"].astype(np.float64) - b[f"c{i}"]).max()
        if dc > 1e-5:
            print(f"scene {i}: colour differs by {dc:.2e}")
    rel, Tall = np.concatenate(rel), np.concatenate(Tall)
    print(f"{n} scenes, {rel.size} pixels with T < 1; pixels with different contributor counters: {bad}")
    for lo, hi in ((0.25, 1.0), (1e-2, 0.25), (1e-3, 1e-2), (0.0, 1e-3)):
        m = (Tall >= lo) & (Tall < hi)
        if m.any():
            r = rel[m]
            print(f"  final T in [{lo:g}, {hi:g}): {int(m.sum()):9d} px  |dT|/T median {np.median(r):.2e}  99% {np.quantile(r, 0.99):.2e}  "
                  f"99.99% {np.quantile(r, 0.9999):.2e}  max {r.max():.2e}")


if __name__ == "__main__":
    if sys.argv[1] == "dump":
        dump(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 40, int(sys.argv[4]) if len(sys.argv) > 4 else 0)
    else:
        cmp(sys.argv[2], sys.argv[3])
