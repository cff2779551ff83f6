"""Developer tool (GPU box): how many pixels the forward blend marks for the exact redo at BASELINE's sizes, and where in their lists."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import HipRender
from materialrefgs_amd.synthetic import make_shell_scene, orbit_camera
dev = torch.device("cuda:0")
for (P, S, H, W) in ((300000, 0, 800, 800), (300000, 8, 800, 800), (20000, 8, 400, 400)):
    scene = make_shell_scene(P, S=S, seed=0, radius_px=7.0 * max(H, W) / 800 if P == 20000 else 7.0, image_size=max(H, W))
    for view in (0, 3):
        hr = HipRender(scene, orbit_camera(view, H, W), dev)
        rl = hr.export("redo_list")
        n = int(rl[0])
        pix = rl[2:2 + n]
        nc = hr.export("n_contrib")[0].ravel()
        print(f"P={P} S={S} {H}x{W} view {view}: R={hr.num_rendered} marked pixels {n} ({n / (H * W):.2e} of the image), last contributor of the marked: "
              f"median {np.median(nc[pix]) if n else 0:.0f} max {nc[pix].max() if n else 0}")
