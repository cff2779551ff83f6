"""Developer check (GPU): the backward blend's contracted recurrence (q - A instead of the reference's per-channel (c - accum_rec) d,
DESIGN.md section 4) under a DEPTH-dominated upstream gradient with the camera far from the scene -- the case in which q and A each
carry depth x dL/ddepth while their difference is of the order of the gap between two surfels.  Prints, per tensor, the distance of the
HIP gradient and of the literal fp32 reading of the reference from the float64 evaluation (oracle variants `lit32`, `f64`)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import HipRender, rel_err
from materialrefgs_amd.synthetic import make_shell_scene, look_at_camera, CAM_DISTANCE, FOV
from oracle import raster_oracle as ro

dev = torch.device("cuda", 0)
P, H, W = 20000, 256, 256
for k in (1.0, 10.0, 50.0):
    scene = make_shell_scene(P, S=0, seed=3, radius_px=7.0, image_size=H)
    scene = scene._replace(means3D=scene.means3D * k, scales=scene.scales * k)
    cam = look_at_camera(17.0, 30.0, CAM_DISTANCE * k, FOV, H, W)
    g_color = torch.full((3, H, W), 0.01)
    g_feat = torch.zeros((0, H, W))
    g_others = torch.zeros((7, H, W)); g_others[0] = 1.0; g_others[1] = 0.01
    hr = HipRender(scene, cam, dev)
    gh = hr.backward(g_color, g_feat, g_others)
    legs = {}
    for v in ("lit32", "f64", None):
        o = ro.render_scene(scene, cam, sh_degree=3, variant=v)
        legs[v] = o.backward(g_color, g_feat, g_others)
        o.close()
    print(f"scale {k:g} (depths ~ {CAM_DISTANCE * k:.0f}):")
    for name in ("means3D", "opacity", "scales", "rotations", "sh"):
        t = legs["f64"][name]
        e_hip, e_lit, e_fused = rel_err(gh[name].reshape(t.shape), t), rel_err(legs["lit32"][name], t), rel_err(legs[None][name], t)
        print(f"   {name:10s} HIP {e_hip:.2e}   literal fp32 {e_lit:.2e}   fused fp32 oracle {e_fused:.2e}   (from float64)")
