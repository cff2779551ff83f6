"""Developer diagnostic: HIP rasterizer vs CPU oracle on a few scenes, printing every comparison (run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from materialrefgs_amd.synthetic import make_shell_scene, orbit_camera, upstream_grads
from oracle import raster_oracle as ro
from helpers import HipRender, rel_err

dev = torch.device("cuda:0")
cases = [(64, 0, 64, 64, 10.0), (1000, 8, 128, 128, 4.0), (2000, 3, 200, 136, 6.0), (20000, 8, 400, 400, 7.0)]
if len(sys.argv) > 1:
    cases = [tuple(float(x) if i == 4 else int(x) for i, x in enumerate(a.split(","))) for a in sys.argv[1:]]
for (P, S, H, W, rpx) in cases:
    print(f"=== P={P} S={S} {H}x{W} radius_px={rpx}")
    sc = make_shell_scene(P, S=S, seed=P, radius_px=rpx, image_size=max(H, W))
    cam = orbit_camera(1, H, W)
    t = time.time(); orc = ro.render_scene(sc, cam); t_or = time.time() - t
    t = time.time(); hr = HipRender(sc, cam, dev); torch.cuda.synchronize(); t_hip = time.time() - t
    print(f"oracle fwd {t_or:.3f}s hip fwd(first) {t_hip:.3f}s  R oracle {orc.R} hip {hr.num_rendered}")
    vis = orc.radii > 0
    print("radii equal:", np.array_equal(orc.radii, hr.radii.cpu().numpy()), "visible", int(vis.sum()))
    for name in ["depths", "means2D", "transMat", "normal_opacity", "rgb", "tiles_touched", "clamped"]:
        a = hr.export(name)[vis]; b = getattr(orc, name)[vis]
        if a.dtype != b.dtype: a = a.astype(b.dtype) if a.dtype.kind != 'f' else a
        eq = np.array_equal(a.view(np.uint8) if a.dtype.kind == 'f' else a.astype(np.int64), b.view(np.uint8) if b.dtype.kind == 'f' else b.astype(np.int64))
        md = float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max()) if a.size else 0.0
        print(f"  {name:15s} bit-exact {eq}  maxdiff {md:.3e}")
    if orc.R == hr.num_rendered:
        print("  point_list equal:", np.array_equal(hr.export("point_list").astype(np.uint32), orc.point_list))
        print("  ranges equal:", np.array_equal(hr.export("ranges").astype(np.uint32), orc.ranges))
    nc_h = hr.export("n_contrib").astype(np.uint32); nc_o = orc.n_contrib
    print("  n_contrib mismatches:", int((nc_h != nc_o).sum()), "of", nc_o.size)
    for name, a, b in [("color", hr.color, orc.color), ("feature", hr.feature, orc.feature), ("others", hr.others, orc.others)]:
        a = a.detach().cpu().numpy()
        if a.size == 0: continue
        d = np.abs(a.astype(np.float64) - b)
        print(f"  {name:8s} max abs {d.max():.3e} rel(max-norm) {rel_err(a, b):.3e}  n>1e-4: {int((d > 1e-4).sum())}")
    for ch in range(7):
        a = hr.others[ch].detach().cpu().numpy(); b = orc.others[ch]
        print(f"    others[{ch}] rel {rel_err(a, b):.3e}")
    g = upstream_grads(S, H, W)
    t = time.time(); go = orc.backward(*g); t_ob = time.time() - t
    gh = hr.backward(*g)
    print(f"  oracle bwd {t_ob:.3f}s")
    names = {"means3D": "means3D", "means2D": "means2D", "opacity": "opacity", "scales": "scales", "rotations": "rotations",
             "features": "features", "sh": "sh"}
    for k, ko in names.items():
        if k in gh:
            print(f"  grad {k:10s} rel {rel_err(gh[k].reshape(go[ko].shape), go[ko]):.3e}  max|ref| {np.abs(go[ko]).max():.3e}")
