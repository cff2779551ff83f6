"""Times the surfel tracer on the C2 scene: hierarchy build, forward, backward (HIP events), for mirror rays of one 800x800 view.

python tools/trace_time.py [P] [H] -> one JSON line.  Rays: every pixel's camera ray is intersected with the unit sphere the shell
scene lives on and mirrored about the sphere's normal there (origin moved 1e-3 along the mirrored ray, as render_indirect does,
envgs_renderer.py:716-731); pixels that miss the sphere get a zero direction (traced as background)."""
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from materialrefgs_amd.surfel_tracing import SurfelTracer, SurfelTracingSettings  # noqa: E402
from materialrefgs_amd.synthetic import make_shell_scene, orbit_camera  # noqa: E402


def mirror_rays(cam, H, W, dev):
    K = torch.as_tensor(cam.HWK[2], dtype=torch.float32, device=dev)
    Wv = cam.world_view_transform.to(dev)
    c2w = Wv.T.inverse()
    ys, xs = torch.meshgrid(torch.arange(H, device=dev, dtype=torch.float32), torch.arange(W, device=dev, dtype=torch.float32), indexing="ij")
    pix = torch.stack([xs, ys, torch.ones_like(xs)], dim=-1)
    d = (pix @ torch.linalg.inv(K).T) @ c2w[:3, :3].T
    d = d / d.norm(dim=-1, keepdim=True)
    o = c2w[:3, 3].expand_as(d)
    b = (o * d).sum(-1)
    disc = b * b - ((o * o).sum(-1) - 1.0)
    hit = disc > 0
    t = -b - torch.sqrt(disc.clamp_min(0))
    x = o + t[..., None] * d
    n = x / x.norm(dim=-1, keepdim=True)
    r = d - 2 * (d * n).sum(-1, keepdim=True) * n
    ro = torch.where(hit[..., None], x + 1e-3 * r, torch.zeros_like(x))
    rd = torch.where(hit[..., None], r, torch.zeros_like(r))
    return ro.contiguous(), rd.contiguous(), hit


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 800
    mode = sys.argv[3] if len(sys.argv) > 3 else "mirror"
    dev = torch.device("cuda:0")
    sc = make_shell_scene(P, seed=0, image_size=H).to(dev)
    cam = orbit_camera(0, H, H).to(dev)
    ro, rd, hit = mirror_rays(cam, H, H, dev)
    if mode == "primary":          # camera rays through the whole shell (two crossings)
        K = torch.as_tensor(cam.HWK[2], dtype=torch.float32, device=dev)
        c2w = cam.world_view_transform.to(dev).T.inverse()
        ys, xs = torch.meshgrid(torch.arange(H, device=dev, dtype=torch.float32), torch.arange(H, device=dev, dtype=torch.float32), indexing="ij")
        rd = ((torch.stack([xs, ys, torch.ones_like(xs)], dim=-1) @ torch.linalg.inv(K).T) @ c2w[:3, :3].T).contiguous()
        ro = c2w[:3, 3].expand_as(rd).contiguous()
    means = sc.means3D.clone().requires_grad_(True)
    scales, rots, opac = sc.scales.clone().requires_grad_(True), sc.rotations.clone().requires_grad_(True), sc.opacities.clone().requires_grad_(True)
    colors = torch.rand(P, 3, device=dev).requires_grad_(True)
    others = torch.full((P, 2), 0.01, device=dev)
    eye = torch.eye(4, device=dev)
    ts = SurfelTracingSettings(H, H, 1.0, 1.0, torch.zeros(3, device=dev), 1.0, eye, eye, 0, torch.zeros(3, device=dev), False, False)
    tr = SurfelTracer()

    def quads():
        from materialrefgs_amd.gs_utils import build_rotation
        R = build_rotation(rots.detach())
        su, sv = scales.detach()[:, 0:1] * R[:, :, 0], scales.detach()[:, 1:2] * R[:, :, 1]
        m = means.detach()
        return torch.stack([m - 3 * su + 3 * sv, m - 3 * su - 3 * sv, m + 3 * su + 3 * sv, m + 3 * su - 3 * sv], dim=1).reshape(-1, 3)

    v = quads()
    ev = lambda: torch.cuda.Event(enable_timing=True)
    res = {}
    for it in range(4):
        e = [ev() for _ in range(4)]
        e[0].record()
        tr.build_acceleration_structure(v, None)
        e[1].record()
        out = tr(ro, rd, v, means3D=means, grads3D=None, shs=None, colors_precomp=colors, others_precomp=others, opacities=opac, scales=scales,
                 rotations=rots, cov3D_precomp=None, tracer_settings=ts)
        e[2].record()
        (out[0].sum() + 0.1 * out[1].sum() + 0.01 * out[4].sum()).backward()
        e[3].record()
        torch.cuda.synchronize()
        res = dict(build_ms=e[0].elapsed_time(e[1]), forward_ms=e[1].elapsed_time(e[2]), backward_ms=e[2].elapsed_time(e[3]))
    wet = out[7]
    from materialrefgs_amd.surfel_tracing import _Trace  # noqa: F401
    res.update(P=P, rays=int(ro.shape[0] * ro.shape[1]), rays_with_direction=int((rd.abs().sum(-1) > 0).sum()), mode=mode,
               mean_alpha=float(out[2].mean()), surfels_touched=int((wet > 0).sum()))
    st = tr.last_state
    live = st[:, 3] != 0
    res.update(rays_in_packets=int((st[:, 3] < 0).sum()), mean_passes=float(st[:, 3].abs()[live].mean()), mean_hits=float(st[:, 2][live].mean()),
               max_hits=int(st[:, 2].max()))
    res["Mrays_per_s_forward"] = res["rays"] / res["forward_ms"] / 1e3
    print(json.dumps(res))


main()
