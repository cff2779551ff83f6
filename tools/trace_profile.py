"""Developer tool (GPU): what a packet walk of the surfel tracer consists of.  Needs a library whose mrgs_surfel_trace.hip was compiled
with -DST_PROFILE (the per-ray `state` then carries the wave's counters instead of its usual contents; outputs of the backward are wrong):
   tools/trace_profile.sh builds it into tools/scratch/libmrgs_stprof.so and runs this with MRGS_LIB pointing there.
Prints, over the rays that walked in packets of the first pass structure: hierarchy visits, candidate surfels tested one by one, lanes
that hit per candidate, and the wave's wall time per candidate."""
import sys, os, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "mirror"
    from materialrefgs_amd.surfel_tracing import SurfelTracer, SurfelTracingSettings
    from materialrefgs_amd.synthetic import make_shell_scene, orbit_camera
    src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "trace_time.py")).read().replace("\nmain()\n", "\n")
    ns = {"__name__": "trace_time_lib", "__file__": os.path.join(os.path.dirname(os.path.abspath(__file__)), "trace_time.py")}
    exec(compile(src, "trace_time.py", "exec"), ns)
    P, H = 300000, 800
    dev = torch.device("cuda:0")
    sc = make_shell_scene(P, seed=0, image_size=H).to(dev)
    cam = orbit_camera(0, H, H).to(dev)
    ro, rd, hit = ns["mirror_rays"](cam, H, H, dev)
    if mode == "primary":
        K = torch.as_tensor(cam.HWK[2], dtype=torch.float32, device=dev)
        c2w = cam.world_view_transform.to(dev).T.inverse()
        ys, xs = torch.meshgrid(torch.arange(H, device=dev, dtype=torch.float32), torch.arange(H, device=dev, dtype=torch.float32), indexing="ij")
        rd = ((torch.stack([xs, ys, torch.ones_like(xs)], dim=-1) @ torch.linalg.inv(K).T) @ c2w[:3, :3].T).contiguous()
        ro = c2w[:3, 3].expand_as(rd).contiguous()
    from materialrefgs_amd.gs_utils import build_rotation
    R = build_rotation(sc.rotations)
    su, sv = sc.scales[:, 0:1] * R[:, :, 0], sc.scales[:, 1:2] * R[:, :, 1]
    m = sc.means3D
    v = torch.stack([m - 3 * su + 3 * sv, m - 3 * su - 3 * sv, m + 3 * su + 3 * sv, m + 3 * su - 3 * sv], dim=1).reshape(-1, 3)
    eye = torch.eye(4, device=dev)
    ts = SurfelTracingSettings(H, H, 1.0, 1.0, torch.zeros(3, device=dev), 1.0, eye, eye, 0, torch.zeros(3, device=dev), False, False)
    tr = SurfelTracer()
    tr.build_acceleration_structure(v, None)
    with torch.no_grad():
        tr(ro, rd, v, means3D=sc.means3D, grads3D=None, shs=None, colors_precomp=torch.rand(P, 3, device=dev), others_precomp=torch.full((P, 2), 0.01, device=dev),
           opacities=sc.opacities, scales=sc.scales, rotations=sc.rotations, cov3D_precomp=None, tracer_settings=ts)
    torch.cuda.synchronize()
    st = tr.last_state[: H * H].reshape(H // 8, 8, H // 8, 8, 4).permute(0, 2, 1, 3, 4).reshape(-1, 64, 4).cpu()
    # a block's lanes carry the counters of the wave that finished them: whole-block packets have one value, split blocks several
    rows = []
    for b in range(st.shape[0]):
        blk = st[b]
        walked = blk[:, 3] < 0                     # walked in a packet
        if not walked.any():
            continue
        u = torch.unique(blk[walked][:, [0, 1, 2, 3]], dim=0)
        for r in u:
            n_l = int(((blk[:, :4] == r).all(dim=1)).sum())
            rows.append((float(r[0]), float(r[1]), float(r[2]), -float(r[3]), n_l))
    t = torch.tensor(rows)
    ticks, lanes, nodes, tests, nl = t[:, 0], t[:, 1], t[:, 2], t[:, 3], t[:, 4]
    full = nl == 64
    for name, sel in (("whole-block packets", full), ("smaller packets", ~full)):
        if sel.sum() == 0:
            continue
        print(f"{mode} {name}: {int(sel.sum())} waves, rays per wave {nl[sel].mean():.1f}; per wave: candidates {tests[sel].mean():.1f}, wall {ticks[sel].mean() / 100:.1f} us, "
              f"of it node / leaf fetch + beam test {nodes[sel].mean() / 100:.1f} us, candidate loops {lanes[sel].mean() / 100:.1f} us")

main()
