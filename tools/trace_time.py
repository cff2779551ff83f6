"""Times mrgs_bvh_trace: reflection-like rays against a ~1 M triangle sphere mesh.  Developer tool."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tests.test_raytracing import sphere_mesh
from materialrefgs_amd.raytracing import RayTracer

n_lat, n_lon = (700, 720) if len(sys.argv) < 2 else (int(sys.argv[1]), int(sys.argv[2]))
res = 800 if len(sys.argv) < 4 else int(sys.argv[3])
v, t = sphere_mesh(n_lat, n_lon, 1.0, 0.01)
t0 = time.perf_counter()
rt = RayTracer(v, t)
print(f"{len(t)} triangles: host build + upload {time.perf_counter() - t0:.2f} s, blob {rt.blob.numel() / 1e6:.1f} MB")
# primary rays of a res x res pinhole camera at (0,-4,0) looking at the origin (coherent, like the per-pixel rays of a view)
ys, xs = torch.meshgrid(torch.linspace(-0.36, 0.36, res, device="cuda"), torch.linspace(-0.36, 0.36, res, device="cuda"), indexing="ij")
d = torch.nn.functional.normalize(torch.stack([xs, torch.ones_like(xs), ys], -1).reshape(-1, 3), dim=-1)
o = torch.tensor([0.0, -4.0, 0.0], device="cuda").expand_as(d).contiguous()
pos, nrm, depth = rt.trace(o, d)
hit = depth < 10
print(f"primary: {int(hit.sum())} of {len(d)} hit")
# reflection rays from the hit points (what get_specular_color_surfel traces)
r = torch.nn.functional.normalize(d - 2 * (d * nrm).sum(-1, keepdim=True) * nrm, dim=-1)
for name, (oo, dd) in {"primary": (o, d), "reflection": (pos[hit].contiguous(), r[hit].contiguous())}.items():
    for _ in range(3):
        rt.trace(oo, dd)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        rt.trace(oo, dd)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{name}: {len(oo)} rays {ms:.3f} ms  {len(oo) / ms / 1e3:.1f} Mrays/s")
