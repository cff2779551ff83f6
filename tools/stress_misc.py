"""Randomised soak run of the widened path (loss, mesh rays, Adam, row compaction) against their checkers.  GPU box, one-off."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import loss_oracle, trace_oracle  # noqa: E402
from materialrefgs_amd import densify, losses  # noqa: E402
from materialrefgs_amd.optim import Adam  # noqa: E402
from materialrefgs_amd.raytracing import RayTracer  # noqa: E402
from materialrefgs_amd.synthetic import sphere_mesh  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = "cuda"
bad = 0
t0 = time.time()


def rel(a, b):
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(np.asarray(a, dtype=np.float64) - b).max() / (np.abs(b).max() + 1e-30))


for i in range(n):
    msgs = []
    # ---- loss
    H, W, C = int(rng.integers(1, 200)), int(rng.integers(1, 200)), int(rng.integers(1, 5))
    mode = int(rng.integers(0, 3))
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    gt = torch.rand(C, H, W, generator=g)
    img = (gt + 0.1 * torch.randn(C, H, W, generator=g)).clamp(0, 1)
    rn, sn, dist, wt = torch.randn(3, H, W, generator=g), torch.randn(3, H, W, generator=g), torch.rand(1, H, W, generator=g), torch.rand(H, W, generator=g)
    lam_n, lam_d = (0.0, 0.0) if mode == 0 else (0.05, 100.0)
    leaves = [t.to(dev).requires_grad_(True) for t in (img, rn, sn, dist)]
    loss, terms = losses.fused_loss(leaves[0], gt.to(dev), leaves[1], leaves[2], leaves[3], wt.to(dev) if mode == 1 else None, 0.2, lam_n, lam_d)
    loss.backward()
    t_o, g_o = loss_oracle.calculate_loss(img.numpy(), gt.numpy(), rn.numpy(), sn.numpy(), dist.numpy(), wt.numpy() if mode == 1 else None, 0.2, lam_n, lam_d)
    if abs(float(loss) - t_o["loss"]) > 2e-5 * max(1.0, abs(t_o["loss"])):
        msgs.append(f"loss value {float(loss)} vs {t_o['loss']}")
    if rel(leaves[0].grad.cpu().numpy(), g_o["image"]) > 5e-4:
        msgs.append("loss dL/dimage")
    if mode and (rel(leaves[1].grad.cpu().numpy(), g_o["rend_normal"]) > 1e-5 or rel(leaves[3].grad.cpu().numpy().reshape(g_o["rend_dist"].shape), g_o["rend_dist"]) > 1e-5):
        msgs.append("loss normal/dist gradients")
    # ---- mesh rays
    nl, no = int(rng.integers(3, 30)), int(rng.integers(4, 40))
    v, t = sphere_mesh(nl, no, float(rng.uniform(0.3, 2.0)), float(rng.uniform(0, 0.05)), int(rng.integers(1 << 30)))
    if len(t) > 8:
        nr = int(rng.integers(1, 3000))
        o = rng.normal(size=(nr, 3)).astype(np.float32) * 2.5
        d = rng.normal(size=(nr, 3)).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        rt = RayTracer(v, t)
        pos, nrm, depth, ids = rt.trace(torch.from_numpy(o).to(dev), torch.from_numpy(d.astype(np.float32)).to(dev), return_faceids=True)
        rpos, rnrm, rdepth, rids = trace_oracle.trace(v, t, o, d.astype(np.float32))
        if not np.array_equal(depth.cpu().numpy(), rdepth):
            msgs.append("trace depth")
    # ---- Adam + compaction
    P = int(rng.choice([1, 5, 63, 64, 65, 1023, 1025, 5000, 70001]))
    shapes = [(P, 3), (P, 1), (P, 15, 3), (P, 4), (int(rng.integers(1, 50)),)]
    pa = [torch.nn.Parameter(torch.randn(*s, generator=g).to(dev)) for s in shapes]
    pb = [torch.nn.Parameter(p.data.clone()) for p in pa]
    oa = Adam([{"params": [p], "lr": 10.0 ** -rng.integers(1, 5), "name": f"g{k}"} for k, p in enumerate(pa)], lr=0.0, eps=1e-15)
    ob = torch.optim.Adam([{"params": [p], "lr": grp["lr"]} for p, grp in zip(pb, oa.param_groups)], lr=0.0, eps=1e-15)
    for _ in range(3):
        for x, y in zip(pa, pb):
            gr = (torch.randn(*x.shape, generator=g) * 10.0 ** -float(rng.integers(0, 4))).to(dev)
            x.grad, y.grad = gr.clone(), gr.clone()
        oa.step(); ob.step()
    for x, y in zip(pa, pb):
        if not torch.allclose(x.data, y.data, rtol=1e-5, atol=1e-6):
            msgs.append("adam")
            break
    keep = (torch.rand(P, generator=g) < float(rng.uniform(0, 1))).to(dev)
    out, m = densify.compact_rows([p.data for p in pa[:4]], keep)
    if m != int(keep.sum()) or any(not torch.equal(a, p.data[keep]) for a, p in zip(out, pa[:4])):
        msgs.append("compaction")
    bad += bool(msgs)
    print(f"[{i:3d}] loss {C}x{H}x{W} mode {mode}; mesh {len(t)} tris; P={P}: {'ok' if not msgs else 'FAIL ' + ', '.join(msgs)}", flush=True)
print(f"{n - bad} of {n} rounds passed in {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
