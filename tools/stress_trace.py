"""Randomised soak of the surfel ray tracer against its dense statement (oracle/surfel_trace_oracle.py): scene size, splat size, opacity
range, ray set (image-shaped coherent rays, random rays, mixtures with missing directions), forward outputs and all gradients.
python tools/stress_trace.py [n_cases] [first_seed] -> one JSON line per failing case and a summary line.  Needs a GPU."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import surfel_trace_oracle as sto  # noqa: E402
from materialrefgs_amd.surfel_tracing import SurfelTracer, SurfelTracingSettings  # noqa: E402
from materialrefgs_amd.synthetic import make_shell_scene  # noqa: E402


def one_case(seed, dev):
    g = torch.Generator().manual_seed(seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    rf = lambda lo, hi: float(lo + (hi - lo) * torch.rand(1, generator=g))
    P = [1, 3, 63, 64, 65, 700, 4097, 9000][ri(0, 7)]
    sc = make_shell_scene(P, seed=seed, radius_px=rf(15.0, 120.0), image_size=400, scale_sigma=rf(0.2, 0.9))
    opac = (sc.opacities * rf(0.05, 1.0)).clamp(1e-3, 0.999)
    colors, others = torch.rand(P, 3, generator=g), torch.rand(P, 2, generator=g)
    kind = ri(0, 3)
    if kind == 0:       # an image of rays from one point (coherent blocks)
        H, W = ri(5, 40), ri(5, 40)
        o0 = torch.randn(3, generator=g) * rf(0.0, 2.5)
        ys, xs = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")
        tgt = torch.stack([xs, ys, torch.zeros_like(xs)], -1) * rf(0.2, 1.2) + torch.randn(3, generator=g) * 0.2
        d = tgt - o0
        o = o0.expand_as(d).clone()
        shape = (H, W, 3)
    elif kind == 1:     # random rays
        n = ri(1, 1500)
        o = torch.randn(n, 3, generator=g) * rf(0.1, 2.0)
        d = torch.randn(n, 3, generator=g)
        shape = (n, 3)
    elif kind == 2:     # image of mirror-like rays from a sphere, some without a direction
        H, W = ri(8, 36), ri(8, 36)
        ys, xs = torch.meshgrid(torch.linspace(-1.2, 1.2, H), torch.linspace(-1.2, 1.2, W), indexing="ij")
        r2 = xs * xs + ys * ys
        hit = r2 < 1.0
        z = torch.sqrt((1.0 - r2).clamp_min(0))
        nrm = torch.stack([xs, ys, z], -1)
        view = torch.tensor([0.0, 0.0, -1.0])
        refl = view - 2 * (nrm * view).sum(-1, keepdim=True) * nrm
        o = torch.where(hit[..., None], nrm * rf(0.9, 1.1), torch.zeros(3))
        d = torch.where(hit[..., None], refl * rf(0.3, 3.0), torch.zeros(3))
        shape = (H, W, 3)
    else:               # image with noisy directions (blocks that split)
        H, W = ri(8, 30), ri(8, 30)
        o = torch.randn(H, W, 3, generator=g) * 0.05
        d = torch.tensor([0.0, 0.0, 1.0]) + torch.randn(H, W, 3, generator=g) * rf(0.01, 0.5)
        shape = (H, W, 3)
    o, d = o.reshape(shape).contiguous(), d.reshape(shape).contiguous()
    n = o.numel() // 3
    bg = torch.rand(3, generator=g)
    mod = rf(0.7, 1.5)
    leaf = lambda t, dt, dv: t.to(dt).to(dv).clone().requires_grad_(True)
    def run(hip, ref_dtype=torch.float64):
        dt, dv = (torch.float32, dev) if hip else (ref_dtype, "cpu")
        L = dict(means=leaf(sc.means3D, dt, dv), scales=leaf(sc.scales, dt, dv), rot=leaf(sc.rotations, dt, dv), op=leaf(opac, dt, dv),
                 col=leaf(colors, dt, dv), oth=leaf(others, dt, dv), o=leaf(o, dt, dv), d=leaf(d, dt, dv))
        if hip:
            tr = SurfelTracer()
            tr.build_acceleration_structure(sto.quad_vertices(L["means"].detach(), L["scales"].detach(), L["rot"].detach(), mod).reshape(-1, 3), None)
            eye = torch.eye(4, device=dev)
            ts = SurfelTracingSettings(1, n, 1.0, 1.0, bg.to(dev), mod, eye, eye, 0, torch.zeros(3, device=dev), False, False)
            rgb, dpt, acc, norm, dist, aux, mid, wet = tr(L["o"], L["d"], None, means3D=L["means"], grads3D=None, shs=None, colors_precomp=L["col"],
                                                           others_precomp=L["oth"], opacities=L["op"], scales=L["scales"], rotations=L["rot"],
                                                           cov3D_precomp=None, tracer_settings=ts)
            out = dict(rgb=rgb.reshape(n, 3), dpt=dpt.reshape(n), acc=acc.reshape(n), norm=norm.reshape(n, 3), dist=dist.reshape(n), aux=aux.reshape(n, 2))
        else:
            dd = L["d"].reshape(n, 3)
            ok = (dd.abs().sum(-1) > 0)
            safe_d = torch.where(ok[:, None], dd, torch.ones_like(dd))
            r = sto.trace_dense(L["o"].reshape(n, 3), safe_d, L["means"], L["scales"], L["rot"], L["op"], L["col"], L["oth"], bg.to(dt), mod)
            okf = ok.to(dt)
            out = dict(rgb=r["rgb"] * okf[:, None] + bg.to(dt) * (1 - okf[:, None]), dpt=r["dpt"] * okf, acc=r["acc"] * okf, norm=r["norm"] * okf[:, None],
                       dist=r["dist"] * okf, aux=r["aux"] * okf[:, None])
        return L, out
    Lh, oh = run(True)
    Lr, orf = run(False)
    same = torch.ones(n, dtype=torch.bool)
    worst = 0.0
    for k in oh:
        err = (oh[k].detach().cpu().double() - orf[k].detach()).abs().reshape(n, -1).max(dim=1).values
        tol = 3e-4 * max(1.0, float(orf[k].detach().abs().max()))
        same &= err <= tol
        worst = max(worst, float(err[err <= tol].max()) if bool((err <= tol).any()) else 0.0)
    n_off = int((~same).sum())
    up = {k: torch.randn(oh[k].shape, generator=g) for k in oh}
    m = same.double()
    shp = lambda k: m.reshape(n, *([1] * (up[k].dim() - 1)))
    sum((oh[k] * (up[k] * shp(k).float()).to(dev)).sum() for k in oh).backward()
    sum((orf[k] * (up[k].double() * shp(k))).sum() for k in oh).backward()
    gworst, gname = 0.0, ""
    for k in Lh:
        a, b = Lh[k].grad.cpu().double(), Lr[k].grad
        if k == "d":   # rays without a direction have no gradient in either
            pass
        e = float((a - b).abs().max()) / max(float(b.abs().max()), 1e-9)
        if not (e == e):
            e = float("inf")
        if e > gworst:
            gworst, gname = e, k
    res = dict(seed=seed, P=P, kind=kind, rays=n, off=n_off, fwd_worst=worst, grad_worst=gworst, grad_of=gname, hits_max=int(0))
    if gworst > 5e-3:
        # how well conditioned is this gradient?  The same dense statement evaluated in float32 on the CPU, against its float64 self.
        try:
            L32, o32 = run(False, torch.float32)
            sum((o32[k] * (up[k] * shp(k).float())).sum() for k in oh).backward()
            a, b = L32[gname].grad.double(), Lr[gname].grad
            res["dense_f32_vs_f64"] = float((a - b).abs().max()) / max(float(b.abs().max()), 1e-9)
        except Exception as ex:       # noqa: BLE001
            res["dense_f32_error"] = repr(ex)[:200]
    return res


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    dev = torch.device("cuda:0")
    bad = 0
    for s in range(first, first + n_cases):
        try:
            r = one_case(s, dev)
        except Exception as ex:       # noqa: BLE001
            r = dict(seed=s, error=repr(ex)[:300])
        fail = "error" in r or r["off"] > max(2, r["rays"] // 150) or r["grad_worst"] > 5e-3
        if fail:
            bad += 1
            print(json.dumps(r))
    print(json.dumps(dict(cases=n_cases, first_seed=first, failed=bad)))
    # the identity of the kernels this log speaks for (tools/publish_profiles.sh refuses a log whose digest is not the tree's)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import kernel_source_digest
    print(f"kernel_source_digest: {kernel_source_digest()}")


main()
