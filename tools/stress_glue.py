"""Randomised soak of the glue epilogue (MrgsRasterGrads::glue_params): render_surfel without opt.indirect, both flavours, random sizes incl.
partial and single-lane waves -- the rasterizer's per-gaussian backward carrying on through the glue's backward in one kernel against the
two-kernel path on the same scene: identical maps, every leaf gradient within 1e-5 of its largest element, exact zeros at the indirect
coefficients.  (The blend backward's float atomics reorder their sums between any two runs: the log also carries, per case, the distance
between two runs of the SAME two-kernel path -- the noise floor the comparison sits on, up to ~3e-6 on lone or tiny surfels -- and the
epilogue is the same arithmetic compiled without FMA contraction.)
    python tools/stress_glue.py <cases> <seed>"""
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import materialrefgs_amd.renderer as renderer_mod  # noqa: E402
from materialrefgs_amd.renderer import render_surfel  # noqa: E402
from materialrefgs_amd.synthetic import orbit_camera  # noqa: E402
from test_render_e2e import PARAMS, _loss, _models  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
bad, t0 = 0, time.time()
for i in range(n):
    P = int(rng.choice([1, 2, 63, 64, 65, 127, 500, 3000, 20000]))
    H, W = int(rng.integers(17, 260)), int(rng.integers(17, 260))
    flag = str(rng.choice(["2dgs", "pgsr"]))
    view, seed, srgb = int(rng.integers(0, 8)), int(rng.integers(1 << 30)), bool(rng.integers(0, 2))
    cam = orbit_camera(view, H, W).to(dev)
    pipe = SimpleNamespace(depth_ratio=float(rng.choice([0.0, 1.0])), debug=False)
    bg = torch.rand(3, generator=torch.Generator().manual_seed(seed)).to(dev)
    res = {}
    try:
        for fused in (True, False, None):             # None: the two-kernel path a second time (the run-to-run noise floor)
            renderer_mod._FUSE_GLUE = bool(fused)
            _pc_o, _base_o, pc_h, env = _models(P, H, W, seed=seed % 100000, dev=dev)
            env.build_mips()
            out = render_surfel(cam, pc_h, pipe, bg, srgb=srgb, opt=SimpleNamespace(indirect=False), flag=flag)
            loss = _loss(out, H, W, False, dev)
            if flag != "2dgs":
                g = torch.Generator().manual_seed(17)
                loss = loss + (out["rend_distance"] * torch.rand(out["rend_distance"].shape, generator=g).to(dev)).sum()
            loss.backward()
            grads = {k: getattr(pc_h, k).grad.detach().clone() for k in PARAMS}
            grads["env.base"] = env.base.grad.detach().clone()
            grads["viewspace_points"] = out["viewspace_points"].grad.detach().clone()
            res[fused] = ({k: v.detach().clone() for k, v in out.items() if torch.is_tensor(v)}, grads)
        (m1, g1), (m0, g0), (_m, g00) = res[True], res[False], res[None]
        for k in m0:
            assert torch.equal(m0[k], m1[k]), f"map {k}"
        worst, noise = 0.0, 0.0
        for k in g0:
            assert bool(torch.isfinite(g1[k]).all()), f"non-finite {k}"
            m = max(float(g0[k].abs().max()), 1e-30)
            e, e00 = float((g0[k] - g1[k]).abs().max()) / m, float((g0[k] - g00[k]).abs().max()) / m
            worst, noise = max(worst, e), max(noise, e00)
            assert e <= 1e-5, f"grad {k}: {e:.3e} (same path twice: {e00:.3e})"
        for k in ("_indirect_dc", "_indirect_rest"):
            assert float(g1[k].abs().max()) == 0.0, k
        status = f"ok (epilogue vs two kernels {worst:.1e}, two kernels twice {noise:.1e})"
    except AssertionError as e:
        bad += 1
        status = "FAIL " + str(e)[:200]
    finally:
        renderer_mod._FUSE_GLUE = True
    print(f"[{i:3d}] P={P:6d} {H}x{W} {flag} srgb={int(srgb)} view={view} seed={seed}: {status}", flush=True)
print(f"{n - bad} of {n} cases passed in {time.time() - t0:.0f} s")
from bench import kernel_source_digest  # noqa: E402
print(f"kernel_source_digest: {kernel_source_digest()}")
sys.exit(1 if bad else 0)
