#!/usr/bin/env python3
"""bench.py -- full-render fwd+bwd views/s of the surfel renderer hot path on MI355X (BASELINE.json metric).

A "step" is one FULL render of one view, forward AND backward.  The number of record (default workload, C3full) is the reference's
`render_surfel` (gaussian_renderer/__init__.py:225-483) at BASELINE.json's 800x800 / 300 000 surfels: per-gaussian material features ->
the drop-in rasterizer with S = 8 material channels (preprocess -> binning -> per-tile blend) -> 2DGS map post-processing -> deferred
split-sum shading (FG LUT + prefiltered cubemap, rebuilt for the view as the reference's training loop does) -> compositing, and the
backward of all of it, on the synthetic shell scene of SURVEY.md section 8d with the inputs resident in HBM.  `--workload C2` is the
rasterizer alone with S = 0 (BASELINE.json configs[1]); the default run reports it as `secondary_raster` next to C2heavy, the traced
last training stage (C3trace) and BASELINE.json configs[3] (C4trace).  With --gpus N (one rank per GPU) every rank renders its own view
of the step and the per-gaussian gradients are summed through the factored exchange of materialrefgs_amd/dist.py ("weak" scaling).

Prints ONE JSON line (rank 0).  `roofline` is computed for the dominant kernel from HIP-event durations recorded on the launch stream
during the timed region; `cpu_baseline` times the CPU checkers (a port: the reference has no CPU path) on one view of the same workload.
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (P, H, W, S, description)
    "C2": (300000, 800, 800, 0, "C2 shell scene: P=300000 surfels, 800x800, SH deg 3, S=0 (diffuse-only surfel raster), fwd+bwd"),
    "C3": (300000, 800, 800, 8, "C3 shell scene: P=300000 surfels, 800x800, SH deg 3, S=8 material channels, fwd+bwd"),
    # BASELINE.json configs[2]: the full render_surfel path = per-gaussian material features -> rasterizer (S=8) -> 2DGS map
    # post-processing -> deferred split-sum shading (FG LUT + 5-level 128^2 cubemap) -> compositing, forward and backward
    "C3full": (300000, 800, 800, 8, "C3 shell scene through render_surfel: P=300000, 800x800, S=8 + deferred BRDF shading, fwd+bwd"),
    # C3full with the reference's per-view loss (calculate_loss: L1 + SSIM + normal consistency + distortion) producing the upstream
    # gradients instead of constants: one training view without the optimizer step
    "C3train": (300000, 800, 800, 8, "C3 shell scene: render_surfel + calculate_loss + Adam step, P=300000, 800x800, S=8, one full training view"),
    # BASELINE.json configs[3]: render_surfel with opt.indirect: visibility (mirror) rays of all H*W pixels against a ~1 M triangle mesh
    "C4full": (1000000, 1600, 1600, 8, "C4 shell scene: render_surfel + visibility rays vs a 1 M triangle mesh + calculate_loss + Adam step, P=1000000, 1600x1600, S=8, one full training view"),
    # raster part of BASELINE.json configs[3]
    "C4raster": (1000000, 1600, 1600, 8, "C4-size shell scene: P=1000000 surfels, 1600x1600, SH deg 3, S=8 material channels, raster fwd+bwd"),
    # same P / image as C2 with the pair count the survey assumed (R ~ 6 P instead of 3.8 P) and heavy-tailed splat sizes (log-normal
    # sigma 0.8 instead of 0.35: splats from sub-pixel to ~450 px); reported as the secondary line of the default run
    "C2heavy": (300000, 800, 800, 0, "C2heavy shell scene: P=300000 surfels, 800x800, SH deg 3, S=0, log-normal splat sizes sigma 0.8, R ~ 6 P, fwd+bwd"),
    # the last training stage of the reference (train_refnerf.py:1501-1504): render_surfel, then the same surfels traced along every pixel's
    # mirror ray (HardwareRendering, gaussian_renderer/__init__.py:486-520) and blended in; hierarchy rebuilt every view as in training
    "C3trace": (300000, 800, 800, 8, "C3 shell scene through render_surfel_with_envgs: render_surfel + surfel-traced mirror rays of all 800x800 pixels (hierarchy rebuilt per view), fwd+bwd"),
    # BASELINE.json configs[3] with the traced reflection term: C4 size through render_surfel_with_envgs
    "C4trace": (1000000, 1600, 1600, 8, "C4-size shell scene through render_surfel_with_envgs: P=1000000, 1600x1600, render_surfel + surfel-traced mirror rays of all 2.56 M pixels (hierarchy rebuilt per view), fwd+bwd"),
    # C3full in the flavour the reference SHIPS (arguments/config.py:1 FLAG = "pgsr"): + the plane distance as a ninth rasterized channel,
    # surf_depth / surf_normal from the flavour's unbiased depth (gaussian_renderer/__init__.py:64-69, 348-357); parity of allmap[7] unpinned
    "C3full-pgsr": (300000, 800, 800, 8, "C3 shell scene through render_surfel(flag='pgsr'): P=300000, 800x800, S=8+1 (plane distance) + deferred BRDF shading, fwd+bwd"),
    "tiny": (20000, 400, 400, 8, "tiny debug scene (not a benchmark configuration)"),
    "tinyfull": (20000, 400, 400, 8, "tiny debug scene through render_surfel (not a benchmark configuration)"),
}
SCENE_KW = {"C2heavy": dict(radius_px=6.5, scale_sigma=0.8)}
HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes(kernel, P, R, HW, S):
    """Algorithmic HBM bytes of ONE launch (DESIGN.md section 'Kernels'); each datum moved once per kernel."""
    if kernel == "render_bwd":
        # list ids + cull bits (5 B), staged records (80 B + features), per-pixel inputs (dL_dpix 3+S, dL_dothers 7, final_T 3,
        # n_contrib 2), one RMW of the (18+S)-float gradient row per (tile, gaussian) pair
        return (5 + 80 + 4 * S) * R + (60 + 4 * S) * HW + 8 * (18 + S) * R
    if kernel == "render_fwd":
        # list ids + cull bits (5 B), staged records (80 B + features), per-pixel outputs (color 3, feature S, others 7,
        # final_T 3, n_contrib 2)
        return (5 + 80 + 4 * S) * R + (60 + 4 * S) * HW
    if kernel == "surfel_trace_fwd":
        # P = surfels, R = blended hits of all rays, HW = rays.  Per ray: origin + direction in (24 B), rgb / dpt / acc / norm / dist / aux out
        # (44 B), its four state words (16 B); per blended hit: the surfel's geometry and attribute records gathered (64 + 32 B) and its id in
        # the replay record (4 B); per surfel: its records laid out in leaf order once (96 B read + 96 B written)
        return 84 * HW + 100 * R + 192 * P
    raise KeyError(kernel)


def kernel_source_digest():
    """sha256[:16] over the HIP sources and headers of libmrgs.so, in name order: the identity of the kernels a counter was measured on
    (the repository's .git does not travel to the GPU box, so a commit id is not available there)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "materialrefgs_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")) + [os.path.join(csrc, "Makefile")]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def _load_json(path):
    try:
        return json.load(open(path))
    except Exception:
        return {}


def spawn_ranks(n):
    """Start `n` ranks of this script under torch.distributed.run (one per GPU, rendezvous on 127.0.0.1) and return its exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(cmd, env=env).returncode


def plumbing_only(args, world, rank):
    """Launch-path check (tests/test_dist_cpu.py): every rank joins the group, one flat gradient bucket goes through the
    all-reduce, rank 0 prints a line that says what it is.  Nothing is rendered and nothing here is a measurement."""
    from materialrefgs_amd import dist as mdist
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))) if torch.cuda.is_available() and \
        os.environ.get("MRGS_DIST_BACKEND") != "gloo" else torch.device("cpu")
    shapes = [torch.Size((1000, 3)), torch.Size((1000, 1)), torch.Size((1000, 4))]
    bucket = mdist.GradBucket(shapes, dev)
    grads = [torch.full(tuple(s), float(rank + 1), device=dev) for s in shapes]
    summed = mdist.allreduce_gradients(bucket, grads)
    expect = world * (world + 1) / 2
    ok = all(bool((t == expect).all()) for t in summed)
    if world > 1:
        torch.distributed.barrier()
    if rank == 0:
        print(json.dumps({"metric": "plumbing check only (no render, not a benchmark result)", "value": None, "unit": None,
                          "n_gpus": world, "plumbing_only": True, "allreduce_ok": ok,
                          "backend": torch.distributed.get_backend() if world > 1 else None}))
    if world > 1:
        torch.distributed.destroy_process_group()
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)     # ~1 s timed region at C3full
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="C3full", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU oracle leg (also skips grad_max_rel_err)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary lines of the default run (C2, C2heavy, C3trace, C4trace)")
    ap.add_argument("--dump-grads", default=None, metavar="PATH",
                    help="after the measurement render step --dump-step once more and let rank 0 save the (reduced) gradient tensors to PATH "
                         "(.npz): tests/test_dist_gpu.py compares a 2-rank run with the sum of two single-rank runs")
    ap.add_argument("--dump-step", type=int, default=0, help="step index of --dump-grads (rank r renders view (step * world + r) mod 8)")
    ap.add_argument("--plumbing-only", action="store_true",
                    help="no render: launch the ranks, check the world size and push one gradient bucket through the collective "
                         "(works without a GPU over gloo; the line it prints is NOT a benchmark result)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves.  A fresh child process, started before
        # anything here has touched the GPU (never an exec of this one); rank 0 of the child prints the JSON line on our stdout.
        return spawn_ranks(args.gpus)

    from materialrefgs_amd import dist as mdist
    env = mdist.init_from_env(backend=os.environ.get("MRGS_DIST_BACKEND"))   # default: RCCL ("nccl"); "gloo" for plumbing checks on one GPU
    world, rank, local = env["world"], env["rank"], env["local"]
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if world > 1:
        world = torch.distributed.get_world_size()   # what the process group (RCCL) reports, not what the environment claimed
    if args.plumbing_only:
        return plumbing_only(args, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (there is no CPU fallback)")
    dev = torch.device("cuda", local % max(torch.cuda.device_count(), 1))
    torch.cuda.set_device(dev)

    from materialrefgs_amd import _lib
    from materialrefgs_amd._lib import MrgsKernelTimes
    from materialrefgs_amd import rasterizer as rasterizer_mod
    from materialrefgs_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    from materialrefgs_amd.synthetic import make_shell_scene, orbit_camera, upstream_grads
    L = _lib.lib()

    P, H, W, S, desc = WORKLOADS[args.workload]
    scene_kw = dict(radius_px=7.0 * max(H, W) / 800.0 if args.workload in ("tiny", "tinyfull") else 7.0)
    scene_kw.update(SCENE_KW.get(args.workload, {}))
    scene_cpu = make_shell_scene(P, S=S, seed=0, image_size=max(H, W), **scene_kw)
    scene = scene_cpu.to(dev)
    # Eight orbit cameras, or fewer when the warm-up is too short to visit each of them TWICE: the timed steps are then steady-state
    # visits -- a camera's second visit still launches the ordering kernel (the first visit has no measured work to order by), from the
    # third on the queues are reused (rasterizer._hint_flags).  With the driver's --warmup 5 that is two cameras; round 5 cycled five
    # and had every camera's second visit inside its 20 timed steps.  What a FIRST visit costs (every camera once per training run and
    # again after each densification: the normal case of the reference's first 25 000-30 000 iterations, arguments/__init__.py:159-162)
    # is measured apart and reported as `cold_ms_per_step`.  MRGS_BENCH_VIEWS overrides (developer A/B).
    n_views = int(os.environ.get("MRGS_BENCH_VIEWS", "0")) or min(8, max(1, args.warmup // 2) * world)
    cams = [orbit_camera(v, H, W, n_views=8) for v in range(n_views)]
    settings = []
    for cam in cams:
        settings.append(GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
            bg=torch.zeros(3, device=dev), scale_modifier=1.0, viewmatrix=cam.world_view_transform.to(dev),
            projmatrix=cam.full_proj_transform.to(dev), sh_degree=3, campos=cam.camera_center.to(dev), prefiltered=False,
            debug=False))
    g_color, g_feat, g_others = upstream_grads(S, H, W, device=dev)

    surfel_mode = args.workload in ("C3full", "C3full-pgsr", "C3train", "C4full", "C3trace", "C4trace", "tinyfull")
    flavour = "pgsr" if args.workload.endswith("-pgsr") else "2dgs"
    traced = args.workload in ("C3trace", "C4trace")
    use_loss = args.workload in ("C3train", "C4full")
    indirect = args.workload == "C4full"
    if surfel_mode:
        from types import SimpleNamespace
        from materialrefgs_amd.renderer import SurfelModel, render_surfel
        from materialrefgs_amd.shading import EnvLight
        from materialrefgs_amd.synthetic import make_surfel_model
        gen = torch.Generator().manual_seed(1)
        pc, env, surfel_params = make_surfel_model(P, max(H, W), dev, seed=0, radius_px=scene_kw["radius_px"])
        pipe = SimpleNamespace(depth_ratio=0.0, debug=False, compute_cov3D_python=False, convert_SHs_python=False)
        bg_color = torch.zeros(3, device=dev)
        if traced:
            from materialrefgs_amd.renderer import render_surfel_with_envgs
            from materialrefgs_amd.surfel_tracing import HardwareRendering
            hw_tracer = HardwareRendering().train()
        cams_dev = [c.to(dev) for c in cams]
        if indirect:
            from materialrefgs_amd.raytracing import RayTracer
            from materialrefgs_amd.synthetic import make_occluder_mesh
            pc.ray_tracer = RayTracer(*make_occluder_mesh(1_000_000), device=dev)
        if use_loss:
            from materialrefgs_amd import losses
            gt_cams = []            # synthetic ground truth + its edge weight, once per camera (train_refnerf.py:1176-1179)
            for _c in cams_dev:
                gt = torch.rand(3, H, W, generator=gen).to(dev)
                gt_cams.append(SimpleNamespace(original_image=gt, image_weight=losses.image_weight(gt)))
            from materialrefgs_amd.optim import Adam
            # learning rates of GaussianModel.training_setup (scene/gaussian_model.py:422-443, arguments/__init__.py defaults)
            # (one group per tensor as there; the rate is kept tiny so that the synthetic scene stays the benchmark's scene --
            # the work of an Adam step does not depend on it)
            optimizer = Adam([{"params": [t_], "lr": 1e-6} for t_ in surfel_params], lr=0.0, eps=1e-15)
            loss_opt = SimpleNamespace(lambda_dssim=0.2, lambda_normal_render_depth=0.05, normal_loss_start=0, lambda_dist=100.0,
                                       dist_loss_start=3000, lambda_normal_smooth=0.0, lambda_depth_smooth=0.0,
                                       normal_smooth_from_iter=0, normal_smooth_until_iter=18000, use_perceptual_loss=False)
    params = {"means3D": scene.means3D, "opacity": scene.opacities, "scales": scene.scales, "rotations": scene.rotations,
              "sh": scene.shs}
    if S > 0:
        params["features"] = scene.features
    params = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    means2D = torch.zeros_like(scene.means3D, requires_grad=True)   # screenspace_points (gaussian_renderer/__init__.py:229)
    grad_names = list(params.keys()) + ["means2D"]
    # view-parallel exchange: dense all-reduce of 13 gradient floats per gaussian + factored exchange of the SH gradient
    # (materialrefgs_amd/dist.py: FactoredGradReducer)
    reducer = mdist.FactoredGradReducer([params[k].shape for k in params] + [means2D.shape], list(params.keys()).index("sh"), dev) \
        if world > 1 else None
    state = {"R": 0}
    if reducer is not None and not surfel_mode and not os.environ.get("MRGS_BENCH_NO_EARLY_GATHER"):
        # the all-gather of the colour-gradient factor starts in the middle of the rasterizer's backward (under the per-gaussian backward)
        rasterizer_mod.set_after_blend_hook(lambda drgb: reducer.begin_early(drgb, state["campos"]))

    # render_surfel's parameter set: both SH families (96 of 111 floats per gaussian) travel factored (dist.SurfelGradReducer)
    surfel_names = ["xyz", "scaling", "rotation", "opacity", "features_dc", "features_rest", "refl_strength", "roughness", "ori_color",
                    "indirect_dc", "indirect_rest", "env_base"]
    # (the dense part as the union of the rows the ranks' views touched: dist.TouchedRowsExchange; MRGS_BENCH_DENSE_EXCHANGE=1 for A/B)
    surfel_reducer = mdist.SurfelGradReducer([t_.shape for t_ in surfel_params], surfel_names, dev,
                                             touched_rows_only=not os.environ.get("MRGS_BENCH_DENSE_EXCHANGE")) if (surfel_mode and world > 1) else None
    if surfel_reducer is not None and not traced and not os.environ.get("MRGS_BENCH_NO_EARLY_GATHER"):
        # both factors of the SH exchange leave the backward as soon as they are final: the colour factor between the rasterizer's blend
        # backward and its per-gaussian backward, the indirect-radiance factor after the per-gaussian glue's backward.  (Not with the
        # traced term: the tracer evaluates the same colour SH a second time, and the sum of two clamp-masked terms is what reduce() reads.)
        import materialrefgs_amd.renderer as renderer_mod
        rasterizer_mod.set_after_blend_hook(lambda drgb: surfel_reducer.begin_early_rgb(drgb, state["campos"]))
        renderer_mod.set_after_features_hook(lambda d_ind: surfel_reducer.begin_early_ind(d_ind))

    def reduce_surfel(view):
        summed = surfel_reducer.reduce([t_.grad for t_ in surfel_params], pc._xyz, pc._rotation, cams_dev[view].camera_center, pc.active_sh_degree)
        for t_, g_ in zip(surfel_params, summed):
            t_.grad = g_

    parts = {}              # diagnostic steps only: torch event pairs around the parts of a surfel view (name -> [(start, end), ...])

    def mark(name):
        if not state.get("parts_on"):
            return lambda: None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        def done():
            e1.record()
            parts.setdefault(name, []).append((e0, e1))
        return done

    def surfel_forward(i):
        view = (i * world + rank) % len(settings)
        state["campos"] = cams_dev[view].camera_center
        for t_ in surfel_params:
            t_.grad = None
        done = mark("env_prefilter_fwd")
        env.build_mips()                                       # every iteration in the reference (train_refnerf.py:1157-1163)
        done()
        done = mark("forward_total")
        if traced:
            out = render_surfel_with_envgs(hw_tracer, cams_dev[view], pc, pipe, bg_color, srgb=False, opt=SimpleNamespace(indirect=False))
        else:
            out = render_surfel(cams_dev[view], pc, pipe, bg_color, srgb=False, opt=SimpleNamespace(indirect=indirect), flag=flavour)
        done()
        return view, out

    def surfel_backward(view, out):
        """Loss (or the fixed upstream gradients) and the backward pass; with world > 1 its hooks start the early all-gathers."""
        if use_loss:
            loss, _tb = losses.calculate_loss(gt_cams[view], pc, out, loss_opt, 5000, gt_cams[view].image_weight, None)
            loss.backward()
            return
        # the maps calculate_loss consumes (utils/loss_utils.py:147-152,166), with fixed upstream gradients
        outs = [out["render"], out["rend_alpha"], out["rend_normal"], out["rend_dist"], out["surf_depth"], out["surf_normal"]]
        if "g" not in state:   # constant upstream gradients, built once (a loss would produce them in training)
            state["g"] = [torch.ones_like(outs[0])] + [torch.full_like(o, 0.1) for o in outs[1:]]
        done = mark("backward_total")
        torch.autograd.backward(outs, state["g"])
        done()

    def surfel_exchange_and_update(view):
        """What couples the ranks and what changes state: the gradient exchange and the optimizer step.  NEVER inside a deferred_count
        box: a workspace overflow surfaces at box.finish(), and by then a rank must not have joined a collective with the empty
        render's zero gradients, nor stepped Adam on them."""
        if world > 1:
            reduce_surfel(view)
        if use_loss:
            optimizer.step()

    _PHASES = [] if os.environ.get("MRGS_BENCH_STEP_TIMES") else None
    _SYNC_COUNT = bool(os.environ.get("MRGS_BENCH_SYNC_COUNT"))     # developer A/B: every forward waits for its own pair count (rounds 1-4)

    def raster_forward(i):
        view = (i * world + rank) % len(settings)
        state["campos"] = settings[view].campos
        for t in list(params.values()) + [means2D]:
            t.grad = None
        rast = GaussianRasterizer(settings[view])
        return view, rast(means3D=params["means3D"], means2D=means2D, opacities=params["opacity"], shs=params["sh"],
                          features=params.get("features"), scales=params["scales"], rotations=params["rotations"])

    def raster_backward(view, res):
        contrib, color, feature, radii, allmap = res
        outs, grads = [color, allmap], [g_color, g_others]
        if S > 0:
            outs.append(feature)
            grads.append(g_feat)
        torch.autograd.backward(outs, grads)

    def raster_exchange(view):
        if world > 1:
            state["reduced"] = reducer.reduce([params[k].grad for k in params] + [means2D.grad], params["means3D"], settings[view].campos, 3)

    forward_fn, backward_fn, exchange_fn = (surfel_forward, surfel_backward, surfel_exchange_and_update) if surfel_mode else \
        (raster_forward, raster_backward, raster_exchange)

    def step(i):
        """One view, forward and backward, then the exchange and the optimizer step.
        ONE rank: forward AND backward inside ONE rasterizer.deferred_count() box -- the view's pair count, which the GPU produces only
        when it gets to this view's tile scan, i.e. after the previous view's backward, is collected AFTER the backward has been queued.
        (Rounds 3-4 collected it at the end of the forward: the host then queued the backward's ~25 nodes (0.33 ms) while the GPU ran the
        rest of the forward (0.28 ms) and the GPU idled whenever the host was the slower of the two.)  An overflow of the guessed
        workspace (its render is the EMPTY render, its gradients zeros: include/mrgs.h) surfaces at box.finish(); the view is then done
        again, exactly sized -- before anything was exchanged or stepped.
        SEVERAL ranks: the backward's hooks start collectives (the early all-gathers of the SH factors), which every rank must enter
        exactly once per step.  The count is therefore collected between forward and backward -- an overflowing rank renders its forward
        again ALONE (no collective is inside a forward), then all ranks run backward + exchange in step.  That costs the deferral's gain
        (<= 6 % on a slow host) where the exchange is exposed anyway, and keeps the rank sequence matched by construction."""
        ta = time.perf_counter()
        if _SYNC_COUNT:
            view, res = forward_fn(i)
            backward_fn(view, res)
        elif world > 1:
            box = rasterizer_mod.deferred_count()
            with box:
                view, res = forward_fn(i)
            try:
                box.finish()
            except rasterizer_mod.RasterWorkspaceOverflow:
                view, res = forward_fn(i)
            backward_fn(view, res)
        else:
            box = rasterizer_mod.deferred_count()
            with box:
                view, res = forward_fn(i)
                tb = time.perf_counter()
                backward_fn(view, res)
            try:
                box.finish()
            except rasterizer_mod.RasterWorkspaceOverflow:
                view, res = forward_fn(i)
                backward_fn(view, res)
            if _PHASES is not None:
                _PHASES.append((ta, tb, time.perf_counter()))
        exchange_fn(view)
        state["R"] = rasterizer_mod.LAST_NUM_RENDERED

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    # Host-loop hygiene: after `import torch` the interpreter tracks ~170k container objects, and a full cyclic-GC pass over them
    # costs ~40 ms; the per-view Python glue allocates enough containers to trigger one every few dozen views.  Moving the
    # start-up objects to the permanent generation keeps collections proportional to what a view allocates.  (Before the warm-up, not
    # between it and the timed region: 40 ms of idle GPU in front of the first timed step lets the clocks fall back.)
    # One process drives one GPU: autograd's per-device worker thread buys nothing here.  The backward of a full render is ~25
    # Python-level nodes (0.6 ms of host work per view); on the worker thread they run on whatever core the scheduler woke it on, and
    # the view, which is paced by the host, moves with that (C3full, identical runs on one box: 802 ... 1 009 views/s with the worker
    # thread, 1 058 ... 1 073 with the backward on the calling thread).  The raw rasterizer's backward is ONE node and measures the
    # same either way (C2: 1 916 ... 1 957), so torch's default stays there.  The line a training script adds once;
    # MRGS_BENCH_MT_AUTOGRAD=1 restores torch's default for A/B runs.
    if surfel_mode and not os.environ.get("MRGS_BENCH_MT_AUTOGRAD"):
        torch.autograd.set_multithreading_enabled(False)
    import gc
    gc.collect()
    gc.freeze()
    # ... and the automatic collector stays off until the timed region is over: with nothing left to find, a generation-1/2 pass still
    # costs ~0.12 ms of host time every ~25 views (it sat on step 19 of every 20-step run).  A training loop would collect between epochs.
    if not os.environ.get("MRGS_BENCH_GC_ON"):
        gc.disable()
    # (the warm-up runs with the same sampled event pairs as the timed region: the first timing events on a stream cost the runtime
    # a one-time set-up that belongs in front of the measurement)
    L.mrgs_set_profiling(0 if os.environ.get("MRGS_BENCH_NO_KERNEL_EVENTS") else 3)
    for i in range(args.warmup):
        step(i)
    fence()
    # HIP events around the dominant kernel (backward blend, feeds `roofline`) on every fourth launch of the timed region: an event
    # pair costs two ~6 us bubbles on the stream, which every-launch timing would charge to the throughput figure
    L.mrgs_set_profiling(0 if os.environ.get("MRGS_BENCH_NO_KERNEL_EVENTS") else 3)
    mem0 = (torch.cuda.memory_allocated(dev), torch.cuda.memory_reserved(dev))
    wait0 = rasterizer_mod.COUNT_WAIT_SECONDS
    t0 = time.perf_counter()
    step_marks = [] if os.environ.get("MRGS_BENCH_STEP_TIMES") else None     # developer diagnostic: host time per step of the timed region
    host_prof = None
    if os.environ.get("MRGS_BENCH_CPROFILE"):        # developer diagnostic: where the host's time per step goes (the figures of this run are then not results)
        import cProfile
        host_prof = cProfile.Profile()
        host_prof.enable()
    for i in range(args.steps):
        step(args.warmup + i)
        if step_marks is not None:
            step_marks.append(time.perf_counter())
    if host_prof is not None:
        host_prof.disable()
        import pstats
        with open(os.environ["MRGS_BENCH_CPROFILE"], "w") as f_:
            st_ = pstats.Stats(host_prof, stream=f_)
            st_.sort_stats("tottime").print_stats(70)
            st_.sort_stats("cumulative").print_stats(70)
    issued = time.perf_counter() - t0        # the host is done queueing
    # ... of which it spent this long blocked on the GPU (every view waits once for its pair count, which the GPU produces after the
    # previous view's backward): what is left is the host's own work per step, the figure that says whether Python paces the step
    waited = rasterizer_mod.COUNT_WAIT_SECONDS - wait0
    host_work = issued - waited
    fence()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if step_marks is not None and rank == 0:
        mem1 = (torch.cuda.memory_allocated(dev), torch.cuda.memory_reserved(dev))
        print("device memory over the timed region (allocated, reserved) MiB: %.1f, %.1f -> %.1f, %.1f" %
              (mem0[0] / 2**20, mem0[1] / 2**20, mem1[0] / 2**20, mem1[1] / 2**20), file=sys.stderr)
        d = [1e3 * (b - a) for a, b in zip([t0] + step_marks[:-1], step_marks)]
        order = sorted(range(len(d)), key=lambda k: -d[k])[:6]
        if _PHASES:
            ph = _PHASES[-len(d):]
            med = lambda v: sorted(v)[len(v) // 2]
            print("host phases (median us): forward incl. the wait for the count %.0f, backward %.0f, between steps %.0f" %
                  (1e6 * med([b - a for a, b, c in ph]), 1e6 * med([c - b for a, b, c in ph]),
                   1e6 * med([n[0] - p[2] for p, n in zip(ph[:-1], ph[1:])])), file=sys.stderr)
        print("first 40 steps: " + " ".join("%.3f" % v for v in d[:40]), file=sys.stderr)
        print("host ms per step by position: " + " ".join("%d-%d: %.3f" % (a, min(a + 20, len(d)), sum(d[a:a + 20]) / len(d[a:a + 20])) for a in range(0, len(d), 20)), file=sys.stderr)
        print("host ms per step: median %.3f, longest %s, tail after the last step %.3f ms" %
              (sorted(d)[len(d) // 2], [(k, round(d[k], 2)) for k in order], 1e3 * (elapsed - (step_marks[-1] - t0))), file=sys.stderr)
    if os.environ.get("MRGS_BENCH_TORCH_PROFILE") and rank == 0:
        # developer diagnostic: which torch operators (fills, copies, accumulations) a step still launches, with their call sites
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof_:
            for i in range(4):
                step(args.warmup + args.steps + 200 + i)
            fence()
        with open(os.environ["MRGS_BENCH_TORCH_PROFILE"], "w") as f_:
            f_.write(prof_.key_averages(group_by_stack_n=6).table(sort_by="cuda_time_total", row_limit=60, max_name_column_width=60, max_src_column_width=110))
    times = MrgsKernelTimes()
    L.mrgs_get_kernel_times(times)
    # per-stage breakdown (diagnostic `stage_ms`): a few extra, untimed steps with an event pair around every stage
    L.mrgs_set_profiling(1)
    for i in range(min(args.steps, 10)):
        step(args.warmup + args.steps + i)
    fence()
    stage_times = MrgsKernelTimes()
    L.mrgs_get_kernel_times(stage_times)
    L.mrgs_set_profiling(0)

    # surfel workloads: torch events around the parts of a view and, with the tracer, around its two native calls (they run on torch's
    # current stream); a few extra untimed steps
    trace_ms = {}
    if surfel_mode:
        tr_ev = {"surfel_trace_fwd": [], "surfel_trace_bwd": []}
        if traced:
            from materialrefgs_amd import surfel_tracing as st_mod
            orig_f, orig_b = st_mod._Trace.forward, st_mod._Trace.backward

            def timed(name, fn):
                def wrapper(ctx, *a):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    r = fn(ctx, *a)
                    e1.record()
                    tr_ev[name].append((e0, e1))
                    return r
                return staticmethod(wrapper)
            st_mod._Trace.forward, st_mod._Trace.backward = timed("surfel_trace_fwd", orig_f), timed("surfel_trace_bwd", orig_b)
        state["parts_on"] = True
        for i in range(min(args.steps, 8)):
            step(args.warmup + args.steps + 16 + i)
        fence()
        state["parts_on"] = False
        if traced:
            st_mod._Trace.forward, st_mod._Trace.backward = staticmethod(orig_f), staticmethod(orig_b)
            ls = hw_tracer.tracer.last_state                       # per ray: sum w t^2, final T, hits blended, passes
            state["trace_hits"] = int(ls[:, 2].sum().item())
        for name, evs in list(parts.items()) + list(tr_ev.items()):
            if evs:
                trace_ms[name] = sum(a.elapsed_time(b) for a, b in evs) / len(evs)

    # First visits: a camera rendered for the first time has no work hint (the forward's queues come from the cull counts) and its
    # backward orders by itself -- every camera once per training run, and again after each densification.  Timed per step between
    # fences (so is the warm figure next to it: a fence per step takes the host's launch latency out of hiding, which `ms_per_step` of
    # the back-to-back timed region does not pay).
    def fenced_ms(first_index, n):
        total = 0.0
        for i in range(n):
            fence()
            t_ = time.perf_counter()
            step(first_index + i)
            fence()
            total += time.perf_counter() - t_
        return 1000.0 * total / n
    n_cam = len(settings)
    base_i = ((args.warmup + args.steps + 64) // n_cam + 1) * n_cam
    warm_fenced = fenced_ms(base_i, n_cam)
    # ... and the first visit after a densification / pruning step, the normal visit of the reference's first 25 000-30 000 iterations
    # (every 100 iterations the surfel set changes, and with ~100 training cameras each camera is then seen about once per set): the
    # camera's measured work is still there (the hint cache is keyed by the camera, not by P), only the deal of its blend waves is redone
    rasterizer_mod.note_surfel_set_changed()
    densified_fenced = fenced_ms(base_i, n_cam)
    rasterizer_mod.reset_work_hints()
    cold_fenced = fenced_ms(base_i, n_cam)
    for i in range(2 * n_cam):                                  # leave the hints warm again (second visit: the deal from measured work)
        step(base_i + i)
    fence()

    if args.dump_grads:
        step(args.dump_step)
        fence()
        if rank == 0:
            import numpy as np
            if surfel_mode:
                dump = {n: t_.grad.detach().cpu().numpy() for n, t_ in zip(surfel_names, surfel_params)}
            elif world > 1:
                dump = {n: t_.detach().cpu().numpy() for n, t_ in zip(grad_names, state["reduced"])}
            else:
                dump = {n: t_.grad.detach().cpu().numpy() for n, t_ in zip(grad_names, list(params.values()) + [means2D])}
            np.savez(args.dump_grads, **dump)

    # What one view-parallel step puts on the links and what every rank computes locally for it (materialrefgs_amd/dist.py), as a model
    # for V = 8 ranks: bytes from the tensor shapes, the time of the local SH expansion measured here by feeding it 8 gathered rows.
    def exchange_model(V=8):
        if surfel_mode:
            dense_floats = sum(int(t_.numel()) for t_ in surfel_params)
            sh_floats = sum(int(t_.numel()) for n, t_ in zip(surfel_names, surfel_params) if n in mdist.SurfelGradReducer.SH_NAMES)
            row = 6 * P + 3
            g_rgb, g_ind = torch.randn(V, 3 * P + 3, device=dev), torch.randn(V, 3 * P, device=dev)
            fn = lambda: mdist.expand_surfel_sh_gradient_rows(g_rgb, g_ind, pc._xyz, pc._rotation, 3, family="rgb")
            fn_b = lambda: mdist.expand_surfel_sh_gradient_rows(g_rgb, g_ind, pc._xyz, pc._rotation, 3, family="ind")
        else:
            dense_floats = sum(int(v.numel()) for v in params.values()) + int(means2D.numel())
            sh_floats = int(params["sh"].numel())
            row = 3 * P + 3
            gathered = torch.randn(V, row, device=dev)
            fn = lambda: mdist.expand_sh_gradients(gathered, params["means3D"], 16, 3)
        def timed(f):
            f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                f()
            e1.record()
            torch.cuda.synchronize(dev)
            return e0.elapsed_time(e1) / 10
        expand_ms = timed(fn)
        ag_row, ar_bytes = 4 * row, 4 * (dense_floats - sh_floats)
        m = {"V": V, "allgather_bytes_per_rank_sent": ag_row, "allgather_bytes_per_rank_received": ag_row * V,
             "allreduce_bytes": ar_bytes, "dense_allreduce_bytes_avoided": 4 * dense_floats,
             "floats_per_gaussian_on_the_wire": round((row - 3 + dense_floats - sh_floats) / P, 2),
             "floats_per_gaussian_dense": round(dense_floats / P, 2), "sh_expand_ms_at_V": round(expand_ms, 4),
             "_ag_row": ag_row, "_ar_bytes": ar_bytes, "_expand_ms": expand_ms}
        if surfel_mode:
            # the two factors travel apart (dist.SurfelGradReducer.begin_early_rgb / _ind) and are expanded apart; the per-gaussian glue's
            # backward (the kernel between the two hand-outs) is timed here on a graph of its own
            import materialrefgs_amd.renderer as renderer_mod
            expand_b_ms = timed(fn_b)
            o_ = renderer_mod.surfel_features(pc, cams_dev[0].camera_center, pass_xyz=True)
            gs_ = [torch.ones_like(t_) for t_ in o_]
            leaves_ = [t_ for t_ in surfel_params[:11]]
            feat_bwd_ms = timed(lambda: torch.autograd.grad(o_, leaves_, gs_, retain_graph=True, allow_unused=True))
            m.update({"allgather_rows_bytes": [4 * (3 * P + 3), 4 * 3 * P], "sh_expand_ms_at_V": [round(expand_ms, 4), round(expand_b_ms, 4)],
                      "surfel_features_bwd_ms": round(feat_bwd_ms, 4), "_expand_b_ms": expand_b_ms, "_feat_bwd_ms": feat_bwd_ms})
            # Which rows a view touches, measured: the dense part of the exchange travels as the UNION of the rows the V ranks' views touched
            # (dist.TouchedRowsExchange).  One backward per orbit camera here; rank r of a V-rank step renders camera (step V + r) mod 8, i.e.
            # V neighbouring cameras of the orbit.
            masks = []
            for v_ in range(len(cams_dev)):
                for t_ in surfel_params:
                    t_.grad = None
                env.build_mips()
                o2 = render_surfel(cams_dev[v_], pc, pipe, bg_color, srgb=False, opt=SimpleNamespace(indirect=indirect), flag=flavour)
                outs2 = [o2[k_] for k_ in ("render", "rend_alpha", "rend_normal", "rend_dist", "surf_depth", "surf_normal")]
                torch.autograd.backward(outs2, state.get("g") or ([torch.ones_like(outs2[0])] + [torch.full_like(o_x, 0.1) for o_x in outs2[1:]]))
                rows_ = torch.cat([t_.grad.reshape(P, -1) for n_, t_ in zip(surfel_names, surfel_params)
                                   if n_ not in mdist.SurfelGradReducer.SH_NAMES and t_.shape[0] == P], dim=1)
                masks.append((rows_ != 0).any(dim=1))
            n_cam_ = len(masks)
            row_floats = int(rows_.shape[1])
            tail_floats = dense_floats - sh_floats - row_floats * P
            union = {}
            for V_ in (2, 4, 8):
                if n_cam_ >= V_:
                    fr = [float(torch.stack(masks[s_:s_ + V_]).any(dim=0).float().mean()) for s_ in range(0, n_cam_ - V_ + 1, V_)]
                    union[f"V{V_}"] = round(sum(fr) / len(fr), 4)
            m.update({"touched_rows_fraction_per_view": round(float(torch.stack(masks).float().mean()), 4), "union_rows_fraction": union,
                      "dense_row_floats": row_floats, "dense_tail_floats": tail_floats, "mask_bytes_per_rank": P,
                      "indirect_factor_gathered": bool(indirect), "_union": union, "_row_floats": row_floats, "_tail_floats": tail_floats})
        return m

    def predicted_scaling(xm, step_ms, overlap_ms):
        """Step efficiency of the view-parallel step at V = 2 / 4 / 8 from the wire bytes and the measured local kernels (no multi-GPU
        box was available to any round: a MODEL, every input of which is printed).  Links: the 8 GPUs of a node are fully connected, 7
        xGMI links per GPU at ~153 GB/s each, counted as 76.8 GB/s per direction and derated to 80 %.  Both collectives go directly
        between peers (every rank sends its all-gather row to each peer over that peer's link; the all-reduce as reduce-scatter +
        all-gather of 1/V slices), so one link carries row + 2 x allreduce / V bytes per step and direction; ~10 us of latency per
        collective phase.  The SH expansion needs the gathered rows (its time is measured here for 8 rows and scaled by V / 8: it is one
        pass over V rows); the all-gather is issued as soon as the blend backward has produced the colour gradients
        (rasterizer.set_after_blend_hook), i.e. `overlap_ms` of it (the per-gaussian backward that follows) is hidden."""
        link = 76.8e9 * 0.8
        tab = {}
        for V in (2, 4, 8):
            t_ar = 2.0 * xm["_ar_bytes"] / V / link * 1e3 + 0.020
            ar_bytes_V = xm["_ar_bytes"]
            if xm.get("_union", {}).get(f"V{V}") is not None:
                # the dense part as the union of the touched rows (dist.TouchedRowsExchange): the measured union's share of the per-surfel rows
                # + the tail (the cubemap) through reduce-scatter + all-gather, in front of it the all-gather of the masks (P bytes per rank,
                # one more collective phase) -- and one host read of the union's size, which the model does not price
                ar_bytes_V = 4.0 * (xm["_union"][f"V{V}"] * P * xm["_row_floats"] + xm["_tail_floats"])
                t_ar = 2.0 * ar_bytes_V / V / link * 1e3 + 0.020 + (P / link * 1e3 + 0.010)
            if "_expand_b_ms" in xm:
                # render_surfel's exchange (dist.SurfelGradReducer): the colour factor's gather starts `overlap_ms` before the backward ends
                # (after the blend backward: the rasterizer's per-gaussian backward and the glue's backward follow), the indirect factor's
                # gather at its end; links: gather 1 (what is left of it), gather 2, then reduce-scatter + all-gather of the dense bucket;
                # compute stream: expansion 1 when gather 1 has landed, expansion 2 when gather 2 has landed and expansion 1 is done
                t_ag1, t_ag2 = 4 * (3 * P + 3) / link * 1e3 + 0.010, 4 * 3 * P / link * 1e3 + 0.010
                t_e1, t_e2 = xm["_expand_ms"] * V / 8.0, xm["_expand_b_ms"] * V / 8.0
                if not xm.get("indirect_factor_gathered", True):
                    # render_surfel without opt.indirect: the indirect factor is zero by the structure of the step (renderer.surfel_features,
                    # indirect_live) -- not gathered, not expanded (dist.SurfelGradReducer.begin_early_ind(None))
                    t_ag2, t_e2 = 0.0, 0.0
                left1 = max(t_ag1 - overlap_ms, 0.0)
                links_done = left1 + t_ag2 + t_ar
                compute_done = max(left1 + t_e1, left1 + t_ag2) + t_e2
                exposed, t_ag, t_exp = max(links_done, compute_done), t_ag1 + t_ag2, t_e1 + t_e2
            else:
                t_ag = xm["_ag_row"] / link * 1e3 + 0.010
                t_exp = xm["_expand_ms"] * V / 8.0
                # the all-gather starts `overlap_ms` before the backward ends; the all-reduce follows it on the same links; the expansion
                # runs on the compute stream as soon as the gather has landed, next to the all-reduce
                exposed = max(t_ag - overlap_ms, 0.0) + max(t_ar, t_exp)
            eff = step_ms / (step_ms + exposed)
            tab[f"V{V}"] = {"allgather_ms": round(t_ag, 4), "allreduce_ms": round(t_ar, 4), "dense_bytes_exchanged": int(ar_bytes_V), "sh_expand_ms": round(t_exp, 4),
                            "exposed_exchange_ms": round(exposed, 4), "efficiency": round(eff, 3), "speedup": round(V * eff, 2)}
        return {"link_GBs_per_direction_derated": round(link / 1e9, 1), "single_gpu_step_ms": round(step_ms, 4),
                "allgather_overlapped_with_preprocess_bwd_ms": round(overlap_ms, 4), **tab}
    # (the hooks of the view-parallel step come off first: the model below runs a per-gaussian backward on rank 0 ALONE, and a hook that
    #  starts a collective there would wait for the other ranks forever)
    rasterizer_mod.set_after_blend_hook(None)
    if surfel_mode:
        import materialrefgs_amd.renderer as _renderer_mod
        _renderer_mod.set_after_features_hook(None)
    xmodel = exchange_model() if rank == 0 and torch.cuda.is_available() else None

    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tmax.item())
    views = args.steps * world
    value = views / elapsed

    out = {
        "metric": "full-render fwd+bwd views/sec at 800x800/300k surfels; grad max-rel-err vs ref",
        "value": round(value, 3), "unit": "views/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1000.0 * elapsed / args.steps, 4), "timed_region_s": round(elapsed, 4),
        "host_issue_ms_per_step": round(1000.0 * issued / args.steps, 4),
        "host_wait_ms_per_step": round(1000.0 * waited / args.steps, 4), "host_work_ms_per_step": round(1000.0 * host_work / args.steps, 4),
        "cold_ms_per_step": round(cold_fenced, 4), "after_densify_ms_per_step": round(densified_fenced, 4),
        "warm_ms_per_step_fenced": round(warm_fenced, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": desc, "P": P, "H": H, "W": W, "S": S, "sh_degree": 3, "num_rendered": int(state["R"]),
                   "views_per_step": world, "cameras_cycled": n_views, "parallelism": f"view-parallel x{world}" if world > 1 else "single GPU"},
    }
    if surfel_mode:
        # (the rasterizer's per-gaussian backward carries on through the glue's backward in one kernel where nothing reads the blended
        #  indirect radiance -- MrgsRasterGrads::glue_params; MRGS_NO_GLUE_EPILOGUE=1 restores the two-kernel backward)
        import materialrefgs_amd.renderer as _rm
        out["config"]["glue_epilogue"] = bool(_rm._FUSE_GLUE and not indirect and not getattr(pipe, "use_asg", False))
    if rank == 0:
        R, HW = int(state["R"]), H * W
        stage_ms = {"preprocess_fwd": stage_times.preprocess_ms, "depth_sort_scan": stage_times.sort_ms,
                    "duplicate_tilesort_ranges": stage_times.duplicate_ms, "render_fwd": stage_times.render_fwd_ms,
                    "render_bwd": times.render_bwd_ms, "preprocess_bwd": stage_times.preprocess_bwd_ms}
        stage_ms.update(trace_ms)
        # the backward blend is the dominant kernel of the raster / shaded workloads (stage_ms confirms it); its duration comes from
        # the events of the timed region.  With the tracer the two launches of its forward walk are longer (one "launch" = the mirror
        # rays of one view); units there: rays, blended hits, surfels.
        dom = "render_bwd"
        dom_ms = stage_ms[dom]
        nbytes = algorithmic_bytes(dom, P, R, HW, S)
        if traced and trace_ms.get("surfel_trace_fwd", 0.0) > dom_ms:
            dom, dom_ms = "surfel_trace_fwd", trace_ms["surfel_trace_fwd"]
            nbytes = algorithmic_bytes(dom, P, state["trace_hits"], HW, S)
        achieved = nbytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        # HBM traffic (FETCH_SIZE / WRITE_SIZE passes) and the VALU issue rate (SQ pass) of the dominant kernel come from separate
        # rocprofv3 --pmc runs of this same command (tools/pmc_traffic.py, tools/pmc_sq.py), which stamp their JSON with the digest of
        # the kernel sources they measured.  A figure whose digest differs from the sources of THIS run is stale and reported as null.
        digest = kernel_source_digest()
        traffic, traffic_src = None, None
        pmc = _load_json(os.path.join(ROOT, "profiles", "pmc_traffic.json"))
        if pmc.get("kernel_source_digest") == digest:
            traffic = pmc.get(args.workload, {}).get(dom)
            traffic_src = pmc.get("measured_by")
        out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                           "algorithmic_bytes_per_launch": int(nbytes), "avg_launch_ms": round(dom_ms, 4),
                           "kernel_source_digest": digest, "traffic_source": traffic_src}
        # The dominant kernel is fp32 VALU work (no matrix shape in it) and sits at the VALU issue rate, not at the HBM rate the
        # required fields above price it against: the SQ counter pass says how close.
        sqj = _load_json(os.path.join(ROOT, "profiles", "pmc_sq.json"))
        sq = sqj.get(args.workload, {}).get(dom) if sqj.get("kernel_source_digest") == digest else None
        out["roofline"]["valu_issue"] = None if not sq else {
            "cycles_per_inst_per_simd": sq["cycles_per_valu_inst_per_simd"], "peak": 4.0, "frac": round(4.0 / sq["cycles_per_valu_inst_per_simd"], 3),
            "insts_per_launch": sq["valu_insts_per_launch"], "source": sqj.get("measured_by", "profiles/pmc_sq.json")}
        out["stage_ms"] = {k: round(v, 4) for k, v in stage_ms.items()}
        if xmodel is not None:
            if world == 1:
                # (render_surfel: the colour factor leaves after the blend backward -- the rasterizer's per-gaussian backward and the glue's
                #  backward run under its gather --, the indirect factor after the glue's backward; not with the traced term, see above)
                ov = float(stage_ms.get("preprocess_bwd", 0.0))
                if surfel_mode:
                    ov = 0.0 if traced else ov + float(xmodel.get("_feat_bwd_ms", 0.0))
                xmodel["predicted_scaling"] = predicted_scaling(xmodel, 1000.0 * elapsed / args.steps, ov)
                # the model's bottom line at top level (the driver's record keeps top-level keys): speed-up of V ranks over one
                out["predicted_scaling_model"] = {k: v["speedup"] for k, v in xmodel["predicted_scaling"].items() if k.startswith("V")}
                out["predicted_scaling_model"]["note"] = ("MODEL, not a measurement (no multi-GPU box was available to any round): wire bytes of "
                                                          "materialrefgs_amd/dist.py over 7 point-to-point xGMI links at 61 GB/s per direction, local "
                                                          "kernels measured here; inputs under exchange_model")
            xmodel = {k: v for k, v in xmodel.items() if not k.startswith("_")}
        out["exchange_model"] = xmodel
        out["host_bound"] = bool(host_work > 0.9 * elapsed)
        if out["host_bound"]:
            # the host's own work (queueing, not waiting for the GPU) filled (nearly) the whole timed region: the figure is then a host figure
            print(f"bench.py: WARNING host_work_ms_per_step {1000.0 * host_work / args.steps:.4f} > 0.9 x ms_per_step {1000.0 * elapsed / args.steps:.4f}: "
                  "the step is paced by the host's launches (Python), not by the GPU", file=sys.stderr)

        if world > 1:
            pass                         # the CPU baseline and the oracle comparison belong to the N = 1 run only
        elif not args.no_cpu_baseline and surfel_mode and (traced or use_loss or indirect or flavour != "2dgs"):
            out["cpu_baseline"] = None   # CPU legs exist for the raster workloads and for C3full
        elif not args.no_cpu_baseline and surfel_mode:
            # render_surfel of view 0 on the host cores through the checkers (oracle/render_oracle.py: per-gaussian glue in torch float64,
            # the rasterizer in oracle/mrgs_oracle.c over OpenMP, map post-processing + split-sum shading + compositing in torch float64),
            # forward and backward with the bench's upstream gradients.  The environment prefilter is NOT in the CPU figure: its checker is
            # a dense (6 N^2)^2 float64 operator that does not exist at N = 128; the CPU leg shades with the levels the HIP prefilter
            # produced.  The same pass gives the gradient parity of this very configuration (`grad_max_rel_err`).
            from oracle import render_oracle
            from oracle import raster_oracle as ro
            import numpy as np
            import materialrefgs_amd.renderer as renderer_mod
            cam0 = cams[0]
            for t_ in surfel_params:
                t_.grad = None
            env.build_mips()
            # Both rasterizers get the SAME per-gaussian inputs: the product's own fp32 activations / feature rows (captured from inside
            # render_surfel).  Evaluated in float64 on the CPU they differ from the fp32 kernel's in the last bit, and one ulp of an
            # opacity moves a handful of the image's 640 000 pixels across the alpha = 1/255 and T = 1e-4 tests -- a comparison of
            # inputs, not of renderers.  The gradient that arrives at those inputs is then pulled back through the checker's float64
            # glue to the raw leaves, and the gradient of the prefiltered levels through the float64 prefilter operators to the texels
            # (oracle/render_oracle.surfel_leaf_gradients): EVERY leaf of the headline workload is compared, xyz and viewspace_points
            # (what densification reads, backward.cu:665-668), the raw material parameters and env.base included.
            stash, glue = {}, renderer_mod.surfel_features
            def capturing(pc_, campos_, **kw):
                o = glue(pc_, campos_, **kw)
                for t_ in o[:4]:
                    t_.retain_grad()
                stash["o"] = o[:4]
                return o
            renderer_mod.surfel_features = capturing
            try:
                out_h = render_surfel(cams_dev[0], pc, pipe, bg_color, srgb=False, opt=SimpleNamespace(indirect=False))
            finally:
                renderer_mod.surfel_features = glue
            keys = ["render", "rend_alpha", "rend_normal", "rend_dist", "surf_depth", "surf_normal"]
            torch.autograd.backward([out_h[k] for k in keys], state["g"])
            torch.cuda.synchronize(dev)
            R_hip = int(rasterizer_mod.LAST_NUM_RENDERED)
            # (the two prefilter levels above 32^2 are applied blocked in float64 on the GPU, as a calculator -- 98 304^2 weights do not
            #  fit a host; they are outside the CPU figure, as is the glue)
            # (shaded with the product's own levels on both sides; the prefilter is compared apart, `env_level_*`: at roughness 0.08 the
            #  reference's fp32 filter weights are ill-conditioned and levels built in fp32 and in float64 differ by per cents)
            levels_h = [m_.detach().cpu().double().numpy() for m_ in env.specular]
            out_o, g_o, info = render_oracle.surfel_leaf_gradients(cam0, surfel_params[:11], env.base, stash["o"], keys, state["g"], pipe, bg_color,
                                                                   env_min_res=env.min_res, mips_device=dev, shade_levels=levels_h, literal32_prefilter=True)
            cpu_s = info["raster_shading_seconds"]
            out["cpu_baseline"] = {"value": round(1.0 / cpu_s, 5), "unit": "views/s", "cores": ro.num_threads(), "kind": "port",
                                   "sample": f"1 view fwd+bwd of the same workload ({args.workload}, view 0) through oracle/render_oracle.py: oracle/mrgs_oracle.c "
                                             f"rasterizer (OpenMP) + torch float64 map post-processing, split-sum shading and compositing; per-gaussian glue "
                                             f"and environment prefilter excluded ({cpu_s:.1f} s)"}
            hip = {n: t_.grad.detach().cpu().numpy() for n, t_ in zip(render_oracle.LEAF_NAMES, surfel_params[:11])}
            hip["env_base"] = env.base.grad.detach().cpu().numpy()
            hip["viewspace_points"] = out_h["viewspace_points"].grad.detach().cpu().numpy()
            # (the rasterizer's per-gaussian inputs have gradient tensors of their own only when the glue's backward is a kernel of its own: with
            #  the glue epilogue -- the default for this workload -- the chain runs from the gradient rows to the raw leaves inside one kernel)
            mid = [n for n, t_ in zip(render_oracle.RASTER_INPUT_NAMES, stash["o"]) if t_.grad is not None]
            for n, t_ in zip(render_oracle.RASTER_INPUT_NAMES, stash["o"]):
                if t_.grad is not None:
                    hip[n] = t_.grad.detach().cpu().numpy()
            names = list(render_oracle.LEAF_NAMES) + ["env_base", "viewspace_points"] + mid
            rows, ok = render_oracle.leaf_gradient_report(hip, g_o, names, bar=1e-4)
            maps = {}
            for k in keys + ["specular_map", "diffuse_map", "roughness_map", "base_color_map", "refl_strength_map"]:
                a, b = out_h[k].detach().cpu().double().numpy(), out_o[k].detach().numpy()
                maps[k] = float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
            lrows, lok = render_oracle.level_report(levels_h, info["levels"], info["levels_lit32"])
            out["env_level_rel_err"] = {f"{r_['res']}^2": (float(f"{r_['err']:.3e}") if r_["rule"] == "bar" else
                                                         f"{r_['err']:.3e} {r_['rule']} (fp32 filter weights {r_['lit32_err']:.2e})") for r_ in lrows}
            ok = ok and lok
            out["grad_max_rel_err"] = round(max(r_["err"] for r_ in rows.values()), 8)
            out["grad_rel_err"] = {k: float(f"{v['err']:.3e}") for k, v in rows.items()}
            out["grad_rule"] = {k: (v["rule"] if "lit32_err" not in v else f"{v['rule']} (fp32 torch glue {v['lit32_err']:.2e})") for k, v in rows.items()
                                if v["rule"] != "bar"}
            out["grad_all_leaves_within_1e-4_or_truth_leg"] = bool(ok)
            out["grad_rel_err_note"] = ("every leaf of render_surfel (raw xyz / scaling / rotation / opacity / material parameters, both SH families, "
                                        "env_base) + viewspace_points (+ the rasterizer's per-gaussian inputs where they exist as tensors: not with the glue epilogue), through glue + rasterizer + maps + prefilter "
                                        "+ shading + compositing; max-norm relative to the tensor's largest element; identical fp32 rasterizer inputs on both sides")
            out["map_rel_err"] = {k: float(f"{v:.3e}") for k, v in maps.items()}
            # two maps are ill-conditioned functions of what the rasterizer blends, in the reference's formulas as much as here: rend_dist
            # (m^2 A + M2 - 2 m M1, forward.cu:412: O(1) terms cancel to a value of order 1e-5; the tests hold it to an ABSOLUTE 5e-6) and
            # surf_normal (normalised cross product of finite differences of neighbouring surface points, utils/point_utils.py:26-39)
            out["rend_dist_abs_err"] = float(f"{np.abs(out_h['rend_dist'].detach().cpu().double().numpy() - out_o['rend_dist'].detach().numpy()).max():.3e}")
            sn_h, sn_o = out_h["surf_normal"].detach().cpu().double().numpy(), out_o["surf_normal"].detach().numpy()
            out["surf_normal_frac_pixels_over_1e-4"] = float(f"{(np.abs(sn_h - sn_o).max(axis=0) > 1e-4).mean():.3e}")
            out["num_rendered_matches_oracle"] = bool(render_oracle.LAST_NUM_RENDERED == R_hip)
        elif not args.no_cpu_baseline:
            from oracle import raster_oracle as ro
            cam = cams[0]
            t = time.perf_counter()
            orc = ro.render_scene(scene_cpu, cam)
            go = orc.backward(g_color.cpu(), g_feat.cpu(), g_others.cpu())
            cpu_s = time.perf_counter() - t
            out["cpu_baseline"] = {"value": round(1.0 / cpu_s, 5), "unit": "views/s", "cores": ro.num_threads(), "kind": "port",
                                   "sample": f"1 view fwd+bwd of the same workload ({args.workload}, view 0) through oracle/mrgs_oracle.c "
                                             f"(OpenMP, {cpu_s:.1f} s)"}
            # gradient parity of this very configuration against the oracle (view 0)
            for t_ in list(params.values()) + [means2D]:
                t_.grad = None
            rast = GaussianRasterizer(settings[0])
            contrib, color, feature, radii, allmap = rast(
                means3D=params["means3D"], means2D=means2D, opacities=params["opacity"], shs=params["sh"],
                features=params.get("features"), scales=params["scales"], rotations=params["rotations"])
            outs, grads = [color, allmap], [g_color, g_others]
            if S > 0:
                outs.append(feature)
                grads.append(g_feat)
            torch.autograd.backward(outs, grads)
            torch.cuda.synchronize(dev)
            import numpy as np
            errs = {}
            for k, ko in [("means3D", "means3D"), ("opacity", "opacity"), ("scales", "scales"), ("rotations", "rotations"),
                          ("sh", "sh"), ("features", "features")]:
                if k in params:
                    a = params[k].grad.detach().cpu().numpy().astype(np.float64).reshape(go[ko].shape)
                    b = go[ko].astype(np.float64)
                    errs[k] = float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
            a = means2D.grad.detach().cpu().numpy().astype(np.float64)
            errs["means2D"] = float(np.abs(a - go["means2D"]).max() / max(np.abs(go["means2D"]).max(), 1e-30))
            out["grad_max_rel_err"] = round(max(errs.values()), 8)
            out["grad_rel_err"] = {k: float(f"{v:.3e}") for k, v in errs.items()}
            out["num_rendered_matches_oracle"] = bool(orc.R == color.grad_fn.num_rendered)
            orc.close()
            # truth leg (tests/test_truth_leg.py at this size): distance of the HIP gradients, of the oracle with the kernels' FMA pattern
            # and of the literal un-fused fp32 reading from the SAME formulas evaluated in float64 -- max-norm and, over the elements
            # above 1e-3 max|g|, the median / 99th percentile of the per-element relative error
            from oracle import compare
            hip = {k: params[k].grad.detach().cpu().numpy() for k in params}
            hip["means2D"] = means2D.grad.detach().cpu().numpy()
            legs = {}
            for v in ("lit32", "f64"):
                o = ro.render_scene(scene_cpu, cam, variant=v)
                legs[v] = o.backward(g_color.cpu(), g_feat.cpu(), g_others.cpu())
                o.close()
            out["grad_err_vs_f64"] = compare.three_way(hip, go, legs["lit32"], legs["f64"])
            out["hip_no_further_from_f64_than_literal_fp32"] = bool(all(
                r["hip"]["max_norm"] <= max(1.5 * r["literal32"]["max_norm"], 2e-5) for r in out["grad_err_vs_f64"].values()))
            # the north star's pure-PyTorch CPU alpha-blend baseline (forward only; BASELINE.json configs[0] and a down-scaled C2)
            # (a child process under a timeout, a handful of threads: the blend is thousands of small tensor ops, and a thread per
            # visible core on a box whose container may own far fewer makes every one of them wait for descheduled threads)
            import subprocess
            try:
                avail = len(os.sched_getaffinity(0))
            except Exception:
                avail = os.cpu_count() or 1
            tb_threads = max(1, min(8, avail))
            try:
                r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "torch_blend.py"), str(tb_threads)], capture_output=True, text=True,
                                   timeout=180)
                tb = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
                out["cpu_baseline_torch"] = {"kind": "port", "what": "forward-only alpha blend, pure torch tensor ops (oracle/torch_blend.py)",
                                             "threads": tb["threads"], "runs": tb["runs"],
                                             "value": round(1.0 / tb["runs"]["C2/10 (P=30000, 256x256)"]["seconds"], 4),
                                             "unit": "forward views/s at C2/10"}
            except Exception as ex:      # incl. subprocess.TimeoutExpired
                out["cpu_baseline_torch"] = {"error": type(ex).__name__}
        if world == 1 and args.workload == "C3full" and not args.no_secondary:
            # secondary lines, each a fresh child process after everything here is done: the rasterizer alone (C2 = BASELINE.json configs[1],
            # with its own CPU leg, oracle gradient comparison and truth leg), the heavier raster scene (R ~ 6 P, heavy-tailed splat
            # sizes), the last training stage (render_surfel + surfel-traced mirror rays, C3trace) and BASELINE.json configs[3] with the
            # traced reflection term (C4trace: 1 M surfels, 1600 x 1600)
            import subprocess

            def child(workload, steps, warmup, cpu_leg=False):
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--workload", workload, "--steps", str(steps), "--warmup", str(warmup),
                                    "--no-secondary"] + ([] if cpu_leg else ["--no-cpu-baseline"]), capture_output=True, text=True)
                for line in r.stdout.splitlines():
                    if line.startswith("{"):
                        j = json.loads(line)
                        d = {"workload": j["config"]["workload"], "value": j["value"], "unit": j["unit"], "ms_per_step": j["ms_per_step"],
                             "host_issue_ms_per_step": j.get("host_issue_ms_per_step"), "host_work_ms_per_step": j.get("host_work_ms_per_step"),
                             "cold_ms_per_step": j.get("cold_ms_per_step"), "after_densify_ms_per_step": j.get("after_densify_ms_per_step"),
                             "warm_ms_per_step_fenced": j.get("warm_ms_per_step_fenced"),
                             "steps": j["steps"], "warmup": j["warmup"], "num_rendered": j["config"]["num_rendered"], "stage_ms": j["stage_ms"],
                             "exchange_model": j.get("exchange_model"),
                             "roofline": {k: j["roofline"].get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic",
                                                                              "algorithmic_bytes_per_launch", "avg_launch_ms", "valu_issue")},
                             "roofline_frac": j["roofline"]["frac"]}
                        if cpu_leg:
                            for k in ("cpu_baseline", "grad_max_rel_err", "grad_rel_err", "num_rendered_matches_oracle", "grad_err_vs_f64",
                                      "hip_no_further_from_f64_than_literal_fp32", "cpu_baseline_torch"):
                                d[k] = j.get(k)
                        return d
                return {"error": (r.stderr or r.stdout)[-300:]}
            out["secondary_raster"] = child("C2", max(args.steps, 200), min(args.warmup, 50), cpu_leg=not args.no_cpu_baseline)
            out["secondary"] = child("C2heavy", min(max(args.steps, 100), 500), min(args.warmup, 30))
            out["secondary_traced"] = child("C3trace", 100, 16)
            out["secondary_c4"] = child("C4trace", 30, 8)
            out["secondary_pgsr"] = child("C3full-pgsr", 200, 16)        # the flavour the reference ships (arguments/config.py:1)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
