"""The loop the reference actually drives (train_refnerf.py:1093-1480): render -> loss -> backward -> Adam, with a prune + clone of the
surfel set every 100 iterations (densify_and_prune, scene/gaussian_model.py:1043; densification_interval = 100,
arguments/__init__.py:159-162), over a handful of cameras -- closed, for 300 iterations, on a synthetic target.  Every piece has its
own parity test; this one is about what only the loop shows: the loss falls, no workspace overflow escapes the render functions when the
surfel count jumps, and the per-camera caches (work hints, workspace guesses, camera copies) neither grow nor go cold when P changes."""
from types import SimpleNamespace

import pytest
import torch

from materialrefgs_amd.synthetic import make_surfel_model, orbit_camera

pytestmark = pytest.mark.gpu

GROUPS = [("xyz", "_xyz"), ("f_dc", "_features_dc"), ("f_rest", "_features_rest"), ("opacity", "_opacity"), ("scaling", "_scaling"),
          ("rotation", "_rotation"), ("refl_strength", "_refl_strength"), ("roughness", "_roughness"), ("ori_color", "_ori_color"),
          ("ind_dc", "_indirect_dc"), ("ind_rest", "_indirect_rest")]          # names of GaussianModel.training_setup (gaussian_model.py:422-443)


def _rebind(pc, optimizable):
    for name, attr in GROUPS:
        setattr(pc, attr, optimizable[name])


def test_three_hundred_iterations_with_densification(gpu_device):
    from materialrefgs_amd import densify, losses
    from materialrefgs_amd import rasterizer as rz
    from materialrefgs_amd.optim import Adam
    from materialrefgs_amd.renderer import render_surfel
    dev = gpu_device
    torch.manual_seed(0)
    P, H, W, n_cam = 20000, 256, 256, 6
    pipe = SimpleNamespace(depth_ratio=0.0, debug=False, compute_cov3D_python=False, convert_SHs_python=False)
    bg = torch.zeros(3, device=dev)
    opt_r = SimpleNamespace(indirect=False)
    cams = [orbit_camera(v, H, W, n_views=n_cam).to(dev) for v in range(n_cam)]

    # the target: renders of the scene itself; the model: the same geometry with its colours, materials and environment scrambled
    pc, env, _ = make_surfel_model(P, max(H, W), dev, seed=0, radius_px=6.0, env_res=32, env_min=8)
    targets = []
    with torch.no_grad():
        env.build_mips()
        for c in cams:
            gt = render_surfel(c, pc, pipe, bg, srgb=False, opt=opt_r)["render"].clone()
            targets.append(SimpleNamespace(original_image=gt, image_weight=losses.image_weight(gt)))
        g = torch.Generator().manual_seed(3)
        pc._features_dc.add_(0.6 * torch.randn(pc._features_dc.shape, generator=g).to(dev))
        pc._ori_color.add_(1.0 * torch.randn(pc._ori_color.shape, generator=g).to(dev))
        pc._refl_strength.add_(1.0 * torch.randn(pc._refl_strength.shape, generator=g).to(dev))
        env.base.add_(0.8 * torch.randn(env.base.shape, generator=g).to(dev))
    rates = {"xyz": 1.6e-5, "f_dc": 2.5e-3, "f_rest": 1.25e-4, "opacity": 0.05, "scaling": 5e-3, "rotation": 1e-3, "refl_strength": 0.01,
             "roughness": 0.01, "ori_color": 0.01, "ind_dc": 2.5e-3, "ind_rest": 1.25e-4}
    groups = [{"params": [torch.nn.Parameter(getattr(pc, attr).detach().clone().requires_grad_(True))], "lr": rates[name], "name": name} for name, attr in GROUPS]
    groups.append({"params": [env.base], "lr": 0.01, "name": "env"})
    optimizer = Adam(groups, lr=0.0, eps=1e-15)
    _rebind(pc, {gr["name"]: gr["params"][0] for gr in optimizer.param_groups})
    loss_opt = SimpleNamespace(lambda_dssim=0.2, lambda_normal_render_depth=0.05, normal_loss_start=0, lambda_dist=100.0, dist_loss_start=100,
                               lambda_normal_smooth=0.0, lambda_depth_smooth=0.0, normal_smooth_from_iter=0, normal_smooth_until_iter=0,
                               use_perceptual_loss=False)

    rz.reset_work_hints()
    rz._PAIR_GUESS.clear()
    history, counts, prepared_after_densify = [], [P], []
    overflow_before = 0
    for it in range(1, 301):
        v = it % n_cam
        env.build_mips()
        out = render_surfel(cams[v], pc, pipe, bg, srgb=False, opt=opt_r)           # (a workspace overflow inside is redone inside: nothing escapes)
        loss, tb = losses.calculate_loss(targets[v], pc, out, loss_opt, it, targets[v].image_weight, None)
        loss.backward()
        assert out["viewspace_points"].grad is not None and out["visibility_filter"].shape[0] == counts[-1]
        optimizer.step()
        optimizer.zero_grad(set_to_none=True)
        history.append(float(loss))
        if it % 100 == 0 and it < 300:
            # densify_and_prune in miniature: the 12 % most transparent surfels go, the 20 % with the largest view-space gradient of this view
            # are cloned a little to the side (gaussian_model.py:1001-1057); optimizer surgery through materialrefgs_amd.densify
            n = pc._xyz.shape[0]
            with torch.no_grad():
                keep = torch.ones(n, dtype=torch.bool, device=dev)
                keep[torch.argsort(pc._opacity.reshape(-1))[: int(0.12 * n)]] = False
            optimizable, _ = densify.prune_optimizer(optimizer, keep)
            _rebind(pc, optimizable)
            n2 = pc._xyz.shape[0]
            with torch.no_grad():
                pick = torch.randperm(n2, device=dev)[: int(0.2 * n2)]
                ext = {name: getattr(pc, attr).detach()[pick].clone() for name, attr in GROUPS}
                ext["xyz"] += 0.002 * torch.randn_like(ext["xyz"])
            _rebind(pc, densify.cat_tensors_to_optimizer(optimizer, ext))
            counts.append(pc._xyz.shape[0])
            assert counts[-1] != counts[-2]
            # the first render of every camera after the step is WARM: its hint (image-space) survived the change of P
            assert all(rz._hint_is_warm(_settings_of(rz, c, dev), dev) for c in cams)
    first, last = sum(history[:20]) / 20, sum(history[-20:]) / 20
    print(f"closed loop: loss {first:.4f} -> {last:.4f}; surfel counts {counts}; hints {len(rz._WORK_HINTS)}, guesses {len(rz._PAIR_GUESS)}")
    assert all(map(lambda x: x == x and x < 1e3, history))                        # finite throughout
    assert last < 0.75 * first, (first, last)
    # caches: one hint per camera whatever P did, a workspace guess per surfel count (bounded), one contiguous copy per camera matrix
    assert len(rz._WORK_HINTS) == n_cam
    assert len(rz._PAIR_GUESS) <= min(rz._PAIR_GUESS_MAX, len(counts) + 1)
    assert len(rz._CAM_COPIES) <= 3 * n_cam


def _settings_of(rz, cam, dev):
    """The (contiguous fp32) matrices the rasterizer keys a camera's hint by: what renderer.render_surfel hands it for this Camera."""
    return SimpleNamespace(image_height=cam.image_height, image_width=cam.image_width, viewmatrix=rz._camera_f32c(cam.world_view_transform),
                           projmatrix=rz._camera_f32c(cam.full_proj_transform))
