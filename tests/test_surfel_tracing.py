"""Surfel ray tracer (second half; reference call site gaussian_renderer/optix_utils.py:36-271).

CPU: the dense oracle against closed forms (one surfel, two layers, quad extent = get_disks' corners, termination, ordering), the
host mirror's shapes.  GPU (-m gpu): the HIP tracer -- hierarchy built on the device -- against the dense oracle: outputs, per-surfel
weights, and every gradient against the oracle's autograd in float64.
"""
import math
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import surfel_trace_oracle as sto  # noqa: E402
from materialrefgs_amd.synthetic import make_shell_scene  # noqa: E402


def one_surfel(dtype=torch.float64, scale=(0.2, 0.1), opacity=0.8, mean=(0.0, 0.0, 2.0)):
    return dict(means=torch.tensor([mean], dtype=dtype), scales=torch.tensor([scale], dtype=dtype),
                rotations=torch.tensor([[1.0, 0.0, 0.0, 0.0]], dtype=dtype), opacities=torch.tensor([[opacity]], dtype=dtype),
                colors=torch.tensor([[0.9, 0.5, 0.1]], dtype=dtype), others=torch.tensor([[0.3, 0.7]], dtype=dtype))


BG = torch.tensor([0.2, 0.3, 0.4], dtype=torch.float64)


def test_one_surfel_closed_form():
    """A surfel in the plane z = 2 with axes x, y; rays from the origin.  Direction (x, y, 1) hits at t = 2 (the ray parameter, not
    the distance), local coordinates (2x / s_u, 2y / s_v)."""
    s = one_surfel()
    d = torch.tensor([[0.0, 0.0, 1.0], [0.05, 0.02, 1.0], [0.0, 0.0, 2.0], [0.0, 0.0, -1.0]], dtype=torch.float64)
    o = torch.zeros_like(d)
    out = sto.trace_dense(o, d, bg=BG, **s)
    a0 = 0.8
    u, v = 2 * 0.05 / 0.2, 2 * 0.02 / 0.1
    a1 = 0.8 * math.exp(-0.5 * (u * u + v * v))
    assert torch.allclose(out["acc"], torch.tensor([a0, a1, a0, 0.0], dtype=torch.float64), atol=1e-12)
    assert torch.allclose(out["dpt"], torch.tensor([2 * a0, 2 * a1, 1 * a0, 0.0], dtype=torch.float64), atol=1e-12)     # d = (0,0,2): t = 1
    assert torch.allclose(out["rgb"][0], a0 * s["colors"][0] + (1 - a0) * BG, atol=1e-12)
    assert torch.allclose(out["rgb"][3], BG, atol=0)                                                                     # behind the origin
    assert torch.allclose(out["aux"][1], a1 * s["others"][0], atol=1e-12)
    # the normal is turned against the ray: the surfel's +z normal seen from below gives -z
    assert torch.allclose(out["norm"][0], torch.tensor([0.0, 0.0, -a0], dtype=torch.float64), atol=1e-12)
    assert torch.allclose(out["dist"], torch.zeros(4, dtype=torch.float64), atol=1e-15)
    assert torch.allclose(out["wet"], torch.tensor([a0 + a1 + a0], dtype=torch.float64), atol=1e-12)


def test_quad_extent_is_the_three_sigma_square_of_get_disks():
    """optix_utils.py:44: corners at +-3 in the splat's (u, v).  Inside the square the hit counts (alpha permitting), outside not."""
    s = one_surfel(opacity=1.0)
    v = sto.quad_vertices(s["means"], s["scales"], s["rotations"])
    assert torch.allclose(v[0], torch.tensor([[-0.6, 0.3, 2.0], [-0.6, -0.3, 2.0], [0.6, 0.3, 2.0], [0.6, -0.3, 2.0]], dtype=torch.float64))
    # u = 2.9 -> alpha = exp(-4.205) = 0.0149 > 1/255; u = 3.1 outside the quad although exp(-4.805) = 0.0082 > 1/255
    d = torch.tensor([[0.5 * 2.9 * 0.2, 0.0, 1.0], [0.5 * 3.1 * 0.2, 0.0, 1.0], [0.0, 0.5 * 3.1 * 0.1, 1.0]], dtype=torch.float64)
    out = sto.trace_dense(torch.zeros_like(d), d, bg=BG, **s)
    assert out["hits"].tolist() == [1, 0, 0]
    assert abs(float(out["acc"][0]) - math.exp(-0.5 * 2.9 ** 2)) < 1e-12


def test_two_layers_order_distortion_and_termination():
    dt = torch.float64
    mk = lambda z, o: (torch.tensor([0.0, 0.0, z], dtype=dt), o)
    layers = [mk(3.0, 0.6), mk(2.0, 0.5)]                      # given far first: the order must come from t
    s = dict(means=torch.stack([l[0] for l in layers]), scales=torch.full((2, 2), 0.5, dtype=dt),
             rotations=torch.tensor([[1.0, 0, 0, 0]] * 2, dtype=dt), opacities=torch.tensor([[l[1]] for l in layers], dtype=dt),
             colors=torch.tensor([[1.0, 0, 0], [0, 1.0, 0]], dtype=dt), others=torch.zeros(2, 2, dtype=dt))
    d = torch.tensor([[0.0, 0.0, 1.0]], dtype=dt)
    out = sto.trace_dense(torch.zeros_like(d), d, bg=BG, **s)
    w_near, w_far = 0.5, 0.5 * 0.6
    assert torch.allclose(out["rgb"][0], torch.tensor([w_far, w_near, 0.0], dtype=dt) + 0.5 * 0.4 * BG, atol=1e-12)
    assert abs(float(out["dpt"][0]) - (2 * w_near + 3 * w_far)) < 1e-12
    assert abs(float(out["dist"][0]) - w_near * w_far * 1.0) < 1e-12            # w1 w2 (t1 - t2)^2
    assert torch.allclose(out["wet"], torch.tensor([w_far, w_near], dtype=dt), atol=1e-12)
    # termination (the rasterizer's rule, forward.cu:395-399): alphas 0.99, 0.9, 0.99, 0.99 -> T = 0.01, 0.001, and the third hit would
    # give 1e-5 < 1e-4: it is not blended and nothing behind it either
    s3 = dict(means=torch.tensor([[0, 0, 1.0 + k] for k in range(4)], dtype=dt), scales=torch.full((4, 2), 0.5, dtype=dt),
              rotations=torch.tensor([[1.0, 0, 0, 0]] * 4, dtype=dt), opacities=torch.tensor([[1.0], [0.9], [1.0], [1.0]], dtype=dt),
              colors=torch.ones(4, 3, dtype=dt), others=torch.zeros(4, 2, dtype=dt))
    out = sto.trace_dense(torch.zeros_like(d), d, bg=BG, **s3)
    assert int(out["hits"][0]) == 2 and abs(float(out["T"][0]) - 0.001) < 1e-15
    assert torch.allclose(out["wet"], torch.tensor([0.99, 0.009, 0.0, 0.0], dtype=dt), atol=1e-15)
    # equal depths: index order
    s2 = dict(s)
    s2["means"] = torch.tensor([[0, 0, 2.0], [0, 0, 2.0]], dtype=dt)
    out = sto.trace_dense(torch.zeros_like(d), d, bg=BG, **s2)
    assert torch.allclose(out["wet"], torch.tensor([0.6, 0.4 * 0.5], dtype=dt), atol=1e-12)


def test_distortion_equals_the_pairwise_form_and_autograd_runs():
    torch.manual_seed(0)
    sc = make_shell_scene(60, seed=3, radius_px=60.0, image_size=200)
    f = lambda t: t.double().requires_grad_(True)
    means, scales, rots, opac = f(sc.means3D), f(sc.scales), f(sc.rotations), f(sc.opacities)
    colors, others = f(torch.rand(60, 3)), f(torch.rand(60, 2))
    o = torch.tensor([[0.0, 0.0, 0.0]], dtype=torch.float64).repeat(64, 1)
    d = torch.nn.functional.normalize(torch.randn(64, 3, dtype=torch.float64), dim=1) * 1.7
    out = sto.trace_dense(o, d, means, scales, rots, opac, colors, others, BG)
    assert int(out["hits"].max()) >= 2
    # A M2 - M1^2 with the sums taken over the blended hits
    t = sto.brute_force_hits(o, d, means, scales, rots, opac)
    assert torch.isfinite(t).sum() >= int(out["hits"].sum())
    loss = out["rgb"].sum() + out["dist"].sum() + (out["norm"] ** 2).sum() + out["dpt"].sum()
    loss.backward()
    assert all(torch.isfinite(x.grad).all() for x in (means, scales, rots, opac, colors))


def test_host_mirror_shapes_without_a_gpu():
    from materialrefgs_amd.surfel_tracing import SurfelTracer, SurfelTracingSettings, surfel_records, MID_CHANNELS
    import diff_surfel_tracing
    assert diff_surfel_tracing.SurfelTracer is SurfelTracer and diff_surfel_tracing.SurfelTracingSettings is SurfelTracingSettings
    sc = make_shell_scene(10, seed=1)
    geom, attr = surfel_records(sc.means3D, sc.scales, sc.rotations, sc.opacities, torch.rand(10, 3), torch.rand(10, 2))
    assert geom.shape == (10, 16) and attr.shape == (10, 8) and MID_CHANNELS == 16
    R = sto.rotation_matrix(sc.rotations)
    assert torch.allclose(geom[:, 9:12], R[:, :, 2], atol=1e-6) and torch.allclose(geom[:, 3:6] * sc.scales[:, 0:1], R[:, :, 0], atol=1e-5)
    with pytest.raises(RuntimeError):
        SurfelTracer().build_acceleration_structure(torch.zeros(8, 3), None)          # no CPU path


# ---- GPU ----------------------------------------------------------------------------------------------------------------------

def _scene(P, seed, radius_px=40.0):
    sc = make_shell_scene(P, seed=seed, radius_px=radius_px, image_size=400)
    g = torch.Generator().manual_seed(seed)
    return sc, torch.rand(P, 3, generator=g), torch.rand(P, 2, generator=g)


def _rays(n, seed, inside=True):
    g = torch.Generator().manual_seed(100 + seed)
    if inside:      # from points inside the shell outwards: every ray crosses the shell once
        o = torch.randn(n, 3, generator=g) * 0.2
    else:           # from outside through the whole shell: two crossings, many layers
        o = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=1) * 3.0
    tgt = torch.randn(n, 3, generator=g) * 0.3
    d = (tgt - o) if not inside else torch.randn(n, 3, generator=g)
    d = d / d.norm(dim=1, keepdim=True) * (0.5 + torch.rand(n, 1, generator=g))       # not normalised on purpose
    return o, d


def _hip_trace(dev, sc, colors, others, o, d, bg, need_grad=False, scale_modifier=1.0):
    from materialrefgs_amd.surfel_tracing import SurfelTracer, SurfelTracingSettings
    leaf = lambda t: t.to(dev).clone().requires_grad_(need_grad)
    L = dict(means=leaf(sc.means3D), scales=leaf(sc.scales), rotations=leaf(sc.rotations), opacities=leaf(sc.opacities), colors=leaf(colors),
             others=leaf(others), o=leaf(o), d=leaf(d))
    tr = SurfelTracer()
    v = sto.quad_vertices(L["means"].detach(), L["scales"].detach(), L["rotations"].detach(), scale_modifier).reshape(-1, 3)
    tr.build_acceleration_structure(v, None)
    eye = torch.eye(4, device=dev)
    ts = SurfelTracingSettings(1, o.shape[0], 1.0, 1.0, bg.to(dev).float(), scale_modifier, eye, eye, 0, torch.zeros(3, device=dev), False, False)
    rgb, dpt, acc, norm, dist, aux, mid, wet = tr(L["o"], L["d"], v, means3D=L["means"], grads3D=None, shs=None, colors_precomp=L["colors"],
                                                   others_precomp=L["others"], opacities=L["opacities"], scales=L["scales"],
                                                   rotations=L["rotations"], cov3D_precomp=None, tracer_settings=ts)
    return dict(rgb=rgb, dpt=dpt[..., 0], acc=acc[..., 0], norm=norm, dist=dist[..., 0], aux=aux, wet=wet[:, 0], mid=mid), L


def _oracle(sc, colors, others, o, d, bg, dtype, need_grad=False, scale_modifier=1.0):
    leaf = lambda t: t.to(dtype).clone().requires_grad_(need_grad)
    L = dict(means=leaf(sc.means3D), scales=leaf(sc.scales), rotations=leaf(sc.rotations), opacities=leaf(sc.opacities), colors=leaf(colors),
             others=leaf(others), o=leaf(o), d=leaf(d))
    out = sto.trace_dense(L["o"], L["d"], L["means"], L["scales"], L["rotations"], L["opacities"], L["colors"], L["others"], bg.to(dtype),
                          scale_modifier)
    return out, L


@pytest.mark.gpu
@pytest.mark.parametrize("P,n,inside,radius", [(1, 64, True, 400.0), (3, 256, True, 300.0), (5, 256, False, 200.0), (900, 2048, True, 40.0),
                                               (4000, 4096, False, 30.0), (20000, 4096, False, 12.0)])
def test_hip_tracer_matches_the_dense_oracle(gpu_device, P, n, inside, radius):
    sc, colors, others = _scene(P, P, radius)
    o, d = _rays(n, P, inside)
    bg = torch.tensor([0.1, 0.5, 0.9])
    hip, _ = _hip_trace(gpu_device, sc, colors, others, o, d, bg)
    ref, _ = _oracle(sc, colors, others, o, d, bg, torch.float64)
    # a ray whose fp32 decision at a threshold (quad edge, 1/255, T < 1e-4) differs from float64 may gain / lose one hit: allowed
    # for a handful of rays, everything else agrees to fp32 rounding
    bad = torch.zeros(n, dtype=torch.bool)
    for k in ("rgb", "dpt", "acc", "norm", "dist", "aux"):
        a, b = hip[k].cpu().double(), ref[k]
        err = (a - b).abs().reshape(n, -1).max(dim=1).values
        bad |= ~(err <= 2e-4 * max(1.0, float(b.abs().max())))           # NaN counts as a mismatch
    assert int(bad.sum()) <= max(1, n // 500), (int(bad.sum()), n)
    assert float(ref["acc"].max()) > 0.5 or P < 10                      # the scene is actually hit
    ok = ~bad
    assert float((hip["acc"].cpu().double() - ref["acc"])[ok].abs().max()) < 5e-5
    wet_err = (hip["wet"].cpu().double() - ref["wet"]).abs()
    assert float(wet_err.max()) <= 1e-4 * max(1.0, float(ref["wet"].max())) + 1.0 * int(bad.sum())
    assert hip["mid"].shape == (n, 16)


@pytest.mark.gpu
def test_hip_tracer_many_layers_need_several_passes(gpu_device):
    """60 faint coplanar-stacked layers in front of every ray: more hits than one 16-entry pass holds, in scrambled index order."""
    P, n = 60, 512
    g = torch.Generator().manual_seed(5)
    z = 1.0 + torch.randperm(P, generator=g).float() * 0.05
    sc_means = torch.stack([torch.zeros(P), torch.zeros(P), z], dim=1)
    from types import SimpleNamespace
    sc = SimpleNamespace(means3D=sc_means, scales=torch.full((P, 2), 2.0), rotations=torch.tensor([[1.0, 0, 0, 0]]).repeat(P, 1),
                         opacities=torch.full((P, 1), 0.05))
    colors, others = torch.rand(P, 3, generator=g), torch.rand(P, 2, generator=g)
    o = torch.zeros(n, 3)
    d = torch.cat([torch.randn(n, 2, generator=g) * 0.3, torch.ones(n, 1)], dim=1)
    bg = torch.zeros(3)
    hip, _ = _hip_trace(gpu_device, sc, colors, others, o, d, bg)
    ref, _ = _oracle(sc, colors, others, o, d, bg, torch.float64)
    assert int(ref["hits"].max()) > 48
    for k in ("rgb", "dpt", "acc", "norm", "dist", "aux"):
        assert float((hip[k].cpu().double() - ref[k]).abs().max()) < 2e-5 * max(1.0, float(ref[k].abs().max())), k
    assert float((hip["wet"].cpu().double() - ref["wet"]).abs().max()) < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("P,n,inside,radius", [(3, 128, True, 300.0), (600, 1024, True, 40.0), (3000, 1024, False, 30.0)])
def test_hip_tracer_gradients_match_autograd_of_the_oracle(gpu_device, P, n, inside, radius):
    sc, colors, others = _scene(P, 7 + P, radius)
    o, d = _rays(n, 7 + P, inside)
    bg = torch.tensor([0.3, 0.2, 0.6])
    g = torch.Generator().manual_seed(11)
    up = dict(rgb=torch.randn(n, 3, generator=g), dpt=torch.randn(n, generator=g), acc=torch.randn(n, generator=g),
              norm=torch.randn(n, 3, generator=g), dist=torch.randn(n, generator=g), aux=torch.randn(n, 2, generator=g))
    hip, Lh = _hip_trace(gpu_device, sc, colors, others, o, d, bg, need_grad=True)
    ref, Lr = _oracle(sc, colors, others, o, d, bg, torch.float64, need_grad=True)
    # rays on which the two disagree about a threshold are taken out of BOTH losses
    same = torch.ones(n, dtype=torch.bool)
    for k in up:
        err = (hip[k].detach().cpu().double() - ref[k].detach()).abs().reshape(n, -1).max(dim=1).values
        same &= err <= 2e-4 * max(1.0, float(ref[k].detach().abs().max()))
    assert int((~same).sum()) <= max(1, n // 200)
    m = same.double()
    loss_h = sum((hip[k] * (up[k] * m.float().reshape(n, *([1] * (up[k].dim() - 1)))).to(gpu_device)).sum() for k in up)
    loss_r = sum((ref[k] * (up[k].double() * m.reshape(n, *([1] * (up[k].dim() - 1))))).sum() for k in up)
    loss_h.backward()
    loss_r.backward()
    for k in ("means", "scales", "rotations", "opacities", "colors", "others", "o", "d"):
        a, b = Lh[k].grad.cpu().double(), Lr[k].grad
        scale = max(float(b.abs().max()), 1e-12)
        assert float((a - b).abs().max()) <= 3e-4 * scale, (k, float((a - b).abs().max()) / scale)


@pytest.mark.gpu
def test_hardware_rendering_mirror(gpu_device):
    """render_gaussians of the HardwareRendering mirror: dictionary keys and shapes of optix_utils.py:218-233 on a small scene."""
    from types import SimpleNamespace
    from materialrefgs_amd.surfel_tracing import HardwareRendering
    from materialrefgs_amd.renderer import SurfelModel
    from materialrefgs_amd.synthetic import orbit_camera
    dev = gpu_device
    sc = make_shell_scene(2000, seed=2, radius_px=25.0, image_size=64).to(dev)
    pc = SurfelModel(sc.means3D.clone().requires_grad_(True), torch.log(sc.scales), sc.rotations, torch.logit(sc.opacities),
                     sc.shs[:, :1].contiguous(), sc.shs[:, 1:].contiguous())
    H = W = 64
    cam = orbit_camera(0, H, W).to(dev)
    hr = HardwareRendering().train()
    g = torch.Generator().manual_seed(3)
    ray_o = (torch.randn(H, W, 3, generator=g) * 0.1).to(dev)
    ray_d = torch.nn.functional.normalize(torch.randn(H, W, 3, generator=g), dim=-1).to(dev)
    out = hr.render_gaussians(cam, ray_o, ray_d, pc, SimpleNamespace(compute_cov3D_python=False, convert_SHs_python=False),
                              torch.zeros(3, device=dev))
    for k, c in (("render", 3), ("rend_alpha", 1), ("rend_normal", 3), ("rend_dist", 1), ("surf_depth", 1), ("surf_normal", 3), ("specular", 1),
                 ("roughness", 1)):
        assert out[k].shape == (c, H, W), k
    assert out["weight_accumulate"].shape == (2000, 1) and out["visibility_filter"].shape == (2000,)
    v, f = hr.get_disks(pc)                                   # optix_utils.py:36-66: corners and the two triangles per surfel
    ref_v = sto.quad_vertices(pc.get_xyz.detach().cpu().double(), pc.get_scaling.detach().cpu().double(), pc._rotation.detach().cpu().double())
    assert v.shape == (8000, 3) and f.shape == (4000, 3) and float((v.detach().cpu().double() - ref_v.reshape(-1, 3)).abs().max()) < 1e-5
    assert f[:2].tolist() == [[0, 1, 2], [1, 2, 3]] and f.dtype == torch.int32
    assert float(out["rend_alpha"].mean()) > 0.3
    import glue_oracle
    sn_ref = glue_oracle.depth_to_normal(cam, out["surf_depth"].detach()).permute(2, 0, 1) * out["rend_alpha"].detach()
    assert float((out["surf_normal"].detach() - sn_ref).abs().max()) < 2e-4              # optix_utils.py:236-242 through the maps kernel
    # every surfel carries others = 0.01 (optix_utils.py:173-177): specular = 0.01 * alpha
    assert torch.allclose(out["specular"], 0.01 * out["rend_alpha"], atol=1e-6)
    out["render"].sum().backward()
    assert out["viewspace_points"].grad is not None and torch.allclose(out["viewspace_points"].grad, pc._xyz.grad)


@pytest.mark.gpu
def test_render_surfel_with_envgs_composes_raster_and_traced_light(gpu_device):
    """gaussian_renderer/__init__.py:486-520: the traced dictionary is what the dense oracle gives for the mirror rays of the rendered
    view, the final image is the documented blend, and the gradient reaches the model through both paths."""
    from types import SimpleNamespace
    from test_render_e2e import _models
    from materialrefgs_amd import renderer
    from materialrefgs_amd.surfel_tracing import HardwareRendering
    from materialrefgs_amd.synthetic import orbit_camera
    dev = gpu_device
    P, H, W = 1500, 40, 56
    _, _, pc, env = _models(P, H, W, seed=4, dev=dev)
    cam = orbit_camera(2, H, W).to(dev)
    pipe = SimpleNamespace(depth_ratio=0.0, debug=False, compute_cov3D_python=False, convert_SHs_python=False)
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    opt = SimpleNamespace(indirect=False)
    hr = HardwareRendering().train()
    out = renderer.render_surfel_with_envgs(hr, cam, pc, pipe, bg, srgb=False, opt=opt)
    base = renderer.render_surfel(cam, pc, pipe, bg, srgb=False, opt=opt)
    tr = out["indirect_out"]
    spec = tr["specular"]
    assert torch.allclose(out["render"], base["render"] * (1 - spec) + spec * tr["render"], atol=1e-6)
    # the traced part against the dense oracle on the same rays
    alpha = base["rend_alpha"].permute(1, 2, 0)
    nmap = renderer.safe_normalize(base["rend_normal"].permute(1, 2, 0) / alpha.clamp_min(1e-6))
    ro, rd = renderer._mirror_rays(cam, nmap, base["surf_depth"])
    dirs = pc.get_xyz - cam.camera_center.reshape(1, 3)
    from materialrefgs_amd.gs_utils import eval_sh
    colors = torch.clamp_min(eval_sh(pc.active_sh_degree, pc.get_features.transpose(1, 2), dirs / dirs.norm(dim=1, keepdim=True)) + 0.5, 0.0)
    f = lambda t: t.detach().cpu().double()
    ref = sto.trace_dense(f(ro).reshape(-1, 3), f(rd).reshape(-1, 3), f(pc.get_xyz), f(pc.get_scaling), f(pc.get_rotation), f(pc.get_opacity),
                          f(colors), torch.full((P, 2), 0.01, dtype=torch.float64), f(bg))
    err = (f(tr["render"]).permute(1, 2, 0).reshape(-1, 3) - ref["rgb"]).abs().max(dim=1).values
    assert int((err > 2e-4).sum()) <= 3 and float(ref["acc"].mean()) > 0.05
    same = renderer.render_indirect(hr, cam, pc, pipe, bg, normal_map=nmap, surf_depth=base["surf_depth"])
    assert torch.allclose(same["render"], tr["render"], atol=1e-6)
    out["render"].sum().backward()
    assert pc._xyz.grad is not None and float(pc._xyz.grad.abs().sum()) > 0
    assert tr["viewspace_points"].grad is not None and float(tr["viewspace_points"].grad.abs().sum()) > 0


@pytest.mark.gpu
def test_render_surfel2_wiring(gpu_device):
    """gaussian_renderer/envgs_renderer.py:461-715 on the HIP pieces: (1) with the "2dgs" channel set and no indirect term it is
    render_surfel; (2) the two extra "pgsr" channels are the rasterized blend weight and plane distance (checked against a separate
    rasterization of each); (3) with opt.indirect the traced light of the ENVIRONMENT surfels stands where the mesh occludes the
    environment map, and its gradient reaches the environment surfels."""
    from types import SimpleNamespace
    from test_render_e2e import _models
    from materialrefgs_amd import renderer
    from materialrefgs_amd.rasterizer import GaussianRasterizer
    from materialrefgs_amd.raytracing import RayTracer
    from materialrefgs_amd.surfel_tracing import HardwareRendering
    from materialrefgs_amd.synthetic import orbit_camera, make_occluder_mesh
    dev = gpu_device
    P, H, W = 1500, 40, 56
    _, _, pc, _env = _models(P, H, W, seed=6, dev=dev)
    _, _, envgs, _ = _models(700, H, W, seed=8, dev=dev)                       # the second surfel set (EnvGaussianModel's role)
    with torch.no_grad():
        envgs._xyz.mul_(2.5)                                                   # a shell around the object
        pc._metalness.copy_(torch.randn(P, 1, device=dev))
    pc._metalness.requires_grad_(True)
    cam = orbit_camera(3, H, W).to(dev)
    pipe = SimpleNamespace(depth_ratio=0.0, debug=False, compute_cov3D_python=False, convert_SHs_python=False, use_asg=False)
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    hr = HardwareRendering().train()
    plain = renderer.render_surfel(cam, pc, pipe, bg, srgb=True, opt=SimpleNamespace(indirect=False))
    two = renderer.render_surfel2(hr, envgs, cam, pc, pipe, bg, srgb=True, opt=SimpleNamespace(indirect=False), flag="2dgs")
    for k in ("render", "specular_map", "diffuse_map", "base_color_map", "rend_alpha", "rend_normal", "surf_depth", "surf_normal", "rend_dist"):
        assert float((two[k] - plain[k]).abs().max()) < 2e-6, k
    assert two["blend_weight"].shape[0] == 0
    pg = renderer.render_surfel2(hr, envgs, cam, pc, pipe, bg, srgb=False, opt=SimpleNamespace(indirect=False))
    rast = GaussianRasterizer(raster_settings=renderer._raster_settings(cam, pc, pipe, torch.zeros_like(bg), 1.0))
    args = dict(means3D=pc.get_xyz, means2D=torch.zeros_like(pc.get_xyz), shs=pc.get_features, opacities=pc.get_opacity, scales=pc.get_scaling,
                rotations=pc.get_rotation)
    only_w = rast(features=pc.get_specular, **args)[2]
    only_d = rast(features=renderer.get_distance(1.0, pc.get_xyz, cam, pc), **args)[2]
    assert float((pg["blend_weight"] - only_w).abs().max()) < 2e-6 and float((pg["rend_distance"] - only_d).abs().max()) < 2e-5
    assert float(only_w.max()) > 0.05
    # the mesh occluder makes part of the mirror rays "not visible": there the traced environment surfels are the specular light
    pc.ray_tracer = RayTracer(*make_occluder_mesh(4000), device=dev)
    ind = renderer.render_surfel2(hr, envgs, cam, pc, pipe, bg, srgb=False, opt=SimpleNamespace(indirect=True))
    vis = ind["visibility"]
    assert 0.02 < float(1 - vis[ind["rend_alpha"] > 0.5].mean()) < 0.98
    assert torch.allclose(ind["indirect_light"], ind["indirect_out"]["render"], atol=1e-6)
    expect = (ind["direct_light"] * vis + (1 - vis) * ind["indirect_out"]["render"]) * ind["rend_alpha"] * ind["specular_weight"].permute(2, 0, 1)     # the reference leaves this one [H,W,3] (refl_utils.py:357)
    assert float((ind["specular_map"] - expect).abs().max()) < 1e-5
    ind["render"].sum().backward()
    assert envgs._xyz.grad is not None and float(envgs._xyz.grad.abs().sum()) > 0            # through the tracer into the environment set
    assert pc._metalness.grad is None or float(pc._metalness.grad.abs().sum()) == 0          # the blend weight is rasterized but not consumed
    assert float(pc._xyz.grad.abs().sum()) > 0
    # render_surfel_with_envgs_sep (envgs_renderer.py:771-807): render_surfel blended with the traced environment set by specular_weight
    sep = renderer.render_surfel_with_envgs_sep(hr, envgs, cam, pc, pipe, bg, srgb=False, opt=SimpleNamespace(indirect=True))
    base = renderer.render_surfel(cam, pc, pipe, bg, srgb=False, opt=SimpleNamespace(indirect=True))
    wgt = base["specular_weight"].permute(2, 0, 1)
    assert torch.allclose(sep["render"], base["render"] * (1 - wgt) + wgt * sep["indirect_out"]["render"], atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("use_sh,degree", [(True, 3), (True, 1), (False, 0)])
def test_record_kernel_matches_the_torch_statement(gpu_device, use_sh, degree):
    """mrgs_surfel_trace_prep_*: records, get_disks' corners and every gradient against surfel_records + eval_sh in torch (float64)."""
    from materialrefgs_amd.surfel_tracing import _Prep, surfel_records
    from materialrefgs_amd.gs_utils import eval_sh
    dev = gpu_device
    P = 3000
    sc = make_shell_scene(P, seed=12)
    g = torch.Generator().manual_seed(3)
    raw = dict(means=sc.means3D, scales=sc.scales, rotations=sc.rotations * 1.7, opacities=sc.opacities, shs=sc.shs + 0.3 * torch.randn(P, 16, 3, generator=g),
               colors=torch.rand(P, 3, generator=g), others=torch.rand(P, 2, generator=g))
    campos = torch.tensor([0.3, -2.0, 1.1])
    mod = 1.3
    def run(conv, hip):
        L = {k: conv(v) for k, v in raw.items()}
        cp = conv(campos).detach()
        if hip:
            geom, attr, quads = _Prep.apply(L["means"], L["scales"], L["rotations"], L["opacities"], L["shs"] if use_sh else None,
                                            None if use_sh else L["colors"], L["others"], cp, degree, mod)
        else:
            if use_sh:
                d = L["means"] - cp.reshape(1, 3)
                col = torch.clamp_min(eval_sh(degree, L["shs"].transpose(1, 2), d / d.norm(dim=1, keepdim=True)) + 0.5, 0.0)
            else:
                col = L["colors"]
            geom, attr = surfel_records(L["means"], L["scales"], L["rotations"], L["opacities"], col, L["others"], mod)
            quads = sto.quad_vertices(L["means"], L["scales"], L["rotations"], mod).reshape(-1, 3)
        return L, geom, attr, quads
    Lh, gh, ah, qh = run(lambda t: t.to(dev).float().clone().requires_grad_(True), True)
    Lr, gr, ar, qr = run(lambda t: t.double().clone().requires_grad_(True), False)
    assert float((gh.cpu().double()[:, :13] - gr[:, :13]).abs().max()) < 2e-5 * float(gr.abs().max())
    assert float((ah.cpu().double()[:, :5] - ar[:, :5]).abs().max()) < 2e-6
    assert float((qh.cpu().double() - qr).abs().max()) < 2e-6
    wg, wa = torch.randn(P, 16, generator=g), torch.randn(P, 8, generator=g)
    wg[:, 13:] = 0; wa[:, 5:] = 0
    ((gh * wg.to(dev)).sum() + (ah * wa.to(dev)).sum()).backward()
    ((gr * wg.double()).sum() + (ar * wa.double()).sum()).backward()
    for k in ("means", "scales", "rotations", "opacities", "others") + (("shs",) if use_sh else ("colors",)):
        a, b = Lh[k].grad.cpu().double(), Lr[k].grad
        assert float((a - b).abs().max()) <= 2e-5 * max(float(b.abs().max()), 1e-12), k


@pytest.mark.gpu
def test_backward_walks_again_when_the_record_overflows(gpu_device):
    """300 faint layers in front of every ray: 19 passes of 16 hits, more than a wave's record holds (15).  The forward raises the
    overflow flag and the backward walks the hierarchy again instead of replaying -- same gradients as the dense statement either way."""
    P, n = 300, 256
    g = torch.Generator().manual_seed(9)
    z = 1.0 + torch.randperm(P, generator=g).float() * 0.01
    from types import SimpleNamespace
    sc = SimpleNamespace(means3D=torch.stack([torch.zeros(P), torch.zeros(P), z], dim=1), scales=torch.full((P, 2), 2.0),
                         rotations=torch.tensor([[1.0, 0, 0, 0]]).repeat(P, 1), opacities=torch.full((P, 1), 0.02))
    colors, others = torch.rand(P, 3, generator=g), torch.rand(P, 2, generator=g)
    o = torch.zeros(n, 3)
    d = torch.cat([torch.randn(n, 2, generator=g) * 0.05, torch.ones(n, 1)], dim=1)
    bg = torch.tensor([0.2, 0.4, 0.6])
    hip, Lh = _hip_trace(gpu_device, sc, colors, others, o, d, bg, need_grad=True)
    ref, Lr = _oracle(sc, colors, others, o, d, bg, torch.float64, need_grad=True)
    assert int(ref["hits"].min()) > 15 * 16
    for k in ("rgb", "dpt", "acc", "norm", "dist", "aux"):
        assert float((hip[k].detach().cpu().double() - ref[k].detach()).abs().max()) < 5e-5 * max(1.0, float(ref[k].detach().abs().max())), k
    w = torch.randn(n, 3, generator=g)
    (hip["rgb"] * w.to(gpu_device)).sum().backward()
    (ref["rgb"] * w.double()).sum().backward()
    for k in ("means", "scales", "rotations", "opacities", "colors", "o", "d"):
        a, b = Lh[k].grad.cpu().double(), Lr[k].grad
        assert float((a - b).abs().max()) <= 1e-3 * max(float(b.abs().max()), 1e-12), k


@pytest.mark.gpu
def test_degenerate_ray_sets(gpu_device):
    """No rays, rays without a direction, non-finite rays, a ray set that is not a multiple of anything: background and zeros, no hang."""
    sc, colors, others = _scene(200, 3, 60.0)
    bg = torch.tensor([0.3, 0.6, 0.9])
    hip, _ = _hip_trace(gpu_device, sc, colors, others, torch.zeros(0, 3), torch.zeros(0, 3), bg)
    assert hip["rgb"].shape == (0, 3) and hip["wet"].shape == (200,) and float(hip["wet"].abs().sum()) == 0.0
    o = torch.zeros(7, 3)
    d = torch.zeros(7, 3)
    d[1] = torch.tensor([float("nan"), 0.0, 1.0]); d[2] = torch.tensor([float("inf"), 0.0, 1.0]); d[3] = torch.tensor([0.0, 0.0, 1.0])
    o[4] = torch.tensor([float("nan"), 0.0, 0.0]); d[4] = torch.tensor([0.0, 0.0, 1.0])
    hip, L = _hip_trace(gpu_device, sc, colors, others, o, d, bg, need_grad=True)
    ref, _ = _oracle(sc, colors, others, o[3:4], d[3:4], bg, torch.float64)
    for r in (0, 1, 2, 4, 5, 6):
        assert torch.allclose(hip["rgb"][r].cpu(), bg) and float(hip["acc"][r]) == 0.0, r
    assert float((hip["rgb"][3].detach().cpu().double() - ref["rgb"][0]).abs().max()) < 1e-5 and float(ref["acc"][0]) > 0
    hip["rgb"].sum().backward()
    assert torch.isfinite(L["means"].grad).all() and torch.isfinite(L["d"].grad[3]).all()


@pytest.mark.gpu
def test_empty_model_and_forward_only_state(gpu_device):
    """(1) An empty surfel set (everything pruned) traces to the background instead of raising.  (2) Under torch.no_grad() the forward
    allocates the state without the replay record (mrgs_surfel_trace_state_floats_norecord) and returns the same outputs.  (3) A state
    without the record handed to the C backward makes it walk again: same gradients as with the record."""
    from types import SimpleNamespace
    from materialrefgs_amd import _lib
    L = _lib.lib()
    bg = torch.tensor([0.3, 0.6, 0.9])
    empty = SimpleNamespace(means3D=torch.zeros(0, 3), scales=torch.zeros(0, 2), rotations=torch.zeros(0, 4), opacities=torch.zeros(0, 1))
    o, d = _rays(130, 1)
    hip, _ = _hip_trace(gpu_device, empty, torch.zeros(0, 3), torch.zeros(0, 2), o, d, bg)
    assert torch.allclose(hip["rgb"].cpu(), bg.expand(130, 3)) and float(hip["acc"].abs().max()) == 0.0 and hip["wet"].shape == (0,)
    assert float(hip["dpt"].abs().max()) == 0.0 and float(hip["norm"].abs().max()) == 0.0 and float(hip["dist"].abs().max()) == 0.0

    sc, colors, others = _scene(900, 4, 40.0)
    o, d = _rays(2048, 4)
    full = L.mrgs_surfel_trace_state_floats(2048, 0)
    small = L.mrgs_surfel_trace_state_floats_norecord(2048, 0)
    assert 0 < small < full // 8
    with_grad, Lg = _hip_trace(gpu_device, sc, colors, others, o, d, bg, need_grad=True)
    with torch.no_grad():
        no_grad, _ = _hip_trace(gpu_device, sc, colors, others, o, d, bg)
    for k in ("rgb", "dpt", "acc", "norm", "dist", "aux"):
        assert torch.equal(with_grad[k].detach(), no_grad[k]), k
    assert float((with_grad["wet"].detach() - no_grad["wet"]).abs().max()) <= 1e-5 * float(no_grad["wet"].abs().max())   # summed with atomics
    # the autograd node of the no-grad trace kept nothing; the one with gradients kept the full state
    def trace_node(t):
        fn = t.grad_fn
        while not hasattr(fn, "saved_tensors"):      # the tracer reshapes its outputs to the rays' shape: a view node sits on top
            fn = fn.next_functions[0][0]
        return fn
    state = trace_node(with_grad["rgb"]).saved_tensors[-1]
    assert state.numel() == full
    # C level: backward on a record-less state == backward on the recorded one
    w = torch.randn(2048, 3, generator=torch.Generator().manual_seed(5)).to(gpu_device)
    (with_grad["rgb"] * w).sum().backward()
    replayed = {k: Lg[k].grad.clone() for k in ("means", "scales", "rotations", "opacities", "colors", "o", "d")}
    import os
    os.environ["MRGS_TRACE_NO_RECORD"] = "0"      # (the developer switch is read once per process; this test does not rely on it)
    from materialrefgs_amd import surfel_tracing as st
    keep = L.mrgs_surfel_trace_state_floats
    try:
        st._lib.lib().mrgs_surfel_trace_state_floats = L.mrgs_surfel_trace_state_floats_norecord     # force the small state with gradients on
        walked, Lw = _hip_trace(gpu_device, sc, colors, others, o, d, bg, need_grad=True)
    finally:
        st._lib.lib().mrgs_surfel_trace_state_floats = keep
    assert trace_node(walked["rgb"]).saved_tensors[-1].numel() == small
    (walked["rgb"] * w).sum().backward()
    for k, ref in replayed.items():
        a = Lw[k].grad
        assert float((a - ref).abs().max()) <= 2e-5 * max(float(ref.abs().max()), 1e-12), k


@pytest.mark.gpu
def test_two_traces_on_one_hierarchy_keep_their_own_backward(gpu_device):
    """a trace saves the hierarchy blob for its backward, and the trace call writes the surfel records into it.  A second
    surfel set of the same size traced (or built) through the same SurfelTracer before the first backward ran must not change the first
    one's gradients."""
    from materialrefgs_amd.surfel_tracing import SurfelTracer, SurfelTracingSettings
    dev = gpu_device
    bg = torch.tensor([0.1, 0.2, 0.3])
    o, d = _rays(1024, 2)
    eye = torch.eye(4, device=dev)
    ts = SurfelTracingSettings(1, 1024, 1.0, 1.0, bg.to(dev), 1.0, eye, eye, 0, torch.zeros(3, device=dev), False, False)

    def run(tr, sc, colors, others, build):
        leaf = lambda t: t.to(dev).clone().requires_grad_(True)
        Lf = dict(means=leaf(sc.means3D), scales=leaf(sc.scales), rotations=leaf(sc.rotations), opacities=leaf(sc.opacities), colors=leaf(colors))
        if build:
            tr.build_acceleration_structure(sto.quad_vertices(Lf["means"].detach(), Lf["scales"].detach(), Lf["rotations"].detach()).reshape(-1, 3), None)
        out = tr(o.to(dev), d.to(dev), None, means3D=Lf["means"], grads3D=None, shs=None, colors_precomp=Lf["colors"], others_precomp=others.to(dev),
                 opacities=Lf["opacities"], scales=Lf["scales"], rotations=Lf["rotations"], cov3D_precomp=None, tracer_settings=ts)
        return out[0], Lf

    from types import SimpleNamespace
    A = _scene(700, 11, 45.0)
    B = _scene(700, 12, 45.0)
    # the same boxes with other contents (a hierarchy stays valid for it): what an eval-mode tracer (has_bvh) would be handed
    B_same = (SimpleNamespace(means3D=A[0].means3D, scales=A[0].scales, rotations=A[0].rotations, opacities=A[0].opacities * 0.5), B[1], B[2])
    w = torch.randn(1024, 3, generator=torch.Generator().manual_seed(3)).to(dev)
    # reference: A alone
    rgbA, LA = run(SurfelTracer(), *A, build=True)
    (rgbA * w).sum().backward()
    want = {k: v.grad.clone() for k, v in LA.items()}
    for rebuild in (True, False):
        tr = SurfelTracer()
        rgb1, L1 = run(tr, *A, build=True)
        blob1 = tr._blob
        rgb2, L2 = run(tr, *(B if rebuild else B_same), build=rebuild)   # same P: second build, or second trace on the first hierarchy
        assert tr._blob.data_ptr() != blob1.data_ptr()
        (rgb2 * w).sum().backward()
        (rgb1 * w).sum().backward()
        for k in want:
            assert torch.equal(L1[k].grad, want[k]) or float((L1[k].grad - want[k]).abs().max()) <= 1e-6 * float(want[k].abs().max()), (rebuild, k)


def test_size_functions_of_the_c_abi_without_a_gpu():
    """mrgs_surfel_bvh_bytes / _ws_bytes / _trace_state_floats are pure host arithmetic: monotone, and large enough for what the header
    says they hold (64-wide nodes over ceil(P / 64) groups; per ray 4 state floats + a list slot; per block of 64 rays two 4 KB chunks)."""
    from materialrefgs_amd import _lib
    L = _lib.lib()
    prev = (0, 0)
    for P in (0, 1, 63, 64, 65, 4096, 4097, 300000, 1 << 20):
        b, w = L.mrgs_surfel_bvh_bytes(P), L.mrgs_surfel_bvh_ws_bytes(P)
        groups = max(1, (P + 63) // 64)
        assert b >= groups * (1536 + 8) + groups * 64 * 64 + P * 4 and w >= P * (24 + 16)
        assert b >= prev[0] and w >= prev[1]
        prev = (b, w)
    for n, width in ((0, 0), (1, 0), (640000, 800), (640000, 0), (2560000, 1600), (1000, 37)):
        f = L.mrgs_surfel_trace_state_floats(n, width)
        blocks = ((width + 7) // 8) * ((n // width + 7) // 8) if width and n % width == 0 else (n + 63) // 64
        assert f >= 5 * n + 32 + blocks * 2 * 1024, (n, width, f)
    assert L.mrgs_surfel_trace_state_floats(640000, 800) >= L.mrgs_surfel_trace_state_floats(640000, 0)       # 8x8 blocks pad the image's edges


@pytest.mark.gpu
def test_mirror_ray_kernel_matches_the_torch_statement(gpu_device):
    """mrgs_mirror_rays_*: ray origins / directions and the gradients to the normal map and the depth against renderer.mirror_rays_torch in
    float64 (utils/refl_utils.py:75-98, envgs_renderer.py:717-724), incl. a strided normal map and pixels with a zero normal."""
    from materialrefgs_amd import renderer
    from materialrefgs_amd.synthetic import orbit_camera
    H, W = 37, 53
    g = torch.Generator().manual_seed(2)
    cam = orbit_camera(5, H, W)
    n_chw = torch.randn(3, H, W, generator=g)
    n_chw[:, :3, :5] = 0.0                                              # uncovered pixels: normal 0 -> the ray looks back along the view ray
    depth = torch.rand(1, H, W, generator=g) * 3 + 0.5
    def run(dev, dt, fn):
        n = n_chw.to(dev).to(dt).clone().requires_grad_(True)
        d = depth.to(dev).to(dt).clone().requires_grad_(True)
        c = cam.to(dev) if dev != "cpu" else cam
        o, r = fn(c, n.permute(1, 2, 0), d)                             # a permuted (strided) view, as render_surfel hands it over
        return n, d, o, r
    nh, dh, oh, rh = run(gpu_device, torch.float32, renderer._mirror_rays)
    nr, dr, orf, rr = run("cpu", torch.float64, renderer.mirror_rays_torch)
    assert float((oh.cpu().double() - orf).detach().abs().max()) < 2e-5 and float((rh.cpu().double() - rr).detach().abs().max()) < 1e-5
    wo, wr = torch.randn(H, W, 3, generator=g), torch.randn(H, W, 3, generator=g)
    ((oh * wo.to(gpu_device)).sum() + (rh * wr.to(gpu_device)).sum()).backward()
    ((orf * wo.double()).sum() + (rr * wr.double()).sum()).backward()
    assert float((nh.grad.cpu().double() - nr.grad).abs().max()) <= 3e-5 * float(nr.grad.abs().max())
    assert float((dh.grad.cpu().double() - dr.grad).abs().max()) <= 3e-5 * float(dr.grad.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("switch", ["MRGS_TRACE_NO_DEFER", "MRGS_TRACE_NO_PACKETS", "MRGS_TRACE_NO_RECORD"])
def test_fallback_paths_of_the_tracer(gpu_device, switch):
    """The paths taken when a list is full (a block walks its own packets), when no rays run together (every ray a wave of its own) and
    when the record is not kept (the backward walks again) are selected by developer switches read once per process: the gradient and
    forward parity tests are run again in a child process with each switch set."""
    import subprocess
    env = dict(os.environ, **{switch: "1"})
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-k",
                        "gradients_match or many_layers or matches_the_dense_oracle and not 20000"], env=env, cwd=ROOT, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]
    assert " passed" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("P,degree", [(3000, 3), (130, 1), (64, 0)])
def test_raw_record_kernel_equals_the_getters_path(gpu_device, P, degree):
    """mrgs_surfel_trace_prep_raw_*: the model's own tensors (raw scaling / rotation / opacity, split SH) against GaussianModel's getters in
    torch float64 (exp / normalize / sigmoid / cat, scene/gaussian_model.py:56-78, 236-259) followed by the torch statement of the records:
    same records and corners, every raw-parameter gradient and the gradient of the densification proxy (= the means' gradient)."""
    from materialrefgs_amd.surfel_tracing import _PrepRaw, surfel_records, _zero_leaf
    from materialrefgs_amd.gs_utils import eval_sh
    dev = gpu_device
    sc = make_shell_scene(P, seed=4)
    g = torch.Generator().manual_seed(8)
    raw = dict(xyz=sc.means3D, scaling=torch.log(sc.scales), rotation=sc.rotations * 0.6, opacity=torch.randn(P, 1, generator=g),
               f_dc=sc.shs[:, :1] + 0.2 * torch.randn(P, 1, 3, generator=g), f_rest=0.3 * torch.randn(P, 15, 3, generator=g))
    others = torch.full((P, 2), 0.01)
    campos = torch.tensor([0.3, -2.0, 1.1])
    mod = 0.9
    Lh = {k: v.to(dev).float().clone().contiguous().requires_grad_(True) for k, v in raw.items()}
    proxy = _zero_leaf(Lh["xyz"])
    gh, ah, qh = _PrepRaw.apply(Lh["xyz"], Lh["scaling"], Lh["rotation"], Lh["opacity"], Lh["f_dc"], Lh["f_rest"], others.to(dev), proxy, campos.to(dev),
                                degree, mod)
    Lr = {k: v.double().clone().requires_grad_(True) for k, v in raw.items()}
    shs = torch.cat([Lr["f_dc"], Lr["f_rest"]], dim=1)
    d = Lr["xyz"] - campos.double().reshape(1, 3)
    col = torch.clamp_min(eval_sh(degree, shs.transpose(1, 2), d / d.norm(dim=1, keepdim=True)) + 0.5, 0.0)
    scales, rot, op = torch.exp(Lr["scaling"]), torch.nn.functional.normalize(Lr["rotation"]), torch.sigmoid(Lr["opacity"])
    gr, ar = surfel_records(Lr["xyz"], scales, rot, op, col, others.double(), mod)
    qr = sto.quad_vertices(Lr["xyz"], scales, rot, mod).reshape(-1, 3)
    assert float((gh.cpu().double()[:, :13] - gr[:, :13]).abs().max()) < 2e-5 * float(gr.abs().max())
    assert float((ah.cpu().double()[:, :5] - ar[:, :5]).abs().max()) < 2e-6
    assert float((qh.cpu().double() - qr.detach()).abs().max()) < 2e-6
    wg, wa = torch.randn(P, 16, generator=g), torch.randn(P, 8, generator=g)
    wg[:, 13:] = 0; wa[:, 5:] = 0
    ((gh * wg.to(dev)).sum() + (ah * wa.to(dev)).sum()).backward()
    ((gr * wg.double()).sum() + (ar * wa.double()).sum()).backward()
    for k in raw:
        a, b = Lh[k].grad.cpu().double(), Lr[k].grad
        assert float((a - b).abs().max()) <= 3e-5 * max(float(b.abs().max()), 1e-12), k
    assert torch.equal(proxy.grad, Lh["xyz"].grad)


@pytest.mark.gpu
def test_blended_mirror_rays_and_traced_blend(gpu_device):
    """mrgs_mirror_rays_blended_* (reflecting normal = safe_normalize(rend_normal / clamp_min(alpha, 1e-6)) inside the ray kernel) and
    mrgs_traced_blend_* against the torch expressions of render_surfel_with_envgs (gaussian_renderer/__init__.py:493-517) in float64, with the
    tracer's strided [H,W,C] views and pixels of zero alpha."""
    from materialrefgs_amd import renderer
    from materialrefgs_amd.gs_utils import safe_normalize
    from materialrefgs_amd.synthetic import orbit_camera
    H, W = 37, 53
    g = torch.Generator().manual_seed(4)
    cam = orbit_camera(3, H, W)
    rn0 = torch.randn(3, H, W, generator=g) * 0.7
    al0 = torch.rand(1, H, W, generator=g)
    al0[:, :4, :6] = 0.0
    rn0[:, :4, :6] = 0.0
    sd0 = torch.rand(1, H, W, generator=g) * 3 + 0.5
    dev = gpu_device
    rn, al, sd = (t.to(dev).clone().requires_grad_(True) for t in (rn0, al0, sd0))
    oh, rh = renderer._mirror_rays_blended(cam.to(dev), rn, al, sd)
    rn64, al64, sd64 = (t.double().clone().requires_grad_(True) for t in (rn0, al0, sd0))
    nm = safe_normalize(rn64.permute(1, 2, 0) / al64.permute(1, 2, 0).clamp_min(1e-6))
    orf, rr = renderer.mirror_rays_torch(cam, nm, sd64)
    assert float((oh.cpu().double() - orf).detach().abs().max()) < 2e-5 and float((rh.cpu().double() - rr).detach().abs().max()) < 1e-5
    wo, wr = torch.randn(H, W, 3, generator=g), torch.randn(H, W, 3, generator=g)
    ((oh * wo.to(dev)).sum() + (rh * wr.to(dev)).sum()).backward()
    ((orf * wo.double()).sum() + (rr * wr.double()).sum()).backward()
    for a, b, name in ((rn, rn64, "rend_normal"), (sd, sd64, "surf_depth")):
        assert float((a.grad.cpu().double() - b.grad).abs().max()) <= 5e-5 * float(b.grad.abs().max()), name
    # the unit normal does not depend on alpha: the float64 gradient is rounding noise around zero, the kernel writes exact zeros
    assert float(al64.grad.abs().max()) < 1e-9 and float(al.grad.abs().max()) == 0.0
    # the blend: a [3,H,W] contiguous, traced / specular as channel-first views of [H,W,3] / [H,W,2]
    a0, b0, s0 = torch.rand(3, H, W, generator=g), torch.rand(H, W, 3, generator=g), torch.rand(H, W, 2, generator=g)
    a, b, s = (t.to(dev).clone().requires_grad_(True) for t in (a0, b0, s0))
    out = renderer._TracedBlend.apply(a, b.permute(2, 0, 1), s[..., :1].permute(2, 0, 1))
    a6, b6, s6 = (t.double().clone().requires_grad_(True) for t in (a0, b0, s0))
    spec = s6[..., :1].permute(2, 0, 1)
    ref = a6 * (1 - spec) + spec * b6.permute(2, 0, 1)
    assert float((out.cpu().double() - ref).detach().abs().max()) < 1e-6
    w = torch.randn(3, H, W, generator=g)
    (out * w.to(dev)).sum().backward()
    (ref * w.double()).sum().backward()
    for x, y in ((a, a6), (b, b6), (s, s6)):
        assert float((x.grad.cpu().double() - y.grad).abs().max()) <= 2e-6 * max(1.0, float(y.grad.abs().max()))
