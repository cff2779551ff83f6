"""End-to-end checker of `render_surfel` -- TEST INFRASTRUCTURE ONLY (tests/ may import it; the product never does).

Composes, on the CPU, the checkers of the individual stages in the order of the reference's
gaussian_renderer/__init__.py:225-483 (`render_surfel`, "2dgs" flavour, SH-indirect branch):

    glue_oracle.surfel_features_reference   :338-355   (torch, float64, autograd)
    raster oracle (mrgs_oracle.c)           :359-370   (C, any arithmetic mode; wrapped as an autograd node below)
    glue_oracle.compute_2dgs_normal_...     :42-90, 392  (torch, float64, autograd)
    normal_map, get_specular_color_surfel   :419-433, utils/refl_utils.py:364-419  (shading_oracle; visibility by trace_oracle)
    compositing, sRGB, background           :436-449
    EnvLight.build_mips                     scene/light.py:72-86 (envfilter_oracle, dense float64 operators)

so that the wiring of the product's render_surfel -- feature-channel order (refl, roughness, albedo 3, indirect 3), strided views,
which maps feed which stage, gradient routing back to every parameter incl. the environment cubemap -- is compared as a whole.
Parity status of the parts: see the headers of the individual oracle modules.
"""
import math

import numpy as np
import torch

from materialrefgs_amd.gs_utils import linear_to_srgb
from . import envfilter_oracle as ef
from . import glue_oracle as go
from . import raster_oracle as ro
from . import shading_oracle as so
from . import trace_oracle as to


LAST_NUM_RENDERED = None


class _OracleRaster(torch.autograd.Function):
    """diff_surfel_rasterization forward/backward through oracle/mrgs_oracle.c (inputs rounded to fp32 at the boundary, as the
    reference's fp32 tensors are)."""

    @staticmethod
    def forward(ctx, means3D, means2D, opacities, shs, features, scales, rotations, cam, sh_degree, variant, precomp=False):
        # precomp: `shs` holds precomputed colours [P,3] (render_volume: colors_precomp = specular + diffuse)
        r = ro.OracleRender(means3D=means3D, opacities=opacities, H=cam.image_height, W=cam.image_width,
                            tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), viewmatrix=cam.world_view_transform,
                            projmatrix=cam.full_proj_transform, campos=cam.camera_center, shs=None if precomp else shs,
                            colors_precomp=shs if precomp else None, features=features, scales=scales,
                            rotations=rotations, sh_degree=sh_degree, variant=variant)
        ctx.precomp = precomp
        ctx.r = r
        global LAST_NUM_RENDERED
        LAST_NUM_RENDERED = int(r.R)          # diagnostics: the (tile, surfel) pair count of the most recent oracle rasterization
        ctx.dtype = means3D.dtype
        ctx.shapes = (means3D.shape, opacities.shape, shs.shape, features.shape, scales.shape, rotations.shape)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(means3D.dtype)
        radii = torch.from_numpy(r.radii.copy())
        ctx.mark_non_differentiable(radii)
        return t(r.color), t(r.feature), t(r.others), radii

    @staticmethod
    def backward(ctx, g_color, g_feature, g_others, _g_radii):
        r = ctx.r
        z = lambda g, shp: np.zeros(shp, np.float32) if g is None else g.detach().numpy()
        H, W, S = r.H, r.W, r.S
        g = r.backward(z(g_color, (3, H, W)), z(g_feature, (S, H, W)), z(g_others, (7, H, W)))
        t = lambda a, shp: torch.from_numpy(np.ascontiguousarray(a)).to(ctx.dtype).reshape(shp)
        s = ctx.shapes
        return (t(g["means3D"], s[0]), t(g["means2D"], (s[0][0], 3)), t(g["opacity"], s[1]), t(g["colors" if ctx.precomp else "sh"], s[2]),
                t(g["features"], s[3]), t(g["scales"], s[4]), t(g["rotations"], s[5]), None, None, None, None)


class _OracleBuildMips(torch.autograd.Function):
    """EnvLight.build_mips (scene/light.py:72-86) with the dense float64 operators of envfilter_oracle: (specular levels..., diffuse)."""

    @staticmethod
    def forward(ctx, base, min_res, min_roughness, max_roughness):
        spec, diffuse, ops = ef.build_mips(base.detach().numpy(), min_res, min_roughness, max_roughness)
        ctx.ops = ops
        ctx.dtype = base.dtype
        return tuple(torch.from_numpy(np.ascontiguousarray(s)).to(base.dtype) for s in list(spec) + [diffuse])

    @staticmethod
    def backward(ctx, *g_all):
        g_spec, g_diff = g_all[:-1], g_all[-1]
        shapes = [(6, op.res, op.res, 3) for op in ctx.ops]
        g = [np.zeros(s) if gi is None else gi.detach().numpy() for gi, s in zip(g_spec, shapes)]
        gd = None if g_diff is None else g_diff.detach().numpy()
        return torch.from_numpy(ef.build_mips_backward(ctx.ops, g, gd)).to(ctx.dtype), None, None, None


def sample_camera_rays_unnormalize(H, W, K, R, T):
    """utils/refl_utils.py:75-93."""
    R = R.T
    i, j = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing="xy")
    xy1 = np.stack([i, j, np.ones_like(i)], axis=2)
    pixel_camera = torch.tensor(np.dot(xy1, np.linalg.inv(K.astype(np.float32)).T))
    rays_o = (-R.T @ T.unsqueeze(-1)).flatten()
    pixel_world = (pixel_camera - T[None, None]).reshape(-1, 3) @ R
    return (pixel_world - rays_o[None]).reshape(H, W, 3), rays_o


def render_surfel_oracle(cam, pc, env_base, env_min_res, pipe, bg_color, srgb=False, indirect=False, mesh=None, variant="fused",
                         min_roughness=0.08, max_roughness=0.5, lut=None, visibility_bits=None, mips=None, flag="2dgs", raster_inputs=None):
    """`pc`: a SurfelModel on the CPU (float64 leaves recommended); `env_base`: [6,N,N,3] pre-sigmoid texels (leaf);
    `mesh`: (vertices, triangles) for opt.indirect.  Returns the reference's dictionary (CPU tensors, autograd attached).
    `visibility_bits` [H,W]: use these bits instead of the own trace in the blend (the trace result is still returned under
    "visibility_traced"): visibility is a step function of the mirror ray, a pixel on a silhouette flips with the last bit of the
    ray set-up, and a test that wants to compare GRADIENTS first checks the two bit maps against each other and then removes
    that source of difference.
    `mips`: prefiltered specular levels to shade with instead of building them from `env_base` (bench.py's CPU leg at 128^2 texels, where
    the dense float64 operators of envfilter_oracle -- (6 N^2)^2 entries -- do not fit; the environment then receives no gradient).
    `raster_inputs` = (opacities, scales, rotations, features): use THESE per-gaussian tensors instead of evaluating the glue of :338-355
    (bench.py hands over the product's own fp32 activations, so that both rasterizers see identical inputs: activations evaluated in
    float64 here and in fp32 there differ in the last bit, which moves a handful of threshold pixels of a 640 000-pixel image)."""
    from materialrefgs_amd.shading import load_fg_lut
    dt = pc._xyz.dtype
    H, W = cam.image_height, cam.image_width
    lut = load_fg_lut("cpu") if lut is None else lut
    means2D = torch.zeros_like(pc._xyz, requires_grad=True)                                           # :229-233
    if raster_inputs is not None:
        opacities, scales, rotations, features = raster_inputs
    else:
        opacities, scales, rotations, features = go.surfel_features_reference(pc, cam.camera_center.to(dt))   # :338-355
    shs = pc.get_features
    if flag != "2dgs":            # "pgsr": + the plane distance as the last channel (:352-357)
        features = torch.cat((features, go.get_distance(pc, cam)), dim=-1)
    color, feat, allmap, radii = _OracleRaster.apply(pc.get_xyz, means2D, opacities, shs, features, scales, rotations, cam,
                                                     pc.active_sh_degree, variant)                   # :359-370
    rend_distance = None
    if flag != "2dgs":
        rend_distance = feat[-1:]
        allmap = go.pgsr_allmap8(allmap, rend_distance, cam)
    base_color = color
    refl_strength, roughness = feat[:1], feat[1:2]                                                    # :372-378 ("2dgs")
    albedo, indirect_light = feat[2:5], feat[5:8]
    cam_dt = cam._replace(world_view_transform=cam.world_view_transform.to(dt), full_proj_transform=cam.full_proj_transform.to(dt))
    reg = go.compute_2dgs_normal_and_regularizations_reference(allmap, cam_dt, pipe)                  # :392
    render_alpha, render_normal = reg["render_alpha"], reg["render_normal"]
    normal_map = render_normal.permute(1, 2, 0) / render_alpha.permute(1, 2, 0).clamp_min(1e-6)       # :419-421
    if mips is None:
        *mips, _diffuse_tex = _OracleBuildMips.apply(env_base, env_min_res, min_roughness, max_roughness)
    _H, _W, K = cam.HWK
    R32, T32 = cam.R.float(), cam.T.float()
    a_hw, r_hw, ro_hw = render_alpha.permute(1, 2, 0), refl_strength.permute(1, 2, 0), roughness.permute(1, 2, 0)
    specular, direct_light, specular_weight = so.specular_color_surfel(list(mips), lut, albedo.permute(1, 2, 0), H, W, K, R32, T32, normal_map,
                                                                       a_hw, r_hw, ro_hw, min_roughness, max_roughness)
    extra = {"direct_light": direct_light, "specular_weight": specular_weight}
    if indirect:                                                                                       # utils/refl_utils.py:379-401
        visibility = torch.ones_like(a_hw)
        mask = (a_hw > 0)[..., 0]
        rays_cam, rays_o = sample_camera_rays_unnormalize(H, W, K, R32, T32)
        w_o = -rays_cam / torch.clamp(torch.linalg.norm(rays_cam, dim=-1, keepdim=True), min=1e-20)
        rays_refl = 2 * normal_map * torch.sum(w_o * normal_map, dim=-1, keepdim=True) - w_o
        rays_refl = rays_refl / torch.clamp(torch.linalg.norm(rays_refl, dim=-1, keepdim=True), min=1e-20)
        inter = rays_o + reg["surf_depth"].permute(1, 2, 0) * rays_cam
        if mesh is not None:
            _, _, depth, _ = to.trace(mesh[0], mesh[1], inter[mask].detach().numpy(), rays_refl[mask].detach().numpy())
            visibility[mask] = torch.from_numpy((depth >= 10).astype(np.float64)).to(dt).unsqueeze(-1)
        traced = visibility
        if visibility_bits is not None:
            visibility = visibility_bits.to(dt).reshape(H, W, 1)
        ind_hw = indirect_light.permute(1, 2, 0)
        light = direct_light.permute(1, 2, 0) * visibility + (1 - visibility) * ind_hw
        specular = (light * a_hw * specular_weight).permute(2, 0, 1)
        indirect_color = ((1 - visibility) * ind_hw * a_hw * specular_weight).permute(2, 0, 1)
        extra.update({"visibility": visibility.permute(2, 0, 1), "visibility_traced": traced.permute(2, 0, 1), "indirect_light": indirect_light,
                      "indirect_color": indirect_color})
    diffuse = (1 - refl_strength) * base_color
    final_image = diffuse + specular                                                                   # :436
    if srgb:                                                                                           # :442-445
        final_image, albedo, specular = linear_to_srgb(final_image), linear_to_srgb(albedo), linear_to_srgb(specular)
    background = bg_color.to(dt)[:, None, None] * (1 - render_alpha)
    final_image = final_image + background                                                             # :448
    out = {"render": final_image, "refl_strength_map": refl_strength, "diffuse_map": diffuse, "diffuse_map_ori": base_color,
           "specular_map": specular, "base_color_map": albedo, "roughness_map": roughness, "viewspace_points": means2D,
           "visibility_filter": radii > 0, "radii": radii, "rend_alpha": render_alpha, "rend_normal": render_normal,
           "rend_dist": reg["render_dist"], "surf_depth": reg["surf_depth"], "surf_normal": reg["surf_normal"]}
    if rend_distance is not None:
        out["rend_distance"] = rend_distance
    if indirect:
        out.update(extra)
        out["indirect_color"] = diffuse + extra["indirect_color"] + background                        # :449-452
    return out


def render_volume_oracle(cam, pc, env_base, env_min_res, pipe, bg_color, srgb=False, indirect=False, mesh=None, variant="fused",
                         min_roughness=0.08, max_roughness=0.5, lut=None, visibility_bits=None, flag="pgsr"):
    """gaussian_renderer/__init__.py:521-749 (`render_volume`, SH-indirect branch; it only runs under the shipped "pgsr" flag, so that is
    the default here: plane distance as the last feature channel, surf_depth from the flavour's eighth all-map channel) with utils/refl_utils.py:426-484 for the
    per-gaussian shading, INCLUDING the reference's `fg[0]` indexing (appendix B-27: every gaussian gets the split-sum table value of
    gaussian 0).  `pc`: SurfelModel on the CPU; `env_base`: the texels of pc.get_envmap_2."""
    from materialrefgs_amd.gs_utils import eval_sh
    from materialrefgs_amd.shading import load_fg_lut
    dt = pc._xyz.dtype
    lut = (load_fg_lut("cpu") if lut is None else lut).to(dt)
    means2D = torch.zeros_like(pc._xyz, requires_grad=True)
    means3D, opacity = pc.get_xyz, pc.get_opacity
    refl, ori_color, roughness = pc.get_refl, pc.get_ori_color, pc.get_rough
    dir_pp = means3D - cam.camera_center.to(dt)
    dir_n = dir_pp / dir_pp.norm(dim=1, keepdim=True)
    normals = pc.get_normal(1.0, dir_n)
    w_o = -dir_n
    reflection = 2 * torch.sum(normals * w_o, dim=1, keepdim=True) * normals - w_o                  # :631-634
    shs_indirect = pc.get_indirect.transpose(1, 2).reshape(-1, 3, (pc.max_sh_degree + 1) ** 2)
    indirect_light = torch.clamp_min(eval_sh(3, shs_indirect, reflection), 0.0)
    # get_full_color_volume[_indirect], utils/refl_utils.py:426-484
    *mips, diffuse_tex = _OracleBuildMips.apply(env_base, env_min_res, min_roughness, max_roughness)
    Rt = cam.R.to(dt).T
    rays_o = (-Rt.T @ cam.T.to(dt).unsqueeze(-1)).flatten()
    wo2 = rays_o - means3D
    wo2 = wo2 / torch.clamp(torch.linalg.norm(wo2, dim=-1, keepdim=True), min=1e-20)
    NdotV = torch.sum(wo2 * normals, dim=-1, keepdim=True)
    rays_refl = 2 * normals * NdotV - wo2
    rays_refl = rays_refl / torch.clamp(torch.linalg.norm(rays_refl, dim=-1, keepdim=True), min=1e-20)
    fg = so.lut_fetch(lut, torch.cat([NdotV, roughness], -1).clamp(0, 1))                               # [N,2]
    fg0 = fg[0]                                                                                          # the reference's fg[0] (B-27)
    diffuse = so.env_lookup([diffuse_tex], normals) * (1 - refl) * ori_color
    direct_light = so.env_lookup(list(mips), rays_refl, roughness.reshape(-1), min_roughness, max_roughness)
    specular_weight = (0.04 * (1 - refl) + ori_color * refl) * fg0[0:1] + fg0[1:2]
    if indirect:
        visibility = torch.ones_like(opacity)
        mask = (opacity > 0).squeeze(-1)
        traced = visibility.clone()
        if mesh is not None:
            _, _, depth, _ = to.trace(mesh[0], mesh[1], means3D[mask].detach().numpy(), rays_refl[mask].detach().numpy())
            traced[mask] = torch.from_numpy((depth >= 10).astype(np.float64)).to(dt).unsqueeze(-1)
        visibility = traced if visibility_bits is None else visibility_bits.to(dt).reshape(-1, 1)
        specular = (direct_light * visibility + (1 - visibility) * indirect_light) * specular_weight
        features = torch.cat((roughness, refl, diffuse, specular, ori_color, visibility, indirect_light, direct_light), dim=-1)
    else:
        specular = direct_light * specular_weight
        features = torch.cat((roughness, refl, diffuse, specular, ori_color), dim=-1)
    colors_precomp = specular + diffuse
    if flag != "2dgs":
        features = torch.cat((features, go.get_distance(pc, cam)), dim=-1)                           # :657-661
    color, feat, allmap, radii = _OracleRaster.apply(means3D, means2D, opacity, colors_precomp, features, pc.get_scaling, pc.get_rotation, cam,
                                                     pc.active_sh_degree, variant, True)
    if flag != "2dgs":
        allmap = go.pgsr_allmap8(allmap, feat[-1:], cam)
    cam_dt = cam._replace(world_view_transform=cam.world_view_transform.to(dt), full_proj_transform=cam.full_proj_transform.to(dt))
    reg = go.compute_2dgs_normal_and_regularizations_reference(allmap, cam_dt, pipe)
    full_color, d_map, s_map = color, feat[2:5], feat[5:8]
    if srgb:
        d_map, s_map, full_color = linear_to_srgb(d_map), linear_to_srgb(s_map), linear_to_srgb(full_color)
    final_image = full_color + bg_color.to(dt)[:, None, None] * (1 - reg["render_alpha"])
    out = {"render": final_image, "refl_strength_map": feat[1:2], "diffuse_map": d_map, "specular_map": s_map, "base_color_map": feat[8:11],
           "roughness_map": feat[:1], "viewspace_points": means2D, "visibility_filter": radii > 0, "radii": radii,
           "rend_alpha": reg["render_alpha"], "rend_normal": reg["render_normal"], "rend_dist": reg["render_dist"],
           "surf_depth": reg["surf_depth"], "surf_normal": reg["surf_normal"]}
    if indirect:
        out.update({"visibility": feat[11:12], "indirect_light": feat[12:15], "direct_light": feat[15:18], "visibility_traced": traced})
    if flag != "2dgs":
        out["rend_distance"] = feat[-1:]
    return out


LEAF_NAMES = ("xyz", "scaling", "rotation", "opacity", "features_dc", "features_rest", "refl_strength", "roughness", "ori_color",
              "indirect_dc", "indirect_rest")
RASTER_INPUT_NAMES = ("opacities", "scales", "rotations", "features")


def surfel_leaf_gradients(cam, leaves, env_base, raster_inputs, keys, upstream, pipe, bg_color, env_min_res=16, mips_device=None,
                          min_roughness=0.08, max_roughness=0.5, shade_levels=None, literal32_prefilter=False):
    """render_surfel of ONE view through the checkers, forward and backward, down to EVERY leaf -- the full-size form of the end-to-end
    comparison (bench.py's CPU leg at C3full, tests/test_full_size.py):

      `leaves`          the 11 raw parameter tensors in LEAF_NAMES order (any dtype / device; evaluated here in float64 on the CPU)
      `env_base`        the environment's pre-sigmoid texels [6,N,N,3]
      `raster_inputs`   the product's OWN fp32 per-gaussian rasterizer inputs (activated opacity / scale / rotation, the material rows)
                        captured from inside its render: both rasterizers then see identical numbers, and no alpha = 1/255 or
                        T = 1e-4 decision of a 640 000-pixel image hangs on the last bit of an activation (render_surfel_oracle)
      `keys`, `upstream` the output maps the scalar reads and their fixed upstream gradients

      `shade_levels`    prefiltered levels to SHADE with instead of the checker's own float64 ones (the product's levels at full size:
                        the reference's fp32 filter weights are ill-conditioned at roughness 0.08 -- envfilter_oracle.BlockedSpecular --
                        so levels built in fp32 and in float64 differ by per cents of a level's range at 128^2, which is a statement
                        about the prefilter, checked on its own below, and must not leak into the comparison of the shading)
      `literal32_prefilter`  also evaluate the prefilter's truth leg: the levels and the texel gradient with the filter weights formed in
                        the reference's fp32 arithmetic (info["levels_lit32"], grads["lit32"]["env_base"])

    Chain: prefiltered levels from `env_base` (envfilter_oracle.build_mips; levels above 32^2 blocked in float64 on `mips_device`) ->
    rasterizer (mrgs_oracle.c) + maps + shading + compositing at `raster_inputs` -> the gradient arriving at `raster_inputs` is pulled
    back through the float64 glue of gaussian_renderer/__init__.py:338-355 (glue_oracle.surfel_features_reference, evaluated at the raw
    leaves) and added to what reaches the leaves directly (xyz through the rasterizer and the SH view direction; the colour SH); the
    gradient arriving at the levels through build_mips_backward to the texels.
    Returns (out, grads, info): `out` the checker's dictionary; `grads` name -> float64 numpy for LEAF_NAMES + "env_base" +
    "viewspace_points" + RASTER_INPUT_NAMES + "env_levels" (list) + "lit32" (name -> the same leaf gradient with the glue's pull-back
    evaluated as the reference evaluates it -- torch float32, the reference's own ops -- from the SAME upstream gradient: the truth leg of
    the glue, see leaf_gradient_report; with literal32_prefilter also "env_base" through the fp32-weight operators); info = {"raster_shading_seconds": the part a CPU baseline may be quoted on (rasterizer + maps +
    shading + compositing, forward and backward), "levels": the float64 levels}."""
    import time
    from materialrefgs_amd.renderer import SurfelModel
    f64 = lambda t_: t_.detach().cpu().double().requires_grad_(True)
    pc = SurfelModel(*[f64(t_) for t_ in leaves[:6]], **{n: f64(t_) for n, t_ in zip(LEAF_NAMES[6:], leaves[6:11])})
    inter = [f64(t_) for t_ in raster_inputs]
    spec, _diffuse, ops = ef.build_mips(env_base.detach().cpu().double().numpy(), env_min_res, min_roughness, max_roughness, device=mips_device)
    levels = [torch.from_numpy(np.ascontiguousarray(np.asarray(s, dtype=np.float64))).requires_grad_(True) for s in (shade_levels if shade_levels is not None else spec)]
    cam = cam.to("cpu") if hasattr(cam, "to") else cam
    t0 = time.perf_counter()
    out = render_surfel_oracle(cam, pc, None, None, pipe, bg_color.detach().cpu(), srgb=False, mips=levels, raster_inputs=tuple(inter),
                               min_roughness=min_roughness, max_roughness=max_roughness)
    torch.autograd.backward([out[k] for k in keys], [g_.detach().cpu().double() for g_ in upstream])
    seconds = time.perf_counter() - t0
    g_inter = [t_.grad if t_.grad is not None else torch.zeros_like(t_) for t_ in inter]
    # the glue, float64, at the raw leaves: its Jacobian transposed applied to what arrived at the rasterizer's inputs
    glue_out = go.surfel_features_reference(pc, cam.camera_center.double())
    torch.autograd.backward(list(glue_out), g_inter)
    grads = {n: (getattr(pc, "_" + n).grad if getattr(pc, "_" + n).grad is not None else torch.zeros_like(getattr(pc, "_" + n))).numpy()
             for n in LEAF_NAMES}
    lit = glue_lit32_leg(leaves, cam.camera_center, g_inter, grads)
    g_levels = [np.zeros(tuple(l_.shape)) if l_.grad is None else l_.grad.numpy() for l_ in levels]
    grads["env_base"] = ef.build_mips_backward(ops, g_levels)
    grads["env_levels"] = g_levels
    info = {"raster_shading_seconds": seconds, "levels": [np.asarray(s) for s in spec]}
    if literal32_prefilter:
        spec32, _d32, ops32 = ef.build_mips(env_base.detach().cpu().double().numpy(), env_min_res, min_roughness, max_roughness, device=mips_device,
                                            literal32=True)
        lit["env_base"] = ef.build_mips_backward(ops32, g_levels)
        info["levels_lit32"] = [np.asarray(s) for s in spec32]
    grads["lit32"] = lit
    grads["viewspace_points"] = out["viewspace_points"].grad.numpy()
    for n, g_ in zip(RASTER_INPUT_NAMES, g_inter):
        grads[n] = g_.numpy()
    return out, grads, info


def glue_lit32_leg(leaves, campos, g_inter, total):
    """The truth leg of the per-gaussian glue: `total` (name -> float64 leaf gradient of the whole chain) with the glue's pull-back of
    `g_inter` (the gradient at the rasterizer's per-gaussian inputs) evaluated the way the REFERENCE evaluates it -- its own torch ops
    (gaussian_renderer/__init__.py:338-355, the GaussianModel getters) on fp32 tensors -- instead of in float64:
    lit[n] = total[n] - pullback_f64[n] + pullback_f32[n].  Everything else (rasterizer, maps, shading: the upstream gradient) is common
    to both, so the distance lit - total is exactly what fp32 arithmetic costs the glue's backward for this scene."""
    from materialrefgs_amd.renderer import SurfelModel
    parts = {}
    for dt in (torch.float64, torch.float32):
        conv = lambda t_: t_.detach().cpu().to(dt).requires_grad_(True)
        pc_ = SurfelModel(*[conv(t_) for t_ in leaves[:6]], **{n: conv(t_) for n, t_ in zip(LEAF_NAMES[6:], leaves[6:11])})
        torch.autograd.backward(list(go.surfel_features_reference(pc_, campos.detach().cpu().to(dt))), [g_.detach().to(dt) for g_ in g_inter])
        parts[dt] = {n: getattr(pc_, "_" + n).grad for n in LEAF_NAMES}
    lit = {}
    for n in LEAF_NAMES:
        if parts[torch.float32][n] is not None:
            lit[n] = np.asarray(total[n], dtype=np.float64) - parts[torch.float64][n].numpy() + parts[torch.float32][n].double().numpy()
    return lit


def leaf_gradient_report(hip, oracle_grads, names, bar=1e-4):
    """Per leaf: max-norm distance of the product's gradient from the checker's, relative to the checker's largest element; a leaf
    above `bar` passes only by the truth-leg rule of tests/test_gpu_parity.py:97-116, applied to the stage whose fp32 arithmetic is the
    reference's own (oracle_grads["lit32"]: the glue's pull-back in the reference's fp32 torch ops; the prefilter's with its fp32 filter
    weights): the product may be no further from the float64 value than that literal fp32 reading is (x 1.5), and the reading must
    itself be beyond half the bar (an ill-conditioned stage -- the quaternion normalisation's projector applied to a mostly radial
    gradient, GGX weights at roughness 0.08 -- not a loose kernel).  hip: name -> array-like.  Returns (rows, ok): rows name -> {"err", "lit32_err"?, "rule"}."""
    rows, ok = {}, True
    for n in names:
        b = np.asarray(oracle_grads[n], dtype=np.float64)
        a = np.asarray(hip[n], dtype=np.float64).reshape(b.shape)
        scale = max(float(np.abs(b).max()), 1e-30)
        e = float(np.abs(a - b).max() / scale)
        row = {"err": e, "rule": "bar"}
        if e > bar:
            lit = oracle_grads.get("lit32", {}).get(n)
            if lit is None:
                row["rule"], ok = "FAIL", False
            else:
                e_lit = float(np.abs(np.asarray(lit, dtype=np.float64).reshape(b.shape) - b).max() / scale)
                row["lit32_err"] = e_lit
                good = e <= 1.5 * e_lit and e_lit > 0.5 * bar
                row["rule"] = "truth-leg" if good else "FAIL"
                ok = ok and good
        rows[n] = row
    return rows, ok


def level_report(product_levels, levels64, levels_lit32, bar=2e-5):
    """The prefiltered levels of the product against the float64 ones: per level max-norm relative error; above `bar` a level passes by
    the truth-leg rule only (no further from float64 than the levels built with the reference's fp32 filter weights, x 1.5, and those
    beyond half the bar themselves).  Returns (rows, ok)."""
    rows, ok = [], True
    for i, (a, b) in enumerate(zip(product_levels, levels64)):
        b = np.asarray(b, dtype=np.float64)
        scale = max(float(np.abs(b).max()), 1e-30)
        e = float(np.abs(np.asarray(a, dtype=np.float64).reshape(b.shape) - b).max() / scale)
        row = {"res": int(b.shape[1]), "err": e, "rule": "bar"}
        if e > bar:
            e_lit = float(np.abs(np.asarray(levels_lit32[i], dtype=np.float64) - b).max() / scale) if levels_lit32 is not None else 0.0
            row["lit32_err"] = e_lit
            good = e <= 1.5 * e_lit and e_lit > 0.5 * bar
            row["rule"] = "truth-leg" if good else "FAIL"
            ok = ok and good
        rows.append(row)
    return rows, ok
