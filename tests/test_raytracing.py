"""Mesh ray queries: host-built hierarchy (CPU checks) and GPU traversal vs the brute-force oracle."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import trace_oracle  # noqa: E402


from materialrefgs_amd.synthetic import sphere_mesh  # noqa: E402


def scene(seed=0, n_lat=16, n_lon=24):
    """Bumpy unit sphere + a big ground quad grid + a small inner sphere: hits, misses, grazing rays, occlusion."""
    v1, t1 = sphere_mesh(n_lat, n_lon, 1.0, 0.03, seed)
    v2, t2 = sphere_mesh(6, 8, 0.3, 0.0, seed + 1)
    v2 = v2 + np.array([1.8, 0.2, 0.1], dtype=np.float32)
    g = np.linspace(-3, 3, 9).astype(np.float32)
    gx, gy = np.meshgrid(g, g, indexing="ij")
    v3 = np.stack([gx, gy, np.full_like(gx, -1.2)], -1).reshape(-1, 3)
    t3 = []
    for i in range(8):
        for j in range(8):
            a, b, c, d = i * 9 + j, i * 9 + j + 1, (i + 1) * 9 + j, (i + 1) * 9 + j + 1
            t3 += [(a, c, b), (b, c, d)]                        # normal +z
    t3 = np.asarray(t3, dtype=np.int32)
    v = np.concatenate([v1, v2, v3]).astype(np.float32)
    t = np.concatenate([t1, t2 + len(v1), t3 + len(v1) + len(v2)]).astype(np.int32)
    return v, t


def rays(n, seed=0):
    rng = np.random.default_rng(seed)
    o = rng.normal(size=(n, 3)).astype(np.float32)
    o = (o / np.linalg.norm(o, axis=1, keepdims=True) * rng.uniform(1.5, 4.0, size=(n, 1))).astype(np.float32)
    target = rng.normal(size=(n, 3)).astype(np.float32) * 0.8
    d = target - o
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    # special rays: axis-parallel (zero direction components), origins on the surface, rays from inside (back faces only)
    o[:8] = np.array([[3, 0, 0], [0, 3, 0], [0, 0, 3], [-3, 0, 0], [0.5, 0.5, 3], [0, 0, 0], [0, 0, 0], [0.1, 0.1, -1.2]], dtype=np.float32)
    d[:8] = np.array([[-1, 0, 0], [0, -1, 0], [0, 0, -1], [1, 0, 0], [0, 0, -1], [0, 0, 1], [-1, 0, 0], [0, 0, -1]], dtype=np.float32)
    return o, d


def _lib():
    from materialrefgs_amd import _lib as L
    return L


def _build(v, t):
    L = _lib()
    lib = L.lib()
    nbytes = lib.mrgs_bvh_bytes(len(t))
    blob = np.zeros(nbytes, dtype=np.uint8)
    rc = lib.mrgs_bvh_build(ctypes.c_void_p(v.ctypes.data), len(v), ctypes.c_void_p(t.ctypes.data), len(t),
                            ctypes.c_void_p(blob.ctypes.data), nbytes)
    return rc, blob


def _views(blob, n):
    cap = n // 3 + 2
    align = lambda x: (x + 255) // 256 * 256   # noqa: E731
    tris_off = align(cap * 128)
    perm_off = align(tris_off + n * 48)
    nodes = blob[:cap * 128].view(np.float32).reshape(cap, 32)
    codes = blob[:cap * 128].view(np.int32).reshape(cap, 32)[:, 24:28]
    rec = blob[tris_off:tris_off + n * 48].view(np.float32).reshape(n, 12)
    perm = blob[perm_off:perm_off + n * 4].view(np.int32)
    return nodes, codes, rec, perm


# ---------------------------------------------------------------- CPU: the host-side build
@pytest.mark.parametrize("n_lat,n_lon", [(3, 4), (16, 24), (40, 60)])
def test_bvh_build_partitions_every_triangle_and_boxes_enclose(n_lat, n_lon):
    v, t = scene(1, n_lat, n_lon)
    rc, blob = _build(v, t)
    assert rc == 0
    n = len(t)
    nodes, codes, rec, perm = _views(blob, n)
    assert sorted(perm.tolist()) == list(range(n))                      # a permutation: every triangle exactly once
    a = v[t[perm, 0]]
    np.testing.assert_array_equal(rec[:, 0:3], a)
    np.testing.assert_array_equal(rec[:, 3:6], v[t[perm, 1]] - a)
    np.testing.assert_array_equal(rec[:, 6:9], v[t[perm, 2]] - a)
    seen = np.zeros(n, dtype=np.int32)
    max_depth = 0

    def walk(node, lo, hi, depth):
        nonlocal max_depth
        max_depth = max(max_depth, depth)
        for c in range(4):
            code = int(codes[node, c])
            if code == 0x7FFFFFFF:
                continue
            clo = nodes[node, [0 + c, 4 + c, 8 + c]]
            chi = nodes[node, [12 + c, 16 + c, 20 + c]]
            assert np.all(clo >= lo - 1e-4) and np.all(chi <= hi + 1e-4)          # nested (up to the padding)
            if code < 0:
                first, cnt = (~code) >> 3, ((~code) & 7) + 1
                assert cnt <= 4
                seen[first:first + cnt] += 1
                tv = v[t[perm[first:first + cnt]]].reshape(-1, 3)
                assert np.all(tv >= clo) and np.all(tv <= chi)                    # leaf box encloses its triangles
            else:
                walk(code, clo, chi, depth + 1)

    walk(0, np.full(3, -np.inf), np.full(3, np.inf), 1)
    assert np.all(seen == 1)
    assert 3 * max_depth + 1 <= 32


def test_bvh_build_rejects_bad_input():
    v, t = scene(0, 3, 4)
    L = _lib()
    lib = L.lib()
    bad = t.copy()
    bad[5, 1] = len(v)                                                   # vertex index out of range
    rc, _ = _build(v, bad)
    assert rc == 1
    blob = np.zeros(64, dtype=np.uint8)
    rc = lib.mrgs_bvh_build(ctypes.c_void_p(v.ctypes.data), len(v), ctypes.c_void_p(t.ctypes.data), len(t),
                            ctypes.c_void_p(blob.ctypes.data), 64)
    assert rc == L.MRGS_E_WORKSPACE
    assert lib.mrgs_bvh_bytes(0) == 0


def test_trace_oracle_known_answers():
    """One triangle in the plane z = 0, normal +z: hit from above, back face from below, miss outside, t >= 10 dropped."""
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], dtype=np.float32)
    t = np.array([[0, 1, 2]], dtype=np.int32)
    o = np.array([[0.25, 0.25, 2], [0.25, 0.25, -2], [2, 2, 2], [0.25, 0.25, 11], [0.25, 0.25, 0.0]], dtype=np.float32)
    d = np.array([[0, 0, -1], [0, 0, 1], [0, 0, -1], [0, 0, -1], [0, 0, -1]], dtype=np.float32)
    pos, nrm, depth, ids = trace_oracle.trace(v, t, o, d)
    np.testing.assert_array_equal(depth, np.array([2, 10, 10, 10, 0], dtype=np.float32))
    np.testing.assert_array_equal(ids, [0, -1, -1, -1, 0])
    np.testing.assert_array_equal(nrm[0], [0, 0, 1])
    np.testing.assert_array_equal(nrm[1], [0, 0, 0])
    np.testing.assert_array_equal(pos[0], [0.25, 0.25, 0])
    np.testing.assert_array_equal(pos[1], [0.25, 0.25, 8])               # a miss still advances by MAX_DIST (bvh.cu:706-708)


# ---------------------------------------------------------------- GPU: traversal vs brute force
def _check_against_oracle(v, t, o, d, inplace=False):
    from materialrefgs_amd.raytracing import RayTracer
    rt = RayTracer(v, t)
    to, td = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
    pos, nrm, depth, ids = rt.trace(to.clone(), td.clone(), inplace=inplace, return_faceids=True)
    pos, nrm, depth, ids = pos.cpu().numpy(), nrm.cpu().numpy(), depth.cpu().numpy(), ids.cpu().numpy()
    rpos, rnrm, rdepth, rids = trace_oracle.trace(v, t, o, d)
    np.testing.assert_array_equal(depth, rdepth)                         # bit-exact closest distance
    np.testing.assert_array_equal(pos, rpos)
    same = ids == rids
    if not same.all():                                                   # only exact ties may pick another triangle
        idx = np.nonzero(~same)[0]
        tt = trace_oracle.hit_time(v, t, o[idx], d[idx], ids[idx])
        np.testing.assert_array_equal(tt, rdepth[idx])
    np.testing.assert_array_equal(nrm[same], rnrm[same])
    return depth, ids


@pytest.mark.gpu
@pytest.mark.parametrize("inplace", [False, True])
def test_trace_matches_brute_force(inplace):
    v, t = scene(0)
    o, d = rays(6000, 1)
    depth, ids = _check_against_oracle(v, t, o, d, inplace)
    assert (depth < 10).mean() > 0.3 and (depth >= 10).mean() > 0.02     # the case has both hits and misses
    assert depth[5] >= 10 and depth[6] >= 10                             # rays from inside the sphere: back faces only


@pytest.mark.gpu
def test_trace_reflection_rays_from_the_surface():
    """The way the shading pass uses it (utils/refl_utils.py:381-391): origins on the mesh, mirrored view directions."""
    v, t = scene(2, 24, 36)
    rng = np.random.default_rng(5)
    f = rng.integers(0, len(t), size=4000)
    w = rng.dirichlet([1, 1, 1], size=4000).astype(np.float32)
    p = (v[t[f, 0]] * w[:, :1] + v[t[f, 1]] * w[:, 1:2] + v[t[f, 2]] * w[:, 2:]).astype(np.float32)
    n = np.cross(v[t[f, 1]] - v[t[f, 0]], v[t[f, 2]] - v[t[f, 0]])
    n = (n / np.linalg.norm(n, axis=1, keepdims=True)).astype(np.float32)
    cam = np.array([0, -4, 1.5], dtype=np.float32)
    wo = cam - p
    wo = wo / np.linalg.norm(wo, axis=1, keepdims=True)
    r = (2 * n * (n * wo).sum(1, keepdims=True) - wo).astype(np.float32)
    r = (r / np.linalg.norm(r, axis=1, keepdims=True)).astype(np.float32)
    _check_against_oracle(v, t, p, r)


@pytest.mark.gpu
def test_trace_large_mesh_properties():
    """~1 M triangles, 640 000 rays (the C4 mesh size, C2 image size): identities that need no oracle."""
    from materialrefgs_amd.raytracing import RayTracer
    v, t = sphere_mesh(700, 720, 1.0, 0.0)
    assert len(t) > 1_000_000
    rt = RayTracer(v, t)
    g = torch.Generator(device="cuda").manual_seed(0)
    d = torch.nn.functional.normalize(torch.randn(640_000, 3, device="cuda", generator=g), dim=-1)
    o = -3.0 * d                                                          # towards the centre from radius 3
    pos, nrm, depth, ids = rt.trace(o, d, return_faceids=True)
    polar = d[:, 2].abs() > 0.9995                                        # the open caps at the poles
    hit = depth < 10
    missed = (~hit) & (~polar)
    assert float(missed.float().mean()) < 1e-3                            # the reference's edge test is not watertight: a few rays slip through
    # a sample of rays (all of the slipped ones first) against the brute-force oracle at full mesh size
    pick = torch.cat([torch.nonzero(missed)[:16, 0], torch.arange(0, 640_000, 13_337, device="cuda")]).cpu().numpy()
    _, _, rdepth, rids = trace_oracle.trace(v, t, o[pick].cpu().numpy(), d[pick].cpu().numpy(), chunk=8)
    np.testing.assert_array_equal(depth[pick].cpu().numpy(), rdepth)
    assert float((depth[hit] - 2.0).abs().max()) < 2e-3                   # unit sphere seen from radius 3 (faceted)
    assert float((pos[hit].norm(dim=-1) - 1.0).abs().max()) < 2e-3
    assert float((nrm[hit] * (-d[hit])).sum(-1).min()) > 0.99             # outward normals face the ray origin
    tv = torch.from_numpy(v).cuda()[torch.from_numpy(t).cuda()[ids[hit].long()].long()]        # [n,3,3]
    nn = torch.linalg.cross(tv[:, 1] - tv[:, 0], tv[:, 2] - tv[:, 0])
    assert float(((pos[hit] - tv[:, 0]) * torch.nn.functional.normalize(nn, dim=-1)).sum(-1).abs().max()) < 1e-4   # hit lies in its triangle's plane
    # rays leaving the sphere from inside see only back faces: all miss
    _, _, depth2 = rt.trace(torch.zeros_like(d), d)
    assert bool((depth2 >= 10).all())


def _camera_rays_unnormalized(H, W, K, R, T):
    """sample_camera_rays_unnormalize (utils/refl_utils.py:75-93) restated with torch fp32 ops for the test."""
    i, j = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing="xy")
    xy1 = np.stack([i, j, np.ones_like(i)], axis=2)
    pixel_camera = torch.tensor(np.dot(xy1, np.linalg.inv(K.astype(np.float32)).T)).to(R.device)
    Rt = R.T
    rays_o = (-Rt.T @ T.unsqueeze(-1)).flatten()
    pixel_world = (pixel_camera - T[None, None]).reshape(-1, 3) @ Rt
    return (pixel_world - rays_o[None]).reshape(H, W, 3), rays_o


@pytest.mark.gpu
def test_fused_visibility_matches_traced_mirror_rays():
    """mrgs_bvh_visibility against the reference's sequence (utils/refl_utils.py:379-391) executed with torch ops + RayTracer.trace."""
    from materialrefgs_amd.raytracing import RayTracer
    from materialrefgs_amd.synthetic import orbit_camera
    H, W = 120, 160
    v, t = scene(3, 24, 36)
    rt = RayTracer(v, t)
    cam = orbit_camera(2, H, W).to("cuda")
    g = torch.Generator(device="cuda").manual_seed(11)
    rays_cam, rays_o = _camera_rays_unnormalized(H, W, np.asarray(cam.HWK[2]), cam.R.float(), cam.T.float())
    # surface points = first hit of the pixel rays against the mesh itself; z-depth = t (rays_cam has unit camera-space z)
    _, nrm, t_hit = rt.trace(rays_o.expand(H * W, 3).contiguous(), rays_cam.reshape(-1, 3).contiguous())
    # the tracer wants unit directions for distances, but t is a parameter along rays_cam here, which is what surf_depth multiplies
    hit = (t_hit < 10).view(H, W)
    surf_depth = torch.where(hit, t_hit.view(H, W), torch.full((H, W), 3.0, device="cuda")).view(1, H, W) * 0.999   # just in front of the surface
    normal = torch.where(hit[..., None], nrm.view(H, W, 3), torch.nn.functional.normalize(torch.randn(H, W, 3, device="cuda", generator=g), dim=-1))
    normal = torch.nn.functional.normalize(normal + 0.2 * torch.randn(H, W, 3, device="cuda", generator=g), dim=-1)
    alpha = (torch.rand(H, W, 1, device="cuda", generator=g) > 0.2).float() * torch.rand(H, W, 1, device="cuda", generator=g)
    vis = rt.visibility(cam.HWK, cam.R, cam.T, normal, alpha, surf_depth)
    # reference sequence
    mask = (alpha > 0)[..., 0]
    w_o = torch.nn.functional.normalize(-rays_cam, dim=-1)
    refl = torch.nn.functional.normalize(2 * normal * (w_o * normal).sum(-1, keepdim=True) - w_o, dim=-1)
    inter = rays_o + surf_depth.permute(1, 2, 0) * rays_cam
    _, _, depth = rt.trace(inter[mask], refl[mask])
    ref = torch.ones_like(alpha)
    ref[mask] = (depth >= 10).float().unsqueeze(-1)
    assert vis.shape == (H, W, 1)
    assert bool((vis[~mask] == 1).all())
    mism = float((vis != ref).float().mean())
    assert mism < 2e-3, mism                                            # rounding of the ray set-up at silhouettes only
    blocked = float((ref[mask] == 0).float().mean())
    assert 0.02 < blocked < 0.98, blocked                               # the case exercises both outcomes


@pytest.mark.gpu
def test_render_surfel_indirect_branch():
    """opt.indirect with a mesh tracer: reference keys, visibility in {0,1}, blend formula, gradients to the indirect SH."""
    from types import SimpleNamespace
    from materialrefgs_amd.raytracing import RayTracer
    from materialrefgs_amd.renderer import SurfelModel, render_surfel
    from materialrefgs_amd.shading import EnvLight
    from materialrefgs_amd.synthetic import make_shell_scene, orbit_camera
    dev = "cuda"
    P, H, W = 3000, 96, 128
    sc = make_shell_scene(P, S=0, seed=1, radius_px=6.0, image_size=128).to(dev)
    g = torch.Generator().manual_seed(0)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)   # noqa: E731
    env = EnvLight(device=dev, trainable=True)
    with torch.no_grad():
        env.base.copy_(rnd(6, 128, 128, 3))
    env.build_mips()
    inv_sig = lambda x: torch.log(x / (1 - x))   # noqa: E731
    pc = SurfelModel(sc.means3D.clone(), torch.log(sc.scales), sc.rotations.clone(), inv_sig(sc.opacities.clamp(1e-4, 1 - 1e-4)),
                     sc.shs[:, :1].clone(), sc.shs[:, 1:].clone(), refl_strength=rnd(P, 1), roughness=rnd(P, 1), ori_color=rnd(P, 3),
                     indirect_dc=rnd(P, 1, 3).abs() * 0.5, indirect_rest=rnd(P, 15, 3) * 0.01, envmap=env)
    for p_ in pc.parameters():
        p_.requires_grad_(True)
    # the shell scene lives on the unit sphere: a slightly larger inward-facing... the tracer culls back faces, so use an
    # outward-facing sphere of radius 0.9 plus a blocker sphere off to the side
    v1, t1 = sphere_mesh(24, 36, 0.9)
    v2, t2 = sphere_mesh(12, 16, 0.8)
    v2 = v2 + np.array([0.0, 2.2, 0.0], dtype=np.float32)
    pc.ray_tracer = RayTracer(np.concatenate([v1, v2]), np.concatenate([t1, t2 + len(v1)]))
    cam = orbit_camera(1, H, W).to(dev)
    pipe = SimpleNamespace(depth_ratio=0.0, debug=False)
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    out = render_surfel(cam, pc, pipe, bg, srgb=False, opt=SimpleNamespace(indirect=True))
    for k in ("render", "specular_map", "visibility", "indirect_light", "direct_light", "indirect_color", "specular_weight", "surf_depth"):
        assert k in out, k
    vis = out["visibility"]
    assert vis.shape == (1, H, W) and bool(((vis == 0) | (vis == 1)).all())
    assert 0.0 < float((vis == 0).float().mean()) < 1.0
    alpha = out["rend_alpha"]
    light = out["direct_light"] * vis + (1 - vis) * out["indirect_light"]
    want = light * alpha * out["specular_weight"].permute(2, 0, 1)
    assert torch.allclose(out["specular_map"], want, atol=1e-6)
    out["render"].mean().backward()
    assert pc._indirect_dc.grad is not None and float(pc._indirect_dc.grad.abs().sum()) > 0
    assert env.base.grad is not None and torch.isfinite(env.base.grad).all()


@pytest.mark.gpu
def test_indirect_blend_kernel_matches_torch_ops():
    """mrgs_indirect_blend_* against the reference's expressions (utils/refl_utils.py:393-401) evaluated with torch autograd."""
    from materialrefgs_amd.shading import _IndirectBlend
    H, W = 37, 53
    g = torch.Generator(device="cuda").manual_seed(4)
    r = lambda *s: torch.rand(*s, device="cuda", generator=g)   # noqa: E731
    direct, weight = r(3, H, W).requires_grad_(True), r(H, W, 3).requires_grad_(True)
    feat = r(8, H, W).requires_grad_(True)
    alpha_chw = r(1, H, W).requires_grad_(True)
    vis = (r(H, W, 1) > 0.4).float()
    indirect, alpha = feat[5:8].permute(1, 2, 0), alpha_chw.permute(1, 2, 0)            # strided views, as render_surfel passes them
    spec, ic = _IndirectBlend.apply(direct, weight, indirect, alpha, vis)
    light = direct.permute(1, 2, 0) * vis + (1 - vis) * indirect
    spec_ref = (light * alpha * weight).permute(2, 0, 1)
    ic_ref = ((1 - vis) * indirect * alpha * weight).permute(2, 0, 1)
    torch.testing.assert_close(spec, spec_ref, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(ic, ic_ref, rtol=1e-6, atol=1e-7)
    gs, gi = r(3, H, W), r(3, H, W)
    got = torch.autograd.grad([spec, ic], [direct, weight, feat, alpha_chw], [gs, gi])
    want = torch.autograd.grad([spec_ref, ic_ref], [direct, weight, feat, alpha_chw], [gs, gi])
    for a, b in zip(got, want):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_trace_tiny_and_degenerate_meshes():
    """Nine triangles (the reference's minimum is > 8) incl. zero-area and duplicated ones, rays parallel to faces, zero rays."""
    from materialrefgs_amd.raytracing import RayTracer
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0], [0, 0, 1], [1, 0, 1], [0, 1, 1], [1, 1, 1], [0.5, 0.5, 0.5], [2, 2, 2]], dtype=np.float32)
    t = np.array([[0, 2, 1], [1, 2, 3], [4, 5, 6], [5, 7, 6], [0, 1, 4], [1, 5, 4], [8, 8, 8], [9, 9, 1], [0, 2, 1]], dtype=np.int32)
    o, d = rays(500, 3)
    o[:4] = np.array([[0.25, 0.25, 3], [0.25, 0.25, -3], [-1, 0.5, 0.0], [0.5, 0.5, 0.5]], dtype=np.float32)
    d[:4] = np.array([[0, 0, -1], [0, 0, 1], [1, 0, 0], [0, 0, 1]], dtype=np.float32)        # ray 2 runs inside the plane z = 0
    _check_against_oracle(v, t, o * 0.7, d)
    rt = RayTracer(v, t)
    pos, nrm, depth = rt.trace(torch.zeros(0, 3, device="cuda"), torch.zeros(0, 3, device="cuda"))
    assert pos.shape == (0, 3) and depth.shape == (0,)
    with pytest.raises(AssertionError):
        RayTracer(v, t[:8])
