"""Dense torch statement of the surfel ray tracer -- TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke, bench.py's cpu leg).

PARITY UNPINNED against the reference's binary: the tracer behind gaussian_renderer/optix_utils.py:185-197 is the un-vendored OptiX
extension `diff_surfel_tracing`; no source, test or vector of it exists in /root/reference.  What the reference does fix is restated
here from its call site:
  * the primitive: the quad  mean +- 3 s_u r_u +- 3 s_v r_v  of every surfel   (get_disks, optix_utils.py:36-66 -> quad_vertices)
  * ray directions are not normalised, depth is the ray parameter               (:121-123)
  * outputs rgb, dpt, acc, norm, dist, aux, wet                                 (:185-197, 218-233)
The compositing is DEFINED (csrc/mrgs_surfel_trace.hip states the same) as the vendored 2DGS rasterizer's per-pixel loop
(submodules/diff-surfel-rasterization/cuda_rasterizer/forward.cu:366-420) applied along a ray: gaussian weight from the hit's local
(u, v), alpha = min(0.99, o G), skip below 1/255, front to back by (t, index), the hit that would push T below 1e-4 ends the ray.
Every ray is tested against every surfel (no hierarchy); torch autograd differentiates it, which is the independent check of the
kernels' hand-written backward.
"""
import torch


def rotation_matrix(q):
    """utils/general_utils.py:80-99 build_rotation: [P,4] (w,x,y,z), normalised inside."""
    q = q / q.norm(dim=1, keepdim=True)
    w, x, y, z = q.unbind(1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                     2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                     2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=1)
    return R.reshape(-1, 3, 3)


def quad_vertices(means, scales, rotations, scale_modifier=1.0):
    """get_disks (optix_utils.py:36-66): splat2world = [s_u r_u, s_v r_v, 0, mean]; corners (-3,3), (-3,-3), (3,3), (3,-3) -> [P,4,3]."""
    R = rotation_matrix(rotations)
    su = (scales[:, 0:1] * scale_modifier) * R[:, :, 0]
    sv = (scales[:, 1:2] * scale_modifier) * R[:, :, 1]
    corners = [(-3.0, 3.0), (-3.0, -3.0), (3.0, 3.0), (3.0, -3.0)]
    return torch.stack([means + cu * su + cv * sv for cu, cv in corners], dim=1)


def trace_dense(ray_o, ray_d, means, scales, rotations, opacities, colors, others, bg, scale_modifier=1.0):
    """ray_o / ray_d [R,3]; means [P,3], scales [P,2], rotations [P,4], opacities [P,1], colors [P,3], others [P,2], bg [3].
    Returns dict(rgb [R,3], dpt [R], acc [R], norm [R,3], dist [R], aux [R,2], wet [P], T [R], hits [R])."""
    Rm = rotation_matrix(rotations)
    s = scales * scale_modifier
    a = Rm[:, :, 0] / s[:, 0:1]
    b = Rm[:, :, 1] / s[:, 1:2]
    n = Rm[:, :, 2]
    o, d = ray_o[:, None, :], ray_d[:, None, :]                    # [R,1,3]
    den = (n[None] * d).sum(-1)                                    # [R,P]
    num = (n[None] * (means[None] - o)).sum(-1)
    t = num / den
    p = (o + t[..., None] * d) - means[None]
    u = (a[None] * p).sum(-1)
    v = (b[None] * p).sum(-1)
    G = torch.exp(-0.5 * (u * u + v * v))
    alpha = torch.clamp(opacities[:, 0][None] * G, max=0.99)
    valid = (t > 0) & (u.abs() <= 3.0) & (v.abs() <= 3.0) & (alpha >= 1.0 / 255.0) & torch.isfinite(t)
    t_key = torch.where(valid, t, torch.full_like(t, float("inf"))).detach()
    order = torch.argsort(t_key, dim=1, stable=True)               # ties by surfel index
    take = lambda x: torch.gather(x, 1, order)
    vs = take(valid)
    al = torch.where(vs, take(alpha), torch.zeros_like(t))
    ts = torch.where(vs, take(t), torch.zeros_like(t))
    one_m = 1.0 - al
    T_before = torch.cumprod(torch.cat([torch.ones_like(one_m[:, :1]), one_m[:, :-1]], dim=1), dim=1)
    stop = vs & ((T_before * one_m) < 1e-4)
    # T only moves on blended hits; after the first stop nothing is blended, so the cumprod above is exact up to that point
    blended = vs & (torch.cumsum(stop.to(torch.int64), dim=1) == 0)
    w = torch.where(blended, al * T_before, torch.zeros_like(t))
    T_final = torch.where(blended, one_m, torch.ones_like(t)).prod(dim=1)
    sgn = torch.where(take(den) > 0, -torch.ones_like(t), torch.ones_like(t))
    nf = sgn[..., None] * n[order]                                  # [R,P,3]
    rgb = (w[..., None] * colors[order]).sum(1) + T_final[:, None] * bg[None]
    aux = (w[..., None] * others[order]).sum(1)
    norm = (w[..., None] * nf).sum(1)
    dpt = (w * ts).sum(1)
    acc = w.sum(1)
    ex = lambda x: torch.cumsum(x, dim=1) - x                      # sums over the hits before
    dist = (w * (ts * ts * ex(w) + ex(w * ts * ts) - 2.0 * ts * ex(w * ts))).sum(1)
    wet = torch.zeros(means.shape[0], dtype=t.dtype, device=t.device).scatter_add(0, order.reshape(-1), w.reshape(-1))
    return dict(rgb=rgb, dpt=dpt, acc=acc, norm=norm, dist=dist, aux=aux, wet=wet, T=T_final, hits=blended.sum(1))


def brute_force_hits(ray_o, ray_d, means, scales, rotations, opacities, scale_modifier=1.0):
    """(t [R,P] with inf where there is no accepted hit): what a hierarchy may only prune, never change."""
    with torch.no_grad():
        Rm = rotation_matrix(rotations)
        s = scales * scale_modifier
        a, b, n = Rm[:, :, 0] / s[:, 0:1], Rm[:, :, 1] / s[:, 1:2], Rm[:, :, 2]
        o, d = ray_o[:, None, :], ray_d[:, None, :]
        t = (n[None] * (means[None] - o)).sum(-1) / (n[None] * d).sum(-1)
        p = (o + t[..., None] * d) - means[None]
        u, v = (a[None] * p).sum(-1), (b[None] * p).sum(-1)
        alpha = torch.clamp(opacities[:, 0][None] * torch.exp(-0.5 * (u * u + v * v)), max=0.99)
        valid = (t > 0) & (u.abs() <= 3.0) & (v.abs() <= 3.0) & (alpha >= 1.0 / 255.0) & torch.isfinite(t)
        return torch.where(valid, t, torch.full_like(t, float("inf")))
