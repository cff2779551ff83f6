"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy, float32, brute force over all triangles) of the reference's mesh ray
query, the checker of csrc/mrgs_bvh.hip.  Never imported by the product path.

Follows submodules/raytracing: Triangle::ray_intersect (include/raytracing/triangle.cuh:27-45), the closest-hit selection of
TriangleBvh4::ray_intersect (src/bvh.cu:259-302: mint starts at MAX_DIST = 10, strict `t < mint`), raytrace_kernel
(src/bvh.cu:694-720: depth, position = o + depth d, unit face normal or zero).  The hierarchy of the reference only prunes, so
its answer is this minimum over all triangles.  PARITY UNPINNED: the reference is CUDA + Eigen (cannot be built in this image) and
ships no test vectors for this path; every float32 operation below is written in the order of the cited lines.
"""
import numpy as np

MAX_DIST = np.float32(10.0)        # bvh.cu:36
f32 = np.float32


def _cross(a, b):
    return (a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1], a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
            a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0])


def _dot(ax, ay, az, bx, by, bz):
    return (ax * bx + ay * by) + az * bz


def trace(vertices, triangles, rays_o, rays_d, chunk=256):
    """Returns (positions [N,3], normals [N,3], depth [N], face_ids [N], tmat-free)."""
    v = np.asarray(vertices, dtype=np.float32)
    t = np.asarray(triangles, dtype=np.int64)
    a, b, c = v[t[:, 0]], v[t[:, 1]], v[t[:, 2]]
    e1, e2 = b - a, c - a                                           # v1v0, v2v0
    nx, ny, nz = _cross(e1, e2)                                     # n = v1v0 x v2v0
    ro = np.asarray(rays_o, dtype=np.float32).reshape(-1, 3)
    rd = np.asarray(rays_d, dtype=np.float32).reshape(-1, 3)
    N = ro.shape[0]
    depth = np.full(N, MAX_DIST, dtype=np.float32)
    ids = np.full(N, -1, dtype=np.int64)
    allt = None
    with np.errstate(all="ignore"):
        for s in range(0, N, chunk):
            o, d = ro[s:s + chunk, None, :], rd[s:s + chunk, None, :]
            rx, ry, rz = o[..., 0] - a[None, :, 0], o[..., 1] - a[None, :, 1], o[..., 2] - a[None, :, 2]      # rov0
            dx, dy, dz = d[..., 0], d[..., 1], d[..., 2]
            dn = _dot(dx, dy, dz, nx[None], ny[None], nz[None])
            qx, qy, qz = ry * dz - rz * dy, rz * dx - rx * dz, rx * dy - ry * dx                              # rov0 x rd
            inv = f32(1.0) / dn
            u = inv * -_dot(qx, qy, qz, e2[None, :, 0], e2[None, :, 1], e2[None, :, 2])
            w = inv * _dot(qx, qy, qz, e1[None, :, 0], e1[None, :, 1], e1[None, :, 2])
            tt = inv * -_dot(nx[None], ny[None], nz[None], rx, ry, rz)
            miss = (dn >= 0) | (u < 0) | (u > 1) | (w < 0) | ((u + w) > 1) | (tt < 0)
            miss |= ~(tt < MAX_DIST)                                                                           # also drops NaN
            tt = np.where(miss, np.float32(np.inf), tt).astype(np.float32)
            j = np.argmin(tt, axis=1)
            best = tt[np.arange(tt.shape[0]), j]
            hit = best < MAX_DIST
            depth[s:s + chunk] = np.where(hit, best, MAX_DIST)
            ids[s:s + chunk] = np.where(hit, j, -1)
    pos = (ro + depth[:, None] * rd).astype(np.float32)
    nrm = np.zeros((N, 3), dtype=np.float32)
    h = ids >= 0
    n3 = np.stack([nx, ny, nz], axis=1)[ids[h]]
    ln = np.sqrt((n3[:, 0] * n3[:, 0] + n3[:, 1] * n3[:, 1]) + n3[:, 2] * n3[:, 2]).astype(np.float32)
    nrm[h] = n3 / ln[:, None]
    return pos, nrm, depth, ids


def hit_time(vertices, triangles, rays_o, rays_d, face_ids):
    """t of ray i against triangle face_ids[i] (inf for a rejected hit): lets a test accept a different id at an exact tie."""
    v = np.asarray(vertices, dtype=np.float32)
    t = np.asarray(triangles, dtype=np.int64)
    out = np.full(len(face_ids), np.inf, dtype=np.float32)
    for i, f in enumerate(face_ids):
        if f < 0:
            continue
        _, _, dpt, ids = trace(v, t[f:f + 1], rays_o[i:i + 1], rays_d[i:i + 1])
        out[i] = dpt[0] if ids[0] >= 0 else np.inf
    return out
