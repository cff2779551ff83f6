"""Mesh ray queries behind the reference's `RayTracer` (submodules/raytracing/raytracing/raytracer.py:8-56 and the variant the
training code constructs, raytracing_brdf/raytracer.py:18-123: `RayTracer(vertices, triangles)`, `.trace(rays_o, rays_d,
inplace=False[, return_faceids=False]) -> positions, face_normals, depth[, triangle ids]`).  The hierarchy is built on the host
by libmrgs.so (`mrgs_bvh_build`, csrc/mrgs_bvh.hip) and traversed on the GPU (`mrgs_bvh_trace`); depth = 10 marks a miss
(utils/refl_utils.py:390-391 tests `depth >= 10`).  Face ids index the `triangles` array that was passed in (the reference
returns build-order ids and maps them back with `trans_ids`; here the mapping happens inside the kernel)."""
import ctypes

import numpy as np
import torch

from . import _lib


class RayTracer:
    def __init__(self, vertices, triangles, device=None):
        if torch.is_tensor(vertices):
            vertices = vertices.detach().cpu().numpy()
        if torch.is_tensor(triangles):
            triangles = triangles.detach().cpu().numpy()
        vertices = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 3)
        triangles = np.ascontiguousarray(triangles, dtype=np.int32).reshape(-1, 3)
        assert triangles.shape[0] > 8, "BVH needs at least 8 triangles."          # raytracer.py:15
        self.n_triangles = int(triangles.shape[0])
        lib = _lib.lib()
        nbytes = lib.mrgs_bvh_bytes(self.n_triangles)
        blob = torch.empty(nbytes, dtype=torch.uint8)
        _lib.check(lib.mrgs_bvh_build(ctypes.c_void_p(vertices.ctypes.data), vertices.shape[0], ctypes.c_void_p(triangles.ctypes.data),
                                      self.n_triangles, ctypes.c_void_p(blob.data_ptr()), nbytes))
        self.blob_host = blob
        self.device = torch.device(device) if device is not None else (torch.device("cuda") if torch.cuda.is_available() else None)
        self.blob = blob.to(self.device) if self.device is not None else None

    def trace(self, rays_o, rays_d, inplace=False, return_faceids=False):
        if self.blob is None:
            raise RuntimeError("materialrefgs_amd.raytracing needs a GPU (libmrgs.so has no CPU traversal)")
        rays_o = rays_o.float().contiguous()
        rays_d = rays_d.float().contiguous()
        if not rays_o.is_cuda:
            rays_o = rays_o.to(self.device)
        if not rays_d.is_cuda:
            rays_d = rays_d.to(self.device)
        prefix = rays_o.shape[:-1]
        rays_o, rays_d = rays_o.view(-1, 3), rays_d.view(-1, 3)
        N = rays_o.shape[0]
        positions = rays_o if inplace else torch.empty_like(rays_o)
        face_normals = rays_d if inplace else torch.empty_like(rays_d)
        depth = torch.empty(N, dtype=torch.float32, device=rays_o.device)
        ids = torch.empty(N, dtype=torch.int32, device=rays_o.device) if return_faceids else None
        p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None   # noqa: E731
        with _lib.guard(rays_o.device):
            st = _lib.stream_ptr(rays_o.device)
            _lib.check(_lib.lib().mrgs_bvh_trace(p(self.blob), self.n_triangles, N, p(rays_o), p(rays_d), p(positions), p(face_normals),
                                                 p(depth), p(ids), st))
        positions, face_normals, depth = positions.view(*prefix, 3), face_normals.view(*prefix, 3), depth.view(*prefix)
        if not return_faceids:
            return positions, face_normals, depth
        return positions, face_normals, depth, ids.view(*prefix)

    def visibility(self, HWK, R, T, normal_map, render_alpha, surf_depth):
        """The visibility block of get_specular_color_surfel (utils/refl_utils.py:379-391) as one launch: normal_map [H,W,3],
        render_alpha [H,W,1], surf_depth [1,H,W] -> visibility [H,W,1] (1 = the mirror ray is free for 10 units or alpha <= 0).
        Not differentiable (the reference's is a comparison)."""
        if self.blob is None:
            raise RuntimeError("materialrefgs_amd.raytracing needs a GPU (libmrgs.so has no CPU traversal)")
        H, W, K = HWK
        Kinv = np.linalg.inv(np.asarray(K, dtype=np.float32)).astype(np.float32).reshape(-1)
        kin = (ctypes.c_float * 9)(*[float(x) for x in Kinv])
        nm, al = normal_map.detach().float(), render_alpha.detach().float()
        sd = surf_depth.detach().float().contiguous()
        Rc, Tc = R.detach().float().contiguous(), T.detach().float().contiguous()
        vis = torch.empty((H, W, 1), dtype=torch.float32, device=nm.device)
        m_n = _lib.MrgsStridedMap(nm.data_ptr(), nm.stride(0), nm.stride(1), nm.stride(2))
        m_a = _lib.MrgsStridedMap(al.data_ptr(), al.stride(0), al.stride(1), al.stride(2) if al.dim() > 2 else 0)
        with _lib.guard(nm.device):
            st = _lib.stream_ptr(nm.device)
            _lib.check(_lib.lib().mrgs_bvh_visibility(ctypes.c_void_p(self.blob.data_ptr()), self.n_triangles, H, W, kin,
                                                      ctypes.c_void_p(Rc.data_ptr()), ctypes.c_void_p(Tc.data_ptr()), ctypes.byref(m_n),
                                                      ctypes.byref(m_a), ctypes.c_void_p(sd.data_ptr()), ctypes.c_void_p(vis.data_ptr()), st))
        return vis
