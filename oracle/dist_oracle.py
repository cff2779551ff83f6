"""Torch restatement of the factored SH-gradient expansion of materialrefgs_amd/dist.py -- TEST INFRASTRUCTURE ONLY.

The product expands the gathered factors in libmrgs.so (mrgs_sh_grad_expand, mrgs_sh_grad_expand_surfel) and rejects CPU
tensors.  The world-size-2 gloo tests of the collective plumbing (tests/test_dist_cpu.py) run without a GPU, so they hand these
functions to the reducers through `expand_fn`; the GPU tests compare the kernels with them.

The quantity restated is the reference's own SH colour gradient, dL/dsh[p][k][c] = B_k(dir(p)) * dL/dRGB[p][c]
(submodules/diff-surfel-rasterization/cuda_rasterizer/backward.cu:22-141), summed over views; the basis B_k is
materialrefgs_amd.gs_utils.sh_basis, which tests/test_golden_helpers.py pins to the reference's utils/sh_utils.py:eval_sh.
"""
import torch

from materialrefgs_amd.gs_utils import sh_basis


def expand_sh_gradients(gathered, means3D, M, sh_degree):
    """gathered [V, 3P + 3] rows [dRGB_v | campos_v] -> sum_v B_k(normalize(means3D - campos_v)) dRGB_v, [P, M, 3]."""
    V, P = gathered.shape[0], means3D.shape[0]
    assert gathered.shape[1] == 3 * P + 3
    out = torch.zeros((P, M, 3), dtype=torch.float32)
    n = (sh_degree + 1) ** 2
    for v in range(V):
        drgb, cam = gathered[v, :3 * P].view(P, 3), gathered[v, 3 * P:]
        d = means3D.detach() - cam
        basis = sh_basis(sh_degree, d / d.norm(dim=1, keepdim=True))          # [P, n]
        out[:, :n] += basis.unsqueeze(-1) * drgb.unsqueeze(1)
    return out


def view_and_mirror_dirs(xyz, rotation_raw, cam):
    """Unit view direction and mirror direction of the facing normal (gaussian_renderer/__init__.py:338-347 with
    scene/gaussian_model.py:269-285) with torch ops."""
    q = rotation_raw / rotation_raw.norm(dim=1, keepdim=True)
    w, x, y, z = q.unbind(1)
    nr = torch.stack([2 * (x * z + w * y), 2 * (y * z - w * x), 1 - 2 * (x * x + y * y)], dim=1)
    d = xyz - cam
    v = d / d.norm(dim=1, keepdim=True)
    flip = torch.where(-(nr * v).sum(1, keepdim=True) >= 0, 1.0, -1.0)
    nn = nr * flip
    nn = nn / nn.norm(dim=1, keepdim=True).clamp_min(1e-20)
    c = -(nn * v).sum(1, keepdim=True)
    return v, 2 * c * nn + v


def expand_surfel_sh_gradients(gathered, xyz, rotation_raw, sh_degree):
    """gathered [V, 6P + 3] rows [dRGB_v | dIND_v | campos_v] -> summed gradients of (features_dc [P,1,3], features_rest [P,15,3],
    indirect_dc [P,1,3], indirect_rest [P,15,3])."""
    V, P = gathered.shape[0], xyz.shape[0]
    assert gathered.shape[1] == 6 * P + 3
    sh, ind = torch.zeros((P, 16, 3)), torch.zeros((P, 16, 3))
    n = (sh_degree + 1) ** 2
    for v in range(V):
        g, h, cam = gathered[v, :3 * P].view(P, 3), gathered[v, 3 * P:6 * P].view(P, 3), gathered[v, 6 * P:]
        vd, rd = view_and_mirror_dirs(xyz.detach(), rotation_raw.detach(), cam)
        sh[:, :n] += sh_basis(sh_degree, vd).unsqueeze(-1) * g.unsqueeze(1)
        ind += sh_basis(3, rd).unsqueeze(-1) * h.unsqueeze(1)
    return [sh[:, :1].contiguous(), sh[:, 1:].contiguous(), ind[:, :1].contiguous(), ind[:, 1:].contiguous()]
