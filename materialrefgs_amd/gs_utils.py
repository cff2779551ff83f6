"""Small per-gaussian / per-pixel torch helpers used by the render functions (host side of the hot path).

Behavioural mirrors (checked against golden vectors generated from the reference, tests/golden/):
  sh_basis / eval_sh      <-> utils/sh_utils.py:57-112 (real SH, degrees 0..3, 3DGS sign convention)
  RGB2SH / SH2RGB         <-> utils/sh_utils.py:114-118
  build_rotation          <-> utils/general_utils.py:78-100 (quaternion w,x,y,z; normalised inside)
  build_scaling_rotation  <-> utils/general_utils.py:102-112
  safe_normalize          <-> utils/general_utils.py:178-181
  flip_align_view         <-> utils/general_utils.py:184-190
  linear_to_srgb / srgb_to_linear <-> utils/graphics_utils.py:102-120
  geom_transform_points   <-> utils/graphics_utils.py:22-29
Everything is device-agnostic (runs on whatever device the inputs live on).
"""
import torch

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
         1.445305721320277, -0.5900435899266435)


def sh_basis(deg: int, dirs: torch.Tensor) -> torch.Tensor:
    """Real SH basis values [..., (deg+1)^2] at unit directions [..., 3] (constants and signs folded in)."""
    assert 0 <= deg <= 3
    x, y, z = dirs[..., 0], dirs[..., 1], dirs[..., 2]
    cols = [torch.full_like(x, SH_C0)]
    if deg > 0:
        cols += [-SH_C1 * y, SH_C1 * z, -SH_C1 * x]
    if deg > 1:
        xx, yy, zz = x * x, y * y, z * z
        cols += [SH_C2[0] * (x * y), SH_C2[1] * (y * z), SH_C2[2] * (2.0 * zz - xx - yy), SH_C2[3] * (x * z), SH_C2[4] * (xx - yy)]
    if deg > 2:
        cols += [SH_C3[0] * y * (3 * xx - yy), SH_C3[1] * (x * y) * z, SH_C3[2] * y * (4 * zz - xx - yy),
                 SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy), SH_C3[4] * x * (4 * zz - xx - yy), SH_C3[5] * z * (xx - yy),
                 SH_C3[6] * x * (xx - 3 * yy)]
    return torch.stack(cols, dim=-1)


def eval_sh(deg: int, sh: torch.Tensor, dirs: torch.Tensor) -> torch.Tensor:
    """sh: [..., C, >=(deg+1)^2] coefficients, dirs: [..., 3] unit vectors -> [..., C]."""
    n = (deg + 1) ** 2
    assert sh.shape[-1] >= n
    basis = sh_basis(deg, dirs)
    return (sh[..., :n] * basis.unsqueeze(-2)).sum(dim=-1)


def RGB2SH(rgb):
    return (rgb - 0.5) / SH_C0


def SH2RGB(sh):
    return sh * SH_C0 + 0.5


def build_rotation(r: torch.Tensor) -> torch.Tensor:
    """[N,4] quaternions (w,x,y,z), not necessarily unit -> [N,3,3] rotation matrices."""
    q = r / torch.sqrt((r * r).sum(dim=1, keepdim=True))
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    rows = [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
            2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
            2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]
    return torch.stack(rows, dim=1).reshape(-1, 3, 3)


def build_scaling_rotation(s: torch.Tensor, r: torch.Tensor) -> torch.Tensor:
    """R(q) @ diag(s) for s [N,3]."""
    return build_rotation(r) * s.unsqueeze(1)


def safe_normalize(x: torch.Tensor, eps: float = 1e-20) -> torch.Tensor:
    return x / torch.clamp(torch.linalg.norm(x, dim=-1, keepdim=True), min=eps)


def flip_align_view(normal: torch.Tensor, viewdir: torch.Tensor):
    """Flip normals that face away from the viewer (n . -viewdir < 0).  Returns (flipped, non_flip mask)."""
    non_flip = (normal * -viewdir).sum(dim=-1, keepdim=True) >= 0
    return normal * torch.where(non_flip, 1, -1), non_flip


def linear_to_srgb(linear: torch.Tensor, eps=None) -> torch.Tensor:
    if eps is None:
        eps = torch.finfo(linear.dtype).eps
    lo = 323 / 25 * linear
    hi = (211 * linear.clamp_min(eps) ** (5 / 12) - 11) / 200
    return torch.where(linear <= 0.0031308, lo, hi)


def srgb_to_linear(srgb: torch.Tensor, eps=None) -> torch.Tensor:
    if eps is None:
        eps = torch.finfo(srgb.dtype).eps
    lo = 25 / 323 * srgb
    hi = ((200 * srgb + 11) / 211).clamp_min(eps) ** (12 / 5)
    return torch.where(srgb <= 0.04045, lo, hi)


def geom_transform_points(points: torch.Tensor, transf_matrix: torch.Tensor) -> torch.Tensor:
    """Row-vector homogeneous transform with perspective divide (w + 1e-7)."""
    hom = torch.cat([points, torch.ones_like(points[:, :1])], dim=1) @ transf_matrix
    return hom[:, :3] / (hom[:, 3:] + 0.0000001)


# ---- anisotropic spherical gaussians of the indirect term (pipe.use_asg, off by default) ------------------------------------------------
def predefined_asg_axes(n_theta=4, n_phi=8, device=None):
    """The fixed lobe frames of GaussianModel.asg_param (utils/graphics_utils.py:196-229, init_predefined_omega(4, 8)): per lobe the axis
    omega on the upper hemisphere (theta at the centres of n_theta bands, phi at the centres of n_phi sectors), the tangent
    omega_lambda = the same phi at theta + pi/2, and omega_mu = omega_lambda turned a quarter about omega (q p q^-1 with
    q = (cos pi/4, sin pi/4 omega)), which for perpendicular unit vectors is omega x omega_lambda.  Three [n_theta n_phi, 3] tensors."""
    import math
    i = torch.arange(n_theta, dtype=torch.float32)
    j = torch.arange(n_phi, dtype=torch.float32)
    theta = (i * (0.5 * math.pi / n_theta) + 0.5 * math.pi / (2 * n_theta)).repeat_interleave(n_phi)
    phi = (j * (2 * math.pi / n_phi) + 2 * math.pi / (2 * n_phi)).repeat(n_theta)
    ball = lambda t, p: torch.stack([torch.cos(p) * torch.sin(t), torch.sin(p) * torch.sin(t), torch.cos(t)], dim=-1)
    omega, lam = ball(theta, phi), ball(theta + 0.5 * math.pi, phi)
    mu = torch.linalg.cross(omega, lam, dim=-1)
    return tuple(t.to(device) if device is not None else t for t in (omega, lam, mu))


def rotate_z_frame_inverse(n, v):
    """rotation_between_z(n)^T v (utils/graphics_utils.py:121-153): v expressed in the frame whose z axis is the unit vector n, the frame
    being the shortest rotation that takes z to n (Rodrigues with axis z x n; n = -z: minus the identity)."""
    nx, ny, nz = n[..., 0], n[..., 1], n[..., 2]
    c = (nz + 1).clamp_min(1e-7)
    vx, vy, vz = v[..., 0], v[..., 1], v[..., 2]
    # rows of R^T = columns of R = I + [a]x + [a]x^2 / (1 + n.z), a = (-n.y, n.x, 0)
    ox = (1 - nx * nx / c) * vx + (-nx * ny / c) * vy + (-nx) * vz
    oy = (-nx * ny / c) * vx + (1 - ny * ny / c) * vy + (-ny) * vz
    oz = nx * vx + ny * vy + (1 - (nx * nx + ny * ny) / c) * vz
    out = torch.stack([ox, oy, oz], dim=-1)
    return torch.where((nz + 1 > 0)[..., None], out, -v)


def asg_indirect(asg, axes, normals, reflection):
    """Indirect radiance [P,3] from the per-gaussian lobes `asg` [P,32,5] = (amplitude 3, lambda, mu) along `reflection`, expressed in the
    frame of `normals` (gaussian_renderer/__init__.py:312-336): sum over lobes of exp(a - 3) * relu(r . omega) *
    exp(-softplus(la - 1) (r . omega_lambda)^2 - softplus(mu - 1) (r . omega_mu)^2), clamped at 0."""
    omega, omega_la, omega_mu = axes
    r = rotate_z_frame_inverse(normals, reflection)[:, None, :]                       # [P,1,3]
    ep, la, mu = torch.split(asg, [3, 1, 1], dim=-1)
    smooth = torch.relu((r * omega[None]).sum(dim=-1, keepdim=True))
    la, mu = torch.nn.functional.softplus(la - 1), torch.nn.functional.softplus(mu - 1)
    arg = -la * (omega_la[None] * r).sum(dim=-1, keepdim=True).pow(2) - mu * (omega_mu[None] * r).sum(dim=-1, keepdim=True).pow(2)
    return (torch.exp(ep - 3) * smooth * torch.exp(arg)).sum(dim=1).clamp_min(0.0)
