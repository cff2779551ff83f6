"""Generates tests/golden/reference_helpers.npz by IMPORTING the reference's pure-PyTorch helper modules
(no stubs needed: utils/sh_utils.py, utils/general_utils.py, utils/graphics_utils.py) in the build container.
The reference source never travels; only these input/output vectors are committed.

    python tests/golden/gen_reference_vectors.py       # needs /root/reference (absent on the GPU box)
"""
import math
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
sys.path.insert(0, REF)
from utils import general_utils, graphics_utils, sh_utils  # noqa: E402

torch.manual_seed(1234)
rng = np.random.default_rng(1234)
out = {}

# eval_sh (utils/sh_utils.py:57-112) for degrees 0..3; sh layout [..., C, (deg+1)^2]
N = 64
sh = torch.randn(N, 3, 16) * 0.3
dirs = torch.nn.functional.normalize(torch.randn(N, 3), dim=-1)
out["sh_coeffs"] = sh.numpy()
out["sh_dirs"] = dirs.numpy()
for deg in range(4):
    out[f"eval_sh_deg{deg}"] = sh_utils.eval_sh(deg, sh[..., : (deg + 1) ** 2], dirs).numpy()
rgb = torch.rand(8, 3)
out["rgb"] = rgb.numpy()
out["rgb2sh"] = sh_utils.RGB2SH(rgb).numpy()
out["sh2rgb"] = sh_utils.SH2RGB(sh_utils.RGB2SH(rgb)).numpy()

# build_rotation (utils/general_utils.py:78-100): quaternion (w,x,y,z), normalised inside
q = torch.randn(N, 4)
out["quat"] = q.numpy()
_cuda = torch.Tensor.cuda
_zeros = torch.zeros
torch.zeros = lambda *a, **k: _zeros(*a, **{kk: vv for kk, vv in k.items() if kk != "device"})  # helpers hard-code device='cuda'
try:
    out["build_rotation"] = general_utils.build_rotation(q).numpy()
    s = torch.rand(N, 3) + 0.1
    out["scal3"] = s.numpy()
    out["build_scaling_rotation"] = general_utils.build_scaling_rotation(s, q).numpy()
finally:
    torch.zeros = _zeros

# camera matrices (utils/graphics_utils.py:37-71)
Rs, Ts, W2V = [], [], []
for _ in range(6):
    A = rng.normal(size=(3, 3))
    Q, _r = np.linalg.qr(A)
    if np.linalg.det(Q) < 0:
        Q[:, 0] = -Q[:, 0]
    t = rng.normal(size=3) * 2
    Rs.append(Q); Ts.append(t)
    W2V.append(graphics_utils.getWorld2View2(Q, t))
out["cam_R"] = np.stack(Rs); out["cam_T"] = np.stack(Ts); out["world2view2"] = np.stack(W2V)
fovs = np.array([[0.6911, 0.6911], [1.2, 0.9], [0.3, 0.5]])
out["fovs"] = fovs
out["projection"] = np.stack([graphics_utils.getProjectionMatrix(0.01, 100.0, fx, fy).numpy() for fx, fy in fovs])
out["fov2focal"] = np.array([graphics_utils.fov2focal(f, 800) for f in fovs[:, 0]])
pts = torch.randn(32, 3)
M = torch.tensor(W2V[0]).T @ graphics_utils.getProjectionMatrix(0.01, 100.0, 0.6911, 0.6911).T
out["gtp_points"] = pts.numpy(); out["gtp_matrix"] = M.numpy()
out["geom_transform_points"] = graphics_utils.geom_transform_points(pts, M).numpy()
lin = torch.rand(256)
out["linear"] = lin.numpy()
out["linear_to_srgb"] = graphics_utils.linear_to_srgb(lin).numpy()
out["srgb_to_linear"] = graphics_utils.srgb_to_linear(lin).numpy()

dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_helpers.npz")
np.savez_compressed(dst, **out)
print("wrote", dst, {k: v.shape for k, v in out.items()})
