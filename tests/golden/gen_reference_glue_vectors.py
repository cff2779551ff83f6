"""Generates tests/golden/reference_glue.npz by IMPORTING the reference's utils/point_utils.py and utils/refl_utils.py in the build
container and running their own functions (depths_to_points, depth_to_normal, sample_camera_rays, sample_camera_rays_unnormalize,
reflection; cube_to_dir and cubemap_mip.forward of scene/light_utils.py, loaded by file path) on seeded inputs, plus probe texels / statistics of the split-sum table the reference ships as a data file
(assets/bsdf_256_256.bin, loaded by refl_utils at import).  Modules the image lacks and these functions never call (cv2, kornia,
nvdiffrast, ...) are registered as empty placeholders so the import statements succeed; `.cuda()` is the identity.  Only inputs and
outputs are committed; the reference source never travels.

    python tests/golden/gen_reference_glue_vectors.py       # needs /root/reference (absent on the GPU box)
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)
for name in ("kornia", "kornia.filters", "cv2", "lpips", "nvdiffrast", "nvdiffrast.torch", "plyfile", "open3d", "mediapy", "torchvision", "ipdb"):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules["nvdiffrast"].torch = sys.modules["nvdiffrast.torch"]
torch.Tensor.cuda = lambda self, *a, **k: self
_arange = torch.arange
torch.arange = lambda *a, **k: _arange(*a, **{kk: vv for kk, vv in k.items() if kk != "device"})   # point_utils hard-codes device='cuda'
os.chdir(REF)                                                        # refl_utils opens ./assets/bsdf_256_256.bin at import
from utils import point_utils, refl_utils  # noqa: E402
os.chdir(ROOT)
from materialrefgs_amd.synthetic import orbit_camera  # noqa: E402

out = {}
g = torch.Generator().manual_seed(2024)
for tag, (view, H, W) in {"a": (1, 40, 56), "b": (5, 33, 47)}.items():
    cam = orbit_camera(view, H, W)
    out[f"{tag}_view"] = np.array([view, H, W])
    depth = 2.5 + 2.0 * torch.rand(1, H, W, generator=g)
    out[f"{tag}_depth"] = depth.numpy()
    out[f"{tag}_points"] = point_utils.depths_to_points(cam, depth).numpy()                 # utils/point_utils.py:9-24
    out[f"{tag}_normal"] = point_utils.depth_to_normal(cam, depth).numpy()                  # :26-37
    refl_utils.pixel_camera = None                                                          # module-level cache keyed on H only
    rays_d, rays_o = refl_utils.sample_camera_rays(cam.HWK, cam.R, cam.T)                   # utils/refl_utils.py:54-73
    out[f"{tag}_rays_d"], out[f"{tag}_rays_o"] = rays_d.numpy(), rays_o.numpy()
    refl_utils.pixel_camera = None
    rays_u, _ = refl_utils.sample_camera_rays_unnormalize(cam.HWK, cam.R, cam.T)            # :75-93
    out[f"{tag}_rays_unnormalized"] = rays_u.numpy()
    n = torch.nn.functional.normalize(torch.randn(H, W, 3, generator=g), dim=-1)
    out[f"{tag}_n"] = n.numpy()
    wk, ndv = refl_utils.reflection(-rays_d, n)                                              # :95-98
    out[f"{tag}_refl"], out[f"{tag}_ndotv"] = wk.numpy(), ndv.numpy()

# cube face convention and the box mip of the environment map (scene/light_utils.py:24-31, 66-69)
import importlib.util  # noqa: E402
spec = importlib.util.spec_from_file_location("ref_light_utils", os.path.join(REF, "scene", "light_utils.py"))   # the `scene` package pulls
light_utils = importlib.util.module_from_spec(spec)                                                             # dataset readers in
spec.loader.exec_module(light_utils)
gy, gx = torch.meshgrid(torch.linspace(-0.9, 0.9, 5), torch.linspace(-0.8, 0.8, 4), indexing="ij")
out["cube_xy"] = torch.stack((gx, gy)).numpy()
out["cube_dirs"] = torch.stack([light_utils.cube_to_dir(s_, gx, gy) for s_ in range(6)]).numpy()               # [6,5,4,3]
cube = torch.randn(6, 8, 8, 3, generator=g)
out["mip_in"] = cube.numpy()
out["mip_out"] = light_utils.cubemap_mip.forward(None, cube).numpy()

lut = refl_utils.FG_LUT[0].numpy()                                                           # [256,256,2], :9
probes = [(0, 0), (0, 255), (255, 0), (255, 255), (128, 128), (10, 200), (200, 10), (64, 32), (32, 64), (3, 3), (250, 5), (5, 250),
          (100, 100), (180, 220), (220, 180), (127, 0)]
out["lut_probe_idx"] = np.array(probes)
out["lut_probe_val"] = np.stack([lut[y, x] for y, x in probes])
out["lut_row_means"] = lut.mean(axis=1)                                                       # [256,2]
out["lut_col_means"] = lut.mean(axis=0)
out["lut_minmax"] = np.array([lut.min(), lut.max()])
out["lut_coarse"] = lut[::8, ::8].copy()                                                      # 32 x 32 sub-sample of the table

dst = os.path.join(HERE, "reference_glue.npz")
np.savez_compressed(dst, **out)
print("wrote", dst, len(out), "arrays", os.path.getsize(dst), "bytes")
