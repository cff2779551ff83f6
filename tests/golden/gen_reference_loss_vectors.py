"""Generates tests/golden/reference_loss.npz by IMPORTING the reference's utils/loss_utils.py in the build container and
running its own l1_loss / ssim / calculate_loss (+ torch.autograd) on seeded inputs.  Modules the image lacks and this path
never calls (kornia, cv2, lpips) are registered as empty placeholders so the import statement succeeds; every function that is
executed is the reference's.  Only inputs and outputs are committed; the reference source never travels.

    python tests/golden/gen_reference_loss_vectors.py       # needs /root/reference (absent on the GPU box)
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
sys.path.insert(0, REF)
for name in ("kornia", "kornia.filters", "cv2", "lpips"):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules["kornia.filters"].spatial_gradient = None
sys.modules["kornia"].filters = sys.modules["kornia.filters"]
torch.Tensor.cuda = lambda self, *a, **k: self                       # calculate_loss calls .cuda() on the ground truth
from utils import loss_utils  # noqa: E402

torch.manual_seed(77)
out = {"window_1d": loss_utils.gaussian(11, 1.5).numpy(), "window_2d": loss_utils.create_window(11, 3).numpy()[0, 0]}


class _Obj:
    pass


def scene(H, W, dtype):
    g = torch.Generator().manual_seed(H * 1000 + W)
    yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
    base = torch.stack([0.5 + 0.4 * torch.sin(6 * xx + 3 * yy), 0.5 + 0.4 * torch.cos(5 * yy), 0.3 + 0.5 * xx * yy])
    gt = (base + 0.05 * torch.randn(3, H, W, generator=g)).clamp(0, 1)
    gt[:, : H // 4, : W // 3] = 0.0                                   # flat background block (sigma = 0 region)
    img = (gt + 0.1 * torch.randn(3, H, W, generator=g)).clamp(0, 1)
    img[:, : H // 8, : W // 6] = 0.0                                  # exact-equality region: sign(0) = 0 in the L1 gradient
    rn = torch.nn.functional.normalize(torch.randn(3, H, W, generator=g), dim=0) * torch.rand(1, H, W, generator=g)
    sn = torch.nn.functional.normalize(rn + 0.3 * torch.randn(3, H, W, generator=g), dim=0)
    dist = torch.rand(1, H, W, generator=g) * 0.01
    return [t.to(dtype) for t in (img, gt, rn, sn, dist)]


for tag, (H, W) in {"a": (37, 45), "b": (64, 48), "c": (16, 16)}.items():
    for dname, dtype in (("f32", torch.float32), ("f64", torch.float64)):
        img, gt, rn, sn, dist = scene(H, W, dtype)
        if dname == "f32":
            for k, v in (("img", img), ("gt", gt), ("rn", rn), ("sn", sn), ("dist", dist)):
                out[f"{tag}_{k}"] = v.numpy()
            wt = (1.0 - loss_utils.get_img_grad_weight(gt)).clamp(0, 1) ** 2      # train_refnerf.py:1178-1179
            out[f"{tag}_weight"] = wt.numpy()
        wt_t = torch.from_numpy(out[f"{tag}_weight"]).to(dtype)
        x = img.clone().requires_grad_(True)
        l1 = loss_utils.l1_loss(x, gt)
        out[f"{tag}_{dname}_l1"] = l1.detach().numpy()
        out[f"{tag}_{dname}_l1_grad"] = torch.autograd.grad(l1, x)[0].numpy()
        x = img.clone().requires_grad_(True)
        s = loss_utils.ssim(x, gt)
        out[f"{tag}_{dname}_ssim"] = s.detach().numpy()
        out[f"{tag}_{dname}_ssim_grad"] = torch.autograd.grad(s, x)[0].numpy()
        for mode, weight in (("w", wt_t), ("cos", None)):
            cam, pc, opt = _Obj(), _Obj(), _Obj()
            cam.original_image = gt
            pc.get_xyz = torch.zeros(5, 3)
            opt.lambda_dssim, opt.lambda_normal_render_depth, opt.normal_loss_start = 0.2, 0.05, 0
            opt.lambda_dist, opt.dist_loss_start = 100.0, 3000
            opt.lambda_normal_smooth = opt.lambda_depth_smooth = 0.0
            opt.normal_smooth_from_iter, opt.normal_smooth_until_iter = 0, 18000
            opt.use_perceptual_loss = False
            leaves = [t.clone().requires_grad_(True) for t in (img, rn, sn, dist)]
            pkg = {"render": leaves[0], "rend_alpha": None, "surf_depth": None, "rend_normal": leaves[1], "surf_normal": leaves[2],
                   "visibility_filter": None, "rend_dist": leaves[3]}
            loss, tb = loss_utils.calculate_loss(cam, pc, pkg, opt, 5000, weight, None)
            grads = torch.autograd.grad(loss, leaves)
            key = f"{tag}_{dname}_{mode}"
            out[key + "_loss"] = loss.detach().numpy()
            out[key + "_terms"] = np.array([tb["loss_l1"], tb["ssim"], tb["loss0"], float(tb["loss_normal_render_depth"]),
                                            float(tb["loss_dist"]), tb["psnr"]], dtype=np.float64)
            for n, gval in zip(("g_img", "g_rn", "g_sn", "g_dist"), grads):
                out[key + "_" + n] = gval.numpy()

dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_loss.npz")
np.savez_compressed(dst, **out)
print("wrote", dst, len(out), "arrays", os.path.getsize(dst), "bytes")
