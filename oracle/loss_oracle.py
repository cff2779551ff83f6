"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy, float64) of the reference's per-view training loss, the checker of
csrc/mrgs_loss.hip.  Never imported by the product path.

Follows utils/loss_utils.py: l1_loss :22-23, gaussian/create_window :28-30,83-87, _ssim :100-117 (F.conv2d with zero
padding 5, depthwise), calculate_loss :142-228 (terms L1, SSIM, normal consistency :166-175, distortion :180-182), psnr
(utils/image_utils.py).  Gradients are the analytic derivatives of exactly those expressions (what torch.autograd returns for
the reference).  PINNED: tests/test_losses.py checks values and gradients against tests/golden/reference_loss.npz, produced by
importing the reference's own loss_utils in the build container (tests/golden/gen_reference_loss_vectors.py).
"""
import math

import numpy as np


def gaussian_window(window_size=11, sigma=1.5):
    # :28-30 -- exp in python double, stored as fp32, normalised in fp32; then used in the image dtype
    g = np.array([math.exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)], dtype=np.float32)
    g = g / np.float32(g.astype(np.float64).sum())               # torch's fp32 sum of these 11 values == correctly rounded sum
    w2 = (g[:, None] * g[None, :]).astype(np.float32)            # _1D_window.mm(_1D_window.t()).float()
    return w2.astype(np.float64)


def _corr2(x, w2):
    """Depthwise F.conv2d(x, window, padding=r) (cross-correlation) for x [C,H,W]."""
    r = w2.shape[0] // 2
    C, H, W = x.shape
    xp = np.zeros((C, H + 2 * r, W + 2 * r), dtype=np.float64)
    xp[:, r:r + H, r:r + W] = x
    out = np.zeros((C, H, W), dtype=np.float64)
    for i in range(w2.shape[0]):
        for j in range(w2.shape[1]):
            out += w2[i, j] * xp[:, i:i + H, j:j + W]
    return out


def ssim_map_and_grad(img1, img2):
    """Returns (ssim_map [C,H,W], d(sum ssim_map)/d img1)."""
    w2 = gaussian_window()
    x, y = img1.astype(np.float64), img2.astype(np.float64)
    mu1, mu2 = _corr2(x, w2), _corr2(y, w2)
    e11, e22, e12 = _corr2(x * x, w2), _corr2(y * y, w2), _corr2(x * y, w2)
    s1, s2, s12 = e11 - mu1 * mu1, e22 - mu2 * mu2, e12 - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    A1, A2, B1, B2 = 2 * mu1 * mu2 + C1, 2 * s12 + C2, mu1 * mu1 + mu2 * mu2 + C1, s1 + s2 + C2
    S = A1 * A2 / (B1 * B2)
    # chain rule through mu1, sigma1_sq, sigma12 as separate nodes (the way autograd walks the reference graph)
    dS_dA1, dS_dA2, dS_dB1, dS_dB2 = A2 / (B1 * B2), A1 / (B1 * B2), -S / B1, -S / B2
    d_s12 = 2 * dS_dA2
    d_s1 = dS_dB2
    d_mu1 = dS_dA1 * 2 * mu2 + dS_dB1 * 2 * mu1 + d_s1 * (-2 * mu1) + d_s12 * (-mu2)
    d_e11, d_e12 = d_s1, d_s12
    w2t = w2[::-1, ::-1]                                          # adjoint of a correlation = correlation with the flipped window
    grad = _corr2(d_mu1, w2t) + 2 * x * _corr2(d_e11, w2t) + y * _corr2(d_e12, w2t)
    return S, grad


def calculate_loss(image, gt, rend_normal=None, surf_normal=None, rend_dist=None, image_weight=None, lambda_dssim=0.2,
                   lambda_normal=0.0, lambda_dist=0.0):
    """Returns (terms dict, grads dict) for dL/dloss = 1."""
    x, y = image.astype(np.float64), gt.astype(np.float64)
    C, H, W = x.shape
    N, HW = C * H * W, H * W
    S, gS = ssim_map_and_grad(x, y)
    Ll1, ssim = np.abs(x - y).mean(), S.mean()
    loss0 = (1.0 - lambda_dssim) * Ll1 + lambda_dssim * (1.0 - ssim)
    g_img = (1.0 - lambda_dssim) * np.sign(x - y) / N - lambda_dssim * gS / N
    terms = {"Ll1": Ll1, "ssim": ssim, "loss0": loss0, "normal": 0.0, "dist": 0.0}
    grads = {"image": g_img}
    loss = loss0
    if lambda_normal > 0:
        rn, sn = rend_normal.astype(np.float64), surf_normal.astype(np.float64)
        if image_weight is not None:
            wt = image_weight.astype(np.float64)
            terms["normal"] = (wt * np.abs(sn - rn).sum(0)).mean()                       # :170
            sg = np.sign(sn - rn)
            grads["surf_normal"] = lambda_normal * wt[None] * sg / HW
            grads["rend_normal"] = -grads["surf_normal"]
        else:
            terms["normal"] = (1 - (rn * sn).sum(0)).mean()                              # :172-173
            grads["rend_normal"] = -lambda_normal * sn / HW
            grads["surf_normal"] = -lambda_normal * rn / HW
        loss = loss + lambda_normal * terms["normal"]
    if lambda_dist > 0:
        terms["dist"] = lambda_dist * rend_dist.astype(np.float64).mean()                # :181
        grads["rend_dist"] = np.full(rend_dist.shape, lambda_dist / HW)
        loss = loss + terms["dist"]
    mse = ((x - y) ** 2).reshape(C, -1).mean(1)
    terms["psnr"] = float(np.mean(20 * np.log10(1.0 / np.sqrt(mse))))
    terms["mse"] = mse
    terms["loss"] = loss
    return terms, grads
