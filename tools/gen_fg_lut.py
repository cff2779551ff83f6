"""Generates materialrefgs_amd/assets/fg_lut_256.npy: the 256x256x2 split-sum "FG" table used by the specular shading
(scale, bias of F0 in the pre-integrated GGX BRDF; Karis 2013).  Row v = roughness, column u = N.V, texel centres at
(i + 0.5) / 256.  The reference ships assets/bsdf_256_256.bin (nvdiffrec); this is an independent generator -- run with
--compare /root/reference/assets/bsdf_256_256.bin in the build container to print the difference (nothing is copied).
"""
import argparse
import os

import numpy as np


def radical_inverse_vdc(n):
    bits = np.arange(n, dtype=np.uint32)
    bits = (bits << 16) | (bits >> 16)
    bits = ((bits & 0x55555555) << 1) | ((bits & 0xAAAAAAAA) >> 1)
    bits = ((bits & 0x33333333) << 2) | ((bits & 0xCCCCCCCC) >> 2)
    bits = ((bits & 0x0F0F0F0F) << 4) | ((bits & 0xF0F0F0F0) >> 4)
    bits = ((bits & 0x00FF00FF) << 8) | ((bits & 0xFF00FF00) >> 8)
    return bits.astype(np.float64) * 2.3283064365386963e-10


def integrate(res=256, n_samples=4096):
    xi1 = (np.arange(n_samples) + 0.5) / n_samples
    xi2 = radical_inverse_vdc(n_samples)
    phi = 2 * np.pi * xi1
    out = np.zeros((res, res, 2), np.float64)
    nov = (np.arange(res) + 0.5) / res
    V = np.stack([np.sqrt(1 - nov ** 2), np.zeros(res), nov], -1)            # [res,3]
    for j in range(res):
        rough = (j + 0.5) / res
        a = rough * rough
        cos_t = np.sqrt((1 - xi2) / (1 + (a * a - 1) * xi2))
        sin_t = np.sqrt(np.maximum(0, 1 - cos_t ** 2))
        Hh = np.stack([sin_t * np.cos(phi), sin_t * np.sin(phi), cos_t], -1)  # [n,3]
        VoH = V @ Hh.T                                                        # [res,n]
        L = 2 * VoH[..., None] * Hh[None] - V[:, None, :]
        NoL = np.clip(L[..., 2], 0, 1)
        NoH = np.clip(Hh[:, 2], 0, 1)[None]
        VoHc = np.clip(VoH, 0, 1)
        # height-correlated Smith GGX masking-shadowing, alpha = roughness^2 (the variant that reproduces the reference
        # asset; the separable Schlick-GGX k = alpha/2 form is off by up to 0.35 at grazing angles)
        nv = nov[:, None]
        lam_v = NoL * np.sqrt(a * a + (1 - a * a) * nv * nv)
        lam_l = nv * np.sqrt(a * a + (1 - a * a) * NoL * NoL)
        G = 2 * NoL * nv / (lam_v + lam_l + 1e-30)
        with np.errstate(divide="ignore", invalid="ignore"):
            G_vis = np.where(NoL > 0, G * VoHc / (NoH * nov[:, None]), 0.0)
        Fc = (1 - VoHc) ** 5
        out[j, :, 0] = ((1 - Fc) * G_vis).mean(1)
        out[j, :, 1] = (Fc * G_vis).mean(1)
    return out.astype(np.float32)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--compare")
    ap.add_argument("--samples", type=int, default=4096)
    a = ap.parse_args()
    lut = integrate(256, a.samples)
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "materialrefgs_amd", "assets", "fg_lut_256.npy")
    np.save(dst, lut)
    print("wrote", dst, lut.shape, "range", lut.min(), lut.max())
    if a.compare:
        ref = np.fromfile(a.compare, dtype=np.float32).reshape(256, 256, 2)
        d = np.abs(ref - lut)
        print("vs reference asset: max abs diff %.4f mean abs diff %.5f; ch0 max %.4f ch1 max %.4f" % (d.max(), d.mean(), d[..., 0].max(), d[..., 1].max()))
        print("ref corners", ref[0, 0], ref[0, -1], ref[-1, 0], ref[-1, -1], "mine", lut[0, 0], lut[0, -1], lut[-1, 0], lut[-1, -1])
