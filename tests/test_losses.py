"""Per-view training loss: oracle vs the reference's own loss_utils (golden vectors), HIP vs both."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import loss_oracle  # noqa: E402

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "reference_loss.npz"))
CASES = ("a", "b", "c")


def _inputs(tag):
    return {k: GOLD[f"{tag}_{k}"] for k in ("img", "gt", "rn", "sn", "dist", "weight")}


def _rel(a, b):
    return float(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)).max() / (np.abs(b).max() + 1e-30))


# ---------------------------------------------------------------- CPU: the oracle is pinned to the reference
def test_oracle_window_is_the_reference_window_bit_for_bit():
    assert np.array_equal(loss_oracle.gaussian_window().astype(np.float32), GOLD["window_2d"])
    assert abs(float(GOLD["window_1d"].astype(np.float64).sum()) - 1.0) < 1e-7


@pytest.mark.parametrize("tag", CASES)
def test_oracle_matches_reference_l1_and_ssim(tag):
    d = _inputs(tag)
    t, g = loss_oracle.calculate_loss(d["img"], d["gt"], lambda_dssim=0.0)
    assert abs(t["Ll1"] - GOLD[f"{tag}_f64_l1"]) < 1e-13
    assert _rel(g["image"], GOLD[f"{tag}_f64_l1_grad"]) < 1e-12
    S, gS = loss_oracle.ssim_map_and_grad(d["img"], d["gt"])
    assert abs(S.mean() - GOLD[f"{tag}_f64_ssim"]) < 1e-12
    assert _rel(gS / S.size, GOLD[f"{tag}_f64_ssim_grad"]) < 1e-10


@pytest.mark.parametrize("tag", CASES)
@pytest.mark.parametrize("mode", ("w", "cos"))
def test_oracle_matches_reference_calculate_loss(tag, mode):
    d = _inputs(tag)
    t, g = loss_oracle.calculate_loss(d["img"], d["gt"], d["rn"], d["sn"], d["dist"], d["weight"] if mode == "w" else None,
                                      lambda_dssim=0.2, lambda_normal=0.05, lambda_dist=100.0)
    key = f"{tag}_f64_{mode}"
    assert abs(t["loss"] - GOLD[key + "_loss"]) < 1e-12
    ref_terms = GOLD[key + "_terms"]                      # l1, ssim, loss0, normal, dist, psnr
    got = [t["Ll1"], t["ssim"], t["loss0"], t["normal"], t["dist"], t["psnr"]]
    np.testing.assert_allclose(got, ref_terms, rtol=1e-10, atol=1e-12)
    assert _rel(g["image"], GOLD[key + "_g_img"]) < 1e-10
    assert _rel(g["rend_normal"], GOLD[key + "_g_rn"]) < 1e-12
    assert _rel(g["surf_normal"], GOLD[key + "_g_sn"]) < 1e-12
    assert _rel(g["rend_dist"], GOLD[key + "_g_dist"]) < 1e-12


def test_reference_fp32_is_close_to_its_fp64():
    """Sizes the tolerance of the GPU tests: the reference's own fp32 run differs from its fp64 run by this much."""
    worst = 0.0
    for tag in CASES:
        for mode in ("w", "cos"):
            worst = max(worst, _rel(GOLD[f"{tag}_f32_{mode}_g_img"], GOLD[f"{tag}_f64_{mode}_g_img"]))
    assert worst < 5e-4, worst


# ---------------------------------------------------------------- GPU: HIP path vs reference vectors and oracle
def _dev(d):
    return {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in d.items()}


@pytest.mark.gpu
@pytest.mark.parametrize("tag", CASES)
@pytest.mark.parametrize("mode", ("w", "cos"))
def test_hip_calculate_loss_matches_reference(tag, mode):
    from materialrefgs_amd import losses
    d = _dev(_inputs(tag))
    img = d["img"].clone().requires_grad_(True)
    rn, sn, dist = (d[k].clone().requires_grad_(True) for k in ("rn", "sn", "dist"))

    class O:
        pass
    cam, pc, opt = O(), O(), O()
    cam.original_image = d["gt"]
    pc.get_xyz = torch.zeros(5, 3)
    opt.lambda_dssim, opt.lambda_normal_render_depth, opt.normal_loss_start = 0.2, 0.05, 0
    opt.lambda_dist, opt.dist_loss_start = 100.0, 3000
    opt.lambda_normal_smooth = opt.lambda_depth_smooth = 0.0
    opt.normal_smooth_from_iter, opt.normal_smooth_until_iter = 0, 18000
    opt.use_perceptual_loss = False
    pkg = {"render": img, "rend_normal": rn, "surf_normal": sn, "rend_dist": dist}
    loss, tb = losses.calculate_loss(cam, pc, pkg, opt, 5000, d["weight"] if mode == "w" else None, None)
    (loss * 1.0).backward()
    key = f"{tag}_f64_{mode}"
    assert abs(float(loss) - float(GOLD[key + "_loss"])) < 2e-6 * max(1.0, abs(float(GOLD[key + "_loss"])))
    got = [float(tb[k]) for k in ("loss_l1", "ssim", "loss0", "loss_normal_render_depth", "loss_dist", "psnr")]
    np.testing.assert_allclose(got, GOLD[key + "_terms"], rtol=5e-6, atol=1e-6)
    # fp32 tolerance: the reference's own fp32 run is within 5e-4 of its fp64 run on these inputs (test above)
    assert _rel(img.grad.cpu().numpy(), GOLD[key + "_g_img"]) < 2e-4
    assert _rel(rn.grad.cpu().numpy(), GOLD[key + "_g_rn"]) < 1e-6
    assert _rel(sn.grad.cpu().numpy(), GOLD[key + "_g_sn"]) < 1e-6
    assert _rel(dist.grad.cpu().numpy(), GOLD[key + "_g_dist"]) < 1e-6
    # iteration gates: before dist_loss_start the distortion term is off and its map gets no gradient
    img2 = d["img"].clone().requires_grad_(True)
    dist2 = d["dist"].clone().requires_grad_(True)
    pkg2 = {"render": img2, "rend_normal": d["rn"], "surf_normal": d["sn"], "rend_dist": dist2}
    loss2, tb2 = losses.calculate_loss(cam, pc, pkg2, opt, 100, None, None)
    loss2.backward()
    assert dist2.grad is None and float(tb2["loss_dist"]) == 0.0


@pytest.mark.gpu
def test_perceptual_term_is_gated_by_iteration_and_pluggable():
    """utils/loss_utils.py:212-215 with the reference's default options (use_perceptual_loss = True, start 18 000): before the start
    iteration the loss is what it is without the term (a warning, once); after it the registered network's distance is added with
    lambda_perceptual_loss and reported as tb_dict["perceptual_loss"]; with nothing registered (and no `lpips` package) it raises THEN."""
    from materialrefgs_amd import losses
    d = _dev(_inputs(CASES[0]))

    class O:
        pass
    cam, pc, opt = O(), O(), O()
    cam.original_image = d["gt"]
    pc.get_xyz = torch.zeros(5, 3)
    opt.lambda_dssim, opt.lambda_normal_render_depth, opt.normal_loss_start = 0.2, 0.0, 0
    opt.lambda_dist, opt.dist_loss_start = 0.0, 3000
    opt.lambda_normal_smooth = opt.lambda_depth_smooth = 0.0
    opt.normal_smooth_from_iter, opt.normal_smooth_until_iter = 0, 18000
    opt.use_perceptual_loss, opt.lambda_perceptual_loss, opt.perceptual_loss_start_iter = True, 0.1, 18000      # arguments/__init__.py:223-225
    pkg = {"render": d["img"], "rend_normal": d["rn"], "surf_normal": d["sn"], "rend_dist": d["dist"]}
    losses._LPIPS_WARNED = False
    with pytest.warns(UserWarning, match="perceptual"):
        early, tb = losses.calculate_loss(cam, pc, pkg, opt, 17999, None, None)
    assert "perceptual_loss" not in tb
    opt.use_perceptual_loss = False
    plain, _ = losses.calculate_loss(cam, pc, pkg, opt, 17999, None, None)
    assert float(early) == float(plain)
    opt.use_perceptual_loss = True
    have_lpips = True
    try:
        import lpips  # noqa: F401
    except ImportError:
        have_lpips = False
    if not have_lpips:
        with pytest.raises(NotImplementedError, match="set_lpips_fn"):
            losses.calculate_loss(cam, pc, pkg, opt, 18001, None, None)
    losses.set_lpips_fn(lambda x, y: (x - y).pow(2).mean((1, 2, 3)))
    try:
        img = d["img"].clone().requires_grad_(True)
        late, tb = losses.calculate_loss(cam, pc, {**pkg, "render": img}, opt, 18001, None, None)
        want = 4.0 * float((d["img"] - d["gt"]).pow(2).mean())
        assert abs(float(tb["perceptual_loss"]) - want) < 1e-6 * max(1.0, want)
        assert abs(float(late) - (float(plain) + 0.1 * want)) < 1e-6
        late.backward()
        assert torch.isfinite(img.grad).all() and float(img.grad.abs().max()) > 0
    finally:
        losses.set_lpips_fn(None)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", CASES)
def test_hip_l1_and_ssim_match_reference(tag):
    from materialrefgs_amd import losses
    d = _dev(_inputs(tag))
    x = d["img"].clone().requires_grad_(True)
    l1 = losses.l1_loss(x, d["gt"])
    l1.backward()
    assert abs(float(l1) - float(GOLD[f"{tag}_f64_l1"])) < 1e-6
    assert _rel(x.grad.cpu().numpy(), GOLD[f"{tag}_f64_l1_grad"]) < 1e-6
    x = d["img"].clone().requires_grad_(True)
    s = losses.ssim(x, d["gt"])
    (3.0 * s).backward()                                   # upstream scale passes through
    assert abs(float(s) - float(GOLD[f"{tag}_f64_ssim"])) < 2e-6
    assert _rel(x.grad.cpu().numpy() / 3.0, GOLD[f"{tag}_f64_ssim_grad"]) < 2e-4
    w = losses.image_weight(d["gt"])
    assert _rel(w.cpu().numpy(), GOLD[f"{tag}_weight"]) < 1e-6


@pytest.mark.gpu
def test_hip_loss_full_size_properties():
    """800 x 800 (C2/C3 image size): identities that need no oracle, and run-to-run identical sums."""
    from materialrefgs_amd import losses
    g = torch.Generator(device="cuda").manual_seed(3)
    gt = torch.rand(3, 800, 800, device="cuda", generator=g)
    loss, terms = losses.fused_loss(gt, gt, lambda_dssim=0.2)
    assert float(terms[1]) == 0.0 and abs(float(terms[2]) - 1.0) < 1e-6 and abs(float(loss)) < 1e-6
    img = (gt + 0.05 * torch.randn(3, 800, 800, device="cuda", generator=g)).clamp(0, 1).requires_grad_(True)
    l_a, t_a = losses.fused_loss(img, gt, lambda_dssim=0.2)
    l_b, t_b = losses.fused_loss(img, gt, lambda_dssim=0.2)
    assert torch.equal(t_a, t_b)                                             # fixed-order reduction
    # loss0 is affine in lambda_dssim: loss(0.2) = 0.8 * l1 + 0.2 * (1 - ssim)
    assert abs(float(l_a) - (0.8 * float(t_a[1]) + 0.2 * (1 - float(t_a[2])))) < 1e-6
    ga, = torch.autograd.grad(l_a, img)
    g1, = torch.autograd.grad(losses.l1_loss(img, gt), img)
    gs, = torch.autograd.grad(losses.ssim(img, gt), img)
    assert _rel(ga.cpu().numpy(), (0.8 * g1 - 0.2 * gs).cpu().numpy()) < 1e-5
    # directional derivative of the SSIM against a finite difference in fp64 on the oracle is too slow at this size; use the kernel
    # itself: d/de ssim(img + e v) ~ <grad, v>
    v = gs.sign()                                                            # along the gradient: <grad, v> = sum |grad|, well above fp32 noise
    e = 1e-3
    sp, sm = losses.ssim((img + e * v).detach(), gt), losses.ssim((img - e * v).detach(), gt)
    fd = (float(sp) - float(sm)) / (2 * e)
    an = float((gs.double() * v.double()).sum())
    assert abs(fd - an) < 2e-2 * abs(an) + 1e-6, (fd, an)


@pytest.mark.gpu
def test_loss_needs_the_device():
    from materialrefgs_amd import losses
    with pytest.raises(RuntimeError):
        losses.l1_loss(torch.rand(3, 8, 8), torch.rand(3, 8, 8))


@pytest.mark.parametrize("tag", CASES)
def test_image_weight_matches_reference_get_img_grad_weight(tag):
    """losses.image_weight (plain torch ops, evaluated once per camera) = (1 - get_img_grad_weight(gt)).clamp(0, 1) ** 2 of
    utils/loss_utils.py:127-140 / train_refnerf.py:1178-1179."""
    from materialrefgs_amd import losses
    w = losses.image_weight(torch.from_numpy(GOLD[f"{tag}_gt"]))
    np.testing.assert_allclose(w.numpy(), GOLD[f"{tag}_weight"], rtol=0, atol=1e-6)


def test_edge_aware_terms_and_the_set_up_check_of_the_perceptual_loss():
    """first_order_edge_aware_loss / smooth_loss (utils/loss_utils.py:121-125) on kornia's documented Sobel: zero for constants, the
    known value of a ramp, damped by image edges; check_loss_config (the optional set-up check) refuses use_perceptual_loss while no
    LPIPS network is registered."""
    from types import SimpleNamespace
    from materialrefgs_amd import losses
    H, W = 12, 16
    ramp = torch.arange(W, dtype=torch.float64).repeat(H, 1)[None]             # d/dx = 1 everywhere inside (Sobel / 8 normalised), 0 in y
    g = losses.spatial_gradient(ramp[None])[0, 0]
    assert torch.allclose(g[0][:, 1:-1], torch.ones(H, W - 2, dtype=torch.float64)) and torch.allclose(g[1], torch.zeros(H, W, dtype=torch.float64))
    assert torch.allclose(g[0][:, 0], torch.full((H,), 0.5, dtype=torch.float64))   # replicate padding halves the border difference
    const_img = torch.zeros(3, H, W, dtype=torch.float64)
    assert abs(float(losses.smooth_loss(ramp)) - float(g[0].mean())) < 1e-12
    assert abs(float(losses.first_order_edge_aware_loss(ramp, const_img)) - float(g[0].mean())) < 1e-12
    edgy = torch.zeros(3, H, W, dtype=torch.float64); edgy[:, :, W // 2:] = 8.0
    assert float(losses.first_order_edge_aware_loss(ramp, edgy)) < float(losses.first_order_edge_aware_loss(ramp, const_img))
    with pytest.raises(NotImplementedError, match="no-use_perceptual_loss"):
        losses.check_loss_config(SimpleNamespace(use_perceptual_loss=True, perceptual_loss_start_iter=18000))
    losses.set_lpips_fn(lambda x, y: (x - y).abs().mean((1, 2, 3)))
    try:
        losses.check_loss_config(SimpleNamespace(use_perceptual_loss=True, perceptual_loss_start_iter=18000))      # a network is registered
        x, y = torch.rand(1, 3, 8, 8, dtype=torch.float64), torch.rand(1, 3, 8, 8, dtype=torch.float64)
        assert abs(float(losses.lpips_loss(x, y)) - 2.0 * float((x - y).abs().mean())) < 1e-12                    # images go in scaled to [-1, 1] (:43)
    finally:
        losses.set_lpips_fn(None)
