"""Training loss of one view behind the reference's names (utils/loss_utils.py): `l1_loss` (:22-23), `ssim` (:83-119),
`get_img_grad_weight` (:127-140) and `calculate_loss` (:142-228).  The pixel work runs in libmrgs.so (csrc/mrgs_loss.hip:
three launches per view for the value and one for all gradient maps, instead of five depthwise conv2d launches, ~30
elementwise kernels and their autograd mirror); there is no torch fallback -- CPU tensors raise.

Differences a caller sees: the entries of `tb_dict` are 0-d device tensors instead of python floats (the reference calls
`.item()` five times per iteration, each a host sync; `float(x)` still works), and the terms the vendored tree cannot run here
(`first_order_edge_aware_loss` needs kornia, `lpips_loss` needs network weights; both default to off before iteration 18 000,
arguments/__init__.py:142-143,223-225) raise NotImplementedError when enabled.
"""
import ctypes

import torch

from . import _lib


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _f32(t):
    return t.detach().float().contiguous()


class _FusedLoss(torch.autograd.Function):
    """(loss scalar, terms[16]) of image [C,H,W] against gt; optional normal-consistency and distortion terms."""

    @staticmethod
    def forward(ctx, image, gt, rend_normal, surf_normal, rend_dist, image_weight, lambda_dssim, lambda_normal, lambda_dist):
        ctx.set_materialize_grads(False)   # unused outputs arrive as None in backward, not as zero-filled tensors
        if not image.is_cuda:
            raise RuntimeError("materialrefgs_amd.losses needs device tensors (libmrgs.so has no CPU path)")
        C, H, W = image.shape[-3:]
        cfg = _lib.MrgsLossConfig(H, W, C, float(lambda_dssim), float(lambda_normal), float(lambda_dist))
        img, g = _f32(image).view(C, H, W), _f32(gt).view(C, H, W)
        use_n = lambda_normal > 0 and rend_normal is not None and surf_normal is not None
        use_d = lambda_dist > 0 and rend_dist is not None
        if not use_n:
            cfg.lambda_normal = 0.0
        if not use_d:
            cfg.lambda_dist = 0.0
        rn = _f32(rend_normal) if use_n else None
        sn = _f32(surf_normal) if use_n else None
        rd = _f32(rend_dist) if use_d else None
        wt = _f32(image_weight) if (use_n and image_weight is not None) else None
        lib = _lib.lib()
        with _lib.guard(image.device):
            ws = torch.empty(lib.mrgs_loss_ws_bytes(H, W, C), dtype=torch.uint8, device=image.device)
            terms = torch.empty(16, dtype=torch.float32, device=image.device)
            loss = torch.empty((), dtype=torch.float32, device=image.device)   # its own tensor: a view of `terms` as a second output
            #                                                                    ties both outputs into a cycle only the GC can free
            st = _lib.stream_ptr(image.device)
            _lib.check(lib.mrgs_loss_forward(ctypes.byref(cfg), _p(img), _p(g), _p(rn), _p(sn), _p(rd), _p(wt), _p(ws), ws.numel(),
                                             _p(terms), _p(loss), st))
        ctx.cfg = cfg
        ctx.shapes = (image.shape, None if rend_normal is None else rend_normal.shape, None if surf_normal is None else surf_normal.shape,
                      None if rend_dist is None else rend_dist.shape)
        ctx.use = (use_n, use_d)
        ctx.save_for_backward(img, g, rn, sn, wt, ws)
        ctx.mark_non_differentiable(terms)
        return loss, terms

    @staticmethod
    def backward(ctx, g_loss, _g_terms):
        img, g, rn, sn, wt, ws = ctx.saved_tensors
        use_n, use_d = ctx.use
        shp_i, shp_rn, shp_sn, shp_d = ctx.shapes
        cfg = ctx.cfg
        if g_loss is None:
            return (None,) * 9
        gl = g_loss.detach().float().contiguous()
        g_img = torch.empty_like(img)
        g_rn = torch.empty_like(rn) if use_n else None
        g_sn = torch.empty_like(sn) if use_n else None
        g_d = torch.empty((cfg.H, cfg.W), dtype=torch.float32, device=img.device) if use_d else None
        with _lib.guard(img.device):
            st = _lib.stream_ptr(img.device)
            _lib.check(_lib.lib().mrgs_loss_backward(ctypes.byref(cfg), _p(img), _p(g), _p(rn), _p(sn), _p(wt), _p(ws), _p(gl), _p(g_img),
                                                     _p(g_rn), _p(g_sn), _p(g_d), st))
        return (g_img.view(shp_i), None, g_rn.view(shp_rn) if use_n else None, g_sn.view(shp_sn) if use_n else None,
                g_d.view(shp_d) if use_d else None, None, None, None, None)


def fused_loss(image, gt, rend_normal=None, surf_normal=None, rend_dist=None, image_weight=None, lambda_dssim=0.2, lambda_normal=0.0,
               lambda_dist=0.0):
    """Returns (loss, terms): terms = [loss, Ll1, ssim, loss0, normal term, lambda_dist*mean(dist), psnr, mse per channel...]."""
    return _FusedLoss.apply(image, gt, rend_normal, surf_normal, rend_dist, image_weight, lambda_dssim, lambda_normal, lambda_dist)


def l1_loss(network_output, gt):
    """utils/loss_utils.py:22-23 (images [C,H,W], C <= 4)."""
    return fused_loss(network_output, gt, lambda_dssim=0.0)[0]


def ssim(img1, img2, window_size=11, size_average=True):
    """utils/loss_utils.py:93-119, the configuration the training loop uses (11x11 window, mean over the image)."""
    if window_size != 11 or not size_average:
        raise NotImplementedError("only window_size=11, size_average=True (the training configuration) is built")
    return 1.0 - fused_loss(img1, img2, lambda_dssim=1.0)[0]


def get_img_grad_weight(img, beta=2.0):
    """utils/loss_utils.py:127-140.  A function of the ground-truth image only: callers should evaluate it once per camera
    (`image_weight`, train_refnerf.py:1178-1179) instead of once per iteration."""
    _, hd, wd = img.shape
    bottom, top = img[..., 2:hd, 1:wd - 1], img[..., 0:hd - 2, 1:wd - 1]
    right, left = img[..., 1:hd - 1, 2:wd], img[..., 1:hd - 1, 0:wd - 2]
    gx = torch.mean(torch.abs(right - left), 0, keepdim=True)
    gy = torch.mean(torch.abs(top - bottom), 0, keepdim=True)
    g, _ = torch.max(torch.cat((gx, gy), dim=0), dim=0)
    g = (g - g.min()) / (g.max() - g.min())
    return torch.nn.functional.pad(g[None, None], (1, 1, 1, 1), mode="constant", value=1.0).squeeze()


def image_weight(gt_image):
    """train_refnerf.py:1178-1179."""
    return (1.0 - get_img_grad_weight(gt_image)).clamp(0, 1).detach() ** 2


def spatial_gradient(x):
    """kornia.filters.spatial_gradient(x, mode="sobel", order=1, normalized=True) for x [B,C,H,W] -> [B,C,2,H,W] (d/dx, d/dy): the 3x3
    Sobel pair divided by the sum of its absolute weights (8), replicate padding.  kornia is a pip dependency of the reference
    (utils/loss_utils.py:12) that is not installed here: restated from its documented behaviour, parity unpinned; only the absolute
    value of the result enters the losses below, so the sign convention of the kernel does not matter."""
    B, C, H, W = x.shape
    kx = torch.tensor([[-1.0, 0.0, 1.0], [-2.0, 0.0, 2.0], [-1.0, 0.0, 1.0]], dtype=x.dtype, device=x.device) / 8.0
    k = torch.stack([kx, kx.t()])[:, None]                                     # [2,1,3,3]
    xp = torch.nn.functional.pad(x.reshape(B * C, 1, H, W), (1, 1, 1, 1), mode="replicate")
    return torch.nn.functional.conv2d(xp, k).reshape(B, C, 2, H, W)


def first_order_edge_aware_loss(data, img):
    """utils/loss_utils.py:121-122: |grad data| damped by exp(-|grad img|), summed over the two directions, mean."""
    return (spatial_gradient(data[None])[0].abs() * torch.exp(-spatial_gradient(img[None])[0].abs())).sum(1).mean()


def smooth_loss(data):
    """utils/loss_utils.py:124-125."""
    return spatial_gradient(data[None])[0].abs().sum(1).mean()


_LPIPS_FN = None          # the network behind lpips_loss: callable(x, y) on [1,3,H,W] images scaled to [-1, 1] -> per-image distances
_LPIPS_WARNED = False


def set_lpips_fn(fn):
    """Register the perceptual network of utils/loss_utils.py:35-43 (`lpips.LPIPS(net='vgg')` in the reference).  Its VGG weights are a
    network download and not part of this build, so the term is pluggable: `set_lpips_fn(lpips.LPIPS(net='vgg').cuda())` where the
    package is installed, or any callable with that signature.  None unregisters."""
    global _LPIPS_FN
    _LPIPS_FN = fn


def lpips_loss(x, y, net="vgg"):
    """utils/loss_utils.py:35-43: the registered network on the images mapped from [0, 1] to [-1, 1], mean over the batch."""
    global _LPIPS_FN
    if _LPIPS_FN is None:
        try:
            import lpips as lpips_module           # the reference's own lazy construction (:38-41), where the package exists
        except ImportError as ex:
            raise NotImplementedError(
                "lpips_loss (utils/loss_utils.py:35-43) needs the LPIPS network: `pip install lpips` is not possible here and its weights are "
                "not part of this build.  Register a network with materialrefgs_amd.losses.set_lpips_fn(...) or run with "
                "--no-use_perceptual_loss (INTEGRATION.md)") from ex
        _LPIPS_FN = lpips_module.LPIPS(net=net, verbose=False).to(x.device)
    return _LPIPS_FN(x * 2.0 - 1.0, y * 2.0 - 1.0).mean()


def check_loss_config(opt):
    """Optional set-up check (a training script may call it before iteration 1): the reference enables the LPIPS term by default
    (arguments/__init__.py:223-225: use_perceptual_loss = True, active after perceptual_loss_start_iter = 18 000); raise now rather
    than 18 000 iterations into a run when no network is registered and the `lpips` package is absent."""
    if getattr(opt, "use_perceptual_loss", False) and _LPIPS_FN is None:
        try:
            import lpips  # noqa: F401
        except ImportError as ex:
            raise NotImplementedError("opt.use_perceptual_loss is set but no LPIPS network is available: materialrefgs_amd.losses.set_lpips_fn(...) "
                                      "or --no-use_perceptual_loss (INTEGRATION.md)") from ex


def calculate_loss(viewpoint_camera, pc, render_pkg, opt, iteration, image_weight=None, bg_mask=None):
    """utils/loss_utils.py:142-228: same arguments, same keys in tb_dict (values are 0-d tensors, see module docstring)."""
    use_perc = bool(getattr(opt, "use_perceptual_loss", False))
    global _LPIPS_WARNED
    if use_perc and _LPIPS_FN is None and not _LPIPS_WARNED:
        # the reference's defaults reach this state (the term is on by default and becomes active at iteration 18 000): training runs
        # as the reference does until then; say once what will happen at that iteration instead of failing at iteration 1
        _LPIPS_WARNED = True
        import warnings
        warnings.warn(f"opt.use_perceptual_loss is set and no LPIPS network is registered (materialrefgs_amd.losses.set_lpips_fn): the loss "
                      f"will raise once iteration > {getattr(opt, 'perceptual_loss_start_iter', '?')} unless the `lpips` package is importable")
    image = render_pkg["render"]
    gt_image = viewpoint_camera.original_image
    if not gt_image.is_cuda:
        gt_image = gt_image.to(image.device)
    use_normal = opt.lambda_normal_render_depth > 0 and iteration > opt.normal_loss_start
    use_dist = opt.lambda_dist > 0 and iteration > opt.dist_loss_start
    use_nsmooth = getattr(opt, "lambda_normal_smooth", 0) > 0 and opt.normal_smooth_from_iter < iteration < opt.normal_smooth_until_iter
    use_dsmooth = getattr(opt, "lambda_depth_smooth", 0) > 0 and iteration > 3000
    loss, terms = fused_loss(image, gt_image,
                             render_pkg["rend_normal"] if use_normal else None, render_pkg["surf_normal"] if use_normal else None,
                             render_pkg["rend_dist"] if use_dist else None, image_weight if use_normal else None,
                             lambda_dssim=opt.lambda_dssim, lambda_normal=opt.lambda_normal_render_depth if use_normal else 0.0,
                             lambda_dist=opt.lambda_dist if use_dist else 0.0)
    zero = terms.new_zeros(())
    tb_dict = {
        "num_points": pc.get_xyz.shape[0],
        "loss_l1": terms[1], "psnr": terms[6], "ssim": terms[2], "loss0": terms[3],
        "loss_normal_render_depth": terms[4] if use_normal else zero,
        "loss_dist": terms[5] if use_dist else zero,
        "loss_normal_smooth": zero, "loss_depth_smooth": zero,
        "loss": terms[0],
    }
    # the two edge-aware smoothness terms (:185-197; lambda = 0 in the reference's defaults): torch ops on top of the fused loss
    if use_nsmooth:
        tb_dict["loss_normal_smooth"] = first_order_edge_aware_loss(render_pkg["rend_normal"], gt_image)
        loss = loss + opt.lambda_normal_smooth * tb_dict["loss_normal_smooth"]
    if use_dsmooth:
        tb_dict["loss_depth_smooth"] = first_order_edge_aware_loss(render_pkg["surf_depth"], gt_image)
        loss = loss + opt.lambda_depth_smooth * tb_dict["loss_depth_smooth"]
    if use_perc and iteration > opt.perceptual_loss_start_iter:                         # :212-215
        perc_loss = lpips_loss(image.unsqueeze(0), gt_image.unsqueeze(0))
        loss = loss + opt.lambda_perceptual_loss * perc_loss
        tb_dict["perceptual_loss"] = perc_loss.detach()
        tb_dict["loss"] = loss.detach()
    if use_nsmooth or use_dsmooth:
        tb_dict["loss"] = loss.detach()
    return loss, tb_dict
