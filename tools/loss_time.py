"""Times the fused per-view loss (libmrgs.so) next to the same expression written with torch ops (conv2d + elementwise, the way
the reference's utils/loss_utils.py evaluates it) on the GPU.  Developer tool; prints ms per forward+backward."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from materialrefgs_amd import losses

H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 800
g = torch.Generator(device="cuda").manual_seed(0)
gt = torch.rand(3, H, W, device="cuda", generator=g)
img = (gt + 0.05 * torch.randn(3, H, W, device="cuda", generator=g)).clamp(0, 1).requires_grad_(True)
rn = torch.randn(3, H, W, device="cuda", generator=g).requires_grad_(True)
sn = torch.randn(3, H, W, device="cuda", generator=g).requires_grad_(True)
dist = torch.rand(1, H, W, device="cuda", generator=g).requires_grad_(True)
wt = torch.rand(H, W, device="cuda", generator=g)
g1 = torch.tensor([__import__("math").exp(-(x - 5) ** 2 / 4.5) for x in range(11)], device="cuda")
g1 = g1 / g1.sum()
win = (g1[:, None] * g1[None, :]).expand(3, 1, 11, 11).contiguous()


def torch_loss():
    l1 = (img - gt).abs().mean()
    mu1, mu2 = F.conv2d(img, win, padding=5, groups=3), F.conv2d(gt, win, padding=5, groups=3)
    s1 = F.conv2d(img * img, win, padding=5, groups=3) - mu1 * mu1
    s2 = F.conv2d(gt * gt, win, padding=5, groups=3) - mu2 * mu2
    s12 = F.conv2d(img * gt, win, padding=5, groups=3) - mu1 * mu2
    ss = (((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 * mu1 + mu2 * mu2 + 1e-4) * (s1 + s2 + 9e-4))).mean()
    return 0.8 * l1 + 0.2 * (1 - ss) + 0.05 * (wt * (sn - rn).abs().sum(0)).mean() + 100.0 * dist.mean()


def fused():
    return losses.fused_loss(img, gt, rn, sn, dist, wt, 0.2, 0.05, 100.0)[0]


def bench(fn, n=50):
    for _ in range(5):
        for t in (img, rn, sn, dist):
            t.grad = None
        fn().backward()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        for t in (img, rn, sn, dist):
            t.grad = None
        fn().backward()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


a, b = float(torch_loss()), float(fused())
print(f"loss torch {a:.7f} fused {b:.7f}")
print(f"{H}x{W}: torch ops {bench(torch_loss):.3f} ms   fused {bench(fused):.3f} ms  (forward + backward, wall)")
