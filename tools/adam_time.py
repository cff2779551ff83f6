"""Times the one-launch Adam step against torch.optim.Adam on the C3 parameter set (P = 300 000).  Developer tool."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from materialrefgs_amd.optim import Adam
P = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
dev = "cuda"
shapes = [(P, 3), (P, 1, 3), (P, 15, 3), (P, 1), (P, 2), (P, 4), (P, 1), (P, 3), (P, 3), (P, 1), (P, 1), (P, 1, 3), (P, 15, 3), (P, 32, 5),
          (6, 128, 128, 3), (6, 128, 128, 3)]
n = sum(torch.Size(s).numel() for s in shapes)
for name, cls, kw in (("fused", Adam, {}), ("torch foreach", torch.optim.Adam, {"foreach": True}), ("torch fused", torch.optim.Adam, {"fused": True})):
    ps = [torch.nn.Parameter(torch.randn(*s, device=dev)) for s in shapes]
    opt = cls([{"params": [p], "lr": 1e-3} for p in ps], lr=0.0, eps=1e-15, **kw)
    for p in ps:
        p.grad = torch.randn_like(p)
    for _ in range(3):
        opt.step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        opt.step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{name}: {n / 1e6:.1f} M elements  {ms:.3f} ms/step  {28 * n / ms / 1e6:.0f} GB/s of the 28 B/element stream")
