"""View-parallel data parallelism for the render hot path (new capability; the reference is single-process,
batch = 1 view per iteration, train_refnerf.py:1166-1173 -- SURVEY.md section 8e).

One process per GPU, gaussians replicated, rank r renders view r of the step.  The path has exactly one real
exchange: the sum of the dense per-gaussian gradient tensors.  They are packed into ONE flat bucket and reduced
with a single all-reduce (RCCL over xGMI on the GPU box, gloo in the CPU tests); xGMI is point-to-point, so one
large collective per step beats many small ones.  The two densification statistics of the reference
(train_refnerf.py:1416-1418, gaussian_model.py:1059-1061) need a sum and a max reduction, provided below.
"""
from typing import Dict, List, Optional, Sequence

import torch
import torch.distributed as dist


class GradBucket:
    """Flat fp32 bucket that packs a fixed list of gradient tensors for a single all-reduce."""

    def __init__(self, shapes: Sequence[torch.Size], device, dtype=torch.float32):
        self.shapes = [torch.Size(s) for s in shapes]
        self.numels = [int(torch.Size(s).numel()) for s in self.shapes]
        self.offsets = [0]
        for n in self.numels:
            self.offsets.append(self.offsets[-1] + n)
        self.flat = torch.zeros(self.offsets[-1], dtype=dtype, device=device)

    def pack(self, tensors: Sequence[Optional[torch.Tensor]]):
        assert len(tensors) == len(self.shapes)
        for t, o, n in zip(tensors, self.offsets, self.numels):
            if t is None:
                self.flat[o:o + n].zero_()
            else:
                self.flat[o:o + n].copy_(t.reshape(-1))
        return self.flat

    def views(self) -> List[torch.Tensor]:
        return [self.flat[o:o + n].view(s) for o, n, s in zip(self.offsets, self.numels, self.shapes)]


def allreduce_gradients(bucket: GradBucket, tensors: Sequence[Optional[torch.Tensor]], group=None, average: bool = False):
    """Sum (or mean) the per-view gradients over all ranks with one collective; returns views into the bucket."""
    flat = bucket.pack(tensors)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        if average:
            flat.div_(dist.get_world_size(group))
    return bucket.views()


def reduce_densification_stats(viewspace_grad_norm: torch.Tensor, visible: torch.Tensor, radii: torch.Tensor, group=None):
    """Reductions the reference's densification bookkeeping needs when views are spread over ranks:
    sum of ||viewspace grad|| and of the visibility count (add_densification_stats, gaussian_model.py:1059-1061)
    and max of the screen radii (max_radii2D update, train_refnerf.py:1416-1417)."""
    stats = torch.stack([viewspace_grad_norm.float(), visible.float()], dim=0).contiguous()
    radii = radii.clone()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(radii, op=dist.ReduceOp.MAX, group=group)
    return stats[0], stats[1], radii


def init_from_env(backend: Optional[str] = None) -> Dict[str, int]:
    """torchrun-style initialisation (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment)."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"   # "nccl" IS RCCL on ROCm
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return {"world": world, "rank": rank, "local": local}
