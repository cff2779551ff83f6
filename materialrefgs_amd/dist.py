"""View-parallel data parallelism for the render hot path (new capability; the reference is single-process,
batch = 1 view per iteration, train_refnerf.py:1166-1173 -- SURVEY.md section 8e).

One process per GPU, gaussians replicated, rank r renders view r of the step.  The path has exactly one real
exchange: the sum of the dense per-gaussian gradient tensors.  They are packed into ONE flat bucket and reduced
with a single all-reduce (RCCL over xGMI on the GPU box, gloo in the CPU tests); xGMI is point-to-point, so one
large collective per step beats many small ones.  The two densification statistics of the reference
(train_refnerf.py:1416-1418, gaussian_model.py:1059-1061) need a sum and a max reduction, provided below.

The SH colour gradient is 48 of the 61 gradient floats per gaussian, but it has rank-one structure per view:
dL/dsh_v[p][k][c] = B_k(dir_v(p)) * dRGB_v[p][c] (backward.cu:22-141).  `FactoredGradReducer` therefore all-gathers the three
floats dRGB_v[p] (= dL/dsh_v[p][0] / SH_C0) plus the camera centre of every rank and lets each rank rebuild
sum_v dL/dsh_v locally (csrc/mrgs_surfel.hip: sh_grad_expand_kernel), and all-reduces only the other 13 floats: 2.4x fewer
bytes over the point-to-point xGMI links than the dense all-reduce, which at 300k surfels costs about as much as the render.
`SurfelGradReducer` does the same for render_surfel's parameter set (111 floats per gaussian, BASELINE config 5), whose second SH
family (indirect radiance along the mirror direction) factors the same way: 21 floats per gaussian on the wire instead of 111.
"""
import ctypes
from typing import Dict, List, Optional, Sequence

import torch
import torch.distributed as dist

SH_C0 = 0.28209479177387814


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
        if all(t is not None and t.dtype == self.flat.dtype and t.device == self.flat.device for t in tensors):
            # one concatenation kernel instead of a copy launch per tensor (a dozen small launches per step at N > 1)
            torch.cat([t.reshape(-1) for t in tensors], out=self.flat)
            return self.flat
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


def expand_sh_gradients(gathered: torch.Tensor, means3D: torch.Tensor, M: int, sh_degree: int) -> torch.Tensor:
    """sum_v B_k(normalize(means3D - campos_v)) * dRGB_v  ->  [P, M, 3].  gathered: [V, 3P + 3] rows [dRGB_v | campos_v].
    Runs in libmrgs.so (mrgs_sh_grad_expand); there is no CPU path -- the gloo tests of the collective plumbing inject their own
    checker through the reducers' `expand_fn` argument."""
    V, P = gathered.shape[0], means3D.shape[0]
    assert gathered.shape[1] == 3 * P + 3 and gathered.is_contiguous()
    if not means3D.is_cuda:
        raise RuntimeError("expand_sh_gradients needs CUDA(HIP) tensors: the expansion runs in libmrgs.so, there is no CPU path")
    from . import _lib
    out = torch.empty((P, M, 3), dtype=torch.float32, device=means3D.device)
    m3 = means3D.detach().float().contiguous()
    with torch.cuda.device(means3D.device):
        st = ctypes.c_void_p(torch.cuda.current_stream(means3D.device).cuda_stream)
        _lib.check(_lib.lib().mrgs_sh_grad_expand(P, M, int(sh_degree), V, ctypes.c_void_p(m3.data_ptr()),
                                                  ctypes.c_void_p(gathered.data_ptr()), gathered.stride(0),
                                                  ctypes.c_void_p(out.data_ptr()), st))
    return out


class FactoredGradReducer:
    """Sum of the per-view gradients over all ranks with the SH gradient sent in factored form (module docstring).
    `shapes`: shapes of the gradient tensors in the order they will be passed; `sh_index`: position of dL/dsh [P, M, 3]."""

    def __init__(self, shapes: Sequence[torch.Size], sh_index: int, device, expand_fn=None):
        self.expand_fn = expand_fn or expand_sh_gradients   # tests on CPU tensors pass oracle.dist_oracle.expand_sh_gradients
        self.sh_index = sh_index
        self.sh_shape = torch.Size(shapes[sh_index])
        self.small = GradBucket([s for i, s in enumerate(shapes) if i != sh_index], device)
        P = self.sh_shape[0]
        self.row = torch.empty(3 * P + 3, dtype=torch.float32, device=device)
        self.gathered = None
        self._early = None

    def _gather(self, world, group):
        P = self.sh_shape[0]
        if self.gathered is None or self.gathered.shape[0] != world:
            self.gathered = torch.empty((world, 3 * P + 3), dtype=torch.float32, device=self.row.device)
        return dist.all_gather_into_tensor(self.gathered.view(-1), self.row, group=group, async_op=True)   # flat output: gloo insists

    def begin_early(self, dRGB: torch.Tensor, campos: torch.Tensor, group=None):
        """Start the all-gather of this view's factor BEFORE its backward has finished: `dRGB` [P,3] is the clamp-masked colour gradient
        the rasterizer hands out between the blend backward and the per-gaussian backward (rasterizer.set_after_blend_hook;
        mrgs_rasterize_backward_blend).  The collective is issued behind what the current stream holds at this moment, so it runs next
        to the per-gaussian backward queued afterwards; the following reduce() picks the gathered rows up instead of gathering
        dL/dsh[:, 0, :] / SH_C0 (the same numbers to one rounding)."""
        world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        if world == 1:
            return
        P = self.sh_shape[0]
        self.row[:3 * P].view(P, 3).copy_(dRGB)
        self.row[3 * P:].copy_(campos.reshape(-1))
        self._early = self._gather(world, group)

    def reduce(self, tensors: Sequence[Optional[torch.Tensor]], means3D: torch.Tensor, campos: torch.Tensor, sh_degree: int, group=None):
        """Returns the summed gradients in the order of `tensors` (views into internal buffers)."""
        world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        sh = tensors[self.sh_index]
        P, M = self.sh_shape[0], self.sh_shape[1]
        small = [t for i, t in enumerate(tensors) if i != self.sh_index]
        flat = self.small.pack(small)
        if world == 1:
            views = self.small.views()
            return views[:self.sh_index] + [sh] + views[self.sh_index:]
        w1, self._early = self._early, None
        if w1 is None:
            if sh is None:
                self.row[:3 * P].zero_()
            else:
                torch.div(sh[:, 0, :], SH_C0, out=self.row[:3 * P].view(P, 3))
            self.row[3 * P:].copy_(campos.reshape(-1))
            w1 = self._gather(world, group)
        w2 = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True)
        w1.wait()
        sh_sum = self.expand_fn(self.gathered, means3D, M, sh_degree)
        w2.wait()
        views = self.small.views()
        return views[:self.sh_index] + [sh_sum] + views[self.sh_index:]


def expand_surfel_sh_gradients(gathered: torch.Tensor, xyz: torch.Tensor, rotation_raw: torch.Tensor, sh_degree: int):
    """gathered [V, 6P + 3] rows [dRGB_v | dIND_v | campos_v] -> summed gradients of (features_dc [P,1,3], features_rest [P,15,3],
    indirect_dc [P,1,3], indirect_rest [P,15,3]) in libmrgs.so (mrgs_sh_grad_expand_surfel); no CPU path (see expand_sh_gradients)."""
    V, P = gathered.shape[0], xyz.shape[0]
    assert gathered.shape[1] == 6 * P + 3 and gathered.is_contiguous()
    if not xyz.is_cuda:
        raise RuntimeError("expand_surfel_sh_gradients needs CUDA(HIP) tensors: the expansion runs in libmrgs.so, there is no CPU path")
    from . import _lib
    o = dict(dtype=torch.float32, device=xyz.device)
    out = [torch.empty((P, 1, 3), **o), torch.empty((P, 15, 3), **o), torch.empty((P, 1, 3), **o), torch.empty((P, 15, 3), **o)]
    x3, q4 = xyz.detach().float().contiguous(), rotation_raw.detach().float().contiguous()
    with torch.cuda.device(xyz.device):
        st = ctypes.c_void_p(torch.cuda.current_stream(xyz.device).cuda_stream)
        _lib.check(_lib.lib().mrgs_sh_grad_expand_surfel(P, int(sh_degree), V, ctypes.c_void_p(x3.data_ptr()), ctypes.c_void_p(q4.data_ptr()),
                                                         ctypes.c_void_p(gathered.data_ptr()), gathered.stride(0),
                                                         *[ctypes.c_void_p(t.data_ptr()) for t in out], st))
    return out


class SurfelGradReducer:
    """View-parallel gradient sum for render_surfel's parameter set (BASELINE config 5).  `names` labels the gradient tensors in the
    order they are passed and must contain "xyz", "rotation", "features_dc", "features_rest", "indirect_dc", "indirect_rest"; the four
    SH tensors (96 of 111 floats per gaussian) are exchanged as 6 floats per gaussian (all-gather) and rebuilt locally, everything else
    (incl. tensors that are not per-gaussian, e.g. the environment cubemap) goes through one flat all-reduce."""
    SH_NAMES = ("features_dc", "features_rest", "indirect_dc", "indirect_rest")

    def __init__(self, shapes: Sequence[torch.Size], names: Sequence[str], device, expand_fn=None):
        self.expand_fn = expand_fn or expand_surfel_sh_gradients   # tests on CPU tensors pass the oracle.dist_oracle restatement
        self.names = list(names)
        assert all(n in self.names for n in self.SH_NAMES + ("xyz", "rotation"))
        self.sh_pos = [self.names.index(n) for n in self.SH_NAMES]
        self.dense_pos = [i for i in range(len(self.names)) if i not in self.sh_pos]
        self.dense = GradBucket([shapes[i] for i in self.dense_pos], device)
        self.P = int(shapes[self.names.index("xyz")][0])
        self.row = torch.empty(6 * self.P + 3, dtype=torch.float32, device=device)
        self.gathered = None

    def reduce(self, tensors: Sequence[Optional[torch.Tensor]], xyz: torch.Tensor, rotation_raw: torch.Tensor, campos: torch.Tensor,
               sh_degree: int, group=None):
        """Returns the summed gradients in the order of `tensors`."""
        world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        if world == 1:
            return list(tensors)
        P = self.P
        flat = self.dense.pack([tensors[i] for i in self.dense_pos])
        for k, name in enumerate(("features_dc", "indirect_dc")):
            t = tensors[self.names.index(name)]
            seg = self.row[3 * P * k:3 * P * (k + 1)]
            if t is None:
                seg.zero_()
            else:
                torch.div(t[:, 0, :], SH_C0, out=seg.view(P, 3))
        self.row[6 * P:].copy_(campos.reshape(-1))
        if self.gathered is None or self.gathered.shape[0] != world:
            self.gathered = torch.empty((world, 6 * P + 3), dtype=torch.float32, device=self.row.device)
        w1 = dist.all_gather_into_tensor(self.gathered.view(-1), self.row, group=group, async_op=True)
        w2 = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True)
        w1.wait()
        sh = self.expand_fn(self.gathered, xyz, rotation_raw, sh_degree)
        w2.wait()
        out = [None] * len(self.names)
        for i, v in zip(self.dense_pos, self.dense.views()):
            out[i] = v
        for i, v in zip(self.sh_pos, sh):
            out[i] = v
        return out


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
            torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return {"world": world, "rank": rank, "local": local}
