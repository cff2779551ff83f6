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
_MAX_SHARDS = 64


class _Done:
    def wait(self):
        return True


def sum_reduce_scatter_gather(buf: torch.Tensor, group=None):
    """In-place sum of `buf` (length a multiple of the world size) over all ranks as an EXPLICIT reduce-scatter followed by an
    all-gather of the reduced shards, queued back to back; returns a handle whose wait() orders the caller's stream (host, for gloo)
    behind both.  Why not dist.all_reduce: on the 8-GPU node every pair of GPUs has its own xGMI link, and with the two halves spelled
    out every rank sends 1/V of the bucket to each peer and receives 1/V from each, twice -- (V - 1) / V of the bucket per link and
    direction in total, whatever algorithm the library would pick for an all-reduce of this size (a ring all-reduce moves the same
    bytes through ONE link per rank, 7x the time on this topology; the scaling model of bench.py assumes the direct form)."""
    world = dist.get_world_size(group)
    assert buf.is_contiguous() and buf.numel() % world == 0
    backend = dist.get_backend(group)
    if backend != "nccl" and buf.is_cuda:
        # gloo on device tensors (the one-GPU plumbing test, tests/test_dist_gpu.py): it stages through the host and has no reduce-scatter
        # for them -- the library's all-reduce; the links this function is about do not exist there
        return dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group, async_op=True)
    n = buf.numel() // world
    shard = torch.empty(n, dtype=buf.dtype, device=buf.device)
    w = dist.reduce_scatter_tensor(shard, buf, op=dist.ReduceOp.SUM, group=group, async_op=True)
    if backend != "nccl":
        w.wait()          # gloo runs its operations on worker threads: the gather must not start before the scatter has finished
    # (RCCL: both collectives are queued on the process group's own stream, in this order; `shard` is kept alive by the work object)
    return dist.all_gather_into_tensor(buf, shard, group=group, async_op=True)


class TouchedRowsExchange:
    """Sum over ranks of a [P, C] row matrix (+ a small dense tail) in which every rank holds non-zero rows only for the surfels ITS view
    touched -- a view blends the surfels in front of its camera, not the whole model -- by exchanging the union of the touched rows only:

        1. all-gather of the ranks' touched-row masks (a byte per surfel: P bytes per rank against 4 C P of the dense bucket);
        2. union of the masks -> row index list, the same on every rank (one host read of its length: the collectives need sizes);
        3. the union's rows compacted into one buffer [n_union, C] | tail, summed with sum_reduce_scatter_gather, scattered back.

    Exact: rows outside the union are zero on every rank, and zero is what comes back for them.  What it saves is the union's share of
    P -- two neighbouring views of an orbit share most of what they see (measured on the bench scene: bench.py `exchange_model`), eight
    views around an object cover it and save nothing -- so the wire bytes are reported per V, not claimed as a constant.
    `last` holds the byte counts of the most recent call."""

    def __init__(self, P: int, C: int, tail_numel: int, device, dtype=torch.float32):
        self.P, self.C, self.tail = int(P), int(C), int(tail_numel)
        self.device, self.dtype = device, dtype
        self.masks = None
        self.last = {}

    def exchange(self, rows: torch.Tensor, tail: Optional[torch.Tensor], group=None):
        """rows [P, C] (this rank's, zero where untouched), tail [tail_numel] or None -> (summed rows [P, C], summed tail)."""
        world = dist.get_world_size(group)
        P, C = self.P, self.C
        assert rows.shape == (P, C) and rows.is_contiguous()
        mine = (rows != 0).any(dim=1).to(torch.uint8)
        if self.masks is None or self.masks.shape[0] != world:
            self.masks = torch.empty((world, P), dtype=torch.uint8, device=rows.device)
        dist.all_gather_into_tensor(self.masks.view(-1), mine, group=group)
        idx = self.masks.any(dim=0).nonzero(as_tuple=False).squeeze(1)          # (one host read: the union's size)
        n = int(idx.numel())
        total = n * C + self.tail
        padded = (total + world - 1) // world * world
        buf = torch.zeros(max(padded, world), dtype=self.dtype, device=rows.device)
        if n:
            buf[:n * C].view(n, C).copy_(rows.index_select(0, idx))
        if self.tail:
            buf[n * C:n * C + self.tail].copy_(tail.reshape(-1))
        sum_reduce_scatter_gather(buf, group).wait()
        out = torch.zeros_like(rows)
        if n:
            out.index_copy_(0, idx, buf[:n * C].view(n, C))
        self.last = {"touched_rows_this_rank": int(mine.sum()), "union_rows": n, "rows": P, "mask_bytes_per_rank": P,
                     "dense_bytes": 4 * (P * C + self.tail), "exchanged_bytes": 4 * padded}
        return out, (buf[n * C:n * C + self.tail] if self.tail else None)


class GradBucket:
    """Flat fp32 bucket that packs a fixed list of gradient tensors for a single all-reduce."""

    def __init__(self, shapes: Sequence[torch.Size], device, dtype=torch.float32):
        self.shapes = [torch.Size(s) for s in shapes]
        self.numels = [int(torch.Size(s).numel()) for s in self.shapes]
        self.offsets = [0]
        for n in self.numels:
            self.offsets.append(self.offsets[-1] + n)
        # (room behind the last tensor so that the bucket can be cut into `world` equal shards for any world <= _MAX_SHARDS:
        #  sum_reduce_scatter_gather)
        self._store = torch.zeros(self.offsets[-1] + _MAX_SHARDS, dtype=dtype, device=device)
        self.flat = self._store[:self.offsets[-1]]

    def padded(self, world: int):
        """The bucket as a tensor whose length is a multiple of `world` (the tail beyond the packed tensors is zeros nobody reads)."""
        n = self.offsets[-1]
        return self._store[:(n + world - 1) // world * world]

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
        world = dist.get_world_size(group)
        sum_reduce_scatter_gather(bucket.padded(world), group).wait()
        if average:
            flat.div_(world)
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
        self._early_void = False     # begin_early was called twice in one step: the early row is not the whole factor (see there)

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
        if self._early is not None:
            # A second rasterizer backward before reduce() (two renders share the SH tensor in one step, or a step was abandoned before
            # its reduce): the row of the first call is NOT the whole factor any more -- dL/dsh then sums two rank-one terms with different
            # directions -- so the early gather is dropped (after it has finished with the buffers) and reduce() gathers dL/dsh[:, 0, :] / SH_C0 as
            # without the hook: exact for one camera per step, and the only form that exists for several.
            self._early.wait()
            self._early, self._early_void = None, True
            return
        if getattr(self, "_early_void", False):
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
        w1, self._early, self._early_void = self._early, None, False
        if w1 is None:
            if sh is None:
                self.row[:3 * P].zero_()
            else:
                torch.div(sh[:, 0, :], SH_C0, out=self.row[:3 * P].view(P, 3))
            self.row[3 * P:].copy_(campos.reshape(-1))
            w1 = self._gather(world, group)
        del flat
        w2 = sum_reduce_scatter_gather(self.small.padded(world), group)
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


def expand_surfel_sh_gradient_rows(gathered_rgb: torch.Tensor, gathered_ind: Optional[torch.Tensor], xyz: torch.Tensor, rotation_raw: torch.Tensor,
                                   sh_degree: int, family: str = "both"):
    """expand_surfel_sh_gradients from the two buffers the two all-gathers fill: gathered_rgb [V, 3P + 3] rows [dRGB_v | campos_v],
    gathered_ind [V, 3P] rows dIND_v (mrgs_sh_grad_expand_surfel_rows); no CPU path.  family "rgb" / "ind": only that family's two tensors
    (dc [P,1,3], rest [P,15,3]) -- a step expands each family when its gather has landed; "both": all four."""
    V, P = gathered_rgb.shape[0], xyz.shape[0]
    assert gathered_rgb.shape == (V, 3 * P + 3) and gathered_rgb.is_contiguous()
    assert family == "rgb" or (gathered_ind.shape == (V, 3 * P) and gathered_ind.is_contiguous())
    if not xyz.is_cuda:
        raise RuntimeError("expand_surfel_sh_gradient_rows needs CUDA(HIP) tensors: the expansion runs in libmrgs.so, there is no CPU path")
    from . import _lib
    o = dict(dtype=torch.float32, device=xyz.device)
    a = [torch.empty((P, 1, 3), **o), torch.empty((P, 15, 3), **o)] if family != "ind" else [None, None]
    b = [torch.empty((P, 1, 3), **o), torch.empty((P, 15, 3), **o)] if family != "rgb" else [None, None]
    x3, q4 = xyz.detach().float().contiguous(), rotation_raw.detach().float().contiguous()
    cam = gathered_rgb[:, 3 * P:]
    ptr = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
    with torch.cuda.device(xyz.device):
        st = ctypes.c_void_p(torch.cuda.current_stream(xyz.device).cuda_stream)
        _lib.check(_lib.lib().mrgs_sh_grad_expand_surfel_rows(
            P, int(sh_degree), V, ptr(x3), ptr(q4), ptr(gathered_rgb) if family != "ind" else None, gathered_rgb.stride(0),
            ptr(gathered_ind) if family != "rgb" else None, gathered_ind.stride(0) if family != "rgb" else 0, ptr(cam), gathered_rgb.stride(0),
            ptr(a[0]), ptr(a[1]), ptr(b[0]), ptr(b[1]), st))
    return [t for t in a + b if t is not None]


class SurfelGradReducer:
    """View-parallel gradient sum for render_surfel's parameter set (BASELINE config 5).  `names` labels the gradient tensors in the
    order they are passed and must contain "xyz", "rotation", "features_dc", "features_rest", "indirect_dc", "indirect_rest"; the four
    SH tensors (96 of 111 floats per gaussian) are exchanged as 6 floats per gaussian (all-gather) and rebuilt locally, everything else
    (incl. tensors that are not per-gaussian, e.g. the environment cubemap) goes through one flat all-reduce."""
    SH_NAMES = ("features_dc", "features_rest", "indirect_dc", "indirect_rest")

    def __init__(self, shapes: Sequence[torch.Size], names: Sequence[str], device, expand_fn=None, touched_rows_only=False):
        """touched_rows_only: the dense part (everything but the four SH tensors) travels as the UNION OF THE TOUCHED ROWS of the ranks'
        views instead of all P rows (TouchedRowsExchange: one more small collective and one host read per step; exact)."""
        self.expand_fn = expand_fn or expand_surfel_sh_gradients   # tests on CPU tensors pass the oracle.dist_oracle restatement
        self.names = list(names)
        assert all(n in self.names for n in self.SH_NAMES + ("xyz", "rotation"))
        self.sh_pos = [self.names.index(n) for n in self.SH_NAMES]
        self.dense_pos = [i for i in range(len(self.names)) if i not in self.sh_pos]
        self.dense = GradBucket([shapes[i] for i in self.dense_pos], device)
        self.P = int(shapes[self.names.index("xyz")][0])
        self.shapes = [torch.Size(s_) for s_ in shapes]
        # the dense tensors that are one row per surfel (their columns side by side make the row matrix), and the rest (the cubemap)
        self.row_pos = [i for i in self.dense_pos if len(shapes[i]) >= 1 and int(shapes[i][0]) == self.P]
        self.tail_pos = [i for i in self.dense_pos if i not in self.row_pos]
        self.row_cols = [int(torch.Size(shapes[i]).numel()) // self.P for i in self.row_pos]
        self.touched = TouchedRowsExchange(self.P, sum(self.row_cols), sum(int(torch.Size(shapes[i]).numel()) for i in self.tail_pos), device) \
            if touched_rows_only else None
        self._ind_zero = False
        # the two factors travel apart (they are final at different times of a backward): [dRGB | campos] and [dIND]
        self.row_rgb = torch.empty(3 * self.P + 3, dtype=torch.float32, device=device)
        self.row_ind = torch.empty(3 * self.P, dtype=torch.float32, device=device)
        self.gathered_rgb = self.gathered_ind = None
        self._early_rgb = self._early_ind = None
        self._void = False

    def _buffers(self, world):
        if self.gathered_rgb is None or self.gathered_rgb.shape[0] != world:
            self.gathered_rgb = torch.empty((world, 3 * self.P + 3), dtype=torch.float32, device=self.row_rgb.device)
            self.gathered_ind = torch.empty((world, 3 * self.P), dtype=torch.float32, device=self.row_rgb.device)

    def _world(self, group):
        return dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1

    def begin_early_rgb(self, dRGB: torch.Tensor, campos: torch.Tensor, group=None):
        """Start the all-gather of this view's colour factor in the middle of the backward: `dRGB` [P,3] is what the rasterizer hands out
        between its blend backward and its per-gaussian backward (rasterizer.set_after_blend_hook) -- final there, because render_surfel
        rasterizes the colour SH exactly once.  The collective runs next to the per-gaussian backward, the per-gaussian glue's backward
        and whatever follows.  A second call before reduce() (two rasterizations in one step share the SH tensors: the sum of two rank-one
        terms is not one row) voids the early path for this step: reduce() then gathers dL/dfeatures_dc[:, 0, :] / SH_C0 as without hooks."""
        world = self._world(group)
        if world == 1:
            return
        if self._early_rgb is not None or self._void:
            if self._early_rgb is not None:
                self._early_rgb.wait()
            self._early_rgb, self._void = None, True
            return
        P = self.P
        self._buffers(world)
        self.row_rgb[:3 * P].view(P, 3).copy_(dRGB)
        self.row_rgb[3 * P:].copy_(campos.reshape(-1))
        self._early_rgb = dist.all_gather_into_tensor(self.gathered_rgb.view(-1), self.row_rgb, group=group, async_op=True)

    def begin_early_ind(self, d_indirect_dc: torch.Tensor, group=None):
        """The second factor, dIND = dL/dindirect_dc[:, 0, :] / SH_C0, as soon as the per-gaussian glue's backward has produced it
        (renderer.set_after_features_hook): its all-gather runs under the packing of the dense bucket and the first collective."""
        world = self._world(group)
        if world == 1:
            return
        if d_indirect_dc is None:
            # the indirect radiance is not looked at in this step (render_surfel without opt.indirect: the blended indirect light feeds no
            # output) -- its factor is zero on EVERY rank by the structure of the step, not by its data: nothing is gathered, nothing
            # expanded, the two indirect SH gradients come back as the zeros they are
            self._ind_zero = True
            return
        if self._early_ind is not None or self._void:
            if self._early_ind is not None:
                self._early_ind.wait()
            self._early_ind, self._void = None, True
            return
        P = self.P
        self._buffers(world)
        torch.div(d_indirect_dc.reshape(P, 3), SH_C0, out=self.row_ind.view(P, 3))
        self._early_ind = dist.all_gather_into_tensor(self.gathered_ind.view(-1), self.row_ind, group=group, async_op=True)

    def reduce(self, tensors: Sequence[Optional[torch.Tensor]], xyz: torch.Tensor, rotation_raw: torch.Tensor, campos: torch.Tensor,
               sh_degree: int, group=None):
        """Returns the summed gradients in the order of `tensors`."""
        world = self._world(group)
        if world == 1:
            return list(tensors)
        P = self.P
        self._buffers(world)
        w_rgb, w_ind, void = self._early_rgb, self._early_ind, self._void
        self._early_rgb = self._early_ind = None
        self._void = False
        if void:                     # (whatever was started early has been waited for in begin_early_*)
            w_rgb = w_ind = None

        def seg(name, out):
            t = tensors[self.names.index(name)]
            if t is None:
                out.zero_()
            else:
                torch.div(t[:, 0, :], SH_C0, out=out.view(P, 3))
        ind_zero, self._ind_zero = self._ind_zero, False
        if w_rgb is None:
            seg("features_dc", self.row_rgb[:3 * P])
            self.row_rgb[3 * P:].copy_(campos.reshape(-1))
            w_rgb = dist.all_gather_into_tensor(self.gathered_rgb.view(-1), self.row_rgb, group=group, async_op=True)
        if w_ind is None and not ind_zero:
            seg("indirect_dc", self.row_ind)
            w_ind = dist.all_gather_into_tensor(self.gathered_ind.view(-1), self.row_ind, group=group, async_op=True)
        dense_out = None
        if self.touched is not None:
            rows = torch.cat([(tensors[i] if tensors[i] is not None else torch.zeros(self.shapes[i], dtype=torch.float32, device=self.row_rgb.device))
                              .reshape(P, -1) for i in self.row_pos], dim=1).contiguous()
            tail = torch.cat([(tensors[i] if tensors[i] is not None else torch.zeros(self.shapes[i], dtype=torch.float32, device=self.row_rgb.device))
                              .reshape(-1) for i in self.tail_pos]) if self.tail_pos else None
            rows_sum, tail_sum = self.touched.exchange(rows, tail, group)
            dense_out, c0, t0 = {}, 0, 0
            for i, c in zip(self.row_pos, self.row_cols):
                dense_out[i] = rows_sum[:, c0:c0 + c].reshape(self.shapes[i])
                c0 += c
            for i in self.tail_pos:
                n_ = int(self.shapes[i].numel())
                dense_out[i] = tail_sum[t0:t0 + n_].view(self.shapes[i])
                t0 += n_
            w2 = _Done()
        else:
            self.dense.pack([tensors[i] for i in self.dense_pos])
            w2 = sum_reduce_scatter_gather(self.dense.padded(world), group)
        w_rgb.wait()
        zeros_ind = lambda: [torch.zeros((P, 1, 3), dtype=torch.float32, device=self.row_rgb.device),
                             torch.zeros((P, 15, 3), dtype=torch.float32, device=self.row_rgb.device)]
        if self.expand_fn is expand_surfel_sh_gradients:
            # each family as soon as ITS rows are there: the first expansion runs under the second gather and the dense exchange
            sh = expand_surfel_sh_gradient_rows(self.gathered_rgb, None, xyz, rotation_raw, sh_degree, family="rgb")
            if ind_zero:
                sh = sh + zeros_ind()
            else:
                w_ind.wait()
                sh = sh + expand_surfel_sh_gradient_rows(self.gathered_rgb, self.gathered_ind, xyz, rotation_raw, sh_degree, family="ind")
        else:       # (an injected checker takes the rows in one piece: [dRGB | dIND | campos])
            if ind_zero:
                self.gathered_ind.zero_()
            else:
                w_ind.wait()
            rows_ = torch.cat((self.gathered_rgb[:, :3 * P], self.gathered_ind, self.gathered_rgb[:, 3 * P:]), dim=1).contiguous()
            sh = self.expand_fn(rows_, xyz, rotation_raw, sh_degree)
        w2.wait()
        out = [None] * len(self.names)
        if dense_out is not None:
            for i in self.dense_pos:
                out[i] = dense_out[i]
        else:
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
