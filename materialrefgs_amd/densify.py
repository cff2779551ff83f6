"""Row compaction for densify / prune and the optimizer surgery around it, mirroring
GaussianModel._prune_optimizer / prune_points / cat_tensors_to_optimizer / replace_tensor_to_optimizer
(scene/gaussian_model.py:856-960).  `compact_rows` drops the rows of many tensors at once (csrc/mrgs_optim.hip: one scan of the
mask, one host read of the survivor count, one gather launch) where the reference runs `tensor[mask]` ~50 times; the functions
below keep the reference's dictionary-of-named-groups protocol so that its densification logic can call them unchanged.
No CPU path: the tensors must live on the GPU."""
import ctypes
from typing import Dict, Iterable, List, Sequence, Tuple

import torch
from torch import nn

from . import _lib

SKIP_GROUPS = ("mlp", "env", "env2")          # not per-gaussian (gaussian_model.py:858, 909)


def compact_rows(tensors: Sequence[torch.Tensor], keep: torch.Tensor) -> Tuple[List[torch.Tensor], int]:
    """[t[keep] for t in tensors] for 4-byte-element tensors sharing their first dimension; returns (new tensors, survivor count)."""
    if len(tensors) == 0:
        return [], int(keep.sum())
    n = int(keep.shape[0])
    dev = keep.device
    if not keep.is_cuda:
        raise RuntimeError("materialrefgs_amd.densify needs device tensors (libmrgs.so has no CPU path)")
    k8 = keep.to(torch.uint8).contiguous()
    src = []
    for t in tensors:
        if t.shape[0] != n or t.element_size() != 4 or t.device != dev:
            raise ValueError("compact_rows: tensors must have 4-byte elements, live on the mask's device and share dim 0 with it")
        src.append(t.detach().contiguous())
    lib = _lib.lib()
    with _lib.guard(dev):
        st = _lib.stream_ptr(dev)
        ws = torch.empty(lib.mrgs_compact_ws_bytes(n), dtype=torch.uint8, device=dev)
        cnt = torch.empty(1, dtype=torch.int64, device=dev)
        _lib.check(lib.mrgs_compact_count(n, ctypes.c_void_p(k8.data_ptr()), ctypes.c_void_p(ws.data_ptr()), ws.numel(),
                                          ctypes.c_void_p(cnt.data_ptr()), st))
        m = int(cnt.item())                                    # the one host read of the whole operation
        out = [torch.empty((m,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev) for t in src]
        arr = (_lib.MrgsCompactTensor * len(src))()
        for i, (s, d) in enumerate(zip(src, out)):
            arr[i] = _lib.MrgsCompactTensor(s.data_ptr(), d.data_ptr(), int(s.numel() // max(n, 1)))
        if m > 0:
            _lib.check(lib.mrgs_compact_rows(n, ctypes.c_void_p(k8.data_ptr()), ctypes.c_void_p(ws.data_ptr()), arr, len(src), st))
    return out, m


def _per_gaussian_groups(optimizer) -> Iterable[dict]:
    for group in optimizer.param_groups:
        if group.get("name") in SKIP_GROUPS:
            continue
        assert len(group["params"]) == 1
        yield group


def prune_optimizer(optimizer, keep: torch.Tensor, extra: Sequence[torch.Tensor] = ()):
    """_prune_optimizer (:856-874) for every per-gaussian group at once.  `extra`: further per-gaussian tensors to compact with the same
    mask (xyz_gradient_accum, denom, max_radii2D of prune_points :900-905).  Returns (optimizable_tensors, compacted extras)."""
    groups = list(_per_gaussian_groups(optimizer))
    tensors, slots = [], []
    for g in groups:
        p = g["params"][0]
        st = optimizer.state.get(p, None)
        tensors.append(p.data); slots.append((g, "param"))
        if st is not None and "exp_avg" in st:
            tensors.append(st["exp_avg"]); slots.append((g, "exp_avg"))
            tensors.append(st["exp_avg_sq"]); slots.append((g, "exp_avg_sq"))
    n_model = len(tensors)
    out, _m = compact_rows(tensors + list(extra), keep)
    new = {}
    for (g, kind), t in zip(slots, out[:n_model]):
        new.setdefault(id(g), {})[kind] = t
    optimizable = {}
    for g in groups:
        old = g["params"][0]
        st = optimizer.state.get(old, None)
        vals = new[id(g)]
        if st is not None:
            if "exp_avg" in vals:
                st["exp_avg"], st["exp_avg_sq"] = vals["exp_avg"], vals["exp_avg_sq"]
            del optimizer.state[old]
        g["params"][0] = nn.Parameter(vals["param"].requires_grad_(True))
        if st is not None:
            optimizer.state[g["params"][0]] = st
        optimizable[g["name"]] = g["params"][0]
    return optimizable, out[n_model:]


def cat_tensors_to_optimizer(optimizer, tensors_dict: Dict[str, torch.Tensor]):
    """cat_tensors_to_optimizer (:907-929): append rows, zero moments for the new ones."""
    optimizable = {}
    for g in _per_gaussian_groups(optimizer):
        ext = tensors_dict[g["name"]]
        old = g["params"][0]
        st = optimizer.state.get(old, None)
        if st is not None:
            st["exp_avg"] = torch.cat((st["exp_avg"], torch.zeros_like(ext)), dim=0)
            st["exp_avg_sq"] = torch.cat((st["exp_avg_sq"], torch.zeros_like(ext)), dim=0)
            del optimizer.state[old]
        g["params"][0] = nn.Parameter(torch.cat((old.data, ext), dim=0).requires_grad_(True))
        if st is not None:
            optimizer.state[g["params"][0]] = st
        optimizable[g["name"]] = g["params"][0]
    return optimizable


def replace_tensor_to_optimizer(optimizer, tensor: torch.Tensor, name: str):
    """replace_tensor_to_optimizer (:840-854): new values for one group, moments reset."""
    optimizable = {}
    for g in optimizer.param_groups:
        if g.get("name") == name:
            old = g["params"][0]
            st = optimizer.state.get(old, None)
            if st is not None:
                st["exp_avg"], st["exp_avg_sq"] = torch.zeros_like(tensor), torch.zeros_like(tensor)
                del optimizer.state[old]
            g["params"][0] = nn.Parameter(tensor.requires_grad_(True))
            if st is not None:
                optimizer.state[g["params"][0]] = st
            optimizable[name] = g["params"][0]
    return optimizable
