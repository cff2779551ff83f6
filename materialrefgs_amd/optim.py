"""Optimizer of the training loop: `Adam` is a drop-in for the `torch.optim.Adam(l, lr=0.0, eps=1e-15)` that
GaussianModel.training_setup builds (scene/gaussian_model.py:417-453).  Same constructor, `param_groups` and per-parameter
`state` keys ("step", "exp_avg", "exp_avg_sq"), so the reference's densification helpers that edit the optimizer in place
(`replace_tensor_to_optimizer`, `_prune_optimizer`, `cat_tensors_to_optimizer`, gaussian_model.py:866-960) and its checkpoints
(`optimizer.state_dict()` inside chkpnt*.pth) keep working.  `step()` updates every parameter tensor of every group with ONE
kernel launch (csrc/mrgs_optim.hip) instead of 6-8 elementwise passes per tensor; CPU parameters raise (no fallback)."""
import ctypes

import torch

from . import _lib


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if weight_decay != 0 or amsgrad:
            raise NotImplementedError("the reference trains with weight_decay=0, amsgrad=False; nothing else is built")
        if not 0.0 <= lr or not 0.0 <= eps or not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        by_key = {}
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise RuntimeError("materialrefgs_amd.optim.Adam updates device tensors only (libmrgs.so has no CPU path)")
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("Adam: parameters must be contiguous fp32 tensors")
                if p.grad.is_sparse:
                    raise RuntimeError("Adam does not support sparse gradients")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                step = int(st["step"]) + 1
                if torch.is_tensor(st["step"]):
                    st["step"].fill_(step)
                else:
                    st["step"] = step
                g = p.grad if (p.grad.dtype == torch.float32 and p.grad.is_contiguous()) else p.grad.float().contiguous()
                m, v = st["exp_avg"], st["exp_avg_sq"]
                if not (m.is_contiguous() and v.is_contiguous() and m.dtype == torch.float32 and v.dtype == torch.float32):
                    raise RuntimeError("Adam: optimizer state must be contiguous fp32")
                key = (p.device, float(beta1), float(beta2), float(group["eps"]))
                by_key.setdefault(key, []).append((p, g, m, v, float(group["lr"]), step))
        lib = _lib.lib()
        for (dev, beta1, beta2, eps), items in by_key.items():
            arr = (_lib.MrgsAdamTensor * len(items))()
            for i, (p, g, m, v, lr, step) in enumerate(items):
                arr[i] = _lib.MrgsAdamTensor(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, step)
            with _lib.guard(dev):
                stream = _lib.stream_ptr(dev)
                _lib.check(lib.mrgs_adam_step(arr, len(items), beta1, beta2, eps, stream))
            del items          # the gradient copies (if any) stay alive until the launch is queued on the same stream
        return loss
