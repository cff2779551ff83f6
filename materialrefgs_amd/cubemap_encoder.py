"""Drop-in for the reference's `cubemapencoder` Python module (submodules/cubemapencoder/cubemapencoder/cubemap_encoder.py):
`cubemap_encode` (autograd function, :17-64), `CubemapEncoder` (:79-110), `MipCubemapEncoder` (:113-171) -- same constructor
arguments, parameter names and shapes ([6,C,L,L] textures, [C] fail value), output layout [C,B] of the raw function and [B,3] /
[B,levels*C] of the modules.  The native side is libmrgs.so (mrgs_cubemap_encode_forward / _backward, include/mrgs.h); CPU tensors
are rejected -- there is no fallback.
"""
import ctypes

import numpy as np
import torch
import torch.nn as nn

from . import _lib

def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t.numel() else None


class _cubemap_encode(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inputs, embeddings, fail_value, interpolation, enable_seamless):
        if not embeddings.is_cuda:
            raise RuntimeError("cubemap_encode needs CUDA(HIP) tensors: the fetch runs in libmrgs.so, there is no CPU path")
        inputs = inputs.detach().float().contiguous()
        embeddings = embeddings.detach().float().contiguous()
        fail_value = fail_value.detach().float().contiguous()
        C, L, B = embeddings.shape[1], embeddings.shape[2], inputs.shape[0]
        outputs = torch.empty([C, B], dtype=torch.float32, device=embeddings.device)
        with _lib.guard(embeddings.device):
            st = _lib.stream_ptr(embeddings.device)
            _lib.check(_lib.lib().mrgs_cubemap_encode_forward(_p(inputs), _p(embeddings), _p(fail_value), _p(outputs), int(interpolation),
                                                              int(enable_seamless), B, C, L, st))
        ctx.save_for_backward(inputs, embeddings)
        ctx.params = (int(interpolation), int(enable_seamless))
        return outputs

    @staticmethod
    def backward(ctx, grad_outputs):
        inputs, embeddings = ctx.saved_tensors
        grad_outputs = grad_outputs.float().contiguous()
        C, L, B = embeddings.shape[1], embeddings.shape[2], inputs.shape[0]
        grad_embeddings = torch.zeros_like(embeddings)
        grad_inputs = torch.empty_like(inputs)
        grad_fail = torch.zeros([C], dtype=embeddings.dtype, device=embeddings.device)
        with _lib.guard(embeddings.device):
            st = _lib.stream_ptr(embeddings.device)
            _lib.check(_lib.lib().mrgs_cubemap_encode_backward(_p(grad_outputs), _p(inputs), _p(embeddings), _p(grad_embeddings), _p(grad_inputs),
                                                               _p(grad_fail), ctx.params[0], ctx.params[1], B, C, L, st))
        return grad_inputs, grad_embeddings, grad_fail, None, None


cubemap_encode = _cubemap_encode.apply


def _fetch(dirs, texture, fail_value, nearest, seamless):
    """[B,3] directions -> [B,C] texels of one [6,C,L,L] cubemap (the raw function returns channel-major [C,B])."""
    return cubemap_encode(dirs, texture, fail_value, 0 if nearest else 1, 1 if seamless else 0).t()


class _CubeModule(nn.Module):
    """What the two modules share: the sampling switches.  `interpolation` is "linear" or "nearest"; `seamless` (on, as in the
    reference) lets bilinear taps cross cube edges.  The contractual surface towards the reference's callers and checkpoints is the
    constructor signature and the parameter names (`params.Cubemap_texture`, `params.Cubemap_failv`; `params_list.<i>`, `fail_value`:
    they are the state_dict keys of scene/gaussian_model.py's env_map) -- everything else here is this module's own."""

    def __init__(self, interpolation):
        super().__init__()
        if interpolation not in ("linear", "nearest"):
            raise KeyError(interpolation)
        self.interpolation = interpolation
        self.seamless = True

    @property
    def n_elems(self):
        return sum(p.numel() for p in self.parameters())

    def extra_repr(self):
        return f"{self.interpolation}, seamless={bool(self.seamless)}, {self.n_elems} parameters"


class CubemapEncoder(_CubeModule):
    """One [6, output_dim, resolution, resolution] cubemap; forward(dirs[..., 3]) -> sigmoid of the fetched texel, [..., 3]."""

    def __init__(self, output_dim=6, resolution=256, interpolation='linear'):
        super().__init__(interpolation)
        self.output_dim, self.resolution = int(output_dim), int(resolution)
        texels = torch.empty(6, self.output_dim, self.resolution, self.resolution).uniform_(-5.0, 5.0)     # pre-sigmoid: U(-5, 5)
        self.params = nn.ParameterDict({'Cubemap_texture': nn.Parameter(texels), 'Cubemap_failv': nn.Parameter(torch.zeros(self.output_dim))})

    def extra_repr(self):
        return f"{self.output_dim} x {self.resolution}^2 x 6, " + super().extra_repr()

    def forward(self, inputs):
        flat = inputs.reshape(-1, 3)
        texel = _fetch(flat, self.params['Cubemap_texture'], self.params['Cubemap_failv'], self.interpolation == "nearest", self.seamless)
        return torch.sigmoid(texel).reshape(*inputs.shape[:-1], 3)      # three channels, as the reference's reshape fixes them (:109)


def mip_resolutions(num_levels, base_resolution, per_level_scale):
    """Side lengths of the levels: ceil(base * scale^i), i = 0 .. num_levels-1 (default 4, 16, 64, 256)."""
    return [int(np.ceil(float(base_resolution) * float(per_level_scale) ** i)) for i in range(num_levels)]


class MipCubemapEncoder(_CubeModule):
    """A pyramid of cubemaps fetched along the same direction; the per-level features are concatenated ([B, levels * level_dim]) or
    summed ([B, level_dim])."""

    def __init__(self, num_levels=4, level_dim=6, per_level_scale=4, base_resolution=4, interpolation='linear', concat=True):
        super().__init__(interpolation)
        self.num_levels, self.level_dim, self.concat = int(num_levels), int(level_dim), bool(concat)
        self.per_level_scale, self.base_resolution = per_level_scale, base_resolution
        self.output_dim = self.num_levels * self.level_dim if self.concat else self.level_dim
        self.params_list = nn.ParameterList([nn.Parameter(torch.empty(6, self.level_dim, r, r))
                                             for r in mip_resolutions(self.num_levels, base_resolution, per_level_scale)])
        self.fail_value = nn.Parameter(torch.zeros(self.level_dim))
        self.reset_parameters()

    def reset_parameters(self, amplitude=1e-4):
        for level in self.params_list:
            nn.init.uniform_(level, -amplitude, amplitude)

    def extra_repr(self):
        return f"levels {[int(p.shape[-1]) for p in self.params_list]} x {self.level_dim}, concat={self.concat}, " + super().extra_repr()

    def forward(self, inputs):
        nearest = self.interpolation == "nearest"
        per_level = [_fetch(inputs, level, self.fail_value, nearest, self.seamless) for level in self.params_list]
        return torch.cat(per_level, dim=1) if self.concat else torch.stack(per_level).sum(0)
