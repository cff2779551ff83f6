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

_interp_to_id = {"nearest": 0, "linear": 1}


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
        with torch.cuda.device(embeddings.device):
            st = ctypes.c_void_p(torch.cuda.current_stream(embeddings.device).cuda_stream)
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
        with torch.cuda.device(embeddings.device):
            st = ctypes.c_void_p(torch.cuda.current_stream(embeddings.device).cuda_stream)
            _lib.check(_lib.lib().mrgs_cubemap_encode_backward(_p(grad_outputs), _p(inputs), _p(embeddings), _p(grad_embeddings), _p(grad_inputs),
                                                               _p(grad_fail), ctx.params[0], ctx.params[1], B, C, L, st))
        return grad_inputs, grad_embeddings, grad_fail, None, None


cubemap_encode = _cubemap_encode.apply


class CubemapEncoder(nn.Module):
    def __init__(self, output_dim=6, resolution=256, interpolation='linear'):
        super().__init__()
        self.input_dim = 3
        self.resolution = resolution
        self.output_dim = output_dim
        self.interpolation = interpolation
        self.interp_id = _interp_to_id[interpolation]
        self.seamless = 1
        self.params = nn.ParameterDict({
            'Cubemap_texture': nn.Parameter(torch.rand(6, self.output_dim, resolution, resolution) * 10 - 5),
            'Cubemap_failv': nn.Parameter(torch.zeros(self.output_dim)),
        })
        self.n_elems = 6 * self.output_dim * resolution * resolution + self.output_dim

    def __repr__(self):
        return (f"CubemapEncoder: input_dim={self.input_dim} output_dim={self.output_dim} resolution={self.resolution} -> {self.n_elems} "
                f"interpolation={self.interpolation} seamless={self.seamless}")

    def forward(self, inputs):
        pre_shape = inputs.shape[:-1]
        outputs = cubemap_encode(inputs.reshape(-1, 3), self.params['Cubemap_texture'], self.params['Cubemap_failv'], self.interp_id,
                                 self.seamless).permute(1, 0)
        return torch.sigmoid(outputs).reshape(*pre_shape, 3)      # (the reference hard-codes 3 channels here, :109)


class MipCubemapEncoder(nn.Module):
    def __init__(self, num_levels=4, level_dim=6, per_level_scale=4, base_resolution=4, interpolation='linear', concat=True):
        super().__init__()
        self.input_dim = 3
        self.num_levels = num_levels
        self.level_dim = level_dim
        self.per_level_scale = per_level_scale
        self.base_resolution = base_resolution
        self.concat = concat
        self.output_dim = num_levels * level_dim if concat else level_dim
        self.interpolation = interpolation
        self.interp_id = _interp_to_id[interpolation]
        self.seamless = 1
        params_list, L, n_elems = [], float(base_resolution), 0
        for _ in range(num_levels):
            iL = int(np.ceil(L))
            params_list.append(nn.Parameter(torch.empty(6, self.level_dim, iL, iL)))
            n_elems += 6 * self.level_dim * iL * iL
            L = L * per_level_scale
        self.params_list = nn.ParameterList(params_list)
        self.fail_value = nn.Parameter(torch.zeros(self.level_dim))
        self.n_elems = n_elems + self.level_dim
        self.reset_parameters()

    def reset_parameters(self):
        std = 1e-4
        for ii in range(self.num_levels):
            self.params_list[ii].data.uniform_(-std, std)

    def __repr__(self):
        return (f"MipCubemapEncoder: input_dim={self.input_dim} num_levels={self.num_levels} level_dim={self.level_dim} "
                f"base_resolution={self.base_resolution} -> {self.n_elems} per_level_scale={self.per_level_scale:.4f} "
                f"interpolation={self.interpolation} seamless={self.seamless}")

    def forward(self, inputs):
        outputs = [cubemap_encode(inputs, self.params_list[ii], self.fail_value, self.interp_id, self.seamless) for ii in range(self.num_levels)]
        outputs = torch.cat(outputs, dim=0) if self.concat else sum(outputs)
        return outputs.permute(1, 0)      # CxN -> NxC
