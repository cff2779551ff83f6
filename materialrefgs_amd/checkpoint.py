"""Training checkpoints in the reference's format: `GaussianModel.capture()` / `restore()`
(scene/gaussian_model.py:124-177) and the `chkpnt<iteration>.pth` files train_refnerf.py:1482-1484 writes with
`torch.save((gaussians.capture(), iteration), path)` and reads back at :1037-1040.

The tuple has 22 entries in a fixed order; entry 20 is `optimizer.state_dict()` of the Adam built by `training_setup` (:417-453),
whose parameter groups -- names, order, learning rates -- are reproduced by `param_groups` so that a state dict written by the
reference loads into `materialrefgs_amd.optim.Adam` (same state keys) and vice versa.
"""
from types import SimpleNamespace

import torch
import torch.nn as nn

# order of GaussianModel.capture() (scene/gaussian_model.py:124-148); entries that are model attributes
CAPTURE_FIELDS = ("active_sh_degree", "_xyz", "_refl_strength", "_metalness", "_roughness", "_ori_color", "_diffuse_color", "_features_dc",
                  "_features_rest", "_indirect_dc", "_indirect_rest", "_indirect_asg", "_scaling", "_rotation", "_opacity", "_normal1", "_normal2",
                  "max_radii2D", "xyz_gradient_accum", "denom")          # then optimizer.state_dict(), spatial_lr_scale


def param_groups(model, training_args):
    """The parameter groups of training_setup (:422-446), in its order (the optimizer state dict indexes parameters by position)."""
    a, s = training_args, getattr(model, "spatial_lr_scale", 1.0)
    env = lambda m: list(m.parameters()) if m is not None else []
    groups = [
        {"params": [model._xyz], "lr": a.position_lr_init * s, "name": "xyz"},
        {"params": [model._features_dc], "lr": a.features_lr, "name": "f_dc"},
        {"params": [model._features_rest], "lr": a.features_lr / 20.0, "name": "f_rest"},
        {"params": [model._opacity], "lr": a.opacity_lr, "name": "opacity"},
        {"params": [model._scaling], "lr": a.scaling_lr, "name": "scaling"},
        {"params": [model._rotation], "lr": a.rotation_lr, "name": "rotation"},
        {"params": env(getattr(model, "env_map", None)), "lr": a.envmap_cubemap_lr, "name": "env"},
        {"params": env(getattr(model, "env_map_2", None)), "lr": a.envmap_cubemap_lr, "name": "env2"},
        {"params": [model._refl_strength], "lr": a.refl_strength_lr, "name": "refl_strength"},
        {"params": [model._ori_color], "lr": a.ori_color_lr, "name": "ori_color"},
        {"params": [model._diffuse_color], "lr": a.ori_color_lr, "name": "diffuse_color"},
        {"params": [model._roughness], "lr": a.roughness_lr, "name": "roughness"},
        {"params": [model._metalness], "lr": a.metalness_lr, "name": "metalness"},
        {"params": [model._normal1], "lr": a.normal_lr, "name": "normal1"},
        {"params": [model._normal2], "lr": a.normal_lr, "name": "normal2"},
        {"params": [model._indirect_dc], "lr": a.indirect_lr, "name": "ind_dc"},
        {"params": [model._indirect_rest], "lr": a.indirect_lr / 20.0, "name": "ind_rest"},
        {"params": [model._indirect_asg], "lr": a.asg_lr, "name": "ind_asg"},
    ]
    return groups


def training_setup(model, training_args, optimizer_cls=None):
    """training_setup (:417-453) without the learning-rate scheduler: statistics buffers + the Adam (lr = 0, eps = 1e-15)."""
    if optimizer_cls is None:
        from .optim import Adam as optimizer_cls
    dev = model._xyz.device
    P = model._xyz.shape[0]
    model.percent_dense = getattr(training_args, "percent_dense", 0.01)
    model.xyz_gradient_accum = torch.zeros((P, 1), device=dev)
    model.denom = torch.zeros((P, 1), device=dev)
    model._normal1.requires_grad_(False)
    model._normal2.requires_grad_(False)
    model.optimizer = optimizer_cls(param_groups(model, training_args), lr=0.0, eps=1e-15)
    return model.optimizer


def capture(model):
    """GaussianModel.capture(): the 22-tuple."""
    return tuple(getattr(model, f) for f in CAPTURE_FIELDS) + (model.optimizer.state_dict(), model.spatial_lr_scale)


def restore(model, model_args, training_args, optimizer_cls=None):
    """GaussianModel.restore() (:150-177): assigns the tuple, re-creates `_indirect_asg` as zeros [P,32,5] exactly as the reference
    does (:173), rebuilds the optimizer and loads its state."""
    if len(model_args) != len(CAPTURE_FIELDS) + 2:
        raise ValueError(f"checkpoint tuple has {len(model_args)} entries, expected {len(CAPTURE_FIELDS) + 2}")
    *fields, opt_dict, spatial_lr_scale = model_args
    xyz_gradient_accum, denom = fields[18], fields[19]
    for name, value in zip(CAPTURE_FIELDS[:18], fields[:18]):
        setattr(model, name, value)
    model.spatial_lr_scale = spatial_lr_scale
    model._indirect_asg = nn.Parameter(torch.zeros(model._rotation.shape[0], 32, 5, device=model._rotation.device).requires_grad_(True))
    training_setup(model, training_args, optimizer_cls)
    model.xyz_gradient_accum = xyz_gradient_accum
    model.denom = denom
    model.optimizer.load_state_dict(opt_dict)
    return model


def save_checkpoint(path, model, iteration):
    """train_refnerf.py:1482-1484."""
    torch.save((capture(model), iteration), path)


def load_checkpoint(path, model, training_args, optimizer_cls=None, map_location=None):
    """train_refnerf.py:1037-1040: returns the iteration the checkpoint was written at."""
    model_params, first_iter = torch.load(path, map_location=map_location, weights_only=False)
    restore(model, model_params, training_args, optimizer_cls)
    return first_iter


def default_training_args():
    """The learning rates of arguments/__init__.py (OptimizationParams) that training_setup reads."""
    return SimpleNamespace(position_lr_init=0.00016, position_lr_final=0.0000016, position_lr_delay_mult=0.01, position_lr_max_steps=30000,
                           features_lr=0.0075, indirect_lr=0.0075, asg_lr=0.0075, opacity_lr=0.05, scaling_lr=0.005, rotation_lr=0.001,
                           ori_color_lr=0.0075, refl_strength_lr=0.005, roughness_lr=0.005, metalness_lr=0.01, normal_lr=0.006,
                           envmap_cubemap_lr=0.01, percent_dense=0.01)      # arguments/__init__.py:108-133
