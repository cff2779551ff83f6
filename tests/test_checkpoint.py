"""chkpnt*.pth: the 22-tuple of GaussianModel.capture() (scene/gaussian_model.py:124-148) and its round trip through
materialrefgs_amd.checkpoint, incl. interoperability of the optimizer state with torch.optim.Adam built the reference's way."""
import os
import tempfile
from types import SimpleNamespace

import torch
import torch.nn as nn

from materialrefgs_amd import checkpoint as ck


def _model(P=37, seed=0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: nn.Parameter(torch.randn(*s, generator=g))
    m = SimpleNamespace(active_sh_degree=2, _xyz=r(P, 3), _refl_strength=r(P, 1), _metalness=r(P, 1), _roughness=r(P, 1), _ori_color=r(P, 3),
                        _diffuse_color=r(P, 3), _features_dc=r(P, 1, 3), _features_rest=r(P, 15, 3), _indirect_dc=r(P, 1, 3),
                        _indirect_rest=r(P, 15, 3), _indirect_asg=r(P, 32, 5), _scaling=r(P, 2), _rotation=r(P, 4), _opacity=r(P, 1),
                        _normal1=r(P, 3), _normal2=r(P, 3), max_radii2D=torch.rand(P, generator=g), spatial_lr_scale=1.7,
                        env_map=nn.ParameterList([r(6, 4, 4, 3)]), env_map_2=nn.ParameterList([r(6, 4, 4, 3)]))
    return m


def test_tuple_layout_is_the_reference_order():
    m = _model()
    args = ck.default_training_args()
    ck.training_setup(m, args, optimizer_cls=torch.optim.Adam)
    t = ck.capture(m)
    assert len(t) == 22
    assert t[0] == 2 and t[1] is m._xyz and t[2] is m._refl_strength and t[3] is m._metalness and t[4] is m._roughness
    assert t[5] is m._ori_color and t[6] is m._diffuse_color and t[7] is m._features_dc and t[8] is m._features_rest
    assert t[9] is m._indirect_dc and t[10] is m._indirect_rest and t[11] is m._indirect_asg and t[12] is m._scaling
    assert t[13] is m._rotation and t[14] is m._opacity and t[15] is m._normal1 and t[16] is m._normal2 and t[17] is m.max_radii2D
    assert t[18] is m.xyz_gradient_accum and t[19] is m.denom and isinstance(t[20], dict) and t[21] == 1.7
    names = [g["name"] for g in m.optimizer.param_groups]
    # both lists as the reference's own GaussianModel produced them (tests/golden/gen_reference_render_vectors.py: capture(), training_setup())
    import numpy as np
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_render.npz"))
    assert names == [str(x) for x in gold["G_optimizer_groups"]]                                          # training_setup :422-446
    assert list(ck.CAPTURE_FIELDS) + ["optimizer.state_dict", "spatial_lr_scale"] == [str(x) for x in gold["G_capture_fields"]]
    lr = {g["name"]: g["lr"] for g in m.optimizer.param_groups}
    assert abs(lr["xyz"] - 0.00016 * 1.7) < 1e-12 and abs(lr["f_rest"] - 0.0075 / 20) < 1e-12 and lr["env"] == 0.01
    assert m.optimizer.defaults["eps"] == 1e-15 and not m._normal1.requires_grad


def test_checkpoint_round_trip_and_optimizer_interop():
    from materialrefgs_amd.optim import Adam
    m = _model(seed=1)
    args = ck.default_training_args()
    ck.training_setup(m, args, optimizer_cls=torch.optim.Adam)       # "the reference": torch's Adam, two steps of state
    for _ in range(2):
        for g in m.optimizer.param_groups:
            for p in g["params"]:
                if p.requires_grad:
                    p.grad = torch.randn_like(p)
        m.optimizer.step()
    m.xyz_gradient_accum += 0.5
    m.denom += 2
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "chkpnt7000.pth")
        ck.save_checkpoint(path, m, 7000)
        saved, it = torch.load(path, weights_only=False)
        assert it == 7000 and len(saved) == 22
        m2 = _model(seed=9)                                           # different values everywhere
        first = ck.load_checkpoint(path, m2, args, optimizer_cls=Adam)   # restored into THIS library's Adam
    assert first == 7000
    for f in ck.CAPTURE_FIELDS:
        if f == "_indirect_asg":
            assert float(m2._indirect_asg.abs().max()) == 0 and tuple(m2._indirect_asg.shape) == (37, 32, 5)   # restore() re-creates it (:173)
            continue
        a, b = getattr(m2, f), getattr(m, f)
        assert (a == b) if not torch.is_tensor(a) else torch.equal(a, b), f
    s1, s2 = m.optimizer.state_dict(), m2.optimizer.state_dict()
    assert [g["name"] for g in s2["param_groups"]] == [g["name"] for g in s1["param_groups"]]
    for k, st in s1["state"].items():
        for kk in ("step", "exp_avg", "exp_avg_sq"):
            assert torch.equal(torch.as_tensor(st[kk]).float(), torch.as_tensor(s2["state"][k][kk]).float()), (k, kk)
