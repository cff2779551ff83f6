"""Adam step: one-launch HIP kernel vs torch.optim.Adam -- the optimizer the reference itself uses
(scene/gaussian_model.py:445), available on both machines, so the comparison is against the reference's own arithmetic."""
import copy
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _groups(dev, seed=0):
    g = torch.Generator().manual_seed(seed)
    shapes = {"xyz": (5000, 3), "f_dc": (5000, 1, 3), "f_rest": (5000, 15, 3), "opacity": (5000, 1), "scaling": (5000, 2),
              "rotation": (5000, 4), "env": (6, 16, 16, 3), "odd": (4099,), "tiny": (1,), "big": (300001, 5), "empty": (0, 3)}
    lrs = {"xyz": 1.6e-4, "f_dc": 2.5e-3, "f_rest": 2.5e-3 / 20, "opacity": 0.05, "scaling": 5e-3, "rotation": 1e-3, "env": 0.01,
           "odd": 1e-2, "tiny": 1e-1, "big": 1e-3, "empty": 1e-3}
    params = {k: torch.randn(*s, generator=g).to(dev) for k, s in shapes.items()}
    return params, lrs


def _make(opt_cls, params, lrs):
    ps = {k: torch.nn.Parameter(v.clone()) for k, v in params.items()}
    groups = [{"params": [ps[k]], "lr": lrs[k], "name": k} for k in ps]
    return ps, opt_cls(groups, lr=0.0, eps=1e-15)


def test_adam_rejects_cpu_parameters_and_unsupported_modes():
    from materialrefgs_amd.optim import Adam
    p = torch.nn.Parameter(torch.zeros(4))
    opt = Adam([p], lr=1e-3)
    p.grad = torch.ones(4)
    with pytest.raises(RuntimeError):
        opt.step()
    with pytest.raises(NotImplementedError):
        Adam([p], amsgrad=True)
    with pytest.raises(ValueError):
        Adam([p], lr=-1.0)


@pytest.mark.gpu
def test_adam_matches_torch_adam_over_steps():
    from materialrefgs_amd.optim import Adam
    dev = "cuda"
    params, lrs = _groups(dev)
    pa, oa = _make(Adam, params, lrs)
    pb, ob = _make(torch.optim.Adam, params, lrs)
    g = torch.Generator().manual_seed(5)
    for it in range(12):
        for k in pa:
            if k == "odd" and it % 3 == 1:                      # a parameter without a gradient this step is skipped (own step count)
                pa[k].grad = pb[k].grad = None
                continue
            scale = 10.0 ** ((it % 4) - 3)
            gr = (torch.randn(*pa[k].shape, generator=g) * scale).to(dev)
            if k == "opacity" and it == 4:
                gr.zero_()                                     # all-zero gradient: the moments decay, the step is m / (sqrt(v) + 1e-15)
            pa[k].grad, pb[k].grad = gr.clone(), gr.clone()
        if it == 6:                                            # learning-rate schedule edits param_groups in place (gaussian_model.py:455-460)
            for grp_a, grp_b in zip(oa.param_groups, ob.param_groups):
                if grp_a["name"] == "xyz":
                    grp_a["lr"] = grp_b["lr"] = 3.3e-5
        oa.step()
        ob.step()
    for k in pa:
        sa, sb = oa.state[pa[k]], ob.state[pb[k]]
        if pa[k].numel() == 0:
            continue
        assert int(sa["step"]) == int(sb["step"]), k
        # the moments are sums of gradients of very different sizes: tolerance relative to the largest entry
        torch.testing.assert_close(sa["exp_avg"], sb["exp_avg"], rtol=2e-6, atol=5e-7 * float(sb["exp_avg"].abs().max()), msg=lambda m, k=k: f"{k}: {m}")
        torch.testing.assert_close(sa["exp_avg_sq"], sb["exp_avg_sq"], rtol=2e-6, atol=5e-7 * float(sb["exp_avg_sq"].abs().max()), msg=lambda m, k=k: f"{k}: {m}")
        torch.testing.assert_close(pa[k].data, pb[k].data, rtol=1e-5, atol=2e-6 * lrs[k] * 12, msg=lambda m, k=k: f"{k}: {m}")


@pytest.mark.gpu
def test_adam_state_is_interchangeable_with_torch_adam():
    """state_dict round trip in both directions + the reference's in-place state surgery (replace_tensor_to_optimizer,
    scene/gaussian_model.py:866-877) on our optimizer."""
    from materialrefgs_amd.optim import Adam
    dev = "cuda"
    params, lrs = _groups(dev, 1)
    pa, oa = _make(Adam, params, lrs)
    pb, ob = _make(torch.optim.Adam, params, lrs)
    for it in range(3):
        for k in pa:
            gr = torch.randn_like(pa[k]) * 0.01
            pa[k].grad, pb[k].grad = gr.clone(), gr.clone()
        oa.step(); ob.step()
    # ours -> torch and torch -> ours
    pc, oc = _make(torch.optim.Adam, {k: v.data for k, v in pa.items()}, lrs)
    oc.load_state_dict(copy.deepcopy(oa.state_dict()))        # load_state_dict aliases same-device tensors: copy first
    pd, od = _make(Adam, {k: v.data for k, v in pb.items()}, lrs)
    od.load_state_dict(copy.deepcopy(ob.state_dict()))
    for k in pa:
        gr = torch.randn_like(pa[k]) * 0.01
        for p_ in (pa, pb, pc, pd):
            p_[k].grad = gr.clone()
    for o in (oa, ob, oc, od):
        o.step()
    for k in pa:
        if pa[k].numel() == 0:
            continue
        torch.testing.assert_close(pc[k].data, pa[k].data, rtol=1e-5, atol=1e-7, msg=lambda m, k=k: f"{k}: {m}")
        torch.testing.assert_close(pd[k].data, pb[k].data, rtol=1e-5, atol=1e-7, msg=lambda m, k=k: f"{k}: {m}")
    # reset the moments of one group in place the way reset_opacity does
    grp = [g_ for g_ in oa.param_groups if g_["name"] == "opacity"][0]
    old = grp["params"][0]
    stored = oa.state.get(old)
    new_val = torch.full_like(old, -2.0)
    stored["exp_avg"] = torch.zeros_like(new_val)
    stored["exp_avg_sq"] = torch.zeros_like(new_val)
    del oa.state[old]
    grp["params"][0] = torch.nn.Parameter(new_val.requires_grad_(True))
    oa.state[grp["params"][0]] = stored
    grp["params"][0].grad = torch.ones_like(new_val)
    oa.step()
    assert int(oa.state[grp["params"][0]]["step"]) == 5
    assert bool((grp["params"][0].data < -2.0).all())


@pytest.mark.gpu
def test_adam_full_size_properties():
    """C3 parameter set (P = 300 000, 275 floats per gaussian): sign/size identities of the first step, no oracle needed."""
    from materialrefgs_amd.optim import Adam
    dev = "cuda"
    P = 300000
    shapes = [(P, 3), (P, 1, 3), (P, 15, 3), (P, 1), (P, 2), (P, 4), (P, 1), (P, 3), (P, 3), (P, 1), (P, 1), (P, 1, 3), (P, 15, 3), (P, 32, 5)]
    ps = [torch.nn.Parameter(torch.zeros(*s, device=dev)) for s in shapes]
    opt = Adam([{"params": [p], "lr": 1e-3 * (i + 1)} for i, p in enumerate(ps)], lr=0.0, eps=1e-15)
    for p in ps:
        p.grad = torch.randn_like(p)
    opt.step()
    for i, p in enumerate(ps):                                  # first Adam step = -lr * sign(g) (bias-corrected m / sqrt(v) = g / |g|)
        lr = 1e-3 * (i + 1)
        torch.testing.assert_close(p.data, -lr * torch.sign(p.grad), rtol=1e-5, atol=1e-9)


def _reference_prune(optimizer, mask):
    """_prune_optimizer as the reference writes it (scene/gaussian_model.py:856-874), torch boolean indexing."""
    out = {}
    for group in optimizer.param_groups:
        if group["name"] in ("mlp", "env", "env2"):
            continue
        stored = optimizer.state.get(group["params"][0], None)
        if stored is not None:
            stored["exp_avg"] = stored["exp_avg"][mask]
            stored["exp_avg_sq"] = stored["exp_avg_sq"][mask]
            del optimizer.state[group["params"][0]]
            group["params"][0] = torch.nn.Parameter(group["params"][0][mask].requires_grad_(True))
            optimizer.state[group["params"][0]] = stored
        else:
            group["params"][0] = torch.nn.Parameter(group["params"][0][mask].requires_grad_(True))
        out[group["name"]] = group["params"][0]
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("P,frac", [(5000, 0.7), (1, 1.0), (1025, 0.0), (300001, 0.93)])
def test_prune_matches_boolean_indexing(P, frac):
    from materialrefgs_amd import densify
    from materialrefgs_amd.optim import Adam
    dev = "cuda"
    g = torch.Generator().manual_seed(P)
    shapes = {"xyz": (P, 3), "f_dc": (P, 1, 3), "f_rest": (P, 15, 3), "opacity": (P, 1), "scaling": (P, 2), "rotation": (P, 4),
              "ind_asg": (P, 32, 5), "nostate": (P, 7), "env": (6, 8, 8, 3)}
    vals = {k: torch.randn(*s, generator=g).to(dev) for k, s in shapes.items()}

    def make():
        ps = {k: torch.nn.Parameter(v.clone()) for k, v in vals.items()}
        opt = Adam([{"params": [ps[k]], "lr": 1e-3, "name": k} for k in ps], lr=0.0, eps=1e-15)
        for k, p in ps.items():
            if k != "nostate":                                    # a group that never received a gradient has no state yet
                p.grad = torch.randn(*p.shape, generator=torch.Generator().manual_seed(3)).to(dev)
        opt.step()
        return ps, opt
    pa, oa = make()
    pb, ob = make()
    keep = (torch.rand(P, generator=g) < frac).to(dev)
    accum, radii = torch.randn(P, 1, generator=g).to(dev), torch.randint(0, 50, (P,), generator=g, dtype=torch.int32).to(dev)
    got, (accum2, radii2) = densify.prune_optimizer(oa, keep, extra=[accum, radii])
    want = _reference_prune(ob, keep)
    assert set(got) == set(want) and "env" not in got
    for k in want:
        assert got[k].shape == want[k].shape and got[k].requires_grad
        assert torch.equal(got[k].data, want[k].data), k
        sa, sb = oa.state.get(got[k], None), ob.state.get(want[k], None)
        assert (sa is None) == (sb is None), k
        if sb is not None:
            assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"]), k
            assert int(sa["step"]) == int(sb["step"])
    assert torch.equal(accum2, accum[keep]) and torch.equal(radii2, radii[keep]) and radii2.dtype == torch.int32
    # the optimizer keeps working on the compacted tensors; appended rows get zero moments (cat_tensors_to_optimizer)
    n_new = 17
    add = {k: torch.randn(n_new, *shapes[k][1:], generator=g).to(dev) for k in shapes if k != "env"}
    grown = densify.cat_tensors_to_optimizer(oa, add)
    m = int(keep.sum())
    for k, p in grown.items():
        assert p.shape[0] == m + n_new
        st = oa.state.get(p, None)
        if st is not None:
            assert float(st["exp_avg"][m:].abs().sum()) == 0.0
        p.grad = torch.ones_like(p)
    oa.step()
    rep = densify.replace_tensor_to_optimizer(oa, torch.full((m + n_new, 1), -3.0, device=dev), "opacity")
    assert float(oa.state[rep["opacity"]]["exp_avg"].abs().sum()) == 0.0
