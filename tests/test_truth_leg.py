"""Truth leg of the gradient parity.

The parity tests compare the HIP kernels with `oracle/mrgs_oracle.c` in its default mode, whose blend arithmetic carries the kernels'
FMA pattern.  This file measures all of them against the SAME formulas evaluated in float64 (`variant="f64"`, the reference's
expression trees as written) and against the literal un-fused fp32 reading (`variant="lit32"`):

    err(HIP vs f64)  <=  max(1.5 x err(literal fp32 vs f64), floor)        per gradient tensor

i.e. the kernels are no further from the true value of the reference's formulas than a literal fp32 build of the reference is.
The ray/splat intersection cancels catastrophically for grazing surfels, so the fp32 readings themselves sit ~1e-3 (max-norm) from the
float64 value on the geometry gradients; what is asserted is the ORDER, what is printed is the whole table."""
import json

import numpy as np
import pytest
import torch

from materialrefgs_amd.synthetic import make_shell_scene, orbit_camera, upstream_grads
from oracle import compare
from oracle import raster_oracle as ro

FLOOR = 2e-5     # below this every leg is at fp32 rounding of the sums; ordering is noise


def _legs(P, S, H, W, seed, radius_px, view=2):
    scene = make_shell_scene(P, S=S, seed=seed, radius_px=radius_px, image_size=max(H, W))
    cam = orbit_camera(view, H, W)
    g = upstream_grads(S, H, W)
    out = {}
    for v in ("fused", "lit32", "f64"):
        r = ro.render_scene(scene, cam, variant=v)
        out[v] = r.backward(*g)
        out[v + "_R"] = r.R
        r.close()
    return scene, cam, g, out


def test_fused_pattern_is_not_further_from_float64_than_the_literal_reading():
    scene, cam, g, legs = _legs(20000, 8, 256, 256, seed=5, radius_px=6.0)
    assert legs["fused_R"] == legs["lit32_R"] == legs["f64_R"]          # same discrete state in all three modes on this scene
    table = compare.three_way(None, legs["fused"], legs["lit32"], legs["f64"])
    print(json.dumps(table, indent=1))
    for k, row in table.items():
        assert row["fused32"]["max_norm"] <= max(2.0 * row["literal32"]["max_norm"], FLOOR), (k, row)
    # the exact-derivative tensors (colour / feature / opacity chain) are well conditioned: both readings sit at the level of one
    # (pixel, surfel) pair falling on the other side of the alpha >= 1/255 test in float64
    for k in ("sh", "features", "opacity"):
        assert table[k]["fused32"]["max_norm"] < 1e-4 and table[k]["literal32"]["max_norm"] < 1e-4


@pytest.mark.parametrize("scale", [10.0, 50.0])
def test_contracted_recurrence_under_a_depth_dominated_gradient_far_from_the_camera(scale):
    """ADVICE (round 3): the backward's ONE contracted recurrence (q - A) subtracts two sums that each carry
    depth x dL/ddepth where the reference sums per-channel differences of the order of the gap between two surfels.  The case that
    would show it: the scene 10x / 50x as far from the camera (depths 40 / 200) and an upstream gradient that is all depth.  The fused
    pattern (the kernels' arithmetic, carried by the oracle's default mode) stays within twice the literal fp32 reading's distance
    from float64, or under half the parity bar where the literal reading is closer than that (measured at 50x: `sh` 1.8e-5 against
    3.5e-6, everything geometric closer than the literal reading; tools/depth_grad_check.py prints the table with the HIP leg)."""
    from materialrefgs_amd.synthetic import CAM_DISTANCE, FOV, look_at_camera
    H = W = 192
    scene = make_shell_scene(8000, S=0, seed=3, radius_px=7.0, image_size=H)
    scene = scene._replace(means3D=scene.means3D * scale, scales=scene.scales * scale)
    cam = look_at_camera(17.0, 30.0, CAM_DISTANCE * scale, FOV, H, W)
    g_others = torch.zeros((7, H, W))
    g_others[0], g_others[1] = 1.0, 0.01
    g = (torch.full((3, H, W), 0.01), torch.zeros((0, H, W)), g_others)
    legs = {}
    for v in ("fused", "lit32", "f64"):
        r = ro.render_scene(scene, cam, variant=v)
        legs[v] = r.backward(*g)
        r.close()
    table = compare.three_way(None, legs["fused"], legs["lit32"], legs["f64"])
    for k, row in table.items():
        assert row["fused32"]["max_norm"] <= max(2.0 * row["literal32"]["max_norm"], 5e-5), (k, row)


@pytest.mark.gpu
@pytest.mark.parametrize("P,S,H,W,seed,radius_px", [(20000, 8, 256, 256, 5, 6.0), (50000, 0, 400, 400, 0, 7.0)])
def test_hip_is_no_further_from_float64_than_the_literal_fp32_reading(P, S, H, W, seed, radius_px):
    from helpers import HipRender
    scene, cam, g, legs = _legs(P, S, H, W, seed, radius_px)
    hr = HipRender(scene, cam, torch.device("cuda:0"))
    assert hr.num_rendered == legs["fused_R"]
    gh = hr.backward(*g)
    table = compare.three_way(gh, legs["fused"], legs["lit32"], legs["f64"])
    print(json.dumps(table, indent=1))
    for k, row in table.items():
        assert row["hip"]["max_norm"] <= max(1.5 * row["literal32"]["max_norm"], FLOOR), (k, row)
        # per-element relative error above 1e-3 max|g|: the bulk of the elements, not only the largest one
        assert row["hip"]["elem_median"] <= max(2.0 * row["literal32"]["elem_median"], 1e-6), (k, row)
        assert row["hip"]["elem_p99"] <= max(2.0 * row["literal32"]["elem_p99"], 1e-4), (k, row)


# ---- lone tiny surfels: the truth-leg rule holds in distribution, not per draw ------------------------------------------------------
# Round 6's soak on a fresh seed (profiles/r6_v2_soak_raster_seed448141_2000.txt, case 1478) met a lone 1.5-pixel surfel whose `scales`
# gradient -- a 1e-3 residue of per-pixel terms that cancel -- was 5.3e-4 from float64 in the kernels and 5.2e-5 in the literal fp32
# reading: the x 1.5 rule compares with ONE draw of the literal's rounding.  Over a set of such surfels both readings spread over
# four decades with the same median; that is what is asserted here, together with the case itself.
LONE_CASE = dict(P=1, S=0, H=80, W=412, deg=3, rpx=1.5, view=5, seed=491724934)      # case 1478 of seed 448141 (tools/scratch/diag_case.py)


def _lone_scenes(n, rng_seed=5):
    rng = np.random.default_rng(rng_seed)
    for _ in range(n):
        H, W = int(rng.integers(40, 300)), int(rng.integers(40, 300))
        yield dict(P=1, S=0, H=H, W=W, deg=3, rpx=1.5, view=int(rng.integers(0, 8)), seed=int(rng.integers(1 << 30)))


def _lone_errors(case, hip_dev=None):
    """(err of the fused oracle, err of the literal reading, err of the kernels or None) of the `scales` gradient against float64;
    None when the surfel is not rendered or its gradient is zero."""
    scene = make_shell_scene(case["P"], S=case["S"], seed=case["seed"], radius_px=case["rpx"], image_size=max(case["H"], case["W"]))
    cam = orbit_camera(case["view"], case["H"], case["W"])
    g = upstream_grads(case["S"], case["H"], case["W"])
    legs = {}
    for v in ("fused", "lit32", "f64"):
        r = ro.render_scene(scene, cam, sh_degree=case["deg"], variant=v)
        if r.R == 0:
            r.close()
            return None
        legs[v] = r.backward(*g)["scales"]
        r.close()
    if float(np.abs(legs["f64"]).max()) == 0.0:
        return None
    rel = lambda a: float(np.abs(np.asarray(a, np.float64).reshape(legs["f64"].shape) - legs["f64"]).max() / np.abs(legs["f64"]).max())
    e_hip = None
    if hip_dev is not None:
        from helpers import HipRender
        hr = HipRender(scene, cam, hip_dev, sh_degree=case["deg"])
        e_hip = rel(hr.backward(*g)["scales"])
    return rel(legs["fused"]), rel(legs["lit32"]), e_hip


def test_lone_tiny_surfels_fused_and_literal_readings_spread_alike():
    errs = [e for e in (_lone_errors(c) for c in _lone_scenes(40)) if e is not None]
    assert len(errs) >= 20
    fused, lit = np.array([e[0] for e in errs]), np.array([e[1] for e in errs])
    print(f"lone 1.5-pixel surfels, scales gradient vs float64 over {len(errs)} scenes: literal fp32 min {lit.min():.1e} median {np.median(lit):.1e} "
          f"max {lit.max():.1e}; fused min {fused.min():.1e} median {np.median(fused):.1e} max {fused.max():.1e}")
    assert lit.max() / max(lit.min(), 1e-12) > 100.0                     # the quantity IS ill-conditioned: one draw says little
    ratio = np.exp(np.mean(np.log(np.maximum(fused, 1e-12) / np.maximum(lit, 1e-12))))
    assert 1 / 1.5 <= ratio <= 1.5, ratio                               # ... and the two readings are equally far from the truth on average
    f, l, _ = _lone_errors(LONE_CASE)
    assert l < 0.2 * np.median(lit) and f <= np.median(lit)             # the soak's case: the literal at the lucky end, the fused reading below the median


@pytest.mark.gpu
def test_lone_tiny_surfels_kernels_and_literal_reading_spread_alike(gpu_device):
    errs = [e for e in (_lone_errors(c, gpu_device) for c in _lone_scenes(40)) if e is not None]
    hip, lit = np.array([e[2] for e in errs]), np.array([e[1] for e in errs])
    ratio = np.exp(np.mean(np.log(np.maximum(hip, 1e-12) / np.maximum(lit, 1e-12))))
    print(f"kernels: min {hip.min():.1e} median {np.median(hip):.1e} max {hip.max():.1e}; geometric-mean ratio to the literal reading {ratio:.2f}")
    assert 1 / 1.5 <= ratio <= 1.5, ratio
    _, l, h = _lone_errors(LONE_CASE, gpu_device)
    assert h <= np.median(lit) and h <= 1e-3, (h, l, float(np.median(lit)))
