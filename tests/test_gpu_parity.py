"""GPU parity tests: the HIP rasterizer (through its reference-shaped Python API and the C ABI) against the CPU oracle.

Bar:
  * geometry state and every integer of the binning state -- radii, tiles_touched, num_rendered, point_list, ranges --
    are BIT-EXACT (the preprocess kernel and the oracle evaluate the same un-fused IEEE expressions);
  * blended maps: |hip - oracle| <= 2e-5 * max|oracle| per map; the distortion channel is held to an ABSOLUTE 5e-6 instead,
    because the reference's one-pass formula (m^2 A + M2 - 2 m M1, forward.cu:412) subtracts O(1) terms to produce a
    value of order 1e-5: its result carries the rounding noise of the O(1) terms whatever the implementation;
  * per-pixel contributor counters (last, median) BIT-EXACT: every decision of the blend -- alpha >= 1/255, depth >= 0.2, rho3d <= rho2d,
    T (1 - alpha) < 1e-4, T > 0.5 -- is taken with the oracle's arithmetic wherever the fast value is within its error band of the
    threshold (mrgs_blend_math.h "Exact decisions"), so no pixel of any map may sit outside the tolerances either;
  * gradients: max|hip - oracle| / max|oracle| <= 1e-4 per tensor (BASELINE.json north star).
"""
import math

import numpy as np
import pytest
import torch

from helpers import HipRender, rel_err
from materialrefgs_amd.synthetic import make_shell_scene, orbit_camera, upstream_grads

pytestmark = pytest.mark.gpu

MAP_TOL = 2e-5
DIST_ABS_TOL = 5e-6
GRAD_TOL = 1e-4

CASES = [
    # P, S, H, W, radius_px, sh_degree, view
    (64, 0, 64, 64, 10.0, 3, 0),
    (1000, 8, 128, 128, 4.0, 3, 1),
    (2000, 3, 200, 136, 6.0, 2, 2),       # ragged tiles, S not a multiple of 4
    (5000, 11, 97, 211, 5.0, 1, 3),       # render_volume's S = 11, odd image size
    (3000, 24, 64, 64, 6.0, 0, 4),        # MAX_FEATURES
    (20000, 8, 400, 400, 7.0, 3, 5),
]


def _map_ok(a, b, tol, absolute=False):
    """[..., H, W] boolean map of the pixels within `tol` (relative to max|b| unless absolute)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    den = 1.0 if absolute else max(float(np.abs(b).max()), 1e-30)
    d = np.abs(a - b) / den
    return (d <= tol).reshape(-1, *d.shape[-2:]).all(0), float(d.max())


def compare_all(scene, cam, dev, sh_degree=3, scale_modifier=1.0, colors_precomp=None, bg=None, check_grads=True, pixel_allowance=0, features_live=0):
    """`pixel_allowance`: number of pixels that may sit outside the map tolerances.  Zero everywhere in the suite since round 4 (the
    kernels take the blend's decisions exactly); the parameter remains for developer builds that switch that off."""
    from oracle import raster_oracle as ro
    H, W = cam.image_height, cam.image_width
    S = scene.features.shape[1]
    orc = ro.render_scene(scene, cam, sh_degree=sh_degree, scale_modifier=scale_modifier, colors_precomp=colors_precomp, bg=bg)
    hr = HipRender(scene, cam, dev, sh_degree=sh_degree, scale_modifier=scale_modifier,
                   colors_precomp=None if colors_precomp is None else torch.as_tensor(colors_precomp), bg=None if bg is None else torch.as_tensor(bg),
                   features_live=features_live)     # (features_live: the caller has zeroed the scene's padding channels; the oracle blends all of them)
    # ---- integer / geometry state: bit exact
    assert hr.num_rendered == orc.R
    np.testing.assert_array_equal(hr.radii.cpu().numpy(), orc.radii)
    vis = orc.radii > 0
    np.testing.assert_array_equal(hr.export("tiles_touched").astype(np.uint32), orc.tiles_touched)
    for name in ("depths", "means2D", "transMat", "normal_opacity") + (("rgb",) if colors_precomp is None else ()):
        a, b = hr.export(name)[vis], getattr(orc, name)[vis]
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), name
    if colors_precomp is None:
        np.testing.assert_array_equal(hr.export("clamped")[vis], orc.clamped[vis])
    np.testing.assert_array_equal(hr.export("point_list").astype(np.uint32), orc.point_list)
    np.testing.assert_array_equal(hr.export("ranges").astype(np.uint32), orc.ranges)
    # ---- maps
    nc_bad = int((hr.export("n_contrib").astype(np.uint32) != orc.n_contrib).sum())
    assert nc_bad <= 2 * pixel_allowance, nc_bad
    others = hr.others.detach().cpu().numpy()
    checks = [("color", hr.color.detach().cpu().numpy(), orc.color, MAP_TOL, False)]
    if S:
        checks.append(("feature", hr.feature.detach().cpu().numpy(), orc.feature, MAP_TOL, False))
    for ch in range(7):
        checks.append((f"others[{ch}]", others[ch], orc.others[ch], DIST_ABS_TOL if ch == 6 else MAP_TOL, ch == 6))
    checks.append(("final_T", hr.export("final_T"), orc.final_T, MAP_TOL, False))
    good = np.ones((H, W), bool)
    for name, a, b, tol, absolute in checks:
        ok, worst = _map_ok(a, b, tol, absolute)
        good &= ok
        assert worst <= (tol if pixel_allowance == 0 else 5e-3), (name, worst)   # one pair at the alpha = 1/255 threshold: <= 3.9e-3 of a map's range
    assert int((~good).sum()) <= pixel_allowance, int((~good).sum())
    assert int(hr.contrib.abs().sum()) == 0   # out_contrib is allocated and returned but never written
    # ---- gradients
    if check_grads:
        g = upstream_grads(S, H, W)
        if features_live:            # the padding maps have no consumer: their upstream gradient is zero (and not read by the kernels)
            g = list(g)
            g[1] = g[1].clone()
            g[1][features_live:] = 0.0
            g = tuple(g)
        gh = hr.backward(*g)
        go = orc.backward(*g)
        names = ["means3D", "means2D", "opacity", "scales", "rotations"] + (["features"] if S else []) + \
                (["sh"] if colors_precomp is None else ["colors"])
        legs = None
        for k in names:
            e = rel_err(gh[k].reshape(go[k].shape), go[k])
            if e <= GRAD_TOL:
                continue
            # Above the bar: either a real difference or a sum no fp32 evaluation can hold to 1e-4 -- per-pixel terms of order 1e2 that
            # cancel to 1e-3 (a lone surfel seen edge-on: soak case 1289 of seed 10000, where the oracle's own two fp32 readings sit 3e-3
            # from the float64 value of the same formulas).  The truth leg decides (tests/test_truth_leg.py): the
            # kernels may be no further from the float64 evaluation than the literal fp32 reading of the reference is (x 1.5).
            if legs is None:
                legs = {}
                for v in ("lit32", "f64"):
                    o = ro.render_scene(scene, cam, sh_degree=sh_degree, scale_modifier=scale_modifier, colors_precomp=colors_precomp, bg=bg, variant=v)
                    legs[v] = o.backward(*g)
                    o.close()
            e_hip = rel_err(gh[k].reshape(go[k].shape), legs["f64"][k])
            e_lit = rel_err(legs["lit32"][k], legs["f64"][k])
            assert e_hip <= 1.5 * e_lit and e_lit > 0.5 * GRAD_TOL, (k, e, e_hip, e_lit)
            print(f"note: {k}: {e:.2e} from the fp32 oracle, {e_hip:.2e} from float64 where the literal fp32 reading is {e_lit:.2e} from it (ill-conditioned sum)")
    orc.close()
    return hr


@pytest.mark.parametrize("P,S,H,W,rpx,deg,view", CASES)
def test_parity_against_oracle(gpu_device, P, S, H, W, rpx, deg, view):
    scene = make_shell_scene(P, S=S, seed=P + S, radius_px=rpx, image_size=max(H, W))
    compare_all(scene, orbit_camera(view, H, W), gpu_device, sh_degree=deg)


def test_scale_modifier_background_and_precomputed_colours(gpu_device):
    scene = make_shell_scene(1500, S=4, seed=11, radius_px=6.0, image_size=128)
    cam = orbit_camera(6, 128, 128)
    compare_all(scene, cam, gpu_device, scale_modifier=0.7, bg=np.array([0.2, 0.5, 0.9], np.float32))
    cols = torch.rand(1500, 3, generator=torch.Generator().manual_seed(3)).numpy()
    compare_all(scene, cam, gpu_device, colors_precomp=cols)


def test_edge_cases(gpu_device):
    """Surfels behind the camera / outside the frustum, opaque giants (alpha clamp 0.99, early termination), ragged
    image sizes, exact depth ties."""
    scene = make_shell_scene(300, S=2, seed=21, radius_px=9.0, image_size=64)
    cam = orbit_camera(0, 33, 47)
    m = scene.means3D.clone()
    m[0] = cam.camera_center + 3.0 * (cam.camera_center / cam.camera_center.norm())   # behind
    m[1] = torch.tensor([50.0, 50.0, 0.0])                                            # off-screen
    m[10:20] = m[10]                                                                  # identical depth -> ties by index
    opa = scene.opacities.clone(); opa[2:6] = 1.0
    scl = scene.scales.clone(); scl[2:6] = 0.6
    compare_all(scene._replace(means3D=m, opacities=opa, scales=scl), cam, gpu_device)


def test_empty_scene_and_nothing_visible(gpu_device):
    from materialrefgs_amd.rasterizer import GaussianRasterizer
    from helpers import raster_settings
    cam = orbit_camera(0, 32, 48)
    rs = raster_settings(cam, gpu_device)
    e = lambda *s: torch.zeros(*s, device=gpu_device)
    contrib, color, feat, radii, allmap = GaussianRasterizer(rs)(means3D=e(0, 3), means2D=e(0, 3), opacities=e(0, 1), shs=e(0, 16, 3),
                                                                scales=e(0, 2), rotations=e(0, 4))
    assert color.shape == (3, 32, 48) and float(color.abs().sum()) == 0 and radii.numel() == 0 and allmap.shape == (7, 32, 48)
    # all gaussians behind the camera: num_rendered = 0, T = 1 everywhere
    scene = make_shell_scene(100, S=0, seed=2, radius_px=5.0, image_size=48)
    behind = scene._replace(means3D=(cam.camera_center + 2.0 * cam.camera_center / cam.camera_center.norm()).repeat(100, 1).contiguous())
    hr = HipRender(behind, cam, gpu_device)
    assert hr.num_rendered == 0 and int(hr.radii.sum()) == 0
    assert float(hr.color.detach().abs().sum()) == 0 and float(hr.others.detach().abs().sum()) == 0
    g = hr.backward(*upstream_grads(0, 32, 48))
    assert all(float(np.abs(v).sum()) == 0 for v in g.values())


def test_untouched_pixels_ignore_their_upstream_gradients(gpu_device):
    """The reference never reads dL/dpixel of a pixel nothing was blended into (backward.cu:266 loops over contributors only), so
    callers may leave NaN/inf there (e.g. depth / alpha with alpha = 0).  The blend backward must give the same gradients."""
    S, H, W = 3, 96, 96
    scene = make_shell_scene(400, S=S, seed=5, radius_px=4.0, image_size=96)
    cam = orbit_camera(2, H, W)
    hr = HipRender(scene, cam, gpu_device)
    empty = torch.from_numpy(hr.export("n_contrib")[0] == 0)
    assert 0.05 < float(empty.float().mean()) < 0.95
    g_color, g_feat, g_others = upstream_grads(S, H, W)
    clean = hr.backward(g_color, g_feat, g_others)
    hr2 = HipRender(scene, cam, gpu_device)
    bad = [g.clone() for g in (g_color, g_feat, g_others)]
    for i, g in enumerate(bad):
        g[:, empty] = float("nan") if i != 1 else float("inf")
    poisoned = hr2.backward(*bad)
    for k in clean:
        assert np.isfinite(poisoned[k]).all(), k
        assert rel_err(poisoned[k], clean[k]) <= 1e-5, k   # atomics: summation order differs between runs


def test_forward_on_a_capacity_guess_matches_the_two_phase_forward(gpu_device):
    """mrgs_rasterize_forward (no host round trip, binning workspace sized from a guess, pair count read on the device) against
    the two-phase path; a guess that is too small must fall back transparently."""
    from materialrefgs_amd import rasterizer as rz
    S, H, W = 4, 160, 120
    scene = make_shell_scene(3000, S=S, seed=9, radius_px=6.0, image_size=160)
    cam = orbit_camera(3, H, W)
    grads = upstream_grads(S, H, W)
    results = []
    key = (gpu_device.index, 3000, H, W)                       # the guess is kept per (device, P, H, W)
    paths = []
    for guess in (None, 10, "half", 10_000_000, "exact"):      # too small (twice), far too large, exact
        if guess is None:
            rz._PAIR_GUESS.pop(key, None)
        else:
            rz._PAIR_GUESS[key] = results[0][0] if guess == "exact" else results[0][0] // 2 if guess == "half" else guess
        hr = HipRender(scene, cam, gpu_device)
        paths.append(hr.fn.binning_pairs)
        assert key in rz._PAIR_GUESS and rz._PAIR_GUESS[key] >= hr.num_rendered      # ... and refreshed by every forward
        pl = hr.export("point_list")
        rng = hr.export("ranges")
        results.append((hr.num_rendered, pl, rng, hr.color.detach().cpu().numpy(), hr.others.detach().cpu().numpy(), hr.backward(*grads)))
    ref = results[0]
    assert ref[0] > 1000
    # the workspace was carved for the exact count where the guess was missing or too small, for the guess where it fitted
    assert paths == [ref[0], ref[0], ref[0], 10_000_000, ref[0]], paths
    # another image size or surfel count on the same device does not disturb this configuration's guess
    rz._PAIR_GUESS[key] = ref[0]
    HipRender(make_shell_scene(500, S=S, seed=1, radius_px=6.0, image_size=160), orbit_camera(3, 64, 64), gpu_device)
    assert rz._PAIR_GUESS[key] == ref[0]
    for r in results[1:]:
        assert r[0] == ref[0]
        assert np.array_equal(r[1], ref[1]) and np.array_equal(r[2], ref[2])
        assert np.array_equal(r[3], ref[3]) and np.array_equal(r[4], ref[4])
        for k in ref[5]:
            assert rel_err(r[5][k], ref[5][k]) <= 1e-5, k


def raster_settings_of(render):
    """Fresh camera tensors with the values of an earlier render's settings (a new camera as far as the hint cache is concerned)."""
    rs = render.rs
    return rs._replace(viewmatrix=rs.viewmatrix.clone(), projmatrix=rs.projmatrix.clone())


def test_work_hints_only_change_the_schedule(gpu_device):
    """MrgsRasterInputs::work_hint (per-camera work of the previous visit) reorders the blend waves and nothing else: first visit
    (hint all zero), second visit (hint from the first) and a visit with a deliberately wrong hint give identical images."""
    from materialrefgs_amd import rasterizer as rz
    S, H, W = 4, 160, 120
    scene = make_shell_scene(3000, S=S, seed=11, radius_px=6.0, image_size=160)
    cam = orbit_camera(5, H, W)
    rz.reset_work_hints()
    a = HipRender(scene, cam, gpu_device)                   # hint zero -> cull counts
    assert len(rz._WORK_HINTS) == 1
    ent, = rz._WORK_HINTS.values()
    hint = ent.buf
    assert ent.vm is a.rs.viewmatrix and ent.pm is a.rs.projmatrix             # the entry keeps the camera's tensors alive
    assert int(hint.sum()) > 0 and ent.visits == 1                             # the forward stored its measured work
    b = HipRender(scene, cam, gpu_device, rs=a.rs)          # same camera tensors: ordered by the measured work, backward prepared
    assert len(rz._WORK_HINTS) == 1 and ent.visits == 2 and b.fn.prepared_grad_ws is not None and a.fn.prepared_grad_ws is None
    n_work = 4 * ((W + 15) // 16) * ((H + 15) // 16)      # the per-(tile, quadrant) work; the dealt queues of the camera lie behind it
    hint[:n_work].copy_(torch.randint(1, 4000, (n_work,), device=hint.device, dtype=torch.int32))
    c = HipRender(scene, cam, gpu_device, rs=a.rs)          # garbage hint: still only a schedule
    # a camera whose matrices were written in place is a new camera: its old hint is not used
    a.rs.viewmatrix.add_(0.0)
    assert rz._hint_entry(a.rs, gpu_device) is None
    # other camera tensors (even with equal values) get their own entry
    d = HipRender(scene, cam, gpu_device)
    assert len(rz._WORK_HINTS) == 2 and d.fn.prepared_grad_ws is None
    for r in (b, c, d):
        assert r.num_rendered == a.num_rendered
        assert torch.equal(r.color, a.color) and torch.equal(r.others, a.others) and torch.equal(r.feature, a.feature)
    # from its third visit on a camera's forward deals its waves as its last ordering did (MRGS_HINT_REUSE_ORDER: no ordering launch,
    # the tile sort resets the queues and clears the prepared backward's gradient rows): same images, same gradients
    g = upstream_grads(S, H, W)
    ref = a.backward(*g)
    rz.reset_work_hints()
    rs = raster_settings_of(a)
    runs = [HipRender(scene, cam, gpu_device, rs=rs) for _ in range(5)]
    P_ = scene.means3D.shape[0]
    assert rz._hint_entry(rs, gpu_device).dealt_for == (P_, rz._GENERATION[0])
    assert rz._hint_flags(rs, gpu_device, P_, forward=True) == rz._lib.MRGS_HINT_REUSE_ORDER and rz._hint_flags(rs, gpu_device, P_) == 0
    for k, r in enumerate(runs):
        assert torch.equal(r.color, a.color) and torch.equal(r.others, a.others), k
        assert (r.fn.prepared_grad_ws is not None) == (k >= 1), k
    for r in (runs[2], runs[4]):
        got = r.backward(*g)
        for name in ref:
            assert rel_err(got[name], ref[name]) <= 1e-5, name
    # a prepared backward reads the forward's queues from the hint buffer THAT forward used (kept in its ctx): forgetting every hint
    # between a forward and its backward (reset_work_hints(), an eviction, an in-place pose change) must not hand it a fresh, zeroed one
    assert runs[3].fn.prepared_grad_ws is not None and runs[3].fn.work_hint is not None
    rz.reset_work_hints()
    got = runs[3].backward(*g)
    for name in ref:
        assert rel_err(got[name], ref[name]) <= 1e-5, name
    # The cache is keyed by the camera alone: after densification / pruning (another surfel count) the camera's measured work serves on --
    # the visit is WARM (ordered by that work, backward prepared) -- but the deal made for the old set is not reused: that visit orders anew,
    # the ones after it reuse again.  The same after note_surfel_set_changed().  Results: those of a first visit, bit for bit.
    rs2 = raster_settings_of(a)
    [HipRender(scene, cam, gpu_device, rs=rs2) for _ in range(3)]
    n_before = len(rz._WORK_HINTS)
    other = make_shell_scene(1500, S=S, seed=5, radius_px=6.0, image_size=160)
    cold = HipRender(other, cam, gpu_device)                                  # the other scene through fresh camera tensors: a first visit
    assert rz._hint_flags(rs2, gpu_device, 1500, forward=True) == 0           # (decides for the next forward: orders anew, deals for P = 1500)
    warm = HipRender(other, cam, gpu_device, rs=rs2)
    assert len(rz._WORK_HINTS) == n_before + 1 and warm.fn.prepared_grad_ws is not None and cold.fn.prepared_grad_ws is None
    assert torch.equal(warm.color, cold.color) and torch.equal(warm.others, cold.others) and warm.num_rendered == cold.num_rendered
    assert rz._hint_entry(rs2, gpu_device).dealt_for == (1500, rz._GENERATION[0])
    again = HipRender(other, cam, gpu_device, rs=rs2)
    assert torch.equal(again.color, cold.color)
    assert rz._hint_flags(rs2, gpu_device, 1500, forward=True) == rz._lib.MRGS_HINT_REUSE_ORDER
    rz.note_surfel_set_changed()
    assert rz._hint_flags(rs2, gpu_device, 1500, forward=True) == 0
    got, want = again.backward(*g), cold.backward(*g)
    for name in want:
        assert rel_err(got[name], want[name]) <= 1e-5, name


def test_backward_in_two_halves_hands_out_the_colour_factor(gpu_device):
    """mrgs_rasterize_backward_blend / _finish through rasterizer.set_after_blend_hook: the hook sees dL/dRGB of every surfel, clamp
    mask applied (= dL/dsh[:, 0, :] / SH_C0, the factor a view-parallel step all-gathers), in the middle of the backward; the gradients
    are those of the one-call backward."""
    from materialrefgs_amd import rasterizer as rz
    from materialrefgs_amd.dist import SH_C0
    S, H, W = 3, 120, 160
    scene = make_shell_scene(4000, S=S, seed=31, radius_px=6.0, image_size=160)
    cam = orbit_camera(4, H, W)
    g = upstream_grads(S, H, W)
    ref = HipRender(scene, cam, gpu_device).backward(*g)
    seen = []
    rz.set_after_blend_hook(lambda d: seen.append(d.clone()))
    try:
        got = HipRender(scene, cam, gpu_device).backward(*g)
    finally:
        rz.set_after_blend_hook(None)
    assert len(seen) == 1 and tuple(seen[0].shape) == (4000, 3)
    for k in ref:
        assert rel_err(got[k], ref[k]) <= 1e-5, k            # (atomics reorder the blend's sums)
    factor = got["sh"][:, 0, :] / SH_C0
    assert rel_err(seen[0].cpu().numpy(), factor) <= 1e-6
    assert float(np.abs(factor).max()) > 0
    dead = got["means3D"].__abs__().sum(1) == 0                 # surfels that received no gradient at all (culled, or never blended)
    assert float(np.abs(seen[0].cpu().numpy()[dead]).sum()) == 0.0


def test_transposed_camera_matrices_reach_the_warm_path(gpu_device):
    """The reference's Camera keeps `world_view_transform` / `full_proj_transform` as TRANSPOSED views (scene/cameras.py:77-79), i.e.
    not contiguous.  The rasterizer must hand the same contiguous copy to every render of that camera, or the per-camera work hints
    (keyed by the matrices' addresses) never repeat: the second render of such a camera is warm and has its backward prepared."""
    from helpers import raster_settings
    from materialrefgs_amd import rasterizer as rz
    H, W = 96, 128
    scene = make_shell_scene(2000, S=0, seed=4, radius_px=6.0, image_size=128)
    cam = orbit_camera(1, H, W)
    rs0 = raster_settings(cam, gpu_device)
    # as scene/cameras.py builds them: row-major data of the transposed matrix, viewed through .transpose(0, 1)
    vm = rs0.viewmatrix.t().contiguous().transpose(0, 1)
    pm = rs0.projmatrix.t().contiguous().transpose(0, 1)
    assert not vm.is_contiguous() and torch.equal(vm, rs0.viewmatrix)
    rs = rs0._replace(viewmatrix=vm, projmatrix=pm)
    rz.reset_work_hints()
    a = HipRender(scene, cam, gpu_device, rs=rs)
    n_hints = len(rz._WORK_HINTS)
    b = HipRender(scene, cam, gpu_device, rs=rs)
    assert len(rz._WORK_HINTS) == n_hints == 1                      # one camera, one entry
    assert a.fn.prepared_grad_ws is None and b.fn.prepared_grad_ws is not None
    assert torch.equal(a.color, b.color)
    ref = HipRender(scene, cam, gpu_device, rs=rs0)
    assert torch.equal(ref.color, a.color)


@pytest.mark.parametrize("P,M,deg", [(3000, 16, 3), (4096, 16, 2), (777, 4, 1), (64, 9, 2)])
def test_split_sh_layout_is_bit_identical(gpu_device, P, M, deg):
    """shs = (features_dc [P,1,3], features_rest [P,M-1,3]) -- GaussianModel's own tensors -- gives the images and gradients of the
    concatenated [P,M,3] tensor: images bit for bit, gradients to the reordering of the blend's atomics (full waves, a ragged last
    wave, M < 16)."""
    from helpers import raster_settings
    from materialrefgs_amd.rasterizer import GaussianRasterizer
    H, W = 96, 128
    sc = make_shell_scene(P, S=0, seed=P, radius_px=6.0, image_size=128).to(gpu_device)
    cam = orbit_camera(2, H, W)
    rs = raster_settings(cam, gpu_device, deg, 1.0, None)
    g_color, _, g_others = upstream_grads(0, H, W, device=gpu_device)
    res = []
    for split in (False, True):
        leaves = [t.clone().requires_grad_(True) for t in (sc.means3D, torch.zeros_like(sc.means3D), sc.opacities, sc.scales, sc.rotations)]
        sh = sc.shs[:, :M].contiguous()
        if split:
            dc, rest = sh[:, :1].clone().requires_grad_(True), sh[:, 1:].clone().requires_grad_(True)
            shs = (dc, rest)
        else:
            full = sh.clone().requires_grad_(True)
            shs = full
        _, color, _, radii, others = GaussianRasterizer(rs)(means3D=leaves[0], means2D=leaves[1], opacities=leaves[2], shs=shs,
                                                            scales=leaves[3], rotations=leaves[4])
        torch.autograd.backward([color, others], [g_color, g_others])
        g_sh = torch.cat((dc.grad, rest.grad), dim=1) if split else full.grad
        res.append((color.detach(), others.detach(), radii, g_sh, [t.grad for t in leaves]))
    a, b = res
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    # gradients: same kernels, but the blend's fp32 atomics reorder sums from run to run
    assert float(a[3].abs().max()) > 0 and rel_err(b[3].cpu().numpy(), a[3].cpu().numpy()) <= 1e-5
    for x, y in zip(a[4], b[4]):
        assert rel_err(y.cpu().numpy(), x.cpu().numpy()) <= 1e-5


def test_mark_visible(gpu_device):
    from materialrefgs_amd.rasterizer import GaussianRasterizer
    from helpers import raster_settings
    from oracle import raster_oracle as ro
    scene = make_shell_scene(5000, S=0, seed=4, radius_px=5.0, image_size=64)
    cam = orbit_camera(2, 64, 64)
    pts = scene.means3D * 6.0   # some behind the camera
    vis = GaussianRasterizer(raster_settings(cam, gpu_device)).markVisible(pts.to(gpu_device))
    assert vis.dtype == torch.bool
    np.testing.assert_array_equal(vis.cpu().numpy(), ro.mark_visible(pts, cam.world_view_transform, cam.full_proj_transform))


@pytest.mark.parametrize("P,S,H,W", [(300000, 0, 800, 800), (300000, 8, 800, 800), (1000000, 8, 1600, 1600)])
def test_full_size_properties(gpu_device, P, S, H, W):
    """BASELINE.json's C2 itself (300k surfels, 800x800, S=0: the S=0 instances of the blend kernels), the C3 size (S=8) and the raster
    part of C4 (1M surfels, 1600x1600): size-independent properties, no oracle needed."""
    scene = make_shell_scene(P, S=S, seed=0, radius_px=7.0, image_size=max(H, W))
    cam = orbit_camera(0, H, W)
    hr = HipRender(scene, cam, gpu_device)
    R = hr.num_rendered
    tt = hr.export("tiles_touched").astype(np.int64)
    assert tt.sum() == R                                         # checksum of the pair emission
    ranges = hr.export("ranges").astype(np.int64)
    assert (ranges[:, 1] - ranges[:, 0]).sum() == R              # ranges partition the list
    pl = hr.export("point_list").astype(np.int64)
    assert np.array_equal(np.bincount(pl, minlength=P), tt)      # every gaussian appears exactly tiles_touched times
    depth = hr.export("depths")
    nonempty = np.where(ranges[:, 1] > ranges[:, 0])[0]
    for t in nonempty[:: max(1, len(nonempty) // 200)]:          # sortedness inside tiles: depth, then index
        a, b = ranges[t]
        d, idx = depth[pl[a:b]], pl[a:b]
        assert np.all(d[1:] >= d[:-1])
        tie = d[1:] == d[:-1]
        assert np.all(idx[1:][tie] > idx[:-1][tie])
    alpha = hr.others[1].detach().cpu().numpy()
    assert alpha.min() >= 0 and alpha.max() <= 1 and np.isfinite(hr.others.detach().cpu().numpy()).all()
    # idempotence of the forward: a second render is bit-identical
    hr2 = HipRender(scene, cam, gpu_device)
    assert torch.equal(hr.color, hr2.color) and torch.equal(hr.others, hr2.others) and torch.equal(hr.feature, hr2.feature)
    # linearity of the backward in the upstream gradients
    g = upstream_grads(S, H, W)
    g1 = hr.backward(*g)
    g2 = hr2.backward(*[2 * x for x in g])
    for k in g1:
        assert rel_err(g2[k], 2 * g1[k]) < 1e-5, k
        assert np.isfinite(g1[k]).all()
    # culled gaussians receive exactly zero gradient (backward.cu:643)
    dead = hr.radii.cpu().numpy() == 0
    if dead.any():
        assert float(np.abs(g1["means3D"][dead]).sum()) == 0


@pytest.mark.parametrize("S", [0, 8])
def test_full_size_against_oracle(gpu_device, S):
    """BASELINE.json's configuration itself against the C oracle, every bar of compare_all: 300 000 surfels, 800 x 800, S = 0 (C2) and
    S = 8 (the rasterizer call of C3) -- bit-exact binning state and contributor counters, maps, all gradients.  The oracle needs the host
    cores for this (OpenMP over tiles: ~10 s on the GPU box's cores, a few minutes on eight), hence a gpu test of its own and view 0 only."""
    scene = make_shell_scene(300000, S=S, seed=0, radius_px=7.0, image_size=800)
    compare_all(scene, orbit_camera(0, 800, 800), gpu_device)


def test_medium_scene_against_oracle(gpu_device):
    """50k surfels at 400x400 with S=8: the largest case the oracle finishes in a few seconds on 8 cores."""
    scene = make_shell_scene(50000, S=8, seed=7, radius_px=7.0, image_size=400)
    compare_all(scene, orbit_camera(7, 400, 400), gpu_device)


def test_factored_sh_gradient_equals_the_sum_over_views(gpu_device):
    """View-parallel exchange (dist.FactoredGradReducer): the sum over views of the rasterizer's dL/dsh is rebuilt exactly from each
    view's dL/dsh[:, 0, :] / SH_C0 and camera centre (mrgs_sh_grad_expand); also the HIP kernel against its torch restatement."""
    from materialrefgs_amd import dist as mdist
    S, H, W, P = 0, 96, 96, 3000
    scene = make_shell_scene(P, S=S, seed=4, radius_px=5.0, image_size=96)
    dense, rows = None, []
    for view in range(3):
        cam = orbit_camera(view, H, W)
        hr = HipRender(scene, cam, gpu_device)
        g = hr.backward(*upstream_grads(S, H, W))
        sh = torch.from_numpy(g["sh"]).reshape(P, 16, 3)
        dense = sh.clone() if dense is None else dense + sh
        rows.append(torch.cat([(sh[:, 0, :] / mdist.SH_C0).reshape(-1), cam.camera_center.reshape(-1).float()]))
    gathered = torch.stack(rows).contiguous()
    m3 = scene.means3D.float()
    from oracle import dist_oracle
    out_cpu = dist_oracle.expand_sh_gradients(gathered, m3, 16, 3)
    out_gpu = mdist.expand_sh_gradients(gathered.to(gpu_device), m3.to(gpu_device), 16, 3).cpu()
    scale = float(dense.abs().max())
    assert scale > 0
    assert float((out_gpu - out_cpu).abs().max()) <= 2e-6 * scale
    assert float((out_gpu - dense).abs().max()) <= 2e-5 * scale
    out_deg1 = mdist.expand_sh_gradients(gathered.to(gpu_device), m3.to(gpu_device), 16, 1).cpu()
    assert float(out_deg1[:, 4:].abs().max()) == 0.0 and float((out_deg1[:, :4] - out_cpu[:, :4]).abs().max()) <= 2e-6 * scale


def _soak_cases(n, seed, only=None):
    """The generator of tools/stress_parity.py with a fixed seed: scene sizes 1 ... 40 000, 17 ... 420 px, 0 ... 24 channels,
    SH degree 0 ... 3, splat radii 1.5 ... 40 px."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        P = int(rng.choice([1, 7, 63, 64, 65, 500, 3000, 12000, 40000]))
        S = int(rng.choice([0, 1, 3, 4, 8, 11, 12, 24]))
        H, W = int(rng.integers(17, 420)), int(rng.integers(17, 420))
        deg = int(rng.integers(0, 4))
        rpx = float(rng.choice([1.5, 4.0, 7.0, 15.0, 40.0]))
        view = int(rng.integers(0, 8))
        out.append((P, S, H, W, deg, rpx, view, int(rng.integers(1 << 30))))
    return out if only is None else [out[i] for i in only]


# Scenes of the 2 000-scene run `tools/stress_parity.py 2000 10000` that had a pixel outside the bars before round 4: 63 ... 1657 a pair
# whose alpha sat within a few ulp of 1/255 (v_rcp_f32 / v_exp_f32 against the oracle's quotient and exp; now decided exactly), 283 / 828
# contributor counters off by one (the same, at T (1 - alpha) = 1e-4), 329 / 1460 a pair with alpha = 3/255 ... 6/255 dropped by the
# block-level cull (a grazing surfel's needle-shaped conic evaluated as A x^2 + 2 B x y + C y^2 in fp32; now a completed square).
_FORMERLY_OFF = [63, 193, 283, 329, 771, 828, 1385, 1460, 1657]


@pytest.mark.parametrize("case", _soak_cases(20, 0) + _soak_cases(2000, 10000, _FORMERLY_OFF),
                         ids=lambda c: f"P{c[0]}-S{c[1]}-{c[2]}x{c[3]}-d{c[4]}-r{c[5]}-s{c[7]}")
def test_randomised_soak_fixed_seed(gpu_device, case):
    """Twenty fixed scenes of the randomised soak run (tools/stress_parity.py runs thousands) and nine that used to have one pixel
    outside the bars: full bit-exact binning state and contributor counters, maps and gradients, NO pixel allowance."""
    P, S, H, W, deg, rpx, view, seed = case
    scene = make_shell_scene(P, S=S, seed=seed, radius_px=rpx, image_size=max(H, W))
    compare_all(scene, orbit_camera(view, H, W), gpu_device, sh_degree=deg)


@pytest.mark.parametrize("P,H,W,rpx", [(6000, 64, 64, 40.0),      # every tile holds ~6 000 pairs: tile_sort_big_kernel, keys in LDS
                                       (24000, 48, 64, 60.0)])    # ~24 000 pairs per tile: beyond its LDS, outer network stages on global memory
def test_dense_tiles_take_the_big_sort_path(gpu_device, P, H, W, rpx):
    """Tiles whose segment exceeds the per-tile sort kernel (4 096 keys) go through the device-side list of oversized tiles
    (mrgs_binning.hip); point_list / ranges / n_contrib must stay bit-exact and the images within the usual bars."""
    scene = make_shell_scene(P, S=2, seed=77, radius_px=rpx, image_size=max(H, W))
    hr = compare_all(scene, orbit_camera(3, H, W), gpu_device, check_grads=(P <= 6000))
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    assert hr.num_rendered / tiles > (4096 if P <= 6000 else 16384)


def test_image_with_more_tiles_than_the_slice_histograms_hold(gpu_device):
    """Above 20 480 tiles (here 3 200 x 3 200 px = 40 000) the binning falls back to the global radix path of mrgs_sort.hip:
    same bit-exact binning state."""
    H = W = 3200
    scene = make_shell_scene(3000, S=0, seed=5, radius_px=30.0, image_size=W)
    compare_all(scene, orbit_camera(2, H, W), gpu_device, check_grads=False)


def test_backward_prepared_by_the_forward_equals_the_self_contained_backward(gpu_device):
    """The forward clears the gradient rows and sets the backward's queues up (MrgsRasterInputs::bwd_grad_ws) for ONE backward; a
    second backward of the same graph orders and clears by itself.  Both must give the same gradients (up to the order of the
    float atomics)."""
    from helpers import HipRender
    S, H, W = 8, 160, 208
    scene = make_shell_scene(6000, S=S, seed=9, radius_px=6.0, image_size=W)
    cold = HipRender(scene, orbit_camera(4, H, W), gpu_device)
    assert cold.fn.prepared_grad_ws is None          # first render of a camera: no measured work to build the backward's queues from
    hr = HipRender(scene, orbit_camera(4, H, W), gpu_device, rs=cold.rs)
    g = [t.to(gpu_device) for t in upstream_grads(S, H, W)]
    outs, grads = [hr.color, hr.others, hr.feature], [g[0], g[2], g[1]]
    assert hr.fn.prepared_grad_ws is not None
    torch.autograd.backward(outs, grads, retain_graph=True)
    first = {k: v.grad.clone() for k, v in hr.leaves.items() if v.grad is not None}
    assert hr.fn.prepared_grad_ws is None
    for v in hr.leaves.values():
        v.grad = None
    torch.autograd.backward(outs, grads)
    for k, a in first.items():
        b = hr.leaves[k].grad
        assert float((a - b).abs().max()) <= 2e-6 * max(float(b.abs().max()), 1e-20), k


def test_more_begun_renders_than_ticket_slots_in_one_box(gpu_device):
    """A render function may begin any number of rasterizer calls before it asks for their pair counts (deferred_count): the library
    keeps 16 landing slots per thread and device, the wrapper collects its oldest counts before a ticket would go stale."""
    from materialrefgs_amd.rasterizer import deferred_count
    scene = make_shell_scene(800, S=0, seed=5, radius_px=6.0, image_size=64)
    cam = orbit_camera(1, 64, 64)
    first = HipRender(scene, cam, gpu_device)              # (outside a box: sizes the workspace guess)
    ref = first.color.detach().clone()
    renders = []
    with deferred_count() as box:
        for _ in range(40):
            renders.append(HipRender(scene, cam, gpu_device, rs=first.rs))
    box.finish()
    for r in renders:
        assert r.num_rendered == first.num_rendered
        assert torch.equal(r.color.detach(), ref)


@pytest.mark.gpu
def test_padding_channels_left_out_of_the_blend(gpu_device):
    """MrgsRasterInputs::features_live (GaussianRasterizer.features_live): rows of nine channels padded to twelve floats -- the "pgsr" flavour's
    eight material channels and plane distance -- rendered with the hint (the <8, true, 9> kernel instances: eight channels staged as for S = 8, the
    ninth carried in the spare float of the surfel record, 24 + 1 values through the gradient reductions) against the same rows rendered as
    twelve channels.  The forward is the same arithmetic per live
    channel: bit-identical maps, the padding maps zero; the backward sums the ninth channel's gradient in another order: 2e-6 of the largest
    entry; the padding columns of dL_dfeatures stay zero and the gradient of the padding maps is not read (NaN there changes nothing)."""
    S, H, W = 12, 176, 144
    scene = make_shell_scene(6000, S=S, seed=23, radius_px=7.0, image_size=176)
    scene.features[:, 9:] = 0.0
    cam = orbit_camera(3, H, W)
    g = torch.Generator().manual_seed(7)
    gc, gf, go = torch.randn(3, H, W, generator=g), torch.randn(S, H, W, generator=g), torch.randn(7, H, W, generator=g) * 0.1
    a = HipRender(scene, cam, gpu_device)
    b = HipRender(scene, cam, gpu_device, features_live=9)
    assert torch.equal(a.color, b.color) and torch.equal(a.feature, b.feature) and torch.equal(a.others, b.others)
    assert float(b.feature[9:].detach().abs().max()) == 0.0
    gf_nan = gf.clone()
    gf_nan[9:] = float("nan")
    gf_zero = gf.clone()
    gf_zero[9:] = 0.0
    ga = a.backward(gc, gf_zero, go)
    gb = b.backward(gc, gf_nan, go)
    for name in ga:
        x, y = ga[name], gb[name]
        assert bool(np.isfinite(y).all()), name
        assert float(np.abs(x - y).max()) <= 2e-6 * max(1e-30, float(np.abs(x).max())), name
    assert float(np.abs(gb["features"][:, 9:]).max()) == 0.0


@pytest.mark.gpu
def test_visibility_bytes_equal_radii_positive(gpu_device):
    """MRGS_HINT_VISIBLE_BYTES: the forward writes radii > 0 as a byte per gaussian behind the radii (GaussianRasterizer.visible, what the
    render functions return as "visibility_filter") -- equal to the torch comparison, culled gaussians included, for several counts."""
    from materialrefgs_amd.rasterizer import GaussianRasterizer
    from helpers import raster_settings
    for P in (1, 63, 64, 65, 3000):
        scene = make_shell_scene(P, S=0, seed=P, radius_px=5.0, image_size=96).to(gpu_device)
        scene.means3D[::3] *= 40.0                      # a third of them far outside the frustum
        rast = GaussianRasterizer(raster_settings(orbit_camera(2, 96, 80), gpu_device, 3, 1.0, None))
        out = rast(means3D=scene.means3D, means2D=torch.zeros_like(scene.means3D), opacities=scene.opacities, shs=scene.shs,
                   scales=scene.scales, rotations=scene.rotations)
        radii = out[3]
        assert rast.visible is not None and rast.visible.dtype == torch.bool and rast.visible.shape == radii.shape
        assert torch.equal(rast.visible, radii > 0) and (P < 3 or (bool((radii > 0).any()) and bool((radii == 0).any())))


@pytest.mark.gpu
def test_needle_surfel_is_not_culled_from_a_block_it_reaches(gpu_device):
    """Case 1376 of the soak sequence of seed 4242 (tools/stress_parity.py): a surfel seen edge-on to 3e-4 rad reaches alpha = 1.0096 / 255
    at one pixel; the ellipse preprocess_fwd built for its block cull came out at 0.68 of its size (its value at the centre is a cancelled
    fp64 sum) and the pair was culled -- the one miss in 5 500 random scenes across three seeds.  Such needles are
    no longer culled; the whole case at zero allowance."""
    rng = np.random.default_rng(4242)
    for i in range(1377):
        P = int(rng.choice([1, 7, 63, 64, 65, 500, 3000, 12000, 40000]))
        S = int(rng.choice([0, 1, 3, 4, 8, 11, 12, 24]))
        H, W = int(rng.integers(17, 420)), int(rng.integers(17, 420))
        deg = int(rng.integers(0, 4))
        rpx = float(rng.choice([1.5, 4.0, 7.0, 15.0, 40.0]))
        view = int(rng.integers(0, 8))
        scene_seed = int(rng.integers(1 << 30))
    assert (P, S, H, W, deg, rpx, view) == (40000, 0, 341, 294, 2, 40.0, 5)
    scene = make_shell_scene(P, S=S, seed=scene_seed, radius_px=rpx, image_size=max(H, W))
    compare_all(scene, orbit_camera(view, H, W), gpu_device, sh_degree=deg)
