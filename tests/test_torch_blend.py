"""BASELINE.json configs[0]: 1k synthetic gaussians, 128x128, forward-only alpha blend through the pure-PyTorch CPU loop
(oracle/torch_blend.py, the north star's CPU baseline), checked against the C oracle on the same scene."""
import time

import numpy as np
import pytest
import torch

from materialrefgs_amd.synthetic import make_shell_scene, orbit_camera
from oracle import raster_oracle as ro
from oracle import torch_blend as tb


@pytest.mark.parametrize("P,S,H,W,rpx,deg", [(1000, 0, 128, 128, 7.0, 3), (3000, 8, 96, 112, 5.0, 2)])
def test_pure_torch_forward_matches_the_c_oracle(P, S, H, W, rpx, deg):
    scene = make_shell_scene(P, S=S, seed=4, radius_px=rpx, image_size=max(H, W))
    cam = orbit_camera(3, H, W)
    t0 = time.perf_counter()
    color, feature, others, n_contrib, R = tb.render(scene, cam, sh_degree=deg, dtype=torch.float64)
    dt = time.perf_counter() - t0
    orc = ro.render_scene(scene, cam, sh_degree=deg, variant="f64")
    assert R == orc.R
    np.testing.assert_allclose(color.numpy(), orc.color, atol=1e-6)   # float-literal constants (SH, 0.99f, 1/255) are fp32-rounded in the C oracle, doubles here
    np.testing.assert_allclose(others.numpy(), orc.others, atol=1e-6)   # float-literal constants (SH, 0.99f, 1/255) are fp32-rounded in the C oracle, doubles here
    if S:
        np.testing.assert_allclose(feature.numpy(), orc.feature, atol=1e-6)   # float-literal constants (SH, 0.99f, 1/255) are fp32-rounded in the C oracle, doubles here
    assert np.array_equal(n_contrib.numpy().astype(np.uint32), orc.n_contrib)
    assert dt < 60
    # the fp32 run (what the baseline times) agrees with the fp32 oracle to blend rounding
    c32, f32_, o32, n32, R32 = tb.render(scene, cam, sh_degree=deg, dtype=torch.float32)
    o = ro.render_scene(scene, cam, sh_degree=deg, variant="lit32")
    assert R32 == o.R
    assert float(np.abs(c32.numpy() - o.color).max()) < 2e-4 and float(np.abs(o32.numpy()[:5] - o.others[:5]).max()) < 1e-3
