"""Randomised sweep of the rasterizer parity check (tests/test_gpu_parity.py: compare_all) over scene sizes, image sizes, channel
counts, SH degrees and views -- a one-off soak run for the GPU box, not part of the test suite.

    python tools/stress_parity.py [n_cases] [seed] [i,j,k: run only these cases of the sequence]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_parity import compare_all  # noqa: E402
from materialrefgs_amd.synthetic import make_shell_scene, orbit_camera  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
only = set(int(v) for v in sys.argv[3].split(",")) if len(sys.argv) > 3 else None
dev = torch.device("cuda:0")
bad = 0
t0 = time.time()
for i in range(n):
    P = int(rng.choice([1, 7, 63, 64, 65, 500, 3000, 12000, 40000]))
    S = int(rng.choice([0, 1, 3, 4, 8, 11, 12, 24]))
    H, W = int(rng.integers(17, 420)), int(rng.integers(17, 420))
    deg = int(rng.integers(0, 4))
    rpx = float(rng.choice([1.5, 4.0, 7.0, 15.0, 40.0]))
    view = int(rng.integers(0, 8))
    scene_seed = int(rng.integers(1 << 30))
    if only is not None and i not in only:
        continue
    scene = make_shell_scene(P, S=S, seed=scene_seed, radius_px=rpx, image_size=max(H, W))
    live = 0
    if S == 12 and scene_seed % 2 == 0:      # every other twelve-channel case as rows of nine channels + padding (MrgsRasterInputs::features_live)
        scene.features[:, 9:] = 0.0
        live = 9
    try:
        compare_all(scene, orbit_camera(view, H, W), dev, sh_degree=deg, features_live=live)
        status = "ok"
    except AssertionError as e:
        bad += 1
        import traceback
        tb = traceback.extract_tb(e.__traceback__)[-1]
        status = f"FAIL line {tb.lineno}: {tb.line} | " + str(e)[:160].replace("\n", " ")
    print(f"[{i:3d}] P={P:6d} S={S:2d}{' live=9' if live else ''} {H}x{W} deg={deg} r={rpx:4.1f} view={view}: {status}", flush=True)
print(f"{n - bad} of {n} cases passed in {time.time() - t0:.0f} s")
# the identity of the kernels this log speaks for (tools/publish_profiles.sh refuses a log whose digest is not the tree's)
sys.path.insert(0, ROOT)
from bench import kernel_source_digest  # noqa: E402
print(f"kernel_source_digest: {kernel_source_digest()}")
sys.exit(1 if bad else 0)
