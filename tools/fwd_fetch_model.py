"""What the forward blend must fetch at C2, counted from the binning state (developer tool, GPU):
   R list entries, Q = sum over entries of the quadrants (8x8 blocks) of their tile their box touches.
   One wave per block stages its own candidates: ids + masks of the whole tile list per block, cull record + blend record per candidate."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from helpers import HipRender
from materialrefgs_amd.synthetic import make_shell_scene, orbit_camera

S = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dev = torch.device("cuda", 0)
scene = make_shell_scene(300000, S=S, seed=0, image_size=800, radius_px=7.0)
cam = orbit_camera(0, 800, 800, n_views=8)
r = HipRender(scene, cam, dev)
R = r.num_rendered
q = r.export("qmask")
pop = np.unpackbits(q[:, None], axis=1)[:, 4:].sum()
ranges = r.export("ranges").astype(np.int64)
T = ranges.shape[0]
rec = 80 + 4 * S          # MRGS_REC_F4 = 5 float4 + the feature row
cull = 48
hw_out = 800 * 800 * 4 * (3 + 7 + S + 3 + 2)
per_tile_once = R * (4 + 1 + cull + rec)
per_block = 4 * R * (4 + 1) + pop * (cull + rec)
print(f"S={S} R={R} quadrant candidates Q={int(pop)} ({pop / R:.2f} per entry), tiles {T}")
print(f"records fetched once per TILE (a 256-thread workgroup sharing its stage): {per_tile_once / 1e6:.1f} MB")
print(f"records fetched once per 8x8 BLOCK (a wave per block, its own stage):    {per_block / 1e6:.1f} MB   (+ {hw_out / 1e6:.1f} MB of pixel outputs written)")
