"""Developer diagnostic: per-wave timeline of render_bwd_kernel (needs `make -C materialrefgs_amd/csrc stats`; GPU box)."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, numpy as np
from materialrefgs_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'tools', 'scratch', 'libmrgs_stats.so')
from helpers import HipRender
from materialrefgs_amd.synthetic import make_shell_scene, orbit_camera, upstream_grads

dev = torch.device('cuda:0')
S = int(sys.argv[1]) if len(sys.argv) > 1 else 0
sc = make_shell_scene(300000, S=S, seed=0, radius_px=7.0, image_size=800)
cam = orbit_camera(0, 800, 800)
L = _lib.lib()
nb = 20032
buf = (ctypes.c_ulonglong * (6 * nb))()
for rep in range(3):
    hr = HipRender(sc, cam, dev); torch.cuda.synchronize()
    hr.backward(*upstream_grads(S, 800, 800)); torch.cuda.synchronize()
    L.mrgs_wave_stats(buf, ctypes.c_int(6 * nb))
a = np.array(buf[:], dtype=np.uint64).reshape(-1, 6)
a = a[a[:, 1] > 0]
a = a[a[:, 0] > a[:, 1].max() - np.uint64(500000)]   # records of the last launch only (5 ms window)
t0 = a[:, 0].astype(np.float64); t1 = a[:, 1].astype(np.float64)
base = t0.min(); t0 = (t0 - base) * 0.01; t1 = (t1 - base) * 0.01   # 100 MHz -> us
dur = t1 - t0
cyc = a[:, 2].astype(np.float64)
iters = (a[:, 3] >> np.uint64(32)).astype(np.float64); act = (a[:, 3] & np.uint64(0xffffffff)).astype(np.float64)
chunks = (a[:, 4] >> np.uint64(32)).astype(np.float64); maxc = (a[:, 4] & np.uint64(0xffffffff)).astype(np.float64)
hw = a[:, 5]
print(f"waves with work {len(a)}  span {t1.max():.1f} us  dur mean {dur.mean():.1f} p50 {np.median(dur):.1f} p99 {np.percentile(dur,99):.1f} max {dur.max():.1f} us")
print(f"sum dur {dur.sum():.0f} us -> avg concurrency {dur.sum()/t1.max():.0f} waves ({dur.sum()/t1.max()/1024:.2f} per SIMD)")
print(f"totals: chunks {chunks.sum():.0f} iters {iters.sum():.0f} active {act.sum():.0f}; cycles/active iter {cyc.sum()/act.sum():.0f}; cycles/iter {cyc.sum()/iters.sum():.0f}")
nbins = 24
edges = np.linspace(0, t1.max(), nbins + 1)
conc = [(np.minimum(t1, edges[i + 1]) - np.maximum(t0, edges[i])).clip(min=0).sum() / (edges[i + 1] - edges[i]) for i in range(nbins)]
print("concurrency timeline (waves resident):", " ".join(f"{c:.0f}" for c in conc))
o = np.argsort(-dur)[:8]
for i in o:
    print(f"  wave start {t0[i]:.1f} dur {dur[i]:.1f} us cycles {cyc[i]:.0f} chunks {chunks[i]:.0f} iters {iters[i]:.0f} active {act[i]:.0f} max_contrib {maxc[i]:.0f} cyc/act {cyc[i]/max(act[i],1):.0f}")
late = np.argsort(-t1)[:8]
print("last finishing:")
for i in late:
    print(f"  wave start {t0[i]:.1f} end {t1[i]:.1f} dur {dur[i]:.1f} active {act[i]:.0f} max_contrib {maxc[i]:.0f}")
# start-time distribution
print("start time percentiles (us):", [round(float(np.percentile(t0, p)), 1) for p in (10, 25, 50, 75, 90, 99, 100)])
