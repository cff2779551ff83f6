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
FWD = len(sys.argv) > 2 and sys.argv[2] == 'fwd'
sc = make_shell_scene(300000, S=S, seed=0, radius_px=7.0, image_size=800)
cam = orbit_camera(0, 800, 800)
L = _lib.lib()
nb = 65536
buf = (ctypes.c_ulonglong * (8 * nb))()
for rep in range(3):
    hr = HipRender(sc, cam, dev); torch.cuda.synchronize()
    hr.backward(*upstream_grads(S, 800, 800)); torch.cuda.synchronize()
    (L.mrgs_wave_stats_fwd if FWD else L.mrgs_wave_stats)(buf, ctypes.c_int(8 * nb))
a = np.array(buf[:], dtype=np.uint64).reshape(-1, 8)
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

# per-SIMD load (HW_ID: wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13]; XCC_ID[3:0] in the high word)
hwid = (hw & np.uint64(0xffffffff)).astype(np.int64); xcc = ((hw >> np.uint64(32)) & np.uint64(0xf)).astype(np.int64)
simd = (hwid >> 4) & 3; cu = (hwid >> 8) & 15; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 7
key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
uk, inv = np.unique(key, return_inverse=True)
load = np.bincount(inv, weights=act); cnt = np.bincount(inv); busy = np.bincount(inv, weights=dur)
last = np.zeros(len(uk)); np.maximum.at(last, inv, t1)
print(f"SIMDs used {len(uk)}; waves/SIMD mean {cnt.mean():.2f} max {cnt.max()}; active iters/SIMD mean {load.mean():.0f} p10 {np.percentile(load,10):.0f} p90 {np.percentile(load,90):.0f} max {load.max():.0f}")
print(f"SIMD finish time (us): p10 {np.percentile(last,10):.0f} p50 {np.percentile(last,50):.0f} p90 {np.percentile(last,90):.0f} max {last.max():.0f}")
cukey = key // 4
ucu, invc = np.unique(cukey, return_inverse=True)
loadc = np.bincount(invc, weights=act)
print(f"CUs used {len(ucu)}; active iters/CU mean {loadc.mean():.0f} p10 {np.percentile(loadc,10):.0f} p90 {np.percentile(loadc,90):.0f} max {loadc.max():.0f}")
xl = np.bincount(xcc, weights=act)
print("active iters per XCC:", [int(v) for v in xl])

print("corr(SIMD load in active iters, SIMD finish time) = %.3f   most loaded SIMD / mean = %.3f" % (np.corrcoef(load, last)[0, 1], load.max() / load.mean()))
wcu = np.bincount(invc)
print("waves per CU histogram:", {int(k): int(v) for k, v in zip(*np.unique(wcu, return_counts=True))})
if not FWD:
    # backward build: w[7] = time at kernel entry, w[6] = (own-queue ticket done, pull done) relative to it | scan rounds
    te = (a[:, 7].astype(np.int64) - a[:, 0].min().astype(np.int64)) * 0.01
    d0 = (a[:, 6] >> np.uint64(40)).astype(np.float64) * 0.01
    d2 = ((a[:, 6] >> np.uint64(16)) & np.uint64(0xffffff)).astype(np.float64) * 0.01
    rounds = (a[:, 6] & np.uint64(0xffff)).astype(np.int64)
    print("kernel entry (us) p0/50/100:", np.round(np.percentile(te, [0, 50, 100]), 1), " own ticket done after:", np.round(np.percentile(d0, [0, 50, 100]), 1),
          " pull done after:", np.round(np.percentile(d2, [0, 50, 99, 100]), 1), " waves that looked in other queues:", int((rounds > 0).sum()))
for lo, hi in ((0, 50), (50, 100), (100, 150), (150, 200), (200, 300), (300, 400), (400, 100000)):
    m = (act >= lo) & (act < hi)
    if m.sum():
        print(f"  active [{lo},{hi}): waves {m.sum()} duration mean {dur[m].mean():.1f} max {dur[m].max():.1f} us, end max {t1[m].max():.1f}, us per active iter {dur[m].sum() / max(act[m].sum(), 1):.3f}")
