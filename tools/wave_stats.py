"""Developer diagnostic: per-wave timeline of render_bwd_kernel (needs `make -C materialrefgs_amd/csrc stats`; GPU box)."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, numpy as np
from materialrefgs_amd import _lib
_lib.LIB_PATH = os.environ.get('MRGS_STATS_LIB', os.path.join(ROOT, 'tools', 'scratch', 'libmrgs_stats.so'))
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
if FWD and len(sys.argv) > 3:
    L.mrgs_wave_stats_min_total(ctypes.c_int(int(sys.argv[3])))
rs = None
for rep in range(4):
    hr = HipRender(sc, cam, dev, rs=rs); torch.cuda.synchronize()
    rs = hr.rs   # the same camera tensors again: from the second render on the work hint of this camera is warm, as in a training loop
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

if FWD:
    # how thin the forward waves run: entries tested while only a few pixels of the block are still alive
    le4 = (a[:, 7] >> np.uint64(40)).astype(np.float64); le8 = ((a[:, 7] >> np.uint64(20)) & np.uint64(0xfffff)).astype(np.float64)
    le16 = (a[:, 7] & np.uint64(0xfffff)).astype(np.float64); bl8 = chunks   # upper word of w[4] in the forward build
    tested = iters
    print(f"entries tested {tested.sum():.0f}: with <=4 live px {le4.sum():.0f}  <=8 {le8.sum():.0f}  <=16 {le16.sum():.0f}; blended with <=8 live {bl8.sum():.0f} of {act.sum():.0f}")
    o = np.argsort(-dur)[:200]
    print(f"  200 longest waves: tested {tested[o].sum():.0f}  <=4 {le4[o].sum():.0f}  <=8 {le8[o].sum():.0f}  <=16 {le16[o].sum():.0f}  blended<=8 {bl8[o].sum():.0f} of {act[o].sum():.0f}")
    for i in o[:10]:
        print(f"   dur {dur[i]:.1f} tested {tested[i]:.0f} blended {act[i]:.0f} le4 {le4[i]:.0f} le8 {le8[i]:.0f} le16 {le16[i]:.0f} bl8 {bl8[i]:.0f}")
    cost = iters + 3 * act; est = a[:, 6].astype(np.float64)
    sc_, se_ = np.bincount(inv, weights=cost), np.bincount(inv, weights=est)
    print(f"per SIMD: measured cost max/mean {sc_.max()/sc_.mean():.3f} p90/mean {np.percentile(sc_,90)/sc_.mean():.3f}; estimate max/mean {se_.max()/se_.mean():.3f}; "
          f"corr(cost, finish) {np.corrcoef(sc_, last)[0,1]:.3f} corr(est, finish) {np.corrcoef(se_, last)[0,1]:.3f} corr(waves, finish) {np.corrcoef(cnt, last)[0,1]:.3f}")
    print(f"estimate vs measured per wave: corr {np.corrcoef(est, cost)[0,1]:.4f}, sum est {est.sum():.0f} sum cost {cost.sum():.0f}, |diff| mean {np.abs(est-cost).mean():.1f}")
    first = np.full(len(uk), 1e9); np.minimum.at(first, inv, t0)
    print("first wave start per SIMD p0/50/100:", np.round(np.percentile(first, [0, 50, 100]), 1))
    # cycles per cost unit on each SIMD
    rate = (last - first) / sc_
    print("us per cost unit per SIMD p10/50/90/max:", np.round(np.percentile(rate, [10, 50, 90, 100]), 4))
    for q in (0, 1, 2, 3):
        mq = (uk % 4) == q
        print(f"  simd {q}: finish mean {last[mq].mean():.1f} cost mean {sc_[mq].mean():.0f}")
    xs = uk // (4 * 16 * 2 * 8)
    for x in range(8):
        mx = xs == x
        print(f"  xcc {x}: SIMDs {mx.sum()} finish mean {last[mx].mean():.1f} max {last[mx].max():.1f} cost mean {sc_[mx].mean():.0f} max {sc_[mx].max():.0f}")
