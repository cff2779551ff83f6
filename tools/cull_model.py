"""CPU model check of the blend kernels' block cull (no GPU): the cull record of mrgs_preprocess.hip ("Cull conic for the blend kernels")
and the block test of mrgs_blend_math.h (mrgs_block_may_touch / mrgs_edge_min) restated in numpy -- fp64 where the kernel computes in fp64,
float32 where it computes in float32 -- and run against the oracle's geometry of random scenes: whenever a pixel of an 8 x 8 block reaches
alpha >= 1/255 for a surfel (the blend's own formulas, evaluated in float64), the block test must let the surfel through.  Only NEEDLES are
examined (det / (Qxx Qyy) below --ratio): the two misses the zero-allowance soak ever had (rounds 4 and 5) were surfels
seen edge-on, and an ordinary ellipse is checked by every parity test anyway.

    python tools/cull_model.py [n_scenes] [seed] [--guard 1e-5] [--ratio 1e-3]

A restatement of kernel code: it follows mrgs_preprocess.hip / mrgs_blend_math.h and has to be kept in step with them."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
f32 = np.float32
NEAR = 0.2
A_MIN = 1.0 / 255.0


def cull_record(T, opa, mean2d, guard):
    """(a, b, c) float32 triples of one surfel as preprocess_fwd writes them; a[2] == 0: "not an ellipse, always a candidate"."""
    never = (np.array([1e30] * 4, f32), np.array([0, 0, 1e30, 1e30], f32), np.array([1e30, 0, -1, 0], f32))
    oa = f32(255.0) * f32(opa)
    if oa < f32(0.999):
        return never, None
    lg = np.log(oa).astype(f32)
    tau = float(f32(2.0) * (lg if lg > 0 else f32(0)) * f32(1.0001) + f32(1e-3))
    u, v, w = T[0:3].astype(np.float64), T[3:6].astype(np.float64), T[6:9].astype(np.float64)
    c0, c1, c2 = np.cross(v, w), np.cross(w, u), np.cross(u, v)
    q = lambda x, y: x[0] * y[0] + x[1] * y[1] - tau * x[2] * y[2]
    Qxx, Qxy, Qyy, Qx1, Qy1, Q11 = q(c0, c0), q(c0, c1), q(c1, c1), q(c0, c2), q(c1, c2), q(c2, c2)
    det = Qxx * Qyy - Qxy * Qxy
    a, b = np.zeros(4, f32), np.zeros(4, f32)
    ratio = det / (Qxx * Qyy) if (Qxx > 0 and Qyy > 0) else None
    if Qxx > 0 and Qyy > 0 and det > guard * Qxx * Qyy:
        xc, yc = -(Qyy * Qx1 - Qxy * Qy1) / det, -(Qxx * Qy1 - Qxy * Qx1) / det
        fp = Q11 + Qx1 * xc + Qy1 * yc
        if fp < 0:
            sc = -1.0 / fp
            A, B, C, dn = Qxx * sc, Qxy * sc, Qyy * sc, det * sc * sc
            with np.errstate(over="ignore"):
                a = np.array([xc, yc, A, C]).astype(f32)
                b = np.array([B / C, B / A, dn / C, dn / A]).astype(f32)
            chk = float(a.astype(np.float64).sum() + b.astype(np.float64).sum())
            if not np.isfinite(chk) or not a[2] > 0:
                a, b = np.zeros(4, f32), np.zeros(4, f32)
    rr = f32(np.sqrt(f32(0.5) * f32(tau))) + f32(0.05)
    c = np.array([mean2d[0], mean2d[1], rr * rr, 0], f32)
    return (a, b, c), (ratio, tau)


EPS = 2.0 ** -53


def cull_record_bounded(T, opa, mean2d, W, H, lg=None):
    """The cull record as preprocess_fwd writes it since round 6: every fp64 quantity with a running bound of its rounding error, the
    ellipse a provable superset of the exact level set inside the image (mrgs_preprocess.hip, "Cull conic").
    lg: logf(255 opacity) as the device evaluated it (a test that compares with the kernel bit for bit hands it over; numpy's differs
    in the last bit now and then, and a needle's numbers amplify that by 1 / (det / (Qxx Qyy)))."""
    never = (np.array([1e30] * 4, f32), np.array([0, 0, 1e30, 1e30], f32), np.array([1e30, 0, -1, 0], f32))
    oa = f32(255.0) * f32(opa)
    if oa < f32(0.999):
        return never, None
    lg = np.log(oa).astype(f32) if lg is None else f32(lg)
    tau = float(f32(2.0) * (lg if lg > 0 else f32(0)) * f32(1.0001) + f32(1e-3))
    u, v, w = [np.asarray(T[i:i + 3], dtype=np.float64) for i in (0, 3, 6)]

    def crossE(a_, b_):
        c_, e_ = np.zeros(3), np.zeros(3)
        for i in range(3):
            j, k = (i + 1) % 3, (i + 2) % 3
            t0, t1 = a_[j] * b_[k], a_[k] * b_[j]
            c_[i], e_[i] = t0 - t1, 2.0 * EPS * (abs(t0) + abs(t1))
        return c_, e_

    def qE(x_, ex_, y_, ey_):
        t0, t1, t2 = x_[0] * y_[0], x_[1] * y_[1], tau * (x_[2] * y_[2])
        p = [abs(x_[i]) * ey_[i] + abs(y_[i]) * ex_[i] + ex_[i] * ey_[i] for i in range(3)]
        return t0 + t1 - t2, 2.0 * (4.0 * EPS * (abs(t0) + abs(t1) + abs(t2)) + p[0] + p[1] + tau * p[2])
    (c0, e0), (c1, e1), (c2, e2) = crossE(v, w), crossE(w, u), crossE(u, v)
    (Qxx, Exx), (Qxy, Exy), (Qyy, Eyy) = qE(c0, e0, c0, e0), qE(c0, e0, c1, e1), qE(c1, e1, c1, e1)
    (Qx1, Ex1), (Qy1, Ey1), (Q11, E11) = qE(c0, e0, c2, e2), qE(c1, e1, c2, e2), qE(c2, e2, c2, e2)
    a, b, slack = np.zeros(4, f32), np.zeros(4, f32), f32(0)
    lam = 2.0 * max(Exx + Exy, Exy + Eyy)
    sigma = 2.0 * (E11 + 2.0 * (Ex1 * float(W) + Ey1 * float(H)))
    Qxx_, Qyy_, Q11_ = Qxx - lam, Qyy - lam, Q11 - sigma
    ratio = (Qxx * Qyy - Qxy * Qxy) / (Qxx * Qyy) if (Qxx > 0 and Qyy > 0) else None
    if Qxx_ > 0 and Qyy_ > 0:
        dp, dq = Qxx_ * Qyy_, Qxy * Qxy
        det_ = dp - dq
        det_lo = det_ - 2.0 * (2.0 * EPS * (dp + dq))
        if det_lo > 0:
            n1, n2, m1, m2 = Qyy_ * Qx1, Qxy * Qy1, Qxx_ * Qy1, Qxy * Qx1
            xc, yc = -(n1 - n2) / det_, -(m1 - m2) / det_
            e_det = det_ - det_lo
            dxc = 2.0 * ((2.0 * EPS * (abs(n1) + abs(n2)) + abs(xc) * e_det) / det_lo + 2.0 * EPS * abs(xc))
            dyc = 2.0 * ((2.0 * EPS * (abs(m1) + abs(m2)) + abs(yc) * e_det) / det_lo + 2.0 * EPS * abs(yc))
            a1, a2, a3, a4, a5 = 2.0 * Qx1 * xc, 2.0 * Qy1 * yc, Qxx_ * xc * xc, 2.0 * Qxy * xc * yc, Qyy_ * yc * yc
            fc = ((Q11_ + a1) + a2) + ((a3 + a4) + a5)
            e_f = 2.0 * (8.0 * EPS * (abs(Q11_) + abs(a1) + abs(a2) + abs(a3) + abs(a4) + abs(a5)))
            f_lo = fc - e_f - (Qxx_ + Qyy_) * (dxc * dxc + dyc * dyc)
            if f_lo < 0:
                sc = -1.0 / f_lo
                A, B, C, dn = Qxx_ * sc, Qxy * sc, Qyy_ * sc, det_lo * sc * sc
                with np.errstate(over="ignore"):
                    a = np.array([xc, yc, A, C]).astype(f32)
                    b = np.array([B / C, B / A, dn / C, dn / A]).astype(f32)
                    slack = f32(f32(max(dxc, dyc)) * f32(1.000001) + f32(1e-30))
                chk = float(a.astype(np.float64).sum() + b.astype(np.float64).sum() + float(slack))
                if not np.isfinite(chk) or not a[2] > 0:
                    a, b, slack = np.zeros(4, f32), np.zeros(4, f32), f32(0)
    rr = f32(np.sqrt(f32(0.5) * f32(tau))) + f32(0.05)
    c = np.array([mean2d[0], mean2d[1], rr * rr, slack], f32)
    return (a, b, c), (ratio, tau)


def fma32(x, y, z):
    return f32(np.float64(x) * np.float64(y) + np.float64(z))


def edge_min(D, Q, slope, X, lo, hi, errX):
    Xs = max(f32(abs(X) - errX), f32(0))
    t0 = f32(-slope * X)
    tc = min(max(t0, lo), hi)
    dt = max(f32(f32(abs(f32(tc - t0))) - f32(f32(abs(slope) * errX) + f32(f32(4e-7) * abs(t0)))), f32(0))
    return fma32(f32(D * Xs), Xs, f32(f32(Q * dt) * dt))


def block_may_touch(rec, x0, y0, w, h):
    a, b, c = rec
    x0, y0, w, h = f32(x0), f32(y0), f32(w), f32(h)
    ex_err, ey_err = f32(f32(f32(2e-7) * abs(a[0]) + f32(1e-5)) + c[3]), f32(f32(f32(2e-7) * abs(a[1]) + f32(1e-5)) + c[3])
    dx0 = f32(x0 - a[0]); dx1 = f32(dx0 + w); dy0 = f32(y0 - a[1]); dy1 = f32(dy0 + h)
    A, C = a[2], a[3]
    inside = (dx0 <= ex_err) and (dx1 >= -ex_err) and (dy0 <= ey_err) and (dy1 >= -ey_err)
    with np.errstate(over="ignore", invalid="ignore"):
        g = edge_min(b[2], C, b[0], dx0, f32(dy0 - ey_err), f32(dy1 + ey_err), ex_err)
        g = min(g, edge_min(b[2], C, b[0], dx1, f32(dy0 - ey_err), f32(dy1 + ey_err), ex_err))
        g = min(g, edge_min(b[3], A, b[1], dy0, f32(dx0 - ex_err), f32(dx1 + ex_err), ey_err))
        g = min(g, edge_min(b[3], A, b[1], dy1, f32(dx0 - ex_err), f32(dx1 + ex_err), ey_err))
    ellipse = inside or not (g > f32(1.01)) or A == 0
    ex = max(max(f32(x0 - c[0]), f32(c[0] - f32(x0 + w))), f32(0))
    ey = max(max(f32(y0 - c[1]), f32(c[1] - f32(y0 + h))), f32(0))
    disc = fma32(ex, ex, f32(ey * ey)) <= c[2]
    return bool(ellipse or disc), float(g)


def alpha_block(T, opa, mean2d, x0, y0):
    """alpha of the surfel at the 64 pixel centres of the block, the blend's formulas in float64 (forward.cu:366-398); 0 where rejected."""
    xs, ys = np.meshgrid(np.arange(x0, x0 + 8, dtype=np.float64), np.arange(y0, y0 + 8, dtype=np.float64))
    Tu, Tv, Tw = T[0:3].astype(np.float64), T[3:6].astype(np.float64), T[6:9].astype(np.float64)
    k = xs[..., None] * Tw - Tu
    l = ys[..., None] * Tw - Tv
    p = np.cross(k, l)
    with np.errstate(divide="ignore", invalid="ignore"):
        s = p[..., :2] / p[..., 2:3]
        rho3 = (s * s).sum(-1)
        d = np.stack([mean2d[0] - xs, mean2d[1] - ys], -1)
        rho2 = 2.0 * (d * d).sum(-1)
        rho = np.minimum(rho3, rho2)
        depth = np.where(rho3 <= rho2, s[..., 0] * Tw[0] + s[..., 1] * Tw[1] + Tw[2], Tw[2])
        alpha = np.minimum(0.99, float(opa) * np.exp(-0.5 * rho))
    ok = (p[..., 2] != 0) & (depth >= NEAR) & np.isfinite(alpha)
    return np.where(ok, alpha, 0.0)


def check_scene(scene, cam, guard, ratio_max, sh_degree=0, sample=0, rng=None):
    """sample: also examine that many randomly chosen visible surfels that are NOT needles (ordinary, huge, off-image ellipses)."""
    from oracle import raster_oracle as ro
    o = ro.render_scene(scene, cam, sh_degree=sh_degree)
    H, W = cam.image_height, cam.image_width
    T, m2, no, radii = o.transMat, o.means2D, o.normal_opacity, o.radii
    stats = dict(needles=0, kept_ellipse=0, blocks=0, reach=0, misses=[])
    vis = np.nonzero(radii > 0)[0]
    extra = set((rng or np.random.default_rng(0)).choice(vis, size=min(sample, len(vis)), replace=False).tolist()) if sample and len(vis) else set()
    for g in vis:
        rec, info = cull_record_bounded(T[g], no[g][3], m2[g], W, H) if guard is None else cull_record(T[g], no[g][3], m2[g], guard)
        if info is None or info[0] is None:
            continue
        if not (info[0] < ratio_max) and int(g) not in extra:
            continue
        stats["needles"] += 1
        stats["kept_ellipse"] += int(rec[0][2] != 0)
        r = int(radii[g])
        bx0, bx1 = max(0, int((m2[g][0] - r) // 8)), min((W + 7) // 8, int((m2[g][0] + r) // 8) + 1)
        by0, by1 = max(0, int((m2[g][1] - r) // 8)), min((H + 7) // 8, int((m2[g][1] + r) // 8) + 1)
        for by in range(by0, by1):
            for bx in range(bx0, bx1):
                al = alpha_block(T[g], no[g][3], m2[g], bx * 8, by * 8)
                inimg = (np.arange(bx * 8, bx * 8 + 8)[None, :] < W) & (np.arange(by * 8, by * 8 + 8)[:, None] < H)
                al = np.where(inimg, al, 0.0)
                stats["blocks"] += 1
                if al.max() < A_MIN * (1.0 + 1e-5):
                    continue
                stats["reach"] += 1
                touch, gmin = block_may_touch(rec, bx * 8, by * 8, 7.0, 7.0)
                if not touch:
                    stats["misses"].append(dict(surfel=int(g), block=(bx, by), alpha255=float(al.max() * 255), g=gmin, ratio=info[0]))
    o.close()
    return stats


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("n", type=int, nargs="?", default=20)
    ap.add_argument("seed", type=int, nargs="?", default=0)
    ap.add_argument("--guard", type=float, default=None, help="the constant-guard record of rounds 1-5 with this det / (Qxx Qyy) threshold (default: the bounded record)")
    ap.add_argument("--ratio", type=float, default=1e-3)
    ap.add_argument("--sample", type=int, default=0, help="per scene, also this many random surfels that are not needles")
    a = ap.parse_args()
    from materialrefgs_amd.synthetic import make_shell_scene, orbit_camera
    rng = np.random.default_rng(a.seed)
    tot = dict(needles=0, kept_ellipse=0, blocks=0, reach=0, misses=0)
    for i in range(a.n):
        P = int(rng.choice([12000, 40000]))
        H, W = int(rng.integers(120, 420)), int(rng.integers(120, 420))
        rpx = float(rng.choice([15.0, 40.0]))
        view, sseed = int(rng.integers(0, 8)), int(rng.integers(1 << 30))
        st = check_scene(make_shell_scene(P, S=0, seed=sseed, radius_px=rpx, image_size=max(H, W)), orbit_camera(view, H, W), a.guard, a.ratio, sample=a.sample, rng=rng)
        for k in ("needles", "kept_ellipse", "blocks", "reach"):
            tot[k] += st[k]
        tot["misses"] += len(st["misses"])
        print(f"[{i}] P={P} {H}x{W} r={rpx} view={view}: needles {st['needles']} (ellipse kept for {st['kept_ellipse']}), blocks {st['blocks']}, "
              f"reached {st['reach']}, MISSES {len(st['misses'])} {st['misses'][:3]}", flush=True)
    print(tot)


if __name__ == "__main__":
    main()
