"""numpy restatement of the reference's cubemapencoder CUDA extension -- TEST INFRASTRUCTURE ONLY.

Follows /root/reference/submodules/cubemapencoder/src/cubemapencoder.cu (LEFT_TOP_AS_ORIGIN is defined there, :20):
  Compute_Cubemap_UV            :147-187   direction -> (face, u, v), faces +x 0, -x 1, +y 2, -y 3, +z 4, -z 5
  EdgeTable                     :66-105    the texel across a face edge (flag 1: u < 0, 2: u >= L, 4: v < 0, 8: v >= L)
  Compute_Seamless_Index        :189-262   the four taps of a seamless bilinear fetch, edge and vertex cases
  Compute_Cubemap_UV_Backward   :264-291
  Cubemap_Bilinear_Seamless_Kernel / Cubemap_Bilinear_Kernel / Cubemap_Nearest_Kernel                 :297-428
  ..._Backward_Kernel                                                                                 :510-706
Layouts: inputs [B,3], cubemap [6,C,L,L], fail_value [C], outputs [C,B] (cubemap_encoder.py:22-39).
PARITY: the CUDA extension cannot be built in this image and ships no test vectors, so this restatement is unpinned against its
binary; it is float64 numpy written line by line from the cited source, and the HIP kernels are compared with it.  (The reference
builds with -use_fast_math, setup.py:10: its own divisions and the floor are approximate, so its fp32 results carry ~1e-6 relative
noise of their own.)
"""
import numpy as np


def compute_uv(d):
    """:147-187 -> face [B] int, uv [B,2]."""
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    ax, ay, az = np.abs(x), np.abs(y), np.abs(z)
    max_dim = np.zeros(len(d), np.int64)
    mv = ax.copy()
    m = ay > mv; max_dim[m] = 1; mv[m] = ay[m]
    m = az > mv; max_dim[m] = 2; mv[m] = az[m]
    face = np.zeros(len(d), np.int64)
    u, v = np.zeros(len(d)), np.zeros(len(d))
    with np.errstate(all="ignore"):
        s = max_dim == 0
        u[s], v[s] = (z / x)[s], (y / x)[s]
        p = s & (x >= 0); face[p] = 0; u[p] = -u[p]; v[p] = -v[p]
        n = s & ~(x >= 0); face[n] = 1; u[n] = -u[n]
        s = max_dim == 1
        u[s], v[s] = (x / y)[s], (z / y)[s]
        p = s & (y >= 0); face[p] = 2
        n = s & ~(y >= 0); face[n] = 3; u[n] = -u[n]; v[n] = -v[n]
        s = max_dim == 2
        u[s], v[s] = (x / z)[s], (y / z)[s]
        p = s & (z >= 0); face[p] = 4; v[p] = -v[p]
        n = s & ~(z >= 0); face[n] = 5
    return face, np.stack([u, v], 1)


def edge_table(L, flag, face, x, y):
    """:66-105 (LEFT_TOP_AS_ORIGIN branch), scalar."""
    t = {
        0: {1: (4, L - 1, y), 2: (5, 0, y), 4: (3, L - 1, x), 8: (2, L - 1, x)},
        1: {1: (5, L - 1, y), 2: (4, 0, y), 4: (3, 0, L - 1 - x), 8: (2, 0, L - 1 - x)},
        2: {1: (1, L - 1 - y, L - 1), 2: (0, y, L - 1), 4: (4, x, L - 1), 8: (5, L - 1 - x, L - 1)},
        3: {1: (1, L - 1 - y, 0), 2: (0, y, 0), 4: (4, x, 0), 8: (5, L - 1 - x, 0)},
        4: {1: (1, L - 1, y), 2: (0, 0, y), 4: (3, x, 0), 8: (2, x, 0)},
        5: {1: (0, L - 1, y), 2: (1, 0, y), 4: (3, L - 1 - x, L - 1), 8: (2, L - 1 - x, L - 1)},
    }
    key = flag if flag in (1, 2, 4) else 8          # the table's last branch is "else"
    return t[face][key]


def seamless_index(face, L, uv):
    """:189-262, scalar: (taps [(face, x, y)] x 4 (3 used when vertex), kx, ky, flag, is_vertex)."""
    u = (uv[0] * 0.5 + 0.5) * L
    v = (-uv[1] * 0.5 + 0.5) * L
    ux0, uy0 = int(np.floor(u - 0.5)), int(np.floor(v - 0.5))
    ux1, uy1 = ux0 + 1, uy0 + 1
    kx, ky = u - ux0 - 0.5, v - uy0 - 0.5
    cl = lambda a: min(max(a, 0), L - 1)
    ux0, ux1, uy0, uy1 = cl(ux0), cl(ux1), cl(uy0), cl(uy1)
    flag = 0
    if u < 0.5:
        flag |= 1; kx = 0.5 - u
    elif u >= L - 0.5:
        flag |= 2
    if v < 0.5:
        flag |= 4; ky = 0.5 - v
    elif v >= L - 0.5:
        flag |= 8
    if (flag & 3) and (flag & 12):
        taps = [(face, ux0, uy0), edge_table(L, flag & 3, face, ux0, uy0), edge_table(L, flag & 12, face, ux0, uy0), None]
        return taps, kx, ky, flag, True
    if flag & 3:
        taps = [(face, ux0, uy0), edge_table(L, flag, face, ux0, uy0), (face, ux0, uy1), edge_table(L, flag, face, ux0, uy1)]
    elif flag & 12:
        taps = [(face, ux0, uy0), (face, ux1, uy0), edge_table(L, flag, face, ux0, uy0), edge_table(L, flag, face, ux1, uy0)]
    else:
        taps = [(face, ux0, uy0), (face, ux1, uy0), (face, ux0, uy1), (face, ux1, uy1)]
    return taps, kx, ky, flag, False


def plain_index(face, L, uv):
    """Cubemap_Bilinear_Kernel :357-378 (clamped taps on the same face)."""
    u = (uv[0] * 0.5 + 0.5) * L
    v = (-uv[1] * 0.5 + 0.5) * L
    ux0, uy0 = int(np.floor(u - 0.5)), int(np.floor(v - 0.5))
    kx, ky = u - ux0 - 0.5, v - uy0 - 0.5
    cl = lambda a: min(max(a, 0), L - 1)
    x0, x1, y0, y1 = cl(ux0), cl(ux0 + 1), cl(uy0), cl(uy0 + 1)
    return [(face, x0, y0), (face, x1, y0), (face, x0, y1), (face, x1, y1)], kx, ky, 0, False


def uv_backward(face, d, g_uv):
    """:264-291: g_uv = dL/d(u, v) of Compute_Cubemap_UV's outputs -> dL/d(x, y, z)."""
    x, y, z = d
    gu, gv = g_uv
    if face // 2 == 0:
        if face == 0: gu, gv = -gu, -gv
        else: gu = -gu
        return np.array([-(z * gu + y * gv) / (x * x), gv / x, gu / x])
    if face // 2 == 1:
        if face == 3: gu, gv = -gu, -gv
        return np.array([gu / y, -(x * gu + z * gv) / (y * y), gv / y])
    if face == 4: gv = -gv
    return np.array([gu / z, gv / z, -(x * gu + y * gv) / (z * z)])


def encode(inputs, cubemap, fail_value, interp=1, seamless=1, grad_outputs=None):
    """Forward (and, with grad_outputs [C,B], backward) of cubemap_encode.  Returns outputs [C,B] or
    (outputs, grad_inputs [B,3], grad_cubemap [6,C,L,L], grad_fail [C])."""
    inputs = np.asarray(inputs, np.float64); cubemap = np.asarray(cubemap, np.float64); fail_value = np.asarray(fail_value, np.float64)
    B, C, L = inputs.shape[0], cubemap.shape[1], cubemap.shape[2]
    out = np.zeros((C, B))
    bw = grad_outputs is not None
    if bw:
        go = np.asarray(grad_outputs, np.float64)
        g_in, g_cm, g_fail = np.zeros((B, 3)), np.zeros_like(cubemap), np.zeros(C)
    face, uv = compute_uv(inputs)
    for n in range(B):
        d = inputs[n]
        if d[0] == 0 and d[1] == 0 and d[2] == 0:
            out[:, n] = fail_value
            if bw:
                g_fail += go[:, n]
            continue
        f = int(face[n])
        if interp == 0:                                   # Cubemap_Nearest_Kernel :380-428
            u = (uv[n, 0] * 0.5 + 0.5) * L
            v = (-uv[n, 1] * 0.5 + 0.5) * L
            ux, uy = min(max(int(u), 0), L - 1), min(max(int(v), 0), L - 1)     # int(): truncation toward zero
            out[:, n] = cubemap[f, :, uy, ux]
            if bw:
                g_cm[f, :, uy, ux] += go[:, n]
            continue
        taps, kx, ky, flag, vertex = (seamless_index if seamless else plain_index)(f, L, uv[n])
        val = lambda t: cubemap[t[0], :, t[2], t[1]]
        v00, v01, v10 = val(taps[0]), val(taps[1]), val(taps[2])
        v11 = (v00 + v01 + v10) / 3.0 if vertex else val(taps[3])
        out[:, n] = (1 - ky) * ((1 - kx) * v00 + kx * v01) + ky * ((1 - kx) * v10 + kx * v11)
        if bw:
            g = go[:, n]
            w = [(1 - ky) * (1 - kx), (1 - ky) * kx, ky * (1 - kx), ky * kx]
            if vertex:
                extra = ky * kx / 3.0
                for k in range(3):
                    t = taps[k]
                    g_cm[t[0], :, t[2], t[1]] += (w[k] + extra) * g
            else:
                for k in range(4):
                    t = taps[k]
                    g_cm[t[0], :, t[2], t[1]] += w[k] * g
            lg0 = ((1 - ky) * (v01 - v00) + ky * (v11 - v10)) * 0.5 * L * g
            lg1 = ((1 - kx) * (v10 - v00) + kx * (v11 - v01)) * 0.5 * L * g
            if flag & 1: lg0 = -lg0
            if flag & 4: lg1 = -lg1
            lg1 = -lg1                                     # LEFT_TOP_AS_ORIGIN
            for c in range(C):
                g_in[n] += uv_backward(f, d, (lg0[c], lg1[c]))
    if bw:
        return out, g_in, g_cm, g_fail
    return out
