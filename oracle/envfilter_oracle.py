"""Dense CPU restatement of EnvLight.build_mips -- TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench cpu_baseline).

Restates, literally and densely (one weight for every (output texel, source texel) pair, float64):
  * cube_to_dir / pixel_area / ndfGGX                       scene/renderutils/c_src/cubemap.cu:17-46,171-176
  * SpecularCubemapFwdKernel + `out[..., 0:3] / out[..., 3:]` scene/renderutils/c_src/cubemap.cu:238-290, ops.py:459
  * DiffuseCubemapFwdKernel                                  scene/renderutils/c_src/cubemap.cu:110-137
  * __ndfBounds cut-off angle                                scene/renderutils/ops.py:426-437
  * cubemap_mip forward (box) and backward (bilinear cube fetch of 0.25 * dout)   scene/light_utils.py:66-81
  * EnvLight.build_mips                                      scene/light.py:72-86
PARITY UNPINNED: renderutils is a CUDA extension (cannot be built here) and the reference has no test or fixture for it; the
cube fetch of the mip backward is nvdiffrast's `dr.texture` (not vendored), restated in shading_oracle.cube_fetch.  The
specular filter's window is {dot(L, V) >= cos_cutoff} INSIDE the per-face bounding boxes of SpecularBoundsKernel
(cubemap.cu:183-236), whose 16x16-tile interval test is restated in `bounds_mask`: it is conservative while no tile straddles a
face axis (resolutions that are multiples of 32) and drops qualifying texels below that -- at 16x16 and roughness 0.08 a whole
face is one tile whose corner directions all have the same major component, the test fails for the face's own centre, the window
is empty and the reference divides 0 by 0.  (Rows with an empty window are NaN here, as in the reference; the HIP operator leaves
them zero.  The reference's defaults, 128 -> 16 with roughness 1 on the 16x16 level, never get there.)
The dense operators are usable for cubemaps up to ~32x32 (6144^2 weights).  For the reference's default chain (128 -> 16) the levels whose
resolution is a multiple of 32 -- where the bounding boxes are conservative and the window is exactly {dot(L, V) >= cos_cutoff} -- are
applied by `BlockedSpecular`: the same weights, formed a block of output texels at a time in float64 torch and never stored (on
whatever device the caller names; at 128^2 the full-size checks run it on the GPU as a float64 calculator -- it shares no code with the
product's sparse / MFMA operators).  tests/test_shading.py checks the blocked form against the dense one at 32^2.
"""
import numpy as np
import torch

from . import shading_oracle as so


def cube_to_dir(N, normalize=True):
    """[6*N*N, 3] unit directions of the texel centres, index (s*N + y)*N + x (cubemap.cu:32-46); normalize=False: before the division."""
    x = (2.0 * ((np.arange(N) + 0.5) / N) - 1.0)
    fx, fy = np.meshgrid(x, x, indexing="xy")          # fx varies along x (columns), fy along y (rows)
    one = np.ones_like(fx)
    faces = [(one, -fy, -fx), (-one, -fy, fx), (fx, one, fy), (fx, -one, -fy), (fx, -fy, one), (-fx, -fy, -one)]
    d = np.stack([np.stack(f, -1) for f in faces], 0).reshape(-1, 3)
    if not normalize:
        return d
    return d / np.sqrt(np.maximum((d * d).sum(-1, keepdims=True), 1e-20))


def pixel_area(N, dtype=np.float64):
    """[6*N*N] (cubemap.cu:17-30); dtype float32: evaluated in fp32 as the kernel does."""
    if N <= 1:
        return np.ones(6, dtype)
    H = N // 2
    i = np.abs(np.arange(N) - H).astype(dtype)
    d = np.arctan((i + 1) / dtype(H)) - np.arctan(i / dtype(H))
    a = d[None, :] * d[:, None]                          # [y, x]
    return np.tile(a.reshape(-1), 6)


def cos_cutoff(roughness, cutoff=0.99):
    def ndf(a2, c):
        c = np.clip(c, 0.0, 1.0)
        d = (c * a2 - c) * c + 1.0
        return a2 / (d * d * np.pi)
    costheta = np.cos(np.linspace(0, np.pi / 2.0, 1000000))
    D = np.cumsum(ndf(roughness ** 4, costheta))
    return float(costheta[np.argmax(D >= D[..., -1] * cutoff)])


def _corner_dir(x, y, s, N):
    """cube_to_dir at integer texel coordinates that may equal N (tile corners, cubemap.cu:208-209)."""
    fx, fy = 2.0 * ((x + 0.5) / N) - 1.0, 2.0 * ((y + 0.5) / N) - 1.0
    d = np.array([(1.0, -fy, -fx), (-1.0, -fy, fx), (fx, 1.0, fy), (fx, -1.0, -fy), (fx, -fy, 1.0), (-fx, -fy, -1.0)][s])
    return d / np.sqrt(max(float(d @ d), 1e-20))


def bounds_mask(N, cosc, TS=16):
    """[6N^2 (output texel), 6N^2 (source texel)] bool: the source texel lies inside the bounding box SpecularBoundsKernel
    (cubemap.cu:183-236) stores for (output texel, source face): tiles are kept when the interval bound of dot(L, VNR) over the
    tile's four corner directions reaches the cut-off, the box is spanned by the qualifying texels of the kept tiles."""
    D = cube_to_dir(N)
    dots = D @ D.T
    NT = 6 * N * N
    mask = np.zeros((NT, NT), bool)
    ys, xs = np.meshgrid(np.arange(N), np.arange(N), indexing="ij")
    for s in range(6):
        kept = np.zeros((NT, N, N), bool)                    # qualifying texels of the kept tiles, per output texel
        for ty in range((N + TS - 1) // TS):
            for tx in range((N + TS - 1) // TS):
                tsx, tsy, tex, tey = tx * TS, ty * TS, min((tx + 1) * TS, N), min((ty + 1) * TS, N)
                L = np.stack([_corner_dir(tsx, tsy, s, N), _corner_dir(tex, tsy, s, N), _corner_dir(tsx, tey, s, N), _corner_dir(tex, tey, s, N)])
                lo, hi = L.min(0), L.max(0)
                maxdp = np.maximum(lo[None] * D, hi[None] * D).sum(1)              # [NT]
                keep = maxdp >= cosc
                sub = dots[:, s * N * N:(s + 1) * N * N].reshape(NT, N, N)[:, tsy:tey, tsx:tex] >= cosc
                kept[:, tsy:tey, tsx:tex] = sub & keep[:, None, None]
        any_ = kept.any((1, 2))
        big = N + 1
        xmin = np.where(kept, xs[None], big).min((1, 2)); xmax = np.where(kept, xs[None], -1).max((1, 2))
        ymin = np.where(kept, ys[None], big).min((1, 2)); ymax = np.where(kept, ys[None], -1).max((1, 2))
        box = (xs[None] >= xmin[:, None, None]) & (xs[None] <= xmax[:, None, None]) & (ys[None] >= ymin[:, None, None]) & \
              (ys[None] <= ymax[:, None, None]) & any_[:, None, None]
        mask[:, s * N * N:(s + 1) * N * N] = box.reshape(NT, N * N)
    return mask


_SPEC_CACHE = {}


def specular_matrix(N, roughness, cutoff=0.99):
    """Row-normalised dense operator [6N^2, 6N^2] of specular_cubemap (cached per (N, roughness, cutoff): it is a constant)."""
    key = (int(N), float(roughness), float(cutoff))
    if key not in _SPEC_CACHE:
        _SPEC_CACHE[key] = _specular_matrix(N, roughness, cutoff)
    return _SPEC_CACHE[key]


def specular_weights(N, roughness, cosc):
    """Un-normalised dense weights [6N^2 (output texel), 6N^2 (source texel)] of SpecularCubemapFwdKernel (cubemap.cu:238-290) for a
    given cut-off cosine (ops.py hands the kernel the float32 value of __ndfBounds): out[..., :3] = W @ cube, out[..., 3] = W.sum(1)."""
    cosc = np.float32(cosc)
    D = cube_to_dir(N)
    dots = D @ D.T                                        # [out t, src s] = dot(L_s, VNR_t)
    a2 = (roughness * roughness) ** 2
    Hh = D[None, :, :] + D[:, None, :]                   # L + VNR
    Hh = Hh / np.sqrt(np.maximum((Hh * Hh).sum(-1, keepdims=True), 1e-20))
    vh = np.maximum((Hh * D[:, None, :]).sum(-1), 0.0)
    c = np.clip(vh, 0.0, 1.0)
    dd = (c * a2 - c) * c + 1.0
    ndf = a2 / (dd * dd * np.pi)
    W = np.maximum(dots, 0.0) * ndf * pixel_area(N)[None, :] / 4.0
    return np.where((dots >= cosc) & bounds_mask(N, cosc), W, 0.0)


def _specular_matrix(N, roughness, cutoff):
    W = specular_weights(N, roughness, cos_cutoff(roughness, cutoff))
    with np.errstate(invalid="ignore", divide="ignore"):
        return W / W.sum(1, keepdims=True)


class DenseOp:
    """A dense operator behind the matvec / rmatvec interface of BlockedSpecular."""

    def __init__(self, P):
        self.P = P
        self.shape = P.shape
        self.res = int(round(np.sqrt(P.shape[0] / 6)))

    def matvec(self, x):
        return self.P @ np.asarray(x, dtype=np.float64).reshape(-1, 3)

    def rmatvec(self, g):
        return self.P.T @ np.asarray(g, dtype=np.float64).reshape(-1, 3)


class BlockedSpecular:
    """The row-normalised operator of specular_matrix(N, roughness) for N a multiple of 32, never stored: weights of `block` output texels
    against all source texels at a time, torch on `device`.  Same formulas as specular_weights (cubemap.cu:238-290).
    literal32 = False: float64; with unit directions the half vector's cosine  dot(normalize(L + V), V)  is  (1 + dot(L, V)) / sqrt(2 + 2 dot(L, V)).
    literal32 = True: THE WEIGHTS as the reference's kernel forms them -- float32, its expression tree (directions normalised in fp32, the half
    vector normalised, ndfGGX as  a2 / (d d pi)  with  d = (c a2 - c) c + 1) -- and then applied in float64.  At roughness 0.08 (a2 = 4e-5)
    the window is a cone of 1.2 degrees and  d = 1 - c^2 (1 - a2)  is a difference of two numbers near 1 that comes out around 5e-5: fp32
    holds it to ~1e-3, the weight to a few 1e-3, and a texel at the rim of the window is in or out by the last bit of a dot product --
    the fp32 formula of the REFERENCE is ill-conditioned there (measured at 128^2: the literal fp32 weights sit up to several per cent
    of a level's range from the float64 ones).  This is the truth leg of the prefilter: a product operator is held to be no further
    from the float64 levels than this literal reading is (render_oracle.leaf_gradient_report)."""

    def __init__(self, N, roughness, cutoff=0.99, device="cpu", block=1024, literal32=False):
        assert N % 32 == 0, "the bounding boxes of SpecularBoundsKernel are conservative for multiples of 32 only (module docstring)"
        self.res, self.block, self.device, self.literal32 = int(N), int(block), torch.device(device), bool(literal32)
        self.shape = (6 * N * N, 6 * N * N)
        self.a2 = float((roughness * roughness) ** 2)
        self.cosc = float(np.float32(cos_cutoff(roughness, cutoff)))
        self.D = torch.from_numpy(cube_to_dir(N)).to(self.device)
        self.area = torch.from_numpy(pixel_area(N)).to(self.device)
        if literal32:
            d = torch.from_numpy(cube_to_dir(N, normalize=False).astype(np.float32)).to(self.device)
            self.D = d / torch.sqrt(torch.clamp((d * d).sum(-1, keepdim=True), min=1e-20))
            self.area = torch.from_numpy(pixel_area(N, np.float32)).to(self.device)

    def _weights(self, r0, r1):
        if self.literal32:
            f = torch.float32
            a2, pi = torch.tensor(self.a2, dtype=f, device=self.device), torch.tensor(3.14159265358979323846, dtype=f, device=self.device)
            V = self.D[r0:r1]
            dots = V @ self.D.T                                                  # dot(L, VNR), fp32
            Hh = self.D[None, :, :] + V[:, None, :]
            Hh = Hh / torch.sqrt(torch.clamp((Hh * Hh).sum(-1, keepdim=True), min=1e-20))
            c = torch.clamp((Hh * V[:, None, :]).sum(-1), min=0.0).clamp_(0.0, 1.0)
            dd = (c * a2 - c) * c + 1.0
            W = torch.clamp(dots, min=0.0) * (a2 / (dd * dd * pi)) * self.area[None, :] / 4.0
            W = torch.where(dots >= torch.tensor(self.cosc, dtype=f, device=self.device), W, torch.zeros((), dtype=f, device=self.device))
            return W.double()
        dots = self.D[r0:r1] @ self.D.T                                       # [block (output texel), source texel]
        vh = ((1.0 + dots) / torch.sqrt(torch.clamp(2.0 + 2.0 * dots, min=1e-20))).clamp_(0.0, 1.0)
        dd = (vh * self.a2 - vh) * vh + 1.0
        W = torch.clamp(dots, min=0.0) * (self.a2 / (dd * dd * np.pi)) * self.area[None, :] / 4.0
        return torch.where(dots >= self.cosc, W, torch.zeros((), dtype=W.dtype, device=W.device))

    def _blocks(self):
        blk = max(64, self.block // 8) if self.literal32 else self.block       # (the literal half vectors are a [block, 6 N^2, 3] tensor)
        for r0 in range(0, self.shape[0], blk):
            yield r0, self._weights(r0, min(r0 + blk, self.shape[0]))

    def matvec(self, x):
        x = torch.as_tensor(np.asarray(x, dtype=np.float64).reshape(-1, 3)).to(self.device)
        out = torch.empty_like(x)
        for r0, W in self._blocks():
            out[r0:r0 + W.shape[0]] = (W @ x) / W.sum(1, keepdim=True)
        return out.cpu().numpy()

    def rmatvec(self, g):
        g = torch.as_tensor(np.asarray(g, dtype=np.float64).reshape(-1, 3)).to(self.device)
        out = torch.zeros_like(g)
        for r0, W in self._blocks():
            out += W.T @ (g[r0:r0 + W.shape[0]] / W.sum(1, keepdim=True))
        return out.cpu().numpy()


def specular_operator(N, roughness, cutoff=0.99, device=None, literal32=False):
    """The operator of one level: dense (cached) up to 32^2 or when no device is named, blocked above.  literal32: the weights in the
    reference's fp32 arithmetic (BlockedSpecular; levels it cannot serve -- resolutions that are not multiples of 32 -- stay float64)."""
    if literal32 and N % 32 == 0:
        return BlockedSpecular(N, roughness, cutoff, device or "cpu", literal32=True)
    if device is None or N <= 32 or N % 32 != 0:
        return DenseOp(specular_matrix(N, roughness, cutoff))
    return BlockedSpecular(N, roughness, cutoff, device)


def diffuse_matrix(N):
    D = cube_to_dir(N)
    ct = np.minimum(np.maximum(D @ D.T, 0.0), 0.999)
    return ct * pixel_area(N)[None, :] / 3.141592       # diffuse_cubemap is NOT normalised by the weight sum


def mip_forward(cube):
    """[6,N,N,3] -> [6,N/2,N/2,3] (avg_pool2d 2x2)."""
    N = cube.shape[1]
    return cube.reshape(6, N // 2, 2, N // 2, 2, 3).mean(axis=(2, 4))


def mip_backward(dout):
    """The reference's rule (light_utils.py:71-80): bilinear cube fetch of 0.25 * dout at the finer level's texel centres."""
    r = dout.shape[1] * 2
    lin = torch.linspace(-1.0 + 1.0 / r, 1.0 - 1.0 / r, r, dtype=torch.float64)
    gy, gx = torch.meshgrid(lin, lin, indexing="ij")
    out = np.zeros((6, r, r, 3))
    t = torch.from_numpy(np.ascontiguousarray(dout * 0.25))
    for s in range(6):
        v = so.cube_to_dir(s, gx, gy)
        v = v / torch.sqrt(torch.clamp((v * v).sum(-1, keepdim=True), min=1e-20))
        out[s] = so.cube_fetch(t, v.reshape(-1, 3)).reshape(r, r, 3).numpy()
    return out


def build_mips(base, min_res, min_roughness=0.08, max_roughness=0.5, cutoff=0.99, device=None, literal32=False):
    """EnvLight.build_mips forward: returns (specular levels, diffuse, per-level operators -- objects with matvec / rmatvec).
    device: where the levels above 32^2 are evaluated (BlockedSpecular); None: dense everywhere (small cubemaps only).
    literal32: the filter weights of the levels that are multiples of 32 as the reference's fp32 kernel forms them (the truth leg)."""
    raw = [np.asarray(base, dtype=np.float64)]
    while raw[-1].shape[1] > min_res:
        raw.append(mip_forward(raw[-1]))
    n = len(raw)
    ops = []
    for idx in range(n - 1):
        rough = (idx / (n - 2)) * (max_roughness - min_roughness) + min_roughness
        ops.append(specular_operator(raw[idx].shape[1], rough, cutoff, device, literal32))
    ops.append(specular_operator(raw[-1].shape[1], 1.0, cutoff, device, literal32))
    spec = [P.matvec(r).reshape(r.shape) for P, r in zip(ops, raw)]
    diffuse = (diffuse_matrix(raw[-1].shape[1]) @ raw[-1].reshape(-1, 3)).reshape(raw[-1].shape)
    return spec, diffuse, ops


def build_mips_backward(ops, g_spec, g_diffuse=None):
    """Gradient w.r.t. base of sum_l <g_spec[l], specular[l]> (+ <g_diffuse, diffuse>) under the reference's autograd rules."""
    g = None
    for l in range(len(ops) - 1, -1, -1):
        gl = ops[l].rmatvec(g_spec[l]).reshape(g_spec[l].shape)
        if l == len(ops) - 1 and g_diffuse is not None:      # diffuse_cubemap reads the raw last level (scene/light.py:84-86)
            N = g_spec[l].shape[1]
            gl = gl + (diffuse_matrix(N).T @ np.asarray(g_diffuse, dtype=np.float64).reshape(-1, 3)).reshape(g_spec[l].shape)
        if g is not None:
            gl = gl + mip_backward(g)
        g = gl
    return g
