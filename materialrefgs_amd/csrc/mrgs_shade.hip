// mrgs_shade.hip -- environment-cubemap mip lookup and deferred split-sum specular shading on gfx950.
//
//   envmap_lookup_{fwd,bwd}_kernel : EnvLight.__call__ (scene/light.py:99-129): roughness -> mip level (get_mip, :88-96),
//                                    seamless trilinear cubemap fetch over the mip chain, sigmoid.
//   shade_specular_{fwd,bwd}_kernel: get_specular_color_surfel (utils/refl_utils.py:364-419) fused into one pass per pixel:
//                                    camera ray (sample_camera_rays :54-73), mirror direction (reflection :95-98), split-sum
//                                    FG LUT fetch (:373-374), environment lookup (:376), Fresnel-style weight (:377),
//                                    specular = light * alpha * weight (:400-401).
//
// The reference does this with ~25 elementwise torch kernels plus two nvdiffrast `dr.texture` launches.  nvdiffrast is not
// vendored (requirements.txt:57 pins a local path), so its sampling rules are RESTATED here and in oracle/shading_oracle.py
// -- parity unpinned for the lookups: texel centres at (i + 0.5) / res, bilinear taps, `clamp` boundary for the 2D LUT,
// seamless cube edges (a tap that falls off a face is taken from the face across the edge; the non-existent corner tap gets
// zero weight and the other three are renormalised), mip level = mip_level_bias clamped to [0, levels-1] with linear
// interpolation between the two nearest levels.  Face/orientation convention: cube_to_dir (scene/light_utils.py:24-31).
//
// Memory-bound by design: one pixel per lane, ~100 B/pixel in, 36 B/pixel out; the 1.6 MB mip chain and the 512 KB LUT stay
// in L2.  Texel gradients are accumulated with fp32 atomics (mirror directions of neighbouring pixels hit the same texels).
#include "mrgs_internal.h"

struct EnvMips {   // by-value kernel argument
    int n;
    int res[MRGS_MAX_MIPS];
    const float* tex[MRGS_MAX_MIPS];
    float* grad[MRGS_MAX_MIPS];
    int copies[MRGS_MAX_MIPS];
    int lds_off[MRGS_MAX_MIPS];   // backward of the deferred shading: float offset of the level's LDS accumulator, -1 = global only
    float min_roughness, max_roughness;
};

struct Map {   // strided [H,W,C] view
    const float* p;
    long long sh, sw, sc;
};
struct MapOut {
    float* p;
    long long sh, sw, sc;
};

// v_rcp_f32 (1 ulp) instead of the IEEE quotient (ten instructions and two mode switches each): the lookups and the split-sum shading are
// compared to 2e-5 of a map's range and their sampling rule is a restatement to begin with (header); the filter-building kernels further
// down, which restate the reference's weights operation for operation, keep the IEEE quotient.  Every kernel of a fetch (forward, the
// backward's per-pixel launch, its scatter launch) goes through the same inline functions, so the three agree on a pixel's taps.
__device__ __forceinline__ float shade_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

struct f3 { float x, y, z; };
__device__ __forceinline__ f3 mk(float x, float y, float z) { f3 r = {x, y, z}; return r; }
__device__ __forceinline__ float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// direction -> (face, u, v) with u,v in [-1,1]; inverse of cube_to_dir (scene/light_utils.py:24-31)
struct FaceUV { int face; float u, v, inv_ma; int axis; float sgn; };
__device__ __forceinline__ FaceUV dir_to_face(f3 d)
{   // branch-free (selects): the callers inline it up to eight times per pixel
    FaceUV r;
    const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    const bool mx = ax >= ay && ax >= az, my = !mx && ay >= az;
    const float ma = mx ? ax : (my ? ay : az), mc = mx ? d.x : (my ? d.y : d.z);
    const bool pos = mc >= 0.f;
    r.axis = mx ? 0 : (my ? 1 : 2);
    r.sgn = pos ? 1.f : -1.f;
    r.inv_ma = shade_rcp(ma);
    r.face = 2 * r.axis + (pos ? 0 : 1);
    // x-major: u = -+z, v = -y; y-major: u = x, v = +-z; z-major: u = +-x, v = -y
    const float un = mx ? (pos ? -d.z : d.z) : (my ? d.x : (pos ? d.x : -d.x));
    const float vn = my ? (pos ? d.z : -d.z) : -d.y;
    r.u = un * r.inv_ma;
    r.v = vn * r.inv_ma;
    return r;
}
__device__ __forceinline__ f3 face_to_dir(int s, float x, float y)
{
    switch (s) {
    case 0: return mk(1.f, -y, -x);
    case 1: return mk(-1.f, -y, x);
    case 2: return mk(x, 1.f, y);
    case 3: return mk(x, -1.f, -y);
    case 4: return mk(x, -y, 1.f);
    default: return mk(-x, -y, -1.f);
    }
}

// the four bilinear taps of one mip level: linear texel indices (face*res*res + y*res + x) and weights
struct Taps { int idx[4]; float w[4]; float wx, wy, norm; bool corner; };
__device__ __forceinline__ int wrap_tap(int face, int x, int y, int res)
{
    if (x >= 0 && x < res && y >= 0 && y < res) return (face * res + y) * res + x;
    // texel centre of the tap in the face's plane; the overshooting coordinate is pulled just across the edge so that the
    // re-projection lands on the border texel of the adjacent face with the edge-parallel coordinate unchanged
    const float eps = 1.0f / 4096.0f;
    float u = ((float)x + 0.5f) / (float)res * 2.0f - 1.0f;
    float v = ((float)y + 0.5f) / (float)res * 2.0f - 1.0f;
    if (x < 0) u = -1.0f - eps; else if (x >= res) u = 1.0f + eps;
    if (y < 0) v = -1.0f - eps; else if (y >= res) v = 1.0f + eps;
    const FaceUV f = dir_to_face(face_to_dir(face, u, v));
    int xi = (int)floorf((f.u * 0.5f + 0.5f) * (float)res);
    int yi = (int)floorf((f.v * 0.5f + 0.5f) * (float)res);
    xi = min(max(xi, 0), res - 1);
    yi = min(max(yi, 0), res - 1);
    return (f.face * res + yi) * res + xi;
}
__device__ __forceinline__ Taps cube_taps(const FaceUV& f, int res)
{
    Taps t;
    const float fx = (f.u * 0.5f + 0.5f) * (float)res - 0.5f;
    const float fy = (f.v * 0.5f + 0.5f) * (float)res - 0.5f;
    const float x0f = floorf(fx), y0f = floorf(fy);
    const int x0 = (int)x0f, y0 = (int)y0f;
    t.wx = fx - x0f; t.wy = fy - y0f;
    const bool ox0 = x0 < 0, ox1 = x0 + 1 >= res, oy0 = y0 < 0, oy1 = y0 + 1 >= res;
    t.w[0] = (1.f - t.wx) * (1.f - t.wy); t.w[1] = t.wx * (1.f - t.wy); t.w[2] = (1.f - t.wx) * t.wy; t.w[3] = t.wx * t.wy;
    const bool c0 = ox0 && oy0, c1 = ox1 && oy0, c2 = ox0 && oy1, c3 = ox1 && oy1;
    t.corner = c0 || c1 || c2 || c3;
    t.norm = 1.0f;
    if (t.corner) {   // three faces meet: the diagonal tap does not exist
        if (c0) t.w[0] = 0.f; if (c1) t.w[1] = 0.f; if (c2) t.w[2] = 0.f; if (c3) t.w[3] = 0.f;
        const float s = shade_rcp(t.w[0] + t.w[1] + t.w[2] + t.w[3]);
        t.norm = s;
        t.w[0] *= s; t.w[1] *= s; t.w[2] *= s; t.w[3] *= s;
    }
    t.idx[0] = c0 ? 0 : wrap_tap(f.face, x0, y0, res);
    t.idx[1] = c1 ? 0 : wrap_tap(f.face, x0 + 1, y0, res);
    t.idx[2] = c2 ? 0 : wrap_tap(f.face, x0, y0 + 1, res);
    t.idx[3] = c3 ? 0 : wrap_tap(f.face, x0 + 1, y0 + 1, res);
    return t;
}

// get_mip, scene/light.py:88-96 (n = number of specular levels); dlevel = d level / d roughness
__device__ __forceinline__ float mip_level(const EnvMips& m, float r, float& dlevel)
{
    const float lo = m.min_roughness, hi = m.max_roughness;
    const float n2 = (float)(m.n - 2);
    // (the two slopes are uniform: one scalar-sourced reciprocal each instead of a quotient per lane)
    const float inv_a = shade_rcp(hi - lo), inv_b = shade_rcp(1.0f - hi);
    if (r < hi) {
        const float c = fminf(fmaxf(r, lo), hi);
        dlevel = (r >= lo && r <= hi) ? n2 * inv_a : 0.f;
        return (c - lo) * inv_a * n2;
    }
    const float c = fminf(fmaxf(r, hi), 1.0f);
    dlevel = (r >= hi && r <= 1.0f) ? inv_b : 0.f;
    return (c - hi) * inv_b + n2;
}

struct __attribute__((aligned(4))) Tex3 { float x, y, z; };     // one texel: a 12-byte load at 4-byte alignment
struct EnvSample { float L[3]; float dlev[3]; float du[2][3], dv[2][3]; int l0, l1; float f; bool lev_in; };

// per-lane level -> its resolution / texels by a select chain over the (scalar) kernel arguments: indexing the by-value struct with a
// lane-dependent level is a memory round trip of its own in front of the texel loads
__device__ __forceinline__ int level_res(const EnvMips& m, int l)
{
    int r = m.res[0];
#pragma unroll
    for (int i = 1; i < MRGS_MAX_MIPS; i++) r = (l == i) ? m.res[i] : r;
    return r;
}
__device__ __forceinline__ const float* level_tex(const EnvMips& m, int l)
{
    const float* p = m.tex[0];
#pragma unroll
    for (int i = 1; i < MRGS_MAX_MIPS; i++) p = (l == i) ? m.tex[i] : p;
    return p;
}

// the two levels a fetch blends and the blend factor (s.l0, s.l1, s.f, s.lev_in)
__device__ __forceinline__ void env_levels(const EnvMips& m, float level, bool use_mips, EnvSample& s)
{
    const int top = use_mips ? m.n - 1 : 0;
    const float lc = fminf(fmaxf(level, 0.f), (float)top);
    s.lev_in = level >= 0.f && level <= (float)top;
    s.l0 = min((int)floorf(lc), top);
    s.l1 = min(s.l0 + 1, top);
    s.f = lc - (float)s.l0;
}

// trilinear seamless cube fetch (pre-sigmoid); fills what the backward needs
__device__ __forceinline__ void env_fetch(const EnvMips& m, const FaceUV& fu, float level, bool use_mips, EnvSample& s, Taps tp[2])
{
    env_levels(m, level, use_mips, s);
    float v[2][3];
    // the taps of both levels first, then all eight 12-byte texel loads in flight together (unconditional: a missing corner tap has
    // index 0 and weight 0 -- behind a condition every load waited for the one before)
    const int res0 = level_res(m, s.l0), res1 = level_res(m, s.l1);
    tp[0] = cube_taps(fu, res0);
    tp[1] = cube_taps(fu, res1);
    const float* tex0 = level_tex(m, s.l0);
    const float* tex1 = level_tex(m, s.l1);
    float t[2][4][3];
#pragma unroll
    for (int k = 0; k < 2; k++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const Tex3 v3 = *reinterpret_cast<const Tex3*>((k == 0 ? tex0 : tex1) + (size_t)tp[k].idx[q] * 3);
            t[k][q][0] = v3.x; t[k][q][1] = v3.y; t[k][q][2] = v3.z;
        }
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int res = k == 0 ? res0 : res1;
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int c = 0; c < 3; c++) t[k][q][c] = tp[k].w[q] != 0.f ? t[k][q][c] : 0.f;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            v[k][c] = tp[k].w[0] * t[k][0][c] + tp[k].w[1] * t[k][1][c] + tp[k].w[2] * t[k][2][c] + tp[k].w[3] * t[k][3][c];
            // d value / d fx, d value / d fy (bilinear; the corner renormalisation is treated as constant)
            s.du[k][c] = ((1.f - tp[k].wy) * (t[k][1][c] - t[k][0][c]) + tp[k].wy * (t[k][3][c] - t[k][2][c])) * 0.5f * (float)res * tp[k].norm;
            s.dv[k][c] = ((1.f - tp[k].wx) * (t[k][2][c] - t[k][0][c]) + tp[k].wx * (t[k][3][c] - t[k][1][c])) * 0.5f * (float)res * tp[k].norm;
        }
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
        s.L[c] = (1.f - s.f) * v[0][c] + s.f * v[1][c];
        s.dlev[c] = (s.lev_in && s.l1 != s.l0) ? v[1][c] - v[0][c] : 0.f;
    }
}

__device__ __forceinline__ float sigmoidf(float x) { return shade_rcp(1.0f + __expf(-x)); }

// wave64 sum (DPP), result valid in every lane after the final readlane
__device__ __forceinline__ float wave_sum_all(float v)
{
#define DPP_ADD(CTRL, RM) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, RM, 0xf, false))
    DPP_ADD(0xb1, 0xf); DPP_ADD(0x4e, 0xf); DPP_ADD(0x124, 0xf); DPP_ADD(0x128, 0xf); DPP_ADD(0x142, 0xa); DPP_ADD(0x143, 0xc);
#undef DPP_ADD
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Add v[0..2] to the texel at `addr` (nullptr / all-zero v = nothing to add).  Neighbouring pixels mirror into the same few
// texels (every lane of a wave into ONE texel of the coarse mip levels), and same-address atomics of one instruction
// serialise in L2: combine the lanes that share a texel inside the wave first (leader election by ballot, DPP sums), for at
// most MAX_ROUNDS distinct texels, and let the rest fall through to plain atomics.  Convergent: every lane of the wave calls it.
__device__ __forceinline__ void texel_scatter(float* addr, const float v[3])
{
    const int MAX_ROUNDS = 4;
    bool pending = addr != nullptr && (v[0] != 0.f || v[1] != 0.f || v[2] != 0.f);
    const unsigned lo = (unsigned)(uintptr_t)addr, hi = (unsigned)((uintptr_t)addr >> 32);
    for (int r = 0; r < MAX_ROUNDS; r++) {
        const unsigned long long pm = __ballot(pending);
        if (pm == 0ull) return;
        const int leader = __builtin_ctzll(pm);
        const unsigned klo = __builtin_amdgcn_readlane(lo, leader), khi = __builtin_amdgcn_readlane(hi, leader);
        const bool match = pending && lo == klo && hi == khi;
        const float s0 = wave_sum_all(match ? v[0] : 0.f), s1 = wave_sum_all(match ? v[1] : 0.f), s2 = wave_sum_all(match ? v[2] : 0.f);
        if ((int)(threadIdx.x & 63) == leader) {
            atomicAdd(addr + 0, s0);
            atomicAdd(addr + 1, s1);
            atomicAdd(addr + 2, s2);
        }
        pending = pending && !match;
    }
    if (pending) {
        atomicAdd(addr + 0, v[0]);
        atomicAdd(addr + 1, v[1]);
        atomicAdd(addr + 2, v[2]);
    }
}

// gradient of the fetch: scatter to the texels (global atomics, lanes sharing a texel combined by texel_scatter; SCATTER = false leaves the
// texels to the caller), return d/d dir and d/d level
template <bool SCATTER = true>
__device__ __forceinline__ void env_fetch_bwd(const EnvMips& m, const FaceUV& fu, f3 dir, const EnvSample& s, const Taps tp[2],
                                              const float gL[3], f3& g_dir, float& g_level)
{
    float gu = 0.f, gv = 0.f;
    g_level = 0.f;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const float wk = k == 0 ? 1.f - s.f : s.f;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float g = gL[c] * wk;
            gu += g * s.du[k][c];
            gv += g * s.dv[k][c];
        }
        if constexpr (SCATTER) {
            const int lk = k == 0 ? s.l0 : s.l1;
            float* gt = m.grad[lk];
            if (gt != nullptr) {   // privatised copy of this workgroup
                const unsigned wg = blockIdx.x + blockIdx.y * gridDim.x;
                gt += (size_t)(wg % (unsigned)m.copies[lk]) * (size_t)(6 * m.res[lk] * m.res[lk] * 3);
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float wq = wk * tp[k].w[q];
                const float vq[3] = {gL[0] * wq, gL[1] * wq, gL[2] * wq};
                texel_scatter((gt != nullptr && wq != 0.f) ? gt + (size_t)tp[k].idx[q] * 3 : nullptr, vq);     // wave-convergent
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 3; c++) g_level += gL[c] * s.dlev[c];
    // (u, v) = (su * a, sv * b) / |major|  ->  d/d dir
    float ga = gu * fu.inv_ma, gb = gv * fu.inv_ma;                       // w.r.t. the signed minor components
    const float gm = -(gu * fu.u + gv * fu.v) * fu.sgn * fu.inv_ma;        // w.r.t. the major component
    g_dir = mk(0.f, 0.f, 0.f);
    switch (fu.face) {
    case 0: g_dir = mk(gm, -gb, -ga); break;   // u = -z/|x|, v = -y/|x|
    case 1: g_dir = mk(gm, -gb, ga); break;    // u = +z/|x|, v = -y/|x|
    case 2: g_dir = mk(ga, gm, gb); break;     // u = +x/|y|, v = +z/|y|
    case 3: g_dir = mk(ga, gm, -gb); break;    // u = +x/|y|, v = -z/|y|
    case 4: g_dir = mk(ga, -gb, gm); break;    // u = +x/|z|, v = -y/|z|
    default: g_dir = mk(-ga, -gb, gm); break;  // u = -x/|z|, v = -y/|z|
    }
    (void)dir;
}

// ---- texel-gradient accumulation of the deferred shading backward ----------------------------------------------------------
// Measured on MI355X (tools/ubench/lds_atomic, tools/shade_stats.py): a scattered global float atomic costs the chip
// ~1/25 ns, an LDS float atomic (ds_add_f32) ~3 cycles PER ACTIVE LANE of its CU's LDS (200 G/s chip-wide; integer compare-and-swap is
// ten times faster), a VALU instruction 4 cycles per wave on one of four SIMDs -- so lanes are merged on the VALU first, then in LDS, and
// only what is left leaves the CU:
//   1. run_merge: neighbouring pixels mirror into the same texels in RUNS along a row (a 64 x 1 wave footprint touches ~9 distinct
//      texel footprints of a 64 x 64 level on the bench scene, 49 lanes using it): a segmented sum over runs of equal key inside each
//      16-lane row (four row_shr DPP steps) leaves one lane per run;
//   2. levels that fit the workgroup's LDS (coarsest first, MRGS_SHADE_LDS_FLOATS) accumulate in a dense LDS copy;
//   3. the finer levels go through a 4 096-entry LDS hash table (key = level | texel, CAS insert with linear probing, three float
//      accumulators per entry): a 64 x 12-pixel tile touches ~35 footprints of the 64 x 64 level, 590 pixels using it.  The table is
//      flushed with global atomics when half full (checked between tiles) and at the end; a lane that finds no slot in 8 probes adds
//      straight to global memory.
#define MRGS_SHADE_HASH_BITS 12
#define MRGS_SHADE_HASH_SIZE (1 << MRGS_SHADE_HASH_BITS)
#define MRGS_SHADE_KEY_NONE 0xffffffffu

// Sum v over the run of consecutive lanes (inside a 16-lane row) that hold the same key; true in the lane that ends a run, which then
// holds the run's total.  Hillis-Steele segmented scan: (v, head) o (v', head') = head' ? (v', 1) : (v + v', head).
__device__ __forceinline__ bool run_merge(unsigned key, float v[3])
{
#define SHR(x, n, old) __builtin_amdgcn_update_dpp((int)(old), (int)(x), 0x110 + (n), 0xf, 0xf, false)
    const unsigned prev = (unsigned)SHR(key, 1, ~key);                    // lane 0 of a row: no source -> old = ~key != key
    int head = prev != key;
    const int next_head = __builtin_amdgcn_update_dpp(1, head, 0x101, 0xf, 0xf, false);     // row_shl:1; lane 15 of a row: 1
#pragma unroll
    for (int d = 1; d <= 8; d <<= 1) {
        const float m = head ? 0.f : 1.f;
        float p0, p1, p2;
        int ph;
        if (d == 1) { p0 = __int_as_float(SHR(__float_as_int(v[0]), 1, 0)); p1 = __int_as_float(SHR(__float_as_int(v[1]), 1, 0)); p2 = __int_as_float(SHR(__float_as_int(v[2]), 1, 0)); ph = SHR(head, 1, 1); }
        else if (d == 2) { p0 = __int_as_float(SHR(__float_as_int(v[0]), 2, 0)); p1 = __int_as_float(SHR(__float_as_int(v[1]), 2, 0)); p2 = __int_as_float(SHR(__float_as_int(v[2]), 2, 0)); ph = SHR(head, 2, 1); }
        else if (d == 4) { p0 = __int_as_float(SHR(__float_as_int(v[0]), 4, 0)); p1 = __int_as_float(SHR(__float_as_int(v[1]), 4, 0)); p2 = __int_as_float(SHR(__float_as_int(v[2]), 4, 0)); ph = SHR(head, 4, 1); }
        else { p0 = __int_as_float(SHR(__float_as_int(v[0]), 8, 0)); p1 = __int_as_float(SHR(__float_as_int(v[1]), 8, 0)); p2 = __int_as_float(SHR(__float_as_int(v[2]), 8, 0)); ph = SHR(head, 8, 1); }
        v[0] = fmaf(p0, m, v[0]); v[1] = fmaf(p1, m, v[1]); v[2] = fmaf(p2, m, v[2]);
        head |= ph;
    }
#undef SHR
    return next_head != 0;
}

struct ShadeAcc {          // the workgroup's LDS accumulators
    float* dense;          // coarse levels, EnvMips::lds_off
    unsigned* keys;        // hash table of the finer levels
    float* vals;
    unsigned* count;       // entries taken so far (never reset: the kernel remembers its value at the last flush)
};

// one merged contribution (the lane ends a run; key = level << 24 | texel) into the workgroup's accumulators; loff = the level's
// offset in the dense LDS copy or < 0
__device__ __forceinline__ void acc_add(const EnvMips& m, const ShadeAcc& A, int lk, int loff, int idx, const float v[3])
{
    if (loff >= 0) {
        float* a = A.dense + loff + idx * 3;
#ifdef MRGS_X_INT_ATOMICS   // developer timing build (results are wrong): what the accumulation would cost with integer LDS atomics
        atomicAdd((int*)a, (int)(v[0] * 65536.f)); atomicAdd((int*)a + 1, (int)(v[1] * 65536.f)); atomicAdd((int*)a + 2, (int)(v[2] * 65536.f));
#else
        atomicAdd(a, v[0]); atomicAdd(a + 1, v[1]); atomicAdd(a + 2, v[2]);
#endif
        return;
    }
    const unsigned key = ((unsigned)lk << 24) | (unsigned)idx;
    unsigned h = (key * 2654435761u) >> (32 - MRGS_SHADE_HASH_BITS);
    int slot = -1;
    for (int p = 0; p < 8; ++p) {
        const unsigned old = atomicCAS(&A.keys[h], MRGS_SHADE_KEY_NONE, key);
        if (old == MRGS_SHADE_KEY_NONE) atomicAdd(A.count, 1u);
        if (old == MRGS_SHADE_KEY_NONE || old == key) { slot = (int)h; break; }
        h = (h + 1) & (MRGS_SHADE_HASH_SIZE - 1);
    }
    if (slot >= 0) {
        float* a = A.vals + slot * 3;
#ifdef MRGS_X_INT_ATOMICS
        atomicAdd((int*)a, (int)(v[0] * 65536.f)); atomicAdd((int*)a + 1, (int)(v[1] * 65536.f)); atomicAdd((int*)a + 2, (int)(v[2] * 65536.f));
#else
        atomicAdd(a, v[0]); atomicAdd(a + 1, v[1]); atomicAdd(a + 2, v[2]);
#endif
    } else {
        float* g = m.grad[lk] + (size_t)(blockIdx.x % (unsigned)m.copies[lk]) * (size_t)(6 * m.res[lk] * m.res[lk] * 3) + (size_t)idx * 3;
        atomicAdd(g, v[0]); atomicAdd(g + 1, v[1]); atomicAdd(g + 2, v[2]);
    }
}

// the eight taps of one pixel's fetch: merged along the row, then accumulated (wave-convergent: every lane of the wave calls it).
// grad_mask: bit l = level l takes gradients (scalar, built once per kernel: the per-lane level must not index the argument struct here)
__device__ __forceinline__ void env_scatter_tile(const EnvMips& m, unsigned grad_mask, const ShadeAcc& A, const EnvSample& s, const Taps tp[2],
                                                 const float gL[3])
{
    const bool any = gL[0] != 0.f || gL[1] != 0.f || gL[2] != 0.f;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const float wk = k == 0 ? 1.f - s.f : s.f;
        const int lk = k == 0 ? s.l0 : s.l1;
        const bool lev = any && wk != 0.f && ((grad_mask >> lk) & 1u);
        int loff = m.lds_off[0];
#pragma unroll
        for (int i = 1; i < MRGS_MAX_MIPS; i++) loff = (lk == i) ? m.lds_off[i] : loff;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float wq = wk * tp[k].w[q];
            float v[3] = {gL[0] * wq, gL[1] * wq, gL[2] * wq};
            const bool live = lev && tp[k].w[q] != 0.f;
            const unsigned key = live ? (((unsigned)lk << 24) | (unsigned)tp[k].idx[q]) : MRGS_SHADE_KEY_NONE;
            if (!live) { v[0] = 0.f; v[1] = 0.f; v[2] = 0.f; }
#ifdef MRGS_X_NO_MERGE
            const bool last = true;
#else
            const bool last = run_merge(key, v);
#endif
#ifdef MRGS_X_NO_ACC       // developer timing build: everything but the accumulation (results are wrong)
            if (last && key == 0x12345678u && v[0] == 1e30f) acc_add(m, A, lk, loff, tp[k].idx[q], v);
#else
            if (last && key != MRGS_SHADE_KEY_NONE) acc_add(m, A, lk, loff, tp[k].idx[q], v);
#endif
        }
    }
}

// every entry of the hash table -> global memory, table emptied (all threads of the workgroup; barriers are the caller's)
__device__ __forceinline__ void acc_flush_hash(const EnvMips& m, const ShadeAcc& A, int nthreads)
{
    for (int i = threadIdx.x; i < MRGS_SHADE_HASH_SIZE; i += nthreads) {
        const unsigned key = A.keys[i];
        if (key == MRGS_SHADE_KEY_NONE) continue;
        const int lk = (int)(key >> 24), idx = (int)(key & 0xffffffu);
        float* g = m.grad[lk] + (size_t)(blockIdx.x % (unsigned)m.copies[lk]) * (size_t)(6 * m.res[lk] * m.res[lk] * 3) + (size_t)idx * 3;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float v = A.vals[i * 3 + c];
            if (v != 0.f) atomicAdd(g + c, v);
            A.vals[i * 3 + c] = 0.f;
        }
        A.keys[i] = MRGS_SHADE_KEY_NONE;
    }
}

// ---- standalone environment lookup ---------------------------------------------------------------------
__global__ void __launch_bounds__(256) envmap_lookup_fwd_kernel(EnvMips m, long long N, const float* __restrict__ dirs,
                                                                const float* __restrict__ roughness, float* __restrict__ out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const f3 d = mk(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]);
    const FaceUV fu = dir_to_face(d);
    float dl;
    const bool use_mips = roughness != nullptr;
    const float level = use_mips ? mip_level(m, roughness[i], dl) : 0.f;
    EnvSample s;
    Taps tp[2];
    env_fetch(m, fu, level, use_mips, s, tp);
#pragma unroll
    for (int c = 0; c < 3; c++) out[3 * i + c] = sigmoidf(s.L[c]);
}

__global__ void __launch_bounds__(256) envmap_lookup_bwd_kernel(EnvMips m, long long N, const float* __restrict__ dirs,
                                                                const float* __restrict__ roughness, const float* __restrict__ g_out,
                                                                float* __restrict__ g_dirs, float* __restrict__ g_rough)
{
    const long long i_ = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i_ < N;              // out-of-range lanes stay alive (the texel scatter is wave-convergent)
    const long long i = valid ? i_ : N - 1;
    const f3 d = mk(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]);
    const FaceUV fu = dir_to_face(d);
    float dl = 0.f;
    const bool use_mips = roughness != nullptr;
    const float level = use_mips ? mip_level(m, roughness[i], dl) : 0.f;
    EnvSample s;
    Taps tp[2];
    env_fetch(m, fu, level, use_mips, s, tp);
    float gL[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float y = sigmoidf(s.L[c]);
        gL[c] = valid ? g_out[3 * i + c] * y * (1.f - y) : 0.f;
    }
    f3 gd;
    float glev;
    env_fetch_bwd<true>(m, fu, d, s, tp, gL, gd, glev);
    if (valid && g_dirs) { g_dirs[3 * i] = gd.x; g_dirs[3 * i + 1] = gd.y; g_dirs[3 * i + 2] = gd.z; }
    if (valid && g_rough) g_rough[i] = use_mips ? glev * dl : 0.f;
}

// ---- fused deferred specular shading ---------------------------------------------------------------------
struct ShadeCam { float Kinv[9]; const float* R; const float* T; };   // R: Camera.R (c2w rotation, [3,3]); T: w2c translation

struct ShadePix {
    f3 wo, n, r, rn;
    float ndv, rlen, u, v, rough, refl, alpha;
    float albedo[3], fg[2], dfg_du[2], dfg_dv[2];
    bool u_in, v_in;
};

__device__ __forceinline__ void lut_fetch(const float* __restrict__ lut, int lres, float u, float v, float fg[2], float du[2], float dv[2])
{
    const float fx = u * (float)lres - 0.5f, fy = v * (float)lres - 0.5f;
    const float x0f = floorf(fx), y0f = floorf(fy);
    const float wx = fx - x0f, wy = fy - y0f;
    const int x0 = min(max((int)x0f, 0), lres - 1), x1 = min(max((int)x0f + 1, 0), lres - 1);
    const int y0 = min(max((int)y0f, 0), lres - 1), y1 = min(max((int)y0f + 1, 0), lres - 1);
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const float t00 = lut[((size_t)y0 * lres + x0) * 2 + c], t10 = lut[((size_t)y0 * lres + x1) * 2 + c];
        const float t01 = lut[((size_t)y1 * lres + x0) * 2 + c], t11 = lut[((size_t)y1 * lres + x1) * 2 + c];
        fg[c] = (1.f - wy) * ((1.f - wx) * t00 + wx * t10) + wy * ((1.f - wx) * t01 + wx * t11);
        du[c] = ((1.f - wy) * (t10 - t00) + wy * (t11 - t01)) * (float)lres;
        dv[c] = ((1.f - wx) * (t01 - t00) + wx * (t11 - t10)) * (float)lres;
    }
}

// the camera of a launch in scalar registers: K^-1, Camera.R (c2w rotation), the w2c translation and the camera centre -R T
struct ShadeCamS { float Kinv[9], R[9], t[3], ro[3]; };
__device__ __forceinline__ float sreg(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ ShadeCamS shade_cam_load(const ShadeCam& cam)
{
    ShadeCamS c;
#pragma unroll
    for (int i = 0; i < 9; i++) { c.Kinv[i] = cam.Kinv[i]; c.R[i] = sreg(cam.R[i]); }
#pragma unroll
    for (int i = 0; i < 3; i++) c.t[i] = sreg(cam.T[i]);
#pragma unroll
    for (int i = 0; i < 3; i++) c.ro[i] = -(c.R[3 * i] * c.t[0] + c.R[3 * i + 1] * c.t[1] + c.R[3 * i + 2] * c.t[2]);
    return c;
}

// the per-pixel inputs of the shading, fetched in one go (nine independent loads in flight)
struct ShadeIn { float n[3], rough, refl, alpha, albedo[3]; };
__device__ __forceinline__ ShadeIn shade_load(int x, int y, const Map& albedo, const Map& normal, const Map& alpha, const Map& refl, const Map& rough)
{
    ShadeIn in;
    const long long o = (long long)y * normal.sh + (long long)x * normal.sw;
    in.n[0] = normal.p[o]; in.n[1] = normal.p[o + normal.sc]; in.n[2] = normal.p[o + 2 * normal.sc];
    in.rough = rough.p[(long long)y * rough.sh + (long long)x * rough.sw];
    in.refl = refl.p[(long long)y * refl.sh + (long long)x * refl.sw];
    in.alpha = alpha.p[(long long)y * alpha.sh + (long long)x * alpha.sw];
    const long long oa = (long long)y * albedo.sh + (long long)x * albedo.sw;
#pragma unroll
    for (int c = 0; c < 3; c++) in.albedo[c] = albedo.p[oa + c * albedo.sc];
    return in;
}

// w_o of a pixel: sample_camera_rays (utils/refl_utils.py:54-73), pixel centres at integer coordinates
__device__ __forceinline__ f3 shade_view_dir(const ShadeCamS& cam, int x, int y)
{
    const float fx_ = (float)x, fy_ = (float)y;
    const f3 pc = mk(cam.Kinv[0] * fx_ + cam.Kinv[1] * fy_ + cam.Kinv[2], cam.Kinv[3] * fx_ + cam.Kinv[4] * fy_ + cam.Kinv[5],
                     cam.Kinv[6] * fx_ + cam.Kinv[7] * fy_ + cam.Kinv[8]);
    // Camera.R is stored transposed (c2w); the reference re-transposes it: world = (pc - T) @ R^T^T ... = c2w * (pc - T)
    const float* R = cam.R;
    const f3 q = mk(pc.x - cam.t[0], pc.y - cam.t[1], pc.z - cam.t[2]);
    const f3 pw = mk(R[0] * q.x + R[1] * q.y + R[2] * q.z, R[3] * q.x + R[4] * q.y + R[5] * q.z, R[6] * q.x + R[7] * q.y + R[8] * q.z);
    f3 rd = mk(pw.x - cam.ro[0], pw.y - cam.ro[1], pw.z - cam.ro[2]);
    const float inv = __builtin_amdgcn_rsqf(dot3(rd, rd));
    rd = mk(rd.x * inv, rd.y * inv, rd.z * inv);
    return mk(-rd.x, -rd.y, -rd.z);
}

// the unit mirror direction of a pixel (what shade_setup leaves in ShadePix::rn: the scatter launch of the backward forms it again)
__device__ __forceinline__ f3 shade_mirror_dir(const ShadeCamS& cam, int x, int y, f3 n)
{
    const f3 wo = shade_view_dir(cam, x, y);
    const float ndv = dot3(wo, n);                              // reflection(), :95-98
    const f3 r = mk(2.f * n.x * ndv - wo.x, 2.f * n.y * ndv - wo.y, 2.f * n.z * ndv - wo.z);
    const float rlen = fmaxf(__builtin_amdgcn_sqrtf(dot3(r, r)), 1e-20f);        // safe_normalize
    const float irl = shade_rcp(rlen);
    return mk(r.x * irl, r.y * irl, r.z * irl);
}

__device__ __forceinline__ void shade_setup(const ShadeCamS& cam, int x, int y, const ShadeIn& in, const float* __restrict__ lut, int lres, ShadePix& p)
{
    p.wo = shade_view_dir(cam, x, y);
    p.n = mk(in.n[0], in.n[1], in.n[2]);
    p.ndv = dot3(p.wo, p.n);                                    // reflection(), :95-98
    p.r = mk(2.f * p.n.x * p.ndv - p.wo.x, 2.f * p.n.y * p.ndv - p.wo.y, 2.f * p.n.z * p.ndv - p.wo.z);
    p.rlen = fmaxf(__builtin_amdgcn_sqrtf(dot3(p.r, p.r)), 1e-20f);              // safe_normalize
    const float irl = shade_rcp(p.rlen);
    p.rn = mk(p.r.x * irl, p.r.y * irl, p.r.z * irl);
    p.rough = in.rough;
    p.refl = in.refl;
    p.alpha = in.alpha;
#pragma unroll
    for (int c = 0; c < 3; c++) p.albedo[c] = in.albedo[c];
    p.u_in = p.ndv >= 0.f && p.ndv <= 1.f;
    p.v_in = p.rough >= 0.f && p.rough <= 1.f;
    p.u = fminf(fmaxf(p.ndv, 0.f), 1.f);
    p.v = fminf(fmaxf(p.rough, 0.f), 1.f);
    lut_fetch(lut, lres, p.u, p.v, p.fg, p.dfg_du, p.dfg_dv);
}

__device__ __forceinline__ float shade_lin2srgb(float x)        // linear_to_srgb (utils/general_utils / mrgs_maps.hip: the same expression)
{
    const float eps = 1.1920928955078125e-07f;
    return x <= 0.0031308f ? (323.0f / 25.0f) * x : (211.0f * powf(fmaxf(x, eps), 5.0f / 12.0f) - 11.0f) / 200.0f;
}

__device__ __forceinline__ float shade_lin2srgb_grad(float x)   // d linear_to_srgb / dx (mrgs_maps.hip: lin2srgb_grad, the same expression)
{
    const float eps = 1.1920928955078125e-07f;
    if (x <= 0.0031308f) return 323.0f / 25.0f;
    return x >= eps ? (211.0f / 200.0f) * (5.0f / 12.0f) * powf(x, -7.0f / 12.0f) : 0.0f;
}

__global__ void __launch_bounds__(256) shade_specular_fwd_kernel(EnvMips m, ShadeCam cam, int H, int W, Map albedo, Map normal, Map alpha,
                                                                 Map refl, Map rough, const float* __restrict__ lut, int lres,
                                                                 float* __restrict__ specular /*[3,H,W]*/, float* __restrict__ direct /*[3,H,W]*/,
                                                                 float* __restrict__ weight /*[H,W,3]*/,
                                                                 // render_surfel's compositing in the same pass (render != nullptr; mrgs_surfel_composite_forward)
                                                                 const float* __restrict__ base /*[3,H,W]*/, const float* __restrict__ bg, int srgb,
                                                                 float* __restrict__ render /*[3,H,W]*/, float* __restrict__ diffuse /*[3,H,W]*/,
                                                                 float* __restrict__ zero_fill, long long zero_floats)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    {   // side job for the backward of this very render: clear the texel-gradient buffers its shading backward accumulates into (a fill
        // launch of its own otherwise: the backward's first kernel is the one that accumulates)
        const long long tid = ((long long)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x, nthr = (long long)gridDim.x * gridDim.y * 256;
        for (long long i = tid; i < zero_floats; i += nthr) zero_fill[i] = 0.0f;
    }
    if (x >= W || y >= H) return;
    const ShadeCamS cs = shade_cam_load(cam);
    const ShadeIn in = shade_load(x, y, albedo, normal, alpha, refl, rough);
    ShadePix p;
    shade_setup(cs, x, y, in, lut, lres, p);
    const FaceUV fu = dir_to_face(p.rn);
    float dl;
    const float level = mip_level(m, p.rough, dl);
    EnvSample s;
    Taps tp[2];
    env_fetch(m, fu, level, true, s, tp);
    const size_t HW = (size_t)H * W, pix = (size_t)y * W + x;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float light = sigmoidf(s.L[c]);
        const float wgt = (0.04f * (1.f - p.refl) + p.albedo[c] * p.refl) * p.fg[0] + p.fg[1];
        direct[c * HW + pix] = light;
        weight[pix * 3 + c] = wgt;
        const float spec = light * p.alpha * wgt;
        specular[c * HW + pix] = spec;
        if (render != nullptr) {     // diffuse = (1 - refl) base, render = [srgb](diffuse + specular) + bg (1 - alpha)  (gaussian_renderer/__init__.py:436-445)
            const float d = (1.0f - p.refl) * base[c * HW + pix];
            float v = d + spec;
            if (srgb) v = shade_lin2srgb(v);
            render[c * HW + pix] = v + bg[c] * (1.0f - p.alpha);
            diffuse[c * HW + pix] = d;
        }
    }
}

// ---- backward of the deferred shading -----------------------------------------------------------------------------------------
// Persistent workgroups (one per CU, 12 waves): each keeps an LDS copy of the texel gradients of the coarse mip levels
// (<= MRGS_SHADE_LDS_FLOATS floats: 32x32 and 16x16 cubemap levels = 90 KB) and a hash table for the finer ones (64 KB), walks
// 64x12-pixel tiles of the image and flushes both at the end (see "texel-gradient accumulation" above).  COMPOSITE: render_surfel's
// compositing backward (gaussian_renderer/__init__.py:436-445) in front of the shading's, in the same lane: its shares of g_specular,
// g_refl and g_alpha never leave registers (rounds 2-5 ran it as a launch of its own, 10.7 us for 36 bytes a pixel each way).
// Round 6 measured the alternative the register count suggests (168 VGPRs, a dozen spilled, three waves per SIMD) -- a per-pixel launch
// without LDS (134 VGPRs, no spill) that leaves the pixel's d loss / d sample in a scratch map, and a scatter launch that recomputes the
// taps (72 VGPRs): 35 + 45 us against this kernel's 69.  The scatter launch alone is 11 us of loop, loads, accumulator set-up and flush,
// + 10 of recomputed taps, + 4 of run merging, + 20 of LDS float atomics and hash probes (8 with integer atomics): the persistent
// form hides the accumulation's latency behind the fetch of the next pixel, the split form pays both in turn.  Dropped.
#define MRGS_SHADE_LDS_FLOATS 23552
#ifndef MRGS_SHADE_FUSED_THREADS
#define MRGS_SHADE_FUSED_THREADS 768
#endif
template <bool COMPOSITE>
__global__ void __launch_bounds__(MRGS_SHADE_FUSED_THREADS) shade_fused_bwd_kernel(
    EnvMips m, ShadeCam cam, int H, int W, Map albedo, Map normal, Map alpha, Map refl, Map rough, const float* __restrict__ lut, int lres,
    const float* __restrict__ g_specular, const float* __restrict__ g_direct, const float* __restrict__ g_weight,
    float* __restrict__ g_albedo /*[H,W,3]*/, float* __restrict__ g_normal /*[H,W,3]*/, float* __restrict__ g_alpha /*[H,W]*/,
    float* __restrict__ g_refl /*[H,W]*/, float* __restrict__ g_rough /*[H,W]*/, int tiles_x, int ntiles, int lds_floats,
    float* __restrict__ g_features, const float* __restrict__ g_refl_composite, const float* __restrict__ g_alpha_composite,
    int srgb, const float* __restrict__ base, const float* __restrict__ spec_fwd, const float* __restrict__ bg,
    const float* __restrict__ g_render, const float* __restrict__ g_diffuse, float* __restrict__ g_base)
{
    __shared__ float s_grad[MRGS_SHADE_LDS_FLOATS];
    __shared__ unsigned s_keys[MRGS_SHADE_HASH_SIZE];
    __shared__ float s_vals[MRGS_SHADE_HASH_SIZE * 3];
    __shared__ unsigned s_count;
    for (int i = threadIdx.x; i < lds_floats; i += MRGS_SHADE_FUSED_THREADS) s_grad[i] = 0.f;
    for (int i = threadIdx.x; i < MRGS_SHADE_HASH_SIZE; i += MRGS_SHADE_FUSED_THREADS) { s_keys[i] = MRGS_SHADE_KEY_NONE; s_vals[3 * i] = 0.f; s_vals[3 * i + 1] = 0.f; s_vals[3 * i + 2] = 0.f; }
    if (threadIdx.x == 0) s_count = 0u;
    __syncthreads();
    const ShadeAcc A = {s_grad, s_keys, s_vals, &s_count};
    const size_t HW = (size_t)H * W;
    unsigned flushed_at = 0u;                              // value of the (never reset) entry counter at the last flush
    int since_check = 0;
    const ShadeCamS cs = shade_cam_load(cam);
    unsigned grad_mask = 0u;
#pragma unroll
    for (int i = 0; i < MRGS_MAX_MIPS; i++) grad_mask |= (i < m.n && m.grad[i] != nullptr) ? (1u << i) : 0u;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int x_ = (t % tiles_x) * 64 + (threadIdx.x & 63), y_ = (t / tiles_x) * (MRGS_SHADE_FUSED_THREADS / 64) + (threadIdx.x >> 6);
        const bool valid = x_ < W && y_ < H;    // out-of-image lanes stay alive (the texel scatter is wave-convergent)
        const int x = min(x_, W - 1), y = min(y_, H - 1);
        const size_t pix = (size_t)y * W + x;
        // every input of the pixel first: the maps and the upstream gradients are independent loads
        const ShadeIn in = shade_load(x, y, albedo, normal, alpha, refl, rough);
        float gs_[3], gd_[3], gw_[3];
        float g_refl_c = 0.f, g_alpha_c = 0.f;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            gs_[c] = (valid && g_specular) ? g_specular[c * HW + pix] : 0.f;
            gd_[c] = (valid && g_direct) ? g_direct[c * HW + pix] : 0.f;
            gw_[c] = (valid && g_weight) ? g_weight[pix * 3 + c] : 0.f;
        }
        if constexpr (COMPOSITE) {
            // render = [srgb](k base + specular) + bg (1 - alpha), diffuse = k base, k = 1 - refl   (gaussian_renderer/__init__.py:436-445)
            const float k = 1.0f - in.refl;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float b = base[c * HW + pix];
                const float gR = (valid && g_render != nullptr) ? g_render[c * HW + pix] : 0.0f;
                g_alpha_c -= bg[c] * gR;
                float gl = gR;
                if (srgb) gl *= shade_lin2srgb_grad(k * b + spec_fwd[c * HW + pix]);
                const float gd = gl + ((valid && g_diffuse != nullptr) ? g_diffuse[c * HW + pix] : 0.0f);
                if (valid) g_base[c * HW + pix] = k * gd;
                g_refl_c -= b * gd;
                gs_[c] += gl;
            }
        } else if (g_features != nullptr) {
            g_refl_c = g_refl_composite[pix];
            g_alpha_c = g_alpha_composite[pix];
        }
        ShadePix p;
        shade_setup(cs, x, y, in, lut, lres, p);
        const FaceUV fu = dir_to_face(p.rn);
        float dl;
        const float level = mip_level(m, p.rough, dl);
        EnvSample s;
        Taps tp[2];
        env_fetch(m, fu, level, true, s, tp);
        float gL[3], ga = 0.f, gm = 0.f, gfg0 = 0.f, gfg1 = 0.f, galb[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float light = sigmoidf(s.L[c]);
            const float basec = 0.04f * (1.f - p.refl) + p.albedo[c] * p.refl;
            const float wgt = basec * p.fg[0] + p.fg[1];
            const float gs = gs_[c];
            const float gd = gd_[c] + gs * p.alpha * wgt;
            const float gw = gw_[c] + gs * light * p.alpha;
            ga += gs * light * wgt;
            gL[c] = gd * light * (1.f - light);
            galb[c] = gw * p.refl * p.fg[0];
            gm += gw * (p.albedo[c] - 0.04f) * p.fg[0];
            gfg0 += gw * basec;
            gfg1 += gw;
        }
        f3 g_rn;
        float g_level;
        env_fetch_bwd<false>(m, fu, p.rn, s, tp, gL, g_rn, g_level);
        // safe_normalize backward (the 1e-20 clamp never binds for finite normals)
        const float rg = dot3(p.rn, g_rn);
        const float irl = shade_rcp(p.rlen);
        const f3 g_r = mk((g_rn.x - p.rn.x * rg) * irl, (g_rn.y - p.rn.y * rg) * irl, (g_rn.z - p.rn.z * rg) * irl);
        // r = 2 n (n.wo) - wo ; NdotV = n.wo feeds the LUT's u coordinate
        const float g_ndv = p.u_in ? gfg0 * p.dfg_du[0] + gfg1 * p.dfg_du[1] : 0.f;
        const float grn = dot3(g_r, p.n);
        const f3 gn = mk(2.f * p.ndv * g_r.x + (2.f * grn + g_ndv) * p.wo.x, 2.f * p.ndv * g_r.y + (2.f * grn + g_ndv) * p.wo.y,
                         2.f * p.ndv * g_r.z + (2.f * grn + g_ndv) * p.wo.z);
        const float g_rough_v = (p.v_in ? gfg0 * p.dfg_dv[0] + gfg1 * p.dfg_dv[1] : 0.f) + g_level * dl;
        if (valid) {
            g_normal[pix * 3] = gn.x; g_normal[pix * 3 + 1] = gn.y; g_normal[pix * 3 + 2] = gn.z;
            if (g_features != nullptr) {
                g_alpha[pix] = g_alpha_c + ga;
                g_features[pix] = g_refl_c + gm;
                g_features[HW + pix] = g_rough_v;
#pragma unroll
                for (int c = 0; c < 3; c++) { g_features[(2 + c) * HW + pix] = galb[c]; g_features[(5 + c) * HW + pix] = 0.0f; }
            } else {
#pragma unroll
                for (int c = 0; c < 3; c++) g_albedo[pix * 3 + c] = galb[c];
                g_alpha[pix] = ga;
                g_refl[pix] = gm;
                g_rough[pix] = g_rough_v;
            }
        }
        if (!valid) { gL[0] = 0.f; gL[1] = 0.f; gL[2] = 0.f; }
        // (a wave of pixels that saw no environment light -- background: alpha = 0 makes gL an exact zero -- has nothing to add)
        if (grad_mask != 0u && __builtin_amdgcn_ballot_w64(gL[0] != 0.f || gL[1] != 0.f || gL[2] != 0.f) != 0ull)
            env_scatter_tile(m, grad_mask, A, s, tp, gL);
        // between tiles: a hash table more than half full goes out
        if (++since_check < 4) continue;
        since_check = 0;
        __syncthreads();
        const unsigned taken = s_count;                    // same value in every thread: entries are only taken before the barrier above
        if (taken - flushed_at > MRGS_SHADE_HASH_SIZE / 2) {
            acc_flush_hash(m, A, MRGS_SHADE_FUSED_THREADS);
            flushed_at = taken;
        }
        __syncthreads();
    }
    // flush: the hash table, then the LDS-resident levels into one of the level's global copies (untouched texels are skipped)
    __syncthreads();                 // every wave has left the tile loop (the loop only meets at a barrier every fourth tile)
    acc_flush_hash(m, A, MRGS_SHADE_FUSED_THREADS);
    for (int l = 0; l < m.n; l++) {
        if (m.lds_off[l] < 0 || m.grad[l] == nullptr) continue;
        const int n = 6 * m.res[l] * m.res[l] * 3;
        float* dst = m.grad[l] + (size_t)(blockIdx.x % (unsigned)m.copies[l]) * (size_t)n;
        for (int i = threadIdx.x; i < n; i += MRGS_SHADE_FUSED_THREADS) {
            const float v = s_grad[m.lds_off[l] + i];
            if (v != 0.f) atomicAdd(dst + i, v);
        }
    }
}

// ---- environment prefilter (EnvLight.build_mips, scene/light.py:72-86) -------------------------------------------
// The reference runs, every iteration, renderutils' specular_cubemap on each mip level (scene/renderutils/c_src/
// cubemap.cu:238-354: per output texel a loop over a bounds window with GGX weights, backward = the same loop with three
// atomics per (texel, window texel) pair) and diffuse_cubemap on the last one (:110-166).  For a given (resolution,
// roughness, cutoff) those filters are FIXED linear operators on the cubemap: out[t] = sum_s w(t, s) cube[s] / sum_s w(t, s).
// Here the operator is materialised once as a sparse matrix (CSR, normalised weights; a second CSR holds its transpose) and
// every iteration is a 3-channel SpMV each way -- no transcendental per pair, no atomics, deterministic, and bound by
// streaming 8 bytes per non-zero from HBM (26 M non-zeros for the 128/64/32/16 chain of the reference's defaults).
__device__ __forceinline__ float cm_pixel_area(int x, int y, int N)
{   // cubemap.cu:17-30
    if (N <= 1) return 1.0f;
    const int H = N / 2;
    x = abs(x - H); y = abs(y - H);
    const float dx = atanf((float)(x + 1) / (float)H) - atanf((float)x / (float)H);
    const float dy = atanf((float)(y + 1) / (float)H) - atanf((float)y / (float)H);
    return dx * dy;
}
__device__ __forceinline__ f3 cm_normalize(f3 v)
{   // safeNormalize of renderutils (vec3f.h): v / sqrt(max(dot, tiny))
    const float l = sqrtf(fmaxf(dot3(v, v), 1e-20f));
    return mk(v.x / l, v.y / l, v.z / l);
}
__device__ __forceinline__ f3 cm_cube_to_dir(int x, int y, int side, int N)
{   // cubemap.cu:32-46 (same face convention as face_to_dir)
    const float fx = 2.0f * (((float)x + 0.5f) / (float)N) - 1.0f;
    const float fy = 2.0f * (((float)y + 0.5f) / (float)N) - 1.0f;
    return cm_normalize(face_to_dir(side, fx, fy));
}
__device__ __forceinline__ float cm_ndf_ggx(float alphaSqr, float cosTheta)
{   // cubemap.cu:171-176
    const float c = fminf(fmaxf(cosTheta, 0.0f), 1.0f);
    const float d = (c * alphaSqr - c) * c + 1.0f;
    return alphaSqr / (d * d * 3.14159265358979323846f);
}
// weight of source texel (x, y, s) for the output direction VNR; <= 0 when the texel does not take part
__device__ __forceinline__ float cm_weight(int kind, f3 VNR, int x, int y, int s, int N, float alphaSqr, float cos_cutoff, bool& take)
{
    const f3 L = cm_cube_to_dir(x, y, s, N);
    const float d = dot3(L, VNR);
    if (kind == 0) {   // specular (cubemap.cu:262-276)
        take = d >= cos_cutoff;
        if (!take) return 0.0f;
        const f3 Hh = cm_normalize(mk(L.x + VNR.x, L.y + VNR.y, L.z + VNR.z));
        const float wiDotN = fmaxf(d, 0.0f), vh = fmaxf(dot3(VNR, Hh), 0.0f);
        return wiDotN * cm_ndf_ggx(alphaSqr, vh) * cm_pixel_area(x, y, N) / 4.0f;
    }
    take = true;       // diffuse (cubemap.cu:124-136): every texel, cosine clamped to [0, 0.999]
    const float ct = fminf(fmaxf(d, 0.0f), 0.999f);
    return ct * cm_pixel_area(x, y, N) / 3.141592f;
}

// pass 0: count the non-zeros of each row and sum its weights; pass 1: write column indices and normalised weights.
// One thread per output texel; 16x16 source tiles are culled with the interval test of SpecularBoundsKernel (cubemap.cu:203-214).
template <int PASS>
__global__ void __launch_bounds__(256) cubemap_filter_build_kernel(int N, int kind, float roughness, float cos_cutoff, uint32_t* __restrict__ row_count,
                                                                   float* __restrict__ row_wsum, const uint32_t* __restrict__ row_ptr,
                                                                   uint32_t* __restrict__ col, float* __restrict__ val)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int NT = 6 * N * N;
    if (t >= NT) return;
    const int px = t % N, py = (t / N) % N, pz = t / (N * N);
    const f3 VNR = cm_cube_to_dir(px, py, pz, N);
    const float alpha = roughness * roughness, alphaSqr = alpha * alpha;
    uint32_t cnt = 0;
    float wsum = 0.0f;
    uint32_t at = PASS == 1 ? row_ptr[t] : 0u;
    const float inv = PASS == 1 ? (kind == 0 ? 1.0f / row_wsum[t] : 1.0f) : 0.0f;   // only specular_cubemap divides by the weight sum (ops.py:459)
    const int TS = 16;
    for (int s = 0; s < 6; s++)
        for (int ty = 0; ty < (N + TS - 1) / TS; ty++)
            for (int tx = 0; tx < (N + TS - 1) / TS; tx++) {
                const int tsx = tx * TS, tsy = ty * TS, tex = min((tx + 1) * TS, N), tey = min((ty + 1) * TS, N);
                if (kind == 0) {
                    const f3 L0 = cm_cube_to_dir(tsx, tsy, s, N), L1 = cm_cube_to_dir(tex, tsy, s, N);
                    const f3 L2 = cm_cube_to_dir(tsx, tey, s, N), L3 = cm_cube_to_dir(tex, tey, s, N);
                    const float minx = fminf(fminf(L0.x, L1.x), fminf(L2.x, L3.x)), maxx = fmaxf(fmaxf(L0.x, L1.x), fmaxf(L2.x, L3.x));
                    const float miny = fminf(fminf(L0.y, L1.y), fminf(L2.y, L3.y)), maxy = fmaxf(fmaxf(L0.y, L1.y), fmaxf(L2.y, L3.y));
                    const float minz = fminf(fminf(L0.z, L1.z), fminf(L2.z, L3.z)), maxz = fmaxf(fmaxf(L0.z, L1.z), fmaxf(L2.z, L3.z));
                    const float maxdp = fmaxf(minx * VNR.x, maxx * VNR.x) + fmaxf(miny * VNR.y, maxy * VNR.y) + fmaxf(minz * VNR.z, maxz * VNR.z);
                    if (maxdp < cos_cutoff) continue;
                }
                for (int y = tsy; y < tey; y++)
                    for (int x = tsx; x < tex; x++) {
                        bool take;
                        const float w = cm_weight(kind, VNR, x, y, s, N, alphaSqr, cos_cutoff, take);
                        if (!take) continue;
                        if (PASS == 0) { cnt++; wsum += w; }
                        else { col[at] = (uint32_t)((s * N + y) * N + x); val[at] = w * inv; at++; }
                    }
            }
    if (PASS == 0) { row_count[t] = cnt; row_wsum[t] = wsum; }
}

// y[r, 0..2] = sum_k val[k] x[col[k], 0..2] over the non-zeros of row r; G lanes per row (4: short rows, 64: long rows).
// The kernel streams its matrix from HBM once per call, so the matrix is kept small: 16-bit column indices when they fit and
// 16-bit fixed-point weights with one fp32 scale per row (weight = q * scale[r], q in [0, 65535]: absolute error <= row
// maximum / 131070 per weight, ~1e-6 of the result for the rows of hundreds of similar weights these filters have).
#ifndef MRGS_SPMV_ROUNDS
#define MRGS_SPMV_ROUNDS 4
#endif
template <int G, typename IDX, typename WT>
__device__ __forceinline__ void csr_spmv3_body(int block, int nrows, const uint32_t* __restrict__ row_ptr, const IDX* __restrict__ col,
                                               const WT* __restrict__ val, const float* __restrict__ row_scale, const float* __restrict__ x,
                                               float* __restrict__ y)
{
    const int gid = (block * 256 + threadIdx.x) / G, sub = threadIdx.x % G;
    const int r = min(gid, nrows - 1);
    const uint32_t a = row_ptr[r], b = gid < nrows ? row_ptr[r + 1] : a;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
    uint32_t k = a + sub;
#ifndef MRGS_SPMV_NO_UNROLL
    // The matrix streams from HBM (it does not survive in the last-level cache between two iterations of a training step) and a wave
    // has only 2 x 128 bytes of it in flight per round: with every wave slot of the chip taken that is 2 MB against the ~8 MB that
    // 8 TB/s x 1 us of latency want.  MRGS_SPMV_ROUNDS rounds of (index, weight) loads are issued before the first gather.
    for (; k + (MRGS_SPMV_ROUNDS - 1) * G < b; k += MRGS_SPMV_ROUNDS * G) {
        uint32_t c[MRGS_SPMV_ROUNDS];
        float w[MRGS_SPMV_ROUNDS], v[MRGS_SPMV_ROUNDS][3];
#pragma unroll
        for (int u = 0; u < MRGS_SPMV_ROUNDS; ++u) { c[u] = col[k + u * G]; w[u] = (float)val[k + u * G]; }
#pragma unroll
        for (int u = 0; u < MRGS_SPMV_ROUNDS; ++u) {
            const float* xv = x + 3 * (size_t)c[u];
            v[u][0] = xv[0]; v[u][1] = xv[1]; v[u][2] = xv[2];
        }
#pragma unroll
        for (int u = 0; u < MRGS_SPMV_ROUNDS; ++u) { s0 += w[u] * v[u][0]; s1 += w[u] * v[u][1]; s2 += w[u] * v[u][2]; }
    }
#endif
    for (; k < b; k += G) {
        const float w = (float)val[k];
        const float* xv = x + 3 * (size_t)col[k];
        s0 += w * xv[0]; s1 += w * xv[1]; s2 += w * xv[2];
    }
#pragma unroll
    for (int d = G / 2; d >= 1; d >>= 1) {
        s0 += __shfl_xor(s0, d, 64); s1 += __shfl_xor(s1, d, 64); s2 += __shfl_xor(s2, d, 64);
    }
    if (sub == 0 && gid < nrows) {
        const float sc = row_scale != nullptr ? row_scale[r] : 1.0f;
        y[3 * (size_t)r] = s0 * sc; y[3 * (size_t)r + 1] = s1 * sc; y[3 * (size_t)r + 2] = s2 * sc;
    }
}

// Blocked rows ("val_bytes == 8"): the long rows of the GGX / cosine filters touch runs of ~10-17 consecutive texels, so a row is stored
// as blocks of four consecutive columns -- a 16-bit block index (column / 4) and four 16-bit fixed-point weights (zero where the row has
// no entry; 0.88-0.93 of the slots are used) = 2.8 bytes per non-zero instead of 4, and ONE 48-byte gather of x (three 16-byte loads,
// 4 texels x 3 channels) instead of four gathers of 12 bytes.  64 lanes per row, one block per lane and round.  row_ptr counts blocks.
#ifndef MRGS_SPMV_BLK_ROUNDS
#define MRGS_SPMV_BLK_ROUNDS 2      // measured in the batched prefilter launch: 2 rounds 43.5 us, 4 rounds 51.5 (66 VGPRs), 8 rounds 47.7; plain CSR 51.4
#endif
// (Also measured: a wave owning 2 / 4 / 8 consecutive rows and walking their blocks as one range, products added to the row's accumulators
// with 0/1 factors -- fewer, longer waves without the partly empty last round per row: 45.7 / 54.6 / 74.7 us against 42.6 for a row per
// wave.  Round 4: a wave owning 2 / 4 / 8 rows strided by the wave count, software-pipelined ACROSS rows -- the next row's extent fetched
// two rows ahead, its first round of (index, weights) one row ahead, so that a row starts with its gathers instead of three dependent
// round trips: the same time at 2 rows, +5 us (forward) and +15 us (transposed call) at 4, +16 us (forward) at 8, bit-identical sums.  The launch wants MANY short waves;
// what holds it at 1.7 TB/s was not found.)
__device__ __forceinline__ void csr_spmv3_blk4_body(int block, int nrows, const uint32_t* __restrict__ row_ptr, const uint16_t* __restrict__ bcol,
                                                    const uint2* __restrict__ bval, const float* __restrict__ row_scale,
                                                    const float* __restrict__ x, float* __restrict__ y)
{
    const int gid = (block * 256 + threadIdx.x) / 64, sub = threadIdx.x % 64;
    const int r = min(gid, nrows - 1);
    const uint32_t a = row_ptr[r], b = gid < nrows ? row_ptr[r + 1] : a;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
    auto consume = [&](uint32_t c, uint2 w, const float4& v0, const float4& v1, const float4& v2) {
        const float w0 = (float)(w.x & 0xFFFFu), w1 = (float)(w.x >> 16), w2 = (float)(w.y & 0xFFFFu), w3 = (float)(w.y >> 16);
        s0 = fmaf(w0, v0.x, s0); s1 = fmaf(w0, v0.y, s1); s2 = fmaf(w0, v0.z, s2);
        s0 = fmaf(w1, v0.w, s0); s1 = fmaf(w1, v1.x, s1); s2 = fmaf(w1, v1.y, s2);
        s0 = fmaf(w2, v1.z, s0); s1 = fmaf(w2, v1.w, s1); s2 = fmaf(w2, v2.x, s2);
        s0 = fmaf(w3, v2.y, s0); s1 = fmaf(w3, v2.z, s1); s2 = fmaf(w3, v2.w, s2);
    };
    uint32_t k = a + sub;
    // MRGS_SPMV_BLK_ROUNDS rounds in flight: a wave has only 64 x 10 bytes of the matrix per round, and the gather of x depends on it
    for (; k + (MRGS_SPMV_BLK_ROUNDS - 1) * 64 < b; k += MRGS_SPMV_BLK_ROUNDS * 64) {
        uint32_t c[MRGS_SPMV_BLK_ROUNDS];
        uint2 w[MRGS_SPMV_BLK_ROUNDS];
        float4 v[MRGS_SPMV_BLK_ROUNDS][3];
#pragma unroll
        for (int u = 0; u < MRGS_SPMV_BLK_ROUNDS; ++u) { c[u] = bcol[k + u * 64]; w[u] = bval[k + u * 64]; }
#pragma unroll
        for (int u = 0; u < MRGS_SPMV_BLK_ROUNDS; ++u) {
            const float4* p = reinterpret_cast<const float4*>(x) + 3 * (size_t)c[u];
            v[u][0] = p[0]; v[u][1] = p[1]; v[u][2] = p[2];
        }
#pragma unroll
        for (int u = 0; u < MRGS_SPMV_BLK_ROUNDS; ++u) consume(c[u], w[u], v[u][0], v[u][1], v[u][2]);
    }
    for (; k < b; k += 64) {
        const uint32_t c0 = bcol[k];
        const uint2 w0 = bval[k];
        const float4* p0 = reinterpret_cast<const float4*>(x) + 3 * (size_t)c0;
        const float4 a0 = p0[0], a1 = p0[1], a2 = p0[2];
        consume(c0, w0, a0, a1, a2);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        s0 += __shfl_xor(s0, d, 64); s1 += __shfl_xor(s1, d, 64); s2 += __shfl_xor(s2, d, 64);
    }
    if (sub == 0 && gid < nrows) {
        const float sc = row_scale[r];
        y[3 * (size_t)r] = s0 * sc; y[3 * (size_t)r + 1] = s1 * sc; y[3 * (size_t)r + 2] = s2 * sc;
    }
}
__global__ void __launch_bounds__(256) csr_spmv3_blk4_kernel(int nrows, const uint32_t* __restrict__ row_ptr, const uint16_t* __restrict__ bcol,
                                                             const uint2* __restrict__ bval, const float* __restrict__ row_scale,
                                                             const float* __restrict__ x, float* __restrict__ y)
{
    csr_spmv3_blk4_body((int)blockIdx.x, nrows, row_ptr, bcol, bval, row_scale, x, y);
}

template <int G, typename IDX, typename WT>
__global__ void __launch_bounds__(256) csr_spmv3_kernel(int nrows, const uint32_t* __restrict__ row_ptr, const IDX* __restrict__ col,
                                                        const WT* __restrict__ val, const float* __restrict__ row_scale,
                                                        const float* __restrict__ x, float* __restrict__ y)
{
    csr_spmv3_body<G, IDX, WT>((int)blockIdx.x, nrows, row_ptr, col, val, row_scale, x, y);
}

// (Tried in round 3 and removed: the long-row levels with their vector resident in LDS -- 16 x 16 / 32 x 32 cubemaps whole, a 64 x 64 face
// as one of six panels, 16 bytes per texel, a workgroup per block of rows walking the panels its rows touch.  Correct, and 80-150 us per
// call against 50 us for the streaming kernel below: one 1 024-thread workgroup per CU (or two of 512) in lock step through
// stage / barrier / consume phases keeps fewer requests in flight than 2 500 independent 256-thread workgroups do, whatever the gathers cost.)
// Several independent SpMVs in ONE launch (the levels of the environment prefilter: four launches of 9-18 us each were mostly the
// latency of streaming each matrix with a fraction of the chip; together the matrices stream with every wave slot busy).  A workgroup
// finds its product from a table in the kernel arguments (wave-uniform scan) and runs the body of its storage format.
struct SpmvSeg {
    int nrows, fmt, first_block, log2n;
    const uint32_t* row_ptr; const void* col; const void* val; const float* row_scale; const float* x; float* y;
    // format 16 (tiles of rows of ONE fundamental domain of the cube's symmetry group, below): the row -> image rows table, the factor of
    // the vector's texels, the tiles' row ranges, their panels (the column blocks a tile's rows touch); `val` = the tiles' dense weights
    const int32_t* image_rows; const float* pre; const uint32_t* tile_ptr; const uint32_t* panel_ptr; const uint16_t* panel_src;
};
struct SpmvBatch { int n; int total_blocks; uint64_t sym[48]; SpmvSeg seg[MRGS_SPMV_MAX_BATCH]; unsigned long long* trace; };

// ---- the filters as operators on ONE fundamental domain of the cube's 48 symmetries -------------------------------------------------
// A filter weight is K(r, c) * area(c) / n(r): K (GGX lobe x cosine x cut-off x the reference's tile cull) depends on the two directions
// only and is invariant under the 48 signed axis permutations g, which map the texel grid of a cube map onto itself:
// K(g r, g c) = K(r, c).  (The solid angle `area` of the reference, cubemap.cu:17-30, is NOT invariant -- it is off by one texel on the
// negative half of a face -- and the row sum n inherits that; both are per-texel factors and stay outside.)  So
//     y[g r0] = post[g r0] * sum_c K'(r0, c) * (pre * x)[g c]
// for the N/2 (N/2 + 1) / 2 rows r0 of a triangle of face 0: the matrix is 1/47 of the full one (60 MB -> 1.3 MB for the 64 x 64 level)
// and stays in the L2 of every XCD instead of streaming from HBM each call and each way.  That alone bought little: with its matrix
// cached the gather kernel is bound by the gather of x itself (48 bytes per block of four columns, 350 MB a call: 33 us against 43 from
// HBM), and with x staged in LDS per (tile of rows, symmetry) by the instructions of 64-lane reductions per row (measured, round 5).
// What the symmetry really gives is a matrix PRODUCT: the rows of a 4 x 4 TILE of the triangle touch the same ~1 000-2 300 texels (the
// tile's PANEL, 60 % of it per row), so a tile is a dense 16 x K matrix, and the 48 symmetries x 3 channels are 144 right-hand sides:
//     Y[16 rows][48 g x 3] = W[16][K] * X[K][48 g x 3],   X[k][g, :] = (pre * x)[g panel_k]
// on v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate: an fmaf chain; the 16-bit fixed-point weights are exact in f32).  A workgroup =
// (tile, four symmetries: 16 columns with the channel padded to four); its four waves split K; a wave stages 16 texels x 4 symmetries
// per step (one (texel, symmetry) per lane: 12 bytes of x and its factor -> 16 bytes of LDS, wave-private, double-buffered) for four MFMAs.
// sym[g] packs, per SOURCE face s (6 bits each): image face | axes exchanged << 3 | row flipped << 4 | column flipped << 5.
struct CubeImage { int s, y, x; };
static inline CubeImage cube_symmetry_image(int g, int N, int s, int y, int x)
{
    static const int PERM[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
    const int u = 2 * x - (N - 1), v = 2 * y - (N - 1);       // texel centre in units of 1 / N, the major axis at +-N (face_to_dir)
    int d[3];
    switch (s) {
    case 0: d[0] = N; d[1] = -v; d[2] = -u; break;
    case 1: d[0] = -N; d[1] = -v; d[2] = u; break;
    case 2: d[0] = u; d[1] = N; d[2] = v; break;
    case 3: d[0] = u; d[1] = -N; d[2] = -v; break;
    case 4: d[0] = u; d[1] = -v; d[2] = N; break;
    default: d[0] = -u; d[1] = -v; d[2] = -N; break;
    }
    int e[3];
    for (int i = 0; i < 3; ++i) e[i] = (((g & 7) >> i) & 1) ? -d[PERM[g >> 3][i]] : d[PERM[g >> 3][i]];
    int uu, vv, ss;
    if (e[0] == N) { ss = 0; vv = -e[1]; uu = -e[2]; }
    else if (e[0] == -N) { ss = 1; vv = -e[1]; uu = e[2]; }
    else if (e[1] == N) { ss = 2; uu = e[0]; vv = e[2]; }
    else if (e[1] == -N) { ss = 3; uu = e[0]; vv = -e[2]; }
    else if (e[2] == N) { ss = 4; uu = e[0]; vv = -e[1]; }
    else { ss = 5; uu = -e[0]; vv = -e[1]; }
    return {ss, (vv + N - 1) / 2, (uu + N - 1) / 2};
}
static void cube_symmetry_table(uint64_t* tab)
{
    for (int g = 0; g < 48; ++g) {
        uint64_t t = 0;
        for (int s = 0; s < 6; ++s) {
            const CubeImage o = cube_symmetry_image(g, 4, s, 0, 0), ax = cube_symmetry_image(g, 4, s, 0, 1), ay = cube_symmetry_image(g, 4, s, 1, 0);
            const bool swap = ax.x == o.x;                       // a step along x moves the image along y
            const bool fc = swap ? ax.y < o.y : ax.x < o.x;      // the image runs backwards along the source's x
            const bool fr = swap ? ay.x < o.x : ay.y < o.y;      // ... along the source's y
            t |= (uint64_t)(o.s | (swap ? 8 : 0) | (fr ? 16 : 0) | (fc ? 32 : 0)) << (6 * s);
        }
        tab[g] = t;
    }
}

#ifndef MRGS_SPMV_BATCH_THREADS
#define MRGS_SPMV_BATCH_THREADS 512
#endif
#define MRGS_SPMV_SYM_WAVES (MRGS_SPMV_BATCH_THREADS / 64)      // the waves of a workgroup split the panel
static_assert(128 * MRGS_SPMV_SYM_WAVES >= MRGS_SPMV_MAX_PANEL, "a wave holds the patch ids of 128 steps");
#define MRGS_SPMV_SYM_BROW 20        // floats per staged texel row: 16 columns + 4 of padding (the 16-byte writes of eight lanes then cover the 32 banks)
#define MRGS_SPMV_SYM_LDS (MRGS_SPMV_SYM_WAVES * (16 * MRGS_SPMV_SYM_BROW * 4 + 1024))  // per wave: a staging buffer of 16 texels x 16 columns, a partial tile
#ifndef MRGS_SPMV_SYM_DEPTH
#define MRGS_SPMV_SYM_DEPTH 2       // steps of gathers in flight ahead of the MFMAs (a step waits for nothing younger than DEPTH steps)
#endif
typedef float spmv_f32x4 __attribute__((ext_vector_type(4)));
extern __shared__ float4 spmv_lds[];
#ifdef MRGS_SPMV_TRACE
#define MRGS_SPMV_TRACE_PTR trace_ptr
#endif
__device__ __forceinline__ void csr_spmv3_sym_body(int block, const SpmvSeg& S, const uint64_t* __restrict__ sym, unsigned long long* trace_ptr)
{
    constexpr int NW = MRGS_SPMV_SYM_WAVES, D = MRGS_SPMV_SYM_DEPTH;
#ifdef MRGS_SPMV_TRACE
    const unsigned long long tr0 = wall_clock64();
#endif
    const int t = block / 12, ig = block - t * 12;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int L = S.log2n, N = 1 << L;
    const uint32_t p0 = S.panel_ptr[t], nq = S.panel_ptr[t + 1] - p0;                  // the tile's panel: 4 x 4 texel patches
    const uint2* __restrict__ A = reinterpret_cast<const uint2*>(S.val) + (size_t)p0 * 64;     // [patch][lane] -> four 16-bit weights, one per row of the patch
    const uint16_t* __restrict__ psrc = S.panel_src + p0;
    // staging role of the lane: texel kk of the patch's 16 (row kk >> 2, column kk & 3) under symmetry 4 ig + (lane >> 4).  The image of a
    // patch is a patch -- four runs of 48 bytes whichever way the symmetry turns the face (a run of 16 texels along x became 16 lines when
    // the axes were exchanged; the lines fetched into L1, 128 bytes for every 12 used, bounded the kernel: measured, round 5)
    const uint32_t kk = (uint32_t)lane & 15u, simg = (uint32_t)lane >> 4;
    const uint64_t tab = sym[4 * ig + (int)simg];
    float* Bs = reinterpret_cast<float*>(spmv_lds) + wave * (16 * MRGS_SPMV_SYM_BROW);     // (LDS serves a wave's accesses in order: no barrier between its write and its reads)
    const int LP = L - 2;                                               // patches per face edge = N / 4
    auto texel_of = [&](uint32_t cb) {
        const uint32_t x = 4u * (cb & (uint32_t)((1 << LP) - 1)) + (kk & 3u), y = 4u * ((cb >> LP) & (uint32_t)((1 << LP) - 1)) + (kk >> 2), s = cb >> (2 * LP);
        const uint32_t e = (uint32_t)(tab >> (6u * s)) & 63u;
        const uint32_t yy = (e & 16u) ? (uint32_t)(N - 1) - y : y, xx = (e & 32u) ? (uint32_t)(N - 1) - x : x;
        return ((((e & 7u) << L) + ((e & 8u) ? xx : yy)) << L) + ((e & 8u) ? yy : xx);
    };
    // software pipeline over the wave's steps: step i = patch wave + NW i of the panel, texels and weights of D steps in flight.  The loop
    // is unrolled D times over statically named stages: a stage's registers are the targets of loads in flight, and moving them down a
    // ring would wait for every one of them (it did: 0.35 us a step, per-wave timestamps, round 5).
    const uint32_t ni = nq > (uint32_t)wave ? (nq - (uint32_t)wave + (uint32_t)NW - 1u) / (uint32_t)NW : 0u;      // the wave's steps (<= 128)
    // the patch ids of all its steps in two VECTOR loads (lane j: steps j and 64 + j), read back per step with v_readlane -- as scalar loads
    // in the loop they would share the LDS reads' counter (lgkmcnt) and make every step wait for a scalar-cache miss
    const uint32_t q0 = (uint32_t)wave + (uint32_t)NW * (uint32_t)lane, q1 = q0 + 64u * (uint32_t)NW;
    const uint32_t ids0 = q0 < nq ? (uint32_t)psrc[q0] : 0u, ids1 = q1 < nq ? (uint32_t)psrc[q1] : 0u;
    auto patch_of = [&](uint32_t i) {      // (i uniform; beyond the wave's steps: patch 0, whose weights the step then zeroes)
        const uint32_t j = i < 127u ? i : 127u;
        return j < 64u ? (uint32_t)__builtin_amdgcn_readlane((int)ids0, (int)j) : (uint32_t)__builtin_amdgcn_readlane((int)ids1, (int)(j - 64u));
    };
    struct Step { float f, v0, v1, v2; uint2 a4; };
    auto gather = [&](uint32_t i) {
        Step st;
        const uint32_t tex = texel_of(patch_of(i));
        st.f = S.pre[tex];
        const float* __restrict__ src = S.x + 3u * tex;
        st.v0 = src[0]; st.v1 = src[1]; st.v2 = src[2];
        const uint32_t qd = (uint32_t)wave + (uint32_t)NW * i;
        st.a4 = A[(size_t)(qd < nq ? qd : nq - 1u) * 64 + lane];
        return st;
    };
    // (the epilogue's row and its factor are asked for now: two dependent loads off the end of the workgroup's life)
    const uint32_t r_first = S.tile_ptr[t], r_end = S.tile_ptr[t + 1];
    const int orow_t = (int)threadIdx.x >> 4, ocol = threadIdx.x & 15;
    int orow = -1;
    if (threadIdx.x < 256 && (ocol & 3) < 3 && r_first + (uint32_t)orow_t < r_end) orow = S.image_rows[(r_first + (uint32_t)orow_t) * 48 + 4 * ig + (ocol >> 2)];
    const float opost = orow >= 0 ? S.row_scale[orow] : 0.0f;
    Step st[D];
#pragma unroll
    for (int j = 0; j < D; ++j) st[j] = gather((uint32_t)j);
    spmv_f32x4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = spmv_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#ifdef MRGS_SPMV_TRACE
    unsigned long long tr1 = 0, tr2 = 0;
    { float probe = st[0].f + st[0].v0 + (float)st[0].a4.x; asm volatile("" :: "v"(probe)); tr1 = wall_clock64(); }
#endif
    float4* Bw = reinterpret_cast<float4*>(Bs + kk * MRGS_SPMV_SYM_BROW) + simg;        // [texel kk][symmetry][4] = [kk][16 columns], rows padded
    const float* Br = Bs + (lane >> 4) * MRGS_SPMV_SYM_BROW + (lane & 15);               // B[k = lane >> 4][column lane & 15] of a patch row
    for (uint32_t i0 = 0; i0 < ni; i0 += (uint32_t)D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
            // this step: the wave's 16 texels x 4 symmetries into its buffer, four MFMAs out of it (a step past the wave's last has zero weights)
            const Step c0 = st[u];
            const bool live = i0 + (uint32_t)u < ni;
            const uint32_t ax = live ? c0.a4.x : 0u, ay = live ? c0.a4.y : 0u;
            *Bw = make_float4(c0.f * c0.v0, c0.f * c0.v1, c0.f * c0.v2, 0.0f);
            const float b0 = Br[0], b1 = Br[4 * MRGS_SPMV_SYM_BROW], b2 = Br[8 * MRGS_SPMV_SYM_BROW], b3 = Br[12 * MRGS_SPMV_SYM_BROW];
            st[u] = gather(i0 + (uint32_t)(u + D));
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32((float)(ax & 0xFFFFu), b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32((float)(ax >> 16), b1, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32((float)(ay & 0xFFFFu), b2, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32((float)(ay >> 16), b3, acc[3], 0, 0, 0);
        }
    }
#ifdef MRGS_SPMV_TRACE
    { float probe = acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0]; asm volatile("" :: "v"(probe)); tr2 = wall_clock64(); }
#endif
    // the waves' partial tiles -> one; C/D of the MFMA: column lane & 15, rows 4 (lane >> 4) + register
    float* red = reinterpret_cast<float*>(spmv_lds) + NW * (16 * MRGS_SPMV_SYM_BROW);
#pragma unroll
    for (int r = 0; r < 4; ++r)
        red[wave * 256 + (4 * (lane >> 4) + r) * 16 + (lane & 15)] = (acc[0][r] + acc[1][r]) + (acc[2][r] + acc[3][r]);
    __syncthreads();
    if (orow >= 0) {      // (-1: a texel on the triangle's diagonal is its own image under one reflection -- written once)
        float sum = 0.0f;
#pragma unroll
        for (int w = 0; w < NW; ++w) sum += red[w * 256 + orow_t * 16 + ocol];
        S.y[3 * (size_t)orow + (ocol & 3)] = sum * opost;
    }
#ifdef MRGS_SPMV_TRACE
    if (MRGS_SPMV_TRACE_PTR != nullptr && lane == 0) {
        unsigned long long* o = MRGS_SPMV_TRACE_PTR + ((size_t)blockIdx.x * NW + wave) * 4;
        o[0] = tr0; o[1] = tr1; o[2] = tr2; o[3] = wall_clock64();
    }
#endif
}

__global__ void __launch_bounds__(MRGS_SPMV_BATCH_THREADS) csr_spmv3_batched_kernel(SpmvBatch B)
{
    int i = 0;
    for (int k = 1; k < B.n; ++k) i = ((int)blockIdx.x >= B.seg[k].first_block) ? k : i;
    const SpmvSeg& S = B.seg[i];
    const int own = (int)blockIdx.x - S.first_block;
    // (the row-per-wave bodies count 256-thread blocks: block 2 own + (threadIdx.x >> 8), which is what their `block * 256 + threadIdx.x` makes of 2 own)
    const int blk = S.fmt == 16 ? own : own * (MRGS_SPMV_BATCH_THREADS / 256);
    switch (S.fmt) {   // bit 2: 64 lanes per row (else 4); bit 1: 32-bit column indices (else 16); bit 0: fp32 weights (else 16-bit fixed point); 8: blocked rows; 16: blocked rows of one fundamental domain
    case 16: csr_spmv3_sym_body(blk, S, B.sym, B.trace); break;
    case 8: csr_spmv3_blk4_body(blk, S.nrows, S.row_ptr, (const uint16_t*)S.col, (const uint2*)S.val, S.row_scale, S.x, S.y); break;
    case 0: csr_spmv3_body<4, uint16_t, uint16_t>(blk, S.nrows, S.row_ptr, (const uint16_t*)S.col, (const uint16_t*)S.val, S.row_scale, S.x, S.y); break;
    case 1: csr_spmv3_body<4, uint16_t, float>(blk, S.nrows, S.row_ptr, (const uint16_t*)S.col, (const float*)S.val, S.row_scale, S.x, S.y); break;
    case 2: csr_spmv3_body<4, uint32_t, uint16_t>(blk, S.nrows, S.row_ptr, (const uint32_t*)S.col, (const uint16_t*)S.val, S.row_scale, S.x, S.y); break;
    case 3: csr_spmv3_body<4, uint32_t, float>(blk, S.nrows, S.row_ptr, (const uint32_t*)S.col, (const float*)S.val, S.row_scale, S.x, S.y); break;
    case 4: csr_spmv3_body<64, uint16_t, uint16_t>(blk, S.nrows, S.row_ptr, (const uint16_t*)S.col, (const uint16_t*)S.val, S.row_scale, S.x, S.y); break;
    case 5: csr_spmv3_body<64, uint16_t, float>(blk, S.nrows, S.row_ptr, (const uint16_t*)S.col, (const float*)S.val, S.row_scale, S.x, S.y); break;
    case 6: csr_spmv3_body<64, uint32_t, uint16_t>(blk, S.nrows, S.row_ptr, (const uint32_t*)S.col, (const uint16_t*)S.val, S.row_scale, S.x, S.y); break;
    default: csr_spmv3_body<64, uint32_t, float>(blk, S.nrows, S.row_ptr, (const uint32_t*)S.col, (const float*)S.val, S.row_scale, S.x, S.y); break;
    }
}

// 2x2 box mip of a [6, N, N, 3] cubemap (cubemap_mip.forward, scene/light_utils.py:68-69)
__global__ void __launch_bounds__(256) cubemap_mip_fwd_kernel(int Nout, const float* __restrict__ in, float* __restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 6 * Nout * Nout * 3) return;
    const int c = i % 3, x = (i / 3) % Nout, y = (i / (3 * Nout)) % Nout, s = i / (3 * Nout * Nout);
    const int N = 2 * Nout;
    const float* p = in + ((size_t)(s * N + 2 * y) * N + 2 * x) * 3 + c;
    out[i] = 0.25f * (p[0] + p[3] + p[(size_t)N * 3] + p[(size_t)N * 3 + 3]);
}
// cubemap_mip.backward (scene/light_utils.py:71-80): NOT the transpose of the box filter -- the reference samples 0.25 * dout
// with a seamless bilinear cube fetch at the texel-centre directions of the finer level; reproduced as is, accumulating
// into g_fine (which already holds the finer level's own gradient).
__global__ void __launch_bounds__(256) cubemap_mip_bwd_kernel(int N, const float* __restrict__ dout /*[6,N/2,N/2,3]*/, float* __restrict__ g_fine)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= 6 * N * N) return;
    const int x = t % N, y = (t / N) % N, s = t / (N * N);
    const float gx = -1.0f + 1.0f / (float)N + (float)x * ((2.0f - 2.0f / (float)N) / (float)(N - 1));   // torch.linspace(-1+1/N, 1-1/N, N)
    const float gy = -1.0f + 1.0f / (float)N + (float)y * ((2.0f - 2.0f / (float)N) / (float)(N - 1));
    const f3 v = cm_normalize(face_to_dir(s, gx, gy));
    const FaceUV fu = dir_to_face(v);
    const Taps tp = cube_taps(fu, N / 2);
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float a = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (tp.w[q] != 0.f) a += tp.w[q] * dout[(size_t)tp.idx[q] * 3 + c];
        g_fine[(size_t)t * 3 + c] += 0.25f * a;
    }
}

// The whole box-mip chain below one level in ONE launch (cubemap_mip.forward applied `steps` <= 3 times, scene/light.py:74-76): a
// thread owns one channel of one texel of the coarsest level and averages its 2^steps x 2^steps block bottom-up, writing every level on the
// way -- the same 2x2 sums in the same order as the level-by-level kernel above, so the values are bit-identical to it.
template <int LVL>
__device__ __forceinline__ float mip_block(const float* __restrict__ in, int N0, int s, int y, int x, int c, float* const* outs)
{
    if constexpr (LVL == 0) {
        return in[((size_t)(s * N0 + y) * N0 + x) * 3 + c];
    } else {
        const float p00 = mip_block<LVL - 1>(in, N0, s, 2 * y, 2 * x, c, outs), p01 = mip_block<LVL - 1>(in, N0, s, 2 * y, 2 * x + 1, c, outs);
        const float p10 = mip_block<LVL - 1>(in, N0, s, 2 * y + 1, 2 * x, c, outs), p11 = mip_block<LVL - 1>(in, N0, s, 2 * y + 1, 2 * x + 1, c, outs);
        const float v = 0.25f * (p00 + p01 + p10 + p11);
        const int N = N0 >> LVL;
        outs[LVL - 1][((size_t)(s * N + y) * N + x) * 3 + c] = v;
        return v;
    }
}
struct MipOuts { float* o[3]; };
// Three levels with FOUR lanes per texel of the coarsest one (a DPP quad): each lane averages a 4x4 quarter of the texel's 8x8 block
// bottom-up (16 loads in flight instead of 64 behind each other), lane 0 of the quad folds the four level-2 values in the level-by-level
// kernel's order ((p00 + p01) + p10) + p11 -- bit-identical to it (10 -> 5 us for the 128..16 chain).
__global__ void __launch_bounds__(256) cubemap_mip_chain3_fwd_kernel(int N0, const float* __restrict__ in, MipOuts outs)
{
    const int Nc = N0 >> 3;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const bool live = t < 6 * Nc * Nc * 3 * 4;
    const int q = t & 3, i = live ? t >> 2 : 0;
    const int c = i % 3, x = (i / 3) % Nc, y = (i / (3 * Nc)) % Nc, s = i / (3 * Nc * Nc);
    float* const o[3] = {outs.o[0], outs.o[1], outs.o[2]};
    // my level-2 texel: (2 y + q / 2, 2 x + q % 2); its level-1 texels and their level-0 blocks below
    const int y2 = 2 * y + (q >> 1), x2 = 2 * x + (q & 1);
    const float v2 = live ? mip_block<2>(in, N0, s, y2, x2, c, o) : 0.0f;       // writes levels 1 and 2
    const float p01 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v2), 0x55, 0xf, 0xf, false));   // quad_perm [1,1,1,1]
    const float p10 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v2), 0xAA, 0xf, 0xf, false));   // [2,2,2,2]
    const float p11 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v2), 0xFF, 0xf, 0xf, false));   // [3,3,3,3]
    if (live && q == 0) o[2][((size_t)(s * Nc + y) * Nc + x) * 3 + c] = 0.25f * (v2 + p01 + p10 + p11);
}
__global__ void __launch_bounds__(256) cubemap_mip_chain_fwd_kernel(int N0, int steps, const float* __restrict__ in, MipOuts outs)
{
    const int Nc = N0 >> steps;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 6 * Nc * Nc * 3) return;
    const int c = i % 3, x = (i / 3) % Nc, y = (i / (3 * Nc)) % Nc, s = i / (3 * Nc * Nc);
    float* const o[3] = {outs.o[0], outs.o[1], outs.o[2]};
    if (steps == 1) mip_block<1>(in, N0, s, y, x, c, o);
    else if (steps == 2) mip_block<2>(in, N0, s, y, x, c, o);
    else mip_block<3>(in, N0, s, y, x, c, o);
}

// ---- C ABI ---------------------------------------------------------------------------------------------------
static int make_mips(const MrgsEnvMips* in, EnvMips& m)
{
    if (!in || in->n_levels < 1 || in->n_levels > MRGS_MAX_MIPS) return MRGS_E_BAD_ARG;
    m.n = in->n_levels;
    for (int i = 0; i < MRGS_MAX_MIPS; i++) {
        m.res[i] = i < m.n ? in->res[i] : 0;
        m.tex[i] = i < m.n ? in->tex[i] : nullptr;
        m.grad[i] = i < m.n ? in->grad[i] : nullptr;
        m.copies[i] = (i < m.n && in->grad_copies[i] > 1) ? in->grad_copies[i] : 1;
        m.lds_off[i] = -1;
        if (i < m.n && (!m.tex[i] || m.res[i] < 1)) return MRGS_E_BAD_ARG;
    }
    m.min_roughness = in->min_roughness;
    m.max_roughness = in->max_roughness;
    return MRGS_OK;
}
static Map to_map(const MrgsStridedMap& s) { Map r = {s.ptr, s.stride_h, s.stride_w, s.stride_c}; return r; }

extern "C" {

int mrgs_envmap_lookup_forward(const MrgsEnvMips* mips, int64_t N, const float* dirs, const float* roughness, float* out, void* stream)
{
    EnvMips m;
    int rc = make_mips(mips, m);
    if (rc) return rc;
    if (N < 0 || (N > 0 && (!dirs || !out))) return MRGS_E_BAD_ARG;
    if (N == 0) return MRGS_OK;
    hipLaunchKernelGGL(envmap_lookup_fwd_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, m, (long long)N, dirs,
                       roughness, out);
    return hipGetLastError() == hipSuccess ? MRGS_OK : MRGS_E_HIP;
}

int mrgs_envmap_lookup_backward(const MrgsEnvMips* mips, int64_t N, const float* dirs, const float* roughness, const float* g_out,
                                float* g_dirs, float* g_roughness, void* stream)
{
    EnvMips m;
    int rc = make_mips(mips, m);
    if (rc) return rc;
    if (N < 0 || (N > 0 && (!dirs || !g_out))) return MRGS_E_BAD_ARG;
    if (N == 0) return MRGS_OK;
    hipLaunchKernelGGL(envmap_lookup_bwd_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, m, (long long)N, dirs,
                       roughness, g_out, g_dirs, g_roughness);
    return hipGetLastError() == hipSuccess ? MRGS_OK : MRGS_E_HIP;
}

static int shade_specular_forward_impl(const MrgsEnvMips* mips, const MrgsShadeFrame* fr, float* specular, float* direct_light, float* specular_weight,
                                       const float* base_color, const float* bg, int srgb, float* render, float* diffuse, float* zero_fill,
                                       int64_t zero_floats, void* stream)
{
    EnvMips m;
    int rc = make_mips(mips, m);
    if (rc) return rc;
    if (!fr || fr->H <= 0 || fr->W <= 0 || !fr->R || !fr->T || !fr->lut || fr->lut_res < 1 || !specular || !direct_light || !specular_weight)
        return MRGS_E_BAD_ARG;
    if (zero_floats < 0 || (zero_floats > 0 && !zero_fill)) return MRGS_E_BAD_ARG;
    ShadeCam cam;
    for (int i = 0; i < 9; i++) cam.Kinv[i] = fr->Kinv[i];
    cam.R = fr->R; cam.T = fr->T;
    const dim3 grid((fr->W + 63) / 64, (fr->H + 3) / 4), block(256);
    hipLaunchKernelGGL(shade_specular_fwd_kernel, grid, block, 0, (hipStream_t)stream, m, cam, fr->H, fr->W, to_map(fr->albedo), to_map(fr->normal),
                       to_map(fr->alpha), to_map(fr->refl), to_map(fr->roughness), fr->lut, fr->lut_res, specular, direct_light, specular_weight,
                       base_color, bg, srgb, render, diffuse, zero_fill, (long long)zero_floats);
    return hipGetLastError() == hipSuccess ? MRGS_OK : MRGS_E_HIP;
}

int mrgs_shade_specular_forward(const MrgsEnvMips* mips, const MrgsShadeFrame* fr, float* specular, float* direct_light, float* specular_weight,
                                void* stream)
{
    return shade_specular_forward_impl(mips, fr, specular, direct_light, specular_weight, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0, stream);
}

int mrgs_shade_specular_forward_composite(const MrgsEnvMips* mips, const MrgsShadeFrame* fr, const float* base_color, const float* bg, int32_t srgb,
                                          float* specular, float* direct_light, float* specular_weight, float* render, float* diffuse,
                                          float* zero_fill, int64_t zero_floats, void* stream)
{
    if (!base_color || !bg || !render || !diffuse) return MRGS_E_BAD_ARG;
    return shade_specular_forward_impl(mips, fr, specular, direct_light, specular_weight, base_color, bg, srgb ? 1 : 0, render, diffuse, zero_fill,
                                       zero_floats, stream);
}

// composite != nullptr: render_surfel's compositing backward in the same launch (mrgs_surfel_shade_composite_backward)
struct ShadeCompositeArgs { int srgb; const float *base, *spec_fwd, *bg, *g_render, *g_diffuse; float* g_base; };

static int shade_specular_backward_impl(const MrgsEnvMips* mips, const MrgsShadeFrame* fr, const float* g_specular, const float* g_direct_light,
                                        const float* g_specular_weight, float* g_albedo, float* g_normal, float* g_alpha, float* g_refl,
                                        float* g_roughness, float* g_features, const float* g_refl_composite, const float* g_alpha_composite,
                                        const ShadeCompositeArgs* composite, void* stream)
{
    EnvMips m;
    int rc = make_mips(mips, m);
    if (rc) return rc;
    if (!fr || fr->H <= 0 || fr->W <= 0 || !fr->R || !fr->T || !fr->lut || !g_normal || !g_alpha) return MRGS_E_BAD_ARG;
    if (composite ? (!g_features || !composite->base || !composite->spec_fwd || !composite->bg || !composite->g_base)
                  : (g_features ? (!g_refl_composite || !g_alpha_composite) : (!g_albedo || !g_refl || !g_roughness))) return MRGS_E_BAD_ARG;
    // the scatter keys pack (level << 24 | texel index): a level with 6 res^2 >= 2^24 texels (res >= 1673) would alias into the level bits
    for (int l = 0; l < m.n; l++)
        if (m.grad[l] != nullptr && 6ll * m.res[l] * m.res[l] >= (1ll << 24)) return MRGS_E_UNSUPPORTED;
    ShadeCam cam;
    for (int i = 0; i < 9; i++) cam.Kinv[i] = fr->Kinv[i];
    cam.R = fr->R; cam.T = fr->T;
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n_cu = v;
        else n_cu = 256;
    }
    // LDS-resident levels: from the coarsest up while they fit
    int lds_floats = 0;
    for (int l = m.n - 1; l >= 0; l--) {
        const int n = 6 * m.res[l] * m.res[l] * 3;
        if (m.grad[l] == nullptr || lds_floats + n > MRGS_SHADE_LDS_FLOATS) break;
        m.lds_off[l] = lds_floats;
        lds_floats += n;
    }
    const int rows = MRGS_SHADE_FUSED_THREADS / 64;
    const int tiles_x = (fr->W + 63) / 64, ntiles = tiles_x * ((fr->H + rows - 1) / rows);
    const dim3 grid(ntiles < n_cu ? ntiles : n_cu), block(MRGS_SHADE_FUSED_THREADS);
    if (composite)
        hipLaunchKernelGGL((shade_fused_bwd_kernel<true>), grid, block, 0, (hipStream_t)stream, m, cam, fr->H, fr->W, to_map(fr->albedo), to_map(fr->normal),
                           to_map(fr->alpha), to_map(fr->refl), to_map(fr->roughness), fr->lut, fr->lut_res, g_specular, g_direct_light, g_specular_weight,
                           g_albedo, g_normal, g_alpha, g_refl, g_roughness, tiles_x, ntiles, lds_floats, g_features, g_refl_composite, g_alpha_composite,
                           composite->srgb, composite->base, composite->spec_fwd, composite->bg, composite->g_render, composite->g_diffuse, composite->g_base);
    else
        hipLaunchKernelGGL((shade_fused_bwd_kernel<false>), grid, block, 0, (hipStream_t)stream, m, cam, fr->H, fr->W, to_map(fr->albedo), to_map(fr->normal),
                           to_map(fr->alpha), to_map(fr->refl), to_map(fr->roughness), fr->lut, fr->lut_res, g_specular, g_direct_light, g_specular_weight,
                           g_albedo, g_normal, g_alpha, g_refl, g_roughness, tiles_x, ntiles, lds_floats, g_features, g_refl_composite, g_alpha_composite,
                           0, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (float*)nullptr);
    return hipGetLastError() == hipSuccess ? MRGS_OK : MRGS_E_HIP;
}

int mrgs_shade_specular_backward(const MrgsEnvMips* mips, const MrgsShadeFrame* fr, const float* g_specular, const float* g_direct_light,
                                 const float* g_specular_weight, float* g_albedo, float* g_normal, float* g_alpha, float* g_refl,
                                 float* g_roughness, void* stream)
{
    return shade_specular_backward_impl(mips, fr, g_specular, g_direct_light, g_specular_weight, g_albedo, g_normal, g_alpha, g_refl, g_roughness,
                                        nullptr, nullptr, nullptr, nullptr, stream);
}

int mrgs_shade_specular_backward_features(const MrgsEnvMips* mips, const MrgsShadeFrame* fr, const float* g_specular, const float* g_direct_light,
                                          const float* g_specular_weight, const float* g_refl_composite, const float* g_alpha_composite,
                                          float* g_normal, float* g_features, float* g_alpha, void* stream)
{
    if (!g_features) return MRGS_E_BAD_ARG;
    return shade_specular_backward_impl(mips, fr, g_specular, g_direct_light, g_specular_weight, nullptr, g_normal, g_alpha, nullptr, nullptr,
                                        g_features, g_refl_composite, g_alpha_composite, nullptr, stream);
}

int mrgs_surfel_shade_composite_backward(const MrgsEnvMips* mips, const MrgsShadeFrame* fr, int32_t srgb, const float* base_color, const float* specular,
                                         const float* bg, const float* g_render, const float* g_diffuse, const float* g_specular_extra,
                                         const float* g_direct_light, const float* g_specular_weight, float* g_base, float* g_normal, float* g_features,
                                         float* g_alpha, void* stream)
{
    if (!g_features) return MRGS_E_BAD_ARG;
    const ShadeCompositeArgs ca = {srgb, base_color, specular, bg, g_render, g_diffuse, g_base};
    return shade_specular_backward_impl(mips, fr, g_specular_extra, g_direct_light, g_specular_weight, nullptr, g_normal, g_alpha, nullptr, nullptr,
                                        g_features, nullptr, nullptr, &ca, stream);
}

int mrgs_cubemap_filter_count(int32_t res, int32_t kind, float roughness, float cos_cutoff, uint32_t* row_count, float* row_wsum, void* stream)
{
    if (res < 1 || res > 1024 || (kind != 0 && kind != 1) || !row_count || !row_wsum) return MRGS_E_BAD_ARG;
    const int NT = 6 * res * res;
    hipLaunchKernelGGL(cubemap_filter_build_kernel<0>, dim3((NT + 255) / 256), dim3(256), 0, (hipStream_t)stream, res, kind, roughness, cos_cutoff,
                       row_count, row_wsum, (const uint32_t*)nullptr, (uint32_t*)nullptr, (float*)nullptr);
    return hipGetLastError() == hipSuccess ? MRGS_OK : MRGS_E_HIP;
}

int mrgs_cubemap_filter_fill(int32_t res, int32_t kind, float roughness, float cos_cutoff, const uint32_t* row_ptr, const float* row_wsum,
                             uint32_t* col, float* val, void* stream)
{
    if (res < 1 || res > 1024 || (kind != 0 && kind != 1) || !row_ptr || !row_wsum || !col || !val) return MRGS_E_BAD_ARG;
    const int NT = 6 * res * res;
    hipLaunchKernelGGL(cubemap_filter_build_kernel<1>, dim3((NT + 255) / 256), dim3(256), 0, (hipStream_t)stream, res, kind, roughness, cos_cutoff,
                       (uint32_t*)nullptr, const_cast<float*>(row_wsum), row_ptr, col, val);
    return hipGetLastError() == hipSuccess ? MRGS_OK : MRGS_E_HIP;
}

int mrgs_csr_spmv3(int32_t nrows, const uint32_t* row_ptr, const void* col, int32_t col_bytes, const void* val, int32_t val_bytes,
                   const float* row_scale, const float* x, float* y, int32_t lanes_per_row, void* stream)
{
    if (nrows < 1 || !row_ptr || !col || !val || !x || !y || (col_bytes != 2 && col_bytes != 4) || (val_bytes != 2 && val_bytes != 4 && val_bytes != 8)) return MRGS_E_BAD_ARG;
    if (val_bytes != 4 && !row_scale) return MRGS_E_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (val_bytes == 8) {     // blocked rows: 16-bit block indices, 4 x 16-bit weights per block, x read in 16-byte pieces
        if (col_bytes != 2 || lanes_per_row < 64 || (nrows & 3) || ((uintptr_t)x & 15u) || ((uintptr_t)val & 7u)) return MRGS_E_BAD_ARG;
        hipLaunchKernelGGL(csr_spmv3_blk4_kernel, dim3((unsigned)(((size_t)nrows * 64 + 255) / 256)), dim3(256), 0, st, nrows, row_ptr, (const uint16_t*)col,
                           (const uint2*)val, row_scale, x, y);
        return hipGetLastError() == hipSuccess ? MRGS_OK : MRGS_E_HIP;
    }
    const bool wide = lanes_per_row >= 64;
    const dim3 g(wide ? (unsigned)(((size_t)nrows * 64 + 255) / 256) : (unsigned)(((size_t)nrows * 4 + 255) / 256)), b(256);
#define SPMV(G_, IDX_, WT_) hipLaunchKernelGGL((csr_spmv3_kernel<G_, IDX_, WT_>), g, b, 0, st, nrows, row_ptr, (const IDX_*)col, (const WT_*)val, row_scale, x, y)
    if (wide) {
        if (col_bytes == 2) { if (val_bytes == 2) SPMV(64, uint16_t, uint16_t); else SPMV(64, uint16_t, float); }
        else { if (val_bytes == 2) SPMV(64, uint32_t, uint16_t); else SPMV(64, uint32_t, float); }
    } else {
        if (col_bytes == 2) { if (val_bytes == 2) SPMV(4, uint16_t, uint16_t); else SPMV(4, uint16_t, float); }
        else { if (val_bytes == 2) SPMV(4, uint32_t, uint16_t); else SPMV(4, uint32_t, float); }
    }
#undef SPMV
    return hipGetLastError() == hipSuccess ? MRGS_OK : MRGS_E_HIP;
}

int mrgs_cube_symmetry_rows(int32_t res, int32_t* rows)
{
    if (res < 1 || res > 1024 || !rows) return MRGS_E_BAD_ARG;
    const int n = 6 * res * res;
    for (int g = 0; g < 48; ++g)
        for (int t = 0; t < n; ++t) {
            const CubeImage o = cube_symmetry_image(g, res, t / (res * res), (t / res) % res, t % res);
            rows[(size_t)g * n + t] = (o.s * res + o.y) * res + o.x;
        }
    return MRGS_OK;
}

int mrgs_csr_spmv3_batched(const MrgsSpmvDesc* descs, int32_t n, void* stream)
{
    if (!descs || n < 1 || n > MRGS_SPMV_MAX_BATCH) return MRGS_E_BAD_ARG;
    static SpmvBatch proto = [] { SpmvBatch b = {}; cube_symmetry_table(b.sym); return b; }();
    SpmvBatch B = proto;
    B.n = n;
    int blocks = 0;
    size_t lds = 0;
    for (int i = 0; i < n; ++i) {
        const MrgsSpmvDesc& d = descs[i];
        const bool sym = d.image_rows != nullptr;
        if (d.nrows < 1 || !d.val || !d.x || !d.y) return MRGS_E_BAD_ARG;
        SpmvSeg& S = B.seg[i];
        S.nrows = d.nrows;
        S.first_block = blocks;
        S.log2n = 0;
        S.row_ptr = d.row_ptr; S.col = d.col; S.val = d.val; S.row_scale = d.row_scale; S.x = d.x; S.y = d.y;
        S.image_rows = nullptr; S.pre = nullptr; S.tile_ptr = nullptr; S.panel_ptr = nullptr; S.panel_src = nullptr;
        if (sym) {
            // tiles of rows of one fundamental domain: a power-of-two face of 4 .. 128 texels (16-bit block indices), nrows = 6 res^2 texels
            int L = 0;
            while ((1 << L) < d.res) ++L;
            // (a wave of the product keeps the patch ids of its 128 steps in two registers: panels of at most 128 x its workgroup's waves)
            if (d.res < 4 || d.res > 128 || (1 << L) != d.res || d.nrows != 6 * d.res * d.res || d.n_tiles < 1 || !d.pre_scale || !d.row_scale ||
                !d.tile_ptr || !d.panel_ptr || !d.panel_src || ((uintptr_t)d.val & 7u) || d.max_panel < 1 || d.max_panel > MRGS_SPMV_MAX_PANEL)
                return MRGS_E_BAD_ARG;
            S.fmt = 16;
            S.log2n = L; S.image_rows = d.image_rows; S.pre = d.pre_scale; S.tile_ptr = d.tile_ptr; S.panel_ptr = d.panel_ptr; S.panel_src = d.panel_src;
            blocks += d.n_tiles * 12;
            lds = MRGS_SPMV_SYM_LDS;
            continue;
        }
        if (!d.row_ptr || !d.col || (d.col_bytes != 2 && d.col_bytes != 4) || (d.val_bytes != 2 && d.val_bytes != 4 && d.val_bytes != 8) ||
            (d.val_bytes != 4 && !d.row_scale))
            return MRGS_E_BAD_ARG;
        const bool blk4 = d.val_bytes == 8;
        if (blk4 && (d.col_bytes != 2 || d.lanes_per_row < 64 || (d.nrows & 3) || ((uintptr_t)d.x & 15u) || ((uintptr_t)d.val & 7u))) return MRGS_E_BAD_ARG;
        const bool wide = d.lanes_per_row >= 64;
        S.fmt = blk4 ? 8 : ((wide ? 4 : 0) | (d.col_bytes == 4 ? 2 : 0) | (d.val_bytes == 4 ? 1 : 0));
        blocks += (int)(((size_t)d.nrows * (wide ? 64 : 4) + MRGS_SPMV_BATCH_THREADS - 1) / MRGS_SPMV_BATCH_THREADS);
    }
    B.total_blocks = blocks;
    B.trace = nullptr;
#ifdef MRGS_SPMV_TRACE
    if (const char* tp = getenv("MRGS_SPMV_TRACE_BUF")) B.trace = (unsigned long long*)strtoull(tp, nullptr, 16);     // developer build: per-wave timestamps
#endif
    hipLaunchKernelGGL(csr_spmv3_batched_kernel, dim3((unsigned)blocks), dim3(MRGS_SPMV_BATCH_THREADS), lds, (hipStream_t)stream, B);
    return hipGetLastError() == hipSuccess ? MRGS_OK : MRGS_E_HIP;
}

int mrgs_cubemap_mip_chain_forward(int32_t res_in, int32_t n_steps, const float* in, float* const* outs, void* stream)
{
    if (res_in < 2 || n_steps < 1 || !in || !outs || (res_in >> n_steps) < 1 || ((res_in >> n_steps) << n_steps) != res_in) return MRGS_E_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const float* src = in;
    int res = res_in, done = 0;
    while (done < n_steps) {                                   // three levels per launch
        const int steps = n_steps - done < 3 ? n_steps - done : 3;
        MipOuts o = {{nullptr, nullptr, nullptr}};
        for (int k = 0; k < steps; ++k) {
            if (!outs[done + k]) return MRGS_E_BAD_ARG;
            o.o[k] = outs[done + k];
        }
        const int Nc = res >> steps, n = 6 * Nc * Nc * 3;
        if (steps == 3) hipLaunchKernelGGL(cubemap_mip_chain3_fwd_kernel, dim3((4 * n + 255) / 256), dim3(256), 0, st, res, src, o);
        else hipLaunchKernelGGL(cubemap_mip_chain_fwd_kernel, dim3((n + 255) / 256), dim3(256), 0, st, res, steps, src, o);
        src = outs[done + steps - 1];
        res = Nc;
        done += steps;
    }
    return hipGetLastError() == hipSuccess ? MRGS_OK : MRGS_E_HIP;
}

int mrgs_cubemap_mip_chain_backward(int32_t res0, int32_t n_levels, float* const* g, void* stream)
{
    if (res0 < 2 || n_levels < 1 || n_levels > MRGS_MAX_MIPS || !g || (res0 >> (n_levels - 1)) < 1) return MRGS_E_BAD_ARG;
    for (int k = 0; k < n_levels; ++k) if (!g[k]) return MRGS_E_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    // One launch per level, coarse to fine (3 x 6.6 us for 128..16).  Measured and dropped: the levels up to 64 x 64 in ONE workgroup behind
    // barriers (152 us: 24 576 texels x the cross-face tap arithmetic on one CU); one launch in which every level-0 texel gathers down
    // the pyramid (84 us: 21 tap set-ups per texel, 12.8 k instructions); and, round 5, ONE launch whose levels wait for each other on
    // device counters (blocks of a level behind those of the coarser one in the grid, one lane spinning, release / acquire at agent
    // scope between the levels: correct, bit-identical, also from two streams at once -- and 91 us: an agent-scope release or acquire
    // writes back or invalidates the XCD's WHOLE L2, once per block, in the middle of a backward pass whose other kernels' data it holds).
    for (int k = n_levels - 2; k >= 0; --k) {
        const int N = res0 >> k, n = 6 * N * N;
        hipLaunchKernelGGL(cubemap_mip_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, st, N, g[k + 1], g[k]);
    }
    return hipGetLastError() == hipSuccess ? MRGS_OK : MRGS_E_HIP;
}

int mrgs_cubemap_mip_forward(int32_t res_out, const float* in, float* out, void* stream)
{
    if (res_out < 1 || !in || !out) return MRGS_E_BAD_ARG;
    const int n = 6 * res_out * res_out * 3;
    hipLaunchKernelGGL(cubemap_mip_fwd_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, res_out, in, out);
    return hipGetLastError() == hipSuccess ? MRGS_OK : MRGS_E_HIP;
}

int mrgs_cubemap_mip_backward(int32_t res_fine, const float* dout, float* g_fine, void* stream)
{
    if (res_fine < 2 || (res_fine & 1) || !dout || !g_fine) return MRGS_E_BAD_ARG;
    const int n = 6 * res_fine * res_fine;
    hipLaunchKernelGGL(cubemap_mip_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, res_fine, dout, g_fine);
    return hipGetLastError() == hipSuccess ? MRGS_OK : MRGS_E_HIP;
}

}   // extern "C"
