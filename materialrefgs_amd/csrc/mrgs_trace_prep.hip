// Per-surfel records of the surfel ray tracer from the model's tensors, and the way back (one launch each).
//
// What HardwareRendering.render_gaussians prepares around the tracer call (gaussian_renderer/optix_utils.py:124-183): the quad corners
// of get_disks (:36-66) for the hierarchy, the splat frame from scales / rotations, and -- when SHs are passed -- the colour of every
// surfel seen from the settings' camera position, computeColorFromSH of the rasterizer family (forward.cu:20-81: basis of degree
// <= 3 in the 3DGS sign convention, + 0.5, clamped at 0).  In torch that is ~130 small launches forward and as many backward per view
// (measured: 3.3 ms of a 14 ms traced view, host-bound); here it is one kernel each way:
//   geom  [P,16] = mean (3), r_u / s_u (3), r_v / s_v (3), normal r_w (3), opacity, 3 unused     (s = scale * scale_modifier)
//   attr  [P,8]  = rgb (3), others (2), 3 unused
//   quads [P,4,3] = mean -+ 3 s_u r_u +- 3 s_v r_v in get_disks' corner order (-3,3), (-3,-3), (3,3), (3,-3)
// R = rotation matrix of q / |q| (utils/general_utils.py:80-99: the reference normalises inside build_rotation).
// Backward: g_geom / g_attr (the tracer's outputs) -> gradients of means, scales, rotations (through the normalisation), opacities,
// SH coefficients or colours, others.  The quads carry no gradient (the reference detaches them, optix_utils.py:76).
#include "mrgs_internal.h"

namespace {

__device__ __constant__ float pSH_C0 = 0.28209479177387814f;
__device__ __constant__ float pSH_C1 = 0.4886025119029199f;
__device__ __constant__ float pSH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f,
                                           0.5462742152960396f};
__device__ __constant__ float pSH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                                           -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

struct PrepArgs {
    int P, M, degree;                     // M coefficients per channel in `shs` (layout [P,M,3]), active degree
    float scale_modifier;
    const float *means, *scales, *rotations, *opacities, *shs, *colors, *others, *campos;
    float *geom, *attr, *quads;
    const float *g_geom, *g_attr;
    float *g_means, *g_scales, *g_rotations, *g_opacities, *g_shs, *g_colors, *g_others;
    // raw mode (the model's own tensors, GaussianModel's activations applied here: scene/gaussian_model.py:56-78): scales = exp(raw),
    // opacities = sigmoid(raw), rotations raw anyway (make_rot normalises); SH split as the model stores it: shs = _features_dc [P,1,3],
    // shs_rest = _features_rest [P,15,3], gradients likewise
    int raw;
    const float* shs_rest;
    float* g_shs_rest;
};

#define PREST_L 45           // 15 coefficients x 3 channels of _features_rest
// 64 consecutive rows of 45 floats <-> a per-wave LDS tile with 16-byte global accesses (the run starts 16-byte aligned: the first row
// index is a multiple of 64; row stride 45 floats is odd, so lane-private rows are bank-conflict free); partial waves go scalar
__device__ __forceinline__ void prest_load(float* __restrict__ tile, const float* __restrict__ src, int nrows, int lane)
{
    if (nrows == 64) {
        constexpr int NF4 = 16 * PREST_L;
        const float4* s4 = reinterpret_cast<const float4*>(src);
#pragma unroll
        for (int k = 0; k * 64 < NF4; k++) {
            const int t = k * 64 + lane;
            if (t < NF4) {
                const float4 v = s4[t];
                tile[4 * t] = v.x; tile[4 * t + 1] = v.y; tile[4 * t + 2] = v.z; tile[4 * t + 3] = v.w;
            }
        }
    } else {
        for (int e = lane; e < nrows * PREST_L; e += 64) tile[e] = src[e];
    }
}
__device__ __forceinline__ void prest_store(const float* __restrict__ tile, float* __restrict__ dst, int nrows, int lane)
{
    if (nrows == 64) {
        constexpr int NF4 = 16 * PREST_L;
        float4* d4 = reinterpret_cast<float4*>(dst);
#pragma unroll
        for (int k = 0; k * 64 < NF4; k++) {
            const int t = k * 64 + lane;
            if (t < NF4) d4[t] = make_float4(tile[4 * t], tile[4 * t + 1], tile[4 * t + 2], tile[4 * t + 3]);
        }
    } else {
        for (int e = lane; e < nrows * PREST_L; e += 64) dst[e] = tile[e];
    }
}
__device__ __forceinline__ float psigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// basis values B[0..n) and their derivatives with respect to the unit direction (x, y, z)
__device__ __forceinline__ void sh_basis_and_grad(int degree, float x, float y, float z, float (&B)[16], float (&Bx)[16], float (&By)[16], float (&Bz)[16])
{
#pragma unroll
    for (int i = 0; i < 16; ++i) { B[i] = 0.f; Bx[i] = 0.f; By[i] = 0.f; Bz[i] = 0.f; }
    B[0] = pSH_C0;
    if (degree < 1) return;
    B[1] = -pSH_C1 * y; By[1] = -pSH_C1;
    B[2] = pSH_C1 * z; Bz[2] = pSH_C1;
    B[3] = -pSH_C1 * x; Bx[3] = -pSH_C1;
    if (degree < 2) return;
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    B[4] = pSH_C2[0] * xy; Bx[4] = pSH_C2[0] * y; By[4] = pSH_C2[0] * x;
    B[5] = pSH_C2[1] * yz; By[5] = pSH_C2[1] * z; Bz[5] = pSH_C2[1] * y;
    B[6] = pSH_C2[2] * (2.0f * zz - xx - yy); Bx[6] = -2.0f * pSH_C2[2] * x; By[6] = -2.0f * pSH_C2[2] * y; Bz[6] = 4.0f * pSH_C2[2] * z;
    B[7] = pSH_C2[3] * xz; Bx[7] = pSH_C2[3] * z; Bz[7] = pSH_C2[3] * x;
    B[8] = pSH_C2[4] * (xx - yy); Bx[8] = 2.0f * pSH_C2[4] * x; By[8] = -2.0f * pSH_C2[4] * y;
    if (degree < 3) return;
    B[9] = pSH_C3[0] * y * (3.0f * xx - yy); Bx[9] = pSH_C3[0] * 6.0f * xy; By[9] = pSH_C3[0] * 3.0f * (xx - yy);
    B[10] = pSH_C3[1] * xy * z; Bx[10] = pSH_C3[1] * yz; By[10] = pSH_C3[1] * xz; Bz[10] = pSH_C3[1] * xy;
    B[11] = pSH_C3[2] * y * (4.0f * zz - xx - yy); Bx[11] = -2.0f * pSH_C3[2] * xy; By[11] = pSH_C3[2] * (4.0f * zz - xx - 3.0f * yy); Bz[11] = 8.0f * pSH_C3[2] * yz;
    B[12] = pSH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy); Bx[12] = -6.0f * pSH_C3[3] * xz; By[12] = -6.0f * pSH_C3[3] * yz;
    Bz[12] = pSH_C3[3] * (6.0f * zz - 3.0f * xx - 3.0f * yy);
    B[13] = pSH_C3[4] * x * (4.0f * zz - xx - yy); Bx[13] = pSH_C3[4] * (4.0f * zz - 3.0f * xx - yy); By[13] = -2.0f * pSH_C3[4] * xy; Bz[13] = 8.0f * pSH_C3[4] * xz;
    B[14] = pSH_C3[5] * z * (xx - yy); Bx[14] = 2.0f * pSH_C3[5] * xz; By[14] = -2.0f * pSH_C3[5] * yz; Bz[14] = pSH_C3[5] * (xx - yy);
    B[15] = pSH_C3[6] * x * (xx - 3.0f * yy); Bx[15] = pSH_C3[6] * 3.0f * (xx - yy); By[15] = -6.0f * pSH_C3[6] * xy;
}

struct Rot { float qn[4], len, R[3][3]; };

__device__ __forceinline__ Rot make_rot(const float4 q)
{
    Rot r;
    r.len = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    const float w = q.x / r.len, x = q.y / r.len, y = q.z / r.len, z = q.w / r.len;
    r.qn[0] = w; r.qn[1] = x; r.qn[2] = y; r.qn[3] = z;
    r.R[0][0] = 1.f - 2.f * (y * y + z * z); r.R[0][1] = 2.f * (x * y - w * z); r.R[0][2] = 2.f * (x * z + w * y);
    r.R[1][0] = 2.f * (x * y + w * z); r.R[1][1] = 1.f - 2.f * (x * x + z * z); r.R[1][2] = 2.f * (y * z - w * x);
    r.R[2][0] = 2.f * (x * z - w * y); r.R[2][1] = 2.f * (y * z + w * x); r.R[2][2] = 1.f - 2.f * (x * x + y * y);
    return r;
}

template <bool BWD, bool SPLIT>
__global__ __launch_bounds__(256) void trace_prep_kernel(PrepArgs A)
{
    __shared__ float s_rest[SPLIT ? 4 : 1][SPLIT ? 64 * PREST_L : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wave_base = blockIdx.x * 256 + wave * 64;
    if (wave_base >= A.P) return;                           // wave-uniform
    const int nrows = min(64, A.P - wave_base);
    const bool valid = lane < nrows;                         // lanes past the end stay for the cooperative row transfers
    const int p = valid ? wave_base + lane : A.P - 1;
    float* rest = s_rest[SPLIT ? wave : 0];
    if (SPLIT) {
        prest_load(rest, A.shs_rest + (size_t)wave_base * PREST_L, nrows, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // lanes read rows other lanes wrote
        __builtin_amdgcn_wave_barrier();
    }
    const float mx = A.means[3 * p], my = A.means[3 * p + 1], mz = A.means[3 * p + 2];
    const float s_u = A.raw ? expf(A.scales[2 * p]) : A.scales[2 * p], s_v = A.raw ? expf(A.scales[2 * p + 1]) : A.scales[2 * p + 1];
    const float su = s_u * A.scale_modifier, sv = s_v * A.scale_modifier;
    const Rot r = make_rot(reinterpret_cast<const float4*>(A.rotations)[p]);
    const float ru[3] = {r.R[0][0], r.R[1][0], r.R[2][0]}, rv[3] = {r.R[0][1], r.R[1][1], r.R[2][1]}, rw[3] = {r.R[0][2], r.R[1][2], r.R[2][2]};
    // view direction and colour
    float dirx = 0.f, diry = 0.f, dirz = 0.f, dlen = 1.f;
    float B[16], Bx[16], By[16], Bz[16];
    float rgb[3] = {0.f, 0.f, 0.f};
    const bool from_sh = A.shs != nullptr;
    if (from_sh) {
        const float ex = mx - A.campos[0], ey = my - A.campos[1], ez = mz - A.campos[2];
        dlen = sqrtf(ex * ex + ey * ey + ez * ez);
        dirx = ex / dlen; diry = ey / dlen; dirz = ez / dlen;
        sh_basis_and_grad(A.degree, dirx, diry, dirz, B, Bx, By, Bz);
    }
    const int ncoef = min((A.degree + 1) * (A.degree + 1), A.M);
    if (!BWD) {
        if (from_sh) {
            const float* sh = A.shs + (size_t)p * (SPLIT ? 1 : A.M) * 3;
            if (SPLIT) {              // DC from its own tensor, the 45 "rest" floats of the row from the wave's LDS tile
                float row[48];
                row[0] = sh[0]; row[1] = sh[1]; row[2] = sh[2];
#pragma unroll
                for (int i = 0; i < PREST_L; ++i) row[3 + i] = rest[lane * PREST_L + i];
#pragma unroll
                for (int k = 0; k < 16; ++k) { rgb[0] += B[k] * row[3 * k]; rgb[1] += B[k] * row[3 * k + 1]; rgb[2] += B[k] * row[3 * k + 2]; }   // B[k] = 0 beyond the degree
            } else if (A.M == 16) {          // the 192-byte row with twelve 16-byte loads (a 4-byte load per coefficient walks 48 cache lines per wave)
                float row[48];
                const float4* sh4 = reinterpret_cast<const float4*>(sh);
#pragma unroll
                for (int i = 0; i < 12; ++i) { const float4 v = sh4[i]; row[4 * i] = v.x; row[4 * i + 1] = v.y; row[4 * i + 2] = v.z; row[4 * i + 3] = v.w; }
#pragma unroll
                for (int k = 0; k < 16; ++k) { rgb[0] += B[k] * row[3 * k]; rgb[1] += B[k] * row[3 * k + 1]; rgb[2] += B[k] * row[3 * k + 2]; }   // B[k] = 0 beyond the degree
            } else
            for (int k = 0; k < ncoef; ++k) { rgb[0] += B[k] * sh[3 * k]; rgb[1] += B[k] * sh[3 * k + 1]; rgb[2] += B[k] * sh[3 * k + 2]; }
#pragma unroll
            for (int c = 0; c < 3; ++c) rgb[c] = fmaxf(rgb[c] + 0.5f, 0.0f);
        } else {
            rgb[0] = A.colors[3 * p]; rgb[1] = A.colors[3 * p + 1]; rgb[2] = A.colors[3 * p + 2];
        }
        if (!valid) return;                                  // (no cooperative work follows in the forward)
        float4* g = reinterpret_cast<float4*>(A.geom) + (size_t)p * 4;
        g[0] = make_float4(mx, my, mz, ru[0] / su);
        g[1] = make_float4(ru[1] / su, ru[2] / su, rv[0] / sv, rv[1] / sv);
        g[2] = make_float4(rv[2] / sv, rw[0], rw[1], rw[2]);
        g[3] = make_float4(A.raw ? psigmoid(A.opacities[p]) : A.opacities[p], 0.f, 0.f, 0.f);
        float4* a = reinterpret_cast<float4*>(A.attr) + (size_t)p * 2;
        a[0] = make_float4(rgb[0], rgb[1], rgb[2], A.others ? A.others[2 * p] : 0.f);
        a[1] = make_float4(A.others ? A.others[2 * p + 1] : 0.f, 0.f, 0.f, 0.f);
        if (A.quads) {
            float* q = A.quads + (size_t)p * 12;
            const float cu[4] = {-3.f, -3.f, 3.f, 3.f}, cv[4] = {3.f, -3.f, 3.f, -3.f};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                q[3 * k] = mx + cu[k] * su * ru[0] + cv[k] * sv * rv[0];
                q[3 * k + 1] = my + cu[k] * su * ru[1] + cv[k] * sv * rv[1];
                q[3 * k + 2] = mz + cu[k] * su * ru[2] + cv[k] * sv * rv[2];
            }
        }
        return;
    }
    // ---- backward ----
    const float4* gg = reinterpret_cast<const float4*>(A.g_geom) + (size_t)p * 4;
    const float4 g0 = gg[0], g1 = gg[1], g2 = gg[2], g3 = gg[3];
    const float4* ga = reinterpret_cast<const float4*>(A.g_attr) + (size_t)p * 2;
    const float4 a0 = ga[0], a1 = ga[1];
    float dm[3] = {g0.x, g0.y, g0.z};
    const float da[3] = {g0.w, g1.x, g1.y}, db[3] = {g1.z, g1.w, g2.x}, dn[3] = {g2.y, g2.z, g2.w};
    // a = r_u / s_u: d r_u = da / s_u, d s_u = -(da . r_u) / s_u^2
    float dR[3][3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { dR[k][0] = da[k] / su; dR[k][1] = db[k] / sv; dR[k][2] = dn[k]; }
    const float dsu = -(da[0] * ru[0] + da[1] * ru[1] + da[2] * ru[2]) / (su * su), dsv = -(db[0] * rv[0] + db[1] * rv[1] + db[2] * rv[2]) / (sv * sv);
    // d/d activated scale = dsu * modifier; through exp in raw mode: * exp(raw) = * s_u
    if (valid) {
        A.g_scales[2 * p] = dsu * A.scale_modifier * (A.raw ? s_u : 1.0f);
        A.g_scales[2 * p + 1] = dsv * A.scale_modifier * (A.raw ? s_v : 1.0f);
    }
    // rotation matrix -> unit quaternion -> raw quaternion
    const float w = r.qn[0], x = r.qn[1], y = r.qn[2], z = r.qn[3];
    const float dw = 2.f * (-z * dR[0][1] + y * dR[0][2] + z * dR[1][0] - x * dR[1][2] - y * dR[2][0] + x * dR[2][1]);
    const float dx = 2.f * (y * dR[0][1] + z * dR[0][2] + y * dR[1][0] - 2.f * x * dR[1][1] - w * dR[1][2] + z * dR[2][0] + w * dR[2][1] - 2.f * x * dR[2][2]);
    const float dy = 2.f * (-2.f * y * dR[0][0] + x * dR[0][1] + w * dR[0][2] + x * dR[1][0] + z * dR[1][2] - w * dR[2][0] + z * dR[2][1] - 2.f * y * dR[2][2]);
    const float dz = 2.f * (-2.f * z * dR[0][0] - w * dR[0][1] + x * dR[0][2] + w * dR[1][0] - 2.f * z * dR[1][1] + y * dR[1][2] + x * dR[2][0] + y * dR[2][1]);
    const float dot = w * dw + x * dx + y * dy + z * dz;
    if (valid) {
        reinterpret_cast<float4*>(A.g_rotations)[p] = make_float4((dw - w * dot) / r.len, (dx - x * dot) / r.len, (dy - y * dot) / r.len, (dz - z * dot) / r.len);
        if (A.raw) { const float o = psigmoid(A.opacities[p]); A.g_opacities[p] = g3.x * o * (1.0f - o); }
        else A.g_opacities[p] = g3.x;
        if (A.g_others) { A.g_others[2 * p] = a0.w; A.g_others[2 * p + 1] = a1.x; }
    }
    if (from_sh) {
        const float* sh = A.shs + (size_t)p * (SPLIT ? 1 : A.M) * 3;
        float* gsh = A.g_shs + (size_t)p * (SPLIT ? 1 : A.M) * 3;
        float val[3] = {0.5f, 0.5f, 0.5f};
        float ddx = 0.f, ddy = 0.f, ddz = 0.f;
        if (SPLIT || A.M == 16) {     // vector loads and stores of the two 192-byte rows (split: DC apart, the rest through the LDS tile)
            float row[48];
            if (SPLIT) {
                row[0] = sh[0]; row[1] = sh[1]; row[2] = sh[2];
#pragma unroll
                for (int i = 0; i < PREST_L; ++i) row[3 + i] = rest[lane * PREST_L + i];
            } else {
                const float4* sh4 = reinterpret_cast<const float4*>(sh);
#pragma unroll
                for (int i = 0; i < 12; ++i) { const float4 v = sh4[i]; row[4 * i] = v.x; row[4 * i + 1] = v.y; row[4 * i + 2] = v.z; row[4 * i + 3] = v.w; }
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) { val[0] += B[k] * row[3 * k]; val[1] += B[k] * row[3 * k + 1]; val[2] += B[k] * row[3 * k + 2]; }
            const float dcv[3] = {val[0] >= 0.f ? a0.x : 0.f, val[1] >= 0.f ? a0.y : 0.f, val[2] >= 0.f ? a0.z : 0.f};
            float out[48];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                out[3 * k] = B[k] * dcv[0]; out[3 * k + 1] = B[k] * dcv[1]; out[3 * k + 2] = B[k] * dcv[2];
                const float sdot = row[3 * k] * dcv[0] + row[3 * k + 1] * dcv[1] + row[3 * k + 2] * dcv[2];
                ddx += Bx[k] * sdot; ddy += By[k] * sdot; ddz += Bz[k] * sdot;
            }
            if (SPLIT) {
                if (valid) { gsh[0] = out[0]; gsh[1] = out[1]; gsh[2] = out[2]; }
                // (the lane's row of the tile is its own: read above, overwritten here, no other lane touches it in between)
#pragma unroll
                for (int i = 0; i < PREST_L; ++i) rest[lane * PREST_L + i] = out[3 + i];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                prest_store(rest, A.g_shs_rest + (size_t)wave_base * PREST_L, nrows, lane);
            } else if (valid) {
                float4* g4 = reinterpret_cast<float4*>(gsh);
#pragma unroll
                for (int i = 0; i < 12; ++i) g4[i] = make_float4(out[4 * i], out[4 * i + 1], out[4 * i + 2], out[4 * i + 3]);
            }
        } else if (valid) {
        for (int k = 0; k < ncoef; ++k) { val[0] += B[k] * sh[3 * k]; val[1] += B[k] * sh[3 * k + 1]; val[2] += B[k] * sh[3 * k + 2]; }
        const float dc[3] = {val[0] >= 0.f ? a0.x : 0.f, val[1] >= 0.f ? a0.y : 0.f, val[2] >= 0.f ? a0.z : 0.f};
        for (int k = 0; k < A.M; ++k) {
            const bool live = k < ncoef;
            gsh[3 * k] = live ? B[k] * dc[0] : 0.f; gsh[3 * k + 1] = live ? B[k] * dc[1] : 0.f; gsh[3 * k + 2] = live ? B[k] * dc[2] : 0.f;
            if (live) {
                const float s = sh[3 * k] * dc[0] + sh[3 * k + 1] * dc[1] + sh[3 * k + 2] * dc[2];
                ddx += Bx[k] * s; ddy += By[k] * s; ddz += Bz[k] * s;
            }
        }
        }
        // dir = e / |e|
        const float dd = dirx * ddx + diry * ddy + dirz * ddz;
        dm[0] += (ddx - dirx * dd) / dlen; dm[1] += (ddy - diry * dd) / dlen; dm[2] += (ddz - dirz * dd) / dlen;
    } else if (A.g_colors && valid) {
        A.g_colors[3 * p] = a0.x; A.g_colors[3 * p + 1] = a0.y; A.g_colors[3 * p + 2] = a0.z;
    }
    if (valid) { A.g_means[3 * p] = dm[0]; A.g_means[3 * p + 1] = dm[1]; A.g_means[3 * p + 2] = dm[2]; }
}


// ---- the mirror rays of a rendered view ------------------------------------------------------------------------------------------
// render_indirect / render_surfel_with_envgs (gaussian_renderer/envgs_renderer.py:717-724, __init__.py:496-505): per pixel the surface
// point rays_o + surf_depth * rays_cam (rays_cam: un-normalised pixel ray, sample_camera_rays_unnormalize utils/refl_utils.py:75-93),
// the mirror direction of the view ray about `normal` (reflection :95-98, safe_normalize before and after), origin moved 1e-3 along it.
// ~20 torch launches each way otherwise.  Same camera conventions as bvh_visibility_kernel (Kinv host, R = Camera.R, T = Camera.T).
struct MirrorArgs {
    int H, W;
    float Kinv[9];
    const float *R, *T;
    const float* normal; long long nh, nw, nc;      // [H,W,3], element strides
    const float* alpha;                              // [H,W] or NULL.  Given: `normal` is the blended normal and the reflecting normal is
                                                     // safe_normalize(normal / max(alpha, 1e-6)) (gaussian_renderer/__init__.py:493-495)
    float* g_alpha;                                  // [H,W] (with alpha)
    const float* depth;                              // [H,W]
    float *ray_o, *ray_d;                            // [H,W,3]
    const float *g_ray_o, *g_ray_d;
    float *g_normal, *g_depth;                       // [H,W,3] contiguous, [H,W]
};

template <bool BWD>
__global__ __launch_bounds__(256) void mirror_rays_kernel(MirrorArgs A)
{
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= A.W || y >= A.H) return;
    const size_t pix = (size_t)y * A.W + x;
    const float fx = (float)x, fy = (float)y;
    const float pcx = A.Kinv[0] * fx + A.Kinv[1] * fy + A.Kinv[2], pcy = A.Kinv[3] * fx + A.Kinv[4] * fy + A.Kinv[5],
                pcz = A.Kinv[6] * fx + A.Kinv[7] * fy + A.Kinv[8];
    const float* R = A.R;
    const float tx = A.T[0], ty = A.T[1], tz = A.T[2];
    const float qx = pcx - tx, qy = pcy - ty, qz = pcz - tz;
    const float rox = -(R[0] * tx + R[1] * ty + R[2] * tz), roy = -(R[3] * tx + R[4] * ty + R[5] * tz), roz = -(R[6] * tx + R[7] * ty + R[8] * tz);
    const float cx = (R[0] * qx + R[1] * qy + R[2] * qz) - rox, cy = (R[3] * qx + R[4] * qy + R[5] * qz) - roy,
                cz = (R[6] * qx + R[7] * qy + R[8] * qz) - roz;                          // rays_cam (un-normalised)
    const float cl = fmaxf(sqrtf(cx * cx + cy * cy + cz * cz), 1e-20f);
    const float wx = -cx / cl, wy = -cy / cl, wz = -cz / cl;                            // w_o
    const long long on = (long long)y * A.nh + (long long)x * A.nw;
    float nx = A.normal[on], ny = A.normal[on + A.nc], nz = A.normal[on + 2 * A.nc];
    float ua = 1.f, ul = 1.f, qx_ = 0.f, qy_ = 0.f, qz_ = 0.f;   // with alpha: q = raw / max(alpha, 1e-6), n = q / max(|q|, 1e-20)
    float al = 1.f;
    if (A.alpha) {
        al = A.alpha[pix];
        ua = fmaxf(al, 1e-6f);
        qx_ = nx / ua; qy_ = ny / ua; qz_ = nz / ua;
        ul = fmaxf(sqrtf(qx_ * qx_ + qy_ * qy_ + qz_ * qz_), 1e-20f);
        nx = qx_ / ul; ny = qy_ / ul; nz = qz_ / ul;
    }
    const float ndv = wx * nx + wy * ny + wz * nz;
    const float ax = 2.f * nx * ndv - wx, ay = 2.f * ny * ndv - wy, az = 2.f * nz * ndv - wz;
    const float len = sqrtf(ax * ax + ay * ay + az * az), rl = fmaxf(len, 1e-20f);
    const float rx = ax / rl, ry = ay / rl, rz = az / rl;
    if (!BWD) {
        const float sd = A.depth[pix];
        A.ray_o[3 * pix] = (rox + sd * cx) + 1e-3f * rx; A.ray_o[3 * pix + 1] = (roy + sd * cy) + 1e-3f * ry; A.ray_o[3 * pix + 2] = (roz + sd * cz) + 1e-3f * rz;
        A.ray_d[3 * pix] = rx; A.ray_d[3 * pix + 1] = ry; A.ray_d[3 * pix + 2] = rz;
        return;
    }
    const float gox = A.g_ray_o[3 * pix], goy = A.g_ray_o[3 * pix + 1], goz = A.g_ray_o[3 * pix + 2];
    A.g_depth[pix] = gox * cx + goy * cy + goz * cz;
    const float grx = A.g_ray_d[3 * pix] + 1e-3f * gox, gry = A.g_ray_d[3 * pix + 1] + 1e-3f * goy, grz = A.g_ray_d[3 * pix + 2] + 1e-3f * goz;
    float gax = 0.f, gay = 0.f, gaz = 0.f;                   // through x / max(|x|, eps): the clamp is flat below eps
    if (len > 1e-20f) {
        const float dot = rx * grx + ry * gry + rz * grz;
        gax = (grx - rx * dot) / rl; gay = (gry - ry * dot) / rl; gaz = (grz - rz * dot) / rl;
    } else {
        gax = grx / rl; gay = gry / rl; gaz = grz / rl;
    }
    // a = 2 n (w_o . n) - w_o
    const float ng = nx * gax + ny * gay + nz * gaz;
    float gnx = 2.f * (ndv * gax + ng * wx), gny = 2.f * (ndv * gay + ng * wy), gnz = 2.f * (ndv * gaz + ng * wz);
    if (A.alpha) {
        // n = q / max(|q|, eps): projection when the clamp is not active; q = raw / max(alpha, 1e-6)
        float gqx, gqy, gqz;
        const bool unit = sqrtf(qx_ * qx_ + qy_ * qy_ + qz_ * qz_) > 1e-20f;
        if (unit) {
            const float d = nx * gnx + ny * gny + nz * gnz;
            gqx = (gnx - nx * d) / ul; gqy = (gny - ny * d) / ul; gqz = (gnz - nz * d) / ul;
        } else {
            gqx = gnx / ul; gqy = gny / ul; gqz = gnz / ul;
        }
        gnx = gqx / ua; gny = gqy / ua; gnz = gqz / ua;
        // q / |q| does not depend on alpha (the projected gradient is orthogonal to q: autograd's fp32 evaluation leaves rounding noise
        // there, scaled by 1 / alpha); alpha only matters where the normalisation is clamped, and clamp_min passes nothing below its bound
        A.g_alpha[pix] = (al > 1e-6f && !unit) ? -(gqx * qx_ + gqy * qy_ + gqz * qz_) / ua : 0.f;
    }
    A.g_normal[3 * pix] = gnx; A.g_normal[3 * pix + 1] = gny; A.g_normal[3 * pix + 2] = gnz;
}

// render_surfel_with_envgs' last line (gaussian_renderer/__init__.py:517): out = a (1 - s) + s b per channel, a / b [3,H,W], s [1,H,W]
// b and s by element strides (the tracer hands out [H,W,C] tensors seen as [C,H,W]: channel stride 1, pixel stride C)
__global__ __launch_bounds__(256) void traced_blend_fwd_kernel(int HW, const float* __restrict__ a, const float* __restrict__ b, long long b_cs, long long b_ps,
                                                               const float* __restrict__ s, long long s_ps, float* __restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= HW) return;
    const float w = s[i * s_ps];
#pragma unroll
    for (int c = 0; c < 3; ++c) out[c * HW + i] = a[c * HW + i] * (1.f - w) + w * b[c * b_cs + i * b_ps];
}
__global__ __launch_bounds__(256) void traced_blend_bwd_kernel(int HW, const float* __restrict__ a, const float* __restrict__ b, long long b_cs, long long b_ps,
                                                               const float* __restrict__ s, long long s_ps, const float* __restrict__ g,
                                                               float* __restrict__ ga, float* __restrict__ gb, float* __restrict__ gs)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= HW) return;
    const float w = s[i * s_ps];
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float gc = g[c * HW + i];
        ga[c * HW + i] = gc * (1.f - w);
        gb[c * b_cs + i * b_ps] = gc * w;          // g_b in b's own layout
        acc += gc * (b[c * b_cs + i * b_ps] - a[c * HW + i]);
    }
    gs[i] = acc;
}

}   // namespace

extern "C" {

int mrgs_surfel_trace_prep_forward(int64_t P, const float* means3D, const float* scales, const float* rotations, const float* opacities,
                                   const float* shs, int32_t M, int32_t sh_degree, const float* colors_precomp, const float* others,
                                   const float* campos, float scale_modifier, float* geom, float* attr, float* quad_vertices, void* stream)
{
    if (P < 0 || P > (int64_t)1 << 24) return MRGS_E_UNSUPPORTED;
    if (P == 0) return MRGS_OK;
    if (!means3D || !scales || !rotations || !opacities || !geom || !attr || (shs == nullptr) == (colors_precomp == nullptr)) return MRGS_E_BAD_ARG;
    if (shs && (!campos || M < 1 || M > 16 || sh_degree < 0 || sh_degree > 3)) return MRGS_E_BAD_ARG;
    PrepArgs a = {};
    a.P = (int)P; a.M = M; a.degree = sh_degree; a.scale_modifier = scale_modifier;
    a.means = means3D; a.scales = scales; a.rotations = rotations; a.opacities = opacities; a.shs = shs; a.colors = colors_precomp; a.others = others;
    a.campos = campos; a.geom = geom; a.attr = attr; a.quads = quad_vertices;
    hipLaunchKernelGGL((trace_prep_kernel<false, false>), dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? MRGS_OK : MRGS_E_HIP;
}

int mrgs_surfel_trace_prep_backward(int64_t P, const float* means3D, const float* scales, const float* rotations, const float* shs, int32_t M,
                                    int32_t sh_degree, const float* campos, float scale_modifier, const float* g_geom, const float* g_attr,
                                    float* g_means3D, float* g_scales, float* g_rotations, float* g_opacities, float* g_shs,
                                    float* g_colors_precomp, float* g_others, void* stream)
{
    if (P < 0 || P > (int64_t)1 << 24) return MRGS_E_UNSUPPORTED;
    if (P == 0) return MRGS_OK;
    if (!means3D || !scales || !rotations || !g_geom || !g_attr || !g_means3D || !g_scales || !g_rotations || !g_opacities) return MRGS_E_BAD_ARG;
    if (shs && (!campos || !g_shs || M < 1 || M > 16 || sh_degree < 0 || sh_degree > 3)) return MRGS_E_BAD_ARG;
    PrepArgs a = {};
    a.P = (int)P; a.M = M; a.degree = sh_degree; a.scale_modifier = scale_modifier;
    a.means = means3D; a.scales = scales; a.rotations = rotations; a.shs = shs; a.campos = campos;
    a.g_geom = g_geom; a.g_attr = g_attr; a.g_means = g_means3D; a.g_scales = g_scales; a.g_rotations = g_rotations; a.g_opacities = g_opacities;
    a.g_shs = g_shs; a.g_colors = g_colors_precomp; a.g_others = g_others;
    hipLaunchKernelGGL((trace_prep_kernel<true, false>), dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? MRGS_OK : MRGS_E_HIP;
}

int mrgs_surfel_trace_prep_raw_forward(int64_t P, const float* xyz, const float* scaling_raw, const float* rotation_raw, const float* opacity_raw,
                                       const float* features_dc, const float* features_rest, int32_t sh_degree, const float* others,
                                       const float* campos, float scale_modifier, float* geom, float* attr, float* quad_vertices, void* stream)
{
    if (P < 0 || P > (int64_t)1 << 24) return MRGS_E_UNSUPPORTED;
    if (P == 0) return MRGS_OK;
    if (!xyz || !scaling_raw || !rotation_raw || !opacity_raw || !features_dc || !features_rest || !campos || !geom || !attr) return MRGS_E_BAD_ARG;
    if (sh_degree < 0 || sh_degree > 3) return MRGS_E_BAD_ARG;
    PrepArgs a = {};
    a.P = (int)P; a.M = 16; a.degree = sh_degree; a.scale_modifier = scale_modifier; a.raw = 1;
    a.means = xyz; a.scales = scaling_raw; a.rotations = rotation_raw; a.opacities = opacity_raw; a.shs = features_dc; a.shs_rest = features_rest;
    a.others = others; a.campos = campos; a.geom = geom; a.attr = attr; a.quads = quad_vertices;
    hipLaunchKernelGGL((trace_prep_kernel<false, true>), dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? MRGS_OK : MRGS_E_HIP;
}

int mrgs_surfel_trace_prep_raw_backward(int64_t P, const float* xyz, const float* scaling_raw, const float* rotation_raw, const float* opacity_raw,
                                        const float* features_dc, const float* features_rest, int32_t sh_degree, const float* campos,
                                        float scale_modifier, const float* g_geom, const float* g_attr, float* g_xyz, float* g_scaling_raw,
                                        float* g_rotation_raw, float* g_opacity_raw, float* g_features_dc, float* g_features_rest, float* g_others,
                                        void* stream)
{
    if (P < 0 || P > (int64_t)1 << 24) return MRGS_E_UNSUPPORTED;
    if (P == 0) return MRGS_OK;
    if (!xyz || !scaling_raw || !rotation_raw || !opacity_raw || !features_dc || !features_rest || !campos || !g_geom || !g_attr || !g_xyz ||
        !g_scaling_raw || !g_rotation_raw || !g_opacity_raw || !g_features_dc || !g_features_rest)
        return MRGS_E_BAD_ARG;
    if (sh_degree < 0 || sh_degree > 3) return MRGS_E_BAD_ARG;
    PrepArgs a = {};
    a.P = (int)P; a.M = 16; a.degree = sh_degree; a.scale_modifier = scale_modifier; a.raw = 1;
    a.means = xyz; a.scales = scaling_raw; a.rotations = rotation_raw; a.opacities = opacity_raw; a.shs = features_dc; a.shs_rest = features_rest;
    a.campos = campos; a.g_geom = g_geom; a.g_attr = g_attr; a.g_means = g_xyz; a.g_scales = g_scaling_raw; a.g_rotations = g_rotation_raw;
    a.g_opacities = g_opacity_raw; a.g_shs = g_features_dc; a.g_shs_rest = g_features_rest; a.g_others = g_others;
    hipLaunchKernelGGL((trace_prep_kernel<true, true>), dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? MRGS_OK : MRGS_E_HIP;
}

static void mirror_fill(MirrorArgs& a, int32_t H, int32_t W, const float* Kinv_host, const float* R, const float* T, const MrgsStridedMap* normal)
{
    a.H = H; a.W = W;
    for (int i = 0; i < 9; ++i) a.Kinv[i] = Kinv_host[i];
    a.R = R; a.T = T;
    a.normal = normal->ptr; a.nh = normal->stride_h; a.nw = normal->stride_w; a.nc = normal->stride_c;
}

int mrgs_traced_blend_forward(int32_t H, int32_t W, const float* a, const float* b, int64_t b_channel_stride, int64_t b_pixel_stride, const float* s,
                              int64_t s_pixel_stride, float* out, void* stream)
{
    if (H <= 0 || W <= 0) return MRGS_OK;
    if (!a || !b || !s || !out) return MRGS_E_BAD_ARG;
    const int HW = H * W;
    hipLaunchKernelGGL(traced_blend_fwd_kernel, dim3((HW + 255) / 256), dim3(256), 0, (hipStream_t)stream, HW, a, b, (long long)b_channel_stride,
                       (long long)b_pixel_stride, s, (long long)s_pixel_stride, out);
    return hipGetLastError() == hipSuccess ? MRGS_OK : MRGS_E_HIP;
}

int mrgs_traced_blend_backward(int32_t H, int32_t W, const float* a, const float* b, int64_t b_channel_stride, int64_t b_pixel_stride, const float* s,
                               int64_t s_pixel_stride, const float* g_out, float* g_a, float* g_b, float* g_s, void* stream)
{
    if (H <= 0 || W <= 0) return MRGS_OK;
    if (!a || !b || !s || !g_out || !g_a || !g_b || !g_s) return MRGS_E_BAD_ARG;
    const int HW = H * W;
    hipLaunchKernelGGL(traced_blend_bwd_kernel, dim3((HW + 255) / 256), dim3(256), 0, (hipStream_t)stream, HW, a, b, (long long)b_channel_stride,
                       (long long)b_pixel_stride, s, (long long)s_pixel_stride, g_out, g_a, g_b, g_s);
    return hipGetLastError() == hipSuccess ? MRGS_OK : MRGS_E_HIP;
}

int mrgs_mirror_rays_blended_forward(int32_t H, int32_t W, const float* Kinv_host, const float* R, const float* T, const MrgsStridedMap* rend_normal,
                                     const float* alpha, const float* surf_depth, float* ray_o, float* ray_d, void* stream)
{
    if (H <= 0 || W <= 0) return MRGS_OK;
    if (!Kinv_host || !R || !T || !rend_normal || !rend_normal->ptr || !alpha || !surf_depth || !ray_o || !ray_d) return MRGS_E_BAD_ARG;
    MirrorArgs a = {};
    mirror_fill(a, H, W, Kinv_host, R, T, rend_normal);
    a.alpha = alpha; a.depth = surf_depth; a.ray_o = ray_o; a.ray_d = ray_d;
    hipLaunchKernelGGL(mirror_rays_kernel<false>, dim3((W + 31) / 32, (H + 7) / 8), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? MRGS_OK : MRGS_E_HIP;
}

int mrgs_mirror_rays_blended_backward(int32_t H, int32_t W, const float* Kinv_host, const float* R, const float* T, const MrgsStridedMap* rend_normal,
                                      const float* alpha, const float* g_ray_o, const float* g_ray_d, float* g_rend_normal, float* g_alpha,
                                      float* g_surf_depth, void* stream)
{
    if (H <= 0 || W <= 0) return MRGS_OK;
    if (!Kinv_host || !R || !T || !rend_normal || !rend_normal->ptr || !alpha || !g_ray_o || !g_ray_d || !g_rend_normal || !g_alpha || !g_surf_depth)
        return MRGS_E_BAD_ARG;
    MirrorArgs a = {};
    mirror_fill(a, H, W, Kinv_host, R, T, rend_normal);
    a.alpha = alpha; a.g_alpha = g_alpha;
    a.g_ray_o = g_ray_o; a.g_ray_d = g_ray_d; a.g_normal = g_rend_normal; a.g_depth = g_surf_depth;
    hipLaunchKernelGGL(mirror_rays_kernel<true>, dim3((W + 31) / 32, (H + 7) / 8), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? MRGS_OK : MRGS_E_HIP;
}

int mrgs_mirror_rays_forward(int32_t H, int32_t W, const float* Kinv_host, const float* R, const float* T, const MrgsStridedMap* normal,
                             const float* surf_depth, float* ray_o, float* ray_d, void* stream)
{
    if (H <= 0 || W <= 0) return MRGS_OK;
    if (!Kinv_host || !R || !T || !normal || !normal->ptr || !surf_depth || !ray_o || !ray_d) return MRGS_E_BAD_ARG;
    MirrorArgs a = {};
    mirror_fill(a, H, W, Kinv_host, R, T, normal);
    a.depth = surf_depth; a.ray_o = ray_o; a.ray_d = ray_d;
    hipLaunchKernelGGL(mirror_rays_kernel<false>, dim3((W + 31) / 32, (H + 7) / 8), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? MRGS_OK : MRGS_E_HIP;
}

int mrgs_mirror_rays_backward(int32_t H, int32_t W, const float* Kinv_host, const float* R, const float* T, const MrgsStridedMap* normal,
                              const float* g_ray_o, const float* g_ray_d, float* g_normal, float* g_surf_depth, void* stream)
{
    if (H <= 0 || W <= 0) return MRGS_OK;
    if (!Kinv_host || !R || !T || !normal || !normal->ptr || !g_ray_o || !g_ray_d || !g_normal || !g_surf_depth) return MRGS_E_BAD_ARG;
    MirrorArgs a = {};
    mirror_fill(a, H, W, Kinv_host, R, T, normal);
    a.g_ray_o = g_ray_o; a.g_ray_d = g_ray_d; a.g_normal = g_normal; a.g_depth = g_surf_depth;
    hipLaunchKernelGGL(mirror_rays_kernel<true>, dim3((W + 31) / 32, (H + 7) / 8), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? MRGS_OK : MRGS_E_HIP;
}

}   // extern "C"
