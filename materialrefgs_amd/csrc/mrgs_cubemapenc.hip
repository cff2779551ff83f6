// mrgs_cubemapenc.hip -- the cubemapencoder fetch primitive on gfx950 (BASELINE.json north star: "the cubemapencoder mip lookup").
//
// Replaces the CUDA extension of /root/reference/submodules/cubemapencoder (src/cubemapencoder.cu): direction [B,3] ->
// features [C,B] fetched from a cubemap [6,C,L,L], nearest / bilinear / seamless bilinear (edge texels taken from the neighbouring
// face, the missing corner texel at a cube vertex = mean of the other three, :326-328), and the backward to the texels, the
// directions and the fail value.  Face convention and edge table: Compute_Cubemap_UV (:147-187) and EdgeTable (:66-105, the
// LEFT_TOP_AS_ORIGIN branch the reference compiles).  Same values, own decomposition: the tap set of a sample is resolved once
// (branch-light: the edge table is a 6 x 4 constant table of affine index maps instead of a 24-way if chain) and the channel loop
// runs over four precomputed texel offsets; one sample per lane, outputs [C,B] written coalesced across the wave.
#include "mrgs_internal.h"

namespace {

struct CmTaps {
    int off[4];       // texel offset inside one channel plane of its face: (face * C) * L * L is added per channel -> see tap_base
    int face[4];
    float w[4];       // blend weights of the four taps (vertex: three taps, the fourth weight is folded into them)
    float kx, ky;
    int flag;
    bool vertex;
};

// the texel across the edge `e` (0: u < 0, 1: u >= L, 2: v < 0, 3: v >= L) of face f, as (face, x, y) with x, y affine in the
// in-face coordinates: value = a * in_x + b * in_y + c0 + c1 * (L - 1)   (cubemapencoder.cu:66-105)
struct EdgeMap { signed char face, xa, xb, xc1, ya, yb, yc1; };
__device__ __constant__ EdgeMap kEdge[6][4] = {
    // face 0: +x
    {{4, 0, 0, 1, 0, 1, 0}, {5, 0, 0, 0, 0, 1, 0}, {3, 0, 0, 1, 1, 0, 0}, {2, 0, 0, 1, 1, 0, 0}},
    // face 1: -x
    {{5, 0, 0, 1, 0, 1, 0}, {4, 0, 0, 0, 0, 1, 0}, {3, 0, 0, 0, -1, 0, 1}, {2, 0, 0, 0, -1, 0, 1}},
    // face 2: +y
    {{1, 0, -1, 1, 0, 0, 1}, {0, 0, 1, 0, 0, 0, 1}, {4, 1, 0, 0, 0, 0, 1}, {5, -1, 0, 1, 0, 0, 1}},
    // face 3: -y
    {{1, 0, -1, 1, 0, 0, 0}, {0, 0, 1, 0, 0, 0, 0}, {4, 1, 0, 0, 0, 0, 0}, {5, -1, 0, 1, 0, 0, 0}},
    // face 4: +z
    {{1, 0, 0, 1, 0, 1, 0}, {0, 0, 0, 0, 0, 1, 0}, {3, 1, 0, 0, 0, 0, 0}, {2, 1, 0, 0, 0, 0, 0}},
    // face 5: -z
    {{0, 0, 0, 1, 0, 1, 0}, {1, 0, 0, 0, 0, 1, 0}, {3, -1, 0, 1, 0, 0, 1}, {2, -1, 0, 1, 0, 0, 1}},
};
__device__ __forceinline__ void edge_texel(int L, int f, int e, int x, int y, int& of, int& ox, int& oy)
{
    const EdgeMap m = kEdge[f][e];
    of = m.face;
    ox = m.xa * x + m.xb * y + m.xc1 * (L - 1);
    oy = m.ya * x + m.yb * y + m.yc1 * (L - 1);
}

// Compute_Cubemap_UV (:147-187)
__device__ __forceinline__ void dir_to_uv(float x, float y, float z, int& face, float& u, float& v)
{
    const float ax = fabsf(x), ay = fabsf(y), az = fabsf(z);
    int dim = 0;
    float mv = ax;
    if (ay > mv) { mv = ay; dim = 1; }
    if (az > mv) { mv = az; dim = 2; }
    if (dim == 0) {
        u = z / x; v = y / x;
        if (x >= 0.f) { face = 0; u = -u; v = -v; } else { face = 1; u = -u; }
    } else if (dim == 1) {
        u = x / y; v = z / y;
        if (y >= 0.f) { face = 2; } else { face = 3; u = -u; v = -v; }
    } else {
        u = x / z; v = y / z;
        if (z >= 0.f) { face = 4; v = -v; } else { face = 5; }
    }
}

// tap set of one sample: Compute_Seamless_Index (:189-262) / the clamped taps of Cubemap_Bilinear_Kernel (:357-378)
__device__ __forceinline__ CmTaps resolve_taps(int face, int L, float u, float v, bool seamless)
{
    CmTaps t;
    const float pu = (u * 0.5f + 0.5f) * (float)L;
    const float pv = (-v * 0.5f + 0.5f) * (float)L;           // LEFT_TOP_AS_ORIGIN
    int x0 = (int)floorf(pu - 0.5f), y0 = (int)floorf(pv - 0.5f);
    int x1 = x0 + 1, y1 = y0 + 1;
    float kx = pu - (float)x0 - 0.5f, ky = pv - (float)y0 - 0.5f;
    x0 = min(max(x0, 0), L - 1); x1 = min(max(x1, 0), L - 1);
    y0 = min(max(y0, 0), L - 1); y1 = min(max(y1, 0), L - 1);
    int flag = 0;
    if (seamless) {
        if (pu < 0.5f) { flag |= 1; kx = 0.5f - pu; } else if (pu >= (float)L - 0.5f) flag |= 2;
        if (pv < 0.5f) { flag |= 4; ky = 0.5f - pv; } else if (pv >= (float)L - 0.5f) flag |= 8;
    }
    const int eu = (flag & 2) ? 1 : 0, ev = (flag & 8) ? 3 : 2;   // edge index crossed in u / in v
    int f[4] = {face, face, face, face}, xs[4] = {x0, x1, x0, x1}, ys[4] = {y0, y0, y1, y1};
    t.vertex = (flag & 3) && (flag & 12);
    if (t.vertex) {
        xs[1] = x0; ys[1] = y0; xs[2] = x0; ys[2] = y0;
        edge_texel(L, face, eu, x0, y0, f[1], xs[1], ys[1]);
        edge_texel(L, face, ev, x0, y0, f[2], xs[2], ys[2]);
        f[3] = f[0]; xs[3] = x0; ys[3] = y0;                  // unused (weight folded below)
    } else if (flag & 3) {                                     // crossing a u edge: taps (x0,y0) | across, (x0,y1) | across
        xs[2] = x0; ys[2] = y1;
        edge_texel(L, face, eu, x0, y0, f[1], xs[1], ys[1]);
        edge_texel(L, face, eu, x0, y1, f[3], xs[3], ys[3]);
    } else if (flag & 12) {                                    // crossing a v edge: taps (x0,y0), (x1,y0) | across both
        edge_texel(L, face, ev, x0, y0, f[2], xs[2], ys[2]);
        edge_texel(L, face, ev, x1, y0, f[3], xs[3], ys[3]);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) { t.face[k] = f[k]; t.off[k] = ys[k] * L + xs[k]; }
    t.kx = kx; t.ky = ky; t.flag = flag;
    t.w[0] = (1.f - ky) * (1.f - kx); t.w[1] = (1.f - ky) * kx; t.w[2] = ky * (1.f - kx); t.w[3] = ky * kx;
    return t;
}

__device__ __forceinline__ void nearest_texel(int L, float u, float v, int& x, int& y)
{
    const float pu = (u * 0.5f + 0.5f) * (float)L, pv = (-v * 0.5f + 0.5f) * (float)L;
    x = min(max((int)pu, 0), L - 1);                           // int(): truncation toward zero (:407-411)
    y = min(max((int)pv, 0), L - 1);
}

__global__ void __launch_bounds__(256) cubemap_encode_fwd_kernel(const float* __restrict__ inputs, const float* __restrict__ cubemap,
                                                                 const float* __restrict__ fail_value, float* __restrict__ outputs, int interp,
                                                                 int seamless, long long B, int C, int L)
{
    const long long n = (long long)blockIdx.x * 256 + threadIdx.x;
    if (n >= B) return;
    const float vx = inputs[n * 3], vy = inputs[n * 3 + 1], vz = inputs[n * 3 + 2];
    if (vx == 0.f && vy == 0.f && vz == 0.f) {
        for (int c = 0; c < C; c++) outputs[(size_t)c * B + n] = fail_value[c];
        return;
    }
    int face; float u, v;
    dir_to_uv(vx, vy, vz, face, u, v);
    const size_t plane = (size_t)L * L;
    if (interp == 0) {
        int x, y;
        nearest_texel(L, u, v, x, y);
        const float* p = cubemap + (size_t)face * C * plane + (size_t)y * L + x;
        for (int c = 0; c < C; c++) outputs[(size_t)c * B + n] = p[c * plane];
        return;
    }
    const CmTaps t = resolve_taps(face, L, u, v, seamless != 0);
    const float* p0 = cubemap + (size_t)t.face[0] * C * plane + t.off[0];
    const float* p1 = cubemap + (size_t)t.face[1] * C * plane + t.off[1];
    const float* p2 = cubemap + (size_t)t.face[2] * C * plane + t.off[2];
    const float* p3 = cubemap + (size_t)t.face[3] * C * plane + t.off[3];
    for (int c = 0; c < C; c++) {
        const float v00 = p0[c * plane], v01 = p1[c * plane], v10 = p2[c * plane];
        const float v11 = t.vertex ? (v00 + v01 + v10) / 3.f : p3[c * plane];
        outputs[(size_t)c * B + n] = (1.f - t.ky) * ((1.f - t.kx) * v00 + t.kx * v01) + t.ky * ((1.f - t.kx) * v10 + t.kx * v11);
    }
}

__global__ void __launch_bounds__(256) cubemap_encode_bwd_kernel(const float* __restrict__ grad_outputs, const float* __restrict__ inputs,
                                                                 const float* __restrict__ cubemap, float* __restrict__ grad_cubemap,
                                                                 float* __restrict__ grad_inputs, float* __restrict__ grad_fail, int interp,
                                                                 int seamless, long long B, int C, int L)
{
    const long long n = (long long)blockIdx.x * 256 + threadIdx.x;
    if (n >= B) return;
    const float vx = inputs[n * 3], vy = inputs[n * 3 + 1], vz = inputs[n * 3 + 2];
    float gx = 0.f, gy = 0.f, gz = 0.f;
    if (vx == 0.f && vy == 0.f && vz == 0.f) {
        for (int c = 0; c < C; c++) atomicAdd(grad_fail + c, grad_outputs[(size_t)c * B + n]);
    } else {
        int face; float u, v;
        dir_to_uv(vx, vy, vz, face, u, v);
        const size_t plane = (size_t)L * L;
        if (interp == 0) {
            int x, y;
            nearest_texel(L, u, v, x, y);
            float* g = grad_cubemap + (size_t)face * C * plane + (size_t)y * L + x;
            for (int c = 0; c < C; c++) atomicAdd(g + c * plane, grad_outputs[(size_t)c * B + n]);
        } else {
            const CmTaps t = resolve_taps(face, L, u, v, seamless != 0);
            size_t base[4];
#pragma unroll
            for (int k = 0; k < 4; k++) base[k] = (size_t)t.face[k] * C * plane + t.off[k];
            const float extra = t.vertex ? t.ky * t.kx / 3.f : 0.f;
            float gu = 0.f, gv = 0.f;
            for (int c = 0; c < C; c++) {
                const float go = grad_outputs[(size_t)c * B + n];
                const float v00 = cubemap[base[0] + c * plane], v01 = cubemap[base[1] + c * plane], v10 = cubemap[base[2] + c * plane];
                const float v11 = t.vertex ? (v00 + v01 + v10) / 3.f : cubemap[base[3] + c * plane];
                atomicAdd(grad_cubemap + base[0] + c * plane, (t.w[0] + extra) * go);
                atomicAdd(grad_cubemap + base[1] + c * plane, (t.w[1] + extra) * go);
                atomicAdd(grad_cubemap + base[2] + c * plane, (t.w[2] + extra) * go);
                if (!t.vertex) atomicAdd(grad_cubemap + base[3] + c * plane, t.w[3] * go);
                float l0 = ((1.f - t.ky) * (v01 - v00) + t.ky * (v11 - v10)) * (0.5f * (float)L * go);
                float l1 = ((1.f - t.kx) * (v10 - v00) + t.kx * (v11 - v01)) * (0.5f * (float)L * go);
                if (t.flag & 1) l0 = -l0;
                if (t.flag & 4) l1 = -l1;
                gu += l0;
                gv += -l1;                                      // LEFT_TOP_AS_ORIGIN
            }
            // Compute_Cubemap_UV_Backward (:264-291), applied once to the channel sums (it is linear in the uv gradient)
            if (face < 2) {
                if (face == 0) { gu = -gu; gv = -gv; } else { gu = -gu; }
                gx = -(vz * gu + vy * gv) / (vx * vx); gy = gv / vx; gz = gu / vx;
            } else if (face < 4) {
                if (face == 3) { gu = -gu; gv = -gv; }
                gx = gu / vy; gy = -(vx * gu + vz * gv) / (vy * vy); gz = gv / vy;
            } else {
                if (face == 4) gv = -gv;
                gx = gu / vz; gy = gv / vz; gz = -(vx * gu + vy * gv) / (vz * vz);
            }
        }
    }
    grad_inputs[n * 3] = gx; grad_inputs[n * 3 + 1] = gy; grad_inputs[n * 3 + 2] = gz;
}

}   // namespace

extern "C" {

int mrgs_cubemap_encode_forward(const float* inputs, const float* cubemap, const float* fail_value, float* outputs, int32_t interp,
                                int32_t seamless, int64_t B, int32_t C, int32_t L, void* stream)
{
    if (B < 0 || C < 1 || L < 1 || interp < 0 || interp > 1) return MRGS_E_BAD_ARG;
    if (B == 0) return MRGS_OK;
    if (!inputs || !cubemap || !fail_value || !outputs) return MRGS_E_BAD_ARG;
    hipLaunchKernelGGL(cubemap_encode_fwd_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, inputs, cubemap, fail_value,
                       outputs, interp, seamless, (long long)B, C, L);
    return hipGetLastError() == hipSuccess ? MRGS_OK : MRGS_E_HIP;
}

int mrgs_cubemap_encode_backward(const float* grad_outputs, const float* inputs, const float* cubemap, float* grad_cubemap, float* grad_inputs,
                                 float* grad_fail, int32_t interp, int32_t seamless, int64_t B, int32_t C, int32_t L, void* stream)
{
    if (B < 0 || C < 1 || L < 1 || interp < 0 || interp > 1) return MRGS_E_BAD_ARG;
    if (B == 0) return MRGS_OK;
    if (!grad_outputs || !inputs || !cubemap || !grad_cubemap || !grad_inputs || !grad_fail) return MRGS_E_BAD_ARG;
    hipLaunchKernelGGL(cubemap_encode_bwd_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, grad_outputs, inputs, cubemap,
                       grad_cubemap, grad_inputs, grad_fail, interp, seamless, (long long)B, C, L);
    return hipGetLastError() == hipSuccess ? MRGS_OK : MRGS_E_HIP;
}

}   // extern "C"
