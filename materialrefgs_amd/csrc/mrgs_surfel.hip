// mrgs_surfel.hip -- per-gaussian inputs of the surfel renderer in one kernel (forward) and one kernel (backward).
//
// Replaces the ~50 small torch kernels the reference runs per view before it calls the rasterizer
// (SURVEY.md section 8a rows 1 and 10):
//   GaussianModel getters (scene/gaussian_model.py:236-311): opacity = sigmoid, scaling = exp, rotation = normalize,
//     refl / roughness / ori_color = sigmoid, get_indirect = cat(_indirect_dc, _indirect_rest);
//   get_normal (gaussian_model.py:269-285): third column of R(q) (q normalised inside build_rotation,
//     utils/general_utils.py:78-100), flip_align_view (:184-190), safe_normalize (:178-181);
//   render_surfel (gaussian_renderer/__init__.py:338-355): view direction, mirror direction
//     r = 2 (n . w_o) n - w_o, indirect = clamp_min(eval_sh(3, indirect_sh, r), 0) (utils/sh_utils.py:57-112),
//     features = cat(refl, roughness, ori_color, indirect)  ("2dgs" flavour, S = 8).
// The backward is the exact derivative of this forward (flip sign and clamp mask treated as constants, as autograd does).
//
// One gaussian per lane.  The 15x3 "rest" coefficients of a wave's 64 gaussians are one contiguous 11.5 KB run: it is
// moved with 16-byte loads/stores through a per-wave LDS tile (odd row stride, conflict-free), in both directions.
#include "mrgs_internal.h"

namespace {

__device__ __constant__ float cSH_C0 = 0.28209479177387814f;
__device__ __constant__ float cSH_C1 = 0.4886025119029199f;
__device__ __constant__ float cSH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f,
                                           0.5462742152960396f};
__device__ __constant__ float cSH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                                           -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

#define REST_L 45            // 15 coefficients x 3 channels
#define REST_STRIDE 45       // odd: lane-private rows are bank-conflict free

// 64 consecutive rows of L floats <-> LDS tile, 16 bytes per lane and instruction (the run starts 16-byte aligned because
// the first row index is a multiple of 64); partial waves take the scalar path
template <int L, int STRIDE = REST_STRIDE>
__device__ __forceinline__ void tile_load(float* __restrict__ tile, const float* __restrict__ src, int nrows, int lane)
{
    if (nrows == 64) {
        constexpr int NF4 = 16 * L;
        const float4* s4 = reinterpret_cast<const float4*>(src);
#pragma unroll
        for (int k = 0; k * 64 < NF4; k++) {
            const int t = k * 64 + lane;
            if (t < NF4) {
                const float4 v = s4[t];
                const float a[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int e = 4 * t + i;
                    tile[(e / L) * STRIDE + (e % L)] = a[i];
                }
            }
        }
    } else {
        for (int e = lane; e < nrows * L; e += 64) tile[(e / L) * STRIDE + (e % L)] = src[e];
    }
}
template <int L, int STRIDE = REST_STRIDE>
__device__ __forceinline__ void tile_store(const float* __restrict__ tile, float* __restrict__ dst, int nrows, int lane)
{
    if (nrows == 64) {
        constexpr int NF4 = 16 * L;
        float4* d4 = reinterpret_cast<float4*>(dst);
#pragma unroll
        for (int k = 0; k * 64 < NF4; k++) {
            const int t = k * 64 + lane;
            if (t < NF4) {
                float a[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int e = 4 * t + i;
                    a[i] = tile[(e / L) * STRIDE + (e % L)];
                }
                d4[t] = make_float4(a[0], a[1], a[2], a[3]);
            }
        }
    } else {
        for (int e = lane; e < nrows * L; e += 64) dst[e] = tile[(e / L) * STRIDE + (e % L)];
    }
}

__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

// real SH basis, degree 3, the 3DGS sign convention (utils/sh_utils.py:57-112)
__device__ __forceinline__ void sh_basis16(float x, float y, float z, float (&B)[16])
{
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    B[0] = cSH_C0;
    B[1] = -cSH_C1 * y; B[2] = cSH_C1 * z; B[3] = -cSH_C1 * x;
    B[4] = cSH_C2[0] * xy; B[5] = cSH_C2[1] * yz; B[6] = cSH_C2[2] * (2.0f * zz - xx - yy); B[7] = cSH_C2[3] * xz;
    B[8] = cSH_C2[4] * (xx - yy);
    B[9] = cSH_C3[0] * y * (3.0f * xx - yy); B[10] = cSH_C3[1] * xy * z; B[11] = cSH_C3[2] * y * (4.0f * zz - xx - yy);
    B[12] = cSH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy); B[13] = cSH_C3[4] * x * (4.0f * zz - xx - yy);
    B[14] = cSH_C3[5] * z * (xx - yy); B[15] = cSH_C3[6] * x * (xx - 3.0f * yy);
}

struct Frame {               // everything the forward derives from (xyz, q, campos) and the backward needs again
    float qlen, qn[4];       // |q|, q / |q| (w, x, y, z)
    float nr[3];             // third column of R(q)
    float flip, nflen, nn[3];// facing sign, |nf|, unit normal
    float dlen, v[3];        // |xyz - campos|, unit view direction
    float c, r[3];           // n . w_o, mirror direction
};

__device__ __forceinline__ Frame make_frame(const float p[3], const float4 q, const float* __restrict__ campos)
{
    Frame f;
    f.qlen = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    f.qn[0] = q.x / f.qlen; f.qn[1] = q.y / f.qlen; f.qn[2] = q.z / f.qlen; f.qn[3] = q.w / f.qlen;
    const float w = f.qn[0], x = f.qn[1], y = f.qn[2], z = f.qn[3];
    f.nr[0] = 2.0f * (x * z + w * y);
    f.nr[1] = 2.0f * (y * z - w * x);
    f.nr[2] = 1.0f - 2.0f * (x * x + y * y);
    const float d[3] = {p[0] - campos[0], p[1] - campos[1], p[2] - campos[2]};
    f.dlen = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    f.v[0] = d[0] / f.dlen; f.v[1] = d[1] / f.dlen; f.v[2] = d[2] / f.dlen;
    const float dotp = -(f.nr[0] * f.v[0] + f.nr[1] * f.v[1] + f.nr[2] * f.v[2]);
    f.flip = dotp >= 0.0f ? 1.0f : -1.0f;
    const float nf[3] = {f.nr[0] * f.flip, f.nr[1] * f.flip, f.nr[2] * f.flip};
    f.nflen = fmaxf(sqrtf(nf[0] * nf[0] + nf[1] * nf[1] + nf[2] * nf[2]), 1e-20f);
    f.nn[0] = nf[0] / f.nflen; f.nn[1] = nf[1] / f.nflen; f.nn[2] = nf[2] / f.nflen;
    f.c = -(f.nn[0] * f.v[0] + f.nn[1] * f.v[1] + f.nn[2] * f.v[2]);       // n . w_o, w_o = -v
    f.r[0] = 2.0f * f.c * f.nn[0] + f.v[0];
    f.r[1] = 2.0f * f.c * f.nn[1] + f.v[1];
    f.r[2] = 2.0f * f.c * f.nn[2] + f.v[2];
    return f;
}

// get_distance (gaussian_renderer/envgs_renderer.py:30-38): normal_cam = n @ Wv[:3,:3], centre_cam = p @ Wv[:3,:3] + Wv[3,:3] with the
// world_view_transform as stored; the distance is |normal_cam . centre_cam|.
struct PlaneDist { float nc[3], cc[3], s; };
__device__ __forceinline__ PlaneDist plane_distance(const float* __restrict__ Wv, const float (&n)[3], const float (&p)[3])
{
    PlaneDist d;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        d.nc[j] = n[0] * Wv[j] + n[1] * Wv[4 + j] + n[2] * Wv[8 + j];
        d.cc[j] = p[0] * Wv[j] + p[1] * Wv[4 + j] + p[2] * Wv[8 + j] + Wv[12 + j];
    }
    d.s = d.nc[0] * d.cc[0] + d.nc[1] * d.cc[1] + d.nc[2] * d.cc[2];
    return d;
}

__global__ void __launch_bounds__(256) surfel_features_fwd_kernel(MrgsSurfelParams prm, float* __restrict__ opacity, float* __restrict__ scales,
                                                                  float* __restrict__ rotations, float* __restrict__ features)
{
    __shared__ float s_rest[4][64 * REST_STRIDE];
    const int P = prm.P;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = blockIdx.x * 256 + wave * 64;
    const int nrows = min(64, P - row0);
    if (nrows <= 0) return;
    float* tile = s_rest[wave];
    tile_load<REST_L>(tile, prm.indirect_rest + (size_t)row0 * REST_L, nrows, lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int idx = row0 + lane;
    if (idx >= P) return;

    const float p[3] = {prm.xyz[3 * (size_t)idx], prm.xyz[3 * (size_t)idx + 1], prm.xyz[3 * (size_t)idx + 2]};
    const float4 q = reinterpret_cast<const float4*>(prm.rotation_raw)[idx];
    const Frame f = make_frame(p, q, prm.campos);

    float B[16];
    sh_basis16(f.r[0], f.r[1], f.r[2], B);
    const float* row = tile + lane * REST_STRIDE;
    float ind[3];
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        float s = prm.indirect_dc[3 * (size_t)idx + ch] * B[0];
#pragma unroll
        for (int k = 1; k < 16; k++) s += row[(k - 1) * 3 + ch] * B[k];
        ind[ch] = fmaxf(s, 0.0f);
    }
    opacity[idx] = sigmoidf(prm.opacity_raw[idx]);
    const float2 sr = reinterpret_cast<const float2*>(prm.scaling_raw)[idx];
    reinterpret_cast<float2*>(scales)[idx] = make_float2(expf(sr.x), expf(sr.y));
    // torch.nn.functional.normalize: x / max(|x|, 1e-12)
    const float rl = fmaxf(f.qlen, 1e-12f);
    reinterpret_cast<float4*>(rotations)[idx] = make_float4(q.x / rl, q.y / rl, q.z / rl, q.w / rl);
    const int f4 = prm.viewmatrix != nullptr ? 3 : 2;          // float4 per feature row
    float4* fo = reinterpret_cast<float4*>(features) + f4 * (size_t)idx;
    fo[0] = make_float4(sigmoidf(prm.refl_raw[idx]), sigmoidf(prm.rough_raw[idx]), sigmoidf(prm.ori_color_raw[3 * (size_t)idx]),
                        sigmoidf(prm.ori_color_raw[3 * (size_t)idx + 1]));
    fo[1] = make_float4(sigmoidf(prm.ori_color_raw[3 * (size_t)idx + 2]), ind[0], ind[1], ind[2]);
    if (prm.viewmatrix != nullptr) fo[2] = make_float4(fabsf(plane_distance(prm.viewmatrix, f.nn, p).s), 0.0f, 0.0f, 0.0f);
}

#ifndef MRGS_FEAT_BWD_WAVES
#define MRGS_FEAT_BWD_WAVES 3        // 46 KB of LDS per workgroup: three workgroups (twelve waves) per CU, if the registers allow it
#endif
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MRGS_FEAT_BWD_WAVES, 8))) surfel_features_bwd_kernel(MrgsSurfelParams prm, const float* __restrict__ g_opacity,
                                                                  const float* __restrict__ g_scales, const float* __restrict__ g_rotations,
                                                                  const float* __restrict__ g_features, MrgsSurfelGrads out,
                                                                  const float* __restrict__ g_xyz_upstream)
{
    __shared__ float s_rest[4][64 * REST_STRIDE];
    const int P = prm.P;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = blockIdx.x * 256 + wave * 64;
    const int nrows = min(64, P - row0);
    if (nrows <= 0) return;
    float* tile = s_rest[wave];
    const int idx_ = row0 + lane;
    const bool in_range = idx_ < P;
    const int idx = in_range ? idx_ : P - 1;     // out-of-range lanes stay for the cooperative tile store
    const int f4 = prm.viewmatrix != nullptr ? 3 : 2;          // float4 per feature row
    const float4 gf0 = g_features ? reinterpret_cast<const float4*>(g_features)[f4 * (size_t)idx] : make_float4(0, 0, 0, 0);
    const float4 gf1 = g_features ? reinterpret_cast<const float4*>(g_features)[f4 * (size_t)idx + 1] : make_float4(0, 0, 0, 0);
    // The indirect radiance takes a gradient only where something looked at it (opt.indirect, a loss on "indirect_light"): where the
    // upstream gradient of its three channels is exactly zero for all 64 rows of the wave, every term it feeds is an exact zero -- the
    // 45 coefficient gradients, the DC's, the mirror direction's -- and the wave neither reads its 11.5 KB of coefficients nor evaluates
    // the basis: it writes the zeros (54 of the kernel's 97 MB of reads at 300 k surfels when no row has one).  Same values as the full
    // path for finite coefficients (0 x inf would be NaN there).
    const bool sh_live = __builtin_amdgcn_ballot_w64(in_range && (gf1.y != 0.0f || gf1.z != 0.0f || gf1.w != 0.0f)) != 0ull;
    if (sh_live) {
        tile_load<REST_L>(tile, prm.indirect_rest + (size_t)row0 * REST_L, nrows, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }

    const float p[3] = {prm.xyz[3 * (size_t)idx], prm.xyz[3 * (size_t)idx + 1], prm.xyz[3 * (size_t)idx + 2]};
    const float4 q = reinterpret_cast<const float4*>(prm.rotation_raw)[idx];
    const Frame f = make_frame(p, q, prm.campos);
    // "pgsr": the plane distance |nc . cc| (plane_distance) sends its gradient to the facing normal and to the centre
    float d_nn_pd[3] = {0.0f, 0.0f, 0.0f}, d_p_pd[3] = {0.0f, 0.0f, 0.0f};
    if (prm.viewmatrix != nullptr && g_features != nullptr) {
        const float gd = reinterpret_cast<const float4*>(g_features)[f4 * (size_t)idx + 2].x;
        const PlaneDist pd = plane_distance(prm.viewmatrix, f.nn, p);
        const float sg = pd.s > 0.0f ? gd : (pd.s < 0.0f ? -gd : 0.0f);           // d|s| = sign(s) (torch: 0 at 0)
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const float* Wi = prm.viewmatrix + 4 * i;
            d_nn_pd[i] = sg * (Wi[0] * pd.cc[0] + Wi[1] * pd.cc[1] + Wi[2] * pd.cc[2]);
            d_p_pd[i] = sg * (Wi[0] * pd.nc[0] + Wi[1] * pd.nc[1] + Wi[2] * pd.nc[2]);
        }
    }

    // ---- indirect = clamp_min(sum_k sh[k][ch] B_k(r), 0) ----
    const float x = f.r[0], y = f.r[1], z = f.r[2];
    float* row = tile + lane * REST_STRIDE;
    const float gin[3] = {gf1.y, gf1.z, gf1.w};
    float d_r[3] = {0.0f, 0.0f, 0.0f};
    float d_dc[3] = {0.0f, 0.0f, 0.0f};
    if (sh_live) {
        float B[16];
        sh_basis16(x, y, z, B);
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            const float dc = prm.indirect_dc[3 * (size_t)idx + ch];
            float sh[16];
            sh[0] = dc;
#pragma unroll
            for (int k = 1; k < 16; k++) sh[k] = row[(k - 1) * 3 + ch];
            float s = 0.0f;
#pragma unroll
            for (int k = 0; k < 16; k++) s += sh[k] * B[k];
            const float g = s >= 0.0f ? gin[ch] : 0.0f;          // clamp_min passes the gradient where x >= min
            d_dc[ch] = g * B[0];
#pragma unroll
            for (int k = 1; k < 16; k++) row[(k - 1) * 3 + ch] = g * B[k];   // in place: coefficients of this channel were read above
            // d(sum_k sh_k B_k)/d(x,y,z)
            float dx = -cSH_C1 * sh[3] + cSH_C2[0] * y * sh[4] + cSH_C2[2] * -2.0f * x * sh[6] + cSH_C2[3] * z * sh[7] +
                       cSH_C2[4] * 2.0f * x * sh[8];
            float dy = -cSH_C1 * sh[1] + cSH_C2[0] * x * sh[4] + cSH_C2[1] * z * sh[5] + cSH_C2[2] * -2.0f * y * sh[6] +
                       cSH_C2[4] * -2.0f * y * sh[8];
            float dz = cSH_C1 * sh[2] + cSH_C2[1] * y * sh[5] + cSH_C2[2] * 4.0f * z * sh[6] + cSH_C2[3] * x * sh[7];
            dx += cSH_C3[0] * sh[9] * 6.0f * xy + cSH_C3[1] * sh[10] * yz + cSH_C3[2] * sh[11] * -2.0f * xy +
                  cSH_C3[3] * sh[12] * -6.0f * xz + cSH_C3[4] * sh[13] * (4.0f * zz - 3.0f * xx - yy) + cSH_C3[5] * sh[14] * 2.0f * xz +
                  cSH_C3[6] * sh[15] * 3.0f * (xx - yy);
            dy += cSH_C3[0] * sh[9] * 3.0f * (xx - yy) + cSH_C3[1] * sh[10] * xz + cSH_C3[2] * sh[11] * (4.0f * zz - xx - 3.0f * yy) +
                  cSH_C3[3] * sh[12] * -6.0f * yz + cSH_C3[4] * sh[13] * -2.0f * xy + cSH_C3[5] * sh[14] * -2.0f * yz +
                  cSH_C3[6] * sh[15] * -6.0f * xy;
            dz += cSH_C3[1] * sh[10] * xy + cSH_C3[2] * sh[11] * 8.0f * yz + cSH_C3[3] * sh[12] * (6.0f * zz - 3.0f * xx - 3.0f * yy) +
                  cSH_C3[4] * sh[13] * 8.0f * xz + cSH_C3[5] * sh[14] * (xx - yy);
            d_r[0] += g * dx; d_r[1] += g * dy; d_r[2] += g * dz;
        }
    }
    if (sh_live) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        tile_store<REST_L>(tile, out.d_indirect_rest + (size_t)row0 * REST_L, nrows, lane);
    } else {
        float* dst = out.d_indirect_rest + (size_t)row0 * REST_L;
        if (nrows == 64 && (((uintptr_t)dst) & 15u) == 0) {          // the wave's 64 rows are one run of 64 * 45 floats, 16-byte aligned when the tensor is
            float4* d4 = reinterpret_cast<float4*>(dst);
#pragma unroll
            for (int k = 0; k * 64 < 16 * REST_L; k++)
                if (k * 64 + lane < 16 * REST_L) d4[k * 64 + lane] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        } else {
            for (int e = lane; e < nrows * REST_L; e += 64) dst[e] = 0.0f;
        }
    }
    if (!in_range) return;
#pragma unroll
    for (int ch = 0; ch < 3; ch++) out.d_indirect_dc[3 * (size_t)idx + ch] = d_dc[ch];

    // ---- r = 2 c n + v, c = -(n . v) (w_o = -v) ----
    const float n_dot_dr = f.nn[0] * d_r[0] + f.nn[1] * d_r[1] + f.nn[2] * d_r[2];
    float d_nn[3], d_v[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        d_nn[i] = 2.0f * f.c * d_r[i] - 2.0f * n_dot_dr * f.v[i] + d_nn_pd[i];       // 2 c d_r + 2 (n . d_r) w_o (+ the plane distance's share)
        d_v[i] = d_r[i] - 2.0f * n_dot_dr * f.nn[i];                      // -(d_wo), d_wo = 2 (n . d_r) n - d_r
    }
    // n = nf / |nf|, nf = nr * flip
    const float nn_dot = f.nn[0] * d_nn[0] + f.nn[1] * d_nn[1] + f.nn[2] * d_nn[2];
    float a[3];
#pragma unroll
    for (int i = 0; i < 3; i++) a[i] = (d_nn[i] - f.nn[i] * nn_dot) / f.nflen * f.flip;
    // nr(qn): nr0 = 2 (x z + w y), nr1 = 2 (y z - w x), nr2 = 1 - 2 (x^2 + y^2)
    const float w = f.qn[0], qx = f.qn[1], qy = f.qn[2], qz = f.qn[3];
    float d_qn[4];
    d_qn[0] = 2.0f * qy * a[0] - 2.0f * qx * a[1];
    d_qn[1] = 2.0f * qz * a[0] - 2.0f * w * a[1] - 4.0f * qx * a[2];
    d_qn[2] = 2.0f * w * a[0] + 2.0f * qz * a[1] - 4.0f * qy * a[2];
    d_qn[3] = 2.0f * qx * a[0] + 2.0f * qy * a[1];
    // qn = q / |q| (build_rotation), plus the `rotations` output = normalize(q) with upstream g_rotations
    const float4 gr = g_rotations ? reinterpret_cast<const float4*>(g_rotations)[idx] : make_float4(0, 0, 0, 0);
    const float grv[4] = {gr.x, gr.y, gr.z, gr.w};
    float dot1 = 0.0f, dot2 = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; i++) { dot1 += f.qn[i] * d_qn[i]; dot2 += f.qn[i] * grv[i]; }
    const float rl = fmaxf(f.qlen, 1e-12f);
    float d_q[4];
#pragma unroll
    for (int i = 0; i < 4; i++) d_q[i] = (d_qn[i] - f.qn[i] * dot1) / f.qlen + (grv[i] - f.qn[i] * dot2) / rl;
    reinterpret_cast<float4*>(out.d_rotation)[idx] = make_float4(d_q[0], d_q[1], d_q[2], d_q[3]);
    // v = d / |d|
    const float v_dot = f.v[0] * d_v[0] + f.v[1] * d_v[1] + f.v[2] * d_v[2];
#pragma unroll
    for (int i = 0; i < 3; i++)           // + what reached the centres some other way (the rasterizer's dL/dmeans3D): one sum here instead of a kernel of its own
        out.d_xyz[3 * (size_t)idx + i] = (g_xyz_upstream ? g_xyz_upstream[3 * (size_t)idx + i] : 0.0f) + (d_v[i] - f.v[i] * v_dot) / f.dlen + d_p_pd[i];

    // ---- activations ----
    const float so = sigmoidf(prm.opacity_raw[idx]);
    out.d_opacity[idx] = (g_opacity ? g_opacity[idx] : 0.0f) * so * (1.0f - so);
    const float2 sr = reinterpret_cast<const float2*>(prm.scaling_raw)[idx];
    const float2 gs = g_scales ? reinterpret_cast<const float2*>(g_scales)[idx] : make_float2(0, 0);
    reinterpret_cast<float2*>(out.d_scaling)[idx] = make_float2(gs.x * expf(sr.x), gs.y * expf(sr.y));
    const float s0 = sigmoidf(prm.refl_raw[idx]), s1 = sigmoidf(prm.rough_raw[idx]);
    out.d_refl[idx] = gf0.x * s0 * (1.0f - s0);
    out.d_rough[idx] = gf0.y * s1 * (1.0f - s1);
    const float go[3] = {gf0.z, gf0.w, gf1.x};
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const float s = sigmoidf(prm.ori_color_raw[3 * (size_t)idx + i]);
        out.d_ori_color[3 * (size_t)idx + i] = go[i] * s * (1.0f - s);
    }
}

// Sum over views of the SH colour gradients, rebuilt from each view's masked colour gradient (materialrefgs_amd/dist.py):
// dL/dsh[k][c] of view v = B_k(dir_v) * dRGB_v[c] (backward.cu:22-141), so instead of all-reducing 48 floats per gaussian the
// ranks all-gather 3 and every rank evaluates sum_v B_k(dir_v(p)) dRGB_v[p][c] itself.  gathered: V rows of `row_stride` floats,
// row v = [dRGB_v (P x 3) | campos_v (3)].
// Stores: a lane's 3 M floats are one 12 M-byte row, i.e. written lane by lane the wave's 64 rows take 48 scalar store instructions of
// 64 different cache lines each (0.7 TB/s, 133 us for 300 k gaussians -- a third of the whole exchange at V = 8).  For M = 16 the rows go
// through a per-wave LDS tile (row stride 49 floats: lane-private rows without bank conflicts) and leave as 16-byte pieces of the wave's
// contiguous 12 KB run, as in preprocess_bwd.
#define EXP_L 48
#define EXP_STRIDE 49
template <bool TILE>
__global__ void __launch_bounds__(256) sh_grad_expand_kernel(int P, int M, int D, int V, const float* __restrict__ means3D,
                                                             const float* __restrict__ gathered, long long row_stride,
                                                             float* __restrict__ out)
{
    __shared__ float s_tile[TILE ? 4 : 1][TILE ? 64 * EXP_STRIDE : 1];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, row0 = idx - lane;
    const bool in_range = idx < P;
    if (!TILE && !in_range) return;
    const size_t i3 = 3 * (size_t)(in_range ? idx : 0);
    const float p[3] = {means3D[i3], means3D[i3 + 1], means3D[i3 + 2]};
    float acc[16][3];
#pragma unroll
    for (int k = 0; k < 16; k++) { acc[k][0] = 0.0f; acc[k][1] = 0.0f; acc[k][2] = 0.0f; }
    const int ncoef = (D + 1) * (D + 1);
    for (int v = 0; v < V; v++) {
        const float* row = gathered + (size_t)v * row_stride;
        const float* cam = row + 3 * (size_t)P;
        const float g[3] = {row[i3], row[i3 + 1], row[i3 + 2]};
        if (g[0] == 0.0f && g[1] == 0.0f && g[2] == 0.0f) continue;     // not visible in that view
        const float d[3] = {p[0] - cam[0], p[1] - cam[1], p[2] - cam[2]};
        const float len = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        float B[16];
        sh_basis16(d[0] / len, d[1] / len, d[2] / len, B);
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const float b = k < ncoef ? B[k] : 0.0f;
            acc[k][0] += b * g[0]; acc[k][1] += b * g[1]; acc[k][2] += b * g[2];
        }
    }
    if (TILE) {       // M == 16
        float* tile = s_tile[threadIdx.x >> 6];
#pragma unroll
        for (int k = 0; k < 16; k++)
#pragma unroll
            for (int c = 0; c < 3; c++) tile[lane * EXP_STRIDE + 3 * k + c] = acc[k][c];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (row0 < P) tile_store<EXP_L, EXP_STRIDE>(tile, out + (size_t)row0 * EXP_L, min(64, P - row0), lane);
        return;
    }
    float* o = out + (size_t)idx * M * 3;
    for (int k = 0; k < M; k++) {
        const bool in16 = k < 16;
        o[3 * k] = in16 ? acc[k & 15][0] : 0.0f; o[3 * k + 1] = in16 ? acc[k & 15][1] : 0.0f; o[3 * k + 2] = in16 ? acc[k & 15][2] : 0.0f;
    }
}


// The same factoring for the parameter set of render_surfel (BASELINE config 5: 111 floats per gaussian): both SH families are
// rank one per view -- the colour SH along the view direction (as above) and the indirect-radiance SH along the mirror direction
// make_frame(...).r of this file (d indirect_sh[p][k][c] = B_k(r_v(p)) g_v[p][c], clamp already folded into g) -- and both
// directions can be rebuilt from replicated parameters (xyz, rotation) and the rank's camera centre.  gathered row v =
// [dRGB_v (P x 3) | dIND_v (P x 3) | campos_v (3)]; outputs are the four split tensors of the model (dc [P,1,3], rest [P,15,3]).
// (the three parts of a row by pointer and stride of their own: a view-parallel step gathers the two factors at different times --
// mrgs_sh_grad_expand_surfel_rows -- or as one row, mrgs_sh_grad_expand_surfel)
__global__ void __launch_bounds__(256) sh_grad_expand_surfel_kernel(int P, int D, int V, const float* __restrict__ xyz,
                                                                    const float* __restrict__ rotation_raw, const float* __restrict__ rgb_rows,
                                                                    long long rgb_stride, const float* __restrict__ ind_rows, long long ind_stride,
                                                                    const float* __restrict__ cam_rows, long long cam_stride,
                                                                    float* __restrict__ g_dc, float* __restrict__ g_rest,
                                                                    float* __restrict__ g_ind_dc, float* __restrict__ g_ind_rest)
{
    __shared__ float s_tile[4][64 * REST_STRIDE];       // the wave's 64 rows of 45 "rest" floats, first one family, then the other
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, row0 = idx - lane;
    const bool in_range = idx < P;
    const size_t ic = (size_t)(in_range ? idx : 0);
    const float p[3] = {xyz[3 * ic], xyz[3 * ic + 1], xyz[3 * ic + 2]};
    const float4 q = reinterpret_cast<const float4*>(rotation_raw)[ic];
    float a[16][3], b[16][3];
#pragma unroll
    for (int k = 0; k < 16; k++) { a[k][0] = a[k][1] = a[k][2] = 0.0f; b[k][0] = b[k][1] = b[k][2] = 0.0f; }
    const int ncoef = (D + 1) * (D + 1);
    // (rgb_rows == nullptr / ind_rows == nullptr: that family is left out, its two output tensors untouched -- the step expands each
    // family when ITS all-gather has landed)
    const bool fam_a = rgb_rows != nullptr, fam_b = ind_rows != nullptr;
    for (int v = 0; v < V; v++) {
        const float* cam = cam_rows + (size_t)v * cam_stride;
        float g[3] = {0.0f, 0.0f, 0.0f}, h[3] = {0.0f, 0.0f, 0.0f};
        if (fam_a) {
            const float* row = rgb_rows + (size_t)v * rgb_stride;
            g[0] = row[3 * ic]; g[1] = row[3 * ic + 1]; g[2] = row[3 * ic + 2];
        }
        if (fam_b) {
            const float* ri = ind_rows + (size_t)v * ind_stride;
            h[0] = ri[3 * ic]; h[1] = ri[3 * ic + 1]; h[2] = ri[3 * ic + 2];
        }
        const bool has_g = (g[0] != 0.0f) | (g[1] != 0.0f) | (g[2] != 0.0f), has_h = (h[0] != 0.0f) | (h[1] != 0.0f) | (h[2] != 0.0f);
        if (!has_g && !has_h) continue;
        const Frame f = make_frame(p, q, cam);
        float B[16];
        if (has_g) {
            sh_basis16(f.v[0], f.v[1], f.v[2], B);
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const float w = k < ncoef ? B[k] : 0.0f;
                a[k][0] += w * g[0]; a[k][1] += w * g[1]; a[k][2] += w * g[2];
            }
        }
        if (has_h) {
            sh_basis16(f.r[0], f.r[1], f.r[2], B);          // eval_sh(3, ...) along the mirror direction (__init__.py:350-352)
#pragma unroll
            for (int k = 0; k < 16; k++) { b[k][0] += B[k] * h[0]; b[k][1] += B[k] * h[1]; b[k][2] += B[k] * h[2]; }
        }
    }
    if (in_range) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            if (fam_a) g_dc[3 * (size_t)idx + c] = a[0][c];
            if (fam_b) g_ind_dc[3 * (size_t)idx + c] = b[0][c];
        }
    }
    // the two [P,15,3] tensors: a lane's row is 180 bytes, written lane by lane the wave's 64 rows are 45 scalar stores of 64 cache lines
    // each; through the per-wave tile they leave as 16-byte pieces of one contiguous 11.5 KB run
    float* tile = s_tile[threadIdx.x >> 6];
    const int nrows = min(64, P - row0);
    if (fam_a) {
#pragma unroll
        for (int k = 1; k < 16; k++)
#pragma unroll
            for (int c = 0; c < 3; c++) tile[lane * REST_STRIDE + 3 * (k - 1) + c] = a[k][c];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (row0 < P) tile_store<REST_L>(tile, g_rest + (size_t)row0 * REST_L, nrows, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (fam_b) {
#pragma unroll
        for (int k = 1; k < 16; k++)
#pragma unroll
            for (int c = 0; c < 3; c++) tile[lane * REST_STRIDE + 3 * (k - 1) + c] = b[k][c];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (row0 < P) tile_store<REST_L>(tile, g_ind_rest + (size_t)row0 * REST_L, nrows, lane);
    }
}

}   // namespace

extern "C" {

int mrgs_sh_grad_expand_surfel(int32_t P, int32_t D, int32_t V, const float* xyz, const float* rotation_raw, const float* gathered,
                               int64_t row_stride, float* g_features_dc, float* g_features_rest, float* g_indirect_dc, float* g_indirect_rest,
                               void* stream)
{
    if (P < 0 || D < 0 || D > 3 || V < 1 || row_stride < 6 * (int64_t)P + 3) return MRGS_E_BAD_ARG;
    if (P == 0) return MRGS_OK;
    if (!xyz || !rotation_raw || !gathered || !g_features_dc || !g_features_rest || !g_indirect_dc || !g_indirect_rest) return MRGS_E_BAD_ARG;
    hipLaunchKernelGGL(sh_grad_expand_surfel_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, D, V, xyz, rotation_raw, gathered,
                       (long long)row_stride, gathered + 3 * (size_t)P, (long long)row_stride, gathered + 6 * (size_t)P, (long long)row_stride,
                       g_features_dc, g_features_rest, g_indirect_dc, g_indirect_rest);
    return hipGetLastError() == hipSuccess ? MRGS_OK : MRGS_E_HIP;
}

int mrgs_sh_grad_expand_surfel_rows(int32_t P, int32_t D, int32_t V, const float* xyz, const float* rotation_raw, const float* rgb_rows,
                                    int64_t rgb_stride, const float* ind_rows, int64_t ind_stride, const float* campos_rows, int64_t campos_stride,
                                    float* g_features_dc, float* g_features_rest, float* g_indirect_dc, float* g_indirect_rest, void* stream)
{
    if (P < 0 || D < 0 || D > 3 || V < 1 || (rgb_rows && rgb_stride < 3 * (int64_t)P) || (ind_rows && ind_stride < 3 * (int64_t)P) || campos_stride < 3)
        return MRGS_E_BAD_ARG;
    if (P == 0) return MRGS_OK;
    if (!xyz || !rotation_raw || (!rgb_rows && !ind_rows) || !campos_rows) return MRGS_E_BAD_ARG;
    if ((rgb_rows && (!g_features_dc || !g_features_rest)) || (ind_rows && (!g_indirect_dc || !g_indirect_rest))) return MRGS_E_BAD_ARG;
    hipLaunchKernelGGL(sh_grad_expand_surfel_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, D, V, xyz, rotation_raw, rgb_rows,
                       (long long)rgb_stride, ind_rows, (long long)ind_stride, campos_rows, (long long)campos_stride, g_features_dc, g_features_rest,
                       g_indirect_dc, g_indirect_rest);
    return hipGetLastError() == hipSuccess ? MRGS_OK : MRGS_E_HIP;
}

int mrgs_sh_grad_expand(int32_t P, int32_t M, int32_t D, int32_t V, const float* means3D, const float* gathered, int64_t row_stride,
                        float* dL_dsh, void* stream)
{
    if (P < 0 || M < 1 || D < 0 || D > 3 || V < 1 || row_stride < 3 * (int64_t)P + 3) return MRGS_E_BAD_ARG;
    if (P == 0) return MRGS_OK;
    if (!means3D || !gathered || !dL_dsh) return MRGS_E_BAD_ARG;
    // 16-byte pieces need the output 16-byte aligned (torch allocations are)
    if (M == 16 && ((uintptr_t)dL_dsh & 15u) == 0)
        hipLaunchKernelGGL(sh_grad_expand_kernel<true>, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, M, D, V, means3D, gathered,
                           (long long)row_stride, dL_dsh);
    else
        hipLaunchKernelGGL(sh_grad_expand_kernel<false>, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, M, D, V, means3D, gathered,
                           (long long)row_stride, dL_dsh);
    return hipGetLastError() == hipSuccess ? MRGS_OK : MRGS_E_HIP;
}


int mrgs_surfel_features_forward(const MrgsSurfelParams* p, float* opacity, float* scales, float* rotations, float* features, void* stream)
{
    if (!p || p->P < 0) return MRGS_E_BAD_ARG;
    if (p->P == 0) return MRGS_OK;
    if (!p->xyz || !p->scaling_raw || !p->rotation_raw || !p->opacity_raw || !p->refl_raw || !p->rough_raw || !p->ori_color_raw ||
        !p->indirect_dc || !p->indirect_rest || !p->campos || !opacity || !scales || !rotations || !features)
        return MRGS_E_BAD_ARG;
    hipLaunchKernelGGL(surfel_features_fwd_kernel, dim3((p->P + 255) / 256), dim3(256), 0, (hipStream_t)stream, *p, opacity, scales,
                       rotations, features);
    return hipGetLastError() == hipSuccess ? MRGS_OK : MRGS_E_HIP;
}

int mrgs_surfel_features_backward(const MrgsSurfelParams* p, const float* g_opacity, const float* g_scales, const float* g_rotations,
                                  const float* g_features, const MrgsSurfelGrads* grads, const float* g_xyz_upstream, void* stream)
{
    if (!p || p->P < 0 || !grads) return MRGS_E_BAD_ARG;
    if (p->P == 0) return MRGS_OK;
    if (!p->xyz || !p->scaling_raw || !p->rotation_raw || !p->opacity_raw || !p->refl_raw || !p->rough_raw || !p->ori_color_raw ||
        !p->indirect_dc || !p->indirect_rest || !p->campos)
        return MRGS_E_BAD_ARG;
    if (!grads->d_xyz || !grads->d_scaling || !grads->d_rotation || !grads->d_opacity || !grads->d_refl || !grads->d_rough ||
        !grads->d_ori_color || !grads->d_indirect_dc || !grads->d_indirect_rest)
        return MRGS_E_BAD_ARG;
    hipLaunchKernelGGL(surfel_features_bwd_kernel, dim3((p->P + 255) / 256), dim3(256), 0, (hipStream_t)stream, *p, g_opacity, g_scales,
                       g_rotations, g_features, *grads, g_xyz_upstream);
    return hipGetLastError() == hipSuccess ? MRGS_OK : MRGS_E_HIP;
}

}   // extern "C"
