// mrgs_render_bwd.hip -- per-tile back-to-front replay and gradient accumulation on gfx950.
// Replaces BACKWARD::render / renderCUDA (backward.cu:145-468).
//
// Same tiling as the forward: one workgroup per 16x16 tile, wave w owns an 8x8 quadrant.  The reference
// issues 16+S global fp32 atomicAdds per contributing (pixel, surfel) pair (backward.cu:350-465).  Here the
// 64 per-pixel terms of a wave are first summed inside the wave with DPP row operations (no LDS traffic),
// and only the wave total is added -- one atomic per value per (wave, surfel) -- into a packed
// per-gaussian gradient row (MRGS_GRAD_STRIDE floats) so that all atomics of a pair hit the same lines.
#include "mrgs_internal.h"

#define BWD_THREADS 256
#define BWD_BATCH 256

// wave64 sum via DPP: quad_perm swaps, row rotates, then row_bcast 15/31; lane 63 holds the total.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v)
{
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
    return v + __int_as_float(moved);
}
__device__ __forceinline__ float wave_sum_to_lane63(float v)
{
    v = dpp_add<0xb1, 0xf>(v);    // quad_perm:[1,0,3,2]
    v = dpp_add<0x4e, 0xf>(v);    // quad_perm:[2,3,0,1]
    v = dpp_add<0x124, 0xf>(v);   // row_ror:4
    v = dpp_add<0x128, 0xf>(v);   // row_ror:8
    v = dpp_add<0x142, 0xa>(v);   // row_bcast:15 into rows 1,3
    v = dpp_add<0x143, 0xc>(v);   // row_bcast:31 into rows 2,3
    return v;
}

template <int S_MAX>
__global__ void __launch_bounds__(BWD_THREADS) render_bwd_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, int S, int W, int H, int tiles_x,
    const float4* __restrict__ rec, const float* __restrict__ features, const float* __restrict__ bg,
    const float* __restrict__ final_Ts, const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpixels,
    const float* __restrict__ dL_dpixels_f, const float* __restrict__ dL_dothers, float* __restrict__ grad_rec, int gstride)
{
    __shared__ float4 s_rec[MRGS_REC_F4][BWD_BATCH];
    __shared__ float s_feat[(S_MAX > 0 ? S_MAX : 1) * BWD_BATCH];
    __shared__ uint32_t s_id[BWD_BATCH];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int pxi = tx * MRGS_BLOCK_X + (wave & 1) * 8 + (lane & 7);
    const int pyi = ty * MRGS_BLOCK_Y + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = pxi < W && pyi < H;
    const float px = (float)pxi, py = (float)pyi;
    const int HW = H * W;
    const int pix = inside ? W * pyi + pxi : 0;

    const uint2 range = ranges[tile];
    const int total = (int)(range.y - range.x);
    const int rounds = (total + BWD_BATCH - 1) / BWD_BATCH;

    const float T_final = inside ? final_Ts[pix] : 0.f;
    float T = T_final;
    const int last_contributor = inside ? (int)n_contrib[pix] : 0;
    const int median_contributor = inside ? (int)n_contrib[pix + HW] : 0;

    float accum_rec[3] = {0.f, 0.f, 0.f}, last_color[3] = {0.f, 0.f, 0.f}, dL_dpixel[3] = {0.f, 0.f, 0.f};
    float accum_rec_f[S_MAX > 0 ? S_MAX : 1], last_feature[S_MAX > 0 ? S_MAX : 1], dL_dpixel_f[S_MAX > 0 ? S_MAX : 1];
#pragma unroll
    for (int i = 0; i < (S_MAX > 0 ? S_MAX : 1); i++) { accum_rec_f[i] = 0.f; last_feature[i] = 0.f; dL_dpixel_f[i] = 0.f; }
    float dL_dreg = 0.f, dL_ddepth = 0.f, dL_daccum = 0.f, dL_dnormal2D[3] = {0.f, 0.f, 0.f}, dL_dmedian_depth = 0.f;
    if (inside) {
        dL_ddepth = dL_dothers[0 * HW + pix];
        dL_daccum = dL_dothers[1 * HW + pix];
        dL_dnormal2D[0] = dL_dothers[2 * HW + pix];
        dL_dnormal2D[1] = dL_dothers[3 * HW + pix];
        dL_dnormal2D[2] = dL_dothers[4 * HW + pix];
        dL_dmedian_depth = dL_dothers[5 * HW + pix];
        dL_dreg = dL_dothers[6 * HW + pix];
#pragma unroll
        for (int i = 0; i < 3; i++) dL_dpixel[i] = dL_dpixels[i * HW + pix];
        if (S_MAX > 0) {
#pragma unroll
            for (int i = 0; i < S_MAX; i++)
                if (i < S) dL_dpixel_f[i] = dL_dpixels_f[(size_t)i * HW + pix];
        }
    }
    float last_depth = 0.f, last_normal[3] = {0.f, 0.f, 0.f}, accum_depth_rec = 0.f, accum_alpha_rec = 0.f,
          accum_normal_rec[3] = {0.f, 0.f, 0.f};
    const float final_D = inside ? final_Ts[pix + HW] : 0.f;
    const float final_D2 = inside ? final_Ts[pix + 2 * HW] : 0.f;
    const float final_A = 1 - T_final;
    float last_dL_dT = 0.f, last_alpha = 0.f;
    const float mscale = MRGS_FAR_N / (MRGS_FAR_N - MRGS_NEAR_N);
    const float bg_dot_dpixel = bg[0] * dL_dpixel[0] + bg[1] * dL_dpixel[1] + bg[2] * dL_dpixel[2];

    for (int r = 0; r < rounds; r++) {
        __syncthreads();
        const int base = r * BWD_BATCH;
        if (base + tid < total) {
            const uint32_t g = point_list[range.y - 1 - (uint32_t)(base + tid)];   // back to front
            s_id[tid] = g;
            const float4* src = rec + (size_t)g * MRGS_REC_F4;
#pragma unroll
            for (int k = 0; k < MRGS_REC_F4; k++) s_rec[k][tid] = src[k];
            if (S_MAX > 0) {
                const float* fsrc = features + (size_t)g * S;
                for (int ch = 0; ch < S; ch++) s_feat[tid * S_MAX + ch] = fsrc[ch];
            }
        }
        __syncthreads();
        const int count = min(BWD_BATCH, total - base);
        for (int j = 0; j < count; j++) {
            // list position (0-based, front to back) of this surfel; the forward's contributor is position+1
            const int contributor = total - 1 - (base + j);
            bool active = inside && contributor < last_contributor;
            float G = 0.f, alpha = 0.f, c_d = 0.f, sx = 0.f, sy = 0.f, ppz = 1.f, rho3d = 0.f, rho2d = 0.f, dx = 0.f, dy = 0.f;
            float kx = 0.f, ky = 0.f, kz = 0.f, lx = 0.f, ly = 0.f, lz = 0.f;
            const float4 r0 = s_rec[0][j], r1 = s_rec[1][j], r2 = s_rec[2][j];
            if (active) {
                kx = px * r1.z - r0.x; ky = px * r1.w - r0.y; kz = px * r2.x - r0.z;
                lx = py * r1.z - r0.w; ly = py * r1.w - r1.x; lz = py * r2.x - r1.y;
                const float ppx = ky * lz - kz * ly, ppy = kz * lx - kx * lz;
                ppz = kx * ly - ky * lx;
                if (ppz == 0.0f) active = false;
                else {
                    sx = ppx / ppz; sy = ppy / ppz;
                    rho3d = sx * sx + sy * sy;
                    dx = r2.y - px; dy = r2.z - py;
                    rho2d = MRGS_FILTER_INV_SQUARE * (dx * dx + dy * dy);
                    const float rho = fminf(rho3d, rho2d);
                    c_d = (rho3d <= rho2d) ? (sx * r1.z + sy * r1.w) + r2.x : r2.x;
                    const float power = -0.5f * rho;
                    if (c_d < MRGS_NEAR_N || power > 0.0f) active = false;
                    else {
                        G = MRGS_EXP(power);
                        alpha = fminf(0.99f, r2.w * G);
                        if (alpha < 1.0f / 255.0f) active = false;
                    }
                }
            }
            if (__ballot(active) == 0ull) continue;   // wave-uniform: nobody in this 8x8 quadrant is touched

            // per-lane gradient terms (zero for inactive lanes)
            float g_T[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            float g_m2x = 0.f, g_m2y = 0.f, g_op = 0.f, g_n[3] = {0.f, 0.f, 0.f}, g_c[3] = {0.f, 0.f, 0.f};
            float g_f[S_MAX > 0 ? S_MAX : 1];
#pragma unroll
            for (int i = 0; i < (S_MAX > 0 ? S_MAX : 1); i++) g_f[i] = 0.f;

            if (active) {
                const float4 r3 = s_rec[3][j], r4 = s_rec[4][j];
                const float normal[3] = {r3.x, r3.y, r3.z};
                const float col[3] = {r3.w, r4.x, r4.y};
                T = T / (1.f - alpha);
                const float dchannel_dcolor = alpha * T;
                float dL_dalpha = 0.0f;
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
                    last_color[ch] = col[ch];
                    dL_dalpha += (col[ch] - accum_rec[ch]) * dL_dpixel[ch];
                    g_c[ch] = dchannel_dcolor * dL_dpixel[ch];
                }
                if (S_MAX > 0) {
#pragma unroll
                    for (int ch = 0; ch < S_MAX; ch++)
                        if (ch < S) {
                            const float f = s_feat[j * S_MAX + ch];
                            accum_rec_f[ch] = last_alpha * last_feature[ch] + (1.f - last_alpha) * accum_rec_f[ch];
                            last_feature[ch] = f;
                            dL_dalpha += (f - accum_rec_f[ch]) * dL_dpixel_f[ch];
                            g_f[ch] = dchannel_dcolor * dL_dpixel_f[ch];
                        }
                }
                float dL_dz = 0.0f, dL_dweight = 0.f;
                const float m_d = mscale * (1 - MRGS_NEAR_N / c_d);
                const float dmd_dd = (MRGS_FAR_N * MRGS_NEAR_N) / ((MRGS_FAR_N - MRGS_NEAR_N) * c_d * c_d);
                if (contributor == median_contributor - 1) dL_dz += dL_dmedian_depth;
                dL_dweight += (final_D2 + m_d * m_d * final_A - 2 * m_d * final_D) * dL_dreg;
                dL_dalpha += dL_dweight - last_dL_dT;
                last_dL_dT = dL_dweight * alpha + (1 - alpha) * last_dL_dT;
                const float dL_dmd = 2.0f * (T * alpha) * (m_d * final_A - final_D) * dL_dreg;
                dL_dz += dL_dmd * dmd_dd;
                accum_depth_rec = last_alpha * last_depth + (1.f - last_alpha) * accum_depth_rec;
                last_depth = c_d;
                dL_dalpha += (c_d - accum_depth_rec) * dL_ddepth;
                accum_alpha_rec = last_alpha * 1.0f + (1.f - last_alpha) * accum_alpha_rec;
                dL_dalpha += (1 - accum_alpha_rec) * dL_daccum;
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    accum_normal_rec[ch] = last_alpha * last_normal[ch] + (1.f - last_alpha) * accum_normal_rec[ch];
                    last_normal[ch] = normal[ch];
                    dL_dalpha += (normal[ch] - accum_normal_rec[ch]) * dL_dnormal2D[ch];
                    g_n[ch] = alpha * T * dL_dnormal2D[ch];
                }
                dL_dalpha *= T;
                last_alpha = alpha;
                dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;
                const float dL_dG = r2.w * dL_dalpha;
                dL_dz += alpha * T * dL_ddepth;
                if (rho3d <= rho2d) {
                    const float dL_dsx = dL_dG * -G * sx + dL_dz * r1.z;
                    const float dL_dsy = dL_dG * -G * sy + dL_dz * r1.w;
                    const float dsx_pz = dL_dsx / ppz, dsy_pz = dL_dsy / ppz;
                    const float dpx = dsx_pz, dpy = dsy_pz, dpz = -(dsx_pz * sx + dsy_pz * sy);
                    const float dkx = ly * dpz - lz * dpy, dky = lz * dpx - lx * dpz, dkz = lx * dpy - ly * dpx;   // cross(l, dL_dp)
                    const float dlx = dpy * kz - dpz * ky, dly = dpz * kx - dpx * kz, dlz = dpx * ky - dpy * kx;   // cross(dL_dp, k)
                    g_T[0] = -dkx; g_T[1] = -dky; g_T[2] = -dkz;
                    g_T[3] = -dlx; g_T[4] = -dly; g_T[5] = -dlz;
                    g_T[6] = px * dkx + py * dlx + dL_dz * sx;
                    g_T[7] = px * dky + py * dly + dL_dz * sy;
                    g_T[8] = px * dkz + py * dlz + dL_dz * 1.0f;
                } else {
                    const float dG_ddelx = -G * MRGS_FILTER_INV_SQUARE * dx;
                    const float dG_ddely = -G * MRGS_FILTER_INV_SQUARE * dy;
                    g_m2x = dL_dG * dG_ddelx;
                    g_m2y = dL_dG * dG_ddely;
                    g_T[8] = dL_dz;
                }
                g_op = G * dL_dalpha;
            }

            // wave reduction, then one atomic per value from lane 63
            float* dst = grad_rec + (size_t)s_id[j] * gstride;
#pragma unroll
            for (int i = 0; i < 9; i++) g_T[i] = wave_sum_to_lane63(g_T[i]);
            g_m2x = wave_sum_to_lane63(g_m2x);
            g_m2y = wave_sum_to_lane63(g_m2y);
            g_op = wave_sum_to_lane63(g_op);
#pragma unroll
            for (int i = 0; i < 3; i++) { g_n[i] = wave_sum_to_lane63(g_n[i]); g_c[i] = wave_sum_to_lane63(g_c[i]); }
            if (S_MAX > 0) {
#pragma unroll
                for (int i = 0; i < S_MAX; i++)
                    if (i < S) g_f[i] = wave_sum_to_lane63(g_f[i]);
            }
            if (lane == 63) {
#pragma unroll
                for (int i = 0; i < 9; i++) atomicAdd(dst + i, g_T[i]);
                atomicAdd(dst + 9, g_m2x);
                atomicAdd(dst + 10, g_m2y);
                atomicAdd(dst + 11, g_op);
#pragma unroll
                for (int i = 0; i < 3; i++) { atomicAdd(dst + 12 + i, g_n[i]); atomicAdd(dst + 15 + i, g_c[i]); }
                if (S_MAX > 0) {
#pragma unroll
                    for (int i = 0; i < S_MAX; i++)
                        if (i < S) atomicAdd(dst + 18 + i, g_f[i]);
                }
            }
        }
    }
}

void mrgs_launch_render_bwd(const MrgsRasterConfig& cfg, const MrgsRasterInputs& in, const MrgsGeomWs& g, const uint32_t* plist,
                            const MrgsImgWs& img, const float* dL_dpix, const float* dL_dpix_f, const float* dL_dothers,
                            float* grad_rec, hipStream_t stream)
{
    const int tiles_x = (cfg.W + MRGS_BLOCK_X - 1) / MRGS_BLOCK_X, tiles_y = (cfg.H + MRGS_BLOCK_Y - 1) / MRGS_BLOCK_Y;
    const dim3 grid(tiles_x * tiles_y), block(BWD_THREADS);
    const int gstride = MRGS_GRAD_STRIDE(cfg.S);
#define LAUNCH(SM)                                                                                                              \
    hipLaunchKernelGGL(render_bwd_kernel<SM>, grid, block, 0, stream, img.ranges, plist, cfg.S, cfg.W, cfg.H, tiles_x, g.rec,    \
                       in.features, in.bg, img.final_T, img.n_contrib, dL_dpix, dL_dpix_f, dL_dothers, grad_rec, gstride)
    if (cfg.S == 0) LAUNCH(0);
    else if (cfg.S <= 8) LAUNCH(8);
    else if (cfg.S <= 12) LAUNCH(12);
    else LAUNCH(24);
#undef LAUNCH
}
