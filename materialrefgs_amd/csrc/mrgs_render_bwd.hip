// mrgs_render_bwd.hip -- back-to-front replay and gradient accumulation on gfx950.
// Replaces BACKWARD::render / renderCUDA (backward.cu:145-468): same per-pixel arithmetic, same list order.
//
// Decomposition (see mrgs_render_fwd.hip for the forward twin): one wave64 per 8x8 pixel block, no barriers;
// the wave starts at the block's deepest contributor (wave-max of the forward's last_contributor) instead of
// the end of the tile list, consumes the list 64 entries at a time and walks only the entries whose
// conservative screen box can touch the block (__ballot sub-list, highest bit first).
//
// Gradient accumulation: the reference issues 16+S global fp32 atomicAdds per contributing (pixel, surfel)
// pair (backward.cu:350-465).  Here the K = 18+S per-lane terms of a surfel are reduced across the wave with
// a transposing butterfly -- v_permlane32_swap / v_permlane16_swap halve the register count while they halve
// the lane span, then four DPP row rotates finish inside 16-lane rows (2.5 K instructions instead of 6 K for
// K independent wave reductions) -- and K/4 atomic instructions, each with four lanes writing four different
// floats of the surfel's packed gradient row, replace K single-lane atomics.
#include "mrgs_blend_math.h"

#ifndef MRGS_BWD_STAGES
#define MRGS_BWD_STAGES 1   // see MRGS_FWD_STAGES in mrgs_render_fwd.hip
#endif


__device__ __forceinline__ void swap32_add(float& a, float b)   // a <- [a.lo + a.hi | b.lo + b.hi]
{
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r.x) + __uint_as_float(r.y);
}
__device__ __forceinline__ void swap16_add(float& a, float b)   // rows: [a.r0+a.r1, b.r0+b.r1, a.r2+a.r3, b.r2+b.r3]
{
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r.x) + __uint_as_float(r.y);
}
__device__ __forceinline__ float row_sum(float v)   // every lane of a 16-lane row gets the row total
{
    v = mrgs_dpp_add<0x128, 0xf>(v);   // row_ror:8
    v = mrgs_dpp_add<0x124, 0xf>(v);   // row_ror:4
    v = mrgs_dpp_add<0x122, 0xf>(v);   // row_ror:2
    v = mrgs_dpp_add<0x121, 0xf>(v);   // row_ror:1
    return v;
}

// Reduce K (multiple of 4) per-lane values over the wave and add them to dst[0..K).  After the two swap
// stages register i holds, in its rows 0..3, the 16-lane partial sums of values 4i, 4i+2, 4i+1, 4i+3.
template <int K>
__device__ __forceinline__ void wave_reduce_atomic_add(float (&v)[K], float* __restrict__ dst, int lane)
{
    static_assert(K % 4 == 0, "pad the value count to a multiple of 4");
#pragma unroll
    for (int i = 0; i < K / 2; i++) swap32_add(v[2 * i], v[2 * i + 1]);      // result in v[2i]
#pragma unroll
    for (int i = 0; i < K / 4; i++) swap16_add(v[4 * i], v[4 * i + 2]);      // result in v[4i]
    const int row = lane >> 4;
    const int sub = ((row & 1) << 1) | (row >> 1);                             // rows 0,1,2,3 -> values +0,+2,+1,+3
    // the four row rotates run as K/4 independent chains interleaved step by step (a DPP operand needs two wait states after
    // the VALU write that produced it: back-to-back steps of ONE chain would each cost an s_nop)
#pragma unroll
    for (int i = 0; i < K / 4; i++) v[4 * i] = mrgs_dpp_add<0x128, 0xf>(v[4 * i]);   // row_ror:8
#pragma unroll
    for (int i = 0; i < K / 4; i++) v[4 * i] = mrgs_dpp_add<0x124, 0xf>(v[4 * i]);   // row_ror:4
#pragma unroll
    for (int i = 0; i < K / 4; i++) v[4 * i] = mrgs_dpp_add<0x122, 0xf>(v[4 * i]);   // row_ror:2
#pragma unroll
    for (int i = 0; i < K / 4; i++) v[4 * i] = mrgs_dpp_add<0x121, 0xf>(v[4 * i]);   // row_ror:1
    if ((lane & 15) == 0) {
#pragma unroll
        for (int i = 0; i < K / 4; i++) atomicAdd(dst + 4 * i + sub, v[4 * i]);
    }
}

template <int S_MAX>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(S_MAX == 0 ? 4 : S_MAX <= 8 ? 3 : 2, 8))) render_bwd_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ tile_order, const uint32_t* __restrict__ point_list, int S, int W, int H, int tiles_x, int ntiles,
    const float4* __restrict__ rec, const float* __restrict__ features, const float* __restrict__ bg,
    const float* __restrict__ final_Ts, const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpixels,
    const float* __restrict__ dL_dpixels_f, const float* __restrict__ dL_dothers, float* __restrict__ grad_rec, int gstride)
{
    constexpr int SF = S_MAX > 0 ? S_MAX : 1;
    constexpr int K = (18 + S_MAX + 3) & ~3;
    __shared__ StageBuf<SF> stage[MRGS_BWD_STAGES];

    const int lane = threadIdx.x;
    const int b = blockIdx.x;
    // first half of the grid: one wave per quadrant; second half: the extra wave of each quadrant of a split tile (kept at the
    // end of the grid so that the waves that exit at once do not alternate with working ones in the dispatch order)
    const int nq = (int)(gridDim.x >> 1);
    const int half = b >= nq ? 1 : 0;
    const int bb = b - half * nq;
    const int xcd = bb & 7, seq = bb >> 3;
    const int tile = (int)tile_order[(seq >> 2) * 8 + xcd];   // longest lists first
    const int quad = seq & 3;
    if (tile >= ntiles) return;
    const uint2 range = ranges[tile];
    // Eight waves are launched per tile.  A tile with a short list is blended by four of them (one 8x8 quadrant each, the
    // other four exit here).  The launch lasts as long as its longest wave, so the quadrants of the DENSEST tiles are split
    // into two 8x4 halves: each half sees fewer surfels (the cull rectangle is half as tall), which shortens the critical
    // path at the price of idle lanes in a few waves.
    const bool split = (int)(range.y - range.x) > MRGS_SPLIT_THRESHOLD;
    if (!split && half) return;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int bx = tx * 2 + (quad & 1), by = ty * 2 + (quad >> 1);
    const int rows = split ? 4 : 8;
    const int pxi = bx * 8 + (lane & 7), pyi = by * 8 + half * 4 + (lane >> 3);
    const bool inside = pxi < W && pyi < H && (lane >> 3) < rows;
    const float px = (float)pxi, py = (float)pyi;
    const float blk_x0 = (float)(bx * 8), blk_y0 = (float)(by * 8 + half * 4), blk_h = (float)(rows - 1);   // rectangle of pixel centres
    const int HW = H * W;
    const int pix = inside ? W * pyi + pxi : 0;

    const int last_contributor = inside ? (int)n_contrib[pix] : 0;
    // deepest list position any pixel of this block blended: nothing behind it can receive a gradient
    int max_contrib = last_contributor;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) max_contrib = max(max_contrib, __shfl_xor(max_contrib, d, 64));
    if (max_contrib == 0) return;
    // longest waves first in line for issue slots (see mrgs_render_fwd.hip)
    if (max_contrib > 768) __builtin_amdgcn_s_setprio(3);
    else if (max_contrib > 384) __builtin_amdgcn_s_setprio(2);
    else if (max_contrib > 192) __builtin_amdgcn_s_setprio(1);
    const int median_contributor = inside ? (int)n_contrib[pix + HW] : 0;

    const float T_final = inside ? final_Ts[pix] : 0.f;
    float T = T_final;
    float accum_rec[3] = {0.f, 0.f, 0.f}, last_color[3] = {0.f, 0.f, 0.f}, dL_dpixel[3] = {0.f, 0.f, 0.f};
    float accum_rec_f[SF], last_feature[SF], dL_dpixel_f[SF];
#pragma unroll
    for (int i = 0; i < SF; i++) { accum_rec_f[i] = 0.f; last_feature[i] = 0.f; dL_dpixel_f[i] = 0.f; }
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
    const float final_A = 1.0f - T_final;
    float last_dL_dT = 0.f, last_alpha = 0.f;
    const float mscale = MRGS_FAR_N / (MRGS_FAR_N - MRGS_NEAR_N);
    const float dmd_scale = (MRGS_FAR_N * MRGS_NEAR_N) / (MRGS_FAR_N - MRGS_NEAR_N);
    const float bg_dot_dpixel = fmaf(bg[2], dL_dpixel[2], fmaf(bg[1], dL_dpixel[1], bg[0] * dL_dpixel[0]));

    // chunks of 64 list entries, from the one holding position max_contrib-1 down to chunk 0; lane l <-> position 64c+l.
    // Same staging pipeline as the forward (LDS-DMA double buffer, boxes two chunks and ids three chunks ahead), walked
    // towards the front of the list.
    const uint32_t* plist = point_list + range.x;
    const CullConic kNever = mrgs_cull_never();
    const int c_top = (max_contrib - 1) / MRGS_CHUNK;
    uint32_t id1 = 0, id2 = 0;
    CullConic box1 = kNever;
    uint64_t mask_cur;
    {
        uint32_t id0 = 0;
        CullConic box0 = kNever;
        if (c_top * MRGS_CHUNK + lane < max_contrib) {
            id0 = plist[c_top * MRGS_CHUNK + lane];
            box0 = mrgs_cull_load(rec, id0);
        }
        if (c_top >= 1) {
            id1 = plist[(c_top - 1) * MRGS_CHUNK + lane];
            box1 = mrgs_cull_load(rec, id1);
        }
        if (c_top >= 2) id2 = plist[(c_top - 2) * MRGS_CHUNK + lane];
        const bool cand0 = mrgs_block_may_touch(box0, blk_x0, blk_y0, 7.0f, blk_h);
        mask_cur = __ballot(cand0);
        mrgs_stage_async<S_MAX, SF>(stage[c_top % MRGS_BWD_STAGES], rec, features, S, id0, cand0);
        if (cand0) stage[c_top % MRGS_BWD_STAGES].id[lane] = id0;
    }

    for (int c = c_top; c >= 0; c--) {
        const int base = c * MRGS_CHUNK;
        mrgs_stage_wait();                    // chunk c has landed
        uint64_t mask_nxt = 0ull;
        auto stage_next = [&]() {
            const bool cand1 = mrgs_block_may_touch(box1, blk_x0, blk_y0, 7.0f, blk_h);
            mask_nxt = __ballot(cand1);
            mrgs_stage_async<S_MAX, SF>(stage[(c + 1) % MRGS_BWD_STAGES], rec, features, S, id1, cand1);
            if (cand1) stage[(c + 1) % MRGS_BWD_STAGES].id[lane] = id1;
            id1 = id2;
            box1 = kNever;
            if (c >= 2) box1 = mrgs_cull_load(rec, id1);
            if (c >= 3) id2 = plist[(c - 3) * MRGS_CHUNK + lane];
        };
        if (MRGS_BWD_STAGES == 2) stage_next();

        uint64_t mask = mask_cur;
        const StageBuf<SF>& sb = stage[c % MRGS_BWD_STAGES];
        if (mask != 0ull) {
        int j = 63 - __builtin_clzll(mask);   // back to front
        SurfelGeom sg;
        sg.g0 = sb.rec[0][j]; sg.g1 = sb.rec[1][j]; sg.g2 = sb.rec[2][j];
        while (true) {
            mask &= ~(1ull << j);
            const bool more = mask != 0ull;
            const int jn = more ? 63 - __builtin_clzll(mask) : j;
            SurfelGeom nxt;
            nxt.g0 = sb.rec[0][jn]; nxt.g1 = sb.rec[1][jn]; nxt.g2 = sb.rec[2][jn];
            const int contributor = base + j;           // 0-based list position; the forward's contributor is position+1
            Hit h;
            const bool hit = mrgs_intersect(sg, px, py, h);
            const bool active = hit && inside && contributor < last_contributor;
            if (__ballot(active) != 0ull) {

            float g[K];
#pragma unroll
            for (int i = 0; i < K; i++) g[i] = 0.f;
            if (active) {
                const float4 a0 = sb.rec[3][j], a1 = sb.rec[4][j];
                const float normal[3] = {a0.x, a0.y, a0.z};
                const float col[3] = {a0.w, a1.x, a1.y};
                const float alpha = h.alpha, G = h.G, c_d = h.depth;
                const float inv_1ma = mrgs_rcp(1.0f - alpha);
                T = T * inv_1ma;                                   // backward.cu:330
                const float w = alpha * T;
                const float one_m_la = 1.0f - last_alpha;
                float dL_dalpha = 0.0f;
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    accum_rec[ch] = fmaf(last_alpha, last_color[ch], one_m_la * accum_rec[ch]);
                    last_color[ch] = col[ch];
                    dL_dalpha = fmaf(col[ch] - accum_rec[ch], dL_dpixel[ch], dL_dalpha);
                    g[15 + ch] = w * dL_dpixel[ch];
                }
                if (S_MAX > 0) {
#pragma unroll
                    for (int ch = 0; ch < S_MAX; ch++)
                        if (ch < S) {
                            const float f = sb.feat[ch][j];
                            accum_rec_f[ch] = fmaf(last_alpha, last_feature[ch], one_m_la * accum_rec_f[ch]);
                            last_feature[ch] = f;
                            dL_dalpha = fmaf(f - accum_rec_f[ch], dL_dpixel_f[ch], dL_dalpha);
                            g[18 + ch] = w * dL_dpixel_f[ch];
                        }
                }
                const float inv_cd = mrgs_rcp(c_d);
                const float m_d = mscale * (1.0f - MRGS_NEAR_N * inv_cd);
                const float dmd_dd = dmd_scale * inv_cd * inv_cd;
                float dL_dz = (contributor == median_contributor - 1) ? dL_dmedian_depth : 0.0f;
                const float dL_dweight = fmaf(-2.0f * m_d, final_D, fmaf(m_d * m_d, final_A, final_D2)) * dL_dreg;
                dL_dalpha += dL_dweight - last_dL_dT;
                last_dL_dT = fmaf(dL_dweight, alpha, (1.0f - alpha) * last_dL_dT);
                const float dL_dmd = 2.0f * w * fmaf(m_d, final_A, -final_D) * dL_dreg;
                dL_dz = fmaf(dL_dmd, dmd_dd, dL_dz);
                accum_depth_rec = fmaf(last_alpha, last_depth, one_m_la * accum_depth_rec);
                last_depth = c_d;
                dL_dalpha = fmaf(c_d - accum_depth_rec, dL_ddepth, dL_dalpha);
                accum_alpha_rec = fmaf(one_m_la, accum_alpha_rec, last_alpha);
                dL_dalpha = fmaf(1.0f - accum_alpha_rec, dL_daccum, dL_dalpha);
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    accum_normal_rec[ch] = fmaf(last_alpha, last_normal[ch], one_m_la * accum_normal_rec[ch]);
                    last_normal[ch] = normal[ch];
                    dL_dalpha = fmaf(normal[ch] - accum_normal_rec[ch], dL_dnormal2D[ch], dL_dalpha);
                    g[12 + ch] = w * dL_dnormal2D[ch];
                }
                dL_dalpha *= T;
                last_alpha = alpha;
                dL_dalpha = fmaf(-T_final * inv_1ma, bg_dot_dpixel, dL_dalpha);
                const float dL_dG = sg.g2.w * dL_dalpha;
                dL_dz = fmaf(w, dL_ddepth, dL_dz);
                if (h.rho3d <= h.rho2d) {
                    const float Twx = sg.g1.z, Twy = sg.g1.w;
                    const float dGn = dL_dG * -G;
                    const float dL_dsx = fmaf(dGn, h.sx, dL_dz * Twx);
                    const float dL_dsy = fmaf(dGn, h.sy, dL_dz * Twy);
                    const float dpx = dL_dsx * h.inv_pz, dpy = dL_dsy * h.inv_pz;
                    const float dpz = -fmaf(dpx, h.sx, dpy * h.sy);
                    const float dkx = fmaf(h.ly, dpz, -(h.lz * dpy));   // cross(l, dL_dp)
                    const float dky = fmaf(h.lz, dpx, -(h.lx * dpz));
                    const float dkz = fmaf(h.lx, dpy, -(h.ly * dpx));
                    const float dlx = fmaf(dpy, h.kz, -(dpz * h.ky));   // cross(dL_dp, k)
                    const float dly = fmaf(dpz, h.kx, -(dpx * h.kz));
                    const float dlz = fmaf(dpx, h.ky, -(dpy * h.kx));
                    g[0] = -dkx; g[1] = -dky; g[2] = -dkz;
                    g[3] = -dlx; g[4] = -dly; g[5] = -dlz;
                    g[6] = fmaf(px, dkx, fmaf(py, dlx, dL_dz * h.sx));
                    g[7] = fmaf(px, dky, fmaf(py, dly, dL_dz * h.sy));
                    g[8] = fmaf(px, dkz, fmaf(py, dlz, dL_dz));
                } else {
                    const float dGf = -G * MRGS_FILTER_INV_SQUARE;
                    g[9] = dL_dG * (dGf * h.dx);
                    g[10] = dL_dG * (dGf * h.dy);
                    g[8] = dL_dz;
                }
                g[11] = G * dL_dalpha;
            }
            wave_reduce_atomic_add<K>(g, grad_rec + (size_t)sb.id[j] * gstride, lane);
            }
            if (!more) break;
            sg = nxt;
            j = jn;
        }
        }
        if (MRGS_BWD_STAGES == 1) stage_next();
        mask_cur = mask_nxt;
    }
    mrgs_stage_wait();   // do not retire the wave with LDS-DMA still in flight
}

void mrgs_launch_render_bwd(const MrgsRasterConfig& cfg, const MrgsRasterInputs& in, const MrgsGeomWs& g, const uint32_t* plist,
                            const MrgsImgWs& img, const float* dL_dpix, const float* dL_dpix_f, const float* dL_dothers,
                            float* grad_rec, hipStream_t stream)
{
    const int tiles_x = (cfg.W + MRGS_BLOCK_X - 1) / MRGS_BLOCK_X, tiles_y = (cfg.H + MRGS_BLOCK_Y - 1) / MRGS_BLOCK_Y;
    const int ntiles = tiles_x * tiles_y;
    const int nblocks = ((ntiles + 7) / 8) * 8 * 8;   // 8 waves per tile (4 quadrants x 2 halves), tiles dealt to the 8 XCDs
    const dim3 grid(nblocks), block(64);
#define LAUNCH(SM, GS)                                                                                                       \
    hipLaunchKernelGGL(render_bwd_kernel<SM>, grid, block, 0, stream, img.ranges, img.tile_order, plist, cfg.S, cfg.W, cfg.H, tiles_x, ntiles, \
                       g.rec, in.features, in.bg, img.final_T, img.n_contrib, dL_dpix, dL_dpix_f, dL_dothers, grad_rec, GS)
    // the packed gradient row is as wide as the padded value count of the kernel instance (MRGS_GRAD_STRIDE)
    const int gs = MRGS_GRAD_STRIDE(cfg.S);
    if (cfg.S == 0) LAUNCH(0, gs);
    else if (cfg.S <= 8) LAUNCH(8, gs);
    else if (cfg.S <= 12) LAUNCH(12, gs);
    else LAUNCH(24, gs);
#undef LAUNCH
}
