// mrgs_render_fwd.hip -- front-to-back surfel blend on gfx950.  Replaces FORWARD::render / renderCUDA
// (forward.cu:272-463): same per-pixel arithmetic and the same list order, different decomposition.
//
// MI355X-first design (the reference runs one 256-thread block per 16x16 tile with two barriers per batch):
//   * one wave64 per 8x8 pixel block, each wave a workgroup of its own -> no barriers at all, 4x more (and 4x
//     smaller) work items to balance over 256 CUs, and the 64 lanes of a wave see nearly the same surfels;
//   * the tile's depth-sorted list is consumed 64 entries at a time: lane l fetches entry l's packed record
//     (id + four dwordx4), tests its conservative screen bound against the wave's block, and a 64-bit
//     __ballot gives the sub-list that can touch this block -- the wave then walks only the set bits
//     (s_ff1) instead of all 64 entries.  Skipped entries could not have passed alpha >= 1/255 on any lane;
//   * the surviving records are broadcast from LDS (ds_read_b128, same address on all lanes); colour, normal
//     and feature channels are read only when some lane really blends the surfel;
//   * blockIdx -> (tile, quadrant) keeps the four quadrant-waves of a tile on one XCD (blocks are dealt
//     round-robin to the 8 XCDs) so that the tile's records are fetched into one L2 only.
#include "mrgs_blend_math.h"

#define FWD_CHUNK 64

template <int S_MAX>
__global__ void __launch_bounds__(64) render_fwd_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, int S, int W, int H, int tiles_x, int ntiles,
    const float4* __restrict__ rec, const float* __restrict__ features, const float* __restrict__ bg, float* __restrict__ final_T,
    uint32_t* __restrict__ n_contrib, float* __restrict__ out_color, float* __restrict__ out_feature, float* __restrict__ out_others)
{
    constexpr int SF = S_MAX > 0 ? S_MAX : 1;
    __shared__ float4 s_geo[3][FWD_CHUNK];
    __shared__ float4 s_app[2][FWD_CHUNK];
    __shared__ float s_feat[SF * FWD_CHUNK];

    const int lane = threadIdx.x;
    // XCD-aware mapping: b % 8 selects the XCD; within an XCD consecutive blocks are the 4 quadrants of one tile
    const int b = blockIdx.x;
    const int xcd = b & 7, seq = b >> 3;
    const int tile = (seq >> 2) * 8 + xcd;
    const int quad = seq & 3;
    if (tile >= ntiles) return;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int bx = tx * 2 + (quad & 1), by = ty * 2 + (quad >> 1);
    int pxi, pyi;
    mrgs_block_pixel(bx, by, lane, pxi, pyi);
    const bool inside = pxi < W && pyi < H;
    const float px = (float)pxi, py = (float)pyi;
    const float bcx = (float)(bx * 8) + 3.5f, bcy = (float)(by * 8) + 3.5f;
    const int HW = H * W;
    const int pix = W * pyi + pxi;

    const uint2 range = ranges[tile];
    const int total = (int)(range.y - range.x);

    bool done = !inside;
    float T = 1.0f;
    float C0 = 0.f, C1 = 0.f, C2 = 0.f, N0 = 0.f, N1 = 0.f, N2 = 0.f;
    float F[SF];
#pragma unroll
    for (int i = 0; i < SF; i++) F[i] = 0.f;
    float Dp = 0.f, M1 = 0.f, M2 = 0.f, distortion = 0.f, median_depth = 0.f;
    uint32_t last_contributor = 0, median_contributor = 0;
    const float mscale = MRGS_FAR_N / (MRGS_FAR_N - MRGS_NEAR_N);

    // three-stage software pipeline over 64-entry chunks: ids are fetched two chunks ahead, packed geometry one
    // chunk ahead, so that the dependent gather (id -> record) never stalls the blend loop
    const uint32_t* plist = point_list + range.x;
    const float4 kNever = make_float4(0.f, 0.f, -1e30f, -1e30f);
    uint32_t id_cur = 0, id_nxt = 0;
    float4 q0, q1, q2, q5 = kNever;
    q0 = q1 = q2 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < total) {
        id_cur = plist[lane];
        const float4* src = rec + (size_t)id_cur * MRGS_REC_F4;
        q0 = src[0]; q1 = src[1]; q2 = src[2]; q5 = src[5];
    }
    if (FWD_CHUNK + lane < total) id_nxt = plist[FWD_CHUNK + lane];

    for (int base = 0; base < total; base += FWD_CHUNK) {
        if (__ballot(!done) == 0ull) break;   // every pixel of the block has terminated (forward.cu:342-344, per wave)
        const uint32_t cur_id = id_cur;
        const float4 c0 = q0, c1 = q1, c2 = q2;
        const bool cand = mrgs_block_may_touch(q5, bcx, bcy);
        uint64_t mask = __ballot(cand);
        // appearance of the candidates (needed first), then the prefetches for the following chunks
        float4 a3 = make_float4(0.f, 0.f, 0.f, 0.f), a4 = a3;
        float fch[SF];
        if (cand) {
            const float4* src = rec + (size_t)cur_id * MRGS_REC_F4;
            a3 = src[3];
            a4 = src[4];
            if (S_MAX > 0) {
                const float* fsrc = features + (size_t)cur_id * S;
#pragma unroll
                for (int ch = 0; ch < S_MAX; ch++)
                    if (ch < S) fch[ch] = fsrc[ch];
            }
        }
        id_cur = id_nxt;
        q5 = kNever;
        if (base + FWD_CHUNK + lane < total) {
            const float4* src = rec + (size_t)id_cur * MRGS_REC_F4;
            q0 = src[0]; q1 = src[1]; q2 = src[2]; q5 = src[5];
        }
        if (base + 2 * FWD_CHUNK + lane < total) id_nxt = plist[base + 2 * FWD_CHUNK + lane];
        if (mask == 0ull) continue;
        if (cand) {
            s_geo[0][lane] = c0; s_geo[1][lane] = c1; s_geo[2][lane] = c2;
            s_app[0][lane] = a3;
            s_app[1][lane] = a4;
            if (S_MAX > 0) {
#pragma unroll
                for (int ch = 0; ch < S_MAX; ch++)
                    if (ch < S) s_feat[ch * FWD_CHUNK + lane] = fch[ch];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // single-wave workgroup: LDS ops are ordered in issue order
        while (mask) {
            const int j = __builtin_ctzll(mask);
            mask &= mask - 1;
            if (done) continue;
            SurfelGeom sg;
            sg.g0 = s_geo[0][j]; sg.g1 = s_geo[1][j]; sg.g2 = s_geo[2][j];
            Hit h;
            if (!mrgs_intersect(sg, px, py, h)) continue;
            const float test_T = T * (1.0f - h.alpha);
            if (test_T < MRGS_T_MIN) { done = true; continue; }
            const float w = h.alpha * T;
            const float A = 1.0f - T;
            const float m = mscale * (1.0f - MRGS_NEAR_N * mrgs_rcp(h.depth));
            const float mm = m * m;
            // distortion += (m*m*A + M2 - 2*m*M1) * w   (forward.cu:412), fused as written here and in the oracle
            distortion = fmaf(fmaf(-2.0f * m, M1, fmaf(mm, A, M2)), w, distortion);
            Dp = fmaf(h.depth, w, Dp);
            M1 = fmaf(m, w, M1);
            M2 = fmaf(mm, w, M2);
            const uint32_t contributor = (uint32_t)(base + j + 1);
            if (T > 0.5f) { median_depth = h.depth; median_contributor = contributor; }
            const float4 a0 = s_app[0][j], a1 = s_app[1][j];
            N0 = fmaf(a0.x, w, N0); N1 = fmaf(a0.y, w, N1); N2 = fmaf(a0.z, w, N2);
            C0 = fmaf(a0.w, w, C0); C1 = fmaf(a1.x, w, C1); C2 = fmaf(a1.y, w, C2);
            if (S_MAX > 0) {
#pragma unroll
                for (int ch = 0; ch < S_MAX; ch++)
                    if (ch < S) F[ch] = fmaf(s_feat[ch * FWD_CHUNK + j], w, F[ch]);
            }
            T = test_T;
            last_contributor = contributor;
        }
    }

    if (inside) {
        final_T[pix] = T;
        final_T[pix + HW] = M1;
        final_T[pix + 2 * HW] = M2;
        n_contrib[pix] = last_contributor;
        n_contrib[pix + HW] = median_contributor;
        out_color[pix] = fmaf(T, bg[0], C0);
        out_color[pix + HW] = fmaf(T, bg[1], C1);
        out_color[pix + 2 * HW] = fmaf(T, bg[2], C2);
        if (S_MAX > 0) {
#pragma unroll
            for (int ch = 0; ch < S_MAX; ch++)
                if (ch < S) out_feature[(size_t)ch * HW + pix] = F[ch];
        }
        out_others[pix + 0 * HW] = Dp;
        out_others[pix + 1 * HW] = 1.0f - T;
        out_others[pix + 2 * HW] = N0;
        out_others[pix + 3 * HW] = N1;
        out_others[pix + 4 * HW] = N2;
        out_others[pix + 5 * HW] = median_depth;
        out_others[pix + 6 * HW] = distortion;
    }
}

void mrgs_launch_render_fwd(const MrgsRasterConfig& cfg, const MrgsRasterInputs& in, const MrgsGeomWs& g, const uint32_t* plist,
                            const MrgsImgWs& img, float* out_color, float* out_feature, float* out_others, hipStream_t stream)
{
    const int tiles_x = (cfg.W + MRGS_BLOCK_X - 1) / MRGS_BLOCK_X, tiles_y = (cfg.H + MRGS_BLOCK_Y - 1) / MRGS_BLOCK_Y;
    const int ntiles = tiles_x * tiles_y;
    const int nblocks = ((ntiles + 7) / 8) * 8 * 4;   // 4 quadrant-waves per tile, tiles dealt to the 8 XCDs
    const dim3 grid(nblocks), block(64);
#define LAUNCH(SM)                                                                                                           \
    hipLaunchKernelGGL(render_fwd_kernel<SM>, grid, block, 0, stream, img.ranges, plist, cfg.S, cfg.W, cfg.H, tiles_x, ntiles, \
                       g.rec, in.features, in.bg, img.final_T, img.n_contrib, out_color, out_feature, out_others)
    if (cfg.S == 0) LAUNCH(0);
    else if (cfg.S <= 8) LAUNCH(8);
    else if (cfg.S <= 12) LAUNCH(12);
    else LAUNCH(24);
#undef LAUNCH
}
