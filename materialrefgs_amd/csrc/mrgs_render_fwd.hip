// mrgs_render_fwd.hip -- per-tile front-to-back surfel blend on gfx950.
// Replaces FORWARD::render / renderCUDA (forward.cu:272-463).
//
// One workgroup (256 threads = 4 wave64) per 16x16 tile; wave w owns the 8x8 pixel quadrant
// (w&1, w>>1) so that the 64 lanes of a wave see nearly the same set of contributing surfels (coherent
// skip / early-out decisions, which are wave-level branches on CDNA).  The tile's depth-sorted surfel list is
// streamed through LDS in batches of 256: every thread gathers ONE packed 80-byte record (five dwordx4
// loads) plus the surfel's S feature channels, so the inner loop touches LDS only -- the reference re-reads
// colours and features from global memory per pixel per surfel (forward.cu:427,430).
#include "mrgs_internal.h"

#define FWD_THREADS 256
#define FWD_BATCH 256

template <int S_MAX>
__global__ void __launch_bounds__(FWD_THREADS) render_fwd_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, int S, int W, int H, int tiles_x,
    const float4* __restrict__ rec, const float* __restrict__ features, const float* __restrict__ bg, float* __restrict__ final_T,
    uint32_t* __restrict__ n_contrib, float* __restrict__ out_color, float* __restrict__ out_feature, float* __restrict__ out_others)
{
    __shared__ float4 s_rec[MRGS_REC_F4][FWD_BATCH];
    __shared__ float s_feat[(S_MAX > 0 ? S_MAX : 1) * FWD_BATCH];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int pxi = tx * MRGS_BLOCK_X + (wave & 1) * 8 + (lane & 7);
    const int pyi = ty * MRGS_BLOCK_Y + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = pxi < W && pyi < H;
    const float px = (float)pxi, py = (float)pyi;
    const int HW = H * W;
    const int pix = W * pyi + pxi;

    const uint2 range = ranges[tile];
    const int total = (int)(range.y - range.x);
    const int rounds = (total + FWD_BATCH - 1) / FWD_BATCH;

    bool done = !inside;
    float T = 1.0f;
    float C[3] = {0.f, 0.f, 0.f};
    float F[S_MAX > 0 ? S_MAX : 1];
#pragma unroll
    for (int i = 0; i < (S_MAX > 0 ? S_MAX : 1); i++) F[i] = 0.f;
    float N[3] = {0.f, 0.f, 0.f};
    float Dp = 0.f, M1 = 0.f, M2 = 0.f, distortion = 0.f, median_depth = 0.f;
    uint32_t last_contributor = 0, median_contributor = 0;
    const float mscale = MRGS_FAR_N / (MRGS_FAR_N - MRGS_NEAR_N);

    for (int r = 0; r < rounds; r++) {
        // whole tile finished? (forward.cu:342-344)
        if (__syncthreads_and(done)) break;
        const int base = r * FWD_BATCH;
        if (base + tid < total) {
            const uint32_t g = point_list[range.x + base + tid];
            const float4* src = rec + (size_t)g * MRGS_REC_F4;
#pragma unroll
            for (int k = 0; k < MRGS_REC_F4; k++) s_rec[k][tid] = src[k];
            if (S_MAX > 0) {
                const float* fsrc = features + (size_t)g * S;
                for (int ch = 0; ch < S; ch++) s_feat[tid * S_MAX + ch] = fsrc[ch];
            }
        }
        __syncthreads();
        const int count = min(FWD_BATCH, total - base);
        if (!done) {
            for (int j = 0; j < count; j++) {
                const float4 r0 = s_rec[0][j], r1 = s_rec[1][j], r2 = s_rec[2][j];
                // Tu = r0.xyz, Tv = (r0.w, r1.x, r1.y), Tw = (r1.z, r1.w, r2.x), xy = r2.yz, opacity = r2.w
                const float kx = px * r1.z - r0.x, ky = px * r1.w - r0.y, kz = px * r2.x - r0.z;
                const float lx = py * r1.z - r0.w, ly = py * r1.w - r1.x, lz = py * r2.x - r1.y;
                const float ppx = ky * lz - kz * ly, ppy = kz * lx - kx * lz, ppz = kx * ly - ky * lx;
                if (ppz == 0.0f) continue;
                const float sx = ppx / ppz, sy = ppy / ppz;
                const float rho3d = sx * sx + sy * sy;
                const float dx = r2.y - px, dy = r2.z - py;
                const float rho2d = MRGS_FILTER_INV_SQUARE * (dx * dx + dy * dy);
                const float rho = fminf(rho3d, rho2d);
                const float depth = (rho3d <= rho2d) ? (sx * r1.z + sy * r1.w) + r2.x : r2.x;
                if (depth < MRGS_NEAR_N) continue;
                const float power = -0.5f * rho;
                if (power > 0.0f) continue;
                const float alpha = fminf(0.99f, r2.w * MRGS_EXP(power));
                if (alpha < 1.0f / 255.0f) continue;
                const float test_T = T * (1 - alpha);
                if (test_T < 0.0001f) { done = true; break; }
                const float w = alpha * T;
                const float A = 1 - T;
                const float m = mscale * (1 - MRGS_NEAR_N / depth);
                distortion += (m * m * A + M2 - 2 * m * M1) * w;
                Dp += depth * w;
                M1 += m * w;
                M2 += m * m * w;
                const uint32_t contributor = (uint32_t)(base + j + 1);
                if (T > 0.5f) { median_depth = depth; median_contributor = contributor; }
                const float4 r3 = s_rec[3][j], r4 = s_rec[4][j];
                N[0] += r3.x * w; N[1] += r3.y * w; N[2] += r3.z * w;
                C[0] += r3.w * w; C[1] += r4.x * w; C[2] += r4.y * w;
                if (S_MAX > 0) {
#pragma unroll
                    for (int ch = 0; ch < S_MAX; ch++)
                        if (ch < S) F[ch] += s_feat[j * S_MAX + ch] * w;
                }
                T = test_T;
                last_contributor = contributor;
            }
        }
    }

    if (inside) {
        final_T[pix] = T;
        final_T[pix + HW] = M1;
        final_T[pix + 2 * HW] = M2;
        n_contrib[pix] = last_contributor;
        n_contrib[pix + HW] = median_contributor;
        out_color[pix] = C[0] + T * bg[0];
        out_color[pix + HW] = C[1] + T * bg[1];
        out_color[pix + 2 * HW] = C[2] + T * bg[2];
        if (S_MAX > 0) {
#pragma unroll
            for (int ch = 0; ch < S_MAX; ch++)
                if (ch < S) out_feature[(size_t)ch * HW + pix] = F[ch];
        }
        out_others[pix + 0 * HW] = Dp;
        out_others[pix + 1 * HW] = 1 - T;
        out_others[pix + 2 * HW] = N[0];
        out_others[pix + 3 * HW] = N[1];
        out_others[pix + 4 * HW] = N[2];
        out_others[pix + 5 * HW] = median_depth;
        out_others[pix + 6 * HW] = distortion;
    }
}

void mrgs_launch_render_fwd(const MrgsRasterConfig& cfg, const MrgsRasterInputs& in, const MrgsGeomWs& g, const uint32_t* plist,
                            const MrgsImgWs& img, float* out_color, float* out_feature, float* out_others, hipStream_t stream)
{
    const int tiles_x = (cfg.W + MRGS_BLOCK_X - 1) / MRGS_BLOCK_X, tiles_y = (cfg.H + MRGS_BLOCK_Y - 1) / MRGS_BLOCK_Y;
    const dim3 grid(tiles_x * tiles_y), block(FWD_THREADS);
#define LAUNCH(SM)                                                                                                              \
    hipLaunchKernelGGL(render_fwd_kernel<SM>, grid, block, 0, stream, img.ranges, plist, cfg.S, cfg.W, cfg.H, tiles_x, g.rec,    \
                       in.features, in.bg, img.final_T, img.n_contrib, out_color, out_feature, out_others)
    if (cfg.S == 0) LAUNCH(0);
    else if (cfg.S <= 8) LAUNCH(8);
    else if (cfg.S <= 12) LAUNCH(12);
    else LAUNCH(24);
#undef LAUNCH
}
