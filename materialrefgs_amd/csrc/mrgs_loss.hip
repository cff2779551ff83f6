// Photometric + geometric training loss of one view, fused.
//
// Replaces calculate_loss (utils/loss_utils.py:142-228) with l1_loss (:22-23), ssim/_ssim/create_window (:83-119): the reference
// runs five depthwise 11x11 conv2d launches, ~30 elementwise kernels and their autograd mirror per view; here
//   loss_fwd_kernel   one 32x32 pixel tile per workgroup and channel: both images staged with a 5-pixel halo in LDS, the five
//                     window moments by a separable pass (rows then columns, four outputs per thread from a sliding window), SSIM and its three partial derivatives with respect
//                     to (mu1, E[x^2], E[xy]) written as maps, |x-y|, (x-y)^2, the normal-consistency term and the distortion
//                     term reduced per workgroup (fixed order: results are run-to-run identical),
//   loss_finalize_kernel   one workgroup sums the per-workgroup partials in double and writes the scalar terms,
//   loss_bwd_kernel   same tiling: the three derivative maps are convolved with the (symmetric) window and combined with the
//                     pixel values into dL/dimage; L1 sign term, normal and distortion gradients are added in the same pass.
// Zero padding as F.conv2d(padding=5): pixels outside the image count as 0 in the moments and carry no derivative.
#include "mrgs_internal.h"

namespace {

constexpr int LT = 32;            // tile edge
constexpr int LP = 4;             // outputs per thread and pass (256 threads x 4 = 32 x 32)
constexpr int LR = 5;             // window radius (window_size 11, loss_utils.py:91)
constexpr int LH = LT + 2 * LR;   // tile + halo
constexpr int NPART = 8;          // floats per workgroup partial: ssim, l1, sq, normal, dist

struct LossArgs {
    int H, W, C;
    int normal_mode;              // 0 off, 1 weighted L1 (image_weight given), 2 cosine (no weight)
    float w[2 * LR + 1];
    float lambda_dssim, lambda_normal, lambda_dist;
};

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(256) void loss_fwd_kernel(LossArgs a, const float* __restrict__ img, const float* __restrict__ gt,
                                                        const float* __restrict__ rn, const float* __restrict__ sn,
                                                        const float* __restrict__ dist, const float* __restrict__ weight,
                                                        float* __restrict__ dmaps, float* __restrict__ partials)
{
    __shared__ float s1[LH][LH + 1], s2[LH][LH + 1];
    __shared__ float hz[5][LH][LT + 1];
    __shared__ float red[4][NPART];
    const int H = a.H, W = a.W, c = blockIdx.z;
    const int bx = blockIdx.x * LT, by = blockIdx.y * LT, tid = threadIdx.x;
    const size_t HW = (size_t)H * W;
    const float* I1 = img + c * HW;
    const float* I2 = gt + c * HW;
    {   // all loads of the tile in flight before the first LDS store (a rolled loop waits for every round trip in turn)
        constexpr int NL = (LH * LH + 255) / 256;
        float v1[NL], v2[NL];
#pragma unroll
        for (int n = 0; n < NL; ++n) {
            const int i = tid + n * 256, r = i / LH, cc = i - r * LH, y = by + r - LR, x = bx + cc - LR;
            const bool in = (i < LH * LH) & (x >= 0) & (x < W) & (y >= 0) & (y < H);
            v1[n] = in ? I1[(size_t)y * W + x] : 0.f;
            v2[n] = in ? I2[(size_t)y * W + x] : 0.f;
        }
#pragma unroll
        for (int n = 0; n < NL; ++n) {
            const int i = tid + n * 256, r = i / LH, cc = i - r * LH;
            if (i < LH * LH) { s1[r][cc] = v1[n]; s2[r][cc] = v2[n]; }
        }
    }
    __syncthreads();
    // rows: a thread produces LP consecutive outputs of one row from LP + 10 inputs (sliding window: 3.1 LDS reads per output
    // and quantity instead of 11)
    for (int it = tid; it < LH * (LT / LP); it += 256) {
        const int r = it / (LT / LP), c0 = (it - r * (LT / LP)) * LP;
        float p[LP + 2 * LR], q[LP + 2 * LR];
#pragma unroll
        for (int j = 0; j < LP + 2 * LR; ++j) { p[j] = s1[r][c0 + j]; q[j] = s2[r][c0 + j]; }
#pragma unroll
        for (int o = 0; o < LP; ++o) {
            float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
            for (int k = 0; k <= 2 * LR; ++k) {
                const float w = a.w[k], wp = w * p[o + k], wq = w * q[o + k];
                m1 += wp; m2 += wq; e11 = fmaf(wp, p[o + k], e11); e22 = fmaf(wq, q[o + k], e22); e12 = fmaf(wp, q[o + k], e12);
            }
            hz[0][r][c0 + o] = m1; hz[1][r][c0 + o] = m2; hz[2][r][c0 + o] = e11; hz[3][r][c0 + o] = e22; hz[4][r][c0 + o] = e12;
        }
    }
    __syncthreads();
    // columns: thread (tx, tg) produces the LP outputs (tx, LP tg .. LP tg + LP - 1)
    const int tx = tid & (LT - 1), y0 = (tid / LT) * LP;
    float mom[5][LP];
#pragma unroll
    for (int qn = 0; qn < 5; ++qn) {
        float v[LP + 2 * LR];
#pragma unroll
        for (int j = 0; j < LP + 2 * LR; ++j) v[j] = hz[qn][y0 + j][tx];
#pragma unroll
        for (int o = 0; o < LP; ++o) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k <= 2 * LR; ++k) acc = fmaf(a.w[k], v[o + k], acc);
            mom[qn][o] = acc;
        }
    }
    float part[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    const int x = bx + tx;
#pragma unroll
    for (int o = 0; o < LP; ++o) {
        const int y = by + y0 + o;
        if ((x >= W) | (y >= H)) continue;
        const size_t pix = (size_t)y * W + x;
        const float mu1 = mom[0][o], mu2 = mom[1][o], e11 = mom[2][o], e22 = mom[3][o], e12 = mom[4][o];
        const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;            // loss_utils.py:107-108
        const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
        const float sg1 = e11 - mu1_sq, sg2 = e22 - mu2_sq, sg12 = e12 - mu12;
        const float A1 = 2.f * mu12 + C1, A2 = 2.f * sg12 + C2, B1 = mu1_sq + mu2_sq + C1, B2 = sg1 + sg2 + C2;
        const float inv = 1.f / (B1 * B2);
        const float S = A1 * A2 * inv;                                  // :110
        const size_t CHW = HW * a.C;
        // total derivative with respect to mu1 (sigma1_sq = e11 - mu1^2, sigma12 = e12 - mu1 mu2), to e11 and to e12
        dmaps[c * HW + pix] = 2.f * mu2 * (A2 - A1) * inv + 2.f * mu1 * S * (1.f / B2 - 1.f / B1);
        dmaps[CHW + c * HW + pix] = -S / B2;
        dmaps[2 * CHW + c * HW + pix] = 2.f * A1 * inv;
        const float d = s1[y0 + o + LR][tx + LR] - s2[y0 + o + LR][tx + LR];
        part[0] += S; part[1] += fabsf(d); part[2] += d * d;
        if (c == 0) {
            if (a.normal_mode == 1) {                                    // (image_weight * |surf - rend|.sum(0)).mean()   :170
                float sm = 0.f;
#pragma unroll
                for (int k = 0; k < 3; ++k) sm += fabsf(sn[k * HW + pix] - rn[k * HW + pix]);
                part[3] += weight[pix] * sm;
            } else if (a.normal_mode == 2) {                             // (1 - (rend * surf).sum(0)).mean()   :172-173
                float sm = 0.f;
#pragma unroll
                for (int k = 0; k < 3; ++k) sm = fmaf(rn[k * HW + pix], sn[k * HW + pix], sm);
                part[3] += 1.f - sm;
            }
            if (a.lambda_dist > 0.f) part[4] += dist[pix];
        }
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) part[k] = wave_sum(part[k]);
    if ((tid & 63) == 0)
#pragma unroll
        for (int k = 0; k < 5; ++k) red[tid >> 6][k] = part[k];
    __syncthreads();
    if (tid < NPART) {
        const size_t b = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        partials[b * NPART + tid] = tid < 5 ? (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]) : 0.f;
    }
}

// out[0] loss, [1] Ll1, [2] ssim, [3] loss0, [4] normal term (mean, unscaled), [5] lambda_dist * mean(rend_dist), [6] psnr,
// [7..7+C) per-channel mse
__global__ __launch_bounds__(256) void loss_finalize_kernel(LossArgs a, const float* __restrict__ partials, int blocks_per_channel,
                                                             float* __restrict__ out, float* __restrict__ out_loss)
{
    __shared__ double red[4][8];
    const int tid = threadIdx.x, C = a.C < 4 ? a.C : 4;
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};                              // ssim, l1, normal, dist, sq[0..3]
    const int nb = blocks_per_channel * a.C;
    for (int b0 = tid; b0 < nb; b0 += 4 * 256) {
        // four partial rows (two 16-byte loads each) in flight per thread; the order of the additions is fixed
        float4 lo[4], hi[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int b = b0 + u * 256;
            const float4* p = reinterpret_cast<const float4*>(partials + (size_t)(b < nb ? b : 0) * NPART);
            lo[u] = p[0]; hi[u] = p[1];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int b = b0 + u * 256;
            if (b >= nb) continue;
            const int ch = b / blocks_per_channel;
            acc[0] += lo[u].x; acc[1] += lo[u].y; acc[2] += lo[u].w; acc[3] += hi[u].x;
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[4 + k] += ch == k ? (double)lo[u].z : 0.0;
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        double v = acc[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((tid & 63) == 0) red[tid >> 6][k] = v;
    }
    __syncthreads();
    if (tid == 0) {
        double s[8];
        for (int k = 0; k < 8; ++k) {
            double v = 0;
            for (int w = 0; w < 4; ++w) v += red[w][k];
            s[k] = v;
        }
        const double HW = (double)a.H * a.W, N = HW * a.C;
        const float Ll1 = (float)(s[1] / N), ssim = (float)(s[0] / N);
        const float loss0 = (1.0f - a.lambda_dssim) * Ll1 + a.lambda_dssim * (1.0f - ssim);        // :160
        const float nrm = a.normal_mode ? (float)(s[2] / HW) : 0.f;
        const float dst = a.lambda_dist > 0.f ? a.lambda_dist * (float)(s[3] / HW) : 0.f;         // :181
        float loss = loss0;
        if (a.normal_mode) loss += a.lambda_normal * nrm;
        loss += dst;
        float ps = 0.f;
        for (int k = 0; k < C; ++k) {                                       // utils/image_utils.py psnr: 20 log10(1 / sqrt(mse_c)), mean
            const float mse = (float)(s[4 + k] / HW);
            out[7 + k] = mse;
            ps += 20.f * log10f(1.f / sqrtf(mse));
        }
        for (int k = 7 + C; k < 16; ++k) out[k] = 0.f;
        if (out_loss) out_loss[0] = loss;
        out[0] = loss; out[1] = Ll1; out[2] = ssim; out[3] = loss0; out[4] = nrm; out[5] = dst; out[6] = ps / (float)C;
    }
}

__global__ __launch_bounds__(256) void loss_bwd_kernel(LossArgs a, const float* __restrict__ img, const float* __restrict__ gt,
                                                        const float* __restrict__ rn, const float* __restrict__ sn,
                                                        const float* __restrict__ weight, const float* __restrict__ dmaps,
                                                        const float* __restrict__ g_loss, float* __restrict__ g_img,
                                                        float* __restrict__ g_rn, float* __restrict__ g_sn, float* __restrict__ g_dist)
{
    __shared__ float s[3][LH][LH + 1];
    __shared__ float hz[3][LH][LT + 1];
    const int H = a.H, W = a.W, c = blockIdx.z;
    const int bx = blockIdx.x * LT, by = blockIdx.y * LT, tid = threadIdx.x;
    const size_t HW = (size_t)H * W, CHW = HW * a.C;
    const float gl = g_loss ? g_loss[0] : 1.f;
    const int tx = tid & (LT - 1), y0 = (tid / LT) * LP;
    float conv[3][LP];
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int o = 0; o < LP; ++o) conv[m][o] = 0.f;
    if (a.lambda_dssim != 0.f) {
        {
            constexpr int NL = (LH * LH + 255) / 256;
            float v[3][NL];
#pragma unroll
            for (int n = 0; n < NL; ++n) {
                const int i = tid + n * 256, r = i / LH, cc = i - r * LH, y = by + r - LR, x = bx + cc - LR;
                const bool in = (i < LH * LH) & (x >= 0) & (x < W) & (y >= 0) & (y < H);
                const size_t o = c * HW + (size_t)y * W + x;
                v[0][n] = in ? dmaps[o] : 0.f;
                v[1][n] = in ? dmaps[CHW + o] : 0.f;
                v[2][n] = in ? dmaps[2 * CHW + o] : 0.f;
            }
#pragma unroll
            for (int n = 0; n < NL; ++n) {
                const int i = tid + n * 256, r = i / LH, cc = i - r * LH;
                if (i < LH * LH) { s[0][r][cc] = v[0][n]; s[1][r][cc] = v[1][n]; s[2][r][cc] = v[2][n]; }
            }
        }
        __syncthreads();
        for (int it = tid; it < LH * (LT / LP); it += 256) {
            const int r = it / (LT / LP), c0 = (it - r * (LT / LP)) * LP;
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                float v[LP + 2 * LR];
#pragma unroll
                for (int j = 0; j < LP + 2 * LR; ++j) v[j] = s[m][r][c0 + j];
#pragma unroll
                for (int o = 0; o < LP; ++o) {
                    float acc = 0.f;
#pragma unroll
                    for (int k = 0; k <= 2 * LR; ++k) acc = fmaf(a.w[k], v[o + k], acc);
                    hz[m][r][c0 + o] = acc;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            float v[LP + 2 * LR];
#pragma unroll
            for (int j = 0; j < LP + 2 * LR; ++j) v[j] = hz[m][y0 + j][tx];
#pragma unroll
            for (int o = 0; o < LP; ++o) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k <= 2 * LR; ++k) acc = fmaf(a.w[k], v[o + k], acc);
                conv[m][o] = acc;
            }
        }
    }
    const int x = bx + tx;
    const float invN = 1.f / (float)((double)CHW), invHW = 1.f / (float)((double)HW);
#pragma unroll
    for (int o = 0; o < LP; ++o) {
        const int y = by + y0 + o;
        if ((x >= W) | (y >= H)) continue;
        const size_t pix = (size_t)y * W + x;
        const float p = img[c * HW + pix], q = gt[c * HW + pix];
        float g = 0.f;
        if (a.lambda_dssim != 0.f) g = -a.lambda_dssim * invN * (conv[0][o] + 2.f * p * conv[1][o] + q * conv[2][o]);
        const float d = p - q;
        const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        g = fmaf((1.f - a.lambda_dssim) * invN, sg, g);
        g_img[c * HW + pix] = gl * g;
        if (c != 0) continue;
        if (g_rn && g_sn) {
            const float k = gl * a.lambda_normal * invHW;
            if (a.normal_mode == 1) {
                const float kw = k * weight[pix];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const float dd = sn[j * HW + pix] - rn[j * HW + pix];
                    const float sgn = dd > 0.f ? 1.f : (dd < 0.f ? -1.f : 0.f);
                    g_sn[j * HW + pix] = kw * sgn;
                    g_rn[j * HW + pix] = -kw * sgn;
                }
            } else if (a.normal_mode == 2) {
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    g_rn[j * HW + pix] = -k * sn[j * HW + pix];
                    g_sn[j * HW + pix] = -k * rn[j * HW + pix];
                }
            } else {
#pragma unroll
                for (int j = 0; j < 3; ++j) { g_rn[j * HW + pix] = 0.f; g_sn[j * HW + pix] = 0.f; }
            }
        }
        if (g_dist) g_dist[pix] = a.lambda_dist > 0.f ? gl * a.lambda_dist * invHW : 0.f;
    }
}

int make_args(const MrgsLossConfig* cfg, bool has_weight, LossArgs& a)
{
    if (!cfg || cfg->H <= 0 || cfg->W <= 0 || cfg->C <= 0 || cfg->C > 4) return MRGS_E_BAD_ARG;
    a.H = cfg->H; a.W = cfg->W; a.C = cfg->C;
    a.lambda_dssim = cfg->lambda_dssim; a.lambda_normal = cfg->lambda_normal; a.lambda_dist = cfg->lambda_dist;
    a.normal_mode = cfg->lambda_normal > 0.f ? (has_weight ? 1 : 2) : 0;
    // gaussian(11, 1.5) (loss_utils.py:28-30): exp in double, stored as fp32, divided by their fp32 sum (torch's sum of these 11
    // values equals the correctly rounded one; tests/golden/reference_loss.npz holds the resulting window)
    float g[2 * LR + 1];
    double sum = 0.0;
    for (int i = 0; i <= 2 * LR; ++i) {
        const double d = (double)(i - LR);
        g[i] = (float)exp(-(d * d) / (2.0 * 1.5 * 1.5));
        sum += (double)g[i];
    }
    for (int i = 0; i <= 2 * LR; ++i) a.w[i] = g[i] / (float)sum;
    return MRGS_OK;
}

dim3 loss_grid(const LossArgs& a) { return dim3((a.W + LT - 1) / LT, (a.H + LT - 1) / LT, a.C); }

}   // namespace

extern "C" size_t mrgs_loss_ws_bytes(int32_t H, int32_t W, int32_t C)
{
    if (H <= 0 || W <= 0 || C <= 0) return 0;
    const size_t nb = (size_t)((W + LT - 1) / LT) * ((H + LT - 1) / LT) * C;
    return (((3 * (size_t)C * H * W + 3) & ~(size_t)3) + nb * NPART) * sizeof(float);
}

extern "C" int mrgs_loss_forward(const MrgsLossConfig* cfg, const float* image, const float* gt, const float* rend_normal,
                                 const float* surf_normal, const float* rend_dist, const float* image_weight, void* ws,
                                 size_t ws_bytes, float* out_terms, float* out_loss, void* stream)
{
    LossArgs a;
    const int rc = make_args(cfg, image_weight != nullptr, a);
    if (rc) return rc;
    if (!image || !gt || !ws || !out_terms) return MRGS_E_BAD_ARG;
    if (a.normal_mode && (!rend_normal || !surf_normal)) return MRGS_E_BAD_ARG;
    if (a.lambda_dist > 0.f && !rend_dist) return MRGS_E_BAD_ARG;
    if (ws_bytes < mrgs_loss_ws_bytes(a.H, a.W, a.C)) return MRGS_E_WORKSPACE;
    float* dmaps = (float*)ws;
    float* partials = dmaps + ((3 * (size_t)a.C * a.H * a.W + 3) & ~(size_t)3);   // 16-byte aligned rows (read as float4)
    const dim3 grid = loss_grid(a);
    hipStream_t st = (hipStream_t)stream;
    loss_fwd_kernel<<<grid, 256, 0, st>>>(a, image, gt, rend_normal, surf_normal, rend_dist, image_weight, dmaps, partials);
    loss_finalize_kernel<<<1, 256, 0, st>>>(a, partials, (int)(grid.x * grid.y), out_terms, out_loss);
    return hipGetLastError() == hipSuccess ? MRGS_OK : MRGS_E_HIP;
}

extern "C" int mrgs_loss_backward(const MrgsLossConfig* cfg, const float* image, const float* gt, const float* rend_normal,
                                  const float* surf_normal, const float* image_weight, const void* ws, const float* g_loss,
                                  float* g_image, float* g_rend_normal, float* g_surf_normal, float* g_rend_dist, void* stream)
{
    LossArgs a;
    const int rc = make_args(cfg, image_weight != nullptr, a);
    if (rc) return rc;
    if (!image || !gt || !ws || !g_image) return MRGS_E_BAD_ARG;
    if (a.normal_mode && (!rend_normal || !surf_normal || !g_rend_normal || !g_surf_normal)) return MRGS_E_BAD_ARG;
    if (a.lambda_dist > 0.f && !g_rend_dist) return MRGS_E_BAD_ARG;
    loss_bwd_kernel<<<loss_grid(a), 256, 0, (hipStream_t)stream>>>(a, image, gt, rend_normal, surf_normal, image_weight, (const float*)ws,
                                                                  g_loss, g_image, g_rend_normal, g_surf_normal, g_rend_dist);
    return hipGetLastError() == hipSuccess ? MRGS_OK : MRGS_E_HIP;
}
