// Photometric + geometric training loss of one view, fused.
//
// Replaces calculate_loss (utils/loss_utils.py:142-228) with l1_loss (:22-23), ssim/_ssim/create_window (:83-119): the reference
// runs five depthwise 11x11 conv2d launches, ~30 elementwise kernels and their autograd mirror per view; here
//   loss_fwd_kernel   one 16x16 pixel tile per workgroup and channel: both images staged with a 5-pixel halo in LDS, the five
//                     window moments by a separable pass (rows then columns), SSIM and its three partial derivatives with respect
//                     to (mu1, E[x^2], E[xy]) written as maps, |x-y|, (x-y)^2, the normal-consistency term and the distortion
//                     term reduced per workgroup (fixed order: results are run-to-run identical),
//   loss_finalize_kernel   one workgroup sums the per-workgroup partials in double and writes the scalar terms,
//   loss_bwd_kernel   same tiling: the three derivative maps are convolved with the (symmetric) window and combined with the
//                     pixel values into dL/dimage; L1 sign term, normal and distortion gradients are added in the same pass.
// Zero padding as F.conv2d(padding=5): pixels outside the image count as 0 in the moments and carry no derivative.
#include "mrgs_internal.h"

namespace {

constexpr int LT = 16;            // tile edge
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
    const int bx = blockIdx.x * LT, by = blockIdx.y * LT, tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const size_t HW = (size_t)H * W;
    const float* I1 = img + c * HW;
    const float* I2 = gt + c * HW;
    for (int i = tid; i < LH * LH; i += 256) {
        const int r = i / LH, cc = i - r * LH, y = by + r - LR, x = bx + cc - LR;
        const bool in = (x >= 0) & (x < W) & (y >= 0) & (y < H);
        s1[r][cc] = in ? I1[(size_t)y * W + x] : 0.f;
        s2[r][cc] = in ? I2[(size_t)y * W + x] : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < LH * LT; i += 256) {
        const int r = i >> 4, cc = i & 15;
        float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
        for (int k = 0; k <= 2 * LR; ++k) {
            const float p = s1[r][cc + k], q = s2[r][cc + k], w = a.w[k];
            const float wp = w * p, wq = w * q;
            m1 += wp; m2 += wq; e11 = fmaf(wp, p, e11); e22 = fmaf(wq, q, e22); e12 = fmaf(wp, q, e12);
        }
        hz[0][r][cc] = m1; hz[1][r][cc] = m2; hz[2][r][cc] = e11; hz[3][r][cc] = e22; hz[4][r][cc] = e12;
    }
    __syncthreads();
    float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
    for (int k = 0; k <= 2 * LR; ++k) {
        const float w = a.w[k];
        mu1 = fmaf(w, hz[0][ty + k][tx], mu1); mu2 = fmaf(w, hz[1][ty + k][tx], mu2);
        e11 = fmaf(w, hz[2][ty + k][tx], e11); e22 = fmaf(w, hz[3][ty + k][tx], e22); e12 = fmaf(w, hz[4][ty + k][tx], e12);
    }
    const int x = bx + tx, y = by + ty;
    const bool valid = (x < W) & (y < H);
    const size_t pix = (size_t)y * W + x;
    float part[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (valid) {
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
        const float d = s1[ty + LR][tx + LR] - s2[ty + LR][tx + LR];
        part[0] = S; part[1] = fabsf(d); part[2] = d * d;
        if (c == 0) {
            if (a.normal_mode == 1) {                                    // (image_weight * |surf - rend|.sum(0)).mean()   :170
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < 3; ++k) s += fabsf(sn[k * HW + pix] - rn[k * HW + pix]);
                part[3] = weight[pix] * s;
            } else if (a.normal_mode == 2) {                             // (1 - (rend * surf).sum(0)).mean()   :172-173
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < 3; ++k) s = fmaf(rn[k * HW + pix], sn[k * HW + pix], s);
                part[3] = 1.f - s;
            }
            if (a.lambda_dist > 0.f) part[4] = dist[pix];
        }
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) part[k] = wave_sum(part[k]);
    if ((tid & 63) == 0)
#pragma unroll
        for (int k = 0; k < 5; ++k) red[tid >> 6][k] = part[k];
    __syncthreads();
    if (tid < 5) {
        const size_t b = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        partials[b * NPART + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    }
}

// out[0] loss, [1] Ll1, [2] ssim, [3] loss0, [4] normal term (mean, unscaled), [5] lambda_dist * mean(rend_dist), [6] psnr,
// [7..7+C) per-channel mse
__global__ __launch_bounds__(1024) void loss_finalize_kernel(LossArgs a, const float* __restrict__ partials, int blocks_per_channel,
                                                             float* __restrict__ out, float* __restrict__ out_loss)
{
    __shared__ double red[16][8];
    const int tid = threadIdx.x, C = a.C < 4 ? a.C : 4;
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};                              // ssim, l1, normal, dist, sq[0..3]
    const int nb = blocks_per_channel * a.C;
    for (int b = tid; b < nb; b += 1024) {
        const float* p = partials + (size_t)b * NPART;
        const int ch = b / blocks_per_channel;
        acc[0] += p[0]; acc[1] += p[1]; acc[2] += p[3]; acc[3] += p[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[4 + k] += ch == k ? (double)p[2] : 0.0;
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
            for (int w = 0; w < 16; ++w) v += red[w][k];
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
    const int bx = blockIdx.x * LT, by = blockIdx.y * LT, tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const size_t HW = (size_t)H * W, CHW = HW * a.C;
    const float gl = g_loss ? g_loss[0] : 1.f;
    if (a.lambda_dssim != 0.f) {
        for (int i = tid; i < LH * LH; i += 256) {
            const int r = i / LH, cc = i - r * LH, y = by + r - LR, x = bx + cc - LR;
            const bool in = (x >= 0) & (x < W) & (y >= 0) & (y < H);
            const size_t o = c * HW + (size_t)y * W + x;
            s[0][r][cc] = in ? dmaps[o] : 0.f;
            s[1][r][cc] = in ? dmaps[CHW + o] : 0.f;
            s[2][r][cc] = in ? dmaps[2 * CHW + o] : 0.f;
        }
        __syncthreads();
        for (int i = tid; i < LH * LT; i += 256) {
            const int r = i >> 4, cc = i & 15;
            float v0 = 0.f, v1 = 0.f, v2 = 0.f;
#pragma unroll
            for (int k = 0; k <= 2 * LR; ++k) {
                const float w = a.w[k];
                v0 = fmaf(w, s[0][r][cc + k], v0); v1 = fmaf(w, s[1][r][cc + k], v1); v2 = fmaf(w, s[2][r][cc + k], v2);
            }
            hz[0][r][cc] = v0; hz[1][r][cc] = v1; hz[2][r][cc] = v2;
        }
        __syncthreads();
    }
    const int x = bx + tx, y = by + ty;
    if ((x >= W) | (y >= H)) return;
    const size_t pix = (size_t)y * W + x;
    const float p = img[c * HW + pix], q = gt[c * HW + pix];
    const float invN = 1.f / (float)((double)CHW);
    float g = 0.f;
    if (a.lambda_dssim != 0.f) {
        float v0 = 0.f, v1 = 0.f, v2 = 0.f;
#pragma unroll
        for (int k = 0; k <= 2 * LR; ++k) {
            const float w = a.w[k];
            v0 = fmaf(w, hz[0][ty + k][tx], v0); v1 = fmaf(w, hz[1][ty + k][tx], v1); v2 = fmaf(w, hz[2][ty + k][tx], v2);
        }
        g = -a.lambda_dssim * invN * (v0 + 2.f * p * v1 + q * v2);
    }
    const float d = p - q;
    const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    g = fmaf((1.f - a.lambda_dssim) * invN, sg, g);
    g_img[c * HW + pix] = gl * g;
    if (c != 0) return;
    const float invHW = 1.f / (float)((double)HW);
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
    return (3 * (size_t)C * H * W + nb * NPART) * sizeof(float);
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
    float* partials = dmaps + 3 * (size_t)a.C * a.H * a.W;
    const dim3 grid = loss_grid(a);
    hipStream_t st = (hipStream_t)stream;
    loss_fwd_kernel<<<grid, 256, 0, st>>>(a, image, gt, rend_normal, surf_normal, rend_dist, image_weight, dmaps, partials);
    loss_finalize_kernel<<<1, 1024, 0, st>>>(a, partials, (int)(grid.x * grid.y), out_terms, out_loss);
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
