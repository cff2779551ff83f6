// Optimizer step of the training loop: Adam over every parameter tensor in ONE launch.
//
// Replaces torch.optim.Adam(l, lr=0.0, eps=1e-15).step() as set up by GaussianModel.training_setup
// (scene/gaussian_model.py:417-453: ~18 parameter groups with their own learning rates) -- the default single/multi-tensor
// implementation runs 6-8 elementwise passes per tensor (lerp, mul, addcmul, sqrt, div, add, addcdiv), each re-reading its
// operands from HBM.  Here a table of (param, grad, exp_avg, exp_avg_sq, numel, step size, bias correction) rows travels in the
// kernel arguments, workgroup b looks up its tensor with a short scalar scan and every element is read once (16 B) and written
// once (12 B): the kernel is a pure HBM stream, 28 B per element.
//   exp_avg    <- exp_avg + (grad - exp_avg) * (1 - beta1)
//   exp_avg_sq <- exp_avg_sq * beta2 + (1 - beta2) * grad * grad
//   param      <- param - step_size * exp_avg / (sqrt(exp_avg_sq) / sqrt(1 - beta2^t) + eps),   step_size = lr / (1 - beta1^t)
// (torch/optim/adam.py _single_tensor_adam, non-capturable, amsgrad = False, weight_decay = 0, maximize = False).
#include "mrgs_internal.h"

namespace {

constexpr int ADAM_MAX = MRGS_ADAM_MAX_TENSORS;
constexpr int ADAM_CHUNK = 4096;                     // elements per workgroup: 256 threads x 4 x float4

struct AdamTable {
    float* p[ADAM_MAX];
    const float* g[ADAM_MAX];
    float* m[ADAM_MAX];
    float* v[ADAM_MAX];
    long long numel[ADAM_MAX];
    unsigned chunk_start[ADAM_MAX + 1];
    float step_size[ADAM_MAX], inv_bc2_sqrt[ADAM_MAX];
    int n;
    float w1, beta2, w2, eps;
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float w1, float beta2, float w2, float eps, float step_size,
                                         float inv_bc2_sqrt)
{
    m = fmaf(g - m, w1, m);
    v = fmaf(w2 * g, g, v * beta2);
    const float denom = sqrtf(v) * inv_bc2_sqrt + eps;
    p = p - step_size * (m / denom);
}

template <bool VEC>
__global__ __launch_bounds__(256) void adam_kernel(AdamTable t)
{
    const unsigned b = blockIdx.x;
    int k = 0;
    while (k + 1 < t.n && b >= t.chunk_start[k + 1]) ++k;             // wave-uniform: scalar loop over <= 32 entries
    const long long n = t.numel[k];
    const long long base = (long long)(b - t.chunk_start[k]) * ADAM_CHUNK;
    float* __restrict__ P = t.p[k];
    const float* __restrict__ G = t.g[k];
    float* __restrict__ M = t.m[k];
    float* __restrict__ V = t.v[k];
    const float ss = t.step_size[k], ib = t.inv_bc2_sqrt[k];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long long i = base + ((long long)j * 256 + threadIdx.x) * 4;
        if (i >= n) break;
        if (VEC && i + 3 < n) {
            float4 p = *reinterpret_cast<const float4*>(P + i);
            const float4 g = *reinterpret_cast<const float4*>(G + i);
            float4 m = *reinterpret_cast<const float4*>(M + i);
            float4 v = *reinterpret_cast<const float4*>(V + i);
            adam_one(p.x, g.x, m.x, v.x, t.w1, t.beta2, t.w2, t.eps, ss, ib);
            adam_one(p.y, g.y, m.y, v.y, t.w1, t.beta2, t.w2, t.eps, ss, ib);
            adam_one(p.z, g.z, m.z, v.z, t.w1, t.beta2, t.w2, t.eps, ss, ib);
            adam_one(p.w, g.w, m.w, v.w, t.w1, t.beta2, t.w2, t.eps, ss, ib);
            *reinterpret_cast<float4*>(P + i) = p;
            *reinterpret_cast<float4*>(M + i) = m;
            *reinterpret_cast<float4*>(V + i) = v;
        } else {
            for (int e = 0; e < 4 && i + e < n; ++e) {
                float p = P[i + e], m = M[i + e], v = V[i + e];
                adam_one(p, G[i + e], m, v, t.w1, t.beta2, t.w2, t.eps, ss, ib);
                P[i + e] = p; M[i + e] = m; V[i + e] = v;
            }
        }
    }
}

}   // namespace

extern "C" int mrgs_adam_step(const MrgsAdamTensor* tensors, int32_t n_tensors, double beta1, double beta2, double eps, void* stream)
{
    if (n_tensors < 0 || (n_tensors > 0 && !tensors)) return MRGS_E_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    for (int32_t first = 0; first < n_tensors; first += ADAM_MAX) {
        AdamTable t;
        t.n = 0;
        // the scalars of the update in the precision torch applies them: python doubles rounded to fp32 at the tensor op
        // (1 - beta) must be formed in double: 1 - (float)0.999 is off by 1.3e-5 relative
        t.w1 = (float)(1.0 - beta1); t.beta2 = (float)beta2; t.w2 = (float)(1.0 - beta2); t.eps = (float)eps;
        unsigned chunks = 0;
        bool vec = true;
        for (int32_t i = first; i < n_tensors && i < first + ADAM_MAX; ++i) {
            const MrgsAdamTensor& a = tensors[i];
            if (a.numel < 0 || a.step < 1) return MRGS_E_BAD_ARG;
            if (a.numel == 0) continue;
            if (!a.param || !a.grad || !a.exp_avg || !a.exp_avg_sq) return MRGS_E_BAD_ARG;
            const long long nch = (a.numel + ADAM_CHUNK - 1) / ADAM_CHUNK;
            if (nch + chunks > 0x7FFFFFFFll) return MRGS_E_UNSUPPORTED;
            const int k = t.n++;
            t.p[k] = a.param; t.g[k] = a.grad; t.m[k] = a.exp_avg; t.v[k] = a.exp_avg_sq; t.numel[k] = a.numel;
            t.chunk_start[k] = chunks;
            chunks += (unsigned)nch;
            const double bc1 = 1.0 - pow(beta1, (double)a.step), bc2 = 1.0 - pow(beta2, (double)a.step);
            t.step_size[k] = (float)((double)a.lr / bc1);
            t.inv_bc2_sqrt[k] = (float)(1.0 / sqrt(bc2));
            vec = vec && ((((uintptr_t)a.param | (uintptr_t)a.grad | (uintptr_t)a.exp_avg | (uintptr_t)a.exp_avg_sq) & 15) == 0);
        }
        if (t.n == 0) continue;
        t.chunk_start[t.n] = chunks;
        if (vec) adam_kernel<true><<<dim3(chunks), 256, 0, st>>>(t);
        else adam_kernel<false><<<dim3(chunks), 256, 0, st>>>(t);
    }
    return hipGetLastError() == hipSuccess ? MRGS_OK : MRGS_E_HIP;
}
