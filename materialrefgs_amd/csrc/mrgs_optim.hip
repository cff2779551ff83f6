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

// ---- densify / prune compaction (scene/gaussian_model.py:856-905: _prune_optimizer + prune_points) -------------------------------
// The reference drops pruned gaussians with `tensor[mask]` on each of ~16 parameter tensors, their two Adam moments and three
// statistics vectors: ~50 boolean-index calls, each with its own nonzero() pass and host sync.  Here the keep mask is scanned once
// (count per 1024-row block, one-workgroup scan of the block counts) and ONE launch gathers the surviving rows of every tensor:
// grid = (row blocks, tensors), rows keep their order (stable, like boolean indexing).
namespace {

constexpr int COMPACT_ROWS = 1024;      // rows per workgroup

struct CompactTable {
    const float* src[MRGS_COMPACT_MAX_TENSORS];
    float* dst[MRGS_COMPACT_MAX_TENSORS];
    int row_floats[MRGS_COMPACT_MAX_TENSORS];
};

__device__ __forceinline__ unsigned block_exclusive_scan_256(unsigned v, unsigned* s_wave, unsigned& total)
{
    // v = this thread's count; returns the exclusive prefix over the 256 threads of the workgroup
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned n = __shfl_up(inc, o, 64);
        if (lane >= o) inc += n;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    unsigned base = 0;
    for (int w = 0; w < wave; ++w) base += s_wave[w];
    total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    return base + inc - v;
}

__global__ __launch_bounds__(256) void compact_count_kernel(long long n, const uint8_t* __restrict__ keep, unsigned* __restrict__ block_count)
{
    __shared__ unsigned s_wave[4];
    const long long r0 = (long long)blockIdx.x * COMPACT_ROWS + threadIdx.x * 4;
    unsigned c = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) c += (r0 + j < n && keep[r0 + j]) ? 1u : 0u;
    unsigned total;
    block_exclusive_scan_256(c, s_wave, total);
    if (threadIdx.x == 0) block_count[blockIdx.x] = total;
}

__global__ __launch_bounds__(1024) void compact_scan_kernel(int nblocks, const unsigned* __restrict__ block_count, unsigned* __restrict__ block_off,
                                                            long long* __restrict__ total_out)
{
    __shared__ unsigned s_part[1024];
    const int per = (nblocks + 1023) / 1024, b0 = threadIdx.x * per;
    unsigned s = 0;
    for (int i = 0; i < per && b0 + i < nblocks; ++i) s += block_count[b0 + i];
    s_part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned run = 0;
        for (int i = 0; i < 1024; ++i) { const unsigned v = s_part[i]; s_part[i] = run; run += v; }
        total_out[0] = (long long)run;
    }
    __syncthreads();
    unsigned run = s_part[threadIdx.x];
    for (int i = 0; i < per && b0 + i < nblocks; ++i) { block_off[b0 + i] = run; run += block_count[b0 + i]; }
}

__global__ __launch_bounds__(256) void compact_gather_kernel(long long n, const uint8_t* __restrict__ keep, const unsigned* __restrict__ block_off,
                                                             CompactTable t)
{
    __shared__ unsigned s_wave[4];
    __shared__ unsigned s_dst[COMPACT_ROWS];        // destination row of each kept row of this block, 0xFFFFFFFF for dropped rows
    const long long rb = (long long)blockIdx.x * COMPACT_ROWS, r0 = rb + threadIdx.x * 4;
    unsigned k[4], c = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { k[j] = (r0 + j < n && keep[r0 + j]) ? 1u : 0u; c += k[j]; }
    unsigned total;
    unsigned pre = block_exclusive_scan_256(c, s_wave, total) + block_off[blockIdx.x];
#pragma unroll
    for (int j = 0; j < 4; ++j) { s_dst[threadIdx.x * 4 + j] = k[j] ? pre : 0xFFFFFFFFu; pre += k[j]; }
    __syncthreads();
    if (total == 0) return;
    const int ti = blockIdx.y, L = t.row_floats[ti];
    const float* __restrict__ src = t.src[ti] + rb * L;
    float* __restrict__ dst = t.dst[ti];
    const long long rows = n - rb < COMPACT_ROWS ? n - rb : COMPACT_ROWS;
    const int ne = (int)rows * L;
    for (int e = threadIdx.x; e < ne; e += 256) {
        const int row = e / L, col = e - row * L;
        const unsigned d = s_dst[row];
        if (d != 0xFFFFFFFFu) dst[(size_t)d * L + col] = src[e];
    }
}

}   // namespace

extern "C" size_t mrgs_compact_ws_bytes(int64_t n_rows)
{
    if (n_rows <= 0) return 256;
    const size_t nb = (size_t)((n_rows + COMPACT_ROWS - 1) / COMPACT_ROWS);
    return mrgs_align_up(2 * nb * sizeof(unsigned), 256) + 256;
}

extern "C" int mrgs_compact_count(int64_t n_rows, const uint8_t* keep, void* ws, size_t ws_bytes, int64_t* count_dev, void* stream)
{
    if (n_rows < 0 || !ws || !count_dev || ws_bytes < mrgs_compact_ws_bytes(n_rows)) return MRGS_E_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (n_rows == 0) return hipMemsetAsync(count_dev, 0, sizeof(int64_t), st) == hipSuccess ? MRGS_OK : MRGS_E_HIP;
    if (!keep) return MRGS_E_BAD_ARG;
    const long long nb = (n_rows + COMPACT_ROWS - 1) / COMPACT_ROWS;
    if (nb > (1 << 22)) return MRGS_E_UNSUPPORTED;
    unsigned* counts = (unsigned*)ws;
    unsigned* offs = counts + nb;
    compact_count_kernel<<<dim3((unsigned)nb), 256, 0, st>>>(n_rows, keep, counts);
    compact_scan_kernel<<<1, 1024, 0, st>>>((int)nb, counts, offs, (long long*)count_dev);
    return hipGetLastError() == hipSuccess ? MRGS_OK : MRGS_E_HIP;
}

extern "C" int mrgs_compact_rows(int64_t n_rows, const uint8_t* keep, const void* ws, const MrgsCompactTensor* tensors, int32_t n_tensors,
                                 void* stream)
{
    if (n_rows < 0 || n_tensors < 0 || (n_tensors > 0 && !tensors)) return MRGS_E_BAD_ARG;
    if (n_rows == 0 || n_tensors == 0) return MRGS_OK;
    if (!keep || !ws) return MRGS_E_BAD_ARG;
    const long long nb = (n_rows + COMPACT_ROWS - 1) / COMPACT_ROWS;
    const unsigned* offs = (const unsigned*)ws + nb;
    for (int32_t first = 0; first < n_tensors; first += MRGS_COMPACT_MAX_TENSORS) {
        CompactTable t;
        int m = 0;
        for (int32_t i = first; i < n_tensors && i < first + MRGS_COMPACT_MAX_TENSORS; ++i) {
            if (tensors[i].row_floats < 0 || tensors[i].row_floats > (1 << 20)) return MRGS_E_BAD_ARG;
            if (tensors[i].row_floats == 0) continue;
            if (!tensors[i].src || !tensors[i].dst) return MRGS_E_BAD_ARG;
            t.src[m] = tensors[i].src; t.dst[m] = tensors[i].dst; t.row_floats[m] = tensors[i].row_floats;
            ++m;
        }
        if (m == 0) continue;
        compact_gather_kernel<<<dim3((unsigned)nb, (unsigned)m), 256, 0, (hipStream_t)stream>>>(n_rows, keep, offs, t);
    }
    return hipGetLastError() == hipSuccess ? MRGS_OK : MRGS_E_HIP;
}
