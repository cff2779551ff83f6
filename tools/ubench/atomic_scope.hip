// atomic_scope.hip -- rate of scattered fp32 atomic adds on MI355X: device (agent) scope against XCD-private copies updated with
// workgroup-scope atomics (no sc1: the read-modify-write is done by the issuing XCD's L2, which every CU of that XCD shares), and whether
// the latter loses updates.  Behind the texel-gradient scatter of csrc/mrgs_shade.hip.
//
//   hipcc --offload-arch=gfx950 -O3 -o atomic_scope atomic_scope.hip && ./atomic_scope
//
// Every lane adds 1.0f to `per_lane` pseudo-random texels of a table of `n_texels` floats (3 consecutive floats per texel, as the
// cubemap gradient has).  Checks: the table (or the sum of its 8 copies) must add up to lanes * per_lane * 3 exactly (integers < 2^24
// per texel are exact in fp32).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int MODE>   // 0: agent scope, one table; 1: workgroup scope, table of this wave's XCD; 2: agent scope into the XCD's table (control)
__global__ void __launch_bounds__(256) scatter_kernel(float* __restrict__ table, int n_texels, int per_lane, int locality)
{
    const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 7u;      // HW_REG_XCC_ID
    float* t = MODE == 0 ? table : table + (size_t)xcc * n_texels * 3;
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    for (int k = 0; k < per_lane; ++k) {
        // locality > 0: the lanes of a wave land within a window of `locality` texels (neighbouring pixels mirror into neighbouring texels)
        unsigned h = hash(gid * 977u + k * 131071u);
        unsigned idx = locality > 0 ? (hash((gid >> 6) * 31u + k) + (h % (unsigned)locality)) % (unsigned)n_texels : h % (unsigned)n_texels;
        float* a = t + (size_t)idx * 3;
        if (MODE == 1) {
            __hip_atomic_fetch_add(a + 0, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(a + 1, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(a + 2, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            __hip_atomic_fetch_add(a + 0, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(a + 1, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(a + 2, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__global__ void sum_copies_kernel(const float* __restrict__ table, int n, int copies, float* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int c = 0; c < copies; ++c) s += table[(size_t)c * n + i];
    out[i] = s;
}

template <int MODE>
static void run(const char* name, int n_texels, int lanes, int per_lane, int locality)
{
    const int copies = MODE == 0 ? 1 : 8;
    float *table, *out;
    const size_t n = (size_t)n_texels * 3;
    CHECK(hipMalloc(&table, n * copies * sizeof(float)));
    CHECK(hipMalloc(&out, n * sizeof(float)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    double total = 0;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipMemset(table, 0, n * copies * sizeof(float)));
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(scatter_kernel<MODE>, dim3(lanes / 256), dim3(256), 0, 0, table, n_texels, per_lane, locality);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        hipLaunchKernelGGL(sum_copies_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, table, (int)n, copies, out);
        std::vector<float> h(n);
        CHECK(hipMemcpy(h.data(), out, n * sizeof(float), hipMemcpyDeviceToHost));
        total = 0;
        for (float v : h) total += v;
    }
    const double want = (double)lanes * per_lane * 3;
    printf("%-34s texels %8d lanes %8d x %2d locality %5d : %8.3f ms  %7.2f G atomics/s   sum %.0f / %.0f %s\n", name, n_texels, lanes, per_lane,
           locality, best, want / best / 1e6, total, want, total == want ? "OK" : "LOST UPDATES");
    CHECK(hipFree(table)); CHECK(hipFree(out));
}

int main()
{
    const int lanes = 640000 / 256 * 256;
    for (int locality : {0, 512, 64}) {
        for (int n_texels : {6 * 128 * 128, 6 * 64 * 64, 6 * 16 * 16}) {
            run<0>("agent scope, one table", n_texels, lanes, 4, locality);
            run<2>("agent scope, table per XCD", n_texels, lanes, 4, locality);
            run<1>("workgroup scope, table per XCD", n_texels, lanes, 4, locality);
        }
    }
    return 0;
}
