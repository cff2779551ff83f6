// What FETCH_SIZE counts for the access patterns of the blend kernels (developer micro-benchmark; run under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- tools/ubench/fetch_size
// and compare the counter of each kernel with the bytes it is known to touch, printed below).
//   stream_kernel     every lane 16 consecutive bytes, the wave one contiguous 1 KB run      (what the guide's x2 correction was calibrated on)
//   gather_kernel     every lane 16 bytes at a stride of 80 bytes from a random 80-byte record (global_load_dwordx4)
//   gather_dma_kernel the same gather through the LDS-DMA path (global_load_lds_dwordx4), five pieces per record as render_fwd stages them
//   gather_rep_kernel the same records gathered again by four "quadrant" waves in a row (the re-reads hit L2)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void stream_kernel(const float4* __restrict__ src, float4* __restrict__ out, size_t n)
{
    float4 acc = make_float4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = src[i];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (acc.x == 12345.f) out[0] = acc;
}
__global__ void gather_kernel(const float4* __restrict__ rec, const uint32_t* __restrict__ ids, float4* __restrict__ out, size_t n, int pieces)
{
    float4 acc = make_float4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4* r = rec + (size_t)ids[i] * 5;
        for (int p = 0; p < pieces; p++) { const float4 v = r[p]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    }
    if (acc.x == 12345.f) out[0] = acc;
}
__global__ void __launch_bounds__(64) gather_dma_kernel(const float4* __restrict__ rec, const uint32_t* __restrict__ ids, float4* __restrict__ out, size_t n)
{
    __shared__ float4 stage[5][64];
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 64 + threadIdx.x; i < n; i += (size_t)gridDim.x * 64) {
        const float4* r = rec + (size_t)ids[i] * 5;
#pragma unroll
        for (int p = 0; p < 5; p++) __builtin_amdgcn_global_load_lds(r + p, &stage[p][0], 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        acc += stage[0][threadIdx.x].x + stage[4][threadIdx.x].y;
        __builtin_amdgcn_wave_barrier();
    }
    if (acc == 12345.f) out[0] = make_float4(acc, 0, 0, 0);
}

int main()
{
    const size_t P = 300000, R = 1153513;              // surfel records, list entries (C2)
    std::vector<uint32_t> ids(R), ids4(4 * R);
    srand(1);
    // a tile's list: surfels in depth order, i.e. scattered over the record array; consecutive list entries are unrelated records
    for (size_t i = 0; i < R; i++) ids[i] = (uint32_t)(((uint64_t)rand() * 2654435761ull) % P);
    for (size_t i = 0; i < 4 * R; i++) ids4[i] = ids[((i / 256) * 64 + (i % 64)) % R];      // every 64-entry chunk four times in a row
    float4 *rec, *out, *big; uint32_t *d_ids, *d_ids4;
    const size_t NBIG = (size_t)1 << 26;               // 1 GiB of float4
    hipMalloc(&rec, P * 80); hipMalloc(&out, 64); hipMalloc(&big, NBIG * 16); hipMalloc(&d_ids, R * 4); hipMalloc(&d_ids4, 4 * R * 4);
    hipMemset(rec, 0, P * 80); hipMemset(big, 0, NBIG * 16);
    hipMemcpy(d_ids, ids.data(), R * 4, hipMemcpyHostToDevice); hipMemcpy(d_ids4, ids4.data(), 4 * R * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; rep++) {
        stream_kernel<<<2048, 256>>>(big, out, NBIG);
        gather_kernel<<<2048, 256>>>(rec, d_ids, out, R, 5);
        gather_kernel<<<2048, 256>>>(rec, d_ids, out, R, 1);
        gather_dma_kernel<<<4096, 64>>>(rec, d_ids, out, R);
        gather_dma_kernel<<<4096, 64>>>(rec, d_ids4, out, 4 * R);
    }
    hipDeviceSynchronize();
    printf("stream_kernel:            reads %zu bytes, all of them once\n", NBIG * 16);
    printf("gather_kernel pieces=5:   %zu requests of 80 bytes = %zu bytes (+ %zu of ids); distinct records %zu = %zu bytes of the array\n", R, R * 80, R * 4, P, P * 80);
    printf("gather_kernel pieces=1:   %zu requests of 16 bytes = %zu bytes (+ ids)\n", R, R * 16);
    printf("gather_dma_kernel:        %zu requests of 80 bytes = %zu bytes (+ ids)\n", R, R * 80);
    printf("gather_dma_kernel x4:     %zu requests of 80 bytes = %zu bytes (+ %zu of ids), every record asked for four times in a row\n", 4 * R, 4 * R * 80, 4 * R * 4);
    return 0;
}
