#include <hip/hip_runtime.h>
#include <cstdio>
template<int NACC>
__global__ void k(float* out, int iters, float a, float b) {
    float acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = threadIdx.x + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++)
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = __builtin_fmaf(acc[i], a, b);
    }
    float s = 0; for (int i = 0; i < NACC; i++) s += acc[i];
    if (s == 12345.f) out[0] = s;
}
template<int NACC> void run(int wps, float* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int iters = 20000;
    int grid = 256 * 4 * wps;
    k<NACC><<<grid, 64>>>(d, 10, 1.0001f, 0.5f);
    hipEventRecord(e0);
    k<NACC><<<grid, 64>>>(d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double instr_per_simd = (double)iters * 16 * NACC * wps;
    printf("NACC=%d waves/SIMD=%d: %.3f ms, %.2f cycles/instr/SIMD at 2.4GHz\n", NACC, wps, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
}
int main() {
    float* d; hipMalloc(&d, 4);
    for (int w : {1, 2, 4, 8}) { run<1>(w, d); run<2>(w, d); run<4>(w, d); run<8>(w, d); }
    return 0;
}
