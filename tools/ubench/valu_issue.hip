#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2_ __attribute__((ext_vector_type(2)));
template<int NACC, int MODE>
__global__ void k(float* out, int iters, float a, float b) {
    float2_ acc[NACC];
    for (int i = 0; i < NACC; i++) { acc[i].x = threadIdx.x + i; acc[i].y = threadIdx.x - i; }
    float2_ va = {a, a}, vb = {b, b};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++)
#pragma unroll
            for (int i = 0; i < NACC; i++) {
                if (MODE == 0) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i].x) : "v"(a), "v"(b)); }
                else if (MODE == 1) { asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(va), "v"(vb)); }
                else if (MODE == 2) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(va)); }
                else if (MODE == 3) { asm volatile("v_exp_f32 %0, %0" : "+v"(acc[i].x)); }
                else if (MODE == 4) { asm volatile("v_rcp_f32 %0, %0" : "+v"(acc[i].x)); }
                else if (MODE == 5) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(acc[i].x) : "v"(a)); }
                else if (MODE == 6) { asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(acc[i].x)); }
                else if (MODE == 7) { asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(acc[i].x), "+v"(acc[i].y)); }
                else if (MODE == 8) { asm volatile("v_mov_b32 %0, %1" : "=v"(acc[i].x) : "v"(acc[i].y)); }
                else if (MODE == 9) { asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(acc[i].x) : "v"(a) : "s20", "s21"); }
                else if (MODE == 10) { asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(acc[i].x), "v"(a) : "vcc"); }
                else if (MODE == 11) { asm volatile("v_min_f32 %0, %0, %1" : "+v"(acc[i].x) : "v"(a)); }
                else if (MODE == 12) { asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i].x) : "v"(a), "v"(b)); }
                else if (MODE == 13) { asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(acc[i].x) : "s"(a)); }
                else if (MODE == 14) { asm volatile("v_mul_f32 %0, %0, %1" : "+v"(acc[i].x) : "v"(a)); }
                else if (MODE == 15) { asm volatile("v_mul_f32 %0, %1, %0" : "+v"(acc[i].x) : "s"(a)); }
                else if (MODE == 17) { asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(acc[i].x) : "v"(a)); }
                else if (MODE == 18) { asm volatile("v_cndmask_b32_e64 %0, %1, %2, s[20:21]" : "=v"(acc[i].x) : "v"(acc[i].y), "v"(a) : "s20", "s21"); }
                else if (MODE == 16) { asm volatile("v_cndmask_b32 %0, 0, %0, vcc" : "+v"(acc[i].x)); }
            }
    }
    float s = 0; for (int i = 0; i < NACC; i++) s += acc[i].x + acc[i].y;
    if (s == 12345.f) out[0] = s;
}
template<int MODE> void run(const char* name, float* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, wps = 4, NACC = 8;
    int grid = 256 * 4 * wps;
    k<NACC, MODE><<<grid, 64>>>(d, 10, 1.0001f, 0.5f);
    hipEventRecord(e0);
    k<NACC, MODE><<<grid, 64>>>(d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double instr_per_simd = (double)iters * 16 * NACC * wps;
    printf("%-22s %.2f cycles/instr/SIMD (8 independent, 4 waves/SIMD, 2.4 GHz assumed)\n", name, ms * 1e-3 * 2.4e9 / instr_per_simd);
}
int main() {
    float* d; hipMalloc(&d, 4);
    run<0>("v_fma_f32", d); run<1>("v_pk_fma_f32", d); run<2>("v_pk_mul_f32", d); run<3>("v_exp_f32", d); run<4>("v_rcp_f32", d);
    run<5>("v_cndmask_b32 vcc", d); run<9>("v_cndmask_e64 sgpr", d); run<10>("v_cmp_lt_f32 vcc", d); run<11>("v_min_f32", d); run<12>("v_fmac_f32 vvv", d); run<13>("v_fma_f32 v,s,v(same)", d); run<14>("v_mul_f32 v,v", d); run<15>("v_mul_f32 s,v", d); run<16>("v_cndmask 0,v vcc", d); run<17>("v_cndmask_e64 vcc", d); run<18>("v_cndmask_e64 3 vgpr", d); run<6>("v_add_f32_dpp", d); run<7>("v_permlane32_swap", d); run<8>("v_mov_b32", d);
    return 0;
}
