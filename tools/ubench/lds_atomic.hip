// lds_atomic.hip -- what one LDS float atomic (ds_add_f32, no return) and one LDS compare-and-swap cost on MI355X, per wave instruction,
// as a function of how many lanes are active and how their addresses fall.  Behind the texel-gradient accumulation of the deferred
// shading backward (csrc/mrgs_shade.hip), which keeps the coarse cubemap levels in LDS.
//
//   hipcc --offload-arch=gfx950 -O3 -o lds_atomic lds_atomic.hip && ./lds_atomic
//
// One workgroup of `waves` wavefronts per CU; every wave issues ITER x 8 atomics; clock from s_memtime (100 MHz x ... reported as wall
// time via events instead: cycles = time x 2.4 GHz / (ITER x 8) / waves per CU, i.e. LDS-pipe cycles per wave instruction when the pipe
// is the bound).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// PATTERN 0: lane l -> word l (conflict-free); 1: random words of a 24 K-float table; 2: all lanes one word; 3: groups of 8 lanes share a word;
// 4: random texel * 3 + c (three consecutive words per lane over three instructions, as the cubemap gradient)
template <int OP, int PATTERN>
__global__ void __launch_bounds__(1024) k(float* out, int iters, int active)
{
    __shared__ float tab[24576];
    for (int i = threadIdx.x; i < 24576; i += blockDim.x) tab[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const bool on = lane < active;
    unsigned h = hash(threadIdx.x * 977u + blockIdx.x * 7919u);
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            h = hash(h + u);
            unsigned w;
            if (PATTERN == 0) w = lane + 64 * ((h >> 8) % 300u) - lane + lane;      // lane-linear inside a random 64-word row
            else if (PATTERN == 1) w = h % 24576u;
            else if (PATTERN == 2) w = __builtin_amdgcn_readfirstlane(h) % 24576u;
            else if (PATTERN == 3) w = hash(__builtin_amdgcn_readfirstlane(h) + (lane >> 3)) % 24576u;
            else w = (h % 8192u) * 3u + (u % 3);
            if (PATTERN == 0) w = (w / 64) * 64 + lane;
            if (on) {
                if (OP == 0) atomicAdd(&tab[w], 1.0f);
                else if (OP == 1) acc += atomicAdd(&tab[w], 1.0f);                   // returning
                else if (OP == 2) acc += __int_as_float(atomicCAS((int*)&tab[w], 0, (int)h | 1));
                else if (OP == 4) atomicAdd((unsigned*)&tab[w], 3u);                  // ds_add_u32
                else if (OP == 5) atomicAdd((unsigned long long*)&tab[w & ~1u], 3ull); // ds_add_u64 (12 288 entries of 8 bytes)
                else tab[w] = 1.0f;                                                   // plain store (reference)
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = tab[5] + acc;
}

template <int OP, int PATTERN>
static void run(const char* name, int waves, int active)
{
    float* out;
    CHECK(hipMalloc(&out, 4096));
    const int iters = 2000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<OP, PATTERN>), dim3(256), dim3(waves * 64), 0, 0, out, 10, active);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<OP, PATTERN>), dim3(256), dim3(waves * 64), 0, 0, out, iters, active);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double per = ms * 1e-3 * 2.4e9 / ((double)iters * 8 * waves);
    printf("%-44s waves/CU %2d active %2d: %7.1f cycles per wave instruction (pipe-bound reading), %6.2f G lane-ops/s chip\n", name, waves, active, per,
           (double)iters * 8 * waves * 256 * active / (ms * 1e-3) / 1e9);
    CHECK(hipFree(out));
}

int main()
{
    for (int waves : {4, 12}) {
        run<3, 1>("ds_write_b32 random", waves, 64);
        run<0, 0>("ds_add_f32 lane-linear", waves, 64);
        run<0, 1>("ds_add_f32 random words", waves, 64);
        run<0, 1>("ds_add_f32 random words", waves, 16);
        run<0, 1>("ds_add_f32 random words", waves, 4);
        run<0, 4>("ds_add_f32 random texel*3+c", waves, 64);
        run<0, 2>("ds_add_f32 one word", waves, 64);
        run<0, 2>("ds_add_f32 one word", waves, 8);
        run<0, 3>("ds_add_f32 8 lanes per word", waves, 64);
        run<4, 1>("ds_add_u32 random words", waves, 64);
        run<4, 4>("ds_add_u32 random texel*3+c", waves, 64);
        run<4, 3>("ds_add_u32 8 lanes per word", waves, 64);
        run<5, 1>("ds_add_u64 random words", waves, 64);
        run<5, 4>("ds_add_u64 random texel*3+c", waves, 64);
        run<5, 3>("ds_add_u64 8 lanes per word", waves, 64);
        run<1, 1>("ds_add_rtn_f32 random words", waves, 64);
        run<2, 1>("ds_cmpst_rtn_b32 random words", waves, 64);
        run<2, 3>("ds_cmpst_rtn_b32 8 lanes per word", waves, 64);
    }
    return 0;
}
