// mrgs_preprocess.hip -- per-gaussian kernels of the surfel rasterizer on gfx950 (one gaussian per lane).
//
//   preprocess_fwd_kernel : FORWARD::preprocess / preprocessCUDA (forward.cu:163-266) + compute_transmat (:77-125)
//                           + compute_aabb (:129-159) + computeColorFromSH (:22-73) + getRect (auxiliary.h:68-78)
//   preprocess_bwd_kernel : BACKWARD::preprocess / preprocessCUDA (backward.cu:614-669) + compute_transmat_aabb
//                           (:471-612) + computeColorFromSH backward (:22-141) + quat_to_rotmat_vjp (auxiliary.h:245-289)
//   mark_visible_kernel   : checkFrustum (rasterizer_impl.cu:56-68)
//
// This file is compiled with -ffp-contract=off: every discrete decision of the pipeline (cull, radius, tile
// rect, sort key) is taken here, and keeping IEEE mul/add/div/sqrt un-fused makes the geometry state -- and
// therefore the whole integer binning state -- reproducible bit for bit (tests/test_gpu_parity.py).  The
// kernels are bandwidth-trivial (~320 B/gaussian), so the un-fused arithmetic costs nothing measurable.
// The 35 camera floats are read from the caller's device tensors through wave-uniform (scalar) loads.
#include "mrgs_internal.h"

__device__ __constant__ float kSH_C0 = 0.28209479177387814f;
__device__ __constant__ float kSH_C1 = 0.4886025119029199f;
__device__ __constant__ float kSH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                           -1.0925484305920792f, 0.5462742152960396f};
__device__ __constant__ float kSH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                           0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                                           -0.5900435899266435f};

__device__ __forceinline__ int f2i_sat(float v)
{
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (-2147483647 - 1);
    return (int)v;
}

// auxiliary.h:220-242, column-major R[c][r]; rsqrtf evaluated as 1/sqrt (IEEE)
__device__ __forceinline__ void quat_to_rotmat(const float4 q, float R[3][3])
{
    const float s = 1.0f / sqrtf(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
    const float w = q.x * s, x = q.y * s, y = q.z * s, z = q.w * s;
    R[0][0] = 1.f - 2.f * (y * y + z * z);
    R[0][1] = 2.f * (x * y + w * z);
    R[0][2] = 2.f * (x * z - w * y);
    R[1][0] = 2.f * (x * y - w * z);
    R[1][1] = 1.f - 2.f * (x * x + z * z);
    R[1][2] = 2.f * (y * z + w * x);
    R[2][0] = 2.f * (x * z + w * y);
    R[2][1] = 2.f * (y * z - w * x);
    R[2][2] = 1.f - 2.f * (x * x + y * y);
}

__device__ __forceinline__ void row_times_proj(const float a[4], const float* pm, float out[4])
{
#pragma unroll
    for (int c = 0; c < 4; c++) out[c] = a[0] * pm[0 + c] + a[1] * pm[4 + c] + a[2] * pm[8 + c] + a[3] * pm[12 + c];
}

__device__ __forceinline__ void get_rect(float px, float py, int max_radius, int gx, int gy, int rmin[2], int rmax[2])
{
    const float r = (float)max_radius;
    rmin[0] = min(gx, max(0, f2i_sat((px - r) / (float)MRGS_BLOCK_X)));
    rmin[1] = min(gy, max(0, f2i_sat((py - r) / (float)MRGS_BLOCK_Y)));
    rmax[0] = min(gx, max(0, f2i_sat((px + r + (float)MRGS_BLOCK_X - 1.0f) / (float)MRGS_BLOCK_X)));
    rmax[1] = min(gy, max(0, f2i_sat((py + r + (float)MRGS_BLOCK_Y - 1.0f) / (float)MRGS_BLOCK_Y)));
}

// Wave-cooperative transfer of 64 consecutive rows of L floats (the [P,M,3] SH tensors) between global memory and a
// per-wave LDS tile with an odd row stride: one lane per gaussian would touch 64 rows 192 B apart with every load/store
// instruction, whereas here every instruction moves 256 contiguous bytes and each lane then works on its own (bank
// conflict-free) LDS row.
#define SH_ROW_MAX 48
#define SH_LDS_STRIDE 49
// Coalesced store of K floats per lane (row-major [row][K] destination, one row per lane): through the wave's LDS tile,
// K dword stores each covering 64 consecutive floats instead of K stores that touch 64 cache lines each.
template <int K>
__device__ __forceinline__ void wave_store_rows(float* __restrict__ tile, const float (&v)[K], float* __restrict__ dst_rows, int nrows, int lane)
{
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < K; j++) tile[lane * K + j] = v[j];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < K; j++) {
        const int e = j * 64 + lane;
        if (e < nrows * K) dst_rows[e] = tile[e];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ void wave_rows_load(float* __restrict__ tile, const float* __restrict__ src, int nrows, int L, int lane)
{
    if (L == 48 && nrows == 64) {
        // 3072 floats = 12 float4 per lane, all twelve loads in flight at once (a float4 never straddles a 48-float row)
        const float4* src4 = reinterpret_cast<const float4*>(src);
        float4 v[12];
#pragma unroll
        for (int k = 0; k < 12; k++) v[k] = src4[k * 64 + lane];
#pragma unroll
        for (int k = 0; k < 12; k++) {
            const int e = 4 * (k * 64 + lane);
            const int r = e / 48, c = e - 48 * r;
            float* d = tile + r * SH_LDS_STRIDE + c;
            d[0] = v[k].x; d[1] = v[k].y; d[2] = v[k].z; d[3] = v[k].w;
        }
        return;
    }
    const int total = nrows * L;
    int r = 0, c = lane;
    while (c >= L) { c -= L; r++; }
    for (int t = lane; t < total; t += 64) {
        tile[r * SH_LDS_STRIDE + c] = src[t];
        c += 64;
        while (c >= L) { c -= L; r++; }
    }
}
__device__ __forceinline__ void wave_rows_store(const float* __restrict__ tile, float* __restrict__ dst, int nrows, int L, int lane)
{
    if (L == 48 && nrows == 64) {
        float4* dst4 = reinterpret_cast<float4*>(dst);
#pragma unroll
        for (int k = 0; k < 12; k++) {
            const int e = 4 * (k * 64 + lane);
            const int r = e / 48, c = e - 48 * r;
            const float* q = tile + r * SH_LDS_STRIDE + c;
            dst4[k * 64 + lane] = make_float4(q[0], q[1], q[2], q[3]);
        }
        return;
    }
    const int total = nrows * L;
    int r = 0, c = lane;
    while (c >= L) { c -= L; r++; }
    for (int t = lane; t < total; t += 64) {
        dst[t] = tile[r * SH_LDS_STRIDE + c];
        c += 64;
        while (c >= L) { c -= L; r++; }
    }
}

// Split SH layout (MrgsRasterInputs::shs_rest): the DC coefficients [P,1,3] and the higher orders [P,M-1,3] live in two tensors, as
// GaussianModel stores them (_features_dc / _features_rest, scene/gaussian_model.py:401-402) -- the reference concatenates them for
// every render (get_features, :256-259), a 58 MB copy each way at P = 300k.  A wave's 64 rows of either tensor are one contiguous,
// 16-byte aligned range: copied with 16-byte accesses between global memory and the per-wave LDS tile, whose row layout
// (DC first) is the one the unsplit path uses.
__device__ __forceinline__ void wave_rows_load_split(float* __restrict__ tile, const float* __restrict__ dc, const float* __restrict__ rest,
                                                     int nrows, int Lr, int lane)
{
    for (int t = lane; t < nrows * 3; t += 64) {
        const int r = t / 3;
        tile[r * SH_LDS_STRIDE + (t - 3 * r)] = dc[t];
    }
    if (Lr == 45 && nrows == 64 && (((uintptr_t)rest) & 15) == 0) {
        const float4* src4 = reinterpret_cast<const float4*>(rest);      // 2880 floats = 720 float4
        float4 v[12];
#pragma unroll
        for (int k = 0; k < 12; k++) v[k] = (k * 64 + lane < 720) ? src4[k * 64 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 12; k++) {
            if (k * 64 + lane >= 720) continue;
            const float q[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int e = 4 * (k * 64 + lane) + j, r = e / 45, c = e - 45 * r;
                tile[r * SH_LDS_STRIDE + 3 + c] = q[j];
            }
        }
        return;
    }
    const int total = nrows * Lr;
    for (int t = lane; t < total; t += 64) {
        const int r = t / Lr;
        tile[r * SH_LDS_STRIDE + 3 + (t - Lr * r)] = rest[t];
    }
}
__device__ __forceinline__ void wave_rows_store_split(const float* __restrict__ tile, float* __restrict__ dc, float* __restrict__ rest, int nrows,
                                                      int Lr, int lane)
{
    for (int t = lane; t < nrows * 3; t += 64) {
        const int r = t / 3;
        dc[t] = tile[r * SH_LDS_STRIDE + (t - 3 * r)];
    }
    if (Lr == 45 && nrows == 64 && (((uintptr_t)rest) & 15) == 0) {
        float4* dst4 = reinterpret_cast<float4*>(rest);
#pragma unroll
        for (int k = 0; k < 12; k++) {
            if (k * 64 + lane >= 720) continue;
            float q[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int e = 4 * (k * 64 + lane) + j, r = e / 45, c = e - 45 * r;
                q[j] = tile[r * SH_LDS_STRIDE + 3 + c];
            }
            dst4[k * 64 + lane] = make_float4(q[0], q[1], q[2], q[3]);
        }
        return;
    }
    const int total = nrows * Lr;
    for (int t = lane; t < total; t += 64) {
        const int r = t / Lr;
        rest[t] = tile[r * SH_LDS_STRIDE + 3 + (t - Lr * r)];
    }
}

// Split layout, the common case (64 rows, 15 higher-order coefficients = 45 floats a row, 16-byte aligned tensors): NO transposition.
// A row of the DC tensor is 3 floats and a row of the rest tensor 45 -- both strides are odd, so a verbatim copy of the wave's two
// contiguous runs (768 B + 11 520 B) already gives every lane a bank-conflict-free row of its own: lane l reads word 3 l + c resp.
// 45 l + c, 32 lanes on 32 banks.  The copy is 16-byte global accesses against linear ds_write_b128 / ds_read_b128.  (Rounds 1-4 laid
// the rows out at stride 49 as the unsplit tensor's 48-float rows need, and the 16-byte pieces of 45-float rows then landed 4 words
// apart on 8 of the 32 banks: 4.7-way conflicts on the wave's 96 transposing ds instructions, 3.4 M conflict cycles per launch of the
// backward at P = 300 k -- tools/lds_bank_model.py.)  Tile: [0, 192) the DC rows, [192, 3072) the rest rows.
#define SH_LIN_REST 192
__device__ __forceinline__ bool wave_rows_linear_ok(const void* dc, const void* rest, int nrows, int Lr)
{
    return Lr == 45 && nrows == 64 && ((((uintptr_t)dc) | ((uintptr_t)rest)) & 15u) == 0;
}
__device__ __forceinline__ void wave_rows_load_linear(float* __restrict__ tile, const float* __restrict__ dc, const float* __restrict__ rest, int lane)
{
    const float4* dc4 = reinterpret_cast<const float4*>(dc);            // 48 float4
    const float4* r4 = reinterpret_cast<const float4*>(rest);           // 720 float4
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 d = lane < 48 ? dc4[lane] : zero;
    float4 v[12];
#pragma unroll
    for (int k = 0; k < 12; k++) v[k] = (k * 64 + lane < 720) ? r4[k * 64 + lane] : zero;
    float4* t4 = reinterpret_cast<float4*>(tile);
    if (lane < 48) t4[lane] = d;
#pragma unroll
    for (int k = 0; k < 12; k++)
        if (k * 64 + lane < 720) t4[SH_LIN_REST / 4 + k * 64 + lane] = v[k];
}
__device__ __forceinline__ void wave_rows_store_linear(const float* __restrict__ tile, float* __restrict__ dc, float* __restrict__ rest, int lane)
{
    const float4* t4 = reinterpret_cast<const float4*>(tile);
    if (lane < 48) reinterpret_cast<float4*>(dc)[lane] = t4[lane];
#pragma unroll
    for (int k = 0; k < 12; k++)
        if (k * 64 + lane < 720) reinterpret_cast<float4*>(rest)[k * 64 + lane] = t4[SH_LIN_REST / 4 + k * 64 + lane];
}

#ifndef MRGS_PRE_STAGE_REC
#define MRGS_PRE_STAGE_REC 1
#endif
template <bool SPLIT>
__global__ void __launch_bounds__(256) preprocess_fwd_kernel(
    int P, int D, int M, int W, int H, int tiles_x, int tiles_y, float scale_modifier, const float* __restrict__ means3D,
    const float* __restrict__ scales, const float* __restrict__ rotations, const float* __restrict__ opacities,
    const float* __restrict__ shs, const float* __restrict__ shs_rest, const float* __restrict__ transMat_precomp,
    const float* __restrict__ colors_precomp, const float* __restrict__ viewmatrix, const float* __restrict__ projmatrix,
    const float* __restrict__ campos, int32_t* __restrict__ radii, float4* __restrict__ rec, float4* __restrict__ cull,
    uint32_t* __restrict__ depth_key, uint32_t* __restrict__ order, uint2* __restrict__ rect, uint32_t* __restrict__ tiles_touched,
    uint8_t* __restrict__ clamped, uint32_t* __restrict__ clear_ptr, unsigned clear_words,
    const float* __restrict__ rec_extra, int rec_extra_stride,   // optional: a per-gaussian value for the record's spare float (the blend kernels' ninth channel)
    uint8_t* __restrict__ visible)                                // optional: radii > 0 as a byte per gaussian (MRGS_HINT_VISIBLE_BYTES)
{
    // SPLIT (shs = DC [P,1,3], shs_rest = [P,M-1,3]): the rows of the two tensors are staged through a per-wave LDS tile in the unsplit
    // row layout; 180-byte rows cannot be fetched per lane with 16-byte loads the way the 192-byte rows of the unsplit tensor are
    // (without the SH tile the records still leave through LDS: 64 x 32 floats per wave, see MRGS_PRE_STAGE_REC below)
    __shared__ __attribute__((aligned(16))) float s_sh[SPLIT ? 4 * 64 * SH_LDS_STRIDE : (MRGS_PRE_STAGE_REC ? 4 * 64 * 32 : 1)];
    // first kernel of a forward: clears the per-call state of the later binning kernels (num_rendered, error flag, CU census,
    // tickets / totals / look-back words of the depth sort and the scan) instead of a separate memset launch
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < clear_words; i += gridDim.x * blockDim.x) clear_ptr[i] = 0u;
    // (Round 6 measured persistent workgroups here -- as many as the chip holds, each walking its share of the 256-gaussian blocks --: the
    //  loop form of this body costs the compiler 234 registers (168 with 40 spilled when capped) and the launch 41 - 51 us against 36.)
    bool sh_linear = false;                 // (wave-uniform) the wave's tile is the verbatim copy: wave_rows_load_linear
    if (SPLIT) {
        const int lane_ = threadIdx.x & 63, wave_ = threadIdx.x >> 6;
        const int row0 = blockIdx.x * blockDim.x + wave_ * 64;
        const int nrows = min(64, P - row0);
        if (nrows > 0 && colors_precomp == nullptr) {
            sh_linear = wave_rows_linear_ok(shs, shs_rest, nrows, (M - 1) * 3);
            if (sh_linear) wave_rows_load_linear(s_sh + wave_ * 64 * SH_LDS_STRIDE, shs + (size_t)row0 * 3, shs_rest + (size_t)row0 * 45, lane_);
            else wave_rows_load_split(s_sh + wave_ * 64 * SH_LDS_STRIDE, shs + (size_t)row0 * 3, shs_rest + (size_t)row0 * (M - 1) * 3, nrows, (M - 1) * 3,
                                      lane_);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // (the SH rows are read straight from global memory here: staging them through LDS as the backward does costs more in
    // occupancy -- 50 KB per workgroup -- than the coalescing gains; 0.060 ms staged vs 0.054 ms direct at P = 300k)
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
#if !MRGS_PRE_STAGE_REC
    if (idx >= P) return;
#endif
    float V[16], PM[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { V[i] = viewmatrix[i]; PM[i] = projmatrix[i]; }
    int out_radius = 0;
    uint32_t out_tiles = 0, out_key = 0xFFFFFFFFu;
    uint2 out_rect = make_uint2(0, 0);
#if MRGS_PRE_STAGE_REC
    // The surfel record (80 bytes) and the cull record (48 bytes) of a lane's gaussian leave through the wave's LDS tile: written per lane
    // they are 5 + 3 store instructions that each touch 64 different 128-byte lines 80 / 48 bytes apart; from the tile the wave's 64
    // records are one contiguous run of 5 120 / 3 072 bytes that leaves as 16 bytes per lane and instruction, lane-contiguous.
    bool wrote_rec = false;
    float4* const st4 = reinterpret_cast<float4*>(s_sh + (threadIdx.x >> 6) * (SPLIT ? 64 * SH_LDS_STRIDE : 64 * 32));
    const int st_lane = threadIdx.x & 63;
    if (idx < P)
#endif
    do {
        const float p[3] = {means3D[3 * (size_t)idx], means3D[3 * (size_t)idx + 1], means3D[3 * (size_t)idx + 2]};
        const float pvx = V[0] * p[0] + V[4] * p[1] + V[8] * p[2] + V[12];
        const float pvy = V[1] * p[0] + V[5] * p[1] + V[9] * p[2] + V[13];
        const float pvz = V[2] * p[0] + V[6] * p[1] + V[10] * p[2] + V[14];
        if (pvz <= 0.2f) break;   // in_frustum, auxiliary.h:207 (the NDC test is commented out in the reference)

        float T[9], nx, ny, nz;
        if (scales != nullptr) {
            float R[3][3];
            quat_to_rotmat(reinterpret_cast<const float4*>(rotations)[idx], R);
            const float2 sc = reinterpret_cast<const float2*>(scales)[idx];
            const float sx = scale_modifier * sc.x, sy = scale_modifier * sc.y;
            float L0[3], L1[3], L2[3];
#pragma unroll
            for (int r = 0; r < 3; r++) {
                L0[r] = R[0][r] * sx + R[1][r] * 0.0f + R[2][r] * 0.0f;
                L1[r] = R[0][r] * 0.0f + R[1][r] * sy + R[2][r] * 0.0f;
                L2[r] = R[0][r] * 0.0f + R[1][r] * 0.0f + R[2][r] * 1.0f;
            }
            const float a0[4] = {L0[0], L0[1], L0[2], 0.0f};
            const float a1[4] = {L1[0], L1[1], L1[2], 0.0f};
            const float a2[4] = {p[0], p[1], p[2], 1.0f};
            float c[3][4];
            row_times_proj(a0, PM, c[0]);
            row_times_proj(a1, PM, c[1]);
            row_times_proj(a2, PM, c[2]);
            const float hw = (float)W / 2.0f, ow = (float)(W - 1) / 2.0f;
            const float hh = (float)H / 2.0f, oh = (float)(H - 1) / 2.0f;
#pragma unroll
            for (int i = 0; i < 3; i++) {
                T[0 + i] = c[i][0] * hw + c[i][1] * 0.0f + c[i][2] * 0.0f + c[i][3] * ow;
                T[3 + i] = c[i][0] * 0.0f + c[i][1] * hh + c[i][2] * 0.0f + c[i][3] * oh;
                T[6 + i] = c[i][0] * 0.0f + c[i][1] * 0.0f + c[i][2] * 0.0f + c[i][3] * 1.0f;
            }
            nx = V[0] * L2[0] + V[4] * L2[1] + V[8] * L2[2];
            ny = V[1] * L2[0] + V[5] * L2[1] + V[9] * L2[2];
            nz = V[2] * L2[0] + V[6] * L2[1] + V[10] * L2[2];
        } else {
#pragma unroll
            for (int i = 0; i < 9; i++) T[i] = transMat_precomp[9 * (size_t)idx + i];
            nx = 0.0f; ny = 0.0f; nz = 1.0f;
        }
        // DUAL_VISIABLE, forward.cu:224-229
        const float cosv = -((pvx * nx + pvy * ny) + pvz * nz);
        if (cosv == 0) break;
        const float mult = cosv > 0 ? 1.0f : -1.0f;
        nx = mult * nx; ny = mult * ny; nz = mult * nz;

        // compute_aabb with cutoff 3 (TIGHTBBOX 0), forward.cu:129-159,235
        const float* T0 = T; const float* T1 = T + 3; const float* T3 = T + 6;
        const float t[3] = {9.0f, 9.0f, -1.0f};
        const float distance = ((T3[0] * T3[0]) * t[0] + (T3[1] * T3[1]) * t[1]) + (T3[2] * T3[2]) * t[2];
        const float inv = 1 / distance;
        const float f[3] = {inv * t[0], inv * t[1], inv * t[2]};
        if (distance == 0.0f) break;
        const float cx = ((f[0] * T0[0]) * T3[0] + (f[1] * T0[1]) * T3[1]) + (f[2] * T0[2]) * T3[2];
        const float cy = ((f[0] * T1[0]) * T3[0] + (f[1] * T1[1]) * T3[1]) + (f[2] * T1[2]) * T3[2];
        const float tmp0 = ((f[0] * T0[0]) * T0[0] + (f[1] * T0[1]) * T0[1]) + (f[2] * T0[2]) * T0[2];
        const float tmp1 = ((f[0] * T1[0]) * T1[0] + (f[1] * T1[1]) * T1[1]) + (f[2] * T1[2]) * T1[2];
        const float h0 = cx * cx - tmp0, h1 = cy * cy - tmp1;
        const float floor_ = 1e-4f;
        const float e0 = sqrtf((h0 != h0) ? floor_ : (floor_ > h0 ? floor_ : h0));
        const float e1 = sqrtf((h1 != h1) ? floor_ : (floor_ > h1 ? floor_ : h1));
        const float radius = ceilf(e0 > e1 ? e0 : e1);
        const int iradius = f2i_sat(radius);
        int rmin[2], rmax[2];
        get_rect(cx, cy, iradius, tiles_x, tiles_y, rmin, rmax);
        if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0) break;

        float rgb[3];
        if (colors_precomp == nullptr) {
            // computeColorFromSH, forward.cu:22-73
            const float dx = p[0] - campos[0], dy = p[1] - campos[1], dz = p[2] - campos[2];
            const float len = sqrtf(dx * dx + dy * dy + dz * dz);
            const float x = dx / len, y = dy / len, z = dz / len;
            const float* sh = shs + (size_t)idx * M * 3;
            uint32_t cl = 0;
            // sh_at(i, c) = coefficient i of channel c.  With the full 16-coefficient layout the 192-byte row is fetched with
            // twelve 16-byte loads into registers (a 4-byte load per coefficient makes the texture path walk 64 cache lines
            // 48 times per wave); other layouts take the plain path.
            auto shade = [&](auto sh_at) {
#pragma unroll
                for (int c = 0; c < 3; c++) {
#define SH(i) sh_at(i, c)
                    float r = kSH_C0 * SH(0);
                    if (D > 0) {
                        r = r - kSH_C1 * y * SH(1) + kSH_C1 * z * SH(2) - kSH_C1 * x * SH(3);
                        if (D > 1) {
                            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                            r = r + kSH_C2[0] * xy * SH(4) + kSH_C2[1] * yz * SH(5) + kSH_C2[2] * (2.0f * zz - xx - yy) * SH(6) +
                                kSH_C2[3] * xz * SH(7) + kSH_C2[4] * (xx - yy) * SH(8);
                            if (D > 2) {
                                r = r + kSH_C3[0] * y * (3.0f * xx - yy) * SH(9) + kSH_C3[1] * xy * z * SH(10) +
                                    kSH_C3[2] * y * (4.0f * zz - xx - yy) * SH(11) +
                                    kSH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SH(12) +
                                    kSH_C3[4] * x * (4.0f * zz - xx - yy) * SH(13) + kSH_C3[5] * z * (xx - yy) * SH(14) +
                                    kSH_C3[6] * x * (xx - 3.0f * yy) * SH(15);
                            }
                        }
                    }
#undef SH
                    r += 0.5f;
                    if (r < 0) cl |= 1u << c;
                    rgb[c] = r > 0.0f ? r : 0.0f;
                }
            };
            if (SPLIT) {
                const float* wt = s_sh + (threadIdx.x >> 6) * 64 * SH_LDS_STRIDE;       // this wave's tile
                const int ln = threadIdx.x & 63;
                const float* dcp = sh_linear ? wt + ln * 3 : wt + ln * SH_LDS_STRIDE;
                const float* row = sh_linear ? wt + SH_LIN_REST + ln * 45 - 3 : wt + ln * SH_LDS_STRIDE;
                shade([&](int i, int c) { return i == 0 ? dcp[c] : row[i * 3 + c]; });
            } else if (M == 16) {
                float row[48];
                const float4* sh4 = reinterpret_cast<const float4*>(sh);   // 192-byte rows of a 16-byte aligned tensor
#pragma unroll
                for (int k = 0; k < 12; k++) {
                    const float4 v = sh4[k];
                    row[4 * k] = v.x; row[4 * k + 1] = v.y; row[4 * k + 2] = v.z; row[4 * k + 3] = v.w;
                }
                shade([&](int i, int c) { return row[i * 3 + c]; });
            } else {
                shade([&](int i, int c) { return sh[i * 3 + c]; });
            }
            clamped[idx] = (uint8_t)cl;
        } else {
            rgb[0] = colors_precomp[3 * (size_t)idx];
            rgb[1] = colors_precomp[3 * (size_t)idx + 1];
            rgb[2] = colors_precomp[3 * (size_t)idx + 2];
        }
        const float opa = opacities[idx];
        // Cull conic for the blend kernels (mrgs_blend_math.h).  A pixel can reach alpha >= 1/255 only where
        // min(rho3d, rho2d) <= tau = 2 ln(255 opacity).
        //  * rho3d: the pixel -> splat map of the blend is p = M (x, y, 1) with M = [Tv x Tw | Tw x Tu | Tu x Tv]
        //    (expand k x l of forward.cu:371-373), so {rho3d <= tau} = {(x,y,1) Q (x,y,1)^T <= 0}, Q = M^T diag(1,1,-tau) M:
        //    an exact conic.  When it is an ellipse it is stored centred and normalised, d^T [[A,B],[B,C]] d <= 1; otherwise
        //    (splat crossing the camera plane) A = B = C = 0, which every block passes.  Evaluated in fp64.
        //  * rho2d: a disc of radius sqrt(tau/2) around mean2D.
        //  The ellipse leaves here as (centre, A, C | B/C, B/A, det/C, det/A): the block test evaluates it as a completed square per
        //  rectangle edge (mrgs_block_may_touch), never as A x^2 + 2 B x y + C y^2, whose terms cancel for the needle of a grazing surfel.
        float4 cull_a = make_float4(1e30f, 1e30f, 1e30f, 1e30f);   // never a candidate: opacity < 1/255 can never pass
        float4 cull_b = make_float4(0.0f, 0.0f, 1e30f, 1e30f);
        float4 cull_c = make_float4(1e30f, 0.0f, -1.0f, 0.0f);
        const float oa = 255.0f * opa;
        if (!(oa < 0.999f)) {
            const float lg = logf(oa);
            const double tau = (double)(2.0f * (lg > 0.0f ? lg : 0.0f) * 1.0001f + 1e-3f);
            const double u[3] = {T0[0], T0[1], T0[2]}, v[3] = {T1[0], T1[1], T1[2]}, w[3] = {T3[0], T3[1], T3[2]};
            // Every quantity below is formed in fp64 TOGETHER WITH A BOUND OF ITS ROUNDING ERROR (running error analysis: one rounding of
            // relative size EPS = 2^-53 per operation, first order, each bound doubled for the second-order terms), and the ellipse that
            // leaves here contains the exact level set of the conic the fp32 inputs define, by construction -- DESIGN.md section 3:
            //   1. cross products c_i = a b - c d: |err| <= 2 EPS (|a b| + |c d|);
            //   2. Q entries q(x, y) = x0 y0 + x1 y1 - tau x2 y2: |err| <= E = 4 EPS sum |terms| + the inputs' bounds propagated;
            //   3. for every pixel p = (x, y) of the image, f(p) = p' Q2 p + 2 q' p + Q11 >= p' (Q2~ - lam I) p + 2 q~' p + (Q11~ - sigma) =: f'(p),
            //      lam = ||E2||_inf >= ||E2||_2, sigma = E11 + 2 (Ex1 W + Ey1 H): the level set of f' contains that of f inside the image;
            //   4. f' has fp64 coefficients: its determinant, centre and minimum are formed with their own bounds -- det' >= det_lo > 0 or
            //      "not an ellipse"; the centre to +-(dxc, dyc), which goes into the record and widens the block's rectangle in the test;
            //      the minimum f'* >= f'(centre~) - e_f - trace(Q2') (dxc^2 + dyc^2) =: f_lo (a point's value bounds the minimum from
            //      above, the centre's error enters at second order), and the ellipse is scaled by f_lo.
            // A needle whose determinant drowns in its own rounding (round 5's miss: det = 9e-8 Qxx Qyy, a surfel edge-on to 3e-4 rad) ends
            // in "not an ellipse" at step 4 or with a rectangle widened by its centre's uncertainty -- no constant decides that (rounds 1-5
            // compared det with 1e-9, then 1e-5 Qxx Qyy, thresholds found by soaking).
            const double EPS = 1.1102230246251565e-16;
            auto crossE = [&](const double (&a_)[3], const double (&b_)[3], double (&c_)[3], double (&e_)[3]) {
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    const int j = (i + 1) % 3, k = (i + 2) % 3;
                    const double t0 = a_[j] * b_[k], t1 = a_[k] * b_[j];
                    c_[i] = t0 - t1;
                    e_[i] = 2.0 * EPS * (fabs(t0) + fabs(t1));
                }
            };
            double c0[3], c1[3], c2[3], e0[3], e1[3], e2[3];
            crossE(v, w, c0, e0);   // Tv x Tw
            crossE(w, u, c1, e1);   // Tw x Tu
            crossE(u, v, c2, e2);   // Tu x Tv
            auto qE = [&](const double (&x_)[3], const double (&ex_)[3], const double (&y_)[3], const double (&ey_)[3], double& val, double& err) {
                const double t0 = x_[0] * y_[0], t1 = x_[1] * y_[1], t2 = tau * (x_[2] * y_[2]);
                val = t0 + t1 - t2;
                const double p0 = fabs(x_[0]) * ey_[0] + fabs(y_[0]) * ex_[0] + ex_[0] * ey_[0];
                const double p1 = fabs(x_[1]) * ey_[1] + fabs(y_[1]) * ex_[1] + ex_[1] * ey_[1];
                const double p2 = fabs(x_[2]) * ey_[2] + fabs(y_[2]) * ex_[2] + ex_[2] * ey_[2];
                err = 2.0 * (4.0 * EPS * (fabs(t0) + fabs(t1) + fabs(t2)) + p0 + p1 + tau * p2);
            };
            double Qxx, Qxy, Qyy, Qx1, Qy1, Q11, Exx, Exy, Eyy, Ex1, Ey1, E11;
            qE(c0, e0, c0, e0, Qxx, Exx);
            qE(c0, e0, c1, e1, Qxy, Exy);
            qE(c1, e1, c1, e1, Qyy, Eyy);
            qE(c0, e0, c2, e2, Qx1, Ex1);
            qE(c1, e1, c2, e2, Qy1, Ey1);
            qE(c2, e2, c2, e2, Q11, E11);
            // default: not an ellipse -> always a candidate (A = 0)
            cull_a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            cull_b = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            float centre_slack = 0.0f;
            const double lam = 2.0 * fmax(Exx + Exy, Exy + Eyy);
            const double sigma = 2.0 * (E11 + 2.0 * (Ex1 * (double)W + Ey1 * (double)H));
            const double Qxx_ = Qxx - lam, Qyy_ = Qyy - lam, Q11_ = Q11 - sigma;      // f' (step 3); from here on these ARE the coefficients
            if (Qxx_ > 0.0 && Qyy_ > 0.0) {
                const double dp = Qxx_ * Qyy_, dq = Qxy * Qxy;
                const double det_ = dp - dq, det_lo = det_ - 2.0 * (2.0 * EPS * (dp + dq));
                if (det_lo > 0.0) {
                    const double n1 = Qyy_ * Qx1, n2 = Qxy * Qy1, m1 = Qxx_ * Qy1, m2 = Qxy * Qx1;
                    const double xc = -(n1 - n2) / det_, yc = -(m1 - m2) / det_;
                    const double e_det = det_ - det_lo;
                    const double dxc = 2.0 * ((2.0 * EPS * (fabs(n1) + fabs(n2)) + fabs(xc) * e_det) / det_lo + 2.0 * EPS * fabs(xc));
                    const double dyc = 2.0 * ((2.0 * EPS * (fabs(m1) + fabs(m2)) + fabs(yc) * e_det) / det_lo + 2.0 * EPS * fabs(yc));
                    const double a1 = 2.0 * Qx1 * xc, a2 = 2.0 * Qy1 * yc, a3 = Qxx_ * xc * xc, a4 = 2.0 * Qxy * xc * yc, a5 = Qyy_ * yc * yc;
                    const double fc = ((Q11_ + a1) + a2) + ((a3 + a4) + a5);
                    const double e_f = 2.0 * (8.0 * EPS * (fabs(Q11_) + fabs(a1) + fabs(a2) + fabs(a3) + fabs(a4) + fabs(a5)));
                    const double f_lo = fc - e_f - (Qxx_ + Qyy_) * (dxc * dxc + dyc * dyc);
                    if (f_lo < 0.0) {
                        const double sc_ = -1.0 / f_lo;
                        const double A = Qxx_ * sc_, B = Qxy * sc_, C = Qyy_ * sc_, dn = det_lo * sc_ * sc_;
                        cull_a = make_float4((float)xc, (float)yc, (float)A, (float)C);
                        cull_b = make_float4((float)(B / C), (float)(B / A), (float)(dn / C), (float)(dn / A));
                        centre_slack = (float)fmax(dxc, dyc) * 1.000001f + 1e-30f;        // (rounded up)
                        const float chk = cull_a.x + cull_a.y + cull_a.z + cull_a.w + cull_b.x + cull_b.y + cull_b.z + cull_b.w + centre_slack;
                        if (!(chk - chk == 0.0f) || !(cull_a.z > 0.0f)) {      // inf / NaN / underflow: never cull
                            cull_a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                            cull_b = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                            centre_slack = 0.0f;
                        }
                    }   // else: the level set may be empty -- or not: stay conservative
                }
            }
            const float rr = sqrtf(0.5f * (float)tau) + 0.05f;
            cull_c = make_float4(cx, cy, rr * rr, centre_slack);
        }
#if MRGS_PRE_STAGE_REC
        __builtin_amdgcn_wave_barrier();         // (the SH rows of the tile have been read by every lane that gets here)
        float4* r4 = st4 + st_lane * MRGS_REC_F4;
        float4* c4 = st4 + 64 * MRGS_REC_F4 + st_lane * MRGS_CULL_F4;
        wrote_rec = true;
#else
        float4* r4 = rec + (size_t)idx * MRGS_REC_F4;
        float4* c4 = cull + (size_t)idx * MRGS_CULL_F4;
#endif
        r4[0] = make_float4(T[0], T[1], T[2], T[3]);
        r4[1] = make_float4(T[4], T[5], T[6], T[7]);
        r4[2] = make_float4(T[8], cx, cy, opa);
        r4[3] = make_float4(nx, ny, nz, rgb[0]);
        r4[4] = make_float4(rgb[1], rgb[2], pvz, rec_extra != nullptr ? rec_extra[(size_t)idx * rec_extra_stride] : 0.0f);
        c4[0] = cull_a;
        c4[1] = cull_b;
        c4[2] = cull_c;
        out_radius = iradius;
        out_tiles = (uint32_t)((rmax[1] - rmin[1]) * (rmax[0] - rmin[0]));
        out_key = __float_as_uint(pvz);
        out_rect = make_uint2((uint32_t)rmin[0] | ((uint32_t)rmin[1] << 16), (uint32_t)rmax[0] | ((uint32_t)rmax[1] << 16));
    } while (0);
#if MRGS_PRE_STAGE_REC
    {
        const uint64_t wm = __builtin_amdgcn_ballot_w64(wrote_rec);          // lanes whose records are in the tile (the others' stay unwritten, as ever)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (wm != 0ull) {
            const size_t row0 = (size_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63u);
            float4* g_rec = rec + row0 * MRGS_REC_F4;
            float4* g_cull = cull + row0 * MRGS_CULL_F4;
#pragma unroll
            for (int k = 0; k < MRGS_REC_F4; k++) {
                const int t = k * 64 + st_lane;
                if ((wm >> (t / MRGS_REC_F4)) & 1ull) g_rec[t] = st4[t];
            }
#pragma unroll
            for (int k = 0; k < MRGS_CULL_F4; k++) {
                const int t = k * 64 + st_lane;
                if ((wm >> (t / MRGS_CULL_F4)) & 1ull) g_cull[t] = st4[64 * MRGS_REC_F4 + t];
            }
        }
        if (idx >= P) return;
    }
#endif
    radii[idx] = out_radius;
    if (visible != nullptr) visible[idx] = out_radius > 0 ? (uint8_t)1 : (uint8_t)0;
    tiles_touched[idx] = out_tiles;
    depth_key[idx] = out_key;
    order[idx] = (uint32_t)idx;
    rect[idx] = out_rect;
}

void mrgs_launch_preprocess_fwd(const MrgsRasterConfig& cfg, const MrgsRasterInputs& in, const MrgsGeomWs& g, int32_t* radii,
                                hipStream_t stream)
{
    const int tiles_x = (cfg.W + MRGS_BLOCK_X - 1) / MRGS_BLOCK_X, tiles_y = (cfg.H + MRGS_BLOCK_Y - 1) / MRGS_BLOCK_Y;
    // rows of nine channels in twelve floats (MrgsRasterInputs::features_live): the ninth goes into the record's spare float, where the
    // <8, true, 9> blend instances read it (the same condition as their dispatch in mrgs_launch_render_fwd / _bwd)
    const float* rec_extra = (cfg.S == 12 && in.features != nullptr && in.features_live == 9u && ((uintptr_t)in.features & 15u) == 0) ? in.features + 8 : nullptr;
    // MRGS_HINT_VISIBLE_BYTES: the caller's radii buffer is P int32 followed by P bytes, which take radii > 0 (the renderers' visibility_filter)
    uint8_t* visible = (in.hint_flags & MRGS_HINT_VISIBLE_BYTES) != 0u ? reinterpret_cast<uint8_t*>(radii + cfg.P) : nullptr;
#define LAUNCH_PRE(SPLIT_)                                                                                                              \
    hipLaunchKernelGGL(preprocess_fwd_kernel<SPLIT_>, dim3((cfg.P + 255) / 256), dim3(256), 0, stream, cfg.P, cfg.D, cfg.M, cfg.W, cfg.H, \
                       tiles_x, tiles_y, cfg.scale_modifier, in.means3D, in.scales, in.rotations, in.opacities, in.shs, in.shs_rest,     \
                       in.transMat_precomp, in.colors_precomp, in.viewmatrix, in.projmatrix, in.campos, radii, g.rec, g.cull,              \
                       g.depth_key[0], g.order[0], g.rect, g.tiles_touched, g.clamped, g.counters,                                       \
                       (unsigned)(g.clear_bytes / sizeof(uint32_t)), rec_extra, 12, visible)
    if (in.shs_rest != nullptr && in.colors_precomp == nullptr) LAUNCH_PRE(true);
    else LAUNCH_PRE(false);
#undef LAUNCH_PRE
}

// ------------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------------
struct f3 { float x, y, z; };

// auxiliary.h:129-139
__device__ __forceinline__ f3 dnormvdv(f3 v, f3 dv)
{
    const float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
    const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    f3 r;
    r.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;
    r.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32;
    r.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32;
    return r;
}

#ifndef MRGS_PREB_WAVES
#define MRGS_PREB_WAVES 1      // wavefronts per workgroup of the backward (each owns a 12.5 KB LDS tile): single waves start as slots free up
                               // instead of four in lock step through the load / compute / store phases (46.5 -> 44.9 us at C2)
#endif
// GLUE (MrgsRasterGrads::glue_params / glue_grads): the backward of render_surfel's per-gaussian glue (mrgs_surfel.hip,
// gaussian_renderer/__init__.py:338-355 + the GaussianModel getters) applied to this kernel's results while they are in registers -- the raw
// parameters' gradients leave instead of dL/d(activated opacity, scales, rotations, features, means3D), which are neither written nor read
// back by a second pass over the P rows.  Only for the case in which nothing reads the blended indirect radiance (feature channels 5..7
// have no upstream gradient: render_surfel without opt.indirect) and without the "pgsr" plane distance: the mirror direction then takes no
// gradient and the glue's backward is the activations' (sigmoid, exp, normalize); the indirect coefficients' gradients are zeros.
struct GlueBwd {
    const float *rotation_raw, *opacity_raw, *scaling_raw, *refl_raw, *rough_raw, *ori_color_raw;
    float *d_xyz, *d_scaling, *d_rotation, *d_opacity, *d_refl, *d_rough, *d_ori_color, *d_indirect_dc, *d_indirect_rest;
    const float* plane_view;        // "pgsr" (rows of 12 floats, channel 8 = get_distance): the camera's world_view_transform as stored, else NULL
};
__device__ __forceinline__ float glue_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }     // mrgs_surfel.hip's sigmoidf

// "pgsr": the plane distance |n_cam . c_cam| (gaussian_renderer/envgs_renderer.py:30-38; mrgs_surfel.hip: make_frame, plane_distance) sends its
// gradient gd to the raw rotation (through the facing unit normal = third column of R(q / |q|), flipped towards the camera) and to the
// centre: d_q[4] is ADDED to, d_p[3] is set.  The formulas of surfel_features_bwd_kernel with a zero gradient at the mirror direction.
__device__ __forceinline__ void glue_plane_distance_bwd(const float* __restrict__ Wv, const float (&p)[3], const float4 q, const float* __restrict__ campos,
                                                        float gd, float (&d_q)[4], float (&d_p)[3])
{
    const float qlen = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    const float qn[4] = {q.x / qlen, q.y / qlen, q.z / qlen, q.w / qlen};
    const float w = qn[0], x = qn[1], y = qn[2], z = qn[3];
    const float nr[3] = {2.0f * (x * z + w * y), 2.0f * (y * z - w * x), 1.0f - 2.0f * (x * x + y * y)};
    const float d[3] = {p[0] - campos[0], p[1] - campos[1], p[2] - campos[2]};
    const float dlen = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    const float v[3] = {d[0] / dlen, d[1] / dlen, d[2] / dlen};
    const float flip = -(nr[0] * v[0] + nr[1] * v[1] + nr[2] * v[2]) >= 0.0f ? 1.0f : -1.0f;
    const float nf[3] = {nr[0] * flip, nr[1] * flip, nr[2] * flip};
    const float nflen = fmaxf(sqrtf(nf[0] * nf[0] + nf[1] * nf[1] + nf[2] * nf[2]), 1e-20f);
    const float nn[3] = {nf[0] / nflen, nf[1] / nflen, nf[2] / nflen};
    float nc[3], cc[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        nc[j] = nn[0] * Wv[j] + nn[1] * Wv[4 + j] + nn[2] * Wv[8 + j];
        cc[j] = p[0] * Wv[j] + p[1] * Wv[4 + j] + p[2] * Wv[8 + j] + Wv[12 + j];
    }
    const float sdist = nc[0] * cc[0] + nc[1] * cc[1] + nc[2] * cc[2];
    const float sg = sdist > 0.0f ? gd : (sdist < 0.0f ? -gd : 0.0f);           // d|s| = sign(s) (torch: 0 at 0)
    float d_nn[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const float* Wi = Wv + 4 * i;
        d_nn[i] = sg * (Wi[0] * cc[0] + Wi[1] * cc[1] + Wi[2] * cc[2]);
        d_p[i] = sg * (Wi[0] * nc[0] + Wi[1] * nc[1] + Wi[2] * nc[2]);
    }
    const float nn_dot = nn[0] * d_nn[0] + nn[1] * d_nn[1] + nn[2] * d_nn[2];
    float a[3];
#pragma unroll
    for (int i = 0; i < 3; i++) a[i] = (d_nn[i] - nn[i] * nn_dot) / nflen * flip;
    float d_qn[4];
    d_qn[0] = 2.0f * y * a[0] - 2.0f * x * a[1];
    d_qn[1] = 2.0f * z * a[0] - 2.0f * w * a[1] - 4.0f * x * a[2];
    d_qn[2] = 2.0f * w * a[0] + 2.0f * z * a[1] - 4.0f * y * a[2];
    d_qn[3] = 2.0f * x * a[0] + 2.0f * y * a[1];
    const float dot1 = ((qn[0] * d_qn[0] + qn[1] * d_qn[1]) + qn[2] * d_qn[2]) + qn[3] * d_qn[3];
#pragma unroll
    for (int i = 0; i < 4; i++) d_q[i] += (d_qn[i] - qn[i] * dot1) / qlen;
}

template <bool GLUE>
__global__ void __launch_bounds__(64 * MRGS_PREB_WAVES) preprocess_bwd_kernel(
    int P, int D, int M, int S, int Wimg, int Himg, float tanfovx, float tanfovy, const float* __restrict__ means3D,
    const float* __restrict__ scales, const float* __restrict__ rotations, const float* __restrict__ shs,
    const float* __restrict__ transMat_precomp, const float* __restrict__ viewmatrix, const float* __restrict__ projmatrix,
    const float* __restrict__ campos, const int32_t* __restrict__ radii, const uint8_t* __restrict__ clamped,
    const float4* __restrict__ rec, const float* __restrict__ grad_rec, int gstride, float* __restrict__ dL_dmeans2D,
    float* __restrict__ dL_dcolors, float* __restrict__ dL_dfeatures, float* __restrict__ dL_dopacity,
    float* __restrict__ dL_dmeans3D, float* __restrict__ dL_dtransMat, float* __restrict__ dL_dsh, float* __restrict__ dL_dscales,
    float* __restrict__ dL_drotations, const float* __restrict__ shs_rest, float* __restrict__ dL_dsh_rest, GlueBwd gl)
{
    __shared__ __attribute__((aligned(16))) float s_sh[MRGS_PREB_WAVES][64 * SH_LDS_STRIDE];
    const int idx_ = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int L = M * 3;
    // SH coefficients in, SH gradients out: staged through a per-wave LDS tile so that global traffic is coalesced
    const bool sh_staged = M > 0 && L <= SH_ROW_MAX;
    const int row0 = blockIdx.x * blockDim.x + wave * 64;
    const int nrows = min(64, P - row0);
    // (wave-uniform) split layout in its common shape: the tile is the verbatim copy of the two tensors' runs (wave_rows_load_linear)
    const bool sh_linear = sh_staged && shs != nullptr && shs_rest != nullptr && dL_dsh_rest != nullptr &&
                           wave_rows_linear_ok(shs, shs_rest, nrows, L - 3) && wave_rows_linear_ok(dL_dsh, dL_dsh_rest, nrows, L - 3);
    if (sh_staged && shs != nullptr && nrows > 0) {
        // (shs_rest: split layout, shs = DC rows; otherwise the tile holds the rows in the unsplit layout either way)
        if (sh_linear) wave_rows_load_linear(s_sh[wave], shs + (size_t)row0 * 3, shs_rest + (size_t)row0 * 45, lane);
        else if (shs_rest != nullptr) wave_rows_load_split(s_sh[wave], shs + (size_t)row0 * 3, shs_rest + (size_t)row0 * (L - 3), nrows, L - 3, lane);
        else wave_rows_load(s_sh[wave], shs + (size_t)row0 * L, nrows, L, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    const bool in_range = idx_ < P;
    const int idx = in_range ? idx_ : P - 1;   // out-of-range lanes stay alive for the cooperative tile store below
    const bool live = in_range && radii[idx] > 0;   // backward.cu:643
    const bool precomp = scales == nullptr;
    const float* gr = grad_rec + (size_t)idx * gstride;

    float dT[9], dm3[3] = {0, 0, 0}, dsc[2] = {0, 0}, drot[4] = {0, 0, 0, 0}, dm2[2] = {0, 0}, dcol[3] = {0, 0, 0}, dop = 0.0f;
#pragma unroll
    for (int i = 0; i < 9; i++) dT[i] = 0.0f;
    float dT_out[9];
#pragma unroll
    for (int i = 0; i < 9; i++) dT_out[i] = 0.0f;

    if (live) {
        float V[16], PM[16];
#pragma unroll
        for (int i = 0; i < 16; i++) { V[i] = viewmatrix[i]; PM[i] = projmatrix[i]; }
        // first 16 floats of the row with four 16-byte loads (rows are 16-byte aligned: the stride is a multiple of 4 floats)
        float g16[16];
        {
            const float4* gr4 = reinterpret_cast<const float4*>(gr);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float4 v = gr4[k];
                g16[4 * k] = v.x; g16[4 * k + 1] = v.y; g16[4 * k + 2] = v.z; g16[4 * k + 3] = v.w;
            }
        }
#pragma unroll
        for (int i = 0; i < 9; i++) { dT[i] = g16[MRGS_G_DT + i]; dT_out[i] = dT[i]; }
        const int m2o = MRGS_G_M2(MRGS_SMAX(S));
        const float2 m2v = *reinterpret_cast<const float2*>(gr + m2o);
        const float m2x = m2v.x, m2y = m2v.y;
        dop = g16[MRGS_G_OPA];
        const float dn[3] = {g16[MRGS_G_NRM], g16[MRGS_G_NRM + 1], g16[MRGS_G_NRM + 2]};
        dcol[0] = g16[MRGS_G_COL]; dcol[1] = g16[MRGS_G_COL + 1]; dcol[2] = g16[MRGS_G_COL + 2];

        // rasterizer_impl.cu:398-399 and backward.cu:646-647: W,H are re-derived from the focal lengths
        const float focal_y = Himg / (2.0f * tanfovy), focal_x = Wimg / (2.0f * tanfovx);
        const int W = f2i_sat(focal_x * tanfovx * 2), H = f2i_sat(focal_y * tanfovy * 2);

        float T[9], Pm[4][3], R[3][3];
        float nx = 0, ny = 0, nz = 0, sx0 = 0, sy0 = 0;
        const float p[3] = {means3D[3 * (size_t)idx], means3D[3 * (size_t)idx + 1], means3D[3 * (size_t)idx + 2]};
        if (precomp) {
#pragma unroll
            for (int i = 0; i < 9; i++) T[i] = transMat_precomp[9 * (size_t)idx + i];
        } else {
            const float2 sc = reinterpret_cast<const float2*>(scales)[idx];
            sx0 = sc.x; sy0 = sc.y;
            quat_to_rotmat(reinterpret_cast<const float4*>(rotations)[idx], R);
            const float sx = 1.0f * sx0, sy = 1.0f * sy0;   // scale_to_mat(scale, 1.0f), backward.cu:509
            float L0[3], L1[3], L2[3];
#pragma unroll
            for (int r = 0; r < 3; r++) {
                L0[r] = R[0][r] * sx + R[1][r] * 0.0f + R[2][r] * 0.0f;
                L1[r] = R[0][r] * 0.0f + R[1][r] * sy + R[2][r] * 0.0f;
                L2[r] = R[0][r] * 0.0f + R[1][r] * 0.0f + R[2][r] * 1.0f;
            }
            const float hw = (float)W / 2.0f, ow = (float)(W - 1) / 2.0f;
            const float hh = (float)H / 2.0f, oh = (float)(H - 1) / 2.0f;
#pragma unroll
            for (int r = 0; r < 4; r++) {   // P = world2ndc * ndc2pix, backward.cu:531
                Pm[r][0] = PM[4 * r + 0] * hw + PM[4 * r + 1] * 0.0f + PM[4 * r + 2] * 0.0f + PM[4 * r + 3] * ow;
                Pm[r][1] = PM[4 * r + 0] * 0.0f + PM[4 * r + 1] * hh + PM[4 * r + 2] * 0.0f + PM[4 * r + 3] * oh;
                Pm[r][2] = PM[4 * r + 0] * 0.0f + PM[4 * r + 1] * 0.0f + PM[4 * r + 2] * 0.0f + PM[4 * r + 3] * 1.0f;
            }
            const float a[3][4] = {{L0[0], L0[1], L0[2], 0.0f}, {L1[0], L1[1], L1[2], 0.0f}, {p[0], p[1], p[2], 1.0f}};
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++)
                    T[3 * j + i] = a[i][0] * Pm[0][j] + a[i][1] * Pm[1][j] + a[i][2] * Pm[2][j] + a[i][3] * Pm[3][j];
            nx = V[0] * L2[0] + V[4] * L2[1] + V[8] * L2[2];
            ny = V[1] * L2[0] + V[5] * L2[1] + V[9] * L2[2];
            nz = V[2] * L2[0] + V[6] * L2[1] + V[10] * L2[2];
        }
        bool returned = false;
        if (m2x != 0 || m2y != 0) {   // backward.cu:543-582
            const float* Tu = T; const float* Tv = T + 3; const float* Tw = T + 6;
            const float distance = Tw[0] * Tw[0] + Tw[1] * Tw[1] - Tw[2] * Tw[2];
            const float f = 1 / distance;
            const float dpx_dT00 = f * Tw[0], dpx_dT01 = f * Tw[1], dpx_dT02 = -f * Tw[2];
            const float dpy_dT10 = f * Tw[0], dpy_dT11 = f * Tw[1], dpy_dT12 = -f * Tw[2];
            const float dpx_dT30 = Tu[0] * (f - 2 * f * f * Tw[0] * Tw[0]);
            const float dpx_dT31 = Tu[1] * (f - 2 * f * f * Tw[1] * Tw[1]);
            const float dpx_dT32 = -Tu[2] * (f + 2 * f * f * Tw[2] * Tw[2]);
            const float dpy_dT30 = Tv[0] * (f - 2 * f * f * Tw[0] * Tw[0]);
            const float dpy_dT31 = Tv[1] * (f - 2 * f * f * Tw[1] * Tw[1]);
            const float dpy_dT32 = -Tv[2] * (f + 2 * f * f * Tw[2] * Tw[2]);
            dT[0] += m2x * dpx_dT00; dT[1] += m2x * dpx_dT01; dT[2] += m2x * dpx_dT02;
            dT[3] += m2y * dpy_dT10; dT[4] += m2y * dpy_dT11; dT[5] += m2y * dpy_dT12;
            dT[6] += m2x * dpx_dT30 + m2y * dpy_dT30;
            dT[7] += m2x * dpx_dT31 + m2y * dpy_dT31;
            dT[8] += m2x * dpx_dT32 + m2y * dpy_dT32;
            if (precomp) {
#pragma unroll
                for (int i = 0; i < 9; i++) dT_out[i] = dT[i];
                returned = true;
            }
        }
        if (!precomp && !returned) {
            float dM[3][4];
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int r = 0; r < 4; r++) dM[i][r] = Pm[r][0] * dT[0 + i] + Pm[r][1] * dT[3 + i] + Pm[r][2] * dT[6 + i];
            float dtx = V[0] * dn[0] + V[1] * dn[1] + V[2] * dn[2];
            float dty = V[4] * dn[0] + V[5] * dn[1] + V[6] * dn[2];
            float dtz = V[8] * dn[0] + V[9] * dn[1] + V[10] * dn[2];
            const float pvx = V[0] * p[0] + V[4] * p[1] + V[8] * p[2] + V[12];
            const float pvy = V[1] * p[0] + V[5] * p[1] + V[9] * p[2] + V[13];
            const float pvz = V[2] * p[0] + V[6] * p[1] + V[10] * p[2] + V[14];
            const float cosv = -((pvx * nx + pvy * ny) + pvz * nz);
            const float mult = cosv > 0 ? 1.0f : -1.0f;
            dtx = mult * dtx; dty = mult * dty; dtz = mult * dtz;
            const float dRS[3][3] = {{dM[0][0], dM[0][1], dM[0][2]}, {dM[1][0], dM[1][1], dM[1][2]}, {dtx, dty, dtz}};
            float vR[3][3];
#pragma unroll
            for (int r = 0; r < 3; r++) { vR[0][r] = dRS[0][r] * sx0; vR[1][r] = dRS[1][r] * sy0; vR[2][r] = dRS[2][r]; }
            {   // quat_to_rotmat_vjp, auxiliary.h:245-289
                const float4 q = reinterpret_cast<const float4*>(rotations)[idx];
                const float s = 1.0f / sqrtf(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
                const float w = q.x * s, x = q.y * s, y = q.z * s, z = q.w * s;
                drot[0] = 2.f * (x * (vR[1][2] - vR[2][1]) + y * (vR[2][0] - vR[0][2]) + z * (vR[0][1] - vR[1][0]));
                drot[1] = 2.f * (-2.f * x * (vR[1][1] + vR[2][2]) + y * (vR[0][1] + vR[1][0]) + z * (vR[0][2] + vR[2][0]) +
                                 w * (vR[1][2] - vR[2][1]));
                drot[2] = 2.f * (x * (vR[0][1] + vR[1][0]) - 2.f * y * (vR[0][0] + vR[2][2]) + z * (vR[1][2] + vR[2][1]) +
                                 w * (vR[2][0] - vR[0][2]));
                drot[3] = 2.f * (x * (vR[0][2] + vR[2][0]) + y * (vR[1][2] + vR[2][1]) - 2.f * z * (vR[0][0] + vR[1][1]) +
                                 w * (vR[0][1] - vR[1][0]));
            }
            dsc[0] = (dRS[0][0] * R[0][0] + dRS[0][1] * R[0][1]) + dRS[0][2] * R[0][2];
            dsc[1] = (dRS[1][0] * R[1][0] + dRS[1][1] * R[1][1]) + dRS[1][2] * R[1][2];
            dm3[0] = dM[2][0]; dm3[1] = dM[2][1]; dm3[2] = dM[2][2];
        }
        // densification proxy, backward.cu:665-668 (uses the raw accumulated dL_dtransMat unless precomp updated it)
        const float depth = precomp ? transMat_precomp[9 * (size_t)idx + 8] : rec[(size_t)idx * MRGS_REC_F4 + 2].x;
        dm2[0] = dT_out[2] * depth * 0.5f * (float)W;
        dm2[1] = dT_out[5] * depth * 0.5f * (float)H;
    }

    // SH backward (backward.cu:22-141), also zero-fills dL_dsh for culled gaussians / unused degrees
    if (M > 0) {
        // this lane's row: coefficient 0 (DC) at dsh0[c], coefficient i >= 1 at dsh[i * 3 + c] (one pointer unless the tile is linear)
        float* tile = &s_sh[wave][lane * SH_LDS_STRIDE];
        float* dsh = sh_linear ? &s_sh[wave][SH_LIN_REST + lane * 45 - 3] : sh_staged ? tile : dL_dsh + (size_t)idx * M * 3;
        float* dsh0 = sh_linear ? &s_sh[wave][lane * 3] : dsh;
        if (live && shs != nullptr) {
            const float* sh = sh_staged ? dsh : shs + (size_t)idx * M * 3;
            const float* sh0 = sh_staged ? dsh0 : sh;
            const f3 dir_orig = {means3D[3 * (size_t)idx] - campos[0], means3D[3 * (size_t)idx + 1] - campos[1],
                                 means3D[3 * (size_t)idx + 2] - campos[2]};
            const float len = sqrtf(dir_orig.x * dir_orig.x + dir_orig.y * dir_orig.y + dir_orig.z * dir_orig.z);
            const float x = dir_orig.x / len, y = dir_orig.y / len, z = dir_orig.z / len;
            const uint32_t cl = clamped[idx];
            float dRGB[3], ddx[3], ddy[3], ddz[3];
#pragma unroll
            for (int c = 0; c < 3; c++) dRGB[c] = dcol[c] * (((cl >> c) & 1u) ? 0.0f : 1.0f);
            const int ncoef = (D + 1) * (D + 1);
#pragma unroll
            for (int c = 0; c < 3; c++) {
                // the coefficients of this channel are read before its gradients overwrite them (in-place tile)
                float shv[16];
#pragma unroll
                for (int i = 0; i < 16; i++) shv[i] = (i < ncoef && i < M) ? (i == 0 ? sh0[c] : sh[i * 3 + c]) : 0.0f;
#define SH(i) shv[i]
#define DSH(i) ((i) == 0 ? dsh0 : dsh + (i) * 3)[c]
                float dRGBdx = 0, dRGBdy = 0, dRGBdz = 0;
                DSH(0) = kSH_C0 * dRGB[c];
                if (D > 0) {
                    DSH(1) = (-kSH_C1 * y) * dRGB[c];
                    DSH(2) = (kSH_C1 * z) * dRGB[c];
                    DSH(3) = (-kSH_C1 * x) * dRGB[c];
                    dRGBdx = -kSH_C1 * SH(3);
                    dRGBdy = -kSH_C1 * SH(1);
                    dRGBdz = kSH_C1 * SH(2);
                    if (D > 1) {
                        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                        DSH(4) = (kSH_C2[0] * xy) * dRGB[c];
                        DSH(5) = (kSH_C2[1] * yz) * dRGB[c];
                        DSH(6) = (kSH_C2[2] * (2.f * zz - xx - yy)) * dRGB[c];
                        DSH(7) = (kSH_C2[3] * xz) * dRGB[c];
                        DSH(8) = (kSH_C2[4] * (xx - yy)) * dRGB[c];
                        dRGBdx += kSH_C2[0] * y * SH(4) + kSH_C2[2] * 2.f * -x * SH(6) + kSH_C2[3] * z * SH(7) + kSH_C2[4] * 2.f * x * SH(8);
                        dRGBdy += kSH_C2[0] * x * SH(4) + kSH_C2[1] * z * SH(5) + kSH_C2[2] * 2.f * -y * SH(6) + kSH_C2[4] * 2.f * -y * SH(8);
                        dRGBdz += kSH_C2[1] * y * SH(5) + kSH_C2[2] * 2.f * 2.f * z * SH(6) + kSH_C2[3] * x * SH(7);
                        if (D > 2) {
                            DSH(9) = (kSH_C3[0] * y * (3.f * xx - yy)) * dRGB[c];
                            DSH(10) = (kSH_C3[1] * xy * z) * dRGB[c];
                            DSH(11) = (kSH_C3[2] * y * (4.f * zz - xx - yy)) * dRGB[c];
                            DSH(12) = (kSH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy)) * dRGB[c];
                            DSH(13) = (kSH_C3[4] * x * (4.f * zz - xx - yy)) * dRGB[c];
                            DSH(14) = (kSH_C3[5] * z * (xx - yy)) * dRGB[c];
                            DSH(15) = (kSH_C3[6] * x * (xx - 3.f * yy)) * dRGB[c];
                            dRGBdx += (kSH_C3[0] * SH(9) * 3.f * 2.f * xy + kSH_C3[1] * SH(10) * yz + kSH_C3[2] * SH(11) * -2.f * xy +
                                       kSH_C3[3] * SH(12) * -3.f * 2.f * xz + kSH_C3[4] * SH(13) * (-3.f * xx + 4.f * zz - yy) +
                                       kSH_C3[5] * SH(14) * 2.f * xz + kSH_C3[6] * SH(15) * 3.f * (xx - yy));
                            dRGBdy += (kSH_C3[0] * SH(9) * 3.f * (xx - yy) + kSH_C3[1] * SH(10) * xz +
                                       kSH_C3[2] * SH(11) * (-3.f * yy + 4.f * zz - xx) + kSH_C3[3] * SH(12) * -3.f * 2.f * yz +
                                       kSH_C3[4] * SH(13) * -2.f * xy + kSH_C3[5] * SH(14) * -2.f * yz +
                                       kSH_C3[6] * SH(15) * -3.f * 2.f * xy);
                            dRGBdz += (kSH_C3[1] * SH(10) * xy + kSH_C3[2] * SH(11) * 4.f * 2.f * yz +
                                       kSH_C3[3] * SH(12) * 3.f * (2.f * zz - xx - yy) + kSH_C3[4] * SH(13) * 4.f * 2.f * xz +
                                       kSH_C3[5] * SH(14) * (xx - yy));
                        }
                    }
                }
#undef SH
#undef DSH
                ddx[c] = dRGBdx; ddy[c] = dRGBdy; ddz[c] = dRGBdz;
            }
            for (int i = ncoef; i < M; i++) { dsh[i * 3] = 0.0f; dsh[i * 3 + 1] = 0.0f; dsh[i * 3 + 2] = 0.0f; }      // (ncoef >= 1: never the DC row)
            const f3 dd = {(ddx[0] * dRGB[0] + ddx[1] * dRGB[1]) + ddx[2] * dRGB[2],
                           (ddy[0] * dRGB[0] + ddy[1] * dRGB[1]) + ddy[2] * dRGB[2],
                           (ddz[0] * dRGB[0] + ddz[1] * dRGB[1]) + ddz[2] * dRGB[2]};
            const f3 dm = dnormvdv(dir_orig, dd);
            dm3[0] += dm.x; dm3[1] += dm.y; dm3[2] += dm.z;
        } else if (in_range || sh_staged) {
            for (int i = 0; i < 3; i++) dsh0[i] = 0.0f;
            for (int i = 3; i < M * 3; i++) dsh[i] = 0.0f;
        }
        if (sh_staged) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (nrows > 0) {
                if (sh_linear) wave_rows_store_linear(s_sh[wave], dL_dsh + (size_t)row0 * 3, dL_dsh_rest + (size_t)row0 * 45, lane);
                else if (dL_dsh_rest != nullptr) wave_rows_store_split(s_sh[wave], dL_dsh + (size_t)row0 * 3, dL_dsh_rest + (size_t)row0 * (L - 3), nrows, L - 3, lane);
                else wave_rows_store(s_sh[wave], dL_dsh + (size_t)row0 * L, nrows, L, lane);
            }
        }
    }
    // the per-gaussian outputs leave through the wave's LDS tile as well (every lane of the wave is still here)
    {
        float* tile = s_sh[wave];
        const size_t r0 = (size_t)row0;
        if (nrows > 0) {
            const float m2v[3] = {dm2[0], dm2[1], 0.0f};
            wave_store_rows<3>(tile, m2v, dL_dmeans2D + 3 * r0, nrows, lane);
            if constexpr (!GLUE) {
                wave_store_rows<3>(tile, dcol, dL_dcolors + 3 * r0, nrows, lane);
                wave_store_rows<3>(tile, dm3, dL_dmeans3D + 3 * r0, nrows, lane);
                wave_store_rows<9>(tile, dT_out, dL_dtransMat + 9 * r0, nrows, lane);
                wave_store_rows<2>(tile, dsc, dL_dscales + 2 * r0, nrows, lane);
                wave_store_rows<4>(tile, drot, dL_drotations + 4 * r0, nrows, lane);
            } else {
                // (dL_dcolors / dL_dtransMat: of interest with precomputed colours / transMat only; NULL = not wanted)
                if (dL_dcolors != nullptr) wave_store_rows<3>(tile, dcol, dL_dcolors + 3 * r0, nrows, lane);
                if (dL_dtransMat != nullptr) wave_store_rows<9>(tile, dT_out, dL_dtransMat + 9 * r0, nrows, lane);
                // ---- the glue's backward on this lane's row (surfel_features_bwd_kernel with a zero gradient at the mirror direction) ----
                // rotations = q / max(|q|, 1e-12) (torch.nn.functional.normalize)
                const float4 q = reinterpret_cast<const float4*>(gl.rotation_raw)[idx];
                const float qlen = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
                const float qn[4] = {q.x / qlen, q.y / qlen, q.z / qlen, q.w / qlen};
                const float dot2 = ((qn[0] * drot[0] + qn[1] * drot[1]) + qn[2] * drot[2]) + qn[3] * drot[3];
                const float rl = fmaxf(qlen, 1e-12f);
                float dq_raw[4] = {(drot[0] - qn[0] * dot2) / rl, (drot[1] - qn[1] * dot2) / rl, (drot[2] - qn[2] * dot2) / rl,
                                   (drot[3] - qn[3] * dot2) / rl};
                // centres: what the rasterizer sends them (+ the plane distance's share, "pgsr")
                float dxyz[3] = {dm3[0], dm3[1], dm3[2]};
                if (gl.plane_view != nullptr) {
                    const float pc3[3] = {means3D[3 * (size_t)idx], means3D[3 * (size_t)idx + 1], means3D[3 * (size_t)idx + 2]};
                    float d_p[3];
                    glue_plane_distance_bwd(gl.plane_view, pc3, q, campos, live ? gr[MRGS_G_FEAT + 8] : 0.0f, dq_raw, d_p);
                    dxyz[0] += d_p[0]; dxyz[1] += d_p[1]; dxyz[2] += d_p[2];
                }
                wave_store_rows<3>(tile, dxyz, gl.d_xyz + 3 * r0, nrows, lane);
                wave_store_rows<4>(tile, dq_raw, gl.d_rotation + 4 * r0, nrows, lane);
                // scales = exp(raw)
                const float2 sr = reinterpret_cast<const float2*>(gl.scaling_raw)[idx];
                const float ds_raw[2] = {dsc[0] * expf(sr.x), dsc[1] * expf(sr.y)};
                wave_store_rows<2>(tile, ds_raw, gl.d_scaling + 2 * r0, nrows, lane);
                // material channels: refl, roughness, ori_color = sigmoid(raw); rows 0..4 of the feature block of the gradient row
                float gfe[5];
#pragma unroll
                for (int c = 0; c < 5; c++) gfe[c] = live ? gr[MRGS_G_FEAT + c] : 0.0f;
                float d_ori[3];
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const float so = glue_sigmoid(gl.ori_color_raw[3 * (size_t)idx + c]);
                    d_ori[c] = gfe[2 + c] * so * (1.0f - so);
                }
                wave_store_rows<3>(tile, d_ori, gl.d_ori_color + 3 * r0, nrows, lane);
                const float zero3[3] = {0.0f, 0.0f, 0.0f};
                wave_store_rows<3>(tile, zero3, gl.d_indirect_dc + 3 * r0, nrows, lane);
                {   // the 45 higher indirect coefficients of the wave's rows: one run of nrows * 45 floats
                    float* dst = gl.d_indirect_rest + r0 * 45;
                    if (nrows == 64 && (((uintptr_t)dst) & 15u) == 0) {
                        float4* d4 = reinterpret_cast<float4*>(dst);
#pragma unroll
                        for (int k = 0; k * 64 < 16 * 45; k++)
                            if (k * 64 + lane < 16 * 45) d4[k * 64 + lane] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    } else {
                        for (int e = lane; e < nrows * 45; e += 64) dst[e] = 0.0f;
                    }
                }
                if (in_range) {
                    const float s0 = glue_sigmoid(gl.refl_raw[idx]), s1 = glue_sigmoid(gl.rough_raw[idx]), so = glue_sigmoid(gl.opacity_raw[idx]);
                    gl.d_refl[idx] = gfe[0] * s0 * (1.0f - s0);
                    gl.d_rough[idx] = gfe[1] * s1 * (1.0f - s1);
                    gl.d_opacity[idx] = dop * so * (1.0f - so);
                }
            }
        }
    }
    if (!in_range || GLUE) return;
    if ((S & 3) == 0 && (((uintptr_t)dL_dfeatures) & 15u) == 0) {
        // feature gradients as 16-byte pieces (the row's feature part starts 16-byte aligned: MRGS_G_FEAT = 16 floats, stride % 4 == 0):
        // a 4-byte access per channel touched 64 cache lines per instruction, S times each way
        const float4* g4 = reinterpret_cast<const float4*>(gr + MRGS_G_FEAT);
        float4* o4 = reinterpret_cast<float4*>(dL_dfeatures + (size_t)idx * S);
        for (int c = 0; c < S / 4; c++) o4[c] = live ? g4[c] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    } else {
        for (int c = 0; c < S; c++) dL_dfeatures[(size_t)idx * S + c] = live ? gr[MRGS_G_FEAT + c] : 0.0f;
    }
    dL_dopacity[idx] = dop;
}

void mrgs_launch_preprocess_bwd(const MrgsRasterConfig& cfg, const MrgsRasterInputs& in, const MrgsGeomWs& g,
                                const int32_t* radii, const float* grad_rec, const MrgsRasterGrads& out, hipStream_t stream)
{
    const dim3 grid((cfg.P + 64 * MRGS_PREB_WAVES - 1) / (64 * MRGS_PREB_WAVES)), block(64 * MRGS_PREB_WAVES);
#define PREB_ARGS cfg.P, cfg.D, cfg.M, cfg.S, cfg.W, cfg.H, cfg.tanfovx, cfg.tanfovy, in.means3D, in.scales, in.rotations, in.shs, in.transMat_precomp, \
                  in.viewmatrix, in.projmatrix, in.campos, radii, g.clamped, g.rec, grad_rec, MRGS_GRAD_STRIDE(cfg.S), \
                  out.dL_dmeans2D, out.dL_dcolors, out.dL_dfeatures, out.dL_dopacity, out.dL_dmeans3D, out.dL_dtransMat, \
                  out.dL_dsh, out.dL_dscales, out.dL_drotations, in.shs_rest, out.dL_dsh_rest
    if (out.glue_params != nullptr) {
        const MrgsSurfelParams& p = *out.glue_params;
        const MrgsSurfelGrads& o = *out.glue_grads;
        const GlueBwd gl = {p.rotation_raw, p.opacity_raw, p.scaling_raw, p.refl_raw, p.rough_raw, p.ori_color_raw,
                            o.d_xyz, o.d_scaling, o.d_rotation, o.d_opacity, o.d_refl, o.d_rough, o.d_ori_color, o.d_indirect_dc, o.d_indirect_rest,
                            p.viewmatrix};
        hipLaunchKernelGGL(preprocess_bwd_kernel<true>, grid, block, 0, stream, PREB_ARGS, gl);
    } else {
        hipLaunchKernelGGL(preprocess_bwd_kernel<false>, grid, block, 0, stream, PREB_ARGS, GlueBwd{});
    }
#undef PREB_ARGS
}

// dL/dRGB of every surfel as the SH backward consumes it (backward.cu:33-36: zero where the forward clamped the channel; zero for culled
// surfels), straight from the gradient rows the blend backward has just finished: the factor a view-parallel step all-gathers
// (dL/dsh_v[p][k][c] = B_k(dir_v(p)) dRGB_v[p][c], materialrefgs_amd/dist.py) exists before the per-gaussian backward runs.
__global__ void __launch_bounds__(256) color_grad_extract_kernel(int P, const int32_t* __restrict__ radii, const uint8_t* __restrict__ clamped,
                                                                 const float* __restrict__ grad_rec, int gstride, int from_sh, float* __restrict__ out)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P) return;
    float v[3] = {0.0f, 0.0f, 0.0f};
    if (radii[idx] > 0) {
        const float* gr = grad_rec + (size_t)idx * gstride + MRGS_G_COL;
        const uint32_t cl = from_sh ? clamped[idx] : 0u;
#pragma unroll
        for (int c = 0; c < 3; c++) v[c] = ((cl >> c) & 1u) ? 0.0f : gr[c];
    }
#pragma unroll
    for (int c = 0; c < 3; c++) out[3 * (size_t)idx + c] = v[c];
}

void mrgs_launch_color_grad_extract(const MrgsRasterConfig& cfg, const MrgsGeomWs& g, const int32_t* radii, const float* grad_rec, bool from_sh,
                                    float* out, hipStream_t stream)
{
    hipLaunchKernelGGL(color_grad_extract_kernel, dim3((cfg.P + 255) / 256), dim3(256), 0, stream, cfg.P, radii, g.clamped, grad_rec,
                       MRGS_GRAD_STRIDE(cfg.S), from_sh ? 1 : 0, out);
}

// checkFrustum, rasterizer_impl.cu:56-68
__global__ void __launch_bounds__(256) mark_visible_kernel(int P, const float* __restrict__ means3D, const float* __restrict__ viewmatrix,
                                                           uint8_t* __restrict__ present)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P) return;
    const float z = viewmatrix[2] * means3D[3 * (size_t)idx] + viewmatrix[6] * means3D[3 * (size_t)idx + 1] +
                    viewmatrix[10] * means3D[3 * (size_t)idx + 2] + viewmatrix[14];
    present[idx] = !(z <= 0.2f);
}

void mrgs_launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present, hipStream_t stream)
{
    hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, means3D, viewmatrix, present);
}
