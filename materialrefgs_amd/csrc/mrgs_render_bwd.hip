// mrgs_render_bwd.hip -- back-to-front replay and gradient accumulation on gfx950.
// Replaces BACKWARD::render / renderCUDA (backward.cu:145-468): same per-pixel arithmetic, same list order.
//
// Decomposition (see mrgs_render_fwd.hip for the forward twin): one wave64 per 8x8 pixel block, no barriers;
// the wave starts at the block's deepest contributor (wave-max of the forward's last_contributor) instead of
// the end of the tile list, consumes the list 64 entries at a time and walks only the entries whose
// conservative screen box can touch the block (__ballot sub-list, highest bit first).
//
// Gradient accumulation: the reference issues 16+S global fp32 atomicAdds per contributing (pixel, surfel)
// pair (backward.cu:350-465).  Here the K = 18+S per-lane terms of a surfel are reduced across the wave with
// a transposing butterfly -- v_permlane32_swap / v_permlane16_swap halve the register count while they halve
// the lane span, then four DPP row rotates finish inside 16-lane rows (2.5 K instructions instead of 6 K for
// K independent wave reductions) -- and K/4 atomic instructions, each with four lanes writing four different
// floats of the surfel's packed gradient row, replace K single-lane atomics.
#include "mrgs_blend_math.h"

#ifndef MRGS_BWD_STAGES
#define MRGS_BWD_STAGES 1   // see MRGS_FWD_STAGES in mrgs_render_fwd.hip
#endif


__device__ __forceinline__ void swap32_add(float& a, float b)   // a <- [a.lo + a.hi | b.lo + b.hi]
{
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r.x) + __uint_as_float(r.y);
}
__device__ __forceinline__ void swap16_add(float& a, float b)   // rows: [a.r0+a.r1, b.r0+b.r1, a.r2+a.r3, b.r2+b.r3]
{
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r.x) + __uint_as_float(r.y);
}

// ---- wave reduction of the per-lane gradient terms -------------------------------------------------------
// K (multiple of 4) values per lane.  Two transposing swap stages (swap32_add, swap16_add) leave K/4 registers, each
// holding in its rows 0..3 the 16-lane partial sums of values 4i, 4i+2, 4i+1, 4i+3.  The remaining 16 -> 1 reduction
// inside the rows keeps transposing for two more steps: a DPP add whose bank_mask writes only half of the lanes
// merges two registers into one while it halves the lane span (row_ror:8 into banks {0,1} | {2,3}, then
// row_half_mirror into banks {0,2} | {1,3}), so that four registers (16 values) end up in ONE register in which
// every quad of lanes holds the four partials of one value; two quad_perm adds finish.  8 DPP adds for 16 values
// (the plain row reduction needs 16) and, more important, ONE atomic instruction whose 16 active lanes write 16
// consecutive floats of the surfel's gradient row.
//
// The DPP steps are written as single asm blocks: the masked-write form (old lanes preserved) cannot be expressed
// through __builtin_amdgcn_update_dpp + add, and inside a block the VALU-write -> DPP-read hazard (2 wait states)
// is covered by the instruction order plus explicit s_nop (the compiler does not see hazards inside inline asm;
// the leading s_nop covers the instruction that produced the inputs).
__device__ __forceinline__ float rows_reduce4(float a0, float a1, float a2, float a3)
{   // quad q = 4*row + bank of the result holds value 8*(bank&1) + 4*(bank>>1) + sub(row) of the 16
    float t0, t1, u;
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %1, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %2, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        : "=&v"(t0), "=&v"(t1), "=&v"(u)
        : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
    return u;
}
__device__ __forceinline__ float rows_reduce2(float a0, float a1)
{   // banks {0,1} of the result hold value 0 + sub(row), banks {2,3} value 4 + sub(row) of the 8
    float t;
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        : "=&v"(t)
        : "v"(a0), "v"(a1));
    return t;
}
__device__ __forceinline__ float rows_reduce1(float v)   // every lane of a 16-lane row gets the row total
{
    v = mrgs_dpp_add<0x128, 0xf>(v);   // row_ror:8
    v = mrgs_dpp_add<0x124, 0xf>(v);   // row_ror:4
    v = mrgs_dpp_add<0x122, 0xf>(v);   // row_ror:2
    v = mrgs_dpp_add<0x121, 0xf>(v);   // row_ror:1
    return v;
}

// lane -> (float offset inside a 16-value group, writer flags) for the three block shapes above
struct ReduceLane { uint32_t off16, off8, off4; bool w16, w8, w4; };   // offsets in bytes
__device__ __forceinline__ ReduceLane mrgs_reduce_lane(int lane)
{
    const int row = lane >> 4, bank = (lane >> 2) & 3;
    const int sub = ((row & 1) << 1) | (row >> 1);                  // rows 0,1,2,3 -> values +0,+2,+1,+3
    ReduceLane r;
    r.off16 = 4u * (uint32_t)(8 * (bank & 1) + 4 * (bank >> 1) + sub);
    r.off8 = 4u * (uint32_t)(4 * (bank >> 1) + sub);
    r.off4 = 4u * (uint32_t)sub;
    r.w16 = (lane & 3) == 0;
    r.w8 = (lane & 7) == 0;
    r.w4 = (lane & 15) == 0;
    return r;
}

// the per-lane address is formed as (wave-uniform base pointer) + (32-bit byte offset): one VALU add per atomic
__device__ __forceinline__ void atomic_add_at(float* __restrict__ base, uint32_t byte_off, float v)
{
    atomicAdd((float*)((char*)base + byte_off), v);
}

// Reduce K per-lane values over the wave and add them to the K floats at base + row_off bytes (the surfel's gradient row).
template <int K>
__device__ __forceinline__ void wave_reduce_atomic_add(float (&v)[K], float* __restrict__ base, uint32_t row_off, const ReduceLane& rl)
{
    static_assert(K % 4 == 0, "pad the value count to a multiple of 4");
#pragma unroll
    for (int i = 0; i < K / 2; i++) swap32_add(v[2 * i], v[2 * i + 1]);      // result in v[2i]
#pragma unroll
    for (int i = 0; i < K / 4; i++) swap16_add(v[4 * i], v[4 * i + 2]);      // result in v[4i]
    constexpr int N4 = K / 4;
    constexpr int NB4 = N4 / 4;                 // blocks of four registers
    constexpr int REM = N4 - 4 * NB4;           // 0..3 registers left
#pragma unroll
    for (int t = 0; t < NB4; t++) {
        const float u = rows_reduce4(v[16 * t], v[16 * t + 4], v[16 * t + 8], v[16 * t + 12]);
        if (rl.w16) atomic_add_at(base, row_off + (64u * t + rl.off16), u);
    }
    if (REM >= 2) {
        const float u = rows_reduce2(v[16 * NB4], v[16 * NB4 + 4]);
        if (rl.w8) atomic_add_at(base, row_off + (64u * NB4 + rl.off8), u);
    }
    if (REM == 1 || REM == 3) {
        constexpr int vbase = 16 * NB4 + (REM == 3 ? 8 : 0);
        const float u = rows_reduce1(v[vbase]);
        if (rl.w4) atomic_add_at(base, row_off + (4u * vbase + rl.off4), u);
    }
}

// the dL/dmean2D pair of the low-pass-filter branch (backward.cu:407-413) is non-zero for few pairs: reduced apart,
// and only when some lane of the wave took that branch
__device__ __forceinline__ void wave_reduce_atomic_add2(float a, float b, float* __restrict__ base, uint32_t byte_off, int lane)
{
    swap32_add(a, b);                       // lanes 0..31: partials of a, lanes 32..63: partials of b
    a = rows_reduce1(a);
    // row_bcast:15 -- lane 15 of each row is added into the next row; rows 1 and 3 end up with the totals
    a += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x142, 0xa, 0xf, false));
    if ((lane & 31) == 16) atomic_add_at(base, byte_off + 4u * (uint32_t)(lane >> 5), a);
}

// one value apart (the ninth live feature channel of rows padded to twelve: 16 + 9 values are 24 through the transposing reduction and this one)
__device__ __forceinline__ void wave_reduce_atomic_add1(float a, float* __restrict__ base, uint32_t byte_off, int lane)
{
    float z = 0.0f;
    swap32_add(a, z);                       // lanes 0..31: a[l] + a[l + 32]
    a = rows_reduce1(a);
    a += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x142, 0xa, 0xf, false));      // row_bcast:15: row 1 holds the total
    if (lane == 16) atomic_add_at(base, byte_off, a);
}

#ifdef MRGS_WAVE_STATS   // developer build only (tools/wave_stats.py): per-wave start/end time, iteration counts, placement
__device__ unsigned long long g_wave_stats[8 * 65536];
extern "C" int mrgs_wave_stats(unsigned long long* host, int n)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wave_stats), sizeof(unsigned long long) * n);
}
#define WS_ENTRY() const unsigned long long ws_te = wall_clock64(); unsigned long long ws_dbg[3];
#define WS_DBG , ws_dbg
#define WS_BEGIN() const unsigned long long ws_t0 = wall_clock64(), ws_c0 = __builtin_amdgcn_s_memtime(); unsigned ws_iters = 0, ws_act = 0, ws_chunks = 0;
#define WS_ITER(a) { ws_iters++; ws_act += (a) ? 1 : 0; }
#define WS_CHUNK() ws_chunks++;
#define WS_END() if (lane == 0 && b < 65536) { unsigned long long* w = g_wave_stats + 8 * (size_t)b; w[6] = ((ws_dbg[0] - ws_te) << 40) | ((ws_dbg[2] - ws_te) << 16) | ws_dbg[1]; w[7] = ws_te; w[0] = ws_t0; w[1] = wall_clock64(); \
        w[2] = __builtin_amdgcn_s_memtime() - ws_c0; w[3] = ((unsigned long long)ws_iters << 32) | ws_act; w[4] = ((unsigned long long)ws_chunks << 32) | (unsigned)max_contrib; \
        w[5] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32); }
#else
#define WS_ENTRY()
#define WS_DBG
#define WS_BEGIN()
#define WS_ITER(a)
#define WS_CHUNK()
#define WS_END()
#endif

#ifndef MRGS_BWD_WPE0
#define MRGS_BWD_WPE0 5
#endif
#ifndef MRGS_BWD_WPE8
#define MRGS_BWD_WPE8 4
#endif
// S_LIVE: the leading channels that can be non-zero (MrgsRasterInputs::features_live); the padding channels stay out of the entry's
// arithmetic and of the reduction, their gradient columns keep the zeros they were cleared to
template <int S_MAX, bool FV, int S_LIVE>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(S_MAX == 0 ? MRGS_BWD_WPE0 : S_MAX <= 8 ? MRGS_BWD_WPE8 : 2, 8))) render_bwd_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ bwd_assign, uint32_t* __restrict__ q_bwd, const uint32_t* __restrict__ cu_state, const uint32_t* __restrict__ point_list,
    const uint8_t* __restrict__ cflag, int S, int W, int H, int tiles_x, int ntiles,
    const float4* __restrict__ rec, const float* __restrict__ features, const float* __restrict__ bg,
    const float* __restrict__ final_Ts, const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpixels,
    const float* __restrict__ dL_dpixels_f, const float* __restrict__ dL_dothers, float* __restrict__ grad_rec, int gstride, int slots)
{
    constexpr int SF = S_MAX > 0 ? S_MAX : 1;
    constexpr bool XREC = S_LIVE > S_MAX;                       // the ninth live channel rides in the surfel record (see render_fwd_kernel)
    static_assert(!XREC || (S_MAX == 8 && S_LIVE == 9 && FV), "one record channel: rows of twelve floats, eight staged");
    constexpr int S_ROW = XREC ? 12 : S_MAX;                    // floats per feature row = width of the gradient row's feature block
    constexpr int S_STAGED = S_LIVE < S_MAX ? S_LIVE : S_MAX;
    constexpr int SFA = S_LIVE > SF ? S_LIVE : SF;
    constexpr int K = 16 + S_LIVE;  // gradient values of an entry
    constexpr int KT = K & ~3;      // ... through the transposing reduction (whole groups of four; the dL/dmean2D pair and a remainder go apart)
    __shared__ StageBuf<SF> stage[MRGS_BWD_STAGES];

    const int lane = threadIdx.x;
    const int b = blockIdx.x;
    WS_ENTRY();
    const uint32_t item = mrgs_pull_item(q_bwd, cu_state, bwd_assign, ntiles, b & 7, b >> 3, lane, slots WS_DBG);
    if (item == 0xFFFFFFFFu) return;
    const int tile = (int)((item & 0x1FFFFFFFu) >> 2), quad = (int)(item & 3u);
    const uint32_t prio = (item >> 29) & 3u;
    const uint2 range = ranges[tile];
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int bx = tx * 2 + (quad & 1), by = ty * 2 + (quad >> 1);
    const int pxi = bx * 8 + (lane & 7), pyi = by * 8 + (lane >> 3);
    const bool inside = pxi < W && pyi < H;
    const float px = (float)pxi, py = (float)pyi;
    const int HW = H * W;
    const int pix = inside ? W * pyi + pxi : 0;

    const int last_contributor = inside ? (int)n_contrib[pix] : 0;
    // deepest list position any pixel of this block blended: nothing behind it can receive a gradient
    int max_contrib = last_contributor;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) max_contrib = max(max_contrib, __shfl_xor(max_contrib, d, 64));
    if (max_contrib == 0) return;
    WS_BEGIN();
    // heavy items first in line for issue slots (priority class chosen by bwd_order_kernel)
    if (prio == 3u) __builtin_amdgcn_s_setprio(3);
    else if (prio == 2u) __builtin_amdgcn_s_setprio(2);
    else if (prio == 1u) __builtin_amdgcn_s_setprio(1);
    const int median_contributor = inside ? (int)n_contrib[pix + HW] : 0;

    const float T_final = inside ? final_Ts[pix] : 0.f;
    float T = T_final;
    // accum_* hold the reference's accum_rec recurrences (backward.cu:340-372) WITH the term of the entry processed last already
    // folded in: the reference keeps last_alpha / last_color / last_depth / last_normal / last_feature and folds them at the start of
    // the next entry; folding right after an entry's last use of accum_* is the same expression on the same operands, evaluated
    // earlier -- bit-identical -- and frees 8 + S registers across the gradient reduction, which is where the kernel's register
    // peak sits: 104 -> 96 VGPRs (5 waves per SIMD instead of 4) without feature channels, 140 -> 128 (4 instead of 3) with 8.
    // accum_dot: the reference's accum_rec recurrences (colour, feature and normal channels, depth, accumulated alpha: backward.cu:340-372;
    // last_dL_dT: :436-437) CONTRACTED with the pixel's upstream gradients.  dL/dalpha needs sum_ch (c_ch - accum_rec_ch) dL_dpixel_ch = q - A
    // with q = sum_ch c_ch dL_dpixel_ch of this entry and A = sum_ch accum_rec_ch dL_dpixel_ch, and A obeys the same recurrence as every
    // accum_rec_ch (A' = alpha q + (1 - alpha) A): ONE scalar recurrence instead of 3 + S + 3 + 3 of them -- 4 instructions per channel
    // become 2 and S + 8 registers go.
    // Not the reference's summation order: the difference is rounding of the two sums (measured against the float64 evaluation of the
    // reference's formulas in tests/test_truth_leg.py and in every default bench line, next to the literal fp32 reading).
    float dL_dpixel[3] = {0.f, 0.f, 0.f};
    float dL_dpixel_f[SFA];
#pragma unroll
    for (int i = 0; i < SFA; i++) dL_dpixel_f[i] = 0.f;
    float accum_dot = 0.f;
    float dL_dreg = 0.f, dL_ddepth = 0.f, dL_daccum = 0.f, dL_dnormal2D[3] = {0.f, 0.f, 0.f}, dL_dmedian_depth = 0.f;
    // A pixel nothing was blended into takes no part in any sum; its upstream gradients are not even read (they may
    // hold non-finite values, e.g. from a division by the zero accumulated alpha, and the entry body below multiplies
    // the gradients of non-contributing lanes by exact zeros instead of branching around them).
    if (inside && last_contributor > 0) {
        dL_ddepth = dL_dothers[0 * HW + pix];
        dL_daccum = dL_dothers[1 * HW + pix];
        dL_dnormal2D[0] = dL_dothers[2 * HW + pix];
        dL_dnormal2D[1] = dL_dothers[3 * HW + pix];
        dL_dnormal2D[2] = dL_dothers[4 * HW + pix];
        dL_dmedian_depth = dL_dothers[5 * HW + pix];
        dL_dreg = dL_dothers[6 * HW + pix];
#pragma unroll
        for (int i = 0; i < 3; i++) dL_dpixel[i] = dL_dpixels[i * HW + pix];
        if (S_MAX > 0) {
#pragma unroll
            for (int i = 0; i < S_LIVE; i++)
                if (i < S) dL_dpixel_f[i] = dL_dpixels_f[(size_t)i * HW + pix];
        }
    }
    const float final_D = inside ? final_Ts[pix + HW] : 0.f;
    const float final_D2 = inside ? final_Ts[pix + 2 * HW] : 0.f;
    const float mscale = MRGS_FAR_N / (MRGS_FAR_N - MRGS_NEAR_N);
    const float dmd_scale = (MRGS_FAR_N * MRGS_NEAR_N) / (MRGS_FAR_N - MRGS_NEAR_N);
    const float bg_dot_dpixel = fmaf(bg[2], dL_dpixel[2], fmaf(bg[1], dL_dpixel[1], bg[0] * dL_dpixel[0]));

    // chunks of 64 list entries, from the one holding position max_contrib-1 down to chunk 0; lane l <-> position 64c+l.
    // Same staging pipeline as the forward (LDS-DMA, cull conics two chunks and ids three chunks ahead), walked
    // towards the front of the list.  The id slot of a staged entry holds the BYTE offset of the surfel's gradient row.
    const uint32_t* plist = point_list + range.x;
    // the forward's flag of this quadrant for every list entry: 1 = some pixel of the block blended it.  Only those are walked -- an
    // entry no pixel blended has `active` false in every lane and would leave after its intersection test
    const uint8_t* qm = cflag + (size_t)range.x * 4 + quad;
    const ReduceLane rl = mrgs_reduce_lane(lane);
    const uint32_t row_bytes = (uint32_t)gstride * 4u;
    const int c_top = (max_contrib - 1) / MRGS_CHUNK;
    // prefetched (surfel id, cull bits) of the two chunks ahead, one register each: the four cull bits ride in bits 28-31
    // (P < 2^26: the gradient rows are addressed with 32-bit byte offsets)
    uint32_t idq1 = 0, idq2 = 0;
    uint64_t mask_cur, exact_cur;            // entries of the current chunk to walk / to evaluate with the oracle's arithmetic (flag bit 1)
    {
        uint32_t id0 = 0, q0 = 0;
        if (c_top * MRGS_CHUNK + lane < max_contrib) { id0 = plist[c_top * MRGS_CHUNK + lane]; q0 = qm[(size_t)(c_top * MRGS_CHUNK + lane) * 4]; }
        if (c_top >= 1) idq1 = plist[(c_top - 1) * MRGS_CHUNK + lane] | ((uint32_t)qm[(size_t)((c_top - 1) * MRGS_CHUNK + lane) * 4] << 28);
        if (c_top >= 2) idq2 = plist[(c_top - 2) * MRGS_CHUNK + lane] | ((uint32_t)qm[(size_t)((c_top - 2) * MRGS_CHUNK + lane) * 4] << 28);
        const bool cand0 = q0 & 1u;
        mask_cur = __builtin_amdgcn_ballot_w64(cand0);
        exact_cur = __builtin_amdgcn_ballot_w64((q0 & 2u) != 0u);
        mrgs_stage_async<S_MAX, SF, FV, S_LIVE, S_ROW>(stage[c_top % MRGS_BWD_STAGES], rec, features, S, id0, cand0);
        if (cand0) stage[c_top % MRGS_BWD_STAGES].id[lane] = id0 * row_bytes;
    }

    for (int c = c_top; c >= 0; c--) {
        const int base = c * MRGS_CHUNK;
        mrgs_stage_wait();                    // chunk c has landed
        WS_CHUNK();
        uint64_t mask_nxt = 0ull, exact_nxt = 0ull;
        auto stage_next = [&]() {
            const bool cand1 = (idq1 >> 28) & 1u;
            exact_nxt = __builtin_amdgcn_ballot_w64(((idq1 >> 29) & 1u) != 0u);
            const uint32_t id1 = idq1 & 0x0FFFFFFFu;
            mask_nxt = __builtin_amdgcn_ballot_w64(cand1);
            mrgs_stage_async<S_MAX, SF, FV, S_LIVE, S_ROW>(stage[(c + 1) % MRGS_BWD_STAGES], rec, features, S, id1, cand1);
            if (cand1) stage[(c + 1) % MRGS_BWD_STAGES].id[lane] = id1 * row_bytes;
            idq1 = idq2;
            idq2 = 0;
            if (c >= 3) idq2 = plist[(c - 3) * MRGS_CHUNK + lane] | ((uint32_t)qm[(size_t)((c - 3) * MRGS_CHUNK + lane) * 4] << 28);
        };
        if (MRGS_BWD_STAGES == 2) stage_next();

        uint64_t mask = mask_cur;
        const uint64_t exact_mask = exact_cur;
        const StageBuf<SF>& sb = stage[c % MRGS_BWD_STAGES];

        // One list entry.  The body is branch-free across lanes: a lane that does not contribute (no hit, outside the
        // image, or behind the pixel's last contributor) runs it with alpha = 0, which leaves every running quantity
        // bit-identical (T * rcp(1) = T; the accum_* recurrences absorb the pending term now and add 0 * x next time),
        // and with the operands that could be non-finite for it (G, depth, 1/p.z, s) replaced, so that all of its
        // gradient terms are exact zeros.  The 3D / low-pass-filter choice (backward.cu:380-413) is a select as well.
        auto blend_entry = [&](const SurfelGeom& sg, int j) {
            const int contributor = base + j;           // 0-based list position; the forward's contributor is position+1
            Hit h;
            bool active;
            // The forward took this pair's decision exactly (mrgs_blend_math.h "Exact decisions") and flagged the entries where the
            // fast values could not tell for some pixel of the block (bit 1 of the entry's flag byte, ~1e-6 of the pairs): those are
            // evaluated as the oracle does, by every lane; for all others the fast evaluation IS the exact decision, no band to look at.
            active = mrgs_intersect(sg, px, py, h) & !(h.depth < MRGS_NEAR_LO);
            if (__builtin_expect((exact_mask >> j) & 1ull, 0)) active = mrgs_intersect_exact<true>(sg, px, py, h);     // (falls through otherwise)
            active = active & inside & (contributor < last_contributor);
            const uint64_t amask = __builtin_amdgcn_ballot_w64(active);
            WS_ITER(amask != 0ull);
            if (amask == 0ull) return;
            const bool use3d = h.use3d;
            const bool a3 = active & use3d;
            const float4 a0 = sb.rec[3][j], a1 = sb.rec[4][j];
            const float normal[3] = {a0.x, a0.y, a0.z};
            const float col[3] = {a0.w, a1.x, a1.y};
            const float alpha = active ? h.alpha : 0.0f;
            const float G = active ? h.G : 0.0f;
            const float c_d = active ? h.depth : 1.0f;
            const float sx = a3 ? h.sx : 0.0f, sy = a3 ? h.sy : 0.0f;
            const float inv_pz = a3 ? h.inv_pz : 0.0f;

            float g[K];
            const float inv_1ma = mrgs_rcp(1.0f - alpha);
            T = T * inv_1ma;                                   // backward.cu:330
            const float w = alpha * T;
            const float one_m_a = 1.0f - alpha;                // the next entry's (1 - last_alpha)
            // q = sum over every blended quantity of (the entry's value) x (the pixel's upstream gradient): colours, features, normal ...
            float q = 0.0f;
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                q = fmaf(col[ch], dL_dpixel[ch], q);
                g[MRGS_G_COL + ch] = w * dL_dpixel[ch];
            }
            if (S_MAX > 0) {
#pragma unroll
                for (int ch = 0; ch < S_STAGED; ch++) {
                    // no branch on the runtime S (see the forward): slots beyond S read as 0 and have dL_dpixel_f = 0
                    const float f = (FV || ch < S) ? mrgs_staged_feature<FV>(sb, ch, j) : 0.0f;
                    q = fmaf(f, dL_dpixel_f[ch], q);
                    g[MRGS_G_FEAT + ch] = w * dL_dpixel_f[ch];
                }
                if (XREC) {
                    q = fmaf(a1.w, dL_dpixel_f[S_MAX], q);
                    g[MRGS_G_FEAT + S_MAX] = w * dL_dpixel_f[S_MAX];
                }
            }
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                q = fmaf(normal[ch], dL_dnormal2D[ch], q);
                g[MRGS_G_NRM + ch] = w * dL_dnormal2D[ch];
            }
            const float inv_cd = mrgs_rcp(c_d);
            const float m_d = mscale * (1.0f - MRGS_NEAR_N * inv_cd);
            const float dmd_dd = dmd_scale * inv_cd * inv_cd;
            float dL_dz = (active & (contributor == median_contributor - 1)) ? dL_dmedian_depth : 0.0f;
            const float final_A = 1.0f - T_final;              // recomputed per entry: one instruction for one register
            const float dL_dweight = fmaf(-2.0f * m_d, final_D, fmaf(m_d * m_d, final_A, final_D2)) * dL_dreg;
            const float dL_dmd = 2.0f * w * fmaf(m_d, final_A, -final_D) * dL_dreg;
            dL_dz = fmaf(dL_dmd, dmd_dd, dL_dz);
            // ... and of the three single-channel recurrences of the same form: the depth (value c_d, gradient dL_ddepth), the accumulated
            // alpha (value 1, gradient dL_daccum) and the distortion weight (value dL_dweight, "gradient" 1: last_dL_dT, backward.cu:436-437)
            q = fmaf(c_d, dL_ddepth, q);
            q += dL_daccum;
            q += dL_dweight;
            float dL_dalpha = q - accum_dot;
            accum_dot = fmaf(alpha, q, one_m_a * accum_dot);      // backward.cu:340-372, 436-437 (every recurrence at once), for the next entry
            dL_dalpha *= T;
            dL_dalpha = fmaf(-T_final * inv_1ma, bg_dot_dpixel, dL_dalpha);
            dL_dalpha = active ? dL_dalpha : 0.0f;
            const float dL_dG = sg.g2.w * dL_dalpha;
            dL_dz = fmaf(w, dL_ddepth, dL_dz);
            {   // ray/splat branch (zero for lanes in the low-pass branch: s = 0 and 1/p.z = 0 there)
                const float Twx = sg.g1.z, Twy = sg.g1.w;
                const float dGn = dL_dG * -G;
                const float dL_dsx = fmaf(dGn, sx, dL_dz * Twx);
                const float dL_dsy = fmaf(dGn, sy, dL_dz * Twy);
                const float dpx = dL_dsx * inv_pz, dpy = dL_dsy * inv_pz;
                const float dpz = -fmaf(dpx, sx, dpy * sy);
                // dL_dk = cross(l, dL_dp), dL_dl = cross(dL_dp, k); the rows of dL/dT need their negatives, which are
                // formed directly (fma(-a, b, c*d) is the exact negative of fma(a, b, -(c*d)))
                const float ndkx = fmaf(-h.ly, dpz, h.lz * dpy);
                const float ndky = fmaf(-h.lz, dpx, h.lx * dpz);
                const float ndkz = fmaf(-h.lx, dpy, h.ly * dpx);
                const float ndlx = fmaf(-dpy, h.kz, dpz * h.ky);
                const float ndly = fmaf(-dpz, h.kx, dpx * h.kz);
                const float ndlz = fmaf(-dpx, h.ky, dpy * h.kx);
                g[0] = ndkx; g[1] = ndky; g[2] = ndkz;
                g[3] = ndlx; g[4] = ndly; g[5] = ndlz;
                g[6] = fmaf(px, -ndkx, fmaf(py, -ndlx, dL_dz * sx));
                g[7] = fmaf(px, -ndky, fmaf(py, -ndly, dL_dz * sy));
                g[8] = fmaf(px, -ndkz, fmaf(py, -ndlz, dL_dz));
            }
            g[MRGS_G_OPA] = G * dL_dalpha;
            const uint32_t row_off = sb.id[j];   // byte offset of the surfel's gradient row
            if constexpr (KT == K) {
                wave_reduce_atomic_add<K>(g, grad_rec, row_off, rl);
            } else {
                float gt[KT];
#pragma unroll
                for (int k = 0; k < KT; k++) gt[k] = g[k];
                wave_reduce_atomic_add<KT>(gt, grad_rec, row_off, rl);
#pragma unroll
                for (int k = KT; k < K; k++) wave_reduce_atomic_add1(g[k], grad_rec, row_off + 4u * (uint32_t)k, lane);
            }
            const bool a2 = active & !use3d;
            if (__builtin_amdgcn_ballot_w64(a2) != 0ull) {   // low-pass-filter branch: dL/dmean2D
                const float dL_dG2 = a2 ? dL_dG : 0.0f;
                const float dGf = -G * MRGS_FILTER_INV_SQUARE;
                wave_reduce_atomic_add2(dL_dG2 * (dGf * h.dx), dL_dG2 * (dGf * h.dy), grad_rec, row_off + 4u * MRGS_G_M2(S_ROW), lane);
            }
        };

        if (mask != 0ull) {
            // back to front over the set bits; the geometry of the next entry is fetched from LDS while the current one
            // is processed, in two alternating register sets (no copies at the loop edge)
            int j = 63 - __builtin_clzll(mask);
            SurfelGeom sA, sB;
            sA.g0 = sb.rec[0][j]; sA.g1 = sb.rec[1][j]; sA.g2 = sb.rec[2][j];
            while (true) {
                mask &= ~(1ull << j);
                bool more = mask != 0ull;
                int jn = more ? 63 - __builtin_clzll(mask) : j;
                sB.g0 = sb.rec[0][jn]; sB.g1 = sb.rec[1][jn]; sB.g2 = sb.rec[2][jn];
                blend_entry(sA, j);
                if (!more) break;
                j = jn;
                mask &= ~(1ull << j);
                more = mask != 0ull;
                jn = more ? 63 - __builtin_clzll(mask) : j;
                sA.g0 = sb.rec[0][jn]; sA.g1 = sb.rec[1][jn]; sA.g2 = sb.rec[2][jn];
                blend_entry(sB, j);
                if (!more) break;
                j = jn;
            }
        }
        if (MRGS_BWD_STAGES == 1) stage_next();
        mask_cur = mask_nxt;
        exact_cur = exact_nxt;
    }
    mrgs_stage_wait();   // do not retire the wave with LDS-DMA still in flight
    WS_END();
}

#define MRGS_BWD_KERNEL render_bwd_kernel

void mrgs_launch_render_bwd(const MrgsRasterConfig& cfg, const MrgsRasterInputs& in, const MrgsGeomWs& g, const uint32_t* plist,
                            const uint8_t* cflag, const MrgsImgWs& img, const float* dL_dpix, const float* dL_dpix_f, const float* dL_dothers,
                            float* grad_rec, bool forward_queues, hipStream_t stream)
{
    // forward_queues: the forward set the backward's queue state up as a copy of its own (MrgsRasterInputs::bwd_grad_ws): same dealing,
    // read from the forward's assignment array
    const uint32_t* assign = forward_queues ? img.fwd_assign : img.bwd_assign;
    const int tiles_x = (cfg.W + MRGS_BLOCK_X - 1) / MRGS_BLOCK_X, tiles_y = (cfg.H + MRGS_BLOCK_Y - 1) / MRGS_BLOCK_Y;
    const int ntiles = tiles_x * tiles_y;
    // one wave per slot of the work queues (mrgs_pull_item): items rounded up to whole dealing passes; blockIdx % 8 = XCD list
    const int nblocks = (((ntiles + 7) / 8) * 4 + MRGS_MAX_SIMD_QUEUES) * 8;
    const dim3 grid(nblocks), block(64);
#define LAUNCH(GS, SM, ...)                                                                                                       \
    hipLaunchKernelGGL((MRGS_BWD_KERNEL<SM, __VA_ARGS__>), grid, block, 0, stream, img.ranges, assign, img.q_bwd, img.blend_state + MRGS_CS_BASE, plist, cflag, cfg.S, cfg.W, cfg.H, tiles_x, ntiles, \
                       g.rec, in.features, in.bg, img.final_T, img.n_contrib, dL_dpix, dL_dpix_f, dL_dothers, grad_rec, GS, mrgs_waves_per_simd<MRGS_BWD_KERNEL<SM, __VA_ARGS__>>())
    // the packed gradient row is as wide as the padded value count of the kernel instance (MRGS_GRAD_STRIDE)
    const int gs = MRGS_GRAD_STRIDE(cfg.S);
    const bool fv_ok = ((uintptr_t)in.features & 15u) == 0;   // 16-byte DMA pieces need an aligned feature tensor
    if (cfg.S == 0) LAUNCH(gs, 0, false, 0);
    else if (cfg.S == 8 && fv_ok) LAUNCH(gs, 8, true, 8);
    else if (cfg.S <= 8) LAUNCH(gs, 8, false, 8);
    else if (cfg.S == 12 && fv_ok && in.features_live == 9u) LAUNCH(gs, 8, true, 9);      // rows of 9 channels in 12 floats: 8 staged, the ninth in the surfel record
    else if (cfg.S == 12 && fv_ok) LAUNCH(gs, 12, true, 12);
    else if (cfg.S <= 12) LAUNCH(gs, 12, false, 12);
    else if (cfg.S == 24 && fv_ok) LAUNCH(gs, 24, true, 24);
    else LAUNCH(gs, 24, false, 24);
#undef LAUNCH
}
