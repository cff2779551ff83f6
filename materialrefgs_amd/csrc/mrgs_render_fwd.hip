// mrgs_render_fwd.hip -- front-to-back surfel blend on gfx950.  Replaces FORWARD::render / renderCUDA
// (forward.cu:272-463): same per-pixel arithmetic and the same list order, different decomposition.
//
// MI355X-first design (the reference runs one 256-thread block per 16x16 tile with two barriers per batch):
//   * one wave64 per 8x8 pixel block, each wave a workgroup of its own -> no barriers at all, 4x more (and 4x
//     smaller) work items to balance over 256 CUs, and the 64 lanes of a wave see nearly the same surfels;
//   * the tile's depth-sorted list is consumed 64 entries at a time: lane l tests entry l's conservative screen
//     box against the wave's block and a 64-bit __ballot gives the sub-list that can touch this block -- the
//     wave then walks only the set bits (s_ff1) instead of all 64 entries.  Skipped entries could not have
//     passed alpha >= 1/255 on any lane;
//   * the records of the surviving entries are gathered by LDS-DMA (global_load_lds, per-lane source address,
//     EXEC-masked to the candidates) into a double-buffered stage: the copy for chunk c+1 is in flight while
//     chunk c is blended and the staged data never passes through VGPRs.  Ids run three chunks ahead and cull
//     boxes two chunks ahead, so the dependent gather id -> box -> record is never waited for;
//   * in the blend loop the stage is read with uniform-address ds_read_b128 (LDS broadcast), the next surfel's
//     geometry being fetched while the current one is evaluated;
//   * blockIdx -> (tile, quadrant) keeps the four quadrant-waves of a tile on one XCD (blocks are dealt
//     round-robin to the 8 XCDs) so that the tile's records are fetched into one L2 only;
//   * the launch lasts as long as its longest wave takes ALONE on a SIMD: late in a block's list, when few of
//     its pixels are still alive, the block-level cull is evaluated again against the bounding box of the live pixels (below), and
//     the file is compiled with LLVM's max-ilp scheduling strategy (a lone wave pays ~8 cycles per dependent instruction).
#include <type_traits>

#include "mrgs_blend_math.h"

// MRGS_FWD_REFINE: the cull against the live pixels' bounding box, for chunks that start with at most MRGS_FWD_REFINE_LIVE live pixels
// (8 ... 40 measure the same, 64 = every chunk is 11 % slower than never).
#ifndef MRGS_FWD_REFINE
#define MRGS_FWD_REFINE 1
#endif
#ifndef MRGS_FWD_REFINE_LIVE
#define MRGS_FWD_REFINE_LIVE 24
#endif
// MRGS_FWD_STAGES 1: single stage buffer, the copy of chunk c+1 is issued after chunk c has been blended (5.5 KB LDS per wave -> 7 waves
//    per SIMD; the copy latency is covered by the other waves).  2: double buffer, copy overlapped inside the wave (11 KB).
#ifndef MRGS_FWD_STAGES
#define MRGS_FWD_STAGES 1
#endif

#ifdef MRGS_WAVE_STATS   // developer build only (tools/wave_stats.py fwd)
__device__ unsigned long long g_wave_stats_fwd[8 * 65536];
extern "C" int mrgs_wave_stats_fwd(unsigned long long* host, int n)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wave_stats_fwd), sizeof(unsigned long long) * n);
}
// timing experiment: only the items with at least this many list entries do their work (images are wrong, the duration of a
// heavy wave WITHOUT its neighbours is what gets measured)
__device__ int g_ws_min_total = 0;
extern "C" int mrgs_wave_stats_min_total(int n) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_ws_min_total), &n, sizeof(int)); }
#endif

// One marked pixel per wave, the lanes turned sideways: lane l evaluates list entry base + l against THAT pixel as the oracle does
// (IEEE quotient, correctly rounded exp, the reference's thresholds: mrgs_intersect_exact); the transmittance -- the one quantity whose
// rounding sequence decides anything -- is multiplied up serially in list order, one rounding per blended entry exactly as
// forward.cu:400-441 does; the sums (which decide nothing) are formed per lane and folded across the wave at the end.  Lane 0
// overwrites the pixel's outputs; entries the pixel blends are flagged for the backward like those of the main kernel.
// Called from the tail of the forward kernel (MRGS_FWD_REDO_INLINE, the default: the forward's occupancy is pinned, so what this code
// needs beyond the main loop's registers is spilled around it, in code that runs for one wave in a hundred) or from a launch of its own.
// inclusive prefix sum over the 64 lanes in DPP adds: Hillis-Steele inside the 16-lane rows (a lane whose source falls outside its row adds
// 0), then lane 15 / lane 31 of the rows before into the rows behind.  Every lane of the wave must be active.
__device__ __forceinline__ float wave_inclusive_sum(float v)
{
    v = mrgs_dpp_add<0x111, 0xf>(v);   // row_shr:1
    v = mrgs_dpp_add<0x112, 0xf>(v);   // row_shr:2
    v = mrgs_dpp_add<0x114, 0xf>(v);   // row_shr:4
    v = mrgs_dpp_add<0x118, 0xf>(v);   // row_shr:8
    v = mrgs_dpp_add<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
    v = mrgs_dpp_add<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3
    return v;
}

constexpr int MRGS_REDO_QCAP = 384, MRGS_REDO_REFILL = 4;      // queue entries; a refill scans 4 x 64 list entries with all their loads in flight
template <int S_MAX>
__device__ __forceinline__ void mrgs_redo_pixel(
    int pix, int lane, uint32_t* __restrict__ q_id, uint32_t* __restrict__ q_pos /* LDS, MRGS_REDO_QCAP words each */,
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, const uint8_t* __restrict__ qmask, uint8_t* cflag, int S, int W, int H,
    int tiles_x, const float4* __restrict__ rec, const float* __restrict__ features, const float* __restrict__ bg, float* __restrict__ final_T,
    uint32_t* __restrict__ n_contrib, float* __restrict__ out_color, float* __restrict__ out_feature, float* __restrict__ out_others)
{
    constexpr int SF = S_MAX > 0 ? S_MAX : 1;
    constexpr int QCAP = MRGS_REDO_QCAP, REFILL = MRGS_REDO_REFILL;
    (void)QCAP;
    const int HW = H * W;
    const float mscale = MRGS_FAR_N / (MRGS_FAR_N - MRGS_NEAR_N);
    const int pyi = pix / W, pxi = pix - pyi * W;
    const int tile = (pyi >> 4) * tiles_x + (pxi >> 4), quad = ((pyi >> 3) & 1) * 2 + ((pxi >> 3) & 1);
    const uint2 range = ranges[tile];
    const int total = (int)(range.y - range.x);
    const uint32_t* plist = point_list + range.x;
    const uint8_t* qm = qmask + range.x;
    const float qx = (float)pxi, qy = (float)pyi;
    float xT = 1.0f, xM1 = 0.f, xM2 = 0.f, xmed = 0.f;          // wave-uniform running values of the pixel
    uint32_t xlast = 0, xmedc = 0;
    float sC0 = 0.f, sC1 = 0.f, sC2 = 0.f, sN0 = 0.f, sN1 = 0.f, sN2 = 0.f, sD = 0.f, sDist = 0.f;   // this lane's share of the sums
    float sF[SF];
#pragma unroll
    for (int i = 0; i < SF; i++) sF[i] = 0.f;
    int nq = 0, qh = 0, next = 0;             // the queue holds nq candidates from q_*[qh] on
    bool ended = false, pre_ok = false;
    uint32_t pre_gid = 0;
    float4 p0 = make_float4(0, 0, 0, 0), p1 = p0, p2 = p0, p3 = p0, p4 = p0;
    while (!ended && (next < total || nq > 0)) {
        if (nq < MRGS_CHUNK && next < total) {
            // refill: what is left (< 64) moves to the front, then the candidates of the next 256 list entries are appended in order
            const uint32_t keep_id = q_id[qh + lane], keep_pos = q_pos[qh + lane];
            __builtin_amdgcn_wave_barrier();
            if (lane < nq) { q_id[lane] = keep_id; q_pos[lane] = keep_pos; }
            qh = 0;
            bool cand[REFILL];
            uint32_t id[REFILL];
#pragma unroll
            for (int k = 0; k < REFILL; k++) {
                const int e = next + k * MRGS_CHUNK + lane;
                cand[k] = e < total && ((qm[e < total ? e : 0] >> quad) & 1u);
                id[k] = plist[e < total ? e : 0];
            }
#pragma unroll
            for (int k = 0; k < REFILL; k++) {
                const uint64_t cm = __builtin_amdgcn_ballot_w64(cand[k]);
                if (cand[k]) {
                    const int at = nq + __builtin_amdgcn_mbcnt_hi((uint32_t)(cm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cm, 0u));
                    q_id[at] = id[k]; q_pos[at] = (uint32_t)(next + k * MRGS_CHUNK + lane);
                }
                nq += __builtin_popcountll(cm);
            }
            next += REFILL * MRGS_CHUNK;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (nq < MRGS_CHUNK && next < total) continue;       // sparse stretch of the list: scan on before a partial batch is evaluated
        }
        const int n = nq < MRGS_CHUNK ? nq : MRGS_CHUNK;
        const bool valid = lane < n;
        const uint32_t gid = valid ? q_id[qh + lane] : 0u;
        const int e = valid ? (int)q_pos[qh + lane] : 0;
        // this batch's records: fetched while the batch before was evaluated when the queue held them then already
        float4 r0, r1, r2, r3, r4;
        if (pre_gid == gid && pre_ok) { r0 = p0; r1 = p1; r2 = p2; r3 = p3; r4 = p4; }
        else { const float4* src = rec + (size_t)gid * MRGS_REC_F4; r0 = src[0]; r1 = src[1]; r2 = src[2]; r3 = src[3]; r4 = src[4]; }
        {   // ... and the next one's are requested now
            const int rest = nq - n;
            pre_ok = rest > 0;                                                    // (wave-uniform)
            pre_gid = lane < rest ? q_id[qh + n + lane] : 0u;
            if (pre_ok) { const float4* nx = rec + (size_t)pre_gid * MRGS_REC_F4; p0 = nx[0]; p1 = nx[1]; p2 = nx[2]; p3 = nx[3]; p4 = nx[4]; }
        }
        SurfelGeom sg;
        sg.g0 = r0; sg.g1 = r1; sg.g2 = r2;
        Hit h;
        const bool hit = mrgs_intersect_exact(sg, qx, qy, h) & valid;
        const uint64_t hm = __builtin_amdgcn_ballot_w64(hit);
        const float oma = 1.0f - h.alpha;
        // T in front of every hit of the batch: the serial product, one float multiplication per hit in list order -- formed by every
        // lane for itself (lane l multiplies the factors of lanes 0 .. l-1 in that order, 1.0 for a lane without a hit: the same
        // roundings as one running product), so that the dependent chain is 64 multiplications and nothing else
        // (only the lanes WITH a hit carry a factor other than 1.0, and a multiplication by 1.0 is exact: walking the set bits of the hit
        // mask gives the same roundings as walking all 63 lanes, in a sixth of the steps)
        const float fac = hit ? oma : 1.0f;
        float Tb = xT;
        for (uint64_t left = hm & 0x7FFFFFFFFFFFFFFFull; left != 0ull; left &= left - 1ull) {
            const int i = __builtin_ctzll(left);
            const float d = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(fac), i));
            Tb = Tb * (lane > i ? d : 1.0f);
        }
        const float run = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(Tb * fac), MRGS_CHUNK - 1));
        // the first hit that would take T below 1e-4 ends the pixel and is not blended (forward.cu:400-404); the products
        // formed behind it are not used
        const uint64_t tm = __builtin_amdgcn_ballot_w64(hit & (Tb * oma < MRGS_T_MIN));
        const uint64_t bm = tm != 0ull ? hm & ((1ull << __builtin_ctzll(tm)) - 1ull) : hm;
        const bool bl = (bm >> lane) & 1ull;
        if (tm != 0ull) xT = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(Tb), __builtin_ctzll(tm)));
        else xT = run;
        const float w = bl ? h.alpha * Tb : 0.0f;
        const float depth = bl ? h.depth : 1.0f;
        const float m_ = mscale * (1.0f - MRGS_NEAR_N * (1.0f / depth));
        const float mw = m_ * w, mmw = m_ * m_ * w;
        // M1, M2 in front of this lane's entry: carry of the batches before + exclusive prefix inside the batch
        const float i1 = wave_inclusive_sum(mw), i2 = wave_inclusive_sum(mmw);
        const float pM1 = xM1 + (i1 - mw), pM2 = xM2 + (i2 - mmw);
        xM1 += __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(i1), 63));
        xM2 += __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(i2), 63));
        sDist = fmaf(fmaf(-2.0f * m_, pM1, fmaf(m_ * m_, 1.0f - Tb, pM2)), w, sDist);
        sD = fmaf(depth, w, sD);
        const float4 a0 = r3, a1 = r4;
        sN0 = fmaf(a0.x, w, sN0); sN1 = fmaf(a0.y, w, sN1); sN2 = fmaf(a0.z, w, sN2);
        sC0 = fmaf(a0.w, w, sC0); sC1 = fmaf(a1.x, w, sC1); sC2 = fmaf(a1.y, w, sC2);
        if (S_MAX > 0 && __builtin_amdgcn_ballot_w64(bl) != 0ull) {
            const float* fsrc = features + (size_t)gid * S;
#pragma unroll
            for (int ch = 0; ch < S_MAX; ch++)
                if (ch < S) sF[ch] = fmaf(fsrc[ch], w, sF[ch]);
        }
        if (bm != 0ull) xlast = (uint32_t)__builtin_amdgcn_readlane(e, 63 - __builtin_clzll(bm)) + 1u;
        const uint64_t medm = __builtin_amdgcn_ballot_w64(bl & (Tb > 0.5f));          // forward.cu:417-420
        if (medm != 0ull) {
            const int jm = 63 - __builtin_clzll(medm);
            xmedc = (uint32_t)__builtin_amdgcn_readlane(e, jm) + 1u;
            xmed = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(h.depth), jm));
        }
        // flags for the backward: bit 0 = blended by some pixel, bit 1 = the FAST evaluation of this pair cannot be sure of the hit
        // (the main kernel flagged the entries it walked; this pixel may go further than its wave did).  Bits are only ever set here.
        if (valid) {
            Hit hf;
            bool amb;
            const bool mh = mrgs_intersect(sg, qx, qy, hf);
            (void)mrgs_hit_decide(hf, mh, amb);
            const bool in_reach = tm == 0ull || lane <= __builtin_ctzll(tm);
            const uint32_t fl = (bl ? 1u : 0u) | ((amb && in_reach) ? 3u : 0u);
            // (the four quadrant bytes of a list entry are one aligned word; marked pixels of one block may meet on a byte)
            if (fl) atomicOr(reinterpret_cast<uint32_t*>(cflag) + range.x + e, fl << (8 * quad));
        }
        ended = tm != 0ull;
        qh += n;
        nq -= n;
    }
    auto wave_sum = [&](float v) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
        return v;
    };
    sC0 = wave_sum(sC0); sC1 = wave_sum(sC1); sC2 = wave_sum(sC2);
    sN0 = wave_sum(sN0); sN1 = wave_sum(sN1); sN2 = wave_sum(sN2);
    sD = wave_sum(sD); sDist = wave_sum(sDist);
    if (S_MAX > 0) {
#pragma unroll
        for (int ch = 0; ch < S_MAX; ch++) sF[ch] = wave_sum(sF[ch]);
    }
    if (lane == 0) {
        final_T[pix] = xT;
        final_T[pix + HW] = xM1;
        final_T[pix + 2 * HW] = xM2;
        n_contrib[pix] = xlast;
        n_contrib[pix + HW] = xmedc;
        out_color[pix] = fmaf(xT, bg[0], sC0);
        out_color[pix + HW] = fmaf(xT, bg[1], sC1);
        out_color[pix + 2 * HW] = fmaf(xT, bg[2], sC2);
        if (S_MAX > 0) {
#pragma unroll
            for (int ch = 0; ch < S_MAX; ch++)
                if (ch < S) out_feature[(size_t)ch * HW + pix] = sF[ch];
        }
        out_others[pix + 0 * HW] = sD;
        out_others[pix + 1 * HW] = 1.0f - xT;
        out_others[pix + 2 * HW] = sN0;
        out_others[pix + 3 * HW] = sN1;
        out_others[pix + 4 * HW] = sN2;
        out_others[pix + 5 * HW] = xmed;
        out_others[pix + 6 * HW] = sDist;
    }
}

// The marked pixels of a launch, one per wave (see mrgs_redo_pixel).  The forward kernel renders its own marked pixels at the end of its
// wave by default (MRGS_FWD_REDO_INLINE): measured, the separate launch's duration -- the longest marked list, walked by a wave that is
// alone on its SIMD -- is fully exposed on the stream (26 us at C2), while inside the forward the same work extends a handful of waves
// of which few are among the last to finish.
template <int S_MAX>
__global__ void __launch_bounds__(64) render_fwd_redo_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, const uint8_t* __restrict__ qmask, uint8_t* cflag, int S, int W, int H,
    int tiles_x, const float4* __restrict__ rec, const float* __restrict__ features, const float* __restrict__ bg, float* __restrict__ final_T,
    uint32_t* __restrict__ n_contrib, float* __restrict__ out_color, float* __restrict__ out_feature, float* __restrict__ out_others,
    const uint32_t* __restrict__ redo_list)
{
    __shared__ uint32_t q_id[MRGS_REDO_QCAP], q_pos[MRGS_REDO_QCAP];
    const int lane = threadIdx.x;
    const uint32_t count = redo_list[0];
    for (uint32_t it = blockIdx.x; it < count; it += gridDim.x) {
        mrgs_redo_pixel<S_MAX>((int)redo_list[2 + it], lane, q_id, q_pos, ranges, point_list, qmask, cflag, S, W, H, tiles_x, rec, features, bg, final_T,
                               n_contrib, out_color, out_feature, out_others);
        __builtin_amdgcn_wave_barrier();        // the next pixel's queue starts empty: nothing of this one is read again
    }
}

#ifndef MRGS_FWD_REDO_INLINE
#define MRGS_FWD_REDO_INLINE 1
#endif
// S_LIVE: the leading channels that can be non-zero (MrgsRasterInputs::features_live); the rest of the S_MAX are padding of the row
template <int S_MAX, bool FV, int S_LIVE>
// Waves per SIMD the register allocator is held to.  Round 4 (after the exact-decision logic joined the loop), forward blend stage in ms at
// C2 / C3full: S = 0 at 7 / 6 / 5 / 4 waves 0.172 / 0.170 / 0.176 / 0.174 (72 / 80 / 95 / 96 VGPRs, 21 / 14 / 0 / 0 spilled); S = 8 at 6 / 5 / 4
// waves 0.203 / 0.204 / 0.195 (105 VGPRs and no spills at 4).
// Round 6 (with the running transmittance bound in the loop): S = 0 at 6 / 5 waves 153.1-154.4 / 154.7-156.4 us (80 VGPRs with 12 spilled /
// 94 with none): the same within the run-to-run spread -- 5, the instance without scratch.
#ifndef MRGS_FWD_WAVES_S0
#define MRGS_FWD_WAVES_S0 5
#endif
#ifndef MRGS_FWD_WAVES_S8
#define MRGS_FWD_WAVES_S8 4
#endif
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(S_MAX == 0 ? MRGS_FWD_WAVES_S0 : S_MAX <= 12 ? MRGS_FWD_WAVES_S8 : 4, 8))) render_fwd_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ fwd_assign, uint32_t* __restrict__ blend_state, const uint32_t* __restrict__ point_list,
    const uint8_t* __restrict__ qmask, uint8_t* __restrict__ cflag, int S, int W, int H, int tiles_x, int ntiles,
    const float4* __restrict__ rec, const float4* __restrict__ cull, const float* __restrict__ features, const float* __restrict__ bg, float* __restrict__ final_T,
    uint32_t* __restrict__ n_contrib, float* __restrict__ out_color, float* __restrict__ out_feature, float* __restrict__ out_others,
    uint32_t* __restrict__ item_work, const uint32_t* __restrict__ item_est /* read by the MRGS_WAVE_STATS build only */,
    uint32_t* __restrict__ work_hint, int slots, uint32_t* __restrict__ redo_list)
{
    constexpr int SF = S_MAX > 0 ? S_MAX : 1;
    // S_LIVE > S_MAX (the <8, true, 9> instance, "pgsr" rows of nine channels in twelve floats): eight channels are staged, the ninth rides
    // in the spare float of the surfel record (preprocess copies it there) -- the entry costs two 16-byte pieces and two LDS reads of
    // features, as with eight channels, instead of three
    constexpr bool XREC = S_LIVE > S_MAX;
    static_assert(!XREC || (S_MAX == 8 && S_LIVE == 9 && FV), "one record channel: rows of twelve floats, eight staged");
    constexpr int S_ROW = XREC ? 12 : S_MAX;                    // floats per feature row = channel maps the instance serves
    constexpr int S_STAGED = S_LIVE < S_MAX ? S_LIVE : S_MAX;   // channels accumulated out of the stage buffer
    using RecTail = typename std::conditional<XREC, float4, float2>::type;      // what an entry takes of the record's last 16 bytes
    __shared__ StageBuf<SF> stage[MRGS_FWD_STAGES];

    const int lane = threadIdx.x;
    // XCD-aware mapping: b % 8 selects the XCD; within an XCD consecutive blocks are the 4 quadrants of one tile
    const int b = blockIdx.x;
    const uint32_t item = mrgs_pull_item(blend_state + MRGS_QS_FWD, blend_state + MRGS_CS_BASE, fwd_assign, ntiles, b & 7, b >> 3, lane, slots);
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
    const int pix = W * pyi + pxi;

#ifdef MRGS_WAVE_STATS
    const int total = (int)(range.y - range.x) >= g_ws_min_total ? (int)(range.y - range.x) : 0;
#else
    const int total = (int)(range.y - range.x);
#endif
#ifdef MRGS_WAVE_STATS
    const unsigned long long ws_t0 = wall_clock64(), ws_c0 = __builtin_amdgcn_s_memtime();
    unsigned ws_blend = 0, ws_le4 = 0, ws_le8 = 0, ws_le16 = 0, ws_blend_le8 = 0;
#endif
    // heavy items first in line for issue slots (priority class chosen by blend_order_kernel)
    if (prio == 3u) __builtin_amdgcn_s_setprio(3);
    else if (prio == 2u) __builtin_amdgcn_s_setprio(2);
    else if (prio == 1u) __builtin_amdgcn_s_setprio(1);

    uint64_t done = ~__builtin_amdgcn_ballot_w64(inside);   // lanes (pixels) that take no more entries: a lane MASK in scalar registers
    uint64_t redo = 0ull;            // pixels (lanes) with a decision inside its error band
    int cf_end = 0;                  // list entries whose cflag this wave has written
    uint32_t work = 0;
    float T = 1.0f;
#if MRGS_T1_RUNNING
    float Terr = 0.0f;               // MRGS_T1_SLACK x the bound of |T - the transmittance exact arithmetic has at this point| (mrgs_blend_math.h)
#endif
    float C0 = 0.f, C1 = 0.f, C2 = 0.f, N0 = 0.f, N1 = 0.f, N2 = 0.f;
    // (the record channel's accumulator is a variable of its own, not F[S_MAX]: as a ninth array element it shifted the pairs the compiler
    //  forms for v_pk_fma_f32 by one -- (x, F0), (F1, F2) ... (F7, F8) -- and every blended entry paid eight register moves to line the
    //  staged features up with them: +15 % VALU instructions per launch, +17 us, measured with the SQ counters, round 5)
    float F[SF];
#pragma unroll
    for (int i = 0; i < SF; i++) F[i] = 0.f;
    float Fx = 0.f;
    float Dp = 0.f, M1 = 0.f, M2 = 0.f, distortion = 0.f, median_depth = 0.f;
    uint32_t last_contributor = 0, median_contributor = 0;
    const float mscale = MRGS_FAR_N / (MRGS_FAR_N - MRGS_NEAR_N);

    // ---- pipeline prologue: chunk 0 staged, ids and cull bits of chunks 1, 2 on their way ----------------------
    // (the block-level cull was evaluated once per entry and quadrant by tile_ranges_kernel: bit `quad` of qmask)
    const uint32_t* plist = point_list + range.x;
    const uint8_t* qm = qmask + range.x;
    uint8_t* cf = cflag + (size_t)range.x * 4 + quad;       // this quadrant's "blended by some pixel" flag of every list entry
    uint32_t id1 = 0, id2 = 0, q1 = 0, q2 = 0;
    uint32_t idc = 0, idn = 0;       // ids of the chunk being blended / of the one staged behind it (MRGS_FWD_REFINE)

    uint64_t mask_cur;
    {
        uint32_t id0 = 0, q0 = 0;
        if (lane < total) { id0 = plist[lane]; q0 = qm[lane]; }
        idc = id0;
        if (MRGS_CHUNK + lane < total) { id1 = plist[MRGS_CHUNK + lane]; q1 = qm[MRGS_CHUNK + lane]; }
        if (2 * MRGS_CHUNK + lane < total) { id2 = plist[2 * MRGS_CHUNK + lane]; q2 = qm[2 * MRGS_CHUNK + lane]; }
        const bool cand0 = (q0 >> quad) & 1u;
        mask_cur = __builtin_amdgcn_ballot_w64(cand0);
        mrgs_stage_async<S_MAX, SF, FV, S_LIVE, S_ROW>(stage[0], rec, features, S, id0, cand0);
    }

    for (int base = 0, c = 0; base < total; base += MRGS_CHUNK, c++) {
        if (~done == 0ull) break;             // every pixel of the block has terminated (forward.cu:342-344, per wave)
        mrgs_stage_wait();                    // chunk c has landed
        uint64_t mask_nxt = 0ull;
        auto stage_next = [&]() {
            // stage chunk c+1 (its ids and cull bits arrived during the previous iterations), prefetch those of chunk c+3
            const bool cand1 = (q1 >> quad) & 1u;
            mask_nxt = __builtin_amdgcn_ballot_w64(cand1);
            mrgs_stage_async<S_MAX, SF, FV, S_LIVE, S_ROW>(stage[(c + 1) % MRGS_FWD_STAGES], rec, features, S, id1, cand1);
            idn = id1;
            id1 = id2; q1 = q2;
            id2 = 0; q2 = 0;
            if (base + 3 * MRGS_CHUNK + lane < total) { id2 = plist[base + 3 * MRGS_CHUNK + lane]; q2 = qm[base + 3 * MRGS_CHUNK + lane]; }
        };
        if (MRGS_FWD_STAGES == 2) stage_next();

        uint64_t m = mask_cur;
#if MRGS_FWD_REFINE
        // Few pixels of the block still alive (a silhouette block late in its list): the cull that tile_ranges_kernel evaluated
        // against the whole 8x8 block is evaluated again against the bounding rectangle of the LIVE pixels, all 64 entries of
        // the chunk at once (lane = entry).  An entry it removes reaches alpha >= 1/255 at no live pixel, i.e. every lane would
        // have failed the test below: same images, and the longest waves of the launch -- which set its duration, one
        // issue slot per ~4 cycles each -- walk fewer entries.
        {
            const uint64_t live = ~done;
            if (m != 0ull && __builtin_popcountll(live) <= MRGS_FWD_REFINE_LIVE) {
                const int r0 = __builtin_ctzll(live) >> 3, r1 = (63 - __builtin_clzll(live)) >> 3;
                uint32_t cols = (uint32_t)(live | (live >> 32));
                cols |= cols >> 16; cols |= cols >> 8; cols &= 0xFFu;
                const int c0 = __builtin_ctz(cols), c1 = 31 - __builtin_clz(cols);
                const CullConic cc = mrgs_cull_load(cull, idc);     // (fetching it a chunk ahead into registers: no gain measured)
                const bool touch = mrgs_block_may_touch(cc, (float)(bx * 8 + c0), (float)(by * 8 + r0), (float)(c1 - c0), (float)(r1 - r0));
                m &= __builtin_amdgcn_ballot_w64(touch);
            }
        }
#endif
        work += (uint32_t)__builtin_popcountll(m);
        const StageBuf<SF>& sb = stage[c % MRGS_FWD_STAGES];
        uint64_t contributed = 0ull;          // bit j: some live pixel of the block is hit by entry base + j
        uint64_t unsure = 0ull;               // bit j: ... and for some live pixel the hit itself is a decision inside its error band

        // One list entry (forward.cu:358-442).  Branch-free across lanes: a lane that does not blend this entry (no hit,
        // pixel already terminated, or terminating right now) runs the accumulation with alpha = 0 -- every sum gets
        // + x * 0 with a finite x, T gets * 1 -- and the depth, the only operand that can be non-finite for such a lane, is
        // replaced.  The results are bit-identical to skipping the entry.
        // (a0, a1: normal and color of the entry, fetched together with its geometry one entry ahead -- inside the branch
        // below their LDS latency sat on the critical path of the longest waves, which set the duration of this kernel)
        // MEDIAN: some live pixel of the block may still have T > 0.5 in this chunk (T only falls).  Dense blocks are past that after
        // their first chunk or two, and the entry body then carries neither the T > 0.5 test and its band nor the median selects.
        const bool median_live = (MRGS_BALLOT(T > 0.5f - MRGS_T2_EPS) & ~done) != 0ull;      // wave-uniform, per chunk
        auto blend_entry = [&](const SurfelGeom& sg, const float4& a0, const RecTail& a1, int j) {
            Hit h;
            const uint64_t may_hit = mrgs_intersect_mask(sg, px, py, h) & ~done;
#ifdef MRGS_WAVE_STATS
            const int ws_live = __builtin_popcountll(~done);     // how thin the wave runs near its end
            ws_le4 += ws_live <= 4; ws_le8 += ws_live <= 8; ws_le16 += ws_live <= 16;
#endif
            if (may_hit == 0ull) return;
#ifdef MRGS_WAVE_STATS
            ws_blend++; ws_blend_le8 += ws_live <= 8;
#endif
            work += 3u;   // an entry some pixel blends costs the backward about four times an entry that only gets tested
            contributed |= 1ull << j;          // (a superset of "blended": a pixel may terminate on it instead; the backward sorts that out)
            uint64_t ambiguous;
            const uint64_t ok = mrgs_hit_decide_mask(h, may_hit, ambiguous);
            // (an entry whose hit is ambiguous for some pixel: the backward evaluates it with the oracle's arithmetic for the whole
            // block, flag bit 1.  Resolving it here for the lanes concerned instead of marking their pixels was measured: the exact
            // evaluation as a cold branch of the entry body costs the loop 5 us, the ~20 more marked pixels 3)
            if (ambiguous != 0ull) unsure |= 1ull << j;
            const float oma = 1.0f - h.alpha;
            const float test_T = T * oma;
            // a decision the fast arithmetic cannot be sure of marks the pixel: it is rendered again, exactly, at the end of the wave
            // (mrgs_blend_math.h "Exact decisions"; mrgs_redo_pixel).  Three quarters of the marks are transmittance bands.
            // (the two transmittance tests as two comparisons each, against the near and the far edge of the band: between them the
            // pixel is marked, and what the fast path does with a marked pixel does not matter)
#if MRGS_T1_RUNNING
            // the band of the 1e-4 test: a running BOUND of |T_fast - T_exact| carried per pixel (mrgs_blend_math.h: MRGS_T1_RUNNING)
#if MRGS_T1_RUNNING == 2     // (the recurrence with d_n = 2.1e-7 rho_n + 2.5e-7 carried per pair: a tighter band for three more instructions)
            const float Terr_new = fmaf(Terr, oma, MRGS_T1_SLACK * fmaf(fmaf(2.1e-7f, fminf(h.rho3d, h.rho2d), 2.5e-7f) * h.alpha, T, 2.4e-7f * test_T));
#else
            const float Terr_new = fmaf(Terr, oma, (MRGS_T1_SLACK * MRGS_T_STEP_ERR) * T);
#endif
            const uint64_t below = MRGS_BALLOT(test_T < (MRGS_T_MIN - MRGS_T1_ABS) - Terr_new);
#else
            const uint64_t below = MRGS_BALLOT(test_T < MRGS_T_MIN - MRGS_T1_EPS);
#endif
            // (a T (1 - alpha) inside the band of the 1e-4 test is not looked for here: it does not end the pixel -- `below` is the near
            // edge of the band --, becomes the pixel's T, and nothing but a terminating entry can follow it: the pixel's FINAL T lies in
            // the band exactly when some entry's did, and is tested once, after the list)
            redo |= ambiguous;
            done |= ok & below;                               // forward.cu:400-404: the pixel stops BEFORE blending this entry
            const uint64_t upd_mask = ok & ~below;
            const bool upd = MRGS_LANES(upd_mask);
            const float alpha = upd ? h.alpha : 0.0f;
            const float depth = upd ? h.depth : 1.0f;
            const float w = alpha * T;
            const float A = 1.0f - T;
            const float m_ = mscale * (1.0f - MRGS_NEAR_N * mrgs_rcp(depth));
            const float mm = m_ * m_;
            // distortion += (m*m*A + M2 - 2*m*M1) * w   (forward.cu:412), fused as written here and in the oracle
            distortion = fmaf(fmaf(-2.0f * m_, M1, fmaf(mm, A, M2)), w, distortion);
            Dp = fmaf(depth, w, Dp);
            M1 = fmaf(m_, w, M1);
            M2 = fmaf(mm, w, M2);
            const uint32_t contributor = (uint32_t)(base + j + 1);
            if (median_live) {       // (ONE block for everything the T > 0.5 test feeds: T is still the transmittance in front of this entry)
                const uint64_t t_high = MRGS_BALLOT(T > 0.5f + MRGS_T2_EPS);
                const uint64_t t_band = MRGS_BALLOT(T > 0.5f - MRGS_T2_EPS) & ~t_high;
                redo |= ok & t_band;
                const bool med = MRGS_LANES(upd_mask & t_high);
                median_depth = med ? depth : median_depth;
                median_contributor = med ? contributor : median_contributor;
            }
            N0 = fmaf(a0.x, w, N0); N1 = fmaf(a0.y, w, N1); N2 = fmaf(a0.z, w, N2);
            C0 = fmaf(a0.w, w, C0); C1 = fmaf(a1.x, w, C1); C2 = fmaf(a1.y, w, C2);
            if (S_MAX > 0) {
                // every channel slot of the kernel instance, no per-channel branch on the runtime S: a branch per channel puts
                // each LDS read and its wait into a basic block of its own (measured: +110 us for 8 channels); the slots
                // beyond S accumulate whatever the stage buffer holds and are never written out
#pragma unroll
                for (int ch = 0; ch < S_STAGED; ch++) F[ch] = fmaf(mrgs_staged_feature<FV>(sb, ch, j), w, F[ch]);
                if constexpr (XREC) Fx = fmaf(a1.w, w, Fx);       // (the record's tail arrives one entry ahead with the rest of it)
            }
            T = upd ? test_T : T;
#if MRGS_T1_RUNNING
            Terr = upd ? Terr_new : Terr;
#endif
            last_contributor = upd ? contributor : last_contributor;
        };

        if (m != 0ull) {
            // front to back over the set bits; the geometry of the next entry is fetched from LDS while the current one is
            // evaluated, in two alternating register sets (no copies at the loop edge)
            int j = __builtin_ctzll(m);
            SurfelGeom sA, sB;
            sA.g0 = sb.rec[0][j]; sA.g1 = sb.rec[1][j]; sA.g2 = sb.rec[2][j];
            float4 tA0 = sb.rec[3][j], tB0;
            RecTail tA1 = *reinterpret_cast<const RecTail*>(&sb.rec[4][j]), tB1;
            while (true) {
                m &= m - 1;
                bool more = m != 0ull;
                int jn = more ? __builtin_ctzll(m) : j;
                sB.g0 = sb.rec[0][jn]; sB.g1 = sb.rec[1][jn]; sB.g2 = sb.rec[2][jn];
                tB0 = sb.rec[3][jn]; tB1 = *reinterpret_cast<const RecTail*>(&sb.rec[4][jn]);
                blend_entry(sA, tA0, tA1, j);
                if (!more) break;
                j = jn;
                m &= m - 1;
                more = m != 0ull;
                jn = more ? __builtin_ctzll(m) : j;
                sA.g0 = sb.rec[0][jn]; sA.g1 = sb.rec[1][jn]; sA.g2 = sb.rec[2][jn];
                tA0 = sb.rec[3][jn]; tA1 = *reinterpret_cast<const RecTail*>(&sb.rec[4][jn]);
                blend_entry(sB, tB0, tB1, j);
                if (!more) break;
                j = jn;
            }
        }
        // the backward walks exactly the entries flagged here: for every other entry all of its gradient terms are zeros
        // (no pixel of the block blended it), so skipping it there is bit-identical and saves the staging and the intersection
        if (base + lane < total) cf[(size_t)(base + lane) * 4] = (uint8_t)(((contributed >> lane) & 1ull) | (((unsure >> lane) & 1ull) << 1));
        cf_end = base + MRGS_CHUNK;
        if (MRGS_FWD_STAGES == 1) stage_next();
        mask_cur = mask_nxt;
        idc = idn;
    }
    mrgs_stage_wait();   // do not retire the wave with LDS-DMA still in flight

    // entries this wave tested + 3 x entries it blended: the cost of the backward wave of the same pixel block, which walks
    // the same entries (bwd_order_kernel)
    if (lane == 0) {
        item_work[tile * 4 + quad] = work;
        if (work_hint != nullptr) work_hint[tile * 4 + quad] = work;     // next visit of this camera orders its waves by it
    }
#ifdef MRGS_WAVE_STATS
    if (lane == 0 && b < 65536) {
        unsigned long long* w = g_wave_stats_fwd + 8 * (size_t)b;
        w[0] = ws_t0; w[1] = wall_clock64(); w[2] = __builtin_amdgcn_s_memtime() - ws_c0;
        w[3] = ((unsigned long long)(work - 3u * ws_blend) << 32) | ws_blend; w[4] = ((unsigned long long)ws_blend_le8 << 32) | (unsigned)total;
        w[6] = item_est[tile * 4 + quad];          // the estimate the queues were built from, next to the measured work
        w[5] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32);
        w[7] = ((unsigned long long)ws_le4 << 40) | ((unsigned long long)ws_le8 << 20) | ws_le16;   // entries tested with <= 4 / 8 / 16 live pixels
    }
#endif
    if (inside) {
        final_T[pix] = T;
        final_T[pix + HW] = M1;
        final_T[pix + 2 * HW] = M2;
        n_contrib[pix] = last_contributor;
        n_contrib[pix + HW] = median_contributor;
        out_color[pix] = fmaf(T, bg[0], C0);
        out_color[pix + HW] = fmaf(T, bg[1], C1);
        out_color[pix + 2 * HW] = fmaf(T, bg[2], C2);
        if (S_MAX > 0) {
#pragma unroll
            for (int ch = 0; ch < S_ROW; ch++)
                if (ch < S) out_feature[(size_t)ch * HW + pix] = ch < S_STAGED ? F[ch < SF ? ch : 0] : (XREC && ch == S_MAX) ? Fx : 0.0f;
        }
        out_others[pix + 0 * HW] = Dp;
        out_others[pix + 1 * HW] = 1.0f - T;
        out_others[pix + 2 * HW] = N0;
        out_others[pix + 3 * HW] = N1;
        out_others[pix + 4 * HW] = N2;
        out_others[pix + 5 * HW] = median_depth;
        out_others[pix + 6 * HW] = distortion;
    }

    // ---- marked pixels (mrgs_blend_math.h "Exact decisions"): listed for render_fwd_redo_kernel, which renders them again with the
    // oracle's arithmetic and overwrites what was written above.  ~1e-4 of the pixels.  The backward walks flagged list entries only
    // and the redo may blend entries this wave never reached: the flags it did not write are cleared here, the redo then only sets.
#ifdef MRGS_FWD_REDO_ALL   // developer build: every pixel goes through the exact path (what the margins are measured against)
    redo = ~0ull;
#endif
#if MRGS_T1_RUNNING
    redo |= MRGS_BALLOT(T < (MRGS_T_MIN + MRGS_T1_ABS) + Terr);
#else
    redo |= MRGS_BALLOT(T < MRGS_T_MIN + MRGS_T1_EPS);
#endif
    redo &= __builtin_amdgcn_ballot_w64(inside);
    if (redo != 0ull) {
        for (int e = cf_end + lane; e < total; e += MRGS_CHUNK) cf[(size_t)e * 4] = 0;
#if MRGS_FWD_REDO_INLINE
        if (lane == 0) atomicAdd(&redo_list[0], (uint32_t)__builtin_popcountll(redo));      // (diagnostics: mrgs_debug_export 12 reports the count)
        // (the list staging buffer is free by now: the candidate queue of the redo lives there)
        uint32_t* q = reinterpret_cast<uint32_t*>(&stage[0]);
        static_assert(sizeof(StageBuf<SF>) >= 2 * MRGS_REDO_QCAP * sizeof(uint32_t), "the redo's queue must fit the staging buffer");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        while (redo != 0ull) {
            const int p = __builtin_ctzll(redo);
            redo &= redo - 1;
            mrgs_redo_pixel<S_ROW>(__builtin_amdgcn_readlane(pix, p), lane, q, q + MRGS_REDO_QCAP, ranges, point_list, qmask, cflag, S, W, H, tiles_x, rec, features,
                                   bg, final_T, n_contrib, out_color, out_feature, out_others);
            __builtin_amdgcn_wave_barrier();
        }
#else
        if ((redo >> lane) & 1ull) redo_list[2 + atomicAdd(&redo_list[0], 1u)] = (uint32_t)pix;
#endif
    }
}

#define MRGS_FWD_KERNEL render_fwd_kernel
#ifdef MRGS_FWD_REDO_ALL
#define MRGS_REDO_BLOCKS 8192
#else
#define MRGS_REDO_BLOCKS 512
#endif

void mrgs_launch_render_fwd(const MrgsRasterConfig& cfg, const MrgsRasterInputs& in, const MrgsGeomWs& g, const uint32_t* plist,
                            const uint8_t* qmask, uint8_t* cflag, const MrgsImgWs& img, float* out_color, float* out_feature, float* out_others, hipStream_t stream)
{
    const int tiles_x = (cfg.W + MRGS_BLOCK_X - 1) / MRGS_BLOCK_X, tiles_y = (cfg.H + MRGS_BLOCK_Y - 1) / MRGS_BLOCK_Y;
    const int ntiles = tiles_x * tiles_y;
    // one wave per slot of the work queues (mrgs_pull_item): items rounded up to whole dealing passes; blockIdx % 8 = XCD list
    const int nblocks = (((ntiles + 7) / 8) * 4 + MRGS_MAX_SIMD_QUEUES) * 8;
    const dim3 grid(nblocks), block(64);
#define LAUNCH(SM, ...)                                                                                                           \
    hipLaunchKernelGGL((MRGS_FWD_KERNEL<SM, __VA_ARGS__>), grid, block, 0, stream, img.ranges, img.fwd_assign, img.blend_state, plist, qmask, cflag, cfg.S, cfg.W, cfg.H, tiles_x, ntiles, \
                       g.rec, g.cull, in.features, in.bg, img.final_T, img.n_contrib, out_color, out_feature, out_others, img.item_work, img.item_est, in.work_hint, mrgs_waves_per_simd<MRGS_FWD_KERNEL<SM, __VA_ARGS__>>(), img.redo_list)
    // FV instances: the feature rows are exactly S_MAX floats (16-byte aligned pieces, see mrgs_stage_async)
    const bool fv_ok = ((uintptr_t)in.features & 15u) == 0;   // 16-byte DMA pieces need an aligned feature tensor
    if (cfg.S == 0) LAUNCH(0, false, 0);
    else if (cfg.S == 8 && fv_ok) LAUNCH(8, true, 8);
    else if (cfg.S <= 8) LAUNCH(8, false, 8);
    else if (cfg.S == 12 && fv_ok && in.features_live == 9u) LAUNCH(8, true, 9);       // rows of 9 channels in 12 floats: 8 staged, the ninth in the surfel record
    else if (cfg.S == 12 && fv_ok) LAUNCH(12, true, 12);
    else if (cfg.S <= 12) LAUNCH(12, false, 12);
    else if (cfg.S == 24 && fv_ok) LAUNCH(24, true, 24);
    else LAUNCH(24, false, 24);
#undef LAUNCH
#if !MRGS_FWD_REDO_INLINE
    // the marked pixels again, exactly (an empty list most of the time: the launch is there for the count it reads on the device)
#define REDO(SM) hipLaunchKernelGGL((render_fwd_redo_kernel<SM>), dim3(MRGS_REDO_BLOCKS), block, 0, stream, img.ranges, plist, qmask, cflag, cfg.S, cfg.W, cfg.H, tiles_x, \
                                    g.rec, in.features, in.bg, img.final_T, img.n_contrib, out_color, out_feature, out_others, img.redo_list)
    if (cfg.S == 0) REDO(0);
    else if (cfg.S <= 8) REDO(8);
    else if (cfg.S <= 12) REDO(12);
    else REDO(24);
#undef REDO
#endif
}
