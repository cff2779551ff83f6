// mrgs_render_bwd_pairs.h -- the backward blend with TWO list entries per step in the two halves of packed fp32 instructions.
//
// render_bwd_kernel (mrgs_render_bwd.hip) runs at 0.91-0.95 of the VALU issue rate: only fewer issue cycles per entry make it faster.
// gfx950 has v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 (two fp32 results per instruction, 5.0 issue cycles measured against 2 x 4.0-4.3
// for the scalar forms, tools/ubench).  Packing the three colour channels of ONE entry was tried in round 2 and lost (register-pair
// moves).  Here the pair is two ENTRIES: everything that does not depend on the running transmittance -- the ray/splat intersection, the
// exponential's argument, the geometric part of the gradient, the per-channel products -- is evaluated for entries 2p+1 and 2p of the
// staged chunk in the high and low half of the same instructions; the T / accum_rec recurrences stay serial (hi first: back to front),
// and each entry's 16+S terms go through the same transposing reduction and atomics as before.  Per entry the arithmetic is the SAME
// expression tree on the same operands as in the one-entry kernel (a packed fma is an IEEE fma per half), so gradients agree with it bit
// for bit up to the order of the atomics.
//
// What makes the operands arrive as pairs without register moves:
//   * compaction: the lanes of a chunk whose entry the forward flagged for this quadrant are counted (mbcnt) and the k-th flagged entry is
//     staged by lane k, so the walk is over consecutive slots n-1 ... 0 instead of over the set bits of a mask;
//   * in-place transposition: the LDS-DMA lands records as float4 per slot; every lane then rewrites its slot as 20 (+S) single floats in
//     field-major order (field f of slot s at soa[f][s]), so that fields of slots 2p and 2p+1 are one aligned 8-byte LDS read.
//   An odd count is padded with a zeroed slot whose entry has no active lane.
#pragma once

typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 dup2(float x) { return (f2){x, x}; }
__device__ __forceinline__ f2 sel2(bool lo, bool hi, f2 a, f2 b) { return (f2){lo ? a.x : b.x, hi ? a.y : b.y}; }
__device__ __forceinline__ f2 rcp2(f2 x) { return (f2){mrgs_rcp(x.x), mrgs_rcp(x.y)}; }
__device__ __forceinline__ f2 rcp2_pz(f2 x) { return (f2){mrgs_rcp_pz(x.x), mrgs_rcp_pz(x.y)}; }
__device__ __forceinline__ f2 min2(f2 a, f2 b) { return (f2){fminf(a.x, b.x), fminf(a.y, b.y)}; }
// mrgs_exp on both halves (same operations per half)
__device__ __forceinline__ f2 exp2_pair(f2 x)
{
    const float L2E_HI = 1.44269502162933349609375f, L2E_LO = 1.925963033500011e-8f, LN2 = 0.693147182464599609375f;
    const f2 t = x * L2E_HI;
    f2 e = pk_fma(x, dup2(L2E_HI), -t);
    e = pk_fma(x, dup2(L2E_LO), e);
    const f2 r = (f2){__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
    return pk_fma(r, e * LN2, r);
}

// field-major view of a stage buffer after the transposition: the same bytes as StageBuf<SF>::rec / feat
#define MRGS_SOA_FIELDS 20
template <int SF>
__device__ __forceinline__ float* mrgs_soa(StageBuf<SF>& sb) { return reinterpret_cast<float*>(&sb.rec[0][0]); }
template <int SF>
__device__ __forceinline__ f2 mrgs_soa_pair(const StageBuf<SF>& sb, int field, int p)      // slots (2p, 2p+1) of a field: .x = 2p, .y = 2p+1
{
    return *reinterpret_cast<const f2*>(reinterpret_cast<const float*>(&sb.rec[0][0]) + field * MRGS_CHUNK + 2 * p);
}
template <int SF, bool FV>
__device__ __forceinline__ f2 mrgs_soa_feature_pair(const StageBuf<SF>& sb, int ch, int p)
{
    // FV: transposed like the records (channel ch = float4 #(ch / 4), component ch % 4 -> field ch); else already [channel][slot]
    return *reinterpret_cast<const f2*>(&sb.feat[0][0] + ch * MRGS_CHUNK + 2 * p);
}

template <int S_MAX, bool FV>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(S_MAX == 0 ? MRGS_BWDP_WPE0 : S_MAX <= 8 ? MRGS_BWDP_WPE8 : 2, 8))) render_bwd_pairs_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ bwd_assign, uint32_t* __restrict__ blend_state, const uint32_t* __restrict__ point_list,
    const uint8_t* __restrict__ cflag, int S, int W, int H, int tiles_x, int ntiles,
    const float4* __restrict__ rec, const float* __restrict__ features, const float* __restrict__ bg,
    const float* __restrict__ final_Ts, const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpixels,
    const float* __restrict__ dL_dpixels_f, const float* __restrict__ dL_dothers, float* __restrict__ grad_rec, int gstride, int slots)
{
    constexpr int SF = S_MAX > 0 ? S_MAX : 1;
    constexpr int K = 16 + S_MAX;
    __shared__ StageBuf<SF> stage;
    __shared__ uint32_t s_meta[MRGS_CHUNK];      // compacted entries: list position inside the chunk << 26 | surfel id

    const int lane = threadIdx.x;
    const int b = blockIdx.x;
    const uint32_t item = mrgs_pull_item(blend_state + MRGS_QS_BWD, blend_state + MRGS_CS_BASE, bwd_assign, ntiles, b & 7, b >> 3, lane, slots);
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
    int max_contrib = last_contributor;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) max_contrib = max(max_contrib, __shfl_xor(max_contrib, d, 64));
    if (max_contrib == 0) return;
    if (prio == 3u) __builtin_amdgcn_s_setprio(3);
    else if (prio == 2u) __builtin_amdgcn_s_setprio(2);
    else if (prio == 1u) __builtin_amdgcn_s_setprio(1);
    const int median_contributor = inside ? (int)n_contrib[pix + HW] : 0;

    const float T_final = inside ? final_Ts[pix] : 0.f;
    float T = T_final;
    float accum_rec[3] = {0.f, 0.f, 0.f}, dL_dpixel[3] = {0.f, 0.f, 0.f};
    float accum_rec_f[SF], dL_dpixel_f[SF];
#pragma unroll
    for (int i = 0; i < SF; i++) { accum_rec_f[i] = 0.f; dL_dpixel_f[i] = 0.f; }
    float dL_dreg = 0.f, dL_ddepth = 0.f, dL_daccum = 0.f, dL_dnormal2D[3] = {0.f, 0.f, 0.f}, dL_dmedian_depth = 0.f;
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
            for (int i = 0; i < S_MAX; i++)
                if (i < S) dL_dpixel_f[i] = dL_dpixels_f[(size_t)i * HW + pix];
        }
    }
    float accum_depth_rec = 0.f, accum_alpha_rec = 0.f, accum_normal_rec[3] = {0.f, 0.f, 0.f};
    const float final_D = inside ? final_Ts[pix + HW] : 0.f;
    const float final_D2 = inside ? final_Ts[pix + 2 * HW] : 0.f;
    float last_dL_dT = 0.f;
    const float mscale = MRGS_FAR_N / (MRGS_FAR_N - MRGS_NEAR_N);
    const float dmd_scale = (MRGS_FAR_N * MRGS_NEAR_N) / (MRGS_FAR_N - MRGS_NEAR_N);
    const float bg_dot_dpixel = fmaf(bg[2], dL_dpixel[2], fmaf(bg[1], dL_dpixel[1], bg[0] * dL_dpixel[0]));
    const float final_A = 1.0f - T_final;

    const uint32_t* plist = point_list + range.x;
    const uint8_t* qm = cflag + (size_t)range.x * 4 + quad;
    const ReduceLane rl = mrgs_reduce_lane(lane);
    const uint32_t row_bytes = (uint32_t)gstride * 4u;
    const int c_top = (max_contrib - 1) / MRGS_CHUNK;
    const f2 PX = dup2(px), PY = dup2(py);

    // (id | flag << 28) of the chunks ahead, as in the one-entry kernel
    auto fetch = [&](int cc, bool bounded) -> uint32_t {
        const int pos = cc * MRGS_CHUNK + lane;
        if (bounded && pos >= max_contrib) return 0u;
        return plist[pos] | ((uint32_t)qm[(size_t)pos * 4] << 28);
    };
    // stage the flagged entries of a chunk, compacted: the k-th flagged entry (in list order) lands in slot k.  Returns their number.
    auto stage_chunk = [&](uint32_t idq) -> int {
        const bool cand = (idq >> 28) & 1u;
        const uint64_t m = __builtin_amdgcn_ballot_w64(cand);
        const int n = (int)__popcll(m);
        const int k = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (cand) s_meta[k] = ((uint32_t)lane << 26) | (idq & 0x03FFFFFFu);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t meta = lane < n ? s_meta[lane] : 0u;
        mrgs_stage_async<S_MAX, SF, FV>(stage, rec, features, S, meta & 0x03FFFFFFu, lane < n);
        return n;
    };
    // after the records have landed: every lane rewrites its slot field-major (in place: all reads of the wave precede its writes in the
    // LDS queue); the slot behind an odd count is zeroed (its entry then has no active lane)
    auto transpose = [&](int n) {
        float4 r[5];
        float4 fq[FV ? SF / 4 : 1];
        const bool mine = lane < n;
        const bool pad = (n & 1) && lane == n;
#pragma unroll
        for (int f = 0; f < 5; f++) r[f] = mine ? stage.rec[f][lane] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (S_MAX > 0 && FV) {
#pragma unroll
            for (int q = 0; q < SF / 4; q++) fq[q] = mine ? reinterpret_cast<const float4*>(&stage.feat[0][0])[q * MRGS_CHUNK + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (mine || pad) {
            float* soa = mrgs_soa(stage);
#pragma unroll
            for (int f = 0; f < 5; f++) {
                soa[(4 * f + 0) * MRGS_CHUNK + lane] = r[f].x; soa[(4 * f + 1) * MRGS_CHUNK + lane] = r[f].y;
                soa[(4 * f + 2) * MRGS_CHUNK + lane] = r[f].z; soa[(4 * f + 3) * MRGS_CHUNK + lane] = r[f].w;
            }
            if (S_MAX > 0) {
                if (FV) {
                    float* fs = &stage.feat[0][0];
#pragma unroll
                    for (int q = 0; q < SF / 4; q++) {
                        fs[(4 * q + 0) * MRGS_CHUNK + lane] = fq[q].x; fs[(4 * q + 1) * MRGS_CHUNK + lane] = fq[q].y;
                        fs[(4 * q + 2) * MRGS_CHUNK + lane] = fq[q].z; fs[(4 * q + 3) * MRGS_CHUNK + lane] = fq[q].w;
                    }
                } else if (pad) {
#pragma unroll
                    for (int ch = 0; ch < S_MAX; ch++) stage.feat[ch][lane] = 0.f;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };

    uint32_t idq1 = 0, idq2 = 0;
    int n_cur;
    {
        const uint32_t idq0 = fetch(c_top, true);
        if (c_top >= 1) idq1 = fetch(c_top - 1, false);
        if (c_top >= 2) idq2 = fetch(c_top - 2, false);
        n_cur = stage_chunk(idq0);
    }

    for (int c = c_top; c >= 0; c--) {
        const int base = c * MRGS_CHUNK;
        mrgs_stage_wait();                    // chunk c has landed
        const int n = n_cur;
        transpose(n);
        const StageBuf<SF>& sb = stage;

        for (int p = (n + 1) / 2 - 1; p >= 0; p--) {
            // entries: .y = slot 2p+1 (processed first: back to front), .x = slot 2p
            const uint32_t meta_lo = s_meta[2 * p], meta_hi = (2 * p + 1 < n) ? s_meta[2 * p + 1] : 0u;
            const bool valid_hi = 2 * p + 1 < n;
            const int contributor_lo = base + (int)(meta_lo >> 26), contributor_hi = base + (int)(meta_hi >> 26);
            // ---- intersection (mrgs_intersect on both halves) ----
            // (fetching the next pair's geometry one step ahead, as the one-entry kernel does, needs 24 more registers: 168 VGPRs with
            // spills, 0.299 instead of 0.267 ms)
            const f2 Tux = mrgs_soa_pair(sb, 0, p), Tuy = mrgs_soa_pair(sb, 1, p), Tuz = mrgs_soa_pair(sb, 2, p);
            const f2 Tvx = mrgs_soa_pair(sb, 3, p), Tvy = mrgs_soa_pair(sb, 4, p), Tvz = mrgs_soa_pair(sb, 5, p);
            const f2 Twx = mrgs_soa_pair(sb, 6, p), Twy = mrgs_soa_pair(sb, 7, p), Twz = mrgs_soa_pair(sb, 8, p);
            const f2 m2x = mrgs_soa_pair(sb, 9, p), m2y = mrgs_soa_pair(sb, 10, p), opac = mrgs_soa_pair(sb, 11, p);
            const f2 kx = pk_fma(PX, Twx, -Tux), ky = pk_fma(PX, Twy, -Tuy), kz = pk_fma(PX, Twz, -Tuz);
            const f2 lx = pk_fma(PY, Twx, -Tvx), ly = pk_fma(PY, Twy, -Tvy), lz = pk_fma(PY, Twz, -Tvz);
            const f2 ppx = pk_fma(ky, lz, -(kz * ly));
            const f2 ppy = pk_fma(kz, lx, -(kx * lz));
            const f2 ppz = pk_fma(kx, ly, -(ky * lx));
            const f2 h_inv_pz = rcp2_pz(ppz);
            const f2 h_sx = ppx * h_inv_pz, h_sy = ppy * h_inv_pz;
            const f2 rho3d = pk_fma(h_sx, h_sx, h_sy * h_sy);
            const f2 hdx = m2x - PX, hdy = m2y - PY;
            const f2 rho2d = MRGS_FILTER_INV_SQUARE * pk_fma(hdx, hdx, hdy * hdy);
            const f2 rho = min2(rho3d, rho2d);
            const bool use3d_lo = rho3d.x <= rho2d.x, use3d_hi = rho3d.y <= rho2d.y;
            const f2 depth3 = pk_fma(h_sx, Twx, pk_fma(h_sy, Twy, Twz));
            const f2 h_depth = sel2(use3d_lo, use3d_hi, depth3, Twz);
            const f2 power = -0.5f * rho;
            const f2 h_G = exp2_pair(power);
            const f2 h_alpha = min2(dup2(0.99f), opac * h_G);
            const bool hit_lo = (ppz.x != 0.0f) & !(h_depth.x < MRGS_NEAR_N) & !(power.x > 0.0f) & !(h_alpha.x < MRGS_ALPHA_MIN);
            const bool hit_hi = (ppz.y != 0.0f) & !(h_depth.y < MRGS_NEAR_N) & !(power.y > 0.0f) & !(h_alpha.y < MRGS_ALPHA_MIN);
            const bool act_lo = hit_lo & inside & (contributor_lo < last_contributor);
            const bool act_hi = hit_hi & inside & (contributor_hi < last_contributor) & valid_hi;
            const uint64_t amask_lo = __builtin_amdgcn_ballot_w64(act_lo), amask_hi = __builtin_amdgcn_ballot_w64(act_hi);
            if ((amask_lo | amask_hi) == 0ull) continue;
            const bool a3_lo = act_lo & use3d_lo, a3_hi = act_hi & use3d_hi;
            const f2 zero2 = dup2(0.0f);
            const f2 alpha = sel2(act_lo, act_hi, h_alpha, zero2);
            const f2 G = sel2(act_lo, act_hi, h_G, zero2);
            const f2 c_d = sel2(act_lo, act_hi, h_depth, dup2(1.0f));
            const f2 sx = sel2(a3_lo, a3_hi, h_sx, zero2), sy = sel2(a3_lo, a3_hi, h_sy, zero2);
            const f2 inv_pz = sel2(a3_lo, a3_hi, h_inv_pz, zero2);

            const f2 one_m_a = 1.0f - alpha;
            const f2 inv_1ma = rcp2(one_m_a);
            // the transmittance in front of each entry: hi first
            f2 Tp;
            Tp.y = T * inv_1ma.y;                                  // backward.cu:330
            Tp.x = Tp.y * inv_1ma.x;
            T = Tp.x;
            const f2 w = alpha * Tp;
            f2 g[K];
            f2 dL_dalpha = zero2;
            // appearance: normal = fields 12..14, colour = 15..17 (record float4 #3 = normal.xyz, col.x; #4 = col.yz, depth, -)
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                const f2 col = mrgs_soa_pair(sb, 15 + ch, p);
                f2 acc;                                            // accum_rec in front of each entry
                acc.y = accum_rec[ch];
                acc.x = fmaf(alpha.y, col.y, one_m_a.y * acc.y);   // backward.cu:340-342 after the hi entry
                accum_rec[ch] = fmaf(alpha.x, col.x, one_m_a.x * acc.x);
                dL_dalpha = pk_fma(col - acc, dup2(dL_dpixel[ch]), dL_dalpha);
                g[MRGS_G_COL + ch] = w * dL_dpixel[ch];
            }
            if (S_MAX > 0) {
#pragma unroll
                for (int ch = 0; ch < S_MAX; ch++) {
                    const f2 f = (FV || ch < S) ? mrgs_soa_feature_pair<SF, FV>(sb, ch, p) : zero2;
                    f2 acc;
                    acc.y = accum_rec_f[ch];
                    acc.x = fmaf(alpha.y, f.y, one_m_a.y * acc.y);
                    accum_rec_f[ch] = fmaf(alpha.x, f.x, one_m_a.x * acc.x);
                    dL_dalpha = pk_fma(f - acc, dup2(dL_dpixel_f[ch]), dL_dalpha);
                    g[MRGS_G_FEAT + ch] = w * dL_dpixel_f[ch];
                }
            }
            const f2 inv_cd = rcp2(c_d);
            const f2 m_d = mscale * (1.0f - MRGS_NEAR_N * inv_cd);
            const f2 dmd_dd = dmd_scale * inv_cd * inv_cd;
            f2 dL_dz = (f2){(act_lo & (contributor_lo == median_contributor - 1)) ? dL_dmedian_depth : 0.0f,
                            (act_hi & (contributor_hi == median_contributor - 1)) ? dL_dmedian_depth : 0.0f};
            const f2 dL_dweight = pk_fma(-2.0f * m_d, dup2(final_D), pk_fma(m_d * m_d, dup2(final_A), dup2(final_D2))) * dL_dreg;
            {
                f2 last;                                           // last_dL_dT in front of each entry
                last.y = last_dL_dT;
                last.x = fmaf(dL_dweight.y, alpha.y, (1.0f - alpha.y) * last.y);
                last_dL_dT = fmaf(dL_dweight.x, alpha.x, (1.0f - alpha.x) * last.x);
                dL_dalpha += dL_dweight - last;
            }
            const f2 dL_dmd = 2.0f * w * pk_fma(m_d, dup2(final_A), dup2(-final_D)) * dL_dreg;
            dL_dz = pk_fma(dL_dmd, dmd_dd, dL_dz);
            {
                f2 acc;
                acc.y = accum_depth_rec;
                acc.x = fmaf(alpha.y, c_d.y, one_m_a.y * acc.y);
                accum_depth_rec = fmaf(alpha.x, c_d.x, one_m_a.x * acc.x);
                dL_dalpha = pk_fma(c_d - acc, dup2(dL_ddepth), dL_dalpha);
            }
            {
                f2 acc;
                acc.y = accum_alpha_rec;
                acc.x = fmaf(one_m_a.y, acc.y, alpha.y);
                accum_alpha_rec = fmaf(one_m_a.x, acc.x, alpha.x);
                dL_dalpha = pk_fma(1.0f - acc, dup2(dL_daccum), dL_dalpha);
            }
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                const f2 nrm = mrgs_soa_pair(sb, 12 + ch, p);
                f2 acc;
                acc.y = accum_normal_rec[ch];
                acc.x = fmaf(alpha.y, nrm.y, one_m_a.y * acc.y);
                accum_normal_rec[ch] = fmaf(alpha.x, nrm.x, one_m_a.x * acc.x);
                dL_dalpha = pk_fma(nrm - acc, dup2(dL_dnormal2D[ch]), dL_dalpha);
                g[MRGS_G_NRM + ch] = w * dL_dnormal2D[ch];
            }
            dL_dalpha *= Tp;
            dL_dalpha = pk_fma(-T_final * inv_1ma, dup2(bg_dot_dpixel), dL_dalpha);
            dL_dalpha = sel2(act_lo, act_hi, dL_dalpha, zero2);
            const f2 dL_dG = opac * dL_dalpha;
            dL_dz = pk_fma(w, dup2(dL_ddepth), dL_dz);
            {   // ray/splat branch (zero for lanes in the low-pass branch: s = 0 and 1/p.z = 0 there)
                const f2 dGn = dL_dG * -G;
                const f2 dL_dsx = pk_fma(dGn, sx, dL_dz * Twx);
                const f2 dL_dsy = pk_fma(dGn, sy, dL_dz * Twy);
                const f2 dpx = dL_dsx * inv_pz, dpy = dL_dsy * inv_pz;
                const f2 dpz = -pk_fma(dpx, sx, dpy * sy);
                const f2 ndkx = pk_fma(-ly, dpz, lz * dpy);
                const f2 ndky = pk_fma(-lz, dpx, lx * dpz);
                const f2 ndkz = pk_fma(-lx, dpy, ly * dpx);
                const f2 ndlx = pk_fma(-dpy, kz, dpz * ky);
                const f2 ndly = pk_fma(-dpz, kx, dpx * kz);
                const f2 ndlz = pk_fma(-dpx, ky, dpy * kx);
                g[0] = ndkx; g[1] = ndky; g[2] = ndkz;
                g[3] = ndlx; g[4] = ndly; g[5] = ndlz;
                g[6] = pk_fma(PX, -ndkx, pk_fma(PY, -ndlx, dL_dz * sx));
                g[7] = pk_fma(PX, -ndky, pk_fma(PY, -ndly, dL_dz * sy));
                g[8] = pk_fma(PX, -ndkz, pk_fma(PY, -ndlz, dL_dz));
            }
            g[MRGS_G_OPA] = G * dL_dalpha;
            // ---- the two entries' terms through the transposing reduction, hi first ----
            if (amask_hi != 0ull) {
                float gh[K];
#pragma unroll
                for (int i = 0; i < K; i++) gh[i] = g[i].y;
                const uint32_t row_off = (meta_hi & 0x03FFFFFFu) * row_bytes;
                wave_reduce_atomic_add<K>(gh, grad_rec, row_off, rl);
                const bool a2 = act_hi & !use3d_hi;
                if (__builtin_amdgcn_ballot_w64(a2) != 0ull) {   // low-pass-filter branch: dL/dmean2D
                    const float dL_dG2 = a2 ? dL_dG.y : 0.0f;
                    const float dGf = -G.y * MRGS_FILTER_INV_SQUARE;
                    wave_reduce_atomic_add2(dL_dG2 * (dGf * hdx.y), dL_dG2 * (dGf * hdy.y), grad_rec, row_off + 4u * MRGS_G_M2(S_MAX), lane);
                }
            }
            if (amask_lo != 0ull) {
                float gl[K];
#pragma unroll
                for (int i = 0; i < K; i++) gl[i] = g[i].x;
                const uint32_t row_off = (meta_lo & 0x03FFFFFFu) * row_bytes;
                wave_reduce_atomic_add<K>(gl, grad_rec, row_off, rl);
                const bool a2 = act_lo & !use3d_lo;
                if (__builtin_amdgcn_ballot_w64(a2) != 0ull) {
                    const float dL_dG2 = a2 ? dL_dG.x : 0.0f;
                    const float dGf = -G.x * MRGS_FILTER_INV_SQUARE;
                    wave_reduce_atomic_add2(dL_dG2 * (dGf * hdx.x), dL_dG2 * (dGf * hdy.x), grad_rec, row_off + 4u * MRGS_G_M2(S_MAX), lane);
                }
            }
        }
        // next chunk (one stage buffer: staged after this one has been walked)
        n_cur = 0;
        if (c >= 1) {
            n_cur = stage_chunk(idq1);
            idq1 = idq2;
            idq2 = c >= 3 ? fetch(c - 3, false) : 0u;
        }
    }
    mrgs_stage_wait();   // do not retire the wave with LDS-DMA still in flight
}
