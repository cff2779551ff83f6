// mrgs_render_fwd_pairs.h -- the forward blend with the ray/splat intersection of TWO list entries per step in the halves of packed
// fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32).
//
// The forward's duration is the lifetime of its heaviest waves (~580 tested / ~355 blended entries, one wave alone
// on its SIMD issuing a dependent chain), and about half of an entry's instructions are the intersection, which does not depend on the
// running transmittance.  Here the candidates of a chunk are staged COMPACTED (the k-th candidate of the quadrant lands in slot k), every
// lane rewrites its slot field-major in place, and the walk takes slots (2p, 2p+1) together: one packed intersection for both, then the
// two blends one after the other, each exactly the one-entry kernel's blend (same expressions, same order: bit-identical images).
// Shares the pair helpers with mrgs_render_bwd_pairs.h.
#pragma once

#ifndef MRGS_PAIR_HELPERS
#define MRGS_PAIR_HELPERS
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
// after the records of `n` compacted entries have landed: every lane rewrites its slot field-major (in place: all reads of the wave
// precede its writes in the LDS queue); the slot behind an odd count is zeroed (its entry then hits no pixel)
template <int S_MAX, int SF, bool FV>
__device__ __forceinline__ void mrgs_soa_transpose(StageBuf<SF>& stage, int n, int lane)
{
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
}
#endif   // MRGS_PAIR_HELPERS

#ifndef MRGS_FWDP_WPE0
#define MRGS_FWDP_WPE0 5
#endif
#ifndef MRGS_FWDP_WPE8
#define MRGS_FWDP_WPE8 4
#endif

template <int S_MAX, bool FV>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(S_MAX == 0 ? MRGS_FWDP_WPE0 : S_MAX <= 8 ? MRGS_FWDP_WPE8 : 2, 8))) render_fwd_pairs_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ fwd_assign, uint32_t* __restrict__ blend_state, const uint32_t* __restrict__ point_list,
    const uint8_t* __restrict__ qmask, uint8_t* __restrict__ cflag, int S, int W, int H, int tiles_x, int ntiles,
    const float4* __restrict__ rec, const float* __restrict__ features, const float* __restrict__ bg, float* __restrict__ final_T,
    uint32_t* __restrict__ n_contrib, float* __restrict__ out_color, float* __restrict__ out_feature, float* __restrict__ out_others,
    uint32_t* __restrict__ item_work, const uint32_t* __restrict__ item_est, uint32_t* __restrict__ work_hint, int slots)
{
    constexpr int SF = S_MAX > 0 ? S_MAX : 1;
    __shared__ StageBuf<SF> stage;
    __shared__ uint32_t s_meta[MRGS_CHUNK];      // compacted candidates: list position inside the chunk << 26 | surfel id
    (void)item_est;

    const int lane = threadIdx.x;
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
    const int total = (int)(range.y - range.x);
    if (prio == 3u) __builtin_amdgcn_s_setprio(3);
    else if (prio == 2u) __builtin_amdgcn_s_setprio(2);
    else if (prio == 1u) __builtin_amdgcn_s_setprio(1);

    bool done = !inside;
    uint32_t work = 0;
    float T = 1.0f;
    float C0 = 0.f, C1 = 0.f, C2 = 0.f, N0 = 0.f, N1 = 0.f, N2 = 0.f;
    float F[SF];
#pragma unroll
    for (int i = 0; i < SF; i++) F[i] = 0.f;
    float Dp = 0.f, M1 = 0.f, M2 = 0.f, distortion = 0.f, median_depth = 0.f;
    uint32_t last_contributor = 0, median_contributor = 0;
    const float mscale = MRGS_FAR_N / (MRGS_FAR_N - MRGS_NEAR_N);
    const f2 PX = dup2(px), PY = dup2(py);

    const uint32_t* plist = point_list + range.x;
    const uint8_t* qm = qmask + range.x;
    uint8_t* cf = cflag + (size_t)range.x * 4 + quad;
    // (id, cull bits) of the chunks ahead; the chunk being walked keeps, per lane, whether its entry is a candidate and its slot
    uint32_t id1 = 0, id2 = 0, q1 = 0, q2 = 0;
    bool cand_cur = false;
    int slot_cur = 0, n_cur = 0;
    auto stage_chunk = [&](uint32_t id, uint32_t q) {
        cand_cur = (q >> quad) & 1u;
        const uint64_t m = __builtin_amdgcn_ballot_w64(cand_cur);
        n_cur = (int)__popcll(m);
        slot_cur = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (cand_cur) s_meta[slot_cur] = ((uint32_t)lane << 26) | (id & 0x03FFFFFFu);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t meta = lane < n_cur ? s_meta[lane] : 0u;
        mrgs_stage_async<S_MAX, SF, FV>(stage, rec, features, S, meta & 0x03FFFFFFu, lane < n_cur);
    };
    {
        uint32_t id0 = 0, q0 = 0;
        if (lane < total) { id0 = plist[lane]; q0 = qm[lane]; }
        if (MRGS_CHUNK + lane < total) { id1 = plist[MRGS_CHUNK + lane]; q1 = qm[MRGS_CHUNK + lane]; }
        if (2 * MRGS_CHUNK + lane < total) { id2 = plist[2 * MRGS_CHUNK + lane]; q2 = qm[2 * MRGS_CHUNK + lane]; }
        stage_chunk(id0, q0);
    }

    for (int base = 0; base < total; base += MRGS_CHUNK) {
        if (__builtin_amdgcn_ballot_w64(!done) == 0ull) break;   // every pixel of the block has terminated (forward.cu:342-344, per wave)
        mrgs_stage_wait();                    // the chunk has landed
        const int n = n_cur;
        mrgs_soa_transpose<S_MAX, SF, FV>(stage, n, lane);
        const StageBuf<SF>& sb = stage;
        work += (uint32_t)n;
        uint64_t contributed = 0ull;          // bit k: some live pixel of the block is hit by the entry in slot k

        // one entry's blend: the one-entry kernel's (forward.cu:358-442), branch-free across lanes
        auto blend_one = [&](bool hit, float h_alpha, float h_depth, int slot, float n0, float n1, float n2, float c0, float c1, float c2, int p, bool hi) {
            const bool ok = hit & !done;
            if (__builtin_amdgcn_ballot_w64(ok) == 0ull) return;
            work += 3u;
            contributed |= 1ull << slot;
            const float test_T = T * (1.0f - h_alpha);
            const bool term = ok & (test_T < MRGS_T_MIN);
            done |= term;
            const bool upd = ok & !(test_T < MRGS_T_MIN);
            const float alpha = upd ? h_alpha : 0.0f;
            const float depth = upd ? h_depth : 1.0f;
            const float w = alpha * T;
            const float A = 1.0f - T;
            const float m_ = mscale * (1.0f - MRGS_NEAR_N * mrgs_rcp(depth));
            const float mm = m_ * m_;
            distortion = fmaf(fmaf(-2.0f * m_, M1, fmaf(mm, A, M2)), w, distortion);
            Dp = fmaf(depth, w, Dp);
            M1 = fmaf(m_, w, M1);
            M2 = fmaf(mm, w, M2);
            const uint32_t contributor = (uint32_t)(base + (int)(s_meta[slot] >> 26) + 1);
            const bool med = upd & (T > 0.5f);
            median_depth = med ? depth : median_depth;
            median_contributor = med ? contributor : median_contributor;
            N0 = fmaf(n0, w, N0); N1 = fmaf(n1, w, N1); N2 = fmaf(n2, w, N2);
            C0 = fmaf(c0, w, C0); C1 = fmaf(c1, w, C1); C2 = fmaf(c2, w, C2);
            if (S_MAX > 0) {
#pragma unroll
                for (int ch = 0; ch < S_MAX; ch++) {
                    const f2 f = mrgs_soa_feature_pair<SF, FV>(sb, ch, p);
                    F[ch] = fmaf(hi ? f.y : f.x, w, F[ch]);
                }
            }
            T = upd ? test_T : T;
            last_contributor = upd ? contributor : last_contributor;
        };

        // (fetching the next step's 18 field pairs one step ahead: 96 VGPRs with spills, 0.237 instead of 0.170 ms)
        for (int p = 0; 2 * p < n; p++) {
            const bool valid_hi = 2 * p + 1 < n;
            // ---- mrgs_intersect on both halves ----
            const f2 Tux = mrgs_soa_pair(sb, 0, p), Tuy = mrgs_soa_pair(sb, 1, p), Tuz = mrgs_soa_pair(sb, 2, p);
            const f2 Tvx = mrgs_soa_pair(sb, 3, p), Tvy = mrgs_soa_pair(sb, 4, p), Tvz = mrgs_soa_pair(sb, 5, p);
            const f2 Twx = mrgs_soa_pair(sb, 6, p), Twy = mrgs_soa_pair(sb, 7, p), Twz = mrgs_soa_pair(sb, 8, p);
            const f2 m2x = mrgs_soa_pair(sb, 9, p), m2y = mrgs_soa_pair(sb, 10, p), opac = mrgs_soa_pair(sb, 11, p);
            const f2 nr0 = mrgs_soa_pair(sb, 12, p), nr1 = mrgs_soa_pair(sb, 13, p), nr2 = mrgs_soa_pair(sb, 14, p);
            const f2 cl0 = mrgs_soa_pair(sb, 15, p), cl1 = mrgs_soa_pair(sb, 16, p), cl2 = mrgs_soa_pair(sb, 17, p);
            const f2 kx = pk_fma(PX, Twx, -Tux), ky = pk_fma(PX, Twy, -Tuy), kz = pk_fma(PX, Twz, -Tuz);
            const f2 lx = pk_fma(PY, Twx, -Tvx), ly = pk_fma(PY, Twy, -Tvy), lz = pk_fma(PY, Twz, -Tvz);
            const f2 ppx = pk_fma(ky, lz, -(kz * ly));
            const f2 ppy = pk_fma(kz, lx, -(kx * lz));
            const f2 ppz = pk_fma(kx, ly, -(ky * lx));
            const f2 inv_pz = rcp2_pz(ppz);
            const f2 sx = ppx * inv_pz, sy = ppy * inv_pz;
            const f2 rho3d = pk_fma(sx, sx, sy * sy);
            const f2 hdx = m2x - PX, hdy = m2y - PY;
            const f2 rho2d = MRGS_FILTER_INV_SQUARE * pk_fma(hdx, hdx, hdy * hdy);
            const f2 rho = min2(rho3d, rho2d);
            const f2 depth3 = pk_fma(sx, Twx, pk_fma(sy, Twy, Twz));
            const f2 h_depth = sel2(rho3d.x <= rho2d.x, rho3d.y <= rho2d.y, depth3, Twz);
            const f2 power = -0.5f * rho;
            const f2 h_G = exp2_pair(power);
            const f2 h_alpha = min2(dup2(0.99f), opac * h_G);
            const bool hit_lo = (ppz.x != 0.0f) & !(h_depth.x < MRGS_NEAR_N) & !(power.x > 0.0f) & !(h_alpha.x < MRGS_ALPHA_MIN);
            const bool hit_hi = (ppz.y != 0.0f) & !(h_depth.y < MRGS_NEAR_N) & !(power.y > 0.0f) & !(h_alpha.y < MRGS_ALPHA_MIN) & valid_hi;
            blend_one(hit_lo, h_alpha.x, h_depth.x, 2 * p, nr0.x, nr1.x, nr2.x, cl0.x, cl1.x, cl2.x, p, false);
            blend_one(hit_hi, h_alpha.y, h_depth.y, 2 * p + 1, nr0.y, nr1.y, nr2.y, cl0.y, cl1.y, cl2.y, p, true);
            if (__builtin_amdgcn_ballot_w64(!done) == 0ull) break;     // (the one-entry kernel finishes its chunk: the rest would test and blend nothing)
        }
        // the backward walks exactly the entries flagged here (see the one-entry kernel); a lane knows the slot of its own entry
        if (base + lane < total) cf[(size_t)(base + lane) * 4] = (uint8_t)(cand_cur ? (contributed >> slot_cur) & 1ull : 0ull);
        // next chunk (one stage buffer)
        {
            stage_chunk(id1, q1);
            id1 = id2; q1 = q2;
            id2 = 0; q2 = 0;
            if (base + 3 * MRGS_CHUNK + lane < total) { id2 = plist[base + 3 * MRGS_CHUNK + lane]; q2 = qm[base + 3 * MRGS_CHUNK + lane]; }
        }
    }
    mrgs_stage_wait();   // do not retire the wave with LDS-DMA still in flight

    if (lane == 0) {
        item_work[tile * 4 + quad] = work;
        if (work_hint != nullptr) work_hint[tile * 4 + quad] = work;
    }
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
            for (int ch = 0; ch < S_MAX; ch++)
                if (ch < S) out_feature[(size_t)ch * HW + pix] = F[ch];
        }
        out_others[pix + 0 * HW] = Dp;
        out_others[pix + 1 * HW] = 1.0f - T;
        out_others[pix + 2 * HW] = N0;
        out_others[pix + 3 * HW] = N1;
        out_others[pix + 4 * HW] = N2;
        out_others[pix + 5 * HW] = median_depth;
        out_others[pix + 6 * HW] = distortion;
    }
}
