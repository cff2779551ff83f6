// mrgs_blend_math.h -- per-(pixel, surfel) arithmetic and list staging shared by the forward and the backward
// blend kernels.
//
// Both kernels must reproduce EXACTLY the same alpha for a pair (the backward rebuilds T by dividing the
// forward's products back out), so the ray/splat intersection lives in one place.  The translation units that
// include this header are compiled with -ffp-contract=off: every fused multiply-add below is written
// explicitly, and oracle/mrgs_oracle.c mirrors the same fused expressions with fmaf(), which makes the
// ill-conditioned part (cross product of the two pixel planes) bit-reproducible between CPU and GPU.
// Two operations are NOT bit-reproducible and differ from the oracle by a few ulp: the reciprocal (v_rcp_f32 here, the
// IEEE quotient 1.0f / x there) and exp (v_exp_f32 with a compensated argument here, the correctly rounded exponential there).
// The VALUES they feed move by ~1e-7; the DECISIONS they feed (alpha >= 1/255, depth >= 0.2, rho3d <= rho2d, T (1 - alpha) < 1e-4,
// T > 0.5) are taken exactly all the same -- "Exact decisions" below.
#pragma once
#include "mrgs_internal.h"

#define MRGS_ALPHA_MIN (1.0f / 255.0f)
#define MRGS_T_MIN 0.0001f
#define MRGS_CHUNK 64   // list entries staged per step = one per lane

// waves of a one-wave-workgroup kernel that one SIMD keeps resident (registers and LDS of that instance), asked once
template <auto KERNEL> static int mrgs_waves_per_simd()
{
    static const int n = [] {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, KERNEL, 64, 0) != hipSuccess || per_cu < 4) return 1;
        return per_cu / 4;
    }();
    return n;
}


// Work-item pull shared by the two blend kernels (queues built by blend_order_kernel, mrgs_sort.hip).  Wave `wv` of XCD
// list `xcd` (blockIdx = wv * 8 + xcd: blockIdx % 8 selects the XCD) takes the next item from the queue of the SIMD it runs
// on.  passes * NQ waves of a list take part -- one per queue SLOT, not one per item.  When the whole launch is resident at
// once (C2 backward: 4.47 items per SIMD, 5 wave slots) the hardware fills every SIMD with `passes` waves, each queue is
// drained by waves of its own SIMD, and the load of a SIMD is the load blend_order_kernel dealt to it.  With one wave per item
// the hardware chose which half of the SIMDs got a fifth wave, those waves took whatever fifth item was left anywhere (after
// a lane-0 walk over the queues that cost them 20-65 us), and the most loaded SIMD carried 1.20x the mean load instead of
// 1.08x; the backward blend is issue-bound per SIMD (finish time of a SIMD against its load: r = 0.93), so that was its
// duration.
// A wave that finds nothing in its own queue (the spare wave of a queue whose slot of the last, partly filled pass is empty;
// or uneven placement) looks through the other queues, 64 per round trip, takes what is left and otherwise retires.  Tickets
// only grow, so a wave that saw every queue exhausted leaves nothing behind, and every item is handed out exactly once.
// The planned spare wave waits ~4 us first when the launch fits the machine (`slots` waves per SIMD hold a whole queue), so
// that the owners of the other queues have taken their tickets; when the launch is larger than the machine waves arrive as
// slots free up, that is the dynamic part of the schedule, and nobody waits.
// Returns 0xFFFFFFFF for "no work", otherwise tile << 2 | quadrant, with the wave's issue priority in bits 29-30.
__device__ __forceinline__ uint32_t mrgs_pull_item(uint32_t* __restrict__ qstate, const uint32_t* __restrict__ cu_state,
                                                   const uint32_t* __restrict__ assign_ws, int ntiles, int xcd, int wv, int lane,
                                                   int slots, unsigned long long* dbg = nullptr)
{
    const uint32_t pw = qstate[MRGS_QS_PASSES + xcd];
    const int passes = (int)(pw & 0xFFFFu), NQ = (int)(pw >> 16);
    if (wv >= passes * NQ) return 0xFFFFFFFFu;
    const int per_list = ((ntiles + 7) >> 3) * 4;
    const uint32_t hw_id = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));              // HW_REG_HW_ID
    const uint32_t xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 7u;           // HW_REG_XCC_ID
    const uint32_t dense = cu_state[MRGS_CS_DENSE + xcc * 256 + mrgs_cu_key(hw_id)];
    const int q0 = (int)((dense * 4u + ((hw_id >> 4) & 3u)) % (uint32_t)NQ);
    const uint32_t* assign = assign_ws + (size_t)xcd * (per_list + MRGS_MAX_SIMD_QUEUES);
    uint32_t* tickets = qstate + MRGS_QS_TICKET + xcd * MRGS_MAX_SIMD_QUEUES;
    uint32_t item = 0xFFFFFFFFu;
    int t = 0;
    if (lane == 0) {
        t = (int)atomicAdd(&tickets[q0], 1u);
        if (t < passes) item = assign[t * NQ + q0];
    }
    item = __builtin_amdgcn_readfirstlane(item);
    t = __builtin_amdgcn_readfirstlane(t);
    if (dbg) { dbg[0] = dbg[2] = wall_clock64(); dbg[1] = 0; }
    if (item != 0xFFFFFFFFu) return item;
    if (t < passes && passes <= slots) __builtin_amdgcn_s_sleep(127);
    for (int base = 1; item == 0xFFFFFFFFu && base < NQ;) {
        const int d = base + lane;
        const int q = d < NQ ? (q0 + d < NQ ? q0 + d : q0 + d - NQ) : q0;
        // (a queue whose slot of the last pass is empty is exhausted one ticket earlier)
        const uint32_t tk = __hip_atomic_load(&tickets[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t last = assign[(passes - 1) * NQ + q];
        const bool has = d < NQ && (int)tk < passes - (last == 0xFFFFFFFFu ? 1 : 0);
        const unsigned long long m = __ballot(has);
        if (dbg) dbg[1] += 1 + (m == 0ull ? 256 : 0);
        if (m == 0ull) { base += 64; continue; }
        const int qs = __builtin_amdgcn_readlane(q, (int)__builtin_ctzll(m));
        if (lane == 0) {
            const int ts = (int)atomicAdd(&tickets[qs], 1u);
            if (ts < passes) item = assign[ts * NQ + qs];
        }
        item = __builtin_amdgcn_readfirstlane(item);   // lost the race for that ticket: look again from the same base
    }
    if (dbg) dbg[2] = wall_clock64();
    return item;
}

__device__ __forceinline__ float mrgs_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// exp(x) for x <= 0 through v_exp_f32: 2^(x*log2e) with the rounding error of the product folded back in
__device__ __forceinline__ float mrgs_exp(float x)
{
    const float L2E_HI = 1.44269502162933349609375f, L2E_LO = 1.925963033500011e-8f, LN2 = 0.693147182464599609375f;
    const float t = x * L2E_HI;
    float e = fmaf(x, L2E_HI, -t);
    e = fmaf(x, L2E_LO, e);
    const float r = __builtin_amdgcn_exp2f(t);
    return fmaf(r, e * LN2, r);
}

// geometry part of the packed record (first three float4 of MRGS_REC): Tu = g0.xyz, Tv = (g0.w, g1.x, g1.y),
// Tw = (g1.z, g1.w, g2.x), mean2D = g2.yz, opacity = g2.w
struct SurfelGeom { float4 g0, g1, g2; };

struct Hit {
    float kx, ky, kz, lx, ly, lz;   // the two pixel planes in splat space (forward.cu:371-372)
    float inv_pz, sx, sy;           // intersection in splat coordinates, s = p.xy / p.z
    float rho3d, rho2d, dx, dy;
    float depth, G, alpha;
    bool use3d;                     // rho3d <= rho2d: the ray/splat hit, not the low-pass disc, gives depth and gradients
};

// ---- Exact decisions ---------------------------------------------------------------------------------------------------------
// The reference divides (p.x / p.z, forward.cu:375) and calls expf; the fast path below multiplies by v_rcp_f32 and goes through
// v_exp_f32.  alpha, depth and rho3d of a pair then sit within a few 1e-7 (relative) of the values the oracle computes with the IEEE
// quotient and the correctly rounded exponential -- invisible in the images, except where such a value is compared with a threshold
// and lands on the other side: a pair blended here and skipped there (or a pixel that terminates / takes its median depth one entry
// earlier) is a 1e-3 ... 1e-2 jump of that pixel.  So the fast path never DECIDES anything it cannot be sure of:
//   * a value within the error band of its threshold (the MRGS_*_LO / _HI / _EPS constants; each band is the worst-case distance of
//     the fast value from the exact one, with slack) makes the pair AMBIGUOUS;
//   * the backward re-evaluates an ambiguous pair on the spot with mrgs_intersect_exact (IEEE quotient, correctly rounded exp, the
//     reference's thresholds) -- per lane, behind a wave-uniform branch that is taken for ~1e-6 of the pairs;
//   * the forward, whose transmittance T is a PRODUCT of (1 - alpha) and therefore carries the accumulated difference of every
//     earlier pair, marks the PIXEL instead, carries on, and renders marked pixels again at the end of the wave with exact arithmetic
//     from the first list entry on (mrgs_render_fwd.hip: redo_pixel), which is the oracle's computation of that pixel.  A pixel is
//     marked for an ambiguous pair it blends and for a transmittance within MRGS_T1_EPS / MRGS_T2_EPS of the 1e-4 / 0.5 tests.
// Every decision of an unmarked pixel is then the one exact arithmetic takes; the values stay fast.
// Bands.  v_rcp_f32 is within 1 ulp, s = p.xy * (1/p.z) then within 1.5 ulp (1.8e-7) of the exact product, rho3d within 4.2e-7
// relative, the exponent -rho/2 within 2.1e-7 rho <= 2.4e-6 for rho <= 2 ln 255 (beyond that alpha < 1/255 whatever the opacity);
// exp and the product with the opacity add ~2.5e-7: alpha within 2.7e-6 relative where it matters -> 4e-6.  depth = s . Tw.xy + Tw.z
// moves by 1.8e-7 (|sx Tw.x| + |sy Tw.y|) -> 1e-4 absolute covers terms up to ~250 (the near plane is at 0.2).  rho3d <= rho2d only
// matters for pairs that pass the alpha test, i.e. rho <= 11.1: 4.2e-7 x 11.1 = 4.7e-6 -> 2e-5 absolute.
#define MRGS_ALPHA_LO (MRGS_ALPHA_MIN * (1.0f - 4e-6f))
#define MRGS_ALPHA_HI (MRGS_ALPHA_MIN * (1.0f + 4e-6f))
#define MRGS_NEAR_LO (MRGS_NEAR_N - 1e-4f)
#define MRGS_NEAR_HI (MRGS_NEAR_N + 1e-4f)
#define MRGS_RHO_EPS 2e-5f
// transmittance: the fast product of (1 - alpha) against the exact one.  T > 0.5: every earlier alpha is below 0.5, the relative
// difference is at most sum alpha_i / (1 - alpha_i) x 2.7e-6 <= 2 ln 2 x 2.7e-6 = 3.7e-6 (measured over 1.4 M pixels: 5.5e-7) -> 4e-6.
// T (1 - alpha) < 1e-4: no constant bounds the difference at the END of a list -- a pair with alpha = 0.99 alone moves the relative
// difference by alpha / (1 - alpha) x 2.5e-7 = 2.5e-5 -- so the forward carries the bound per pixel (MRGS_T1_RUNNING, round 6; rounds
// 4-5 used MRGS_T1_EPS = 1.5e-5 relative, three times the largest difference MEASURED over 1.4 M pixels: a measurement, not a bound).
// With F_n >= |T~_n - T_n| (T~: the fast product; T: the oracle's fp32 product of its exactly evaluated factors):
//     T~_n = fl(T~_{n-1} fl(1 - a~_n)),  T_n = fl(T_{n-1} fl(1 - a_n)),  |a~_n - a_n| <= d_n a_n,  d_n = 2.1e-7 rho_n + 2.5e-7 (above)
//     |T~_n - T_n| <= |T~_{n-1} - T_{n-1}| (1 - a_n) + T_{n-1} d_n a_n + 4 x 2^-24 T_n          (two roundings on either side)
// and, since a_n <= exp(-rho_n / 2) and x exp(-x / 2) <= 2 / e:  d_n a_n <= 1.55e-7 + 2.5e-7 a_n, so that
//     T_{n-1} d_n a_n + 2.4e-7 T_{n-1} (1 - a_n) <= T_{n-1} (1.55e-7 + 2.5e-7 (a_n + (1 - a_n))) = 4.05e-7 T_{n-1}:
//     F_n := F_{n-1} (1 - a~_n) + 4.1e-7 T~_{n-1},   F_0 = 0        (relative form: F_n / T_n = sum_i 4.1e-7 / (1 - a_i))
// first order in the errors; the second-order terms (F d a, the fast values standing in for the exact ones on the right-hand side) are
// covered by the factor MRGS_T1_SLACK = 1.25 -- they are 1e-5 of the first-order ones -- plus 1e-11 absolute (1e-7 of the threshold)
// for the rounding of the recurrence itself (the kernel carries MRGS_T1_SLACK F: the recurrence is linear).  A test value T~ (1 - a~)
// within MRGS_T1_SLACK F_n + 1e-11 of 1e-4 marks the pixel.  No division: a multiplication, a multiply-add, a select and a subtraction
// per blended entry.  Typical pixel (six pairs of alpha ~ 0.8): F / T = 1.2e-5; a pair of alpha 0.99: + 4e-5.  Measured at C3full /
// C2 (MI355X, 100 views each, twice): 110-124 marked pixels a view against 67-84 with the constant band, forward blend 182-185 us
// against 178-179 (C2: 156 against 151); carrying rho per pair (MRGS_T1_RUNNING=2: d_n exactly) marks 89-107 and costs the same.
#ifndef MRGS_T1_RUNNING
#define MRGS_T1_RUNNING 1
#endif
#define MRGS_T_STEP_ERR 4.1e-7f
#define MRGS_T1_SLACK 1.25f
#define MRGS_T1_ABS 1e-11f
#ifndef MRGS_T1_EPS
#define MRGS_T1_EPS (MRGS_T_MIN * 1.5e-5f)      // (the constant band of rounds 4-5: MRGS_T1_RUNNING=0 builds)
#endif
#ifndef MRGS_T2_EPS
#define MRGS_T2_EPS (0.5f * 4e-6f)
#endif

// forward.cu:366-398 / backward.cu:296-328, branch-free: everything is evaluated and the reference's chain of `continue`s collapses
// into flags (a zero p.z gives inf/NaN operands, and every comparison with NaN is false, so such pairs are rejected exactly as the
// reference rejects them).  The fast evaluation: fills h and hands back the three quantities the hit test looks at.
struct HitTest { float ppz, d3, power; };
__device__ __forceinline__ HitTest mrgs_intersect_fast(const SurfelGeom& s, float px, float py, Hit& h)
{
    const float Twx = s.g1.z, Twy = s.g1.w, Twz = s.g2.x;
    h.kx = fmaf(px, Twx, -s.g0.x); h.ky = fmaf(px, Twy, -s.g0.y); h.kz = fmaf(px, Twz, -s.g0.z);
    h.lx = fmaf(py, Twx, -s.g0.w); h.ly = fmaf(py, Twy, -s.g1.x); h.lz = fmaf(py, Twz, -s.g1.y);
    const float ppx = fmaf(h.ky, h.lz, -(h.kz * h.ly));
    const float ppy = fmaf(h.kz, h.lx, -(h.kx * h.lz));
    const float ppz = fmaf(h.kx, h.ly, -(h.ky * h.lx));
    h.inv_pz = mrgs_rcp(ppz);
    h.sx = ppx * h.inv_pz;
    h.sy = ppy * h.inv_pz;
    h.rho3d = fmaf(h.sx, h.sx, h.sy * h.sy);
    h.dx = s.g2.y - px;
    h.dy = s.g2.z - py;
    h.rho2d = MRGS_FILTER_INV_SQUARE * fmaf(h.dx, h.dx, h.dy * h.dy);
    const float rho = fminf(h.rho3d, h.rho2d);
    const float d3 = fmaf(h.sx, Twx, fmaf(h.sy, Twy, Twz));
    h.use3d = h.rho3d <= h.rho2d;
    h.depth = h.use3d ? d3 : Twz;
    const float power = -0.5f * rho;
    h.G = mrgs_exp(power);
    h.alpha = fminf(0.99f, s.g2.w * h.G);
    return HitTest{ppz, d3, power};
}

// "May be a hit": the reference's tests with every threshold moved to the far side of its band and the depth test passed by either
// candidate depth -- a superset of the exact hits; mrgs_hit_decide narrows it down.  (Twz is the same for every lane: its test is scalar.)
__device__ __forceinline__ bool mrgs_intersect(const SurfelGeom& s, float px, float py, Hit& h)
{
    const HitTest t = mrgs_intersect_fast(s, px, py, h);
    return (t.ppz != 0.0f) & (!(t.d3 < MRGS_NEAR_LO) | !(s.g2.x < MRGS_NEAR_LO)) & !(t.power > 0.0f) & !(h.alpha < MRGS_ALPHA_LO);
}

// Second half of the test for a lane mrgs_intersect let through.  Returns the fast decision -- which is the exact one unless
// `ambiguous` comes back set: a value inside its band (the band's lower half counts as a hit meanwhile).
__device__ __forceinline__ bool mrgs_hit_decide(const Hit& h, bool may_hit, bool& ambiguous)
{
    const bool hit = may_hit & !(h.depth < MRGS_NEAR_LO);
    ambiguous = (hit & ((h.alpha < MRGS_ALPHA_HI) | (h.depth < MRGS_NEAR_HI))) | (may_hit & (fabsf(h.rho3d - h.rho2d) < MRGS_RHO_EPS));
    return hit;
}

// The same two steps on LANE MASKS, for the forward blend: every comparison is one v_cmp whose result lands in a scalar register pair
// (the ballot of a single comparison is free), all the logic on them is scalar ALU work next to the vector pipe, and a mask becomes a
// per-lane predicate again with inverse_ballot where a select needs one.  Written with per-lane bools the compiler keeps loop-carried
// flags in vector registers and rebuilds masks with v_cndmask / v_cmp_ne pairs: 17 vector instructions more per blended entry, +40 % on
// the launch (measured when the ambiguity tests went in).  All 64 lanes of the blend waves are active.
#define MRGS_BALLOT(cond) __builtin_amdgcn_ballot_w64(cond)
#define MRGS_LANES(mask) __builtin_amdgcn_inverse_ballot_w64(mask)
__device__ __forceinline__ uint64_t mrgs_intersect_mask(const SurfelGeom& s, float px, float py, Hit& h)
{
    const HitTest t = mrgs_intersect_fast(s, px, py, h);
    const uint64_t tw_ok = s.g2.x < MRGS_NEAR_LO ? 0ull : ~0ull;
    return MRGS_BALLOT(t.ppz != 0.0f) & (~MRGS_BALLOT(t.d3 < MRGS_NEAR_LO) | tw_ok) & ~MRGS_BALLOT(t.power > 0.0f) & ~MRGS_BALLOT(h.alpha < MRGS_ALPHA_LO);
}
__device__ __forceinline__ uint64_t mrgs_hit_decide_mask(const Hit& h, uint64_t may_hit, uint64_t& ambiguous)
{
    const uint64_t hit = may_hit & ~MRGS_BALLOT(h.depth < MRGS_NEAR_LO);
    ambiguous = (hit & (MRGS_BALLOT(h.alpha < MRGS_ALPHA_HI) | MRGS_BALLOT(h.depth < MRGS_NEAR_HI))) |
                (may_hit & MRGS_BALLOT(fabsf(h.rho3d - h.rho2d) < MRGS_RHO_EPS));
    return hit;
}

// exp(x), x <= 0, correctly rounded to fp32: 2^(x log2 e) in double -- argument reduced to [-1/2, 1/2], Taylor series of degree 13
// (remainder 4e-18) -- and one rounding.  The double value is within ~1e-15 of the true one, so the result differs from the
// correctly rounded one only when the true value lies that close to the midpoint of two floats (2e-8 of all arguments); the oracle
// rounds glibc's double exp the same way.
// TABLE: the coefficients come from a constant table through VOLATILE loads instead of literals -- for the branches inside the blend
// loops that run for one pair in a million: plain literals are loop invariants the compiler materialises in front of the loop (ten
// more vector registers held across the backward's entry body, three of them spilled at S = 0).  The redo kernel, where this function
// is the main path, keeps the literals (a round trip to memory per batch otherwise: 24 -> 36 us for the kernel).
__device__ const double mrgs_exp_cr_tab[16] = {1.4426950408889634074, 0.69314718055994530942, 1.0 / 6227020800.0, 1.0 / 479001600.0,
                                               1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0, 1.0 / 40320.0, 1.0 / 5040.0, 1.0 / 720.0,
                                               1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0, 0.5, 1.0, 1.0};
__device__ __forceinline__ float mrgs_exp_cr_finish(double p, double n)
{
    // below 2^-1000 the result is far under the smallest float anyway (and ldexp's int argument stays in range)
    const int e = n < -1000.0 ? -1000 : (int)n;
    return (float)__builtin_ldexp(p, e);
}
template <bool TABLE>
__device__ __forceinline__ float mrgs_exp_cr(float x)
{
    if (TABLE) {
        const volatile double* c = mrgs_exp_cr_tab;
        const double t = (double)x * c[0];
        const double n = __builtin_rint(t);
        const double z = (t - n) * c[1];
        double p = c[2];
#pragma unroll
        for (int i = 3; i < 16; i++) p = __builtin_fma(p, z, c[i]);
        return mrgs_exp_cr_finish(p, n);
    }
    const double t = (double)x * 1.4426950408889634074;
    const double n = __builtin_rint(t);
    const double z = (t - n) * 0.69314718055994530942;
    double p = 1.0 / 6227020800.0;
    p = __builtin_fma(p, z, 1.0 / 479001600.0);
    p = __builtin_fma(p, z, 1.0 / 39916800.0);
    p = __builtin_fma(p, z, 1.0 / 3628800.0);
    p = __builtin_fma(p, z, 1.0 / 362880.0);
    p = __builtin_fma(p, z, 1.0 / 40320.0);
    p = __builtin_fma(p, z, 1.0 / 5040.0);
    p = __builtin_fma(p, z, 1.0 / 720.0);
    p = __builtin_fma(p, z, 1.0 / 120.0);
    p = __builtin_fma(p, z, 1.0 / 24.0);
    p = __builtin_fma(p, z, 1.0 / 6.0);
    p = __builtin_fma(p, z, 0.5);
    p = __builtin_fma(p, z, 1.0);
    p = __builtin_fma(p, z, 1.0);
    return mrgs_exp_cr_finish(p, n);
}

// The pair again as the oracle evaluates it (oracle/mrgs_oracle.c: intersect): same expression tree, the IEEE quotient for 1 / p.z,
// the correctly rounded exponential, the reference's thresholds in the reference's order.  Fills h, returns the hit.
template <bool COLD = false>      // COLD: called from a rarely taken branch of a hot loop (see mrgs_exp_cr)
__device__ __forceinline__ bool mrgs_intersect_exact(const SurfelGeom& s, float px, float py, Hit& h)
{
    const float Twx = s.g1.z, Twy = s.g1.w, Twz = s.g2.x;
    h.kx = fmaf(px, Twx, -s.g0.x); h.ky = fmaf(px, Twy, -s.g0.y); h.kz = fmaf(px, Twz, -s.g0.z);
    h.lx = fmaf(py, Twx, -s.g0.w); h.ly = fmaf(py, Twy, -s.g1.x); h.lz = fmaf(py, Twz, -s.g1.y);
    const float ppx = fmaf(h.ky, h.lz, -(h.kz * h.ly));
    const float ppy = fmaf(h.kz, h.lx, -(h.kx * h.lz));
    const float ppz = fmaf(h.kx, h.ly, -(h.ky * h.lx));
    h.inv_pz = 1.0f / ppz;                      // correctly rounded (the translation unit is not built with fast-math)
    h.sx = ppx * h.inv_pz;
    h.sy = ppy * h.inv_pz;
    h.rho3d = fmaf(h.sx, h.sx, h.sy * h.sy);
    h.dx = s.g2.y - px;
    h.dy = s.g2.z - py;
    h.rho2d = MRGS_FILTER_INV_SQUARE * fmaf(h.dx, h.dx, h.dy * h.dy);
    const float rho = fminf(h.rho3d, h.rho2d);
    h.use3d = h.rho3d <= h.rho2d;
    h.depth = h.use3d ? fmaf(h.sx, Twx, fmaf(h.sy, Twy, Twz)) : Twz;
    const float power = -0.5f * rho;
    h.G = mrgs_exp_cr<COLD>(power);
    h.alpha = fminf(0.99f, s.g2.w * h.G);
    return (ppz != 0.0f) & !(h.depth < MRGS_NEAR_N) & !(power > 0.0f) & !(h.alpha < MRGS_ALPHA_MIN);
}

// 8x8 pixel block owned by a wave: lane -> pixel
__device__ __forceinline__ void mrgs_block_pixel(int block_x, int block_y, int lane, int& pxi, int& pyi)
{
    pxi = block_x * 8 + (lane & 7);
    pyi = block_y * 8 + (lane >> 3);
}

// Block-level cull (MrgsGeomWs::cull, three float4 per surfel, written by preprocess).  A surfel can reach alpha >= 1/255 only
// inside the ellipse d^T [[A,B],[B,C]] d <= 1 around (ex, ey) -- the exact pixel-space level set rho3d <= tau, tau = 2 ln(255 opacity)
// -- or inside the low-pass disc of radius r around mean2D.  A surfel is skipped for a whole pixel block only when the block's
// rectangle of pixel centres misses both, i.e. when every lane would have failed the alpha test anyway: exact per-pixel results are
// unaffected.  The minimum of the (convex) quadratic over the rectangle is taken on its four edges unless the centre lies inside.
// The quadratic is NEVER evaluated as A x^2 + 2 B x y + C y^2: a grazing surfel's ellipse is a needle (det / (A C) down to 1e-8), the
// three terms are ~1e7 each and cancel to ~1, and fp32 returns noise -- round 4's soak found two scenes in 2 000 where a pair with
// alpha = 5.7/255 was culled that way.  On an edge x = X the form is a completed square,
//      q(X, t) = (det / C) X^2 + C (t - t0)^2,   t0 = -(B / C) X,
// two non-negative terms whose coefficients det / C, det / A, B / C, B / A are formed in fp64 by the preprocess; what is left of the
// rounding (the centre and t0 are stored / formed in fp32) is taken off the distances before they are squared, so the value returned
// is a LOWER bound of the true minimum.  A = 0 encodes "not an ellipse, always a candidate".
struct CullConic { float4 a, b, c; };    // a = ex, ey, A, C | b = B/C, B/A, det/C, det/A | c = mean2D.xy, r^2 of the disc, bound of the centre's fp64 error
__device__ __forceinline__ float mrgs_edge_min(float D, float Q, float slope, float X, float lo, float hi, float errX)
{   // lower bound of min over t in [lo, hi] of D X^2 + Q (t - t0)^2, t0 = -slope X; X known to +- errX
    const float Xs = fmaxf(fabsf(X) - errX, 0.0f);
    const float t0 = -slope * X;
    const float tc = fminf(fmaxf(t0, lo), hi);
    const float dt = fmaxf(fabsf(tc - t0) - (fabsf(slope) * errX + 4e-7f * fabsf(t0)), 0.0f);
    return fmaf(D * Xs, Xs, Q * dt * dt);
}
__device__ __forceinline__ bool mrgs_block_may_touch(const CullConic& c, float x0, float y0, float w, float h)
{
    // the centre is an fp32 rounding (2^-24 relative: 6e-8 < 2e-7) of an fp64 value that is itself known to +- c.c.w (the preprocess'
    // running error bound: a needle's centre is a quotient by a determinant that may have lost most of its digits), and it can lie far
    // outside the image: the rectangle grows by that much
    const float ex_err = 2e-7f * fabsf(c.a.x) + 1e-5f + c.c.w, ey_err = 2e-7f * fabsf(c.a.y) + 1e-5f + c.c.w;
    const float dx0 = x0 - c.a.x, dx1 = dx0 + w, dy0 = y0 - c.a.y, dy1 = dy0 + h;
    const float A = c.a.z, C = c.a.w;
    const bool inside = (dx0 <= ex_err) & (dx1 >= -ex_err) & (dy0 <= ey_err) & (dy1 >= -ey_err);
    float g = mrgs_edge_min(c.b.z, C, c.b.x, dx0, dy0 - ey_err, dy1 + ey_err, ex_err);
    g = fminf(g, mrgs_edge_min(c.b.z, C, c.b.x, dx1, dy0 - ey_err, dy1 + ey_err, ex_err));
    g = fminf(g, mrgs_edge_min(c.b.w, A, c.b.y, dy0, dx0 - ex_err, dx1 + ex_err, ey_err));
    g = fminf(g, mrgs_edge_min(c.b.w, A, c.b.y, dy1, dx0 - ex_err, dx1 + ex_err, ey_err));
    const bool ellipse = inside | !(g > 1.01f) | (A == 0.0f);      // (1 %: the fp32 evaluation of rho3d itself is noisy for grazing surfels)
    // disc: squared distance from mean2D to the rectangle
    const float ex = fmaxf(fmaxf(x0 - c.c.x, c.c.x - (x0 + w)), 0.0f), ey = fmaxf(fmaxf(y0 - c.c.y, c.c.y - (y0 + h)), 0.0f);
    const bool disc = fmaf(ex, ex, ey * ey) <= c.c.z;
    return ellipse | disc;
}
__device__ __forceinline__ CullConic mrgs_cull_never()
{
    CullConic c;
    c.a = make_float4(1e30f, 1e30f, 1e30f, 1e30f);
    c.b = make_float4(0.0f, 0.0f, 1e30f, 1e30f);
    c.c = make_float4(1e30f, 0.0f, -1.0f, 0.0f);
    return c;
}
__device__ __forceinline__ CullConic mrgs_cull_load(const float4* __restrict__ rec /* MrgsGeomWs::cull */, uint32_t gid)
{
    CullConic c;
    c.a = rec[(size_t)gid * MRGS_CULL_F4];
    c.b = rec[(size_t)gid * MRGS_CULL_F4 + 1];
    c.c = rec[(size_t)gid * MRGS_CULL_F4 + 2];
    return c;
}

// ---- list staging: asynchronous global -> LDS gather ---------------------------------------------------
// One stage = the records of up to 64 list entries, entry l in slot l.  The five float4 that the blend reads
// (geometry 0..2, appearance 3..4) and the S feature floats are moved by LDS-DMA
// (global_load_lds_dwordx4 / _dword: per-lane global address, destination = wave-uniform LDS base + lane*size),
// so the staged data never occupies VGPRs and the copy for the next chunk is in flight while the current one is
// blended.  Only lanes whose surfel can touch the block issue the copy (EXEC-masked DMA).
template <int SF>
struct StageBuf {
    float4 rec[5][MRGS_CHUNK];
    float feat[SF][MRGS_CHUNK];   // FV = false: [channel][slot]; FV = true: the same bytes hold float4 [channel / 4][slot]
    uint32_t id[MRGS_CHUNK];
};

// feature channel ch of staged entry j
template <bool FV, int SF>
__device__ __forceinline__ float mrgs_staged_feature(const StageBuf<SF>& sb, int ch, int j)
{
    if (FV) return reinterpret_cast<const float*>(&sb.feat[0][0])[((ch >> 2) * MRGS_CHUNK + j) * 4 + (ch & 3)];
    return sb.feat[ch][j];
}

// FV ("feature vectors"): the feature row of a gaussian is S = S_MAX floats with S % 4 == 0, i.e. 16-byte aligned 16-byte
// pieces: S/4 DMA instructions of 16 bytes per lane instead of S of 4 bytes (each DMA instruction of a wave gathers from up
// to 64 different cache lines, which is what it costs), and the blend reads four channels with one ds_read_b128.
// (TAG: the instantiating kernel's live-channel count -- two kernels sharing one specialization of this function trip the host pass of
//  clang over its device-only builtins: the second use reports a substitution failure)
// ROW: floats per feature row of the FV layout (S_MAX unless the rows carry more than is staged: the "pgsr" rows of twelve floats, of which
// eight are staged and the ninth rides in the surfel record)
template <int S_MAX, int SF, bool FV, int TAG = 0, int ROW = S_MAX>
__device__ __forceinline__ void mrgs_stage_async(StageBuf<SF>& dst, const float4* __restrict__ rec, const float* __restrict__ features,
                                                 int S, uint32_t gid, bool pred)
{
    if (pred) {
        const float4* src = rec + (size_t)gid * MRGS_REC_F4;
        __builtin_amdgcn_global_load_lds(src + 0, &dst.rec[0][0], 16, 0, 0);
        __builtin_amdgcn_global_load_lds(src + 1, &dst.rec[1][0], 16, 0, 0);
        __builtin_amdgcn_global_load_lds(src + 2, &dst.rec[2][0], 16, 0, 0);
        __builtin_amdgcn_global_load_lds(src + 3, &dst.rec[3][0], 16, 0, 0);
        __builtin_amdgcn_global_load_lds(src + 4, &dst.rec[4][0], 16, 0, 0);
        if (S_MAX > 0) {
            if (FV) {
                const float4* fsrc = reinterpret_cast<const float4*>(features + (size_t)gid * ROW);
                float4* fdst = reinterpret_cast<float4*>(&dst.feat[0][0]);
#pragma unroll
                for (int q = 0; q < S_MAX / 4; q++) __builtin_amdgcn_global_load_lds(fsrc + q, fdst + q * MRGS_CHUNK, 16, 0, 0);
            } else {
                const float* fsrc = features + (size_t)gid * S;
#pragma unroll
                for (int ch = 0; ch < S_MAX; ch++)
                    if (ch < S) __builtin_amdgcn_global_load_lds(fsrc + ch, &dst.feat[ch][0], 4, 0, 0);
            }
        }
    }
}

// all LDS-DMA of this wave has landed (the DMA is tracked by vmcnt; nothing else orders a ds_read behind it)
__device__ __forceinline__ void mrgs_stage_wait()
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// wave64 sum with DPP; every lane of the wave must be active.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float mrgs_dpp_add(float v)
{
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
    return v + __int_as_float(moved);
}
