// Ray tracing of 2D-gaussian surfels (second half): the replacement of the un-vendored OptiX extension
// `diff_surfel_tracing` behind HardwareRendering (gaussian_renderer/optix_utils.py:14-271), whose callers trace the mirror rays of
// every pixel through the surfel set (render_indirect / render_surfel_with_envgs / render_surfel2, envgs_renderer.py:461-804).
//
// What the reference fixes and what it leaves open.  optix_utils.py fixes the primitive -- each surfel is the quad
// mean +- 3 s_u r_u +- 3 s_v r_v as two triangles (get_disks :36-66), rebuilt every training iteration (:68-82) --, the ray
// convention (direction NOT normalised, depth = ray parameter, :121-123), the inputs (means, opacities, scales / rotations, shs or
// colours, two "others" channels :173-183) and the outputs rgb, dpt, acc, norm, dist, aux, mid, wet (:185-197, 218-233).  The
// arithmetic between them lives in the missing extension: PARITY IS UNPINNED for this file.  It is defined here as the 2DGS
// compositing of the vendored rasterizer applied along a ray (forward.cu:366-420 with the ray parameter in place of the view depth):
//   hit of surfel p:  t = n.(m - o) / n.d,  x = o + t d,  u = r_u.(x - m) / s_u,  v = r_v.(x - m) / s_v,  t > 0, |u|,|v| <= 3
//   G = exp(-(u^2 + v^2) / 2),  alpha = min(0.99, opacity G),  skipped when alpha < 1/255
//   hits in order of (t, index);  w = alpha T;  the hit that would take T below 1e-4 is not blended and ends the ray
//   rgb = sum w c + T bg, dpt = sum w t, acc = sum w, norm = sum w n (turned against the ray), aux = sum w others,
//   dist = sum_i w_i (t_i^2 A_i + M2_i - 2 t_i M1_i)  (A, M1, M2: sums of w, w t, w t^2 over the hits before i), wet[p] += w.
// oracle/surfel_trace_oracle.py states the same definition densely (every ray against every surfel) in torch; its autograd is the
// check of the hand-written backward below.
//
// MI355X design.  No RT cores; what the machine has is 64 lanes per wave, scalar registers for what a wave shares, and LDS.
//  * BUILD ON THE DEVICE, every iteration (the reference rebuilds per iteration too): quad boxes + scene bounds (per-block partials,
//    folded by the last block: no same-address atomics) -> 30-bit Morton keys -> the radix sort of mrgs_sort.hip -> a complete 64-ARY
//    tree over the sorted order, one kernel per level (3-4 levels): node n of level l owns nodes 64 n .. 64 n + 63 of level l-1,
//    level 0 owns surfels.  A node is six rows of 64 floats: child c's box is read by LANE c.  ~0.2 ms for 300 k surfels.
//  * 64 RAYS, ONE WALK (st_gather_wide): a wave takes an 8x8 block of rays.  While they run together (directions within ~11 degrees,
//    origins close) the wave walks the tree once for all of them: lane c tests child c against the BEAM of the rays (bounds on offset
//    and slope in the rays' own frame), a ballot names the children to enter, the nearest first (wave-min of the near depths).  At the
//    bottom lane c tests surfel c's own square against the beam; the surviving surfels are visited one by one, their record broadcast
//    from lane c (v_readlane), every lane evaluating the exact hit for ITS ray and inserting into its own sorted 16-entry buffer
//    (LDS, [slot][lane]).  Blocks that do not run together split into 4x4 quadrants, then 2x2 groups; such a packet spreads the
//    exact tests over all 64 lanes, 64 / R candidates a step (st_gather_group).  The wave-wide minima / maxima / sums of the walk
//    (nearest child, far bound, beam) are DPP butterflies, not __shfl_xor (= ds_bpermute) ones.
//  * ONE RAY, ONE WAVE (st_trace_lone_rays): what is left over walks with the lanes turned sideways -- 64 children, then the 64
//    surfels of a leaf group, against the one ray -- instead of alone in a lane (a dependent ~1 us gather per step).
//  * 16-NEAREST PASSES: a ray gathers its 16 nearest not-yet-blended hits, blends them front to back, and continues behind the last
//    one while the buffer came back full and the ray is not saturated; the walk's far bound closes when every ray of the packet has
//    a full buffer.
//  * BACKWARD FRONT TO BACK as well: with the forward's totals at hand, d/d alpha_i = T_i q_i - (Q - Q_i) / (1 - alpha_i)
//    (q: the pixel gradient contracted with hit i's contribution, Q its weighted sum over all hits, Q_i over hits <= i), so the
//    backward re-walks exactly the forward's passes and needs no per-ray hit storage.
// What was measured on the way (800x800 primary rays through the 300 k shell scene, forward): a 4-ary tree walked per lane with the
// stack in registers 42 ms; + leaf-ordered records, 8x8 ray blocks 29; the same tree walked per wave (scalar node loads, every lane
// tests its ray) 10; 64-wide nodes with the beam test 4.8.  Mirror rays off a rendered view: 16 ms with a 25-degree cone, 9.5 with
// 11 degrees, 3.9 with the left-over rays traced one per wave.
// Compiled with -ffp-contract=off: forward and backward evaluate the hit expression identically.
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "mrgs_internal.h"

namespace {

#ifndef ST_GROUP_GATHER
#define ST_GROUP_GATHER 1                 // packets smaller than the wave test 64 / R candidates per step (st_gather_group); 0: one per step (A/B)
#endif
#ifndef ST_REPLAY_MERGE
#define ST_REPLAY_MERGE 1                 // the replaying backward merges neighbouring lanes that hold the same surfel BEFORE the LDS collection; 0: after (A/B)
#endif
#ifndef ST_REC_REFETCH
#define ST_REC_REFETCH 0                  // st_gather_group: candidate records re-read from memory instead of ds_bpermute from the holding lane (A/B)
#endif
#ifndef ST_NO_WET
#define ST_NO_WET 0                       // developer A/B: 1 = the forward leaves the per-surfel weight sums out (wrong `wet`, timing only)
#endif
#ifndef ST_LONE_IN_BLOCK
#define ST_LONE_IN_BLOCK 1                // backward: a single ray whose record suffices is replayed by its block's wave (LDS gradient table); 0: st_replay_lone_rays4 / a wave of its own (A/B)
#endif
#ifndef ST_REPLAY4
#define ST_REPLAY4 1                      // backward: single rays whose record suffices are replayed four to a wave (st_replay_lone_rays4); 0: one per wave (A/B)
#endif
#ifndef ST_NO_STEAL
#define ST_NO_STEAL 0
#endif
#ifndef ST_GLOBAL_ORDER
#define ST_GLOBAL_ORDER 0
#endif
#ifndef ST_REST_SCHED_STATIC_REGION
#define ST_REST_SCHED_STATIC_REGION 0
#endif
#ifndef ST_REST_SCHED
#define ST_REST_SCHED 0                   // second launch of the forward: 0 = items by ticket, own region first; 1 = every list strided over all waves (A/B)
#endif
#ifndef ST_REST_LONE_FIRST
#define ST_REST_LONE_FIRST 0
#endif
#ifndef ST_FWD_WAVES
#define ST_FWD_WAVES 4                    // waves per SIMD the forward walking kernels are compiled for (register budget 512 / that)
#endif
constexpr int ST_K = 16;                  // hits gathered per pass
constexpr int ST_THREADS = 256;
constexpr int ST_MAX_PASSES = 256;        // 4096 hits per ray at most
constexpr float ST_EXTENT = 3.0f;         // optix_utils.py:44 (3-sigma quad)
constexpr int ST_AABB_BLOCKS = 256;

constexpr int ST_REC_STATIC = 3;          // chunks of the hit record every wave owns; further passes draw from a shared pool
#ifndef ST_LONE_CAP_TILES
#define ST_LONE_CAP_TILES 8             // rays per block of rays the single rays' record has room for (x n_tiles / 8 per region); 2 was too few: the regions differ by 2x
#endif
#ifndef ST_LONE_PASSES
#define ST_LONE_PASSES 8
#endif
constexpr int ST_LONE_REC_PASSES = ST_LONE_PASSES;   // passes of a ray traced alone whose 16 sorted hits are kept for the backward (more: it walks again)
constexpr int ST_REC_PASSES = 16;         // passes of a wave the record can hold (beyond: the backward traces again)
constexpr uint32_t ST_REC_NONE = 0xFFFFFFFFu;
// Header of the two per-region lists (StArgs::lone_list / defer_list): ST_LIST_HDR words in front of the entries; region r's count sits
// at word 64 r and its ticket (second launch of the forward) at word 64 r + 32 -- every counter in a 128-byte line of its own.  Round 5
// first kept the sixteen words side by side: every listing wave of the first launch and every ticket of the second then met on ONE line
// at the memory-side atomic unit (~11 ns per operation, tools/ubench), and 20 000 tickets of two operations each put 0.4 ms of
// serialised waiting into a 0.6 ms launch (measured: tickets without stealing 1.24 ms against 0.86 ms for computed indices).
constexpr int ST_LIST_HDR = 512;
#define ST_CNT(hdr, r) ((hdr)[64u * (r)])
#define ST_TKT(hdr, r) ((hdr) + 64u * (r) + 32u)
constexpr int SW_MAX_LEVELS = 4;          // 64^4 surfels
struct StWide {
    int32_t n;                            // levels; level 0 holds surfels (sorted position 64 g + c), the root is node 0 of level n-1
    int32_t off[SW_MAX_LEVELS];           // first node of level l
    int32_t cnt[SW_MAX_LEVELS];
    int32_t total;
};

StWide st_wide(int64_t P)
{
    StWide w;
    std::memset(&w, 0, sizeof(w));
    int64_t c = (P + 63) / 64;
    if (c < 1) c = 1;
    int32_t off = 0;
    for (int l = 0; l < SW_MAX_LEVELS; ++l) {
        w.off[l] = off; w.cnt[l] = (int32_t)c; off += (int32_t)c; w.n = l + 1;
        if (c == 1) break;
        c = (c + 63) / 64;
    }
    w.total = off;
    return w;
}

struct BuildWs {
    size_t aabb, key0, key1, val0, val1, sortws, bounds, total, zero_from, zero_bytes;
};

BuildWs st_build_ws(int64_t P)
{
    BuildWs w;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o = mrgs_align_up(o + bytes, 256); return at; };
    w.aabb = take((size_t)P * 24);
    w.key0 = take((size_t)P * 4); w.key1 = take((size_t)P * 4);
    w.val0 = take((size_t)P * 4); w.val1 = take((size_t)P * 4);
    w.zero_from = o;
    w.sortws = take(mrgs_sort_ws_words(P) * 4);
    w.bounds = take(64 + ST_AABB_BLOCKS * 6 * 4);   // 6 ordered-uint extrema, [8] sort error flag, [9] ticket, [16..] per-block partial extrema
    w.zero_bytes = o - w.zero_from;
    w.total = o;
    return w;
}

__device__ __forceinline__ uint32_t ord_f(float f)          // order-preserving float -> uint
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unord_f(uint32_t u)
{
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}

__device__ __forceinline__ uint32_t ld_agent_u(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent_u(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ uint32_t wave_max_u(uint32_t v)
{
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, s));
    return v;
}

// quad boxes + scene bounds.  bounds[k] = max ord(hi_k), bounds[3 + k] = max ~ord(lo_k).  Same-address atomics serialise at L2
// (one per wave cost 0.32 ms at 300 k surfels): every block leaves one partial row instead and the last block to finish folds them.
__global__ __launch_bounds__(256) void st_aabb_kernel(int P, const float* __restrict__ verts, float* __restrict__ aabb, uint32_t* __restrict__ partial,
                                                      uint32_t* __restrict__ ticket, uint32_t* __restrict__ bounds)
{
    __shared__ uint32_t red[4][6];
    __shared__ bool last;
    uint32_t ext[6] = {0, 0, 0, 0, 0, 0};
    for (int p = blockIdx.x * 256 + threadIdx.x; p < P; p += gridDim.x * 256) {
        const float4* q = reinterpret_cast<const float4*>(verts + (size_t)p * 12);
        const float4 a = q[0], b = q[1], c = q[2];
        const float v[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w};
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        bool ok = true;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            ok = ok && (fabsf(v[i]) < 1e30f);            // false for NaN / inf
            lo[i % 3] = fminf(lo[i % 3], v[i]);
            hi[i % 3] = fmaxf(hi[i % 3], v[i]);
        }
        float* o = aabb + (size_t)p * 6;
        if (ok) {
            o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = hi[0]; o[4] = hi[1]; o[5] = hi[2];
#pragma unroll
            for (int k = 0; k < 3; ++k) { ext[k] = max(ext[k], ord_f(hi[k])); ext[3 + k] = max(ext[3 + k], ~ord_f(lo[k])); }
        } else {
            o[0] = NAN; o[1] = o[2] = o[3] = o[4] = o[5] = 0.f;          // never hit, sorted to key 0
        }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const uint32_t m = wave_max_u(ext[k]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = m;
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = threadIdx.x;
        st_agent_u(partial + blockIdx.x * 6 + k, max(max(red[0][k], red[1][k]), max(red[2][k], red[3][k])));
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    __syncthreads();
    if (!last) return;
    __threadfence();
    // thread b folds block b's row (gridDim.x <= 256 = blockDim.x); six dependent loops over 256 agent-scope loads cost 50 us
    uint32_t mine[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) mine[k] = threadIdx.x < gridDim.x ? ld_agent_u(partial + threadIdx.x * 6 + k) : 0u;
    __syncthreads();                       // `red` is reused
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const uint32_t m = wave_max_u(mine[k]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = m;
    }
    __syncthreads();
    if (threadIdx.x < 6) bounds[threadIdx.x] = max(max(red[0][threadIdx.x], red[1][threadIdx.x]), max(red[2][threadIdx.x], red[3][threadIdx.x]));
}

// Bits of the Morton key per axis.  8: a 24-bit key = THREE 8-bit radix passes instead of four (13 us per traced view); 16.7 M cells for
// leaf groups of 64 surfels is far finer than the groups (measured: the walks take the same time as with 10 bits per axis).
#ifndef ST_MORTON_AXIS_BITS
#define ST_MORTON_AXIS_BITS 8
#endif
__device__ __forceinline__ uint32_t spread10(uint32_t v)      // 10 bits -> every third bit
{
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__global__ __launch_bounds__(256) void st_morton_kernel(int P, const float* __restrict__ aabb, const uint32_t* __restrict__ bounds,
                                                        uint32_t* __restrict__ key, uint32_t* __restrict__ val)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const float* b = aabb + (size_t)p * 6;
    uint32_t code = 0;
    if (b[0] == b[0]) {
        uint32_t q[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float smin = unord_f(~bounds[3 + k]), smax = unord_f(bounds[k]);
            const float ext = smax - smin;
            const float c = 0.5f * (b[k] + b[3 + k]);
            const float f = ext > 0.f ? (c - smin) / ext : 0.f;
            q[k] = (uint32_t)fminf(fmaxf(f * (float)(1 << ST_MORTON_AXIS_BITS), 0.f), (float)((1 << ST_MORTON_AXIS_BITS) - 1));
        }
        code = (spread10(q[0]) << 2) | (spread10(q[1]) << 1) | spread10(q[2]);
    }
    key[p] = code;
    val[p] = (uint32_t)p;
}

// ---- tracing ------------------------------------------------------------------------------------------------------------

struct StArgs {
    int64_t n_rays;
    const float* ray_o; const float* ray_d;
    const float4* geom;                   // [P][4]: (m.xyz, a.x) (a.yz, b.xy) (b.z, n.xyz) (opacity, -, -, -);  a = r_u / s_u, b = r_v / s_v
    const float4* geom_leaf;              // the same records in leaf order: the four surfels of level-0 node i at [4 i .. 4 i + 3]
    int32_t ray_width;                    // > 0: rays form rows of this length and a wave takes an 8x8 block of them
    int32_t packets;                      // waves whose rays run together walk the wide hierarchy as one (st_gather_wide)
    StWide wide;
    // the forward's record of what every wave gathered, pass by pass (ids, [slot][lane] like the LDS buffer): the backward replays it
    // instead of walking the hierarchy again.  hdr[0] chunks drawn from the pool, hdr[1] overflow flag (then the backward traces).
    uint32_t *rec_hdr, *rec_chunks, *rec_arena;
    unsigned long long* lone_rec;         // [8][lone_cap][ST_LONE_REC_PASSES][ST_K] sorted (t, id) keys of the rays traced one per wavefront, or nullptr
    uint32_t lone_cap;                    // per region
    uint32_t rec_pool;                    // chunks in the shared pool (behind the n_tiles * ST_REC_STATIC owned ones)
    uint32_t rec_static;                  // n_tiles * ST_REC_STATIC: where the pool starts
    uint32_t n_tiles;                     // waves of the first launch (one 8x8 block of rays each)
    // Both lists are kept per REGION: block b of a launch runs on XCD b % 8 (observed dispatch order -- used for speed only, nothing below is
    // wrong under another placement), the first launch gives XCD x every eighth 64 x 64 patch of the rays (st_tile_of_wave), and what it
    // lists goes to sub-list x, which the second launch hands to the blocks with b % 8 == x again: the waves resident on one XCD at a time
    // trace neighbouring rays and share the subtrees and leaf records their L2 holds (round 5; before, consecutive blocks of four 8x8 ray
    // blocks went round the eight L2s, an XCD's resident waves were spread over half the image and every L2 saw the whole hierarchy:
    // 4.2 GB fetched by the second launch of a C4-size view).
    uint32_t* defer_list;                 // header (ST_LIST_HDR words: ST_CNT / ST_TKT), then [x * defer_cap ..] (tile << 5 | packet): the packets traced by the second launch, one per wave
    uint32_t defer_cap;                   // per region
    uint32_t* lone_list;                  // header, then [x * lone_list_cap ..] indices of the rays that walk alone (behind the per-ray state)
    uint32_t lone_list_cap;               // per region (= the rays of a region: never full)
    uint32_t* lone_slot;                  // [n_rays]: where the first launch listed a ray that walks alone, region << 28 | index in the region's list (written for exactly those rays)
    uint32_t region_blocks;               // blocks of the first launch per region
    float cone, cone_quad, cone_group;    // 1 - cos of the half-angle within which the directions of a block / quadrant / 2x2 group must stay
    const float4* attr;                   // [P][2]: (rgb, others.x) (others.y, -, -, -)
    float bg[3];
    float *rgb, *dpt, *acc, *norm, *dist, *aux, *wet, *state;     // state [n_rays][4]: M2, T_final, hits blended, passes
    const float *g_rgb, *g_dpt, *g_acc, *g_norm, *g_dist, *g_aux;
    float *g_geom, *g_attr, *g_ray_o, *g_ray_d;                   // [P][16], [P][8] (zeroed by the entry point), [n_rays][3] x 2
};

struct StHit { float t, u, v, G, alpha, den; bool ok; };

__device__ __forceinline__ StHit st_hit(const float4 g0, const float4 g1, const float4 g2, const float opacity,
                                        float ox, float oy, float oz, float dx, float dy, float dz)
{
    StHit h;
    const float mx = g0.x, my = g0.y, mz = g0.z, ax = g0.w, ay = g1.x, az = g1.y, bx = g1.z, by = g1.w, bz = g2.x, nx = g2.y, ny = g2.z, nz = g2.w;
    h.den = nx * dx + ny * dy + nz * dz;
    const float num = nx * (mx - ox) + ny * (my - oy) + nz * (mz - oz);
    h.t = num / h.den;
    const float px = (ox + h.t * dx) - mx, py = (oy + h.t * dy) - my, pz = (oz + h.t * dz) - mz;
    h.u = ax * px + ay * py + az * pz;
    h.v = bx * px + by * py + bz * pz;
    h.G = expf(-0.5f * (h.u * h.u + h.v * h.v));
    h.alpha = fminf(0.99f, opacity * h.G);
    h.ok = h.t > 0.0f && fabsf(h.u) <= ST_EXTENT && fabsf(h.v) <= ST_EXTENT && h.alpha >= 1.0f / 255.0f && h.t < 1e30f;   // false for NaN
    return h;
}

struct StProf { int nodes, tests, lanes; unsigned t_fetch, t_cand; };     // developer counters (-DST_PROFILE writes them into `state`)

// wave-wide reductions; the result is handed back through v_readfirstlane so that the compiler keeps it in a scalar register
// (after the butterfly every lane holds the same value, which it cannot know: the beam's ~40 numbers would sit in vector registers)
__device__ __forceinline__ float st_uniform(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }
// DPP butterflies: xor 1, xor 2 inside the quads, rotations by 4 and 8 inside the 16-lane rows, then lane 15 / lane 31 of the rows before
// into the rows behind (row_bcast) -- lane 63 ends up with the result of all 64.  Six dependent VALU instructions; the __shfl_xor
// butterfly these replace goes through ds_bpermute, six dependent ~120-cycle LDS round trips (measured: a packet wave spends 500 of these
// reductions a view -- nearest child, far bound, beams, packet tests -- i.e. ~40 % of its 470 us in them).  Masked-out rows take `old`.
#define ST_DPP(OLD, V, CTRL, RM) __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(OLD), __float_as_int(V), CTRL, RM, 0xf, false))
__device__ __forceinline__ float st_lane63(float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63)); }
__device__ __forceinline__ float wave_sum_f(float v)
{
    v += ST_DPP(0.0f, v, 0xb1, 0xf); v += ST_DPP(0.0f, v, 0x4e, 0xf); v += ST_DPP(0.0f, v, 0x124, 0xf); v += ST_DPP(0.0f, v, 0x128, 0xf);
    v += ST_DPP(0.0f, v, 0x142, 0xa); v += ST_DPP(0.0f, v, 0x143, 0xc);
    return st_lane63(v);
}
__device__ __forceinline__ float wave_max_f(float v)
{
    v = fmaxf(v, ST_DPP(v, v, 0xb1, 0xf)); v = fmaxf(v, ST_DPP(v, v, 0x4e, 0xf)); v = fmaxf(v, ST_DPP(v, v, 0x124, 0xf)); v = fmaxf(v, ST_DPP(v, v, 0x128, 0xf));
    v = fmaxf(v, ST_DPP(v, v, 0x142, 0xa)); v = fmaxf(v, ST_DPP(v, v, 0x143, 0xc));
    return st_lane63(v);
}
__device__ __forceinline__ float wave_min_f(float v)
{
    v = fminf(v, ST_DPP(v, v, 0xb1, 0xf)); v = fminf(v, ST_DPP(v, v, 0x4e, 0xf)); v = fminf(v, ST_DPP(v, v, 0x124, 0xf)); v = fminf(v, ST_DPP(v, v, 0x128, 0xf));
    v = fminf(v, ST_DPP(v, v, 0x142, 0xa)); v = fminf(v, ST_DPP(v, v, 0x143, 0xc));
    return st_lane63(v);
}

// ---- the wave-wide hierarchy -------------------------------------------------------------------------------------------------
// A second tree over the same Morton order for waves whose 64 rays run close together (an 8x8 block of mirror rays off a smooth
// surface): 64 children per node, stored one child per LANE (six rows of 64 floats), three to four levels for 10^5..10^7 surfels.
// The wave walks it ONCE for all its rays: a node's 64 child boxes arrive with six coalesced loads, lane c tests child c against the
// BEAM of the wave's rays (interval arithmetic over the rays' origins and inverse directions: conservative for every ray), one
// ballot names the children to enter.  Ten dependent round trips per candidate (in a 4-ary tree) become three, and the 4 x 64
// ray-box tests of a packet walking such a tree become one test per lane.  At the bottom the surviving surfels are visited one by one: their
// record comes from a wave-uniform address (scalar loads, the next one in flight while this one is tested), every lane evaluates
// the exact hit for its own ray and inserts into its own buffer.  No ordering of the walk: the far bound of a packet only closes
// when all its lanes have full buffers, which a measurement with 32-entry buffers showed to be rare.
// boxes [node][6][64] (lo.xyz, hi.xyz rows), vmask [node]: the children that exist and hold a finite box
__global__ __launch_bounds__(64) void st_wide_level_kernel(int level, int n_children_total, StWide w, const uint32_t* __restrict__ sorted,
                                                           const float* __restrict__ aabb, float* __restrict__ boxes,
                                                           unsigned long long* __restrict__ vmask)
{
    const int node = blockIdx.x, c = threadIdx.x;
    float* mine = boxes + (size_t)(w.off[level] + node) * 384;
    const int child = 64 * node + c;
    bool ok = child < n_children_total;
    float b[6] = {0, 0, 0, 0, 0, 0};
    if (level == 0) {
        if (ok) {
            const float* src = aabb + (size_t)sorted[child] * 6;
#pragma unroll
            for (int k = 0; k < 6; ++k) b[k] = src[k];
            ok = b[0] == b[0];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float pad = 1e-5f * fmaxf(fabsf(b[k]), fabsf(b[3 + k])) + 1e-30f;
                b[k] -= pad; b[3 + k] += pad;
            }
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) mine[k * 64 + c] = ok ? b[k] : 0.f;
    } else {
        if (ok) {
#pragma unroll
            for (int k = 0; k < 6; ++k) b[k] = mine[k * 64 + c];       // written by the launch of the level below
            ok = b[0] <= b[3];                                           // a subtree without a valid surfel arrives inverted
        }
    }
    const unsigned long long m = __ballot(ok);
    if (c == 0) vmask[w.off[level] + node] = m;
    if (level + 1 < w.n) {
        float* up = boxes + (size_t)(w.off[level + 1] + (node >> 6)) * 384;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float lo = wave_min_f(ok ? b[k] : INFINITY), hi = wave_max_f(ok ? b[3 + k] : -INFINITY);
            if (c == 0) { up[k * 64 + (node & 63)] = lo; up[(3 + k) * 64 + (node & 63)] = hi; }
        }
    }
}

// The beam of a wave's rays in its own frame: m = mean direction, (e1, e2) across it, origin at the mean ray origin.  Ray i is the line
// (u, v)(s) = (u0_i + k1_i s, v0_i + k2_i s) over the depth s along m, so the beam's cross-section at depth s lies inside
// [min u0 + min(k1 s), max u0 + max(k1 s)] x (same in v): interval arithmetic again, but on slopes and offsets that are SMALL for rays
// that run together (in world axes the same bound multiplies O(1) numbers and lets almost every box through -- measured: 2.5x more
// candidates than a walk in which every lane tests its own ray, 20x on diverging mirror rays).
struct StBeam {
    float oc[3], m[3], e1[3], e2[3], am[3], a1[3], a2[3];       // frame and |components| (for the extent of a box along each frame axis)
    float u0min, u0max, v0min, v0max, k1min, k1max, k2min, k2max;
};

__device__ __forceinline__ StBeam st_make_beam(bool on, float ox, float oy, float oz, float dx, float dy, float dz, float& s0, float& dm)
{
    StBeam B;
    const float cnt = wave_sum_f(on ? 1.0f : 0.0f);
    const float il = on ? 1.0f / sqrtf(dx * dx + dy * dy + dz * dz) : 0.0f;
    float mx = wave_sum_f(dx * il), my = wave_sum_f(dy * il), mz = wave_sum_f(dz * il);
    const float ml = 1.0f / sqrtf(mx * mx + my * my + mz * mz);
    mx *= ml; my *= ml; mz *= ml;
    B.oc[0] = wave_sum_f(on ? ox : 0.0f) / cnt; B.oc[1] = wave_sum_f(on ? oy : 0.0f) / cnt; B.oc[2] = wave_sum_f(on ? oz : 0.0f) / cnt;
    // e1 = m x (the axis m is least aligned with), e2 = m x e1
    const float ax = fabsf(mx), ay = fabsf(my), az = fabsf(mz);
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (ax <= ay && ax <= az) qx = 1.f; else if (ay <= az) qy = 1.f; else qz = 1.f;
    float e1x = my * qz - mz * qy, e1y = mz * qx - mx * qz, e1z = mx * qy - my * qx;
    const float el = 1.0f / sqrtf(e1x * e1x + e1y * e1y + e1z * e1z);
    e1x *= el; e1y *= el; e1z *= el;
    const float e2x = my * e1z - mz * e1y, e2y = mz * e1x - mx * e1z, e2z = mx * e1y - my * e1x;
    B.m[0] = mx; B.m[1] = my; B.m[2] = mz; B.e1[0] = e1x; B.e1[1] = e1y; B.e1[2] = e1z; B.e2[0] = e2x; B.e2[1] = e2y; B.e2[2] = e2z;
#pragma unroll
    for (int k = 0; k < 3; ++k) { B.am[k] = fabsf(B.m[k]); B.a1[k] = fabsf(B.e1[k]); B.a2[k] = fabsf(B.e2[k]); }
    dm = dx * mx + dy * my + dz * mz;                                           // > 0 for every ray of a wave that runs together
    const float rx = ox - B.oc[0], ry = oy - B.oc[1], rz = oz - B.oc[2];
    s0 = rx * mx + ry * my + rz * mz;                                           // depth of the ray's origin
    const float tau = -s0 / dm;                                                 // ray parameter where it crosses the plane s = 0
    const float px = rx + tau * dx, py = ry + tau * dy, pz = rz + tau * dz;
    const float u0 = px * e1x + py * e1y + pz * e1z, v0 = px * e2x + py * e2y + pz * e2z;
    const float k1 = (dx * e1x + dy * e1y + dz * e1z) / dm, k2 = (dx * e2x + dy * e2y + dz * e2z) / dm;
    B.u0min = wave_min_f(on ? u0 : INFINITY); B.u0max = wave_max_f(on ? u0 : -INFINITY);
    B.v0min = wave_min_f(on ? v0 : INFINITY); B.v0max = wave_max_f(on ? v0 : -INFINITY);
    B.k1min = wave_min_f(on ? k1 : INFINITY); B.k1max = wave_max_f(on ? k1 : -INFINITY);
    B.k2min = wave_min_f(on ? k2 : INFINITY); B.k2max = wave_max_f(on ? k2 : -INFINITY);
    return B;
}

// May ANY ray of the beam touch the box lo..hi at a depth not in front of s_prev?  Conservative: a ray that meets the box has a point
// in it, whose depth lies in [sa, sb] and whose lateral coordinates lie within the box's extent about its centre.
__device__ __forceinline__ bool st_beam_box(const StBeam& B, const float lo[3], const float hi[3], float s_prev, float s_far, float& near_depth)
{
    float sc = 0.f, uc = 0.f, vc = 0.f, rs = 0.f, r1 = 0.f, r2 = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float c = 0.5f * (lo[k] + hi[k]) - B.oc[k], h = 0.5f * (hi[k] - lo[k]);
        sc += c * B.m[k]; uc += c * B.e1[k]; vc += c * B.e2[k];
        rs += h * B.am[k]; r1 += h * B.a1[k]; r2 += h * B.a2[k];
    }
    const float sa = sc - rs, sb = sc + rs;
    const float umax = B.u0max + fmaxf(fmaxf(B.k1min * sa, B.k1min * sb), fmaxf(B.k1max * sa, B.k1max * sb));
    const float umin = B.u0min + fminf(fminf(B.k1min * sa, B.k1min * sb), fminf(B.k1max * sa, B.k1max * sb));
    const float vmax = B.v0max + fmaxf(fmaxf(B.k2min * sa, B.k2min * sb), fmaxf(B.k2max * sa, B.k2max * sb));
    const float vmin = B.v0min + fminf(fminf(B.k2min * sa, B.k2min * sb), fminf(B.k2max * sa, B.k2max * sb));
    const float eps = 1e-4f * (fabsf(sc) + rs + fabsf(uc) + r1 + fabsf(vc) + r2 + fabsf(umax) + fabsf(umin) + fabsf(vmax) + fabsf(vmin)) + 1e-30f;
    near_depth = sa - eps;
    return uc - r1 <= umax + eps && uc + r1 >= umin - eps && vc - r2 <= vmax + eps && vc + r2 >= vmin - eps && sb + eps >= s_prev && sa - eps <= s_far;
}

// The same question for a surfel itself (level 0): centre = its mean, extent along a frame axis e = ext (|A.e| + |B.e|) with A = s_u r_u,
// B = s_v r_v (the support function of the square |u|, |v| <= ext), ext = min(3, radius at which alpha falls below 1/255).
__device__ __forceinline__ bool st_beam_surfel(const StBeam& B, const float4 g0, const float4 g1, const float4 g2, float opacity, float s_prev, float s_far)
{
    const float ax = g0.w, ay = g1.x, az = g1.y, bx = g1.z, by = g1.w, bz = g2.x;
    const float ia = 1.0f / (ax * ax + ay * ay + az * az), ib = 1.0f / (bx * bx + by * by + bz * bz);      // a = r_u / s_u -> s_u r_u = a / (a.a)
    const float vis = 255.0f * opacity;
    if (!(vis > 1.0f)) return false;                                               // alpha < 1/255 everywhere
    const float ext = fminf(ST_EXTENT, sqrtf(2.0f * logf(vis)) * 1.0001f);
    const float cx = g0.x - B.oc[0], cy = g0.y - B.oc[1], cz = g0.z - B.oc[2];
    const float sc = cx * B.m[0] + cy * B.m[1] + cz * B.m[2], uc = cx * B.e1[0] + cy * B.e1[1] + cz * B.e1[2], vc = cx * B.e2[0] + cy * B.e2[1] + cz * B.e2[2];
    const float rs = ext * (fabsf(ax * B.m[0] + ay * B.m[1] + az * B.m[2]) * ia + fabsf(bx * B.m[0] + by * B.m[1] + bz * B.m[2]) * ib);
    const float r1 = ext * (fabsf(ax * B.e1[0] + ay * B.e1[1] + az * B.e1[2]) * ia + fabsf(bx * B.e1[0] + by * B.e1[1] + bz * B.e1[2]) * ib);
    const float r2 = ext * (fabsf(ax * B.e2[0] + ay * B.e2[1] + az * B.e2[2]) * ia + fabsf(bx * B.e2[0] + by * B.e2[1] + bz * B.e2[2]) * ib);
    const float sa = sc - rs, sb = sc + rs;
    const float umax = B.u0max + fmaxf(fmaxf(B.k1min * sa, B.k1min * sb), fmaxf(B.k1max * sa, B.k1max * sb));
    const float umin = B.u0min + fminf(fminf(B.k1min * sa, B.k1min * sb), fminf(B.k1max * sa, B.k1max * sb));
    const float vmax = B.v0max + fmaxf(fmaxf(B.k2min * sa, B.k2min * sb), fmaxf(B.k2max * sa, B.k2max * sb));
    const float vmin = B.v0min + fminf(fminf(B.k2min * sa, B.k2min * sb), fminf(B.k2max * sa, B.k2max * sb));
    const float eps = 1e-4f * (fabsf(sc) + rs + fabsf(uc) + r1 + fabsf(vc) + r2 + fabsf(umax) + fabsf(umin) + fabsf(vmax) + fabsf(vmin)) + 1e-30f;
    return uc - r1 <= umax + eps && uc + r1 >= umin - eps && vc - r2 <= vmax + eps && vc + r2 >= vmin - eps && sb + eps >= s_prev && sa - eps <= s_far;
}

// (Measured and dropped, round 3: touching the cache lines of the next remaining sibling with one LDS-DMA instruction whenever the walk
// descends into a child -- 12 lines of boxes or the 32 lines of a leaf group, no registers -- so that the visit after this one would find
// them cached: st_trace_kernel<0> 570 -> 590 us, st_trace_rest_kernel<0> 750 -> 789 us; the walk's visits are not waiting for misses that a
// one-visit lead removes, and the extra instruction costs six more spilled registers in kernels that are at their budget.)
__device__ __forceinline__ int st_gather_wide(const StWide& W, const float* __restrict__ boxes, const unsigned long long* __restrict__ vmask,
                                              const float4* __restrict__ leaf, uint32_t (*kb_id)[ST_THREADS], float (*kb_t)[ST_THREADS], int tid,
                                              float ox, float oy, float oz, float dx, float dy, float dz, float ivx, float ivy, float ivz,
                                              float prev_t, uint32_t prev_id, bool first_pass, bool on, StProf& prof)
{
    int n = 0;
    if (__ballot(on) == 0) return 0;
    const int lane = tid & 63;
    float s0, dm;
    const StBeam B = st_make_beam(on, ox, oy, oz, dx, dy, dz, s0, dm);
    const float s_prev = wave_min_f(on ? s0 + prev_t * dm : INFINITY);      // nothing in front of every lane's last blended hit is needed
    // behind s_far no lane needs anything: every active lane's buffer is full and its last entry lies in front (depth = s0 + t dm)
    float s_far = INFINITY;
    unsigned long long mask[SW_MAX_LEVELS] = {0, 0, 0, 0};
    int node[SW_MAX_LEVELS] = {0, 0, 0, 0};
    float near1 = 0.f, near2 = 0.f, near3 = 0.f;                             // per lane: near depth of its child at levels 1..3
    float4 rec0 = make_float4(0, 0, 0, 0), rec1 = rec0, rec2 = rec0, rec3 = rec0;   // lane c: record of surfel c of the current group
    int l = W.n - 1;
    bool fresh = true;                                                       // node[l] has not been tested yet
    for (;;) {
        if (fresh) {
            ++prof.nodes;
            const int nd = __builtin_amdgcn_readfirstlane(W.off[l] + node[l]);
            bool h;
            if (l == 0) {                                                     // the surfels themselves, one per lane
                const float4* g = leaf + ((size_t)node[0] * 64 + lane) * 4;
                rec0 = g[0]; rec1 = g[1]; rec2 = g[2]; rec3 = g[3];
                h = st_beam_surfel(B, rec0, rec1, rec2, rec3.x, s_prev, s_far);
            } else {
                const float* bx = boxes + (size_t)nd * 384 + lane;
                const float lo[3] = {bx[0], bx[64], bx[128]}, hi[3] = {bx[192], bx[256], bx[320]};
                float nd_depth;
                h = st_beam_box(B, lo, hi, s_prev, s_far, nd_depth);
                if (l == 1) near1 = nd_depth; else if (l == 2) near2 = nd_depth; else near3 = nd_depth;
            }
            mask[l] = __ballot(h) & vmask[nd];
            fresh = false;
        }
        if (mask[l] == 0) {
            if (l == W.n - 1) break;
            ++l;
            continue;
        }
        if (l > 0) {                                                          // nearest remaining child first
            const float mine = l == 1 ? near1 : l == 2 ? near2 : near3;
            const float key = ((mask[l] >> lane) & 1ull) ? mine : INFINITY;
            const float nearest = wave_min_f(key);
            if (nearest > s_far) { mask[l] = 0; continue; }                   // and everything else at this node lies behind it
            const int c = __builtin_ctzll(__ballot(key == nearest));
            mask[l] &= ~(1ull << c);
            node[l - 1] = node[l] * 64 + c;
            --l;
            fresh = true;
            continue;
        }
        // level 0: the surviving surfels of this group, one by one.  Lane c holds surfel c's record since the beam test: it is
        // broadcast from there (v_readlane, no memory round trip -- a record fetched per candidate cost ~1 us of latency each)
        unsigned long long m = mask[0];
        mask[0] = 0;
        while (m) {
            const int c = __builtin_ctzll(m);
            m &= m - 1;
#if ST_REC_REFETCH
            const float4* gq = leaf + ((size_t)__builtin_amdgcn_readfirstlane(node[0]) * 64 + (size_t)c) * 4;      // (wave-uniform address)
            const float4 h0 = gq[0], h1 = gq[1], h2 = gq[2], h3 = gq[3];
#else
            auto bc = [c](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), c)); };
            const float4 h0 = make_float4(bc(rec0.x), bc(rec0.y), bc(rec0.z), bc(rec0.w)), h1 = make_float4(bc(rec1.x), bc(rec1.y), bc(rec1.z), bc(rec1.w));
            const float4 h2 = make_float4(bc(rec2.x), bc(rec2.y), bc(rec2.z), bc(rec2.w));
            float4 h3;
            h3.x = bc(rec3.x); h3.y = bc(rec3.y);
#endif
#ifdef ST_PROFILE
            ++prof.tests;
#endif
            if (on) {
                const uint32_t id = __float_as_uint(h3.y);
                const StHit h = st_hit(h0, h1, h2, h3.x, ox, oy, oz, dx, dy, dz);
                bool take = h.ok && (first_pass || h.t > prev_t || (h.t == prev_t && id > prev_id));
#ifdef ST_PROFILE
                prof.lanes += __popcll(__ballot(h.ok));
#endif
                if (take && n == ST_K) {
                    const float lt = kb_t[ST_K - 1][tid];
                    take = h.t < lt || (h.t == lt && id < kb_id[ST_K - 1][tid]);
                }
                if (take) {     // sorted insertion (a register-resident buffer with branch-free insertion was measured: 0.72 instead of
                                // 1.0 us per candidate for the wave, but 243 VGPRs = one wave per SIMD, slower overall)
                    int pos = n < ST_K ? n++ : ST_K - 1;
                    while (pos > 0) {
                        const float pt = kb_t[pos - 1][tid];
                        const uint32_t pid = kb_id[pos - 1][tid];
                        if (pt < h.t || (pt == h.t && pid < id)) break;
                        kb_t[pos][tid] = pt; kb_id[pos][tid] = pid;
                        --pos;
                    }
                    kb_t[pos][tid] = h.t; kb_id[pos][tid] = id;
                }
            }
        }
        s_far = wave_max_f(on ? (n == ST_K ? s0 + kb_t[ST_K - 1][tid] * dm : INFINITY) : -INFINITY) * (1.0f + 1e-5f) + 1e-30f;
    }
    return n;
}

// The same walk for a packet SMALLER than the wave -- a 4x4 quadrant (R = 16 rays) or a 2x2 group (R = 4) --, with the exact tests at
// the bottom spread over all 64 lanes: lane l = g R + i tests the g-th surviving surfel of the leaf group against ray i of the packet, so
// one step evaluates 64 / R candidates (measured on the mirror rays of a rendered view, 95 % of whose waves are such packets: a candidate
// hits 6 of a packet's ~20 rays, i.e. in st_gather_wide's layout two lanes in three idled through every hit evaluation, and a candidate
// cost the wave 1.5 us).  The surfel's record crosses from the lane that loaded it by ds_bpermute, the ray's data from the lane that
// owns it (once per walk).  Hits go into the owning lane's LDS column as before; lanes that hit for the SAME ray take turns, ordered by
// their rank among them (per-ray ranks from one ballot), so a step costs as many insertion rounds as its busiest ray has hits.  The
// buffers end up holding the same 16 nearest (t, id) keys in the same order whatever the order of insertion: outputs and the record are
// those of the one-candidate-a-step walk bit for bit.  kb_n: entries in every column (the owners' `n` of st_gather_wide, shared here).
__device__ __forceinline__ int st_gather_group(const StWide& W, const float* __restrict__ boxes, const unsigned long long* __restrict__ vmask,
                                               const float4* __restrict__ leaf, uint32_t (*kb_id)[ST_THREADS], float (*kb_t)[ST_THREADS],
                                               uint32_t* kb_n, int tid, int pk, bool tiled, float ox, float oy, float oz, float dx, float dy, float dz,
                                               float prev_t, uint32_t prev_id, bool first_pass, bool on, StProf& prof)
{
    if (__ballot(on) == 0) return 0;
    const int lane = tid & 63, wbase = tid & ~63;
    float s0, dm;
    const StBeam B = st_make_beam(on, ox, oy, oz, dx, dy, dz, s0, dm);
    const float s_prev = wave_min_f(on ? s0 + prev_t * dm : INFINITY);
    float s_far = INFINITY;
    // the gather layout: lane = grp * R + i; `own` = the lane that owns ray i of packet pk (st_assign_packets' numbering)
    const bool quad_pk = pk <= 4;
    const int shift = quad_pk ? 4 : 2, G = 64 >> shift;
    const int grp = lane >> shift, i = lane & ((1 << shift) - 1);
    int own;
    if (quad_pk) {
        const int q = pk - 1;
        own = tiled ? 8 * (4 * (q >> 1) + (i >> 2)) + 4 * (q & 1) + (i & 3) : 16 * q + i;
    } else {
        const int q = (pk - 5) >> 2, g2 = (pk - 5) & 3;
        own = tiled ? 8 * (4 * (q >> 1) + 2 * (g2 >> 1) + (i >> 1)) + 4 * (q & 1) + 2 * (g2 & 1) + (i & 1) : 16 * q + 4 * g2 + i;
    }
    auto from = [](int src_lane, float v) { return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_lane << 2, __builtin_bit_cast(int, v))); };
    const float rox = from(own, ox), roy = from(own, oy), roz = from(own, oz), rdx = from(own, dx), rdy = from(own, dy), rdz = from(own, dz);
    const float rprev_t = from(own, prev_t);
    const uint32_t rprev_id = (uint32_t)__builtin_amdgcn_ds_bpermute(own << 2, (int)prev_id);
    const bool ron = __builtin_amdgcn_ds_bpermute(own << 2, (int)on) != 0;
    const int col = wbase + own;
    const unsigned long long sib = (quad_pk ? 0x0001000100010001ull : 0x1111111111111111ull) << i;      // the lanes that share my ray
    const unsigned long long below = (1ull << lane) - 1ull;
    kb_n[tid] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    unsigned long long mask[SW_MAX_LEVELS] = {0, 0, 0, 0};
    int node[SW_MAX_LEVELS] = {0, 0, 0, 0};
    float near1 = 0.f, near2 = 0.f, near3 = 0.f;
    float4 rec0 = make_float4(0, 0, 0, 0), rec1 = rec0, rec2 = rec0, rec3 = rec0;
    int l = W.n - 1;
    bool fresh = true;
    for (;;) {
        if (fresh) {
#ifdef ST_PROFILE
            const unsigned long long tf0 = wall_clock64();
#endif
            ++prof.nodes;
            const int nd = __builtin_amdgcn_readfirstlane(W.off[l] + node[l]);
            bool h;
            if (l == 0) {
                const float4* g = leaf + ((size_t)node[0] * 64 + lane) * 4;
                rec0 = g[0]; rec1 = g[1]; rec2 = g[2]; rec3 = g[3];
                h = st_beam_surfel(B, rec0, rec1, rec2, rec3.x, s_prev, s_far);
            } else {
                const float* bx = boxes + (size_t)nd * 384 + lane;
                const float lo[3] = {bx[0], bx[64], bx[128]}, hi[3] = {bx[192], bx[256], bx[320]};
                float nd_depth;
                h = st_beam_box(B, lo, hi, s_prev, s_far, nd_depth);
                if (l == 1) near1 = nd_depth; else if (l == 2) near2 = nd_depth; else near3 = nd_depth;
            }
            mask[l] = __ballot(h) & vmask[nd];
            fresh = false;
#ifdef ST_PROFILE
            prof.t_fetch += (unsigned)(wall_clock64() - tf0);
#endif
        }
        if (mask[l] == 0) {
            if (l == W.n - 1) break;
            ++l;
            continue;
        }
        if (l > 0) {                                                          // nearest remaining child first
            const float mine = l == 1 ? near1 : l == 2 ? near2 : near3;
            const float key = ((mask[l] >> lane) & 1ull) ? mine : INFINITY;
            const float nearest = wave_min_f(key);
            if (nearest > s_far) { mask[l] = 0; continue; }
            const int c = __builtin_ctzll(__ballot(key == nearest));
            mask[l] &= ~(1ull << c);
            node[l - 1] = node[l] * 64 + c;
            --l;
            fresh = true;
            continue;
        }
        unsigned long long m = mask[0];
        mask[0] = 0;
#ifdef ST_PROFILE
        const unsigned long long tc0 = wall_clock64();
#endif
        while (m) {
            // the next G survivors, one per lane group
            int c = 0;
            bool have = false;
            for (int k = 0; k < G && m; ++k) {
                const int ck = __builtin_ctzll(m);
                m &= m - 1;
                if (grp == k) { c = ck; have = true; }
#ifdef ST_PROFILE
                ++prof.tests;
#endif
            }
#if ST_REC_REFETCH
            // the candidate's record from memory again (the leaf group's 4 KB were fetched a moment ago: L1 / L2) instead of from the lane
            // that holds it: thirteen ds_bpermute less per step and rec0..3 are dead across the candidate loop
            const float4* gq = leaf + ((size_t)node[0] * 64 + (size_t)c) * 4;
            const float4 h0 = gq[0], h1 = gq[1], h2 = gq[2], h3q = gq[3];
            const float opac = h3q.x;
            const uint32_t id = __float_as_uint(h3q.y);
#else
            const float4 h0 = make_float4(from(c, rec0.x), from(c, rec0.y), from(c, rec0.z), from(c, rec0.w));
            const float4 h1 = make_float4(from(c, rec1.x), from(c, rec1.y), from(c, rec1.z), from(c, rec1.w));
            const float4 h2 = make_float4(from(c, rec2.x), from(c, rec2.y), from(c, rec2.z), from(c, rec2.w));
            const float opac = from(c, rec3.x);
            const uint32_t id = __float_as_uint(from(c, rec3.y));
#endif
            const StHit h = st_hit(h0, h1, h2, opac, rox, roy, roz, rdx, rdy, rdz);
            const bool take = have && ron && h.ok && (first_pass || h.t > rprev_t || (h.t == rprev_t && id > rprev_id));
            const unsigned long long takers = __ballot(take);
#ifdef ST_PROFILE
            prof.lanes += __popcll(__ballot(have && ron && h.ok));
#endif
            if (takers == 0) continue;
            const int rank = __popcll(takers & sib & below);                 // takers of my ray in the lane groups before mine
            for (int k = 0;; ++k) {
                const bool now = take && rank == k;
                if (__ballot(now) == 0) break;                                // (ranks are dense per ray)
                if (now) {
                    uint32_t n = kb_n[col];
                    bool ins = true;
                    if (n == ST_K) {
                        const float lt = kb_t[ST_K - 1][col];
                        ins = h.t < lt || (h.t == lt && id < kb_id[ST_K - 1][col]);
                    }
                    if (ins) {
                        int pos = n < ST_K ? (int)n++ : ST_K - 1;
                        while (pos > 0) {
                            const float pt = kb_t[pos - 1][col];
                            const uint32_t pid = kb_id[pos - 1][col];
                            if (pt < h.t || (pt == h.t && pid < id)) break;
                            kb_t[pos][col] = pt; kb_id[pos][col] = pid;
                            --pos;
                        }
                        kb_t[pos][col] = h.t; kb_id[pos][col] = id;
                        kb_n[col] = n;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
#ifdef ST_PROFILE
        prof.t_cand += (unsigned)(wall_clock64() - tc0);
#endif
        s_far = wave_max_f(on ? (kb_n[tid] == ST_K ? s0 + kb_t[ST_K - 1][tid] * dm : INFINITY) : -INFINITY) * (1.0f + 1e-5f) + 1e-30f;
    }
    return on ? (int)kb_n[tid] : 0;
}

// Rays "run together" when their directions stay within a cone of ~5.7 degrees about their mean (1 - cos <= 0.005 for a block or a quadrant, 0.001 = 2.6 degrees for a 2x2 group: wider beams of grazing rays sweep thousands of surfels; measured on mirror rays off a rendered view: 16 ms at 25 degrees, 9.5 at 11 before and 3.65 -> 1.9 at 11 -> 4.4 after the rays left over got waves of their own) and their origins within 2 % of the
// scene's extent of their centre.  `on` selects the rays asked about; the answer is wave-uniform.
__device__ __forceinline__ bool st_run_together(float extent, float cone, bool on, float ox, float oy, float oz, float dx, float dy, float dz)
{
    const float cnt = wave_sum_f(on ? 1.0f : 0.0f);
    if (cnt < 1.0f) return false;
    const float il = on ? 1.0f / sqrtf(dx * dx + dy * dy + dz * dz) : 0.0f;
    const float ux = dx * il, uy = dy * il, uz = dz * il;
    float mx = wave_sum_f(ux), my = wave_sum_f(uy), mz = wave_sum_f(uz);
    const float ml = sqrtf(mx * mx + my * my + mz * mz);
    if (!(ml > 0.5f * cnt)) return false;
    mx /= ml; my /= ml; mz /= ml;
    const float worst = wave_max_f(on ? 1.0f - (ux * mx + uy * my + uz * mz) : 0.0f);
    const float cx = wave_sum_f(on ? ox : 0.0f) / cnt, cy = wave_sum_f(on ? oy : 0.0f) / cnt, cz = wave_sum_f(on ? oz : 0.0f) / cnt;
    const float spread = wave_max_f(on ? fmaxf(fabsf(ox - cx), fmaxf(fabsf(oy - cy), fabsf(oz - cz))) : 0.0f);
    return worst <= cone && spread <= 0.02f * extent;
}

// Packet of every lane: 0 = the whole wave, 1..4 = its quadrant (4x4 rays of the 8x8 block, or 16 consecutive rays), 5..20 = its
// 2x2 group inside the quadrant, -1 = the ray walks alone.  The coarsest grouping whose rays run together wins; `present` gets one bit
// per packet in use.
__device__ __forceinline__ int st_assign_packets(const StArgs& A, const float* __restrict__ boxes, const unsigned long long* __restrict__ vmask, bool on,
                                                 int lane, float ox, float oy, float oz, float dx, float dy, float dz, uint32_t& present)
{
    present = 0;
    if (!A.packets || __ballot(on) == 0) return -1;
    const int root = A.wide.off[A.wide.n - 1];                                // lane c: child c of the root of the wide hierarchy
    float extent = 0.0f;
    if (A.wide.n > 1) {
        const bool there = (vmask[root] >> lane) & 1ull;
        const float* bx = boxes + (size_t)root * 384 + lane;
        for (int k = 0; k < 3; ++k)
            extent = fmaxf(extent, wave_max_f(there ? bx[(3 + k) * 64] : -INFINITY) - wave_min_f(there ? bx[k * 64] : INFINITY));
    } else {
        extent = INFINITY;                                                    // <= 64 surfels: no scale to compare origins with
    }
    if (st_run_together(extent, A.cone, on, ox, oy, oz, dx, dy, dz)) { present = 1u; return 0; }
    const bool tiled = A.ray_width > 0;
    const int quad = tiled ? ((lane >> 2) & 1) + 2 * (lane >> 5) : lane >> 4;
    const int sub = tiled ? ((lane >> 1) & 1) + 2 * ((lane >> 4) & 1) : (lane >> 2) & 3;
    int mine = -1;
    for (int q = 0; q < 4; ++q) {
        const bool in_q = on && quad == q;
        if (__ballot(in_q) == 0) continue;
        if (st_run_together(extent, A.cone_quad, in_q, ox, oy, oz, dx, dy, dz)) {
            if (in_q) mine = 1 + q;
            present |= 1u << (1 + q);
            continue;
        }
        for (int g = 0; g < 4; ++g) {
            const bool in_g = in_q && sub == g;
            if (__ballot(in_g) == 0) continue;
            if (st_run_together(extent, A.cone_group, in_g, ox, oy, oz, dx, dy, dz)) {
                if (in_g) mine = 5 + 4 * q + g;
                present |= 1u << (5 + 4 * q + g);
            }
        }
    }
    return mine;
}

// The surfel records in leaf (= sorted) order: a wide group's 64 candidates are one 4 KB run, the surfel's index rides in the record's
// spare word.
__global__ __launch_bounds__(256) void st_leaf_order_kernel(int n_slots, int P, const uint32_t* __restrict__ perm, const float4* __restrict__ geom,
                                                            float4* __restrict__ geom_leaf)
{
    const int i = blockIdx.x * 256 + threadIdx.x;            // one thread per 16 bytes
    if (i >= n_slots * 4) return;
    const int slot = i >> 2, part = i & 3;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (slot < P) {
        const uint32_t id = perm[slot];
        v = geom[(size_t)id * 4 + part];
        if (part == 3) v.y = __uint_as_float(id);
    }
    geom_leaf[i] = v;
}

// MODE 0: forward (walks, blends, records what it gathered); 1: backward that walks again (no record, or it overflowed);
// 2: backward that replays the forward's record -- same blend arithmetic on the same ids in the same order, no hierarchy.
#ifndef ST_TAB_ENTRIES
#define ST_TAB_ENTRIES 96
#endif
constexpr int ST_TAB = ST_TAB_ENTRIES;                // surfels a wave of the replaying backward accumulates in LDS before its gradients go out
constexpr int ST_TAB_WORDS = ST_TAB * 19; // 18 gradient terms + the surfel's index per entry
constexpr uint32_t ST_REC_DEFERRED = 0xFFFFFFFEu;   // in the last slot of a block's row of the chunk table: its packets went to the second launch

// One 8x8 block of rays (64 consecutive rays when the rays are no image).  only_packet < 0: the first launch -- the block walks as one
// packet if its rays run together; otherwise its packets (quadrants, 2x2 groups) are LISTED for the second launch, one wave each,
// because a wave that walks 16 packets one after the other lasts 16 times as long as its neighbours and the launch as long as that
// wave (measured: total work of 0.2 ms of perfectly spread waves, 3.5 ms launch).  only_packet >= 1: that packet of the block.
template <int MODE>
__device__ __forceinline__ void st_trace_tile(const StArgs& A, const float4* __restrict__ leaf_ro, const float* __restrict__ wide_boxes,
                                              const unsigned long long* __restrict__ wide_vmask, uint32_t (*kb_id)[ST_THREADS],
                                              float (*kb_t)[ST_THREADS], uint32_t* kb_n, uint32_t* tab, int tid, int64_t tile, int only_packet, uint32_t rec_row,
                                              uint32_t region)
{
    constexpr bool BWD = MODE != 0;
    int64_t r = tile * 64 + (tid & 63);
    if (A.ray_width > 0) {                 // 8x8 blocks of neighbouring rays per wave: neighbours walk the same nodes
        const int64_t tiles_x = (A.ray_width + 7) >> 3, rows = A.n_rays / A.ray_width;
        const int64_t px = (tile % tiles_x) * 8 + (tid & 7), py = (tile / tiles_x) * 8 + ((tid >> 3) & 7);
        r = (px < A.ray_width && py < rows) ? py * A.ray_width + px : A.n_rays;
    }
#ifdef ST_PROFILE
    const unsigned long long prof_t0 = wall_clock64();
#endif
    const bool exists = r < A.n_rays;
    if (!exists) r = 0;                    // the lane stays for the wave-wide steps and neither blends nor writes
    const float ox = A.ray_o[3 * r], oy = A.ray_o[3 * r + 1], oz = A.ray_o[3 * r + 2];
    const float dx = A.ray_d[3 * r], dy = A.ray_d[3 * r + 1], dz = A.ray_d[3 * r + 2];
    const float ivx = 1.0f / dx, ivy = 1.0f / dy, ivz = 1.0f / dz;

    float T = 1.0f, C[3] = {0, 0, 0}, D = 0, Aw = 0, N[3] = {0, 0, 0}, X[2] = {0, 0}, dist = 0, M1 = 0, M2 = 0;
    int blended = 0, passes = 0;
    // backward: totals of the forward and the contraction of the pixel gradient with them
    float gc[3] = {0, 0, 0}, gd = 0, ga = 0, gn[3] = {0, 0, 0}, gx[2] = {0, 0}, gdist = 0;
    float fA = 0, fM1 = 0, fM2 = 0, fT = 0, Qtot = 0, Qpre = 0, bgdot = 0;
    float go[3] = {0, 0, 0}, gdir[3] = {0, 0, 0};
    if (BWD) {
        // (an upstream gradient nobody supplied is a null pointer = zeros: no zero-filled maps on the way in)
        if (A.g_rgb) { gc[0] = A.g_rgb[3 * r]; gc[1] = A.g_rgb[3 * r + 1]; gc[2] = A.g_rgb[3 * r + 2]; }
        if (A.g_dpt) gd = A.g_dpt[r];
        if (A.g_acc) ga = A.g_acc[r];
        if (A.g_dist) gdist = A.g_dist[r];
        if (A.g_norm) { gn[0] = A.g_norm[3 * r]; gn[1] = A.g_norm[3 * r + 1]; gn[2] = A.g_norm[3 * r + 2]; }
        if (A.g_aux) { gx[0] = A.g_aux[2 * r]; gx[1] = A.g_aux[2 * r + 1]; }
        fA = A.acc[r]; fM1 = A.dpt[r]; fM2 = A.state[4 * r]; fT = A.state[4 * r + 1];
        bgdot = gc[0] * A.bg[0] + gc[1] * A.bg[1] + gc[2] * A.bg[2];
        Qtot = gc[0] * (A.rgb[3 * r] - fT * A.bg[0]) + gc[1] * (A.rgb[3 * r + 1] - fT * A.bg[1]) + gc[2] * (A.rgb[3 * r + 2] - fT * A.bg[2])
             + gd * fM1 + ga * fA + gn[0] * A.norm[3 * r] + gn[1] * A.norm[3 * r + 1] + gn[2] * A.norm[3 * r + 2]
             + gx[0] * A.aux[2 * r] + gx[1] * A.aux[2 * r + 1] + gdist * 2.0f * (fA * fM2 - fM1 * fM1);
    }

    float prev_t = 0.0f;
    uint32_t prev_id = 0;
    // a ray without a direction (or with a non-finite one) would visit every node: it sees the background
    bool done = !(exists && fabsf(ox) < 1e30f && fabsf(oy) < 1e30f && fabsf(oz) < 1e30f && fabsf(dx) < 1e30f && fabsf(dy) < 1e30f &&
                  fabsf(dz) < 1e30f && (dx != 0.0f || dy != 0.0f || dz != 0.0f));
    const bool no_ray = done;                       // sees the background; written by the first launch
    uint32_t packets_present = 0;
    int packet;
    if (only_packet >= 1) {
        // a listed packet: the first launch found its rays running together (and its block / quadrant not): its rays are the valid rays of
        // its lanes -- no need to repeat the ~200 wave-wide reductions of the assignment for the whole block
        const int ln = tid & 63;
        const bool tiled = A.ray_width > 0;
        const int quad = tiled ? ((ln >> 2) & 1) + 2 * (ln >> 5) : ln >> 4;
        const int sub = tiled ? ((ln >> 1) & 1) + 2 * ((ln >> 4) & 1) : (ln >> 2) & 3;
        const bool member = only_packet <= 4 ? quad == only_packet - 1 : (quad == ((only_packet - 5) >> 2) && sub == ((only_packet - 5) & 3));
        packet = (member && !done) ? only_packet : -1;
    } else {
        packet = st_assign_packets(A, wide_boxes, wide_vmask, !done, tid & 63, ox, oy, oz, dx, dy, dz, packets_present);
    }
    // rays that run with nobody are only listed here; the second launch gives each a wave of its own
    const bool lone = !done && packet < 0;
    bool deferred = false;                          // this block's packets are the second launch's
    uint32_t my_row = rec_row;                      // the row of the chunk table this lane's record lives in
    if (only_packet < 0) {
        const uint32_t dm = packets_present & ~1u;
        if (MODE == 0) {
            const unsigned long long lm = __ballot(lone);
            if (lm) {
                const int first = __builtin_ctzll(lm), lane = tid & 63;
                uint32_t base = 0;
                if (lane == first) base = atomicAdd(&ST_CNT(A.lone_list, region), (uint32_t)__popcll(lm));
                base = (uint32_t)__builtin_amdgcn_readlane((int)base, first);
                if (lone) {
                    const uint32_t at = base + (uint32_t)__popcll(lm & ((1ull << lane) - 1ull));
                    A.lone_list[ST_LIST_HDR + (size_t)region * A.lone_list_cap + at] = (uint32_t)r;
                    A.lone_slot[r] = (region << 28) | at;          // (the replaying backward finds the ray's record from its block's wave)
                }
            }
            if (dm != 0 && A.defer_list != nullptr) {
                const uint32_t cnt = (uint32_t)__popc(dm);
                uint32_t base = 0;
                if ((tid & 63) == 0) base = atomicAdd(&ST_CNT(A.defer_list, region), cnt);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                deferred = base + cnt <= A.defer_cap;                      // else: the region's list is full and the block walks its packets itself
                base += region * A.defer_cap;                              // the item's index over all regions (its row of the chunk table: n_tiles + index)
                if (deferred && (tid & 63) == 0) {
                    uint32_t i = 0;
                    for (uint32_t left = dm; left; left &= left - 1) A.defer_list[ST_LIST_HDR + base + i++] = ((uint32_t)tile << 5) | (uint32_t)__builtin_ctz(left);
                    A.rec_chunks[(size_t)rec_row * ST_REC_PASSES + ST_REC_PASSES - 1] = ST_REC_DEFERRED;
                    A.rec_chunks[(size_t)rec_row * ST_REC_PASSES + ST_REC_PASSES - 2] = base;     // where its packets' rows start (the block's own row holds no record)
                }
            }
        } else {
            deferred = dm != 0 && A.defer_list != nullptr && A.rec_chunks[(size_t)rec_row * ST_REC_PASSES + ST_REC_PASSES - 1] == ST_REC_DEFERRED;
            if (MODE == 2 && deferred) {
                // The replay needs no walk, so nothing ties it to the forward's one-wave-per-packet split: every lane replays ITS packet's
                // record (row n_tiles + first listed item of the block + the packet's rank among the block's packets) and the whole block
                // is one wave again -- 64 busy lanes and one LDS gradient table instead of up to 16 waves of 4-16 rays each.
                const uint32_t base = A.rec_chunks[(size_t)rec_row * ST_REC_PASSES + ST_REC_PASSES - 2];
                if (packet >= 1) my_row = A.n_tiles + base + (uint32_t)__popc(dm & ((1u << packet) - 1u));
                deferred = false;
            }
        }
    } else {
        packets_present = 1u << only_packet;
    }
    const bool mine = only_packet < 0 ? (no_ray || (packet >= 0 && !deferred)) : (packet == only_packet);
    // The replaying backward also takes its block's single rays along when their recorded hits suffice (round 5): the lane follows the
    // ray's own record (the sorted keys st_trace_lone_rays kept) through the same blend loop and the same LDS gradient table as its
    // neighbours, who meet the same surfels.  A wave per such ray -- or per four of them, st_replay_lone_rays4 -- sent 18 float atomics
    // per hit straight to memory: 12 300 rays cost 0.15 ms at C3 size, 52 000 cost 0.56 ms at C4 size, the atomic unit's rate.
    bool lone_here = false;
    const unsigned long long* lrec = nullptr;
    if (MODE == 2 && ST_LONE_IN_BLOCK && only_packet < 0 && lone && A.lone_rec != nullptr) {
        const uint32_t code = A.lone_slot[r], lloc = code & 0x0FFFFFFFu;
        if (lloc < A.lone_cap && A.state[4 * r + 3] <= (float)ST_LONE_REC_PASSES) {
            lone_here = true;
            lrec = A.lone_rec + ((size_t)(code >> 28) * A.lone_cap + lloc) * (ST_LONE_REC_PASSES * ST_K);
        }
    }
    bool want = !done && ((!lone && mine) || lone_here);
    // The replaying backward first collects a surfel's gradient terms in LDS (per wave: ST_TAB entries, open addressing on the surfel's
    // index): the rays of a block meet the same ~100 surfels at different ranks and in different passes, and the global float atomics
    // (18 per hit) are what bounds this kernel.  An entry that finds no place within four probes goes out directly.
    if (MODE == 2) {
        for (int e = tid & 63; e < ST_TAB; e += 64) {
#pragma unroll
            for (int k = 0; k < 18; ++k) tab[e * 19 + k] = 0u;
            tab[e * 19 + 18] = ST_REC_NONE;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    StProf prof = {0, 0, 0, 0u, 0u};
    for (int pass = 0; pass < ST_MAX_PASSES; ++pass) {
        if (__ballot(want) == 0) break;
        int n = 0;
        if (MODE == 2) {
            if (pass >= ST_REC_PASSES - 1) break;
            const uint32_t chunk = (want && !lone_here) ? A.rec_chunks[(size_t)my_row * ST_REC_PASSES + pass] : ST_REC_NONE;
            const bool has_lone = lone_here && want && pass < ST_LONE_REC_PASSES;
            const bool has = chunk < ST_REC_DEFERRED || has_lone;               // (a lane whose packet recorded no further pass: n = 0 ends it below)
            if (__ballot(has) == 0) break;
            const uint32_t* src = A.rec_arena + (size_t)(chunk < ST_REC_DEFERRED ? chunk : 0u) * (ST_K * 64) + (tid & 63);
            const unsigned long long* lsrc = lrec + (has_lone ? pass * ST_K : 0);
#pragma unroll
            for (int j = 0; j < ST_K; ++j) {
                // (a single ray's record holds (t, id) keys, "none" = all ones: its low word is ST_REC_NONE)
                const uint32_t id = has_lone ? (uint32_t)lsrc[j] : (has ? src[j * 64] : ST_REC_NONE);
                kb_id[j][tid] = id;
                if (id != ST_REC_NONE) n = j + 1;
            }
        } else {
            for (uint32_t left = packets_present; left; left &= left - 1) {          // wave-uniform: one walk per packet
                const int pk = __builtin_ctz(left);
                const bool mine = want && packet == pk;
                const int got = (pk == 0 || ST_GROUP_GATHER == 0)
                    ? st_gather_wide(A.wide, wide_boxes, wide_vmask, leaf_ro, kb_id, kb_t, tid, ox, oy, oz, dx, dy, dz, ivx, ivy, ivz, prev_t, prev_id, pass == 0, mine, prof)
                    : st_gather_group(A.wide, wide_boxes, wide_vmask, leaf_ro, kb_id, kb_t, kb_n, tid, pk, A.ray_width > 0, ox, oy, oz, dx, dy, dz, prev_t, prev_id,
                                      pass == 0, mine, prof);
                if (mine) n = got;
            }
        }
        if (MODE == 0 && A.rec_arena != nullptr) {
            // the record of this pass: the wave's own chunks first, then one drawn from the pool (one atomic per wave and late pass)
            uint32_t chunk = ST_REC_NONE;
            if (only_packet < 0 && pass < ST_REC_STATIC) {
                chunk = rec_row * ST_REC_STATIC + pass;
            } else if (pass < ST_REC_PASSES - 1) {
                uint32_t got = 0;
                if ((tid & 63) == 0) got = atomicAdd(A.rec_hdr, 1u);
                got = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
                if (got < A.rec_pool) chunk = A.rec_static + got;
            }
            if (chunk != ST_REC_NONE) {
                if ((tid & 63) == 0) A.rec_chunks[(size_t)rec_row * ST_REC_PASSES + pass] = chunk;
                uint32_t* dst = A.rec_arena + (size_t)chunk * (ST_K * 64) + (tid & 63);
#pragma unroll
                for (int j = 0; j < ST_K; ++j) dst[j * 64] = (want && j < n) ? kb_id[j][tid] : ST_REC_NONE;
            } else if ((tid & 63) == 0) {
                A.rec_hdr[1] = 1u;                 // the backward walks again
            }
        }
        if (!want) n = 0;                  // (every lane stays with the wave: the merge below moves data across lanes)
        passes += want ? 1 : 0;
        // The wave steps through its lanes' buffers together (a lane without an entry j idles): in the backward, neighbouring rays
        // mostly hold the SAME surfel at the same rank, and lanes that do merge their 18 gradient terms before the atomics.
        for (int j = 0; j < ST_K; ++j) {
            bool act = j < n && !done;
            if (__ballot(act) == 0) break;
            const uint32_t id = act ? kb_id[j][tid] : 0u;
            const float4* g = A.geom + (size_t)id * 4;
            const float4 g0 = g[0], g1 = g[1], g2 = g[2];
            const float opacity = g[3].x;
            const StHit h = st_hit(g0, g1, g2, opacity, ox, oy, oz, dx, dy, dz);      // same expression, same operands as in the gather
            const float t = MODE == 2 ? h.t : kb_t[j][tid];                           // (the same number)
            const float alpha = h.alpha;
            const float test_T = T * (1.0f - alpha);
            if (act && test_T < 0.0001f) { done = true; act = false; }
            const float w = act ? alpha * T : 0.0f;
            const float4 a0 = A.attr[(size_t)id * 2], a1 = A.attr[(size_t)id * 2 + 1];
            const float sgn = h.den > 0.0f ? -1.0f : 1.0f;                   // the normal faces the ray's origin
            const float nfx = sgn * g2.y, nfy = sgn * g2.z, nfz = sgn * g2.w;
            if (!BWD) {
                if (act) {
                    C[0] += w * a0.x; C[1] += w * a0.y; C[2] += w * a0.z;
                    N[0] += w * nfx; N[1] += w * nfy; N[2] += w * nfz;
                    X[0] += w * a0.w; X[1] += w * a1.x;
                    dist += w * (t * t * Aw + M2 - 2.0f * t * M1);
                    D += w * t; Aw += w; M1 += w * t; M2 += w * t * t;
                    if (!ST_NO_WET) atomicAdd(A.wet + id, w);
                }
            } else {
                float gv[18];
#pragma unroll
                for (int k = 0; k < 18; ++k) gv[k] = 0.0f;
                if (act) {
                    const float q = gc[0] * a0.x + gc[1] * a0.y + gc[2] * a0.z + gd * t + ga + gn[0] * nfx + gn[1] * nfy + gn[2] * nfz
                                  + gx[0] * a0.w + gx[1] * a1.x + gdist * (t * t * fA - 2.0f * t * fM1 + fM2);
                    Qpre += w * q;
                    const float inv1ma = 1.0f / (1.0f - alpha);
                    const float dalpha = T * q - (Qtot - Qpre) * inv1ma - fT * bgdot * inv1ma;
                    const float dG = opacity * dalpha;                                    // no clamp mask, as backward.cu:411-413
                    const float du = -h.u * h.G * dG, dv = -h.v * h.G * dG;
                    const float ax = g0.w, ay = g1.x, az = g1.y, bx = g1.z, by = g1.w, bz = g2.x, nx = g2.y, ny = g2.z, nz = g2.w;
                    const float px = (ox + t * dx) - g0.x, py = (oy + t * dy) - g0.y, pz = (oz + t * dz) - g0.z;
                    const float dpx = du * ax + dv * bx, dpy = du * ay + dv * by, dpz = du * az + dv * bz;
                    const float dt = w * (gd + gdist * 2.0f * (t * fA - fM1)) + (dpx * dx + dpy * dy + dpz * dz);
                    const float dnum = dt / h.den, dden = -dt * t / h.den;
                    gv[0] = -dpx + dnum * nx; gv[1] = -dpy + dnum * ny; gv[2] = -dpz + dnum * nz;
                    gv[3] = du * px; gv[4] = du * py; gv[5] = du * pz;
                    gv[6] = dv * px; gv[7] = dv * py; gv[8] = dv * pz;
                    gv[9] = dnum * (g0.x - ox) + dden * dx + sgn * w * gn[0];
                    gv[10] = dnum * (g0.y - oy) + dden * dy + sgn * w * gn[1];
                    gv[11] = dnum * (g0.z - oz) + dden * dz + sgn * w * gn[2];
                    gv[12] = h.G * dalpha;
                    gv[13] = w * gc[0]; gv[14] = w * gc[1]; gv[15] = w * gc[2]; gv[16] = w * gx[0]; gv[17] = w * gx[1];
                    go[0] += dpx - dnum * nx; go[1] += dpy - dnum * ny; go[2] += dpz - dnum * nz;
                    gdir[0] += t * dpx + dden * nx; gdir[1] += t * dpy + dden * ny; gdir[2] += t * dpz + dden * nz;
                }
                // merge with the horizontal, then the vertical neighbour of the 8x8 block when both hold the same surfel: the lower
                // lane of a pair carries the sum, the upper one is done (DPP moves: quad_perm [1,0,3,2] = lane ^ 1, [2,3,0,1] = lane ^ 2, row_ror:8 = lane ^ 8).
                // The replay merges FIRST and collects in LDS what is left (ST_REPLAY_MERGE, round 5): neighbouring rays hold the same surfel
                // at the same rank more often than not, and up to eight lanes adding to one LDS row are eight serialised ds_add_f32 per term
                // (688 k / 2.78 M conflict cycles per launch at C3 / C4 size with the LDS pipe the busiest unit of the kernel).
                bool live = act;
                bool placed = false;
                auto collect = [&]() {
                    if (MODE == 2 && live) {
                        int slot = (int)(((id * 2654435761u) >> 16) % (uint32_t)ST_TAB);
                        for (int probe = 0; probe < 4 && !placed; ++probe) {
                            const uint32_t old = atomicCAS(&tab[slot * 19 + 18], ST_REC_NONE, id);
                            if (old == ST_REC_NONE || old == id) placed = true;
                            else slot = slot + 1 == ST_TAB ? 0 : slot + 1;
                        }
                        if (placed) {
                            float* row = reinterpret_cast<float*>(tab) + slot * 19;
#pragma unroll
                            for (int k = 0; k < 18; ++k) atomicAdd(row + k, gv[k]);
                        }
                    }
                };
                if (!ST_REPLAY_MERGE) collect();
                if (MODE != 2 || ST_REPLAY_MERGE || __ballot(act && !placed) != 0) {
                live = act && !placed;                                      // (what is collected in LDS already takes no part in the merge below)
                {
                    const uint32_t pid = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)id, 0xB1, 0xf, 0xf, true);
                    const bool plive = __builtin_amdgcn_update_dpp(0, (int)live, 0xB1, 0xf, 0xf, true) != 0;
                    const bool merge = live && plive && pid == id;
                    const bool lower = (tid & 1) == 0;
#pragma unroll
                    for (int k = 0; k < 18; ++k) {
                        const float pv = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, gv[k]), 0xB1, 0xf, 0xf, true));
                        const float sum = gv[k] + pv;                    // (v_add_f32_dpp: the move folds into the add)
                        gv[k] = (merge && lower) ? sum : gv[k];
                    }
                    live = live && !(merge && !lower);
                }
                {
                    const uint32_t pid = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)id, 0x4E, 0xf, 0xf, true);        // quad_perm [2,3,0,1] = lane ^ 2
                    const bool plive = __builtin_amdgcn_update_dpp(0, (int)live, 0x4E, 0xf, 0xf, true) != 0;
                    const bool merge = live && plive && pid == id;
                    const bool lower = (tid & 2) == 0;
#pragma unroll
                    for (int k = 0; k < 18; ++k) {
                        const float pv = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, gv[k]), 0x4E, 0xf, 0xf, true));
                        const float sum = gv[k] + pv;                    // (v_add_f32_dpp: the move folds into the add)
                        gv[k] = (merge && lower) ? sum : gv[k];
                    }
                    live = live && !(merge && !lower);
                }
                {
                    const uint32_t pid = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)id, 0x128, 0xf, 0xf, true);
                    const bool plive = __builtin_amdgcn_update_dpp(0, (int)live, 0x128, 0xf, 0xf, true) != 0;
                    const bool merge = live && plive && pid == id;
                    const bool lower = (tid & 8) == 0;
#pragma unroll
                    for (int k = 0; k < 18; ++k) {
                        const float pv = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, gv[k]), 0x128, 0xf, 0xf, true));
                        const float sum = gv[k] + pv;                    // (v_add_f32_dpp: the move folds into the add)
                        gv[k] = (merge && lower) ? sum : gv[k];
                    }
                    live = live && !(merge && !lower);
                }
                if (ST_REPLAY_MERGE) collect();
                if (live && !placed) {
                    float* gg = A.g_geom + (size_t)id * 16;
#pragma unroll
                    for (int k = 0; k < 13; ++k) atomicAdd(gg + k, gv[k]);
                    float* ga_ = A.g_attr + (size_t)id * 8;
#pragma unroll
                    for (int k = 0; k < 5; ++k) atomicAdd(ga_ + k, gv[13 + k]);
                }
                }
            }
            if (act) { T = test_T; ++blended; }
        }
        if (n < ST_K || done) want = false;
        if (want && MODE != 2) {
            prev_t = kb_t[ST_K - 1][tid];
            prev_id = kb_id[ST_K - 1][tid];
        }
    }
    if (MODE == 2) {                           // the collected gradients go out: one entry per lane and round
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int e = tid & 63; e < ST_TAB; e += 64) {
            const uint32_t key = tab[e * 19 + 18];
            if (key != ST_REC_NONE) {
                const float* row = reinterpret_cast<const float*>(tab) + e * 19;
                float* gg = A.g_geom + (size_t)key * 16;
#pragma unroll
                for (int k = 0; k < 13; ++k) atomicAdd(gg + k, row[k]);
                float* ga_ = A.g_attr + (size_t)key * 8;
#pragma unroll
                for (int k = 0; k < 5; ++k) atomicAdd(ga_ + k, row[13 + k]);
            }
        }
    }
    if (!exists || !((mine && !lone) || lone_here)) return;
    if (!BWD) {
        A.rgb[3 * r] = C[0] + T * A.bg[0]; A.rgb[3 * r + 1] = C[1] + T * A.bg[1]; A.rgb[3 * r + 2] = C[2] + T * A.bg[2];
        A.dpt[r] = D; A.acc[r] = Aw; A.dist[r] = dist;
        A.norm[3 * r] = N[0]; A.norm[3 * r + 1] = N[1]; A.norm[3 * r + 2] = N[2];
        A.aux[2 * r] = X[0]; A.aux[2 * r + 1] = X[1];
#ifdef ST_PROFILE
        blended = (int)prof.t_fetch; passes = prof.tests; M2 = (float)(wall_clock64() - prof_t0); T = (float)prof.t_cand;   // 100 MHz ticks
#endif
        reinterpret_cast<float4*>(A.state)[r] = make_float4(M2, T, (float)blended, (float)(packet >= 0 ? -passes : passes));   // sign: walked in a packet
    } else {
        A.g_ray_o[3 * r] = go[0]; A.g_ray_o[3 * r + 1] = go[1]; A.g_ray_o[3 * r + 2] = go[2];
        A.g_ray_d[3 * r] = gdir[0]; A.g_ray_d[3 * r + 1] = gdir[1]; A.g_ray_d[3 * r + 2] = gdir[2];
    }
}

// Wave w of region x of the first launch -> its 8x8 block of rays.  The blocks of rays are grouped into SUPERTILES of ST_SUPER x ST_SUPER
// blocks (4 x 4: 32 x 32 rays, 16 waves), supertile s belongs to region s % 8 and a region walks its supertiles in order: the ~500 waves
// an XCD holds at a time cover ~30 compact patches of the image -- whose mirror rays meet compact parts of the scene -- while every XCD
// gets every eighth patch of the WHOLE image, i.e. an equal share of the work.  Measured on the way (round 5, C4 size, first launch /
// replaying backward in us): eight contiguous bands of the image, one per XCD, 2 270 / -- (a view's rays that hit nothing sit in its
// corners: the launch lasts as long as the band through the middle); supertiles of 16 x 16 blocks 1 682 / 1 550, 8 x 8 1 600 / 1 487,
// 4 x 4 1 582 / 1 388; round 4's round-robin of four-block rows 1 604 / 1 074 (+ 1 003 for the single rays the replay now takes along).
// The fetched bytes fall with the patch size (first launch 1 148 -> 541 MB at 8 x 8); the time follows the balance.  Ray sets that are no
// image: runs of ST_SUPER^2 blocks.
#ifndef ST_SUPER_EDGE
#define ST_SUPER_EDGE 4
#endif
constexpr uint32_t ST_SUPER = ST_SUPER_EDGE;
__host__ __device__ __forceinline__ uint32_t st_supertiles(int64_t n_tiles, int32_t ray_width)
{
    if (ray_width <= 0) return (uint32_t)((n_tiles + ST_SUPER * ST_SUPER - 1) / (ST_SUPER * ST_SUPER));
    const uint32_t tiles_x = (uint32_t)(ray_width + 7) >> 3, tiles_y = (uint32_t)(n_tiles / tiles_x);
    return ((tiles_x + ST_SUPER - 1) / ST_SUPER) * ((tiles_y + ST_SUPER - 1) / ST_SUPER);
}
__device__ __forceinline__ int64_t st_tile_of_wave(const StArgs& A, uint32_t region, uint32_t w)
{
    constexpr uint32_t per = ST_SUPER * ST_SUPER;
    const uint32_t s = (w / per) * 8u + region, t = w % per;
    if (A.ray_width <= 0) {
        const uint64_t tile = (uint64_t)s * per + t;
        return tile < A.n_tiles ? (int64_t)tile : -1;
    }
    const uint32_t tiles_x = (uint32_t)(A.ray_width + 7) >> 3, tiles_y = A.n_tiles / tiles_x;
    const uint32_t sxn = (tiles_x + ST_SUPER - 1) / ST_SUPER;
    const uint32_t sy = s / sxn, sx = s - sy * sxn;
    const uint32_t ty = sy * ST_SUPER + t / ST_SUPER, tx = sx * ST_SUPER + t % ST_SUPER;
    if (tx >= tiles_x || ty >= tiles_y) return -1;
    return (int64_t)ty * tiles_x + tx;
}

// first launch: one wave per block of rays.  MODE 0: forward (walks, blends, records what it gathered); 1: backward that walks again
// (no record, or it overflowed); 2: backward that replays the forward's record -- same blend arithmetic on the same ids in the same
// order, no hierarchy.
template <int MODE>
__global__ __launch_bounds__(ST_THREADS) __attribute__((amdgpu_waves_per_eu(MODE == 2 ? 3 : ST_FWD_WAVES, 8))) void st_trace_kernel(StArgs A, const float4* __restrict__ leaf_ro, const float* __restrict__ wide_boxes,
                                                              const unsigned long long* __restrict__ wide_vmask)
{
    __shared__ uint32_t kb_id[ST_K][ST_THREADS];
    __shared__ float kb_t[MODE == 2 ? 1 : ST_K][ST_THREADS];                 // (the replay needs no depths: it recomputes them)
    __shared__ uint32_t kb_n[MODE == 2 ? 1 : ST_THREADS];
    __shared__ uint32_t tab[MODE == 2 ? ST_THREADS / 64 : 1][MODE == 2 ? ST_TAB_WORDS : 1];
    if (MODE != 0 && A.rec_hdr != nullptr && ((A.rec_hdr[1] != 0u) != (MODE == 1))) return;     // the other backward does the work
    const int tid = threadIdx.x;
    // XCD b % 8 takes every eighth supertile of the ray set (st_tile_of_wave)
    const uint32_t region = blockIdx.x & 7u;
    const int64_t tile = st_tile_of_wave(A, region, (blockIdx.x >> 3) * (ST_THREADS / 64) + (tid >> 6));
    if (tile < 0) return;
    st_trace_tile<MODE>(A, leaf_ro, wide_boxes, wide_vmask, kb_id, kb_t, kb_n, tab[MODE == 2 ? tid >> 6 : 0], tid, tile, -1, (uint32_t)tile, region);
}

// ---- rays that run with nobody: one WAVE per ray --------------------------------------------------------------------------------
// A ray whose 2x2 neighbours point elsewhere (silhouettes, normals of barely covered pixels) cannot share a walk.  Walking alone in
// a lane is the slow way on this machine: every step is a dependent ~1 us gather and the wave lasts as long as its slowest lane
// (measured: 4-10 ms for a wave of such lanes, the whole kernel's duration).  So the main kernel only LISTS these rays and this
// kernel gives each of them a wave, with the parallelism turned sideways: the 64 lanes are the 64 children of a wide node, then the
// 64 surfels of a leaf group -- each lane tests ITS surfel against the one ray.  The ray's 16 nearest hits live in lanes 0..15
// (sorted); new hits are merged by rank (rank = entries in front of me, counted with one v_readlane sweep over the other side);
// blending is lane-parallel too: transmittance and the running sums are 16-lane scans, the pixel sums wave reductions, every hit's
// gradient is written by its own lane.
// (Hillis-Steele inside the 16-lane DPP row: row_shr:1/2/4/8 with the identity for lanes whose source falls outside the row -- the same
// products / sums in the same order as the __shfl_up form these replace, without its five ds_bpermute round trips)
__device__ __forceinline__ float scan16_mul_excl(float v, int lane)          // exclusive prefix product over lanes 0..15 (others: don't care)
{
    float inc = v;
    inc *= ST_DPP(1.0f, inc, 0x111, 0xf); inc *= ST_DPP(1.0f, inc, 0x112, 0xf); inc *= ST_DPP(1.0f, inc, 0x114, 0xf); inc *= ST_DPP(1.0f, inc, 0x118, 0xf);
    return ST_DPP(1.0f, inc, 0x111, 0xf);
}
__device__ __forceinline__ float scan16_add_excl(float v, int lane)
{
    float inc = v;
    inc += ST_DPP(0.0f, inc, 0x111, 0xf); inc += ST_DPP(0.0f, inc, 0x112, 0xf); inc += ST_DPP(0.0f, inc, 0x114, 0xf); inc += ST_DPP(0.0f, inc, 0x118, 0xf);
    return ST_DPP(0.0f, inc, 0x111, 0xf);
}
__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long v, int l)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}

template <bool BWD>
__device__ __forceinline__ void st_trace_lone_rays(const StArgs& A, const float4* __restrict__ leaf, const float* __restrict__ boxes,
                                                   const unsigned long long* __restrict__ vmask, const uint32_t* __restrict__ lone_list,
                                                   unsigned long long* slot, int lane, uint32_t region, uint32_t first_item, uint32_t item_stride,
                                                   bool skip_replayable = false)
{
    // item_stride == 0: the one item `first_item` (the forward's second launch hands items out by ticket); otherwise every item_stride-th
    // skip_replayable (backward): rays whose recorded hits suffice are st_replay_lone_rays4's, four to a wave
    const uint32_t listed = ST_CNT(lone_list, region);
    const uint32_t count = listed < A.lone_list_cap ? listed : A.lone_list_cap;
    const StWide& W = A.wide;
    for (uint32_t item = first_item; item < count; item += item_stride) {
        const int64_t r = lone_list[ST_LIST_HDR + (size_t)region * A.lone_list_cap + item];
        if (BWD && skip_replayable && A.lone_rec != nullptr && item < A.lone_cap && A.state[4 * r + 3] <= (float)ST_LONE_REC_PASSES) continue;
        const float ox = A.ray_o[3 * r], oy = A.ray_o[3 * r + 1], oz = A.ray_o[3 * r + 2];
        const float dx = A.ray_d[3 * r], dy = A.ray_d[3 * r + 1], dz = A.ray_d[3 * r + 2];
        const float ivx = 1.0f / dx, ivy = 1.0f / dy, ivz = 1.0f / dz;
        // running totals (uniform) and, for the backward, the forward's totals
        float T = 1.0f, C0 = 0, C1 = 0, C2 = 0, D = 0, Aw = 0, N0 = 0, N1 = 0, N2 = 0, X0 = 0, X1 = 0, dist = 0, M1 = 0, M2 = 0;
        int blended = 0, passes = 0;
        float gc0 = 0, gc1 = 0, gc2 = 0, gd = 0, ga = 0, gn0 = 0, gn1 = 0, gn2 = 0, gx0 = 0, gx1 = 0, gdist = 0;
        float fA = 0, fM1 = 0, fM2 = 0, fT = 0, Qtot = 0, Qpre = 0, bgdot = 0;
        float go0 = 0, go1 = 0, go2 = 0, gv0 = 0, gv1 = 0, gv2 = 0;         // per lane partial sums of the ray's own gradient
        if (BWD) {
            if (A.g_rgb) { gc0 = A.g_rgb[3 * r]; gc1 = A.g_rgb[3 * r + 1]; gc2 = A.g_rgb[3 * r + 2]; }
            if (A.g_dpt) gd = A.g_dpt[r];
            if (A.g_acc) ga = A.g_acc[r];
            if (A.g_dist) gdist = A.g_dist[r];
            if (A.g_norm) { gn0 = A.g_norm[3 * r]; gn1 = A.g_norm[3 * r + 1]; gn2 = A.g_norm[3 * r + 2]; }
            if (A.g_aux) { gx0 = A.g_aux[2 * r]; gx1 = A.g_aux[2 * r + 1]; }
            fA = A.acc[r]; fM1 = A.dpt[r]; fM2 = A.state[4 * r]; fT = A.state[4 * r + 1];
            bgdot = gc0 * A.bg[0] + gc1 * A.bg[1] + gc2 * A.bg[2];
            Qtot = gc0 * (A.rgb[3 * r] - fT * A.bg[0]) + gc1 * (A.rgb[3 * r + 1] - fT * A.bg[1]) + gc2 * (A.rgb[3 * r + 2] - fT * A.bg[2])
                 + gd * fM1 + ga * fA + gn0 * A.norm[3 * r] + gn1 * A.norm[3 * r + 1] + gn2 * A.norm[3 * r + 2]
                 + gx0 * A.aux[2 * r] + gx1 * A.aux[2 * r + 1] + gdist * 2.0f * (fA * fM2 - fM1 * fM1);
        }
        unsigned long long prev_key = 0;
        bool done = false;
        // the forward keeps the sorted hits of the first ST_LONE_REC_PASSES passes of every listed ray (128 bytes a pass); a backward whose
        // ray needed no more than that replays them instead of walking the hierarchy again (the walk was 0.3 of the backward's 0.9 ms)
        unsigned long long* rec = (A.lone_rec != nullptr && item < A.lone_cap) ? A.lone_rec + ((size_t)region * A.lone_cap + item) * (ST_LONE_REC_PASSES * ST_K) : nullptr;
        const bool replay = BWD && rec != nullptr && A.state[4 * r + 3] <= (float)ST_LONE_REC_PASSES;
        for (int pass = 0; pass < ST_MAX_PASSES && !done; ++pass) {
            // ---- gather: the ST_K smallest keys above prev_key, sorted, one per lane 0..ST_K-1 ----
            unsigned long long mine = ~0ull;                                   // lanes >= nb hold "none"
            int nb = 0;
            float t_far = INFINITY;
            if (replay) {
                if (lane < ST_K) mine = rec[pass * ST_K + lane];
                nb = (int)__popcll(__ballot(mine != ~0ull));
            } else {
            unsigned long long mask[SW_MAX_LEVELS] = {0, 0, 0, 0};
            int node[SW_MAX_LEVELS] = {0, 0, 0, 0};
            float near1 = 0.f, near2 = 0.f, near3 = 0.f;
            const float prev_t = __uint_as_float((uint32_t)(prev_key >> 32));
            int l = W.n - 1;
            bool fresh = true;
            for (;;) {
                if (fresh) {
                    const int nd = W.off[l] + node[l];
                    fresh = false;
                    if (l > 0) {
                        const float* bx = boxes + (size_t)nd * 384 + lane;
                        const float ax = (bx[0] - ox) * ivx, bxx = (bx[192] - ox) * ivx, ay = (bx[64] - oy) * ivy, by = (bx[256] - oy) * ivy;
                        const float az = (bx[128] - oz) * ivz, bz = (bx[320] - oz) * ivz;
                        const float t_in = fmaxf(fmaxf(fminf(ax, bxx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
                        const float t_out = fminf(fminf(fmaxf(ax, bxx), fmaxf(ay, by)), fmaxf(az, bz));
                        const bool h = t_in <= t_out * 1.00001f + 1e-30f && t_in * 0.99999f <= t_far && t_out * 1.00001f + 1e-30f >= prev_t;
                        mask[l] = __ballot(h) & vmask[nd];
                        if (l == 1) near1 = t_in; else if (l == 2) near2 = t_in; else near3 = t_in;
                    } else {
                        // 64 surfels against the ray at once
                        const float4* g = leaf + ((size_t)node[0] * 64 + lane) * 4;
                        const float4 g0 = g[0], g1 = g[1], g2 = g[2], g3 = g[3];
                        const StHit h = st_hit(g0, g1, g2, g3.x, ox, oy, oz, dx, dy, dz);
                        const unsigned long long cand = ((unsigned long long)__float_as_uint(h.t) << 32) | __float_as_uint(g3.y);
                        const unsigned long long kth = readlane_u64(mine, ST_K - 1);
                        const bool c_ok = ((vmask[nd] >> lane) & 1ull) && h.ok && (pass == 0 || cand > prev_key) && cand < kth;
                        const unsigned long long cmask = __ballot(c_ok);
                        if (cmask) {
                            // new rank of every item: buffer entries keep their order, candidates slot in by key
                            int rank_b = lane;                                 // for lanes < nb
                            int rank_c = 0;
                            for (unsigned long long m = cmask; m; m &= m - 1) {
                                const unsigned long long k = readlane_u64(cand, __builtin_ctzll(m));
                                rank_b += k < mine ? 1 : 0;
                                rank_c += k < cand ? 1 : 0;
                            }
                            for (int j = 0; j < nb; ++j) rank_c += readlane_u64(mine, j) < cand ? 1 : 0;
                            if (lane < ST_K) slot[lane] = ~0ull;
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                            if (lane < nb && rank_b < ST_K) slot[rank_b] = mine;
                            if (c_ok && rank_c < ST_K) slot[rank_c] = cand;
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                            mine = lane < ST_K ? slot[lane] : ~0ull;
                            nb = min(ST_K, nb + (int)__popcll(cmask));
                            if (nb == ST_K) t_far = __uint_as_float((uint32_t)(readlane_u64(mine, ST_K - 1) >> 32)) * 1.00001f + 1e-30f;
                        }
                        mask[0] = 0;
                    }
                }
                if (mask[l] == 0) {
                    if (l == W.n - 1) break;
                    ++l;
                    continue;
                }
                const float my_near = l == 1 ? near1 : l == 2 ? near2 : near3;
                const float key = ((mask[l] >> lane) & 1ull) ? my_near : INFINITY;
                const float nearest = wave_min_f(key);
                if (nearest * 0.99999f > t_far) { mask[l] = 0; continue; }
                const int c = __builtin_ctzll(__ballot(key == nearest));
                mask[l] &= ~(1ull << c);
                node[l - 1] = node[l] * 64 + c;
                --l;
                fresh = true;
            }
            if (!BWD && rec != nullptr && pass < ST_LONE_REC_PASSES && lane < ST_K) rec[pass * ST_K + lane] = mine;
            }   // (walk)
            ++passes;
            // ---- blend: hit j in lane j ----
            const bool has = lane < nb;
            const uint32_t id = (uint32_t)mine;
            const float t = has ? __uint_as_float((uint32_t)(mine >> 32)) : 0.0f;      // "none" is all ones: a NaN as a float
            float4 g0 = make_float4(0, 0, 0, 0), g1 = g0, g2 = g0, a0 = g0, a1 = g0;
            float opacity = 0.f;
            StHit h;
            h.alpha = 0.f; h.den = 1.f; h.u = h.v = h.G = 0.f;
            if (has) {
                const float4* g = A.geom + (size_t)id * 4;
                g0 = g[0]; g1 = g[1]; g2 = g[2]; opacity = g[3].x;
                a0 = A.attr[(size_t)id * 2]; a1 = A.attr[(size_t)id * 2 + 1];
                h = st_hit(g0, g1, g2, opacity, ox, oy, oz, dx, dy, dz);
            }
            const float alpha = has ? h.alpha : 0.0f;
            const float Tj = T * scan16_mul_excl(1.0f - alpha, lane);          // transmittance in front of hit j
            const unsigned long long stop = __ballot(has && Tj * (1.0f - alpha) < 0.0001f);
            const int n_bl = stop ? min(nb, (int)__builtin_ctzll(stop)) : nb;
            const bool bl = lane < n_bl;
            const float w = bl ? alpha * Tj : 0.0f;
            const float sgn = h.den > 0.0f ? -1.0f : 1.0f;
            const float nfx = sgn * g2.y, nfy = sgn * g2.z, nfz = sgn * g2.w;
            const float Ab = Aw + scan16_add_excl(w, lane), M1b = M1 + scan16_add_excl(w * t, lane), M2b = M2 + scan16_add_excl(w * t * t, lane);
            if (!BWD) {
                C0 += wave_sum_f(w * a0.x); C1 += wave_sum_f(w * a0.y); C2 += wave_sum_f(w * a0.z);
                N0 += wave_sum_f(w * nfx); N1 += wave_sum_f(w * nfy); N2 += wave_sum_f(w * nfz);
                X0 += wave_sum_f(w * a0.w); X1 += wave_sum_f(w * a1.x);
                dist += wave_sum_f(w * (t * t * Ab + M2b - 2.0f * t * M1b));
                if (bl && !ST_NO_WET) atomicAdd(A.wet + id, w);
            } else {
                const float q = gc0 * a0.x + gc1 * a0.y + gc2 * a0.z + gd * t + ga + gn0 * nfx + gn1 * nfy + gn2 * nfz + gx0 * a0.w + gx1 * a1.x
                              + gdist * (t * t * fA - 2.0f * t * fM1 + fM2);
                const float wq = w * q;
                const float Qin = Qpre + scan16_add_excl(wq, lane) + wq;            // inclusive
                if (bl) {
                    const float inv1ma = 1.0f / (1.0f - alpha);
                    const float dalpha = Tj * q - (Qtot - Qin) * inv1ma - fT * bgdot * inv1ma;
                    const float dG = opacity * dalpha;
                    const float du = -h.u * h.G * dG, dv = -h.v * h.G * dG;
                    const float ax = g0.w, ay = g1.x, az = g1.y, bx = g1.z, by = g1.w, bz = g2.x, nx = g2.y, ny = g2.z, nz = g2.w;
                    const float px = (ox + t * dx) - g0.x, py = (oy + t * dy) - g0.y, pz = (oz + t * dz) - g0.z;
                    const float dpx = du * ax + dv * bx, dpy = du * ay + dv * by, dpz = du * az + dv * bz;
                    const float dt = w * (gd + gdist * 2.0f * (t * fA - fM1)) + (dpx * dx + dpy * dy + dpz * dz);
                    const float dnum = dt / h.den, dden = -dt * t / h.den;
                    float* gg = A.g_geom + (size_t)id * 16;
                    atomicAdd(gg + 0, -dpx + dnum * nx); atomicAdd(gg + 1, -dpy + dnum * ny); atomicAdd(gg + 2, -dpz + dnum * nz);
                    atomicAdd(gg + 3, du * px); atomicAdd(gg + 4, du * py); atomicAdd(gg + 5, du * pz);
                    atomicAdd(gg + 6, dv * px); atomicAdd(gg + 7, dv * py); atomicAdd(gg + 8, dv * pz);
                    atomicAdd(gg + 9, dnum * (g0.x - ox) + dden * dx + sgn * w * gn0);
                    atomicAdd(gg + 10, dnum * (g0.y - oy) + dden * dy + sgn * w * gn1);
                    atomicAdd(gg + 11, dnum * (g0.z - oz) + dden * dz + sgn * w * gn2);
                    atomicAdd(gg + 12, h.G * dalpha);
                    float* ga_ = A.g_attr + (size_t)id * 8;
                    atomicAdd(ga_ + 0, w * gc0); atomicAdd(ga_ + 1, w * gc1); atomicAdd(ga_ + 2, w * gc2);
                    atomicAdd(ga_ + 3, w * gx0); atomicAdd(ga_ + 4, w * gx1);
                    go0 += dpx - dnum * nx; go1 += dpy - dnum * ny; go2 += dpz - dnum * nz;
                    gv0 += t * dpx + dden * nx; gv1 += t * dpy + dden * ny; gv2 += t * dpz + dden * nz;
                }
                Qpre += wave_sum_f(wq);
            }
            D += wave_sum_f(w * t);
            const float sw = wave_sum_f(w);
            Aw += sw; M1 += wave_sum_f(w * t); M2 += wave_sum_f(w * t * t);
            // transmittance behind the blended hits
            const float keep = bl ? 1.0f - alpha : 1.0f;
            float prod = keep;          // product over the row's 16 lanes: inside the quads, then the neighbouring quads by rotation
            prod *= ST_DPP(1.0f, prod, 0xb1, 0xf); prod *= ST_DPP(1.0f, prod, 0x4e, 0xf); prod *= ST_DPP(1.0f, prod, 0x124, 0xf); prod *= ST_DPP(1.0f, prod, 0x128, 0xf);
            T *= st_uniform(prod);
            blended += n_bl;
            if (stop || nb < ST_K) done = true;
            else prev_key = readlane_u64(mine, ST_K - 1);
        }
        if (!BWD) {
            if (lane == 0) {
                A.rgb[3 * r] = C0 + T * A.bg[0]; A.rgb[3 * r + 1] = C1 + T * A.bg[1]; A.rgb[3 * r + 2] = C2 + T * A.bg[2];
                A.dpt[r] = D; A.acc[r] = Aw; A.dist[r] = dist;
                A.norm[3 * r] = N0; A.norm[3 * r + 1] = N1; A.norm[3 * r + 2] = N2;
                A.aux[2 * r] = X0; A.aux[2 * r + 1] = X1;
                reinterpret_cast<float4*>(A.state)[r] = make_float4(M2, T, (float)blended, (float)passes);
            }
        } else {
            const float s0 = wave_sum_f(go0), s1 = wave_sum_f(go1), s2 = wave_sum_f(go2), v0 = wave_sum_f(gv0), v1 = wave_sum_f(gv1), v2 = wave_sum_f(gv2);
            if (lane == 0) {
                A.g_ray_o[3 * r] = s0; A.g_ray_o[3 * r + 1] = s1; A.g_ray_o[3 * r + 2] = s2;
                A.g_ray_d[3 * r] = v0; A.g_ray_d[3 * r + 1] = v1; A.g_ray_d[3 * r + 2] = v2;
            }
        }
        if (item_stride == 0u) break;
    }
}

// Backward of the single rays whose recorded hits suffice (ST_LONE_REC_PASSES passes of 16 sorted keys), FOUR RAYS PER WAVE: the replay
// needs no walk, and a ray's 16 hits fill one 16-lane DPP row -- hit j of ray `row` in lane 16 row + j.  Same arithmetic as the replaying
// branch of st_trace_lone_rays<true> (round 3 gave every such ray a wave of its own, 48 of whose 64 lanes idled through the gathers:
// 385 us for 12 300 rays at C3 size, 1.0 ms at C4 size), with the wave-wide scans and sums cut down to the row.
__device__ __forceinline__ float row_sum_f(float v)             // sum over the 16-lane row, in every lane of the row
{
    v += ST_DPP(0.0f, v, 0xb1, 0xf); v += ST_DPP(0.0f, v, 0x4e, 0xf); v += ST_DPP(0.0f, v, 0x124, 0xf); v += ST_DPP(0.0f, v, 0x128, 0xf);
    return v;
}
__device__ __forceinline__ void st_replay_lone_rays4(const StArgs& A, const uint32_t* __restrict__ lone_list, int lane, uint32_t region,
                                                     uint32_t first_quad, uint32_t quad_stride)
{
    if (A.lone_rec == nullptr) return;
    const uint32_t listed = ST_CNT(lone_list, region);
    uint32_t count = listed < A.lone_list_cap ? listed : A.lone_list_cap;
    if (count > A.lone_cap) count = A.lone_cap;                  // (items beyond the record's capacity walk again: st_trace_lone_rays)
    const int row = lane >> 4, j = lane & 15;
    for (uint32_t quad = first_quad; 4u * quad < count; quad += quad_stride) {
        const uint32_t item = 4u * quad + (uint32_t)row;
        bool valid = item < count;
        const int64_t r = valid ? (int64_t)lone_list[ST_LIST_HDR + (size_t)region * A.lone_list_cap + item] : 0;
        valid = valid && A.state[4 * r + 3] <= (float)ST_LONE_REC_PASSES;
        if (__ballot(valid) == 0) continue;
        const float ox = A.ray_o[3 * r], oy = A.ray_o[3 * r + 1], oz = A.ray_o[3 * r + 2];
        const float dx = A.ray_d[3 * r], dy = A.ray_d[3 * r + 1], dz = A.ray_d[3 * r + 2];
        float gc0 = 0, gc1 = 0, gc2 = 0, gd = 0, ga = 0, gn0 = 0, gn1 = 0, gn2 = 0, gx0 = 0, gx1 = 0, gdist = 0;
        if (A.g_rgb) { gc0 = A.g_rgb[3 * r]; gc1 = A.g_rgb[3 * r + 1]; gc2 = A.g_rgb[3 * r + 2]; }
        if (A.g_dpt) gd = A.g_dpt[r];
        if (A.g_acc) ga = A.g_acc[r];
        if (A.g_dist) gdist = A.g_dist[r];
        if (A.g_norm) { gn0 = A.g_norm[3 * r]; gn1 = A.g_norm[3 * r + 1]; gn2 = A.g_norm[3 * r + 2]; }
        if (A.g_aux) { gx0 = A.g_aux[2 * r]; gx1 = A.g_aux[2 * r + 1]; }
        const float fA = A.acc[r], fM1 = A.dpt[r], fM2 = A.state[4 * r], fT = A.state[4 * r + 1];
        const float bgdot = gc0 * A.bg[0] + gc1 * A.bg[1] + gc2 * A.bg[2];
        const float Qtot = gc0 * (A.rgb[3 * r] - fT * A.bg[0]) + gc1 * (A.rgb[3 * r + 1] - fT * A.bg[1]) + gc2 * (A.rgb[3 * r + 2] - fT * A.bg[2])
                         + gd * fM1 + ga * fA + gn0 * A.norm[3 * r] + gn1 * A.norm[3 * r + 1] + gn2 * A.norm[3 * r + 2]
                         + gx0 * A.aux[2 * r] + gx1 * A.aux[2 * r + 1] + gdist * 2.0f * (fA * fM2 - fM1 * fM1);
        float T = 1.0f, Qpre = 0.0f;
        float go0 = 0, go1 = 0, go2 = 0, gv0 = 0, gv1 = 0, gv2 = 0;
        bool done = !valid;
        const unsigned long long* rec = A.lone_rec + ((size_t)region * A.lone_cap + (valid ? item : 0u)) * (ST_LONE_REC_PASSES * ST_K);
        for (int pass = 0; pass < ST_LONE_REC_PASSES; ++pass) {
            if (__ballot(!done) == 0) break;
            const unsigned long long mine = done ? ~0ull : rec[pass * ST_K + j];
            const uint32_t rowmask = (uint32_t)(__ballot(mine != ~0ull) >> (16 * row)) & 0xFFFFu;
            const int nb = (int)__popc(rowmask);
            const bool has = j < nb;
            const uint32_t id = (uint32_t)mine;
            const float t = has ? __uint_as_float((uint32_t)(mine >> 32)) : 0.0f;
            float4 g0 = make_float4(0, 0, 0, 0), g1 = g0, g2 = g0, a0 = g0, a1 = g0;
            float opacity = 0.f;
            StHit h;
            h.alpha = 0.f; h.den = 1.f; h.u = h.v = h.G = 0.f;
            if (has) {
                const float4* g = A.geom + (size_t)id * 4;
                g0 = g[0]; g1 = g[1]; g2 = g[2]; opacity = g[3].x;
                a0 = A.attr[(size_t)id * 2]; a1 = A.attr[(size_t)id * 2 + 1];
                h = st_hit(g0, g1, g2, opacity, ox, oy, oz, dx, dy, dz);
            }
            const float alpha = has ? h.alpha : 0.0f;
            const float Tj = T * scan16_mul_excl(1.0f - alpha, lane);          // transmittance in front of hit j
            const uint32_t stop = (uint32_t)(__ballot(has && Tj * (1.0f - alpha) < 0.0001f) >> (16 * row)) & 0xFFFFu;
            const int n_bl = stop ? min(nb, (int)__builtin_ctz(stop)) : nb;
            const bool bl = j < n_bl;
            const float w = bl ? alpha * Tj : 0.0f;
            const float sgn = h.den > 0.0f ? -1.0f : 1.0f;
            const float nfx = sgn * g2.y, nfy = sgn * g2.z, nfz = sgn * g2.w;
            const float q = gc0 * a0.x + gc1 * a0.y + gc2 * a0.z + gd * t + ga + gn0 * nfx + gn1 * nfy + gn2 * nfz + gx0 * a0.w + gx1 * a1.x
                          + gdist * (t * t * fA - 2.0f * t * fM1 + fM2);
            const float wq = w * q;
            const float Qin = Qpre + scan16_add_excl(wq, lane) + wq;            // inclusive
            if (bl) {
                const float inv1ma = 1.0f / (1.0f - alpha);
                const float dalpha = Tj * q - (Qtot - Qin) * inv1ma - fT * bgdot * inv1ma;
                const float dG = opacity * dalpha;
                const float du = -h.u * h.G * dG, dv = -h.v * h.G * dG;
                const float ax = g0.w, ay = g1.x, az = g1.y, bx = g1.z, by = g1.w, bz = g2.x, nx = g2.y, ny = g2.z, nz = g2.w;
                const float px = (ox + t * dx) - g0.x, py = (oy + t * dy) - g0.y, pz = (oz + t * dz) - g0.z;
                const float dpx = du * ax + dv * bx, dpy = du * ay + dv * by, dpz = du * az + dv * bz;
                const float dt = w * (gd + gdist * 2.0f * (t * fA - fM1)) + (dpx * dx + dpy * dy + dpz * dz);
                const float dnum = dt / h.den, dden = -dt * t / h.den;
                float* gg = A.g_geom + (size_t)id * 16;
                atomicAdd(gg + 0, -dpx + dnum * nx); atomicAdd(gg + 1, -dpy + dnum * ny); atomicAdd(gg + 2, -dpz + dnum * nz);
                atomicAdd(gg + 3, du * px); atomicAdd(gg + 4, du * py); atomicAdd(gg + 5, du * pz);
                atomicAdd(gg + 6, dv * px); atomicAdd(gg + 7, dv * py); atomicAdd(gg + 8, dv * pz);
                atomicAdd(gg + 9, dnum * (g0.x - ox) + dden * dx + sgn * w * gn0);
                atomicAdd(gg + 10, dnum * (g0.y - oy) + dden * dy + sgn * w * gn1);
                atomicAdd(gg + 11, dnum * (g0.z - oz) + dden * dz + sgn * w * gn2);
                atomicAdd(gg + 12, h.G * dalpha);
                float* ga_ = A.g_attr + (size_t)id * 8;
                atomicAdd(ga_ + 0, w * gc0); atomicAdd(ga_ + 1, w * gc1); atomicAdd(ga_ + 2, w * gc2);
                atomicAdd(ga_ + 3, w * gx0); atomicAdd(ga_ + 4, w * gx1);
                go0 += dpx - dnum * nx; go1 += dpy - dnum * ny; go2 += dpz - dnum * nz;
                gv0 += t * dpx + dden * nx; gv1 += t * dpy + dden * ny; gv2 += t * dpz + dden * nz;
            }
            Qpre += row_sum_f(wq);
            const float keep = bl ? 1.0f - alpha : 1.0f;
            float prod = keep;          // product over the row's 16 lanes (the order of st_trace_lone_rays)
            prod *= ST_DPP(1.0f, prod, 0xb1, 0xf); prod *= ST_DPP(1.0f, prod, 0x4e, 0xf); prod *= ST_DPP(1.0f, prod, 0x124, 0xf); prod *= ST_DPP(1.0f, prod, 0x128, 0xf);
            T *= prod;
            if (stop || nb < ST_K) done = true;
        }
        const float s0 = row_sum_f(go0), s1 = row_sum_f(go1), s2 = row_sum_f(go2), v0 = row_sum_f(gv0), v1 = row_sum_f(gv1), v2 = row_sum_f(gv2);
        if (valid && j == 0) {
            A.g_ray_o[3 * r] = s0; A.g_ray_o[3 * r + 1] = s1; A.g_ray_o[3 * r + 2] = s2;
            A.g_ray_d[3 * r] = v0; A.g_ray_d[3 * r + 1] = v1; A.g_ray_d[3 * r + 2] = v2;
        }
    }
}

// second launch: every listed packet and every listed single ray gets a wave.  One launch for both: each kind ends in a tail of a few long
// waves, and the two tails overlap instead of following each other.
// FORWARD (MODE 0): ST_REST_BLOCKS blocks of persistent waves.  A wave of XCD x (block b, x = b % 8) takes the next packet of region x's
// list by ticket -- what the first launch's waves on that XCD listed, front to back, so that the waves an XCD holds at a time trace
// neighbouring rays (StArgs) --, and when that list is exhausted it goes on with the lists of the other regions, then with the single rays
// the same way.  The lists of the regions differ: what gets listed are silhouettes and grazing normals, which sit in a few dozen of an
// 800 x 800 view's 169 supertiles, and a static "XCD x walks list x" ran 40 % longer than the region-blind stride of round 4 while
// fetching a third of its bytes (measured, round 5) -- with the tickets the XCDs that finish early take over the tail of the others.
// A ticket is one L2 atomic per ~150 us walk.  BACKWARD (MODE 1 / 2): no walk to keep local (the replay reads records), and the state
// is not the backward's to write: every list is strided over all waves of the launch.
constexpr int ST_REST_BLOCKS = 2048;
constexpr int ST_REST_PACKET_BLOCKS = 2048;      // ST_REST_SCHED == 2 only (A/B)

// ST_CNT(hdr, r): counts of the eight region lists (final: written by the launch before); ST_TKT(hdr, r): tickets (zeroed by st_init_kernel).  `k`:
// regions this wave has found exhausted (tickets only grow: they stay exhausted).  Wave-uniform result.
__device__ __forceinline__ bool st_next_item(uint32_t* hdr, uint32_t cap, uint32_t own, int lane, uint32_t& k, uint32_t& region, uint32_t& local)
{
    for (; k < (ST_NO_STEAL ? 1u : 8u); ++k) {
        const uint32_t r = (own + k) & 7u;
        const uint32_t listed = ST_CNT(hdr, r);
        const uint32_t n = listed < cap ? listed : cap;
        uint32_t t = n;
        if (lane == 0) t = atomicAdd(ST_TKT(hdr, r), 1u);      // (no look before the leap: a ticket beyond the end costs one atomic, a look costs one per item)
        t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
        if (t < n) { region = r; local = t; return true; }
    }
    return false;
}

template <int MODE>
__global__ __launch_bounds__(ST_THREADS) __attribute__((amdgpu_waves_per_eu(MODE == 2 ? 3 : ST_FWD_WAVES, 8))) void st_trace_rest_kernel(StArgs A, const float4* __restrict__ leaf_ro, const float* __restrict__ wide_boxes,
                                                                   const unsigned long long* __restrict__ wide_vmask, uint32_t* __restrict__ lone_list)
{
    __shared__ uint32_t kb_id[ST_K][ST_THREADS];
    __shared__ float kb_t[MODE == 2 ? 1 : ST_K][ST_THREADS];
    __shared__ uint32_t kb_n[MODE == 2 ? 1 : ST_THREADS];
    __shared__ uint32_t tab[MODE == 2 ? ST_THREADS / 64 : 1][MODE == 2 ? ST_TAB_WORDS : 1];
    __shared__ unsigned long long slot[ST_THREADS / 64][ST_K];
    if (MODE != 0 && A.rec_hdr != nullptr && ((A.rec_hdr[1] != 0u) != (MODE == 1))) return;     // the other backward does the work
    const int tid = threadIdx.x, lane = tid & 63;
    if (MODE == 0 && ST_REST_SCHED == 2) {
        // tickets, but a wave takes only its share of the items and retires (the hardware's dispatcher starts the next block): blocks below
        // ST_REST_PACKET_BLOCKS take packets, the others single rays
        const uint32_t own = ST_GLOBAL_ORDER ? 0u : (blockIdx.x & 7u);
        uint32_t k = 0, region = 0, local = 0;
        if (ST_REST_SCHED_STATIC_REGION) {
            // (A/B) XCD x walks region x's lists, one item per wave and stride
            const uint32_t r = blockIdx.x & 7u;
            if (blockIdx.x < ST_REST_PACKET_BLOCKS) {
                if (A.defer_list == nullptr) return;
                const uint32_t listed = ST_CNT(A.defer_list, r), count = listed < A.defer_cap ? listed : A.defer_cap;
                for (uint32_t l = (blockIdx.x >> 3) * (ST_THREADS / 64) + (tid >> 6); l < count; l += (ST_REST_PACKET_BLOCKS / 8) * (ST_THREADS / 64)) {
                    const uint32_t item = r * A.defer_cap + l;
                    const uint32_t code = A.defer_list[ST_LIST_HDR + item];
                    if (code == ST_REC_NONE) continue;
                    st_trace_tile<MODE>(A, leaf_ro, wide_boxes, wide_vmask, kb_id, kb_t, kb_n, tab[0], tid, (int64_t)(code >> 5), (int)(code & 31u), A.n_tiles + item, r);
                }
            } else {
                st_trace_lone_rays<false>(A, leaf_ro, wide_boxes, wide_vmask, lone_list, slot[tid >> 6], lane, r,
                                          ((blockIdx.x - ST_REST_PACKET_BLOCKS) >> 3) * (ST_THREADS / 64) + (tid >> 6), ((gridDim.x - ST_REST_PACKET_BLOCKS) / 8) * (ST_THREADS / 64));
            }
            return;
        }
        if (blockIdx.x < ST_REST_PACKET_BLOCKS) {
            if (A.defer_list == nullptr) return;
            uint32_t total = 0;
            for (int r = 0; r < 8; ++r) total += min(ST_CNT(A.defer_list, r), A.defer_cap);
            const uint32_t waves = ST_REST_PACKET_BLOCKS * (ST_THREADS / 64);
            for (uint32_t q = (total + waves - 1) / waves; q > 0 && st_next_item(A.defer_list, A.defer_cap, own, lane, k, region, local); --q) {
                const uint32_t item = region * A.defer_cap + local;
                const uint32_t code = A.defer_list[ST_LIST_HDR + item];
                if (code == ST_REC_NONE) continue;
                st_trace_tile<MODE>(A, leaf_ro, wide_boxes, wide_vmask, kb_id, kb_t, kb_n, tab[0], tid, (int64_t)(code >> 5), (int)(code & 31u), A.n_tiles + item, region);
            }
        } else {
            uint32_t total = 0;
            for (int r = 0; r < 8; ++r) total += min(ST_CNT(lone_list, r), A.lone_list_cap);
            const uint32_t waves = (gridDim.x - ST_REST_PACKET_BLOCKS) * (ST_THREADS / 64);
            for (uint32_t q = (total + waves - 1) / waves; q > 0 && st_next_item(lone_list, A.lone_list_cap, own, lane, k, region, local); --q)
                st_trace_lone_rays<false>(A, leaf_ro, wide_boxes, wide_vmask, lone_list, slot[tid >> 6], lane, region, local, 0u);
        }
        return;
    }
    if (MODE == 0 && ST_REST_SCHED == 0) {
        const uint32_t own = blockIdx.x & 7u;
        uint32_t k = 0, region = 0, local = 0;
#if ST_REST_LONE_FIRST
        while (st_next_item(lone_list, A.lone_list_cap, own, lane, k, region, local))
            st_trace_lone_rays<false>(A, leaf_ro, wide_boxes, wide_vmask, lone_list, slot[tid >> 6], lane, region, local, 0u);
        k = 0;
#endif
        if (A.defer_list != nullptr) {
            while (st_next_item(A.defer_list, A.defer_cap, own, lane, k, region, local)) {
                const uint32_t item = region * A.defer_cap + local;
                const uint32_t code = A.defer_list[ST_LIST_HDR + item];
                if (code == ST_REC_NONE) continue;                 // a slot of a block that found the list full
                st_trace_tile<MODE>(A, leaf_ro, wide_boxes, wide_vmask, kb_id, kb_t, kb_n, tab[0], tid, (int64_t)(code >> 5), (int)(code & 31u), A.n_tiles + item, region);
            }
        }
#if !ST_REST_LONE_FIRST
        k = 0;
        while (st_next_item(lone_list, A.lone_list_cap, own, lane, k, region, local))
            st_trace_lone_rays<false>(A, leaf_ro, wide_boxes, wide_vmask, lone_list, slot[tid >> 6], lane, region, local, 0u);
#endif
        return;
    }
    // every stride-th item of the eight lists laid end to end: region r's items start at the sum of the counts before it
    const uint32_t wave = blockIdx.x * (ST_THREADS / 64) + (tid >> 6), stride = gridDim.x * (ST_THREADS / 64);
    if (MODE != 2 && A.defer_list != nullptr) {                    // (MODE 2: the replay of listed packets rides in their blocks' waves of the first launch)
        uint32_t before = 0;
        for (uint32_t region = 0; region < 8u; ++region) {
            const uint32_t listed = ST_CNT(A.defer_list, region);
            const uint32_t count = listed < A.defer_cap ? listed : A.defer_cap;
            for (uint32_t local = (wave + stride - before % stride) % stride; local < count; local += stride) {
                const uint32_t item = region * A.defer_cap + local;
                const uint32_t code = A.defer_list[ST_LIST_HDR + item];
                if (code == ST_REC_NONE) continue;
                st_trace_tile<MODE>(A, leaf_ro, wide_boxes, wide_vmask, kb_id, kb_t, kb_n, tab[MODE == 2 ? tid >> 6 : 0], tid, (int64_t)(code >> 5), (int)(code & 31u), A.n_tiles + item,
                                    region);
            }
            before += count;
        }
    }
    if (MODE == 2 && ST_REPLAY4 && !ST_LONE_IN_BLOCK) {
        // the rays whose record suffices, four to a wave; the others (more passes than the record keeps) walk again below
        uint32_t before4 = 0;
        for (uint32_t region = 0; region < 8u; ++region) {
            const uint32_t listed = ST_CNT(lone_list, region);
            st_replay_lone_rays4(A, lone_list, lane, region, (wave + stride - before4 % stride) % stride, stride);
            before4 += ((listed < A.lone_list_cap ? listed : A.lone_list_cap) + 3u) / 4u;
        }
    }
    uint32_t before = 0;
    for (uint32_t region = 0; region < 8u; ++region) {
        const uint32_t listed = ST_CNT(lone_list, region);
        st_trace_lone_rays<MODE != 0>(A, leaf_ro, wide_boxes, wide_vmask, lone_list, slot[tid >> 6], lane, region, (wave + stride - before % stride) % stride, stride,
                                      MODE == 2 && (ST_REPLAY4 || ST_LONE_IN_BLOCK));
        before += listed < A.lone_list_cap ? listed : A.lone_list_cap;
    }
}

}   // namespace

extern "C" {

struct BlobLayout { size_t perm, leaf, wide_boxes, wide_vmask, total; int n_slots; };

static BlobLayout st_blob(int64_t n_surfels)          // sorted order | surfel records in that order (filled by the trace calls) | wide boxes | wide masks
{
    const StWide w = st_wide(n_surfels);
    BlobLayout b;
    b.n_slots = 64 * w.cnt[0];
    b.perm = 0;
    b.leaf = mrgs_align_up((size_t)b.n_slots * 4, 256);
    b.wide_boxes = mrgs_align_up(b.leaf + (size_t)b.n_slots * 64, 256);
    b.wide_vmask = mrgs_align_up(b.wide_boxes + (size_t)w.total * 1536, 256);
    b.total = mrgs_align_up(b.wide_vmask + (size_t)w.total * 8, 256);
    return b;
}

size_t mrgs_surfel_bvh_bytes(int64_t n_surfels)
{
    if (n_surfels < 0) return 0;
    return st_blob(n_surfels).total;
}

struct StateLayout { int64_t grid, n_tiles; size_t lone_slot, lone, defer, rec_hdr, rec_chunks, rec_arena, lone_rec, total; uint32_t pool, defer_cap, lone_cap, lone_list_cap, region_blocks; };

static StateLayout st_state(int64_t n_rays, int32_t ray_width)      // in 4-byte words
{
    StateLayout L;
    L.n_tiles = (n_rays + 63) / 64;
    if (ray_width > 0 && n_rays % ray_width == 0) L.n_tiles = (int64_t)((ray_width + 7) / 8) * ((n_rays / ray_width + 7) / 8);
    // eight regions (StArgs): the first launch is region_blocks blocks per region (whole supertiles), every list has a part per region
    L.region_blocks = ((st_supertiles(L.n_tiles, (ray_width > 0 && n_rays % ray_width == 0) ? ray_width : 0) + 7u) / 8u) * (ST_SUPER * ST_SUPER * 64u / ST_THREADS);
    L.grid = 8 * (int64_t)L.region_blocks;
    L.defer_cap = (uint32_t)((4 * L.n_tiles + 1024 + 7) / 8);       // per region
    L.pool = (uint32_t)(3 * L.n_tiles + 64);
    L.lone_list_cap = (uint32_t)(n_rays < (int64_t)L.region_blocks * ST_THREADS ? n_rays : (int64_t)L.region_blocks * ST_THREADS);   // per region: every ray of the region
    L.lone_slot = (size_t)4 * n_rays;                                // per ray: where a ray that walks alone was listed
    L.lone = L.lone_slot + (((size_t)n_rays + 63) & ~(size_t)63);                                     // header (counts and tickets per region), ray indices region by region
    L.defer = L.lone + ST_LIST_HDR + (size_t)8 * L.lone_list_cap;             // header, packets of the second launch region by region
    L.rec_hdr = L.defer + ST_LIST_HDR + (size_t)8 * L.defer_cap;
    L.rec_chunks = L.rec_hdr + 16;                                   // one row per block of rays, then one per listed packet
    L.rec_arena = L.rec_chunks + ((size_t)L.n_tiles + (size_t)8 * L.defer_cap) * ST_REC_PASSES;
    L.lone_rec = (L.rec_arena + ((size_t)L.n_tiles * ST_REC_STATIC + L.pool) * (ST_K * 64) + 1) & ~(size_t)1;      // 8-byte keys
    L.lone_cap = (uint32_t)((ST_LONE_CAP_TILES * L.n_tiles + 1024 + 7) / 8);        // per region
    L.total = L.lone_rec + (size_t)8 * L.lone_cap * ST_LONE_REC_PASSES * ST_K * 2;
    return L;
}

size_t mrgs_surfel_trace_state_floats(int64_t n_rays, int32_t ray_width) { return n_rays < 0 ? 0 : st_state(n_rays, ray_width).total; }
size_t mrgs_surfel_trace_state_floats_norecord(int64_t n_rays, int32_t ray_width) { return n_rays < 0 ? 0 : st_state(n_rays, ray_width).rec_arena; }

// Introspection for the tests: word offsets inside `state` of [0] the lists of rays traced one per wavefront (their counts per
// region at a stride of 64 words), [1] the lists of packets handed to the second launch (counts the same way), [2] the record header (word 0: chunks taken from the shared
// pool, word 1: non-zero = the record overflowed / was not kept and the backward walks again), [3] the replay record, [4] the full size.
int mrgs_surfel_trace_state_layout(int64_t n_rays, int32_t ray_width, size_t* offsets5)
{
    if (n_rays < 0 || !offsets5) return MRGS_E_BAD_ARG;
    const StateLayout L = st_state(n_rays, (ray_width > 0 && n_rays % ray_width == 0) ? ray_width : 0);
    offsets5[0] = L.lone; offsets5[1] = L.defer; offsets5[2] = L.rec_hdr; offsets5[3] = L.rec_arena; offsets5[4] = L.total;
    return MRGS_OK;
}

// rays and blocks of rays are carried in 32-bit words (ray indices in the lists, `tile << 5 | packet` codes, the launch grid)
static bool st_ray_count_supported(int64_t n_rays, int32_t ray_width)
{
    if (n_rays >= ((int64_t)1 << 31)) return false;
    return st_state(n_rays, ray_width).n_tiles < ((int64_t)1 << 27);
}

__global__ void st_fill_bg_kernel(int64_t n_rays, float b0, float b1, float b2, float* __restrict__ rgb)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_rays) { rgb[3 * i] = b0; rgb[3 * i + 1] = b1; rgb[3 * i + 2] = b2; }
}

size_t mrgs_surfel_bvh_ws_bytes(int64_t n_surfels)
{
    if (n_surfels < 0) return 0;
    return st_build_ws(n_surfels).total;
}

int mrgs_surfel_bvh_build(const float* quad_vertices, int64_t n_surfels, void* blob, size_t blob_bytes, void* ws, size_t ws_bytes, void* stream)
{
    if (n_surfels < 0 || n_surfels > (int64_t)1 << 24) return MRGS_E_UNSUPPORTED;      // 64^4 surfels
    if (n_surfels == 0) return MRGS_OK;
    if (!quad_vertices || !blob || !ws) return MRGS_E_BAD_ARG;
    const BuildWs w = st_build_ws(n_surfels);
    if (blob_bytes < mrgs_surfel_bvh_bytes(n_surfels) || ws_bytes < w.total) return MRGS_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    char* base = (char*)ws;
    const int P = (int)n_surfels;
    if (hipMemsetAsync(base + w.zero_from, 0, w.zero_bytes, st) != hipSuccess) return MRGS_E_HIP;
    float* aabb = (float*)(base + w.aabb);
    uint32_t* bounds = (uint32_t*)(base + w.bounds);
    uint32_t* key[2] = {(uint32_t*)(base + w.key0), (uint32_t*)(base + w.key1)};
    uint32_t* val[2] = {(uint32_t*)(base + w.val0), (uint32_t*)(base + w.val1)};
    const int nb = (P + 255) / 256;
    hipLaunchKernelGGL(st_aabb_kernel, dim3(nb < ST_AABB_BLOCKS ? nb : ST_AABB_BLOCKS), dim3(256), 0, st, P, quad_vertices, aabb, bounds + 16,
                       bounds + 9, bounds);
    hipLaunchKernelGGL(st_morton_kernel, dim3(nb), dim3(256), 0, st, P, aabb, bounds, key[0], val[0]);
    const int cur = mrgs_radix_sort_pairs(key, val, (uint32_t*)(base + w.sortws), bounds + 8, n_surfels, nullptr, 0, 3 * ST_MORTON_AXIS_BITS <= 24 ? 24 : 32, st);
    if (hipMemcpyAsync(blob, val[cur], (size_t)P * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return MRGS_E_HIP;      // the sorted order
    const StWide wd = st_wide(n_surfels);
    const BlobLayout bl = st_blob(n_surfels);
    for (int l = 0; l < wd.n; ++l)
        hipLaunchKernelGGL(st_wide_level_kernel, dim3(wd.cnt[l]), dim3(64), 0, st, l, l == 0 ? P : wd.cnt[l - 1], wd, val[cur], aabb,
                           (float*)((char*)blob + bl.wide_boxes), (unsigned long long*)((char*)blob + bl.wide_vmask));
    return hipGetLastError() == hipSuccess ? MRGS_OK : MRGS_E_HIP;
}

// start-of-trace initialisation in one launch (see st_launch): `defer` = header (16 words) + list, directly followed by the record header
// (16 words) and the chunk table -- n_ff words from the list's start are set to all ones, then the three headers are written
__global__ __launch_bounds__(256) void st_init_kernel(uint32_t* __restrict__ lone_hdr, uint32_t* __restrict__ defer_hdr, uint32_t* __restrict__ rec_hdr,
                                                      size_t n_ff, uint32_t rec_flag, float* __restrict__ wet, size_t n_wet)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t* ff = defer_hdr + ST_LIST_HDR;
    // (rec_hdr lies inside [ff, ff + n_ff): its 16 words are written by the threads that own them, with the header's values)
    if (i < n_ff) {
        uint32_t* w = ff + i;
        uint32_t v = 0xFFFFFFFFu;
        if (w >= rec_hdr && w < rec_hdr + 16) v = (w == rec_hdr + 1) ? rec_flag : 0u;
        *w = v;
    }
    if (i < (size_t)ST_LIST_HDR) { lone_hdr[i] = 0u; defer_hdr[i] = 0u; }
    if (i < n_wet && wet != nullptr) wet[i] = 0.0f;
}

__global__ __launch_bounds__(256) void st_zero2_kernel(float4* __restrict__ a, size_t na, float4* __restrict__ b, size_t nb)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < na) a[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < nb) b[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

static int st_launch(bool bwd, void* blob, int64_t n_surfels, int64_t n_rays, int32_t ray_width, size_t state_floats, StArgs& a, hipStream_t st)
{
    a.n_rays = n_rays;
    const BlobLayout bl = st_blob(n_surfels);
    float4* leaf = (float4*)((char*)blob + bl.leaf);
    a.geom_leaf = leaf;
    a.wide = st_wide(n_surfels);
    const float* wide_boxes = (const float*)((char*)blob + bl.wide_boxes);
    const unsigned long long* wide_vmask = (const unsigned long long*)((char*)blob + bl.wide_vmask);
    hipLaunchKernelGGL(st_leaf_order_kernel, dim3((bl.n_slots * 4 + 255) / 256), dim3(256), 0, st, bl.n_slots, (int)n_surfels,
                       (const uint32_t*)((char*)blob + bl.perm), a.geom, leaf);
    a.ray_width = (ray_width > 0 && n_rays % ray_width == 0) ? ray_width : 0;
    static const bool no_packets = getenv("MRGS_TRACE_NO_PACKETS") != nullptr;      // developer switch: every ray gets a wave of its own
    a.packets = no_packets ? 0 : 1;
    static const char* cone_env = getenv("MRGS_TRACE_CONE");
    a.cone = cone_env ? (float)atof(cone_env) : 0.005f;
    static const char* cone1_env = getenv("MRGS_TRACE_CONE_QUAD");
    static const char* cone2_env = getenv("MRGS_TRACE_CONE_GROUP");
    a.cone_quad = cone1_env ? (float)atof(cone1_env) : a.cone;
    a.cone_group = cone2_env ? (float)atof(cone2_env) : 0.2f * a.cone;       // 2x2 groups must be tighter still: four rays rarely pay for a wide beam
    const StateLayout SL = st_state(n_rays, a.ray_width);
    if (state_floats < SL.rec_arena) return MRGS_E_WORKSPACE;
    const bool have_arena = state_floats >= SL.total;       // a state without the replay record (forward-only callers): the backward walks again
    const dim3 grid((unsigned)SL.grid), rgrid(ST_REST_SCHED == 2 ? ST_REST_PACKET_BLOCKS + 4096 : ST_REST_BLOCKS);
    uint32_t* words = reinterpret_cast<uint32_t*>(a.state);
    a.lone_list = words + SL.lone;
    a.lone_slot = words + SL.lone_slot;
    a.defer_list = words + SL.defer;
    a.defer_cap = SL.defer_cap;
    a.lone_list_cap = SL.lone_list_cap;
    a.region_blocks = SL.region_blocks;
    a.rec_hdr = words + SL.rec_hdr;
    a.rec_chunks = words + SL.rec_chunks;
    a.rec_arena = words + SL.rec_arena;
    a.rec_pool = SL.pool;
    a.lone_rec = have_arena ? reinterpret_cast<unsigned long long*>(words + SL.lone_rec) : nullptr;
    a.lone_cap = SL.lone_cap;
    a.rec_static = (uint32_t)(SL.n_tiles * ST_REC_STATIC);
    a.n_tiles = (uint32_t)SL.n_tiles;
    static const bool no_record = getenv("MRGS_TRACE_NO_RECORD") != nullptr;          // developer switch: the backward always walks again
    static const bool no_defer = getenv("MRGS_TRACE_NO_DEFER") != nullptr;            // developer switch: every block walks its own packets
    if (no_defer) a.defer_list = nullptr;
    if (no_record) a.lone_rec = nullptr;
    if (!bwd) {
        // the three list / record headers (zeros), the list of deferred packets and the record's chunk table (all ones), the surfels'
        // `wet` sums (zeros): ONE launch instead of six memsets of 4-5 us each
        const bool keep = !(no_record || !have_arena);
        if (!keep) a.rec_arena = nullptr;
        const size_t n_ff = (size_t)8 * SL.defer_cap + ((size_t)SL.n_tiles + (size_t)8 * SL.defer_cap) * ST_REC_PASSES + 16;   // defer lists .. end of rec_chunks (the header between them is rewritten)
        const size_t n_init = n_ff > (size_t)n_surfels ? n_ff : (size_t)n_surfels;
        hipLaunchKernelGGL(st_init_kernel, dim3((unsigned)((n_init + 255) / 256)), dim3(256), 0, st, a.lone_list, words + SL.defer, a.rec_hdr, n_ff,
                           keep ? 0u : 0x01010101u, a.wet, (size_t)n_surfels);
        hipLaunchKernelGGL(st_trace_kernel<0>, grid, dim3(ST_THREADS), 0, st, a, a.geom_leaf, wide_boxes, wide_vmask);
        hipLaunchKernelGGL(st_trace_rest_kernel<0>, rgrid, dim3(ST_THREADS), 0, st, a, a.geom_leaf, wide_boxes, wide_vmask, a.lone_list);
    } else {
        // exactly one of the two pairs does the work: the replay of the forward's record, or -- when the record overflowed -- the walk
        hipLaunchKernelGGL(st_trace_kernel<2>, grid, dim3(ST_THREADS), 0, st, a, a.geom_leaf, wide_boxes, wide_vmask);
        hipLaunchKernelGGL(st_trace_rest_kernel<2>, rgrid, dim3(ST_THREADS), 0, st, a, a.geom_leaf, wide_boxes, wide_vmask, a.lone_list);
        hipLaunchKernelGGL(st_trace_kernel<1>, grid, dim3(ST_THREADS), 0, st, a, a.geom_leaf, wide_boxes, wide_vmask);
        hipLaunchKernelGGL(st_trace_rest_kernel<1>, rgrid, dim3(ST_THREADS), 0, st, a, a.geom_leaf, wide_boxes, wide_vmask, a.lone_list);
    }
    return hipGetLastError() == hipSuccess ? MRGS_OK : MRGS_E_HIP;
}

int mrgs_surfel_trace_forward(void* blob, int64_t n_surfels, int64_t n_rays, int32_t ray_width, const float* ray_o, const float* ray_d, const float* geom,
                              const float* attr, const float* bg_host, float* rgb, float* dpt, float* acc, float* norm, float* dist,
                              float* aux, float* wet, float* state, size_t state_floats, void* stream)
{
    if (n_rays < 0 || n_surfels < 0 || n_surfels > (int64_t)1 << 24 || !st_ray_count_supported(n_rays, ray_width)) return MRGS_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (n_rays == 0) {                                                     // nothing traced: every surfel's weight sum is zero
        if (n_surfels > 0 && wet && hipMemsetAsync(wet, 0, (size_t)n_surfels * 4, st) != hipSuccess) return MRGS_E_HIP;
        return MRGS_OK;
    }
    if (!bg_host || !rgb || !dpt || !acc || !norm || !dist || !aux) return MRGS_E_BAD_ARG;
    if (n_surfels > 0 && (!ray_o || !ray_d || !state || !blob || !geom || !attr || !wet)) return MRGS_E_BAD_ARG;
    if (n_surfels == 0) {                                                  // nothing to hit: background everywhere
        if (hipMemsetAsync(dpt, 0, n_rays * 4, st) != hipSuccess || hipMemsetAsync(acc, 0, n_rays * 4, st) != hipSuccess ||
            hipMemsetAsync(norm, 0, n_rays * 12, st) != hipSuccess || hipMemsetAsync(dist, 0, n_rays * 4, st) != hipSuccess ||
            hipMemsetAsync(aux, 0, n_rays * 8, st) != hipSuccess)
            return MRGS_E_HIP;
        hipLaunchKernelGGL(st_fill_bg_kernel, dim3((unsigned)((n_rays + 255) / 256)), dim3(256), 0, st, n_rays, bg_host[0], bg_host[1], bg_host[2], rgb);
        return hipGetLastError() == hipSuccess ? MRGS_OK : MRGS_E_HIP;    // (an empty model, e.g. after pruning everything; `state` is not touched)
    }
    StArgs a;
    std::memset(&a, 0, sizeof(a));
    a.ray_o = ray_o; a.ray_d = ray_d; a.geom = (const float4*)geom; a.attr = (const float4*)attr;
    a.bg[0] = bg_host[0]; a.bg[1] = bg_host[1]; a.bg[2] = bg_host[2];
    a.rgb = rgb; a.dpt = dpt; a.acc = acc; a.norm = norm; a.dist = dist; a.aux = aux; a.wet = wet; a.state = state;
    return st_launch(false, blob, n_surfels, n_rays, ray_width, state_floats, a, st);
}

int mrgs_surfel_trace_backward(void* blob, int64_t n_surfels, int64_t n_rays, int32_t ray_width, const float* ray_o, const float* ray_d, const float* geom,
                               const float* attr, const float* bg_host, const float* rgb, const float* dpt, const float* acc,
                               const float* norm, const float* aux, const float* state, size_t state_floats, const float* g_rgb, const float* g_dpt,
                               const float* g_acc, const float* g_norm, const float* g_dist, const float* g_aux, float* g_geom,
                               float* g_attr, float* g_ray_o, float* g_ray_d, void* stream)
{
    if (n_rays < 0 || n_surfels <= 0 || n_surfels > (int64_t)1 << 24 || !st_ray_count_supported(n_rays, ray_width)) return MRGS_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (!g_geom || !g_attr) return MRGS_E_BAD_ARG;
    if ((((uintptr_t)g_geom | (uintptr_t)g_attr) & 15u) == 0) {                                             // one launch, not two memsets
        hipLaunchKernelGGL(st_zero2_kernel, dim3((unsigned)(((size_t)n_surfels * 4 + 255) / 256)), dim3(256), 0, st, reinterpret_cast<float4*>(g_geom),
                           (size_t)n_surfels * 4, reinterpret_cast<float4*>(g_attr), (size_t)n_surfels * 2);
        if (hipGetLastError() != hipSuccess) return MRGS_E_HIP;
    } else if (hipMemsetAsync(g_geom, 0, (size_t)n_surfels * 64, st) != hipSuccess || hipMemsetAsync(g_attr, 0, (size_t)n_surfels * 32, st) != hipSuccess) {
        return MRGS_E_HIP;
    }
    if (n_rays == 0) return MRGS_OK;
    if (!blob || !ray_o || !ray_d || !geom || !attr || !bg_host || !rgb || !dpt || !acc || !norm || !aux || !state || !g_ray_o || !g_ray_d)
        return MRGS_E_BAD_ARG;                                  // (any of the six upstream gradients may be NULL = zeros)
    StArgs a;
    std::memset(&a, 0, sizeof(a));
    a.ray_o = ray_o; a.ray_d = ray_d; a.geom = (const float4*)geom; a.attr = (const float4*)attr;
    a.bg[0] = bg_host[0]; a.bg[1] = bg_host[1]; a.bg[2] = bg_host[2];
    a.rgb = const_cast<float*>(rgb); a.dpt = const_cast<float*>(dpt); a.acc = const_cast<float*>(acc); a.norm = const_cast<float*>(norm);
    a.aux = const_cast<float*>(aux); a.state = const_cast<float*>(state);
    a.g_rgb = g_rgb; a.g_dpt = g_dpt; a.g_acc = g_acc; a.g_norm = g_norm; a.g_dist = g_dist; a.g_aux = g_aux;
    a.g_geom = g_geom; a.g_attr = g_attr; a.g_ray_o = g_ray_o; a.g_ray_d = g_ray_d;
    return st_launch(true, blob, n_surfels, n_rays, ray_width, state_floats, a, st);
}

}   // extern "C"
