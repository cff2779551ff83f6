// mrgs_binning.hip -- tile binning of the surfel rasterizer on gfx950 without a global sort.
//
// Reference behaviour being reproduced (rasterizer_impl.cu:283-324): point_list = gaussian ids ordered by
// (tile id, raw depth bits), ties in emission order (gaussian index); ranges[tile] = its slice of the list.  The reference
// (and round 1 of this library, mrgs_sort.hip) gets there with global radix passes over all R (tile, gaussian) pairs plus an
// inclusive scan and a blocking read-back.  The order inside a tile is a TOTAL order on (depth bits, gaussian index), so it
// does not matter in which order the pairs arrive in their tile's segment -- only that the segment is sorted afterwards:
//
//   tile_count_kernel   G workgroups, each over a fixed slice of the surfels: per-tile pair counts of the slice in an LDS
//                       histogram (LDS atomics only), written out as row g of a [G][tiles] matrix
//   tile_scan_kernel    one thread per tile: exclusive prefix of its column (= where each slice's pairs start inside the tile's
//                       segment), tile totals, exclusive scan over the tiles (per 256-tile chunk; the last workgroup to finish
//                       scans the chunk totals), num_rendered for the host
//   tile_emit_kernel    same slices: LDS cursors = segment start + column prefix; every pair takes its slot with ONE LDS atomic and is
//                       stored as a 64-bit key  depth bits << 32 | gaussian index << 4 | quadrant cull bits  -- no global atomic, no
//                       look-back, no dependence between workgroups.  The cull of the surfel against the four 8x8 quadrants of the
//                       tile is evaluated here, where the surfel's conic is in registers (round 1 gathered it per list entry)
//   tile_sort_kernel    one workgroup per tile (up to 4 096 keys): bucket sort in LDS with an order-preserving linear hash of the depth
//                       (count, scan, drop into the bucket's range) and an exact ordering of the handful of keys inside each
//                       bucket; writes point_list, the cull bits, ranges[tile] and the per-quadrant survivor counts (the forward's
//                       work estimate)
//   tile_sort_big_kernel  the rare tiles beyond 4 096 keys (a device-side list): up to 16 384 keys in LDS, beyond that the outer
//                       network stages run on global memory
//
// Five short launches and ~45 MB of traffic at C2 (P = 300k, R = 1.15 M) where the radix pipeline had eleven launches and
// 116 MB; point_list, ranges and n_contrib stay bit-identical to the reference's 64-bit-key sort (tests/test_gpu_parity.py).
#include "mrgs_blend_math.h"

#define BIN_THREADS 1024
#define SORT_SMALL_CAP 4096       // keys per tile the bucket sort of tile_sort_kernel holds in LDS
#define BIN_CHUNK 64              // tiles per workgroup of tile_scan_kernel = granularity of tile_loc / chunk_base
#define SORT_BIG_THREADS 1024
#define SORT_BIG_CAP 16384

namespace {

__device__ __forceinline__ uint32_t ld_agent(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Visits every tile of the rectangles of one wave's surfels: lanes with few tiles walk them serially, a surfel with many tiles
// (a heavy-tailed scene has splats of hundreds of tiles) is spread over the 64 lanes.  own() / take(src) switch the caller's
// "current surfel" to the lane's own one / to the one of lane src; take runs in wave-uniform control flow (all lanes active, src
// uniform), so it may use readlane broadcasts.  f(tile) is then called once per tile of the current surfel.
#define BIN_COOP_MIN 24
template <typename Own, typename Take, typename F>
__device__ __forceinline__ void for_each_tile(bool have, uint2 r, int tiles_x, Own own, Take take, F f)
{
    const int lane = threadIdx.x & 63;
    const int x0 = r.x & 0xFFFF, y0 = r.x >> 16, x1 = r.y & 0xFFFF, y1 = r.y >> 16;
    const int w = x1 - x0, n = have ? w * (y1 - y0) : 0;
    const bool big = n >= BIN_COOP_MIN;
    own();
    if (have && !big)
        for (int y = y0; y < y1; y++)
            for (int x = x0; x < x1; x++) f(y * tiles_x + x);
    uint64_t todo = __builtin_amdgcn_ballot_w64(big);
    while (todo != 0ull) {
        const int src = __builtin_ctzll(todo);
        todo &= todo - 1ull;
        const int sx0 = __builtin_amdgcn_readlane(x0, src), sy0 = __builtin_amdgcn_readlane(y0, src);
        const int sw = __builtin_amdgcn_readlane(w, src), sn = __builtin_amdgcn_readlane(n, src);
        take(src);
        for (int k = lane; k < sn; k += 64) {
            const int yy = k / sw;
            f((sy0 + yy) * tiles_x + sx0 + (k - yy * sw));
        }
    }
}

__global__ void __launch_bounds__(BIN_THREADS) tile_count_kernel(int P, int per_group, const uint32_t* __restrict__ tiles_touched,
                                                                 const uint2* __restrict__ rect, int tiles_x, int T, int Tpad,
                                                                 uint32_t* __restrict__ mat)
{
    extern __shared__ uint32_t s_cnt[];
    for (int t = threadIdx.x; t < T; t += BIN_THREADS) s_cnt[t] = 0u;
    __syncthreads();
    const int beg = blockIdx.x * per_group, end = min(P, beg + per_group);
    for (int i0 = beg; i0 < end; i0 += BIN_THREADS) {          // wave-uniform trip count: the cooperative part needs all lanes
        const int i = i0 + threadIdx.x;
        const bool have = i < end && tiles_touched[i] != 0u;
        const uint2 r = have ? rect[i] : make_uint2(0u, 0u);
        for_each_tile(have, r, tiles_x, [] {}, [](int) {}, [&](int tile) { atomicAdd(&s_cnt[tile], 1u); });
    }
    __syncthreads();
    uint32_t* row = mat + (size_t)blockIdx.x * Tpad;
    for (int t = threadIdx.x; t < T; t += BIN_THREADS) row[t] = s_cnt[t];
}

// Workgroup = 64 consecutive tiles x 16 waves; wave w owns the slice rows [w * rpw, (w + 1) * rpw), lane = tile.
// mat[g][t]: in: pairs of slice g in tile t; out: pairs of slices < g in tile t (all loads of a lane in flight at once).
// tile_cnt[t] = pairs of tile t; tile_loc[t] = exclusive scan of tile_cnt inside the tile's 64-tile chunk; chunk_base[c] = pairs of the
// chunks before c (written by the last workgroup to finish); state[0] = num_rendered, state[2] = ticket (zero on entry).
#define SCAN_ROWS_MAX 16          // rows per wave: mrgs_bin_groups() <= 256 slices / 16 waves
__global__ void __launch_bounds__(1024) tile_scan_kernel(int G, int T, int Tpad, uint32_t* __restrict__ mat, uint32_t* __restrict__ tile_cnt,
                                                         uint32_t* __restrict__ tile_loc, uint32_t* __restrict__ chunk_tot,
                                                         uint32_t* __restrict__ chunk_base, uint32_t* __restrict__ state,
                                                         uint32_t* __restrict__ host_slot)
{
    __shared__ uint32_t part[16][64];
    __shared__ uint32_t wave_sums[16];
    __shared__ int s_last;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = blockIdx.x * BIN_CHUNK + lane;
    const int rpw = (G + 15) >> 4, g0 = wave * rpw;
    uint32_t v[SCAN_ROWS_MAX];
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ROWS_MAX; k++) {
        v[k] = (k < rpw && g0 + k < G && t < T) ? mat[(size_t)(g0 + k) * Tpad + t] : 0u;
    }
#pragma unroll
    for (int k = 0; k < SCAN_ROWS_MAX; k++) sum += v[k];
    part[wave][lane] = sum;
    __syncthreads();
    uint32_t run = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) { const uint32_t pw = part[w][lane]; if (w < wave) run += pw; total += pw; }
#pragma unroll
    for (int k = 0; k < SCAN_ROWS_MAX; k++) {
        if (k < rpw && g0 + k < G && t < T) mat[(size_t)(g0 + k) * Tpad + t] = run;
        run += v[k];
    }
    if (wave == 0) {
        // exclusive scan of the 64 tile totals of the chunk
        uint32_t inc = total;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t u = __shfl_up(inc, d, 64);
            if (lane >= d) inc += u;
        }
        if (t < T) { tile_cnt[t] = total; tile_loc[t] = inc - total; }
        if (lane == 63) {
            st_agent(chunk_tot + blockIdx.x, inc);
            __threadfence();
            s_last = atomicAdd(state + 2, 1u) == gridDim.x - 1 ? 1 : 0;
        }
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    // last workgroup: exclusive scan over the chunk totals (<= 576 chunks for the tile counts this path serves)
    uint32_t carry = 0;
    for (int c0 = 0; c0 < (int)gridDim.x; c0 += 1024) {
        const int c = c0 + threadIdx.x;
        const uint32_t val = c < (int)gridDim.x ? ld_agent(chunk_tot + c) : 0u;
        uint32_t in2 = val;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t u = __shfl_up(in2, d, 64);
            if (lane >= d) in2 += u;
        }
        __syncthreads();
        if (lane == 63) wave_sums[wave] = in2;
        __syncthreads();
        uint32_t o2 = 0, t2 = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) { const uint32_t sw = wave_sums[w]; if (w < wave) o2 += sw; t2 += sw; }
        if (c < (int)gridDim.x) chunk_base[c] = carry + o2 + in2 - val;
        carry += t2;
    }
    if (threadIdx.x == 0) {
        state[0] = carry;                             // num_rendered (pair counts are checked against 2^30 by the host)
        if (host_slot != nullptr) {                   // pinned, device-mapped: the host waits for the event recorded behind this kernel
            __hip_atomic_store(host_slot, carry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(host_slot + 1, state[1], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// Per wave of tile_emit_kernel: the 64 surfels of a step publish their record in LDS, and the (surfel, tile) pairs of the surfels
// with fewer than BIN_COOP_MIN tiles are numbered consecutively (wave prefix sum); lane l then takes pair 64 j + l -- every lane has
// work in every iteration but the last.  (One lane walking the tiles of its own surfel ran the wave for as many iterations as its
// largest surfel has tiles, with a quarter of the lanes busy on average: 190 instructions per iteration, 10.6 M VALU instructions per
// launch at C2.)
#define EMIT_REC_WORDS 16                       // conic 12 | depth bits | rect.x | rect.y | first pair number
#define EMIT_PAIRS_MAX (64 * (BIN_COOP_MIN - 1))
#define EMIT_WAVE_BYTES (64 * EMIT_REC_WORDS * 4 + ((EMIT_PAIRS_MAX + 15) & ~15))
__global__ void __launch_bounds__(BIN_THREADS) tile_emit_kernel(int P, int per_group, const uint32_t* __restrict__ tiles_touched,
                                                                const uint2* __restrict__ rect, const uint32_t* __restrict__ depth_key,
                                                                const float4* __restrict__ cull, int tiles_x, int T, int Tpad,
                                                                const uint32_t* __restrict__ mat, const uint32_t* __restrict__ tile_loc,
                                                                const uint32_t* __restrict__ chunk_base, const uint32_t* __restrict__ state,
                                                                int64_t capacity, unsigned long long* __restrict__ pairs, uint32_t* __restrict__ census,
                                                                uint32_t* __restrict__ big_count)
{
    extern __shared__ uint32_t s_cur[];                 // [T] cursors, then per wave: records [64][12] and owner lane of each pair
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) mrgs_census_mark(census);
    if (blockIdx.x == 0 && threadIdx.x == 0) big_count[0] = 0u;   // list of oversized tiles of tile_sort_kernel: empty
    if ((int64_t)state[0] > capacity) return;          // binning workspace sized from a guess that was too small: the host redoes this phase
    const uint32_t* row = mat + (size_t)blockIdx.x * Tpad;
    for (int t = threadIdx.x; t < T; t += BIN_THREADS) s_cur[t] = chunk_base[t / BIN_CHUNK] + tile_loc[t] + row[t];
    __syncthreads();
    char* wbase = (char*)(s_cur + ((T + 3) & ~3)) + (size_t)wave * EMIT_WAVE_BYTES;
    uint32_t* wrec = (uint32_t*)wbase;
    uint8_t* wown = (uint8_t*)(wbase + 64 * EMIT_REC_WORDS * 4);
    auto emit = [&](const CullConic& cc, uint32_t d, uint32_t id, int tile) {
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
        const float x0 = (float)(tx * MRGS_BLOCK_X), y0 = (float)(ty * MRGS_BLOCK_Y);
        uint32_t m = 0;
#pragma unroll
        for (int q = 0; q < 4; q++)
            m |= mrgs_block_may_touch(cc, x0 + (float)(8 * (q & 1)), y0 + (float)(8 * (q >> 1)), 7.0f, 7.0f) ? (1u << q) : 0u;
        const uint32_t pos = atomicAdd(&s_cur[tile], 1u);
        pairs[pos] = ((unsigned long long)d << 32) | (unsigned long long)((id << 4) | m);
    };
    const int beg = blockIdx.x * per_group, end = min(P, beg + per_group);
    for (int i0 = beg; i0 < end; i0 += BIN_THREADS) {
        const int i = i0 + threadIdx.x;
        const bool have = i < end && tiles_touched[i] != 0u;
        uint2 r = make_uint2(0u, 0u);
        CullConic c = mrgs_cull_never();
        uint32_t dk = 0;
        if (have) { r = rect[i]; c = mrgs_cull_load(cull, (uint32_t)i); dk = depth_key[i]; }
        const int x0 = r.x & 0xFFFF, y0 = r.x >> 16, x1 = r.y & 0xFFFF, y1 = r.y >> 16;
        const int w = x1 - x0, n = have ? w * (y1 - y0) : 0;
        const bool big = n >= BIN_COOP_MIN;
        const uint32_t small_n = big ? 0u : (uint32_t)n;
        uint32_t incl = small_n;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t u = (uint32_t)__shfl_up((int)incl, d, 64);
            if (lane >= d) incl += u;
        }
        const uint32_t excl = incl - small_n;
        const int total = __builtin_amdgcn_readlane((int)incl, 63);
        // records of this step's 64 surfels, and the owner lane of every numbered pair
        uint4* rec4 = (uint4*)(wrec + lane * EMIT_REC_WORDS);
        rec4[0] = make_uint4(__float_as_uint(c.a.x), __float_as_uint(c.a.y), __float_as_uint(c.a.z), __float_as_uint(c.a.w));
        rec4[1] = make_uint4(__float_as_uint(c.b.x), __float_as_uint(c.b.y), __float_as_uint(c.b.z), __float_as_uint(c.b.w));
        rec4[2] = make_uint4(__float_as_uint(c.c.x), __float_as_uint(c.c.y), __float_as_uint(c.c.z), __float_as_uint(c.c.w));
        rec4[3] = make_uint4(dk, r.x, r.y, excl);
        for (uint32_t k = 0; k < small_n; k++) wown[excl + k] = (uint8_t)lane;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int base = 0; base < total; base += 64) {
            const int p = base + lane;
            if (p < total) {
                const int o = (int)wown[p];
                const uint4* q4 = (const uint4*)(wrec + o * EMIT_REC_WORDS);
                const uint4 qa = q4[0], qb = q4[1], qd = q4[2], qc = q4[3];
                CullConic cc;
                cc.a = make_float4(__uint_as_float(qa.x), __uint_as_float(qa.y), __uint_as_float(qa.z), __uint_as_float(qa.w));
                cc.b = make_float4(__uint_as_float(qb.x), __uint_as_float(qb.y), __uint_as_float(qb.z), __uint_as_float(qb.w));
                cc.c = make_float4(__uint_as_float(qd.x), __uint_as_float(qd.y), __uint_as_float(qd.z), __uint_as_float(qd.w));
                const int ox0 = qc.y & 0xFFFF, oy0 = qc.y >> 16, ow = (int)(qc.z & 0xFFFF) - ox0;
                const int k = p - (int)qc.w, yy = k / ow;
                emit(cc, qc.x, (uint32_t)(i0 + (int)(threadIdx.x & ~63u) + o), (oy0 + yy) * tiles_x + ox0 + (k - yy * ow));
            }
        }
        // surfels with many tiles (a heavy-tailed scene has splats of hundreds of tiles): one at a time, spread over the 64 lanes
        uint64_t todo = __builtin_amdgcn_ballot_w64(big);
        auto rl = [](float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); };
        while (todo != 0ull) {
            const int src = __builtin_ctzll(todo);
            todo &= todo - 1ull;
            const int sx0 = __builtin_amdgcn_readlane(x0, src), sy0 = __builtin_amdgcn_readlane(y0, src);
            const int sw = __builtin_amdgcn_readlane(w, src), sn = __builtin_amdgcn_readlane(n, src);
            CullConic cc;
            cc.a = make_float4(rl(c.a.x, src), rl(c.a.y, src), rl(c.a.z, src), rl(c.a.w, src));
            cc.b = make_float4(rl(c.b.x, src), rl(c.b.y, src), rl(c.b.z, src), rl(c.b.w, src));
            cc.c = make_float4(rl(c.c.x, src), rl(c.c.y, src), rl(c.c.z, src), rl(c.c.w, src));
            const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)dk, src);
            const uint32_t id = (uint32_t)(i0 + (int)(threadIdx.x & ~63u) + src);
            for (int k = lane; k < sn; k += 64) {
                const int yy = k / sw;
                emit(cc, d, id, (sy0 + yy) * tiles_x + sx0 + (k - yy * sw));
            }
        }
        __builtin_amdgcn_wave_barrier();                 // the next step overwrites the wave's records
    }
}

// ---- bitonic network with ascending comparators only (first step of a merge level mirrors, the rest are half-cleaners): a
// virtual +inf padding beyond n then never moves, so comparators whose upper element is >= n are skipped and nothing is padded.
template <int THREADS>
__device__ __forceinline__ void lds_levels(unsigned long long* s, int n, int k_from, int k_to)
{
    // merge levels k = k_from .. k_to (powers of two, <= capacity) on s[0, n)
    for (int k = k_from; k <= k_to; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            const bool flip = j == (k >> 1);
            for (int i = threadIdx.x; 2 * i < n + j; i += THREADS) {     // comparators whose lower element can exist
                int lo, hi;
                if (flip) { const int blk = i / j, o = i - blk * j; lo = blk * k + o; hi = blk * k + k - 1 - o; }
                else { lo = ((i & ~(j - 1)) << 1) | (i & (j - 1)); hi = lo | j; }
                if (hi < n) {
                    const unsigned long long a = s[lo], b = s[hi];
                    if (a > b) { s[lo] = b; s[hi] = a; }
                }
            }
            __syncthreads();
        }
    }
}

// half-cleaner steps j = j_from .. 1 of level k on s[0, m) where s holds the elements [base, base + m) of the tile (m = chunk)
template <int THREADS>
__device__ __forceinline__ void lds_tail(unsigned long long* s, int m, int j_from)
{
    for (int j = j_from; j > 0; j >>= 1) {
        for (int i = threadIdx.x; 2 * i < m + j; i += THREADS) {
            const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1)), hi = lo | j;
            if (hi < m) {
                const unsigned long long a = s[lo], b = s[hi];
                if (a > b) { s[lo] = b; s[hi] = a; }
            }
        }
        __syncthreads();
    }
}

__device__ __forceinline__ int next_pow2(int n) { int p = 1; while (p < n) p <<= 1; return p; }

// sorted keys of one tile (in LDS: s, or in global memory: gkeys) -> point_list, cull bits, per-quadrant counts
template <int THREADS>
__device__ __forceinline__ void write_tile(const unsigned long long* src, int n, uint32_t beg, int tile, uint32_t* __restrict__ plist,
                                           uint8_t* __restrict__ qmask, uint32_t* __restrict__ item_est, uint32_t* s_q /*[4]*/)
{
    if (threadIdx.x < 4) s_q[threadIdx.x] = 0u;
    __syncthreads();
    uint32_t cq[4] = {0u, 0u, 0u, 0u};
    for (int e0 = 0; e0 < n; e0 += THREADS) {
        const int e = e0 + threadIdx.x;
        uint32_t m = 0;
        if (e < n) {
            const uint32_t low = (uint32_t)src[e];
            m = low & 15u;
            plist[beg + e] = low >> 4;
            qmask[beg + e] = (uint8_t)m;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) cq[q] += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64((m >> q) & 1u));
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int q = 0; q < 4; q++) if (cq[q]) atomicAdd(&s_q[q], cq[q]);
    }
    __syncthreads();
    if (threadIdx.x < 4) item_est[tile * 4 + threadIdx.x] = s_q[threadIdx.x];
}

// a < b on 64-bit keys through 32-bit compares
__device__ __forceinline__ bool lt64(unsigned long long a, unsigned long long b)
{
    const uint32_t ah = (uint32_t)(a >> 32), bh = (uint32_t)(b >> 32), al = (uint32_t)a, bl = (uint32_t)b;
    return (ah < bh) | ((ah == bh) & (al < bl));
}

// per-quadrant survivor counts of the keys a lane holds, packed 4 x 16 bits (a tile of this kernel holds <= 4 096 keys)
__device__ __forceinline__ unsigned long long quad_counts(uint32_t m)
{
    return (unsigned long long)(m & 1u) | ((unsigned long long)((m >> 1) & 1u) << 16) | ((unsigned long long)((m >> 2) & 1u) << 32) |
           ((unsigned long long)((m >> 3) & 1u) << 48);
}
__device__ __forceinline__ unsigned long long wave_sum64(unsigned long long v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// One workgroup per tile: bucket sort with an order-preserving linear hash.  The keys of a tile are (depth bits, gaussian index,
// cull bits); its depths fill some interval [dmin, dmax].  bucket(key) = floor((depth - dmin) * NB / (dmax - dmin + 1)) is monotone
// in the depth, so sorting = counting the keys of every bucket (LDS atomics), an exclusive scan over the NB counters, dropping every
// key into its bucket's range, and ordering the few keys INSIDE a bucket exactly: a key's final position is its bucket's start plus
// the number of smaller keys in the bucket (they are distinct: they carry the gaussian index), found by walking the bucket.  With
// NB = 4 096 buckets for at most 4 096 keys a bucket holds a handful of keys, and the whole sort is ~100 lane instructions per key --
// a comparator network on 64-bit keys (a bitonic sort in registers with lane-XOR exchanges, or in LDS) measured 52 - 92 us at C2 for
// the same job, of which every variant was instruction issue.  Equal depths share a bucket whatever NB is; the walk makes that exact,
// at quadratic cost in the number of EQUAL depths of one tile.
#define BS_THREADS 512
#define BS_NB 4096
#define BS_PER_THREAD (SORT_SMALL_CAP / BS_THREADS)
#define BS_OWN (BS_NB / BS_THREADS)                                       // consecutive counters a thread owns in the scan
__device__ __forceinline__ int bs_pad(int b) { return b + b / BS_OWN; }    // ... read without bank conflicts (lane stride BS_OWN + 1 words)
// (amdgpu_waves_per_eu: without the cap the compiler has been seen to spend 237 VGPRs on this kernel -- one workgroup per CU -- and the
// sort took three times as long)
#ifndef MRGS_TILE_SORT_WAVES
#define MRGS_TILE_SORT_WAVES 4
#endif
__global__ void __launch_bounds__(BS_THREADS) __attribute__((amdgpu_waves_per_eu(MRGS_TILE_SORT_WAVES, 8))) tile_sort_kernel(int T, const uint32_t* __restrict__ tile_cnt, const uint32_t* __restrict__ tile_loc,
                                                               const uint32_t* __restrict__ chunk_base, uint32_t* __restrict__ state,
                                                               int64_t capacity, const unsigned long long* __restrict__ pairs,
                                                               uint32_t* __restrict__ plist, uint8_t* __restrict__ qmask,
                                                               uint2* __restrict__ ranges, uint32_t* __restrict__ item_est,
                                                               uint32_t* __restrict__ big_list, uint32_t* __restrict__ census,
                                                               uint32_t* __restrict__ reuse_state, uint32_t* __restrict__ reuse_bwd, uint32_t* __restrict__ redo_count,
                                                               uint32_t* __restrict__ item_work,
                                                               float4* __restrict__ bulk_zero, size_t bulk_zero_f4)
{
    // No ordering launch follows when the forward reuses the queues dealt at an earlier visit of the camera (reuse_state = the forward's queue state in the camera's hint buffer, reuse_bwd = the backward's in this render's workspace):
    // its side jobs are done here -- tickets of both blend kernels back to zero, the backward's queue shape = the forward's, the work
    // records of this tile's four items cleared, and this workgroup's share of the coming backward's gradient rows zeroed
    if (reuse_state != nullptr) {
        const int t = (int)blockIdx.x, T_ = (int)gridDim.x;
        for (int w = t * BS_THREADS + (int)threadIdx.x; w < 8 * MRGS_MAX_SIMD_QUEUES; w += T_ * BS_THREADS) {
            reuse_state[MRGS_QS_FWD + MRGS_QS_TICKET + w] = 0u;
            reuse_bwd[MRGS_QS_TICKET + w] = 0u;
        }
        if (t == 0 && threadIdx.x < 16) reuse_bwd[threadIdx.x] = reuse_state[MRGS_QS_FWD + threadIdx.x];     // COUNT[8] | PASSES[8]
        if (t == 0 && threadIdx.x == 16) redo_count[0] = 0u;         // (the list of marked pixels of the MRGS_FWD_REDO_INLINE=0 build)
        if (threadIdx.x < 4) item_work[t * 4 + threadIdx.x] = 0u;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        const size_t per = (bulk_zero_f4 + T_ - 1) / T_, z0 = per * t, z1 = z0 + per < bulk_zero_f4 ? z0 + per : bulk_zero_f4;
        for (size_t i = z0 + threadIdx.x; i < z1; i += BS_THREADS) bulk_zero[i] = z;
    }
    __shared__ unsigned long long s_tmp[SORT_SMALL_CAP];          // the keys in bucket order
    __shared__ uint32_t s_cnt[BS_NB + BS_NB / BS_OWN];             // counts -> bucket starts -> bucket ends (padded index: bs_pad)
    __shared__ uint32_t s_lo[BS_THREADS / 64], s_hi[BS_THREADS / 64], s_wsum[BS_THREADS / 64], s_q[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (lane == 0) mrgs_census_mark(census);
    const int tile = blockIdx.x;
    // binning workspace sized from a guess that was too small: nothing was emitted; the blend kernels queued behind this one must
    // find empty lists (the host redoes the phase on an exactly sized workspace)
    const int n = (int64_t)state[0] > capacity ? 0 : (int)tile_cnt[tile];
    const uint32_t beg = chunk_base[tile / BIN_CHUNK] + tile_loc[tile];
    if (tid == 0) {
        ranges[tile] = n ? make_uint2(beg, beg + (uint32_t)n) : make_uint2(0u, 0u);   // empty tiles read (0, 0), rasterizer_impl.cu:316
        if (n > SORT_SMALL_CAP) big_list[atomicAdd(state + 6, 1u)] = (uint32_t)tile;   // rare: handed to tile_sort_big_kernel
    }
    if (n == 0 && tid < 4) item_est[tile * 4 + tid] = 0u;
    if (n == 0 || n > SORT_SMALL_CAP) return;
    const unsigned long long* src = pairs + beg;
    // 1. keys into registers (consecutive threads read consecutive keys), depth range of the tile
    unsigned long long key[BS_PER_THREAD];
    uint32_t dmin = 0xFFFFFFFFu, dmax = 0u;
#pragma unroll
    for (int i = 0; i < BS_PER_THREAD; i++) {
        const int e = i * BS_THREADS + tid;
        key[i] = 0ull;
        if (i * BS_THREADS < n && e < n) {
            key[i] = src[e];
            const uint32_t d = (uint32_t)(key[i] >> 32);
            dmin = min(dmin, d); dmax = max(dmax, d);
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { dmin = min(dmin, (uint32_t)__shfl_xor((int)dmin, d, 64)); dmax = max(dmax, (uint32_t)__shfl_xor((int)dmax, d, 64)); }
    if (lane == 0) { s_lo[wave] = dmin; s_hi[wave] = dmax; }
#pragma unroll
    for (int i = 0; i < (BS_NB + BS_NB / BS_OWN + BS_THREADS - 1) / BS_THREADS; i++) {
        const int k = i * BS_THREADS + tid;
        if (k < BS_NB + BS_NB / BS_OWN) s_cnt[k] = 0u;
    }
    if (tid < 4) s_q[tid] = 0u;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < BS_THREADS / 64; w++) { dmin = min(dmin, s_lo[w]); dmax = max(dmax, s_hi[w]); }
    const float scale = (float)BS_NB / ((float)(dmax - dmin) + 1.0f);
    // 2. count the keys of every bucket
    int bucket[BS_PER_THREAD];
#pragma unroll
    for (int i = 0; i < BS_PER_THREAD; i++) {
        bucket[i] = 0;
        if (i * BS_THREADS < n && i * BS_THREADS + tid < n) {
            // monotone in the depth: u32 -> float conversion, the product with a positive constant and the truncation are all monotone
            bucket[i] = min((int)((float)((uint32_t)(key[i] >> 32) - dmin) * scale), BS_NB - 1);
            atomicAdd(&s_cnt[bs_pad(bucket[i])], 1u);
        }
    }
    __syncthreads();
    // 3. exclusive scan over the buckets: thread t owns the BS_OWN consecutive buckets [BS_OWN t, BS_OWN t + BS_OWN)
    {
        uint32_t c[BS_OWN], sum = 0;
#pragma unroll
        for (int k = 0; k < BS_OWN; k++) { c[k] = s_cnt[bs_pad(BS_OWN * tid + k)]; sum += c[k]; }
        uint32_t inc = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t u = (uint32_t)__shfl_up((int)inc, d, 64);
            if (lane >= d) inc += u;
        }
        if (lane == 63) s_wsum[wave] = inc;
        __syncthreads();
        uint32_t run = inc - sum;
#pragma unroll
        for (int w = 0; w < BS_THREADS / 64; w++) if (w < wave) run += s_wsum[w];
#pragma unroll
        for (int k = 0; k < BS_OWN; k++) { s_cnt[bs_pad(BS_OWN * tid + k)] = run; run += c[k]; }
    }
    __syncthreads();
    // 4. every key into its bucket's range (any order inside the bucket); the counter of a bucket ends up at the bucket's END
#pragma unroll
    for (int i = 0; i < BS_PER_THREAD; i++) {
        if (i * BS_THREADS < n && i * BS_THREADS + tid < n) {
            const uint32_t pos = atomicAdd(&s_cnt[bs_pad(bucket[i])], 1u);
            s_tmp[pos] = key[i];
        }
    }
    __syncthreads();
    // 5. exact position = bucket start + number of smaller keys in the bucket; straight to the point list
    unsigned long long cnt = 0ull;
#pragma unroll
    for (int i = 0; i < BS_PER_THREAD; i++) {
        if (i * BS_THREADS < n && i * BS_THREADS + tid < n) {
            const int b = bucket[i];
            const uint32_t lo = b ? s_cnt[bs_pad(b - 1)] : 0u, hi = s_cnt[bs_pad(b)];
            uint32_t rank = lo;
            for (uint32_t m = lo; m < hi; m++) rank += lt64(s_tmp[m], key[i]) ? 1u : 0u;
            const uint32_t low = (uint32_t)key[i];
            plist[beg + rank] = low >> 4;
            qmask[beg + rank] = (uint8_t)(low & 15u);
            cnt += quad_counts(low & 15u);
        }
    }
    cnt = wave_sum64(cnt);
    if (lane < 4) {
        const uint32_t c = (uint32_t)(cnt >> (16 * lane)) & 0xFFFFu;
        if (c) atomicAdd(&s_q[lane], c);
    }
    __syncthreads();
    if (tid < 4) item_est[tile * 4 + tid] = s_q[tid];
}

__global__ void __launch_bounds__(SORT_BIG_THREADS) tile_sort_big_kernel(const uint32_t* __restrict__ tile_cnt, const uint32_t* __restrict__ tile_loc,
                                                                         const uint32_t* __restrict__ chunk_base, const uint32_t* __restrict__ state,
                                                                         int64_t capacity, unsigned long long* __restrict__ pairs,
                                                                         uint32_t* __restrict__ plist, uint8_t* __restrict__ qmask,
                                                                         uint32_t* __restrict__ item_est, const uint32_t* __restrict__ big_list)
{
    extern __shared__ unsigned long long s_big[];        // SORT_BIG_CAP keys
    __shared__ uint32_t s_q[4];
    if ((int64_t)state[0] > capacity) return;
    const int n_big = (int)state[6];
    for (int b = blockIdx.x; b < n_big; b += gridDim.x) {
        const int tile = (int)big_list[b];
        const int n = (int)tile_cnt[tile];
        const uint32_t beg = chunk_base[tile / BIN_CHUNK] + tile_loc[tile];
        unsigned long long* g = pairs + beg;
        if (n <= SORT_BIG_CAP) {
            for (int e = threadIdx.x; e < n; e += SORT_BIG_THREADS) s_big[e] = g[e];
            __syncthreads();
            lds_levels<SORT_BIG_THREADS>(s_big, n, 2, next_pow2(n));
            write_tile<SORT_BIG_THREADS>(s_big, n, beg, tile, plist, qmask, item_est, s_q);
            __syncthreads();
            continue;
        }
        // more keys than LDS holds: levels up to SORT_BIG_CAP chunk by chunk in LDS, then per level the wide steps on global
        // memory (one workgroup: __syncthreads orders them) and the narrow steps in LDS again
        for (int c0 = 0; c0 < n; c0 += SORT_BIG_CAP) {
            const int m = min(SORT_BIG_CAP, n - c0);
            for (int e = threadIdx.x; e < m; e += SORT_BIG_THREADS) s_big[e] = g[c0 + e];
            __syncthreads();
            lds_levels<SORT_BIG_THREADS>(s_big, m, 2, SORT_BIG_CAP);
            for (int e = threadIdx.x; e < m; e += SORT_BIG_THREADS) g[c0 + e] = s_big[e];
            __syncthreads();
        }
        const int npad = next_pow2(n);
        for (int k = 2 * SORT_BIG_CAP; k <= npad; k <<= 1) {
            for (int j = k >> 1; j >= SORT_BIG_CAP; j >>= 1) {
                const bool flip = j == (k >> 1);
                __threadfence_block();
                for (int i = threadIdx.x; 2 * i < n + j; i += SORT_BIG_THREADS) {
                    int lo, hi;
                    if (flip) { const int blk = i / j, o = i - blk * j; lo = blk * k + o; hi = blk * k + k - 1 - o; }
                    else { lo = ((i & ~(j - 1)) << 1) | (i & (j - 1)); hi = lo | j; }
                    if (hi < n) {
                        const unsigned long long a = g[lo], bb = g[hi];
                        if (a > bb) { g[lo] = bb; g[hi] = a; }
                    }
                }
                __threadfence_block();
                __syncthreads();
            }
            for (int c0 = 0; c0 < n; c0 += SORT_BIG_CAP) {
                const int m = min(SORT_BIG_CAP, n - c0);
                for (int e = threadIdx.x; e < m; e += SORT_BIG_THREADS) s_big[e] = g[c0 + e];
                __syncthreads();
                lds_tail<SORT_BIG_THREADS>(s_big, m, SORT_BIG_CAP >> 1);
                for (int e = threadIdx.x; e < m; e += SORT_BIG_THREADS) g[c0 + e] = s_big[e];
                __syncthreads();
            }
        }
        __threadfence_block();
        write_tile<SORT_BIG_THREADS>(g, n, beg, tile, plist, qmask, item_est, s_q);
        __syncthreads();
    }
}

}   // namespace

// ---- host side --------------------------------------------------------------------------------------------------------
int mrgs_bin_groups(int P) { const int g = (P + BIN_THREADS - 1) / BIN_THREADS; return g < 1 ? 1 : g > 256 ? 256 : g; }
int mrgs_bin_tpad(int T) { return (T + 255) & ~255; }
// the LDS histogram / cursor array of a slice workgroup holds one word per tile (160 KiB per workgroup on gfx950, shared with the
// emit kernel's per-wave staging)
bool mrgs_bin_supported(int T) { return T <= 16384; }   // 64 KB of cursors + 88 KB of per-wave staging in tile_emit_kernel (up to 2 048 x 2 048 px)

static int per_group(int P) { const int G = mrgs_bin_groups(P); return ((P + G - 1) / G + 63) & ~63; }

void mrgs_launch_tile_count_scan(const MrgsRasterConfig& cfg, const MrgsGeomWs& g, uint32_t* host_slot, hipStream_t stream)
{
    const int tiles_x = (cfg.W + MRGS_BLOCK_X - 1) / MRGS_BLOCK_X, tiles_y = (cfg.H + MRGS_BLOCK_Y - 1) / MRGS_BLOCK_Y;
    const int T = tiles_x * tiles_y, Tpad = mrgs_bin_tpad(T), G = mrgs_bin_groups(cfg.P);
    static bool attr_done[64] = {false};
    int dev = 0;
    (void)hipGetDevice(&dev);
    bool& attr_set = attr_done[dev & 63];
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)tile_count_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
        (void)hipFuncSetAttribute((const void*)tile_emit_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
        (void)hipFuncSetAttribute((const void*)tile_sort_big_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SORT_BIG_CAP * 8);
        attr_set = true;
    }
    hipLaunchKernelGGL(tile_count_kernel, dim3(G), dim3(BIN_THREADS), (size_t)T * sizeof(uint32_t), stream, cfg.P, per_group(cfg.P), g.tiles_touched,
                       g.rect, tiles_x, T, Tpad, g.tile_mat);
    hipLaunchKernelGGL(tile_scan_kernel, dim3(Tpad / BIN_CHUNK), dim3(1024), 0, stream, G, T, Tpad, g.tile_mat, g.tile_cnt, g.tile_loc, g.chunk_tot,
                       g.chunk_base, g.counters, host_slot);
}

void mrgs_launch_tile_emit_sort(const MrgsRasterConfig& cfg, const MrgsGeomWs& g, const MrgsBinWs& b, const MrgsImgWs& img, int64_t capacity,
                                bool reuse_order, void* bulk_zero, size_t bulk_zero_bytes, hipStream_t stream)
{
    const int tiles_x = (cfg.W + MRGS_BLOCK_X - 1) / MRGS_BLOCK_X, tiles_y = (cfg.H + MRGS_BLOCK_Y - 1) / MRGS_BLOCK_Y;
    const int T = tiles_x * tiles_y, Tpad = mrgs_bin_tpad(T), G = mrgs_bin_groups(cfg.P);
    unsigned long long* pairs = (unsigned long long*)b.tile_key[0];
    hipLaunchKernelGGL(tile_emit_kernel, dim3(G), dim3(BIN_THREADS), (size_t)((T + 3) & ~3) * sizeof(uint32_t) + (BIN_THREADS / 64) * EMIT_WAVE_BYTES, stream, cfg.P, per_group(cfg.P), g.tiles_touched,
                       g.rect, g.depth_key[0], g.cull, tiles_x, T, Tpad, g.tile_mat, g.tile_loc, g.chunk_base, g.counters, capacity, pairs,
                       g.counters + 16, g.counters + 6);
    hipLaunchKernelGGL(tile_sort_kernel, dim3(T), dim3(BS_THREADS), 0, stream, T, g.tile_cnt, g.tile_loc, g.chunk_base, g.counters, capacity,
                       pairs, b.plist[0], b.qmask, img.ranges, img.item_est, g.big_list, g.counters + 16,
                       reuse_order ? img.blend_state : (uint32_t*)nullptr, img.q_bwd, img.redo_list, img.item_work, (float4*)(reuse_order ? bulk_zero : nullptr),
                       reuse_order && bulk_zero ? bulk_zero_bytes / sizeof(float4) : (size_t)0);
    hipLaunchKernelGGL(tile_sort_big_kernel, dim3(32), dim3(SORT_BIG_THREADS), (size_t)SORT_BIG_CAP * 8, stream, g.tile_cnt, g.tile_loc, g.chunk_base,
                       g.counters, capacity, pairs, b.plist[0], b.qmask, img.item_est, g.big_list);
}
