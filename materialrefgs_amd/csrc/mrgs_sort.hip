// mrgs_sort.hip -- binning for the surfel rasterizer on gfx950: wave64 ballot-ranked LSD radix sort,
// tiles_touched scan, (tile, gaussian) pair emission and tile ranges.
//
// Reference behaviour being reproduced (rasterizer_impl.cu:283-324): point_list = gaussian ids ordered by
// (tile id, raw depth bits), ties in emission order (gaussian index).  The reference gets it with ONE
// radix sort of R 64-bit keys over 32+log2(tiles) bits (CUB, ~6 passes x 24 B/pair).  Here the same order is
// produced MI355X-first by
//   (1) sorting the P gaussians once by depth bits (4 passes over P 8-byte pairs),
//   (2) emitting the (tile, gaussian) pairs in that order, and
//   (3) a stable sort of the R pairs on the tile id only (2 passes for <= 65536 tiles).
// LSD radix passes are stable, so within a tile the pairs stay in (depth bits, gaussian index) order --
// exactly the reference's order -- while moving ~3.4x fewer bytes than the 64-bit-key sort.
#include "mrgs_internal.h"

#define SORT_THREADS 256
#define SORT_WAVES (SORT_THREADS / 64)

// Each radix pass is ONE kernel ("onesweep" structure): a workgroup ranks its tile of keys locally, publishes its per-digit
// counts, obtains the counts of all preceding tiles by a decoupled look-back over the published status words, and
// scatters.  The only other launch is one kernel that histograms every digit position of all keys up front (digit totals
// do not depend on the order of the keys).  Against a histogram + scatter pair per pass this halves the launches and
// reads the keys once per pass; with 1024-key tiles a 300k-key pass fills the chip (293 workgroups) instead of 74.
//
// Look-back protocol (placement-independent, as the MI355X guide requires): a workgroup takes its tile index from an
// atomic ticket, so "lower tile index" implies "started earlier"; status word = state << 30 | count, written and read with
// agent-scope atomics (the word carries its own payload, no second location has to be ordered); tile 0 publishes an
// inclusive prefix straight away, which ends every look-back.  Spins are bounded: on overrun the workgroup raises the
// error flag of the sort workspace and leaves without writing.
#define OS_STATE_AGG 1u
#define OS_STATE_INC 2u
#define OS_VALUE_MASK 0x3FFFFFFFu
#define OS_SPIN_LIMIT (1 << 24)

// Element count of a binning kernel: the host value, or -- when the launch was sized for a capacity before the count was
// known on the host (mrgs_rasterize_forward) -- the device-resident count clamped to that capacity.
__device__ __forceinline__ int64_t mrgs_count(int64_t n_host, const uint32_t* __restrict__ n_dev)
{
    if (n_dev == nullptr) return n_host;
    const int64_t v = (int64_t)*n_dev;
    return v < n_host ? v : n_host;
}

__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << (threadIdx.x & 63)) - 1ull; }

__device__ __forceinline__ uint32_t os_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void os_store(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// sum of the published counts of tiles [0, blk) for the caller's column; false on spin overrun.  The status words of
// OS_WINDOW predecessors are fetched together (independent loads in flight) and consumed nearest-first: a walk that had
// to take one dependent memory round trip per predecessor made the passes latency-bound.
#define OS_WINDOW 8
__device__ __forceinline__ bool os_look_back(const uint32_t* __restrict__ status, int stride, int column, int blk, uint32_t& excl)
{
    excl = 0;
    int spins = 0;
    int p = blk - 1;
    while (p >= 0) {
        uint32_t s[OS_WINDOW];
#pragma unroll
        for (int k = 0; k < OS_WINDOW; k++) s[k] = os_load(status + (size_t)max(p - k, 0) * stride + column);
        bool stalled = false;
#pragma unroll
        for (int k = 0; k < OS_WINDOW; k++) {
            if (!stalled && p >= 0) {
                const uint32_t st = s[k] >> 30;
                if (st == 0u) {
                    stalled = true;             // not published yet: poll again from here
                } else {
                    excl += s[k] & OS_VALUE_MASK;
                    p = (st == OS_STATE_INC) ? -1 : p - 1;
                }
            }
        }
        if (stalled) {
            if (++spins > OS_SPIN_LIMIT) return false;
            __builtin_amdgcn_s_sleep(1);
        }
    }
    return true;
}

// ---- block-wide exclusive scan helper (wave64 shuffles + one LDS hop) ----------------------------------
template <int THREADS>
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* lds_wave_sums /*[THREADS/64]*/, uint32_t& total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
    }
    if (lane == 63) lds_wave_sums[wave] = inc;
    __syncthreads();
    uint32_t wave_off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < THREADS / 64; w++) {
        uint32_t s = lds_wave_sums[w];
        if (w < wave) wave_off += s;
        tot += s;
    }
    __syncthreads();
    total = tot;
    return wave_off + inc - v;
}

// ---- digit totals of every pass in one sweep over the keys ----------------------------------------------
template <int NPASS>
__global__ void __launch_bounds__(SORT_THREADS) radix_totals_kernel(const uint32_t* __restrict__ keys, int64_t n_host,
                                                                    const uint32_t* __restrict__ n_dev, int bit_lo,
                                                                    uint32_t* __restrict__ totals /*[NPASS][256]*/)
{
    __shared__ uint32_t h[NPASS][256];
    const int tid = threadIdx.x;
    const int64_t n = mrgs_count(n_host, n_dev);
#pragma unroll
    for (int p = 0; p < NPASS; p++) h[p][tid] = 0;
    __syncthreads();
    for (int64_t idx = (int64_t)blockIdx.x * SORT_THREADS + tid; idx < n; idx += (int64_t)gridDim.x * SORT_THREADS) {
        const uint32_t k = keys[idx];
#pragma unroll
        for (int p = 0; p < NPASS; p++) atomicAdd(&h[p][(k >> (bit_lo + 8 * p)) & 255u], 1u);
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < NPASS; p++)
        if (h[p][tid] != 0u) atomicAdd(&totals[p * 256 + tid], h[p][tid]);
}

// ---- one stable radix pass; in-wave ranks from 8 ballots (wave64 multi-split) -----------------------------
// Wave w of the workgroup owns ITEMS consecutive runs of 64 keys; all counters it touches while ranking are its own
// (whist[w]), so the ranking loop has no barrier: lanes of one wave execute LDS instructions in program order.
template <int ITEMS>
__global__ void __launch_bounds__(SORT_THREADS) radix_onesweep_kernel(const uint32_t* __restrict__ kin, const uint32_t* __restrict__ vin,
                                                                      uint32_t* __restrict__ kout, uint32_t* __restrict__ vout,
                                                                      const uint32_t* __restrict__ totals, uint32_t* __restrict__ status,
                                                                      uint32_t* __restrict__ ticket, uint32_t* __restrict__ error_flag,
                                                                      int64_t n_host, const uint32_t* __restrict__ n_dev, int shift)
{
    __shared__ uint32_t whist[SORT_WAVES][256];
    __shared__ uint32_t start[256];
    __shared__ uint32_t scan_tmp[SORT_WAVES];
    __shared__ uint32_t s_blk;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_blk = atomicAdd(ticket, 1u);
#pragma unroll
    for (int w = 0; w < SORT_WAVES; w++) whist[w][tid] = 0;
    __syncthreads();
    const int blk = (int)s_blk;
    const int64_t n = mrgs_count(n_host, n_dev);
    if ((int64_t)blk * (SORT_THREADS * ITEMS) >= n) return;   // launched for the capacity, not needed for the actual count
    const int64_t wbase = (int64_t)blk * (SORT_THREADS * ITEMS) + (int64_t)wave * (64 * ITEMS) + lane;
    const uint64_t lt = lanemask_lt();

    uint32_t key[ITEMS], val[ITEMS], off[ITEMS];
#pragma unroll
    for (int it = 0; it < ITEMS; it++) {
        const int64_t idx = wbase + it * 64;
        const bool valid = idx < n;
        key[it] = valid ? kin[idx] : 0xFFFFFFFFu;
        val[it] = valid ? vin[idx] : 0u;
    }
#pragma unroll
    for (int it = 0; it < ITEMS; it++) {
        const bool valid = wbase + it * 64 < n;
        const uint32_t d = (key[it] >> shift) & 255u;
        uint64_t peers = __builtin_amdgcn_ballot_w64(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const bool bit = (d >> b) & 1u;
            const uint64_t m = __builtin_amdgcn_ballot_w64(bit);
            peers &= bit ? m : ~m;
        }
        const uint32_t rank = __popcll(peers & lt);
        const uint32_t prev = whist[wave][d];              // keys of this digit in the wave's earlier runs
        __builtin_amdgcn_wave_barrier();
        if (valid && rank == 0) whist[wave][d] = prev + (uint32_t)__popcll(peers);
        __builtin_amdgcn_wave_barrier();
        off[it] = prev + rank;
    }
    __syncthreads();

    // thread = digit: tile count, exclusive prefix over the waves (kept in whist), look-back over the preceding tiles
    uint32_t cnt = 0;
#pragma unroll
    for (int w = 0; w < SORT_WAVES; w++) {
        const uint32_t c = whist[w][tid];
        whist[w][tid] = cnt;
        cnt += c;
    }
    uint32_t* mine = status + (size_t)blk * 256 + tid;
    uint32_t excl = 0;
    bool ok = true;
    if (blk == 0) {
        os_store(mine, (OS_STATE_INC << 30) | cnt);
    } else {
        os_store(mine, (OS_STATE_AGG << 30) | cnt);
        ok = os_look_back(status, 256, tid, blk, excl);
        if (ok) os_store(mine, (OS_STATE_INC << 30) | (excl + cnt));
    }
    uint32_t all;
    const uint32_t digit_base = block_exclusive_scan<SORT_THREADS>(totals[tid], scan_tmp, all);
    start[tid] = digit_base + excl;
    if (__syncthreads_or(!ok)) {
        if (tid == 0) atomicExch(error_flag, 1u);
        return;
    }
#pragma unroll
    for (int it = 0; it < ITEMS; it++) {
        if (wbase + it * 64 < n) {
            const uint32_t d = (key[it] >> shift) & 255u;
            const uint32_t pos = start[d] + whist[wave][d] + off[it];
            kout[pos] = key[it];
            vout[pos] = val[it];
        }
    }
}

// tile size of the passes for n keys: enough workgroups to fill 256 CUs for small inputs, longer tiles (shorter look-back
// chains, fewer status words) for large ones
#ifdef MRGS_SORT_ITEMS_FORCE
static int sort_items(int64_t) { return MRGS_SORT_ITEMS_FORCE; }
#else
static int sort_items(int64_t n) { return n <= (128 << 10) ? 4 : n <= (768 << 10) ? 8 : 16; }   // look-back chains of <= ~300 tiles
#endif

size_t mrgs_sort_ws_words(int64_t n)
{
    const int64_t nblk = (n + 1023) / 1024 + 1;
    return (size_t)(MRGS_SORT_WS_HEADER + 4 * 256 + 4 * 256 * nblk);
}

// ws must be zero on entry (the caller clears it together with its other per-call state); layout: [0..3] tickets of the
// passes, then digit totals [4][256], then status words [passes][nblk][256]
int mrgs_radix_sort_pairs(uint32_t* key[2], uint32_t* val[2], uint32_t* ws, uint32_t* error_flag, int64_t n, const uint32_t* n_dev,
                          int bit_lo, int bit_hi, hipStream_t stream)
{
    int cur = 0;
    if (n <= 0 || bit_hi <= bit_lo) return cur;
    const int npass = (bit_hi - bit_lo + 7) / 8;
    const int items = sort_items(n);
    const int tile = SORT_THREADS * items;
    const int nblk = (int)((n + tile - 1) / tile);
    uint32_t* tickets = ws;
    uint32_t* totals = ws + MRGS_SORT_WS_HEADER;
    uint32_t* status = totals + 4 * 256;
    const int tblk = (int)((n + 4095) / 4096 < 1024 ? (n + 4095) / 4096 : 1024);
    switch (npass) {
    case 1: hipLaunchKernelGGL(radix_totals_kernel<1>, dim3(tblk), dim3(SORT_THREADS), 0, stream, key[0], n, n_dev, bit_lo, totals); break;
    case 2: hipLaunchKernelGGL(radix_totals_kernel<2>, dim3(tblk), dim3(SORT_THREADS), 0, stream, key[0], n, n_dev, bit_lo, totals); break;
    case 3: hipLaunchKernelGGL(radix_totals_kernel<3>, dim3(tblk), dim3(SORT_THREADS), 0, stream, key[0], n, n_dev, bit_lo, totals); break;
    default: hipLaunchKernelGGL(radix_totals_kernel<4>, dim3(tblk), dim3(SORT_THREADS), 0, stream, key[0], n, n_dev, bit_lo, totals); break;
    }
    for (int p = 0; p < npass; p++) {
        const int shift = bit_lo + 8 * p;
        uint32_t* st = status + (size_t)p * nblk * 256;
#define OS_LAUNCH(IT)                                                                                                        \
    hipLaunchKernelGGL(radix_onesweep_kernel<IT>, dim3(nblk), dim3(SORT_THREADS), 0, stream, key[cur], val[cur], key[cur ^ 1], \
                       val[cur ^ 1], totals + p * 256, st, tickets + p, error_flag, n, n_dev, shift)
        if (items == 4) OS_LAUNCH(4);
        else if (items == 8) OS_LAUNCH(8);
        else OS_LAUNCH(16);
#undef OS_LAUNCH
        cur ^= 1;
    }
    return cur;
}

// ---- exclusive scan of tiles_touched in depth-sorted order (rasterizer_impl.cu:283, InclusiveSum) -----
// One kernel, same look-back protocol with one status word per workgroup (64-bit: the total may need 31 bits).
#define SCAN_THREADS 256
#define SCAN_PER_THREAD 8
#define SCAN_TILE (SCAN_THREADS * SCAN_PER_THREAD)

__global__ void __launch_bounds__(SCAN_THREADS) scan_tiles_kernel(const uint32_t* __restrict__ tiles_touched,
                                                                  const uint32_t* __restrict__ order,
                                                                  uint32_t* __restrict__ offsets, unsigned long long* __restrict__ status,
                                                                  uint32_t* __restrict__ ticket, uint32_t* __restrict__ total_out,
                                                                  uint32_t* __restrict__ error_flag, int n, int nblk)
{
    __shared__ uint32_t wave_sums[SCAN_THREADS / 64];
    __shared__ uint32_t s_blk;
    __shared__ unsigned long long s_excl;
    __shared__ int s_ok;
    if (threadIdx.x == 0) s_blk = atomicAdd(ticket, 1u);
    __syncthreads();
    const int blk = (int)s_blk;
    const int base = blk * SCAN_TILE + threadIdx.x * SCAN_PER_THREAD;
    uint32_t v[SCAN_PER_THREAD], s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; k++) {
        v[k] = (base + k < n) ? tiles_touched[order[base + k]] : 0u;
        s += v[k];
    }
    uint32_t tot;
    uint32_t ex = block_exclusive_scan<SCAN_THREADS>(s, wave_sums, tot);
    if (threadIdx.x < 64) {
        // wave 0 looks back over 64 predecessors per step: lane l reads the status of tile p - l
        const unsigned long long AGG = 1ull << 62, INC = 2ull << 62, MASK = (1ull << 62) - 1;
        const int lane = threadIdx.x;
        unsigned long long excl = 0;
        bool ok = true;
        if (blk == 0) {
            if (lane == 0) __hip_atomic_store(status, INC | tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (lane == 0) __hip_atomic_store(status + blk, AGG | tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            int p = blk - 1;
            while (p >= 0) {
                const int q = p - lane;
                const unsigned long long w = q >= 0 ? __hip_atomic_load(status + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : INC;
                const unsigned long long st = w >> 62;
                const uint64_t m_inc = __builtin_amdgcn_ballot_w64(st == 2ull);
                const uint64_t m_zero = __builtin_amdgcn_ballot_w64(st == 0ull);
                // usable lanes: everything nearer than the first unpublished one, up to and including the first inclusive one
                const int first_zero = m_zero ? __builtin_ctzll(m_zero) : 64;
                const int first_inc = m_inc ? __builtin_ctzll(m_inc) : 64;
                const int take = min(first_zero, first_inc + 1);       // lanes [0, take)
                unsigned long long part = (lane < take && q >= 0) ? (w & MASK) : 0ull;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
                excl += part;
                if (first_inc < first_zero) break;                      // reached an inclusive prefix
                p -= take;
                if (take == 0) {
                    if (++spins > OS_SPIN_LIMIT) { ok = false; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            if (ok && lane == 0) __hip_atomic_store(status + blk, INC | (excl + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) {
            s_excl = excl;
            s_ok = ok ? 1 : 0;
            if (!ok) atomicExch(error_flag, 1u);
            if (ok && blk == nblk - 1) {
                const unsigned long long total = excl + tot;
                *total_out = total > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)total;   // saturates; the host rejects >= 2^30
            }
        }
    }
    __syncthreads();
    if (!s_ok) return;
    ex += (uint32_t)s_excl;
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; k++) {
        if (base + k < n) offsets[base + k] = ex;
        ex += v[k];
    }
}

size_t mrgs_scan_ws_words(int n) { return 2 * (size_t)((n + SCAN_TILE - 1) / SCAN_TILE + 1) + 2; }

// ws (zero on entry): [0] ticket, [1] pad, then 64-bit status words
void mrgs_scan_tiles(const uint32_t* tiles_touched, const uint32_t* order, uint32_t* offsets, uint32_t* ws, uint32_t* total_out,
                     uint32_t* error_flag, int n, hipStream_t stream)
{
    const int nblk = (n + SCAN_TILE - 1) / SCAN_TILE;
    hipLaunchKernelGGL(scan_tiles_kernel, dim3(nblk), dim3(SCAN_THREADS), 0, stream, tiles_touched, order, offsets,
                       (unsigned long long*)(ws + 2), ws, total_out, error_flag, n, nblk);
}

// ---- pair emission (duplicateWithKeys, rasterizer_impl.cu:72-113) in depth-sorted gaussian order -------
__global__ void __launch_bounds__(256) duplicate_kernel(int P, const uint32_t* __restrict__ order, const uint32_t* __restrict__ tiles_touched,
                                                        const uint32_t* __restrict__ offsets, const uint2* __restrict__ rect,
                                                        int tiles_x, uint32_t* __restrict__ tile_key, uint32_t* __restrict__ plist,
                                                        uint32_t capacity)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const uint32_t g = order[i];
    if (tiles_touched[g] == 0) return;
    uint32_t off = offsets[i];
    if (off + tiles_touched[g] > capacity) return;   // only when the buffers were sized for a guess that proved too small
    const uint2 r = rect[g];
    const int x0 = r.x & 0xFFFF, y0 = r.x >> 16, x1 = r.y & 0xFFFF, y1 = r.y >> 16;
    for (int y = y0; y < y1; y++)
        for (int x = x0; x < x1; x++) {
            tile_key[off] = (uint32_t)(y * tiles_x + x);
            plist[off] = g;
            off++;
        }
}

void mrgs_launch_duplicate(const MrgsRasterConfig& cfg, const MrgsGeomWs& g, const uint32_t* order, uint32_t* tile_key,
                           uint32_t* plist, int64_t capacity, hipStream_t stream)
{
    const int tiles_x = (cfg.W + MRGS_BLOCK_X - 1) / MRGS_BLOCK_X;
    hipLaunchKernelGGL(duplicate_kernel, dim3((cfg.P + 255) / 256), dim3(256), 0, stream, cfg.P, order, g.tiles_touched, g.offsets,
                       g.rect, tiles_x, tile_key, plist, (uint32_t)capacity);
}

// ---- identifyTileRanges (rasterizer_impl.cu:118-140) on the sorted tile ids ---------------------------
__global__ void __launch_bounds__(256) tile_ranges_kernel(const uint32_t* __restrict__ tile_key, int64_t R_host,
                                                          const uint32_t* __restrict__ R_dev, uint2* __restrict__ ranges)
{
    const int64_t R = mrgs_count(R_host, R_dev);
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= R) return;
    const uint32_t cur = tile_key[idx];
    if (idx == 0)
        ranges[cur].x = 0;
    else {
        const uint32_t prev = tile_key[idx - 1];
        if (cur != prev) {
            ranges[prev].y = (uint32_t)idx;
            ranges[cur].x = (uint32_t)idx;
        }
    }
    if (idx == R - 1) ranges[cur].y = (uint32_t)R;
}

// Blend-kernel dispatch order: tiles sorted by decreasing list length (counting sort on length / 16, one workgroup).  The
// blend launches last as long as their longest wave, and not every wave is resident from the start: longest-first keeps the
// heavy tiles off the tail of the launch.  Slots beyond ntiles (grid padding) get the id ntiles (= no tile).
__global__ void __launch_bounds__(1024) tile_order_kernel(const uint2* __restrict__ ranges, int ntiles, int nslots, uint32_t* __restrict__ order,
                                                          uint32_t* __restrict__ item_work, uint32_t* __restrict__ bwd_state)
{
    for (int t = threadIdx.x; t < 8 * nslots; t += 1024) item_work[t] = 0u;   // filled by the forward blend waves that do work
    if (threadIdx.x < 64) bwd_state[MRGS_BS_BITMAP + threadIdx.x] = 0u;       // CU census, filled by the same waves
    __shared__ uint32_t hist[256];
    __shared__ uint32_t wave_sums[16];
    const int tid = threadIdx.x;
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    for (int t = tid; t < ntiles; t += 1024) {
        const uint2 r = ranges[t];
        atomicAdd(&hist[255u - min((r.y - r.x) >> 4, 255u)], 1u);   // bucket 0 = longest lists
    }
    __syncthreads();
    uint32_t tot;
    const uint32_t ex = block_exclusive_scan<1024>(tid < 256 ? hist[tid] : 0u, wave_sums, tot);
    if (tid < 256) hist[tid] = ex;
    // number of leading tiles that the blend kernels split into half-quadrants (mrgs_decode_item)
    if (tid == 256 - MRGS_SPLIT_THRESHOLD / 16) order[nslots] = ex;
    __syncthreads();
    for (int t = tid; t < ntiles; t += 1024) {
        const uint2 r = ranges[t];
        const uint32_t pos = atomicAdd(&hist[255u - min((r.y - r.x) >> 4, 255u)], 1u);
        order[pos] = (uint32_t)t;
    }
    for (int t = ntiles + tid; t < nslots; t += 1024) order[t] = (uint32_t)ntiles;
}

// Dispatch order of the blend backward.  The forward waves leave the number of list entries they walked in item_work; the
// backward walks (almost) the same entries, so this is its work per item.  Kernel time is set by the most loaded SIMD, and
// a random mix of four or five waves per SIMD put 1.44x the mean load on the worst one.  Here, per XCD list (tile p of
// tile_order belongs to list p % 8, as in the forward), the items with work are counting-sorted by decreasing work and
// dealt to one queue per SIMD of the XCD in passes of NQ items: in every pass the heaviest item goes to the queue with the
// smallest load so far (LPT per pass).  A backward wave pulls from the queue of the SIMD it finds itself on and steals
// from the other queues when its own is empty (render_bwd_kernel); nothing depends on how the hardware places waves.
// This kernel also turns the CU census of the forward launch into a dense CU numbering per XCC.
#define BWD_ADAPTIVE_PASSES 64
__global__ void __launch_bounds__(1024) bwd_order_kernel(const uint32_t* __restrict__ tile_order, const uint32_t* __restrict__ item_work,
                                                         int ntiles, int nslots, uint32_t* __restrict__ bwd_items, uint32_t* __restrict__ bwd_work,
                                                         uint32_t* __restrict__ bwd_assign, uint32_t* __restrict__ bwd_state)
{
    __shared__ uint32_t hist[1024];
    __shared__ uint32_t wave_sums[16];
    __shared__ uint32_t load[MRGS_MAX_SIMD_QUEUES];
    __shared__ unsigned long long s_total;
    const int tid = threadIdx.x, x = blockIdx.x;
    hist[tid] = 0;
    if (tid == 0) s_total = 0ull;
    if (tid < MRGS_MAX_SIMD_QUEUES) { bwd_state[MRGS_BS_TICKET + x * MRGS_MAX_SIMD_QUEUES + tid] = 0u; load[tid] = 0u; }
    if (tid < 256) {   // block x numbers the CUs of XCC x
        const uint32_t* bm = bwd_state + MRGS_BS_BITMAP + x * 8;
        uint32_t below = 0;
        for (int w = 0; w < (tid >> 5); w++) below += __popc(bm[w]);
        below += __popc(bm[tid >> 5] & ((1u << (tid & 31)) - 1u));
        bwd_state[MRGS_BS_DENSE + x * 256 + tid] = below;
        if (tid == 255) bwd_state[MRGS_BS_NCU + x] = below + ((bm[7] >> 31) & 1u);
    }
    __syncthreads();
    const int n = nslots;   // (nslots / 8 tiles) x 8 items
    for (int idx = tid; idx < n; idx += 1024) {
        const uint32_t tile = tile_order[(idx >> 3) * 8 + x];
        const uint32_t w = tile < (uint32_t)ntiles ? item_work[tile * 8 + (idx & 7)] : 0u;
        if (w > 0u) atomicAdd(&hist[1023u - min(w >> 2, 1023u)], 1u);   // bucket 0 = most work
    }
    __syncthreads();
    uint32_t tot;
    const uint32_t ex = block_exclusive_scan<1024>(hist[tid], wave_sums, tot);
    hist[tid] = ex;
    __syncthreads();
    uint32_t* items = bwd_items + (size_t)x * nslots;
    uint32_t* work = bwd_work + (size_t)x * nslots;
    for (int idx = tid; idx < n; idx += 1024) {
        const uint32_t tile = tile_order[(idx >> 3) * 8 + x];
        const uint32_t w = tile < (uint32_t)ntiles ? item_work[tile * 8 + (idx & 7)] : 0u;
        if (w > 0u) {
            const uint32_t pos = atomicAdd(&hist[1023u - min(w >> 2, 1023u)], 1u);
            items[pos] = (tile << 3) | (uint32_t)(idx & 7);
            work[pos] = w;
            atomicAdd(&s_total, (unsigned long long)w);
        }
    }
    __threadfence_block();
    __syncthreads();

    // dealing: NQ queues (the SIMDs of the XCC this list runs on; all XCCs of a device have the same CU count)
    int Q = 0;
    for (int k = 0; k < 8; k++) {
        const uint32_t* bm = bwd_state + MRGS_BS_BITMAP + k * 8;
        int c = 0;
        for (int w = 0; w < 8; w++) c += __popc(bm[w]);
        Q = max(Q, c);
    }
    const int NQ = 4 * min(max(Q, 1), MRGS_MAX_SIMD_QUEUES / 4);
    const int n_items = (int)tot;
    const int passes = (n_items + NQ - 1) / NQ;
    const unsigned long long mean5 = n_items > 0 ? 5ull * s_total / (unsigned long long)n_items : 0ull;   // 5 x mean work
    uint32_t* assign = bwd_assign + (size_t)x * (nslots + MRGS_MAX_SIMD_QUEUES);
    for (int p = 0; p < passes; p++) {
        int slot = tid;                                    // position in the pass (0 = heaviest item) this queue receives
        if (tid < NQ) {
            if (p > 0 && p < BWD_ADAPTIVE_PASSES) {
                const uint32_t mine = load[tid];
                int r = 0;
                for (int q = 0; q < NQ; q++) {
                    const uint32_t o = load[q];
                    r += (o < mine || (o == mine && q < tid)) ? 1 : 0;
                }
                slot = r;                                  // lightest queue <- heaviest item
            } else if (p & 1) {
                slot = NQ - 1 - tid;                       // plain snake beyond the adaptive passes
            }
        }
        __syncthreads();
        if (tid < NQ) {
            const int rank = p * NQ + slot;
            const bool valid = rank < n_items;
            uint32_t entry = 0xFFFFFFFFu;
            if (valid) {
                // issue priority of the wave (bits 29-30): the heavy items of a SIMD run ahead of its light ones, so that the
                // SIMD keeps several waves in flight until its work runs out instead of finishing with one long straggler
                const unsigned long long w25 = 25ull * work[rank];
                const uint32_t prio = w25 > 8ull * mean5 ? 3u : w25 > 6ull * mean5 ? 2u : w25 > 4ull * mean5 ? 1u : 0u;
                entry = items[rank] | (prio << 29);
                load[tid] += work[rank];
            }
            assign[p * NQ + tid] = entry;
        }
        __syncthreads();
    }
    if (tid == 0) {
        bwd_state[MRGS_BS_COUNT + x] = tot;
        bwd_state[MRGS_BS_PASSES + x] = (uint32_t)passes | ((uint32_t)NQ << 16);
    }
}

void mrgs_launch_bwd_order(const MrgsImgWs& img, int ntiles, hipStream_t stream)
{
    hipLaunchKernelGGL(bwd_order_kernel, dim3(8), dim3(1024), 0, stream, img.tile_order, img.item_work, ntiles, ((ntiles + 7) / 8) * 8,
                       img.bwd_items, img.bwd_work, img.bwd_assign, img.bwd_state);
}

void mrgs_launch_tile_ranges(const uint32_t* tile_key, int64_t R, const uint32_t* R_dev, const MrgsImgWs& img, int ntiles, hipStream_t stream)
{
    uint2* ranges = img.ranges;
    uint32_t* tile_order = img.tile_order;
    (void)hipMemsetAsync(ranges, 0, sizeof(uint2) * (size_t)ntiles, stream);   // rasterizer_impl.cu:316
    if (R > 0)
        hipLaunchKernelGGL(tile_ranges_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, stream, tile_key, R, R_dev, ranges);
    hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), 0, stream, ranges, ntiles, ((ntiles + 7) / 8) * 8, tile_order, img.item_work, img.bwd_state);
}
