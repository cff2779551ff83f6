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
#include "mrgs_blend_math.h"

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
// known on the host (mrgs_rasterize_forward) -- the device-resident count.  If that count exceeds the capacity the whole
// second phase degenerates to "nothing to do" (0): the pair buffers would hold holes with arbitrary tile / gaussian ids, and
// the host redoes the phase on an exactly sized workspace anyway.
__device__ __forceinline__ int64_t mrgs_count(int64_t n_host, const uint32_t* __restrict__ n_dev)
{
    if (n_dev == nullptr) return n_host;
    const int64_t v = (int64_t)*n_dev;
    return v <= n_host ? v : 0;
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
    // eight keys in flight per thread: a rolled load -> LDS-atomic loop waits for every load's round trip in turn
    const int64_t stride = (int64_t)gridDim.x * SORT_THREADS;
    for (int64_t base = (int64_t)blockIdx.x * SORT_THREADS + tid; base < n; base += 8 * stride) {
        uint32_t k[8];
#pragma unroll
        for (int j = 0; j < 8; j++) k[j] = base + j * stride < n ? keys[base + j * stride] : 0u;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (base + j * stride < n) {
#pragma unroll
                for (int p = 0; p < NPASS; p++) atomicAdd(&h[p][(k[j] >> (bit_lo + 8 * p)) & 255u], 1u);
            }
        }
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
    // All keys share this digit (the sign / high exponent byte of the depths of an ordinary scene): a stable pass is the identity,
    // every workgroup sees it in the digit totals and just copies its tile -- no ranking, no look-back.
    if (__syncthreads_or(totals[tid] == (uint32_t)n)) {
#pragma unroll
        for (int it = 0; it < ITEMS; it++) {
            const int64_t idx = wbase + it * 64;
            if (idx < n) { kout[idx] = kin[idx]; vout[idx] = vin[idx]; }
        }
        return;
    }

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
#ifdef MRGS_SORT_DIRECT_SCATTER
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
#else
    // The tile is put in digit order in LDS first and leaves in runs: consecutive threads then store to consecutive addresses
    // of a run (a wave's 64 stores touch a handful of cache lines), where storing straight from the ranking registers sends the
    // 64 keys of a wave instruction to up to 64 different runs.
    __shared__ uint32_t s_key[SORT_THREADS * ITEMS], s_val[SORT_THREADS * ITEMS];
    __shared__ uint32_t lstart[256];
    uint32_t tile_n;
    const uint32_t lbase = block_exclusive_scan<SORT_THREADS>(cnt, scan_tmp, tile_n);   // digit runs inside the tile
    lstart[tid] = lbase;
    start[tid] = digit_base + excl - lbase;                                              // global position of local position 0 of the run
    if (__syncthreads_or(!ok)) {
        if (tid == 0) atomicExch(error_flag, 1u);
        return;
    }
#pragma unroll
    for (int it = 0; it < ITEMS; it++) {
        if (wbase + it * 64 < n) {
            const uint32_t d = (key[it] >> shift) & 255u;
            const uint32_t lp = lstart[d] + whist[wave][d] + off[it];
            s_key[lp] = key[it];
            s_val[lp] = val[it];
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < ITEMS; it++) {
        const uint32_t e = (uint32_t)(it * SORT_THREADS + tid);
        if (e < tile_n) {
            const uint32_t k = s_key[e];
            const uint32_t pos = start[(k >> shift) & 255u] + e;
            kout[pos] = k;
            vout[pos] = s_val[e];
        }
    }
#endif
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
                                                        uint32_t capacity, const uint32_t* __restrict__ R_dev, uint32_t* __restrict__ census,
                                                        uint32_t* __restrict__ clear_a, unsigned words_a, uint32_t* __restrict__ clear_b,
                                                        unsigned words_b, uint32_t* __restrict__ host_slot)
{
    // mrgs_rasterize_forward: the pair count and the error flag of the first phase go straight into the caller's pinned host slot
    // (a copy engine transfer in the middle of the stream costs the copy and a ~6 us bubble)
    if (host_slot != nullptr && R_dev != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {
        __hip_atomic_store(host_slot, R_dev[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(host_slot + 1, R_dev[1], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // first kernel of the second phase: clears the look-back state of the tile sort (clear_a) and the tile ranges + cull counts
    // (clear_b; rasterizer_impl.cu:316 clears the ranges) instead of two memset launches, and takes the CU census for the
    // work queues of the blend kernels (one flag per CU some wave of this launch runs on)
    for (unsigned k = blockIdx.x * blockDim.x + threadIdx.x; k < words_a; k += gridDim.x * blockDim.x) clear_a[k] = 0u;
    for (unsigned k = blockIdx.x * blockDim.x + threadIdx.x; k < words_b; k += gridDim.x * blockDim.x) clear_b[k] = 0u;
    if ((threadIdx.x & 63) == 0) mrgs_census_mark(census);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    if (R_dev != nullptr && *R_dev > capacity) return;   // capacity guess too small: see mrgs_count
    const uint32_t g = order[i];
    if (tiles_touched[g] == 0) return;
    uint32_t off = offsets[i];
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
                           uint32_t* plist, int64_t capacity, const uint32_t* R_dev, const MrgsBinWs& b, const MrgsImgWs& img, uint32_t* host_slot,
                           hipStream_t stream)
{
    const int tiles_x = (cfg.W + MRGS_BLOCK_X - 1) / MRGS_BLOCK_X;
    hipLaunchKernelGGL(duplicate_kernel, dim3((cfg.P + 255) / 256), dim3(256), 0, stream, cfg.P, order, g.tiles_touched, g.offsets,
                       g.rect, tiles_x, tile_key, plist, (uint32_t)capacity, R_dev, g.counters + 16, b.sort_ws,
                       (unsigned)(b.sort_ws_bytes / sizeof(uint32_t)), (uint32_t*)img.ranges, (unsigned)(img.ranges_est_bytes / sizeof(uint32_t)), host_slot);
}

// ---- identifyTileRanges (rasterizer_impl.cu:118-140) on the sorted tile ids + quadrant cull ------------------
// One thread per list entry.  Besides the tile ranges it evaluates the block-level cull (mrgs_block_may_touch) of the
// entry's surfel against the four 8x8 quadrants of its tile ONCE -- the blend kernels' four quadrant waves used to fetch
// the 32-byte cull conic and run the test each, in the forward and again in the backward -- and leaves
//   qmask[idx]            bit q set when the surfel can touch quadrant q (the blend waves read 1 byte per entry),
//   item_est[tile*4 + q]  number of such entries = the forward's work estimate for that quadrant wave (blend_order_kernel).
__global__ void __launch_bounds__(256) tile_ranges_kernel(const uint32_t* __restrict__ tile_key, const uint32_t* __restrict__ plist,
                                                          int64_t R_host, const uint32_t* __restrict__ R_dev, const float4* __restrict__ cull,
                                                          int tiles_x, uint2* __restrict__ ranges, uint8_t* __restrict__ qmask,
                                                          uint32_t* __restrict__ item_est)
{
    const int64_t R = mrgs_count(R_host, R_dev);
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = idx < R;
    uint32_t cur = 0xFFFFFFFFu;
    uint32_t m = 0;
    if (valid) {
        cur = tile_key[idx];
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
        const CullConic c = mrgs_cull_load(cull, plist[idx]);
        const int tx = (int)(cur % (uint32_t)tiles_x), ty = (int)(cur / (uint32_t)tiles_x);
        const float x0 = (float)(tx * MRGS_BLOCK_X), y0 = (float)(ty * MRGS_BLOCK_Y);
#ifdef TR_NO_CULL
        m = (c.a.x + x0 + y0 > 1e20f) ? 3u : 15u;
#else
#pragma unroll
        for (int q = 0; q < 4; q++)
            m |= mrgs_block_may_touch(c, x0 + (float)(8 * (q & 1)), y0 + (float)(8 * (q >> 1)), 7.0f, 7.0f) ? (1u << q) : 0u;
#endif
        qmask[idx] = (uint8_t)m;
    }
    // per-quadrant counts: summed per wave with ballots, per workgroup in LDS (the 256 sorted entries of a workgroup span one or
    // two tiles, rarely more than 64), then one global atomic per (workgroup, tile, quadrant)
    __shared__ uint32_t s_cnt[64 * 4];
    __shared__ uint32_t s_first;
    s_cnt[threadIdx.x] = 0u;
    if (threadIdx.x == 0) s_first = cur;          // entry 0 of the workgroup is valid whenever any entry is
    __syncthreads();
    const uint32_t first_tile = s_first;
#ifdef TR_NO_ATOMICS
    uint64_t todo = 0;
#else
    uint64_t todo = __builtin_amdgcn_ballot_w64(valid);
#endif
    while (todo != 0ull) {
        const uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)cur, __builtin_ctzll(todo));
        const bool mine = valid && cur == t;
        todo &= ~__builtin_amdgcn_ballot_w64(mine);
        const uint32_t rel = t - first_tile;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int n = __builtin_popcountll(__builtin_amdgcn_ballot_w64(mine && ((m >> q) & 1u)));
            if ((threadIdx.x & 63) == 0 && n > 0) {
                if (rel < 64u) atomicAdd(&s_cnt[rel * 4 + q], (uint32_t)n);
                else atomicAdd(&item_est[t * 4 + q], (uint32_t)n);
            }
        }
    }
    __syncthreads();
    const uint32_t c = s_cnt[threadIdx.x];
    if (c != 0u) atomicAdd(&item_est[(first_tile + (threadIdx.x >> 2)) * 4 + (threadIdx.x & 3)], c);
}

// ---- dispatch order of the blend kernels ---------------------------------------------------------------------
// Kernel time of a blend launch is set by its most loaded SIMD: a wave lasts as long as its list is long, the lists differ
// by an order of magnitude, and a random mix of four to six waves per SIMD put 1.44x the mean load on the worst SIMD.
// This kernel (one workgroup per XCD list; list x = the tiles t with t % 8 == x, so that the four quadrant waves of a tile
// share one L2) counting-sorts the work items (tile, quadrant) with work > 0 by decreasing work and deals them to one
// queue per SIMD of an XCD in passes of NQ items; in every pass the heaviest item goes to the queue with the smallest load
// so far (LPT per pass).  The blend kernels launch one wave per queue slot (passes * NQ per list); a wave pulls from the queue
// of the SIMD it finds itself on and looks through the other queues when its own is empty (mrgs_pull_item): correctness does
// not depend on how the hardware places waves, the balance of a launch that fits the machine relies on it filling all SIMDs
// evenly (it does: 1.09x the mean load on the most loaded SIMD, measured, against 1.08x dealt).
// Work per item: the forward uses the cull counts of tile_ranges_kernel, the backward what the forward waves actually walked.
// The first call of a forward also turns the CU census (bits set by the preprocess waves) into a dense CU numbering.
// Work of an item.  Forward with a hint buffer (MrgsRasterInputs::work_hint): what the item's wave measured the last time this
// camera was rendered (entries tested + 3 x entries blended, the unit of the backward's queues) -- it knows where rays terminate
// early, which no count taken before the blend does; items the hint has never seen fall back to 3 x the cull count (same unit).
__device__ __forceinline__ uint32_t item_cost(const uint32_t* __restrict__ item_src, const uint32_t* __restrict__ hint, int idx)
{
    const uint32_t e = item_src[idx];
    if (hint == nullptr) return e;
    const uint32_t h = hint[idx];
    return (h != 0u && e != 0u) ? h : 3u * e;
}
#define ORDER_ADAPTIVE_PASSES 16
#define ORDER_LDS_ITEMS 4096
__global__ void __launch_bounds__(1024) blend_order_kernel(const uint32_t* __restrict__ item_src, int ntiles, uint32_t* __restrict__ items_ws,
                                                           uint32_t* __restrict__ work_ws, uint32_t* __restrict__ assign_ws,
                                                           uint32_t* __restrict__ qstate, const uint32_t* __restrict__ census,
                                                           uint32_t* __restrict__ cu_state, int forward, uint32_t* __restrict__ zero_this,
                                                           float4* __restrict__ bulk_zero, size_t bulk_zero_f4, const uint32_t* __restrict__ hint,
                                                           uint32_t* __restrict__ qstate_twin, uint32_t* __restrict__ redo_count)
{
    // workgroups beyond the eight that order the lists only clear a buffer for the kernel that follows (the gradient rows of
    // the blend backward, 24 MB at P = 300k): the ordering occupies 8 CUs for ~10 us, the clear runs beside it on the others
    if (blockIdx.x >= 8) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        for (size_t i = (size_t)(blockIdx.x - 8) * 1024 + threadIdx.x; i < bulk_zero_f4; i += (size_t)(gridDim.x - 8) * 1024) bulk_zero[i] = z;
        return;
    }
    __shared__ uint32_t hist[1024];
    __shared__ uint32_t wave_sums[16];
    __shared__ uint32_t load[MRGS_MAX_SIMD_QUEUES];
    __shared__ unsigned long long s_total;
    __shared__ uint32_t s_busy;
    const int tid = threadIdx.x, x = blockIdx.x;
    const int per_list = ((ntiles + 7) >> 3) * 4;           // items of one XCD list (upper bound)
    hist[tid] = 0;
    if (tid == 0) { s_total = 0ull; s_busy = 0u; }
    if (tid < MRGS_MAX_SIMD_QUEUES) {
        qstate[MRGS_QS_TICKET + x * MRGS_MAX_SIMD_QUEUES + tid] = 0u;
        // the forward may set up the backward's queues as a copy of its own (MrgsRasterInputs::bwd_grad_ws)
        if (qstate_twin != nullptr) qstate_twin[MRGS_QS_TICKET + x * MRGS_MAX_SIMD_QUEUES + tid] = 0u;
        load[tid] = 0u;
    }
    if (zero_this != nullptr)
        for (int i = tid + 1024 * x; i < 4 * ntiles; i += 8 * 1024) zero_this[i] = 0u;
    if (redo_count != nullptr && x == 0 && tid == 0) redo_count[0] = 0u;     // the forward blend's list of marked pixels: empty
    // forward: every (tile, quadrant) is an item (idle ones still write their pixels: key = work + 1); backward: only those
    // with work
    const uint32_t idle = forward ? 1u : 0u;
    if (forward && tid < 256) {   // block x numbers the CUs of XCC x (tid = CU key)
        const bool seen = census[x * 256 + tid] != 0u;
        const uint64_t m = __builtin_amdgcn_ballot_w64(seen);
        if ((tid & 63) == 0) wave_sums[tid >> 6] = (uint32_t)__builtin_popcountll(m);
        __syncthreads();
        uint32_t below = (uint32_t)__builtin_popcountll(m & ((1ull << (tid & 63)) - 1ull));
        for (int w = 0; w < (tid >> 6); w++) below += wave_sums[w];
        cu_state[MRGS_CS_DENSE + x * 256 + tid] = below;
        if (tid == 255) cu_state[MRGS_CS_NCU + x] = below + (seen ? 1u : 0u);
    } else if (forward) {
        __syncthreads();
    }
    __syncthreads();
    // (items without work -- all of them in bucket 1023 of the forward -- are counted per wave: several hundred LDS
    // atomics on one address would dominate this kernel)
    for (int i0 = 0; i0 < per_list; i0 += 1024) {
        const int i = i0 + tid;
        const int tile = (i >> 2) * 8 + x;
        const uint32_t w = (i < per_list && tile < ntiles) ? item_cost(item_src, hint, tile * 4 + (i & 3)) + idle : 0u;
        const bool lightest = w > 0u && (w >> 2) == 0u;
        const uint64_t lm = __builtin_amdgcn_ballot_w64(lightest);
        if (lightest) { if ((tid & 63) == __builtin_ctzll(lm)) atomicAdd(&hist[1023], (uint32_t)__builtin_popcountll(lm)); }
        else if (w > 0u) atomicAdd(&hist[1023u - min(w >> 2, 1023u)], 1u);   // bucket 0 = most work
        const uint64_t bm = __builtin_amdgcn_ballot_w64(w > idle);
        if ((tid & 63) == 0 && bm != 0ull) atomicAdd(&s_busy, (uint32_t)__builtin_popcountll(bm));
    }
    __syncthreads();
    uint32_t tot;
    const uint32_t ex = block_exclusive_scan<1024>(hist[tid], wave_sums, tot);
    hist[tid] = ex;
    __syncthreads();
    // sorted items and their work: in LDS when the list is short enough (the dealing passes below read them back one pass
    // after the other: from global memory every pass costs a round trip)
    __shared__ uint32_t s_items[ORDER_LDS_ITEMS], s_work[ORDER_LDS_ITEMS];
    const bool in_lds = per_list <= ORDER_LDS_ITEMS;
    uint32_t* items = in_lds ? s_items : items_ws + (size_t)x * per_list;
    uint32_t* work = in_lds ? s_work : work_ws + (size_t)x * per_list;
    for (int i0 = 0; i0 < per_list; i0 += 1024) {
        const int i = i0 + tid;
        const int tile = (i >> 2) * 8 + x;
        const uint32_t w = (i < per_list && tile < ntiles) ? item_cost(item_src, hint, tile * 4 + (i & 3)) + idle : 0u;
        const bool lightest = w > 0u && (w >> 2) == 0u;
        const uint64_t lm = __builtin_amdgcn_ballot_w64(lightest);
        uint32_t pos = 0;
        if (lightest) {
            const int leader = __builtin_ctzll(lm);
            uint32_t base = 0;
            if ((tid & 63) == leader) base = atomicAdd(&hist[1023], (uint32_t)__builtin_popcountll(lm));
            base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
            pos = base + (uint32_t)__builtin_popcountll(lm & ((1ull << (tid & 63)) - 1ull));
        } else if (w > 0u) {
            pos = atomicAdd(&hist[1023u - min(w >> 2, 1023u)], 1u);
        }
        if (w > 0u) {
            items[pos] = ((uint32_t)tile << 2) | (uint32_t)(i & 3);
            work[pos] = w;
        }
        // total work of the list (for the priority classes): wave sum, one atomic per wave
        unsigned long long ws = w;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) ws += __shfl_xor(ws, d, 64);
        if ((tid & 63) == 0 && ws != 0ull) atomicAdd(&s_total, ws);
    }
    __threadfence_block();
    __syncthreads();

    // NQ queues = the SIMDs of an XCC (all XCCs of a device have the same CU count; XCC x was counted above or, for the
    // backward, by the forward's call)
    __syncthreads();
    const int Q = (int)cu_state[MRGS_CS_NCU + x];
    const int NQ = 4 * min(max(Q, 1), MRGS_MAX_SIMD_QUEUES / 4);
    const int n_items = (int)tot;
    const int passes = (n_items + NQ - 1) / NQ;
    const unsigned long long mean5 = n_items > 0 ? 5ull * s_total / (unsigned long long)n_items : 0ull;   // 5 x mean work
    uint32_t* assign = assign_ws + (size_t)x * (per_list + MRGS_MAX_SIMD_QUEUES);
    __shared__ uint32_t qrank[MRGS_MAX_SIMD_QUEUES];
    // items with work: sequential passes, the heaviest item of a pass to the queue with the least load so far
    const int n_adapt = min(min(passes, ORDER_ADAPTIVE_PASSES), ((int)s_busy + NQ - 1) / NQ);
    for (int p = 0; p < n_adapt; p++) {
        if (p > 0) {
            // rank of every queue by load (ascending, ties by index): thread (part, q) counts the queues of its eighth that
            // come before q
            if (tid < NQ) qrank[tid] = 0u;
            __syncthreads();
            const int q = tid & (MRGS_MAX_SIMD_QUEUES - 1), part = tid >> 7;
            if (q < NQ) {
                const uint32_t mine = load[q];
                uint32_t r = 0;
                for (int o = part; o < NQ; o += 8) {
                    const uint32_t lo = load[o];
                    r += (lo < mine || (lo == mine && o < q)) ? 1u : 0u;
                }
                if (r) atomicAdd(&qrank[q], r);
            }
            __syncthreads();
        }
        if (tid < NQ) {
            const int rank = p * NQ + (p > 0 ? (int)qrank[tid] : tid);
            uint32_t entry = 0xFFFFFFFFu;
            if (rank < n_items) {
                // issue priority of the wave (bits 29-30): the heavy items of a SIMD run ahead of its light ones
                const unsigned long long w25 = 25ull * work[rank];
                const uint32_t prio = w25 > 8ull * mean5 ? 3u : w25 > 6ull * mean5 ? 2u : w25 > 4ull * mean5 ? 1u : 0u;
                entry = items[rank] | (prio << 29);
                load[tid] += work[rank];
            }
            assign[p * NQ + tid] = entry;
        }
        __syncthreads();
    }
    // the rest (idle items of the forward, very long lists): plain snake dealing, all passes at once
    for (int e = n_adapt * NQ + tid; e < passes * NQ; e += 1024) {
        const int p = e / NQ, q = e - p * NQ;
        const int rank = p * NQ + ((p & 1) ? NQ - 1 - q : q);
        assign[e] = rank < n_items ? items[rank] : 0xFFFFFFFFu;
    }
    if (tid == 0) {
        qstate[MRGS_QS_COUNT + x] = tot;
        qstate[MRGS_QS_PASSES + x] = (uint32_t)passes | ((uint32_t)NQ << 16);
        if (qstate_twin != nullptr) {
            qstate_twin[MRGS_QS_COUNT + x] = tot;
            qstate_twin[MRGS_QS_PASSES + x] = (uint32_t)passes | ((uint32_t)NQ << 16);
        }
    }
}

void mrgs_launch_blend_order(const MrgsImgWs& img, const uint32_t* census, int ntiles, int backward, void* bulk_zero, size_t bulk_zero_bytes,
                             const uint32_t* fwd_hint, hipStream_t stream)
{
    const size_t f4 = bulk_zero ? bulk_zero_bytes / sizeof(float4) : 0;   // callers pass multiples of 16 bytes
    const unsigned extra = f4 ? (unsigned)((f4 + 8191) / 8192 < 504 ? (f4 + 8191) / 8192 : 504) : 0u;
    if (backward)
        hipLaunchKernelGGL(blend_order_kernel, dim3(8 + extra), dim3(1024), 0, stream, img.item_work, ntiles, img.order_items, img.order_work,
                           img.bwd_assign, img.q_bwd, census, img.blend_state + MRGS_CS_BASE, 0, (uint32_t*)nullptr,
                           (float4*)bulk_zero, f4, (const uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr);
    else
        hipLaunchKernelGGL(blend_order_kernel, dim3(8 + extra), dim3(1024), 0, stream, img.item_est, ntiles, img.order_items, img.order_work,
                           img.fwd_assign, img.blend_state + MRGS_QS_FWD, census, img.blend_state + MRGS_CS_BASE, 1, img.item_work,
                           (float4*)bulk_zero, f4, fwd_hint, bulk_zero ? img.q_bwd : (uint32_t*)nullptr, img.redo_list);
}

void mrgs_launch_tile_ranges(const uint32_t* tile_key, const uint32_t* plist, int64_t R, const uint32_t* R_dev, const float4* rec,
                             uint8_t* qmask, const MrgsImgWs& img, int tiles_x, int ntiles, hipStream_t stream)
{
    // (ranges and item_est were cleared by duplicate_kernel)
    if (R > 0)
        hipLaunchKernelGGL(tile_ranges_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, stream, tile_key, plist, R, R_dev, rec,
                           tiles_x, img.ranges, qmask, img.item_est);
}
