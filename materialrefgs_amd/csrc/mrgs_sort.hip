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
#define SORT_ITERS (MRGS_SORT_TILE / SORT_THREADS)
#define RADIX_FUSED_MAX_BLOCKS 1024

__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << (threadIdx.x & 63)) - 1ull; }

// ---- radix pass 1/3: per-block digit histogram ----------------------------------------------------
__global__ void __launch_bounds__(SORT_THREADS) radix_hist_kernel(const uint32_t* __restrict__ keys, uint32_t* __restrict__ hist,
                                                                  int64_t n, int shift, int nblk)
{
    __shared__ uint32_t h[256];
    const int tid = threadIdx.x;
    h[tid] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * MRGS_SORT_TILE;
#pragma unroll 4
    for (int it = 0; it < SORT_ITERS; it++) {
        const int64_t idx = base + it * SORT_THREADS + tid;
        if (idx < n) atomicAdd(&h[(keys[idx] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(size_t)blockIdx.x * 256 + tid] = h[tid];   // block-major: digit d of consecutive blocks is read coalesced
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

// ---- radix pass 2/3: exclusive scan of the digit-major histogram (single workgroup) ------------------
// (only used for very large inputs, nblk > RADIX_FUSED_MAX_BLOCKS; otherwise the scatter kernel derives its offsets itself)
__global__ void __launch_bounds__(1024) radix_scan_kernel(uint32_t* __restrict__ hist, int nblk)
{
    __shared__ uint32_t wave_sums[16];
    uint32_t carry = 0;
    const int total = 256 * nblk;
    for (int base = 0; base < total; base += 1024) {
        const int i = base + threadIdx.x;                       // digit-major scan order over the block-major matrix
        const size_t at = i < total ? (size_t)(i % nblk) * 256 + (size_t)(i / nblk) : 0;
        uint32_t v = i < total ? hist[at] : 0u;
        uint32_t tot;
        uint32_t ex = block_exclusive_scan<1024>(v, wave_sums, tot);
        if (i < total) hist[at] = carry + ex;
        carry += tot;
    }
}

// ---- radix pass 3/3: stable scatter; in-wave ranks from 8 ballots (wave64 multi-split) ---------------
__global__ void __launch_bounds__(SORT_THREADS) radix_scatter_kernel(const uint32_t* __restrict__ kin, const uint32_t* __restrict__ vin,
                                                                     uint32_t* __restrict__ kout, uint32_t* __restrict__ vout,
                                                                     const uint32_t* __restrict__ offs, int64_t n, int shift, int nblk,
                                                                     int fused)
{
    __shared__ uint32_t running[256];
    __shared__ uint32_t wcnt[SORT_WAVES][256];
    __shared__ uint32_t scan_tmp[SORT_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (fused) {
        // offs is the raw block-major histogram matrix: this block's start for digit d is
        //   sum_{d' < d} total[d'] + sum_{b' < b} hist[b'][d]   -- nblk coalesced loads per thread, no separate scan launch
        uint32_t below = 0, total = 0;
        const int me = blockIdx.x;
#pragma unroll 8
        for (int bb = 0; bb < nblk; bb++) {
            const uint32_t v = offs[(size_t)bb * 256 + tid];
            total += v;
            below += bb < me ? v : 0u;
        }
        uint32_t tot;
        running[tid] = block_exclusive_scan<SORT_THREADS>(total, scan_tmp, tot) + below;
    } else {
        running[tid] = offs[(size_t)blockIdx.x * 256 + tid];
    }
#pragma unroll
    for (int w = 0; w < SORT_WAVES; w++) wcnt[w][tid] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * MRGS_SORT_TILE;
    const uint64_t lt = lanemask_lt();
    for (int it = 0; it < SORT_ITERS; it++) {
        const int64_t idx = base + it * SORT_THREADS + tid;
        const bool valid = idx < n;
        const uint32_t key = valid ? kin[idx] : 0xFFFFFFFFu;
        const uint32_t val = valid ? vin[idx] : 0u;
        const uint32_t d = (key >> shift) & 255u;
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const bool bit = (d >> b) & 1u;
            const uint64_t m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        const uint32_t rank = __popcll(peers & lt);
        if (valid && rank == 0) wcnt[wave][d] = __popcll(peers);
        __syncthreads();
        if (valid) {
            uint32_t pos = running[d] + rank;
#pragma unroll
            for (int w = 0; w < SORT_WAVES; w++)
                if (w < wave) pos += wcnt[w][d];
            kout[pos] = key;
            vout[pos] = val;
        }
        __syncthreads();
        uint32_t s = 0;
#pragma unroll
        for (int w = 0; w < SORT_WAVES; w++) { s += wcnt[w][tid]; wcnt[w][tid] = 0; }
        running[tid] += s;
        __syncthreads();
    }
    (void)lane;
}

int mrgs_radix_sort_pairs(uint32_t* key[2], uint32_t* val[2], uint32_t* hist, int64_t n, int bit_lo, int bit_hi,
                          hipStream_t stream)
{
    int cur = 0;
    if (n <= 0) return cur;
    const int nblk = (int)((n + MRGS_SORT_TILE - 1) / MRGS_SORT_TILE);
    for (int shift = bit_lo; shift < bit_hi; shift += 8) {
        hipLaunchKernelGGL(radix_hist_kernel, dim3(nblk), dim3(SORT_THREADS), 0, stream, key[cur], hist, n, shift, nblk);
        const int fused = nblk <= RADIX_FUSED_MAX_BLOCKS;
        if (!fused) hipLaunchKernelGGL(radix_scan_kernel, dim3(1), dim3(1024), 0, stream, hist, nblk);
        hipLaunchKernelGGL(radix_scatter_kernel, dim3(nblk), dim3(SORT_THREADS), 0, stream, key[cur], val[cur], key[cur ^ 1],
                           val[cur ^ 1], hist, n, shift, nblk, fused);
        cur ^= 1;
    }
    return cur;
}

// ---- exclusive scan of tiles_touched in depth-sorted order (rasterizer_impl.cu:283, InclusiveSum) -----
#define SCAN_THREADS 256
#define SCAN_PER_THREAD 8
#define SCAN_TILE (SCAN_THREADS * SCAN_PER_THREAD)

__global__ void __launch_bounds__(SCAN_THREADS) scan_block_sums_kernel(const uint32_t* __restrict__ tiles_touched,
                                                                       const uint32_t* __restrict__ order,
                                                                       uint32_t* __restrict__ block_sums, int n)
{
    __shared__ uint32_t wave_sums[SCAN_THREADS / 64];
    const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_PER_THREAD;
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; k++)
        if (base + k < n) s += tiles_touched[order[base + k]];
    uint32_t tot;
    block_exclusive_scan<SCAN_THREADS>(s, wave_sums, tot);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(1024) scan_sums_kernel(uint32_t* __restrict__ block_sums, int nblk, uint32_t* __restrict__ total_out)
{
    __shared__ uint32_t wave_sums[16];
    uint32_t carry = 0;
    for (int base = 0; base < nblk; base += 1024) {
        const int i = base + threadIdx.x;
        uint32_t v = i < nblk ? block_sums[i] : 0u;
        uint32_t tot;
        uint32_t ex = block_exclusive_scan<1024>(v, wave_sums, tot);
        if (i < nblk) block_sums[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) *total_out = carry;
}

__global__ void __launch_bounds__(SCAN_THREADS) scan_final_kernel(const uint32_t* __restrict__ tiles_touched,
                                                                  const uint32_t* __restrict__ order,
                                                                  const uint32_t* __restrict__ block_sums,
                                                                  uint32_t* __restrict__ offsets, int n)
{
    __shared__ uint32_t wave_sums[SCAN_THREADS / 64];
    const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_PER_THREAD;
    uint32_t v[SCAN_PER_THREAD], s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; k++) {
        v[k] = (base + k < n) ? tiles_touched[order[base + k]] : 0u;
        s += v[k];
    }
    uint32_t tot;
    uint32_t ex = block_exclusive_scan<SCAN_THREADS>(s, wave_sums, tot) + block_sums[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; k++) {
        if (base + k < n) offsets[base + k] = ex;
        ex += v[k];
    }
}

void mrgs_scan_tiles(const uint32_t* tiles_touched, const uint32_t* order, uint32_t* offsets, uint32_t* block_sums,
                     uint32_t* total_out, int n, hipStream_t stream)
{
    const int nblk = (n + SCAN_TILE - 1) / SCAN_TILE;
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3(nblk), dim3(SCAN_THREADS), 0, stream, tiles_touched, order, block_sums, n);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, stream, block_sums, nblk, total_out);
    hipLaunchKernelGGL(scan_final_kernel, dim3(nblk), dim3(SCAN_THREADS), 0, stream, tiles_touched, order, block_sums, offsets, n);
}

// ---- pair emission (duplicateWithKeys, rasterizer_impl.cu:72-113) in depth-sorted gaussian order -------
__global__ void __launch_bounds__(256) duplicate_kernel(int P, const uint32_t* __restrict__ order, const uint32_t* __restrict__ tiles_touched,
                                                        const uint32_t* __restrict__ offsets, const uint2* __restrict__ rect,
                                                        int tiles_x, uint32_t* __restrict__ tile_key, uint32_t* __restrict__ plist)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
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
                           uint32_t* plist, hipStream_t stream)
{
    const int tiles_x = (cfg.W + MRGS_BLOCK_X - 1) / MRGS_BLOCK_X;
    hipLaunchKernelGGL(duplicate_kernel, dim3((cfg.P + 255) / 256), dim3(256), 0, stream, cfg.P, order, g.tiles_touched, g.offsets,
                       g.rect, tiles_x, tile_key, plist);
}

// ---- identifyTileRanges (rasterizer_impl.cu:118-140) on the sorted tile ids ---------------------------
__global__ void __launch_bounds__(256) tile_ranges_kernel(const uint32_t* __restrict__ tile_key, int64_t R, uint2* __restrict__ ranges)
{
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
__global__ void __launch_bounds__(1024) tile_order_kernel(const uint2* __restrict__ ranges, int ntiles, int nslots, uint32_t* __restrict__ order)
{
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

void mrgs_launch_tile_ranges(const uint32_t* tile_key, int64_t R, uint2* ranges, uint32_t* tile_order, int ntiles, hipStream_t stream)
{
    (void)hipMemsetAsync(ranges, 0, sizeof(uint2) * (size_t)ntiles, stream);   // rasterizer_impl.cu:316
    if (R > 0)
        hipLaunchKernelGGL(tile_ranges_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, stream, tile_key, R, ranges);
    hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), 0, stream, ranges, ntiles, ((ntiles + 7) / 8) * 8, tile_order);
}
