// mrgs_internal.h -- shared declarations of the libmrgs.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mrgs.h"

#define MRGS_BLOCK_X 16
#define MRGS_BLOCK_Y 16
#define MRGS_NEAR_N 0.2f             // auxiliary.h:39
#define MRGS_FAR_N 100.0f            // auxiliary.h:40
#define MRGS_FILTER_INV_SQUARE 2.0f  // auxiliary.h:41

// Packed per-gaussian render record written by preprocess and gathered by the blend kernels:
// 5 x float4 = 80 B, 16-byte aligned so that a record is fetched with dwordx4 loads.
//   [0] Tu.xyz, Tv.x   [1] Tv.yz, Tw.xy   [2] Tw.z, mean2D.xy, opacity   (geometry)
//   [3] normal.xyz, rgb.r   [4] rgb.gb, depth, 0                          (appearance)
// (the reference keeps these in five separate arrays: transMat, means2D, normal_opacity, rgb; forward.cu:350-357,427)
// The cull conic (mrgs_blend_math.h) lives in its own array, 3 x float4 per gaussian: it is gathered once per (tile, surfel) pair
// by the emission kernel and by the forward blend only for its live-pixel cull.
//   [0] ellipse centre.xy, A, C   [1] B/C, B/A, det/C, det/A   [2] mean2D.xy, disc r^2, 0
#define MRGS_REC_F4 5
#define MRGS_CULL_F4 3

// Packed per-gaussian gradient accumulator of the blend backward (one row per gaussian so that the
// atomics of one (tile, gaussian) pair land in one or two cache lines):
//   [0..8] dL/dT (Tu,Tv,Tw)  [9] dL/dopacity  [10..12] dL/dnormal  [13..15] dL/dcolor  [16..16+SMAX) dL/dfeature
//   [16+SMAX, 17+SMAX] dL/dmean2D.xy ; SMAX = channel capacity of the blend-backward kernel instance that serves S
//   channels (instances exist for 0 / 8 / 12 / 24); row stride = MRGS_GRAD_STRIDE(S) floats.  The first 16 (+SMAX)
//   values are what one transposing wave reduction produces, 16 consecutive floats per atomic instruction.
#define MRGS_SMAX(S) ((S) == 0 ? 0 : (S) <= 8 ? 8 : (S) <= 12 ? 12 : 24)
#define MRGS_G_DT 0
#define MRGS_G_OPA 9
#define MRGS_G_NRM 10
#define MRGS_G_COL 13
#define MRGS_G_FEAT 16
#define MRGS_G_M2(SMAX) (16 + (SMAX))
#define MRGS_GRAD_STRIDE(S) ((18 + MRGS_SMAX(S) + 3) & ~3)

static inline size_t mrgs_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct MrgsGeomWs {   // carved from geom_ws (all offsets 256-B aligned)
    float4* rec;            // [P][MRGS_REC_F4]
    float4* cull;           // [P][MRGS_CULL_F4]
    uint32_t* depth_key[2]; // [P] ping-pong radix keys (depth bits, 0xFFFFFFFF when culled)
    uint32_t* order[2];     // [P] ping-pong payload: gaussian index
    uint2* rect;            // [P] tile rect packed: x = min.x | min.y<<16, y = max.x | max.y<<16
    uint32_t* tiles_touched;// [P]
    uint32_t* offsets;      // [P] exclusive scan of tiles_touched in depth-sorted order
    uint8_t* clamped;       // [P] bit c set when SH colour channel c was clamped
    // cleared at the start of every forward (one memset over [counters, scan_ws end)):
    uint32_t* counters;     // [16] 0: num_rendered, 1: error flag of the look-back kernels; then [MRGS_CENSUS_WORDS] CU census
    uint32_t* sort_ws;      // tickets / digit totals / status words of the depth sort (mrgs_sort_ws_words)
    uint32_t* scan_ws;      // ticket / status words of the tiles_touched scan (mrgs_scan_ws_words)
    size_t clear_bytes;     // size of the region that starts at counters
    // tile binning without a global sort (mrgs_binning.hip); null when the image has more tiles than that path serves
    uint32_t* tile_mat;     // [groups][Tpad] pairs of surfel slice g in tile t -> exclusive prefix over the slices
    uint32_t* tile_cnt;     // [Tpad] pairs per tile
    uint32_t* tile_loc;     // [Tpad] exclusive scan of tile_cnt inside the tile's 64-tile chunk
    uint32_t* chunk_tot;    // [Tpad / 64]
    uint32_t* chunk_base;   // [Tpad / 64] pairs of the chunks before
    uint32_t* big_list;     // [Tpad] tiles whose segment exceeds the small sort kernel's LDS capacity (count: counters[3])
    size_t total;
};
// counters[]: 0 num_rendered, 1 error flag of the look-back kernels, 2 ticket of tile_scan_kernel, 6 length of big_list

struct MrgsImgWs {
    uint2* ranges;       // [tiles]
    uint32_t* item_est;  // [4 * tiles] directly behind ranges (cleared together): entries that pass the cull per (tile, quadrant)
    size_t ranges_est_bytes;
    uint32_t* item_work; // [4 * tiles] cost of each forward wave's walk (item = tile * 4 + quadrant) = work of its backward wave
    uint32_t* order_items, *order_work;   // [8][per_list] scratch of blend_order_kernel (items of an XCD list by decreasing work)
    uint32_t* fwd_assign, *bwd_assign;    // [8][per_list + 128] item handed to ticket t of SIMD queue q at [t * NQ + q]
    uint32_t* blend_state;                // MRGS_BLEND_STATE_WORDS: forward queue state | (backward queue state) | CU numbering -- in the camera's
                                          // hint buffer when the caller passed one (mrgs_api.hip: img_use_hint)
    uint32_t* q_bwd;                      // queue state of the backward blend: always in THIS render's workspace (two renders of one camera
                                          // may have their backwards pending at the same time)
    float* final_T;      // [3][H*W]: T, M1, M2
    uint32_t* n_contrib; // [2][H*W]: last, median
    uint32_t* redo_list; // [2 + H*W]: [0] count (cleared by the forward's ordering launch), [2 + i] pixel index of the i-th pixel whose
                         // decisions the forward blend could not take for sure (mrgs_blend_math.h "Exact decisions")
    size_t total;
};

struct MrgsBinWs {
    uint32_t* tile_key[2];  // [R] ping-pong: tile id of each pair (one contiguous block: the 64-bit keys of mrgs_binning.hip alias it)
    uint32_t* plist[2];     // [R] ping-pong: gaussian index of each pair
    uint32_t* sort_ws;      // [16] 0: error flag; then tickets / digit totals / status words of the tile-id sort; cleared per forward
    size_t sort_ws_bytes;
    uint8_t* qmask;         // [R] bit q: the entry's surfel can touch quadrant q of its tile (tile_ranges_kernel)
    uint8_t* cflag;         // [R][4] written by the forward blend: 1 = some pixel of quadrant q blended the entry (the backward walks only these)
    size_t total;
};

// Per-camera hint buffer (MrgsRasterInputs::work_hint), uint32 words: [0, 4 T) work of every (tile, quadrant) at the last visit |
// blend_state (MRGS_BLEND_STATE_WORDS: queue state of both blend kernels + CU numbering) | fwd_assign (the dealt queues).  With a hint
// buffer the last two live THERE instead of in the per-call image workspace, so that a later visit can reuse them (MRGS_HINT_REUSE_ORDER).
struct MrgsHintLayout { size_t work, blend_state, fwd_assign, total_words; };
MrgsHintLayout mrgs_hint_layout(int H, int W);
MrgsGeomWs mrgs_carve_geom(void* base, int P, int H, int W);
MrgsImgWs mrgs_carve_img(void* base, int H, int W);
MrgsBinWs mrgs_carve_bin(void* base, int64_t R);

// ---- kernel launchers (one per translation unit) ---------------------------------------------------
// radix sort of (u32 key, u32 value) pairs on bits [bit_lo, bit_hi); returns index (0/1) of the buffer
// that holds the sorted result.  hist must hold 256 * ceil(n / SORT_TILE) u32.

#define MRGS_SORT_WS_HEADER 16
// Work queues of the blend kernels (mrgs_sort.hip: blend_order_kernel; mrgs_blend_math.h: mrgs_pull_item).
// MrgsImgWs::blend_state (uint32 words) = queue state of the forward | queue state of the backward | CU numbering.
// A CU is identified by (XCC_ID[2:0], HW_ID se[2:0] sh cu[3:0]) = 3 + 8 bits.
#define MRGS_MAX_SIMD_QUEUES 128             // 32 CUs x 4 SIMDs per XCD
#define MRGS_QS_COUNT 0                      // [8]       work items of each XCD list
#define MRGS_QS_PASSES 8                     // [8]       dealing passes | NQ << 16 of each XCD list
#define MRGS_QS_TICKET 16                    // [8][128]  items handed out per (XCD list, SIMD queue)
#define MRGS_QS_WORDS (16 + 8 * MRGS_MAX_SIMD_QUEUES)
#define MRGS_QS_FWD 0
#define MRGS_QS_BWD MRGS_QS_WORDS
#define MRGS_CS_BASE (2 * MRGS_QS_WORDS)
#define MRGS_CS_NCU 0                        // [8]       CUs per XCC
#define MRGS_CS_DENSE 8                      // [8][256]  CU key -> dense index inside its XCC
#define MRGS_BLEND_STATE_WORDS (2 * MRGS_QS_WORDS + 8 + 2048)
#define MRGS_CENSUS_WORDS 2048               // [8][256] flag per CU key: a preprocess wave ran there; in MrgsGeomWs::counters + 16
__device__ __forceinline__ uint32_t mrgs_cu_key(uint32_t hw_id) { return ((hw_id >> 8) & 0xFFu); }   // cu[3:0] sh se[2:0]
// CU census bit of the calling wave (set by the preprocess waves, read by blend_order_kernel)
__device__ __forceinline__ void mrgs_census_mark(uint32_t* __restrict__ census)
{
    const uint32_t key = mrgs_cu_key(__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)));
    const uint32_t xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 7u;
    census[xcc * 256 + key] = 1u;   // plain idempotent store, nothing in the wave waits for it
}
// stable LSD radix sort of (key, value) pairs on key bits [bit_lo, bit_hi); returns the index of the buffer holding the
// result.  ws: mrgs_sort_ws_words(n) zeroed words; *error_flag is set if a look-back spin overruns (never expected).
size_t mrgs_sort_ws_words(int64_t n);
// n_dev (nullable): device-resident element count; the launches are then sized for n (a capacity) and use min(*n_dev, n)
int mrgs_radix_sort_pairs(uint32_t* key[2], uint32_t* val[2], uint32_t* ws, uint32_t* error_flag, int64_t n, const uint32_t* n_dev,
                          int bit_lo, int bit_hi, hipStream_t stream);
// exclusive scan of tiles_touched[order[i]] -> offsets[i]; total -> *total_out (device); ws: mrgs_scan_ws_words(n) zeroed words
size_t mrgs_scan_ws_words(int n);
void mrgs_scan_tiles(const uint32_t* tiles_touched, const uint32_t* order, uint32_t* offsets, uint32_t* ws, uint32_t* total_out,
                     uint32_t* error_flag, int n, hipStream_t stream);

// tile binning without a global sort (mrgs_binning.hip)
int mrgs_bin_groups(int P);
int mrgs_bin_tpad(int T);
bool mrgs_bin_supported(int T);
void mrgs_launch_tile_count_scan(const MrgsRasterConfig& cfg, const MrgsGeomWs& g, uint32_t* host_slot, hipStream_t stream);
// reuse_order: no ordering launch follows -- the tile sort's workgroups reset the queue tickets, copy the forward's queue shape to the
// backward's, clear item_work and (bulk_zero) the gradient rows of the coming backward
void mrgs_launch_tile_emit_sort(const MrgsRasterConfig& cfg, const MrgsGeomWs& g, const MrgsBinWs& b, const MrgsImgWs& img, int64_t capacity,
                                bool reuse_order, void* bulk_zero, size_t bulk_zero_bytes, hipStream_t stream);

void mrgs_launch_preprocess_fwd(const MrgsRasterConfig& cfg, const MrgsRasterInputs& in, const MrgsGeomWs& g, int32_t* radii,
                                hipStream_t stream);
void mrgs_launch_preprocess_bwd(const MrgsRasterConfig& cfg, const MrgsRasterInputs& in, const MrgsGeomWs& g,
                                const int32_t* radii, const float* grad_rec, const MrgsRasterGrads& out, hipStream_t stream);
void mrgs_launch_color_grad_extract(const MrgsRasterConfig& cfg, const MrgsGeomWs& g, const int32_t* radii, const float* grad_rec, bool from_sh,
                                    float* out, hipStream_t stream);
void mrgs_launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present, hipStream_t stream);

void mrgs_launch_duplicate(const MrgsRasterConfig& cfg, const MrgsGeomWs& g, const uint32_t* order, uint32_t* tile_key,
                           uint32_t* plist, int64_t capacity, const uint32_t* R_dev, const MrgsBinWs& b, const MrgsImgWs& img, uint32_t* host_slot,
                           hipStream_t stream);
void mrgs_launch_tile_ranges(const uint32_t* tile_key, const uint32_t* plist, int64_t R, const uint32_t* R_dev, const float4* rec,
                             uint8_t* qmask, const MrgsImgWs& img, int tiles_x, int ntiles, hipStream_t stream);
// bulk_zero (nullable, 16-byte aligned, size a multiple of 16): cleared by extra workgroups of the same launch
void mrgs_launch_blend_order(const MrgsImgWs& img, const uint32_t* census, int ntiles, int backward, void* bulk_zero, size_t bulk_zero_bytes,
                             const uint32_t* fwd_hint,
                             hipStream_t stream);

void mrgs_launch_render_fwd(const MrgsRasterConfig& cfg, const MrgsRasterInputs& in, const MrgsGeomWs& g, const uint32_t* plist,
                            const uint8_t* qmask, uint8_t* cflag, const MrgsImgWs& img, float* out_color, float* out_feature, float* out_others, hipStream_t stream);
void mrgs_launch_render_bwd(const MrgsRasterConfig& cfg, const MrgsRasterInputs& in, const MrgsGeomWs& g, const uint32_t* plist,
                            const uint8_t* cflag, const MrgsImgWs& img, const float* dL_dpix, const float* dL_dpix_f, const float* dL_dothers,
                            float* grad_rec, bool forward_queues, hipStream_t stream);

#ifndef MRGS_EXP
#define MRGS_EXP(x) expf(x)
#endif
