// mrgs_api.hip -- C-ABI entry points of libmrgs.so (see include/mrgs.h for the contract and the reference
// interfaces each entry replaces).
#include <stdio.h>
#include <string.h>
#include <atomic>
#include <mutex>
#include <vector>
#include "mrgs_internal.h"

static thread_local char g_hip_err[256] = "";
static int g_profiling = 0;

#define HIP_TRY(expr)                                                                                       \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) {                                                                             \
            snprintf(g_hip_err, sizeof(g_hip_err), "%s at %s:%d", hipGetErrorString(e_), __FILE__, __LINE__); \
            return MRGS_E_HIP;                                                                              \
        }                                                                                                   \
    } while (0)

// CHECK_CUDA(A, debug) of the reference (auxiliary.h:303-310): with cfg.debug every stage is synchronised.
#define STAGE_CHECK(cfg, stream)                                 \
    do {                                                         \
        HIP_TRY(hipGetLastError());                              \
        if ((cfg)->debug) HIP_TRY(hipStreamSynchronize(stream)); \
    } while (0)

namespace {
// Non-blocking per-stage timing: while profiling is enabled every stage records a HIP event pair on the
// launch stream (no host synchronisation inside the timed region); mrgs_get_kernel_times() synchronises the
// recorded pairs afterwards and returns the mean duration of each stage since the last reset.
enum { ST_PRE = 0, ST_SORT, ST_DUP, ST_FWD, ST_BWD, ST_PREB, ST_COUNT };
struct EvPair { hipEvent_t a, b; int stage; };
static std::vector<EvPair> g_pairs;
static std::vector<EvPair> g_free;
static std::mutex g_prof_mutex;   // autograd runs the backward (and its StageTimer) on a worker thread

struct StageTimer {
    EvPair p;
    hipStream_t s;
    bool on;
    // level 1: every stage; 2: the two blend stages; 3: the backward blend stage (the dominant kernel) on every fourth call --
    // an event pair costs two ~6 us bubbles on the stream, which a throughput measurement should not pay on every launch
    static bool sampled(int stage)
    {
        static std::atomic<unsigned> calls{0};   // autograd runs the backward on worker threads
        return stage == ST_BWD && (calls.fetch_add(1u, std::memory_order_relaxed) & 3u) == 0u;
    }
    StageTimer(hipStream_t stream, int stage)
        : s(stream), on(g_profiling == 1 || (g_profiling == 2 && (stage == ST_FWD || stage == ST_BWD)) || (g_profiling == 3 && sampled(stage)))
    {
        if (on) {
            bool reuse = false;
            {
                std::lock_guard<std::mutex> lk(g_prof_mutex);
                if (!g_free.empty()) { p = g_free.back(); g_free.pop_back(); reuse = true; }
            }
            if (!reuse) { (void)hipEventCreate(&p.a); (void)hipEventCreate(&p.b); }
            p.stage = stage;
            (void)hipEventRecord(p.a, s);
        }
    }
    void stop()
    {
        if (on) {
            (void)hipEventRecord(p.b, s);
            std::lock_guard<std::mutex> lk(g_prof_mutex);
            g_pairs.push_back(p);
            on = false;
        }
    }
};

struct Carver {
    char* p;
    size_t used = 0;
    explicit Carver(void* base) : p((char*)base) {}
    template <typename T>
    T* take(size_t n)
    {
        used = mrgs_align_up(used, 256);
        T* r = p ? (T*)(p + used) : nullptr;
        used += sizeof(T) * n;
        return r;
    }
};
}   // namespace

MrgsGeomWs mrgs_carve_geom(void* base, int P, int H, int W)
{
    Carver c(base);
    MrgsGeomWs g;
    const size_t n = (size_t)(P > 0 ? P : 1);
    g.rec = c.take<float4>(n * MRGS_REC_F4);
    g.cull = c.take<float4>(n * MRGS_CULL_F4);
    g.depth_key[0] = c.take<uint32_t>(n);
    g.depth_key[1] = c.take<uint32_t>(n);
    g.order[0] = c.take<uint32_t>(n);
    g.order[1] = c.take<uint32_t>(n);
    g.rect = c.take<uint2>(n);
    g.tiles_touched = c.take<uint32_t>(n);
    g.offsets = c.take<uint32_t>(n);
    g.clamped = c.take<uint8_t>(n);
    g.counters = c.take<uint32_t>(16 + MRGS_CENSUS_WORDS);
    const size_t clear_from = c.used - (16 + MRGS_CENSUS_WORDS) * sizeof(uint32_t);
    g.sort_ws = c.take<uint32_t>(mrgs_sort_ws_words((int64_t)n));
    g.scan_ws = c.take<uint32_t>(mrgs_scan_ws_words((int)n));
    g.clear_bytes = c.used - clear_from;
    const int T = ((W + MRGS_BLOCK_X - 1) / MRGS_BLOCK_X) * ((H + MRGS_BLOCK_Y - 1) / MRGS_BLOCK_Y);
    g.tile_mat = g.tile_cnt = g.tile_loc = g.chunk_tot = g.chunk_base = g.big_list = nullptr;
    if (mrgs_bin_supported(T)) {
        const size_t Tpad = (size_t)mrgs_bin_tpad(T);
        g.tile_mat = c.take<uint32_t>((size_t)mrgs_bin_groups(P) * Tpad);
        g.tile_cnt = c.take<uint32_t>(Tpad);
        g.tile_loc = c.take<uint32_t>(Tpad);
        g.chunk_tot = c.take<uint32_t>(Tpad / 64);      // BIN_CHUNK tiles per chunk (mrgs_binning.hip)
        g.chunk_base = c.take<uint32_t>(Tpad / 64);
        g.big_list = c.take<uint32_t>(Tpad);
    }
    g.total = mrgs_align_up(c.used, 256);
    return g;
}

MrgsImgWs mrgs_carve_img(void* base, int H, int W)
{
    Carver c(base);
    MrgsImgWs w;
    const size_t hw = (size_t)H * W;
    const size_t tiles = (size_t)((W + MRGS_BLOCK_X - 1) / MRGS_BLOCK_X) * ((H + MRGS_BLOCK_Y - 1) / MRGS_BLOCK_Y);
    const size_t nt = tiles > 0 ? tiles : 1;
    w.ranges = c.take<uint2>(nt);
    const size_t ranges_from = c.used - sizeof(uint2) * nt;
    c.used = mrgs_align_up(c.used, 256);
    w.item_est = c.take<uint32_t>(4 * nt);
    w.ranges_est_bytes = c.used - ranges_from;
    w.item_work = c.take<uint32_t>(4 * nt);
    const size_t per_list = (nt + 7) / 8 * 4;
    w.order_items = c.take<uint32_t>(8 * per_list);
    w.order_work = c.take<uint32_t>(8 * per_list);
    w.fwd_assign = c.take<uint32_t>(8 * (per_list + MRGS_MAX_SIMD_QUEUES));
    w.bwd_assign = c.take<uint32_t>(8 * (per_list + MRGS_MAX_SIMD_QUEUES));
    w.blend_state = c.take<uint32_t>(MRGS_BLEND_STATE_WORDS);
    w.q_bwd = w.blend_state + MRGS_QS_BWD;
    w.final_T = c.take<float>(3 * hw);
    w.n_contrib = c.take<uint32_t>(2 * hw);
    w.redo_list = c.take<uint32_t>(2 + hw);
    w.total = mrgs_align_up(c.used, 256);
    return w;
}

MrgsBinWs mrgs_carve_bin(void* base, int64_t R)
{
    Carver c(base);
    MrgsBinWs b;
    const size_t n = (size_t)(R > 0 ? R : 1);
    const size_t n64 = mrgs_align_up(n, 64);
    b.tile_key[0] = c.take<uint32_t>(2 * n64);       // one block: [R] 64-bit keys of mrgs_binning.hip, or the two ping-pong halves
    b.tile_key[1] = b.tile_key[0] ? b.tile_key[0] + n64 : nullptr;
    b.plist[0] = c.take<uint32_t>(n);
    b.plist[1] = c.take<uint32_t>(n);
    b.sort_ws = c.take<uint32_t>(16 + mrgs_sort_ws_words((int64_t)n));
    b.sort_ws_bytes = sizeof(uint32_t) * (16 + mrgs_sort_ws_words((int64_t)n));
    b.qmask = c.take<uint8_t>(n);
    b.cflag = c.take<uint8_t>(4 * n);
    b.total = mrgs_align_up(c.used, 256);
    return b;
}

static int tile_bits(int ntiles)
{
    int bits = 1;
    while ((1 << bits) < ntiles) bits++;
    return bits;
}

// which of the two ping-pong buffers holds the final data after sorting `bits` bits in 8-bit passes
static int sorted_buf(int bits) { return ((bits + 7) / 8) & 1; }
// ... and which plist buffer holds the point list of a forward: buffer 0 on the tile-binning path (mrgs_binning.hip)
static int plist_buf(const MrgsGeomWs& g, int ntiles) { return g.tile_mat != nullptr ? 0 : sorted_buf(tile_bits(ntiles)); }

static int check_cfg(const MrgsRasterConfig* cfg, const MrgsRasterInputs* in)
{
    if (!cfg || !in) return MRGS_E_BAD_ARG;
    // a caller built against another revision of mrgs.h (e.g. MrgsRasterInputs without its trailing optional pointers, which the calls act
    // on) is refused before anything is read
    if (cfg->struct_size != sizeof(MrgsRasterConfig) || in->struct_size != sizeof(MrgsRasterInputs)) return MRGS_E_BAD_ARG;
    if (cfg->P < 0 || cfg->H <= 0 || cfg->W <= 0 || cfg->S < 0 || cfg->M < 0) return MRGS_E_BAD_ARG;
    if (cfg->S > MRGS_MAX_FEATURES) return MRGS_E_TOO_MANY_FEATURES;
    if (cfg->P > 0) {
        if (!in->means3D || !in->opacities || !in->viewmatrix || !in->projmatrix || !in->campos || !in->bg) return MRGS_E_BAD_ARG;
        if ((in->shs == nullptr) == (in->colors_precomp == nullptr)) return MRGS_E_NEED_COLORS;
        if (in->shs && cfg->M <= 0) return MRGS_E_BAD_ARG;
        if (in->shs_rest && (!in->shs || cfg->M < 2 || cfg->M > 16)) return MRGS_E_BAD_ARG;   // split SH: DC + 1..15 higher coefficients
        const bool have_sr = in->scales && in->rotations;
        if (have_sr == (in->transMat_precomp != nullptr)) return MRGS_E_BAD_ARG;
        if (cfg->S > 0 && !in->features) return MRGS_E_BAD_ARG;
        if ((cfg->W + 15) / 16 > 65535 || (cfg->H + 15) / 16 > 65535) return MRGS_E_UNSUPPORTED;
        if (cfg->P >= (1 << 28)) return MRGS_E_UNSUPPORTED;   // binning keys carry the gaussian index in 28 bits
        // the blend backward addresses a surfel's gradient row with a 32-bit byte offset
        if ((uint64_t)cfg->P * MRGS_GRAD_STRIDE(cfg->S) * sizeof(float) >= (1ull << 32)) return MRGS_E_UNSUPPORTED;
    }
    return MRGS_OK;
}

MrgsHintLayout mrgs_hint_layout(int H, int W)
{
    const size_t tiles = (size_t)((W + MRGS_BLOCK_X - 1) / MRGS_BLOCK_X) * ((H + MRGS_BLOCK_Y - 1) / MRGS_BLOCK_Y);
    const size_t nt = tiles > 0 ? tiles : 1, per_list = (nt + 7) / 8 * 4;
    MrgsHintLayout l;
    l.work = 0;
    l.blend_state = mrgs_align_up(4 * nt, 64);
    l.fwd_assign = mrgs_align_up(l.blend_state + MRGS_BLEND_STATE_WORDS, 64);
    l.total_words = mrgs_align_up(l.fwd_assign + 8 * (per_list + MRGS_MAX_SIMD_QUEUES), 64);
    return l;
}
// with a per-camera hint buffer the blend kernels' queue state and the forward's dealt queues live in it (see MrgsHintLayout)
static void side_mark_pre_blend(hipStream_t stream);     // (side stream, below) an event in front of the forward blend, when somebody forks from there

static void img_use_hint(MrgsImgWs& img, const MrgsRasterInputs* in, int H, int W)
{
    if (in->work_hint == nullptr) return;
    const MrgsHintLayout l = mrgs_hint_layout(H, W);
    img.blend_state = in->work_hint + l.blend_state;
    img.fwd_assign = in->work_hint + l.fwd_assign;
}

extern "C" {

size_t mrgs_geom_bytes(int32_t P, int32_t H, int32_t W) { return mrgs_carve_geom(nullptr, P, H, W).total; }
size_t mrgs_img_bytes(int32_t H, int32_t W) { return mrgs_carve_img(nullptr, H, W).total; }
size_t mrgs_binning_bytes(int64_t R) { return mrgs_carve_bin(nullptr, R).total; }
size_t mrgs_work_hint_bytes(int32_t H, int32_t W)
{
    if (H <= 0 || W <= 0) return 0;
    return mrgs_hint_layout(H, W).total_words * sizeof(uint32_t);
}
size_t mrgs_grad_bytes(int32_t P, int32_t S) { return mrgs_align_up((size_t)(P > 0 ? P : 1) * MRGS_GRAD_STRIDE(S) * sizeof(float), 256); }

// phase 1: preprocess, depth sort, scan; leaves num_rendered and the look-back error flag in g.counters[0..1]
static int enqueue_geom(const MrgsRasterConfig* cfg, const MrgsRasterInputs* in, MrgsGeomWs g, int32_t* radii, hipStream_t stream,
                        uint32_t* host_slot = nullptr, hipEvent_t slot_event = nullptr)
{
    StageTimer t0(stream, ST_PRE);
    mrgs_launch_preprocess_fwd(*cfg, *in, g, radii, stream);
    t0.stop();
    STAGE_CHECK(cfg, stream);

    StageTimer t1(stream, ST_SORT);
    if (g.tile_mat != nullptr) {
        // per-tile pair counts of fixed surfel slices, column prefixes, tile offsets, num_rendered (mrgs_binning.hip)
        mrgs_launch_tile_count_scan(*cfg, g, host_slot, stream);
        if (slot_event != nullptr) HIP_TRY(hipEventRecord(slot_event, stream));   // the host slot is written when this completes
    } else {
        // images with more tiles than the slice histograms hold: global radix path of round 1 (mrgs_sort.hip).
        // depth sort of the gaussians (32 key bits, 4 passes -> result back in buffer 0)
        const int cur = mrgs_radix_sort_pairs(g.depth_key, g.order, g.sort_ws, g.counters + 1, cfg->P, nullptr, 0, 32, stream);
        STAGE_CHECK(cfg, stream);
        mrgs_scan_tiles(g.tiles_touched, g.order[cur], g.offsets, g.scan_ws, g.counters, g.counters + 1, cfg->P, stream);
    }
    t1.stop();
    STAGE_CHECK(cfg, stream);
    return MRGS_OK;
}

// phase 2: pair emission, tile sort, ranges, blend.  R is the pair count the binning workspace was carved for; with
// R_dev != nullptr it is a capacity and the kernels take the actual count from device memory.
static int enqueue_render(const MrgsRasterConfig* cfg, const MrgsRasterInputs* in, MrgsGeomWs g, MrgsBinWs b,
                          const MrgsImgWs& img, int64_t R, const uint32_t* R_dev, float* out_color, float* out_feature,
                          float* out_others, hipStream_t stream, uint32_t* host_slot = nullptr, hipEvent_t slot_event = nullptr)
{
    const int tiles_x = (cfg->W + MRGS_BLOCK_X - 1) / MRGS_BLOCK_X, tiles_y = (cfg->H + MRGS_BLOCK_Y - 1) / MRGS_BLOCK_Y;
    const int ntiles = tiles_x * tiles_y;
    const int dcur = sorted_buf(32);

    StageTimer t0(stream, ST_DUP);
    int cur = 0;
    bool reuse_order = false;
    if (g.tile_mat != nullptr) {
        // pairs into their tile's segment (one LDS atomic each), per-tile LDS sort -> point_list, cull bits, ranges, work estimates
        // MRGS_HINT_REUSE_ORDER: the queues dealt at an earlier visit of this camera serve again (they are in the hint buffer); what the
        // ordering launch did besides ordering moves into the tile sort
        reuse_order = R > 0 && in->work_hint != nullptr && (in->hint_flags & MRGS_HINT_REUSE_ORDER) != 0u;
        if (R > 0) mrgs_launch_tile_emit_sort(*cfg, g, b, img, R, reuse_order, in->bwd_grad_ws, in->bwd_grad_ws ? mrgs_grad_bytes(cfg->P, cfg->S) : 0, stream);
        else HIP_TRY(hipMemsetAsync(img.ranges, 0, img.ranges_est_bytes, stream));   // nothing visible: the ranges read as empty
        STAGE_CHECK(cfg, stream);
    } else {
        if (R > 0) {
            mrgs_launch_duplicate(*cfg, g, g.order[dcur], b.tile_key[0], b.plist[0], R, R_dev, b, img, host_slot, stream);
            if (slot_event != nullptr) HIP_TRY(hipEventRecord(slot_event, stream));   // the host slot is written when this completes
        } else {   // nothing visible: no kernel touches the pair buffers, only the ranges have to read as empty
            HIP_TRY(hipMemsetAsync(img.ranges, 0, img.ranges_est_bytes, stream));
        }
        STAGE_CHECK(cfg, stream);
        const int bits = tile_bits(ntiles);
        cur = mrgs_radix_sort_pairs(b.tile_key, b.plist, b.sort_ws + 16, b.sort_ws, R, R_dev, 0, bits, stream);
        STAGE_CHECK(cfg, stream);
        mrgs_launch_tile_ranges(b.tile_key[cur], b.plist[cur], R, R_dev, g.cull, b.qmask, img, tiles_x, ntiles, stream);
    }
    // (with MrgsRasterInputs::bwd_grad_ws the ordering launch also prepares the backward: rows cleared, queues copied)
    if (!reuse_order)
        mrgs_launch_blend_order(img, g.counters + 16, ntiles, 0, in->bwd_grad_ws, in->bwd_grad_ws ? mrgs_grad_bytes(cfg->P, cfg->S) : 0, in->work_hint,
                                stream);
    t0.stop();
    STAGE_CHECK(cfg, stream);

    side_mark_pre_blend(stream);
    StageTimer t1(stream, ST_FWD);
    mrgs_launch_render_fwd(*cfg, *in, g, b.plist[cur], b.qmask, b.cflag, img, out_color, out_feature, out_others, stream);
    t1.stop();
    STAGE_CHECK(cfg, stream);
    return MRGS_OK;
}

static int zero_outputs(const MrgsRasterConfig* cfg, float* out_color, float* out_feature, float* out_others, hipStream_t stream)
{
    // the reference returns zero-filled outputs without touching the kernels (rasterize_points.cu:89-93,106)
    const size_t hw = (size_t)cfg->H * cfg->W;
    HIP_TRY(hipMemsetAsync(out_color, 0, sizeof(float) * 3 * hw, stream));
    if (cfg->S > 0) HIP_TRY(hipMemsetAsync(out_feature, 0, sizeof(float) * cfg->S * hw, stream));
    HIP_TRY(hipMemsetAsync(out_others, 0, sizeof(float) * MRGS_NUM_OTHERS * hw, stream));
    return MRGS_OK;
}

static int check_counters(const uint32_t host[2], int64_t* num_rendered_host)
{
    if (host[1] != 0) return MRGS_E_INTERNAL;
    if (host[0] >= (1u << 30)) return MRGS_E_UNSUPPORTED;   // pair counts are carried in 30 bits by the binning kernels
    *num_rendered_host = (int64_t)host[0];
    return MRGS_OK;
}

int mrgs_rasterize_forward_geom(const MrgsRasterConfig* cfg, const MrgsRasterInputs* in, void* geom_ws, size_t geom_bytes,
                                int32_t* radii, int64_t* num_rendered_host, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    int rc = check_cfg(cfg, in);
    if (rc) return rc;
    if (!num_rendered_host) return MRGS_E_BAD_ARG;
    *num_rendered_host = 0;
    if (cfg->P == 0) return MRGS_OK;
    if (!geom_ws || !radii) return MRGS_E_BAD_ARG;
    MrgsGeomWs g = mrgs_carve_geom(geom_ws, cfg->P, cfg->H, cfg->W);
    if (geom_bytes < g.total) return MRGS_E_WORKSPACE;
    rc = enqueue_geom(cfg, in, g, radii, stream);
    if (rc) return rc;
    // blocking read-back of num_rendered, as rasterizer_impl.cu:287 (plus the error flag of the look-back kernels)
    uint32_t host[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(host, g.counters, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return check_counters(host, num_rendered_host);
}

int mrgs_rasterize_forward_render(const MrgsRasterConfig* cfg, const MrgsRasterInputs* in, void* geom_ws, void* binning_ws,
                                  size_t binning_bytes, void* img_ws, int64_t R, float* out_color, float* out_feature,
                                  float* out_others, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    int rc = check_cfg(cfg, in);
    if (rc) return rc;
    if (!out_color || !out_others || (cfg->S > 0 && !out_feature) || !img_ws) return MRGS_E_BAD_ARG;
    if (cfg->P == 0) return zero_outputs(cfg, out_color, out_feature, out_others, stream);
    if (!geom_ws || !binning_ws) return MRGS_E_BAD_ARG;
    MrgsImgWs img = mrgs_carve_img(img_ws, cfg->H, cfg->W);
    img_use_hint(img, in, cfg->H, cfg->W);
    MrgsGeomWs g = mrgs_carve_geom(geom_ws, cfg->P, cfg->H, cfg->W);
    MrgsBinWs b = mrgs_carve_bin(binning_ws, R);
    if (binning_bytes < b.total) return MRGS_E_WORKSPACE;
    return enqueue_render(cfg, in, g, b, img, R, nullptr, out_color, out_feature, out_others, stream);
}

namespace {
// pinned, device-mapped landing slots + events for the num_rendered read-back of mrgs_rasterize_forward[_begin]: a ring per (host thread,
// device) -- the pinned words are mapped into the device that allocated them and the events belong to it.  A ticket names its slot and the
// ring's sequence number at its begin; MRGS_TICKET_RING later begins reuse the slot and the ticket is stale.
#define MRGS_MAX_DEVICES 64
#define MRGS_TICKET_RING 16
struct ReadbackRing {
    uint32_t* host = nullptr;   // MRGS_TICKET_RING x 16 words
    uint32_t* dev = nullptr;    // device address of the same pinned words
    hipEvent_t ev[MRGS_TICKET_RING] = {};
    uint64_t seq[MRGS_TICKET_RING] = {};
    uint64_t next = 1;
};
static thread_local ReadbackRing g_rings[MRGS_MAX_DEVICES];
}   // namespace

int mrgs_rasterize_forward_begin(const MrgsRasterConfig* cfg, const MrgsRasterInputs* in, void* geom_ws, size_t geom_bytes,
                                 void* binning_ws, size_t binning_bytes, int64_t capacity_pairs, void* img_ws, int32_t* radii,
                                 float* out_color, float* out_feature, float* out_others, MrgsRasterTicket* ticket, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    int rc = check_cfg(cfg, in);
    if (rc) return rc;
    if (!ticket || !out_color || !out_others || (cfg->S > 0 && !out_feature) || !img_ws) return MRGS_E_BAD_ARG;
    ticket->device = -1; ticket->slot = -1; ticket->seq = 0; ticket->capacity_pairs = capacity_pairs;
    if (cfg->P == 0) return zero_outputs(cfg, out_color, out_feature, out_others, stream);     // (device < 0: the count is 0)
    if (!geom_ws || !radii || !binning_ws || capacity_pairs < 1 || capacity_pairs >= (1ll << 30)) return MRGS_E_BAD_ARG;
    MrgsGeomWs g = mrgs_carve_geom(geom_ws, cfg->P, cfg->H, cfg->W);
    if (geom_bytes < g.total) return MRGS_E_WORKSPACE;
    MrgsImgWs img = mrgs_carve_img(img_ws, cfg->H, cfg->W);
    img_use_hint(img, in, cfg->H, cfg->W);
    MrgsBinWs b = mrgs_carve_bin(binning_ws, capacity_pairs);
    if (binning_bytes < b.total) return MRGS_E_WORKSPACE;
    int device = 0;
    HIP_TRY(hipGetDevice(&device));
    if (device < 0 || device >= MRGS_MAX_DEVICES) return MRGS_E_UNSUPPORTED;
    ReadbackRing& ring = g_rings[device];
    if (!ring.host) {
        HIP_TRY(hipHostMalloc((void**)&ring.host, MRGS_TICKET_RING * 64, hipHostMallocMapped | hipHostMallocPortable));
        HIP_TRY(hipHostGetDevicePointer((void**)&ring.dev, ring.host, 0));
        for (int i = 0; i < MRGS_TICKET_RING; ++i) HIP_TRY(hipEventCreateWithFlags(&ring.ev[i], hipEventDisableTiming));
    }
    const int slot = (int)(ring.next % MRGS_TICKET_RING);
    ring.seq[slot] = ring.next;
    ticket->device = device; ticket->slot = slot; ticket->seq = ring.next++;
    uint32_t* slot_dev = ring.dev + 16 * slot;
    rc = enqueue_geom(cfg, in, g, radii, stream, slot_dev, ring.ev[slot]);
    if (rc) return rc;
    // phase 2 is queued without waiting for the count: its kernels read it from g.counters; the count (and the error flag of phase 1)
    // reaches the pinned host slot through a kernel -- the tile scan, or on the radix path the first kernel of phase 2 -- with no
    // copy-engine transfer in the middle of the stream
    return enqueue_render(cfg, in, g, b, img, capacity_pairs, g.counters, out_color, out_feature, out_others, stream, slot_dev, ring.ev[slot]);
}

int mrgs_rasterize_forward_finish(const MrgsRasterTicket* ticket, int64_t* num_rendered_host)
{
    if (!ticket || !num_rendered_host) return MRGS_E_BAD_ARG;
    *num_rendered_host = 0;
    if (ticket->device < 0) return MRGS_OK;                          // empty model: nothing was queued behind a count
    if (ticket->device >= MRGS_MAX_DEVICES || ticket->slot < 0 || ticket->slot >= MRGS_TICKET_RING) return MRGS_E_BAD_ARG;
    ReadbackRing& ring = g_rings[ticket->device];
    if (!ring.host || ring.seq[ticket->slot] != ticket->seq) return MRGS_E_BAD_ARG;      // another thread's ticket, or the slot was reused
    HIP_TRY(hipEventSynchronize(ring.ev[ticket->slot]));
    int rc = check_counters(ring.host + 16 * ticket->slot, num_rendered_host);
    if (rc) return rc;
    return *num_rendered_host > ticket->capacity_pairs ? MRGS_E_WORKSPACE : MRGS_OK;
}

int mrgs_rasterize_forward(const MrgsRasterConfig* cfg, const MrgsRasterInputs* in, void* geom_ws, size_t geom_bytes,
                           void* binning_ws, size_t binning_bytes, int64_t capacity_pairs, void* img_ws, int32_t* radii,
                           float* out_color, float* out_feature, float* out_others, int64_t* num_rendered_host, void* stream_)
{
    if (!num_rendered_host) return MRGS_E_BAD_ARG;
    *num_rendered_host = 0;
    MrgsRasterTicket ticket;
    int rc = mrgs_rasterize_forward_begin(cfg, in, geom_ws, geom_bytes, binning_ws, binning_bytes, capacity_pairs, img_ws, radii, out_color,
                                          out_feature, out_others, &ticket, stream_);
    if (rc) return rc;
    return mrgs_rasterize_forward_finish(&ticket, num_rendered_host);
}

// The backward in two halves: the blend backward (+ optionally the clamp-masked colour gradients of the visible surfels, which are final
// once it has run) and the per-gaussian backward.  A view-parallel caller puts its all-gather of those colour gradients on the wire
// between the two (materialrefgs_amd/dist.py), where it overlaps the second half.
// the gradient tensors a backward needs for this configuration (MrgsRasterGrads; with the glue epilogue the five tensors it replaces may be
// NULL): MRGS_OK, MRGS_E_BAD_ARG (a tensor missing) or MRGS_E_UNSUPPORTED (the epilogue asked for a render it does not serve)
static int grads_check(const MrgsRasterConfig* cfg, const MrgsRasterInputs* in, const MrgsRasterGrads* g)
{
    if (!g->dL_dmeans2D || (cfg->M > 0 && !g->dL_dsh) || ((in->shs_rest != nullptr) != (g->dL_dsh_rest != nullptr))) return MRGS_E_BAD_ARG;
    if ((g->glue_params != nullptr) != (g->glue_grads != nullptr)) return MRGS_E_BAD_ARG;
    if (g->glue_params == nullptr)
        return (g->dL_dcolors && g->dL_dopacity && g->dL_dmeans3D && g->dL_dtransMat && g->dL_dscales && g->dL_drotations && (cfg->S == 0 || g->dL_dfeatures))
                   ? MRGS_OK : MRGS_E_BAD_ARG;
    const MrgsSurfelParams* p = g->glue_params;
    const MrgsSurfelGrads* o = g->glue_grads;
    // rows of eight channels, or the "pgsr" rows (nine channels in twelve floats, features_live = 9) with the glue's viewmatrix
    const bool rows8 = cfg->S == 8 && p->viewmatrix == nullptr, rows12 = cfg->S == 12 && in->features_live == 9u && p->viewmatrix != nullptr;
    if (!(rows8 || rows12) || !in->scales || !in->rotations || in->transMat_precomp) return MRGS_E_UNSUPPORTED;
    if (cfg->P != p->P || !p->scaling_raw || !p->rotation_raw || !p->opacity_raw || !p->refl_raw || !p->rough_raw || !p->ori_color_raw) return MRGS_E_BAD_ARG;
    return (o->d_xyz && o->d_scaling && o->d_rotation && o->d_opacity && o->d_refl && o->d_rough && o->d_ori_color && o->d_indirect_dc && o->d_indirect_rest)
               ? MRGS_OK : MRGS_E_BAD_ARG;
}

int mrgs_rasterize_backward_blend(const MrgsRasterConfig* cfg, const MrgsRasterInputs* in, const int32_t* radii, const void* geom_ws,
                                  const void* binning_ws, const void* img_ws, int64_t R, const float* dL_dout_color,
                                  const float* dL_dout_feature, const float* dL_dout_others, void* grad_ws, float* dL_dRGB_masked,
                                  void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    int rc = check_cfg(cfg, in);
    if (rc) return rc;
    if (cfg->P == 0) return MRGS_OK;
    if (!radii || !geom_ws || !binning_ws || !img_ws || !grad_ws || !dL_dout_color || !dL_dout_others) return MRGS_E_BAD_ARG;
    if (cfg->S > 0 && !dL_dout_feature) return MRGS_E_BAD_ARG;
    const int tiles_x = (cfg->W + MRGS_BLOCK_X - 1) / MRGS_BLOCK_X, tiles_y = (cfg->H + MRGS_BLOCK_Y - 1) / MRGS_BLOCK_Y;
    MrgsGeomWs g = mrgs_carve_geom(const_cast<void*>(geom_ws), cfg->P, cfg->H, cfg->W);
    MrgsBinWs b = mrgs_carve_bin(const_cast<void*>(binning_ws), R);
    MrgsImgWs img = mrgs_carve_img(const_cast<void*>(img_ws), cfg->H, cfg->W);
    img_use_hint(img, in, cfg->H, cfg->W);
    const int cur = plist_buf(g, tiles_x * tiles_y);
    float* grad_rec = (float*)grad_ws;

    StageTimer t0(stream, ST_BWD);
    const bool prepared = in->bwd_grad_ws != nullptr && in->bwd_grad_ws == grad_ws;   // the forward cleared the rows and set up the queues
    if (R > 0) {
        // (the gradient rows are cleared by spare workgroups of the ordering launch; mrgs_grad_bytes is a multiple of 256)
        if (!prepared) mrgs_launch_blend_order(img, g.counters + 16, tiles_x * tiles_y, 1, grad_rec, mrgs_grad_bytes(cfg->P, cfg->S), nullptr, stream);
        mrgs_launch_render_bwd(*cfg, *in, g, b.plist[cur], b.cflag, img, dL_dout_color, dL_dout_feature, dL_dout_others, grad_rec, prepared, stream);
    } else if (!prepared) {
        HIP_TRY(hipMemsetAsync(grad_rec, 0, mrgs_grad_bytes(cfg->P, cfg->S), stream));
    }
    t0.stop();
    if (dL_dRGB_masked) mrgs_launch_color_grad_extract(*cfg, g, radii, grad_rec, in->shs != nullptr, dL_dRGB_masked, stream);
    STAGE_CHECK(cfg, stream);
    return MRGS_OK;
}

int mrgs_rasterize_backward_finish(const MrgsRasterConfig* cfg, const MrgsRasterInputs* in, const int32_t* radii, const void* geom_ws,
                                   const void* grad_ws, const MrgsRasterGrads* grads, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    int rc = check_cfg(cfg, in);
    if (rc) return rc;
    if (!grads || grads->struct_size != sizeof(MrgsRasterGrads)) return MRGS_E_BAD_ARG;
    if (cfg->P == 0) return MRGS_OK;   // every gradient tensor has zero elements
    rc = grads_check(cfg, in, grads);
    if (rc) return rc;
    if (!radii || !geom_ws || !grad_ws) return MRGS_E_BAD_ARG;
    MrgsGeomWs g = mrgs_carve_geom(const_cast<void*>(geom_ws), cfg->P, cfg->H, cfg->W);
    StageTimer t1(stream, ST_PREB);
    mrgs_launch_preprocess_bwd(*cfg, *in, g, radii, (const float*)grad_ws, *grads, stream);
    t1.stop();
    STAGE_CHECK(cfg, stream);
    return MRGS_OK;
}

int mrgs_rasterize_backward(const MrgsRasterConfig* cfg, const MrgsRasterInputs* in, const int32_t* radii, const void* geom_ws,
                            const void* binning_ws, const void* img_ws, int64_t R, const float* dL_dout_color,
                            const float* dL_dout_feature, const float* dL_dout_others, void* grad_ws, const MrgsRasterGrads* grads,
                            void* stream_)
{
    // (argument checks of both halves before anything is queued)
    int rc = check_cfg(cfg, in);
    if (rc) return rc;
    if (!grads || grads->struct_size != sizeof(MrgsRasterGrads)) return MRGS_E_BAD_ARG;
    if (cfg->P == 0) return MRGS_OK;
    rc = grads_check(cfg, in, grads);
    if (rc) return rc;
    if (!radii || !geom_ws || !binning_ws || !img_ws || !grad_ws || !dL_dout_color || !dL_dout_others) return MRGS_E_BAD_ARG;
    if (cfg->S > 0 && !dL_dout_feature) return MRGS_E_BAD_ARG;
    rc = mrgs_rasterize_backward_blend(cfg, in, radii, geom_ws, binning_ws, img_ws, R, dL_dout_color, dL_dout_feature, dL_dout_others, grad_ws,
                                       nullptr, stream_);
    if (rc) return rc;
    return mrgs_rasterize_backward_finish(cfg, in, radii, geom_ws, grad_ws, grads, stream_);
}

int mrgs_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present,
                      void* stream_)
{
    (void)projmatrix;
    if (P < 0) return MRGS_E_BAD_ARG;
    if (P == 0) return MRGS_OK;
    if (!means3D || !viewmatrix || !present) return MRGS_E_BAD_ARG;
    mrgs_launch_mark_visible(P, means3D, viewmatrix, present, (hipStream_t)stream_);
    HIP_TRY(hipGetLastError());
    return MRGS_OK;
}

// ---- introspection for the parity tests --------------------------------------------------------------
__global__ void export_rec_kernel(int P, int which, const float4* __restrict__ rec, const uint8_t* __restrict__ clamped, void* dst)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const float4* r = rec + (size_t)i * MRGS_REC_F4;
    const float4 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3], r4 = r[4];
    float* f = (float*)dst;
    switch (which) {
    case 0: f[i] = r4.z; break;
    case 1: f[2 * i] = r2.y; f[2 * i + 1] = r2.z; break;
    case 2: {
        float* t = f + 9 * (size_t)i;
        t[0] = r0.x; t[1] = r0.y; t[2] = r0.z; t[3] = r0.w; t[4] = r1.x; t[5] = r1.y; t[6] = r1.z; t[7] = r1.w; t[8] = r2.x;
    } break;
    case 3: f[4 * i] = r3.x; f[4 * i + 1] = r3.y; f[4 * i + 2] = r3.z; f[4 * i + 3] = r2.w; break;
    case 4: f[3 * i] = r3.w; f[3 * i + 1] = r4.x; f[3 * i + 2] = r4.y; break;
    case 6: {
        uint8_t* c = (uint8_t*)dst;
        const uint32_t cl = clamped[i];
        c[3 * i] = cl & 1; c[3 * i + 1] = (cl >> 1) & 1; c[3 * i + 2] = (cl >> 2) & 1;
    } break;
    default: break;
    }
}

int mrgs_debug_export(const MrgsRasterConfig* cfg, const void* geom_ws, const void* binning_ws, const void* img_ws, int64_t R,
                      int32_t which, void* dst, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!cfg || !dst || cfg->struct_size != sizeof(MrgsRasterConfig)) return MRGS_E_BAD_ARG;
    const int P = cfg->P;
    const int tiles_x = (cfg->W + MRGS_BLOCK_X - 1) / MRGS_BLOCK_X, tiles_y = (cfg->H + MRGS_BLOCK_Y - 1) / MRGS_BLOCK_Y;
    const size_t hw = (size_t)cfg->H * cfg->W;
    MrgsGeomWs g = mrgs_carve_geom(const_cast<void*>(geom_ws), P, cfg->H, cfg->W);
    MrgsImgWs img = mrgs_carve_img(const_cast<void*>(img_ws), cfg->H, cfg->W);
    switch (which) {
    case 0: case 1: case 2: case 3: case 4: case 6:
        if (P > 0) hipLaunchKernelGGL(export_rec_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, which, g.rec, g.clamped, dst);
        break;
    case 5: HIP_TRY(hipMemcpyAsync(dst, g.tiles_touched, sizeof(uint32_t) * P, hipMemcpyDeviceToDevice, stream)); break;
    case 7: {
        MrgsBinWs b = mrgs_carve_bin(const_cast<void*>(binning_ws), R);
        const int cur = plist_buf(g, tiles_x * tiles_y);
        if (R > 0) HIP_TRY(hipMemcpyAsync(dst, b.plist[cur], sizeof(uint32_t) * R, hipMemcpyDeviceToDevice, stream));
    } break;
    case 8: HIP_TRY(hipMemcpyAsync(dst, img.ranges, sizeof(uint2) * tiles_x * tiles_y, hipMemcpyDeviceToDevice, stream)); break;
    case 9: HIP_TRY(hipMemcpyAsync(dst, img.final_T, sizeof(float) * 3 * hw, hipMemcpyDeviceToDevice, stream)); break;
    case 10: HIP_TRY(hipMemcpyAsync(dst, img.n_contrib, sizeof(uint32_t) * 2 * hw, hipMemcpyDeviceToDevice, stream)); break;
    case 11: HIP_TRY(hipMemcpyAsync(dst, g.order[sorted_buf(32)], sizeof(uint32_t) * P, hipMemcpyDeviceToDevice, stream)); break;
    case 12: HIP_TRY(hipMemcpyAsync(dst, img.redo_list, sizeof(uint32_t) * (2 + hw), hipMemcpyDeviceToDevice, stream)); break;   // [0] count, [2..] marked pixels
    case 13: {      // the quadrant masks of the tile lists (one byte per list entry: which 8x8 blocks of its tile the surfel's box touches)
        MrgsBinWs b = mrgs_carve_bin(const_cast<void*>(binning_ws), R);
        if (R > 0) HIP_TRY(hipMemcpyAsync(dst, b.qmask, sizeof(uint8_t) * R, hipMemcpyDeviceToDevice, stream));
    } break;
    case 14: if (P > 0) HIP_TRY(hipMemcpyAsync(dst, g.cull, sizeof(float4) * MRGS_CULL_F4 * (size_t)P, hipMemcpyDeviceToDevice, stream)); break;
    default: return MRGS_E_BAD_ARG;
    }
    HIP_TRY(hipGetLastError());
    return MRGS_OK;
}

int mrgs_set_profiling(int32_t enabled)
{
    std::lock_guard<std::mutex> lk(g_prof_mutex);
    g_profiling = enabled;
    for (auto& p : g_pairs) g_free.push_back(p);   // reset the statistics
    g_pairs.clear();
    // The event pairs a measurement will use exist before it starts: creating timing events in the middle of a stream of launches was
    // measured to stall single steps of bench.py by 2-4 ms in a third of its 20-step runs.
    if (enabled > 0) {
        while (g_free.size() < 64) {
            EvPair p;
            p.stage = 0;
            if (hipEventCreate(&p.a) != hipSuccess) break;
            if (hipEventCreate(&p.b) != hipSuccess) { (void)hipEventDestroy(p.a); break; }
            g_free.push_back(p);
        }
    }
    return MRGS_OK;
}
int mrgs_get_kernel_times(MrgsKernelTimes* out)
{
    if (!out) return MRGS_E_BAD_ARG;
    double sum[ST_COUNT] = {0};
    int cnt[ST_COUNT] = {0};
    std::lock_guard<std::mutex> lk(g_prof_mutex);
    for (auto& p : g_pairs) {
        float ms = 0;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { sum[p.stage] += ms; cnt[p.stage]++; }
    }
    float* dst[ST_COUNT] = {&out->preprocess_ms, &out->sort_ms, &out->duplicate_ms, &out->render_fwd_ms, &out->render_bwd_ms,
                            &out->preprocess_bwd_ms};
    for (int i = 0; i < ST_COUNT; i++) *dst[i] = cnt[i] ? (float)(sum[i] / cnt[i]) : 0.0f;
    return MRGS_OK;
}

// ---- a second stream next to the caller's (mrgs.h: mrgs_side_stream_fork / _join) ---------------------------------------------------
// One non-blocking side stream per device (shared by the host threads: a backward may fork on autograd's worker thread and be joined from
// the thread that called backward()) and a ring of events without timing, created on first use and kept.  fork: the side stream waits
// for what the caller's stream holds NOW; join: the caller's stream waits for what the side stream holds now.
namespace {
#define MRGS_SIDE_EVENTS 32
struct SideStream {
    hipStream_t stream = nullptr;
    hipEvent_t ev[MRGS_SIDE_EVENTS] = {};
    unsigned next = 0;
    hipEvent_t pre_blend = nullptr;     // recorded on the rasterizer's stream right before its forward blend (once somebody has asked for it)
    // The mark is ONE-SHOT: armed by mrgs_side_stream_arm_blend_mark (whoever owes side work says so BEFORE the forward it wants to fork
    // from), recorded by the next forward on `armed_on`, consumed by the first fork_at_blend, dropped by a join.  A mark of an earlier
    // iteration can therefore never order side work (it would not cover what the caller's stream did since, e.g. an optimizer step).
    bool want_pre_blend = false, have_pre_blend = false;
    hipStream_t armed_on = nullptr, marked_on = nullptr;
};
SideStream g_side[MRGS_MAX_DEVICES];
std::mutex g_side_mutex;
static int side_of(SideStream** out)
{
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= MRGS_MAX_DEVICES) return MRGS_E_UNSUPPORTED;
    SideStream& s = g_side[dev];
    if (s.stream == nullptr) {
        // LOWEST priority: the side work is filler -- the dispatcher hands a free wave slot to the caller's stream first, and what runs
        // on the side stream takes what is left (the lone-wave tails of the blend kernels); MRGS_SIDE_PRIORITY=default: same priority
        int least = 0, greatest = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
        const char* pr = getenv("MRGS_SIDE_PRIORITY");
        const bool dflt = pr != nullptr && pr[0] == 'd';
        if (dflt) HIP_TRY(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
        else HIP_TRY(hipStreamCreateWithPriority(&s.stream, hipStreamNonBlocking, least));
        for (int i = 0; i < MRGS_SIDE_EVENTS; i++) HIP_TRY(hipEventCreateWithFlags(&s.ev[i], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&s.pre_blend, hipEventDisableTiming));
    }
    *out = &s;
    return MRGS_OK;
}
}   // namespace

static void side_mark_pre_blend(hipStream_t stream)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MRGS_MAX_DEVICES) return;
    std::lock_guard<std::mutex> lk(g_side_mutex);
    SideStream& s = g_side[dev];
    if (!s.want_pre_blend || s.pre_blend == nullptr || stream != s.armed_on) return;
    s.have_pre_blend = hipEventRecord(s.pre_blend, stream) == hipSuccess;
    s.marked_on = stream;
    s.want_pre_blend = false;                    // the first forward after the arming is the one meant
}

int mrgs_side_stream_arm_blend_mark(void* main_stream)
{
    std::lock_guard<std::mutex> lk(g_side_mutex);
    SideStream* s = nullptr;
    int rc = side_of(&s);
    if (rc) return rc;
    s->want_pre_blend = true;
    s->have_pre_blend = false;                   // whatever an earlier forward marked is history
    s->armed_on = (hipStream_t)main_stream;
    return MRGS_OK;
}

int mrgs_side_stream_fork_at_blend(void* main_stream, void** side_stream)
{
    if (!side_stream) return MRGS_E_BAD_ARG;
    std::lock_guard<std::mutex> lk(g_side_mutex);
    SideStream* s = nullptr;
    int rc = side_of(&s);
    if (rc) return rc;
    if (s->have_pre_blend && s->marked_on == (hipStream_t)main_stream) {
        s->have_pre_blend = false;               // consumed: a second fork without a new arming + forward takes the plain path
        HIP_TRY(hipStreamWaitEvent(s->stream, s->pre_blend, 0));
    } else {                                     // no forward on this stream has marked the point since the arming: everything the caller's stream holds
        s->have_pre_blend = false;
        hipEvent_t e = s->ev[s->next++ % MRGS_SIDE_EVENTS];
        HIP_TRY(hipEventRecord(e, (hipStream_t)main_stream));
        HIP_TRY(hipStreamWaitEvent(s->stream, e, 0));
    }
    *side_stream = (void*)s->stream;
    return MRGS_OK;
}

int mrgs_side_stream_fork(void* main_stream, void** side_stream)
{
    if (!side_stream) return MRGS_E_BAD_ARG;
    std::lock_guard<std::mutex> lk(g_side_mutex);
    SideStream* s = nullptr;
    int rc = side_of(&s);
    if (rc) return rc;
    hipEvent_t e = s->ev[s->next++ % MRGS_SIDE_EVENTS];
    HIP_TRY(hipEventRecord(e, (hipStream_t)main_stream));
    HIP_TRY(hipStreamWaitEvent(s->stream, e, 0));
    *side_stream = (void*)s->stream;
    return MRGS_OK;
}

int mrgs_side_stream_join(void* main_stream)
{
    std::lock_guard<std::mutex> lk(g_side_mutex);
    SideStream* s = nullptr;
    int rc = side_of(&s);
    if (rc) return rc;
    hipEvent_t e = s->ev[s->next++ % MRGS_SIDE_EVENTS];
    HIP_TRY(hipEventRecord(e, s->stream));
    HIP_TRY(hipStreamWaitEvent((hipStream_t)main_stream, e, 0));
    s->have_pre_blend = false;                   // a mark older than a join is stale
    return MRGS_OK;
}

const char* mrgs_strerror(int code)
{
    switch (code) {
    case MRGS_OK: return "ok";
    case MRGS_E_BAD_ARG: return "bad argument (shape / pointer contract violated)";
    case MRGS_E_TOO_MANY_FEATURES: return "more than MRGS_MAX_FEATURES (24) feature channels";
    case MRGS_E_NEED_COLORS: return "provide exactly one of SHs or precomputed colours";
    case MRGS_E_HIP: return "HIP runtime error (see mrgs_last_hip_error)";
    case MRGS_E_WORKSPACE: return "workspace too small";
    case MRGS_E_UNSUPPORTED: return "unsupported configuration";
    case MRGS_E_INTERNAL: return "device-side wait overran in the binning kernels";
    default: return "unknown error";
    }
}
const char* mrgs_last_hip_error(void) { return g_hip_err; }
const char* mrgs_version(void) { return "mrgs 0.3.0 (gfx950)"; }
int32_t mrgs_abi_version(void) { return MRGS_ABI_VERSION; }

}   // extern "C"
