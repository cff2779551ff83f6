/*
 * mrgs.h -- C ABI of libmrgs.so, the MI355X (gfx950) surfel rasterizer + BRDF shading library.
 *
 * Drop-in boundary for the reference's rasterizer extension `diff_surfel_rasterization._C`
 * (pybind: /root/reference/submodules/diff-surfel-rasterization/ext.cpp:15-19).  Every entry point below
 * names the reference interface it replaces.  Plain pointers and sizes only -- no torch types; every
 * pointer is a DEVICE pointer unless its name ends in _host.  The library never allocates or frees device
 * memory: the caller owns the outputs and the three opaque workspaces (sizes from mrgs_*_bytes), exactly as
 * the reference's geomBuffer / binningBuffer / imgBuffer tensors are owned by Python
 * (rasterize_points.cu:95-103, diff_surfel_rasterization/__init__.py:101).  ONE exception, host memory: the forward calls that
 * report the pair count without a blocking copy (mrgs_rasterize_forward, _begin / _finish) read it from a pinned, device-mapped
 * slot; the library allocates a ring of MRGS_TICKET_RING (16) such 64-byte slots per (host thread, device) with hipHostMalloc on
 * first use and keeps it for the life of the process.  Consequence: a ticket of mrgs_rasterize_forward_begin is finished on the
 * SAME host thread that began it (the ring is thread-local; another thread's _finish returns MRGS_E_BAD_ARG) -- a host that
 * finishes renders on a worker thread begins them there too, or uses the _geom / _render pair.  All work is enqueued on the
 * caller's hipStream_t (pass torch.cuda.current_stream().cuda_stream); calls on distinct streams/devices are
 * independent.  Return value: 0 on success, otherwise an MRGS_E_* code (mrgs_strerror gives the text);
 * nothing throws across the ABI.
 */
#ifndef MRGS_H_INCLUDED
#define MRGS_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MRGS_MAX_FEATURES 24 /* cuda_rasterizer/config.h:17 */
#define MRGS_TILE 16         /* cuda_rasterizer/config.h:19-20 */
#define MRGS_NUM_OTHERS 7    /* auxiliary.h:25-29: depth, alpha, normal xyz, median depth, distortion */

enum {
    MRGS_OK = 0,
    MRGS_E_BAD_ARG = 1,       /* shape / pointer contract violated (AT_ERROR in rasterize_points.cu:64-66) */
    MRGS_E_TOO_MANY_FEATURES = 2,
    MRGS_E_NEED_COLORS = 3,   /* neither SH nor precomputed colours (rasterizer_impl.cu:248-251) */
    MRGS_E_HIP = 4,           /* a HIP runtime call failed; see mrgs_last_hip_error() */
    MRGS_E_WORKSPACE = 5,     /* workspace too small for this call */
    MRGS_E_UNSUPPORTED = 6,
    MRGS_E_INTERNAL = 7       /* a bounded device-side wait overran (binning look-back); results were not written */
};

/* Scalar arguments shared by forward and backward; mirrors GaussianRasterizationSettings
 * (diff_surfel_rasterization/__init__.py:167-179) plus the tensor extents the pybind layer derives
 * (rasterize_points.cu:69-72,116-121). */
typedef struct MrgsRasterConfig {
    uint32_t struct_size; /* = sizeof(MrgsRasterConfig) of the header the caller was built against; any other value -> MRGS_E_BAD_ARG
                             (a caller built against another revision of this header is refused instead of being misread) */
    int32_t P;            /* number of gaussians (means3D.size(0)) */
    int32_t S;            /* extra feature channels (features.size(1)), 0..MRGS_MAX_FEATURES */
    int32_t D;            /* active SH degree (raster_settings.sh_degree) */
    int32_t M;            /* SH coefficients per gaussian (sh.size(1)), 0 when colours are precomputed */
    int32_t H, W;         /* image_height, image_width */
    float tanfovx, tanfovy;
    float scale_modifier;
    int32_t prefiltered;
    int32_t debug;        /* nonzero: synchronise the stream and report HIP errors after every kernel (CHECK_CUDA, auxiliary.h:303-310) */
} MrgsRasterConfig;

/* Per-gaussian inputs (device pointers, contiguous fp32, the GaussianModel getter layout):
 * means3D[P,3], shs[P,M,3] or NULL, colors_precomp[P,3] or NULL, features[P,S] (may be NULL when S==0),
 * opacities[P], scales[P,2]+rotations[P,4] (w,x,y,z) or both NULL with transMat_precomp[P,9] given.
 * Camera: viewmatrix[16], projmatrix[16] (row-vector convention tensors, read as stored), campos[3], bg[3]. */
typedef struct MrgsRasterInputs {
    uint64_t struct_size;  /* = sizeof(MrgsRasterInputs); checked like MrgsRasterConfig::struct_size.  The struct has grown optional
                              trailing fields (work_hint, shs_rest, bwd_grad_ws, hint_flags) which the calls ACT on: a shorter struct from
                              an older header would make the library read them past its end */
    const float* bg;
    const float* means3D;
    const float* shs;
    const float* colors_precomp;
    const float* features;
    const float* opacities;
    const float* scales;
    const float* rotations;
    const float* transMat_precomp;
    const float* viewmatrix;
    const float* projmatrix;
    const float* campos;
    uint32_t* work_hint;   /* optional, in/out, device, mrgs_work_hint_bytes(H, W), zero-initialised by the caller once per camera: the
                              forward orders its blend waves by the work each (tile, quadrant) took the last time this buffer was
                              passed (falling back to the cull count where it holds 0) and stores the work of this call.  A training
                              loop revisits the same cameras, so the previous visit predicts where rays terminate early far better than
                              any count available before the blend.  NULL: cull counts only.  Results do not depend on its CONTENTS.
                              The buffer also carries the blend kernels' queue state, tickets and the forward's dealt queues
                              (MrgsHintLayout), so (a) renders that share one buffer -- the same camera -- are issued on ONE stream, one
                              after the other (two forwards of one camera on two streams would hand every item out once ACROSS both
                              kernels; renders with different buffers, or without one, are independent on any streams), and (b) a
                              backward that names a prepared workspace (bwd_grad_ws below) is given the SAME buffer its forward was
                              given, not rewritten in between by anything but this library: a fresh or zeroed buffer there makes every
                              wave pull item 0.  The Python wrapper keeps the forward's tensor in the autograd node for that reason. */
    const float* shs_rest; /* optional: split SH layout.  When set, `shs` holds the DC coefficients [P,1,3] and shs_rest the higher orders
                              [P,M-1,3] (M = 2..16) -- the two tensors GaussianModel stores (_features_dc / _features_rest,
                              scene/gaussian_model.py:401-402), which the reference concatenates for every render (get_features,
                              :256-259); MrgsRasterGrads::dL_dsh_rest must then be set too (dL_dsh receives [P,1,3]). */
    void* bwd_grad_ws;     /* optional.  FORWARD calls: the grad_ws (mrgs_grad_bytes) the caller will hand to mrgs_rasterize_backward for
                              this render.  The forward then prepares the backward while it runs anyway -- the gradient rows are cleared by
                              spare workgroups of its ordering launch and the backward's work queues are set up as a copy of its own -- and the
                              backward needs no launch before its blend kernel.  BACKWARD call: pass the same pointer (and the same
                              grad_ws) to say that this was done; NULL, or a pointer other than grad_ws, makes the backward order and clear
                              by itself.  Valid for ONE backward per forward, with the forward's work_hint buffer (above).  Results do not
                              depend on it. */
    uint32_t hint_flags;   /* MRGS_HINT_REUSE_ORDER: the forward skips the ordering of its blend waves and deals them as the last
                              ordering of this camera did -- the dealt queues live in the work_hint buffer, behind the per-block work
                              (mrgs_work_hint_bytes covers both).  Only with work_hint set, and only after a forward WITHOUT the flag has
                              run on the same buffer; the caller (the Python wrapper: every visit of a camera but the first two and every
                              sixteenth) is responsible for that.  A schedule only: results do not depend on it. */
    uint32_t features_live; /* ABI 8 (was reserved, 0): 0 = every one of the S feature channels may be non-zero.  n in 1 .. S - 1: channels n .. S - 1
                               are PADDING (the caller keeps them zero -- e.g. rows of 9 channels padded to 12 floats so that they are 16-byte
                               pieces): the blend kernels leave them out of their per-entry arithmetic, their output maps are written as zeros,
                               their columns of dL_dfeatures stay zero and the upstream gradient of those maps is not read.  Honoured for
                               S = 12 with n = 9 and a 16-byte aligned tensor (the "pgsr" flavour's rows: the blend kernels stage eight channels
                               as for S = 8 and take the ninth from the spare float of the surfel record, where preprocess copies it); any
                               other combination is treated as 0 -- the results are the same, only slower. */
} MrgsRasterInputs;
#define MRGS_HINT_REUSE_ORDER 1u
#define MRGS_HINT_VISIBLE_BYTES 2u   /* ABI 8: `radii` of the forward calls points at P int32 FOLLOWED BY P bytes, and the forward also writes
                                        radii[i] > 0 into byte i of the tail (every render function of the reference returns that mask as
                                        "visibility_filter": a torch kernel per view otherwise) */

/* Workspace sizes.  geom <-> geomBuffer (GeometryState, rasterizer_impl.cu:157-172), img <-> imgBuffer
 * (ImageState, :174-181), binning <-> binningBuffer (BinningState, :183-196; sized from num_rendered). */
size_t mrgs_geom_bytes(int32_t P, int32_t H, int32_t W);
size_t mrgs_img_bytes(int32_t H, int32_t W);
size_t mrgs_binning_bytes(int64_t num_rendered);
size_t mrgs_work_hint_bytes(int32_t H, int32_t W);

/* Forward, phase 1 of 2.  Replaces the first half of CudaRasterizer::Rasterizer::forward
 * (rasterizer_impl.cu:200-291: preprocess, prefix sum, blocking read-back of num_rendered).
 * Writes radii[P] (int32), fills geom_ws, and returns the number of (tile, gaussian) pairs in
 * *num_rendered_host after synchronising `stream` (the reference does the same blocking 4-byte copy, :287). */
int mrgs_rasterize_forward_geom(const MrgsRasterConfig* cfg, const MrgsRasterInputs* in, void* geom_ws, size_t geom_bytes,
                                int32_t* radii, int64_t* num_rendered_host, void* stream);

/* Forward, phase 2 of 2.  Replaces rasterizer_impl.cu:293-348 (duplicateWithKeys, tile sort, tile ranges,
 * per-tile blend).  Outputs (caller-allocated; fully written, no pre-zeroing needed): out_color[3,H,W],
 * out_feature[S,H,W], out_others[7,H,W].  binning_ws must hold mrgs_binning_bytes(num_rendered). */
int mrgs_rasterize_forward_render(const MrgsRasterConfig* cfg, const MrgsRasterInputs* in, void* geom_ws, void* binning_ws,
                                  size_t binning_bytes, void* img_ws, int64_t num_rendered, float* out_color,
                                  float* out_feature, float* out_others, void* stream);

/* Both phases in one call, without a host round trip in the middle of the GPU work: the binning workspace is carved for
 * `capacity_pairs` (a guess, e.g. the previous call's count plus a margin; binning_bytes >= mrgs_binning_bytes(capacity_pairs)),
 * phase 2 is queued right behind phase 1 and takes the actual pair count from device memory, and the call returns once the
 * count has reached the host (the GPU keeps working).  Returns MRGS_OK with *num_rendered_host set when the count fitted.
 * Returns MRGS_E_WORKSPACE with *num_rendered_host set when it did not: the outputs are then those of an EMPTY render (every tile
 * list reads as empty: background colour, zero maps -- finite, so that work queued behind the call runs on defined data) and the
 * caller redoes phase 2 with mrgs_rasterize_forward_render on a workspace of mrgs_binning_bytes(*num_rendered_host).
 * mrgs_rasterize_backward must be given the pair count the binning workspace was carved for (capacity_pairs here).
 * Same reference interface as the two calls above (rasterize_points.cu:41-144). */
int mrgs_rasterize_forward(const MrgsRasterConfig* cfg, const MrgsRasterInputs* in, void* geom_ws, size_t geom_bytes,
                           void* binning_ws, size_t binning_bytes, int64_t capacity_pairs, void* img_ws, int32_t* radii,
                           float* out_color, float* out_feature, float* out_others, int64_t* num_rendered_host, void* stream);

/* The same call in two halves, for callers that have more work to queue behind the rasterizer before they need the count (a renderer
 * queues its per-pixel kernels; the host never sits between the tile scan and the blend): _begin queues both phases and returns at
 * once, _finish waits until the count has reached the host and reports it -- MRGS_OK, or MRGS_E_WORKSPACE exactly as above (everything
 * computed from the outputs is then undefined as well and the caller redoes the view).  A ticket is valid on the host thread that
 * began it until MRGS_TICKET_RING (16) later begins on the same device; a stale one returns MRGS_E_BAD_ARG. */
typedef struct MrgsRasterTicket {
    int32_t device, slot;
    uint64_t seq;
    int64_t capacity_pairs;
} MrgsRasterTicket;
int mrgs_rasterize_forward_begin(const MrgsRasterConfig* cfg, const MrgsRasterInputs* in, void* geom_ws, size_t geom_bytes,
                                 void* binning_ws, size_t binning_bytes, int64_t capacity_pairs, void* img_ws, int32_t* radii,
                                 float* out_color, float* out_feature, float* out_others, MrgsRasterTicket* ticket, void* stream);
int mrgs_rasterize_forward_finish(const MrgsRasterTicket* ticket, int64_t* num_rendered_host);

/* Gradient outputs of mrgs_rasterize_backward, the tuple returned by RasterizeGaussiansBackwardCUDA
 * (rasterize_points.cu:146-252): all fully written by the call (no pre-zeroing needed). */
typedef struct MrgsRasterGrads {
    uint64_t struct_size;  /* = sizeof(MrgsRasterGrads); checked like MrgsRasterConfig::struct_size */
    float* dL_dmeans2D;    /* [P,3]  (.xy = densification proxy, backward.cu:665-668; .z = 0) */
    float* dL_dcolors;     /* [P,3] */
    float* dL_dfeatures;   /* [P,S] */
    float* dL_dopacity;    /* [P,1] */
    float* dL_dmeans3D;    /* [P,3] */
    float* dL_dtransMat;   /* [P,9] */
    float* dL_dsh;         /* [P,M,3] */
    float* dL_dscales;     /* [P,2] */
    float* dL_drotations;  /* [P,4] */
    float* dL_dsh_rest;    /* [P,M-1,3] with the split SH layout (MrgsRasterInputs::shs_rest; dL_dsh is then [P,1,3]), else NULL */
    /* ABI 10 -- the glue epilogue: both NULL, or both set by render_surfel's caller (declared further down: the raw GaussianModel parameters
     * and their gradient tensors, the arguments of mrgs_surfel_features_backward).  Set: the per-gaussian backward applies the backward of the
     * per-gaussian glue (gaussian_renderer/__init__.py:338-355 and the GaussianModel getters: sigmoid / exp / normalize) to its results while
     * they are in registers and writes glue_grads' nine tensors (fully; d_indirect_dc / d_indirect_rest as zeros) INSTEAD of dL_dopacity,
     * dL_dscales, dL_drotations, dL_dfeatures and dL_dmeans3D, which are not touched and may be NULL; dL_dcolors and dL_dtransMat are written
     * if not NULL.  One pass over the P rows instead of two (mrgs_surfel_features_backward is not called for this render).
     * Contract: the rows mrgs_surfel_features_forward wrote for the same parameters (scales / rotations given, no precomputed transMat) --
     * S = 8 with glue_params->viewmatrix NULL, or the "pgsr" rows (S = 12, features_live = 9) with the viewmatrix they were built with: the
     * plane distance's gradient then goes to the raw rotation and the centre inside the epilogue (campos = MrgsRasterInputs::campos) --
     * and NO upstream gradient at feature channels 5..7 (nothing reads the blended indirect radiance: render_surfel without opt.indirect),
     * so that the mirror direction takes no gradient.  Other row shapes or a precomputed transMat: MRGS_E_UNSUPPORTED; the three channels'
     * zero gradient is the caller's word (it is not read).  glue_params' indirect_dc / indirect_rest / xyz / campos are not read. */
    const struct MrgsSurfelParams* glue_params;
    const struct MrgsSurfelGrads* glue_grads;
} MrgsRasterGrads;

/* Backward.  Replaces CudaRasterizer::Rasterizer::backward (rasterizer_impl.cu:353-462).  The three
 * workspaces must be the ones the forward of the same view filled.  grad_ws: scratch of
 * mrgs_grad_bytes(P,S) bytes for the packed per-gaussian accumulators. */
size_t mrgs_grad_bytes(int32_t P, int32_t S);
int mrgs_rasterize_backward(const MrgsRasterConfig* cfg, const MrgsRasterInputs* in, const int32_t* radii, const void* geom_ws,
                            const void* binning_ws, const void* img_ws, int64_t num_rendered, const float* dL_dout_color,
                            const float* dL_dout_feature, const float* dL_dout_others, void* grad_ws,
                            const MrgsRasterGrads* grads, void* stream);

/* The same backward in two halves (no counterpart in the reference, which is single-process): _blend runs the per-tile blend backward
 * (backward.cu:145-468) and, when dL_dRGB_masked [P,3] is given, writes the colour gradient of every surfel as the SH backward consumes
 * it (zero where the forward clamped the channel, backward.cu:33-36, and for culled surfels) -- final at that point; _finish runs the
 * per-gaussian backward (backward.cu:614-669) and fills `grads`.  A view-parallel step starts its all-gather of dL_dRGB_masked between
 * the two calls (materialrefgs_amd/dist.py: FactoredGradReducer.begin_early), where it overlaps the second half.
 * mrgs_rasterize_backward == _blend(..., NULL) followed by _finish. */
int mrgs_rasterize_backward_blend(const MrgsRasterConfig* cfg, const MrgsRasterInputs* in, const int32_t* radii, const void* geom_ws,
                                  const void* binning_ws, const void* img_ws, int64_t num_rendered, const float* dL_dout_color,
                                  const float* dL_dout_feature, const float* dL_dout_others, void* grad_ws, float* dL_dRGB_masked,
                                  void* stream);
int mrgs_rasterize_backward_finish(const MrgsRasterConfig* cfg, const MrgsRasterInputs* in, const int32_t* radii, const void* geom_ws,
                                   const void* grad_ws, const MrgsRasterGrads* grads, void* stream);

/* Replaces markVisible (rasterize_points.cu:254-273, rasterizer_impl.cu:56-68,143-155). present: uint8[P]. */
int mrgs_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present,
                      void* stream);

/* ---- environment lookup and deferred specular shading ("second boundary") ----------------------
 * The mip chain of scene/light.py:EnvLight: level i is a [6,res_i,res_i,3] fp32 cubemap holding PRE-sigmoid texels
 * (light.py:129); face/orientation convention of cube_to_dir (scene/light_utils.py:24-31).  grad[i] (may be NULL) receives
 * dL/dtexel by atomic accumulation into grad_copies[i] privatised copies -- the caller zero-fills them and sums them. */
#define MRGS_MAX_MIPS 8
typedef struct MrgsEnvMips {
    int32_t n_levels;
    int32_t res[MRGS_MAX_MIPS];
    const float* tex[MRGS_MAX_MIPS];
    float* grad[MRGS_MAX_MIPS];
    int32_t grad_copies[MRGS_MAX_MIPS];   /* >= 1: grad[i] holds this many consecutive copies of the level; workgroups spread
                                             their atomics over the copies (thousands of pixels mirror into each texel of the
                                             coarse levels, and same-address atomics serialise in L2); the caller sums them */
    float min_roughness, max_roughness;   /* EnvLight.min_roughness / max_roughness (0.08 / 0.5) */
} MrgsEnvMips;

/* Replaces EnvLight.__call__(l, roughness=...) (scene/light.py:99-129): out[N,3] = sigmoid(trilinear cube fetch at the mip
 * level get_mip(roughness)).  roughness == NULL selects mode="pure_env" (level 0 only). */
int mrgs_envmap_lookup_forward(const MrgsEnvMips* mips, int64_t N, const float* dirs, const float* roughness, float* out, void* stream);
int mrgs_envmap_lookup_backward(const MrgsEnvMips* mips, int64_t N, const float* dirs, const float* roughness, const float* g_out,
                                float* g_dirs /*[N,3] or NULL*/, float* g_roughness /*[N] or NULL*/, void* stream);

typedef struct MrgsStridedMap { const float* ptr; int64_t stride_h, stride_w, stride_c; } MrgsStridedMap;   /* element strides */
typedef struct MrgsShadeFrame {
    int32_t H, W;
    float Kinv[9];          /* inverse intrinsics, row-major, host values (np.linalg.inv(K), utils/refl_utils.py:64) */
    const float* R;         /* device [3,3]: Camera.R (stored transposed, i.e. camera-to-world rotation) */
    const float* T;         /* device [3]:   Camera.T (world-to-camera translation) */
    MrgsStridedMap albedo;  /* [H,W,3] */
    MrgsStridedMap normal;  /* [H,W,3] world-space shading normal (rend_normal / alpha) */
    MrgsStridedMap alpha;   /* [H,W,1] */
    MrgsStridedMap refl;    /* [H,W,1] refl_strength */
    MrgsStridedMap roughness; /* [H,W,1] */
    const float* lut;       /* device [lut_res, lut_res, 2] split-sum FG table (assets/bsdf_256_256.bin layout, refl_utils.py:9) */
    int32_t lut_res;
} MrgsShadeFrame;

/* Replaces get_specular_color_surfel without visibility tracing (utils/refl_utils.py:364-419): per pixel, in one pass,
 * specular[3,H,W] = direct_light * alpha * specular_weight, direct_light[3,H,W], specular_weight[H,W,3]. */
int mrgs_shade_specular_forward(const MrgsEnvMips* mips, const MrgsShadeFrame* frame, float* specular, float* direct_light,
                                float* specular_weight, void* stream);
/* ... and render_surfel's compositing in the same pass (ABI 6; mrgs_surfel_composite_forward's arithmetic on the specular just computed):
 * diffuse[3,H,W] = (1 - refl) base_color, render[3,H,W] = [linear_to_srgb](diffuse + specular) + bg (1 - alpha).
 * zero_fill / zero_floats (ABI 9): a buffer this launch clears on the side (NULL / 0 = nothing) -- the texel-gradient buffers the
 * backward of this render will accumulate into (mrgs_surfel_shade_composite_backward expects them zeroed and has no launch in front
 * of it that could do it). */
int mrgs_shade_specular_forward_composite(const MrgsEnvMips* mips, const MrgsShadeFrame* frame, const float* base_color, const float* bg,
                                          int32_t srgb, float* specular, float* direct_light, float* specular_weight, float* render,
                                          float* diffuse, float* zero_fill, int64_t zero_floats, void* stream);
/* Gradients w.r.t. the five maps (dense, fully written: g_albedo[H,W,3], g_normal[H,W,3], g_alpha[H,W], g_refl[H,W],
 * g_roughness[H,W]) and, through mips->grad, w.r.t. the cubemap texels (ACCUMULATED into mips->grad: the caller clears them).  Any of
 * the three upstream gradients may be NULL.  A level that receives gradients must have fewer than 2^24 texels (res < 1673; the
 * reference's EnvLight uses 16 ... 128, at most 512): larger ones return MRGS_E_UNSUPPORTED. */
int mrgs_shade_specular_backward(const MrgsEnvMips* mips, const MrgsShadeFrame* frame, const float* g_specular, const float* g_direct_light,
                                 const float* g_specular_weight, float* g_albedo, float* g_normal, float* g_alpha, float* g_refl,
                                 float* g_roughness, void* stream);
/* The same backward as render_surfel needs it (ABI 6): the per-pixel gradients leave as the gradient of the rasterizer's [8,H,W]
 * feature map -- g_features = (g_refl + g_refl_composite, g_roughness, g_albedo[3], 0, 0, 0), channel-major, fully written -- and as
 * the TOTAL alpha gradient g_alpha[H,W] = g_alpha_composite + this kernel's, with mrgs_surfel_composite_backward's g_refl / g_alpha
 * as the two inputs (what mrgs_surfel_feature_grads assembles from five maps in a launch of its own).  g_normal[H,W,3] as above. */
int mrgs_shade_specular_backward_features(const MrgsEnvMips* mips, const MrgsShadeFrame* frame, const float* g_specular,
                                          const float* g_direct_light, const float* g_specular_weight, const float* g_refl_composite,
                                          const float* g_alpha_composite, float* g_normal, float* g_features, float* g_alpha, void* stream);
/* render_surfel's compositing backward AND the shading backward in ONE launch (ABI 9; gaussian_renderer/__init__.py:436-445 then
 * utils/refl_utils.py:364-419 backwards): what mrgs_surfel_composite_backward followed by mrgs_shade_specular_backward_features computes, with
 * the compositing's g_specular / g_refl / g_alpha shares kept in the pixel's registers.  g_render / g_diffuse [3,H,W]: upstream gradients
 * of the two composited maps (either may be NULL); g_specular_extra [3,H,W]: an upstream gradient of the specular map itself (NULL =
 * none); base_color, specular [3,H,W]: the forward's inputs / output; bg[3].  Outputs, fully written: g_base[3,H,W], g_normal[H,W,3],
 * g_features[8,H,W], g_alpha[H,W] (total).  The texel gradients are accumulated into mips->grad, which must be zero on entry
 * (mrgs_shade_specular_forward_composite's zero_fill clears them in the forward of the same render). */
int mrgs_surfel_shade_composite_backward(const MrgsEnvMips* mips, const MrgsShadeFrame* frame, int32_t srgb, const float* base_color,
                                         const float* specular, const float* bg, const float* g_render, const float* g_diffuse,
                                         const float* g_specular_extra, const float* g_direct_light, const float* g_specular_weight,
                                         float* g_base, float* g_normal, float* g_features, float* g_alpha, void* stream);

/* ---- environment prefilter: EnvLight.build_mips (scene/light.py:72-86) ------------------------------------------------
 * renderutils' specular_cubemap / diffuse_cubemap (scene/renderutils/c_src/cubemap.cu:110-354, ops.py:390-459) are fixed linear
 * operators for a given (resolution, roughness, cos_cutoff): out[t] = sum_s w(t,s) cube[s] / sum_s w(t,s).  They are built
 * once as CSR matrices with normalised weights and applied every iteration as 3-channel SpMVs (forward with the matrix,
 * backward with its transpose, which the caller builds from the CSR).  kind: 0 specular GGX lobe, 1 diffuse cosine.
 * Texel index = (face * res + y) * res + x; cubemaps are [6,res,res,3] fp32.
 *   count: row_count[6 res^2] non-zeros per output texel, row_wsum[6 res^2] = sum of its un-normalised weights
 *   fill:  col / val for row_ptr = exclusive prefix sum of row_count (val = weight / row_wsum for kind 0, the plain weight for kind 1:
 *          diffuse_cubemap is not normalised) */
int mrgs_cubemap_filter_count(int32_t res, int32_t kind, float roughness, float cos_cutoff, uint32_t* row_count, float* row_wsum, void* stream);
int mrgs_cubemap_filter_fill(int32_t res, int32_t kind, float roughness, float cos_cutoff, const uint32_t* row_ptr, const float* row_wsum,
                             uint32_t* col, float* val, void* stream);
/* y[nrows,3] = A x for a CSR matrix A (row_ptr[nrows+1]).  The kernel is bound by streaming A from HBM, so A may be stored
 * compactly: col holds uint16 (col_bytes 2, <= 65536 columns) or uint32 (4) column indices; val holds fp32 weights (val_bytes 4,
 * row_scale may be NULL) or 16-bit fixed-point weights q (val_bytes 2) with weight = q * row_scale[row].  lanes_per_row: 4 for
 * short rows, 64 for rows of hundreds of non-zeros.
 * val_bytes 8 = BLOCKED rows (the long rows of these filters touch runs of consecutive texels): row_ptr counts blocks, col holds the
 * uint16 block index (columns 4 b .. 4 b + 3), val four uint16 fixed-point weights per block (lowest column first; 0 where the row has
 * no entry), weight = q * row_scale[row].  Needs col_bytes 2, lanes_per_row 64, nrows % 4 == 0 (square filters: x has nrows texels), x
 * 16-byte and val 8-byte aligned; MRGS_E_BAD_ARG otherwise. */
int mrgs_csr_spmv3(int32_t nrows, const uint32_t* row_ptr, const void* col, int32_t col_bytes, const void* val, int32_t val_bytes,
                   const float* row_scale, const float* x, float* y, int32_t lanes_per_row, void* stream);
/* Up to MRGS_SPMV_MAX_BATCH independent products of the kind above in ONE launch (`descs` is a host array): the levels of
 * EnvLight.build_mips each way. */
#define MRGS_SPMV_MAX_BATCH 8
#define MRGS_SPMV_MAX_PANEL 1024       /* patches of a tile's panel (image_rows form of MrgsSpmvDesc) */
/* image_rows != NULL (ABI 8) = TILES OF ROWS OF ONE FUNDAMENTAL DOMAIN of the cube's symmetry group, as dense matrices.  The weight
 * of a filter is K(r, c) * area(c) / n(r) with K invariant under the 48 signed axis permutations g of the cube (K(g r, g c) = K(r, c);
 * the reference's texel solid angle, cubemap.cu:17-30, and with it the row sum are not -- they are per-texel factors), so the rows of the
 * texels r0 of a triangle of face 0 (y <= x < res / 2) are the whole operator:
 *     y[image_rows[r0 * 48 + g]] = row_scale[that row] * sum_c W(r0, c) * pre_scale[g c] * x[g c]      (image_rows < 0: skipped)
 * with g c as mrgs_cube_symmetry_rows lists it.  The rows are stored tile by tile (tile t = rows tile_ptr[t] .. tile_ptr[t + 1], at most
 * 16: a 4 x 4 patch of the triangle); panel_src[panel_ptr[t] .. panel_ptr[t + 1]) lists the 4 x 4 texel PATCHES ((face * res / 4 +
 * y / 4) * res / 4 + x / 4) the rows of tile t touch -- the image of a patch under a symmetry is a patch; `val` holds, for every
 * panel patch and each of 64 lanes (lane = 16 * (x & 3) + row in tile), the 16-bit fixed-point weights of the patch's four rows y & 3
 * as one 8-byte word (lowest row first) -- the A operand of v_mfma_f32_16x16x4_f32 as its lanes read it; the 48 symmetries x 3
 * channels are the product's right-hand sides.  nrows = 6 res^2 (the length of x, y, pre_scale, row_scale), res a power of two in
 * 4 .. 128; row_ptr, col, lanes_per_row, col_bytes, val_bytes are not used.  The matrices are 1/40 of the full ones and stay in cache;
 * x is read once per (tile, g). */
typedef struct MrgsSpmvDesc {
    int32_t nrows, lanes_per_row, col_bytes, val_bytes;
    const uint32_t* row_ptr;
    const void* col;
    const void* val;
    const float* row_scale;
    const float* x;
    float* y;
    const int32_t* image_rows;
    const float* pre_scale;
    const uint32_t* tile_ptr;
    const uint32_t* panel_ptr;
    const uint16_t* panel_src;
    int32_t res, n_tiles;
    int32_t max_panel, reserved;   /* the longest panel, in patches: 1 .. MRGS_SPMV_MAX_PANEL */
} MrgsSpmvDesc;
int mrgs_csr_spmv3_batched(const MrgsSpmvDesc* descs, int32_t n, void* stream);
/* rows[g * 6 res^2 + t] = the texel ((face * res + y) * res + x) that the g-th of the cube's 48 symmetries (signed axis permutations,
 * the library's order) maps texel t to; `rows` is a HOST array.  g = 0 is the identity. */
int mrgs_cube_symmetry_rows(int32_t res, int32_t* rows);
/* cubemap_mip applied n_steps times below `in` [6,res_in,res_in,3] (scene/light.py:74-76): outs[k] = level k + 1 ([6, res_in >> (k+1), ., 3]),
 * `outs` a host array of device pointers; bit-identical to n_steps calls of mrgs_cubemap_mip_forward, one launch per three levels.
 * The backward chain g[k] += cubemap_mip.backward(g[k + 1]) for k = n_levels - 2 ... 0 in place (g[k]: [6, res0 >> k, ., 3], g a host
 * array), one launch per level. */
int mrgs_cubemap_mip_chain_forward(int32_t res_in, int32_t n_steps, const float* in, float* const* outs, void* stream);
int mrgs_cubemap_mip_chain_backward(int32_t res0, int32_t n_levels, float* const* g, void* stream);
/* cubemap_mip (scene/light_utils.py:66-81): forward = 2x2 box filter [6,2r,2r,3] -> [6,r,r,3]; backward = the reference's own
 * rule (seamless bilinear cube fetch of 0.25 * dout at the finer level's texel-centre directions), ACCUMULATED into g_fine. */
int mrgs_cubemap_mip_forward(int32_t res_out, const float* in, float* out, void* stream);
int mrgs_cubemap_mip_backward(int32_t res_fine, const float* dout, float* g_fine, void* stream);

/* ---- cubemapencoder fetch primitive (BASELINE.json north star: "the cubemapencoder mip lookup") ---------------------------
 * Replaces `_cubemapencoder.cubemap_encode_forward / cubemap_encode_backward` of the reference's CUDA extension
 * (submodules/cubemapencoder/src/cubemapencoder.cu:430-485, 713-775; bindings.cpp), same argument meaning and layouts:
 * inputs [B,3] directions (a zero vector yields fail_value), cubemap [6,C,L,L], fail_value [C], outputs [C,B];
 * interp 0 = nearest (:380-428), 1 = bilinear; seamless != 0 takes edge texels from the neighbouring face and averages the three
 * texels of a cube vertex (:298-334, face / edge tables :66-187).  Backward: grad_cubemap [6,C,L,L] and grad_fail [C] are ACCUMULATED
 * into (the caller passes zeros, as cubemap_encoder.py:55-57 does), grad_inputs [B,3] is written in full (zeros for nearest, which
 * the reference leaves uninitialised). */
int mrgs_cubemap_encode_forward(const float* inputs, const float* cubemap, const float* fail_value, float* outputs, int32_t interp,
                                int32_t seamless, int64_t B, int32_t C, int32_t L, void* stream);
int mrgs_cubemap_encode_backward(const float* grad_outputs, const float* inputs, const float* cubemap, float* grad_cubemap, float* grad_inputs,
                                 float* grad_fail, int32_t interp, int32_t seamless, int64_t B, int32_t C, int32_t L, void* stream);

/* Visibility blend of get_specular_color_surfel (utils/refl_utils.py:393-401), one pass each way:
 * specular = (direct * vis + (1 - vis) * indirect) * alpha * weight, indirect_color = (1 - vis) * indirect * alpha * weight.
 * direct / specular / indirect_color [3,H,W], weight [H,W,3] (layouts of mrgs_shade_specular_forward), indirect [H,W,3] and alpha
 * [H,W,1] strided maps, visibility [H,W] (constant).  Backward writes g_direct [3,H,W], g_weight [H,W,3], g_indirect [H,W,3],
 * g_alpha [H,W] in full; either upstream gradient may be NULL. */
int mrgs_indirect_blend_forward(int32_t H, int32_t W, const float* direct, const float* weight, const MrgsStridedMap* indirect,
                                const MrgsStridedMap* alpha, const float* visibility, float* specular, float* indirect_color, void* stream);
int mrgs_indirect_blend_backward(int32_t H, int32_t W, const float* direct, const float* weight, const MrgsStridedMap* indirect,
                                 const MrgsStridedMap* alpha, const float* visibility, const float* g_specular, const float* g_indirect_color,
                                 float* g_direct, float* g_weight, float* g_indirect, float* g_alpha, void* stream);

/* g_features[8,H,W] of render_surfel's material map (refl, roughness, albedo[3], indirect[3]) assembled from the outputs of
 * mrgs_surfel_composite_backward (g_refl), mrgs_shade_specular_backward (g_refl, g_roughness [H,W]; g_albedo [H,W,3]) and, with the
 * visibility tracer, mrgs_indirect_blend_backward (g_indirect [H,W,3]; NULL = zeros).  In the same pass g_alpha [H,W] (NULL = skip) receives
 * the sum of the up to three alpha gradients of those kernels (g_alpha_a / _b / _c, each [H,W] or NULL). */
int mrgs_surfel_feature_grads(int32_t H, int32_t W, const float* g_refl_composite, const float* g_refl_shade, const float* g_roughness,
                              const float* g_albedo_hwc, const float* g_indirect_hwc, float* g_features, const float* g_alpha_a,
                              const float* g_alpha_b, const float* g_alpha_c, float* g_alpha, void* stream);

/* ---- per-gaussian inputs of the surfel renderer (fused glue) -----------------------------------------------------
 * One kernel instead of the ~50 torch kernels the reference runs per view before the rasterizer call: GaussianModel getters
 * (scene/gaussian_model.py:236-311: sigmoid / exp / normalize activations), get_normal (:269-285) and the feature assembly of
 * render_surfel (gaussian_renderer/__init__.py:338-355): view direction, facing unit normal (third column of R(q)), mirror
 * direction r = 2 (n.w_o) n - w_o, indirect = clamp_min(eval_sh(3, cat(indirect_dc, indirect_rest), r), 0),
 * features[P,8] = (sigmoid refl, sigmoid roughness, sigmoid ori_color[3], indirect[3]).  All pointers are device pointers to
 * contiguous fp32 tensors in the GaussianModel layout: xyz[P,3], scaling_raw[P,2], rotation_raw[P,4] (w,x,y,z), opacity_raw[P,1],
 * refl_raw[P,1], rough_raw[P,1], ori_color_raw[P,3], indirect_dc[P,1,3], indirect_rest[P,15,3], campos[3]. */
typedef struct MrgsSurfelParams {
    int32_t P;
    const float *xyz, *scaling_raw, *rotation_raw, *opacity_raw, *refl_raw, *rough_raw, *ori_color_raw, *indirect_dc, *indirect_rest,
        *campos;
    /* ABI 7, the "pgsr" flavour (arguments/config.py:1): with viewmatrix set (world_view_transform as stored, 16 floats) the feature rows
     * are TWELVE floats -- channel 8 = get_distance (gaussian_renderer/envgs_renderer.py:30-38: |facing normal . centre| in the camera
     * frame, the plane distance the flavour rasterizes as its last channel, __init__.py:348-357), channels 9..11 = 0 (the row is padded
     * to the blend kernels' 16-byte feature pieces) -- and its gradient joins the normal's and the centre's.  NULL: features[P,8]. */
    const float* viewmatrix;
} MrgsSurfelParams;
typedef struct MrgsSurfelGrads {   /* gradients w.r.t. the raw parameters, same shapes, fully written */
    float *d_xyz, *d_scaling, *d_rotation, *d_opacity, *d_refl, *d_rough, *d_ori_color, *d_indirect_dc, *d_indirect_rest;
} MrgsSurfelGrads;
/* outputs: opacity[P,1], scales[P,2], rotations[P,4] (unit), features[P,8] ([P,12] with MrgsSurfelParams::viewmatrix) -- exactly what
 * GaussianRasterizer is fed */
int mrgs_surfel_features_forward(const MrgsSurfelParams* p, float* opacity, float* scales, float* rotations, float* features,
                                 void* stream);
/* upstream gradients of the four outputs (any may be NULL = zero).  d_xyz = the part that flows through the view and mirror directions
 * + g_xyz_upstream [P,3] (NULL = zero): what reached the centres some other way -- the rasterizer's own dL/dmeans3D --, so that the
 * caller's sum of the two is no kernel of its own (ABI 6; must not alias d_xyz).
 * Where the upstream gradient of the three indirect-radiance channels (g_features[:, 5:8]) is exactly zero for all 64 rows of a wave, the
 * wave writes zeros to d_indirect_dc / d_indirect_rest without reading the coefficients: for FINITE coefficients that is what the chain
 * rule gives; a non-finite coefficient, which the reference's torch chain would turn into a NaN gradient (0 x inf), stays unnoticed
 * there.  d_indirect_rest may have any 4-byte alignment (16-byte aligned tensors take the wide stores). */
int mrgs_surfel_features_backward(const MrgsSurfelParams* p, const float* g_opacity, const float* g_scales, const float* g_rotations,
                                  const float* g_features, const MrgsSurfelGrads* grads, const float* g_xyz_upstream, void* stream);

/* ---- per-pixel maps of the surfel renderer (fused glue) ------------------------------------------------------------
 * compute_2dgs_normal_and_regularizations (gaussian_renderer/__init__.py:42-90) + depths_to_points / depth_to_normal
 * (utils/point_utils.py:9-37) + the normal_map of render_surfel (:419-421) in one kernel each way.
 *   view_rot   = world_view_transform[:3,:3] as stored (row-major): rend_normal = view_rot * allmap[2:5]
 *   ray_matrix = c2w[:3,:3] * intrins^-1, ray_origin = c2w[:3,3] with c2w, intrins exactly as depths_to_points builds them:
 *                point(x, y) = surf_depth * (ray_matrix * (x, y, 1)) + ray_origin
 * allmap is the rasterizer's [7,H,W] output.  Outputs: rend_normal[3,H,W], surf_depth[1,H,W], surf_normal[3,H,W] (already
 * multiplied by the detached alpha; NULL = skip, render_surfel(wo_render_img)), normal_map[H,W,3] = rend_normal / max(alpha,
 * 1e-6) (NULL = skip). */
typedef struct MrgsMapsFrame {
    int32_t H, W;
    float view_rot[9], ray_matrix[9], ray_origin[3];
    float depth_ratio;
    /* ABI 7, the "pgsr" flavour (arguments/config.py:1; gaussian_renderer/__init__.py:64-69): with rend_distance set ([H,W], the blended
     * plane distance = the flavour's last feature channel) surf_depth is nan_to_num(rend_distance / -(n . ray)), n = allmap[2:5],
     * ray = ((x - (W - 1) / 2) / pgsr_fx, (y - (H - 1) / 2) / pgsr_fy, 1) -- the depth where the pixel's ray meets the blended plane
     * (allmap[7] of the flavour's rasterizer, which is not in the reference's tree: parity unpinned) -- and surf_normal its finite
     * differences; depth_ratio is then not read.  The backward writes the map's gradient to g_rend_distance ([H,W], may be NULL). */
    float pgsr_fx, pgsr_fy;
    const float* rend_distance;
    float* g_rend_distance;
} MrgsMapsFrame;
int mrgs_surfel_maps_forward(const MrgsMapsFrame* fr, const float* allmap, float* rend_normal, float* surf_depth, float* surf_normal,
                             float* normal_map, float* rend_alpha /*[1,H,W] = allmap[1], NULL = skip*/, float* rend_dist /*[1,H,W] = allmap[6]*/,
                             float* rend_alpha2 /*a second copy of rend_alpha for a second consumer (ABI 6), NULL = skip*/, void* stream);
/* g_allmap[7,H,W] (fully written) from the upstream gradients of the four outputs and of rend_alpha = allmap[1:2] and
 * rend_dist = allmap[6:7], which are plain views in the reference ([1,H,W] each; any of the seven may be NULL = zero). */
int mrgs_surfel_maps_backward(const MrgsMapsFrame* fr, const float* allmap, const float* g_rend_normal, const float* g_surf_depth,
                              const float* g_surf_normal, const float* g_normal_map, const float* g_rend_alpha, const float* g_rend_dist,
                              const float* g_rend_alpha2 /*gradient of the second copy (ABI 6), NULL = zero*/, float* g_allmap, void* stream);

/* render_surfel compositing (gaussian_renderer/__init__.py:436-445): diffuse = (1 - refl) * base, render =
 * [linear_to_srgb]((diffuse + specular)) + bg * (1 - alpha).  base_color / specular / render / diffuse [3,H,W],
 * refl_strength / alpha [1,H,W], bg[3]; all contiguous fp32 device tensors. */
int mrgs_surfel_composite_forward(int32_t H, int32_t W, int32_t srgb, const float* base_color, const float* refl_strength, const float* specular,
                                  const float* alpha, const float* bg, float* render, float* diffuse, void* stream);
int mrgs_surfel_composite_backward(int32_t H, int32_t W, int32_t srgb, const float* base_color, const float* refl_strength,
                                   const float* specular, const float* bg, const float* g_render, const float* g_diffuse, float* g_base,
                                   float* g_refl, float* g_specular, float* g_alpha,
                                   float* zero_fill /*ABI 6: zero_floats floats cleared by the same launch (what the caller's next kernel
                                   accumulates into, e.g. mrgs_shade_specular_backward's texel gradients), NULL / 0 = nothing*/,
                                   int64_t zero_floats, void* stream);

/* ---- training loss of one view (fused) ---------------------------------------------------
 * Replaces calculate_loss (utils/loss_utils.py:142-228) = (1 - lambda_dssim) * l1_loss (:22-23) + lambda_dssim * (1 - ssim)
 * (ssim/_ssim, 11x11 gaussian window sigma 1.5, zero padding, :83-119) + lambda_normal * normal consistency (:166-175) +
 * lambda_dist * mean(rend_dist) (:180-182), and its autograd backward.  image / gt: [C,H,W] (C <= 4), rend_normal /
 * surf_normal: [3,H,W], rend_dist: [H,W], image_weight: [H,W] or NULL.  With image_weight the normal term is
 * mean(image_weight * sum_c |surf - rend|), without it mean(1 - sum_c rend * surf); lambda_normal <= 0 / lambda_dist <= 0
 * switch the terms off (the iteration gates of the reference are the caller's business; the pointers may then be NULL).
 * ws (mrgs_loss_ws_bytes) carries the SSIM derivative maps from forward to backward.
 * out_terms[16] (device): [0] loss, [1] Ll1, [2] ssim, [3] loss0, [4] normal term (unscaled mean), [5] lambda_dist * mean(dist),
 * [6] psnr (utils/image_utils.py psnr, mean over channels), [7..7+C) per-channel mse.  Sums are reduced in a fixed order.
 * out_loss (device scalar, may be NULL) receives the loss as well: a host binding can hand it out as its own tensor.
 * g_loss: device scalar dL/dloss (NULL = 1).  Gradients are written in full (no pre-zeroing by the caller). */
typedef struct MrgsLossConfig {
    int32_t H, W, C;
    float lambda_dssim, lambda_normal, lambda_dist;
} MrgsLossConfig;
size_t mrgs_loss_ws_bytes(int32_t H, int32_t W, int32_t C);
int mrgs_loss_forward(const MrgsLossConfig* cfg, const float* image, const float* gt, const float* rend_normal, const float* surf_normal,
                      const float* rend_dist, const float* image_weight, void* ws, size_t ws_bytes, float* out_terms, float* out_loss,
                      void* stream);
int mrgs_loss_backward(const MrgsLossConfig* cfg, const float* image, const float* gt, const float* rend_normal, const float* surf_normal,
                       const float* image_weight, const void* ws, const float* g_loss, float* g_image, float* g_rend_normal,
                       float* g_surf_normal, float* g_rend_dist, void* stream);

/* ---- closest-hit ray queries against a triangle mesh (visibility rays) --------------------
 * Replaces RayTracer(vertices, triangles).trace (submodules/raytracing/raytracing/raytracer.py:8-56,
 * raytracing_brdf/raytracer.py:18-123) = create_raytracer + TriangleBvh4::build / ray_trace_gpu
 * (submodules/raytracing/src/raytracer.cu:20-53, bvh.cu:259-302,526-609,694-720) with Triangle::ray_intersect
 * (include/raytracing/triangle.cuh:27-45): back faces are ignored, hits need 0 <= t < 10, depth = 10 marks a miss.
 * mrgs_bvh_build runs on the HOST (as the reference's build does): vertices [n_vertices,3] f32 and triangles [n_triangles,3] i32
 * are host arrays, the hierarchy is written into blob_host (mrgs_bvh_bytes, position independent); the caller copies the blob
 * to the device once per mesh.  mrgs_bvh_trace: all pointers are device pointers; rays_o / rays_d [n_rays,3], outputs
 * positions / normals [n_rays,3] (may alias rays_o / rays_d: the reference's inplace mode), depth [n_rays], face_ids [n_rays]
 * (index into `triangles`, -1 for a miss; may be NULL). */
size_t mrgs_bvh_bytes(int64_t n_triangles);
int mrgs_bvh_build(const float* vertices, int64_t n_vertices, const int32_t* triangles, int64_t n_triangles, void* blob_host,
                   size_t blob_bytes);
int mrgs_bvh_trace(const void* blob_dev, int64_t n_triangles, int64_t n_rays, const float* rays_o, const float* rays_d, float* positions,
                   float* normals, float* depth, int32_t* face_ids, void* stream);
/* The visibility block of get_specular_color_surfel (utils/refl_utils.py:379-391) in one launch: per pixel with alpha > 0 the mirror
 * ray of the view direction about `normal` ([H,W,3]) is started at rays_o + surf_depth * rays_cam (un-normalised pixel ray,
 * sample_camera_rays_unnormalize :75-93; Kinv = host inverse intrinsics, R / T = device Camera.R / Camera.T as in MrgsShadeFrame)
 * and visibility[H,W] = 1 if nothing is hit within 10 units (or alpha <= 0), else 0.  No ray buffers, no mask compaction. */
int mrgs_bvh_visibility(const void* blob_dev, int64_t n_triangles, int32_t H, int32_t W, const float* Kinv, const float* R, const float* T,
                        const MrgsStridedMap* normal, const MrgsStridedMap* alpha, const float* surf_depth, float* visibility, void* stream);

/* ---- surfel ray tracer (second half) -----------------------------------------------------------------
 * Replaces the un-vendored OptiX extension `diff_surfel_tracing` behind HardwareRendering (gaussian_renderer/optix_utils.py:14-271):
 * SurfelTracer.build_acceleration_structure(v, f, rebuild) (:76) -> mrgs_surfel_bvh_build, SurfelTracer.forward (:185-197) ->
 * mrgs_surfel_trace_forward, its autograd backward -> mrgs_surfel_trace_backward.  The extension's arithmetic is not in the reference
 * tree: the definition is stated in csrc/mrgs_surfel_trace.hip (2DGS compositing of forward.cu:366-420 along a ray) and restated
 * densely in oracle/surfel_trace_oracle.py; parity with the OptiX binary is unpinned.  All pointers are device pointers except
 * bg_host (3 floats on the host).
 * Build (on the device, every call): quad_vertices [n_surfels,4,3] = the four corners get_disks (:36-66) produces per surfel; blob
 * (mrgs_surfel_bvh_bytes) receives the hierarchy, ws (mrgs_surfel_bvh_ws_bytes) is scratch that may be reused after the call's
 * kernels have run.
 * Trace: ray_o / ray_d [n_rays,3] (direction not normalised: depths are ray parameters); ray_width > 0 says the rays are an image
 * with rows of that length (n_rays a multiple of it): a wavefront then takes an 8x8 block of neighbouring rays instead of 64 of a
 * row (same results, shorter walks); the trace calls write the surfel records in leaf order into the blob (not const); geom [n_surfels,16] per surfel
 * (mean.xyz, a.xyz, b.xyz, n.xyz, opacity, 3 unused) with a = r_u / s_u, b = r_v / s_v, n = r_u x r_v; attr [n_surfels,8] =
 * (rgb, others[2], 3 unused).  Outputs rgb / norm [n_rays,3], dpt / acc / dist [n_rays], aux [n_rays,2], wet [n_surfels] (summed
 * blend weight per surfel, cleared by the call), state [mrgs_surfel_trace_state_floats(n_rays, ray_width)] (per ray: sum w t^2, final transmittance, hits blended, passes -- negative when the ray walked in a packet --; behind them the list of the rays traced one per wavefront and the record of the ids every wavefront gathered, pass by pass, which the
 * backward replays instead of walking the hierarchy again (4 KB per wavefront and pass; when it overflows the backward walks): what the
 * backward needs beyond the outputs).  Backward: g_* of the six outputs in, g_geom [n_surfels,16] / g_attr [n_surfels,8] (cleared by
 * the call, same layout as geom / attr) and g_ray_o / g_ray_d [n_rays,3] out. */
size_t mrgs_surfel_bvh_bytes(int64_t n_surfels);
size_t mrgs_surfel_bvh_ws_bytes(int64_t n_surfels);
size_t mrgs_surfel_trace_state_floats(int64_t n_rays, int32_t ray_width);
/* A state WITHOUT the replay record (the per-ray part and the lists only; ~1/100 of the full size): enough for a forward whose
 * backward will never run (evaluation, torch.no_grad()).  Both trace calls take the size of the state they are handed (state_floats):
 * at least the full size -> the forward records; at least this size -> it does not, and a backward on such a state walks the hierarchy
 * again; smaller -> MRGS_E_WORKSPACE.  n_rays >= 2^31 (or >= 2^27 blocks of rays) -> MRGS_E_UNSUPPORTED.  n_surfels == 0: outputs are
 * the background / zeros, `state` is not touched. */
size_t mrgs_surfel_trace_state_floats_norecord(int64_t n_rays, int32_t ray_width);
/* Introspection for the tests: word offsets inside `state` of [0] the lists of rays traced one per wavefront (their counts, one per
 * XCD region of the launch, at words 0, 64, ..., 448 of the list's 512-word header), [1] the lists of packets handed to the second launch
 * (counts the same way),
 * [2] the record header (word 0: chunks taken from the shared
 * pool, word 1: non-zero = no usable record, the backward walks again), [3] the replay record, [4] the full size. */
int mrgs_surfel_trace_state_layout(int64_t n_rays, int32_t ray_width, size_t* offsets5);
int mrgs_surfel_bvh_build(const float* quad_vertices, int64_t n_surfels, void* blob, size_t blob_bytes, void* ws, size_t ws_bytes, void* stream);
int mrgs_surfel_trace_forward(void* blob, int64_t n_surfels, int64_t n_rays, int32_t ray_width, const float* ray_o, const float* ray_d, const float* geom,
                              const float* attr, const float* bg_host, float* rgb, float* dpt, float* acc, float* norm, float* dist,
                              float* aux, float* wet, float* state, size_t state_floats, void* stream);
int mrgs_surfel_trace_backward(void* blob, int64_t n_surfels, int64_t n_rays, int32_t ray_width, const float* ray_o, const float* ray_d, const float* geom,
                               const float* attr, const float* bg_host, const float* rgb, const float* dpt, const float* acc,
                               const float* norm, const float* aux, const float* state, size_t state_floats, const float* g_rgb, const float* g_dpt,
                               const float* g_acc, const float* g_norm, const float* g_dist, const float* g_aux /* any of the six g_* may be NULL = zeros */,
                               float* g_geom, float* g_attr, float* g_ray_o, float* g_ray_d, void* stream);

/* The per-surfel records of the tracer from the model's tensors, and the way back: what HardwareRendering.render_gaussians prepares
 * around the tracer call (gaussian_renderer/optix_utils.py:36-66 get_disks, :124-183) in one launch each.  scales [P,2] (multiplied by
 * scale_modifier), rotations [P,4] (w,x,y,z; normalised inside as build_rotation does), opacities [P]; exactly one of shs [P,M,3]
 * (colour = max(SH(degree, direction from campos) + 0.5, 0), forward.cu:20-81) and colors_precomp [P,3]; others [P,2] nullable;
 * campos: 3 floats on the device.  quad_vertices [P,4,3] (nullable) receives get_disks' corners for mrgs_surfel_bvh_build.
 * Backward: g_geom [P,16] / g_attr [P,8] in, gradients of every input out (g_shs or g_colors_precomp, g_others nullable). */
int mrgs_surfel_trace_prep_forward(int64_t P, const float* means3D, const float* scales, const float* rotations, const float* opacities,
                                   const float* shs, int32_t M, int32_t sh_degree, const float* colors_precomp, const float* others,
                                   const float* campos, float scale_modifier, float* geom, float* attr, float* quad_vertices, void* stream);
int mrgs_surfel_trace_prep_backward(int64_t P, const float* means3D, const float* scales, const float* rotations, const float* shs, int32_t M,
                                    int32_t sh_degree, const float* campos, float scale_modifier, const float* g_geom, const float* g_attr,
                                    float* g_means3D, float* g_scales, float* g_rotations, float* g_opacities, float* g_shs,
                                    float* g_colors_precomp, float* g_others, void* stream);

/* The same records straight from the model's own tensors (GaussianModel's activations applied inside, scene/gaussian_model.py:56-78,
 * 236-259): scaling_raw [P,2] (scales = exp), rotation_raw [P,4], opacity_raw [P] (opacities = sigmoid), the colour SH split as the model
 * stores it -- features_dc [P,1,3], features_rest [P,15,3] -- and the gradients in the same layout: replaces the getters' ~12 torch
 * kernels (incl. the cat of the two SH tensors) and their ~25 backward kernels per traced view. */
int mrgs_surfel_trace_prep_raw_forward(int64_t P, const float* xyz, const float* scaling_raw, const float* rotation_raw, const float* opacity_raw,
                                       const float* features_dc, const float* features_rest, int32_t sh_degree, const float* others,
                                       const float* campos, float scale_modifier, float* geom, float* attr, float* quad_vertices, void* stream);
int mrgs_surfel_trace_prep_raw_backward(int64_t P, const float* xyz, const float* scaling_raw, const float* rotation_raw, const float* opacity_raw,
                                        const float* features_dc, const float* features_rest, int32_t sh_degree, const float* campos,
                                        float scale_modifier, const float* g_geom, const float* g_attr, float* g_xyz, float* g_scaling_raw,
                                        float* g_rotation_raw, float* g_opacity_raw, float* g_features_dc, float* g_features_rest, float* g_others,
                                        void* stream);

/* The mirror rays of a rendered view, as render_indirect / render_surfel_with_envgs set them up (gaussian_renderer/envgs_renderer.py:717-724,
 * __init__.py:496-505): origin = camera centre + surf_depth * un-normalised pixel ray + 1e-3 * direction, direction = unit mirror
 * direction of the view ray about `normal` ([H,W,3], any strides).  Kinv: inverse intrinsics on the host; R / T: device Camera.R /
 * Camera.T as in MrgsShadeFrame.  Backward: gradients of ray_o / ray_d to normal ([H,W,3] contiguous) and surf_depth. */
int mrgs_mirror_rays_forward(int32_t H, int32_t W, const float* Kinv_host, const float* R, const float* T, const MrgsStridedMap* normal,
                             const float* surf_depth, float* ray_o, float* ray_d, void* stream);
int mrgs_mirror_rays_backward(int32_t H, int32_t W, const float* Kinv_host, const float* R, const float* T, const MrgsStridedMap* normal,
                              const float* g_ray_o, const float* g_ray_d, float* g_normal, float* g_surf_depth, void* stream);
/* The same with the reflecting normal built inside from the BLENDED normal map and alpha, as render_surfel_with_envgs does
 * (gaussian_renderer/__init__.py:493-495): normal = safe_normalize(rend_normal / clamp_min(alpha, 1e-6)); rend_normal [H,W,3] by strides
 * (the [3,H,W] map is passed as strides (W, 1, H W)), alpha [H,W].  Backward: + g_alpha [H,W]. */
int mrgs_mirror_rays_blended_forward(int32_t H, int32_t W, const float* Kinv_host, const float* R, const float* T, const MrgsStridedMap* rend_normal,
                                     const float* alpha, const float* surf_depth, float* ray_o, float* ray_d, void* stream);
int mrgs_mirror_rays_blended_backward(int32_t H, int32_t W, const float* Kinv_host, const float* R, const float* T, const MrgsStridedMap* rend_normal,
                                      const float* alpha, const float* g_ray_o, const float* g_ray_d, float* g_rend_normal, float* g_alpha,
                                      float* g_surf_depth, void* stream);
/* out = a (1 - s) + s b per channel: the traced light blended into the rendered view (gaussian_renderer/__init__.py:517).  a / out / g_*
 * [3,H,W] contiguous; b by element strides (channel, pixel: a [H,W,3] tensor seen as [3,H,W] is (1, 3)), s by its pixel stride; backward:
 * g_a [3,H,W], g_b in b's layout, g_s [H,W], all fully written. */
int mrgs_traced_blend_forward(int32_t H, int32_t W, const float* a, const float* b, int64_t b_channel_stride, int64_t b_pixel_stride, const float* s,
                              int64_t s_pixel_stride, float* out, void* stream);
int mrgs_traced_blend_backward(int32_t H, int32_t W, const float* a, const float* b, int64_t b_channel_stride, int64_t b_pixel_stride, const float* s,
                               int64_t s_pixel_stride, const float* g_out, float* g_a, float* g_b, float* g_s, void* stream);


/* ---- optimizer step -------------------------------------------------------------------------
 * torch.optim.Adam(l, lr=0.0, eps=1e-15).step() of GaussianModel.training_setup (scene/gaussian_model.py:417-453) for every
 * parameter tensor in one launch (per MRGS_ADAM_MAX_TENSORS tensors): amsgrad off, no weight decay.  `tensors` is a HOST array;
 * param / grad / exp_avg / exp_avg_sq are device pointers to `numel` contiguous fp32 values (updated in place, grad read only),
 * lr the group's learning rate, step the 1-based count of this update for this tensor (torch keeps one per parameter).
 * beta1 / beta2 / eps are doubles like torch's python scalars: 1 - beta and the bias corrections are formed in double and
 * rounded to fp32 where torch rounds them. */
#define MRGS_ADAM_MAX_TENSORS 32
typedef struct MrgsAdamTensor {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    int64_t numel;
    float lr;
    int32_t step;
} MrgsAdamTensor;
int mrgs_adam_step(const MrgsAdamTensor* tensors, int32_t n_tensors, double beta1, double beta2, double eps, void* stream);

/* ---- densify / prune compaction ----------------------------------------------------------------
 * Replaces the ~50 boolean-index calls of _prune_optimizer / prune_points (scene/gaussian_model.py:856-905) -- parameter tensors, their
 * two Adam moments, xyz_gradient_accum / denom / max_radii2D -- by one scan of the keep mask and one gather launch for all tensors.
 * keep: device uint8 [n_rows] (non-zero = keep).  mrgs_compact_count scans it into ws (mrgs_compact_ws_bytes) and writes the number
 * of surviving rows to count_dev (device int64): the caller reads it once, allocates the exact-size destinations and calls
 * mrgs_compact_rows with the same keep / ws.  Rows are fp32 rows of `row_floats` values (any 4-byte type can be passed as such);
 * surviving rows keep their order, like tensor[mask]. */
#define MRGS_COMPACT_MAX_TENSORS 64
typedef struct MrgsCompactTensor {
    const float* src;      /* [n_rows, row_floats] */
    float* dst;            /* [count, row_floats] */
    int32_t row_floats;
} MrgsCompactTensor;
size_t mrgs_compact_ws_bytes(int64_t n_rows);
int mrgs_compact_count(int64_t n_rows, const uint8_t* keep, void* ws, size_t ws_bytes, int64_t* count_dev, void* stream);
int mrgs_compact_rows(int64_t n_rows, const uint8_t* keep, const void* ws, const MrgsCompactTensor* tensors, int32_t n_tensors, void* stream);

/* View-parallel training (materialrefgs_amd/dist.py): sum over V views of the SH colour gradients from each view's masked colour
 * gradient dRGB_v = dL/dsh_v[:,0,:] / SH_C0 and camera centre: dL_dsh[p][k][c] = sum_v B_k(normalize(means3D[p] - campos_v)) dRGB_v[p][c]
 * for k < (D+1)^2, 0 beyond (backward.cu:22-141).  gathered = V rows of row_stride floats, row v = [dRGB_v (P x 3) | campos_v (3)]
 * -- the layout an all-gather of the per-rank rows produces.  Replaces the all-reduce of the [P,M,3] gradient (16x the bytes). */
int mrgs_sh_grad_expand(int32_t P, int32_t M, int32_t D, int32_t V, const float* means3D, const float* gathered, int64_t row_stride,
                        float* dL_dsh, void* stream);

/* The same exchange for render_surfel's parameter set (BASELINE config 5): colour SH along the view direction and indirect-radiance SH
 * (eval_sh(3, .) along the mirror direction of the facing normal, gaussian_renderer/__init__.py:338-352) are both rank one per view.
 * gathered row v = [dRGB_v (P x 3) | dIND_v (P x 3) | campos_v (3)] with dRGB_v = d features_dc_v / SH_C0 and
 * dIND_v = d indirect_dc_v / SH_C0; xyz [P,3] and rotation_raw [P,4] (w,x,y,z, un-normalised) are the replicated parameters the
 * directions are rebuilt from.  Writes the summed gradients of features_dc [P,1,3], features_rest [P,15,3] (coefficients beyond
 * (D+1)^2 zero), indirect_dc [P,1,3], indirect_rest [P,15,3]: 96 of the 111 gradient floats per gaussian never cross the links. */
int mrgs_sh_grad_expand_surfel(int32_t P, int32_t D, int32_t V, const float* xyz, const float* rotation_raw, const float* gathered,
                               int64_t row_stride, float* g_features_dc, float* g_features_rest, float* g_indirect_dc, float* g_indirect_rest,
                               void* stream);
/* The same expansion with the three parts of a rank's row in buffers of their own (V rows each, strides in floats): dRGB_v is final after
 * the blend backward and dIND_v only after the per-gaussian backward, so a view-parallel step gathers them at different times
 * (materialrefgs_amd/dist.py: SurfelGradReducer.begin_early_rgb / begin_early_ind) and expands from where the two all-gathers left them.
 * rgb_rows NULL or ind_rows NULL (not both): that family is left out and its two output tensors are not touched (they may be NULL) --
 * one call per family, each as soon as its rows have arrived.  ABI 7. */
int mrgs_sh_grad_expand_surfel_rows(int32_t P, int32_t D, int32_t V, const float* xyz, const float* rotation_raw, const float* rgb_rows,
                                    int64_t rgb_stride, const float* ind_rows, int64_t ind_stride, const float* campos_rows, int64_t campos_stride,
                                    float* g_features_dc, float* g_features_rest, float* g_indirect_dc, float* g_indirect_rest, void* stream);

/* Introspection used by the parity tests: copies of internal state in the reference's layouts.
 * which: 0 depths f32[P], 1 means2D f32[P,2], 2 transMat f32[P,9], 3 normal_opacity f32[P,4], 4 rgb f32[P,3],
 * 5 tiles_touched u32[P], 6 clamped u8[P,3], 7 point_list u32[R], 8 ranges u32[tiles,2], 9 final_T f32[3,H,W],
 * 10 n_contrib u32[2,H,W], 11 depth-sorted gaussian order u32[P], 12 pixels the forward re-rendered exactly u32[2 + H W] ([0] their
 * count, from [2] their indices), 13 quadrant masks of the tile lists u8[R] (which 8x8 blocks of its tile an entry's box touches),
 * 14 the block-cull records f32[P,12] (centre, A, C | B/C, B/A, det/C, det/A | mean2D, r^2 of the low-pass disc, bound of the centre's
 * error -- csrc/mrgs_blend_math.h: CullConic; tools/cull_model.py restates them on the CPU).
 * dst is a device pointer. */
int mrgs_debug_export(const MrgsRasterConfig* cfg, const void* geom_ws, const void* binning_ws, const void* img_ws,
                      int64_t num_rendered, int32_t which, void* dst, void* stream);

/* HIP-event timing of the last forward_render / backward call's dominant kernels on their stream (bench.py). */
typedef struct MrgsKernelTimes {
    float preprocess_ms, sort_ms, duplicate_ms, render_fwd_ms, render_bwd_ms, preprocess_bwd_ms;
} MrgsKernelTimes;
int mrgs_set_profiling(int32_t level);   /* 0 off, 1 every stage, 2 only the two blend kernels, 3 the backward blend kernel on every
                                            fourth call (an event pair costs two ~6 us bubbles on the stream) */
int mrgs_get_kernel_times(MrgsKernelTimes* out);

const char* mrgs_strerror(int code);
const char* mrgs_last_hip_error(void);
const char* mrgs_version(void);
/* A second stream beside the caller's, for work that depends on nothing the caller's stream will produce for a while (the environment
 * prefilter of a view: six latency-bound SpMVs and the mip chain, which only the shading reads -- next to the issue-bound blend kernels;
 * scene/light.py:72-86 is called once per iteration, train_refnerf.py:1157-1163).  The library keeps ONE side stream per
 * device (any host thread may fork or join).  fork: everything queued on `main_stream` so far happens-before whatever is queued on the returned stream afterwards (pass it
 * as the `stream` of the calls to overlap); join: everything queued on the side stream so far happens-before whatever is queued on
 * `main_stream` afterwards.  Two event records and two stream waits, no host synchronisation.  Buffers the side work reads or writes
 * must stay allocated until the join has been queued (a caching allocator would hand them to the main stream's next kernels otherwise).
 * ABI 7. */
int mrgs_side_stream_fork(void* main_stream, void** side_stream);
int mrgs_side_stream_join(void* main_stream);
/* fork from the point of `main_stream` where a rasterizer forward launched its blend kernel: what is queued on the side stream then runs
 * beside the forward blend -- whose duration is the lifetime of a few long waves and which leaves issue slots and memory bandwidth idle --
 * instead of beside the bandwidth-bound kernels in front of it.  The point is a ONE-SHOT mark (ABI 9):
 *   mrgs_side_stream_arm_blend_mark(main_stream)   the caller owes side work and wants it forked from the NEXT forward's blend on this
 *                                                  stream; forgets any older mark.  Call it where the side work's inputs are final and
 *                                                  its buffers are allocated (EnvLight.build_mips does, after the optimizer step);
 *   the next mrgs_rasterize_forward* on that stream records the mark in front of its blend kernel;
 *   mrgs_side_stream_fork_at_blend(main_stream, &side) consumes it.  Without a mark recorded on `main_stream` since the arming (no forward
 *                                                  came, or it ran on another stream) this is a plain fork; a join drops an unconsumed mark.
 * Contract of the caller: every buffer the side work reads or writes was allocated BEFORE the arming call (a caching allocator may hand
 * out, after the forward, memory whose last reader is the very blend kernel the side work runs beside), and the side work's inputs are not
 * written on `main_stream` between the arming and the join. */
int mrgs_side_stream_arm_blend_mark(void* main_stream);
int mrgs_side_stream_fork_at_blend(void* main_stream, void** side_stream);

/* Revision of this header's struct layouts and call signatures; a binding compares it with the MRGS_ABI_VERSION it was written
 * against before the first call (materialrefgs_amd/_lib.py does). */
#define MRGS_ABI_VERSION 10
int32_t mrgs_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MRGS_H_INCLUDED */
