/*
 * mrgs_oracle.c -- CPU restatement of the reference surfel rasterizer (TEST INFRASTRUCTURE ONLY).
 *
 * This file is the parity checker for the HIP library in materialrefgs_amd/csrc.  It is NOT part of the
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * PARITY UNPINNED: the reference rasterizer is CUDA-only
 * (/root/reference/submodules/diff-surfel-rasterization/cuda_rasterizer/{forward,backward,rasterizer_impl}.cu);
 * it needs nvcc + cuda_runtime.h + CUB + cooperative_groups, none of which exist in this image, and the
 * reference ships no test, golden image or known-answer vector for this path.  It is
 * therefore "unbuildable here" and this restatement is anchored on a line-by-line reading of those sources
 * (citations below as file:line, relative to cuda_rasterizer/), on finite-difference checks of the parts of
 * the backward that ARE exact derivatives (opacity / colour / feature, tests/test_oracle.py) and on structural
 * invariants -- not on outputs of the reference itself.
 *
 * Arithmetic: IEEE fp32, compiled with -ffp-contract=off so that nothing fuses implicitly.
 *   - per-gaussian code (preprocess, its backward): evaluated un-fused in the operation order written in the
 *     reference; the HIP preprocess kernel is compiled the same way, so the geometry state and the whole integer
 *     binning state are bit-identical.
 *   - per-(pixel, surfel) blend code: the reference's expressions with the multiply-adds FUSED EXPLICITLY (fmaf)
 *     exactly where materialrefgs_amd/csrc/mrgs_blend_math.h fuses them.  nvcc fuses the reference's own build too
 *     (-fmad=true is its default), in a pattern that cannot be known here; choosing one fixed pattern on both sides
 *     makes the ill-conditioned ray/splat cross product bit-reproducible.  The GPU's reciprocal (v_rcp_f32) and exp
 *     (v_exp_f32) are approximations good to a few ulp; here they are the IEEE quotient 1.0f/x and the correctly rounded
 *     exponential R_EXP(x).  The kernels take every DECISION (alpha >= 1/255, depth >= 0.2, rho3d <= rho2d, T (1 - alpha) < 1e-4,
 *     T > 0.5) with exactly these two operations wherever the fast value is within its error band of the threshold
 *     (mrgs_blend_math.h, "Exact decisions"), so contributor counters and the set of blended pairs are compared bit for bit.
 * Deviations, all documented in DESIGN.md:
 *   - rR_SQRT(x) is evaluated as 1.0f/R_SQRT(x) (CUDA's rsqrtf is a 2-ulp approximation that cannot be restated);
 *   - per-gaussian gradient sums (the reference's fp32 atomicAdd, whose order is nondeterministic) are
 *     accumulated in double and rounded once, so the oracle is the "order-free" value of the same fp32 terms;
 *   - real->int casts saturate like the GPU conversion instructions do (NaN -> 0).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- arithmetic modes (oracle/Makefile builds one library per mode) ------------------------------------------------
 *   default              real = float, blend multiply-adds fused in the kernel's pattern (the bit-level checker above)
 *   MRGS_ORACLE_LITERAL  blend code written with the reference's own expression trees (forward.cu:366-420,
 *                        backward.cu:296-465), nothing fused, divisions where the reference divides: the "literal un-fused
 *                        fp32 reading" of the reference
 *   MRGS_ORACLE_F64      (with LITERAL) the same literal expression trees evaluated in double on the same fp32 inputs: the
 *                        "true value of the reference's formulas".  Thresholds keep their float-literal values (0.2f, 1/255, ...);
 *                        the sort key uses the depth rounded to float, as the reference's 32-bit key field does.
 * tests/test_truth_leg.py measures how far the HIP kernels and the two fp32 readings are from the F64 build. */
#ifdef MRGS_ORACLE_F64
typedef double real;
#define R_SQRT sqrt
#define R_EXP exp
#define R_CEIL ceil
#define R_FMIN fmin
#else
typedef float real;
#define R_SQRT sqrtf
#ifdef MRGS_ORACLE_LITERAL
#define R_EXP expf
#else
/* the bit-level checker takes exp CORRECTLY ROUNDED (glibc's double exp, < 1 ulp of a double, rounded once): a definition the HIP
 * kernels can meet bit for bit where a decision hangs on it (mrgs_blend_math.h: mrgs_exp_cr), which "whatever this libm's expf
 * returns" is not (glibc's expf is within 0.502 ulp, i.e. not always the correctly rounded value) */
#define R_EXP(x) ((float)exp((double)(x)))
#endif
#define R_CEIL ceilf
#define R_FMIN fminf
#endif
#if defined(MRGS_ORACLE_F64) && !defined(MRGS_ORACLE_LITERAL)
#error "MRGS_ORACLE_F64 is only defined together with MRGS_ORACLE_LITERAL"
#endif

#define BLOCK_X 16               /* config.h:19 */
#define BLOCK_Y 16               /* config.h:20 */
#define NEAR_N 0.2f              /* auxiliary.h:39 */
#define FAR_N 100.0f             /* auxiliary.h:40 */
#define FILTER_INV_SQUARE 2.0f   /* auxiliary.h:41 */
#define MAX_FEATURES 24          /* config.h:17 */

static const real SH_C0 = 0.28209479177387814f;                 /* auxiliary.h:44-61 */
static const real SH_C1 = 0.4886025119029199f;
static const real SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const real SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

typedef struct { real x, y, z; } f3;

typedef struct mrgs_oracle_ctx {
    int P, S, D, M, H, W, tiles_x, tiles_y;
    int have_sh, have_scale;
    float tanfovx, tanfovy, scale_modifier;
    real bg[3], view[16], proj[16], campos[3];
    /* geometry state (GeometryState, rasterizer_impl.cu:157-172) */
    real *depths, *means2D, *transMat, *normal_opacity, *rgb;
    int *radii;
    uint32_t *tiles_touched, *point_offsets;
    uint8_t *clamped;
    /* binning state (BinningState, rasterizer_impl.cu:183-196) */
    int64_t R;
    uint64_t *keys;
    uint32_t *point_list;
    /* image state (ImageState, rasterizer_impl.cu:174-181) */
    uint32_t *ranges;      /* [tiles][2] */
    real *final_T;        /* [3][H*W]: T, M1, M2 */
    uint32_t *n_contrib;   /* [2][H*W]: last, median */
    /* outputs */
    real *out_color, *out_feature, *out_others;
    /* inputs kept for backward */
    const real *means3D, *shs, *colors_precomp, *features, *opacities, *scales, *rotations, *transMat_precomp;
} mrgs_oracle_ctx;

/* GPU-style saturating real->int conversion ((int) casts in auxiliary.h:71-76) */
static int f2i(real v)
{
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (-2147483647 - 1);
    return (int)v;
}
static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* auxiliary.h:220-242 (quat stored w,x,y,z; result column-major R[c][r]) */
static void quat_to_rotmat(const real *q, real R[3][3])
{
    real s = 1.0f / R_SQRT(q[3] * q[3] + q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    real w = q[0] * s, x = q[1] * s, y = q[2] * s, z = q[3] * s;
    R[0][0] = 1.f - 2.f * (y * y + z * z);
    R[0][1] = 2.f * (x * y + w * z);
    R[0][2] = 2.f * (x * z - w * y);
    R[1][0] = 2.f * (x * y - w * z);
    R[1][1] = 1.f - 2.f * (x * x + z * z);
    R[1][2] = 2.f * (y * z + w * x);
    R[2][0] = 2.f * (x * z + w * y);
    R[2][1] = 2.f * (y * z - w * x);
    R[2][2] = 1.f - 2.f * (x * x + y * y);
}

/* rows of splat2world^T times world2ndc: clip = (a0,a1,a2,a3) . M  (forward.cu:99-111,122) */
static void row_times_proj(const real a[4], const real *pm, real out[4])
{
    for (int c = 0; c < 4; c++)
        out[c] = a[0] * pm[0 + c] + a[1] * pm[4 + c] + a[2] * pm[8 + c] + a[3] * pm[12 + c];
}

/* forward.cu:77-125.  T is returned as Tu,Tv,Tw (glm columns 0,1,2 == transMats[0..8]) */
static void compute_transmat(const real *p, const real *scale, real mod, const real *rot,
                             const real *proj, const real *view, int W, int H, real T[9], f3 *normal)
{
    real R[3][3];
    quat_to_rotmat(rot, R);
    real sx = mod * scale[0], sy = mod * scale[1];
    /* L = R * S with S = diag(sx, sy, 1): glm column combos, zero terms kept (x*s + y*0 + z*0) */
    real L0[3], L1[3], L2[3];
    for (int r = 0; r < 3; r++) {
        L0[r] = R[0][r] * sx + R[1][r] * 0.0f + R[2][r] * 0.0f;
        L1[r] = R[0][r] * 0.0f + R[1][r] * sy + R[2][r] * 0.0f;
        L2[r] = R[0][r] * 0.0f + R[1][r] * 0.0f + R[2][r] * 1.0f;
    }
    real a0[4] = {L0[0], L0[1], L0[2], 0.0f};
    real a1[4] = {L1[0], L1[1], L1[2], 0.0f};
    real a2[4] = {p[0], p[1], p[2], 1.0f};
    real c0[4], c1[4], c2[4];
    row_times_proj(a0, proj, c0);
    row_times_proj(a1, proj, c1);
    row_times_proj(a2, proj, c2);
    real hw = (real)((double)(real)W / 2.0), ow = (real)((double)(real)(W - 1) / 2.0);
    real hh = (real)((double)(real)H / 2.0), oh = (real)((double)(real)(H - 1) / 2.0);
    const real *cs[3] = {c0, c1, c2};
    for (int i = 0; i < 3; i++) {
        const real *c = cs[i];
        T[0 + i] = c[0] * hw + c[1] * 0.0f + c[2] * 0.0f + c[3] * ow;   /* Tu */
        T[3 + i] = c[0] * 0.0f + c[1] * hh + c[2] * 0.0f + c[3] * oh;   /* Tv */
        T[6 + i] = c[0] * 0.0f + c[1] * 0.0f + c[2] * 0.0f + c[3] * 1.0f; /* Tw */
    }
    /* auxiliary.h:101-109 */
    normal->x = view[0] * L2[0] + view[4] * L2[1] + view[8] * L2[2];
    normal->y = view[1] * L2[0] + view[5] * L2[1] + view[9] * L2[2];
    normal->z = view[2] * L2[0] + view[6] * L2[1] + view[10] * L2[2];
}

/* forward.cu:129-159 */
static int compute_aabb(const real T[9], real cutoff, real center[2], real extent[2])
{
    const real *T0 = T, *T1 = T + 3, *T3 = T + 6;
    real t[3] = {cutoff * cutoff, cutoff * cutoff, -1.0f};
    real distance = ((T3[0] * T3[0]) * t[0] + (T3[1] * T3[1]) * t[1]) + (T3[2] * T3[2]) * t[2];
    real inv = 1 / distance;
    real f[3] = {inv * t[0], inv * t[1], inv * t[2]};
    if (distance == 0.0f) return 0;
    center[0] = ((f[0] * T0[0]) * T3[0] + (f[1] * T0[1]) * T3[1]) + (f[2] * T0[2]) * T3[2];
    center[1] = ((f[0] * T1[0]) * T3[0] + (f[1] * T1[1]) * T3[1]) + (f[2] * T1[2]) * T3[2];
    real tmp0 = ((f[0] * T0[0]) * T0[0] + (f[1] * T0[1]) * T0[1]) + (f[2] * T0[2]) * T0[2];
    real tmp1 = ((f[0] * T1[0]) * T1[0] + (f[1] * T1[1]) * T1[1]) + (f[2] * T1[2]) * T1[2];
    real h0 = center[0] * center[0] - tmp0, h1 = center[1] * center[1] - tmp1;
    const real floor_ = (real)1e-4;
    extent[0] = R_SQRT(floor_ > h0 ? floor_ : h0);   /* max(1e-4, h): NaN h -> 1e-4 like CUDA fmaxf */
    extent[1] = R_SQRT(floor_ > h1 ? floor_ : h1);
    if (h0 != h0) extent[0] = R_SQRT(floor_);
    if (h1 != h1) extent[1] = R_SQRT(floor_);
    return 1;
}

/* auxiliary.h:68-78 */
static void get_rect(const real p[2], int max_radius, int gx, int gy, int rmin[2], int rmax[2])
{
    real r = (real)max_radius;
    rmin[0] = imin(gx, imax(0, f2i((p[0] - r) / (real)BLOCK_X)));
    rmin[1] = imin(gy, imax(0, f2i((p[1] - r) / (real)BLOCK_Y)));
    rmax[0] = imin(gx, imax(0, f2i((p[0] + r + (real)BLOCK_X - 1.0f) / (real)BLOCK_X)));
    rmax[1] = imin(gy, imax(0, f2i((p[1] + r + (real)BLOCK_Y - 1.0f) / (real)BLOCK_Y)));
}

/* forward.cu:22-73 */
static void sh_to_rgb(int idx, int deg, int M, const real *means, const real *campos, const real *shs,
                      uint8_t *clamped, real out[3])
{
    const real *pos = means + 3 * idx;
    real dx = pos[0] - campos[0], dy = pos[1] - campos[1], dz = pos[2] - campos[2];
    real len = R_SQRT(dx * dx + dy * dy + dz * dz);
    real x = dx / len, y = dy / len, z = dz / len;
    const real *sh = shs + (size_t)idx * M * 3;
    for (int c = 0; c < 3; c++) {
#define SH(i) sh[(i) * 3 + c]
        real r = SH_C0 * SH(0);
        if (deg > 0) {
            r = r - SH_C1 * y * SH(1) + SH_C1 * z * SH(2) - SH_C1 * x * SH(3);
            if (deg > 1) {
                real xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                r = r + SH_C2[0] * xy * SH(4) + SH_C2[1] * yz * SH(5) + SH_C2[2] * (2.0f * zz - xx - yy) * SH(6) +
                    SH_C2[3] * xz * SH(7) + SH_C2[4] * (xx - yy) * SH(8);
                if (deg > 2) {
                    r = r + SH_C3[0] * y * (3.0f * xx - yy) * SH(9) + SH_C3[1] * xy * z * SH(10) +
                        SH_C3[2] * y * (4.0f * zz - xx - yy) * SH(11) +
                        SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SH(12) +
                        SH_C3[4] * x * (4.0f * zz - xx - yy) * SH(13) + SH_C3[5] * z * (xx - yy) * SH(14) +
                        SH_C3[6] * x * (xx - 3.0f * yy) * SH(15);
                }
            }
        }
#undef SH
        r += 0.5f;
        clamped[3 * idx + c] = (r < 0);
        out[c] = r > 0.0f ? r : 0.0f;
    }
}

static int cmp_key(const void *a, const void *b)
{
    /* elements are {key, seq}: ordering by key then emission sequence == a stable sort on the key */
    const uint64_t *x = (const uint64_t *)a, *y = (const uint64_t *)b;
    if (x[0] != y[0]) return x[0] < y[0] ? -1 : 1;
    return x[1] < y[1] ? -1 : (x[1] > y[1]);
}

void mrgs_oracle_free(mrgs_oracle_ctx *c)
{
    if (!c) return;
    free(c->depths); free(c->means2D); free(c->transMat); free(c->normal_opacity); free(c->rgb);
    free(c->radii); free(c->tiles_touched); free(c->point_offsets); free(c->clamped);
    free(c->keys); free(c->point_list); free(c->ranges); free(c->final_T); free(c->n_contrib);
    free(c->out_color); free(c->out_feature); free(c->out_others);
    free(c);
}

/* preprocessCUDA, forward.cu:163-266 */
static void preprocess_fwd(mrgs_oracle_ctx *c)
{
    const int P = c->P;
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) {
        c->radii[idx] = 0;
        c->tiles_touched[idx] = 0;
        const real *p = c->means3D + 3 * idx;
        const real *V = c->view;
        /* in_frustum, auxiliary.h:192-217: only the view-space depth test is live */
        f3 pv = {V[0] * p[0] + V[4] * p[1] + V[8] * p[2] + V[12], V[1] * p[0] + V[5] * p[1] + V[9] * p[2] + V[13],
                 V[2] * p[0] + V[6] * p[1] + V[10] * p[2] + V[14]};
        if (pv.z <= 0.2f) continue;
        real T[9];
        f3 n;
        if (c->have_scale) {
            compute_transmat(p, c->scales + 2 * idx, c->scale_modifier, c->rotations + 4 * idx, c->proj, c->view, c->W,
                             c->H, T, &n);
            memcpy(c->transMat + 9 * idx, T, sizeof(T));
        } else {
            memcpy(T, c->transMat_precomp + 9 * idx, sizeof(T));
            n.x = 0.0f; n.y = 0.0f; n.z = 1.0f;
        }
        /* DUAL_VISIABLE, forward.cu:224-229 */
        real cosv = -((pv.x * n.x + pv.y * n.y) + pv.z * n.z);
        if (cosv == 0) continue;
        real mult = cosv > 0 ? 1.0f : -1.0f;
        n.x = mult * n.x; n.y = mult * n.y; n.z = mult * n.z;
        real center[2], extent[2];
        if (!compute_aabb(T, 3.0f, center, extent)) continue;
        real radius = R_CEIL(extent[0] > extent[1] ? extent[0] : extent[1]);
        int rmin[2], rmax[2];
        get_rect(center, f2i(radius), c->tiles_x, c->tiles_y, rmin, rmax);
        if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0) continue;
        if (c->have_sh) sh_to_rgb(idx, c->D, c->M, c->means3D, c->campos, c->shs, c->clamped, c->rgb + 3 * idx);
        c->depths[idx] = pv.z;
        c->radii[idx] = f2i(radius);
        c->means2D[2 * idx] = center[0];
        c->means2D[2 * idx + 1] = center[1];
        real *no = c->normal_opacity + 4 * idx;
        no[0] = n.x; no[1] = n.y; no[2] = n.z; no[3] = c->opacities[idx];
        c->tiles_touched[idx] = (uint32_t)((rmax[1] - rmin[1]) * (rmax[0] - rmin[0]));
    }
}

/* rasterizer_impl.cu:283-324: inclusive scan, duplicateWithKeys, stable sort on [tile|depth bits], tile ranges */
static int binning(mrgs_oracle_ctx *c)
{
    const int P = c->P;
    uint32_t run = 0;
    for (int i = 0; i < P; i++) { run += c->tiles_touched[i]; c->point_offsets[i] = run; }
    c->R = P > 0 ? (int64_t)run : 0;
    const int64_t R = c->R;
    uint64_t *ks = (uint64_t *)malloc(sizeof(uint64_t) * 2 * (size_t)(R > 0 ? R : 1));
    c->keys = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(R > 0 ? R : 1));
    c->point_list = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(R > 0 ? R : 1));
    uint32_t *vals = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(R > 0 ? R : 1));
    if (!ks || !c->keys || !c->point_list || !vals) { free(ks); free(vals); return -1; }
    for (int idx = 0; idx < P; idx++) {
        if (c->radii[idx] <= 0) continue;
        uint32_t off = idx == 0 ? 0 : c->point_offsets[idx - 1];
        int rmin[2], rmax[2];
        get_rect(c->means2D + 2 * idx, c->radii[idx], c->tiles_x, c->tiles_y, rmin, rmax);
        uint32_t dbits;
        float d32 = (float)c->depths[idx];   /* the key field is 32 bits of a float (rasterizer_impl.cu:104-106) */
        memcpy(&dbits, &d32, 4);
        for (int y = rmin[1]; y < rmax[1]; y++)
            for (int x = rmin[0]; x < rmax[0]; x++) {
                uint64_t key = (uint64_t)(y * c->tiles_x + x);
                key <<= 32;
                key |= dbits;
                ks[2 * (size_t)off] = key;
                ks[2 * (size_t)off + 1] = off;
                vals[off] = (uint32_t)idx;
                off++;
            }
    }
    qsort(ks, (size_t)R, 2 * sizeof(uint64_t), cmp_key);
    for (int64_t i = 0; i < R; i++) { c->keys[i] = ks[2 * i]; c->point_list[i] = vals[ks[2 * i + 1]]; }
    free(ks); free(vals);
    memset(c->ranges, 0, sizeof(uint32_t) * 2 * (size_t)c->tiles_x * c->tiles_y);
    for (int64_t i = 0; i < R; i++) {   /* identifyTileRanges, rasterizer_impl.cu:118-140 */
        uint32_t cur = (uint32_t)(c->keys[i] >> 32);
        if (i == 0) c->ranges[2 * cur] = 0;
        else {
            uint32_t prev = (uint32_t)(c->keys[i - 1] >> 32);
            if (cur != prev) { c->ranges[2 * prev + 1] = (uint32_t)i; c->ranges[2 * cur] = (uint32_t)i; }
        }
        if (i == R - 1) c->ranges[2 * cur + 1] = (uint32_t)R;
    }
    return 0;
}

/* ray-splat intersection shared by forward.cu:366-404 and backward.cu:296-328.
 * returns 0 when the pair is skipped before the alpha test */
typedef struct { real sx, sy, pz, inv_pz, rho3d, rho2d, depth, dx, dy, G, alpha; f3 k, l; } hit_t;
#ifdef MRGS_ORACLE_LITERAL
/* the reference's expression trees, operation by operation (forward.cu:366-404 == backward.cu:296-328) */
static int intersect(const real *T, const real *xy, real opa, real px, real py, hit_t *h)
{
    const real *Tu = T, *Tv = T + 3, *Tw = T + 6;
    f3 k = {px * Tw[0] - Tu[0], px * Tw[1] - Tu[1], px * Tw[2] - Tu[2]};                 /* pix.x * Tw - Tu */
    f3 l = {py * Tw[0] - Tv[0], py * Tw[1] - Tv[1], py * Tw[2] - Tv[2]};                 /* pix.y * Tw - Tv */
    f3 p = {k.y * l.z - k.z * l.y, k.z * l.x - k.x * l.z, k.x * l.y - k.y * l.x};        /* cross, auxiliary.h:162 */
    if (p.z == 0.0) return 0;
    h->k = k; h->l = l; h->pz = p.z;
    h->inv_pz = 0;   /* unused in this mode: the reference divides */
    h->sx = p.x / p.z; h->sy = p.y / p.z;
    h->rho3d = (h->sx * h->sx + h->sy * h->sy);
    h->dx = xy[0] - px; h->dy = xy[1] - py;
    h->rho2d = FILTER_INV_SQUARE * (h->dx * h->dx + h->dy * h->dy);
    real rho = R_FMIN(h->rho3d, h->rho2d);
    h->depth = (h->rho3d <= h->rho2d) ? (h->sx * Tw[0] + h->sy * Tw[1]) + Tw[2] : Tw[2];
    if (h->depth < NEAR_N) return 0;
    real power = -0.5f * rho;
    if (power > 0.0f) return 0;
    h->G = R_EXP(power);
    h->alpha = R_FMIN(0.99f, opa * h->G);
#ifndef MRGS_ORACLE_NO_ALPHA_CUTOFF
    if (h->alpha < 1.0f / 255.0f) return 0;
#endif
    return 1;
}
#else
static int intersect(const real *T, const real *xy, real opa, real px, real py, hit_t *h)
{
    const real *Tu = T, *Tv = T + 3, *Tw = T + 6;
    f3 k = {fmaf(px, Tw[0], -Tu[0]), fmaf(px, Tw[1], -Tu[1]), fmaf(px, Tw[2], -Tu[2])};
    f3 l = {fmaf(py, Tw[0], -Tv[0]), fmaf(py, Tw[1], -Tv[1]), fmaf(py, Tw[2], -Tv[2])};
    f3 p = {fmaf(k.y, l.z, -(k.z * l.y)), fmaf(k.z, l.x, -(k.x * l.z)), fmaf(k.x, l.y, -(k.y * l.x))};   /* cross(k, l) */
    if (p.z == 0.0f) return 0;
    h->k = k; h->l = l; h->pz = p.z;
    h->inv_pz = 1.0f / p.z;
    h->sx = p.x * h->inv_pz; h->sy = p.y * h->inv_pz;
    h->rho3d = fmaf(h->sx, h->sx, h->sy * h->sy);
    h->dx = xy[0] - px; h->dy = xy[1] - py;
    h->rho2d = FILTER_INV_SQUARE * fmaf(h->dx, h->dx, h->dy * h->dy);
    real rho = R_FMIN(h->rho3d, h->rho2d);   /* CUDA min(float,float) == fminf */
    h->depth = (h->rho3d <= h->rho2d) ? fmaf(h->sx, Tw[0], fmaf(h->sy, Tw[1], Tw[2])) : Tw[2];
    if (h->depth < NEAR_N) return 0;
    real power = -0.5f * rho;
    if (power > 0.0f) return 0;
    h->G = R_EXP(power);
    h->alpha = R_FMIN(0.99f, opa * h->G);
#ifndef MRGS_ORACLE_NO_ALPHA_CUTOFF   /* test-only build: see tests/test_oracle.py::test_smooth_part_is_exact_derivative */
    if (h->alpha < 1.0f / 255.0f) return 0;
#endif
    return 1;
}
#endif

/* renderCUDA, forward.cu:272-463 (one pixel at a time: a pixel's result does not depend on its neighbours) */
static void render_fwd(mrgs_oracle_ctx *c)
{
    const int H = c->H, W = c->W, S = c->S, HW = H * W;
    const real *colors = c->have_sh ? c->rgb : c->colors_precomp;
    const real *Ts = c->have_scale ? c->transMat : c->transMat_precomp;
    const real mscale = FAR_N / (FAR_N - NEAR_N);
#pragma omp parallel for schedule(dynamic, 4)
    for (int tile = 0; tile < c->tiles_x * c->tiles_y; tile++) {
        const int tx = tile % c->tiles_x, ty = tile / c->tiles_x;
        const uint32_t r0 = c->ranges[2 * tile], r1 = c->ranges[2 * tile + 1];
        for (int ly = 0; ly < BLOCK_Y; ly++)
            for (int lx = 0; lx < BLOCK_X; lx++) {
                const int pxi = tx * BLOCK_X + lx, pyi = ty * BLOCK_Y + ly;
                if (pxi >= W || pyi >= H) continue;
                const int pix = W * pyi + pxi;
                const real px = (real)pxi, py = (real)pyi;
                real T = 1.0f, C[3] = {0, 0, 0}, F[MAX_FEATURES] = {0}, N[3] = {0, 0, 0};
                real Dp = 0, M1 = 0, M2 = 0, distortion = 0, median_depth = 0;
                uint32_t contributor = 0, last_contributor = 0, median_contributor = 0;
                for (uint32_t i = r0; i < r1; i++) {
                    contributor++;
                    const uint32_t g = c->point_list[i];
                    const real *no = c->normal_opacity + 4 * (size_t)g;
                    hit_t h;
                    if (!intersect(Ts + 9 * (size_t)g, c->means2D + 2 * (size_t)g, no[3], px, py, &h)) continue;
                    real test_T = T * (1.0f - h.alpha);
                    if (test_T < 0.0001f) break;   /* done = true */
                    real w = h.alpha * T;
#ifdef MRGS_ORACLE_LITERAL
                    {   /* forward.cu:406-431, as written */
                        real A = 1 - T;
                        real m = FAR_N / (FAR_N - NEAR_N) * (1 - NEAR_N / h.depth);
                        distortion += (m * m * A + M2 - 2 * m * M1) * w;
                        Dp += h.depth * w;
                        M1 += m * w;
                        M2 += m * m * w;
                        if (T > 0.5) { median_depth = h.depth; median_contributor = contributor; }
                        for (int ch = 0; ch < 3; ch++) N[ch] += no[ch] * w;
                        for (int ch = 0; ch < 3; ch++) C[ch] += colors[3 * (size_t)g + ch] * w;
                        for (int ch = 0; ch < S; ch++) F[ch] += c->features[(size_t)g * S + ch] * w;
                        (void)mscale;
                    }
#else
                    real A = 1.0f - T;
                    real m = mscale * (1.0f - NEAR_N * (1.0f / h.depth));   /* deviation: near * (1/depth) for near / depth */
                    real mm = m * m;
                    distortion = fmaf(fmaf(-2.0f * m, M1, fmaf(mm, A, M2)), w, distortion);   /* (m*m*A + M2 - 2*m*M1) * w */
                    Dp = fmaf(h.depth, w, Dp);
                    M1 = fmaf(m, w, M1);
                    M2 = fmaf(mm, w, M2);
                    if (T > 0.5f) { median_depth = h.depth; median_contributor = contributor; }
                    for (int ch = 0; ch < 3; ch++) N[ch] = fmaf(no[ch], w, N[ch]);
                    for (int ch = 0; ch < 3; ch++) C[ch] = fmaf(colors[3 * (size_t)g + ch], w, C[ch]);
                    for (int ch = 0; ch < S; ch++) F[ch] = fmaf(c->features[(size_t)g * S + ch], w, F[ch]);
#endif
                    T = test_T;
                    last_contributor = contributor;
                }
                c->final_T[pix] = T;
                c->final_T[pix + HW] = M1;
                c->final_T[pix + 2 * HW] = M2;
                c->n_contrib[pix] = last_contributor;
                c->n_contrib[pix + HW] = median_contributor;   /* real -1 -> uint32 0 on the GPU (forward.cu:335,453) */
#ifdef MRGS_ORACLE_LITERAL
                for (int ch = 0; ch < 3; ch++) c->out_color[ch * HW + pix] = C[ch] + T * c->bg[ch];
#else
                for (int ch = 0; ch < 3; ch++) c->out_color[ch * HW + pix] = fmaf(T, c->bg[ch], C[ch]);
#endif
                for (int ch = 0; ch < S; ch++) c->out_feature[ch * HW + pix] = F[ch];
                c->out_others[pix + 0 * HW] = Dp;
                c->out_others[pix + 1 * HW] = 1.0f - T;
                for (int ch = 0; ch < 3; ch++) c->out_others[pix + (2 + ch) * HW] = N[ch];
                c->out_others[pix + 5 * HW] = median_depth;
                c->out_others[pix + 6 * HW] = distortion;
            }
    }
}

#define ALLOC(ptr, type, n) do { (ptr) = (type *)calloc((size_t)((n) > 0 ? (n) : 1), sizeof(type)); if (!(ptr)) goto fail; } while (0)

/* CudaRasterizer::Rasterizer::forward, rasterizer_impl.cu:200-349.
 * shs == NULL  <=> colours are precomputed;  scales == NULL <=> transMat_precomp is used.
 * Input pointers must stay valid until mrgs_oracle_free (backward reads them again). */
mrgs_oracle_ctx *mrgs_oracle_forward(int P, int S, int D, int M, int H, int W, const real *bg, const real *means3D,
                                     const real *shs, const real *colors_precomp, const real *features,
                                     const real *opacities, const real *scales, float scale_modifier,
                                     const real *rotations, const real *transMat_precomp, const real *viewmatrix,
                                     const real *projmatrix, const real *campos, float tanfovx, float tanfovy)
{
    mrgs_oracle_ctx *c = (mrgs_oracle_ctx *)calloc(1, sizeof(*c));
    if (!c) return NULL;
    if (S > MAX_FEATURES || S < 0) { free(c); return NULL; }
    c->P = P; c->S = S; c->D = D; c->M = M; c->H = H; c->W = W;
    c->tiles_x = (W + BLOCK_X - 1) / BLOCK_X;
    c->tiles_y = (H + BLOCK_Y - 1) / BLOCK_Y;
    c->have_sh = shs != NULL; c->have_scale = scales != NULL;
    c->tanfovx = tanfovx; c->tanfovy = tanfovy; c->scale_modifier = scale_modifier;
    memcpy(c->bg, bg, 3 * sizeof(real)); memcpy(c->view, viewmatrix, 16 * sizeof(real)); memcpy(c->proj, projmatrix, 16 * sizeof(real));
    memcpy(c->campos, campos, 3 * sizeof(real));
    c->means3D = means3D; c->shs = shs; c->colors_precomp = colors_precomp; c->features = features;
    c->opacities = opacities; c->scales = scales; c->rotations = rotations; c->transMat_precomp = transMat_precomp;
    const size_t HW = (size_t)H * W, NT = (size_t)c->tiles_x * c->tiles_y;
    ALLOC(c->depths, real, P); ALLOC(c->means2D, real, 2 * (size_t)P); ALLOC(c->transMat, real, 9 * (size_t)P);
    ALLOC(c->normal_opacity, real, 4 * (size_t)P); ALLOC(c->rgb, real, 3 * (size_t)P); ALLOC(c->radii, int, P);
    ALLOC(c->tiles_touched, uint32_t, P); ALLOC(c->point_offsets, uint32_t, P); ALLOC(c->clamped, uint8_t, 3 * (size_t)P);
    ALLOC(c->ranges, uint32_t, 2 * NT); ALLOC(c->final_T, real, 3 * HW); ALLOC(c->n_contrib, uint32_t, 2 * HW);
    ALLOC(c->out_color, real, 3 * HW); ALLOC(c->out_feature, real, (size_t)S * HW); ALLOC(c->out_others, real, 7 * HW);
    if (P > 0) {
        preprocess_fwd(c);
        if (binning(c) != 0) goto fail;
        render_fwd(c);
    }
    return c;
fail:
    mrgs_oracle_free(c);
    return NULL;
}

/* accessors for the ctypes wrapper */
int64_t mrgs_oracle_num_rendered(const mrgs_oracle_ctx *c) { return c->R; }
const void *mrgs_oracle_field(const mrgs_oracle_ctx *c, int which)
{
    switch (which) {
    case 0: return c->depths; case 1: return c->radii; case 2: return c->means2D; case 3: return c->transMat;
    case 4: return c->normal_opacity; case 5: return c->rgb; case 6: return c->tiles_touched; case 7: return c->clamped;
    case 8: return c->point_offsets; case 9: return c->keys; case 10: return c->point_list; case 11: return c->ranges;
    case 12: return c->final_T; case 13: return c->n_contrib; case 14: return c->out_color; case 15: return c->out_feature;
    case 16: return c->out_others;
    default: return NULL;
    }
}

/* per-gaussian accumulators of the render backward (the reference's atomicAdd targets, backward.cu:350-465) */
typedef struct { double *color, *feature, *normal, *transMat, *mean2D, *opacity; } acc_t;

/* renderCUDA (backward), backward.cu:145-468, one pixel at a time */
static void render_bwd_pixel(const mrgs_oracle_ctx *c, int tile, int pxi, int pyi, const real *dL_dpix,
                             const real *dL_dpix_f, const real *dL_dothers, acc_t *acc)
{
    const int H = c->H, W = c->W, S = c->S, HW = H * W;
    const int pix = W * pyi + pxi;
    const real px = (real)pxi, py = (real)pyi;
    const real *colors = c->have_sh ? c->rgb : c->colors_precomp;
    const real *Ts = c->have_scale ? c->transMat : c->transMat_precomp;
    const uint32_t r0 = c->ranges[2 * tile], r1 = c->ranges[2 * tile + 1];
    const real T_final = c->final_T[pix];
    real T = T_final;
    uint32_t contributor = r1 - r0;
    const uint32_t last_contributor = c->n_contrib[pix];
    const int median_contributor = (int)c->n_contrib[pix + HW];
    real accum_rec[3] = {0, 0, 0}, accum_rec_f[MAX_FEATURES] = {0}, dL_dpixel[3], dL_dpixel_f[MAX_FEATURES];
    const real dL_ddepth = dL_dothers[0 * HW + pix], dL_daccum = dL_dothers[1 * HW + pix],
                dL_dreg = dL_dothers[6 * HW + pix], dL_dmedian_depth = dL_dothers[5 * HW + pix];
    real dL_dnormal2D[3];
    for (int i = 0; i < 3; i++) dL_dnormal2D[i] = dL_dothers[(2 + i) * HW + pix];
    real last_depth = 0, last_normal[3] = {0, 0, 0}, accum_depth_rec = 0, accum_alpha_rec = 0,
          accum_normal_rec[3] = {0, 0, 0};
    const real final_D = c->final_T[pix + HW], final_D2 = c->final_T[pix + 2 * HW], final_A = 1 - T_final;
    real last_dL_dT = 0;
    for (int i = 0; i < 3; i++) dL_dpixel[i] = dL_dpix[i * HW + pix];
    for (int i = 0; i < S; i++) dL_dpixel_f[i] = dL_dpix_f[i * HW + pix];
    real last_alpha = 0, last_color[3] = {0, 0, 0}, last_feature[MAX_FEATURES] = {0};
    const real mscale = FAR_N / (FAR_N - NEAR_N);
    const real dmd_scale = (FAR_N * NEAR_N) / (FAR_N - NEAR_N);
    (void)mscale; (void)dmd_scale;
    for (uint32_t ii = r1; ii > r0; ii--) {
        contributor--;
        if (contributor >= last_contributor) continue;
        const uint32_t g = c->point_list[ii - 1];
        const real *no = c->normal_opacity + 4 * (size_t)g;
        const real *Tm = Ts + 9 * (size_t)g;
        hit_t h;
        if (!intersect(Tm, c->means2D + 2 * (size_t)g, no[3], px, py, &h)) continue;
#ifdef MRGS_ORACLE_LITERAL
        const real alpha = h.alpha, G = h.G, c_d = h.depth;
        /* backward.cu:330-465, as written (atomicAdd targets -> the double accumulators) */
        T = T / (1.f - alpha);
        const real dchannel_dcolor = alpha * T;
        real dL_dalpha = 0.0f;
        for (int ch = 0; ch < 3; ch++) {
            const real col = colors[3 * (size_t)g + ch];
            accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
            last_color[ch] = col;
            const real dL_dchannel = dL_dpixel[ch];
            dL_dalpha += (col - accum_rec[ch]) * dL_dchannel;
            acc->color[3 * (size_t)g + ch] += (double)(dchannel_dcolor * dL_dchannel);
        }
        for (int ch = 0; ch < S; ch++) {
            const real feature = c->features[(size_t)g * S + ch];
            accum_rec_f[ch] = last_alpha * last_feature[ch] + (1.f - last_alpha) * accum_rec_f[ch];
            last_feature[ch] = feature;
            const real dL_dchannel_f = dL_dpixel_f[ch];
            dL_dalpha += (feature - accum_rec_f[ch]) * dL_dchannel_f;
            acc->feature[(size_t)g * S + ch] += (double)(dchannel_dcolor * dL_dchannel_f);
        }
        real dL_dz = 0.0f;
        real dL_dweight = 0;
        const real m_d = FAR_N / (FAR_N - NEAR_N) * (1 - NEAR_N / c_d);
        const real dmd_dd = (FAR_N * NEAR_N) / ((FAR_N - NEAR_N) * c_d * c_d);
        if (contributor == (uint32_t)(median_contributor - 1)) dL_dz += dL_dmedian_depth;
        dL_dweight += (final_D2 + m_d * m_d * final_A - 2 * m_d * final_D) * dL_dreg;
        dL_dalpha += dL_dweight - last_dL_dT;
        last_dL_dT = dL_dweight * alpha + (1 - alpha) * last_dL_dT;
        const real dL_dmd = 2.0f * (T * alpha) * (m_d * final_A - final_D) * dL_dreg;
        dL_dz += dL_dmd * dmd_dd;
        accum_depth_rec = last_alpha * last_depth + (1.f - last_alpha) * accum_depth_rec;
        last_depth = c_d;
        dL_dalpha += (c_d - accum_depth_rec) * dL_ddepth;
        accum_alpha_rec = (real)((double)last_alpha * 1.0 + (double)((1.f - last_alpha) * accum_alpha_rec));
        dL_dalpha += (1 - accum_alpha_rec) * dL_daccum;
        for (int ch = 0; ch < 3; ch++) {
            accum_normal_rec[ch] = last_alpha * last_normal[ch] + (1.f - last_alpha) * accum_normal_rec[ch];
            last_normal[ch] = no[ch];
            dL_dalpha += (no[ch] - accum_normal_rec[ch]) * dL_dnormal2D[ch];
            acc->normal[3 * (size_t)g + ch] += (double)(alpha * T * dL_dnormal2D[ch]);
        }
        dL_dalpha *= T;
        last_alpha = alpha;
        real bg_dot_dpixel = 0;
        for (int i = 0; i < 3; i++) bg_dot_dpixel += c->bg[i] * dL_dpixel[i];
        dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;
        const real dL_dG = no[3] * dL_dalpha;
        dL_dz += alpha * T * dL_ddepth;
        double *dT = acc->transMat + 9 * (size_t)g;
        if (h.rho3d <= h.rho2d) {
            const real *Tw = Tm + 6;
            const real dL_dsx = dL_dG * -G * h.sx + dL_dz * Tw[0];
            const real dL_dsy = dL_dG * -G * h.sy + dL_dz * Tw[1];
            const real dz_dTw[3] = {h.sx, h.sy, 1.0};
            const real dsx_pz = dL_dsx / h.pz;
            const real dsy_pz = dL_dsy / h.pz;
            const f3 dL_dp = {dsx_pz, dsy_pz, -(dsx_pz * h.sx + dsy_pz * h.sy)};
            const f3 k = h.k, l = h.l;
            const f3 dL_dk = {l.y * dL_dp.z - l.z * dL_dp.y, l.z * dL_dp.x - l.x * dL_dp.z, l.x * dL_dp.y - l.y * dL_dp.x};   /* cross(l, dL_dp) */
            const f3 dL_dl = {dL_dp.y * k.z - dL_dp.z * k.y, dL_dp.z * k.x - dL_dp.x * k.z, dL_dp.x * k.y - dL_dp.y * k.x};   /* cross(dL_dp, k) */
            dT[0] += (double)(-dL_dk.x); dT[1] += (double)(-dL_dk.y); dT[2] += (double)(-dL_dk.z);
            dT[3] += (double)(-dL_dl.x); dT[4] += (double)(-dL_dl.y); dT[5] += (double)(-dL_dl.z);
            dT[6] += (double)(px * dL_dk.x + py * dL_dl.x + dL_dz * dz_dTw[0]);
            dT[7] += (double)(px * dL_dk.y + py * dL_dl.y + dL_dz * dz_dTw[1]);
            dT[8] += (double)(px * dL_dk.z + py * dL_dl.z + dL_dz * dz_dTw[2]);
        } else {
            const real dG_ddelx = -G * FILTER_INV_SQUARE * h.dx;
            const real dG_ddely = -G * FILTER_INV_SQUARE * h.dy;
            acc->mean2D[2 * (size_t)g] += (double)(dL_dG * dG_ddelx);
            acc->mean2D[2 * (size_t)g + 1] += (double)(dL_dG * dG_ddely);
            dT[8] += (double)dL_dz;
        }
#else
        const real alpha = h.alpha, G = h.G, c_d = h.depth;
        const real inv_1ma = 1.0f / (1.0f - alpha);
        T = T * inv_1ma;                                    /* T / (1 - alpha), backward.cu:330 */
        const real w = alpha * T;                          /* dchannel_dcolor */
        const real one_m_la = 1.0f - last_alpha;
        real dL_dalpha = 0.0f;
        for (int ch = 0; ch < 3; ch++) {
            const real col = colors[3 * (size_t)g + ch];
            accum_rec[ch] = fmaf(last_alpha, last_color[ch], one_m_la * accum_rec[ch]);
            last_color[ch] = col;
            dL_dalpha = fmaf(col - accum_rec[ch], dL_dpixel[ch], dL_dalpha);
            acc->color[3 * (size_t)g + ch] += (double)(w * dL_dpixel[ch]);
        }
        for (int ch = 0; ch < S; ch++) {
            const real f = c->features[(size_t)g * S + ch];
            accum_rec_f[ch] = fmaf(last_alpha, last_feature[ch], one_m_la * accum_rec_f[ch]);
            last_feature[ch] = f;
            dL_dalpha = fmaf(f - accum_rec_f[ch], dL_dpixel_f[ch], dL_dalpha);
            acc->feature[(size_t)g * S + ch] += (double)(w * dL_dpixel_f[ch]);
        }
        const real inv_cd = 1.0f / c_d;
        const real m_d = mscale * (1.0f - NEAR_N * inv_cd);
        const real dmd_dd = dmd_scale * inv_cd * inv_cd;   /* (far*near) / ((far-near) * c_d * c_d) */
        real dL_dz = (contributor == (uint32_t)(median_contributor - 1)) ? dL_dmedian_depth : 0.0f;
        const real dL_dweight = fmaf(-2.0f * m_d, final_D, fmaf(m_d * m_d, final_A, final_D2)) * dL_dreg;
        dL_dalpha += dL_dweight - last_dL_dT;
        last_dL_dT = fmaf(dL_dweight, alpha, (1.0f - alpha) * last_dL_dT);
        const real dL_dmd = 2.0f * w * fmaf(m_d, final_A, -final_D) * dL_dreg;
        dL_dz = fmaf(dL_dmd, dmd_dd, dL_dz);
        accum_depth_rec = fmaf(last_alpha, last_depth, one_m_la * accum_depth_rec);
        last_depth = c_d;
        dL_dalpha = fmaf(c_d - accum_depth_rec, dL_ddepth, dL_dalpha);
        accum_alpha_rec = fmaf(one_m_la, accum_alpha_rec, last_alpha);
        dL_dalpha = fmaf(1.0f - accum_alpha_rec, dL_daccum, dL_dalpha);
        for (int ch = 0; ch < 3; ch++) {
            accum_normal_rec[ch] = fmaf(last_alpha, last_normal[ch], one_m_la * accum_normal_rec[ch]);
            last_normal[ch] = no[ch];
            dL_dalpha = fmaf(no[ch] - accum_normal_rec[ch], dL_dnormal2D[ch], dL_dalpha);
            acc->normal[3 * (size_t)g + ch] += (double)(w * dL_dnormal2D[ch]);
        }
        dL_dalpha *= T;
        last_alpha = alpha;
        const real bg_dot_dpixel = fmaf(c->bg[2], dL_dpixel[2], fmaf(c->bg[1], dL_dpixel[1], c->bg[0] * dL_dpixel[0]));
        dL_dalpha = fmaf(-T_final * inv_1ma, bg_dot_dpixel, dL_dalpha);
        const real dL_dG = no[3] * dL_dalpha;
        dL_dz = fmaf(w, dL_ddepth, dL_dz);
        double *dT = acc->transMat + 9 * (size_t)g;
        if (h.rho3d <= h.rho2d) {
            const real *Tw = Tm + 6;
            const real dGn = dL_dG * -G;
            const real dL_dsx = fmaf(dGn, h.sx, dL_dz * Tw[0]);
            const real dL_dsy = fmaf(dGn, h.sy, dL_dz * Tw[1]);
            const real dpx = dL_dsx * h.inv_pz, dpy = dL_dsy * h.inv_pz;
            const real dpz = -fmaf(dpx, h.sx, dpy * h.sy);
            const f3 k = h.k, l = h.l;
            const f3 dL_dk = {fmaf(l.y, dpz, -(l.z * dpy)), fmaf(l.z, dpx, -(l.x * dpz)), fmaf(l.x, dpy, -(l.y * dpx))};
            const f3 dL_dl = {fmaf(dpy, k.z, -(dpz * k.y)), fmaf(dpz, k.x, -(dpx * k.z)), fmaf(dpx, k.y, -(dpy * k.x))};
            dT[0] += (double)(-dL_dk.x); dT[1] += (double)(-dL_dk.y); dT[2] += (double)(-dL_dk.z);
            dT[3] += (double)(-dL_dl.x); dT[4] += (double)(-dL_dl.y); dT[5] += (double)(-dL_dl.z);
            dT[6] += (double)fmaf(px, dL_dk.x, fmaf(py, dL_dl.x, dL_dz * h.sx));
            dT[7] += (double)fmaf(px, dL_dk.y, fmaf(py, dL_dl.y, dL_dz * h.sy));
            dT[8] += (double)fmaf(px, dL_dk.z, fmaf(py, dL_dl.z, dL_dz));
        } else {
            const real dGf = -G * FILTER_INV_SQUARE;
            acc->mean2D[2 * (size_t)g] += (double)(dL_dG * (dGf * h.dx));
            acc->mean2D[2 * (size_t)g + 1] += (double)(dL_dG * (dGf * h.dy));
            dT[8] += (double)dL_dz;
        }
#endif
        acc->opacity[g] += (double)(G * dL_dalpha);
    }
}

/* auxiliary.h:129-139 */
static f3 dnormvdv(f3 v, f3 dv)
{
    real sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
    real invsum32 = 1.0f / R_SQRT(sum2 * sum2 * sum2);
    f3 r;
    r.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;
    r.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32;
    r.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32;
    return r;
}

/* computeColorFromSH (backward), backward.cu:22-141 */
static void sh_backward(int idx, int deg, int M, const real *means, const real *campos, const real *shs,
                        const uint8_t *clamped, const real *dL_dcolor, real *dL_dmeans, real *dL_dshs)
{
    const real *pos = means + 3 * idx;
    f3 dir_orig = {pos[0] - campos[0], pos[1] - campos[1], pos[2] - campos[2]};
    real len = R_SQRT(dir_orig.x * dir_orig.x + dir_orig.y * dir_orig.y + dir_orig.z * dir_orig.z);
    real x = dir_orig.x / len, y = dir_orig.y / len, z = dir_orig.z / len;
    const real *sh = shs + (size_t)idx * M * 3;
    real *dsh = dL_dshs + (size_t)idx * M * 3;
    real dRGB[3];
    for (int c = 0; c < 3; c++) dRGB[c] = dL_dcolor[3 * idx + c] * (clamped[3 * idx + c] ? 0.0f : 1.0f);
    real ddir[3] = {0, 0, 0};   /* dL_ddir accumulated channel by channel = glm::dot(dRGBd*, dL_dRGB) */
    real dx[3], dy[3], dz[3];
    for (int c = 0; c < 3; c++) {
#define SH(i) sh[(i) * 3 + c]
#define DSH(i) dsh[(i) * 3 + c]
        real dRGBdx = 0, dRGBdy = 0, dRGBdz = 0;
        DSH(0) = SH_C0 * dRGB[c];
        if (deg > 0) {
            DSH(1) = (-SH_C1 * y) * dRGB[c];
            DSH(2) = (SH_C1 * z) * dRGB[c];
            DSH(3) = (-SH_C1 * x) * dRGB[c];
            dRGBdx = -SH_C1 * SH(3);
            dRGBdy = -SH_C1 * SH(1);
            dRGBdz = SH_C1 * SH(2);
            if (deg > 1) {
                real xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                DSH(4) = (SH_C2[0] * xy) * dRGB[c];
                DSH(5) = (SH_C2[1] * yz) * dRGB[c];
                DSH(6) = (SH_C2[2] * (2.f * zz - xx - yy)) * dRGB[c];
                DSH(7) = (SH_C2[3] * xz) * dRGB[c];
                DSH(8) = (SH_C2[4] * (xx - yy)) * dRGB[c];
                dRGBdx += SH_C2[0] * y * SH(4) + SH_C2[2] * 2.f * -x * SH(6) + SH_C2[3] * z * SH(7) + SH_C2[4] * 2.f * x * SH(8);
                dRGBdy += SH_C2[0] * x * SH(4) + SH_C2[1] * z * SH(5) + SH_C2[2] * 2.f * -y * SH(6) + SH_C2[4] * 2.f * -y * SH(8);
                dRGBdz += SH_C2[1] * y * SH(5) + SH_C2[2] * 2.f * 2.f * z * SH(6) + SH_C2[3] * x * SH(7);
                if (deg > 2) {
                    DSH(9) = (SH_C3[0] * y * (3.f * xx - yy)) * dRGB[c];
                    DSH(10) = (SH_C3[1] * xy * z) * dRGB[c];
                    DSH(11) = (SH_C3[2] * y * (4.f * zz - xx - yy)) * dRGB[c];
                    DSH(12) = (SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy)) * dRGB[c];
                    DSH(13) = (SH_C3[4] * x * (4.f * zz - xx - yy)) * dRGB[c];
                    DSH(14) = (SH_C3[5] * z * (xx - yy)) * dRGB[c];
                    DSH(15) = (SH_C3[6] * x * (xx - 3.f * yy)) * dRGB[c];
                    dRGBdx += (SH_C3[0] * SH(9) * 3.f * 2.f * xy + SH_C3[1] * SH(10) * yz + SH_C3[2] * SH(11) * -2.f * xy +
                               SH_C3[3] * SH(12) * -3.f * 2.f * xz + SH_C3[4] * SH(13) * (-3.f * xx + 4.f * zz - yy) +
                               SH_C3[5] * SH(14) * 2.f * xz + SH_C3[6] * SH(15) * 3.f * (xx - yy));
                    dRGBdy += (SH_C3[0] * SH(9) * 3.f * (xx - yy) + SH_C3[1] * SH(10) * xz +
                               SH_C3[2] * SH(11) * (-3.f * yy + 4.f * zz - xx) + SH_C3[3] * SH(12) * -3.f * 2.f * yz +
                               SH_C3[4] * SH(13) * -2.f * xy + SH_C3[5] * SH(14) * -2.f * yz +
                               SH_C3[6] * SH(15) * -3.f * 2.f * xy);
                    dRGBdz += (SH_C3[1] * SH(10) * xy + SH_C3[2] * SH(11) * 4.f * 2.f * yz +
                               SH_C3[3] * SH(12) * 3.f * (2.f * zz - xx - yy) + SH_C3[4] * SH(13) * 4.f * 2.f * xz +
                               SH_C3[5] * SH(14) * (xx - yy));
                }
            }
        }
#undef SH
#undef DSH
        dx[c] = dRGBdx; dy[c] = dRGBdy; dz[c] = dRGBdz;
    }
    ddir[0] = (dx[0] * dRGB[0] + dx[1] * dRGB[1]) + dx[2] * dRGB[2];
    ddir[1] = (dy[0] * dRGB[0] + dy[1] * dRGB[1]) + dy[2] * dRGB[2];
    ddir[2] = (dz[0] * dRGB[0] + dz[1] * dRGB[1]) + dz[2] * dRGB[2];
    f3 dd = {ddir[0], ddir[1], ddir[2]};
    f3 dm = dnormvdv(dir_orig, dd);
    dL_dmeans[3 * idx + 0] += dm.x;
    dL_dmeans[3 * idx + 1] += dm.y;
    dL_dmeans[3 * idx + 2] += dm.z;
}

/* auxiliary.h:245-289.  v_R is column-major vR[c][r] */
static void quat_to_rotmat_vjp(const real *q, real vR[3][3], real out[4])
{
    real s = 1.0f / R_SQRT(q[3] * q[3] + q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    real w = q[0] * s, x = q[1] * s, y = q[2] * s, z = q[3] * s;
    out[0] = 2.f * (x * (vR[1][2] - vR[2][1]) + y * (vR[2][0] - vR[0][2]) + z * (vR[0][1] - vR[1][0]));
    out[1] = 2.f * (-2.f * x * (vR[1][1] + vR[2][2]) + y * (vR[0][1] + vR[1][0]) + z * (vR[0][2] + vR[2][0]) +
                    w * (vR[1][2] - vR[2][1]));
    out[2] = 2.f * (x * (vR[0][1] + vR[1][0]) - 2.f * y * (vR[0][0] + vR[2][2]) + z * (vR[1][2] + vR[2][1]) +
                    w * (vR[2][0] - vR[0][2]));
    out[3] = 2.f * (x * (vR[0][2] + vR[2][0]) + y * (vR[1][2] + vR[2][1]) - 2.f * z * (vR[0][0] + vR[1][1]) +
                    w * (vR[0][1] - vR[1][0]));
}

/* preprocessCUDA (backward) + compute_transmat_aabb, backward.cu:471-669 */
static void preprocess_bwd(const mrgs_oracle_ctx *c, real *dL_dtransMat, const real *dL_dnormal, real *dL_dmean2D,
                           const real *dL_dcolors, real *dL_dsh, real *dL_dmean3D, real *dL_dscale, real *dL_drot)
{
    const int P = c->P;
    const real focal_y = c->H / (2.0f * c->tanfovy), focal_x = c->W / (2.0f * c->tanfovx);  /* rasterizer_impl.cu:398-399 */
    const int W = f2i(focal_x * c->tanfovx * 2), H = f2i(focal_y * c->tanfovy * 2);          /* backward.cu:646-647 */
    const real *Ts = c->have_scale ? c->transMat : c->transMat_precomp;
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) {
        if (!(c->radii[idx] > 0)) continue;
        real T[9], Pm[4][3], R[3][3];
        f3 normal = {0, 0, 0};
        const real *p = c->means3D + 3 * idx;
        real scale[2] = {0, 0};
        const int precomp = !c->have_scale;
        if (precomp) {
            memcpy(T, c->transMat_precomp + 9 * idx, sizeof(T));
        } else {
            scale[0] = c->scales[2 * idx]; scale[1] = c->scales[2 * idx + 1];
            quat_to_rotmat(c->rotations + 4 * idx, R);
            real sx = 1.0f * scale[0], sy = 1.0f * scale[1];   /* scale_to_mat(scale, 1.0f): scale_modifier ignored */
            real L0[3], L1[3], L2[3];
            for (int r = 0; r < 3; r++) {
                L0[r] = R[0][r] * sx + R[1][r] * 0.0f + R[2][r] * 0.0f;
                L1[r] = R[0][r] * 0.0f + R[1][r] * sy + R[2][r] * 0.0f;
                L2[r] = R[0][r] * 0.0f + R[1][r] * 0.0f + R[2][r] * 1.0f;
            }
            real hw = (real)((double)(real)W / 2.0), ow = (real)((double)(real)(W - 1) / 2.0);
            real hh = (real)((double)(real)H / 2.0), oh = (real)((double)(real)(H - 1) / 2.0);
            const real *pm = c->proj;
            for (int r = 0; r < 4; r++) {   /* P = world2ndc * ndc2pix, backward.cu:531 */
                Pm[r][0] = pm[4 * r + 0] * hw + pm[4 * r + 1] * 0.0f + pm[4 * r + 2] * 0.0f + pm[4 * r + 3] * ow;
                Pm[r][1] = pm[4 * r + 0] * 0.0f + pm[4 * r + 1] * hh + pm[4 * r + 2] * 0.0f + pm[4 * r + 3] * oh;
                Pm[r][2] = pm[4 * r + 0] * 0.0f + pm[4 * r + 1] * 0.0f + pm[4 * r + 2] * 0.0f + pm[4 * r + 3] * 1.0f;
            }
            real a[3][4] = {{L0[0], L0[1], L0[2], 0.0f}, {L1[0], L1[1], L1[2], 0.0f}, {p[0], p[1], p[2], 1.0f}};
            for (int i = 0; i < 3; i++)       /* T = transpose(M) * P, backward.cu:532 */
                for (int j = 0; j < 3; j++)
                    T[3 * j + i] = a[i][0] * Pm[0][j] + a[i][1] * Pm[1][j] + a[i][2] * Pm[2][j] + a[i][3] * Pm[3][j];
            const real *V = c->view;
            normal.x = V[0] * L2[0] + V[4] * L2[1] + V[8] * L2[2];
            normal.y = V[1] * L2[0] + V[5] * L2[1] + V[9] * L2[2];
            normal.z = V[2] * L2[0] + V[6] * L2[1] + V[10] * L2[2];
        }
        real dT[9];   /* dT[0..2] = d/dTu, [3..5] = d/dTv, [6..8] = d/dTw */
        memcpy(dT, dL_dtransMat + 9 * idx, sizeof(dT));
        const real m2x = dL_dmean2D[3 * idx], m2y = dL_dmean2D[3 * idx + 1];
        int returned = 0;
        if (m2x != 0 || m2y != 0) {   /* backward.cu:543-582 */
            const real *Tu = T, *Tv = T + 3, *Tw = T + 6;
            const real distance = Tw[0] * Tw[0] + Tw[1] * Tw[1] - Tw[2] * Tw[2];
            const real f = 1 / distance;
            const real dpx_dT00 = f * Tw[0], dpx_dT01 = f * Tw[1], dpx_dT02 = -f * Tw[2];
            const real dpy_dT10 = f * Tw[0], dpy_dT11 = f * Tw[1], dpy_dT12 = -f * Tw[2];
            const real dpx_dT30 = Tu[0] * (f - 2 * f * f * Tw[0] * Tw[0]);
            const real dpx_dT31 = Tu[1] * (f - 2 * f * f * Tw[1] * Tw[1]);
            const real dpx_dT32 = -Tu[2] * (f + 2 * f * f * Tw[2] * Tw[2]);
            const real dpy_dT30 = Tv[0] * (f - 2 * f * f * Tw[0] * Tw[0]);
            const real dpy_dT31 = Tv[1] * (f - 2 * f * f * Tw[1] * Tw[1]);
            const real dpy_dT32 = -Tv[2] * (f + 2 * f * f * Tw[2] * Tw[2]);
            dT[0] += m2x * dpx_dT00; dT[1] += m2x * dpx_dT01; dT[2] += m2x * dpx_dT02;
            dT[3] += m2y * dpy_dT10; dT[4] += m2y * dpy_dT11; dT[5] += m2y * dpy_dT12;
            dT[6] += m2x * dpx_dT30 + m2y * dpy_dT30;
            dT[7] += m2x * dpx_dT31 + m2y * dpy_dT31;
            dT[8] += m2x * dpx_dT32 + m2y * dpy_dT32;
            if (precomp) { memcpy(dL_dtransMat + 9 * idx, dT, sizeof(dT)); returned = 1; }
        }
        if (!precomp && !returned) {
            /* dL_dM = P * transpose(dL_dT): dM[i][r] = sum_j Pm[r][j] * dT_j[i], backward.cu:587 */
            real dM[3][4];
            for (int i = 0; i < 3; i++)
                for (int r = 0; r < 4; r++)
                    dM[i][r] = Pm[r][0] * dT[0 + i] + Pm[r][1] * dT[3 + i] + Pm[r][2] * dT[6 + i];
            const real *V = c->view;
            const real *dn = dL_dnormal + 3 * idx;
            f3 dtn = {V[0] * dn[0] + V[1] * dn[1] + V[2] * dn[2], V[4] * dn[0] + V[5] * dn[1] + V[6] * dn[2],
                      V[8] * dn[0] + V[9] * dn[1] + V[10] * dn[2]};
            f3 pv = {V[0] * p[0] + V[4] * p[1] + V[8] * p[2] + V[12], V[1] * p[0] + V[5] * p[1] + V[9] * p[2] + V[13],
                     V[2] * p[0] + V[6] * p[1] + V[10] * p[2] + V[14]};
            real cosv = -((pv.x * normal.x + pv.y * normal.y) + pv.z * normal.z);
            real mult = cosv > 0 ? 1.0f : -1.0f;
            dtn.x = mult * dtn.x; dtn.y = mult * dtn.y; dtn.z = mult * dtn.z;
            real dRS[3][3] = {{dM[0][0], dM[0][1], dM[0][2]}, {dM[1][0], dM[1][1], dM[1][2]}, {dtn.x, dtn.y, dtn.z}};
            real dR[3][3];
            for (int r = 0; r < 3; r++) { dR[0][r] = dRS[0][r] * scale[0]; dR[1][r] = dRS[1][r] * scale[1]; dR[2][r] = dRS[2][r]; }
            quat_to_rotmat_vjp(c->rotations + 4 * idx, dR, dL_drot + 4 * idx);
            dL_dscale[2 * idx] = (dRS[0][0] * R[0][0] + dRS[0][1] * R[0][1]) + dRS[0][2] * R[0][2];
            dL_dscale[2 * idx + 1] = (dRS[1][0] * R[1][0] + dRS[1][1] * R[1][1]) + dRS[1][2] * R[1][2];
            dL_dmean3D[3 * idx] = dM[2][0]; dL_dmean3D[3 * idx + 1] = dM[2][1]; dL_dmean3D[3 * idx + 2] = dM[2][2];
        }
        if (c->have_sh)
            sh_backward(idx, c->D, c->M, c->means3D, c->campos, c->shs, c->clamped, dL_dcolors, dL_dmean3D, dL_dsh);
        /* densification proxy, backward.cu:665-668 */
        real depth = Ts[9 * idx + 8];
        dL_dmean2D[3 * idx] = (real)((double)(dL_dtransMat[9 * idx + 2] * depth) * 0.5 * (double)(real)W);
        dL_dmean2D[3 * idx + 1] = (real)((double)(dL_dtransMat[9 * idx + 5] * depth) * 0.5 * (double)(real)H);
    }
}

/* CudaRasterizer::Rasterizer::backward, rasterizer_impl.cu:353-462.  All outputs are written in full
 * (zero-filled first, as rasterize_points.cu:201-210 does). dL_dnormal is an internal tensor of the
 * reference; it is exposed here so the HIP kernel's intermediate can be checked too. */
int mrgs_oracle_backward(const mrgs_oracle_ctx *c, const real *dL_dpix, const real *dL_dpix_f, const real *dL_dothers,
                         real *dL_dmean2D /*[P,3]*/, real *dL_dnormal /*[P,3]*/, real *dL_dopacity /*[P]*/,
                         real *dL_dcolor /*[P,3]*/, real *dL_dfeature /*[P,S]*/, real *dL_dmean3D /*[P,3]*/,
                         real *dL_dtransMat /*[P,9]*/, real *dL_dsh /*[P,M,3]*/, real *dL_dscale /*[P,2]*/,
                         real *dL_drot /*[P,4]*/)
{
    const int P = c->P, S = c->S;
    const size_t nP = (size_t)(P > 0 ? P : 1);
    memset(dL_dmean2D, 0, sizeof(real) * 3 * (size_t)P); memset(dL_dnormal, 0, sizeof(real) * 3 * (size_t)P);
    memset(dL_dopacity, 0, sizeof(real) * (size_t)P); memset(dL_dcolor, 0, sizeof(real) * 3 * (size_t)P);
    memset(dL_dfeature, 0, sizeof(real) * (size_t)S * P); memset(dL_dmean3D, 0, sizeof(real) * 3 * (size_t)P);
    memset(dL_dtransMat, 0, sizeof(real) * 9 * (size_t)P); memset(dL_dsh, 0, sizeof(real) * 3 * (size_t)c->M * P);
    memset(dL_dscale, 0, sizeof(real) * 2 * (size_t)P); memset(dL_drot, 0, sizeof(real) * 4 * (size_t)P);
    if (P == 0) return 0;
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = omp_get_max_threads();
#endif
    const size_t per = nP * (size_t)(3 + S + 3 + 9 + 2 + 1);
    double *pool = (double *)calloc(per * (size_t)nthreads, sizeof(double));
    if (!pool) return -1;
#pragma omp parallel num_threads(nthreads)
    {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        double *b = pool + per * (size_t)tid;
        acc_t acc;
        acc.color = b; b += 3 * nP; acc.feature = b; b += (size_t)S * nP; acc.normal = b; b += 3 * nP;
        acc.transMat = b; b += 9 * nP; acc.mean2D = b; b += 2 * nP; acc.opacity = b;
#pragma omp for schedule(dynamic, 4)
        for (int tile = 0; tile < c->tiles_x * c->tiles_y; tile++) {
            const int tx = tile % c->tiles_x, ty = tile / c->tiles_x;
            for (int ly = 0; ly < BLOCK_Y; ly++)
                for (int lx = 0; lx < BLOCK_X; lx++) {
                    const int pxi = tx * BLOCK_X + lx, pyi = ty * BLOCK_Y + ly;
                    if (pxi >= c->W || pyi >= c->H) continue;
                    render_bwd_pixel(c, tile, pxi, pyi, dL_dpix, dL_dpix_f, dL_dothers, &acc);
                }
        }
    }
    /* fold the per-thread double accumulators (sum in double, round once) */
#pragma omp parallel for schedule(static)
    for (int g = 0; g < P; g++) {
        double s3[3] = {0, 0, 0}, n3[3] = {0, 0, 0}, t9[9] = {0}, m2[2] = {0, 0}, op = 0, fs[MAX_FEATURES] = {0};
        for (int t = 0; t < nthreads; t++) {
            const double *b = pool + per * (size_t)t;
            const double *color = b; b += 3 * nP; const double *feature = b; b += (size_t)S * nP;
            const double *normal = b; b += 3 * nP; const double *tm = b; b += 9 * nP; const double *m2d = b; b += 2 * nP;
            const double *opac = b;
            for (int i = 0; i < 3; i++) { s3[i] += color[3 * (size_t)g + i]; n3[i] += normal[3 * (size_t)g + i]; }
            for (int i = 0; i < S; i++) fs[i] += feature[(size_t)g * S + i];
            for (int i = 0; i < 9; i++) t9[i] += tm[9 * (size_t)g + i];
            m2[0] += m2d[2 * (size_t)g]; m2[1] += m2d[2 * (size_t)g + 1];
            op += opac[g];
        }
        for (int i = 0; i < 3; i++) { dL_dcolor[3 * g + i] = (real)s3[i]; dL_dnormal[3 * g + i] = (real)n3[i]; }
        for (int i = 0; i < S; i++) dL_dfeature[(size_t)g * S + i] = (real)fs[i];
        for (int i = 0; i < 9; i++) dL_dtransMat[9 * g + i] = (real)t9[i];
        dL_dmean2D[3 * g] = (real)m2[0]; dL_dmean2D[3 * g + 1] = (real)m2[1];
        dL_dopacity[g] = (real)op;
    }
    free(pool);
    preprocess_bwd(c, dL_dtransMat, dL_dnormal, dL_dmean2D, dL_dcolor, dL_dsh, dL_dmean3D, dL_dscale, dL_drot);
    return 0;
}

/* Test hook: BACKWARD::preprocess alone (backward.cu:614-669) on caller-supplied per-gaussian upstream gradients.
 * dL_dtransMat [P,9] and dL_dmean2D [P,3] are updated in place exactly as the reference does; the outputs are
 * zero-filled first.  Used by tests/oracle_derivative_probe.py to check the T -> (mean, scale, rotation) chain. */
void mrgs_oracle_preprocess_backward_only(const mrgs_oracle_ctx *c, real *dL_dtransMat, const real *dL_dnormal,
                                          real *dL_dmean2D, const real *dL_dcolors, real *dL_dsh, real *dL_dmean3D,
                                          real *dL_dscale, real *dL_drot)
{
    const int P = c->P;
    memset(dL_dsh, 0, sizeof(real) * 3 * (size_t)c->M * P); memset(dL_dmean3D, 0, sizeof(real) * 3 * (size_t)P);
    memset(dL_dscale, 0, sizeof(real) * 2 * (size_t)P); memset(dL_drot, 0, sizeof(real) * 4 * (size_t)P);
    preprocess_bwd(c, dL_dtransMat, dL_dnormal, dL_dmean2D, dL_dcolors, dL_dsh, dL_dmean3D, dL_dscale, dL_drot);
}

/* markVisible / checkFrustum, rasterizer_impl.cu:56-68,143-155 */
void mrgs_oracle_mark_visible(int P, const real *means3D, const real *viewmatrix, const real *projmatrix,
                              uint8_t *present)
{
    (void)projmatrix;
    const real *V = viewmatrix;
    for (int i = 0; i < P; i++) {
        const real *p = means3D + 3 * i;
        real z = V[2] * p[0] + V[6] * p[1] + V[10] * p[2] + V[14];
        present[i] = !(z <= 0.2f);
    }
}

/* width of the arithmetic type of this build (4: float, 8: double) -- the ctypes wrapper sizes its arrays by it */
int mrgs_oracle_real_bytes(void) { return (int)sizeof(real); }

int mrgs_oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
