// mrgs_maps.hip -- per-pixel glue of the surfel renderer as fused kernels.
//
//  surfel_maps_*      <-> compute_2dgs_normal_and_regularizations (gaussian_renderer/__init__.py:42-90) with
//                         depths_to_points / depth_to_normal (utils/point_utils.py:9-37) and the normal_map of render_surfel
//                         (gaussian_renderer/__init__.py:419-421): view->world normals, expected / median depth with
//                         nan_to_num, surf_depth, finite-difference normals of the back-projected depth, normal / alpha.
//  surfel_composite_* <-> the compositing lines of render_surfel (:436-445): (1 - refl) * base + specular, optional
//                         linear_to_srgb (utils/graphics_utils.py:102-110), background.
// The reference runs ~40 torch kernels forward and ~80 backward for these on 800x800 maps; here it is one kernel each way,
// one pixel per lane, channel-first maps read and written coalesced.  The backward of the finite-difference normals is a
// gather (each pixel recomputes the four neighbouring normals it contributed to): deterministic, no atomics.
#include "mrgs_internal.h"

namespace {

struct MapsFrameDev {
    int H, W;
    float V[9];      // world_view_transform[:3,:3] as stored: n_world = V * n_view
    float M[9];      // rays_d(x, y) = M * (x, y, 1)
    float o[3];      // rays_o
    float depth_ratio;
    // "pgsr" flavour (MrgsMapsFrame::rend_distance set): surf_depth is the flavour's unbiased depth
    const float* rd;     // blended plane distance [H*W], or nullptr
    float* g_rd;         // backward: its gradient [H*W] (fully written)
    float fx, fy;        // focal lengths in pixels of the rasterizer's image plane (principal point ((W - 1) / 2, (H - 1) / 2))
};

// allmap[7] of the "pgsr" flavour (gaussian_renderer/__init__.py:64-69; PARITY UNPINNED, see renderer.pgsr_unbiased_depth for the
// definition): the depth at which the pixel's ray meets the blended plane, rend_distance / -(n . ray), n = allmap[2:5] (view space),
// ray = ((x - (W - 1) / 2) / fx, (y - (H - 1) / 2) / fy, 1).
struct Unbiased { float ray[3], ndr, v; };
__device__ __forceinline__ Unbiased unbiased_depth(const MapsFrameDev& f, const float* __restrict__ allmap, int HW, int pix)
{
    Unbiased u;
    const int y = pix / f.W, x = pix - y * f.W;
    u.ray[0] = ((float)x - 0.5f * (float)(f.W - 1)) / f.fx;
    u.ray[1] = ((float)y - 0.5f * (float)(f.H - 1)) / f.fy;
    u.ray[2] = 1.0f;
    u.ndr = allmap[2 * HW + pix] * u.ray[0] + allmap[3 * HW + pix] * u.ray[1] + allmap[4 * HW + pix];
    u.v = f.rd[pix] / (-u.ndr);
    return u;
}

__device__ __forceinline__ float nan_to_num0(float x)
{   // torch.nan_to_num(x, 0, 0): nan -> 0, +inf -> 0, -inf -> lowest finite
    if (x != x) return 0.0f;
    if (x == __builtin_inff()) return 0.0f;
    if (x == -__builtin_inff()) return -3.4028234663852886e38f;
    return x;
}
__device__ __forceinline__ bool is_finite(float x) { return (x - x) == 0.0f; }

__device__ __forceinline__ float surf_depth_at(const MapsFrameDev& f, const float* __restrict__ allmap, int HW, int pix)
{
    if (f.rd != nullptr) return nan_to_num0(unbiased_depth(f, allmap, HW, pix).v);       // (empty pixels: 0 / -0 -> 0, as the reference's nan_to_num)
    const float a = allmap[HW + pix];
    const float de = nan_to_num0(allmap[pix] / a);
    float sd = de * (1.0f - f.depth_ratio);
    if (f.depth_ratio != 0.0f) sd += f.depth_ratio * nan_to_num0(allmap[5 * HW + pix]);
    else sd += f.depth_ratio * 0.0f;
    return sd;
}
__device__ __forceinline__ void ray_dir(const MapsFrameDev& f, int x, int y, float (&d)[3])
{
    const float fx = (float)x, fy = (float)y;
#pragma unroll
    for (int i = 0; i < 3; i++) d[i] = f.M[3 * i] * fx + f.M[3 * i + 1] * fy + f.M[3 * i + 2];
}
__device__ __forceinline__ void point_at(const MapsFrameDev& f, const float* __restrict__ allmap, int HW, int x, int y, float (&p)[3])
{
    float d[3];
    ray_dir(f, x, y, d);
    const float sd = surf_depth_at(f, allmap, HW, y * f.W + x);
#pragma unroll
    for (int i = 0; i < 3; i++) p[i] = sd * d[i] + f.o[i];
}
__device__ __forceinline__ void cross3(const float (&a)[3], const float (&b)[3], float (&c)[3])
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

// finite-difference normal at an interior pixel: "dx" = rows difference, "dy" = columns difference (point_utils.py:33-35)
struct FdNormal { float dx[3], dy[3], n[3], len, nn[3]; };
__device__ __forceinline__ FdNormal fd_normal(const MapsFrameDev& f, const float* __restrict__ allmap, int HW, int x, int y)
{
    FdNormal r;
    float pa[3], pb[3];
    point_at(f, allmap, HW, x, y + 1, pa);
    point_at(f, allmap, HW, x, y - 1, pb);
#pragma unroll
    for (int i = 0; i < 3; i++) r.dx[i] = pa[i] - pb[i];
    point_at(f, allmap, HW, x + 1, y, pa);
    point_at(f, allmap, HW, x - 1, y, pb);
#pragma unroll
    for (int i = 0; i < 3; i++) r.dy[i] = pa[i] - pb[i];
    cross3(r.dx, r.dy, r.n);
    r.len = fmaxf(sqrtf(r.n[0] * r.n[0] + r.n[1] * r.n[1] + r.n[2] * r.n[2]), 1e-12f);   // F.normalize eps
#pragma unroll
    for (int i = 0; i < 3; i++) r.nn[i] = r.n[i] / r.len;
    return r;
}

__global__ void __launch_bounds__(256) surfel_maps_fwd_kernel(MapsFrameDev f, const float* __restrict__ allmap, float* __restrict__ rend_normal,
                                                              float* __restrict__ surf_depth, float* __restrict__ surf_normal,
                                                              float* __restrict__ normal_map, float* __restrict__ rend_alpha,
                                                              float* __restrict__ rend_dist, float* __restrict__ rend_alpha2)
{
    const int HW = f.H * f.W;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= HW) return;
    const int y = pix / f.W, x = pix - y * f.W;
    const float a = allmap[HW + pix];
    if (rend_alpha) rend_alpha[pix] = a;                       // the reference's plain slices allmap[1:2], allmap[6:7] as tensors of their own
    if (rend_alpha2) rend_alpha2[pix] = a;                     // (a second copy for a second consumer: its gradient comes back on its own pointer)
    if (rend_dist) rend_dist[pix] = allmap[6 * HW + pix];
    const float nv[3] = {allmap[2 * HW + pix], allmap[3 * HW + pix], allmap[4 * HW + pix]};
    float nw[3];
#pragma unroll
    for (int j = 0; j < 3; j++) nw[j] = f.V[3 * j] * nv[0] + f.V[3 * j + 1] * nv[1] + f.V[3 * j + 2] * nv[2];
#pragma unroll
    for (int j = 0; j < 3; j++) rend_normal[j * HW + pix] = nw[j];
    surf_depth[pix] = surf_depth_at(f, allmap, HW, pix);
    if (normal_map != nullptr) {
        const float inv = 1.0f / fmaxf(a, 1e-6f);
#pragma unroll
        for (int j = 0; j < 3; j++) normal_map[3 * (size_t)pix + j] = nw[j] * inv;
    }
    if (surf_normal != nullptr) {
        float sn[3] = {0.0f, 0.0f, 0.0f};
        if (x >= 1 && x < f.W - 1 && y >= 1 && y < f.H - 1) {
            const FdNormal r = fd_normal(f, allmap, HW, x, y);
#pragma unroll
            for (int i = 0; i < 3; i++) sn[i] = r.nn[i] * a;
        }
#pragma unroll
        for (int j = 0; j < 3; j++) surf_normal[j * HW + pix] = sn[j];
    }
}

// One 32x8 tile of pixels per workgroup.  A pixel's surf_depth feeds the finite-difference normals of its four neighbours, and each of
// those needs four surface points: evaluated per pixel that is 16 points (a division with its nan_to_num each) and four normalised cross
// products, of which neighbouring pixels repeat 15 / 3.  Here the tile computes every point once (tile + 2 pixels of halo, LDS), then
// every centre's (g_dx, g_dy) once (tile + 1 pixel of halo, LDS; zero where the centre is no interior pixel: adding zero is exact, so the
// sums below equal the per-pixel evaluation's term by term), then a pixel adds up its four neighbours' terms in the order it always did.
constexpr int MB_TX = 32, MB_TY = 8;
constexpr int MB_PW = MB_TX + 4, MB_PH = MB_TY + 4, MB_CW = MB_TX + 2, MB_CH = MB_TY + 2;
__global__ void __launch_bounds__(256) surfel_maps_bwd_kernel(MapsFrameDev f, const float* __restrict__ allmap, const float* __restrict__ g_rn,
                                                              const float* __restrict__ g_sd, const float* __restrict__ g_sn,
                                                              const float* __restrict__ g_nm, const float* __restrict__ g_alpha,
                                                              const float* __restrict__ g_dist, const float* __restrict__ g_alpha2,
                                                              float* __restrict__ g_allmap)
{
    __shared__ float s_pt[3][MB_PH * MB_PW];
    __shared__ float s_g[6][MB_CH * MB_CW];
    const int HW = f.H * f.W;
    const int x0 = blockIdx.x * MB_TX, y0 = blockIdx.y * MB_TY;
    const int tid = threadIdx.x;
    if (g_sn != nullptr) {
        for (int i = tid; i < MB_PH * MB_PW; i += 256) {
            const int px = x0 - 2 + i % MB_PW, py = y0 - 2 + i / MB_PW;
            float p[3] = {0.0f, 0.0f, 0.0f};
            if (px >= 0 && px < f.W && py >= 0 && py < f.H) point_at(f, allmap, HW, px, py, p);
            s_pt[0][i] = p[0]; s_pt[1][i] = p[1]; s_pt[2][i] = p[2];
        }
        __syncthreads();
        for (int i = tid; i < MB_CH * MB_CW; i += 256) {
            const int lx = i % MB_CW, ly = i / MB_CW;
            const int cx = x0 - 1 + lx, cy = y0 - 1 + ly;
            float gdx[3] = {0.0f, 0.0f, 0.0f}, gdy[3] = {0.0f, 0.0f, 0.0f};
            if (cx >= 1 && cx < f.W - 1 && cy >= 1 && cy < f.H - 1) {
                const int c = cy * f.W + cx;
                const int q = (ly + 1) * MB_PW + (lx + 1);              // the centre in the point tile
                FdNormal r;
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    r.dx[k] = s_pt[k][q + MB_PW] - s_pt[k][q - MB_PW];  // rows difference (fd_normal)
                    r.dy[k] = s_pt[k][q + 1] - s_pt[k][q - 1];
                }
                cross3(r.dx, r.dy, r.n);
                r.len = fmaxf(sqrtf(r.n[0] * r.n[0] + r.n[1] * r.n[1] + r.n[2] * r.n[2]), 1e-12f);
#pragma unroll
                for (int k = 0; k < 3; k++) r.nn[k] = r.n[k] / r.len;
                const float a = allmap[HW + c];
                const float g_nn[3] = {g_sn[c] * a, g_sn[HW + c] * a, g_sn[2 * HW + c] * a};
                const float dot = r.nn[0] * g_nn[0] + r.nn[1] * g_nn[1] + r.nn[2] * g_nn[2];
                float g_n[3];
#pragma unroll
                for (int k = 0; k < 3; k++) g_n[k] = (g_nn[k] - r.nn[k] * dot) / r.len;
                cross3(r.dy, g_n, gdx);      // n = dx x dy
                cross3(g_n, r.dx, gdy);
            }
#pragma unroll
            for (int k = 0; k < 3; k++) { s_g[k][i] = gdx[k]; s_g[3 + k][i] = gdy[k]; }
        }
        __syncthreads();
    }
    const int lx = tid % MB_TX, ly = tid / MB_TX;
    const int x = x0 + lx, y = y0 + ly;
    if (x >= f.W || y >= f.H) return;
    const int pix = y * f.W + x;
    const float a = allmap[HW + pix];
    const float nv[3] = {allmap[2 * HW + pix], allmap[3 * HW + pix], allmap[4 * HW + pix]};
    float g_nw[3] = {0.0f, 0.0f, 0.0f};
    float g_a = g_alpha != nullptr ? g_alpha[pix] : 0.0f;      // rend_alpha is a plain view of allmap[1]
    if (g_alpha2 != nullptr) g_a += g_alpha2[pix];
    if (g_rn != nullptr) {
#pragma unroll
        for (int j = 0; j < 3; j++) g_nw[j] = g_rn[j * HW + pix];
    }
    if (g_nm != nullptr) {
        const float ac = fmaxf(a, 1e-6f), inv = 1.0f / ac;
        float nw[3], s = 0.0f;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            nw[j] = f.V[3 * j] * nv[0] + f.V[3 * j + 1] * nv[1] + f.V[3 * j + 2] * nv[2];
            const float g = g_nm[3 * (size_t)pix + j];
            g_nw[j] += g * inv;
            s += nw[j] * g;
        }
        if (a >= 1e-6f) g_a -= s * inv * inv;            // clamp_min passes the gradient where x >= min
    }
    float g_nv[3];                                             // gradient of the view-space normal sums allmap[2:5]
#pragma unroll
    for (int i = 0; i < 3; i++) g_nv[i] = f.V[i] * g_nw[0] + f.V[3 + i] * g_nw[1] + f.V[6 + i] * g_nw[2];

    // surf_depth: own upstream gradient + the finite-difference normals of the four neighbours this pixel's point feeds
    float g_depth = g_sd != nullptr ? g_sd[pix] : 0.0f;
    if (g_sn != nullptr) {
        const int c = (ly + 1) * MB_CW + (lx + 1);                  // this pixel in the centre tile
        float gp[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < 3; i++) gp[i] += s_g[i][c - MB_CW];     // c = p - e_y: p is c's "row below" point (+dx)
#pragma unroll
        for (int i = 0; i < 3; i++) gp[i] -= s_g[i][c + MB_CW];     // c = p + e_y: -dx
#pragma unroll
        for (int i = 0; i < 3; i++) gp[i] += s_g[3 + i][c - 1];     // c = p - e_x: +dy
#pragma unroll
        for (int i = 0; i < 3; i++) gp[i] -= s_g[3 + i][c + 1];     // c = p + e_x: -dy
        float d[3];
        ray_dir(f, x, y, d);
        g_depth += gp[0] * d[0] + gp[1] * d[1] + gp[2] * d[2];
    }
    if (f.rd != nullptr) {
        // "pgsr": surf_depth = nan_to_num(rd / -(n . ray)) -- its gradient goes to the plane-distance map and to the normal sums (zero
        // where the quotient is not finite: nan_to_num passes nothing there), none to the expected / median depth channels
        const Unbiased u = unbiased_depth(f, allmap, HW, pix);
        float g_rd = 0.0f;
        if (is_finite(u.v)) {
            g_rd = g_depth / (-u.ndr);
            const float k = g_depth * f.rd[pix] / (u.ndr * u.ndr);
#pragma unroll
            for (int i = 0; i < 3; i++) g_nv[i] += k * u.ray[i];
        }
        if (f.g_rd != nullptr) f.g_rd[pix] = g_rd;
#pragma unroll
        for (int i = 0; i < 3; i++) g_allmap[(2 + i) * HW + pix] = g_nv[i];
        g_allmap[pix] = 0.0f;
        g_allmap[HW + pix] = g_a;
        g_allmap[5 * HW + pix] = 0.0f;
        g_allmap[6 * HW + pix] = g_dist != nullptr ? g_dist[pix] : 0.0f;
        return;
    }
#pragma unroll
    for (int i = 0; i < 3; i++) g_allmap[(2 + i) * HW + pix] = g_nv[i];
    const float d0 = allmap[pix];
    const float q = d0 / a;
    float g0 = 0.0f;
    if (is_finite(q)) {                                        // nan_to_num passes the gradient only where its input is finite
        const float g_de = g_depth * (1.0f - f.depth_ratio);
        g0 = g_de / a;
        g_a -= g_de * d0 / (a * a);
    }
    g_allmap[pix] = g0;
    g_allmap[HW + pix] = g_a;
    const float dmed = allmap[5 * HW + pix];
    g_allmap[5 * HW + pix] = is_finite(dmed) ? g_depth * f.depth_ratio : 0.0f;
    g_allmap[6 * HW + pix] = g_dist != nullptr ? g_dist[pix] : 0.0f;   // rend_dist is a plain view of allmap[6]
}

// ---- compositing ----------------------------------------------------------------------------------------------
__device__ __forceinline__ float lin2srgb(float x)
{
    const float eps = 1.1920928955078125e-07f;
    return x <= 0.0031308f ? (323.0f / 25.0f) * x : (211.0f * powf(fmaxf(x, eps), 5.0f / 12.0f) - 11.0f) / 200.0f;
}
__device__ __forceinline__ float lin2srgb_grad(float x)
{
    const float eps = 1.1920928955078125e-07f;
    if (x <= 0.0031308f) return 323.0f / 25.0f;
    return x >= eps ? (211.0f / 200.0f) * (5.0f / 12.0f) * powf(x, -7.0f / 12.0f) : 0.0f;
}

__global__ void __launch_bounds__(256) surfel_composite_fwd_kernel(int HW, int srgb, const float* __restrict__ base, const float* __restrict__ refl,
                                                                   const float* __restrict__ spec, const float* __restrict__ alpha,
                                                                   const float* __restrict__ bg, float* __restrict__ render,
                                                                   float* __restrict__ diffuse)
{
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= HW) return;
    const float k = 1.0f - refl[pix], oma = 1.0f - alpha[pix];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float d = k * base[c * HW + pix];
        float v = d + spec[c * HW + pix];
        if (srgb) v = lin2srgb(v);
        render[c * HW + pix] = v + bg[c] * oma;
        diffuse[c * HW + pix] = d;
    }
}

__global__ void __launch_bounds__(256) surfel_composite_bwd_kernel(int HW, int srgb, const float* __restrict__ base, const float* __restrict__ refl,
                                                                   const float* __restrict__ spec, const float* __restrict__ bg,
                                                                   const float* __restrict__ g_render, const float* __restrict__ g_diffuse,
                                                                   float* __restrict__ g_base, float* __restrict__ g_refl,
                                                                   float* __restrict__ g_spec, float* __restrict__ g_alpha,
                                                                   float* __restrict__ zero_fill, long long zero_floats)
{
    const int pix = blockIdx.x * 256 + threadIdx.x;
    // a side job for the caller's next kernel: clear a buffer it accumulates into (the shading backward's texel gradients), in the
    // same launch instead of a fill of its own; the grid is sized for whichever of the two is larger
    for (long long i = (long long)pix; i < zero_floats; i += (long long)gridDim.x * 256) zero_fill[i] = 0.0f;
    if (pix >= HW) return;
    const float k = 1.0f - refl[pix];
    float ga = 0.0f, gr = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float b = base[c * HW + pix];
        const float gR = g_render != nullptr ? g_render[c * HW + pix] : 0.0f;
        ga -= bg[c] * gR;
        float gl = gR;
        if (srgb) gl *= lin2srgb_grad(k * b + spec[c * HW + pix]);
        const float gd = gl + (g_diffuse != nullptr ? g_diffuse[c * HW + pix] : 0.0f);
        g_base[c * HW + pix] = k * gd;
        gr -= b * gd;
        g_spec[c * HW + pix] = gl;
    }
    g_refl[pix] = gr;
    g_alpha[pix] = ga;
}

// gradient of the rasterizer's [8,H,W] feature map of render_surfel (channels: refl, roughness, albedo[3], indirect[3]) from
// the pieces the shading and compositing backward kernels produce -- one tensor instead of four slice gradients that autograd
// would each pad to [8,H,W] and add up
__global__ void __launch_bounds__(256) surfel_feature_grads_kernel(int HW, const float* __restrict__ g_refl_composite, const float* __restrict__ g_refl_shade,
                                                                   const float* __restrict__ g_rough, const float* __restrict__ g_albedo_hwc,
                                                                   const float* __restrict__ g_indirect_hwc, float* __restrict__ g_features,
                                                                   const float* __restrict__ g_alpha_a, const float* __restrict__ g_alpha_b,
                                                                   const float* __restrict__ g_alpha_c, float* __restrict__ g_alpha)
{
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= HW) return;
    if (g_alpha) g_alpha[pix] = (g_alpha_a ? g_alpha_a[pix] : 0.0f) + (g_alpha_b ? g_alpha_b[pix] : 0.0f) + (g_alpha_c ? g_alpha_c[pix] : 0.0f);
    g_features[pix] = g_refl_composite[pix] + g_refl_shade[pix];
    g_features[HW + pix] = g_rough[pix];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        g_features[(2 + c) * HW + pix] = g_albedo_hwc[3 * (size_t)pix + c];
        // the indirect radiance map only matters with the visibility tracer (mrgs_indirect_blend_backward)
        g_features[(5 + c) * HW + pix] = g_indirect_hwc ? g_indirect_hwc[3 * (size_t)pix + c] : 0.0f;
    }
}

MapsFrameDev to_dev(const MrgsMapsFrame* fr)
{
    MapsFrameDev f;
    f.H = fr->H; f.W = fr->W;
    for (int i = 0; i < 9; i++) { f.V[i] = fr->view_rot[i]; f.M[i] = fr->ray_matrix[i]; }
    for (int i = 0; i < 3; i++) f.o[i] = fr->ray_origin[i];
    f.depth_ratio = fr->depth_ratio;
    f.rd = fr->rend_distance; f.g_rd = fr->g_rend_distance; f.fx = fr->pgsr_fx; f.fy = fr->pgsr_fy;
    return f;
}

// Visibility blend of get_specular_color_surfel (utils/refl_utils.py:393-401): specular_light = direct * vis + (1 - vis) * indirect,
// specular = specular_light * alpha * weight, indirect_color = (1 - vis) * indirect * alpha * weight.  direct / specular /
// indirect_color are [3,H,W], weight [H,W,3] (the shading kernel's layouts), indirect and alpha arbitrary-stride maps, vis [H,W].
struct BlendMap { const float* p; long long sh, sw, sc; };
__global__ void __launch_bounds__(256) indirect_blend_fwd_kernel(int H, int W, const float* __restrict__ direct, const float* __restrict__ weight,
                                                                 BlendMap indirect, BlendMap alpha, const float* __restrict__ vis,
                                                                 float* __restrict__ specular, float* __restrict__ indirect_color)
{
    const int pix = blockIdx.x * 256 + threadIdx.x, HW = H * W;
    if (pix >= HW) return;
    const int y = pix / W, x = pix - y * W;
    const float v = vis[pix], a = alpha.p[y * alpha.sh + x * alpha.sw];
    const long long oi = y * indirect.sh + x * indirect.sw;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float i = indirect.p[oi + c * indirect.sc], w = weight[pix * 3 + c];
        const float L = direct[c * HW + pix] * v + (1.0f - v) * i;
        specular[c * HW + pix] = L * a * w;
        indirect_color[c * HW + pix] = (1.0f - v) * i * a * w;
    }
}

__global__ void __launch_bounds__(256) indirect_blend_bwd_kernel(int H, int W, const float* __restrict__ direct, const float* __restrict__ weight,
                                                                 BlendMap indirect, BlendMap alpha, const float* __restrict__ vis,
                                                                 const float* __restrict__ g_spec, const float* __restrict__ g_ic,
                                                                 float* __restrict__ g_direct /*[3,H,W]*/, float* __restrict__ g_weight /*[H,W,3]*/,
                                                                 float* __restrict__ g_indirect /*[H,W,3]*/, float* __restrict__ g_alpha /*[H,W]*/)
{
    const int pix = blockIdx.x * 256 + threadIdx.x, HW = H * W;
    if (pix >= HW) return;
    const int y = pix / W, x = pix - y * W;
    const float v = vis[pix], a = alpha.p[y * alpha.sh + x * alpha.sw];
    const long long oi = y * indirect.sh + x * indirect.sw;
    float ga = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float i = indirect.p[oi + c * indirect.sc], w = weight[pix * 3 + c];
        const float L = direct[c * HW + pix] * v + (1.0f - v) * i;
        const float gs = g_spec ? g_spec[c * HW + pix] : 0.0f, gi = g_ic ? g_ic[c * HW + pix] : 0.0f;
        const float ic_aw = (1.0f - v) * i;                       // indirect_color / (a w)
        g_direct[c * HW + pix] = gs * a * w * v;
        g_indirect[pix * 3 + c] = (gs + gi) * (1.0f - v) * a * w;
        g_weight[pix * 3 + c] = gs * L * a + gi * ic_aw * a;
        ga += gs * L * w + gi * ic_aw * w;
    }
    g_alpha[pix] = ga;
}

}   // namespace

extern "C" {

int mrgs_surfel_maps_forward(const MrgsMapsFrame* fr, const float* allmap, float* rend_normal, float* surf_depth, float* surf_normal,
                             float* normal_map, float* rend_alpha, float* rend_dist, float* rend_alpha2, void* stream)
{
    if (!fr || fr->H <= 0 || fr->W <= 0 || !allmap || !rend_normal || !surf_depth) return MRGS_E_BAD_ARG;
    const int HW = fr->H * fr->W;
    hipLaunchKernelGGL(surfel_maps_fwd_kernel, dim3((HW + 255) / 256), dim3(256), 0, (hipStream_t)stream, to_dev(fr), allmap, rend_normal,
                       surf_depth, surf_normal, normal_map, rend_alpha, rend_dist, rend_alpha2);
    return hipGetLastError() == hipSuccess ? MRGS_OK : MRGS_E_HIP;
}

int mrgs_surfel_maps_backward(const MrgsMapsFrame* fr, const float* allmap, const float* g_rend_normal, const float* g_surf_depth,
                              const float* g_surf_normal, const float* g_normal_map, const float* g_rend_alpha, const float* g_rend_dist,
                              const float* g_rend_alpha2, float* g_allmap, void* stream)
{
    if (!fr || fr->H <= 0 || fr->W <= 0 || !allmap || !g_allmap) return MRGS_E_BAD_ARG;
    hipLaunchKernelGGL(surfel_maps_bwd_kernel, dim3((fr->W + MB_TX - 1) / MB_TX, (fr->H + MB_TY - 1) / MB_TY), dim3(256), 0, (hipStream_t)stream,
                       to_dev(fr), allmap, g_rend_normal, g_surf_depth, g_surf_normal, g_normal_map, g_rend_alpha, g_rend_dist, g_rend_alpha2, g_allmap);
    return hipGetLastError() == hipSuccess ? MRGS_OK : MRGS_E_HIP;
}

int mrgs_surfel_composite_forward(int32_t H, int32_t W, int32_t srgb, const float* base_color, const float* refl_strength, const float* specular,
                                  const float* alpha, const float* bg, float* render, float* diffuse, void* stream)
{
    if (H <= 0 || W <= 0 || !base_color || !refl_strength || !specular || !alpha || !bg || !render || !diffuse) return MRGS_E_BAD_ARG;
    const int HW = H * W;
    hipLaunchKernelGGL(surfel_composite_fwd_kernel, dim3((HW + 255) / 256), dim3(256), 0, (hipStream_t)stream, HW, srgb, base_color,
                       refl_strength, specular, alpha, bg, render, diffuse);
    return hipGetLastError() == hipSuccess ? MRGS_OK : MRGS_E_HIP;
}

int mrgs_surfel_composite_backward(int32_t H, int32_t W, int32_t srgb, const float* base_color, const float* refl_strength,
                                   const float* specular, const float* bg, const float* g_render, const float* g_diffuse, float* g_base,
                                   float* g_refl, float* g_specular, float* g_alpha, float* zero_fill, int64_t zero_floats, void* stream)
{
    if (H <= 0 || W <= 0 || !base_color || !refl_strength || !specular || !bg || !g_base || !g_refl || !g_specular || !g_alpha)
        return MRGS_E_BAD_ARG;
    if (zero_floats < 0 || (zero_floats > 0 && !zero_fill)) return MRGS_E_BAD_ARG;
    const int HW = H * W;
    hipLaunchKernelGGL(surfel_composite_bwd_kernel, dim3((HW + 255) / 256), dim3(256), 0, (hipStream_t)stream, HW, srgb, base_color,
                       refl_strength, specular, bg, g_render, g_diffuse, g_base, g_refl, g_specular, g_alpha, zero_fill, (long long)zero_floats);
    return hipGetLastError() == hipSuccess ? MRGS_OK : MRGS_E_HIP;
}

int mrgs_indirect_blend_forward(int32_t H, int32_t W, const float* direct, const float* weight, const MrgsStridedMap* indirect,
                                const MrgsStridedMap* alpha, const float* visibility, float* specular, float* indirect_color, void* stream)
{
    if (H <= 0 || W <= 0 || !direct || !weight || !indirect || !alpha || !indirect->ptr || !alpha->ptr || !visibility || !specular || !indirect_color)
        return MRGS_E_BAD_ARG;
    const BlendMap mi = {indirect->ptr, (long long)indirect->stride_h, (long long)indirect->stride_w, (long long)indirect->stride_c};
    const BlendMap ma = {alpha->ptr, (long long)alpha->stride_h, (long long)alpha->stride_w, (long long)alpha->stride_c};
    hipLaunchKernelGGL(indirect_blend_fwd_kernel, dim3((H * W + 255) / 256), dim3(256), 0, (hipStream_t)stream, H, W, direct, weight, mi, ma,
                       visibility, specular, indirect_color);
    return hipGetLastError() == hipSuccess ? MRGS_OK : MRGS_E_HIP;
}

int mrgs_indirect_blend_backward(int32_t H, int32_t W, const float* direct, const float* weight, const MrgsStridedMap* indirect,
                                 const MrgsStridedMap* alpha, const float* visibility, const float* g_specular, const float* g_indirect_color,
                                 float* g_direct, float* g_weight, float* g_indirect, float* g_alpha, void* stream)
{
    if (H <= 0 || W <= 0 || !direct || !weight || !indirect || !alpha || !indirect->ptr || !alpha->ptr || !visibility || !g_direct || !g_weight ||
        !g_indirect || !g_alpha)
        return MRGS_E_BAD_ARG;
    const BlendMap mi = {indirect->ptr, (long long)indirect->stride_h, (long long)indirect->stride_w, (long long)indirect->stride_c};
    const BlendMap ma = {alpha->ptr, (long long)alpha->stride_h, (long long)alpha->stride_w, (long long)alpha->stride_c};
    hipLaunchKernelGGL(indirect_blend_bwd_kernel, dim3((H * W + 255) / 256), dim3(256), 0, (hipStream_t)stream, H, W, direct, weight, mi, ma,
                       visibility, g_specular, g_indirect_color, g_direct, g_weight, g_indirect, g_alpha);
    return hipGetLastError() == hipSuccess ? MRGS_OK : MRGS_E_HIP;
}

int mrgs_surfel_feature_grads(int32_t H, int32_t W, const float* g_refl_composite, const float* g_refl_shade, const float* g_roughness,
                              const float* g_albedo_hwc, const float* g_indirect_hwc, float* g_features, const float* g_alpha_a,
                              const float* g_alpha_b, const float* g_alpha_c, float* g_alpha, void* stream)
{
    if (H <= 0 || W <= 0 || !g_refl_composite || !g_refl_shade || !g_roughness || !g_albedo_hwc || !g_features) return MRGS_E_BAD_ARG;
    const int HW = H * W;
    hipLaunchKernelGGL(surfel_feature_grads_kernel, dim3((HW + 255) / 256), dim3(256), 0, (hipStream_t)stream, HW, g_refl_composite, g_refl_shade,
                       g_roughness, g_albedo_hwc, g_indirect_hwc, g_features, g_alpha_a, g_alpha_b, g_alpha_c, g_alpha);
    return hipGetLastError() == hipSuccess ? MRGS_OK : MRGS_E_HIP;
}

}   // extern "C"
