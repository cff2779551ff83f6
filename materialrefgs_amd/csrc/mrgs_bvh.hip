// Closest-hit ray queries against a triangle mesh: the visibility rays of the shading pass.
//
// Replaces RayTracer.trace (submodules/raytracing/raytracing/raytracer.py:20-56, raytracing_brdf/raytracer.py:83-123) =
// raytrace_kernel (submodules/raytracing/src/bvh.cu:694-720) over TriangleBvh4::ray_intersect (:259-302) with
// Triangle::ray_intersect (include/raytracing/triangle.cuh:27-45): per ray the nearest front-facing triangle with
// 0 <= t < MAX_DIST = 10 (bvh.cu:36); depth = t or 10 for a miss ("depth >= 10" is the caller's miss test,
// utils/refl_utils.py:390-391), position = o + depth * d, normal = unit face normal or 0.
//
// The answer of the reference is the minimum over all triangles (the hierarchy only prunes), so the hierarchy is ours:
//  * host build (C++, the reference builds on the host too): object-median splits on the widest centroid axis, four children per
//    node, <= 4 triangles per leaf; balanced, so the traversal stack is bounded by 3 * depth + 1 <= 32 entries (4 M triangles);
//  * node = 128 bytes: the four child boxes as six float4 (SoA) + four child codes -- eight 16-byte loads fetch all a visit needs;
//    triangles are stored in leaf order as (a, b-a, c-a, cross) = three float4;
//  * one ray per lane, stack in LDS ([entry][lane]: conflict-free), children visited near to far, boxes padded by a few ulp and
//    the slab test widened so that pruning is conservative: the result equals the brute-force minimum bit for bit
//    (oracle/trace_oracle.py), except for the id/normal when two triangles are hit at exactly the same t.
// Compiled with -ffp-contract=off: the triangle test is evaluated exactly as written (and as the oracle evaluates it).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "mrgs_internal.h"

namespace {

constexpr int BVH_LEAF = 4;
constexpr int BVH_STACK = 32;
constexpr float BVH_MAX_DIST = 10.0f;                 // bvh.cu:36
constexpr int32_t BVH_EMPTY = 0x7FFFFFFF;

struct BvhNode {                                      // 128 bytes
    float lo[3][4], hi[3][4];
    int32_t child[4];                                 // >= 0 inner node, < 0 leaf ~((first << 3) | (count - 1)), BVH_EMPTY none
    int32_t pad[4];
};
static_assert(sizeof(BvhNode) == 128, "node layout");

struct Layout {
    size_t nodes_off, tris_off, perm_off, total;
    int64_t node_cap;
};

Layout bvh_layout(int64_t n)
{
    Layout l;
    l.node_cap = n / 3 + 2;                           // every inner node has four children, every leaf >= 1 triangle
    l.nodes_off = 0;
    l.tris_off = mrgs_align_up((size_t)l.node_cap * sizeof(BvhNode), 256);
    l.perm_off = mrgs_align_up(l.tris_off + (size_t)n * 48, 256);
    l.total = mrgs_align_up(l.perm_off + (size_t)n * 4, 256);
    return l;
}

struct Builder {
    const float* v;
    const int32_t* t;
    int64_t n;
    std::vector<float> cent, tlo, thi;
    std::vector<int32_t> idx;
    BvhNode* nodes;
    int64_t n_nodes = 0, cap = 0;
    int max_depth = 0;

    void range_box(int64_t b, int64_t e, float lo[3], float hi[3], bool centroids) const
    {
        for (int k = 0; k < 3; ++k) { lo[k] = INFINITY; hi[k] = -INFINITY; }
        for (int64_t i = b; i < e; ++i) {
            const int64_t j = idx[i];
            for (int k = 0; k < 3; ++k) {
                const float l = centroids ? cent[3 * j + k] : tlo[3 * j + k], h = centroids ? cent[3 * j + k] : thi[3 * j + k];
                lo[k] = std::min(lo[k], l);
                hi[k] = std::max(hi[k], h);
            }
        }
    }

    int64_t split(int64_t b, int64_t e)
    {
        float lo[3], hi[3];
        range_box(b, e, lo, hi, true);
        int axis = 0;
        float ext = hi[0] - lo[0];
        for (int k = 1; k < 3; ++k)
            if (hi[k] - lo[k] > ext) { ext = hi[k] - lo[k]; axis = k; }
        const int64_t m = b + (e - b) / 2;
        std::nth_element(idx.begin() + b, idx.begin() + m, idx.begin() + e, [&](int32_t x, int32_t y) {
            const float cx = cent[3 * (int64_t)x + axis], cy = cent[3 * (int64_t)y + axis];
            return cx < cy || (cx == cy && x < y);
        });
        return m;
    }

    int32_t build(int64_t b, int64_t e, int depth, bool force_inner)
    {
        if (e - b <= BVH_LEAF && !force_inner) return ~(int32_t)((b << 3) | (e - b - 1));
        const int64_t me = n_nodes++;
        max_depth = std::max(max_depth, depth + 1);
        int64_t cb[4], ce[4];
        int nc;
        if (e - b <= BVH_LEAF) {
            nc = 1; cb[0] = b; ce[0] = e;
        } else {
            const int64_t m = split(b, e), m0 = split(b, m), m1 = split(m, e);
            nc = 4;
            cb[0] = b; ce[0] = m0; cb[1] = m0; ce[1] = m; cb[2] = m; ce[2] = m1; cb[3] = m1; ce[3] = e;
        }
        BvhNode nd;
        std::memset(&nd, 0, sizeof(nd));
        for (int c = 0; c < 4; ++c) {
            if (c >= nc) {
                for (int k = 0; k < 3; ++k) { nd.lo[k][c] = INFINITY; nd.hi[k][c] = -INFINITY; }
                nd.child[c] = BVH_EMPTY;
                continue;
            }
            float lo[3], hi[3];
            range_box(cb[c], ce[c], lo, hi, false);
            for (int k = 0; k < 3; ++k) {
                const float pad = 1e-5f * std::max(std::fabs(lo[k]), std::fabs(hi[k])) + 1e-30f;
                nd.lo[k][c] = lo[k] - pad;
                nd.hi[k][c] = hi[k] + pad;
            }
            nd.child[c] = build(cb[c], ce[c], depth + 1, false);
        }
        nodes[me] = nd;
        return (int32_t)me;
    }
};

__device__ __forceinline__ void cswap(float& da, int32_t& ca, float& db, int32_t& cb)
{
    const bool sw = db < da;
    const float d0 = sw ? db : da, d1 = sw ? da : db;
    const int32_t c0 = sw ? cb : ca, c1 = sw ? ca : cb;
    da = d0; db = d1; ca = c0; cb = c1;
}

// Walks the hierarchy for one ray per lane.  ANY_HIT: stops at the first accepted triangle (enough for "depth >= 10" tests).
template <bool ANY_HIT>
__device__ __forceinline__ void bvh_traverse(const BvhNode* __restrict__ nodes, const float4* __restrict__ tris, int32_t (*stack)[256], int tid,
                                             float ox, float oy, float oz, float dx, float dy, float dz, float& mint, int32_t& hit)
{
    const float ivx = 1.0f / dx, ivy = 1.0f / dy, ivz = 1.0f / dz;
    int sp = 0;
    int32_t cur = 0;
    for (;;) {
        if (cur >= 0) {
            const float4* nd = reinterpret_cast<const float4*>(nodes + cur);
            const float4 lx = nd[0], ly = nd[1], lz = nd[2], hx = nd[3], hy = nd[4], hz = nd[5];
            const int4 ch = reinterpret_cast<const int4*>(nd)[6];
            float dist[4];
            int32_t code[4] = {ch.x, ch.y, ch.z, ch.w};
            const float lxs[4] = {lx.x, lx.y, lx.z, lx.w}, lys[4] = {ly.x, ly.y, ly.z, ly.w}, lzs[4] = {lz.x, lz.y, lz.z, lz.w};
            const float hxs[4] = {hx.x, hx.y, hx.z, hx.w}, hys[4] = {hy.x, hy.y, hy.z, hy.w}, hzs[4] = {hz.x, hz.y, hz.z, hz.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float ax = (lxs[c] - ox) * ivx, bx = (hxs[c] - ox) * ivx;
                const float ay = (lys[c] - oy) * ivy, by = (hys[c] - oy) * ivy;
                const float az = (lzs[c] - oz) * ivz, bz = (hzs[c] - oz) * ivz;
                // fminf/fmaxf drop a NaN operand (0 * inf when the origin lies on a slab plane of an axis-parallel ray)
                const float t_in = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
                const float t_out = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
                // conservative: widen the interval by 1e-5 relative (the boxes are padded at build time as well)
                const bool h = (t_in <= t_out * 1.00001f + 1e-30f) & (t_in * 0.99999f < mint) & (code[c] != BVH_EMPTY);
                dist[c] = h ? t_in : INFINITY;
            }
            cswap(dist[0], code[0], dist[1], code[1]);
            cswap(dist[2], code[2], dist[3], code[3]);
            cswap(dist[0], code[0], dist[2], code[2]);
            cswap(dist[1], code[1], dist[3], code[3]);
            cswap(dist[1], code[1], dist[2], code[2]);
            if (dist[3] < INFINITY) stack[sp++][tid] = code[3];
            if (dist[2] < INFINITY) stack[sp++][tid] = code[2];
            if (dist[1] < INFINITY) stack[sp++][tid] = code[1];
            if (dist[0] < INFINITY) { cur = code[0]; continue; }
        } else {
            const int32_t lf = ~cur;
            const int first = lf >> 3, cnt = (lf & 7) + 1;
            for (int i = 0; i < cnt; ++i) {
                const float4 q0 = tris[3 * (int64_t)(first + i)], q1 = tris[3 * (int64_t)(first + i) + 1], q2 = tris[3 * (int64_t)(first + i) + 2];
                // a = q0.xyz, v1v0 = (q0.w, q1.x, q1.y), v2v0 = (q1.z, q1.w, q2.x), n = (q2.y, q2.z, q2.w)   (triangle.cuh:27-45)
                const float e1x = q0.w, e1y = q1.x, e1z = q1.y, e2x = q1.z, e2y = q1.w, e2z = q2.x, nx = q2.y, ny = q2.z, nz = q2.w;
                const float rx = ox - q0.x, ry = oy - q0.y, rz = oz - q0.z;
                const float dn = dx * nx + dy * ny + dz * nz;
                const float qx = ry * dz - rz * dy, qy = rz * dx - rx * dz, qz = rx * dy - ry * dx;      // rov0 x rd
                const float d = 1.0f / dn;
                const float u = d * -(qx * e2x + qy * e2y + qz * e2z);
                const float v = d * (qx * e1x + qy * e1y + qz * e1z);
                float t = d * -(nx * rx + ny * ry + nz * rz);
                if ((dn >= 0.0f) | (u < 0.0f) | (u > 1.0f) | (v < 0.0f) | ((u + v) > 1.0f) | (t < 0.0f)) t = 1e6f;
                if (t < mint) { mint = t; hit = first + i; }
            }
            if (ANY_HIT && hit >= 0) return;
        }
        if (sp == 0) return;
        cur = stack[--sp][tid];
    }
}

__global__ __launch_bounds__(256) void bvh_trace_kernel(const BvhNode* __restrict__ nodes, const float4* __restrict__ tris,
                                                         const int32_t* __restrict__ perm, int64_t n_rays,
                                                         const float* rays_o, const float* rays_d, float* positions, float* normals,
                                                         float* __restrict__ depth, int32_t* __restrict__ face_ids)   // positions / normals may alias the rays (inplace)
{
    __shared__ int32_t stack[BVH_STACK][256];
    const int tid = threadIdx.x;
    const int64_t r = (int64_t)blockIdx.x * 256 + tid;
    if (r >= n_rays) return;
    const float ox = rays_o[3 * r], oy = rays_o[3 * r + 1], oz = rays_o[3 * r + 2];
    const float dx = rays_d[3 * r], dy = rays_d[3 * r + 1], dz = rays_d[3 * r + 2];
    float mint = BVH_MAX_DIST;
    int32_t hit = -1;
    bvh_traverse<false>(nodes, tris, stack, tid, ox, oy, oz, dx, dy, dz, mint, hit);
    depth[r] = mint;
    positions[3 * r] = ox + mint * dx; positions[3 * r + 1] = oy + mint * dy; positions[3 * r + 2] = oz + mint * dz;
    float nx = 0.f, ny = 0.f, nz = 0.f;
    int32_t id = -1;
    if (hit >= 0) {
        const float4 q2 = tris[3 * (int64_t)hit + 2];
        const float len = sqrtf(q2.y * q2.y + q2.z * q2.z + q2.w * q2.w);
        nx = q2.y / len; ny = q2.z / len; nz = q2.w / len;
        id = perm[hit];
    }
    normals[3 * r] = nx; normals[3 * r + 1] = ny; normals[3 * r + 2] = nz;
    if (face_ids) face_ids[r] = id;
}

struct VisCam { float Kinv[9]; const float* R; const float* T; };
struct VisMap { const float* p; long long sh, sw, sc; };

// Visibility of the environment along the mirror direction of every pixel (get_specular_color_surfel, utils/refl_utils.py:379-391):
// rays_cam = un-normalised pixel ray (sample_camera_rays_unnormalize :75-93), origin = rays_o + surf_depth * rays_cam,
// direction = safe_normalize(reflection(safe_normalize(-rays_cam), normal)); visibility = 1 where alpha <= 0 or the ray is free for
// 10 units, 0 where it is blocked.  8 x 32 pixel tiles per workgroup: neighbouring rays walk the same nodes.
__global__ __launch_bounds__(256) void bvh_visibility_kernel(const BvhNode* __restrict__ nodes, const float4* __restrict__ tris, VisCam cam,
                                                              int H, int W, VisMap normal, VisMap alpha, const float* __restrict__ surf_depth,
                                                              float* __restrict__ visibility)
{
    __shared__ int32_t stack[BVH_STACK][256];
    const int tid = threadIdx.x;
    const int x = blockIdx.x * 32 + (tid & 31), y = blockIdx.y * 8 + (tid >> 5);
    if (x >= W || y >= H) return;
    const size_t pix = (size_t)y * W + x;
    const float a = alpha.p[(long long)y * alpha.sh + (long long)x * alpha.sw];
    float vis = 1.0f;
    if (a > 0.0f) {
        const float fx = (float)x, fy = (float)y;
        const float pcx = cam.Kinv[0] * fx + cam.Kinv[1] * fy + cam.Kinv[2], pcy = cam.Kinv[3] * fx + cam.Kinv[4] * fy + cam.Kinv[5],
                    pcz = cam.Kinv[6] * fx + cam.Kinv[7] * fy + cam.Kinv[8];
        const float* R = cam.R;
        const float tx = cam.T[0], ty = cam.T[1], tz = cam.T[2];
        const float qx = pcx - tx, qy = pcy - ty, qz = pcz - tz;
        const float rox = -(R[0] * tx + R[1] * ty + R[2] * tz), roy = -(R[3] * tx + R[4] * ty + R[5] * tz), roz = -(R[6] * tx + R[7] * ty + R[8] * tz);
        const float cx = (R[0] * qx + R[1] * qy + R[2] * qz) - rox, cy = (R[3] * qx + R[4] * qy + R[5] * qz) - roy,
                    cz = (R[6] * qx + R[7] * qy + R[8] * qz) - roz;                          // rays_cam (un-normalised)
        const float cl = fmaxf(sqrtf(cx * cx + cy * cy + cz * cz), 1e-20f);                  // safe_normalize(-rays_cam)
        const float wx = -cx / cl, wy = -cy / cl, wz = -cz / cl;
        const long long on = (long long)y * normal.sh + (long long)x * normal.sw;
        const float nx = normal.p[on], ny = normal.p[on + normal.sc], nz = normal.p[on + 2 * normal.sc];
        const float ndv = wx * nx + wy * ny + wz * nz;
        float rx = 2.f * nx * ndv - wx, ry = 2.f * ny * ndv - wy, rz = 2.f * nz * ndv - wz;  // reflection() :95-98
        const float rl = fmaxf(sqrtf(rx * rx + ry * ry + rz * rz), 1e-20f);
        rx /= rl; ry /= rl; rz /= rl;
        const float sd = surf_depth[pix];
        float mint = BVH_MAX_DIST;
        int32_t hit = -1;
        bvh_traverse<true>(nodes, tris, stack, tid, rox + sd * cx, roy + sd * cy, roz + sd * cz, rx, ry, rz, mint, hit);
        vis = hit >= 0 ? 0.0f : 1.0f;                                                        // (depth >= 10).float()  :391
    }
    visibility[pix] = vis;
}

}   // namespace

extern "C" size_t mrgs_bvh_bytes(int64_t n_triangles)
{
    if (n_triangles <= 0) return 0;
    return bvh_layout(n_triangles).total;
}

extern "C" int mrgs_bvh_build(const float* vertices, int64_t n_vertices, const int32_t* triangles, int64_t n_triangles, void* blob_host,
                              size_t blob_bytes)
{
    if (!vertices || !triangles || !blob_host || n_vertices <= 0 || n_triangles <= 0) return MRGS_E_BAD_ARG;
    if (n_triangles >= (1 << 28)) return MRGS_E_UNSUPPORTED;
    const Layout l = bvh_layout(n_triangles);
    if (blob_bytes < l.total) return MRGS_E_WORKSPACE;
    for (int64_t i = 0; i < 3 * n_triangles; ++i)
        if (triangles[i] < 0 || triangles[i] >= n_vertices) return MRGS_E_BAD_ARG;
    Builder bd;
    bd.v = vertices; bd.t = triangles; bd.n = n_triangles;
    bd.cent.resize(3 * n_triangles); bd.tlo.resize(3 * n_triangles); bd.thi.resize(3 * n_triangles); bd.idx.resize(n_triangles);
    for (int64_t i = 0; i < n_triangles; ++i) {
        bd.idx[i] = (int32_t)i;
        const float* a = vertices + 3 * (int64_t)triangles[3 * i];
        const float* b = vertices + 3 * (int64_t)triangles[3 * i + 1];
        const float* c = vertices + 3 * (int64_t)triangles[3 * i + 2];
        for (int k = 0; k < 3; ++k) {
            bd.cent[3 * i + k] = (a[k] + b[k] + c[k]) / 3.0f;
            bd.tlo[3 * i + k] = std::min(a[k], std::min(b[k], c[k]));
            bd.thi[3 * i + k] = std::max(a[k], std::max(b[k], c[k]));
        }
    }
    std::memset(blob_host, 0, l.total);
    bd.nodes = reinterpret_cast<BvhNode*>((char*)blob_host + l.nodes_off);
    bd.cap = l.node_cap;
    bd.build(0, n_triangles, 0, true);
    if (bd.n_nodes > l.node_cap) return MRGS_E_INTERNAL;
    if (3 * bd.max_depth + 1 > BVH_STACK) return MRGS_E_UNSUPPORTED;
    float* rec = reinterpret_cast<float*>((char*)blob_host + l.tris_off);
    int32_t* perm = reinterpret_cast<int32_t*>((char*)blob_host + l.perm_off);
    for (int64_t i = 0; i < n_triangles; ++i) {
        const int64_t j = bd.idx[i];
        perm[i] = (int32_t)j;
        const float* a = vertices + 3 * (int64_t)triangles[3 * j];
        const float* b = vertices + 3 * (int64_t)triangles[3 * j + 1];
        const float* c = vertices + 3 * (int64_t)triangles[3 * j + 2];
        float* o = rec + 12 * i;
        const float e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
        o[0] = a[0]; o[1] = a[1]; o[2] = a[2];
        o[3] = e1[0]; o[4] = e1[1]; o[5] = e1[2];
        o[6] = e2[0]; o[7] = e2[1]; o[8] = e2[2];
        o[9] = e1[1] * e2[2] - e1[2] * e2[1];                       // n = v1v0 x v2v0
        o[10] = e1[2] * e2[0] - e1[0] * e2[2];
        o[11] = e1[0] * e2[1] - e1[1] * e2[0];
    }
    return MRGS_OK;
}

extern "C" int mrgs_bvh_trace(const void* blob_dev, int64_t n_triangles, int64_t n_rays, const float* rays_o, const float* rays_d,
                              float* positions, float* normals, float* depth, int32_t* face_ids, void* stream)
{
    if (!blob_dev || n_triangles <= 0 || n_rays < 0) return MRGS_E_BAD_ARG;
    if (n_rays == 0) return MRGS_OK;
    if (!rays_o || !rays_d || !positions || !normals || !depth) return MRGS_E_BAD_ARG;
    const Layout l = bvh_layout(n_triangles);
    const char* base = (const char*)blob_dev;
    const int64_t nblk = (n_rays + 255) / 256;
    if (nblk > 0x7FFFFFFF) return MRGS_E_UNSUPPORTED;
    bvh_trace_kernel<<<dim3((unsigned)nblk), 256, 0, (hipStream_t)stream>>>(
        reinterpret_cast<const BvhNode*>(base + l.nodes_off), reinterpret_cast<const float4*>(base + l.tris_off),
        reinterpret_cast<const int32_t*>(base + l.perm_off), n_rays, rays_o, rays_d, positions, normals, depth, face_ids);
    return hipGetLastError() == hipSuccess ? MRGS_OK : MRGS_E_HIP;
}

extern "C" int mrgs_bvh_visibility(const void* blob_dev, int64_t n_triangles, int32_t H, int32_t W, const float* Kinv, const float* R,
                                   const float* T, const MrgsStridedMap* normal, const MrgsStridedMap* alpha, const float* surf_depth,
                                   float* visibility, void* stream)
{
    if (!blob_dev || n_triangles <= 0 || H <= 0 || W <= 0 || !Kinv || !R || !T || !normal || !alpha || !normal->ptr || !alpha->ptr ||
        !surf_depth || !visibility)
        return MRGS_E_BAD_ARG;
    const Layout l = bvh_layout(n_triangles);
    const char* base = (const char*)blob_dev;
    VisCam cam;
    for (int i = 0; i < 9; ++i) cam.Kinv[i] = Kinv[i];
    cam.R = R; cam.T = T;
    const VisMap nm = {normal->ptr, (long long)normal->stride_h, (long long)normal->stride_w, (long long)normal->stride_c};
    const VisMap am = {alpha->ptr, (long long)alpha->stride_h, (long long)alpha->stride_w, (long long)alpha->stride_c};
    bvh_visibility_kernel<<<dim3((W + 31) / 32, (H + 7) / 8), 256, 0, (hipStream_t)stream>>>(
        reinterpret_cast<const BvhNode*>(base + l.nodes_off), reinterpret_cast<const float4*>(base + l.tris_off), cam, H, W, nm, am, surf_depth,
        visibility);
    return hipGetLastError() == hipSuccess ? MRGS_OK : MRGS_E_HIP;
}
