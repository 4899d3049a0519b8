// ntx_instancer.hip -- the patch instancer on the GPU: C_Instancer::GetModelInput (instancer/src/instancer.cpp:751-1037) behind
// Instancer.get_model_input (instancer/instancer.pyx:38-54).  gfx950 only.
//
// The reference traces every ray through an Embree scene of instanced boxes on ONE CPU thread (the loop at instancer.cpp:772),
// sorts the face crossings, marches the union of the boxes in steps of `step_size` and maps every sample into the patch it
// falls in.  Here the buffers of instancer.pyx:41-50 are produced in HBM, where ntx_render_instanced reads them:
//
//   inst_hits_kernel   ray per lane, instances wave-uniform (their 3x4 matrices arrive through the scalar cache, no vector
//                      memory in the loop): slab test of the ray in patch coordinates against every instance -- for the few
//                      thousand patches of a scene, all pairs on the VALUs cost less than one BVH build (spheres cull per
//                      wave) -- a crossed box goes onto the ray's list as one record {t_in, t_out, instance}
//   inst_mesh_kernel   the same against the triangles of the meshes (closest crossing, with its triangle)
//   inst_shade_kernel  the closing sample of rays that end on an auxiliary mesh (shadeMesh, instancer.cpp:716-743)
//   inst_march_kernel  wave per ray.  The reference's walk over the sorted crossings with a std::set of open patches
//                      (instancer.cpp:800-826, 870-1010) is taken apart into steps that are parallel over crossings, gaps or
//                      marching steps (see WaveLds below); emission is lane per marching step, every output row is written
//                      whole (emitted samples + the defaults of instancer.pyx:41-50), dense, once.  inst_march_shadow_kernel:
//                      the same with shadow rays (occlusion queries by the wave, see `occluded`).
//                      Bound: HBM writes, (3+3+1+1+1+1+P) * 4 bytes per (ray, step); measured at 0.36-0.49 of the peak, the
//                      rest is per-ray event work.
//
// Float32 operations are spelled in the order of oracle/instancer_oracle.py (-ffp-contract=off, IEEE divide and sqrt), so that
// the two agree bit for bit on the same instance matrices.
#include "nerftex.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <thread>
#include <type_traits>
#include <utility>
#include <vector>

extern "C" int ntx_set_error(int code, const char *fmt, ...);   // nerftex.hip

#define INST_TRY(expr)                                                                                   \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) return ntx_set_error(NTX_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// development knob (-DNTX_INST_DEBUG and NERFTEX_INST_DEBUG=bits): parts of the march kernel left out to time the rest; compiled out otherwise
#ifdef NTX_INST_DEBUG
#define NTX_DBG_SKIP(a, bit) ((a).debug_skip & (bit))
#else
#define NTX_DBG_SKIP(a, bit) false
#endif

namespace ntx_inst {

constexpr int MAX_HITS = 200;            // MAX_TOTAL_HITS, instancer.cpp:22
constexpr int MAX_PARAMS = 32;
constexpr float T_FAR = 100.0f;          // init_ray(..., 0, 100, ...), instancer.cpp:776
constexpr uint32_t INF_BITS = 0x7f800000u;

struct Box { float b0[3], b1[3]; };
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c0;
}
__device__ __forceinline__ float uniform01(uint32_t x) { return __builtin_bit_cast(float, (x & 0x7fffffu) | 0x3f800000u) - 1.0f; }
__device__ __forceinline__ int64_t global_index(int64_t idx0, uint32_t run, int64_t stride, int64_t k) {
    const uint32_t r = (uint32_t)k, q = r / run;
    return idx0 + (int64_t)q * stride + (int64_t)(r - q * run);
}

// block<3,3>(0,0) * p + block<3,1>(0,3) (instancer.cpp:556-558), products summed left to right
__device__ __forceinline__ void affine(const float *m, float x, float y, float z, float *out) {
#pragma unroll
    for (int r = 0; r < 3; ++r) out[r] = ((m[4 * r] * x + m[4 * r + 1] * y) + m[4 * r + 2] * z) + m[4 * r + 3];
}
__device__ __forceinline__ void linear34(const float *m, float x, float y, float z, float *out) {
#pragma unroll
    for (int r = 0; r < 3; ++r) out[r] = (m[4 * r] * x + m[4 * r + 1] * y) + m[4 * r + 2] * z;
}
__device__ __forceinline__ void linear33(const float *m, float x, float y, float z, float *out) {
#pragma unroll
    for (int r = 0; r < 3; ++r) out[r] = (m[3 * r] * x + m[3 * r + 1] * y) + m[3 * r + 2] * z;
}
// ... and as EIGEN evaluates them (getPt, getDir: a coefficient of a 3x3 * 3 product is the unrolled reduction x0 + (x1 + x2))
__device__ __forceinline__ void affine_e(const float *m, float x, float y, float z, float *out) {
#pragma unroll
    for (int r = 0; r < 3; ++r) out[r] = (m[4 * r] * x + (m[4 * r + 1] * y + m[4 * r + 2] * z)) + m[4 * r + 3];
}
__device__ __forceinline__ void linear33_e(const float *m, float x, float y, float z, float *out) {
#pragma unroll
    for (int r = 0; r < 3; ++r) out[r] = m[3 * r] * x + (m[3 * r + 1] * y + m[3 * r + 2] * z);
}
__device__ __forceinline__ void normalized(float &x, float &y, float &z) {   // Eigen's normalized(): squaredNorm = x0^2 + (x1^2 + x2^2)
    const float n2 = x * x + (y * y + z * z);
    if (n2 > 0.0f) { const float n = __builtin_sqrtf(n2); x = x / n; y = y / n; z = z / n; }
}

// A cone around the 64 rays of a wave: apex = mean origin (the rays of a camera share theirs), axis = mean direction, opening =
// the widest ray; `reach` = how far an origin lies from the apex.  A sphere (c, r^2) that stays further than r + reach from the
// cone cannot be met by any of the rays: tested lane per sphere, 64 spheres per ballot.  Conservative (widened; rays pointing
// more than 60 degrees apart switch it off), so it never changes a result.
struct WaveCone {
    float ax, ay, az, ux, uy, uz, cos_t, sin_t, reach;
    float wx, wy, wz, ws, wo;       // ... and a wedge: the 64 rays of a pixel row fan out in ONE plane (unit normal w); ws = the largest sine of a ray
    bool on, wedge;                 // out of that plane, wo = how far an origin lies off it.  A sphere further from the plane than r + wo + |v| ws is out.
    __device__ __forceinline__ bool reaches(const float *c, float r2) const {
        if (!on) return true;
        const float vx = c[0] - ax, vy = c[1] - ay, vz = c[2] - az;
        const float along = (vx * ux + vy * uy) + vz * uz;
        const float v2 = (vx * vx + vy * vy) + vz * vz;
        const float perp = __builtin_sqrtf(fmaxf(v2 - along * along, 0.0f));
        const float rs = __builtin_sqrtf(r2) * 1.001f, vn = __builtin_sqrtf(v2);
        const float tol = 1e-4f * (vn + 1.0f);
        if (wedge && fabsf((vx * wx + vy * wy) + vz * wz) > rs + wo + vn * ws + tol) return false;
        return perp * cos_t - along * sin_t <= rs + reach + tol;
    }
};
__device__ __forceinline__ float lane_value(float v, int src) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src)); }
__device__ __forceinline__ float wave_sum(float v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }
__device__ __forceinline__ float wave_max(float v) { for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o)); return v; }
__device__ __forceinline__ float wave_min(float v) { for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o)); return v; }
__device__ __forceinline__ WaveCone wave_cone(float ox, float oy, float oz, float dx, float dy, float dz) {
    WaveCone c;
    c.ax = wave_sum(ox) * (1.0f / 64.0f); c.ay = wave_sum(oy) * (1.0f / 64.0f); c.az = wave_sum(oz) * (1.0f / 64.0f);
    const float dn = __builtin_sqrtf((dx * dx + dy * dy) + dz * dz);
    const float inv = dn > 0.0f ? 1.0f / dn : 0.0f;
    const float nx = dx * inv, ny = dy * inv, nz = dz * inv;
    float ux = wave_sum(nx), uy = wave_sum(ny), uz = wave_sum(nz);
    const float un = __builtin_sqrtf((ux * ux + uy * uy) + uz * uz);
    const float uinv = un > 0.0f ? 1.0f / un : 0.0f;
    c.ux = ux * uinv; c.uy = uy * uinv; c.uz = uz * uinv;
    const float cmin = wave_min((nx * c.ux + ny * c.uy) + nz * c.uz) - 1e-5f;
    const float ex = ox - c.ax, ey = oy - c.ay, ez = oz - c.az;
    c.reach = wave_max(__builtin_sqrtf(ex * ex + (ey * ey + ez * ez))) * 1.001f;
    c.on = cmin > 0.5f && wave_min(dn) > 0.0f;
    c.cos_t = cmin; c.sin_t = __builtin_sqrtf(fmaxf(1.0f - cmin * cmin, 0.0f));
    // the wedge's plane: through the apex, along the first and the last ray's directions
    const float fx = lane_value(nx, 0), fy = lane_value(ny, 0), fz = lane_value(nz, 0), gx = lane_value(nx, 63), gy = lane_value(ny, 63), gz = lane_value(nz, 63);
    float px = fy * gz - fz * gy, py = fz * gx - fx * gz, pz = fx * gy - fy * gx;
    const float pn = __builtin_sqrtf((px * px + py * py) + pz * pz);
    c.wedge = c.on && pn > 1e-6f;
    const float pinv = c.wedge ? 1.0f / pn : 0.0f;
    c.wx = px * pinv; c.wy = py * pinv; c.wz = pz * pinv;
    c.ws = wave_max(fabsf((nx * c.wx + ny * c.wy) + nz * c.wz)) * 1.001f + 1e-7f;
    c.wo = wave_max(fabsf((ex * c.wx + ey * c.wy) + ez * c.wz)) * 1.001f;
    return c;
}

// ---------------------------------------------------------------------------------------------------------------------------
// the two-level cull every kernel walks: the objects (instanced boxes or triangles, each behind a sphere) sit in Morton order behind a
// permutation, a sphere around every block of 64 consecutive ones.  A wave tests 64 block spheres per ballot, then the 64 objects of
// every block that passed, lane per object: `body(id, near, record)` is called in uniform control flow with the lane's object (its index
// in the ORIGINAL order, -1 behind the end), whether its own sphere passed, and its record (already in registers).  The tests are culls only: what they let through gets the
// exact test, so neither the order nor the grouping changes a result.
// ---------------------------------------------------------------------------------------------------------------------------
struct Scene {
    const float *recs;                      // object at Morton position idx: recs[idx * 16 ..]: an instance = {sphere (centre, r^2), world -> patch 3x4};
    int sph;                                //   a triangle = {v0, v1 - v0, v2 - v0, sphere, padding}; sph = where the sphere sits (0 or 9)
    const int32_t *perm; const float *bspheres;   // perm[idx] = the object's index in the ORIGINAL order; a sphere per block of 64
    int n, nb;
    int top_min;                            // with at most this many blocks the walk goes straight to the objects (4; tests force the block level on with -1)
};
struct Rec { f32x4 q[4]; __device__ __forceinline__ float at(int i) const { return q[i >> 2][i & 3]; } };
template <typename Reach, typename Body>
__device__ __forceinline__ void walk_blocks(const Scene &sc, int lane, int part, int parts, Reach reach, Body body) {
    // `parts` waves share one walk: all of them cull the block spheres (cheap), the blocks that pass are dealt out round robin
    auto fetch = [&](int blk, Rec &r, int &id) {                                  // the lane's object of a block: its record and its id, one round of loads
        const int idx = blk * 64 + lane;
        id = -1;
        if (idx < sc.n) {
            const f32x4 *g = reinterpret_cast<const f32x4 *>(sc.recs) + (size_t)idx * 4;
            r.q[0] = g[0]; r.q[1] = g[1]; r.q[2] = g[2]; r.q[3] = g[3];
            id = sc.perm[idx];
        }
    };
    int dealt = 0;                                                                // blocks that passed so far (the same count in every wave of the walk)
    for (int bb = 0; bb < sc.nb; bb += 64) {
        const int b = bb + lane;
        bool pass = b < sc.nb;
        if (pass && sc.nb > sc.top_min) { const float *bsp = sc.bspheres + (size_t)b * 4; pass = reach(bsp, bsp[3]); }   // (a handful of blocks: straight to their objects)
        uint64_t all = __ballot(pass), bm = all;
        if (parts > 1) {                                                          // this wave's share: every parts-th set bit
            bm = 0;
            int rank = dealt;
            for (uint64_t w = all; w; w &= w - 1, ++rank)
                if (rank % parts == part) bm |= w & (~w + 1);
        }
        dealt += __builtin_popcountll(all);
        Rec cur{}, nxt{};
        int cur_id = -1, nxt_id = -1;
        if (bm) fetch(bb + __builtin_ctzll(bm), cur, cur_id);
        while (bm) {
            bm &= bm - 1;
            if (bm) fetch(bb + __builtin_ctzll(bm), nxt, nxt_id);                 // the next block's loads fly while this one is worked on
            bool near = false;
            if (cur_id >= 0) {
                const bool tri = sc.sph != 0;                                      // (selects, not indexing: the record stays in registers)
                const float sp[3] = {tri ? cur.q[2].y : cur.q[0].x, tri ? cur.q[2].z : cur.q[0].y, tri ? cur.q[2].w : cur.q[0].z};
                near = reach(sp, tri ? cur.q[3].x : cur.q[0].w);
            }
            body(cur_id, near, cur);
            cur = nxt; cur_id = nxt_id;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// (ray, instance) pairs: what rtcIntersect1 with the all-hits filter reports (instancer.cpp:779, 526-541)
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int TILE_WAVES = 16;            // most waves that share the walk of one 64-ray tile (hit and mesh kernels); the launch picks 4 or 16
__global__ __launch_bounds__(64 * TILE_WAVES) void inst_hits_kernel(const float *__restrict__ rays_o, const float *__restrict__ rays_d, int n_rays,
                                                        Scene sc, Box box,
                                                        uint32_t *__restrict__ count, uint4 *__restrict__ hits) {
    // a workgroup = 64 rays; its waves share the walk over the instances and hand out a ray's list slots from a counter in LDS
    __shared__ uint32_t slots[64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int ray = blockIdx.x * 64 + lane;
    const bool live = ray < n_rays;
    const int r = live ? ray : n_rays - 1;
    if (threadIdx.x < 64) slots[threadIdx.x] = 0u;
    __syncthreads();
    const float ox = rays_o[3 * r], oy = rays_o[3 * r + 1], oz = rays_o[3 * r + 2];
    const float dx = rays_d[3 * r], dy = rays_d[3 * r + 1], dz = rays_d[3 * r + 2];
    const float dd2 = (dx * dx + dy * dy) + dz * dz;
    const WaveCone cone = wave_cone(ox, oy, oz, dx, dy, dz);
    // spheres around the instanced boxes (centre, radius^2 widened) against the wave's cone -- a wave holds neighbouring rays, so few
    // instances survive -- and the survivors once more against each ray
    walk_blocks(sc, lane, wave, (int)(blockDim.x >> 6), [&](const float *c, float r2) { return cone.reaches(c, r2); }, [&](int id, bool near, const Rec &own) {
      // the lane's own instance sits in registers and the survivors are handed round by readlane: no memory access, and so no latency,
      // inside the loop over them
      uint64_t mk = __ballot(near);
      while (mk) {
        const int src = __builtin_ctzll(mk);
        mk &= mk - 1;
        const int k = __builtin_amdgcn_readlane(id, src);
        const float sp[4] = {lane_value(own.q[0].x, src), lane_value(own.q[0].y, src), lane_value(own.q[0].z, src), lane_value(own.q[0].w, src)};
        const float cx = sp[0] - ox, cy = sp[1] - oy, cz = sp[2] - oz;
        const float qx = cy * dz - cz * dy, qy = cz * dx - cx * dz, qz = cx * dy - cy * dx;
        if (!__any((qx * qx + qy * qy) + qz * qz <= sp[3] * dd2)) continue;
        const float m[12] = {lane_value(own.q[1].x, src), lane_value(own.q[1].y, src), lane_value(own.q[1].z, src), lane_value(own.q[1].w, src),
                             lane_value(own.q[2].x, src), lane_value(own.q[2].y, src), lane_value(own.q[2].z, src), lane_value(own.q[2].w, src),
                             lane_value(own.q[3].x, src), lane_value(own.q[3].y, src), lane_value(own.q[3].z, src), lane_value(own.q[3].w, src)};
        float ol[3], dl[3];
        affine(m, ox, oy, oz, ol);
        linear34(m, dx, dy, dz, dl);
        float t_in = -INFINITY, t_out = INFINITY;
        bool miss = false;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (dl[a] == 0.0f) {
                miss = miss || ol[a] < box.b0[a] || ol[a] > box.b1[a];
            } else {
                const float inv = 1.0f / dl[a];
                const float t0 = (box.b0[a] - ol[a]) * inv, t1 = (box.b1[a] - ol[a]) * inv;
                const float lo = t0 < t1 ? t0 : t1, hi = t0 < t1 ? t1 : t0;
                t_in = lo > t_in ? lo : t_in;
                t_out = hi < t_out ? hi : t_out;
            }
        }
        if (live && !miss && t_in < t_out) {
            // the two face crossings of the box, each reported when tnear < t <= tfar: one record {t_in, t_out, instance}, 0 = not reported
            const bool in_ok = t_in > 0.0f && t_in <= T_FAR, out_ok = t_out > 0.0f && t_out <= T_FAR;
            if (in_ok || out_ok) {
                const uint32_t slot = atomicAdd(&slots[lane], 1u);
                if (slot < (uint32_t)MAX_HITS)
                    hits[(size_t)ray * MAX_HITS + slot] = make_uint4(in_ok ? __builtin_bit_cast(uint32_t, t_in) : 0u, out_ok ? __builtin_bit_cast(uint32_t, t_out) : 0u, (uint32_t)k, 0u);
            }
        }
      }
    });
    __syncthreads();
    if (threadIdx.x < 64 && live) count[ray] = slots[lane];
}

// closest crossing of the meshes per ray (Moeller-Trumbore, no culling); tris[f] = {v0, v1 - v0, v2 - v0, sphere centre, radius^2}
__global__ __launch_bounds__(64 * TILE_WAVES) void inst_mesh_kernel(const float *__restrict__ rays_o, const float *__restrict__ rays_d, int n_rays,
                                                        Scene sc, unsigned long long *__restrict__ t_mesh) {
    __shared__ unsigned long long nearest[64];                                   // per ray: (t bits << 32 | triangle) of the closest crossing, over the tile's waves
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int ray = blockIdx.x * 64 + lane;
    const bool live = ray < n_rays;
    const int r = live ? ray : n_rays - 1;
    if (threadIdx.x < 64) nearest[threadIdx.x] = (unsigned long long)INF_BITS << 32 | INF_BITS;
    __syncthreads();
    const float o[3] = {rays_o[3 * r], rays_o[3 * r + 1], rays_o[3 * r + 2]};
    const float d[3] = {rays_d[3 * r], rays_d[3 * r + 1], rays_d[3 * r + 2]};
    float best = INFINITY;
    int best_f = 0;
    const float dd2 = (d[0] * d[0] + d[1] * d[1]) + d[2] * d[2];
    const WaveCone cone = wave_cone(o[0], o[1], o[2], d[0], d[1], d[2]);
    walk_blocks(sc, lane, wave, (int)(blockDim.x >> 6), [&](const float *c, float r2) { return cone.reaches(c, r2); }, [&](int id, bool near, const Rec &own) {
      uint64_t mk = __ballot(near);                                              // (the lane's own triangle sits in registers, handed round by readlane)
      while (mk) {
        const int src = __builtin_ctzll(mk);
        mk &= mk - 1;
        const int f = __builtin_amdgcn_readlane(id, src);
        {   // the triangle's sphere, as in inst_hits_kernel
            const float cx = lane_value(own.at(9), src) - o[0], cy = lane_value(own.at(10), src) - o[1], cz = lane_value(own.at(11), src) - o[2];
            const float qx = cy * d[2] - cz * d[1], qy = cz * d[0] - cx * d[2], qz = cx * d[1] - cy * d[0];
            if (!__any((qx * qx + qy * qy) + qz * qz <= lane_value(own.at(12), src) * dd2)) continue;
        }
        const float tr[9] = {lane_value(own.at(0), src), lane_value(own.at(1), src), lane_value(own.at(2), src), lane_value(own.at(3), src), lane_value(own.at(4), src),
                             lane_value(own.at(5), src), lane_value(own.at(6), src), lane_value(own.at(7), src), lane_value(own.at(8), src)};
        const float *v0 = tr, *e1 = tr + 3, *e2 = tr + 6;
        const float p[3] = {d[1] * e2[2] - d[2] * e2[1], d[2] * e2[0] - d[0] * e2[2], d[0] * e2[1] - d[1] * e2[0]};
        const float det = (e1[0] * p[0] + e1[1] * p[1]) + e1[2] * p[2];
        if (det == 0.0f) continue;
        const float inv_det = 1.0f / det;
        const float s[3] = {o[0] - v0[0], o[1] - v0[1], o[2] - v0[2]};
        const float u = ((s[0] * p[0] + s[1] * p[1]) + s[2] * p[2]) * inv_det;
        if (u < 0.0f || u > 1.0f) continue;
        const float q[3] = {s[1] * e1[2] - s[2] * e1[1], s[2] * e1[0] - s[0] * e1[2], s[0] * e1[1] - s[1] * e1[0]};
        const float v = ((d[0] * q[0] + d[1] * q[1]) + d[2] * q[2]) * inv_det;
        if (v < 0.0f || u + v > 1.0f) continue;
        const float tt = ((e2[0] * q[0] + e2[1] * q[1]) + e2[2] * q[2]) * inv_det;
        // ties go to the lower triangle whatever order they are met in
        if (tt > 0.0f && tt <= T_FAR && (tt < best || (tt == best && f < best_f))) { best = tt; best_f = f; }
      }
    });
    // (t, triangle) as one 64-bit key: positive floats order like their bits, ties go to the lower triangle
    if (best < INFINITY) atomicMin(&nearest[lane], ((unsigned long long)__builtin_bit_cast(uint32_t, best) << 32) | (uint32_t)best_f);
    __syncthreads();
    if (threadIdx.x < 64 && live) t_mesh[ray] = nearest[lane];                  // (no crossing: the high word stays +inf)
}

// ---------------------------------------------------------------------------------------------------------------------------
// image textures (instancer.cpp:34-50, 605-667)
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int MAX_TEX_FILES = 4;         // texture FILES in the constructor's list (each multiplies one parameter, :656-662)
struct TexTable { int32_t offset, rows, cols, pad; };      // one channel matrix: texels[offset + r * cols + c], r = x, c = y from the bottom
struct TexArgs {
    int n_tex;                              // 0 = no parameter textures
    int32_t par_idx[MAX_TEX_FILES];         // texture_parameter_idxs
    int min_samples, n_samples;             // min_texture_samples, n_texture_samples
    float radius;                           // patch_max_extent (:69, :246)
    const float *texels; const TexTable *table;             // the first n_tex tables are the parameter textures
    // the instancer mesh behind a uniform grid of CANDIDATE lists: cell -> [start, end) into cand[] = the triangles that can be the
    // closest one for some point of the cell (worked out when the textures are set, see ntx_instancer_set_parameter_textures)
    const int32_t *cell_start; const int32_t *cand; const float *tris; const float *face_uv;   // tris[primID][9] corners, face_uv[primID][6]
    float gmin[3], inv_cell;
    int32_t dim[3]; int32_t n_faces;
};

// interpolate2d (instancer.cpp:605-625): bilinear between the four texels around x * (rows - 1, cols - 1); indices truncate, weights are
// x - floor(x); indices outside the matrix are clamped (the reference reads past it; at u or v = 1 that texel's weight is 0)
__device__ __forceinline__ float interpolate2d(const float *__restrict__ texels, const TexTable tb, float u, float v) {
    const float x0 = u * ((float)tb.rows - 1.0f), x1 = v * ((float)tb.cols - 1.0f);
    const int i = (int)x0, j = (int)x1;
    const float w0 = x0 - floorf(x0), w1 = x1 - floorf(x1);
    const int i0 = min(max(i, 0), tb.rows - 1), i1 = min(max(i + 1, 0), tb.rows - 1);
    const int j0 = min(max(j, 0), tb.cols - 1), j1 = min(max(j + 1, 0), tb.cols - 1);
    const float *y = texels + tb.offset;
    const float y00 = y[(size_t)i0 * tb.cols + j0], y01 = y[(size_t)i0 * tb.cols + j1], y10 = y[(size_t)i1 * tb.cols + j0], y11 = y[(size_t)i1 * tb.cols + j1];
    return ((y00 * (1.0f - w0) * (1.0f - w1) + y01 * (1.0f - w0) * w1) + y10 * w0 * (1.0f - w1)) + y11 * w0 * w1;
}

__device__ __forceinline__ float dot3e(const float *a, const float *b) { return a[0] * b[0] + (a[1] * b[1] + a[2] * b[2]); }   // Eigen pairs x0 + (x1 + x2)

// closest_point_triangle (instancer.cpp:154-198): distance of p to triangle abc, barycentrics of the closest point
__device__ __forceinline__ float closest_point_triangle(const float *p, const float *a, const float *b, const float *c, float *uvw) {
    const float ab[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, ac[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]}, ap[3] = {p[0] - a[0], p[1] - a[1], p[2] - a[2]};
    float q[3];
    const float d1 = dot3e(ab, ap), d2 = dot3e(ac, ap);
    const float bp[3] = {p[0] - b[0], p[1] - b[1], p[2] - b[2]};
    const float d3 = dot3e(ab, bp), d4 = dot3e(ac, bp);
    const float cp[3] = {p[0] - c[0], p[1] - c[1], p[2] - c[2]};
    const float d5 = dot3e(ab, cp), d6 = dot3e(ac, cp);
    const float vc = d1 * d4 - d3 * d2, vb = d5 * d2 - d1 * d6, va = d3 * d6 - d5 * d4;
    if (d1 <= 0.0f && d2 <= 0.0f) { q[0] = a[0]; q[1] = a[1]; q[2] = a[2]; uvw[0] = 1.0f; uvw[1] = 0.0f; uvw[2] = 0.0f; }
    else if (d3 >= 0.0f && d4 <= d3) { q[0] = b[0]; q[1] = b[1]; q[2] = b[2]; uvw[0] = 0.0f; uvw[1] = 1.0f; uvw[2] = 0.0f; }
    else if (d6 >= 0.0f && d5 <= d6) { q[0] = c[0]; q[1] = c[1]; q[2] = c[2]; uvw[0] = 0.0f; uvw[1] = 0.0f; uvw[2] = 1.0f; }
    else if (vc <= 0.0f && d1 >= 0.0f && d3 <= 0.0f) {
        const float v = d1 / (d1 - d3);
        q[0] = a[0] + v * ab[0]; q[1] = a[1] + v * ab[1]; q[2] = a[2] + v * ab[2]; uvw[0] = 1.0f - v; uvw[1] = v; uvw[2] = 0.0f;
    } else if (vb <= 0.0f && d2 >= 0.0f && d6 <= 0.0f) {
        const float v = d2 / (d2 - d6);
        q[0] = a[0] + v * ac[0]; q[1] = a[1] + v * ac[1]; q[2] = a[2] + v * ac[2]; uvw[0] = 1.0f - v; uvw[1] = 0.0f; uvw[2] = v;
    } else if (va <= 0.0f && (d4 - d3) >= 0.0f && (d5 - d6) >= 0.0f) {
        const float v = (d4 - d3) / ((d4 - d3) + (d5 - d6));
        q[0] = b[0] + v * (c[0] - b[0]); q[1] = b[1] + v * (c[1] - b[1]); q[2] = b[2] + v * (c[2] - b[2]); uvw[0] = 0.0f; uvw[1] = 1.0f - v; uvw[2] = v;
    } else {
        const float denom = 1.0f / ((va + vb) + vc);
        const float v = vb * denom, w = vc * denom;
        q[0] = (a[0] + v * ab[0]) + w * ac[0]; q[1] = (a[1] + v * ab[1]) + w * ac[1]; q[2] = (a[2] + v * ab[2]) + w * ac[2];
        uvw[0] = (1.0f - v) - w; uvw[1] = v; uvw[2] = w;
    }
    const float e[3] = {p[0] - q[0], p[1] - q[1], p[2] - q[2]};
    return __builtin_sqrtf(e[0] * e[0] + (e[1] * e[1] + e[2] * e[2]));                      // (q - p).norm()
}

// getParameters' point query (instancer.cpp:644-654): the triangle of the instancer mesh whose closest point lies nearest to p, strictly
// within the radius; of several at one distance the lowest primID (the restatement's order).  The reference walks Embree's BVH; here
// the cell of p names every triangle that can win for a point of that cell (its list was cut down with exact distance bounds on the
// host), and each of them gets the reference's closest_point_triangle: the result is that of testing every triangle of the mesh.
// Lane = one query; no wave collectives inside.
__device__ __forceinline__ bool closest_uv(const TexArgs &T, float px, float py, float pz, float *u_out, float *v_out) {
    const float p[3] = {px, py, pz};
    float best = T.radius;
    int best_f = -1;
    float bw[3] = {0.0f, 0.0f, 0.0f};
    const float fx = (px - T.gmin[0]) * T.inv_cell, fy = (py - T.gmin[1]) * T.inv_cell, fz = (pz - T.gmin[2]) * T.inv_cell;
    if (fx >= 0.0f && fy >= 0.0f && fz >= 0.0f && fx < (float)T.dim[0] && fy < (float)T.dim[1] && fz < (float)T.dim[2]) {   // (the grid reaches the radius beyond the mesh)
        const size_t cell = ((size_t)(int)fz * T.dim[1] + (int)fy) * T.dim[0] + (int)fx;
        const int s0 = T.cell_start[cell], s1 = T.cell_start[cell + 1];
        for (int e = s0; e < s1; ++e) {
            const int f = T.cand[e];
            const float *tr = T.tris + (size_t)f * 9;
            const float a[3] = {tr[0], tr[1], tr[2]}, b[3] = {tr[3], tr[4], tr[5]}, c[3] = {tr[6], tr[7], tr[8]};
            float w[3];
            const float d = closest_point_triangle(p, a, b, c, w);
            if (d < best || (d == best && best_f >= 0 && f < best_f)) { best = d; best_f = f; bw[0] = w[0]; bw[1] = w[1]; bw[2] = w[2]; }
        }
    }
    if (best_f < 0) return false;
    const float *uv = T.face_uv + (size_t)best_f * 6;                                          // UV.row(f0) * w0 + UV.row(f1) * w1 + UV.row(f2) * w2 (:661)
    *u_out = (uv[0] * bw[0] + uv[2] * bw[1]) + uv[4] * bw[2];
    *v_out = (uv[1] * bw[0] + uv[3] * bw[1]) + uv[5] * bw[2];
    return true;
}

// getParameters (instancer.cpp:640-667) for the texture parameters only: val[q] = parameter par_idx[q] * texture q at the closest point
// of the mesh, or the parameter as given when nothing lies within reach
__device__ __forceinline__ void texture_values(const TexArgs &T, const float *par, float px, float py, float pz, float *val) {
    float u = 0.0f, v = 0.0f;
    const bool found = closest_uv(T, px, py, pz, &u, &v);
#pragma unroll
    for (int q = 0; q < MAX_TEX_FILES; ++q) {
        if (q < T.n_tex) {
            const float p0 = par[T.par_idx[q]];
            val[q] = found ? p0 * interpolate2d(T.texels, T.table[q], u, v) : p0;
        } else val[q] = 0.0f;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// marching (instancer.cpp:787-1030), wave per ray
// ---------------------------------------------------------------------------------------------------------------------------
struct MarchArgs {
    const float *rays_o, *rays_d, *params;
    const float *mats, *origins, *xforms;   // xforms[K][24]: world -> patch matrix (12), direction map (9), padding: what a step gathers
    const uint32_t *count; const uint4 *hits; const unsigned long long *t_mesh;   // (t bits << 32 | triangle) of the closest mesh hit; NULL = no mesh
    float *rays_d_map, *pts, *t, *dists, *color_last, *alpha_last, *alpha_weight, *params_map;
    int32_t *instance_id; uint8_t *hit; int32_t *status;
    int n_rays, n_pts, n_params;
    int light_dir_idx, light_strength_idx, method, use_mean;
    float step_size, blend_range;
    uint32_t seed_lo, seed_hi;
    int64_t idx0, idx_stride; uint32_t idx_run;
    const float *tris; Box box; Scene inst_scene, tri_scene;      // shadow rays: the scene again, behind its two-level cull
    int min_shadow, n_shadow;
    const uint8_t *kind;                       // per triangle: bit 0 = auxiliary mesh (shaded), bit 1 = primID 1 of its own mesh (shadow filter, :553)
    TexArgs tex;                               // parameter textures on the instancer mesh (getParameters, :640-667); n_tex = 0: none
    int sparse;       // NTX_OPT_INSTANCER_SPARSE: no defaults behind a ray's last step (the rows whose dists are 0)
    int debug_skip;   // development build only (-DNTX_INST_DEBUG + NERFTEX_INST_DEBUG): leave parts of the kernel out to time the rest
};

// The reference walks the sorted crossings with a std::set of the patches it is inside (instancer.cpp:800-826, 870-1010): a crossing
// of a patch that is in the set takes it out, any other puts it in.  Here the walk is taken apart so that nothing but one float
// sum per segment is left sequential:
//   records    what the hit kernel left: a crossed box with its entry and exit parameter (lane per record)
//   crossings  the rank of a crossing in the sorted list = the number of crossings in front of it, counted over the records; the
//              first crossing of a record ENTERS its patch, the second leaves it, a lone one enters and never leaves
//   intervals  the ranks (b, e') of a record's own crossings (e' = the end of the list when there is no second): the patch is in the
//              set in the GAPS b < j <= e', gap j being the stretch in front of crossing j; in ascending patch order (std::set's
//              order) unless the rule is 'nearest'
//   gaps       lane per gap: patches in the set = 2 * (entering crossings before j) - j; segments of the union compacted by ballot,
//              their lengths summed in order (the one sequential loop); the first marching step of a gap as a running maximum
//   steps      lane per step: its gap by bisection of the gaps' first steps, then one pass over the intervals that reach into the
//              64 steps' gaps for the patch it is given to.  64 consecutive steps per pass whatever gaps they fall in.
struct WaveLds {
    float ev_t[MAX_HITS];
    uint32_t ev_info[MAX_HITS + 4];       // bit 0: the crossing enters its patch
    float g_off[MAX_HITS + 4];            // segment_offset in force in gap j (instancer.cpp:1001)
    union {
        int g_step0[MAX_HITS + 4];        // first step emitted in gap j; [n_gaps] = number of steps emitted
        struct { float ts[MAX_HITS / 2 + 2], te[MAX_HITS / 2]; } seg;   // before that: start (then offset) and end of the segments
    } gs;
    union {
        struct { float t_in[MAX_HITS], t_out[MAX_HITS]; uint32_t id[MAX_HITS]; } raw;        // the records of the hit kernel: a crossed box each
        struct { uint32_t id[MAX_HITS], be[MAX_HITS]; float ox[MAX_HITS], oy[MAX_HITS], oz[MAX_HITS]; } iv;   // intervals, by patch
    } u;
    float par[MAX_PARAMS];
};

__device__ __forceinline__ float mean_distance(float mu_f, float hw_f) {   // instancer.cpp:746-748 (double inside)
    const double mu = mu_f, hw = hw_f;
    return (float)(mu + 2 * mu * (hw * hw) / (3 * (mu * mu) + hw * hw));
}

// ---- shadow rays (instancer.cpp:591-602, filter :543-554) ---------------------------------------------------------------------
constexpr int MAX_SHADOW_ENTRIES = 4096;
struct SegLds {                               // the segments of the ray (shadow and texture samples are spaced along them)
    float ts[MAX_HITS / 2 + 2], len[MAX_HITS / 2 + 2];   // start and length of a segment (segment_lengths, :802-821)
    float off[MAX_HITS / 2 + 2];              // its segment_offset (:1001)
    int step0[MAX_HITS / 2 + 2];              // the first marching step emitted in it; [n_segments] = number of steps
    uint8_t g_seg[MAX_HITS + 8];              // the segment gap j lies in
};
struct ShadowLds {
    uint32_t bits[MAX_SHADOW_ENTRIES / 32];   // the shadow queries of the ray, one bit each: its shadow samples, or its marching steps
    int32_t cand_id[128]; float cand_a0[128], cand_a1[128];   // occluders waiting for their exact tests: id and the stretch of the ray they can shadow
    float sl[MAX_HITS / 2 + 2];               // the spacing of a segment's shadow samples
    uint16_t base[MAX_HITS / 2 + 4];          // first entry of segment i in `bits`; [n_segments] = number of entries
};
constexpr int TEX_RING = 128;                 // texture samples of the ray held at a time (two chunks of 64, produced in order)
struct TexLds {
    float ring[MAX_TEX_FILES][TEX_RING];      // value of texture parameter q at sample x: ring[q][x % TEX_RING]
    float sl[MAX_HITS / 2 + 2];               // the spacing of a segment's texture samples
    uint16_t base[MAX_HITS / 2 + 4];          // first sample of segment i; [n_segments] = number of samples
};
struct NoLds {};
template <bool SHADOW, bool TEX> struct MarchLds {
    WaveLds w;
    typename std::conditional<SHADOW || TEX, SegLds, NoLds>::type sg;
    typename std::conditional<SHADOW, ShadowLds, NoLds>::type sh;
    typename std::conditional<TEX, TexLds, NoLds>::type tx;
};

// ---- shadow rays (instancer.cpp:591-602, filter :543-554) ---------------------------------------------------------------------
// Every shadow ray of a primary ray (o, d) starts on it, at o + alpha d, and runs along the one light direction l: they all lie in the
// plane through the ray along l, and sweep the strip { o + alpha d + beta l : alpha_min <= alpha <= alpha_max, beta >= 0 } of it.
struct Strip {
    float ox, oy, oz, dx, dy, dz, lx, ly, lz;
    float nx, ny, nz, nn, ll, ddq, dl_, inv_nn, ra, rb, ta, tb;
    bool strip;
    __device__ __forceinline__ Strip(float ox_, float oy_, float oz_, float dx_, float dy_, float dz_, float lx_, float ly_, float lz_, float ta_, float tb_)
        : ox(ox_), oy(oy_), oz(oz_), dx(dx_), dy(dy_), dz(dz_), lx(lx_), ly(ly_), lz(lz_), ta(ta_), tb(tb_) {
        nx = dy * lz - dz * ly; ny = dz * lx - dx * lz; nz = dx * ly - dy * lx;                   // normal of the plane, not normalised
        nn = (nx * nx + ny * ny) + nz * nz;                                                       // = |d|^2 |l|^2 - (d.l)^2
        ll = (lx * lx + ly * ly) + lz * lz; ddq = (dx * dx + dy * dy) + dz * dz; dl_ = (dx * lx + dy * ly) + dz * lz;
        strip = nn > 1e-10f * ddq * ll;                                                           // d and l not parallel
        inv_nn = strip ? 1.0f / nn : 0.0f;
        const float inv_rt = strip ? 1.0f / __builtin_sqrtf(nn) : 0.0f;
        ra = __builtin_sqrtf(ll) * inv_rt * 1.01f; rb = __builtin_sqrtf(ddq) * inv_rt * 1.01f;
    }
    // can the sphere (c, r^2) meet the strip?  Inside the plane a sphere whose centre has oblique coordinates (alpha, beta) reaches
    // alpha +- r |l| / sqrt(nn), beta +- r |d| / sqrt(nn).  Conservative.
    __device__ __forceinline__ bool reaches(const float *c, float r2) const {
        const float wx = c[0] - ox, wy = c[1] - oy, wz = c[2] - oz;
        const float h0 = (wx * nx + wy * ny) + wz * nz;
        if (!(h0 * h0 <= r2 * nn * 1.001f)) return false;
        if (!strip) {                     // light along the ray: no strip to cull with -- unless it is EXACTLY along it (one point only,
            if (nn != 0.0f) return true;  // inst_shade_kernel passes the shadow ray itself): then every shadow ray lies on the line (o, l)
            const float qx = wy * lz - wz * ly, qy = wz * lx - wx * lz, qz = wx * ly - wy * lx;
            return (qx * qx + qy * qy) + qz * qz <= r2 * ll * 1.001f + 1e-12f;
        }
        const float b1 = (wx * dx + wy * dy) + wz * dz, b2 = (wx * lx + wy * ly) + wz * lz;
        const float alpha = (b1 * ll - b2 * dl_) * inv_nn, beta = (b2 * ddq - b1 * dl_) * inv_nn;
        const float r = __builtin_sqrtf(r2);
        const float tol = 1e-4f * (fabsf(alpha) + fabsf(beta) + 1.0f);
        return alpha + r * ra + tol >= ta && alpha - r * ra - tol <= tb && beta + r * rb + tol >= 0.0f;
    }
};

// the EXACT tests, spelled like the restatement: is the shadow ray from p along l stopped by instance k / by triangle f?
// Accepted (filter :543-554): the top face of a patch box entered from outside, its bottom face either way (primID 4 / 1 of createAABB),
// a triangle hit from its front or primID 1 of its mesh from either side.
__device__ __forceinline__ bool instance_occludes(const float *__restrict__ mats, const Box &box, int k, float px, float py, float pz, float lx, float ly, float lz) {
    const float *m = mats + (size_t)k * 12;
    float ol[3], dl[3];
    affine(m, px, py, pz, ol);
    linear34(m, lx, ly, lz, dl);
    bool occ = false;
    if (dl[2] != 0.0f) {
        const float inv = 1.0f / dl[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float z = e ? box.b0[2] : box.b1[2];
            const float tt = (z - ol[2]) * inv;
            const float x = ol[0] + tt * dl[0], y = ol[1] + tt * dl[1];
            const bool on_face = tt > 0.0f && tt <= T_FAR && box.b0[0] <= x && x <= box.b1[0] && box.b0[1] <= y && y <= box.b1[1];
            occ = occ || (on_face && (e == 1 || dl[2] < 0.0f));
        }
    }
    return occ;
}
__device__ __forceinline__ bool triangle_occludes(const float *__restrict__ tris, const uint8_t *__restrict__ kind, int f, float px, float py, float pz, float lx, float ly, float lz) {
    const float *tr = tris + (size_t)f * 13;
    const float *v0 = tr, *e1 = tr + 3, *e2 = tr + 6;
    const float p[3] = {ly * e2[2] - lz * e2[1], lz * e2[0] - lx * e2[2], lx * e2[1] - ly * e2[0]};
    const float det = (e1[0] * p[0] + e1[1] * p[1]) + e1[2] * p[2];
    const float inv_det = 1.0f / det;
    const float sv[3] = {px - v0[0], py - v0[1], pz - v0[2]};
    const float u = ((sv[0] * p[0] + sv[1] * p[1]) + sv[2] * p[2]) * inv_det;
    const float q[3] = {sv[1] * e1[2] - sv[2] * e1[1], sv[2] * e1[0] - sv[0] * e1[2], sv[0] * e1[1] - sv[1] * e1[0]};
    const float v = ((lx * q[0] + ly * q[1]) + lz * q[2]) * inv_det;
    const float tt = ((e2[0] * q[0] + e2[1] * q[1]) + e2[2] * q[2]) * inv_det;
    const float ng[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    const bool front = (lx * ng[0] + ly * ng[1]) + lz * ng[2] < 0.0f;
    // ... or the triangle is primID 1 of its mesh: the filter's last clause does not ask for the geometry (instancer.cpp:553)
    return det != 0.0f && !(u < 0.0f || u > 1.0f) && !(v < 0.0f || u + v > 1.0f) && tt > 0.0f && tt <= T_FAR && (front || (kind[f] & 2));
}

// isShadowed for ONE point (all lanes the same p and l): the closing sample of a ray that ends on an auxiliary mesh (shadeMesh, :736).
// The wave walks the two-level cull with the shadow ray's own line, lane per occluder, and votes.
__device__ __forceinline__ bool occluded_point(const MarchArgs &a, int lane, float px, float py, float pz, float lx, float ly, float lz) {
    const Strip st(px, py, pz, lx, ly, lz, lx, ly, lz, 0.0f, 0.0f);                               // d = l: nn == 0, the line (p, l)
    bool occ = false;
    walk_blocks(a.inst_scene, lane, 0, 1, [&](const float *c, float r2) { return st.reaches(c, r2); },
                [&](int id, bool near, const Rec &) { if (near) occ = occ || instance_occludes(a.mats, a.box, id, px, py, pz, lx, ly, lz); });
    walk_blocks(a.tri_scene, lane, 0, 1, [&](const float *c, float r2) { return st.reaches(c, r2); },
                [&](int id, bool near, const Rec &) { if (near) occ = occ || triangle_occludes(a.tris, a.kind, id, px, py, pz, lx, ly, lz); });
    return __any(occ);
}

// Keep the alphas with lo <= A + B alpha <= hi in [a0, a1], WIDENED by a margin far above what float32 rounding can move either this
// linear form or the exact test it stands in for: a cull, never a decision.
__device__ __forceinline__ void clip_alpha(float A, float B, float lo, float hi, float amax, float &a0, float &a1) {
    const float m = 1e-4f * (((fabsf(A) + fabsf(B) * amax) + (fabsf(lo) + fabsf(hi))) + 1.0f);
    const float l = lo - m, h = hi + m;
    if (fabsf(B) * amax <= m) {                                         // flat over the range: A decides (NaN: nothing passes, as in the exact test)
        if (!(A >= l - m && A <= h + m)) { a0 = INFINITY; a1 = -INFINITY; }
        return;
    }
    const float x0 = (l - A) / B, x1 = (h - A) / B;
    const float lo_a = x0 < x1 ? x0 : x1, hi_a = x0 < x1 ? x1 : x0;
    a0 = lo_a > a0 ? lo_a : a0;
    a1 = hi_a < a1 ? hi_a : a1;
}

// The stretch [a0, a1] of the primary ray from which a shadow ray can be stopped by instance k (empty: a0 > a1): in patch coordinates the
// point is ol0 + alpha dd and the shadow ray runs along dl, so where it meets the plane of a face -- tau, x, y -- is LINEAR in alpha.
__device__ __forceinline__ void instance_interval(const float *__restrict__ mats, const Box &box, int k, const Strip &st, float amax, float &a0, float &a1) {
    const float *m = mats + (size_t)k * 12;
    float ol0[3], dd[3], dl[3];
    affine(m, st.ox, st.oy, st.oz, ol0);
    linear34(m, st.dx, st.dy, st.dz, dd);
    linear34(m, st.lx, st.ly, st.lz, dl);
    float lo_all = INFINITY, hi_all = -INFINITY;
    if (dl[2] != 0.0f) {
        const float inv = 1.0f / dl[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            if (e == 0 && !(dl[2] < 0.0f)) continue;                    // the top face counts only when entered from outside
            const float z = e ? box.b0[2] : box.b1[2];
            const float At = (z - ol0[2]) * inv, Bt = -dd[2] * inv;     // tau(alpha)
            float f0 = st.ta, f1 = st.tb;
            clip_alpha(At, Bt, 0.0f, T_FAR, amax, f0, f1);
            clip_alpha(ol0[0] + At * dl[0], dd[0] + Bt * dl[0], box.b0[0], box.b1[0], amax, f0, f1);
            clip_alpha(ol0[1] + At * dl[1], dd[1] + Bt * dl[1], box.b0[1], box.b1[1], amax, f0, f1);
            if (f0 <= f1) { lo_all = f0 < lo_all ? f0 : lo_all; hi_all = f1 > hi_all ? f1 : hi_all; }
        }
    }
    a0 = lo_all; a1 = hi_all;
}
// ... and by triangle f: Moeller-Trumbore's u, v, tau for the origin o + alpha d are linear in alpha too
__device__ __forceinline__ void triangle_interval(const float *__restrict__ tris, const uint8_t *__restrict__ kind, int f, const Strip &st, float amax, float &a0, float &a1) {
    const float *tr = tris + (size_t)f * 13;
    const float *v0 = tr, *e1 = tr + 3, *e2 = tr + 6;
    const float lx = st.lx, ly = st.ly, lz = st.lz;
    const float p[3] = {ly * e2[2] - lz * e2[1], lz * e2[0] - lx * e2[2], lx * e2[1] - ly * e2[0]};
    const float det = (e1[0] * p[0] + e1[1] * p[1]) + e1[2] * p[2];
    const float ng[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    const bool front = (lx * ng[0] + ly * ng[1]) + lz * ng[2] < 0.0f;
    a0 = INFINITY; a1 = -INFINITY;
    if (det == 0.0f || !(front || (kind[f] & 2))) return;              // (the same floats as the exact test: the same decision)
    const float inv_det = 1.0f / det;
    const float s0[3] = {st.ox - v0[0], st.oy - v0[1], st.oz - v0[2]}, d[3] = {st.dx, st.dy, st.dz};
    const float q0[3] = {s0[1] * e1[2] - s0[2] * e1[1], s0[2] * e1[0] - s0[0] * e1[2], s0[0] * e1[1] - s0[1] * e1[0]};
    const float q1[3] = {d[1] * e1[2] - d[2] * e1[1], d[2] * e1[0] - d[0] * e1[2], d[0] * e1[1] - d[1] * e1[0]};
    const float Au = ((s0[0] * p[0] + s0[1] * p[1]) + s0[2] * p[2]) * inv_det, Bu = ((d[0] * p[0] + d[1] * p[1]) + d[2] * p[2]) * inv_det;
    const float Av = ((lx * q0[0] + ly * q0[1]) + lz * q0[2]) * inv_det, Bv = ((lx * q1[0] + ly * q1[1]) + lz * q1[2]) * inv_det;
    const float At = ((e2[0] * q0[0] + e2[1] * q0[1]) + e2[2] * q0[2]) * inv_det, Bt = ((e2[0] * q1[0] + e2[1] * q1[1]) + e2[2] * q1[2]) * inv_det;
    float f0 = st.ta, f1 = st.tb;
    clip_alpha(Au, Bu, 0.0f, 1.0f, amax, f0, f1);
    clip_alpha(Av, Bv, 0.0f, 1.0f, amax, f0, f1);
    clip_alpha(Au + Av, Bu + Bv, 0.0f, 1.0f, amax, f0, f1);
    clip_alpha(At, Bt, 0.0f, T_FAR, amax, f0, f1);
    a0 = f0; a1 = f1;
}

// fill row[f0 .. f1) with pattern[f % period] (period <= MAX_PARAMS, pattern in LDS or registers through `at`)
template <typename At>
__device__ __forceinline__ void fill_pattern(float *row, int f0, int f1, int period, int lane, At at) {
    // head up to a 16-byte boundary, body as float4, tail
    const int head = f0 + (int)(((16u - (uint32_t)((uintptr_t)(row + f0) & 15u)) & 15u) >> 2);
    const int h1 = head < f1 ? head : f1;
    for (int f = f0 + lane; f < h1; f += 64) __builtin_nontemporal_store(at(f % period), row + f);
    const int nvec = (f1 - h1) >> 2;
    if (nvec > 0) {
        f32x4 *vp = reinterpret_cast<f32x4 *>(row + h1);
        int r = (h1 + 4 * lane) % period;
        const int inc = 256 % period;
        for (int v = lane; v < nvec; v += 64) {
            f32x4 x;
            int q = r;
            x.x = at(q); q = q + 1 == period ? 0 : q + 1;
            x.y = at(q); q = q + 1 == period ? 0 : q + 1;
            x.z = at(q); q = q + 1 == period ? 0 : q + 1;
            x.w = at(q);
            __builtin_nontemporal_store(x, vp + v);
            r += inc; r = r >= period ? r - period : r;
        }
    }
    for (int f = h1 + 4 * nvec + lane; f < f1; f += 64) __builtin_nontemporal_store(at(f % period), row + f);
}

template <bool SHADOW, bool TEX>
__device__ __forceinline__ void march_ray(const MarchArgs &a, MarchLds<SHADOW, TEX> *lds) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int ray = blockIdx.x * 4 + wave;
    if (ray >= a.n_rays) return;
    WaveLds &L = lds[wave].w;
    auto &SG = lds[wave].sg;      // segment tables (SHADOW or TEX)
    auto &SH = lds[wave].sh;      // shadow tables (SHADOW only)
    auto &TX = lds[wave].tx;      // texture samples (TEX only)
    (void)SG; (void)SH; (void)TX;
    const int S = a.n_pts, P = a.n_params;
    const float h = a.step_size;
    const float ox = a.rays_o[3 * ray], oy = a.rays_o[3 * ray + 1], oz = a.rays_o[3 * ray + 2];
    const float dx = a.rays_d[3 * ray], dy = a.rays_d[3 * ray + 1], dz = a.rays_d[3 * ray + 2];
    float ndx = dx, ndy = dy, ndz = dz;
    normalized(ndx, ndy, ndz);                                             // getDir: dir.normalized(), instancer.cpp:562
    if (lane < P) L.par[lane] = a.params[(size_t)ray * P + lane];

    // ---- the crossings, sorted by (t, instID) (instancer.cpp:441-452, 787) ---------------------------------------------------
    // A record of the hit kernel is a crossed box: its entry and exit parameter (0 = not reported: behind the origin or beyond
    // tfar).  The rank of a crossing among all crossings of the ray = the number of crossings in front of it, counted over the
    // records (lane per record); the ranks of a record's own two crossings are its interval -- no pairing to search for: with both
    // reported the first ENTERS the patch and the second leaves it, a lone one enters and never leaves (the toggle of :812-824).
    const uint32_t raw = a.count[ray];
    const int n_rec = raw < (uint32_t)MAX_HITS ? (int)raw : MAX_HITS;
    for (int i = lane; i < n_rec; i += 64) {
        const uint4 r = a.hits[(size_t)ray * MAX_HITS + i];
        L.u.raw.t_in[i] = __builtin_bit_cast(float, r.x);
        L.u.raw.t_out[i] = __builtin_bit_cast(float, r.y);
        L.u.raw.id[i] = r.z;
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t tm_bits = a.t_mesh ? (uint32_t)(a.t_mesh[ray] >> 32) : INF_BITS;
    const bool has_mesh = tm_bits != INF_BITS;
    const float t_mesh = __builtin_bit_cast(float, tm_bits);
    const bool any_hit = n_rec > 0 || has_mesh;
    float r_in[4], r_out[4];
    uint32_t r_id[4];
    int rk_in[4], rk_out[4];
    int n_events = 0, n_front = 0;                // crossings of the ray; those not behind the mesh hit (it sorts behind its own t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int i = lane + 64 * g;
        r_in[g] = 0.0f; r_out[g] = 0.0f; r_id[g] = 0; rk_in[g] = 0; rk_out[g] = 0;
        if (64 * g >= n_rec) continue;
        if (i < n_rec) { r_in[g] = L.u.raw.t_in[i]; r_out[g] = L.u.raw.t_out[i]; r_id[g] = L.u.raw.id[i]; }
        const float xi = r_in[g], xo = r_out[g];
        const uint32_t id = r_id[g];
        int ci = 0, co = 0;
        #pragma unroll 4
        for (int k = 0; k < n_rec; ++k) {
            const float ki = L.u.raw.t_in[k], ko = L.u.raw.t_out[k];
            const uint32_t ik = L.u.raw.id[k];
            const bool lt = ik < id;
            ci += (ki != 0.0f && (ki < xi || (ki == xi && lt))) ? 1 : 0;
            ci += (ko != 0.0f && (ko < xi || (ko == xi && lt))) ? 1 : 0;
            co += (ki != 0.0f && (ki < xo || (ki == xo && (lt || ik == id)))) ? 1 : 0;      // a record's own entry lies in front of its exit
            co += (ko != 0.0f && (ko < xo || (ko == xo && lt))) ? 1 : 0;
        }
        rk_in[g] = ci; rk_out[g] = co;
        n_events += (xi != 0.0f ? 1 : 0) + (xo != 0.0f ? 1 : 0);
        n_front += (xi != 0.0f && xi <= t_mesh ? 1 : 0) + (xo != 0.0f && xo <= t_mesh ? 1 : 0);
    }
    for (int o = 32; o > 0; o >>= 1) { n_events += __shfl_xor(n_events, o); n_front += __shfl_xor(n_front, o); }
    const bool overflow_hits = raw > (uint32_t)MAX_HITS || n_events > MAX_HITS;
    // at most MAX_TOTAL_HITS crossings are kept (:539): the first ones of the sorted list; the mesh hit ends the walk (:804-811, 988)
    const int m = (has_mesh ? n_front : n_events) < MAX_HITS ? (has_mesh ? n_front : n_events) : MAX_HITS;
    bool has_iv[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const bool vi = r_in[g] != 0.0f, vo = r_out[g] != 0.0f;
        if (vi && rk_in[g] < m) { L.ev_t[rk_in[g]] = r_in[g]; L.ev_info[rk_in[g]] = 1u; }
        if (vo && rk_out[g] < m) { L.ev_t[rk_out[g]] = r_out[g]; L.ev_info[rk_out[g]] = vi ? 0u : 1u; }
        has_iv[g] = (vi && rk_in[g] < m) || (!vi && vo && rk_out[g] < m);
    }
    __builtin_amdgcn_wave_barrier();

    if (NTX_DBG_SKIP(a, 4)) return;
    uint32_t my_enter[4];                          // by position in the sorted list
    uint64_t enter_mask[4];
    int n_int = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = lane + 64 * q;
        my_enter[q] = e < m ? (L.ev_info[e] & 1u) : 0u;
        enter_mask[q] = __ballot(my_enter[q] != 0);
        n_int += __builtin_popcountll(enter_mask[q]);
    }
    if (NTX_DBG_SKIP(a, 8)) return;
    // intervals: (entering crossing b, leaving crossing e' or m) per record, in ascending patch order (std::set's order) unless the
    // rule is 'nearest', which does not depend on the order it meets the patches in once ties go to the smaller id (below)
    int iv_at[4];
    {
        uint64_t iv_mask[4];
        int before = 0;
#pragma unroll
        for (int g = 0; g < 4; ++g) { iv_mask[g] = __ballot(has_iv[g]); iv_at[g] = before + __builtin_popcountll(iv_mask[g] & ((1ull << lane) - 1ull)); before += __builtin_popcountll(iv_mask[g]); }
        if (a.method != 1) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if (has_iv[g]) L.u.raw.id[lane + 64 * g] = r_id[g] | 0x80000000u;          // (ids are below 2^29)
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (64 * g >= n_rec) continue;
                int c = 0;
                const uint32_t key = r_id[g] | 0x80000000u;
                #pragma unroll 8
                for (int k = 0; k < n_rec; ++k) {
                    const uint32_t ik = L.u.raw.id[k];
                    c += (ik >= 0x80000000u && ik < key) ? 1 : 0;
                }
                iv_at[g] = c;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();               // every read of the records is done: their memory takes the intervals
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (has_iv[g]) {
            const bool vi = r_in[g] != 0.0f;
            const int b_ = vi ? rk_in[g] : rk_out[g];
            const int e_ = (vi && r_out[g] != 0.0f && rk_out[g] < m) ? rk_out[g] : m;
            const float *og = a.origins + (size_t)r_id[g] * 3;
            const int at = iv_at[g];
            L.u.iv.be[at] = (uint32_t)b_ | ((uint32_t)e_ << 16);
            L.u.iv.id[at] = r_id[g];
            L.u.iv.ox[at] = og[0]; L.u.iv.oy[at] = og[1]; L.u.iv.oz[at] = og[2];
        }
    }

    if (NTX_DBG_SKIP(a, 16)) return;
    // ---- gaps and segments, lane per gap ------------------------------------------------------------------------------------
    // gap j = the stretch in front of event j (j = m: in front of the mesh hit).  Patches in the set there = 2 * (entering events
    // before j) - j.  A SEGMENT of the union of the boxes (instancer.cpp:800-826) starts at an entering event that finds the set
    // empty and ends at a leaving event that empties it; starts and ends are compacted by ballot and the one sequential thing
    // left is the sum over segments (`cleared`, :996, and with the stretch up to the mesh `total_segment_length`, :808, 817).
    const uint64_t lt = (1ull << lane) - 1ull;
    int cnt_q[4], seg_q[4];                       // patches in the set in gap e, the segment the gap lies in
    int n_starts = 0, n_ends = 0;
    {
        int enters_before = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = lane + 64 * q;
            cnt_q[q] = 2 * (enters_before + __builtin_popcountll(enter_mask[q] & lt)) - e;
            enters_before += __builtin_popcountll(enter_mask[q]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = lane + 64 * q;
            const bool is_start = e < m && my_enter[q] && cnt_q[q] == 0;
            const bool is_end = e < m && !my_enter[q] && cnt_q[q] == 1;
            const uint64_t sm = __ballot(is_start), em = __ballot(is_end);
            const float te = e < m ? L.ev_t[e] : 0.0f;
            if (is_start) L.gs.seg.ts[n_starts + __builtin_popcountll(sm & lt)] = te;
            if (is_end) L.gs.seg.te[n_ends + __builtin_popcountll(em & lt)] = te;
            seg_q[q] = n_starts + __builtin_popcountll(sm & lt) - 1;         // starts among the events before e, less one
            n_starts += __builtin_popcountll(sm);
            n_ends += __builtin_popcountll(em);
        }
    }
    __builtin_amdgcn_wave_barrier();
    float total = 0.0f;
    {
        float cleared = 0.0f;
        for (int i = 0; i < n_starts; ++i) {
            const float ts = L.gs.seg.ts[i];
            L.gs.seg.ts[i] = ts - cleared;                                   // segment_offset of the segment, :1001
            if constexpr (SHADOW || TEX) {                                   // segment_lengths, :809, 818
                SG.ts[i] = ts;
                SG.len[i] = i < n_ends ? L.gs.seg.te[i] - ts : (has_mesh ? t_mesh - ts : 0.0f);
            }
            if (i < n_ends) cleared = cleared + (L.gs.seg.te[i] - ts);      // :996 = :817
            else if (has_mesh) { total = cleared + (t_mesh - ts); }         // :808: the mesh closes the open segment
        }
        if (!(n_starts > n_ends && has_mesh)) total = cleared;
    }
    __builtin_amdgcn_wave_barrier();

    // ---- number of steps, dists (instancer.cpp:840-859) -------------------------------------------------------------------
    const int64_t gray = global_index(a.idx0, a.idx_run, a.idx_stride, ray);
    int n_steps = 0;
    float t_offset = 0.0f, last_dist = 0.0f;
    if (total > 0.0f) {
        const float u = uniform01(philox4x32_10(0u, (uint32_t)gray, (uint32_t)((uint64_t)gray >> 32), 2u, a.seed_lo, a.seed_hi));
        const uint32_t necessary = (uint32_t)(total / h);
        n_steps = necessary < (uint32_t)S ? (int)necessary : S;
        if (n_steps == 0) {
            last_dist = total; t_offset = u * total; n_steps = 1;
        } else {
            last_dist = (h + total) - (float)n_steps * h;
            t_offset = u * h;
        }
    }
    {
        float *row = a.dists + (size_t)ray * S;
        for (int s = lane; s < S; s += 64) __builtin_nontemporal_store(s < n_steps - 1 ? h : (s == n_steps - 1 ? last_dist : 0.0f), row + s);
    }

    if (NTX_DBG_SKIP(a, 32)) return;
    // ---- first step of every gap (the loop of instancer.cpp:870-1010 without its body) -------------------------------------
    // The reference emits, in gap j, the steps from its running counter on while t_pt(step) < t_j.  t_pt rises with the step, so
    // with F_j = the number of steps s >= 0 with t_pt(s) < t_j under the gap's segment offset, the counter behind gap j is
    // max(counter, min(F_j, n_steps)): a running maximum over the gaps that lie inside a patch.
    const int n_gaps = m + (has_mesh ? 1 : 0);
    int step = 0;
    {
        float off_q[4], off_next[4];
        int f_q[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = lane + 64 * q;
            const bool gap = j < n_gaps && cnt_q[q] > 0;
            off_q[q] = gap ? L.gs.seg.ts[seg_q[q]] : 0.0f;
            off_next[q] = (j < m && my_enter[q] && cnt_q[q] == 0) ? L.gs.seg.ts[seg_q[q] + 1] : 0.0f;   // event j starts a segment: that segment's offset
            f_q[q] = 0;
            if (gap) {
                const float tj = j == m ? t_mesh : L.ev_t[j];
                const float off = off_q[q];
                auto before = [&](int s) {
                    const float t_mu = ((float)s * h + t_offset) + off;
                    return (a.use_mean ? mean_distance(t_mu, h) : t_mu) < tj;
                };
                const float x = ((tj - off) - t_offset) / h;
                int s0 = x > 0.0f ? (x < (float)n_steps ? (int)x : n_steps) : 0;
                while (s0 > 0 && !before(s0 - 1)) --s0;
                while (s0 < n_steps && before(s0)) ++s0;
                f_q[q] = s0;
            }
        }
        __builtin_amdgcn_wave_barrier();                                     // the segment table is dead: g_step0 takes its place
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = lane + 64 * q;
            int v = f_q[q];
            for (int o = 1; o < 64; o <<= 1) {                                // running maximum over the lanes
                const int w = __shfl_up(v, o);
                if (lane >= o) v = w > v ? w : v;
            }
            v = v > step ? v : step;
            int ex = __shfl_up(v, 1);
            if (lane == 0) ex = step;
            if (j <= n_gaps) { L.gs.g_step0[j] = ex; L.g_off[j] = off_q[q]; }
            if constexpr (SHADOW || TEX) { if (j <= n_gaps) SG.g_seg[j] = (uint8_t)(seg_q[q] < 0 ? 0 : seg_q[q]); }
            step = __shfl(v, 63);
        }
        if (lane == 0) L.gs.g_step0[n_gaps] = step;
        if constexpr (SHADOW) {                                              // per segment: its first step and its offset (read off its first gap)
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j = lane + 64 * q;
                if (j < m && my_enter[q] && cnt_q[q] == 0) { SG.step0[seg_q[q] + 1] = L.gs.g_step0[j + 1]; SG.off[seg_q[q] + 1] = off_next[q]; }
            }
            if (lane == 0) SG.step0[n_starts] = step;
        }
    }
    __builtin_amdgcn_wave_barrier();

    // ---- shadow queries (instancer.cpp:861, 945-961, 1018-1027).  With N = max(min, n * total) < n_pts the ray has shadow SAMPLES: a
    // segment of length len gets max(min, N * len / total) of them spaced len / (n - 1) from its start (sample k of segment i = bit
    // base[i] + k), and a step takes the nearer of the two around it; else every STEP makes its own query (bit = the step).  Either way
    // the queries are points o + alpha d of this ray asking along one light direction, and they are all answered here, before the steps:
    //   1. the two-level cull finds the occluders whose sphere meets the strip the shadow rays sweep (lane per occluder)
    //   2. for each, the stretch [a0, a1] of the ray from which a shadow ray can be stopped by it -- where the shadow ray meets a face
    //      is linear in alpha -- widened well beyond rounding; mostly empty (a low sun meets a patch's top face from few places)
    //   3. the queries inside [a0, a1] pair up with the occluder; the pairs of 64 occluders are dealt out lane per pair, each gets the
    //      EXACT test of the restatement, and a hit sets the query's bit.
    // Culls 1 and 2 only ever drop pairs the exact test would turn down, so the bits are those of testing every pair.
    bool interpolate = false;
    bool overflow_shadow = false;
    if constexpr (SHADOW) {
        if (total > 0.0f) {
            const uint32_t n_ray = (uint32_t)((float)(uint32_t)a.n_shadow * total);
            const uint32_t n_shadow = n_ray > (uint32_t)a.min_shadow ? n_ray : (uint32_t)a.min_shadow;
            interpolate = n_shadow < (uint32_t)S;
            int n_targets = step;
            float a_min = INFINITY, a_max = -INFINITY;
            if (interpolate) {
                int entries = 0;
                for (int i = 0; i < n_starts; ++i) {
                    const float len = SG.len[i];
                    const uint32_t ns = (uint32_t)(((float)n_shadow * len) / total);
                    const int n_seg = (int)(ns > (uint32_t)a.min_shadow ? ns : (uint32_t)a.min_shadow);
                    const float sl = len / (float)(uint32_t)(n_seg - 1);
                    SH.sl[i] = sl;
                    SH.base[i] = (uint16_t)entries;
                    entries += n_seg + 1;
                    if (entries > MAX_SHADOW_ENTRIES) { overflow_shadow = true; entries = MAX_SHADOW_ENTRIES; }
                    const float t_first = SG.ts[i], t_last = SG.ts[i] + (float)(uint32_t)n_seg * sl;
                    a_min = fminf(a_min, fminf(t_first, t_last)); a_max = fmaxf(a_max, fmaxf(t_first, t_last));
                }
                SH.base[n_starts] = (uint16_t)entries;
                n_targets = entries;
            }
            for (int w = lane; w < (n_targets + 31) / 32; w += 64) SH.bits[w] = 0u;
            __builtin_amdgcn_wave_barrier();
            // the query's parameter on the ray, as the steps below and the restatement work it out
            auto target_alpha = [&](int x) -> float {
                if (interpolate) {
                    int i = 0;
                    for (int q = 1; q < n_starts; ++q) i += (int)SH.base[q] <= x ? 1 : 0;
                    const int k = x - (int)SH.base[i];
                    return SG.ts[i] + (float)(uint32_t)k * SH.sl[i];
                }
                int lo = 0, hi = n_gaps;
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (L.gs.g_step0[mid] <= x) lo = mid; else hi = mid;
                }
                const float t_mu = ((float)x * h + t_offset) + L.g_off[lo];
                return a.use_mean ? mean_distance(t_mu, h) : t_mu;
            };
            if (!interpolate && n_targets > 0) { a_min = target_alpha(0); a_max = target_alpha(n_targets - 1); }
            if (n_targets > 0 && a_min <= a_max) {
                const float lx0 = L.par[a.light_dir_idx], ly0 = L.par[a.light_dir_idx + 1], lz0 = L.par[a.light_dir_idx + 2];
                const float pad = 1e-5f * (fabsf(a_min) + fabsf(a_max)) + 1e-30f;
                const Strip st(ox, oy, oz, dx, dy, dz, lx0, ly0, lz0, a_min - pad, a_max + pad);
                const float amax = fmaxf(fabsf(a_min), fabsf(a_max)) + pad;
                // the queries that can lie in [a0, a1], as a range of bits (generous by a sample or step at either end)
                auto target_range = [&](float a0, float a1, int &x0, int &x1) {
                    x0 = 0x7fffffff; x1 = -1;
                    for (int i = 0; i < n_starts; ++i) {
                        int lo_x, hi_x;                                                      // the segment's queries lo_x .. hi_x, at first + k * spacing
                        float first, spacing, lead;
                        if (interpolate) {
                            lo_x = (int)SH.base[i]; hi_x = (int)SH.base[i + 1] - 1;
                            if (hi_x >= n_targets) hi_x = n_targets - 1;
                            first = SG.ts[i]; spacing = SH.sl[i]; lead = 1.0f;
                        } else {
                            lo_x = SG.step0[i]; hi_x = SG.step0[i + 1] - 1;
                            first = ((float)lo_x * h + t_offset) + SG.off[i]; spacing = h; lead = a.use_mean ? 4.0f : 2.0f;   // (a mean distance lies up to 2 h behind t_mu)
                        }
                        if (hi_x < lo_x) continue;
                        int k0 = 0, k1 = hi_x - lo_x;
                        if (spacing > 0.0f) {
                            const float q0 = (a0 - first) / spacing - lead, q1 = (a1 - first) / spacing + 2.0f;
                            if (q1 < 0.0f || q0 > (float)k1) continue;
                            k0 = q0 > 0.0f ? (int)q0 : 0;
                            k1 = q1 < (float)k1 ? (int)q1 : k1;
                        }
                        x0 = lo_x + k0 < x0 ? lo_x + k0 : x0;
                        x1 = lo_x + k1 > x1 ? lo_x + k1 : x1;
                    }
                };
                int n_cand = 0;
                const uint64_t ltm = (1ull << lane) - 1ull;
                // the exact tests of the first `count` (<= 64) occluders waiting in the list, lane per (occluder, query) pair
                auto flush = [&](bool is_tri, int count) {
                    int id = 0, x0 = 0, cnt = 0;
                    if (lane < count) {
                        int x1;
                        id = SH.cand_id[lane];
                        target_range(SH.cand_a0[lane], SH.cand_a1[lane], x0, x1);
                        cnt = x1 >= x0 ? x1 - x0 + 1 : 0;
                    }
                    int incl = cnt;
                    for (int o = 1; o < 64; o <<= 1) { const int w = __shfl_up(incl, o); if (lane >= o) incl += w; }
                    const int total_pairs = __shfl(incl, 63);
                    for (int p0 = 0; p0 < total_pairs; p0 += 64) {
                        const int pp = p0 + lane;
                        int cl = 0;                                                          // the first occluder whose inclusive count exceeds pp
#pragma unroll
                        for (int bit = 32; bit > 0; bit >>= 1) { const int v = __shfl(incl, cl + bit - 1); if (v <= pp) cl += bit; }
                        cl = cl < 63 ? cl : 63;
                        const int pid = __shfl(id, cl), px0 = __shfl(x0, cl), pincl = __shfl(incl, cl), pcnt = __shfl(cnt, cl);
                        if (pp < total_pairs) {
                            const int x = px0 + (pp - (pincl - pcnt));
                            const float tk = target_alpha(x);
                            const float px = ox + tk * dx, py = oy + tk * dy, pz = oz + tk * dz;
                            const bool occ = is_tri ? triangle_occludes(a.tris, a.kind, pid, px, py, pz, lx0, ly0, lz0)
                                                    : instance_occludes(a.mats, a.box, pid, px, py, pz, lx0, ly0, lz0);
                            if (occ) atomicOr(&SH.bits[x >> 5], 1u << (x & 31));
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    const int rest = n_cand - count;                                         // (< 64) the others move to the front
                    int rid = 0; float r0 = 0.0f, r1 = 0.0f;
                    if (lane < rest) { rid = SH.cand_id[count + lane]; r0 = SH.cand_a0[count + lane]; r1 = SH.cand_a1[count + lane]; }
                    __builtin_amdgcn_wave_barrier();
                    if (lane < rest) { SH.cand_id[lane] = rid; SH.cand_a0[lane] = r0; SH.cand_a1[lane] = r1; }
                    __builtin_amdgcn_wave_barrier();
                    n_cand = rest;
                };
                auto gather = [&](const Scene &sc, bool is_tri) {
                    walk_blocks(sc, lane, 0, 1, [&](const float *c, float r2) { return st.reaches(c, r2); }, [&](int id, bool near, const Rec &) {
                        float a0 = INFINITY, a1 = -INFINITY;
                        if (near) {
                            if (is_tri) triangle_interval(a.tris, a.kind, id, st, amax, a0, a1);
                            else instance_interval(a.mats, a.box, id, st, amax, a0, a1);
                        }
                        const bool keep = near && a0 <= a1;
                        const uint64_t km = __ballot(keep);
                        if (keep) {
                            const int at = n_cand + __builtin_popcountll(km & ltm);
                            SH.cand_id[at] = id; SH.cand_a0[at] = a0; SH.cand_a1[at] = a1;
                        }
                        n_cand += __builtin_popcountll(km);
                        __builtin_amdgcn_wave_barrier();
                        if (n_cand >= 64) flush(is_tri, 64);
                    });
                    if (n_cand > 0) flush(is_tri, n_cand);
                };
                gather(a.inst_scene, false);
                gather(a.tri_scene, true);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }

    // ---- the texture samples of the ray (instancer.cpp:863, 989-998): spaced like the shadow samples, with their own counts; their
    // values (getParameters at the sample's point) are produced 64 at a time, in order, into a ring the steps read from (below) ----
    bool tex_interp = false;
    int tex_entries = 0, tex_produced = 0;
    if constexpr (TEX) {
        if (total > 0.0f && a.tex.n_tex > 0) {
            const uint32_t n_ray = (uint32_t)((float)(uint32_t)a.tex.n_samples * total);
            const uint32_t n_tex = n_ray > (uint32_t)a.tex.min_samples ? n_ray : (uint32_t)a.tex.min_samples;
            tex_interp = n_tex < (uint32_t)S;
            if (tex_interp) {
                for (int i = 0; i < n_starts; ++i) {
                    const float len = SG.len[i];
                    const uint32_t ns = (uint32_t)(((float)n_tex * len) / total);
                    const int n_seg = (int)(ns > (uint32_t)a.tex.min_samples ? ns : (uint32_t)a.tex.min_samples);
                    TX.sl[i] = len / (float)(uint32_t)(n_seg - 1);
                    TX.base[i] = (uint16_t)tex_entries;
                    tex_entries += n_seg + 1;
                }
                TX.base[n_starts] = (uint16_t)tex_entries;
                __builtin_amdgcn_wave_barrier();
            }
        }
    }

    if (NTX_DBG_SKIP(a, 64)) return;
    // ---- steps: lane per step (instancer.cpp:878-986) ----------------------------------------------------------------------
    const float lx = a.light_dir_idx >= 0 ? L.par[a.light_dir_idx] : 0.0f;
    const float ly = a.light_dir_idx >= 0 ? L.par[a.light_dir_idx + 1] : 0.0f;
    const float lz = a.light_dir_idx >= 0 ? L.par[a.light_dir_idx + 2] : 0.0f;
    const float lstr = a.light_strength_idx >= 0 ? L.par[a.light_strength_idx] : 0.0f;
    for (int base = 0; base < (NTX_DBG_SKIP(a, 2) ? 0 : step); base += 64) {
        const int s = base + lane;
        const bool live = s < step;
        // the gap of the step: g_step0[j] <= s < g_step0[j + 1]
        int lo = 0, hi = n_gaps;                                               // g_step0[lo] <= s < g_step0[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (L.gs.g_step0[mid] <= s) lo = mid; else hi = mid;
        }
        const int j = lo;
        const float t_mu = ((float)s * h + t_offset) + L.g_off[j];
        const float t_pt = a.use_mean ? mean_distance(t_mu, h) : t_mu;
        const float px = ox + t_pt * dx, py = oy + t_pt * dy, pz = oz + t_pt * dz;              // getPtOnRay
        // the patches the point lies in, in ascending order: b < j <= e'.  Only intervals that reach into the gaps of THESE 64
        // steps are looked at: a ballot over the intervals (lane per interval), then a loop over its set bits
        const int n_here = step - base < 64 ? step - base : 64;
        const int jlo = __builtin_amdgcn_readfirstlane(j), jhi = __builtin_amdgcn_readlane(j, n_here - 1);
        uint64_t cand[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            cand[g] = 0;
            if (64 * g >= n_int) continue;
            const int q = lane + 64 * g;
            bool c = false;
            if (q < n_int) {
                const uint32_t be = L.u.iv.be[q];
                c = (int)(be & 0xffffu) < jhi && jlo <= (int)(be >> 16);
            }
            cand[g] = __ballot(c);
        }
        auto for_each_candidate = [&](auto body) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint64_t mk = cand[g];
                while (mk) {
                    const int q = 64 * g + __builtin_ctzll(mk);
                    mk &= mk - 1;
                    const uint32_t be = L.u.iv.be[q];
                    body(q, (int)(be & 0xffffu) < j && j <= (int)(be >> 16));
                }
            }
        };
        uint32_t inst = 0;
        float weight = 1.0f;
        int cnt = 0;
        if (a.method == 0) {                                                                    // sampleRandom, :672-677
            for_each_candidate([&](int, bool in) { cnt += in ? 1 : 0; });
            const float uc = uniform01(philox4x32_10((uint32_t)s, (uint32_t)gray, (uint32_t)((uint64_t)gray >> 32), 3u, a.seed_lo, a.seed_hi));
            int pick = cnt > 1 ? (int)(uc * (float)cnt) : 0;
            pick = pick < cnt - 1 ? pick : cnt - 1;
            int seen = 0;
            for_each_candidate([&](int q, bool in) {
                if (in && seen == pick) inst = L.u.iv.id[q];
                seen += in ? 1 : 0;
            });
            weight = cnt > 1 ? (float)cnt : 1.0f;
        } else {
            float best = INFINITY;
            for_each_candidate([&](int q, bool in) {                                            // sampleNearest, :681-692
                const float ex = px - L.u.iv.ox[q], ey = py - L.u.iv.oy[q], ez = pz - L.u.iv.oz[q];
                const float dd = __builtin_sqrtf(ex * ex + (ey * ey + ez * ez));
                if (in) {
                    // the reference keeps the FIRST patch of its ascending std::set at the smallest distance (strict <, :687): the
                    // smallest id among equals, whatever order the intervals come in
                    const uint32_t id = L.u.iv.id[q];
                    if (cnt == 0 || dd < best || (dd == best && id < inst)) { inst = id; }
                    best = dd < best ? dd : best;
                    ++cnt;
                }
            });
            if (a.method == 2 && __any(cnt > 1)) {                                              // sampleNearestBlend, :696-713
                float tot = 0.0f;
                for_each_candidate([&](int q, bool in) {
                    const float ex = px - L.u.iv.ox[q], ey = py - L.u.iv.oy[q], ez = pz - L.u.iv.oz[q];
                    const float w = (a.blend_range + best) - __builtin_sqrtf(ex * ex + (ey * ey + ez * ez));
                    if (in) tot = tot + (w > 0.0f ? w : 0.0f);
                });
                const float uc = uniform01(philox4x32_10((uint32_t)s, (uint32_t)gray, (uint32_t)((uint64_t)gray >> 32), 3u, a.seed_lo, a.seed_hi));
                const float target = uc * tot;
                float acc = 0.0f, wp = 0.0f;
                bool found = false;
                int seen = 0;
                uint32_t pick_id = inst;
                for_each_candidate([&](int q, bool in) {
                    const float ex = px - L.u.iv.ox[q], ey = py - L.u.iv.oy[q], ez = pz - L.u.iv.oz[q];
                    float w = (a.blend_range + best) - __builtin_sqrtf(ex * ex + (ey * ey + ez * ez));
                    w = w > 0.0f ? w : 0.0f;
                    if (in) {
                        acc = acc + w;
                        ++seen;
                        if (!found && (target < acc || seen == cnt)) { found = true; pick_id = L.u.iv.id[q]; wp = w; }
                    }
                });
                if (cnt > 1) { inst = pick_id; weight = tot / wp; }                             // 1 / probability
            }
        }
        // every lane computes its sample (lanes behind the last step work on a clamped patch index and store nothing); the rows go
        // out DENSE: element f of the 64 samples' flat [64 x 3] / [64 x P] block is fetched from lane f / 3 (f / P) by a shuffle,
        // so that a store instruction writes 256 consecutive bytes whatever the width of the row
        float mi[12], di[9];                                   // the patch's world -> patch matrix and direction map: six 16-byte loads
        {
            const f32x4 *xf = reinterpret_cast<const f32x4 *>(a.xforms) + (size_t)inst * 6;
            const f32x4 x0 = xf[0], x1 = xf[1], x2 = xf[2], x3 = xf[3], x4 = xf[4], x5 = xf[5];
            mi[0] = x0.x; mi[1] = x0.y; mi[2] = x0.z; mi[3] = x0.w; mi[4] = x1.x; mi[5] = x1.y; mi[6] = x1.z; mi[7] = x1.w;
            mi[8] = x2.x; mi[9] = x2.y; mi[10] = x2.z; mi[11] = x2.w;
            di[0] = x3.x; di[1] = x3.y; di[2] = x3.z; di[3] = x3.w; di[4] = x4.x; di[5] = x4.y; di[6] = x4.z; di[7] = x4.w; di[8] = x5.x;
        }
        // parameter textures (:910-927): the row between the two texture samples around t_pt (linear), or a query at the point itself
        float tv0[MAX_TEX_FILES] = {0.0f, 0.0f, 0.0f, 0.0f}, tv1[MAX_TEX_FILES] = {0.0f, 0.0f, 0.0f, 0.0f}, tw = 0.0f;
        if constexpr (TEX) {
            if (a.tex.n_tex > 0) {
                if (tex_interp) {
                    const int i = SG.g_seg[j];
                    const float ts = SG.ts[i], sl = TX.sl[i];
                    const int n_seg = (int)TX.base[i + 1] - (int)TX.base[i] - 1;
                    const float xk = (t_pt - ts) / sl;
                    int k = xk > 1.0f ? (xk < (float)n_seg ? (int)xk : n_seg) : 1;              // smallest k >= 1 with !(t_pt > ts + k * sl), :914-919
                    while (k > 1 && !(t_pt > ts + (float)(uint32_t)(k - 1) * sl)) --k;
                    while (k < n_seg && t_pt > ts + (float)(uint32_t)k * sl) ++k;
                    const float t0 = ts + (float)(uint32_t)(k - 1) * sl;
                    tw = (t_pt - t0) / sl;                                                      // :922
                    const int e0 = (int)TX.base[i] + k - 1;
                    // samples e0 and e0 + 1 from the ring, which holds [tex_produced - TEX_RING, tex_produced): the steps need them in
                    // ascending order, so the ring only ever moves forward and every sample is worked out once, 64 to a pass
                    bool pending = live;
                    while (__any(pending)) {
                        int lo = pending ? e0 : 0x7fffffff;
                        for (int o = 32; o > 0; o >>= 1) { const int w = __shfl_xor(lo, o); lo = w < lo ? w : lo; }
                        while (tex_produced < tex_entries && tex_produced + 64 - TEX_RING <= lo) {
                            const int x = tex_produced + lane;
                            if (x < tex_entries) {
                                int si = 0;
                                for (int q = 1; q < n_starts; ++q) si += (int)TX.base[q] <= x ? 1 : 0;
                                const int kk = x - (int)TX.base[si];
                                const float tk = SG.ts[si] + (float)(uint32_t)kk * TX.sl[si];
                                float val[MAX_TEX_FILES];
                                texture_values(a.tex, L.par, ox + tk * dx, oy + tk * dy, oz + tk * dz, val);
#pragma unroll
                                for (int q = 0; q < MAX_TEX_FILES; ++q) TX.ring[q][x & (TEX_RING - 1)] = val[q];
                            }
                            tex_produced += 64;
                            __builtin_amdgcn_wave_barrier();
                        }
                        if (pending && e0 + 1 < tex_produced) {
#pragma unroll
                            for (int q = 0; q < MAX_TEX_FILES; ++q) { tv0[q] = TX.ring[q][e0 & (TEX_RING - 1)]; tv1[q] = TX.ring[q][(e0 + 1) & (TEX_RING - 1)]; }
                            pending = false;
                        }
                    }
                } else {
                    texture_values(a.tex, L.par, px, py, pz, tv0);                               // :926
                }
            }
        }
        float p3[3], d3[3], l3[3] = {0.0f, 0.0f, 0.0f}, lst = 0.0f;
        affine_e(mi, px, py, pz, p3);                                                           // getPt
        linear33_e(di, ndx, ndy, ndz, d3);                                                      // getDir
        if (a.light_dir_idx >= 0) {                                                             // getShadowedLightDir, :571-581
            float sx = lx, sy = ly, sz = lz;
            if (a.light_strength_idx >= 0) { sx = lx - px; sy = ly - py; sz = lz - pz; }
            normalized(sx, sy, sz);
            linear33_e(di, sx, sy, sz, l3);
            if constexpr (SHADOW) {
                bool shadowed;
                if (interpolate) {                                                              // :946-958: the nearer of the two shadow samples around t_pt
                    const int i = SG.g_seg[j];
                    const float ts = SG.ts[i], sl = SH.sl[i];
                    const int n_seg = (int)SH.base[i + 1] - (int)SH.base[i] - 1;
                    const float xk = (t_pt - ts) / sl;
                    int k = xk > 1.0f ? (xk < (float)n_seg ? (int)xk : n_seg) : 1;              // smallest k >= 1 with !(t_pt > ts + k * sl)
                    while (k > 1 && !(t_pt > ts + (float)(uint32_t)(k - 1) * sl)) --k;
                    while (k < n_seg && t_pt > ts + (float)(uint32_t)k * sl) ++k;
                    const float t0 = ts + (float)(uint32_t)(k - 1) * sl;
                    const int e = (int)SH.base[i] + ((t_pt - t0) / sl >= 0.5f ? k : k - 1);
                    shadowed = e < MAX_SHADOW_ENTRIES && ((SH.bits[e >> 5] >> (e & 31)) & 1u);
                } else {                                                                        // :959-961: a query per step, answered above
                    shadowed = live && ((SH.bits[s >> 5] >> (s & 31)) & 1u);
                }
                if (shadowed) { l3[0] = 0.0f; l3[1] = 0.0f; l3[2] = -1.0f; }
            }
        }
        if (a.light_strength_idx >= 0) {                                                        // getLightStrength, :583-588
            const float ex = lx - px, ey = ly - py, ez = lz - pz;
            const float d2 = ex * ex + (ey * ey + ez * ez);                                       // squaredNorm
            lst = (float)((double)lstr / (4 * M_PI * (double)d2 + (double)1e-6f));
        }
        const size_t k0 = (size_t)ray * S + base;
        if (live) {
            a.t[k0 + lane] = t_mu;
            a.alpha_weight[k0 + lane] = weight;
            a.instance_id[k0 + lane] = (int32_t)inst;
        }
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int f = it * 64 + lane;
            const int src = (int)(((uint32_t)f * 21846u) >> 16);                                 // f / 3 for f < 192
            const int comp = f - 3 * src;
            const float pa = __shfl(p3[0], src), pb = __shfl(p3[1], src), pc = __shfl(p3[2], src);
            const float da = __shfl(d3[0], src), db = __shfl(d3[1], src), dc = __shfl(d3[2], src);
            if (src < n_here) {
                a.pts[3 * k0 + f] = comp == 0 ? pa : (comp == 1 ? pb : pc);
                a.rays_d_map[3 * k0 + f] = comp == 0 ? da : (comp == 1 ? db : dc);
            }
        }
        if (P > 0) {
            const float inv_p = 1.0f / (float)P;
            float *prow = a.params_map + k0 * P;
            for (int it = 0; it < P; ++it) {
                const int f = it * 64 + lane;
                const int src = (int)(((float)f + 0.5f) * inv_p);                               // f / P, exact for f < 2048, P <= 32
                const int comp = f - P * src;
                const float la = __shfl(l3[0], src), lb = __shfl(l3[1], src), lc = __shfl(l3[2], src), ls = __shfl(lst, src);
                float v = L.par[comp];
                if constexpr (TEX) {
                    if (a.tex.n_tex > 0) {
                        const float ws = __shfl(tw, src);
                        float v0 = v, v1 = v;
#pragma unroll
                        for (int q = 0; q < MAX_TEX_FILES; ++q) {
                            const float a0 = __shfl(tv0[q], src), a1 = __shfl(tv1[q], src);
                            if (q < a.tex.n_tex && comp == a.tex.par_idx[q]) { v0 = a0; v1 = a1; }
                        }
                        v = tex_interp ? v0 * (1.0f - ws) + v1 * ws : v0;                       // :923 (every column), :926
                    }
                }
                const int lc_i = comp - a.light_dir_idx;
                if (a.light_dir_idx >= 0 && lc_i >= 0 && lc_i < 3) v = lc_i == 0 ? la : (lc_i == 1 ? lb : lc);
                if (comp == a.light_strength_idx) v = ls;
                if (src < n_here) prow[f] = v;
            }
        }
    }

    // ---- what instancer.pyx:41-50 leaves in the rows behind the last emitted step -----------------------------------------
    if (!NTX_DBG_SKIP(a, 1)) {
        const size_t base = (size_t)ray * S;
        // (sparse: only the rows a consumer that goes by dists > 0 reads -- the steps the loop above did not reach, if any)
        const int E = a.sparse ? (n_steps > step ? n_steps : step) : S;
        fill_pattern(a.t + base, step, E, 1, lane, [](int) { return 0.0f; });
        fill_pattern(a.alpha_weight + base, step, E, 1, lane, [](int) { return 1.0f; });
        fill_pattern(reinterpret_cast<float *>(a.instance_id + base), step, E, 1, lane, [](int) { return 0.0f; });
        fill_pattern(a.pts + base * 3, step * 3, E * 3, 1, lane, [](int) { return 0.0f; });
        fill_pattern(a.rays_d_map + base * 3, step * 3, E * 3, 3, lane, [&](int q) { return q == 0 ? dx : (q == 1 ? dy : dz); });
        if (P > 0) fill_pattern(a.params_map + base * P, step * P, E * P, P, lane, [&](int q) { return L.par[q]; });
    }
    if (lane == 0) {
        // the closing sample (:1013-1027): the instancer mesh is black and opaque, no mesh = nothing
        a.color_last[3 * ray] = 0.0f; a.color_last[3 * ray + 1] = 0.0f; a.color_last[3 * ray + 2] = 0.0f;
        a.alpha_last[ray] = has_mesh ? 1.0f : 0.0f;
        a.hit[ray] = any_hit ? 1 : 0;
        if (a.status && (overflow_hits || overflow_shadow)) atomicOr(a.status, (overflow_hits ? 1 : 0) | (overflow_shadow ? 4 : 0));
    }
}

// four kernels around the one body.  Rays without shadow queries run at 5 waves per SIMD (what the 29.5 KB of LDS per workgroup allow;
// the compiler is told to stay within 102 registers for it); the shadow flavour, with its tables in LDS, fits 4 workgroups per CU and
// is held to the 128 registers of 4 waves per SIMD.  The texture flavours add the sample ring and the closest-point walk.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 8))) void inst_march_kernel(MarchArgs a) {
    __shared__ MarchLds<false, false> lds[4];
    march_ray<false, false>(a, lds);
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void inst_march_shadow_kernel(MarchArgs a) {
    __shared__ MarchLds<true, false> lds[4];
    march_ray<true, false>(a, lds);
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void inst_march_tex_kernel(MarchArgs a) {
    __shared__ MarchLds<false, true> lds[4];
    march_ray<false, true>(a, lds);
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 8))) void inst_march_shadow_tex_kernel(MarchArgs a) {
    __shared__ MarchLds<true, true> lds[4];
    march_ray<true, true>(a, lds);
}

// The closing sample of a ray that ends on an AUXILIARY mesh (instancer.cpp:1013-1022 -> shadeMesh, :716-743): wave per ray; the hit
// triangle's barycentrics once more, the interpolated vertex normal, one shadow query from just above the surface (`occluded`: all
// lanes the same point), diffuse + 0.2 ambient on albedo 0.8 or the mesh's texture.  Rays that end on the instancer mesh keep the black the march kernel wrote.
struct ShadeArgs {
    const float *normals; const int32_t *faces; const uint8_t *kind;
    const float *uv; const int32_t *face_tex;        // texture coordinates per vertex; per face the first of its mesh's three channel tables, -1 = none
    const float *texels; const TexTable *table;
};
__global__ __launch_bounds__(256) void inst_shade_kernel(MarchArgs a, ShadeArgs sh) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int ray = blockIdx.x * 4 + wave;
    if (ray >= a.n_rays) return;
    const unsigned long long key = a.t_mesh[ray];
    if ((uint32_t)(key >> 32) == INF_BITS) return;
    const int f = (int)(uint32_t)key;
    if ((sh.kind[f] & 1) == 0) return;
    const float tm = __builtin_bit_cast(float, (uint32_t)(key >> 32));
    const float o[3] = {a.rays_o[3 * ray], a.rays_o[3 * ray + 1], a.rays_o[3 * ray + 2]};
    const float d[3] = {a.rays_d[3 * ray], a.rays_d[3 * ray + 1], a.rays_d[3 * ray + 2]};
    const float *tr = a.tris + (size_t)f * 13;
    const float *v0 = tr, *e1 = tr + 3, *e2 = tr + 6;
    const float p[3] = {d[1] * e2[2] - d[2] * e2[1], d[2] * e2[0] - d[0] * e2[2], d[0] * e2[1] - d[1] * e2[0]};
    const float det = (e1[0] * p[0] + e1[1] * p[1]) + e1[2] * p[2];
    const float inv_det = 1.0f / det;
    const float s[3] = {o[0] - v0[0], o[1] - v0[1], o[2] - v0[2]};
    const float u = ((s[0] * p[0] + s[1] * p[1]) + s[2] * p[2]) * inv_det;
    const float q[3] = {s[1] * e1[2] - s[2] * e1[1], s[2] * e1[0] - s[0] * e1[2], s[0] * e1[1] - s[1] * e1[0]};
    const float v = ((d[0] * q[0] + d[1] * q[1]) + d[2] * q[2]) * inv_det;
    const float w0 = (1.0f - u) - v;                                                   // Vector3f(1 - u - v, u, v), :1020
    const int32_t *fv = sh.faces + (size_t)f * 3;
    float n[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) n[c] = (sh.normals[3 * fv[0] + c] * w0 + sh.normals[3 * fv[1] + c] * u) + sh.normals[3 * fv[2] + c] * v;
    normalized(n[0], n[1], n[2]);
    const float *par = a.params + (size_t)ray * a.n_params;
    const float lx = par[a.light_dir_idx], ly = par[a.light_dir_idx + 1], lz = par[a.light_dir_idx + 2];
    const float px = (o[0] + tm * d[0]) + n[0] * 1e-6f, py = (o[1] + tm * d[1]) + n[1] * 1e-6f, pz = (o[2] + tm * d[2]) + n[2] * 1e-6f;
    const bool dark = occluded_point(a, lane, px, py, pz, lx, ly, lz);
    float nlx = lx, nly = ly, nlz = lz;
    normalized(nlx, nly, nlz);
    const float nd = n[0] * nlx + (n[1] * nly + n[2] * nlz);                          // n.dot(dir.normalized())
    const float diffuse = dark ? 0.0f : 1.0f * (nd > 0.0f ? nd : 0.0f);
    const float sum = diffuse + 0.2f;
    const float shade = sum < 1.0f ? sum : 1.0f;
    float albedo[3] = {0.8f, 0.8f, 0.8f};                                             // :728
    const int tex = sh.face_tex ? sh.face_tex[f] : -1;
    if (tex >= 0) {                                                                    // :730-732: the mesh's texture at the hit's texture coordinates
        const float tu = (sh.uv[2 * fv[0]] * w0 + sh.uv[2 * fv[1]] * u) + sh.uv[2 * fv[2]] * v;
        const float tv = (sh.uv[2 * fv[0] + 1] * w0 + sh.uv[2 * fv[1] + 1] * u) + sh.uv[2 * fv[2] + 1] * v;
#pragma unroll
        for (int c = 0; c < 3; ++c) albedo[c] = interpolate2d(sh.texels, sh.table[tex + c], tu, tv);
    }
    if (lane == 0) { a.color_last[3 * ray] = albedo[0] * shade; a.color_last[3 * ray + 1] = albedo[1] * shade; a.color_last[3 * ray + 2] = albedo[2] * shade; }
}

}   // namespace ntx_inst

// ---------------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------------
struct ntx_instancer {
    int device = 0;
    ntx_instancer_desc desc{};
    int64_t n_inst = 0, n_tri = 0, cap_rays = 0;
    std::vector<float> h_mats, h_dirs, h_org;          // world -> patch [K,12], direction maps [K,9], origins [K,3]
    std::vector<float> h_spheres;                      // [K,4] centre and squared radius of the instanced box, world
    float *d_mats = nullptr, *d_org = nullptr, *d_tris = nullptr, *d_spheres = nullptr, *d_xforms = nullptr;
    uint32_t *d_count = nullptr;
    unsigned long long *d_tmesh = nullptr;
    float *d_normals = nullptr; int32_t *d_faces = nullptr; uint8_t *d_kind = nullptr; bool has_aux = false;   // auxiliary meshes
    int64_t n_mesh_vertices = 0;
    int top_min = 4;                                   // NERFTEX_INST_FORCE_BLOCKS=1 (read at create): the block level of the cull also for tiny scenes (tests)
    // the cull's hierarchy: objects in Morton order behind a permutation, a sphere around every block of 64 of them
    int32_t *d_iperm = nullptr, *d_tperm = nullptr; float *d_ibs = nullptr, *d_tbs = nullptr, *d_irecs = nullptr, *d_trecs = nullptr;
    float *d_uv = nullptr; int32_t *d_face_tex = nullptr; float *d_atexels = nullptr; ntx_inst::TexTable *d_atable = nullptr;   // their textures
    // parameter textures on the instancer mesh: texels + tables, the mesh in its grid
    ntx_inst::TexArgs tex{};
    float *d_ptexels = nullptr, *d_gtris = nullptr, *d_face_uv = nullptr; ntx_inst::TexTable *d_ptable = nullptr; int32_t *d_cell_start = nullptr, *d_cand = nullptr;
    uint4 *d_hits = nullptr;
};

namespace {

// 4x4 inverse in double (Gauss-Jordan, partial pivoting); false = singular
bool invert4(const float *m, double *out) {
    double a[4][8];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 8; ++c) a[r][c] = c < 4 ? (double)m[4 * r + c] : (c - 4 == r ? 1.0 : 0.0);
    for (int c = 0; c < 4; ++c) {
        int piv = c;
        for (int r = c + 1; r < 4; ++r)
            if (std::fabs(a[r][c]) > std::fabs(a[piv][c])) piv = r;
        if (a[piv][c] == 0.0) return false;
        if (piv != c)
            for (int k = 0; k < 8; ++k) std::swap(a[piv][k], a[c][k]);
        const double d = a[c][c];
        for (int k = 0; k < 8; ++k) a[c][k] /= d;
        for (int r = 0; r < 4; ++r) {
            if (r == c) continue;
            const double f = a[r][c];
            if (f != 0.0)
                for (int k = 0; k < 8; ++k) a[r][k] -= f * a[c][k];
        }
    }
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) out[4 * r + c] = a[r][c + 4];
    return true;
}

void release(ntx_instancer *p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    for (void *q : {(void *)p->d_mats, (void *)p->d_org, (void *)p->d_tris, (void *)p->d_spheres, (void *)p->d_xforms, (void *)p->d_normals, (void *)p->d_faces, (void *)p->d_kind, (void *)p->d_count, (void *)p->d_tmesh, (void *)p->d_hits,
                    (void *)p->d_uv, (void *)p->d_face_tex, (void *)p->d_atexels, (void *)p->d_atable, (void *)p->d_ptexels, (void *)p->d_gtris, (void *)p->d_face_uv, (void *)p->d_ptable, (void *)p->d_cell_start, (void *)p->d_cand,
                    (void *)p->d_iperm, (void *)p->d_tperm, (void *)p->d_ibs, (void *)p->d_tbs, (void *)p->d_irecs, (void *)p->d_trecs})
        if (q) (void)hipFree(q);
    delete p;
}

// Morton order of sphere centres + a sphere around every block of 64 consecutive ones: what the kernels' two-level cull walks.
// spheres: n entries of `stride` floats, (centre, r^2) at the front.
void build_blocks(const float *spheres, int64_t n, int stride, std::vector<int32_t> &perm, std::vector<float> &bs) {
    perm.resize((size_t)n);
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int64_t i = 0; i < n; ++i)
        for (int c = 0; c < 3; ++c) { const double x = spheres[i * stride + c]; if (x < lo[c]) lo[c] = x; if (x > hi[c]) hi[c] = x; }
    std::vector<std::pair<uint64_t, int32_t>> key((size_t)n);
    auto spread = [](uint64_t v) { v &= 0x1fffff; v = (v | v << 32) & 0x1f00000000ffffULL; v = (v | v << 16) & 0x1f0000ff0000ffULL; v = (v | v << 8) & 0x100f00f00f00f00fULL;
                                   v = (v | v << 4) & 0x10c30c30c30c30c3ULL; v = (v | v << 2) & 0x1249249249249249ULL; return v; };
    for (int64_t i = 0; i < n; ++i) {
        uint64_t code = 0;
        for (int c = 0; c < 3; ++c) {
            const double w = hi[c] > lo[c] ? (spheres[i * stride + c] - lo[c]) / (hi[c] - lo[c]) : 0.0;
            const uint64_t q = (uint64_t)(w * 2097151.0);
            code |= spread(std::isfinite(w) ? q : 0) << c;
        }
        key[(size_t)i] = {code, (int32_t)i};
    }
    std::sort(key.begin(), key.end());
    for (int64_t i = 0; i < n; ++i) perm[(size_t)i] = key[(size_t)i].second;
    const int64_t nb = (n + 63) / 64;
    bs.assign((size_t)(nb ? nb : 1) * 4, 0.0f);
    for (int64_t b = 0; b < nb; ++b) {
        const int64_t i0 = b * 64, i1 = i0 + 64 < n ? i0 + 64 : n;
        double c[3] = {0, 0, 0}, r = 0.0;
        for (int64_t i = i0; i < i1; ++i)
            for (int k = 0; k < 3; ++k) c[k] += spheres[(int64_t)perm[(size_t)i] * stride + k];
        for (int k = 0; k < 3; ++k) c[k] /= (double)(i1 - i0);
        for (int64_t i = i0; i < i1; ++i) {
            const float *sp = spheres + (int64_t)perm[(size_t)i] * stride;
            double d2 = 0.0;
            for (int k = 0; k < 3; ++k) d2 += (sp[k] - c[k]) * (sp[k] - c[k]);
            const double rr = std::sqrt(d2) + std::sqrt((double)sp[3]);
            r = rr > r ? rr : r;
        }
        for (int k = 0; k < 3; ++k) bs[(size_t)b * 4 + k] = (float)c[k];
        bs[(size_t)b * 4 + 3] = (float)(r * r * 1.002 + 1e-12);
    }
}

int upload_blocks(const std::vector<int32_t> &perm, const std::vector<float> &bs, const std::vector<float> &recs, int32_t **d_perm, float **d_bs, float **d_recs) {
    for (void **q : {(void **)d_perm, (void **)d_bs, (void **)d_recs})
        if (*q) { (void)hipFree(*q); *q = nullptr; }
    INST_TRY(hipMalloc((void **)d_recs, (recs.empty() ? 16 : recs.size()) * sizeof(float)));
    if (!recs.empty()) INST_TRY(hipMemcpy(*d_recs, recs.data(), recs.size() * sizeof(float), hipMemcpyHostToDevice));
    INST_TRY(hipMalloc((void **)d_perm, (perm.empty() ? 1 : perm.size()) * sizeof(int32_t)));
    if (!perm.empty()) INST_TRY(hipMemcpy(*d_perm, perm.data(), perm.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    INST_TRY(hipMalloc((void **)d_bs, bs.size() * sizeof(float)));
    INST_TRY(hipMemcpy(*d_bs, bs.data(), bs.size() * sizeof(float), hipMemcpyHostToDevice));
    return NTX_OK;
}

int reserve(ntx_instancer *p, int64_t max_rays) {
    if (max_rays <= p->cap_rays) return NTX_OK;
    INST_TRY(hipSetDevice(p->device));
    if (p->d_count) { (void)hipFree(p->d_count); p->d_count = nullptr; }
    if (p->d_tmesh) { (void)hipFree(p->d_tmesh); p->d_tmesh = nullptr; }
    if (p->d_hits) { (void)hipFree(p->d_hits); p->d_hits = nullptr; }
    p->cap_rays = 0;
    INST_TRY(hipMalloc((void **)&p->d_count, (size_t)max_rays * sizeof(uint32_t)));
    INST_TRY(hipMalloc((void **)&p->d_tmesh, (size_t)max_rays * sizeof(unsigned long long)));
    INST_TRY(hipMalloc((void **)&p->d_hits, (size_t)max_rays * ntx_inst::MAX_HITS * sizeof(uint4)));
    p->cap_rays = max_rays;
    return NTX_OK;
}

}   // namespace

extern "C" {

int ntx_instancer_create(const ntx_instancer_desc *desc, const float *transformations, int64_t n_instances, int device,
                         ntx_instancer **out) {
    if (!out) return ntx_set_error(NTX_E_INVALID, "out is NULL");
    *out = nullptr;
    if (!desc || desc->size < sizeof(ntx_instancer_desc)) return ntx_set_error(NTX_E_INVALID, "ntx_instancer_desc is NULL or its size field is not sizeof(ntx_instancer_desc)");
    if (n_instances < 0 || n_instances >= (1 << 29) || (n_instances > 0 && !transformations)) return ntx_set_error(NTX_E_INVALID, "bad instance list (0 <= n_instances < 2^29)");
    if (desc->n_parameters < 0 || desc->n_parameters > ntx_inst::MAX_PARAMS) return ntx_set_error(NTX_E_INVALID, "n_parameters %d outside [0, %d]", desc->n_parameters, ntx_inst::MAX_PARAMS);
    if (desc->instance_sample_method < 0 || desc->instance_sample_method > 2) return ntx_set_error(NTX_E_INVALID, "instance_sample_method %d is not 0 (random), 1 (nearest) or 2 (nearest_blend)", desc->instance_sample_method);
    const int ld = desc->light_dir_parameter_idx, ls = desc->light_strength_parameter_idx;
    if (ld < -1 || (ld >= 0 && ld + 3 > desc->n_parameters) || ls < -1 || ls >= desc->n_parameters || (ls >= 0 && ld < 0))
        return ntx_set_error(NTX_E_INVALID, "light parameter indices (%d, %d) do not fit %d parameters", ld, ls, desc->n_parameters);
    if (desc->cast_shadow_rays && (desc->min_shadow_samples < 2 || desc->n_shadow_samples < 0))
        return ntx_set_error(NTX_E_INVALID, "cast_shadow_rays needs min_shadow_samples >= 2 (the spacing is length / (n - 1), instancer.cpp:1021) and n_shadow_samples >= 0");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ntx_set_error(NTX_E_NODEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return ntx_set_error(NTX_E_INVALID, "device %d out of range [0,%d)", device, ndev);
    ntx_instancer *p = new ntx_instancer();
    p->device = device; p->desc = *desc; p->n_inst = n_instances;
    { const char *force = getenv("NERFTEX_INST_FORCE_BLOCKS"); if (force && force[0] == '1') p->top_min = -1; }
    p->h_mats.resize((size_t)n_instances * 12); p->h_dirs.resize((size_t)n_instances * 9); p->h_org.resize((size_t)n_instances * 3); p->h_spheres.resize((size_t)n_instances * 4);
    for (int64_t k = 0; k < n_instances; ++k) {                        // AddInstance, instancer.cpp:124-141
        const float *m = transformations + k * 16;
        double inv[16];
        if (!invert4(m, inv)) { delete p; return ntx_set_error(NTX_E_INVALID, "transformation %lld is singular", (long long)k); }
        for (int i = 0; i < 12; ++i) p->h_mats[k * 12 + i] = (float)inv[i];
        {   // sphere around the box's eight corners in world coordinates
            double c[3], r2 = 0.0;
            const double mid[3] = {0.5 * ((double)desc->b_0[0] + desc->b_1[0]), 0.5 * ((double)desc->b_0[1] + desc->b_1[1]), 0.5 * ((double)desc->b_0[2] + desc->b_1[2])};
            for (int r = 0; r < 3; ++r) c[r] = m[4 * r] * mid[0] + m[4 * r + 1] * mid[1] + m[4 * r + 2] * mid[2] + m[4 * r + 3];
            for (int corner = 0; corner < 8; ++corner) {
                const double q[3] = {corner & 1 ? desc->b_1[0] : desc->b_0[0], corner & 2 ? desc->b_1[1] : desc->b_0[1], corner & 4 ? desc->b_1[2] : desc->b_0[2]};
                double d2 = 0.0;
                for (int r = 0; r < 3; ++r) {
                    const double w = m[4 * r] * q[0] + m[4 * r + 1] * q[1] + m[4 * r + 2] * q[2] + m[4 * r + 3] - c[r];
                    d2 += w * w;
                }
                r2 = d2 > r2 ? d2 : r2;
            }
            for (int r = 0; r < 3; ++r) p->h_spheres[k * 4 + r] = (float)c[r];
            p->h_spheres[k * 4 + 3] = (float)(r2 * 1.002 + 1e-12);
        }
        for (int r = 0; r < 3; ++r) {                                  // block<3,3>.transpose().rowwise().normalized()
            const double c0 = m[r], c1 = m[4 + r], c2 = m[8 + r];
            const double n = std::sqrt(c0 * c0 + c1 * c1 + c2 * c2);
            p->h_dirs[k * 9 + 3 * r] = (float)(c0 / n); p->h_dirs[k * 9 + 3 * r + 1] = (float)(c1 / n); p->h_dirs[k * 9 + 3 * r + 2] = (float)(c2 / n);
            p->h_org[k * 3 + r] = m[4 * r + 3];
        }
    }
    auto up = [&](float **dst, const std::vector<float> &src) -> int {
        const size_t bytes = (src.empty() ? 1 : src.size()) * sizeof(float);
        INST_TRY(hipMalloc((void **)dst, bytes));
        if (!src.empty()) INST_TRY(hipMemcpy(*dst, src.data(), src.size() * sizeof(float), hipMemcpyHostToDevice));
        return NTX_OK;
    };
    int rc = hipSetDevice(device) == hipSuccess ? NTX_OK : ntx_set_error(NTX_E_HIP, "hipSetDevice(%d) failed", device);
    if (rc == NTX_OK) rc = up(&p->d_mats, p->h_mats);
    if (rc == NTX_OK) rc = up(&p->d_org, p->h_org);
    if (rc == NTX_OK) rc = up(&p->d_spheres, p->h_spheres);
    if (rc == NTX_OK) {
        std::vector<float> xf((size_t)n_instances * 24, 0.0f);
        for (int64_t k = 0; k < n_instances; ++k) {
            std::memcpy(xf.data() + k * 24, p->h_mats.data() + k * 12, 12 * sizeof(float));
            std::memcpy(xf.data() + k * 24 + 12, p->h_dirs.data() + k * 9, 9 * sizeof(float));
        }
        rc = up(&p->d_xforms, xf);
    }
    if (rc == NTX_OK) {
        std::vector<int32_t> perm; std::vector<float> bs;
        build_blocks(p->h_spheres.data(), n_instances, 4, perm, bs);
        std::vector<float> recs((size_t)n_instances * 16);                 // in Morton order: {sphere, world -> patch 3x4}
        for (int64_t i = 0; i < n_instances; ++i) {
            std::memcpy(&recs[(size_t)i * 16], &p->h_spheres[(size_t)perm[(size_t)i] * 4], 4 * sizeof(float));
            std::memcpy(&recs[(size_t)i * 16 + 4], &p->h_mats[(size_t)perm[(size_t)i] * 12], 12 * sizeof(float));
        }
        rc = upload_blocks(perm, bs, recs, &p->d_iperm, &p->d_ibs, &p->d_irecs);
    }
    if (rc == NTX_OK) rc = reserve(p, NTX_INSTANCER_DEFAULT_MAX_RAYS);
    if (rc != NTX_OK) { release(p); return rc; }
    *out = p;
    return NTX_OK;
}

int ntx_instancer_destroy(ntx_instancer *inst) {
    release(inst);
    return NTX_OK;
}

int ntx_instancer_reserve(ntx_instancer *inst, int64_t max_rays) {
    if (!inst) return ntx_set_error(NTX_E_INVALID, "inst is NULL");
    if (max_rays < 1 || max_rays > (1 << 24)) return ntx_set_error(NTX_E_INVALID, "max_rays %lld outside [1, 2^24]", (long long)max_rays);
    return reserve(inst, max_rays);
}

int64_t ntx_instancer_count(const ntx_instancer *inst) { return inst ? inst->n_inst : -1; }

int ntx_instancer_matrices(const ntx_instancer *inst, float *world_to_patch, float *directions, float *origins) {
    if (!inst) return ntx_set_error(NTX_E_INVALID, "inst is NULL");
    for (int64_t k = 0; k < inst->n_inst; ++k) {
        if (world_to_patch) {
            std::memcpy(world_to_patch + k * 16, inst->h_mats.data() + k * 12, 12 * sizeof(float));
            world_to_patch[k * 16 + 12] = 0.0f; world_to_patch[k * 16 + 13] = 0.0f; world_to_patch[k * 16 + 14] = 0.0f; world_to_patch[k * 16 + 15] = 1.0f;
        }
        if (directions) std::memcpy(directions + k * 9, inst->h_dirs.data() + k * 9, 9 * sizeof(float));
        if (origins) std::memcpy(origins + k * 3, inst->h_org.data() + k * 3, 3 * sizeof(float));
    }
    return NTX_OK;
}

int ntx_instancer_set_mesh(ntx_instancer *inst, const float *vertices, int64_t n_vertices, const int32_t *faces, int64_t n_faces) {
    return ntx_instancer_set_meshes(inst, vertices, nullptr, n_vertices, faces, nullptr, n_faces);
}

int ntx_instancer_set_meshes(ntx_instancer *inst, const float *vertices, const float *normals, int64_t n_vertices, const int32_t *faces,
                             const uint8_t *face_kind, int64_t n_faces) {
    if (!inst) return ntx_set_error(NTX_E_INVALID, "inst is NULL");
    bool aux = false;
    for (int64_t f = 0; face_kind && f < n_faces; ++f) aux = aux || (face_kind[f] & 1) != 0;
    if (aux && !normals) return ntx_set_error(NTX_E_INVALID, "auxiliary meshes are shaded with their vertex normals (instancer.cpp:722-724): normals is NULL");
    if (aux && inst->desc.light_dir_parameter_idx < 0) return ntx_set_error(NTX_E_INVALID, "auxiliary meshes are shaded from the light parameter (instancer.cpp:1020): the textures list has none");
    if (n_faces < 0 || n_faces > 0x7fffffff || n_vertices < 0 || (n_faces > 0 && (!vertices || !faces))) return ntx_set_error(NTX_E_INVALID, "bad mesh");
    INST_TRY(hipSetDevice(inst->device));
    std::vector<float> tris((size_t)n_faces * 13);
    for (int64_t f = 0; f < n_faces; ++f) {
        for (int c = 0; c < 3; ++c)
            if (faces[3 * f + c] < 0 || faces[3 * f + c] >= n_vertices) return ntx_set_error(NTX_E_INVALID, "face %lld names vertex %d of %lld", (long long)f, faces[3 * f + c], (long long)n_vertices);
        const float *v0 = vertices + 3 * (int64_t)faces[3 * f], *v1 = vertices + 3 * (int64_t)faces[3 * f + 1], *v2 = vertices + 3 * (int64_t)faces[3 * f + 2];
        double r2 = 0.0, cen[3];
        for (int c = 0; c < 3; ++c) {
            tris[f * 13 + c] = v0[c]; tris[f * 13 + 3 + c] = v1[c] - v0[c]; tris[f * 13 + 6 + c] = v2[c] - v0[c];
            cen[c] = ((double)v0[c] + v1[c] + v2[c]) / 3.0;
        }
        for (const float *vv : {v0, v1, v2}) {
            double d2 = 0.0;
            for (int c = 0; c < 3; ++c) d2 += (vv[c] - cen[c]) * (vv[c] - cen[c]);
            r2 = d2 > r2 ? d2 : r2;
        }
        for (int c = 0; c < 3; ++c) tris[f * 13 + 9 + c] = (float)cen[c];
        tris[f * 13 + 12] = (float)(r2 * 1.002 + 1e-12);
    }
    for (void **q : {(void **)&inst->d_tris, (void **)&inst->d_normals, (void **)&inst->d_faces, (void **)&inst->d_kind,
                     (void **)&inst->d_uv, (void **)&inst->d_face_tex, (void **)&inst->d_atexels, (void **)&inst->d_atable})
        if (*q) { (void)hipFree(*q); *q = nullptr; }
    inst->n_tri = 0; inst->has_aux = false; inst->n_mesh_vertices = 0;
    if (n_faces > 0) {
        INST_TRY(hipMalloc((void **)&inst->d_tris, tris.size() * sizeof(float)));
        INST_TRY(hipMemcpy(inst->d_tris, tris.data(), tris.size() * sizeof(float), hipMemcpyHostToDevice));
        {
            std::vector<int32_t> perm; std::vector<float> bs;
            build_blocks(tris.data() + 9, n_faces, 13, perm, bs);
            std::vector<float> recs((size_t)n_faces * 16, 0.0f);                // in Morton order: {v0, e1, e2, sphere, padding}
            for (int64_t i = 0; i < n_faces; ++i) std::memcpy(&recs[(size_t)i * 16], &tris[(size_t)perm[(size_t)i] * 13], 13 * sizeof(float));
            const int rcb = upload_blocks(perm, bs, recs, &inst->d_tperm, &inst->d_tbs, &inst->d_trecs);
            if (rcb != NTX_OK) return rcb;
        }
        {   // bit 0: auxiliary; bit 1: primID 1 of its own mesh (one mesh when no kinds are given)
            std::vector<uint8_t> kind((size_t)n_faces, 0);
            for (int64_t f = 0; f < n_faces; ++f) kind[f] = face_kind ? face_kind[f] : (uint8_t)(f == 1 ? 2 : 0);
            INST_TRY(hipMalloc((void **)&inst->d_kind, (size_t)n_faces));
            INST_TRY(hipMemcpy(inst->d_kind, kind.data(), (size_t)n_faces, hipMemcpyHostToDevice));
        }
        inst->n_mesh_vertices = n_vertices;
        if (aux) {
            INST_TRY(hipMalloc((void **)&inst->d_normals, (size_t)n_vertices * 3 * sizeof(float)));
            INST_TRY(hipMemcpy(inst->d_normals, normals, (size_t)n_vertices * 3 * sizeof(float), hipMemcpyHostToDevice));
            INST_TRY(hipMalloc((void **)&inst->d_faces, (size_t)n_faces * 3 * sizeof(int32_t)));
            INST_TRY(hipMemcpy(inst->d_faces, faces, (size_t)n_faces * 3 * sizeof(int32_t), hipMemcpyHostToDevice));
        }
        inst->n_tri = n_faces; inst->has_aux = aux;
    }
    return NTX_OK;
}

namespace {

// concatenate channel matrices into one texel buffer + table on the device
int upload_textures(const ntx_texture *textures, int n, float **d_texels, ntx_inst::TexTable **d_table) {
    std::vector<ntx_inst::TexTable> table((size_t)n);
    size_t total = 0;
    for (int i = 0; i < n; ++i) {
        if (!textures[i].texels || textures[i].rows < 1 || textures[i].cols < 1 || (int64_t)textures[i].rows * textures[i].cols > (1 << 28))
            return ntx_set_error(NTX_E_INVALID, "texture %d: NULL texels or bad size %d x %d", i, textures[i].rows, textures[i].cols);
        table[i] = ntx_inst::TexTable{(int32_t)total, textures[i].rows, textures[i].cols, 0};
        total += (size_t)textures[i].rows * textures[i].cols;
        if (total > (size_t)0x7fffffff) return ntx_set_error(NTX_E_INVALID, "textures: more than 2^31 texels");
    }
    std::vector<float> texels(total);
    for (int i = 0; i < n; ++i) std::memcpy(texels.data() + table[i].offset, textures[i].texels, (size_t)textures[i].rows * textures[i].cols * sizeof(float));
    INST_TRY(hipMalloc((void **)d_texels, (total ? total : 1) * sizeof(float)));
    INST_TRY(hipMemcpy(*d_texels, texels.data(), total * sizeof(float), hipMemcpyHostToDevice));
    INST_TRY(hipMalloc((void **)d_table, (size_t)(n ? n : 1) * sizeof(ntx_inst::TexTable)));
    INST_TRY(hipMemcpy(*d_table, table.data(), (size_t)n * sizeof(ntx_inst::TexTable), hipMemcpyHostToDevice));
    return NTX_OK;
}


// Distance of point p to triangle abc (Ericson's regions, double): what the device's closest_point_triangle computes in float32.
double point_triangle_distance(const double *p, const double *a, const double *b, const double *c) {
    double ab[3], ac[3], ap[3], bp[3], cp[3], q[3];
    for (int i = 0; i < 3; ++i) { ab[i] = b[i] - a[i]; ac[i] = c[i] - a[i]; ap[i] = p[i] - a[i]; bp[i] = p[i] - b[i]; cp[i] = p[i] - c[i]; }
    auto dot = [](const double *x, const double *y) { return x[0] * y[0] + x[1] * y[1] + x[2] * y[2]; };
    const double d1 = dot(ab, ap), d2 = dot(ac, ap), d3 = dot(ab, bp), d4 = dot(ac, bp), d5 = dot(ab, cp), d6 = dot(ac, cp);
    const double vc = d1 * d4 - d3 * d2, vb = d5 * d2 - d1 * d6, va = d3 * d6 - d5 * d4;
    if (d1 <= 0 && d2 <= 0) { for (int i = 0; i < 3; ++i) q[i] = a[i]; }
    else if (d3 >= 0 && d4 <= d3) { for (int i = 0; i < 3; ++i) q[i] = b[i]; }
    else if (d6 >= 0 && d5 <= d6) { for (int i = 0; i < 3; ++i) q[i] = c[i]; }
    else if (vc <= 0 && d1 >= 0 && d3 <= 0) { const double v = d1 / (d1 - d3); for (int i = 0; i < 3; ++i) q[i] = a[i] + v * ab[i]; }
    else if (vb <= 0 && d2 >= 0 && d6 <= 0) { const double v = d2 / (d2 - d6); for (int i = 0; i < 3; ++i) q[i] = a[i] + v * ac[i]; }
    else if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) { const double v = (d4 - d3) / ((d4 - d3) + (d5 - d6)); for (int i = 0; i < 3; ++i) q[i] = b[i] + v * (c[i] - b[i]); }
    else { const double den = 1.0 / (va + vb + vc), v = vb * den, w = vc * den; for (int i = 0; i < 3; ++i) q[i] = a[i] + v * ab[i] + w * ac[i]; }
    double e2 = 0.0;
    for (int i = 0; i < 3; ++i) e2 += (p[i] - q[i]) * (p[i] - q[i]);
    return std::isfinite(e2) ? std::sqrt(e2) : 1e300;          // (a degenerate triangle can give NaN: never the closest, on the device neither)
}

// The grid behind getParameters' point query.  A point p of cell C (centre m, half diagonal r) lies within r of m, so for every
// triangle T:  d(m, T) - r <= d(p, T) <= d(m, T) + r.  With D = the distance of m to the mesh, the triangle closest to p is at most
// D + r away, hence has d(m, T) <= D + 2 r; and only triangles with d(m, T) - r < radius can lie within the query radius.  The list of C
// is exactly that set (with a rounding margin), found by a ring walk over a coarser grid of centroids: the device then tests the list
// and nothing else, and gets what testing every triangle would give.  Cells are half an average edge wide (at most 2^21 of them) and
// cover the mesh's bounding box grown by the radius; built on all host cores.
int build_candidate_grid(const float *vertices, int64_t n_vertices, const int32_t *faces, int64_t n_faces, double radius, ntx_inst::TexArgs &T,
                         std::vector<int32_t> &start, std::vector<int32_t> &cand) {
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300}, edge = 0.0, r_max = 0.0;
    for (int64_t v = 0; v < n_vertices; ++v)
        for (int c = 0; c < 3; ++c) {
            const double x = vertices[3 * v + c];
            if (!std::isfinite(x)) return ntx_set_error(NTX_E_INVALID, "vertex %lld is not finite", (long long)v);
            lo[c] = x < lo[c] ? x : lo[c]; hi[c] = x > hi[c] ? x : hi[c];
        }
    std::vector<double> tri((size_t)n_faces * 9), cen((size_t)n_faces * 3);
    for (int64_t f = 0; f < n_faces; ++f) {
        for (int j = 0; j < 3; ++j)
            for (int c = 0; c < 3; ++c) tri[f * 9 + 3 * j + c] = vertices[3 * (int64_t)faces[3 * f + j] + c];
        for (int c = 0; c < 3; ++c) cen[3 * f + c] = (tri[f * 9 + c] + tri[f * 9 + 3 + c] + tri[f * 9 + 6 + c]) / 3.0;
        for (int j = 0; j < 3; ++j) {
            double e2 = 0.0, d2 = 0.0;
            for (int c = 0; c < 3; ++c) {
                const double e = tri[f * 9 + 3 * j + c] - tri[f * 9 + 3 * ((j + 1) % 3) + c], d = tri[f * 9 + 3 * j + c] - cen[3 * f + c];
                e2 += e * e; d2 += d * d;
            }
            edge += std::sqrt(e2);
            r_max = std::sqrt(d2) > r_max ? std::sqrt(d2) : r_max;
        }
    }
    edge /= 3.0 * (double)n_faces;
    // the coarse grid of centroids (host only): cells one average edge wide, at most 2^18
    double cc = edge > 0.0 ? edge : 1.0;
    {
        const double vol = (hi[0] - lo[0] + cc) * (hi[1] - lo[1] + cc) * (hi[2] - lo[2] + cc), fl = std::cbrt(vol / (double)(1 << 18));
        cc = cc > fl ? cc : fl;
    }
    int cdim[3]; int64_t n_coarse = 1;
    for (int c = 0; c < 3; ++c) { cdim[c] = (int)std::floor((hi[c] - lo[c]) / cc) + 1; n_coarse *= cdim[c]; }
    std::vector<int32_t> cstart((size_t)n_coarse + 1, 0), clist((size_t)n_faces), cof((size_t)n_faces);
    for (int64_t f = 0; f < n_faces; ++f) {
        int64_t id = 0, mul = 1;
        for (int c = 0; c < 3; ++c) {
            int64_t q = (int64_t)std::floor((cen[3 * f + c] - lo[c]) / cc);
            q = q < 0 ? 0 : (q >= cdim[c] ? cdim[c] - 1 : q);
            id += q * mul; mul *= cdim[c];
        }
        cof[f] = (int32_t)id; ++cstart[id + 1];
    }
    for (int64_t c = 0; c < n_coarse; ++c) cstart[c + 1] += cstart[c];
    {
        std::vector<int32_t> at(cstart.begin(), cstart.end() - 1);
        for (int64_t f = 0; f < n_faces; ++f) clist[(size_t)at[cof[f]]++] = (int32_t)f;
    }
    // the fine grid
    double cell = edge > 0.0 ? 0.5 * edge : 1.0;
    {
        const double vol = (hi[0] - lo[0] + 2 * radius + cell) * (hi[1] - lo[1] + 2 * radius + cell) * (hi[2] - lo[2] + 2 * radius + cell), fl = std::cbrt(vol / (double)(1 << 21));
        cell = cell > fl ? cell : fl;
    }
    int64_t n_cells = 1;
    T.inv_cell = (float)(1.0 / cell);
    const double cellf = 1.0 / (double)T.inv_cell;                  // the cell width the device's arithmetic implies
    for (int c = 0; c < 3; ++c) {
        T.gmin[c] = (float)(lo[c] - radius - cellf);
        T.dim[c] = (int32_t)std::ceil((hi[c] + radius + cellf - (double)T.gmin[c]) / cellf) + 1;
        n_cells *= T.dim[c];
    }
    if (n_cells > (int64_t)1 << 23) return ntx_set_error(NTX_E_INVALID, "the texture grid would have %lld cells", (long long)n_cells);
    const double r_cell = 0.5 * std::sqrt(3.0) * cellf * 1.01 + 1e-6 * (std::fabs(lo[0]) + std::fabs(hi[0]) + std::fabs(lo[1]) + std::fabs(hi[1]) + std::fabs(lo[2]) + std::fabs(hi[2]) + 1.0);
    std::vector<std::vector<int32_t>> lists((size_t)T.dim[2] * T.dim[1]);     // per row of cells: (count per cell, entries), flattened below
    std::vector<std::vector<int32_t>> counts((size_t)T.dim[2] * T.dim[1]);
    const int n_threads = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    std::atomic<int64_t> next_row{0};
    auto work = [&]() {
        std::vector<std::pair<double, int32_t>> near;
        for (;;) {
            const int64_t row = next_row.fetch_add(1);
            if (row >= (int64_t)T.dim[2] * T.dim[1]) break;
            const int z = (int)(row / T.dim[1]), y = (int)(row % T.dim[1]);
            auto &out = lists[(size_t)row]; auto &cnt = counts[(size_t)row];
            cnt.assign((size_t)T.dim[0], 0);
            for (int x = 0; x < T.dim[0]; ++x) {
                const double m[3] = {(double)T.gmin[0] + (x + 0.5) * cellf, (double)T.gmin[1] + (y + 0.5) * cellf, (double)T.gmin[2] + (z + 0.5) * cellf};
                // ring walk over the coarse grid around m: every triangle whose centroid cell lies in ring <= k; behind ring k no
                // unvisited triangle is nearer than k * cc - r_max (minus how far m lies outside its clamped cell)
                int ci[3]; double out_of = 0.0;
                for (int c = 0; c < 3; ++c) {
                    const double q = std::floor((m[c] - lo[c]) / cc);
                    ci[c] = (int)(q < 0 ? 0 : (q >= cdim[c] ? cdim[c] - 1 : q));
                    const double lo_c = lo[c] + ci[c] * cc, hi_c = lo_c + cc;
                    const double o = m[c] < lo_c ? lo_c - m[c] : (m[c] > hi_c ? m[c] - hi_c : 0.0);
                    out_of = o > out_of ? o : out_of;
                }
                near.clear();
                double best = 1e300;
                const double reach = radius + r_cell;                      // nothing beyond this can matter
                const int kmax = std::max(cdim[0], std::max(cdim[1], cdim[2]));
                for (int k = 0; k <= kmax; ++k) {
                    for (int dz = -k; dz <= k; ++dz) {
                        const int zz = ci[2] + dz; if (zz < 0 || zz >= cdim[2]) continue;
                        for (int dy = -k; dy <= k; ++dy) {
                            const int yy = ci[1] + dy; if (yy < 0 || yy >= cdim[1]) continue;
                            const bool shell = dz == -k || dz == k || dy == -k || dy == k;
                            for (int dx = -k; dx <= k; dx += (shell || k == 0) ? 1 : 2 * k) {
                                const int xx = ci[0] + dx; if (xx < 0 || xx >= cdim[0]) continue;
                                const int64_t id = ((int64_t)zz * cdim[1] + yy) * cdim[0] + xx;
                                for (int32_t e = cstart[id]; e < cstart[id + 1]; ++e) {
                                    const int32_t f = clist[(size_t)e];
                                    const double d = point_triangle_distance(m, &tri[(size_t)f * 9], &tri[(size_t)f * 9 + 3], &tri[(size_t)f * 9 + 6]);
                                    best = d < best ? d : best;
                                    near.emplace_back(d, f);
                                }
                            }
                        }
                    }
                    const double behind = k * cc - r_max - out_of;
                    const double need = std::min(best + 2.0 * r_cell, reach);
                    if (behind > need * (1.0 + 1e-9) + 1e-12) break;
                }
                const double cut = std::min(best + 2.0 * r_cell, reach) * (1.0 + 1e-9) + 1e-12;
                size_t n0 = out.size();
                for (const auto &pr : near)
                    if (pr.first <= cut && pr.first - r_cell < radius) out.push_back(pr.second);
                std::sort(out.begin() + (std::ptrdiff_t)n0, out.end());        // ascending primID
                cnt[(size_t)x] = (int32_t)(out.size() - n0);
            }
        }
    };
    {
        std::vector<std::thread> pool;
        for (int i = 1; i < n_threads; ++i) pool.emplace_back(work);
        work();
        for (auto &th : pool) th.join();
    }
    start.assign((size_t)n_cells + 1, 0);
    size_t total = 0;
    for (size_t row = 0; row < lists.size(); ++row)
        for (int x = 0; x < T.dim[0]; ++x) { start[row * T.dim[0] + x] = (int32_t)total; total += (size_t)counts[row][(size_t)x]; if (total > 0x7fffffffu) return ntx_set_error(NTX_E_INVALID, "the texture grid's lists overflow"); }
    start[(size_t)n_cells] = (int32_t)total;
    cand.resize(total);
    for (size_t row = 0; row < lists.size(); ++row)
        if (!lists[row].empty()) std::memcpy(cand.data() + start[row * T.dim[0]], lists[row].data(), lists[row].size() * sizeof(int32_t));
    return NTX_OK;
}

}   // namespace

int ntx_instancer_set_parameter_textures(ntx_instancer *inst, const float *vertices, const float *uv, int64_t n_vertices, const int32_t *faces,
                                         int64_t n_faces, float patch_max_extent, int n_textures, const int32_t *parameter_idx,
                                         const ntx_texture *textures, int min_texture_samples, int n_texture_samples) {
    using namespace ntx_inst;
    if (!inst) return ntx_set_error(NTX_E_INVALID, "inst is NULL");
    INST_TRY(hipSetDevice(inst->device));
    for (void **q : {(void **)&inst->d_ptexels, (void **)&inst->d_ptable, (void **)&inst->d_gtris, (void **)&inst->d_face_uv, (void **)&inst->d_cell_start, (void **)&inst->d_cand})
        if (*q) { (void)hipFree(*q); *q = nullptr; }
    inst->tex = TexArgs{};
    if (n_textures == 0) return NTX_OK;
    if (n_textures < 0 || n_textures > MAX_TEX_FILES) return ntx_set_error(NTX_E_UNSUPPORTED, "%d texture files in the textures list: at most %d are built", n_textures, MAX_TEX_FILES);
    if (!vertices || !uv || !faces || !parameter_idx || !textures || n_vertices < 1 || n_faces < 1 || n_faces > 0x7fffffff)
        return ntx_set_error(NTX_E_INVALID, "parameter textures need the instancer mesh with texture coordinates (getParameters, instancer.cpp:640-667)");
    if (!(patch_max_extent > 0.0f) || std::isinf(patch_max_extent)) return ntx_set_error(NTX_E_INVALID, "patch_max_extent must be finite and > 0");
    if (min_texture_samples < 2 || min_texture_samples > 512 || n_texture_samples < 0)
        return ntx_set_error(NTX_E_INVALID, "min_texture_samples must lie in [2, 512] (the spacing is length / (n - 1), instancer.cpp:992) and n_texture_samples be >= 0");
    for (int i = 0; i < n_textures; ++i)
        if (parameter_idx[i] < 0 || parameter_idx[i] >= inst->desc.n_parameters) return ntx_set_error(NTX_E_INVALID, "texture %d multiplies parameter %d of %d", i, parameter_idx[i], inst->desc.n_parameters);
    for (int64_t f = 0; f < 3 * n_faces; ++f)
        if (faces[f] < 0 || faces[f] >= n_vertices) return ntx_set_error(NTX_E_INVALID, "face %lld names vertex %d of %lld", (long long)(f / 3), faces[f], (long long)n_vertices);
    std::vector<int32_t> start, cand;
    TexArgs T{};
    {
        const int rc_grid = build_candidate_grid(vertices, n_vertices, faces, n_faces, (double)patch_max_extent, T, start, cand);
        if (rc_grid != NTX_OK) return rc_grid;
    }
    std::vector<float> gtris((size_t)n_faces * 9), fuv((size_t)n_faces * 6);
    for (int64_t f = 0; f < n_faces; ++f)
        for (int j = 0; j < 3; ++j) {
            for (int c = 0; c < 3; ++c) gtris[f * 9 + 3 * j + c] = vertices[3 * (int64_t)faces[3 * f + j] + c];
            fuv[f * 6 + 2 * j] = uv[2 * (int64_t)faces[3 * f + j]]; fuv[f * 6 + 2 * j + 1] = uv[2 * (int64_t)faces[3 * f + j] + 1];
        }
    int rc = upload_textures(textures, n_textures, &inst->d_ptexels, &inst->d_ptable);
    if (rc != NTX_OK) return rc;
    INST_TRY(hipMalloc((void **)&inst->d_gtris, gtris.size() * sizeof(float)));
    INST_TRY(hipMemcpy(inst->d_gtris, gtris.data(), gtris.size() * sizeof(float), hipMemcpyHostToDevice));
    INST_TRY(hipMalloc((void **)&inst->d_face_uv, fuv.size() * sizeof(float)));
    INST_TRY(hipMemcpy(inst->d_face_uv, fuv.data(), fuv.size() * sizeof(float), hipMemcpyHostToDevice));
    INST_TRY(hipMalloc((void **)&inst->d_cell_start, start.size() * sizeof(int32_t)));
    INST_TRY(hipMemcpy(inst->d_cell_start, start.data(), start.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    INST_TRY(hipMalloc((void **)&inst->d_cand, (cand.empty() ? 1 : cand.size()) * sizeof(int32_t)));
    INST_TRY(hipMemcpy(inst->d_cand, cand.data(), cand.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    T.n_tex = n_textures;
    for (int i = 0; i < n_textures; ++i) T.par_idx[i] = parameter_idx[i];
    T.min_samples = min_texture_samples; T.n_samples = n_texture_samples; T.radius = patch_max_extent;
    T.texels = inst->d_ptexels; T.table = inst->d_ptable; T.cell_start = inst->d_cell_start; T.cand = inst->d_cand; T.tris = inst->d_gtris; T.face_uv = inst->d_face_uv;
    T.n_faces = (int32_t)n_faces;
    inst->tex = T;
    return NTX_OK;
}

int ntx_instancer_set_mesh_textures(ntx_instancer *inst, const float *uv, int64_t n_vertices, const int32_t *face_texture, int64_t n_faces,
                                    int n_sets, const ntx_texture *textures) {
    if (!inst) return ntx_set_error(NTX_E_INVALID, "inst is NULL");
    INST_TRY(hipSetDevice(inst->device));
    for (void **q : {(void **)&inst->d_uv, (void **)&inst->d_face_tex, (void **)&inst->d_atexels, (void **)&inst->d_atable})
        if (*q) { (void)hipFree(*q); *q = nullptr; }
    if (n_sets == 0) return NTX_OK;
    if (n_sets < 0 || !uv || !face_texture || !textures) return ntx_set_error(NTX_E_INVALID, "bad mesh textures");
    if (n_vertices != inst->n_mesh_vertices || n_faces != inst->n_tri)
        return ntx_set_error(NTX_E_INVALID, "mesh textures for %lld vertices / %lld faces, ntx_instancer_set_meshes got %lld / %lld", (long long)n_vertices, (long long)n_faces,
                             (long long)inst->n_mesh_vertices, (long long)inst->n_tri);
    std::vector<int32_t> ft((size_t)n_faces);
    for (int64_t f = 0; f < n_faces; ++f) {
        if (face_texture[f] < -1 || face_texture[f] >= n_sets) return ntx_set_error(NTX_E_INVALID, "face %lld names texture set %d of %d", (long long)f, face_texture[f], n_sets);
        ft[f] = face_texture[f] < 0 ? -1 : 3 * face_texture[f];
    }
    int rc = upload_textures(textures, 3 * n_sets, &inst->d_atexels, &inst->d_atable);
    if (rc != NTX_OK) return rc;
    INST_TRY(hipMalloc((void **)&inst->d_uv, (size_t)n_vertices * 2 * sizeof(float)));
    INST_TRY(hipMemcpy(inst->d_uv, uv, (size_t)n_vertices * 2 * sizeof(float), hipMemcpyHostToDevice));
    INST_TRY(hipMalloc((void **)&inst->d_face_tex, (size_t)n_faces * sizeof(int32_t)));
    INST_TRY(hipMemcpy(inst->d_face_tex, ft.data(), (size_t)n_faces * sizeof(int32_t), hipMemcpyHostToDevice));
    return NTX_OK;
}

int ntx_instancer_model_input(ntx_instancer *inst, const float *rays_o, const float *rays_d, const float *parameters, int64_t n_rays,
                              int n_pts, float step_size, uint64_t seed, const ntx_render_opts *opts, float *rays_d_map, float *pts,
                              float *t, float *dists, float *color_last, float *alpha_last, float *alpha_weight,
                              int32_t *instance_id, uint8_t *hit, float *params_map, int32_t *status_flag, ntx_stream stream) {
    using namespace ntx_inst;
    if (!inst) return ntx_set_error(NTX_E_INVALID, "inst is NULL");
    if (n_rays < 0 || n_rays > 0x7fffffff) return ntx_set_error(NTX_E_INVALID, "n_rays %lld outside [0, 2^31)", (long long)n_rays);
    if (n_pts < 1 || n_pts > 4096) return ntx_set_error(NTX_E_INVALID, "n_pts %d outside [1, 4096]", n_pts);
    if (!(step_size > 0.0f) || std::isinf(step_size)) return ntx_set_error(NTX_E_INVALID, "step_size must be finite and > 0");
    const int P = inst->desc.n_parameters;
    if (n_rays == 0) return NTX_OK;
    if (!rays_o || !rays_d || !rays_d_map || !pts || !t || !dists || !color_last || !alpha_last || !alpha_weight || !instance_id || !hit ||
        (P > 0 && (!parameters || !params_map)))
        return ntx_set_error(NTX_E_INVALID, "NULL buffer");
    int64_t idx0 = 0, idx_stride = 0;
    uint32_t idx_run = 0xffffffffu;
    bool sparse = false;
    if (opts) {
        if (opts->size < NTX_RENDER_OPTS_V3_SIZE) return ntx_set_error(NTX_E_INVALID, "ntx_render_opts.size %u < %u", opts->size, NTX_RENDER_OPTS_V3_SIZE);
        sparse = opts->size >= sizeof(ntx_render_opts) && (opts->flags & NTX_OPT_INSTANCER_SPARSE) != 0;
        if (!(opts->ray_index0 == 0 && opts->ray_run_length == 0 && opts->ray_run_stride == 0)) {
            if (opts->ray_index0 < 0 || opts->ray_run_length < 1 || opts->ray_run_stride < opts->ray_run_length)
                return ntx_set_error(NTX_E_INVALID, "bad ray index map");
            idx0 = opts->ray_index0; idx_stride = opts->ray_run_stride;
            // a call that lies inside its first run is a plain offset (what a renderer's chunk k0 .. k0 + n passes as (k0, n, n)): it can
            // then be cut into pieces anywhere
            idx_run = (opts->ray_run_length > 0xffffffffLL || opts->ray_run_length >= n_rays) ? 0xffffffffu : (uint32_t)opts->ray_run_length;
        }
    }
    // a call longer than the workspace is cut into pieces; every piece must start on a run boundary of the index map (checked
    // here, before anything is launched: a call that fails has written nothing)
    if (n_rays > inst->cap_rays && idx_run != 0xffffffffu && inst->cap_rays % idx_run != 0)
        return ntx_set_error(NTX_E_INVALID, "ray_run_length %u must divide the reserved %lld rays when a call is split (ntx_instancer_reserve)", idx_run, (long long)inst->cap_rays);
    INST_TRY(hipSetDevice(inst->device));
    hipStream_t st = (hipStream_t)stream;
    Box box;
    for (int c = 0; c < 3; ++c) { box.b0[c] = inst->desc.b_0[c]; box.b1[c] = inst->desc.b_1[c]; }
    const int K = (int)inst->n_inst, F = (int)inst->n_tri;
    // the call's rays in pieces of the reserved workspace; a piece's local ray k is ray c0 + k of the call
    for (int64_t c0 = 0; c0 < n_rays; c0 += inst->cap_rays) {
        const int n = (int)(n_rays - c0 < inst->cap_rays ? n_rays - c0 : inst->cap_rays);
        const float *ro = rays_o + c0 * 3, *rd = rays_d + c0 * 3;
        const int tiles = (n + 63) / 64;
        const Scene isc{inst->d_irecs, 0, inst->d_iperm, inst->d_ibs, K, (K + 63) / 64, inst->top_min};
        const Scene tsc{inst->d_trecs, 9, inst->d_tperm, inst->d_tbs, F, (F + 63) / 64, inst->top_min};
        // a workgroup per 64 rays; its waves share the walk: 16 of them while that still leaves CUs idle, else 4
        const int tile_waves = tiles <= 512 ? TILE_WAVES : 4;
        if (K > 0) hipLaunchKernelGGL(inst_hits_kernel, dim3(tiles), dim3(64 * tile_waves), 0, st, ro, rd, n, isc, box, inst->d_count, inst->d_hits);
        else INST_TRY(hipMemsetAsync(inst->d_count, 0, (size_t)n * sizeof(uint32_t), st));
        if (F > 0) hipLaunchKernelGGL(inst_mesh_kernel, dim3(tiles), dim3(64 * tile_waves), 0, st, ro, rd, n, tsc, inst->d_tmesh);
        MarchArgs a{};
        a.rays_o = ro; a.rays_d = rd; a.params = P > 0 ? parameters + c0 * P : nullptr;
        a.mats = inst->d_mats; a.origins = inst->d_org; a.xforms = inst->d_xforms;
        a.count = inst->d_count; a.hits = inst->d_hits; a.t_mesh = F > 0 ? inst->d_tmesh : nullptr;
        const size_t so = (size_t)c0 * n_pts;
        a.rays_d_map = rays_d_map + so * 3; a.pts = pts + so * 3; a.t = t + so; a.dists = dists + so;
        a.color_last = color_last + c0 * 3; a.alpha_last = alpha_last + c0; a.alpha_weight = alpha_weight + so;
        a.params_map = P > 0 ? params_map + so * P : nullptr;
        a.instance_id = instance_id + so; a.hit = hit + c0; a.status = status_flag;
        a.n_rays = n; a.n_pts = n_pts; a.n_params = P; a.sparse = sparse ? 1 : 0;
        a.light_dir_idx = inst->desc.light_dir_parameter_idx; a.light_strength_idx = inst->desc.light_strength_parameter_idx;
        a.method = inst->desc.instance_sample_method; a.use_mean = inst->desc.use_mean_distance ? 1 : 0;
        a.step_size = step_size; a.blend_range = 0.2f * inst->desc.patch_scale;
        a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32);
#ifdef NTX_INST_DEBUG
        { const char *dbg = getenv("NERFTEX_INST_DEBUG"); a.debug_skip = dbg ? atoi(dbg) : 0; }   // development builds only: results are then wrong
#endif
        // the piece's rays continue the call's index map: local k of the piece = local c0 + k of the call
        if (idx_run == 0xffffffffu) { a.idx0 = idx0 + c0; a.idx_run = 0xffffffffu; a.idx_stride = 0; }
        else { a.idx0 = idx0 + (c0 / idx_run) * idx_stride; a.idx_run = idx_run; a.idx_stride = idx_stride; }
        a.tris = inst->d_tris; a.box = box; a.inst_scene = isc; a.tri_scene = tsc;
        a.min_shadow = inst->desc.min_shadow_samples; a.n_shadow = inst->desc.n_shadow_samples;
        a.kind = inst->d_kind; a.tex = inst->tex;
        const bool shadows = inst->desc.cast_shadow_rays && a.light_dir_idx >= 0, textured = inst->tex.n_tex > 0;
        if (shadows && textured) hipLaunchKernelGGL(inst_march_shadow_tex_kernel, dim3((n + 3) / 4), dim3(256), 0, st, a);
        else if (shadows) hipLaunchKernelGGL(inst_march_shadow_kernel, dim3((n + 3) / 4), dim3(256), 0, st, a);
        else if (textured) hipLaunchKernelGGL(inst_march_tex_kernel, dim3((n + 3) / 4), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(inst_march_kernel, dim3((n + 3) / 4), dim3(256), 0, st, a);
        if (F > 0 && inst->has_aux)
            hipLaunchKernelGGL(inst_shade_kernel, dim3((n + 3) / 4), dim3(256), 0, st, a,
                               ShadeArgs{inst->d_normals, inst->d_faces, inst->d_kind, inst->d_uv, inst->d_face_tex, inst->d_atexels, inst->d_atable});
    }
    INST_TRY(hipGetLastError());
    return NTX_OK;
}

}   // extern "C"
