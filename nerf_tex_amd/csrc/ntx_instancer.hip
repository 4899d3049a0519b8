// ntx_instancer.hip -- the patch instancer on the GPU: C_Instancer::GetModelInput (instancer/src/instancer.cpp:751-1037) behind
// Instancer.get_model_input (instancer/instancer.pyx:38-54).  gfx950 only.
//
// The reference traces every ray through an Embree scene of instanced boxes on ONE CPU thread (the loop at instancer.cpp:772),
// sorts the face crossings, marches the union of the boxes in steps of `step_size` and maps every sample into the patch it
// falls in.  Here the buffers of instancer.pyx:41-50 are produced in HBM, where ntx_render_instanced reads them:
//
//   inst_hits_kernel   ray per lane, instances wave-uniform (their 3x4 matrices arrive through the scalar cache, no vector
//                      memory in the loop): slab test of the ray in patch coordinates against every instance -- for the few
//                      thousand patches of a scene, all pairs on the VALUs cost less than one BVH build -- face crossings
//                      appended to the ray's hit list (<= 200, instancer.cpp:22)
//   inst_mesh_kernel   the same against the triangles of the instancer mesh (closest crossing)
//   inst_march_kernel  wave per ray, lane = marching step: rank sort of the hit list, the active set of instancer.cpp:800-826
//                      one id per lane (insert / erase by ballot + lane shift), steps handed out 64 at a time between two
//                      events, every output row written whole (emitted samples + the defaults of instancer.pyx:41-50).
//                      Bound: HBM writes, (3+3+1+1+1+1+P) * 4 bytes per (ray, step).
//
// Float32 operations are spelled in the order of oracle/instancer_oracle.py (-ffp-contract=off, IEEE divide and sqrt), so that
// the two agree bit for bit on the same instance matrices.
#include "nerftex.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

extern "C" int ntx_set_error(int code, const char *fmt, ...);   // nerftex.hip

#define INST_TRY(expr)                                                                                   \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) return ntx_set_error(NTX_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

namespace ntx_inst {

constexpr int MAX_HITS = 200;            // MAX_TOTAL_HITS, instancer.cpp:22
constexpr int SORT_SLOTS = 256;          // MAX_HITS rounded up to whole waves
constexpr int MAX_ACTIVE = 64;           // patches a sample may lie in at once: one id per lane
constexpr int MAX_PARAMS = 32;
constexpr float T_FAR = 100.0f;          // init_ray(..., 0, 100, ...), instancer.cpp:776
constexpr uint32_t INF_BITS = 0x7f800000u;

struct Box { float b0[3], b1[3]; };

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
__device__ __forceinline__ void normalized(float &x, float &y, float &z) {   // Eigen's normalized()
    const float n2 = (x * x + y * y) + z * z;
    if (n2 > 0.0f) { const float n = __builtin_sqrtf(n2); x = x / n; y = y / n; z = z / n; }
}

// ---------------------------------------------------------------------------------------------------------------------------
// all (ray, instance) pairs: what rtcIntersect1 with the all-hits filter reports (instancer.cpp:779, 526-541)
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void inst_hits_kernel(const float *__restrict__ rays_o, const float *__restrict__ rays_d, int n_rays,
                                                        const float *__restrict__ mats, int n_inst, int per_wave, Box box,
                                                        uint32_t *__restrict__ count, uint2 *__restrict__ hits) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int ray = blockIdx.x * 64 + lane;
    const bool live = ray < n_rays;
    const int r = live ? ray : n_rays - 1;
    const float ox = rays_o[3 * r], oy = rays_o[3 * r + 1], oz = rays_o[3 * r + 2];
    const float dx = rays_d[3 * r], dy = rays_d[3 * r + 1], dz = rays_d[3 * r + 2];
    const int k0 = (blockIdx.y * 4 + wave) * per_wave;
    const int k1 = k0 + per_wave < n_inst ? k0 + per_wave : n_inst;
    for (int k = k0; k < k1; ++k) {
        const float *m = mats + (size_t)k * 12;          // wave-uniform: scalar loads
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
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float tt = e ? t_out : t_in;
                if (tt > 0.0f && tt <= T_FAR) {          // tnear < t <= tfar
                    const uint32_t slot = atomicAdd(&count[ray], 1u);
                    if (slot < (uint32_t)MAX_HITS) hits[(size_t)ray * MAX_HITS + slot] = make_uint2(__builtin_bit_cast(uint32_t, tt), (uint32_t)k);
                }
            }
        }
    }
}

// closest crossing of the instancer mesh per ray (Moeller-Trumbore, no culling); tris[f] = {v0, v1 - v0, v2 - v0}
__global__ __launch_bounds__(256) void inst_mesh_kernel(const float *__restrict__ rays_o, const float *__restrict__ rays_d, int n_rays,
                                                        const float *__restrict__ tris, int n_tri, int per_wave, uint32_t *__restrict__ t_mesh) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int ray = blockIdx.x * 64 + lane;
    const bool live = ray < n_rays;
    const int r = live ? ray : n_rays - 1;
    const float o[3] = {rays_o[3 * r], rays_o[3 * r + 1], rays_o[3 * r + 2]};
    const float d[3] = {rays_d[3 * r], rays_d[3 * r + 1], rays_d[3 * r + 2]};
    const int f0 = (blockIdx.y * 4 + wave) * per_wave;
    const int f1 = f0 + per_wave < n_tri ? f0 + per_wave : n_tri;
    float best = INFINITY;
    for (int f = f0; f < f1; ++f) {
        const float *tr = tris + (size_t)f * 9;
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
        if (tt > 0.0f && tt <= T_FAR && tt < best) best = tt;
    }
    if (live && best < INFINITY) atomicMin(&t_mesh[ray], __builtin_bit_cast(uint32_t, best));   // positive floats order like their bits
}

// ---------------------------------------------------------------------------------------------------------------------------
// marching (instancer.cpp:787-1030), wave per ray
// ---------------------------------------------------------------------------------------------------------------------------
struct MarchArgs {
    const float *rays_o, *rays_d, *params;
    const float *mats, *dirs, *origins;
    const uint32_t *count; const uint2 *hits; const uint32_t *t_mesh;   // t_mesh NULL = no mesh
    float *rays_d_map, *pts, *t, *dists, *color_last, *alpha_last, *alpha_weight, *params_map;
    int32_t *instance_id; uint8_t *hit; int32_t *status;
    int n_rays, n_pts, n_params;
    int light_dir_idx, light_strength_idx, method, use_mean;
    float step_size, blend_range;
    uint32_t seed_lo, seed_hi;
    int64_t idx0, idx_stride; uint32_t idx_run;
};

// the active set of instancer.cpp:800-826 / 989-1009: ascending ids, one per lane (std::set order)
struct Active {
    uint32_t id;      // lane l < n holds the l-th smallest id
    int n;
    __device__ __forceinline__ bool toggle(uint32_t x, int lane, bool *overflow) {   // true = x was inside and left
        const uint64_t in = __ballot(lane < n && id == x);
        if (in) {
            const int pos = __builtin_ctzll(in);
            const uint32_t nxt = __shfl_down(id, 1);
            if (lane >= pos) id = nxt;
            --n;
            return true;
        }
        if (n >= MAX_ACTIVE) { *overflow = true; return false; }
        const int pos = __builtin_popcountll(__ballot(lane < n && id < x));
        const uint32_t prv = __shfl_up(id, 1);
        if (lane > pos) id = prv;
        if (lane == pos) id = x;
        ++n;
        return false;
    }
};

__device__ __forceinline__ float mean_distance(float mu_f, float hw_f) {   // instancer.cpp:746-748 (double inside)
    const double mu = mu_f, hw = hw_f;
    return (float)(mu + 2 * mu * (hw * hw) / (3 * (mu * mu) + hw * hw));
}

__global__ __launch_bounds__(256) void inst_march_kernel(MarchArgs a) {
    __shared__ float s_t[4][SORT_SLOTS];
    __shared__ uint32_t s_id[4][SORT_SLOTS];
    __shared__ float s_ts[4][MAX_HITS];
    __shared__ uint32_t s_ids[4][MAX_HITS];
    __shared__ float s_par[4][MAX_PARAMS];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int ray = blockIdx.x * 4 + wave;
    if (ray >= a.n_rays) return;
    const int S = a.n_pts, P = a.n_params;
    const float h = a.step_size;
    const float ox = a.rays_o[3 * ray], oy = a.rays_o[3 * ray + 1], oz = a.rays_o[3 * ray + 2];
    const float dx = a.rays_d[3 * ray], dy = a.rays_d[3 * ray + 1], dz = a.rays_d[3 * ray + 2];
    float ndx = dx, ndy = dy, ndz = dz;
    normalized(ndx, ndy, ndz);                                             // getDir: dir.normalized(), instancer.cpp:562
    if (lane < P) s_par[wave][lane] = a.params[(size_t)ray * P + lane];

    // ---- the hit list, sorted by (t, instID) (instancer.cpp:441-452, 787) -------------------------------------------------
    const uint32_t raw = a.count[ray];
    const int m = raw < (uint32_t)MAX_HITS ? (int)raw : MAX_HITS;
    bool overflow_hits = raw > (uint32_t)MAX_HITS, overflow_active = false;
    for (int e = lane; e < m; e += 64) {
        const uint2 hv = a.hits[(size_t)ray * MAX_HITS + e];
        s_t[wave][e] = __builtin_bit_cast(float, hv.x);
        s_id[wave][e] = hv.y;
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    for (int e = lane; e < m; e += 64) {
        const float te = s_t[wave][e];
        const uint32_t ie = s_id[wave][e];
        int rank = 0;
        for (int k = 0; k < m; ++k) {
            const float tk = s_t[wave][k];
            const uint32_t ik = s_id[wave][k];
            rank += (tk < te || (tk == te && (ik < ie || (ik == ie && k < e)))) ? 1 : 0;
        }
        s_ts[wave][rank] = te;
        s_ids[wave][rank] = ie;
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    const float *ts = s_ts[wave];
    const uint32_t *ids = s_ids[wave];
    const uint32_t tm_bits = a.t_mesh ? a.t_mesh[ray] : INF_BITS;
    const bool has_mesh = tm_bits != INF_BITS;
    const float t_mesh = __builtin_bit_cast(float, tm_bits);
    const bool any_hit = m > 0 || has_mesh;

    // ---- ray segments inside the union of the boxes (instancer.cpp:800-826) -----------------------------------------------
    Active act{0u, 0};
    float total = 0.0f, t_entry = 0.0f;
    for (int j = 0; j < m; ++j) {
        const float tj = ts[j];
        if (has_mesh && tj > t_mesh) break;             // the mesh hit sorts in front of this one (instID = invalid sorts last on ties)
        const bool had = act.n == 0;
        if (act.toggle(ids[j], lane, &overflow_active)) {
            if (act.n == 0) total = total + (tj - t_entry);
        } else if (had) {
            t_entry = tj;
        }
    }
    if (has_mesh && act.n > 0) total = total + (t_mesh - t_entry);
    act.n = 0;

    // ---- number of steps, dists (instancer.cpp:840-859) -------------------------------------------------------------------
    int n_steps = 0;
    float t_offset = 0.0f, last_dist = 0.0f;
    bool single = false;
    if (total > 0.0f) {
        const int64_t gray = global_index(a.idx0, a.idx_run, a.idx_stride, ray);
        const float u = uniform01(philox4x32_10(0u, (uint32_t)gray, (uint32_t)((uint64_t)gray >> 32), 2u, a.seed_lo, a.seed_hi));
        const uint32_t necessary = (uint32_t)(total / h);
        n_steps = necessary < (uint32_t)S ? (int)necessary : S;
        if (n_steps == 0) {
            single = true; last_dist = total; t_offset = u * total; n_steps = 1;
        } else {
            last_dist = (h + total) - (float)n_steps * h;
            t_offset = u * h;
        }
    }
    {
        float *row = a.dists + (size_t)ray * S;
        for (int s = lane; s < S; s += 64) row[s] = s < n_steps - 1 ? h : (s == n_steps - 1 ? last_dist : 0.0f);
    }
    (void)single;

    // ---- marching: steps handed out 64 at a time between two events (instancer.cpp:870-1010) ------------------------------
    const int64_t gray = global_index(a.idx0, a.idx_run, a.idx_stride, ray);
    const float lx = a.light_dir_idx >= 0 ? s_par[wave][a.light_dir_idx] : 0.0f;
    const float ly = a.light_dir_idx >= 0 ? s_par[wave][a.light_dir_idx + 1] : 0.0f;
    const float lz = a.light_dir_idx >= 0 ? s_par[wave][a.light_dir_idx + 2] : 0.0f;
    const float lstr = a.light_strength_idx >= 0 ? s_par[wave][a.light_strength_idx] : 0.0f;
    float segment_offset = 0.0f, cleared = 0.0f;
    t_entry = 0.0f;
    int step = 0;
    for (int j = 0; j <= m && step < n_steps; ++j) {
        const bool is_mesh = j == m || (has_mesh && ts[j] > t_mesh);
        if (is_mesh && !has_mesh) break;
        const float tj = is_mesh ? t_mesh : ts[j];
        while (act.n > 0) {
            const int s = step + lane;
            const float t_mu = ((float)s * h + t_offset) + segment_offset;
            const float t_pt = a.use_mean ? mean_distance(t_mu, h) : t_mu;
            const uint64_t ok = __ballot(s < n_steps && t_pt < tj);
            const int c = ~ok == 0 ? 64 : __builtin_ctzll(~ok);
            if (c == 0) break;
            // every lane computes (shuffles and readlanes stay in uniform control flow); lanes < c store
            const float px = ox + t_pt * dx, py = oy + t_pt * dy, pz = oz + t_pt * dz;          // getPtOnRay
            uint32_t inst = (uint32_t)__builtin_amdgcn_readlane((int)act.id, 0);
            float weight = 1.0f;
            if (act.n > 1) {
                if (a.method == 0) {                                                            // sampleRandom, :672-677
                    const float uc = uniform01(philox4x32_10((uint32_t)s, (uint32_t)gray, (uint32_t)((uint64_t)gray >> 32), 3u, a.seed_lo, a.seed_hi));
                    int pick = (int)(uc * (float)act.n);
                    pick = pick < act.n - 1 ? pick : act.n - 1;
                    inst = __shfl(act.id, pick);
                    weight = (float)act.n;
                } else {
                    float best = INFINITY;
                    for (int q = 0; q < act.n; ++q) {                                           // sampleNearest, :681-692
                        const uint32_t cid = (uint32_t)__builtin_amdgcn_readlane((int)act.id, q);
                        const float *og = a.origins + (size_t)cid * 3;
                        const float ex = px - og[0], ey = py - og[1], ez = pz - og[2];
                        const float dd = __builtin_sqrtf((ex * ex + ey * ey) + ez * ez);
                        if (dd < best) { best = dd; inst = cid; }
                    }
                    if (a.method == 2) {                                                        // sampleNearestBlend, :696-713
                        float tot = 0.0f;
                        for (int q = 0; q < act.n; ++q) {
                            const uint32_t cid = (uint32_t)__builtin_amdgcn_readlane((int)act.id, q);
                            const float *og = a.origins + (size_t)cid * 3;
                            const float ex = px - og[0], ey = py - og[1], ez = pz - og[2];
                            const float w = (a.blend_range + best) - __builtin_sqrtf((ex * ex + ey * ey) + ez * ez);
                            tot = tot + (w > 0.0f ? w : 0.0f);
                        }
                        const float uc = uniform01(philox4x32_10((uint32_t)s, (uint32_t)gray, (uint32_t)((uint64_t)gray >> 32), 3u, a.seed_lo, a.seed_hi));
                        const float target = uc * tot;
                        float acc = 0.0f, wp = 0.0f;
                        bool found = false;
                        for (int q = 0; q < act.n; ++q) {
                            const uint32_t cid = (uint32_t)__builtin_amdgcn_readlane((int)act.id, q);
                            const float *og = a.origins + (size_t)cid * 3;
                            const float ex = px - og[0], ey = py - og[1], ez = pz - og[2];
                            float w = (a.blend_range + best) - __builtin_sqrtf((ex * ex + ey * ey) + ez * ez);
                            w = w > 0.0f ? w : 0.0f;
                            acc = acc + w;
                            if (!found && (target < acc || q == act.n - 1)) { found = true; inst = cid; wp = w; }
                        }
                        weight = tot / wp;                                                      // 1 / probability
                    }
                }
            }
            if (lane < c) {
                const float *mi = a.mats + (size_t)inst * 12;
                const float *di = a.dirs + (size_t)inst * 9;
                float o3[3];
                const size_t k = (size_t)ray * S + s;
                a.t[k] = t_mu;
                a.alpha_weight[k] = weight;
                a.instance_id[k] = (int32_t)inst;
                affine(mi, px, py, pz, o3);                                                     // getPt
                a.pts[3 * k] = o3[0]; a.pts[3 * k + 1] = o3[1]; a.pts[3 * k + 2] = o3[2];
                linear33(di, ndx, ndy, ndz, o3);                                                // getDir
                a.rays_d_map[3 * k] = o3[0]; a.rays_d_map[3 * k + 1] = o3[1]; a.rays_d_map[3 * k + 2] = o3[2];
                float *prow = a.params_map + k * P;
                for (int p = 0; p < P; ++p) prow[p] = s_par[wave][p];
                if (a.light_dir_idx >= 0) {                                                     // getShadowedLightDir(false, ...), :571-581
                    float sx = lx, sy = ly, sz = lz;
                    if (a.light_strength_idx >= 0) { sx = lx - px; sy = ly - py; sz = lz - pz; }
                    normalized(sx, sy, sz);
                    linear33(di, sx, sy, sz, o3);
                    prow[a.light_dir_idx] = o3[0]; prow[a.light_dir_idx + 1] = o3[1]; prow[a.light_dir_idx + 2] = o3[2];
                }
                if (a.light_strength_idx >= 0) {                                                // getLightStrength, :583-588
                    const float ex = lx - px, ey = ly - py, ez = lz - pz;
                    const float d2 = (ex * ex + ey * ey) + ez * ez;
                    prow[a.light_strength_idx] = (float)((double)lstr / (4 * M_PI * (double)d2 + (double)1e-6f));
                }
            }
            step += c;
            if (c < 64) break;
        }
        if (is_mesh) break;                                                                     // :988
        const bool had = act.n == 0;
        if (act.toggle(ids[j], lane, &overflow_active)) {
            if (act.n == 0) cleared = cleared + (tj - t_entry);                                 // :996
        } else if (had) {
            segment_offset = tj - cleared;                                                      // :1001
            t_entry = tj;
        }
    }

    // ---- what instancer.pyx:41-50 leaves in the rows behind the last emitted step -----------------------------------------
    {
        const size_t base = (size_t)ray * S;
        for (int s = step + lane; s < S; s += 64) {
            a.t[base + s] = 0.0f;
            a.alpha_weight[base + s] = 1.0f;
            a.instance_id[base + s] = 0;
        }
        float *pr = a.pts + base * 3, *dr = a.rays_d_map + base * 3;
        const float d3[3] = {dx, dy, dz};
        int rem = (step * 3 + lane) % 3;
        for (int f = step * 3 + lane; f < S * 3; f += 64) {
            pr[f] = 0.0f;
            dr[f] = rem == 0 ? d3[0] : (rem == 1 ? d3[1] : d3[2]);
            rem = (rem + 1) % 3;                                                                // 64 % 3 = 1
        }
        if (P > 0) {
            float *qr = a.params_map + base * P;
            const int inc = 64 % P;
            int pm = (step * P + lane) % P;
            for (int f = step * P + lane; f < S * P; f += 64) {
                qr[f] = s_par[wave][pm];
                pm += inc; pm = pm >= P ? pm - P : pm;
            }
        }
    }
    if (lane == 0) {
        // the closing sample (:1013-1027): the instancer mesh is black and opaque, no mesh = nothing
        a.color_last[3 * ray] = 0.0f; a.color_last[3 * ray + 1] = 0.0f; a.color_last[3 * ray + 2] = 0.0f;
        a.alpha_last[ray] = has_mesh ? 1.0f : 0.0f;
        a.hit[ray] = any_hit ? 1 : 0;
        if (a.status && (overflow_hits || overflow_active)) atomicOr(a.status, (overflow_hits ? 1 : 0) | (overflow_active ? 2 : 0));
    }
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
    float *d_mats = nullptr, *d_dirs = nullptr, *d_org = nullptr, *d_tris = nullptr;
    uint32_t *d_count = nullptr, *d_tmesh = nullptr;
    uint2 *d_hits = nullptr;
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
    for (void *q : {(void *)p->d_mats, (void *)p->d_dirs, (void *)p->d_org, (void *)p->d_tris, (void *)p->d_count, (void *)p->d_tmesh, (void *)p->d_hits})
        if (q) (void)hipFree(q);
    delete p;
}

int reserve(ntx_instancer *p, int64_t max_rays) {
    if (max_rays <= p->cap_rays) return NTX_OK;
    INST_TRY(hipSetDevice(p->device));
    if (p->d_count) { (void)hipFree(p->d_count); p->d_count = nullptr; }
    if (p->d_tmesh) { (void)hipFree(p->d_tmesh); p->d_tmesh = nullptr; }
    if (p->d_hits) { (void)hipFree(p->d_hits); p->d_hits = nullptr; }
    p->cap_rays = 0;
    INST_TRY(hipMalloc((void **)&p->d_count, (size_t)max_rays * sizeof(uint32_t)));
    INST_TRY(hipMalloc((void **)&p->d_tmesh, (size_t)max_rays * sizeof(uint32_t)));
    INST_TRY(hipMalloc((void **)&p->d_hits, (size_t)max_rays * ntx_inst::MAX_HITS * sizeof(uint2)));
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
    if (n_instances < 0 || n_instances > 0x7fffffff || (n_instances > 0 && !transformations)) return ntx_set_error(NTX_E_INVALID, "bad instance list");
    if (desc->n_parameters < 0 || desc->n_parameters > ntx_inst::MAX_PARAMS) return ntx_set_error(NTX_E_INVALID, "n_parameters %d outside [0, %d]", desc->n_parameters, ntx_inst::MAX_PARAMS);
    if (desc->instance_sample_method < 0 || desc->instance_sample_method > 2) return ntx_set_error(NTX_E_INVALID, "instance_sample_method %d is not 0 (random), 1 (nearest) or 2 (nearest_blend)", desc->instance_sample_method);
    const int ld = desc->light_dir_parameter_idx, ls = desc->light_strength_parameter_idx;
    if (ld < -1 || (ld >= 0 && ld + 3 > desc->n_parameters) || ls < -1 || ls >= desc->n_parameters || (ls >= 0 && ld < 0))
        return ntx_set_error(NTX_E_INVALID, "light parameter indices (%d, %d) do not fit %d parameters", ld, ls, desc->n_parameters);
    if (desc->cast_shadow_rays)
        return ntx_set_error(NTX_E_UNSUPPORTED, "cast_shadow_rays (instancer.cpp:591-602) is not built: occlusion queries per shadow sample need a hierarchy over the instances");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ntx_set_error(NTX_E_NODEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return ntx_set_error(NTX_E_INVALID, "device %d out of range [0,%d)", device, ndev);
    ntx_instancer *p = new ntx_instancer();
    p->device = device; p->desc = *desc; p->n_inst = n_instances;
    p->h_mats.resize((size_t)n_instances * 12); p->h_dirs.resize((size_t)n_instances * 9); p->h_org.resize((size_t)n_instances * 3);
    for (int64_t k = 0; k < n_instances; ++k) {                        // AddInstance, instancer.cpp:124-141
        const float *m = transformations + k * 16;
        double inv[16];
        if (!invert4(m, inv)) { delete p; return ntx_set_error(NTX_E_INVALID, "transformation %lld is singular", (long long)k); }
        for (int i = 0; i < 12; ++i) p->h_mats[k * 12 + i] = (float)inv[i];
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
    if (rc == NTX_OK) rc = up(&p->d_dirs, p->h_dirs);
    if (rc == NTX_OK) rc = up(&p->d_org, p->h_org);
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
    if (!inst) return ntx_set_error(NTX_E_INVALID, "inst is NULL");
    if (n_faces < 0 || n_faces > 0x7fffffff || n_vertices < 0 || (n_faces > 0 && (!vertices || !faces))) return ntx_set_error(NTX_E_INVALID, "bad mesh");
    INST_TRY(hipSetDevice(inst->device));
    std::vector<float> tris((size_t)n_faces * 9);
    for (int64_t f = 0; f < n_faces; ++f) {
        for (int c = 0; c < 3; ++c)
            if (faces[3 * f + c] < 0 || faces[3 * f + c] >= n_vertices) return ntx_set_error(NTX_E_INVALID, "face %lld names vertex %d of %lld", (long long)f, faces[3 * f + c], (long long)n_vertices);
        const float *v0 = vertices + 3 * (int64_t)faces[3 * f], *v1 = vertices + 3 * (int64_t)faces[3 * f + 1], *v2 = vertices + 3 * (int64_t)faces[3 * f + 2];
        for (int c = 0; c < 3; ++c) { tris[f * 9 + c] = v0[c]; tris[f * 9 + 3 + c] = v1[c] - v0[c]; tris[f * 9 + 6 + c] = v2[c] - v0[c]; }
    }
    if (inst->d_tris) { (void)hipFree(inst->d_tris); inst->d_tris = nullptr; }
    inst->n_tri = 0;
    if (n_faces > 0) {
        INST_TRY(hipMalloc((void **)&inst->d_tris, tris.size() * sizeof(float)));
        INST_TRY(hipMemcpy(inst->d_tris, tris.data(), tris.size() * sizeof(float), hipMemcpyHostToDevice));
        inst->n_tri = n_faces;
    }
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
    if (opts) {
        if (opts->size < sizeof(ntx_render_opts)) return ntx_set_error(NTX_E_INVALID, "ntx_render_opts.size %u < %zu", opts->size, sizeof(ntx_render_opts));
        if (!(opts->ray_index0 == 0 && opts->ray_run_length == 0 && opts->ray_run_stride == 0)) {
            if (opts->ray_index0 < 0 || opts->ray_run_length < 1 || opts->ray_run_stride < opts->ray_run_length)
                return ntx_set_error(NTX_E_INVALID, "bad ray index map");
            idx0 = opts->ray_index0; idx_stride = opts->ray_run_stride;
            idx_run = opts->ray_run_length > 0xffffffffLL ? 0xffffffffu : (uint32_t)opts->ray_run_length;
        }
    }
    INST_TRY(hipSetDevice(inst->device));
    hipStream_t st = (hipStream_t)stream;
    Box box;
    for (int c = 0; c < 3; ++c) { box.b0[c] = inst->desc.b_0[c]; box.b1[c] = inst->desc.b_1[c]; }
    const int K = (int)inst->n_inst, F = (int)inst->n_tri;
    // the call's rays in pieces of the reserved workspace; a piece's local ray k is ray c0 + k of the call
    for (int64_t c0 = 0; c0 < n_rays; c0 += inst->cap_rays) {
        const int n = (int)(n_rays - c0 < inst->cap_rays ? n_rays - c0 : inst->cap_rays);
        const float *ro = rays_o + c0 * 3, *rd = rays_d + c0 * 3;
        INST_TRY(hipMemsetAsync(inst->d_count, 0, (size_t)n * sizeof(uint32_t), st));
        const int tiles = (n + 63) / 64;
        if (K > 0) {
            // enough waves to fill 256 CUs: a wave takes 64 rays x per_wave instances
            int per_wave = 256;
            while (per_wave > 32 && (int64_t)tiles * ((K + per_wave - 1) / per_wave) < 4096) per_wave >>= 1;
            const int gy = (K + 4 * per_wave - 1) / (4 * per_wave);
            hipLaunchKernelGGL(inst_hits_kernel, dim3(tiles, gy), dim3(256), 0, st, ro, rd, n, inst->d_mats, K, per_wave, box, inst->d_count, inst->d_hits);
        }
        if (F > 0) {
            INST_TRY(hipMemsetD32Async((hipDeviceptr_t)inst->d_tmesh, (int)INF_BITS, (size_t)n, st));
            int per_wave = 256;
            while (per_wave > 32 && (int64_t)tiles * ((F + per_wave - 1) / per_wave) < 4096) per_wave >>= 1;
            const int gy = (F + 4 * per_wave - 1) / (4 * per_wave);
            hipLaunchKernelGGL(inst_mesh_kernel, dim3(tiles, gy), dim3(256), 0, st, ro, rd, n, inst->d_tris, F, per_wave, inst->d_tmesh);
        }
        MarchArgs a{};
        a.rays_o = ro; a.rays_d = rd; a.params = P > 0 ? parameters + c0 * P : nullptr;
        a.mats = inst->d_mats; a.dirs = inst->d_dirs; a.origins = inst->d_org;
        a.count = inst->d_count; a.hits = inst->d_hits; a.t_mesh = F > 0 ? inst->d_tmesh : nullptr;
        const size_t so = (size_t)c0 * n_pts;
        a.rays_d_map = rays_d_map + so * 3; a.pts = pts + so * 3; a.t = t + so; a.dists = dists + so;
        a.color_last = color_last + c0 * 3; a.alpha_last = alpha_last + c0; a.alpha_weight = alpha_weight + so;
        a.params_map = P > 0 ? params_map + so * P : nullptr;
        a.instance_id = instance_id + so; a.hit = hit + c0; a.status = status_flag;
        a.n_rays = n; a.n_pts = n_pts; a.n_params = P;
        a.light_dir_idx = inst->desc.light_dir_parameter_idx; a.light_strength_idx = inst->desc.light_strength_parameter_idx;
        a.method = inst->desc.instance_sample_method; a.use_mean = inst->desc.use_mean_distance ? 1 : 0;
        a.step_size = step_size; a.blend_range = 0.2f * inst->desc.patch_scale;
        a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32);
        // the piece's rays continue the call's index map: local k of the piece = local c0 + k of the call
        if (idx_run == 0xffffffffu) { a.idx0 = idx0 + c0; a.idx_run = 0xffffffffu; a.idx_stride = 0; }
        else if (c0 % idx_run == 0) { a.idx0 = idx0 + (c0 / idx_run) * idx_stride; a.idx_run = idx_run; a.idx_stride = idx_stride; }
        else return ntx_set_error(NTX_E_INVALID, "ray_run_length %u must divide the reserved %lld rays when a call is split (ntx_instancer_reserve)", idx_run, (long long)inst->cap_rays);
        hipLaunchKernelGGL(inst_march_kernel, dim3((n + 3) / 4), dim3(256), 0, st, a);
    }
    INST_TRY(hipGetLastError());
    return NTX_OK;
}

}   // extern "C"
