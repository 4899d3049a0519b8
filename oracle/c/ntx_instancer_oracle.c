/* ntx_instancer_oracle.c -- a SECOND restatement of the reference's patch instancer, C_Instancer::GetModelInput
 * (/root/reference/instancer/src/instancer.cpp:751-1037) with everything it calls, in plain C.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in nerf_tex_amd/ or bench.py's timed region links or calls this file; tests/ load it through
 * oracle/c_instancer.py and require it to agree ELEMENT FOR ELEMENT with the first restatement (oracle/instancer_oracle.py, sequential
 * Python) on random scenes.  It was written from instancer.cpp, not from the Python: the two share no code and no structure -- this
 * one keeps the reference's shape (a HitList that is sorted and de-duplicated, the std::set of active instances as a sorted array, the
 * marching loop with its running state for shadow and texture samples) where the Python follows its own -- so a misreading of the
 * reference's control flow would have to be made twice, in two forms, to go unnoticed.
 *
 * PARITY UNPINNED, like the first: Embree, Eigen, libigl are absent, so neither can be checked against the reference's output.  What
 * both take from contracts instead of code, and therefore share: (1) rtcIntersect1 on instanced boxes reports every face crossing with
 * tnear < t <= tfar at the world ray's parameter: the slab test of the ray taken into patch coordinates; (2) triangles by
 * Moeller-Trumbore without culling, closest hit, Ng = cross(v1 - v0, v2 - v0); (3) the filter is handed ray and normal in patch
 * coordinates; (4) of more than 200 crossings the first 200 of the sorted list are kept; (5) rtcPointQuery reports the closest triangle,
 * the lowest primID among equals; (6) float32 evaluation orders: what Eigen evaluates is paired x0 + (x1 + x2) (its unrolled reduction),
 * what Embree evaluates is summed left to right; (7) std::pow(float, int) computes in double.
 * The reference's std::mt19937 draws are replaced by caller-supplied uniforms (one per ray for the offset, one per (ray, step) for the
 * patch choice), like in the first restatement.
 *
 * Build: oracle/Makefile (gcc -O2 -ffp-contract=off: every operation rounds once, in source order). */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MAX_TOTAL_HITS 200              /* instancer.cpp:22 */
#define INVALID_ID 0xFFFFFFFFu          /* RTC_INVALID_GEOMETRY_ID */
#define INSTANCE_GEOM 0u                /* this->instance_geomID: the quads of the proxy box (rtcAttachGeometry on the empty proxy scene, :119) */
#define MESH_GEOM 1u                    /* any geomID that is not the instances' (the instancer mesh or an auxiliary one) */

typedef struct {
    int n_inst; const float *inv; const float *dir_t; const float *origins;       /* transformations [K][16], dir_transformations [K][9], instance_origins [K][3] */
    float b0[3], b1[3];
    int n_parameters, light_dir_idx, light_strength_idx, sample_method, use_mean_distance;
    float patch_scale;
    int n_mesh_v, n_mesh_f; const float *mesh_v; const int32_t *mesh_f; const float *mesh_n; const uint8_t *mesh_kind; const int32_t *mesh_prim;
    const float *mesh_uv; const int32_t *mesh_tex;
    int cast_shadow_rays, min_shadow_samples, n_shadow_samples;
    int n_tex_files; const int32_t *tex_idx;                                         /* texture_parameter_idxs */
    int n_textures; const float *const *tex_data; const int32_t *tex_rows, *tex_cols;   /* this->textures: every channel of every file */
    int has_instancer; int n_inst_v, n_inst_f; const float *inst_v; const int32_t *inst_f; const float *inst_uv;
    float patch_max_extent; int min_texture_samples, n_texture_samples;
    int n_aux_sets; const int32_t *aux_first, *aux_channels; const float *const *aux_data; const int32_t *aux_rows, *aux_cols;
} io_spec;

typedef struct { float t; uint32_t primID, geomID, instID; float u, v; } hit_t;      /* HitList::Hit, :437-475 */

/* lexicographical order (t, instID, geomID, primID), :444-453 */
static int hit_less(const hit_t *a, const hit_t *b) {
    if (a->t == b->t) {
        if (a->instID == b->instID) {
            if (a->geomID == b->geomID) return a->primID < b->primID;
            return a->geomID < b->geomID;
        }
        return a->instID < b->instID;
    }
    return a->t < b->t;
}
static int hit_equal(const hit_t *a, const hit_t *b) { return a->t == b->t && a->primID == b->primID && a->geomID == b->geomID && a->instID == b->instID; }
static int hit_cmp(const void *a, const void *b) { return hit_less((const hit_t *)a, (const hit_t *)b) ? -1 : (hit_less((const hit_t *)b, (const hit_t *)a) ? 1 : 0); }

/* ---- Eigen's side ------------------------------------------------------------------------------------------------------------- */
static float dot3(const float *a, const float *b) { return a[0] * b[0] + (a[1] * b[1] + a[2] * b[2]); }
static float norm3(const float *a) { return sqrtf(dot3(a, a)); }
static void normalize3(float *a) {                        /* normalized(): only when the squared norm is positive */
    const float n2 = dot3(a, a);
    if (n2 > 0.0f) { const float n = sqrtf(n2); a[0] = a[0] / n; a[1] = a[1] / n; a[2] = a[2] / n; }
}
/* getPt, :556-558: transformations[instID].block<3,3>(0,0) * pt + .block<3,1>(0,3) */
static void get_pt(const io_spec *s, const float *pt, unsigned inst, float *out) {
    const float *m = s->inv + (size_t)inst * 16;
    for (int r = 0; r < 3; ++r) out[r] = (m[4 * r] * pt[0] + (m[4 * r + 1] * pt[1] + m[4 * r + 2] * pt[2])) + m[4 * r + 3];
}
/* getDir, :561-563: dir_transformations[instID] * dir.normalized() */
static void get_dir(const io_spec *s, const float *dir, unsigned inst, float *out) {
    const float *m = s->dir_t + (size_t)inst * 9;
    float d[3] = {dir[0], dir[1], dir[2]};
    normalize3(d);
    for (int r = 0; r < 3; ++r) out[r] = m[3 * r] * d[0] + (m[3 * r + 1] * d[1] + m[3 * r + 2] * d[2]);
}
static void pt_on_ray(const float *o, const float *d, float t, float *out) { for (int c = 0; c < 3; ++c) out[c] = o[c] + t * d[c]; }   /* :566-568 */
static float mean_distance(float mu, float hw) {          /* :746-748; std::pow(float, int) is double */
    const double m = mu, h = hw;
    return (float)(m + 2 * m * (h * h) / (3 * (m * m) + h * h));
}

/* ---- Embree's side (contracts (1)-(3) of the header) --------------------------------------------------------------------------- */
static void to_patch(const io_spec *s, int k, const float *o, const float *d, float *ol, float *dl) {
    const float *m = s->inv + (size_t)k * 16;
    for (int r = 0; r < 3; ++r) {
        ol[r] = ((m[4 * r] * o[0] + m[4 * r + 1] * o[1]) + m[4 * r + 2] * o[2]) + m[4 * r + 3];
        dl[r] = (m[4 * r] * d[0] + m[4 * r + 1] * d[1]) + m[4 * r + 2] * d[2];
    }
}
/* Moeller-Trumbore, no culling; 1 = the ray (tnear 0, tfar 100) crosses triangle f at *t with barycentrics (*u, *v) */
static int tri_hit(const io_spec *s, int f, const float *o, const float *d, float *t, float *u, float *v, int *front) {
    const int32_t *fv = s->mesh_f + 3 * (size_t)f;
    const float *v0 = s->mesh_v + 3 * (size_t)fv[0], *v1 = s->mesh_v + 3 * (size_t)fv[1], *v2 = s->mesh_v + 3 * (size_t)fv[2];
    const float e1[3] = {v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2]}, e2[3] = {v2[0] - v0[0], v2[1] - v0[1], v2[2] - v0[2]};
    const float p[3] = {d[1] * e2[2] - d[2] * e2[1], d[2] * e2[0] - d[0] * e2[2], d[0] * e2[1] - d[1] * e2[0]};
    const float det = (e1[0] * p[0] + e1[1] * p[1]) + e1[2] * p[2];
    if (det == 0.0f) return 0;
    const float inv_det = 1.0f / det;
    const float sv[3] = {o[0] - v0[0], o[1] - v0[1], o[2] - v0[2]};
    const float uu = ((sv[0] * p[0] + sv[1] * p[1]) + sv[2] * p[2]) * inv_det;
    if (uu < 0.0f || uu > 1.0f) return 0;
    const float q[3] = {sv[1] * e1[2] - sv[2] * e1[1], sv[2] * e1[0] - sv[0] * e1[2], sv[0] * e1[1] - sv[1] * e1[0]};
    const float vv = ((d[0] * q[0] + d[1] * q[1]) + d[2] * q[2]) * inv_det;
    if (vv < 0.0f || uu + vv > 1.0f) return 0;
    const float tt = ((e2[0] * q[0] + e2[1] * q[1]) + e2[2] * q[2]) * inv_det;
    if (!(tt > 0.0f && tt <= 100.0f)) return 0;
    const float ng[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    *t = tt; *u = uu; *v = vv;
    if (front) *front = (d[0] * ng[0] + d[1] * ng[1]) + d[2] * ng[2] < 0.0f;
    return 1;
}

/* rtcIntersect1 under primary_ray_filter_function (:526-541, 779): every crossing of an instanced box goes onto the list (the filter
 * turns them down, so the traversal goes on) and the closest crossing of a mesh, which ends the ray.  Returns the number of hits. */
static int intersect_all(const io_spec *s, const float *o, const float *d, hit_t **list, int *cap) {
    int n = 0;
    for (int k = 0; k < s->n_inst; ++k) {
        float ol[3], dl[3];
        to_patch(s, k, o, d, ol, dl);
        float t_in = -INFINITY, t_out = INFINITY;
        int miss = 0;
        for (int a = 0; a < 3; ++a) {
            if (dl[a] == 0.0f) { if (ol[a] < s->b0[a] || ol[a] > s->b1[a]) miss = 1; continue; }
            const float inv = 1.0f / dl[a];
            const float t0 = (s->b0[a] - ol[a]) * inv, t1 = (s->b1[a] - ol[a]) * inv;
            const float lo = t0 < t1 ? t0 : t1, hi = t0 < t1 ? t1 : t0;
            if (lo > t_in) t_in = lo;
            if (hi < t_out) t_out = hi;
        }
        if (miss || !(t_in < t_out)) continue;
        const float ts[2] = {t_in, t_out};
        for (int e = 0; e < 2; ++e) {
            if (!(ts[e] > 0.0f && ts[e] <= 100.0f)) continue;
            if (n == *cap) { *cap = *cap ? *cap * 2 : 256; *list = (hit_t *)realloc(*list, (size_t)*cap * sizeof(hit_t)); }
            (*list)[n++] = (hit_t){ts[e], 0u, INSTANCE_GEOM, (unsigned)k, 0.0f, 0.0f};
        }
    }
    /* contract (4): the first MAX_TOTAL_HITS of the sorted crossings (in the reference: whichever Embree's traversal meets first, :536) */
    qsort(*list, (size_t)n, sizeof(hit_t), hit_cmp);
    if (n > MAX_TOTAL_HITS) n = MAX_TOTAL_HITS;
    int best = -1; float bt = 0, bu = 0, bv = 0;
    for (int f = 0; f < s->n_mesh_f; ++f) {
        float t, u, v;
        if (tri_hit(s, f, o, d, &t, &u, &v, 0) && (best < 0 || t < bt)) { best = f; bt = t; bu = u; bv = v; }
    }
    if (best >= 0) {
        if (n + 1 > *cap) { *cap = n + 64; *list = (hit_t *)realloc(*list, (size_t)*cap * sizeof(hit_t)); }
        (*list)[n++] = (hit_t){bt, (unsigned)best, MESH_GEOM, INVALID_ID, bu, bv};
    }
    return n;
}

/* isShadowed (:591-602) under shadow_ray_filter_function (:543-554): a hit counts when
 *   (primID == 4 && hit_outside) || (geomID != instance_geomID && hit_outside) || primID == 1
 * with hit_outside = dot(ray.dir, Ng) < 0.  createAABB's quads (:109-116): primID 1 = {0,2,6,4} = the face z = b_0, primID 4 = {7,3,1,5}
 * = the face z = b_1; their outward normals are -z and +z, so `outside` on the top face is dir_z < 0.  For a mesh primID is the
 * triangle's index inside ITS mesh. */
static int is_shadowed(const io_spec *s, const float *pt, const float *dir) {
    for (int k = 0; k < s->n_inst; ++k) {
        float ol[3], dl[3];
        to_patch(s, k, pt, dir, ol, dl);
        if (dl[2] == 0.0f) continue;
        const float inv = 1.0f / dl[2];
        for (int top = 1; top >= 0; --top) {
            const float z = top ? s->b1[2] : s->b0[2];
            const float tt = (z - ol[2]) * inv;
            if (!(tt > 0.0f && tt <= 100.0f)) continue;
            const float x = ol[0] + tt * dl[0], y = ol[1] + tt * dl[1];
            if (!(s->b0[0] <= x && x <= s->b1[0] && s->b0[1] <= y && y <= s->b1[1])) continue;
            const int hit_outside = top ? dl[2] < 0.0f : dl[2] > 0.0f;
            const unsigned primID = top ? 4u : 1u;
            if ((primID == 4u && hit_outside) || primID == 1u) return 1;
        }
    }
    for (int f = 0; f < s->n_mesh_f; ++f) {
        float t, u, v; int front;
        if (!tri_hit(s, f, pt, dir, &t, &u, &v, &front)) continue;
        const unsigned primID = s->mesh_prim ? (unsigned)s->mesh_prim[f] : (unsigned)f;
        if (front || primID == 1u) return 1;            /* (geomID != instance_geomID && hit_outside) || primID == 1 */
    }
    return 0;
}

/* ---- textures ------------------------------------------------------------------------------------------------------------------- */
/* interpolate2d, :605-625; y_ref(r, c) = data[r * cols + c].  Indices outside the matrix are clamped (see the first restatement). */
static float interpolate2d(float u, float v, const float *data, int rows, int cols) {
    const float x0 = u * ((float)rows - 1.0f), x1 = v * ((float)cols - 1.0f);
    const int i = (int)x0, j = (int)x1;
    const float w0 = x0 - floorf(x0), w1 = x1 - floorf(x1);
#define CL(a, n) ((a) < 0 ? 0 : ((a) > (n) - 1 ? (n) - 1 : (a)))
    const float y00 = data[(size_t)CL(i, rows) * cols + CL(j, cols)], y01 = data[(size_t)CL(i, rows) * cols + CL(j + 1, cols)];
    const float y10 = data[(size_t)CL(i + 1, rows) * cols + CL(j, cols)], y11 = data[(size_t)CL(i + 1, rows) * cols + CL(j + 1, cols)];
#undef CL
    return ((y00 * (1.0f - w0) * (1.0f - w1) + y01 * (1.0f - w0) * w1) + y10 * w0 * (1.0f - w1)) + y11 * w0 * w1;
}
/* closest_point_triangle, :154-198: returns the distance |q - p| and the barycentrics */
static float closest_point_triangle(const float *p, const float *a, const float *b, const float *c, float *uvw) {
    float ab[3], ac[3], ap[3], bp[3], cp[3], q[3];
    for (int i = 0; i < 3; ++i) { ab[i] = b[i] - a[i]; ac[i] = c[i] - a[i]; ap[i] = p[i] - a[i]; }
    const float d1 = dot3(ab, ap), d2 = dot3(ac, ap);
    if (d1 <= 0.f && d2 <= 0.f) { memcpy(q, a, sizeof q); uvw[0] = 1; uvw[1] = 0; uvw[2] = 0; goto done; }
    for (int i = 0; i < 3; ++i) bp[i] = p[i] - b[i];
    const float d3 = dot3(ab, bp), d4 = dot3(ac, bp);
    if (d3 >= 0.f && d4 <= d3) { memcpy(q, b, sizeof q); uvw[0] = 0; uvw[1] = 1; uvw[2] = 0; goto done; }
    for (int i = 0; i < 3; ++i) cp[i] = p[i] - c[i];
    const float d5 = dot3(ab, cp), d6 = dot3(ac, cp);
    if (d6 >= 0.f && d5 <= d6) { memcpy(q, c, sizeof q); uvw[0] = 0; uvw[1] = 0; uvw[2] = 1; goto done; }
    const float vc = d1 * d4 - d3 * d2;
    if (vc <= 0.f && d1 >= 0.f && d3 <= 0.f) {
        const float v = d1 / (d1 - d3);
        for (int i = 0; i < 3; ++i) q[i] = a[i] + v * ab[i];
        uvw[0] = 1 - v; uvw[1] = v; uvw[2] = 0; goto done;
    }
    const float vb = d5 * d2 - d1 * d6;
    if (vb <= 0.f && d2 >= 0.f && d6 <= 0.f) {
        const float v = d2 / (d2 - d6);
        for (int i = 0; i < 3; ++i) q[i] = a[i] + v * ac[i];
        uvw[0] = 1 - v; uvw[1] = 0; uvw[2] = v; goto done;
    }
    const float va = d3 * d6 - d5 * d4;
    if (va <= 0.f && (d4 - d3) >= 0.f && (d5 - d6) >= 0.f) {
        const float v = (d4 - d3) / ((d4 - d3) + (d5 - d6));
        for (int i = 0; i < 3; ++i) q[i] = b[i] + v * (c[i] - b[i]);
        uvw[0] = 0; uvw[1] = 1 - v; uvw[2] = v; goto done;
    }
    {
        const float denom = 1.f / (va + vb + vc);
        const float v = vb * denom, w = vc * denom;
        for (int i = 0; i < 3; ++i) q[i] = a[i] + v * ab[i] + w * ac[i];
        uvw[0] = 1 - v - w; uvw[1] = v; uvw[2] = w;
    }
done:;
    const float e[3] = {p[0] - q[0], p[1] - q[1], p[2] - q[2]};
    return norm3(e);
}
/* getParameters, :640-667: parameter_map = parameters; every texture FILE i: parameter_map(idx[i]) *= interpolate2d(uv, textures[i]) */
static void get_parameters(const io_spec *s, const float *pt, const float *parameters, float *out) {
    memcpy(out, parameters, (size_t)s->n_parameters * sizeof(float));
    float radius = s->patch_max_extent;                 /* query.radius, shrunk by the callback (:222-227) */
    int prim = -1; float uvw[3] = {0, 0, 0};
    for (int f = 0; f < s->n_inst_f; ++f) {
        const int32_t *fv = s->inst_f + 3 * (size_t)f;
        float w[3];
        const float d = closest_point_triangle(pt, s->inst_v + 3 * (size_t)fv[0], s->inst_v + 3 * (size_t)fv[1], s->inst_v + 3 * (size_t)fv[2], w);
        if (d < radius) { radius = d; prim = f; memcpy(uvw, w, sizeof uvw); }
    }
    if (prim < 0) return;
    const int32_t *fv = s->inst_f + 3 * (size_t)prim;
    float uv[2];
    for (int c = 0; c < 2; ++c) uv[c] = (s->inst_uv[2 * fv[0] + c] * uvw[0] + s->inst_uv[2 * fv[1] + c] * uvw[1]) + s->inst_uv[2 * fv[2] + c] * uvw[2];
    for (int i = 0; i < s->n_tex_files; ++i) out[s->tex_idx[i]] *= interpolate2d(uv[0], uv[1], s->tex_data[i], s->tex_rows[i], s->tex_cols[i]);
}

/* shadeMesh, :716-743 */
static void shade_mesh(const io_spec *s, const float *pt, unsigned prim, const float *uvw, const float *dir, float *rgb) {
    const int32_t *f = s->mesh_f + 3 * (size_t)prim;
    float n[3];
    for (int c = 0; c < 3; ++c) n[c] = (s->mesh_n[3 * f[0] + c] * uvw[0] + s->mesh_n[3 * f[1] + c] * uvw[1]) + s->mesh_n[3 * f[2] + c] * uvw[2];
    normalize3(n);
    float albedo[3] = {.8f, .8f, .8f};
    const int set = s->mesh_tex ? s->mesh_tex[prim] : -1;
    if (set >= 0) {
        float uv[2];
        for (int c = 0; c < 2; ++c) uv[c] = (s->mesh_uv[2 * f[0] + c] * uvw[0] + s->mesh_uv[2 * f[1] + c] * uvw[1]) + s->mesh_uv[2 * f[2] + c] * uvw[2];
        const int first = s->aux_first[set], nch = s->aux_channels[set];
        float val[4];
        for (int c = 0; c < nch && c < 4; ++c) val[c] = interpolate2d(uv[0], uv[1], s->aux_data[first + c], s->aux_rows[first + c], s->aux_cols[first + c]);
        if (nch == 3) { albedo[0] = val[0]; albedo[1] = val[1]; albedo[2] = val[2]; }
        else albedo[0] = albedo[1] = albedo[2] = val[0];                       /* Constant(albedo(0)), :732 */
    }
    float diffuse = 1.0f;
    const float above[3] = {pt[0] + n[0] * 1e-6f, pt[1] + n[1] * 1e-6f, pt[2] + n[2] * 1e-6f};
    if (!is_shadowed(s, above, dir)) {
        float l[3] = {dir[0], dir[1], dir[2]};
        normalize3(l);
        const float nd = dot3(n, l);
        diffuse *= nd > 0.f ? nd : 0.f;
    } else diffuse = 0;
    const float sum = diffuse + 0.2f, shade = sum < 1.f ? sum : 1.f;
    for (int c = 0; c < 3; ++c) rgb[c] = albedo[c] * shade;
}

/* ---- the std::set<unsigned> of active instances: a sorted array ------------------------------------------------------------------ */
typedef struct { unsigned *id; int n, cap; } set_t;
static int set_find(const set_t *a, unsigned x) { for (int i = 0; i < a->n; ++i) if (a->id[i] == x) return i; return -1; }
static void set_erase(set_t *a, unsigned x) { const int i = set_find(a, x); if (i >= 0) { memmove(a->id + i, a->id + i + 1, (size_t)(a->n - i - 1) * sizeof(unsigned)); --a->n; } }
static void set_insert(set_t *a, unsigned x) {
    if (set_find(a, x) >= 0) return;
    if (a->n == a->cap) { a->cap = a->cap ? 2 * a->cap : 64; a->id = (unsigned *)realloc(a->id, (size_t)a->cap * sizeof(unsigned)); }
    int i = a->n;
    while (i > 0 && a->id[i - 1] > x) { a->id[i] = a->id[i - 1]; --i; }
    a->id[i] = x; ++a->n;
}

/* GetModelInput, :751-1037.  The buffers arrive as instancer.pyx:41-50 fills them (rays_d repeated, t / dists / pts / color / density
 * zero, density_weight one, instance_id zero, hit false, parameters repeated); u_offset[n_rays], u_choice[n_rays * n_pts] stand in for
 * the std::mt19937 draws.  Returns 0, or 1 when a ray walked off the end of its segment list (the reference reads past it there). */
int io_get_model_input(const io_spec *s, int n_rays, int n_pts, float step_size, const float *rays_o, float *rays_d, float *t, float *dists,
                       float *pts, float *color, float *density, float *density_weight, int32_t *instance_id, uint8_t *hit, float *parameters,
                       const float *u_offset, const float *u_choice) {
    const int P = s->n_parameters;
    hit_t *hits = NULL; int cap = 0, rc = 0;
    set_t active = {NULL, 0, 0};
    float *segment_lengths = NULL; int seg_cap = 0;
    float *default_parameters = (float *)malloc((size_t)(P > 0 ? P : 1) * sizeof(float));
    float *sample_0_texture = (float *)malloc((size_t)(P > 0 ? P : 1) * sizeof(float)), *sample_1_texture = (float *)malloc((size_t)(P > 0 ? P : 1) * sizeof(float));
    float *weights = NULL; int w_cap = 0;
    for (int i = 0; i < n_rays; ++i) {
        const float *org = rays_o + 3 * (size_t)i;
        float default_raydir[3];
        memcpy(default_raydir, rays_d + 3 * (size_t)n_pts * i, sizeof default_raydir);
        const int n_hits = intersect_all(s, org, default_raydir, &hits, &cap);
        if (!n_hits) continue;                                                  /* :782 */
        hit[i] = 1;
        /* :787-798: sorted, out of order duplicates dropped (intersect_all sorted the boxes; the mesh hit joins them here) */
        qsort(hits, (size_t)n_hits, sizeof(hit_t), hit_cmp);
        int end = 0;
        {
            int a = 0;
            for (int j = 1; j < n_hits; ++j) { if (hit_equal(&hits[a], &hits[j])) continue; hits[++a] = hits[j]; }
            end = a + 1;
        }
        /* :800-826: the length of the ray inside the union of the boxes, segment by segment */
        int n_seg = 0;
        float total_segment_length = 0, t_entry = 0;
        const hit_t *mesh_hit = NULL;
        active.n = 0;
        for (int j = 0; j < end; ++j) {
            if (hits[j].geomID != INSTANCE_GEOM) {
                if (active.n) {
                    const float len = hits[j].t - t_entry;
                    total_segment_length += len;
                    if (n_seg == seg_cap) { seg_cap = seg_cap ? 2 * seg_cap : 64; segment_lengths = (float *)realloc(segment_lengths, (size_t)seg_cap * sizeof(float)); }
                    segment_lengths[n_seg++] = len;
                }
                mesh_hit = &hits[j];
                break;
            }
            const unsigned inst = hits[j].instID;
            if (set_find(&active, inst) >= 0) {
                set_erase(&active, inst);
                if (!active.n) {
                    const float len = hits[j].t - t_entry;
                    total_segment_length += len;
                    if (n_seg == seg_cap) { seg_cap = seg_cap ? 2 * seg_cap : 64; segment_lengths = (float *)realloc(segment_lengths, (size_t)seg_cap * sizeof(float)); }
                    segment_lengths[n_seg++] = len;
                }
            } else {
                if (!active.n) t_entry = hits[j].t;
                set_insert(&active, inst);
            }
        }
        active.n = 0;                                                           /* :829 */
        float default_lightdir[3] = {0, 0, 0}, default_lightstr = 0;
        if (s->light_dir_idx >= 0) memcpy(default_lightdir, parameters + (size_t)n_pts * i * P + s->light_dir_idx, sizeof default_lightdir);
        if (s->light_strength_idx >= 0) default_lightstr = parameters[(size_t)n_pts * i * P + s->light_strength_idx];

        if (total_segment_length > 0) {
            const uint32_t neccessary_steps = (uint32_t)(total_segment_length / step_size);
            uint32_t n_steps = neccessary_steps < (uint32_t)n_pts ? neccessary_steps : (uint32_t)n_pts;
            float t_offset = 0;
            if (n_steps == 0) {
                dists[(size_t)n_pts * i] = total_segment_length;
                t_offset = u_offset[i] * total_segment_length;
                n_steps = 1;
            } else {
                for (uint32_t j = 0; j + 1 < n_steps; ++j) dists[(size_t)n_pts * i + j] = step_size;
                dists[(size_t)n_pts * i + n_steps - 1] = step_size + total_segment_length - n_steps * step_size;       /* :856 */
                t_offset = u_offset[i] * step_size;
            }
            const uint32_t n_shadow_samples = (uint32_t)s->min_shadow_samples > (uint32_t)(s->n_shadow_samples * total_segment_length)
                                                  ? (uint32_t)s->min_shadow_samples : (uint32_t)(s->n_shadow_samples * total_segment_length);
            const uint32_t n_texture_samples = (uint32_t)s->min_texture_samples > (uint32_t)(s->n_texture_samples * total_segment_length)
                                                   ? (uint32_t)s->min_texture_samples : (uint32_t)(s->n_texture_samples * total_segment_length);
            const int textured = s->has_instancer && s->n_tex_files > 0;        /* instancer_geomID valid && !texture_parameter_idxs.empty() */
            float segment_offset = 0, cleared_segment_length = 0; t_entry = 0;
            uint32_t k_shadow = 0, k_texture = 0;
            float t_0_shadow = 0, t_1_shadow = 0, t_0_texture = 0, t_1_texture = 0, step_length_shadow = 0, step_length_texture = 0;
            int sample_0_shadow = 0, sample_1_shadow = 0;
            if (P > 0) memcpy(default_parameters, parameters + (size_t)n_pts * i * P, (size_t)P * sizeof(float));
            uint32_t step = 0; int l = 0;
            for (int j = 0; j < end && step < n_steps; ++j) {
                float t_mu = step * step_size + t_offset + segment_offset, t_pt;
                t_pt = s->use_mean_distance ? mean_distance(t_mu, step_size) : t_mu;
                while (active.n && (t_pt < hits[j].t) && step < n_steps) {
                    const size_t k = (size_t)n_pts * i + step;
                    t[k] = t_mu;
                    float pt[3];
                    pt_on_ray(org, default_raydir, t_pt, pt);
                    unsigned inst;
                    if (active.n == 1) {
                        inst = active.id[0]; density_weight[k] = 1.0f; instance_id[k] = (int)inst;
                    } else {
                        const int m = active.n;
                        if (s->sample_method == 0) {                              /* sampleRandom, :671-676 (a caller-supplied uniform for the draw) */
                            int pick = (int)(u_choice[k] * (float)m);
                            if (pick > m - 1) pick = m - 1;
                            inst = active.id[pick]; density_weight[k] = (float)m;
                        } else if (s->sample_method == 1) {                       /* sampleNearest, :680-691 */
                            float min_dist = INFINITY; int at = 0;
                            for (int q = 0; q < m; ++q) {
                                const float *og = s->origins + 3 * (size_t)active.id[q];
                                const float e[3] = {pt[0] - og[0], pt[1] - og[1], pt[2] - og[2]};
                                const float dist = norm3(e);
                                if (dist < min_dist) { at = q; min_dist = dist; }
                            }
                            inst = active.id[at]; density_weight[k] = 1.f;
                        } else {                                                  /* sampleNearestBlend, :695-713 */
                            const float transition_range = 0.2f * s->patch_scale;
                            if (m > w_cap) { w_cap = 2 * m; weights = (float *)realloc(weights, (size_t)w_cap * sizeof(float)); }
                            float min_dist = INFINITY;
                            for (int q = 0; q < m; ++q) {
                                const float *og = s->origins + 3 * (size_t)active.id[q];
                                const float e[3] = {pt[0] - og[0], pt[1] - og[1], pt[2] - og[2]};
                                weights[q] = norm3(e);
                                if (weights[q] < min_dist) min_dist = weights[q];
                            }
                            float tot = 0;
                            for (int q = 0; q < m; ++q) { const float w = transition_range + min_dist - weights[q]; weights[q] = w > 0.f ? w : 0.f; tot = tot + weights[q]; }
                            /* std::discrete_distribution with a caller-supplied uniform: the first index whose running sum exceeds u * total */
                            const float target = u_choice[k] * tot;
                            float acc = 0; int pick = m - 1;
                            for (int q = 0; q < m; ++q) { acc = acc + weights[q]; if (target < acc) { pick = q; break; } }
                            inst = active.id[pick]; density_weight[k] = tot / weights[pick];         /* 1 / probability */
                        }
                        instance_id[k] = (int)inst;
                    }
                    float *row = parameters + k * P;
                    if (textured && n_texture_samples < (uint32_t)n_pts) {          /* :911-923 */
                        while (t_pt > t_1_texture) {
                            t_0_texture = t_1_texture;
                            t_1_texture = t_entry + ++k_texture * step_length_texture;
                            memcpy(sample_0_texture, sample_1_texture, (size_t)P * sizeof(float));
                            float q[3];
                            pt_on_ray(org, default_raydir, t_1_texture, q);
                            get_parameters(s, q, default_parameters, sample_1_texture);
                        }
                        const float w = (t_pt - t_0_texture) / step_length_texture;
                        for (int c = 0; c < P; ++c) row[c] = sample_0_texture[c] * (1 - w) + sample_1_texture[c] * w;
                    } else if (textured) {
                        get_parameters(s, pt, default_parameters, row);              /* :926 */
                    }
                    if (s->light_dir_idx >= 0) {                                    /* :930-951 */
                        int shadowed = 0;
                        if (s->cast_shadow_rays && n_shadow_samples < (uint32_t)n_pts) {
                            while (t_pt > t_1_shadow) {
                                t_0_shadow = t_1_shadow;
                                t_1_shadow = t_entry + ++k_shadow * step_length_shadow;
                                sample_0_shadow = sample_1_shadow;
                                float q[3];
                                pt_on_ray(org, default_raydir, t_1_shadow, q);
                                sample_1_shadow = is_shadowed(s, q, default_lightdir);
                            }
                            const int w = (t_pt - t_0_shadow) / step_length_shadow >= 0.5f;
                            shadowed = (!w && sample_0_shadow) || (w && sample_1_shadow);
                        } else if (s->cast_shadow_rays) {
                            shadowed = is_shadowed(s, pt, default_lightdir);
                        }
                        float *ld = row + s->light_dir_idx;                          /* getShadowedLightDir, :571-582 */
                        if (shadowed) { ld[0] = 0; ld[1] = 0; ld[2] = -1; }
                        else if (s->light_strength_idx >= 0) { const float dd[3] = {default_lightdir[0] - pt[0], default_lightdir[1] - pt[1], default_lightdir[2] - pt[2]}; get_dir(s, dd, inst, ld); }
                        else get_dir(s, default_lightdir, inst, ld);
                    }
                    if (s->light_strength_idx >= 0) {                               /* getLightStrength, :584-590: float eps, double arithmetic through M_PI */
                        const float e[3] = {default_lightdir[0] - pt[0], default_lightdir[1] - pt[1], default_lightdir[2] - pt[2]};
                        const float dist_squared = dot3(e, e);
                        const float eps = 1e-6f;
                        row[s->light_strength_idx] = (float)(default_lightstr / (4 * M_PI * dist_squared + eps));
                    }
                    get_pt(s, pt, inst, pts + 3 * k);                                /* :959-960 */
                    get_dir(s, default_raydir, inst, rays_d + 3 * k);
                    ++step;
                    t_mu = step * step_size + t_offset + segment_offset;
                    t_pt = s->use_mean_distance ? mean_distance(t_mu, step_size) : t_mu;
                }
                if (hits[j].geomID != INSTANCE_GEOM) break;                          /* :972 */
                const unsigned inst = hits[j].instID;
                if (set_find(&active, inst) >= 0) {
                    set_erase(&active, inst);
                    if (!active.n) cleared_segment_length += hits[j].t - t_entry;
                } else {
                    if (!active.n) {
                        segment_offset = hits[j].t - cleared_segment_length;
                        t_entry = hits[j].t;
                        const int wants_tex = textured && n_texture_samples < (uint32_t)n_pts;
                        const int wants_sh = s->light_dir_idx >= 0 && s->cast_shadow_rays && n_shadow_samples < (uint32_t)n_pts;
                        if ((wants_tex || wants_sh) && l >= n_seg) { rc = 1; goto next_ray; }   /* the reference reads segment_lengths[l] past its end here */
                        if (wants_tex) {                                             /* :989-998 */
                            const float segment_length = segment_lengths[l];
                            const uint32_t per = (uint32_t)(n_texture_samples * segment_length / total_segment_length);
                            const uint32_t n_seg_samples = (uint32_t)s->min_texture_samples > per ? (uint32_t)s->min_texture_samples : per;
                            step_length_texture = segment_length / (n_seg_samples - 1);
                            k_texture = 1;
                            t_0_texture = t_entry;
                            t_1_texture = t_entry + step_length_texture;
                            float q[3];
                            pt_on_ray(org, default_raydir, t_0_texture, q); get_parameters(s, q, default_parameters, sample_0_texture);
                            pt_on_ray(org, default_raydir, t_1_texture, q); get_parameters(s, q, default_parameters, sample_1_texture);
                        }
                        if (wants_sh) {                                              /* :1000-1009 */
                            const float segment_length = segment_lengths[l];
                            const uint32_t per = (uint32_t)(n_shadow_samples * segment_length / total_segment_length);
                            const uint32_t n_seg_samples = (uint32_t)s->min_shadow_samples > per ? (uint32_t)s->min_shadow_samples : per;
                            step_length_shadow = segment_length / (n_seg_samples - 1);
                            k_shadow = 1;
                            t_0_shadow = t_entry;
                            t_1_shadow = t_entry + step_length_shadow;
                            float q[3];
                            pt_on_ray(org, default_raydir, t_0_shadow, q); sample_0_shadow = is_shadowed(s, q, default_lightdir);
                            pt_on_ray(org, default_raydir, t_1_shadow, q); sample_1_shadow = is_shadowed(s, q, default_lightdir);
                        }
                        ++l;
                    }
                    set_insert(&active, inst);
                }
            }
        }
    next_ray:
        /* :1018-1029: black behind the instancer mesh, shaded behind an auxiliary one, nothing else */
        if (mesh_hit) {
            float *c = color + 3 * (size_t)i;
            if (!s->mesh_kind || s->mesh_kind[mesh_hit->primID] == 0) { c[0] = 0; c[1] = 0; c[2] = 0; }
            else {
                float pt[3];
                pt_on_ray(org, default_raydir, mesh_hit->t, pt);
                const float uvw[3] = {1 - mesh_hit->u - mesh_hit->v, mesh_hit->u, mesh_hit->v};
                shade_mesh(s, pt, mesh_hit->primID, uvw, default_lightdir, c);
            }
            density[i] = 1;
        } else {
            color[3 * (size_t)i] = 0; color[3 * (size_t)i + 1] = 0; color[3 * (size_t)i + 2] = 0;
            density[i] = 0;
        }
        active.n = 0;
    }
    free(hits); free(active.id); free(segment_lengths); free(default_parameters); free(sample_0_texture); free(sample_1_texture); free(weights);
    return rc;
}

/* Eigen's inverse of a 4x4 float matrix, as compute_inverse_size4 evaluates it without SSE (Inverse_SSE / the generic cofactor path give
 * results that differ in the last place; this is the textbook cofactor expansion in float32): what `transform_mat.inverse()` of
 * AddInstance (:130) is checked against -- the product computes the inverse in double and rounds. */
static float det3(const float *m, int r0, int r1, int r2, int c0, int c1, int c2) {
#define M(r, c) m[4 * (r) + (c)]
    return M(r0, c0) * (M(r1, c1) * M(r2, c2) - M(r1, c2) * M(r2, c1)) - M(r0, c1) * (M(r1, c0) * M(r2, c2) - M(r1, c2) * M(r2, c0)) + M(r0, c2) * (M(r1, c0) * M(r2, c1) - M(r1, c1) * M(r2, c0));
#undef M
}
int io_inverse4_float(const float *m, float *out) {
    float cof[16];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            const int rr[3] = {(r + 1) % 4, (r + 2) % 4, (r + 3) % 4}, cc[3] = {(c + 1) % 4, (c + 2) % 4, (c + 3) % 4};
            int a[3] = {rr[0], rr[1], rr[2]}, b[3] = {cc[0], cc[1], cc[2]};
            for (int i = 0; i < 2; ++i) for (int j = 0; j < 2 - i; ++j) { if (a[j] > a[j + 1]) { int x = a[j]; a[j] = a[j + 1]; a[j + 1] = x; } if (b[j] > b[j + 1]) { int x = b[j]; b[j] = b[j + 1]; b[j + 1] = x; } }
            const float minor = det3(m, a[0], a[1], a[2], b[0], b[1], b[2]);
            cof[4 * r + c] = ((r + c) & 1) ? -minor : minor;
        }
    const float det = m[0] * cof[0] + m[1] * cof[1] + m[2] * cof[2] + m[3] * cof[3];
    if (det == 0.0f) return 1;
    const float inv = 1.0f / det;
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) out[4 * r + c] = cof[4 * c + r] * inv;      /* adjugate = transposed cofactors */
    return 0;
}
