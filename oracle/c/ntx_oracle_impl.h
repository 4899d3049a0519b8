/* ntx_oracle_impl.h -- body of the C restatement, included twice (REAL = float / double).
 * TEST INFRASTRUCTURE ONLY (see oracle/nerftex_oracle.py for the rules and the "parity unpinned"
 * statement).  Scalar, single-threaded, k-ordered sums: deliberately the dumbest possible reading of
 * the reference source, written independently of the numpy restatement so the two can be compared.
 * Every function cites the file:line of /root/reference it follows. */

/* layer.FourierFeatures.call (layer.py:8-23): out = [x | sin(2^0 x) | cos(2^0 x) | ...] */
static void FN(fourier)(const REAL *x, int d, int nf, REAL *out) {
    int p = 0;
    for (int c = 0; c < d; ++c) out[p++] = x[c];
    REAL freq = 1;
    for (int k = 0; k < nf; ++k) {
        for (int c = 0; c < d; ++c) out[p++] = SIN(freq * x[c]);
        for (int c = 0; c < d; ++c) out[p++] = COS(freq * x[c]);
        freq *= 2;
    }
}

/* Dense (Keras): y = act(x . kernel[in,out] + bias) */
static const float *FN(dense)(const float *w, const REAL *x, int in, int out, int relu, REAL *y) {
    const float *b = w + (size_t)in * out;
    for (int o = 0; o < out; ++o) {
        REAL s = 0;
        for (int i = 0; i < in; ++i) s += x[i] * (REAL)w[(size_t)i * out + o];
        s += (REAL)b[o];
        y[o] = (relu && s < 0) ? 0 : s;
    }
    return b + out;
}

/* network.model.ParamNerf (model.py:58-125) / Nerf (model.py:9-45) on ONE sample.
 * desc: {kind, n_geo, n_app, pos_freq, dir_freq, param_freq, depth, width, skip, color_depth} */
static void FN(model_one)(const int *desc, const float *w, const REAL *pos, const REAL *dir, const REAL *par,
                          REAL *color, REAL *alpha) {
    const int kind = desc[0], g = kind ? 0 : desc[1], a = kind ? 0 : desc[2];
    const int pf = desc[3], df = desc[4], qf = desc[5], depth = desc[6], width = desc[7], skip = desc[8];
    const int cd = kind ? 0 : desc[9];
    REAL pos_map[512], dir_map[512], h[1024], y[1024];
    int pm = 3 * (1 + 2 * pf), dm = 3 * (1 + 2 * df);
    FN(fourier)(pos, 3, pf, pos_map);                                  /* model.py:77 */
    FN(fourier)(dir, 3, df, dir_map);                                  /* model.py:78 */
    if (g > 0) { FN(fourier)(par, g, qf, pos_map + pm); pm += g * (1 + 2 * qf); }       /* :88-93 */
    if (a > 0) { FN(fourier)(par + g, a, qf, dir_map + dm); dm += a * (1 + 2 * qf); }   /* :96-101 */
    int k = pm;
    for (int i = 0; i < k; ++i) h[i] = pos_map[i];
    for (int i = 0; i < depth; ++i) {                                  /* :104-108 */
        w = FN(dense)(w, h, k, width, 1, y);
        if (i == skip) {
            for (int j = 0; j < pm; ++j) h[j] = pos_map[j];
            for (int j = 0; j < width; ++j) h[pm + j] = y[j];
            k = pm + width;
        } else {
            for (int j = 0; j < width; ++j) h[j] = y[j];
            k = width;
        }
    }
    {   /* :111 -- the alpha head is the LAST layer of the get_weights() blob (Keras orders layers by graph depth) */
        const int c1_in = dm + width;
        size_t off = (size_t)k * width + width;                                     /* feature */
        for (int i = 0; i < cd; ++i) off += (size_t)(i ? width : c1_in) * width + width;   /* colour layers */
        off += (size_t)(cd ? width : c1_in) * (width / 2) + width / 2;              /* colour half */
        off += (size_t)(width / 2) * 3 + 3;                                         /* color */
        FN(dense)(w + off, h, k, 1, 0, alpha);
    }
    w = FN(dense)(w, h, k, width, 0, y);                               /* :114 */
    for (int j = 0; j < dm; ++j) h[j] = dir_map[j];                    /* :115 */
    for (int j = 0; j < width; ++j) h[dm + j] = y[j];
    k = dm + width;
    for (int i = 0; i < cd; ++i) {                                     /* :118-119 */
        w = FN(dense)(w, h, k, width, 1, y);
        for (int j = 0; j < width; ++j) h[j] = y[j];
        k = width;
    }
    w = FN(dense)(w, h, k, width / 2, 1, y);                           /* :122 */
    FN(dense)(w, y, width / 2, 3, 0, color);                           /* :123 */
}

/* Renderer.map_model_output (renderer.py:170-213) on ONE ray */
static void FN(composite_one)(const REAL *color, const REAL *sigma, const REAL *z, const REAL *rays_d, int S,
                              int map_exr, int composite_bkgd, const REAL *bkgd, REAL *color_out, REAL *alpha_out,
                              REAL *weights) {
    const REAL nrm = SQRT(rays_d[0] * rays_d[0] + rays_d[1] * rays_d[1] + rays_d[2] * rays_d[2]);
    REAL T = 1, acc[3] = {0, 0, 0}, A = 0;
    for (int i = 0; i < S; ++i) {
        REAL dist = (i < S - 1) ? z[i + 1] - z[i] : z[S - 1] - z[S - 2];    /* :174-177 */
        dist = dist * nrm;                                                   /* :180 */
        const REAL sg = sigma[i] > 0 ? sigma[i] : 0;
        const REAL a = 1 - EXP(-sg * dist);                                  /* :195 */
        const REAL wgt = a * T;                                              /* :198 exclusive cumprod */
        T = T * ((1 - a) + (REAL)1e-10);
        for (int c = 0; c < 3; ++c) {
            const REAL raw = color[3 * i + c];
            const REAL m = map_exr ? ((raw > 0 ? raw : EXP(raw) - 1) + 1) : 1 / (1 + EXP(-raw));   /* :182-187 */
            acc[c] += wgt * m;                                               /* :201 */
        }
        A += wgt;                                                            /* :207 */
        if (weights) weights[i] = wgt;
    }
    for (int c = 0; c < 3; ++c) color_out[c] = composite_bkgd ? acc[c] + (1 - A) * bkgd[c] : acc[c];   /* :210-211 */
    *alpha_out = A;
}

/* Renderer.render_rays + evaluate_model (renderer.py:92-168), perturb=False, one ray at a time */
void FN(ntxo_render_rays)(const int *desc, const float *w, const float *rays_o, const float *rays_d, const float *t,
                          const float *params, const float *cone, long n_rays, int S, int blur_idx, int map_exr,
                          int composite_bkgd, const double *bkgd, REAL *color_out, REAL *alpha_out) {
    const int np = desc[0] ? 0 : desc[1] + desc[2];
    REAL *col = (REAL *)malloc(sizeof(REAL) * 3 * S), *sg = (REAL *)malloc(sizeof(REAL) * S),
         *z = (REAL *)malloc(sizeof(REAL) * S);
    const REAL delta = (REAL)1 / (REAL)(S - 1);                              /* tf.linspace step */
    REAL bk[3] = {(REAL)bkgd[0], (REAL)bkgd[1], (REAL)bkgd[2]};
    for (long r = 0; r < n_rays; ++r) {
        const REAL o[3] = {rays_o[3 * r], rays_o[3 * r + 1], rays_o[3 * r + 2]};
        const REAL d[3] = {rays_d[3 * r], rays_d[3 * r + 1], rays_d[3 * r + 2]};
        const REAL nrm = SQRT(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        const REAL dn[3] = {d[0] / nrm, d[1] / nrm, d[2] / nrm};            /* :98 */
        for (int i = 0; i < S; ++i) {
            const REAL tv = i == 0 ? 0 : (i == S - 1 ? 1 : delta * (REAL)i);
            z[i] = (REAL)t[2 * r] * (1 - tv) + (REAL)t[2 * r + 1] * tv;      /* :102 */
            const REAL p[3] = {o[0] + d[0] * z[i], o[1] + d[1] * z[i], o[2] + d[2] * z[i]};   /* :114 */
            REAL par[16];
            for (int k = 0; k < np; ++k) {
                par[k] = params[np * r + k];
                if (k == blur_idx) par[k] = par[k] * ((REAL)cone[r] * z[i]);  /* :155-158 */
            }
            FN(model_one)(desc, w, p, dn, par, col + 3 * i, sg + i);
        }
        FN(composite_one)(col, sg, z, d, S, map_exr, composite_bkgd, bk, color_out + 3 * r, alpha_out + r, 0);
    }
    free(col); free(sg); free(z);
}

void FN(ntxo_model)(const int *desc, const float *w, const float *pos, const float *dir, const float *params, long m,
                    REAL *color, REAL *alpha) {
    const int np = desc[0] ? 0 : desc[1] + desc[2];
    for (long i = 0; i < m; ++i) {
        const REAL p[3] = {pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]};
        const REAL d[3] = {dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]};
        REAL par[16];
        for (int k = 0; k < np; ++k) par[k] = params[np * i + k];
        FN(model_one)(desc, w, p, d, par, color + 3 * i, alpha + i);
    }
}

void FN(ntxo_composite)(const float *color, const float *sigma, const float *z, const float *rays_d, long n, int S,
                        int map_exr, int composite_bkgd, const double *bkgd, REAL *color_out, REAL *alpha_out,
                        REAL *weights) {
    REAL *c = (REAL *)malloc(sizeof(REAL) * 3 * S), *s = (REAL *)malloc(sizeof(REAL) * S),
         *zz = (REAL *)malloc(sizeof(REAL) * S);
    REAL bk[3] = {(REAL)bkgd[0], (REAL)bkgd[1], (REAL)bkgd[2]};
    for (long r = 0; r < n; ++r) {
        for (int i = 0; i < S; ++i) {
            s[i] = sigma[r * S + i]; zz[i] = z[r * S + i];
            for (int k = 0; k < 3; ++k) c[3 * i + k] = color[(r * S + i) * 3 + k];
        }
        const REAL d[3] = {rays_d[3 * r], rays_d[3 * r + 1], rays_d[3 * r + 2]};
        FN(composite_one)(c, s, zz, d, S, map_exr, composite_bkgd, bk, color_out + 3 * r, alpha_out + r,
                          weights ? weights + r * S : 0);
    }
    free(c); free(s); free(zz);
}
