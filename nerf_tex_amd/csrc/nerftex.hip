// nerftex.hip -- host side of libnerftex_hip.so: weight packing, kernel dispatch and the C ABI
// declared in include/nerftex.h.  gfx950 only.
#include "nerftex.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>

#include "ntx_device_x3.h"
#include "ntx_small_kernels.h"

using namespace ntx;

// ---------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// the same for the other translation units of the library (ntx_comm.hip); not part of the public ABI
extern "C" __attribute__((visibility("hidden"))) int ntx_set_error(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return fail(NTX_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// ---------------------------------------------------------------------------------------------
// supported architectures = the kernels instantiated below
// ---------------------------------------------------------------------------------------------
struct Variant {
    int n_geo, n_app, cd, ipe;   // the kernel family's layout (for the generic family: its parameter SLOTS)
    int gen;
    int flex;                    // the architecture is read from the model descriptor (ntx_layout.h "flex family")
};
static const Variant kVariants[] = {
    {1, 6, 1, 0, 0},   // carpet          (configs/config_carpet_render.py:59-72)
    {1, 4, 1, 0, 0},   // grass, fur, plush
    {2, 3, 1, 0, 0},   // grass_filtered
    {0, 0, 0, 0, 0},   // plain Nerf      (model.py:9-45)
    {1, 3, 1, 1, 0},   // mip variant of grass_filtered: IPE on (mean, cov), blur parameter spliced out (renderer.py:385-386)
    {GEN_NGEO, GEN_NAPP, 1, 0, 1},   // generic: any other ParamNerf n_parameters = [g <= 4, a <= 8]; absent parameters = zero rows
    {GEN_NGEO, GEN_NAPP, 1, 0, 1, 1},   // flex: any depth <= 24, width <= 256, skips, color_depth <= 4 (model.py:58), float32 kernels only
    {GEN_NGEO, GEN_NAPP, 1, 0, 1, 2},   // flex with param_depth 1..4: Dense(param_width <= 128) layers on the parameter features (model.py:88-101)
};
constexpr int kFlexVariant = 6, kFlexParamVariant = 7;
static_assert(NTX_SKIP_MASK == (unsigned)NTX_SKIP_MASK_BIT, "skip encoding of the ABI header and of ntx_layout.h");

// the model's own parameter counts (the generic family has more slots than the model has parameters)
struct Dims {
    int g, a;
    int pf, df, qf;   // n_freq_bands of the model's position / direction / parameter embeddings (layer.py:11)
};
static Dims dims_of(const ntx_model_desc *d) {
    const bool nerf = d->kind == NTX_MODEL_NERF;
    return Dims{nerf ? 0 : d->n_geo, nerf ? 0 : d->n_app, d->pos_freq, d->dir_freq, nerf || d->n_geo + d->n_app <= 0 ? PAR_FREQ : d->param_freq};
}
// FEWER frequency bands than the kernels' 10 / 4 / 4 (FourierFeatures(n_freq_bands), layer.py:8-23): the kernels evaluate all of
// theirs, the packers give the bands the model does not have zero weight rows -- exact, like the parameters the generic family does
// not have.  Widths of the model's own encodings, and the model's row for row r of a 10/4/4 model's map (-1: no such band).
static int pos_emb_m(Dims m, int ipe) { return ipe ? 6 * m.pf : 3 * (1 + 2 * m.pf); }
static int dir_emb_m(Dims m) { return 3 * (1 + 2 * m.df); }
static int pos_map_m(Dims m, int ipe) { return pos_emb_m(m, ipe) + m.g * (1 + 2 * m.qf); }
static int dir_map_m(Dims m) { return dir_emb_m(m) + m.a * (1 + 2 * m.qf); }
static int par_row_m(int idx, int n_act, int qf) {        // idx into [p (n_act) | sin f0, cos f0 (n_act each) | ...] of 4 bands
    if (idx < n_act) return idx;
    return (idx - n_act) / (2 * n_act) < qf ? idx : -1;
}
static int pos_row_m(int r, Dims m, int ipe) {
    if (r < 0) return r;
    const int full = pos_emb_dim(ipe);
    if (r >= full) { const int q = par_row_m(r - full, m.g, m.qf); return q < 0 ? -1 : pos_emb_m(m, ipe) + q; }
    if (ipe) { const int h = r / (3 * POS_FREQ), q = r % (3 * POS_FREQ); return q / 3 < m.pf ? h * 3 * m.pf + q : -1; }
    return r < 3 || (r - 3) / 6 < m.pf ? r : -1;
}
static int dir_row_m(int r, Dims m) {
    if (r < 0) return r;
    const int full = 3 * (1 + 2 * DIR_FREQ);
    if (r >= full) { const int q = par_row_m(r - full, m.a, m.qf); return q < 0 ? -1 : dir_emb_m(m) + q; }
    return r < 3 || (r - 3) / 6 < m.df ? r : -1;
}

// the model's `skips` as a mask of layer indices: ntx_model_desc.skip is one index (-1: none) or NTX_SKIP_MASK | mask
static unsigned skip_mask_of(const ntx_model_desc *d) {
    if (d->skip < 0) return 0u;
    if (d->skip & NTX_SKIP_MASK) return (unsigned)d->skip & (NTX_SKIP_MASK - 1u);
    return d->skip < 30 ? 1u << d->skip : 0u;
}
// param_depth / param_width of the model: fields of the extended descriptor (kind NTX_MODEL_PARAMNERF_EX); a model without
// parameters has no branches whatever param_depth says (model.py:88, 96)
static int param_depth_of(const ntx_model_desc *d) {
    if (d->kind != NTX_MODEL_PARAMNERF_EX || d->n_geo + d->n_app <= 0) return 0;
    return reinterpret_cast<const ntx_model_desc_ex *>(d)->param_depth;
}
static int param_width_of(const ntx_model_desc *d) {
    return d->kind == NTX_MODEL_PARAMNERF_EX ? reinterpret_cast<const ntx_model_desc_ex *>(d)->param_width : 0;
}
static FlexArch flex_arch_of(const ntx_model_desc *d) {
    // a skip index >= depth - 1 .. : `i in skips` never fires for i >= depth (model.py:107); i = depth - 1 is refused in find_variant
    const int pd = param_depth_of(d);
    return FlexArch{d->depth, d->width, d->kind == NTX_MODEL_NERF ? 0 : d->color_depth, skip_mask_of(d) & ((1u << (d->depth > 1 ? d->depth - 1 : 0)) - 1u),
                    pd, pd > 0 ? param_width_of(d) : 0, pd > 0 && d->n_geo > 0, pd > 0 && d->n_app > 0};
}
static bool default_arch(const ntx_model_desc *d) {
    const bool nerf = d->kind == NTX_MODEL_NERF;
    return d->depth == DEPTH && d->width == WIDTH && d->skip == SKIP && (nerf || d->color_depth == 1) && param_depth_of(d) == 0;
}

static int find_variant(const ntx_model_desc *d) {
    if (!d) return -1;
    const int ipe = d->pos_encoding == NTX_POS_IPE;
    if (d->pos_encoding != NTX_POS_FOURIER && d->pos_encoding != NTX_POS_IPE) return -1;
    if (d->n_pos != (ipe ? 6 : 3) || d->pos_freq < 0 || d->pos_freq > POS_FREQ || d->dir_freq < 0 || d->dir_freq > DIR_FREQ) return -1;
    if (d->kind != NTX_MODEL_PARAMNERF && d->kind != NTX_MODEL_NERF && d->kind != NTX_MODEL_PARAMNERF_EX) return -1;
    const bool nerf = d->kind == NTX_MODEL_NERF;
    const int g = nerf ? 0 : d->n_geo, a = nerf ? 0 : d->n_app, cd = nerf ? 0 : d->color_depth;
    if (g < 0 || a < 0) return -1;
    if (!nerf && (g + a > 0) && (d->param_freq < 0 || d->param_freq > PAR_FREQ)) return -1;
    const bool force_flex = getenv("NERFTEX_FORCE_FLEX") != nullptr;         // A/B knobs for tests: a tuned family's model on the
    const bool force_generic = getenv("NERFTEX_FORCE_GENERIC") != nullptr;   // flex / generic kernels
    if (default_arch(d) && !(force_flex && !ipe)) {
        for (size_t i = 0; i < sizeof(kVariants) / sizeof(kVariants[0]) && !(force_generic && !nerf && !ipe); ++i)
            if (!kVariants[i].gen && kVariants[i].n_geo == g && kVariants[i].n_app == a && kVariants[i].cd == cd && kVariants[i].ipe == ipe) return (int)i;
        for (size_t i = 0; i < sizeof(kVariants) / sizeof(kVariants[0]); ++i)
            if (kVariants[i].gen && !kVariants[i].flex && !nerf && g <= kVariants[i].n_geo && a <= kVariants[i].n_app && kVariants[i].cd == cd && kVariants[i].ipe == ipe) return (int)i;
        return -1;
    }
    // any other architecture: the layer loop of the flex family
    if (ipe || g > GEN_NGEO || a > GEN_NAPP) return -1;
    if (d->depth < 1 || d->depth > FLEX_MAX_DEPTH || d->width < 2 || d->width > WIDTH || cd < 0 || cd > FLEX_MAX_COLOR) return -1;
    if (d->skip >= 0 && !(d->skip & NTX_SKIP_MASK) && d->skip >= 30) return -1;
    // a skip behind the LAST trunk layer widens the inputs of the alpha head and of the feature layer (model.py:107-114): not built
    if ((skip_mask_of(d) >> (d->depth - 1)) & 1u) return -1;
    if (d->kind == NTX_MODEL_PARAMNERF_EX && reinterpret_cast<const ntx_model_desc_ex *>(d)->param_depth < 0) return -1;
    if (const int pd = param_depth_of(d)) {
        if (pd > FLEX_MAX_PARAM_DEPTH || param_width_of(d) < 2 || param_width_of(d) > 2 * BRANCH_K) return -1;
        return kFlexParamVariant;
    }
    return kFlexVariant;
}

static int unsupported(const ntx_model_desc *d) {
    if (!d) return fail(NTX_E_INVALID, "model descriptor is NULL");
    return fail(NTX_E_UNSUPPORTED,
                "unsupported model: kind=%d n_parameters=[%d,%d] n_pos=%d freqs=%d/%d/%d depth=%d width=%d "
                "skip=%d color_depth=%d pos_encoding=%d param_depth=%d param_width=%d (built: ParamNerf with n_parameters [g<=4, a<=8] -- tuned kernels for [1,6] [1,4] "
                "[2,3] at 8x256 / skips [4] / color_depth 1 --, Nerf, and ParamNerf [1,3] with IntegratedPositionalEncoding on 6-D positions; "
                "other architectures (FourierFeatures only): depth 1..24, width 2..256, color_depth 0..4, skips below depth-1, "
                "param_depth 0..4 with param_width 2..128; n_freq_bands <= 10 / 4 / 4)",
                d->kind, d->n_geo, d->n_app, d->n_pos, d->pos_freq, d->dir_freq, d->param_freq, d->depth,
                d->width, d->skip, d->color_depth, d->pos_encoding,
                d->kind == NTX_MODEL_PARAMNERF_EX ? reinterpret_cast<const ntx_model_desc_ex *>(d)->param_depth : 0, param_width_of(d));
}

// ---------------------------------------------------------------------------------------------
// reference-layout blob (Keras get_weights() order) -> layer views (model.py:104-125)
// ---------------------------------------------------------------------------------------------
struct Layer {
    const float *w, *b;
    int in, out;
};

struct Net {
    Layer trunk[DEPTH], alpha, feature, c1, c2, rgb;
    bool has_c1;
    size_t count;
};

static Net view_blob(const Variant &v, Dims m, const float *blob) {
    Net n{};
    const int pm = pos_map_m(m, v.ipe), dm = dir_map_m(m);
    size_t p = 0;
    auto take = [&](int in, int out) {
        Layer l{blob ? blob + p : nullptr, blob ? blob + p + (size_t)in * out : nullptr, in, out};
        p += (size_t)in * out + out;
        return l;
    };
    int k = pm;
    for (int i = 0; i < DEPTH; ++i) {
        n.trunk[i] = take(k, WIDTH);
        k = WIDTH + (i == SKIP ? pm : 0);
    }
    // tf.keras.Model orders its layers by graph depth, ties by traversal from outputs=[color, alpha] (model.py:125), so
    // get_weights() has the alpha head LAST although it is created before the feature layer (model.py:111-123)
    n.feature = take(WIDTH, WIDTH);
    n.has_c1 = v.cd > 0;
    if (n.has_c1) {
        n.c1 = take(WIDTH + dm, WIDTH);
        n.c2 = take(WIDTH, WIDTH / 2);
    } else {
        n.c2 = take(WIDTH + dm, WIDTH / 2);
    }
    n.rgb = take(WIDTH / 2, 3);
    n.alpha = take(WIDTH, 1);
    n.count = p;
    return n;
}

// one segment of the stream: for every k-step, NMT/4 records of [lane][4 consecutive M-tiles]
template <class RowFn>
static void emit_segment(float *&dst, const Layer &l, int nsteps, int nmt, int row_offset, RowFn rowfn) {
    for (int s = 0; s < nsteps; ++s)
        for (int q = 0; q < nmt / 4; ++q)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 4; ++e) {
                    const int row = rowfn(s, lane >> 5);
                    const int col = 32 * (4 * q + e) + (lane & 31);
                    float val = 0.0f;
                    if (row >= 0 && col < l.out) val = l.w[(size_t)(row_offset + row) * l.out + col];
                    *dst++ = val;
                }
}

static void pack(const Variant &v, Dims m, const float *blob, float *out) {
    const Geometry g = make_geometry(v.n_geo, v.n_app, v.cd, v.ipe);
    const Net n = view_blob(v, m, blob);
    const int pm = pos_map_m(m, v.ipe), dm = dir_map_m(m);
    float *dst = out;
    auto posrow = [&](int s, int h) { return pos_row_m(pos_row(v.n_geo, s, h, v.ipe, m.g), m, v.ipe); };
    auto dirrow = [&](int s, int h) { return dir_row_m(dir_row(v.n_app, s, h, m.a), m); };
    auto hidrow = [&](int s, int h) { return hidden_row(s, h); };

    emit_segment(dst, n.trunk[0], g.pos_steps, 8, 0, posrow);
    for (int i = 1; i < DEPTH; ++i) {
        if (i == SKIP + 1) {
            emit_segment(dst, n.trunk[i], g.pos_steps, 8, 0, posrow);
            emit_segment(dst, n.trunk[i], HSTEPS, 8, pm, hidrow);
        } else {
            emit_segment(dst, n.trunk[i], HSTEPS, 8, 0, hidrow);
        }
    }
    emit_segment(dst, n.feature, HSTEPS, 8, 0, hidrow);
    if (n.has_c1) {
        emit_segment(dst, n.c1, g.dir_steps, 8, 0, dirrow);
        emit_segment(dst, n.c1, HSTEPS, 8, dm, hidrow);
        emit_segment(dst, n.c2, HSTEPS, 4, 0, hidrow);
    } else {
        emit_segment(dst, n.c2, g.dir_steps, 4, 0, dirrow);
        emit_segment(dst, n.c2, HSTEPS, 4, dm, hidrow);
    }
    // zero pad up to a whole number of ring turns, then the wrap-around tail: the first RING records again
    for (int i = g.stream_records; i < g.padded_records; ++i) { memset(dst, 0, sizeof(float) * REC_FLOATS); dst += REC_FLOATS; }
    memcpy(dst, out, sizeof(float) * RING * REC_FLOATS);
    dst += RING * REC_FLOATS;

    // aux block
    float *aux = dst;
    memset(aux, 0, sizeof(float) * g.aux_floats);
    auto put_bias = [&](int layer, const Layer &l) {
        for (int h = 0; h < 2; ++h)
            for (int s = 0; s < l.out / 2; ++s) aux[layer * AUX_BIAS_STRIDE + h * 128 + s] = l.b[hidden_row(s, h)];
    };
    for (int i = 0; i < DEPTH; ++i) put_bias(i, n.trunk[i]);
    put_bias(8, n.feature);
    if (n.has_c1) put_bias(9, n.c1);
    put_bias(10, n.c2);
    for (int h = 0; h < 2; ++h)
        for (int s = 0; s < HSTEPS; ++s) aux[aux_alpha_off() + h * 128 + s] = n.alpha.w[hidden_row(s, h)];
    aux[aux_alpha_off() + 256] = n.alpha.b[0];
    for (int c = 0; c < 3; ++c) {
        for (int h = 0; h < 2; ++h)
            for (int s = 0; s < 64; ++s) aux[aux_rgb_off() + (c * 2 + h) * 64 + s] = n.rgb.w[hidden_row(s, h) * 3 + c];
        aux[aux_rgb_off() + 384 + c] = n.rgb.b[c];
    }
}

// ---- flex family (ntx_layout.h): any depth / width <= 256 / skips / color_depth ---------------------------------------------------
struct FlexNet {
    std::vector<Layer> trunk, colour;   // colour: the color_depth hidden colour layers
    std::vector<Layer> pgeo, papp;      // param_depth > 0: the Dense layers of the geometry / appearance branch
    Layer alpha, feature, c2, rgb;
    int pos_map, dir_map;               // widths of pos_map / dir_map as the trunk / the first colour layer see them
    size_t count;
};
// get_weights() order of the functional model for ANY architecture (layer_table of nerf_tex_amd/model.py, checked against a
// restatement of Keras' rule in tests/test_oracle.py): every Dense layer in the order a depth-first traversal from outputs = [color,
// alpha] first meets it, with its graph depth (concat nodes take a level), then by decreasing depth, ties in traversal order.
// Without branches: trunk, feature, colour layers, colour half, color, alpha.  With param_depth > 0 the geometry branch comes
// before the trunk and the appearance branch interleaves with the trunk layers of equal depth, ahead of them.
static FlexNet view_blob_flex(const FlexArch &f, Dims m, const float *blob) {
    FlexNet n{};
    const int pd = f.param_depth, pw = f.param_width, w = f.width;
    const int ffdim_g = m.g * (1 + 2 * m.qf), ffdim_a = m.a * (1 + 2 * m.qf);
    n.pos_map = pos_emb_m(m, 0) + (m.g > 0 ? (pd > 0 ? pw : ffdim_g) : 0);
    n.dir_map = dir_emb_m(m) + (m.a > 0 ? (pd > 0 ? pw : ffdim_a) : 0);
    struct Slot { Layer *l; int in, out, depth; };
    std::vector<Slot> seq;
    n.trunk.resize(f.depth); n.colour.resize(f.color_depth);
    n.pgeo.resize(f.has_geo ? pd : 0); n.papp.resize(f.has_app ? pd : 0);
    const int cd = f.color_depth;
    seq.push_back({&n.rgb, w / 2, 3, 0});
    seq.push_back({&n.c2, cd > 0 ? w : w + n.dir_map, w / 2, 1});
    for (int i = cd - 1; i >= 0; --i) seq.push_back({&n.colour[i], i == 0 ? w + n.dir_map : w, w, 1 + cd - i});
    int d = cd + 2;
    for (int i = (int)n.papp.size() - 1; i >= 0; --i) seq.push_back({&n.papp[i], i == 0 ? ffdim_a : pw, pw, d + 2 + (pd - 1 - i)});
    d += 1;
    seq.push_back({&n.feature, w, w, d});                                    // (a skip behind the last trunk layer is refused)
    for (int i = f.depth - 1; i >= 0; --i) {
        d += 1 + (((f.skip_mask >> i) & 1u) ? 1 : 0);
        const int in = i == 0 ? n.pos_map : w + (((f.skip_mask >> (i - 1)) & 1u) ? n.pos_map : 0);
        seq.push_back({&n.trunk[i], in, w, d});
    }
    for (int i = (int)n.pgeo.size() - 1; i >= 0; --i) seq.push_back({&n.pgeo[i], i == 0 ? ffdim_g : pw, pw, d + 2 + (pd - 1 - i)});
    seq.push_back({&n.alpha, w, 1, 0});
    std::vector<int> order(seq.size());
    for (size_t j = 0; j < order.size(); ++j) order[j] = (int)j;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return seq[x].depth > seq[y].depth; });
    size_t p = 0;
    for (int j : order) {
        const Slot &sl = seq[j];
        *sl.l = Layer{blob ? blob + p : nullptr, blob ? blob + p + (size_t)sl.in * sl.out : nullptr, sl.in, sl.out};
        p += (size_t)sl.in * sl.out + sl.out;
    }
    n.count = p;
    return n;
}

// emit_segment with the rows of a narrower layer (width < 256: hidden rows >= `rows` are zero) and zero records up to PADREC
template <class RowFn>
static void emit_segment_flex(float *&dst, const Layer &l, int nsteps, int nmt, int row_offset, int rows, RowFn rowfn) {
    float *const start = dst;
    emit_segment(dst, l, nsteps, nmt, row_offset, [&](int s, int h) { const int r = rowfn(s, h); return r < rows ? r : -1; });
    const int pad = flex_seg_records(nsteps, nmt) - nsteps * (nmt / 4);
    memset(dst, 0, sizeof(float) * REC_FLOATS * pad);
    dst += (size_t)REC_FLOATS * pad;
    (void)start;
}

static size_t packed_floats_flex(const FlexArch &f) {
    return (size_t)(flex_stream_records(f) + RING) * REC_FLOATS + aux_total() + flex_floats();
}

static void pack_flex(const FlexArch &f, Dims m, const float *blob, float *out) {
    const FlexNet n = view_blob_flex(f, m, blob);
    const bool pb = f.param_depth > 0;
    const int pe = pos_emb_m(m, 0), de = dir_emb_m(m);                       // FF(pos), FF(dir) as the model has them
    const int pm = n.pos_map, dm = n.dir_map;
    // without branches: the position / direction segments of the generic family (parameter features in them); with: FF(pos) /
    // FF(dir) alone, each followed by 64 k-steps over its branch's output
    const int ps = pb ? pos_steps(0) : pos_steps(GEN_NGEO), ds = pb ? dir_steps(0) : dir_steps(GEN_NAPP);
    float *dst = out;
    const Dims m0{0, 0, m.pf, m.df, m.qf};                                      // with branches the segments hold FF(pos) / FF(dir) alone
    auto posrow = [&](int s, int h) { return pb ? pos_row_m(pos_row(0, s, h), m0, 0) : pos_row_m(pos_row(GEN_NGEO, s, h, 0, m.g), m, 0); };
    auto dirrow = [&](int s, int h) { return pb ? dir_row_m(dir_row(0, s, h), m0) : dir_row_m(dir_row(GEN_NAPP, s, h, m.a), m); };
    auto hidrow = [&](int s, int h) { return hidden_row(s, h); };
    const int W = f.width, PW = f.param_width;
    auto branch = [&](const std::vector<Layer> &ls, int n_slots, int n_act) {   // a branch's own layers, 4 tiles
        if (ls.empty()) return;
        emit_segment_flex(dst, ls[0], parff_steps(n_slots), 4, 0, ls[0].in, [&](int s, int h) { const int r = parff_row(n_slots, s, h, n_act); return r < 0 ? r : par_row_m(r, n_act, m.qf); });
        for (size_t i = 1; i < ls.size(); ++i) emit_segment_flex(dst, ls[i], BRANCH_K, 4, 0, PW, hidrow);
    };
    auto pos_input = [&](const Layer &l) {                                      // concat[FF(pos) (+ parameter features) | G]
        emit_segment_flex(dst, l, ps, 8, 0, pb ? pe : pm, posrow);
        if (pb && f.has_geo) emit_segment_flex(dst, l, BRANCH_K, 8, pe, PW, hidrow);
    };
    auto dir_input = [&](const Layer &l, int nmt) {                            // concat[FF(dir) (+ parameter features) | A]
        emit_segment_flex(dst, l, ds, nmt, 0, pb ? de : dm, dirrow);
        if (pb && f.has_app) emit_segment_flex(dst, l, BRANCH_K, nmt, de, PW, hidrow);
    };
    branch(n.pgeo, GEN_NGEO, m.g);
    pos_input(n.trunk[0]);
    for (int i = 1; i < f.depth; ++i) {
        const bool skip_in = (f.skip_mask >> (i - 1)) & 1u;
        if (skip_in) pos_input(n.trunk[i]);
        emit_segment_flex(dst, n.trunk[i], HSTEPS, 8, skip_in ? pm : 0, W, hidrow);
    }
    emit_segment_flex(dst, n.feature, HSTEPS, 8, 0, W, hidrow);
    branch(n.papp, GEN_NAPP, m.a);
    if (f.color_depth > 0) {
        dir_input(n.colour[0], 8);
        emit_segment_flex(dst, n.colour[0], HSTEPS, 8, dm, W, hidrow);
        for (int i = 1; i < f.color_depth; ++i) emit_segment_flex(dst, n.colour[i], HSTEPS, 8, 0, W, hidrow);
        emit_segment_flex(dst, n.c2, HSTEPS, 4, 0, W, hidrow);
    } else {
        dir_input(n.c2, 4);
        emit_segment_flex(dst, n.c2, HSTEPS, 4, dm, W, hidrow);
    }
    memcpy(dst, out, sizeof(float) * RING * REC_FLOATS);   // wrap-around tail
    dst += RING * REC_FLOATS;

    // aux block: the tuned layout (only its alpha / rgb heads are used), then [descriptor | bias slots]
    float *aux = dst;
    memset(aux, 0, sizeof(float) * (aux_total() + flex_floats()));
    for (int h = 0; h < 2; ++h)
        for (int s = 0; s < HSTEPS; ++s) {
            const int r = hidden_row(s, h);
            aux[aux_alpha_off() + h * 128 + s] = r < W ? n.alpha.w[r] : 0.0f;
        }
    aux[aux_alpha_off() + 256] = n.alpha.b[0];
    for (int c = 0; c < 3; ++c) {
        for (int h = 0; h < 2; ++h)
            for (int s = 0; s < 64; ++s) {
                const int r = hidden_row(s, h);
                aux[aux_rgb_off() + (c * 2 + h) * 64 + s] = r < W / 2 ? n.rgb.w[r * 3 + c] : 0.0f;
            }
        aux[aux_rgb_off() + 384 + c] = n.rgb.b[c];
    }
    int32_t *desc = reinterpret_cast<int32_t *>(aux + aux_total());
    desc[0] = f.depth; desc[1] = (int32_t)f.skip_mask; desc[2] = f.color_depth;
    desc[3] = f.param_depth; desc[4] = f.has_geo; desc[5] = f.has_app;
    float *bias = aux + aux_total() + FLEX_DESC_FLOATS;
    auto put_bias = [&](int slot, const Layer &l) {
        for (int h = 0; h < 2; ++h)
            for (int s = 0; s < HSTEPS; ++s) {
                const int r = hidden_row(s, h);
                bias[slot * AUX_BIAS_STRIDE + h * 128 + s] = r < l.out ? l.b[r] : 0.0f;
            }
    };
    int slot = 0;
    for (int i = 0; i < f.depth; ++i) put_bias(slot++, n.trunk[i]);
    put_bias(slot++, n.feature);
    for (int i = 0; i < f.color_depth; ++i) put_bias(slot++, n.colour[i]);
    put_bias(slot++, n.c2);
    // branch layers: geometry at n8 + 1 .., appearance at n8 + 1 + FLEX_MAX_PARAM_DEPTH .. (mlp_flex)
    for (size_t i = 0; i < n.pgeo.size(); ++i) put_bias(slot + (int)i, n.pgeo[i]);
    for (size_t i = 0; i < n.papp.size(); ++i) put_bias(slot + FLEX_MAX_PARAM_DEPTH + (int)i, n.papp[i]);
    static_assert(FLEX_MAX_DEPTH + 1 + FLEX_MAX_COLOR + 1 + 2 * FLEX_MAX_PARAM_DEPTH <= FLEX_MAX_LAYERS, "bias slots");
}

// ---- fp16x3 stream (ntx_layout.h: one record = the A operand of one (k16-step, M-tile), hi record then lo record) ----
// float32 -> IEEE half, round to nearest even, subnormals kept, overflow to inf (what v_cvt_f16_f32 does)
static uint16_t f16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    const uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
    u &= 0x7fffffffu;
    if (u >= 0x47800000u) return sign | (u > 0x7f800000u ? 0x7e00 : 0x7c00);   // >= 65536: inf, or NaN
    if (u < 0x38800000u) {                                                       // < 2^-14: subnormal half or zero
        // adding 0.5f aligns the value so that float addition rounds it (RNE) to a multiple of 2^-24
        float t;
        memcpy(&t, &u, 4);
        t += 0.5f;
        uint32_t r;
        memcpy(&r, &t, 4);
        return sign | (uint16_t)(r - 0x3f000000u);
    }
    const uint32_t odd = (u >> 13) & 1u;
    u += 0xc8000fffu + odd;        // rebias the exponent by -112 and round the 13 dropped bits to nearest even
    return sign | (uint16_t)(u >> 13);                                           // 65520..65535.99 carries into inf
}
static float f16_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, em = h & 0x7fffu;
    uint32_t u;
    if (em >= 0x7c00u) u = sign | 0x7f800000u | ((em & 0x3ffu) << 13);
    else if (em >= 0x0400u) u = sign | ((em << 13) + 0x38000000u);
    else {                                                                       // subnormal: em * 2^-24
        const float t = (float)em * 5.9604644775390625e-08f;
        memcpy(&u, &t, 4);
        u |= sign;
    }
    float f;
    memcpy(&f, &u, 4);
    return f;
}

template <class RowFn>
static void emit_segment16(uint16_t *&dst, const Layer &l, int nsteps16, int nmt, int row_offset, RowFn rowfn) {
    for (int u = 0; u < nsteps16; ++u)
        for (int mt = 0; mt < nmt; ++mt) {
            uint16_t *hi = dst, *lo = dst + 512;
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int row = rowfn(8 * u + e, lane >> 5);
                    const int col = 32 * mt + (lane & 31);
                    float val = 0.0f;
                    if (row >= 0 && col < l.out) val = l.w[(size_t)(row_offset + row) * l.out + col];
                    const uint16_t h = f16_rne(val);
                    hi[lane * 8 + e] = h;
                    lo[lane * 8 + e] = f16_rne(val - f16_f32(h));
                }
            dst += 1024;
        }
}

static size_t packed16_bytes(const Variant &v, int with_dir = 0) {
    return (size_t)stream16_padded(v.n_geo, v.n_app, v.cd, with_dir, v.ipe) * 1024;
}

// hidden segment first, encoder segment second within a pass (ntx_device_x3.h: Cfg16)
// with_dir: the instanced kernel's stream, where C1 keeps its direction segment (directions are per sample there)
static void pack16(const Variant &v, Dims m, const float *blob, uint16_t *out, int with_dir = 0) {
    const Net n = view_blob(v, m, blob);
    const int pm = pos_map_m(m, v.ipe), dm = dir_map_m(m);
    const int ps = steps16(pos_steps(v.n_geo, v.ipe)), ds = steps16(dir_steps(v.n_app)), hs = HSTEPS / 8;
    uint16_t *dst = out;
    auto posrow = [&](int s, int h) { return s < pos_steps(v.n_geo, v.ipe) ? pos_row_m(pos_row(v.n_geo, s, h, v.ipe, m.g), m, v.ipe) : -1; };
    auto dirrow = [&](int s, int h) { return s < dir_steps(v.n_app) ? dir_row_m(dir_row(v.n_app, s, h, m.a), m) : -1; };
    auto hidrow = [&](int s, int h) { return hidden_row(s, h); };
    emit_segment16(dst, n.trunk[0], ps, 8, 0, posrow);
    for (int i = 1; i < DEPTH; ++i) {
        if (i == SKIP + 1) {
            emit_segment16(dst, n.trunk[i], hs, 8, pm, hidrow);
            emit_segment16(dst, n.trunk[i], ps, 8, 0, posrow);
        } else {
            emit_segment16(dst, n.trunk[i], hs, 8, 0, hidrow);
        }
    }
    emit_segment16(dst, n.feature, hs, 8, 0, hidrow);
    if (n.has_c1) {
        emit_segment16(dst, n.c1, hs, 8, dm, hidrow);   // render kernel: its direction rows are applied per ray by dir_block (float32)
        if (with_dir) emit_segment16(dst, n.c1, ds, 8, 0, dirrow);
        emit_segment16(dst, n.c2, hs, 4, 0, hidrow);
    } else {
        emit_segment16(dst, n.c2, hs, 4, dm, hidrow);
        emit_segment16(dst, n.c2, ds, 4, 0, dirrow);
    }
    const int rec = stream16_records(v.n_geo, v.n_app, v.cd, with_dir, v.ipe), pad = stream16_padded(v.n_geo, v.n_app, v.cd, with_dir, v.ipe);
    memset(dst, 0, (size_t)(pad - rec) * 1024);
}

static size_t packed_floats(const Variant &v) {
    const Geometry g = make_geometry(v.n_geo, v.n_app, v.cd, v.ipe);
    return (size_t)(g.padded_records + RING) * REC_FLOATS + g.aux_floats;
}
// the same through the model descriptor, whose architecture sizes the flex family's image
static size_t weight_count_of(int v, const ntx_model_desc *d) {
    return kVariants[v].flex ? view_blob_flex(flex_arch_of(d), dims_of(d), nullptr).count : view_blob(kVariants[v], dims_of(d), nullptr).count;
}
static size_t packed_floats_of(int v, const ntx_model_desc *d) {
    return kVariants[v].flex ? packed_floats_flex(flex_arch_of(d)) : packed_floats(kVariants[v]);
}
static size_t aux_floats_of_variant(int v) {
    const Variant &k = kVariants[v];
    return k.flex ? (size_t)aux_total() + flex_floats() : (size_t)make_geometry(k.n_geo, k.n_app, k.cd, k.ipe).aux_floats;
}
static int no_fp16x3(const Variant &v) {
    return v.flex ? fail(NTX_E_UNSUPPORTED, "fp16x3 is built for the 8x256 / skips [4] / color_depth 1 families only; this model's architecture runs on "
                                            "the float32 layer-loop kernels") : NTX_OK;
}

// ---------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------
struct ntx_ctx {
    int variant;
    int device;
    int n_cus;
    int n_wgs;
    float *packed;        // device: stream | tail | aux
    size_t stream_floats; // incl. tail
    size_t n_packed;
    ntx_model_desc_ex descx;   // (the base descriptor, and param_depth / param_width when kind = NTX_MODEL_PARAMNERF_EX)
    uint16_t *packed16;   // device: fp16x3 stream; shares the f32 aux block
    size_t packed16_bytes;
    uint16_t *packed16i;  // device: fp16x3 stream of the kernels with per-sample directions (C1 with its direction segment); ParamNerf only
    size_t packed16i_bytes;
    int32_t *hit_list;    // device scratch of ntx_render_rays: compacted hit-ray indices, sized by ntx_reserve
    size_t hit_cap;
    int32_t *hit_count;   // device int32[8]: [0] number of hit rays, [1] work counter of the instance kernel, [2..5] its chunk table (inst_order_kernel)
    bool hoist_dir;       // false when NERFTEX_NO_DIR_HOIST is set at ntx_create (A/B knob for tests: same bits either way)
    uint16_t *inst_sidx;  // device scratch of ntx_render_instanced: per wave, the execution list of its bundle of rays in flight (8.5 KiB each)
    // ntx_set_weights_device: where every float of the packed image comes from -- an index into the weight blob, or -1 and a constant
    int32_t *gather_idx = nullptr; float *gather_const = nullptr;
    bool x3_stale = false;   // the fp16x3 images were not remade by the last ntx_set_weights_device
};

// The big kernels of each model family live in their own translation units (ntx_variant.hip / ntx_variant_x3.hip
// compiled with -DNTX_VARIANT=k, the hoisted render kernels with -DNTX_HOIST=1|2) so that the build parallelises; this file
// only dispatches to them through one table.
namespace ntx {
#define NTX_DECL(k)                                                                  \
    hipError_t launch_render_v##k(int n_wgs, RenderArgs &a, hipStream_t st);         \
    hipError_t launch_mlp_v##k(int n_wgs, MlpArgs &a, hipStream_t st);               \
    hipError_t launch_instance_v##k(int n_wgs, InstanceArgs &a, hipStream_t st);     \
    hipError_t launch_render_hoist_v##k(int n_wgs, RenderArgs &a, hipStream_t st);   \
    hipError_t launch_render_hoist2_v##k(int n_wgs, RenderArgs &a, hipStream_t st);  \
    hipError_t launch_render_hoist3_v##k(int n_wgs, RenderArgs &a, hipStream_t st);  \
    hipError_t launch_render_x3_v##k(int n_wgs, RenderArgs &a, hipStream_t st);      \
    hipError_t launch_mlp_x3_v##k(int n_wgs, MlpArgs &a, hipStream_t st);            \
    hipError_t launch_instance_x3_v##k(int n_wgs, InstanceArgs &a, hipStream_t st);
NTX_DECL(0) NTX_DECL(1) NTX_DECL(2) NTX_DECL(3) NTX_DECL(4) NTX_DECL(5) NTX_DECL(6) NTX_DECL(7)
#undef NTX_DECL
}  // namespace ntx

struct Launchers {
    hipError_t (*render)(int, RenderArgs &, hipStream_t);
    hipError_t (*render_hoist)(int, RenderArgs &, hipStream_t);   // NULL: plain Nerf has no per-ray direction segment to hoist out of C1
    hipError_t (*render_hoist2)(int, RenderArgs &, hipStream_t);  // + the geometry-parameter blocks of L0 / L5 per ray; NULL: not built for the family
    hipError_t (*render_hoist3)(int, RenderArgs &, hipStream_t);  // + all of them but parameter 0's (blur_idx = 0); NULL: not built
    hipError_t (*mlp)(int, MlpArgs &, hipStream_t);
    hipError_t (*instance)(int, InstanceArgs &, hipStream_t);
    hipError_t (*render_x3)(int, RenderArgs &, hipStream_t);
    hipError_t (*mlp_x3)(int, MlpArgs &, hipStream_t);
    hipError_t (*instance_x3)(int, InstanceArgs &, hipStream_t);
};
#define NTX_ROW(k, hoist, hoist2, hoist3) {launch_render_v##k, hoist, hoist2, hoist3, launch_mlp_v##k, launch_instance_v##k, launch_render_x3_v##k, launch_mlp_x3_v##k, launch_instance_x3_v##k}
static const Launchers kLaunch[] = {   // indexed like kVariants
    NTX_ROW(0, launch_render_hoist_v0, launch_render_hoist2_v0, nullptr),
#ifndef NTX_DEV_ONLY_CARPET   // development builds link only the carpet family (compile time)
    NTX_ROW(1, launch_render_hoist_v1, launch_render_hoist2_v1, nullptr), NTX_ROW(2, launch_render_hoist_v2, nullptr, launch_render_hoist3_v2),
    NTX_ROW(3, nullptr, nullptr, nullptr), NTX_ROW(4, launch_render_hoist_v4, nullptr, nullptr), NTX_ROW(5, launch_render_hoist_v5, nullptr, nullptr),
    {launch_render_v6, nullptr, nullptr, nullptr, launch_mlp_v6, launch_instance_v6, nullptr, nullptr, nullptr},   // flex: float32, everything per sample
    {launch_render_v7, nullptr, nullptr, nullptr, launch_mlp_v7, launch_instance_v7, nullptr, nullptr, nullptr},   // flex with parameter branches
#else
    {}, {}, {}, {}, {}, {}, {},
#endif
};
#undef NTX_ROW
template <class Fn, class Args>
static hipError_t launch(Fn fn, const ntx_ctx *c, Args &a, hipStream_t st) {
    return fn ? fn(c->n_wgs, a, st) : hipErrorNotSupported;
}

// parameter slots of the kernel family <- columns of the caller's parameter rows (identity for the tuned families; the
// generic family has GEN_NGEO + GEN_NAPP slots and feeds 0 into those the model does not have)
template <class Args>
static void fill_param_map(const ntx_ctx *c, Args &a) {
    const Variant &v = kVariants[c->variant];
    const Dims m = dims_of(&c->descx.base);
    a.np_in = m.g + m.a + v.ipe;
    for (int k = 0; k < MAX_PARAM_SLOTS; ++k) a.pmap[k] = -1;
    for (int k = 0; k < v.n_geo && k < MAX_PARAM_SLOTS; ++k) a.pmap[k] = k < m.g ? (int8_t)k : (int8_t)-1;
    for (int j = 0; j < v.n_app && v.n_geo + j < MAX_PARAM_SLOTS; ++j) a.pmap[v.n_geo + j] = j < m.a ? (int8_t)(m.g + j) : (int8_t)-1;
}
// blur_idx (a column of the caller's rows) -> the slot the kernel compares with
static int blur_slot(const ntx_ctx *c, int blur_idx) {
    const Variant &v = kVariants[c->variant];
    const Dims m = dims_of(&c->descx.base);
    if (blur_idx < 0 || !v.gen) return blur_idx;
    return blur_idx < m.g ? blur_idx : v.n_geo + (blur_idx - m.g);
}

// ntx_render_opts (ABI v3) -> the generator's ray index map; identity when opts is NULL or the map is all zero
struct IndexMap {
    int64_t idx0, stride;
    uint32_t run;
};
static int index_map_of(const ntx_render_opts *o, IndexMap *m) {
    *m = IndexMap{0, 0, 0xffffffffu};
    if (!o) return NTX_OK;
    if (o->size < NTX_RENDER_OPTS_V3_SIZE) return fail(NTX_E_INVALID, "ntx_render_opts.size %u < %u: set it to sizeof(ntx_render_opts)", o->size, NTX_RENDER_OPTS_V3_SIZE);
    if (o->ray_index0 == 0 && o->ray_run_length == 0 && o->ray_run_stride == 0) return NTX_OK;
    if (o->ray_index0 < 0 || o->ray_run_length < 1 || o->ray_run_stride < o->ray_run_length)
        return fail(NTX_E_INVALID, "bad ray index map: index0 %lld run_length %lld run_stride %lld", (long long)o->ray_index0,
                    (long long)o->ray_run_length, (long long)o->ray_run_stride);
    m->idx0 = o->ray_index0; m->stride = o->ray_run_stride;
    m->run = o->ray_run_length > 0xffffffffLL ? 0xffffffffu : (uint32_t)o->ray_run_length;   // local rays are < 2^31: one run then
    return NTX_OK;
}
static int noise_of(const ntx_render_opts *o, uint32_t flags, float *std_out) {
    *std_out = 0.0f;
    if (!(flags & NTX_FLAG_RAW_NOISE)) return NTX_OK;
    if (!o) return fail(NTX_E_INVALID, "NTX_FLAG_RAW_NOISE needs ntx_render_opts.raw_noise_std");
    if (!(o->raw_noise_std >= 0.0f) || std::isinf(o->raw_noise_std)) return fail(NTX_E_INVALID, "raw_noise_std must be finite and >= 0");
    *std_out = o->raw_noise_std;
    return NTX_OK;
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" {

int ntx_abi_version(void) { return NTX_ABI_VERSION; }
const char *ntx_last_error(void) { return g_err; }

size_t ntx_weight_count(const ntx_model_desc *desc) {
    const int v = find_variant(desc);
    if (v < 0) { unsupported(desc); return 0; }
    return weight_count_of(v, desc);
}

size_t ntx_packed_count(const ntx_model_desc *desc) {
    const int v = find_variant(desc);
    if (v < 0) { unsupported(desc); return 0; }
    return packed_floats_of(v, desc);
}

int ntx_pack_weights(const ntx_model_desc *desc, const float *weights_host, size_t n_floats, float *packed_out,
                     size_t n_packed) {
    const int v = find_variant(desc);
    if (v < 0) return unsupported(desc);
    if (!weights_host || !packed_out) return fail(NTX_E_INVALID, "NULL buffer");
    if (n_floats != weight_count_of(v, desc))
        return fail(NTX_E_INVALID, "weight blob has %zu floats, model needs %zu", n_floats, weight_count_of(v, desc));
    if (n_packed != packed_floats_of(v, desc))
        return fail(NTX_E_INVALID, "packed buffer has %zu floats, needs %zu", n_packed, packed_floats_of(v, desc));
    if (kVariants[v].flex) pack_flex(flex_arch_of(desc), dims_of(desc), weights_host, packed_out);
    else pack(kVariants[v], dims_of(desc), weights_host, packed_out);
    return NTX_OK;
}

size_t ntx_packed_fp16x3_bytes(const ntx_model_desc *desc) {
    const int v = find_variant(desc);
    if (v < 0) { unsupported(desc); return 0; }
    if (no_fp16x3(kVariants[v])) return 0;
    return packed16_bytes(kVariants[v]);
}

int ntx_pack_weights_fp16x3(const ntx_model_desc *desc, const float *weights_host, size_t n_floats, uint16_t *packed_out,
                            size_t n_bytes) {
    const int v = find_variant(desc);
    if (v < 0) return unsupported(desc);
    if (int rc = no_fp16x3(kVariants[v])) return rc;
    if (!weights_host || !packed_out) return fail(NTX_E_INVALID, "NULL buffer");
    if (n_floats != view_blob(kVariants[v], dims_of(desc), nullptr).count)
        return fail(NTX_E_INVALID, "weight blob has %zu floats, model needs %zu", n_floats,
                    view_blob(kVariants[v], dims_of(desc), nullptr).count);
    if (n_bytes != packed16_bytes(kVariants[v]))
        return fail(NTX_E_INVALID, "packed buffer has %zu bytes, needs %zu", n_bytes, packed16_bytes(kVariants[v]));
    pack16(kVariants[v], dims_of(desc), weights_host, packed_out);
    return NTX_OK;
}

int ntx_create(const ntx_model_desc *desc, const float *weights_host, size_t n_floats, int device, ntx_ctx **out) {
    if (!out) return fail(NTX_E_INVALID, "out is NULL");
    *out = nullptr;
    const int v = find_variant(desc);
    if (v < 0) return unsupported(desc);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(NTX_E_NODEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return fail(NTX_E_INVALID, "device %d out of range [0,%d)", device, ndev);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(NTX_E_NODEVICE, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    HIP_TRY(hipSetDevice(device));
    ntx_ctx *c = new ntx_ctx();
    c->variant = v;
    c->device = device;
    c->n_cus = prop.multiProcessorCount;
    c->n_wgs = prop.multiProcessorCount;   // one 4-wave workgroup per CU: each wave owns a SIMD's register file
    memset(&c->descx, 0, sizeof(c->descx));
    if (desc->kind == NTX_MODEL_PARAMNERF_EX) c->descx = *reinterpret_cast<const ntx_model_desc_ex *>(desc);
    else c->descx.base = *desc;
    c->n_packed = packed_floats_of(v, desc);
    c->stream_floats = c->n_packed - aux_floats_of_variant(v);
    c->packed = nullptr;
    hipError_t e = hipMalloc((void **)&c->packed, c->n_packed * sizeof(float));
    if (e != hipSuccess) {
        const size_t bytes = c->n_packed * sizeof(float);
        delete c;
        return fail(NTX_E_HIP, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
    }
    c->packed16 = nullptr;
    c->packed16_bytes = 0;
    c->packed16i = nullptr; c->packed16i_bytes = 0;
    c->hit_list = nullptr; c->hit_cap = 0; c->hit_count = nullptr; c->inst_sidx = nullptr;
    c->hoist_dir = getenv("NERFTEX_NO_DIR_HOIST") == nullptr;
    if (!kVariants[v].flex) {   // (the flex family has float32 kernels only)
        c->packed16_bytes = packed16_bytes(kVariants[v]);
        e = hipMalloc((void **)&c->packed16, c->packed16_bytes);
        if (e != hipSuccess) {
            const size_t bytes = c->packed16_bytes;
            (void)hipFree(c->packed);
            delete c;
            return fail(NTX_E_HIP, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
        }
    }
    if (c->packed16 && kVariants[v].cd) {
        c->packed16i_bytes = packed16_bytes(kVariants[v], 1);
        e = hipMalloc((void **)&c->packed16i, c->packed16i_bytes);
        if (e != hipSuccess) {
            const size_t bytes = c->packed16i_bytes;
            (void)hipFree(c->packed); (void)hipFree(c->packed16);
            delete c;
            return fail(NTX_E_HIP, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
        }
    }
    *out = c;
    {   // all the device scratch the entry points will ever use: allocated here (and by ntx_reserve), never per call
        int rc = NTX_OK;
        if (hipMalloc((void **)&c->hit_count, 8 * sizeof(int32_t)) != hipSuccess) rc = fail(NTX_E_HIP, "hipMalloc(hit_count)");
        if (rc == NTX_OK && hipMalloc((void **)&c->inst_sidx, (size_t)c->n_wgs * 4 * INST_EXEC_CAP * sizeof(uint16_t)) != hipSuccess)
            rc = fail(NTX_E_HIP, "hipMalloc(inst_sidx)");
        if (rc == NTX_OK) rc = ntx_reserve(c, NTX_DEFAULT_MAX_RAYS);
        if (rc != NTX_OK) { ntx_destroy(c); *out = nullptr; return rc; }
    }
    if (weights_host) {
        const int rc = ntx_set_weights(c, weights_host, n_floats);
        if (rc != NTX_OK) {
            ntx_destroy(c);
            *out = nullptr;
            return rc;
        }
    } else if (kVariants[v].flex) {   // all-zero weights, but the image carries the architecture
        const std::vector<float> zeros(weight_count_of(v, desc), 0.0f);
        const int rc = ntx_set_weights(c, zeros.data(), zeros.size());
        if (rc != NTX_OK) { ntx_destroy(c); *out = nullptr; return rc; }
    } else {
        HIP_TRY(hipMemset(c->packed, 0, c->n_packed * sizeof(float)));
        if (c->packed16) HIP_TRY(hipMemset(c->packed16, 0, c->packed16_bytes));
        if (c->packed16i) HIP_TRY(hipMemset(c->packed16i, 0, c->packed16i_bytes));
    }
    return NTX_OK;
}

int ntx_reserve(ntx_ctx *ctx, int64_t max_rays) {
    if (!ctx) return fail(NTX_E_INVALID, "ctx is NULL");
    if (max_rays < 0 || max_rays > 0x7fffffff) return fail(NTX_E_INVALID, "max_rays %lld outside [0, 2^31)", (long long)max_rays);
    if ((size_t)max_rays == ctx->hit_cap && (ctx->hit_list || max_rays == 0)) return NTX_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipDeviceSynchronize());   // a launch may still be walking the old list
    if (ctx->hit_list) HIP_TRY(hipFree(ctx->hit_list));
    ctx->hit_list = nullptr; ctx->hit_cap = 0;
    // ntx_render_rays: hit_list[max_rays]; ntx_render_instanced: order[max_rays] | count[max_rays]
    if (max_rays > 0) HIP_TRY(hipMalloc((void **)&ctx->hit_list, (size_t)max_rays * 2 * sizeof(int32_t)));
    ctx->hit_cap = (size_t)max_rays;
    return NTX_OK;
}

// packed[i] = idx[i] >= 0 ? w[idx[i]] : konst[i]: the weight image remade where the weights are
__global__ void gather_weights_kernel(const float *__restrict__ w, const int32_t *__restrict__ idx, const float *__restrict__ konst, size_t n, float *__restrict__ packed) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t g = idx[i];
    packed[i] = g >= 0 ? w[g] : konst[i];
}

int ntx_set_weights_device(ntx_ctx *ctx, const float *weights_dev, size_t n_floats, ntx_stream stream) {
    if (!ctx || !weights_dev) return fail(NTX_E_INVALID, "NULL argument");
    const size_t want = weight_count_of(ctx->variant, &ctx->descx.base);
    if (n_floats != want) return fail(NTX_E_INVALID, "weight blob has %zu floats, model needs %zu", n_floats, want);
    HIP_TRY(hipSetDevice(ctx->device));
    if (!ctx->gather_idx) {
        // The packer only PLACES weights (and a few constants: the flex family's descriptor, zero padding).  Packing two blobs of the weights'
        // own numbers -- i + 1 and 2 (i + 1), exact in float32 below 2^23 -- tells every packed float's source: doubled = weight i, equal = constant.
        if (n_floats >= (size_t)1 << 22) return fail(NTX_E_UNSUPPORTED, "ntx_set_weights_device: the model has more than 2^22 weights");
        std::vector<float> b1(n_floats), b2(n_floats), p1(ctx->n_packed), p2(ctx->n_packed);
        for (size_t i = 0; i < n_floats; ++i) { b1[i] = (float)(i + 1); b2[i] = (float)(2 * (i + 1)); }
        int rc = ntx_pack_weights(&ctx->descx.base, b1.data(), n_floats, p1.data(), p1.size());
        if (rc == NTX_OK) rc = ntx_pack_weights(&ctx->descx.base, b2.data(), n_floats, p2.data(), p2.size());
        if (rc != NTX_OK) return rc;
        std::vector<int32_t> idx(ctx->n_packed);
        for (size_t i = 0; i < ctx->n_packed; ++i) {
            const float a = p1[i], b = p2[i];
            if (a >= 1.0f && a <= (float)n_floats && b == 2.0f * a && a == std::floor(a)) idx[i] = (int32_t)a - 1;
            else if (memcmp(&a, &b, sizeof(float)) == 0) idx[i] = -1;
            else return fail(NTX_E_UNSUPPORTED, "ntx_set_weights_device: packed float %zu is neither a weight nor a constant", i);
        }
        HIP_TRY(hipMalloc((void **)&ctx->gather_idx, ctx->n_packed * sizeof(int32_t)));
        HIP_TRY(hipMalloc((void **)&ctx->gather_const, ctx->n_packed * sizeof(float)));
        HIP_TRY(hipMemcpy(ctx->gather_idx, idx.data(), ctx->n_packed * sizeof(int32_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(ctx->gather_const, p1.data(), ctx->n_packed * sizeof(float), hipMemcpyHostToDevice));
    }
    gather_weights_kernel<<<dim3((unsigned)((ctx->n_packed + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(weights_dev, ctx->gather_idx, ctx->gather_const, ctx->n_packed,
                                                                                                                 ctx->packed);
    HIP_TRY(hipGetLastError());
    ctx->x3_stale = ctx->packed16 != nullptr;
    return NTX_OK;
}

int ntx_set_weights(ntx_ctx *ctx, const float *weights_host, size_t n_floats) {
    if (!ctx || !weights_host) return fail(NTX_E_INVALID, "NULL argument");
    std::vector<float> packed(ctx->n_packed);
    const int rc = ntx_pack_weights(&ctx->descx.base, weights_host, n_floats, packed.data(), packed.size());
    if (rc != NTX_OK) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpy(ctx->packed, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice));
    if (ctx->packed16) {
        std::vector<uint16_t> p16(ctx->packed16_bytes / 2);
        pack16(kVariants[ctx->variant], dims_of(&ctx->descx.base), weights_host, p16.data());
        HIP_TRY(hipMemcpy(ctx->packed16, p16.data(), ctx->packed16_bytes, hipMemcpyHostToDevice));
    }
    if (ctx->packed16i) {
        std::vector<uint16_t> p16(ctx->packed16i_bytes / 2);
        pack16(kVariants[ctx->variant], dims_of(&ctx->descx.base), weights_host, p16.data(), 1);
        HIP_TRY(hipMemcpy(ctx->packed16i, p16.data(), ctx->packed16i_bytes, hipMemcpyHostToDevice));
    }
    ctx->x3_stale = false;
    return NTX_OK;
}

int ntx_destroy(ntx_ctx *ctx) {
    if (!ctx) return NTX_OK;
    if (ctx->packed) (void)hipFree(ctx->packed);
    if (ctx->packed16) (void)hipFree(ctx->packed16);
    if (ctx->packed16i) (void)hipFree(ctx->packed16i);
    if (ctx->hit_list) (void)hipFree(ctx->hit_list);
    if (ctx->hit_count) (void)hipFree(ctx->hit_count);
    if (ctx->inst_sidx) (void)hipFree(ctx->inst_sidx);
    if (ctx->gather_idx) (void)hipFree(ctx->gather_idx);
    if (ctx->gather_const) (void)hipFree(ctx->gather_const);
    delete ctx;
    return NTX_OK;
}

int ntx_kernel_info(ntx_ctx *ctx, int *n_workgroups, int *threads_per_workgroup, int *n_cus) {
    if (!ctx) return fail(NTX_E_INVALID, "ctx is NULL");
    if (n_workgroups) *n_workgroups = ctx->n_wgs;
    if (threads_per_workgroup) *threads_per_workgroup = 256;
    if (n_cus) *n_cus = ctx->n_cus;
    return NTX_OK;
}

int ntx_generate_rays_strided(const float *c2w, int height, int width, float focal, int64_t pixel0, int64_t n_pixels,
                              int64_t run_length, int64_t run_stride, int mode, const float *b0, const float *b1,
                              float near_t, float far_t, float *rays_o, float *rays_d, float *t, float *cone_scale,
                              ntx_stream stream) {
    if (!c2w || !rays_o || !rays_d || !t || !cone_scale) return fail(NTX_E_INVALID, "NULL buffer");
    if (height <= 0 || width <= 0 || n_pixels < 0 || pixel0 < 0 || run_length < 1 || run_stride < run_length)
        return fail(NTX_E_INVALID, "bad pixel set: pixel0 %lld n %lld run_length %lld run_stride %lld", (long long)pixel0,
                    (long long)n_pixels, (long long)run_length, (long long)run_stride);
    if (n_pixels > 0) {
        const int64_t last = pixel0 + ((n_pixels - 1) / run_length) * run_stride + (n_pixels - 1) % run_length;
        if (last >= (int64_t)height * width)
            return fail(NTX_E_INVALID, "pixel set [%lld .. %lld] outside %dx%d", (long long)pixel0, (long long)last, height, width);
    }
    if (mode != 0 && mode != 1) return fail(NTX_E_INVALID, "mode must be 0 (Proxy/AABB) or 1 (Frustum)");
    if (mode == 0 && (!b0 || !b1)) return fail(NTX_E_INVALID, "AABB bounds are NULL");
    if (n_pixels == 0) return NTX_OK;
    RaygenArgs a{};
    memcpy(a.c2w, c2w, sizeof(a.c2w));
    if (mode == 0) { memcpy(a.b0, b0, sizeof(a.b0)); memcpy(a.b1, b1, sizeof(a.b1)); }
    a.focal = focal;
    a.half_w = (float)(.5 * width);   // `.5 * width` is evaluated by python, then cast (ray_sampler.py:41)
    a.half_h = (float)(.5 * height);
    a.near_t = near_t; a.far_t = far_t;
    a.width = width; a.mode = mode;
    a.pixel0 = pixel0; a.n = n_pixels;
    a.run_length = run_length; a.run_stride = run_stride;
    a.rays_o = rays_o; a.rays_d = rays_d; a.t = t; a.cone = cone_scale;
    const int64_t nb = (n_pixels + 255) / 256;
    raygen_kernel<<<dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream>>>(a);
    HIP_TRY(hipGetLastError());
    return NTX_OK;
}

int ntx_generate_rays_at(const float *c2w, int height, int width, float focal, const float *image_plane_loc, int64_t n_rays, int mode,
                         const float *b0, const float *b1, float near_t, float far_t, float *rays_o, float *rays_d, float *t,
                         float *cone_scale, ntx_stream stream) {
    if (!c2w || !rays_o || !rays_d || !t || !cone_scale) return fail(NTX_E_INVALID, "NULL buffer");
    if (height <= 0 || width <= 0 || n_rays < 0) return fail(NTX_E_INVALID, "bad shape %dx%d, n_rays %lld", height, width, (long long)n_rays);
    if (mode != 0 && mode != 1) return fail(NTX_E_INVALID, "mode must be 0 (Proxy/AABB) or 1 (Frustum)");
    if (mode == 0 && (!b0 || !b1)) return fail(NTX_E_INVALID, "AABB bounds are NULL");
    if (n_rays == 0) return NTX_OK;
    if (!image_plane_loc) return fail(NTX_E_INVALID, "image_plane_loc is NULL");
    RaygenArgs a{};
    memcpy(a.c2w, c2w, sizeof(a.c2w));
    if (mode == 0) { memcpy(a.b0, b0, sizeof(a.b0)); memcpy(a.b1, b1, sizeof(a.b1)); }
    a.focal = focal;
    a.half_w = (float)(.5 * width); a.half_h = (float)(.5 * height);          // ray_sampler.py:41
    a.near_t = near_t; a.far_t = far_t;
    a.width = width; a.mode = mode;
    a.pixel0 = 0; a.n = n_rays; a.run_length = 1; a.run_stride = 1;
    a.loc = image_plane_loc;
    a.rays_o = rays_o; a.rays_d = rays_d; a.t = t; a.cone = cone_scale;
    raygen_kernel<<<dim3((unsigned)((n_rays + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(a);
    HIP_TRY(hipGetLastError());
    return NTX_OK;
}

int ntx_aabb_intersect(const float *rays_o, const float *rays_d, int64_t n_rays, const float *b0, const float *b1, float *t,
                       ntx_stream stream) {
    if (n_rays < 0) return fail(NTX_E_INVALID, "n_rays < 0");
    if (!b0 || !b1) return fail(NTX_E_INVALID, "AABB bounds are NULL");
    if (n_rays == 0) return NTX_OK;
    if (!rays_o || !rays_d || !t) return fail(NTX_E_INVALID, "NULL buffer");
    aabb_kernel<<<dim3((unsigned)((n_rays + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(rays_o, rays_d, n_rays, b0[0], b0[1], b0[2],
                                                                                                b1[0], b1[1], b1[2], t);
    HIP_TRY(hipGetLastError());
    return NTX_OK;
}

int ntx_generate_rays(const float *c2w, int height, int width, float focal, int64_t pixel0, int64_t n_pixels,
                      int mode, const float *b0, const float *b1, float near_t, float far_t, float *rays_o,
                      float *rays_d, float *t, float *cone_scale, ntx_stream stream) {
    const int64_t run = n_pixels > 0 ? n_pixels : 1;
    return ntx_generate_rays_strided(c2w, height, width, focal, pixel0, n_pixels, run, run, mode, b0, b1, near_t, far_t, rays_o,
                                     rays_d, t, cone_scale, stream);
}

int ntx_fourier_features(const float *x, int64_t m, int d, int n_freq, float *out, ntx_stream stream) {
    if (m < 0 || d <= 0 || n_freq < 0 || n_freq > 30) return fail(NTX_E_INVALID, "bad shape m=%lld d=%d n_freq=%d", (long long)m, d, n_freq);
    if (m == 0) return NTX_OK;
    if (!x || !out) return fail(NTX_E_INVALID, "NULL buffer");
    const int64_t nb = (m * d + 255) / 256;
    fourier_kernel<<<dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream>>>(x, m, d, n_freq, out);
    HIP_TRY(hipGetLastError());
    return NTX_OK;
}

int ntx_mlp_forward(ntx_ctx *ctx, const float *pos, const float *dirs, const float *params, int64_t m, uint32_t flags,
                    float *color_out, float *sigma_out, ntx_stream stream) {
    if (!ctx) return fail(NTX_E_INVALID, "ctx is NULL");
    if (m < 0) return fail(NTX_E_INVALID, "m < 0");
    if (flags & ~NTX_FLAG_FP16X3) return fail(NTX_E_INVALID, "ntx_mlp_forward takes NTX_FLAG_FP16X3 or 0, got 0x%x", flags);
    if (m == 0) return NTX_OK;
    const Variant &v = kVariants[ctx->variant];
    if (flags & NTX_FLAG_FP16X3) if (int rc = no_fp16x3(v)) return rc;
    if ((flags & NTX_FLAG_FP16X3) && ctx->x3_stale)
        return fail(NTX_E_UNSUPPORTED, "the fp16x3 images are stale: the weights last came from device memory (ntx_set_weights_device remakes the float32 image only); "
                                       "ntx_set_weights remakes all of them");
    const Dims dm_ = dims_of(&ctx->descx.base);
    if (!pos || !dirs || !color_out || !sigma_out || (!params && dm_.g + dm_.a > 0))
        return fail(NTX_E_INVALID, "NULL buffer");
    HIP_TRY(hipSetDevice(ctx->device));   // the launch goes to the context's device whatever the caller's current one is
    MlpArgs a{};
    a.wstream = reinterpret_cast<const f32x4 *>(ctx->packed);
    a.stream_bytes = (uint32_t)(ctx->stream_floats * sizeof(float));
    a.aux = ctx->packed + ctx->stream_floats;
    a.pos = pos; a.dirs = dirs; a.params = params;
    a.color_out = color_out; a.sigma_out = sigma_out;
    a.m = m;
    fill_param_map(ctx, a);
    if (flags & NTX_FLAG_FP16X3) {
        // directions are per sample: ParamNerf uses the stream that keeps C1's direction segment; plain Nerf's one stream
        // has it in C2 anyway
        a.wstream = reinterpret_cast<const f32x4 *>(v.cd ? ctx->packed16i : ctx->packed16);
        a.stream_bytes = (uint32_t)(v.cd ? ctx->packed16i_bytes : ctx->packed16_bytes);
        HIP_TRY(launch(kLaunch[ctx->variant].mlp_x3, ctx, a, (hipStream_t)stream));
        return NTX_OK;
    }
    HIP_TRY(launch(kLaunch[ctx->variant].mlp, ctx, a, (hipStream_t)stream));
    return NTX_OK;
}

int ntx_composite(const float *color, const float *sigma, const float *z_vals, const float *rays_d, int64_t n_rays,
                  int n_samples, uint32_t flags, const float *bkgd, float *color_out, float *alpha_out,
                  float *weights_out, ntx_stream stream) {
    if (n_rays < 0) return fail(NTX_E_INVALID, "n_rays < 0");
    if (n_samples < 2) return fail(NTX_E_INVALID, "n_samples must be >= 2 (renderer.py:174-177 needs a previous step)");
    if (n_rays == 0) return NTX_OK;
    if (!color || !sigma || !z_vals || !rays_d || !color_out || !alpha_out) return fail(NTX_E_INVALID, "NULL buffer");
    CompositeArgs a{};
    a.color = color; a.sigma = sigma; a.z = z_vals; a.rays_d = rays_d;
    a.color_out = color_out; a.alpha_out = alpha_out; a.weights_out = weights_out;
    a.n_rays = n_rays; a.n_samples = n_samples; a.flags = flags;
    for (int k = 0; k < 3; ++k) a.bkgd[k] = bkgd ? bkgd[k] : 1.0f;
    int64_t nb = (n_rays + 3) / 4;
    if (nb > 256 * 8) nb = 256 * 8;   // 8 workgroups per CU, grid-stride over rays
    composite_kernel<<<dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream>>>(a);
    HIP_TRY(hipGetLastError());
    return NTX_OK;
}

int ntx_render_rays(ntx_ctx *ctx, const float *rays_o, const float *rays_d, const float *t, const float *params,
                    int64_t rays_per_param_row, const float *cone_scale, int64_t n_rays, int n_samples, int blur_idx,
                    uint32_t flags, const float *bkgd, const float *z_vals, uint64_t perturb_seed, const ntx_render_opts *opts,
                    float *color_out, float *alpha_out, float *weights_out, int32_t *status_flag, ntx_stream stream) {
    // every check comes before the first launch: a call that fails has written nothing
    if (!ctx) return fail(NTX_E_INVALID, "ctx is NULL");
    if (n_rays < 0) return fail(NTX_E_INVALID, "n_rays < 0");
    if (n_samples < 2) return fail(NTX_E_INVALID, "n_samples must be >= 2 (renderer.py:174-177 needs a previous step)");
    if (n_rays == 0) return NTX_OK;
    const Variant &v = kVariants[ctx->variant];
    if (flags & NTX_FLAG_FP16X3) if (int rc = no_fp16x3(v)) return rc;
    if ((flags & NTX_FLAG_FP16X3) && ctx->x3_stale)
        return fail(NTX_E_UNSUPPORTED, "the fp16x3 images are stale: the weights last came from device memory (ntx_set_weights_device remakes the float32 image only); "
                                       "ntx_set_weights remakes all of them");
    const Dims dm_ = dims_of(&ctx->descx.base);
    const int np = dm_.g + dm_.a + v.ipe;   // parameters per row at the ABI (mip: incl. the spliced-out blur parameter)
    if (!rays_o || !rays_d || !t || !color_out || !alpha_out || (!params && np > 0))
        return fail(NTX_E_INVALID, "NULL buffer");
    if (rays_per_param_row < 1) return fail(NTX_E_INVALID, "rays_per_param_row must be >= 1");
    if (blur_idx < -1 || blur_idx >= np) return fail(NTX_E_INVALID, "blur_idx %d outside [-1,%d)", blur_idx, np);
    if (v.ipe && blur_idx < 0) return fail(NTX_E_INVALID, "an IPE (mip) model needs blur_idx: the cone radius parameter (renderer.py:385)");
    if (blur_idx >= 0 && !cone_scale) return fail(NTX_E_INVALID, "blur_idx set but cone_scale is NULL");
    if ((size_t)n_rays > ctx->hit_cap)
        return fail(NTX_E_INVALID, "n_rays %lld exceeds the %zu rays this context reserved; call ntx_reserve first", (long long)n_rays, ctx->hit_cap);
    IndexMap im;
    float noise_std;
    if (int rc = index_map_of(opts, &im)) return rc;
    if (int rc = noise_of(opts, flags, &noise_std)) return rc;
    const bool x3 = (flags & NTX_FLAG_FP16X3) != 0;
    // the per-ray direction vector is valid unless the blur scaling hits an APPEARANCE parameter per sample (renderer.py:155-158)
    const bool dir_const = v.cd && (blur_idx < 0 || blur_idx < dm_.g || v.ipe);
    if (x3 && v.cd && !dir_const)
        return fail(NTX_E_UNSUPPORTED, "fp16x3: blur_idx %d scales an appearance parameter per sample; use float32", blur_idx);
    RenderArgs a{};
    a.wstream = reinterpret_cast<const f32x4 *>(ctx->packed);
    a.stream_bytes = (uint32_t)(ctx->stream_floats * sizeof(float));
    a.aux = ctx->packed + ctx->stream_floats;
    a.rays_o = rays_o; a.rays_d = rays_d; a.t = t; a.params = params; a.cone = cone_scale; a.z_vals = z_vals;
    a.color_out = color_out; a.alpha_out = alpha_out; a.weights_out = weights_out; a.status = status_flag;
    a.n_rays = n_rays; a.rays_per_row = rays_per_param_row;
    a.n_samples = n_samples; a.blur_idx = blur_slot(ctx, blur_idx); a.flags = flags;
    fill_param_map(ctx, a);
    a.delta = (1.0f - 0.0f) / (float)(n_samples - 1 + v.ipe);   // mip: S+1 segment edges (renderer.py:374)
    a.seed_lo = (uint32_t)perturb_seed; a.seed_hi = (uint32_t)(perturb_seed >> 32);
    a.raw_noise_std = noise_std; a.idx0 = im.idx0; a.idx_run = im.run; a.idx_stride = im.stride;
    for (int k = 0; k < 3; ++k) a.bkgd[k] = bkgd ? bkgd[k] : 1.0f;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ctx->device));
    // Hit-ray compaction: culled rays get their final value here, the render kernel walks the list.  The list lives in the
    // context, so launches on one context must be stream-ordered.
    HIP_TRY(hipMemsetAsync(ctx->hit_count, 0, sizeof(int32_t), st));
    compact_hits_kernel<<<dim3((unsigned)((n_rays + 255) / 256)), dim3(256), 0, st>>>(
        t, n_rays, ctx->hit_list, ctx->hit_count, color_out, alpha_out, flags, a.bkgd[0], a.bkgd[1], a.bkgd[2]);
    HIP_TRY(hipGetLastError());
    a.hit_list = ctx->hit_list; a.hit_count = ctx->hit_count;
    const Launchers &L = kLaunch[ctx->variant];
    if (x3) {
        // ParamNerf: C1's direction segment always enters as the per-ray vector dir_block computes in float32 from the float32 stream
        a.dir_wstream = a.wstream; a.dir_stream_bytes = a.stream_bytes;
        a.wstream = reinterpret_cast<const f32x4 *>(ctx->packed16);
        a.stream_bytes = (uint32_t)ctx->packed16_bytes;
        HIP_TRY(launch(L.render_x3, ctx, a, st));
        return NTX_OK;
    }
    // float32: direction features and appearance parameters are per-ray constants (renderer.py:152-154): the HOIST kernel
    // evaluates the colour layer's direction segment once per ray (dir_block) instead of once per sample -- same bits
    // -- and without a blur_idx the geometry parameters are per-ray constants as well: HOIST = 2 also starts L0 and L5 from
    // per-ray rows (bias + the geometry block of their position segments)
    // -- and with blur_idx = 0 all of them but parameter 0's: HOIST = 3 (its block comes last in the layout)
    if (ctx->hoist_dir && dir_const && blur_idx < 0 && L.render_hoist2) HIP_TRY(launch(L.render_hoist2, ctx, a, st));
    else if (ctx->hoist_dir && dir_const && blur_idx == 0 && dm_.g >= 2 && L.render_hoist3) HIP_TRY(launch(L.render_hoist3, ctx, a, st));
    else if (ctx->hoist_dir && dir_const && L.render_hoist) HIP_TRY(launch(L.render_hoist, ctx, a, st));
    else HIP_TRY(launch(L.render, ctx, a, st));
    return NTX_OK;
}

int ntx_render_instanced(ntx_ctx *ctx, const float *rays_d_map, const float *pts, const float *t, const float *dists,
                         const float *color_last, const float *alpha_last, const float *alpha_weight,
                         const int32_t *instance_id, const uint8_t *hit, const float *params_map, const float *cone_scale,
                         int64_t n_rays, int n_samples, int blur_idx, float patch_scale, float density_scale,
                         uint32_t flags, const float *bkgd, const float *instance_color, const ntx_render_opts *opts,
                         float *color_out, float *alpha_out, int32_t *status_flag, ntx_stream stream) {
    if (!ctx) return fail(NTX_E_INVALID, "ctx is NULL");
    if (n_rays < 0) return fail(NTX_E_INVALID, "n_rays < 0");
    if (n_samples < 1 || n_samples > MAX_INSTANCE_SAMPLES)
        return fail(NTX_E_INVALID, "n_samples %d outside [1,%d]", n_samples, MAX_INSTANCE_SAMPLES);
    if (n_rays == 0) return NTX_OK;
    const Variant &v = kVariants[ctx->variant];
    if (flags & NTX_FLAG_FP16X3) if (int rc = no_fp16x3(v)) return rc;
    if ((flags & NTX_FLAG_FP16X3) && ctx->x3_stale)
        return fail(NTX_E_UNSUPPORTED, "the fp16x3 images are stale: the weights last came from device memory (ntx_set_weights_device remakes the float32 image only); "
                                       "ntx_set_weights remakes all of them");
    const Dims dm_ = dims_of(&ctx->descx.base);
    const int np = dm_.g + dm_.a + v.ipe;
    if (v.ipe && (blur_idx < 0 || !t)) return fail(NTX_E_INVALID, "an IPE (mip) model needs blur_idx and t (renderer.py:511, 575)");
    if (!rays_d_map || !pts || !dists || !color_last || !alpha_last || !hit || !color_out || !alpha_out ||
        (!params_map && np > 0))
        return fail(NTX_E_INVALID, "NULL buffer");
    if (blur_idx < -1 || blur_idx >= np) return fail(NTX_E_INVALID, "blur_idx %d outside [-1,%d)", blur_idx, np);
    if (blur_idx >= 0 && (!cone_scale || !t)) return fail(NTX_E_INVALID, "blur_idx set but cone_scale / t is NULL");
    if (instance_color && !instance_id) return fail(NTX_E_INVALID, "instance_color given without instance_id");
    if (!(patch_scale > 0.0f)) return fail(NTX_E_INVALID, "patch_scale must be > 0");
    if (flags & NTX_FLAG_PERTURB) return fail(NTX_E_INVALID, "NTX_FLAG_PERTURB: the instancer places the samples of this path, there is nothing to jitter");
    IndexMap im;
    float noise_std;
    if (int rc = index_map_of(opts, &im)) return rc;
    if (int rc = noise_of(opts, flags, &noise_std)) return rc;
    InstanceArgs a{};
    a.raw_noise_std = noise_std; a.idx0 = im.idx0; a.idx_run = im.run; a.idx_stride = im.stride;
    a.seed_lo = opts ? (uint32_t)opts->noise_seed : 0u; a.seed_hi = opts ? (uint32_t)(opts->noise_seed >> 32) : 0u;
    a.run_hoist = ctx->hoist_dir ? 1 : 0;
    if (const char *dbg = getenv("NERFTEX_DEBUG_RUNS")) a.run_hoist = atoi(dbg);   // development: ntx_device.h instance_kernel
    a.sidx_scratch = ctx->inst_sidx;
    a.wstream = reinterpret_cast<const f32x4 *>(ctx->packed);
    a.stream_bytes = (uint32_t)(ctx->stream_floats * sizeof(float));
    a.aux = ctx->packed + ctx->stream_floats;
    a.rays_d_map = rays_d_map; a.pts = pts; a.t = t; a.dists = dists; a.color_last = color_last;
    a.alpha_last = alpha_last; a.alpha_weight = alpha_weight; a.params_map = params_map; a.cone = cone_scale;
    a.instance_color = instance_color; a.instance_id = instance_id; a.hit = hit;
    a.color_out = color_out; a.alpha_out = alpha_out; a.status = status_flag;
    a.n_rays = n_rays; a.n_samples = n_samples; a.blur_idx = blur_slot(ctx, blur_idx); a.flags = flags;
    fill_param_map(ctx, a);
    a.patch_scale = patch_scale; a.density_scale = density_scale;
    for (int k = 0; k < 3; ++k) a.bkgd[k] = bkgd ? bkgd[k] : 1.0f;
    if (n_rays > 0x7fffffff) return fail(NTX_E_INVALID, "n_rays %lld exceeds int32", (long long)n_rays);
    // dynamic ray hand-out: a device counter owned by the context (stream-ordered use, like the other scratch)
    if ((size_t)n_rays > ctx->hit_cap)
        return fail(NTX_E_INVALID, "n_rays %lld exceeds the %zu rays this context reserved; call ntx_reserve first", (long long)n_rays, ctx->hit_cap);
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    a.work_counter = ctx->hit_count + 1;   // [0] hits of ntx_render_rays, [1] this counter (inst_order_kernel zeroes it)
    {   // hand the rays out costliest first (ntx_small_kernels.h: inst_*_kernel); scratch reserved in the context
        int32_t *order = ctx->hit_list, *count = ctx->hit_list + ctx->hit_cap;
        inst_count_kernel<<<dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, st>>>(dists, hit, n_rays, n_samples, count);
        // chunks of the hand-out (float32 kernel): the last ta rays per wave not in fours, the last tb single; development knobs in
        // NERFTEX_DEBUG_RUNS: bit 3 = single rays throughout, bits 8-12 / 16-20 = ta / tb
        int ta = 6, tb = 3;
        if ((a.run_hoist >> 8) & 31) ta = (a.run_hoist >> 8) & 31;
        if ((a.run_hoist >> 16) & 31) tb = (a.run_hoist >> 16) & 31;
        if (a.run_hoist & 8) ta = -1;
        a.chunk_tab = ctx->hit_count + 2;
        inst_order_kernel<<<dim3(1), dim3(INST_ORDER_THREADS), 0, st>>>(count, n_rays, order, a.work_counter, ctx->n_wgs * 4, ta, tb, ctx->hit_count + 2);
        HIP_TRY(hipGetLastError());
        a.order = order; a.count = count;
    }
    if (flags & NTX_FLAG_FP16X3) {
        // directions are per sample: ParamNerf uses the stream that keeps C1's direction segment; plain Nerf's one stream
        // has it in C2 anyway
        a.wstream = reinterpret_cast<const f32x4 *>(v.cd ? ctx->packed16i : ctx->packed16);
        a.stream_bytes = (uint32_t)(v.cd ? ctx->packed16i_bytes : ctx->packed16_bytes);
        HIP_TRY(launch(kLaunch[ctx->variant].instance_x3, ctx, a, (hipStream_t)stream));
        return NTX_OK;
    }
    HIP_TRY(launch(kLaunch[ctx->variant].instance, ctx, a, (hipStream_t)stream));
    return NTX_OK;
}

int ntx_sample_depths(const float *t, int64_t n_rays, int n_points, uint32_t flags, uint64_t perturb_seed,
                      const ntx_render_opts *opts, float *z_out, ntx_stream stream) {
    if (n_rays < 0) return fail(NTX_E_INVALID, "n_rays < 0");
    if (n_points < 2) return fail(NTX_E_INVALID, "n_points must be >= 2");
    IndexMap im;
    if (int rc = index_map_of(opts, &im)) return rc;
    if (n_rays > 0x7fffffff) return fail(NTX_E_INVALID, "n_rays %lld exceeds int32", (long long)n_rays);
    if (n_rays == 0) return NTX_OK;
    if (!t || !z_out) return fail(NTX_E_INVALID, "NULL buffer");
    const int64_t n = n_rays * n_points;
    sample_depths_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(
        t, n_rays, n_points, 1.0f / (float)(n_points - 1), flags, (uint32_t)perturb_seed, (uint32_t)(perturb_seed >> 32), im.idx0, im.run,
        im.stride, z_out);
    HIP_TRY(hipGetLastError());
    return NTX_OK;
}

// noise_out[ray][i] = raw_noise_std * N(0,1): the very draws the render kernels add to the density (ntx_device.h normal01)
__global__ __launch_bounds__(256) void sample_noise_kernel(int64_t n_rays, int npts, float noise_std, uint32_t seed_lo, uint32_t seed_hi, int64_t idx0,
                                                           uint32_t idx_run, int64_t idx_stride, float *noise_out) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_rays * npts) return;
    const int64_t ray = k / npts;
    noise_out[k] = noise_std * normal01(global_index(idx0, idx_run, idx_stride, ray), (int)(k % npts), seed_lo, seed_hi);
}

int ntx_sample_noise(int64_t n_rays, int n_points, uint64_t seed, const ntx_render_opts *opts, float *noise_out, ntx_stream stream) {
    if (n_rays < 0 || n_points < 1) return fail(NTX_E_INVALID, "n_rays < 0 or n_points < 1");
    IndexMap im;
    if (int rc = index_map_of(opts, &im)) return rc;
    float noise_std = 0.0f;
    if (int rc = noise_of(opts, NTX_FLAG_RAW_NOISE, &noise_std)) return rc;
    if (n_rays > 0x7fffffff) return fail(NTX_E_INVALID, "n_rays %lld exceeds int32", (long long)n_rays);
    if (n_rays == 0) return NTX_OK;
    if (!noise_out) return fail(NTX_E_INVALID, "NULL buffer");
    const int64_t n = n_rays * n_points;
    sample_noise_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(n_rays, n_points, noise_std, (uint32_t)seed, (uint32_t)(seed >> 32), im.idx0,
                                                                                                   im.run, im.stride, noise_out);
    HIP_TRY(hipGetLastError());
    return NTX_OK;
}

int ntx_sample_pdf(const float *t, const float *z_vals, const float *weights, const float *u, int64_t n_rays,
                   int n_samples, int n_importance, uint32_t flags, uint64_t perturb_seed, const ntx_render_opts *opts, float *z_out,
                   ntx_stream stream) {
    if (n_rays < 0) return fail(NTX_E_INVALID, "n_rays < 0");
    IndexMap im;
    if (int rc = index_map_of(opts, &im)) return rc;
    if (n_rays > 0x7fffffff) return fail(NTX_E_INVALID, "n_rays %lld exceeds int32", (long long)n_rays);
    if (n_samples < 3 || n_samples > MAX_PDF_SAMPLES) return fail(NTX_E_INVALID, "n_samples %d outside [3,%d]", n_samples, MAX_PDF_SAMPLES);
    if (n_importance < 1 || n_importance > MAX_PDF_SAMPLES) return fail(NTX_E_INVALID, "n_importance %d outside [1,%d]", n_importance, MAX_PDF_SAMPLES);
    if (n_rays == 0) return NTX_OK;
    if (!t || !weights || !z_out) return fail(NTX_E_INVALID, "NULL buffer");
    SamplePdfArgs a{};
    a.t = t; a.z_vals = z_vals; a.weights = weights; a.u = u; a.z_out = z_out;
    a.n_rays = n_rays; a.n_samples = n_samples; a.n_imp = n_importance;
    a.delta = 1.0f / (float)(n_samples - 1);
    a.delta_u = n_importance > 1 ? 1.0f / (float)(n_importance - 1) : 0.0f;
    a.flags = flags; a.seed_lo = (uint32_t)perturb_seed; a.seed_hi = (uint32_t)(perturb_seed >> 32);
    a.idx0 = im.idx0; a.idx_run = im.run; a.idx_stride = im.stride;
    int64_t nb = (n_rays + 3) / 4;
    if (nb > 256 * 8) nb = 256 * 8;
    sample_pdf_kernel<<<dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream>>>(a);
    HIP_TRY(hipGetLastError());
    return NTX_OK;
}

int ntx_image_epilogue(const float *rgba, int height, int width, int downsampling_factor, int unpremultiply,
                       float *out_f32, uint8_t *out_u8, ntx_stream stream) {
    if (!rgba || (!out_f32 && !out_u8)) return fail(NTX_E_INVALID, "NULL buffer");
    if (height <= 0 || width <= 0) return fail(NTX_E_INVALID, "bad image size %dx%d", height, width);
    const int f = downsampling_factor;
    if (f < 1 || f * 3 > MAX_EPILOGUE_TAPS) return fail(NTX_E_INVALID, "downsampling_factor %d outside [1,%d]", f, MAX_EPILOGUE_TAPS / 3);
    EpilogueArgs a{};
    a.rgba = rgba; a.out_f32 = out_f32; a.out_u8 = out_u8;
    a.h = height; a.w = width; a.factor = f; a.unpremultiply = unpremultiply;
    a.oh = (height + f - 1) / f; a.ow = (width + f - 1) / f;
    if (f > 1) {
        const float stdv = (float)(f * .5);                     // filtered_downsample(std=.5): factor * std
        const int K = (int)(f * .5 * 6);                        // interpolate.py:81
        a.taps = K;
        float sum = 0.0f;
        for (int i = 0; i < K; ++i) {                           // interpolate.py:71-72 (+0.5 shift for even sizes)
            const float x = (float)(-(K - 1) / 2.0 + i) + (K % 2 == 0 ? 0.5f : 0.0f);
            const float q = x / stdv;
            a.k1[i] = expf(-.5f * (q * q));
            sum += a.k1[i];
        }
        for (int i = 0; i < K; ++i) a.k1[i] /= sum;             // (k1 (x) k1) / sum(k1 (x) k1) = (k1/S) (x) (k1/S)
        const int ph = (a.oh - 1) * f + K - height, pw = (a.ow - 1) * f + K - width;   // TF 'SAME'
        a.pad_top = (ph > 0 ? ph : 0) / 2; a.pad_left = (pw > 0 ? pw : 0) / 2;
    }
    const int n = a.oh * a.ow;
    epilogue_kernel<<<dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(a);
    HIP_TRY(hipGetLastError());
    return NTX_OK;
}

}  // extern "C"
