// ntx_layout.h -- operand layout shared by the host weight packer and the gfx950 kernels.
//
// The fused MLP keeps a batch of 32 samples per wave64 entirely in registers and computes every
// Dense layer TRANSPOSED with v_mfma_f32_32x32x2_f32:   out^T[feat, sample] = W^T[feat, k] * h^T[k, sample]
//   A operand (1 VGPR) : lane l holds W[row(step, half=l>>5)][32*mtile + (l&31)]      (weights)
//   B operand (1 VGPR) : lane l holds h[sample l&31][feature row(step, half=l>>5)]   (activations)
//   C/D (16 regs)      : lane l, reg r holds out feature 32*mtile + (r&3) + 8*(r>>2) + 4*(l>>5)
//                        of sample l&31
// Because the order of the k-summation is free, the k-steps of the NEXT layer are defined so that
// step s = 16*mtile + r pairs exactly the two features a lane pair (l, l+32) holds in register r of
// tile mtile: an accumulator register, after bias+ReLU, IS the next layer's B operand -- activations
// never move between lanes, never touch LDS and never leave the register file.
// `hidden_row` is that map; `pos_row` / `dir_row` are the analogous maps for the positional-encoding
// segments, chosen so that one k-step = {sin(2^f x), cos(2^f x)} = one sin() evaluation per lane
// with a quadrant shift in the upper half-wave.
#pragma once

#if defined(__HIPCC__)
#define NTX_HD __host__ __device__
#else
#define NTX_HD
#endif

namespace ntx {

constexpr int POS_FREQ = 10;   // n_freq_bands of pos_embedding in every reference config
constexpr int DIR_FREQ = 4;
constexpr int PAR_FREQ = 4;
constexpr int WIDTH = 256;
constexpr int DEPTH = 8;
constexpr int SKIP = 4;
constexpr int HSTEPS = WIDTH / 2;       // k-steps of a 256-wide hidden input
#ifndef NTX_RING
#define NTX_RING 8
#endif
constexpr int RING = NTX_RING;          // weight records (64 lanes x float4) kept in flight per wave
constexpr int REC_FLOATS = 256;         // one record = 64 lanes x 4 floats = 1 KiB

NTX_HD constexpr int round_up(int a, int b) { return (a + b - 1) / b * b; }

// feature index (within a hidden activation vector) that half-wave `h` holds for k-step `s`
NTX_HD constexpr int hidden_row(int s, int h) {
    return 32 * (s >> 4) + (s & 3) + 8 * ((s & 15) >> 2) + 4 * h;
}

// ---- position segment: pos_map = [FF(pos,10) (63) | FF(params[:n_geo],4) (9*n_geo)]  (model.py:77,88-93)
// ipe = 1: the position embedding is mip-NeRF's IntegratedPositionalEncoding (layer.py:25-41) of a 6-D input
// (mean, diagonal covariance): 6*POS_FREQ features [sin(y) e^(-var/2) (3L) | sin(y + pi/2) e^(-var/2) (3L)] with
// y index f*3+c, NO identity block.
NTX_HD constexpr int pos_emb_dim(int ipe) { return ipe ? 6 * POS_FREQ : 3 * (1 + 2 * POS_FREQ); }
// k-steps of the position segment, GEOMETRY PARAMETERS FIRST (the order of the k-summation is free), one block of
// GEO_BLOCK = 1 + PAR_FREQ steps per parameter, LAST parameter first:
//   block b = parameter p = n_geo - 1 - b:   (g_p, pad), then {sin, cos}(2^f g_p), f = 0 .. PAR_FREQ-1
//   pos identity  2 steps                 (x, y), (z, pad)            [IPE: none]
//   pos sin/cos   3 * POS_FREQ steps      {sin, cos}(2^f x_c)         [IPE: damped by exp(-4^f var_c / 2)]
// The geometry parameters are constant along a ray (renderer.py:154) unless blur_idx scales one per sample (:155-158), so
// their blocks lead the segment: a kernel that evaluates a PREFIX of them once per ray starts its accumulators from bias +
// those blocks and runs the rest per sample -- same summation order, same bits.  render_kernel<CFG, 2> hoists all of them
// (no blur_idx); <CFG, 3> all but the block of parameter 0, which therefore comes last (blur_idx = 0: grass_filtered).
constexpr int GEO_BLOCK = 1 + PAR_FREQ;
NTX_HD constexpr int pos_geo_steps(int n_geo) { return n_geo * GEO_BLOCK; }
NTX_HD constexpr int pos_steps(int n_geo, int ipe = 0) { return pos_geo_steps(n_geo) + (ipe ? 0 : 2) + 3 * POS_FREQ; }
NTX_HD constexpr int pos_map_dim(int n_geo, int ipe = 0) { return pos_emb_dim(ipe) + n_geo * (1 + 2 * PAR_FREQ); }

// row of the reference's pos_map that (step s, half h) carries, or -1 for a zero pad.  n_geo lays out the k-steps; a model
// with FEWER geometry parameters (n_act < n_geo, the generic family) leaves the steps of the missing ones as zero rows.
NTX_HD constexpr int pos_row(int n_geo, int s, int h, int ipe = 0, int n_act = -1) {
    if (n_act < 0) n_act = n_geo;
    const int base = pos_emb_dim(ipe);
    if (s < pos_geo_steps(n_geo)) {
        const int p = n_geo - 1 - s / GEO_BLOCK, j = s % GEO_BLOCK;
        if (p >= n_act) return -1;
        if (j == 0) return h == 0 ? base + p : -1;
        return base + n_act + 2 * (j - 1) * n_act + h * n_act + p;
    }
    int q = s - pos_geo_steps(n_geo);
    if (!ipe) {
        if (q < 2) {
            const int v = 2 * q + h;
            return v < 3 ? v : -1;
        }
        q -= 2;
    }
    if (q < 3 * POS_FREQ) return ipe ? h * 3 * POS_FREQ + q : 3 + 6 * (q / 3) + 3 * h + (q % 3);
    return -1;
}

// ---- direction segment: dir_map = [FF(dir,4) (27) | FF(params[n_geo:],4) (9*n_app)]  (model.py:78,96-101)
NTX_HD constexpr int dir_id_values(int n_app) { return 3 + n_app; }
NTX_HD constexpr int dir_id_steps(int n_app) { return (dir_id_values(n_app) + 1) / 2; }
NTX_HD constexpr int dir_steps_raw(int n_app) { return dir_id_steps(n_app) + 3 * DIR_FREQ + n_app * PAR_FREQ; }
NTX_HD constexpr int dir_steps(int n_app) { return dir_steps_raw(n_app); }
NTX_HD constexpr int dir_map_dim(int n_app) { return 3 * (1 + 2 * DIR_FREQ) + n_app * (1 + 2 * PAR_FREQ); }

NTX_HD constexpr int dir_row(int n_app, int s, int h, int n_act = -1) {   // n_act < n_app: as pos_row
    if (n_act < 0) n_act = n_app;
    const int nid = dir_id_steps(n_app);
    if (s < nid) {
        const int v = 2 * s + h;
        if (v >= dir_id_values(n_app)) return -1;
        if (v < 3) return v;
        return v - 3 < n_act ? 3 * (1 + 2 * DIR_FREQ) + (v - 3) : -1;
    }
    int q = s - nid;
    if (q < 3 * DIR_FREQ) return 3 + 6 * (q / 3) + 3 * h + (q % 3);
    q -= 3 * DIR_FREQ;
    if (q < n_app * PAR_FREQ) {
        const int f = q / n_app, a = q % n_app;
        return a < n_act ? 3 * (1 + 2 * DIR_FREQ) + n_act + 2 * f * n_act + h * n_act + a : -1;
    }
    return -1;
}

// the generic family: any ParamNerf n_parameters = [g, a] with g <= GEN_NGEO, a <= GEN_NAPP runs on the kernels of
// Cfg<GEN_NGEO, GEN_NAPP>; the rows of the parameters it does not have are zero and their inputs are fed as 0
constexpr int GEN_NGEO = 4, GEN_NAPP = 8;
constexpr int MAX_PARAM_SLOTS = 16;

// ---- packed image geometry -----------------------------------------------------------------
// Weight STREAM (consumed strictly in order by every wave, RING records ahead):
//   L0   : pos segment, 8 M-tiles            pos_steps * 2 records
//   L1-4 : hidden segment                    HSTEPS * 2 records each
//   L5   : pos segment then hidden segment
//   L6-7 : hidden
//   F    : hidden (linear "feature" layer, model.py:114)
//   ParamNerf: C1 = dir segment (8 tiles) + hidden ; C2 = hidden input, 4 M-tiles (HSTEPS records)
//   Nerf     : C2 = dir segment (4 tiles) + hidden (4 tiles)
//   pad  : zero records up to a multiple of RING (skipped, never multiplied), so that the compile-time
//          ring phase is 0 again when the next batch starts
//   tail : a copy of the first RING records (so the prefetch ring wraps into the next batch)
// AUX block (read through LDS): biases in accumulator order, alpha head, rgb head.
struct Geometry {
    int n_geo, n_app, color_depth, ipe;   // color_depth: 1 = ParamNerf, 0 = Nerf
    int pos_steps, dir_steps;
    int stream_records;              // without pad and wrap-around tail
    int padded_records;              // rounded up to a multiple of RING
    int aux_floats;
};

constexpr int N_BIAS_LAYERS_MAX = 12;   // L0..L7, F, C1, C2 (+1 spare)
constexpr int AUX_BIAS_STRIDE = 2 * 128;  // [half][128] per layer (C2 uses the first 64 of each half)

NTX_HD constexpr int aux_alpha_off() { return N_BIAS_LAYERS_MAX * AUX_BIAS_STRIDE; }   // [half][128] + bias
NTX_HD constexpr int aux_rgb_off() { return aux_alpha_off() + 2 * 128 + 4; }             // [3][half][64] + bias[3]
NTX_HD constexpr int aux_total() { return round_up(aux_rgb_off() + 3 * 2 * 64 + 4, 64); }

NTX_HD constexpr Geometry make_geometry(int n_geo, int n_app, int color_depth, int ipe = 0) {
    Geometry g{};
    g.n_geo = n_geo; g.n_app = n_app; g.color_depth = color_depth; g.ipe = ipe;
    g.pos_steps = pos_steps(n_geo, ipe);
    g.dir_steps = dir_steps(n_app);
    int rec = 0;
    rec += g.pos_steps * 2;                 // L0
    rec += 4 * HSTEPS * 2;                  // L1-4
    rec += g.pos_steps * 2 + HSTEPS * 2;    // L5
    rec += 2 * HSTEPS * 2;                  // L6-7
    rec += HSTEPS * 2;                      // F
    if (color_depth) {
        rec += g.dir_steps * 2 + HSTEPS * 2;  // C1
        rec += HSTEPS;                         // C2
    } else {
        rec += g.dir_steps + HSTEPS;           // C2 (4 tiles)
    }
    g.stream_records = rec;
    g.padded_records = round_up(rec, RING);
    g.aux_floats = aux_total();
    return g;
}

// ---- flex family: the ARCHITECTURE is a run-time fact ------------------------------------------------------------------
// model.py:58 (ParamNerf) / :9 (Nerf): depth, width, skips, color_depth.  One kernel set (Cfg<GEN_NGEO, GEN_NAPP, 1, 0, 1, 1>, the
// parameter slots of the generic family) runs a LOOP over 256-wide layers whose bodies are the straight-line segments of the tuned
// kernels; what varies per model is read from a descriptor in the aux block:
//   trunk layer 0            position segment
//   trunk layers 1 .. D-1    [position segment, when the previous layer's index is in `skips`] + hidden segment
//   F (linear)               hidden segment; the alpha head rides on its input relu(trunk D-1)
//   colour layers 1 .. CD    the first one [direction segment] + hidden segment on the LINEAR F, the others hidden segments
//   colour half (4 tiles)    hidden segment; with CD = 0 (also plain Nerf): [direction segment] + hidden segment on the linear F
// width < 256: the missing rows / columns / biases are zero (relu(0) = 0 feeds zero rows of the next layer: exact).
// Every segment of the flex STREAM is padded with zero records to a multiple of RING, so each body starts at ring phase 0 whatever
// came before it; the record offset is a run-time scalar that advances body by body.  The stream ends with the same wrap-around
// tail (the first RING records again).  Bias slots (aux block, behind the tuned aux layout whose alpha / rgb heads it keeps):
// slot l = layer l in the order above.
constexpr int FLEX_MAX_DEPTH = 24;      // trunk layers
constexpr int FLEX_MAX_COLOR = 4;       // color_depth
constexpr int FLEX_MAX_PARAM_DEPTH = 4; // param_depth (the flavour with parameter branches, Cfg FLEX = 2)
constexpr int FLEX_MAX_LAYERS = 40;     // bias slots: D trunk + F + CD colour + colour half <= 24 + 1 + 4 + 1, then 4 + 4 branch layers
constexpr int FLEX_DESC_FLOATS = 64;    // int32 words at the head of the flex block: [0] depth, [1] skip mask, [2] color_depth,
                                        // [3] param_depth, [4] geometry branch present, [5] appearance branch present
constexpr int NTX_SKIP_MASK_BIT = 0x40000000;   // ntx_model_desc.skip = NTX_SKIP_MASK_BIT | mask of the indices in `skips`

NTX_HD constexpr int flex_floats() { return FLEX_DESC_FLOATS + FLEX_MAX_LAYERS * AUX_BIAS_STRIDE; }
NTX_HD constexpr int flex_seg_records(int steps, int nmt) { return round_up(steps * (nmt / 4), RING); }

struct FlexArch {
    int depth, width, color_depth;   // color_depth of the MODEL (0 for plain Nerf)
    unsigned skip_mask;              // bit i: trunk layer i + 1 takes concat[pos_map, h]  (model.py:107-108), i < depth - 1
    // param_depth > 0 (model.py:88-101; Cfg FLEX = 2): Dense(param_width <= 128, relu) layers on FF(params[:g]) / FF(params[g:]) before
    // they are concatenated to FF(pos) / FF(dir); a branch exists when the model has parameters of its kind
    int param_depth, param_width, has_geo, has_app;
};
constexpr int BRANCH_K = 64;         // k-steps of a branch activation (128 wide)
NTX_HD constexpr int parff_steps(int n_slots) { return n_slots * GEO_BLOCK; }
// row of FF(params[0..n_act)) (layer.py:8-23: [x | sin(2^0 x) | cos(2^0 x) | ...], every block n_act wide) that (step s, half h) of a
// branch's first segment carries: one block of GEO_BLOCK steps per parameter SLOT, last slot first, as the position segment's
NTX_HD constexpr int parff_row(int n_slots, int s, int h, int n_act) {
    const int p = n_slots - 1 - s / GEO_BLOCK, j = s % GEO_BLOCK;
    if (p >= n_act) return -1;
    if (j == 0) return h == 0 ? p : -1;
    return n_act + 2 * (j - 1) * n_act + h * n_act + p;
}
// records of the flex stream without the wrap-around tail (a multiple of RING by construction)
NTX_HD constexpr int flex_stream_records(const FlexArch &f) {
    const bool pb = f.param_depth > 0;
    const int psteps = pb ? pos_steps(0) : pos_steps(GEN_NGEO), dsteps = pb ? dir_steps(0) : dir_steps(GEN_NAPP);
    const int g8 = pb && f.has_geo ? flex_seg_records(BRANCH_K, 8) : 0;
    const int a8 = pb && f.has_app ? flex_seg_records(BRANCH_K, 8) : 0, a4 = pb && f.has_app ? flex_seg_records(BRANCH_K, 4) : 0;
    const int ps8 = flex_seg_records(psteps, 8) + g8, ds8 = flex_seg_records(dsteps, 8) + a8, ds4 = flex_seg_records(dsteps, 4) + a4,
              h8 = flex_seg_records(HSTEPS, 8), h4 = flex_seg_records(HSTEPS, 4), b4 = flex_seg_records(BRANCH_K, 4);
    int rec = ps8;
    for (int i = 1; i < f.depth; ++i) rec += (((f.skip_mask >> (i - 1)) & 1u) ? ps8 : 0) + h8;
    rec += h8;                                             // F
    if (f.color_depth > 0) rec += ds8 + h8 + (f.color_depth - 1) * h8 + h4;
    else rec += ds4 + h4;
    if (pb && f.has_geo) rec += flex_seg_records(parff_steps(GEN_NGEO), 4) + (f.param_depth - 1) * b4;
    if (pb && f.has_app) rec += flex_seg_records(parff_steps(GEN_NAPP), 4) + (f.param_depth - 1) * b4;
    return rec;
}

// ---- fp16x3 precision (ntx_device_x3.h): the same network on v_mfma_f32_32x32x16_f16 -----------------
// One k16-step = 8 consecutive k2-steps of the maps above (element e of half h in step u = k2-step 8u+e), segments
// padded with zero rows to whole k16-steps.  One record = 64 lanes x 8 halves (1 KiB) = the A operand of one
// (k16-step, M-tile); the stream holds, per k16-step and tile, the hi record then the lo record of the split
// w = hi + lo (hi = fp16_rne(w), lo = fp16_rne(w - hi), IEEE half with subnormals).  Within a pass the hidden segment comes first, the encoder
// segment second; the direction segment of ParamNerf's colour layer C1 is NOT in the stream: it is a per-ray constant
// and enters through the per-ray start vector of dir_block (float32, ntx_device.h).  The 4 waves of a workgroup share the stream through an LDS ring of NSTAGE16 STAGES of STAGE16
// records (one k16-step of an 8-tile layer); the stream is zero-padded to a whole number of ring turns, so the stage ->
// ring-slot map is the same for every batch and the prefetch simply wraps to stage 0.  The aux block is the f32 one.
constexpr int STAGE16 = 16;    // records per stage
constexpr int NSTAGE16 = 4;    // stages in the LDS ring (64 KiB)
NTX_HD constexpr int steps16(int k2_steps) { return (k2_steps + 7) / 8; }
// with_dir: the stream of the INSTANCED kernel, whose directions are per sample (renderer.py:247-262): C1 keeps its
// direction segment (after its hidden segment)
NTX_HD constexpr int stream16_records(int n_geo, int n_app, int color_depth, int with_dir = 0, int ipe = 0) {
    const int ps = steps16(pos_steps(n_geo, ipe)), ds = steps16(dir_steps(n_app)), hs = HSTEPS / 8;
    int rec = ps * 16 + 4 * hs * 16 + (ps + hs) * 16 + 2 * hs * 16 + hs * 16;
    if (color_depth) rec += hs * 16 + (with_dir ? ds * 16 : 0) + hs * 8;   // ParamNerf: C1 (direction segment hoisted per ray unless with_dir), C2
    else rec += (ds + hs) * 8;                  // plain Nerf: C2 = hidden + direction segment, 4 tiles
    return rec;
}
NTX_HD constexpr int stream16_padded(int n_geo, int n_app, int color_depth, int with_dir = 0, int ipe = 0) {
    return round_up(stream16_records(n_geo, n_app, color_depth, with_dir, ipe), STAGE16 * NSTAGE16);
}

}  // namespace ntx
