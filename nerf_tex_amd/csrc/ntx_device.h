// ntx_device.h -- gfx950 device code of the NeRF-Tex render path (included by nerftex.hip only).
//
// Kernels (all float32, one wave64 = one batch of 32 samples, activations in registers):
//   render_kernel<CFG,HOIST> rays -> premultiplied RGBA  (renderer.py:47-213 fused)
//   mlp_kernel<CFG>         (pos, dir, params) -> (raw rgb, raw sigma)   (model.py:58-125)
//   composite_kernel        map_model_output alone        (renderer.py:170-213)
//   raygen_kernel           rays_from_camera + Proxy/AABB (ray_sampler.py:23-48, proxy.py:13-35)
//   fourier_kernel          FourierFeatures alone         (layer.py:8-23)
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#include <utility>

#include "nerftex.h"   // NTX_FLAG_*
#include "ntx_layout.h"

// Measured on MI355X (carpet 800x800x64, ms per launch) while deciding where VALU work goes:
//   layer epilogue and encoder as BLOCKS of VALU work                    377.4
//   encoder sin() spread over the MFMA slots of the previous k-step       383.3
//   layer epilogue spread over the next layer's k-steps                   388.6
//   both spread                                                           397.5
// A wave's VALU instructions do not hide under v_mfma_f32_32x32x2_f32 (nor do a second wave's:
// tools/ubench/mfma_valu_overlap.hip) -- the f32 MFMA runs on the FP32 lanes -- so VALU work is kept in
// few, dense, well-pipelined blocks.

// The kernel objects do not depend on include/nerftex.h in the Makefile (a comment there must not cost an hour of
// hipcc); the flag values they compile in are frozen instead -- appended to, never renumbered:
static_assert(NTX_FLAG_MAP_EXR == 1u && NTX_FLAG_COMPOSITE_BKGD == 2u && NTX_FLAG_CHECK_NUMERICS == 4u && NTX_FLAG_FP16X3 == 8u &&
              NTX_FLAG_PERTURB == 16u && NTX_FLAG_RAW_NOISE == 32u, "NTX_FLAG_* values are part of the built kernels");

namespace ntx {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define NTX_DEV __device__ __forceinline__

template <class F, int... I>
NTX_DEV void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
// compile-time unrolled loop: f receives std::integral_constant<int, i>, so every register-array
// index inside is a constant expression (runtime-indexed register arrays would go to scratch)
template <int N, class F>
NTX_DEV void static_for(F &&f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// ---------------------------------------------------------------------------------------------
// math
// ---------------------------------------------------------------------------------------------
// sin(x + q*pi/2): 3-term Cody-Waite reduction by pi/2 with FMA (exact first step for |x| < ~2^11,
// <= 1.5 ulp abs error measured out to |x| = 2^17), cephes minimax polynomials on [-pi/4, pi/4].
// q = 0 gives sin, q = 1 gives cos: one evaluation serves a {sin, cos} k-step pair.
NTX_DEV float sin_q(float x, int q) {
    const float n = __builtin_rintf(x * 0x1.45f306p-1f);
    float r = __builtin_fmaf(-n, 0x1.921fb6p+0f, x);
    r = __builtin_fmaf(-n, -0x1.777a5cp-25f, r);
    r = __builtin_fmaf(-n, -0x1.ee59dap-50f, r);
    const int qq = (int)n + q;
    const float r2 = r * r;
    float ps = __builtin_fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = __builtin_fmaf(r2, ps, -1.6666654611e-1f);
    const float s = __builtin_fmaf(r * r2, ps, r);
    float pc = __builtin_fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = __builtin_fmaf(r2, pc, 4.166664568298827e-2f);
    const float c = __builtin_fmaf(r2 * r2, pc, __builtin_fmaf(r2, -0.5f, 1.0f));
    float v = (qq & 1) ? c : s;
    return (qq & 2) ? -v : v;
}

NTX_DEV float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
NTX_DEV float elu1f_(float x) { return (x > 0.0f ? x : expf(x) - 1.0f) + 1.0f; }

// ---------------------------------------------------------------------------------------------
// weight stream: every wave reads the packed stream strictly in order, RING records ahead.
// One record = 64 lanes x float4 (1 KiB); a k-step of an NMT-tile layer consumes NMT/4 records.
// ---------------------------------------------------------------------------------------------
// Addressing: one buffer descriptor (SGPRs) over the whole stream, a wave-uniform byte offset in
// an SGPR that advances record by record, and a fixed per-lane offset lane*16 in one VGPR -- no
// per-load 64-bit address arithmetic on the VALU.
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct WStream {
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t voff;   // lane * 16
    f32x4 ring[RING];   // record n of the stream lives in slot n % RING (all indices are compile-time)
};

// the whole network is straight-line code, so every record index is a constant: the scalar offset of a
// load is an immediate / s_mov, not arithmetic
NTX_DEV f32x4 ws_load(const WStream &ws, uint32_t rec) {
    const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ws.rsrc, ws.voff, rec * 1024u, 0);
    return __builtin_bit_cast(f32x4, v);
}

// M = the record map of the kernel flavour (RecMap<CFG, HOIST> below): every wave consumes a LOGICAL stream, the packed
// (physical) stream with the segments its flavour evaluates per ray left out; M::phys(logical) is a compile-time constant,
// so skipping a segment costs nothing at run time -- fetching the skipped records through the ring to keep its phase, as
// the first hoisting kernels did, put a burst of up to 82 loads and an exposed L2 latency in front of the next MFMA.
template <class M>
NTX_DEV void ws_prime(WStream &ws, const f32x4 *base, uint32_t stream_bytes, int lane) {
    ws.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<f32x4 *>(base), 0, stream_bytes, 0x00020000);
    ws.voff = (uint32_t)lane * 16u;
    static_for<RING>([&](auto I) { ws.ring[I] = ws_load(ws, M::phys(I)); });
}

// logical records [REC0, REC0+N) are padding: keep the ring turning without multiplying them
template <class M, int N, int REC0>
NTX_DEV void skip_records(WStream &ws) {
    static_for<N>([&](auto I) {
        constexpr int rec = REC0 + decltype(I)::value;
        ws.ring[rec % RING] = ws_load(ws, M::phys(rec + RING));
    });
}

NTX_DEV f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------
// B-operand generators.  A generator hands run_segment this lane's B value for k-step S.  Encoder
// generators evaluate PE_GROUP consecutive k-steps at once: the f32 MFMA and the VALU share the FP32 lanes
// on gfx950 (VALU time simply adds to MFMA time, measured in tools/ubench/mfma_valu_overlap.hip), so the
// only thing to gain is VALU issue efficiency -- PE_GROUP independent sin() chains interleave instead of
// one 24-deep dependent chain per step.
// ---------------------------------------------------------------------------------------------
constexpr int PE_GROUP = 4;

// activations of the previous layer, already in registers
struct HiddenGen {
    const float (&hin)[128];
    template <int S, int N>
    NTX_DEV void prepare() {}
    template <int S>
    NTX_DEV float value() const { return hin[S]; }
};

// accumulator tile <- bias, straight from the LDS copy of the aux block ([half][128] per layer).
// A tile is always (re)defined as ONE 16-wide value: defining it a few registers at a time is a partial
// write of a register tuple and makes the allocator copy whole tiles around.
template <int MT>
NTX_DEV void init_bias_tile(f32x16 (&acc)[8], const float *aux, int layer, int h) {
    const f32x4 *b = reinterpret_cast<const f32x4 *>(aux + layer * AUX_BIAS_STRIDE + h * 128) + MT * 4;
    const f32x4 v0 = b[0], v1 = b[1], v2 = b[2], v3 = b[3];
    acc[MT] = f32x16{v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
}
// same from a 256-float row in LDS ([half][128], accumulator order): the colour layer's bias, which for hoisted direction
// features is a per-ray vector (dir_block)
template <int MT>
NTX_DEV void init_bias_tile_row(f32x16 (&acc)[8], const float *row, int h) {
    const f32x4 *b = reinterpret_cast<const f32x4 *>(row + h * 128) + MT * 4;
    const f32x4 v0 = b[0], v1 = b[1], v2 = b[2], v3 = b[3];
    acc[MT] = f32x16{v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
}
template <int NMT>
NTX_DEV void init_bias(f32x16 (&acc)[8], const float *aux, int layer, int h) {
    static_for<NMT>([&](auto MT) { init_bias_tile<decltype(MT)::value>(acc, aux, layer, h); });
}

// hin[V0..V0+8) <- relu(previous layer's accumulators): 8 v_accvgpr_read into 8 different registers, then 8
// v_max_f32 in place.  (fmaxf() on an MFMA result makes hipcc emit a canonicalising v_max x,x,x first, and a
// per-value read/max pair re-uses one temporary 128 times: a serial chain.)  Register r of tile V/16 is
// feature hidden_row(V, half).
template <int V0, bool RELU>
NTX_DEV void convert8(float (&hin)[128], const f32x16 (&prev)[8]) {
    constexpr int T = V0 >> 4, R = V0 & 15;
    static_assert(R % 8 == 0, "blocks of 8 within a tile");
    if constexpr (RELU) {
        asm("v_accvgpr_read_b32 %0, %8\n\tv_accvgpr_read_b32 %1, %9\n\tv_accvgpr_read_b32 %2, %10\n\t"
            "v_accvgpr_read_b32 %3, %11\n\tv_accvgpr_read_b32 %4, %12\n\tv_accvgpr_read_b32 %5, %13\n\t"
            "v_accvgpr_read_b32 %6, %14\n\tv_accvgpr_read_b32 %7, %15\n\t"
            "v_max_f32 %0, 0, %0\n\tv_max_f32 %1, 0, %1\n\tv_max_f32 %2, 0, %2\n\tv_max_f32 %3, 0, %3\n\t"
            "v_max_f32 %4, 0, %4\n\tv_max_f32 %5, 0, %5\n\tv_max_f32 %6, 0, %6\n\tv_max_f32 %7, 0, %7"
            : "=&v"(hin[V0 + 0]), "=&v"(hin[V0 + 1]), "=&v"(hin[V0 + 2]), "=&v"(hin[V0 + 3]), "=&v"(hin[V0 + 4]),
              "=&v"(hin[V0 + 5]), "=&v"(hin[V0 + 6]), "=&v"(hin[V0 + 7])
            : "a"(prev[T][R + 0]), "a"(prev[T][R + 1]), "a"(prev[T][R + 2]), "a"(prev[T][R + 3]), "a"(prev[T][R + 4]),
              "a"(prev[T][R + 5]), "a"(prev[T][R + 6]), "a"(prev[T][R + 7]));
    } else {
        static_for<8>([&](auto K) { hin[V0 + decltype(K)::value] = prev[T][R + decltype(K)::value]; });
    }
}

template <int NMT, bool RELU>
NTX_DEV void store_act(float (&hin)[128], const f32x16 (&acc)[8]) {
    static_for<NMT * 2>([&](auto V) { convert8<decltype(V)::value * 8, RELU>(hin, acc); });
}

// per-ray rows (dir_block, below) live in LDS, 32 ray slots of a workgroup at a time
#ifndef NTX_DIR_BLOCK_ITERS
#define NTX_DIR_BLOCK_ITERS 8
#endif
constexpr int DIR_BLOCK_ITERS = NTX_DIR_BLOCK_ITERS;   // rays per wave and block
constexpr int DIR_BLOCK_RAYS = 4 * DIR_BLOCK_ITERS;   // a multiple of 32 = one MFMA's worth of B columns
constexpr int DIR_ROW_STRIDE = 256 + 4;            // floats; +4: the 32 lanes' b128 stores spread over the LDS banks
constexpr int DIR_BLOCK_FLOATS = DIR_BLOCK_RAYS * DIR_ROW_STRIDE;

// ---------------------------------------------------------------------------------------------
// one segment: acc[mt] += W_segment^T * B over NSTEPS k-steps, one MFMA per (step, tile) SLOT.
//   gen   : B-operand generator; stages of step S+1 run in the slots of step S
//   extra : extra(S, MT) = additional VALU/LDS work to place in the shadow of slot (S, MT)
// A sched_barrier after every slot pins the order hipcc emits: without it the scheduler sinks the
// prefetch loads to their first use (collapsing the ring) and bunches the VALU work in front of the MFMAs.
// REC0 = LOGICAL index of the segment's first record (M::log_of(physical index)).
// ---------------------------------------------------------------------------------------------
template <class M, int NSTEPS, int NMT, int REC0, class Gen, class Extra>
NTX_DEV void run_segment(f32x16 (&acc)[8], WStream &ws, Gen &gen, Extra &&extra) {
    constexpr int RPS = NMT / 4;
    gen.template prepare<0, (NSTEPS < PE_GROUP ? NSTEPS : PE_GROUP)>();   // first group: before the first MFMA
    float b = gen.template value<0>();
    static_for<NSTEPS>([&](auto S) {
        constexpr int s = S;
        f32x4 w;
        static_for<NMT>([&](auto MT) {
            constexpr int mt = MT;
            if constexpr (mt % 4 == 0) {
                constexpr int rec = REC0 + s * RPS + mt / 4;
                w = ws.ring[rec % RING];
                ws.ring[rec % RING] = ws_load(ws, M::phys(rec + RING));
            }
            acc[mt] = mfma32(w[mt % 4], b, acc[mt]);
            // the B values of the next PE_GROUP k-steps, as ONE block of VALU work after this step's first MFMA
            if constexpr (mt == 0 && (s + 1) % PE_GROUP == 0 && s + 1 < NSTEPS)
                gen.template prepare<s + 1, (NSTEPS - s - 1 < PE_GROUP ? NSTEPS - s - 1 : PE_GROUP)>();
            extra(S, MT);
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (s + 1 < NSTEPS) b = gen.template value<s + 1>();
    });
}

// ---------------------------------------------------------------------------------------------
// per-lane inputs of one sample and the positional-encoding generators (layer.py:8-23)
// ---------------------------------------------------------------------------------------------
template <int NGEO, int NAPP>
struct SampleIn {
    float pos[3];
    float cov[3];   // diagonal covariance of the cone-segment gaussian (IPE models only; dead otherwise)
    float dir[3];
    float par[NGEO + NAPP > 0 ? NGEO + NAPP : 1];
};

// Copy of the inputs whose values the optimiser cannot relate to the original: without it, CSE keeps
// the encoded position features alive from layer 0 to the skip layer (~40 VGPRs) instead of
// recomputing them in the shadow of the MFMAs.
template <int NGEO, int NAPP>
NTX_DEV SampleIn<NGEO, NAPP> launder(const SampleIn<NGEO, NAPP> &in) {
    SampleIn<NGEO, NAPP> o = in;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        asm volatile("" : "+v"(o.pos[k]));
        asm volatile("" : "+v"(o.cov[k]));
        asm volatile("" : "+v"(o.dir[k]));
    }
#pragma unroll
    for (int k = 0; k < NGEO + NAPP; ++k) asm volatile("" : "+v"(o.par[k]));
    return o;
}

// identity value v of the direction segment: [dir(3) | appearance params]
template <int NGEO, int NAPP, int V>
NTX_DEV float dir_id_value(const SampleIn<NGEO, NAPP> &in) {
    if constexpr (V < 3) return in.dir[V];
    else if constexpr (V - 3 < NAPP) return in.par[NGEO + V - 3];
    else return 0.0f;
}

// k-step S of the position segment (ntx_layout.h: one block per geometry parameter, last parameter first, then the position block).
// IPE (layer.py:25-41): no identity for the position, and every position pair is damped by exp(-0.5 * 4^f * cov_c).
template <int NGEO, int NAPP, int IPE, int S>
NTX_DEV float pos_feature(const SampleIn<NGEO, NAPP> &in, int h) {
    constexpr int ngid = 0, ngs = pos_geo_steps(NGEO), npid = IPE ? 0 : 2;
    if constexpr (S < ngs) {                                   // block of parameter p: (g_p, pad), {sin, cos}(2^f g_p)
        constexpr int p = NGEO - 1 - S / GEO_BLOCK, j = S % GEO_BLOCK;
        if constexpr (j == 0) return h ? 0.0f : in.par[p];
        else return sin_q(in.par[p] * (float)(1 << (j - 1)), h);
    } else if constexpr (S - ngid - ngs < npid) {
        constexpr int q = S - ngid - ngs;
        float lo = in.pos[2 * q < 3 ? 2 * q : 0], hi = 0.0f;
        if constexpr (2 * q + 1 < 3) hi = in.pos[2 * q + 1];
        return h ? hi : lo;
    } else if constexpr (S - ngid - ngs - npid < 3 * POS_FREQ) {
        constexpr int q = S - ngid - ngs - npid, f = q / 3, c = q % 3;
        const float v = sin_q(in.pos[c] * (float)(1 << f), h);
        if constexpr (IPE) return v * expf(-0.5f * (in.cov[c] * (float)(1 << (2 * f))));
        else return v;
    } else {
        return 0.0f;
    }
}

template <int NGEO, int NAPP, int S>
NTX_DEV float dir_feature(const SampleIn<NGEO, NAPP> &in, int h) {
    constexpr int nid = dir_id_steps(NAPP);
    if constexpr (S < nid) {
        const float lo = dir_id_value<NGEO, NAPP, 2 * S>(in), hi = dir_id_value<NGEO, NAPP, 2 * S + 1>(in);
        return h ? hi : lo;
    } else if constexpr (S - nid < 3 * DIR_FREQ) {
        constexpr int q = S - nid, f = q / 3, c = q % 3;
        return sin_q(in.dir[c] * (float)(1 << f), h);
    } else if constexpr (S - nid - 3 * DIR_FREQ < NAPP * PAR_FREQ) {
        constexpr int q = S - nid - 3 * DIR_FREQ, f = q / (NAPP > 0 ? NAPP : 1), a = q % (NAPP > 0 ? NAPP : 1);
        return sin_q(in.par[NGEO + a] * (float)(1 << f), h);
    } else {
        return 0.0f;
    }
}

// KEEP = 1: evaluate and keep every value in this lane's LDS column pe[step * 64]; KEEP = 2: read them back instead of
// evaluating again (the skip layer re-concatenates the same pos_map, model.py:107-108: one sin() per k-step saved, and
// on gfx950 VALU time is not hidden behind the f32 MFMAs); KEEP = 0: evaluate, no LDS.
// OFF: the segment starts at k-step OFF (the geometry-parameter block before it is evaluated per ray, HOIST = 2).
template <int NGEO, int NAPP, int IPE, int KEEP = 0, int OFF = 0>
struct PosGen {
    const SampleIn<NGEO, NAPP> &in;
    int h;
    float vals[PE_GROUP];
    float *pe;
    template <int S, int N>
    NTX_DEV void prepare() {   // B values of k-steps S .. S+N-1 (S is a multiple of PE_GROUP)
        static_for<N>([&](auto K) {
            constexpr int s = S + decltype(K)::value;
            if constexpr (KEEP == 2) {
                vals[s % PE_GROUP] = pe[(s + OFF) * 64];
            } else {
                vals[s % PE_GROUP] = pos_feature<NGEO, NAPP, IPE, s + OFF>(in, h);
                if constexpr (KEEP == 1) pe[(s + OFF) * 64] = vals[s % PE_GROUP];
            }
        });
    }
    template <int S>
    NTX_DEV float value() const { return vals[S % PE_GROUP]; }
};

template <int NGEO, int NAPP>
struct DirGen {
    const SampleIn<NGEO, NAPP> &in;
    int h;
    float vals[PE_GROUP];
    template <int S, int N>
    NTX_DEV void prepare() {
        static_for<N>([&](auto K) { vals[(S + decltype(K)::value) % PE_GROUP] = dir_feature<NGEO, NAPP, S + decltype(K)::value>(in, h); });
    }
    template <int S>
    NTX_DEV float value() const { return vals[S % PE_GROUP]; }
};

// ---------------------------------------------------------------------------------------------
// the MLP on one batch of 32 samples (model.py:58-125 / 9-45); lanes l and l+32 hold sample l&31
// ---------------------------------------------------------------------------------------------
// GEN: the generic family (ntx_layout.h): the model may have fewer parameters than the NGEO_ + NAPP_ slots of the kernel;
// slot k reads column pmap[k] of the caller's parameter rows (or 0), rows are np_in wide (both in the kernel arguments)
// FLEX: the flex family (ntx_layout.h): depth, skips and color_depth are run-time facts read from the aux block; the compile-time
// geometry below (records of the 8 x 256 network) then only serves the code that is shared with the tuned families
template <int NGEO_, int NAPP_, int CD_, int IPE_ = 0, int GEN_ = 0, int FLEX_ = 0>
struct Cfg {
    static constexpr int NGEO = NGEO_, NAPP = NAPP_, CD = CD_, IPE = IPE_, GEN = GEN_, FLEX = FLEX_;
    static constexpr int NP = NGEO_ + NAPP_;          // parameters the MODEL sees
    static constexpr int NP_IN = NP + IPE_;           // parameters per row at the ABI: mip renderers splice the blur
                                                      // parameter out before the model (renderer.py:385-386, 511-512)
    static constexpr int PS = pos_steps(NGEO_, IPE_);
    static constexpr int DS = dir_steps(NAPP_);
    // first record of hidden pass li (1..8 = L1..L7, F; 9 = C1) and of the colour-half layer
    static constexpr int rec_pass(int li) {
        return PS * 2 + (li - 1) * HSTEPS * 2 + (li > SKIP + 1 ? PS * 2 : 0) + (CD_ && li > 9 ? DS * 2 : 0);
    }
    static constexpr int REC_C2 = rec_pass(9 + (CD_ ? 1 : 0));
    static constexpr int REC_END = make_geometry(NGEO_, NAPP_, CD_, IPE_).stream_records;
    static constexpr int REC_PAD = make_geometry(NGEO_, NAPP_, CD_, IPE_).padded_records;
};

// Logical <-> physical record indices of one kernel flavour.  Removed from the logical stream: with HOIST >= 1 the direction
// segment of C1 (ParamNerf), with HOIST = 2 also the geometry-parameter blocks that lead the position segments of L0 and L5.
// The logical stream is padded to whole ring turns and wraps into itself (no use of the packed stream's tail copy).
// leading k-steps of the position segments that a HOIST level evaluates once per ray: all geometry blocks (2), or all but
// the last one, parameter 0's, which blur_idx = 0 scales per sample (3)
template <class CFG, int HOIST>
constexpr int hoisted_geo_steps() {
    return HOIST == 2 ? pos_geo_steps(CFG::NGEO) : HOIST == 3 ? pos_geo_steps(CFG::NGEO - 1) : 0;
}

template <class CFG, int HOIST>
struct RecMap {
    static constexpr int GS2 = hoisted_geo_steps<CFG, HOIST>() * 2;
    static constexpr int DS2 = (HOIST != 0 && CFG::CD != 0 ? CFG::DS : 0) * 2;
    static constexpr int A5 = CFG::rec_pass(SKIP + 1), A9 = CFG::rec_pass(9);   // physical starts of L5's and C1's leading segments
    static constexpr int LOG_END = CFG::REC_END - 2 * GS2 - DS2;
    static constexpr int LOG_PAD = round_up(LOG_END, RING);
    static constexpr int phys(int l) {
        l %= LOG_PAD;
        if (l >= LOG_END) return 0;          // padding of the last ring turn: fetched, never multiplied
        int p = l + GS2;
        if (p >= A5) p += GS2;
        if (p >= A9) p += DS2;
        return p;
    }
    static constexpr int log_of(int p) {     // p: a physical record outside the removed ranges
        return p - (p >= GS2 ? GS2 : 0) - (p >= A5 + GS2 ? GS2 : 0) - (p >= A9 + DS2 ? DS2 : 0);
    }
};

// Two accumulator sets (2 x 128 AGPRs) alternate between layers: layer n's result is moved out of one set
// (bias already in, ReLU, into `hin`) in one dense block before layer n+1 starts accumulating into the
// other, and the drained set is re-initialised tile by tile with the bias of layer n+2 while layer n+1 runs.
// HOIST (render kernel, ParamNerf): the accumulators of the colour layer C1 start from `c1_row` in LDS
// ([half][128], accumulator order) = the per-ray vector bias_C1 + W_dir^T dir_map that dir_block computed with the same
// instructions, and the direction segment is skipped (its records are still fetched, to keep the ring phase).  A
// compile-time variant, not a run-time branch: a branch around the segment made hipcc spill 1 KiB per lane.
// HOIST = 2 (no blur_idx) / 3 (blur_idx = 0): the geometry-parameter blocks of the position segments of L0 and L5 (all / all
// but parameter 0's) are per-ray constant too; L0 and L5 start from the rows dir_block left behind the C1 row (c1_row +
// DIR_BLOCK_FLOATS, + 2 DIR_BLOCK_FLOATS) and run only the rest.
// HOIST = 4 (instance kernel): as 1, but every LANE has its own row -- c1_row is the wave's block of rows (leader_rows) and
// lane_slots[sample] the row of each sample of the batch, read from LDS where the row is needed (nothing more lives across
// the network).
// KEEP_PE = false: no LDS column for the position features, the skip layer evaluates them again (the instance kernel, whose LDS
// holds 32 rows per wave instead).
// MID (instance kernel): called once behind the position segment of the skip layer, the last use of the sample's inputs -- where
// the kernel issues the gathers of the NEXT batch's inputs into the registers those leave (the loads return under the rest of
// the network; the compiler's vmcnt bookkeeping counts them into the ring's waits).
struct NoMid { NTX_DEV void operator()() const {} };
template <class CFG, int HOIST = 0, bool KEEP_PE = true, class Mid = NoMid>
NTX_DEV void mlp_batch_tuned(const SampleIn<CFG::NGEO, CFG::NAPP> &in, WStream &ws,
                             const float *aux_in, int lane, float &sigma, float (&rgb)[3],
                             const float *c1_row = nullptr, float *pe = nullptr, const uint8_t *lane_slots = nullptr, Mid &&mid = Mid{}) {
    constexpr int NGEO = CFG::NGEO, NAPP = CFG::NAPP;
    constexpr bool GEO_ROWS = HOIST == 2 || HOIST == 3;
    constexpr int GS = hoisted_geo_steps<CFG, HOIST>();        // k-steps of the position segments evaluated per ray
    using M = RecMap<CFG, HOIST>;
    const int h = lane >> 5;
    // The aux block in LDS never changes, so the optimiser would hoist every bias / head-weight
    // read out of the batch loop and then spill ~600 values to scratch.  An opaque OFFSET (not an
    // opaque pointer, which would lose the LDS address space) keeps each ds_read next to its use.
    uint32_t opaque_zero = 0;
    asm volatile("" : "+v"(opaque_zero));
    const float *aux = aux_in + opaque_zero;

    f32x16 accA[8], accB[8];
    float hin[128];
    auto none = [](auto, auto) {};
    // v_max_f32(0, NaN) is 0, so the first ReLU would swallow a NaN/Inf input that TensorFlow's relu propagates
    // (and tf.debugging.check_numerics then reports, renderer.py:140-141): chk - chk is 0 for finite inputs, NaN else
    auto input_check = [&]() {
        float chk = in.pos[0] + in.pos[1] + in.pos[2] + in.dir[0] + in.dir[1] + in.dir[2];
        if constexpr (CFG::IPE != 0) chk += in.cov[0] + in.cov[1] + in.cov[2];
#pragma unroll
        for (int k = 0; k < CFG::NP; ++k) chk += in.par[k];
        return chk - chk;
    };
    float chk_early = 0.0f;
    if constexpr (HOIST == 4) chk_early = input_check();   // (instance kernel: one register across the network instead of the inputs)

    // ---- trunk layer 0: pos_map -> 256 (model.py:104-106) into set A; set B <- bias of layer 1
    if constexpr (GEO_ROWS) {
        static_for<8>([&](auto T) { init_bias_tile_row<decltype(T)::value>(accA, c1_row + DIR_BLOCK_FLOATS + opaque_zero, h); });
    } else {
        init_bias<8>(accA, aux, 0, h);
    }
    {
        PosGen<NGEO, NAPP, CFG::IPE, KEEP_PE ? 1 : 0, GS> gen{in, h, {}, pe};
        static_assert(CFG::PS - GS >= 29, "layer-1 bias initialised behind layer 0");
        run_segment<M, CFG::PS - GS, 8, M::log_of(GS * 2)>(accA, ws, gen, [&](auto S, auto MT) {
            constexpr int s = decltype(S)::value, mt = decltype(MT)::value;
            if constexpr (mt == 1 && s % 4 == 0 && s < 32) init_bias_tile<s / 4>(accB, aux, 1, h);
        });
    }

    // ---- hidden passes li = 1..NPASS: L1..L7, F (linear), and for ParamNerf C1
    constexpr int NPASS = 8 + (CFG::CD ? 1 : 0);
    float sig_part = 0.0f;
    auto hidden_pass = [&](auto LI, f32x16 (&cur)[8], f32x16 (&prev)[8]) {
        constexpr int li = decltype(LI)::value;
        constexpr bool relu_in = li != 9;          // the input of C1 is the linear "feature" layer (model.py:114)
        constexpr int rec0 = CFG::rec_pass(li);
        constexpr bool has_pos = li == SKIP + 1, has_dir = CFG::CD != 0 && li == 9;
        constexpr int pre_steps = has_pos ? CFG::PS : (has_dir ? CFG::DS : 0);
        constexpr int next_bias = li + 1;          // layer that will accumulate into `prev` next (8 = F, 9 = C1, 10 = C2)
        constexpr bool init_next = li < NPASS;     // the colour-half layer initialises its own 4 tiles
        // re-initialise tile T of the drained set (free from k-step 16 T on) with the next layer's bias
        auto reinit = [&](auto S, auto MT) {
            constexpr int s = decltype(S)::value, mt = decltype(MT)::value;
            if constexpr (init_next && mt == 1 && (s & 15) == 0) {
                if constexpr (HOIST == 4 && CFG::CD != 0 && next_bias == 9)
                    init_bias_tile_row<(s >> 4)>(prev, c1_row + (int)lane_slots[(lane & 31) + opaque_zero] * DIR_ROW_STRIDE, h);
                else if constexpr (HOIST != 0 && CFG::CD != 0 && next_bias == 9) init_bias_tile_row<(s >> 4)>(prev, c1_row + opaque_zero, h);
                else if constexpr (GEO_ROWS && next_bias == SKIP + 1) init_bias_tile_row<(s >> 4)>(prev, c1_row + 2 * DIR_BLOCK_FLOATS + opaque_zero, h);
                else init_bias_tile<(s >> 4)>(prev, aux, next_bias, h);
            }
        };
        // alpha head (model.py:111) rides on pass F, which consumes the same activations relu(L7)
        auto alpha_head = [&](auto S, auto MT) {
            constexpr int s = decltype(S)::value, mt = decltype(MT)::value;
            if constexpr (li == DEPTH && mt == 2)
                sig_part = __builtin_fmaf(hin[s], aux[aux_alpha_off() + h * 128 + s], sig_part);
        };
        if constexpr (pre_steps > 0) {
            store_act<8, relu_in>(hin, prev);
            auto conv = none;
            const SampleIn<NGEO, NAPP> in2 = launder(in);
            if constexpr (has_pos) {   // input = concat[pos_map, h]  (model.py:107-108)
                PosGen<NGEO, NAPP, CFG::IPE, KEEP_PE ? 2 : 0, GS> gen{in2, h, {}, pe};   // the values layer 0 kept
                run_segment<M, pre_steps - GS, 8, M::log_of(rec0 + GS * 2)>(cur, ws, gen, conv);
                if constexpr (HOIST == 4) mid();
            } else {                   // input = concat[dir_map, feature]  (model.py:115)
                if constexpr (HOIST == 0) {   // (hoisted: already in the accumulators through c1_row, and not in the logical stream)
                    DirGen<NGEO, NAPP> gen{in2, h, {}};
                    run_segment<M, pre_steps, 8, M::log_of(rec0)>(cur, ws, gen, conv);
                }
            }
            HiddenGen hg{hin};
            run_segment<M, HSTEPS, 8, M::log_of(rec0 + pre_steps * 2)>(cur, ws, hg, [&](auto S, auto MT) { reinit(S, MT); alpha_head(S, MT); });
        } else {
            store_act<8, relu_in>(hin, prev);
            HiddenGen hg{hin};
            run_segment<M, HSTEPS, 8, M::log_of(rec0)>(cur, ws, hg, [&](auto S, auto MT) { reinit(S, MT); alpha_head(S, MT); });
        }
    };
    static_for<NPASS>([&](auto I) {
        constexpr int li = decltype(I)::value + 1;
        if constexpr (li & 1) hidden_pass(std::integral_constant<int, li>{}, accB, accA);
        else hidden_pass(std::integral_constant<int, li>{}, accA, accB);
    });
    sigma = sig_part + __shfl_xor(sig_part, 32, 64) + aux[aux_alpha_off() + 256];

    // ---- colour half layer (-> 128, relu; model.py:122 / 42), 4 M-tiles.
    // ParamNerf: input relu(C1) from set B, output set A.  Nerf: input [dir_map, F (linear, set A)], output set B.
    auto color_half = [&](f32x16 (&cur)[8], f32x16 (&prev)[8]) {
        init_bias<4>(cur, aux, 10, h);
        if constexpr (CFG::CD == 0) {   // plain Nerf: input = concat[dir_map, feature]  (model.py:39-42)
            store_act<8, false>(hin, prev);
            const SampleIn<NGEO, NAPP> in2 = launder(in);
            DirGen<NGEO, NAPP> gen{in2, h, {}};
            run_segment<M, CFG::DS, 4, M::log_of(CFG::REC_C2)>(cur, ws, gen, none);
            HiddenGen hg{hin};
            run_segment<M, HSTEPS, 4, M::log_of(CFG::REC_C2 + CFG::DS)>(cur, ws, hg, none);
        } else {
            store_act<8, true>(hin, prev);
            HiddenGen hg{hin};
            run_segment<M, HSTEPS, 4, M::log_of(CFG::REC_C2)>(cur, ws, hg, none);
        }
        store_act<4, true>(hin, cur);
    };
    if constexpr (NPASS & 1) color_half(accA, accB);   // last pass wrote set B
    else color_half(accB, accA);
    static_assert(CFG::REC_C2 + (CFG::CD == 0 ? CFG::DS : 0) + HSTEPS == CFG::REC_END, "stream bookkeeping");
    static_assert(M::log_of(CFG::REC_END) == M::LOG_END, "logical stream bookkeeping");
    skip_records<M, M::LOG_PAD - M::LOG_END, M::LOG_END>(ws);

    // ---- rgb head (128 -> 3, linear; model.py:123) on the VALU
    static_for<3>([&](auto C) {
        constexpr int c = C;
        const f32x4 *wc = reinterpret_cast<const f32x4 *>(aux + aux_rgb_off() + (c * 2 + h) * 64);
        float p = 0.0f;
        static_for<16>([&](auto I) {
            constexpr int i = I;
            const f32x4 w = wc[i];
            p = __builtin_fmaf(hin[4 * i + 0], w.x, p);
            p = __builtin_fmaf(hin[4 * i + 1], w.y, p);
            p = __builtin_fmaf(hin[4 * i + 2], w.z, p);
            p = __builtin_fmaf(hin[4 * i + 3], w.w, p);
        });
        rgb[c] = p + __shfl_xor(p, 32, 64) + aux[aux_rgb_off() + 384 + c];
    });
    float chk = chk_early;
    if constexpr (HOIST != 4) chk = input_check();
    sigma += chk;
#pragma unroll
    for (int c = 0; c < 3; ++c) rgb[c] += chk;
    // LOG_PAD % RING == 0 and the logical stream wraps into itself, so the ring now holds the first RING logical records
    // in slots 0..RING-1: the next batch starts without a bubble
}

// ---------------------------------------------------------------------------------------------
// flex family: the same segments in a LOOP over layers (ntx_layout.h "flex family")
// ---------------------------------------------------------------------------------------------
// aux block of a family in LDS: the tuned layout, for the flex family followed by [descriptor | bias slots]
template <class CFG>
constexpr int aux_floats_of() { return aux_total() + (CFG::FLEX != 0 ? flex_floats() : 0); }

// run_segment with the segment's first record at byte offset `sbase` of the stream (a wave-uniform run-time value) and at ring
// phase 0; PADREC >= NSTEPS * NMT / 4 records are consumed (the pad is fetched to keep the ring turning, never multiplied) and
// sbase is advanced past them
template <int NSTEPS, int NMT, int PADREC, class Gen, class Extra>
NTX_DEV void run_segment_rt(f32x16 (&acc)[8], WStream &ws, uint32_t &sbase, Gen &gen, Extra &&extra) {
    constexpr int RPS = NMT / 4;
    static_assert(PADREC % RING == 0 && PADREC >= NSTEPS * RPS, "whole ring turns");
    auto load = [&](int rec) {
        const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ws.rsrc, ws.voff, sbase + (uint32_t)rec * 1024u, 0);
        return __builtin_bit_cast(f32x4, v);
    };
    gen.template prepare<0, (NSTEPS < PE_GROUP ? NSTEPS : PE_GROUP)>();
    float b = gen.template value<0>();
    static_for<NSTEPS>([&](auto S) {
        constexpr int s = S;
        f32x4 w;
        static_for<NMT>([&](auto MT) {
            constexpr int mt = MT;
            if constexpr (mt % 4 == 0) {
                constexpr int rec = s * RPS + mt / 4;
                w = ws.ring[rec % RING];
                ws.ring[rec % RING] = load(rec + RING);
            }
            acc[mt] = mfma32(w[mt % 4], b, acc[mt]);
            if constexpr (mt == 0 && (s + 1) % PE_GROUP == 0 && s + 1 < NSTEPS)
                gen.template prepare<s + 1, (NSTEPS - s - 1 < PE_GROUP ? NSTEPS - s - 1 : PE_GROUP)>();
            extra(S, MT);
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (s + 1 < NSTEPS) b = gen.template value<s + 1>();
    });
    static_for<PADREC - NSTEPS * RPS>([&](auto I) {
        constexpr int rec = NSTEPS * RPS + decltype(I)::value;
        ws.ring[rec % RING] = load(rec + RING);
    });
    sbase += (uint32_t)PADREC * 1024u;
}

// hin[V0..V0+8) <- max(lo, accumulators): lo = 0 is the ReLU, lo = -inf passes a linear layer through (convert8 with the bound
// in a scalar register; a NaN accumulator becomes lo, as under the ReLU -- mlp_batch's input check restores it)
template <int V0, int NH>
NTX_DEV void convert8_rt(float (&hin)[NH], const f32x16 (&prev)[8], float lo) {
    constexpr int T = V0 >> 4, R = V0 & 15;
    asm("v_accvgpr_read_b32 %0, %8\n\tv_accvgpr_read_b32 %1, %9\n\tv_accvgpr_read_b32 %2, %10\n\t"
        "v_accvgpr_read_b32 %3, %11\n\tv_accvgpr_read_b32 %4, %12\n\tv_accvgpr_read_b32 %5, %13\n\t"
        "v_accvgpr_read_b32 %6, %14\n\tv_accvgpr_read_b32 %7, %15\n\t"
        "v_max_f32 %0, %16, %0\n\tv_max_f32 %1, %16, %1\n\tv_max_f32 %2, %16, %2\n\tv_max_f32 %3, %16, %3\n\t"
        "v_max_f32 %4, %16, %4\n\tv_max_f32 %5, %16, %5\n\tv_max_f32 %6, %16, %6\n\tv_max_f32 %7, %16, %7"
        : "=&v"(hin[V0 + 0]), "=&v"(hin[V0 + 1]), "=&v"(hin[V0 + 2]), "=&v"(hin[V0 + 3]), "=&v"(hin[V0 + 4]),
          "=&v"(hin[V0 + 5]), "=&v"(hin[V0 + 6]), "=&v"(hin[V0 + 7])
        : "a"(prev[T][R + 0]), "a"(prev[T][R + 1]), "a"(prev[T][R + 2]), "a"(prev[T][R + 3]), "a"(prev[T][R + 4]),
          "a"(prev[T][R + 5]), "a"(prev[T][R + 6]), "a"(prev[T][R + 7]), "s"(lo));
}
template <int NMT, int NH>
NTX_DEV void store_act_rt(float (&hin)[NH], const f32x16 (&acc)[8], float lo) {
    static_assert(NMT * 16 <= NH, "one register per accumulator value");
    static_for<NMT * 2>([&](auto V) { convert8_rt<decltype(V)::value * 8, NH>(hin, acc, lo); });
}

// ---- param_depth > 0 (model.py:88-101; Cfg FLEX = 2): the parameter features pass `param_depth` Dense(param_width <= 128, relu)
// layers before they are concatenated to pos_map / dir_map.  A branch = FF(params) (one block of GEO_BLOCK k-steps per parameter
// slot, last slot first, as the geometry blocks of the position segment) -> 4 tiles, then param_depth - 1 hidden segments of 64
// k-steps x 4 tiles; its output stays in 64 registers and is the B operand of 64 more k-steps of the consuming layer.
template <int N>
struct ParFFGen {               // FF(par[0..N)), PAR_FREQ bands
    const float *par;           // N values of this lane's sample (registers)
    int h;
    float vals[PE_GROUP];
    template <int S>
    NTX_DEV float feature() const {
        constexpr int p = N - 1 - S / GEO_BLOCK, j = S % GEO_BLOCK;
        if constexpr (j == 0) return h ? 0.0f : par[p];
        else return sin_q(par[p] * (float)(1 << (j - 1)), h);
    }
    template <int S, int NN>
    NTX_DEV void prepare() {
        static_for<NN>([&](auto K) { vals[(S + decltype(K)::value) % PE_GROUP] = feature<S + decltype(K)::value>(); });
    }
    template <int S>
    NTX_DEV float value() const { return vals[S % PE_GROUP]; }
};
struct Reg64Gen {               // 64 k-steps out of 64 registers (a branch's hidden activations / its output)
    const float (&hb)[64];
    template <int S, int N>
    NTX_DEV void prepare() {}
    template <int S>
    NTX_DEV float value() const { return hb[S]; }
};
struct DirOnlyGen {             // FF(dir, DIR_FREQ) alone: (dx, dy), (dz, pad), {sin, cos}(2^f d_c)  =  dir_row(0, s, h)
    const float (&dir)[3];
    int h;
    float vals[PE_GROUP];
    template <int S>
    NTX_DEV float feature() const {
        if constexpr (S == 0) return h ? dir[1] : dir[0];
        else if constexpr (S == 1) return h ? 0.0f : dir[2];
        else return sin_q(dir[(S - 2) % 3] * (float)(1 << ((S - 2) / 3)), h);
    }
    template <int S, int NN>
    NTX_DEV void prepare() {
        static_for<NN>([&](auto K) { vals[(S + decltype(K)::value) % PE_GROUP] = feature<S + decltype(K)::value>(); });
    }
    template <int S>
    NTX_DEV float value() const { return vals[S % PE_GROUP]; }
};
constexpr int BRANCH_STEPS = 64;    // k-steps of a 128-wide branch activation
struct LdsColGen {              // 64 k-steps out of this lane's LDS column col[step * 64] (a branch's OUTPUT: it outlives the registers)
    const float *col;
    float vals[PE_GROUP];
    template <int S, int N>
    NTX_DEV void prepare() {
        static_for<N>([&](auto K) { vals[(S + decltype(K)::value) % PE_GROUP] = col[(S + decltype(K)::value) * 64]; });
    }
    template <int S>
    NTX_DEV float value() const { return vals[S % PE_GROUP]; }
};

// relu(accumulator tiles 0..3) -> this lane's LDS column, 8 values at a time (no 64 registers held next to a live `hin`)
NTX_DEV void drain_to_col(float *col, const f32x16 (&acc)[8]) {
    static_for<8>([&](auto V) {
        constexpr int v0 = decltype(V)::value * 8;
        float t[128];   // (only t[v0 .. v0+8) exist after SROA: convert8_rt indexes the array it is given)
        convert8_rt<v0, 128>(t, acc, 0.0f);
        static_for<8>([&](auto K) { col[(v0 + decltype(K)::value) * 64] = t[v0 + decltype(K)::value]; });
    });
}

template <int N>
NTX_DEV void param_branch(const float *par, int pdepth, f32x16 (&acc)[8], float *col, WStream &ws, uint32_t &sbase,
                          const float *fbias, int slot0, int h) {
    auto none = [](auto, auto) {};
    init_bias<4>(acc, fbias, slot0, h);
    {
        ParFFGen<N> gen{par, h, {}};
        run_segment_rt<N * GEO_BLOCK, 4, flex_seg_records(N * GEO_BLOCK, 4)>(acc, ws, sbase, gen, none);
    }
    for (int i = 1; i < pdepth; ++i) {
        drain_to_col(col, acc);              // every layer's activations pass through the column: written and read by this lane only
        init_bias<4>(acc, fbias, slot0 + i, h);
        LdsColGen hg{col, {}};
        run_segment_rt<BRANCH_STEPS, 4, flex_seg_records(BRANCH_STEPS, 4)>(acc, ws, sbase, hg, none);
    }
    drain_to_col(col, acc);
}

// The MLP of a flex model on one batch of 32 samples (model.py:58-125 / 9-45 with depth, skips, color_depth, width <= 256 as the
// model has them).  One accumulator set; a layer = [drain the set into `hin` under the lower bound of the input's activation]
// [bias of the layer] [its leading encoder segment, if it has one] [hidden segment].  For the 8 x 256 / skip 4 / color_depth 1
// model every accumulator sees the same bias, the same products in the same order as in mlp_batch_tuned: the same bits.
template <class CFG, bool KEEP_PE>
NTX_DEV void mlp_flex(const SampleIn<CFG::NGEO, CFG::NAPP> &in, WStream &ws, const float *aux_in, int lane, float &sigma,
                      float (&rgb)[3], float *pe) {
    constexpr int NGEO = CFG::NGEO, NAPP = CFG::NAPP;
    static_assert(CFG::IPE == 0 && CFG::GEN != 0, "flex family: FourierFeatures, generic parameter slots");
    constexpr bool PB = CFG::FLEX == 2;    // param_depth > 0: parameter branches (above); pos_map = [FF(pos) | G], dir_map = [FF(dir) | A]
    constexpr int GOFF = PB ? pos_geo_steps(NGEO) : 0;             // PB: the position segment is its FF(pos) part alone
    constexpr int PSTEPS = CFG::PS - GOFF, DSTEPS = PB ? dir_steps(0) : CFG::DS;
    constexpr int PS8 = flex_seg_records(PSTEPS, 8), DS8 = flex_seg_records(DSTEPS, 8), DS4 = flex_seg_records(DSTEPS, 4);
    constexpr int H8 = flex_seg_records(HSTEPS, 8), H4 = flex_seg_records(HSTEPS, 4);
    constexpr int B8 = flex_seg_records(BRANCH_STEPS, 8), B4 = flex_seg_records(BRANCH_STEPS, 4);
    const int h = lane >> 5;
    uint32_t opaque_zero = 0;
    asm volatile("" : "+v"(opaque_zero));   // as mlp_batch_tuned: keeps the reads of the (constant) aux block next to their use
    const float *aux = aux_in + opaque_zero;
    const float *fbias = aux + aux_total() + FLEX_DESC_FLOATS;
    const int *desc = reinterpret_cast<const int *>(aux_in + aux_total());
    const int depth = __builtin_amdgcn_readfirstlane(desc[0]);
    const uint32_t skip_mask = (uint32_t)__builtin_amdgcn_readfirstlane(desc[1]);
    const int cdepth = __builtin_amdgcn_readfirstlane(desc[2]);
    const int pdepth = PB ? __builtin_amdgcn_readfirstlane(desc[3]) : 0;
    const bool has_geo = PB && __builtin_amdgcn_readfirstlane(desc[4]) != 0, has_app = PB && __builtin_amdgcn_readfirstlane(desc[5]) != 0;
    const float neg_inf = __builtin_bit_cast(float, 0xff800000u);

    f32x16 acc[8];
    float hin[128];
    // PB: the output of the geometry branch, later of the appearance branch, lives in this lane's LDS column `pe` (64 k-steps x 64
    // lanes per wave: the position features are not kept there then, the skip layers evaluate them again)
    constexpr int K1 = (KEEP_PE && !PB) ? 1 : 0, K2 = (KEEP_PE && !PB) ? 2 : 0;
    uint32_t sbase = 0;
    auto none = [](auto, auto) {};
    const int n8 = depth + 1 + cdepth;
    // the part of pos_map behind FF(pos): 64 k-steps out of the geometry branch's registers (PB)
    auto geo_part = [&]() {
        if constexpr (PB) {
            if (has_geo) {
                LdsColGen gg{pe, {}};
                run_segment_rt<BRANCH_STEPS, 8, B8>(acc, ws, sbase, gg, none);
            }
        }
    };

    if constexpr (PB) {   // geometry branch first: its output feeds trunk layer 0 and every skip layer (bias slots behind the colour half's)
        if (has_geo) param_branch<NGEO>(in.par, pdepth, acc, pe, ws, sbase, fbias, n8 + 1, h);
    }
    // ---- trunk layer 0: pos_map -> width (model.py:104-106)
    init_bias<8>(acc, fbias, 0, h);
    {
        PosGen<NGEO, NAPP, 0, K1, GOFF> gen{in, h, {}, pe};
        run_segment_rt<PSTEPS, 8, PS8>(acc, ws, sbase, gen, none);
        geo_part();
    }
    // ---- the other 8-tile layers: trunk 1 .. depth-1, F (l = depth), colour layers (l = depth + 1 .. depth + cdepth)
    float sig_part = 0.0f;
    // the appearance branch, evaluated when the feature layer has been drained and the accumulators are free (PB)
    auto app_branch = [&]() {
        if constexpr (PB) {
            if (has_app) {
                const SampleIn<NGEO, NAPP> in2 = launder(in);
                param_branch<NAPP>(in2.par + NGEO, pdepth, acc, pe, ws, sbase, fbias, n8 + 1 + FLEX_MAX_PARAM_DEPTH, h);
            }
        }
    };
    auto dir_part = [&](auto NMT_) {   // [FF(dir) | A] (PB) or the direction segment with the appearance features in it
        constexpr int nmt = decltype(NMT_)::value;
        const SampleIn<NGEO, NAPP> in2 = launder(in);
        if constexpr (PB) {
            DirOnlyGen gen{in2.dir, h, {}};
            run_segment_rt<DSTEPS, nmt, (nmt == 8 ? DS8 : DS4)>(acc, ws, sbase, gen, none);
            if (has_app) {
                LdsColGen ag{pe, {}};
                run_segment_rt<BRANCH_STEPS, nmt, (nmt == 8 ? B8 : B4)>(acc, ws, sbase, ag, none);
            }
        } else {
            DirGen<NGEO, NAPP> gen{in2, h, {}};
            run_segment_rt<DSTEPS, nmt, (nmt == 8 ? DS8 : DS4)>(acc, ws, sbase, gen, none);
        }
    };
    for (int l = 1; l < n8; ++l) {
        const bool first_colour = l == depth + 1;            // its input is the LINEAR feature layer (model.py:114-115)
        store_act_rt<8>(hin, acc, first_colour ? neg_inf : 0.0f);
        if (l == depth) {                                    // alpha head on relu(trunk depth-1) (model.py:111)
            const float *wa = aux + aux_alpha_off() + h * 128;
            static_for<128>([&](auto S) { sig_part = __builtin_fmaf(hin[decltype(S)::value], wa[decltype(S)::value], sig_part); });
        }
        if (first_colour) app_branch();
        init_bias<8>(acc, fbias, l, h);
        if (l < depth && ((skip_mask >> (l - 1)) & 1u)) {    // input = concat[pos_map, h]  (model.py:107-108)
            const SampleIn<NGEO, NAPP> in2 = launder(in);
            PosGen<NGEO, NAPP, 0, K2, GOFF> gen{in2, h, {}, pe};
            run_segment_rt<PSTEPS, 8, PS8>(acc, ws, sbase, gen, none);
            geo_part();
        } else if (first_colour) {                           // input = concat[dir_map, feature]  (model.py:115)
            dir_part(std::integral_constant<int, 8>{});
        }
        HiddenGen hg{hin};
        run_segment_rt<HSTEPS, 8, H8>(acc, ws, sbase, hg, none);
    }
    sigma = sig_part + __shfl_xor(sig_part, 32, 64) + aux[aux_alpha_off() + 256];

    // ---- colour half layer (-> width / 2, relu; model.py:122 / 42), 4 tiles; color_depth = 0 (and plain Nerf): on [dir_map, feature]
    store_act_rt<8>(hin, acc, cdepth > 0 ? 0.0f : neg_inf);
    if (cdepth == 0) app_branch();
    init_bias<4>(acc, fbias, n8, h);
    if (cdepth == 0) dir_part(std::integral_constant<int, 4>{});
    {
        HiddenGen hg{hin};
        run_segment_rt<HSTEPS, 4, H4>(acc, ws, sbase, hg, none);
    }
    store_act<4, true>(hin, acc);
    // the prefetches of the last segment ran into the wrap-around tail: the ring holds the first RING records again

    // ---- rgb head (width / 2 -> 3, linear; model.py:123) on the VALU, and the input check, as mlp_batch_tuned
    static_for<3>([&](auto C) {
        constexpr int c = C;
        const f32x4 *wc = reinterpret_cast<const f32x4 *>(aux + aux_rgb_off() + (c * 2 + h) * 64);
        float p = 0.0f;
        static_for<16>([&](auto I) {
            constexpr int i = I;
            const f32x4 w = wc[i];
            p = __builtin_fmaf(hin[4 * i + 0], w.x, p);
            p = __builtin_fmaf(hin[4 * i + 1], w.y, p);
            p = __builtin_fmaf(hin[4 * i + 2], w.z, p);
            p = __builtin_fmaf(hin[4 * i + 3], w.w, p);
        });
        rgb[c] = p + __shfl_xor(p, 32, 64) + aux[aux_rgb_off() + 384 + c];
    });
    float chk = in.pos[0] + in.pos[1] + in.pos[2] + in.dir[0] + in.dir[1] + in.dir[2];
#pragma unroll
    for (int k = 0; k < CFG::NP; ++k) chk += in.par[k];
    chk = chk - chk;
    sigma += chk;
#pragma unroll
    for (int c = 0; c < 3; ++c) rgb[c] += chk;
}

// the network of a family on one batch: the straight-line 8 x 256 kernels, or the layer loop of the flex family
template <class CFG, int HOIST = 0, bool KEEP_PE = true, class Mid = NoMid>
NTX_DEV void mlp_batch(const SampleIn<CFG::NGEO, CFG::NAPP> &in, WStream &ws,
                       const float *aux_in, int lane, float &sigma, float (&rgb)[3],
                       const float *c1_row = nullptr, float *pe = nullptr, const uint8_t *lane_slots = nullptr, Mid &&mid = Mid{}) {
    if constexpr (CFG::FLEX != 0) {
        static_assert(HOIST == 0, "the flex family evaluates everything per sample");
        mlp_flex<CFG, KEEP_PE>(in, ws, aux_in, lane, sigma, rgb, pe);
    } else {
        mlp_batch_tuned<CFG, HOIST, KEEP_PE>(in, ws, aux_in, lane, sigma, rgb, c1_row, pe, lane_slots, static_cast<Mid &&>(mid));
    }
}

NTX_DEV void load_aux(float *lds, const float *aux_g, int n) {
    for (int i = threadIdx.x; i < n / 4; i += blockDim.x)
        reinterpret_cast<f32x4 *>(lds)[i] = reinterpret_cast<const f32x4 *>(aux_g)[i];
    __syncthreads();
}

// LDS behind the aux block: one column of position-segment values per lane and wave (PosGen KEEP)
template <class CFG>
constexpr int pe_keep_floats() { return (CFG::FLEX == 2 ? 64 /* BRANCH_STEPS: the branch outputs' column */ : CFG::PS) * 64; }
template <class CFG>
NTX_DEV float *pe_column(float *aux, int wave_in_wg, int lane) {
    return aux + aux_floats_of<CFG>() + wave_in_wg * pe_keep_floats<CFG>() + lane;
}

// slot k of the kernel's parameter vector from a row of the caller's parameters, and the width of those rows
template <class CFG, class Args>
NTX_DEV float param_at(const Args &a, const float *row, int k) {
    if constexpr (CFG::GEN != 0) {
        const int c = a.pmap[k];
        return c >= 0 ? row[c] : 0.0f;
    } else {
        return row[k];
    }
}
template <class CFG, class Args>
NTX_DEV int param_stride(const Args &a) {
    if constexpr (CFG::GEN != 0) return a.np_in;
    else return CFG::NP_IN;
}

// ---------------------------------------------------------------------------------------------
// alpha compositing of one batch of <= 32 consecutive samples of one ray (renderer.py:170-213),
// lane j = l&31 holds sample j (both half-waves hold the same values); carry across batches in `ra`
// ---------------------------------------------------------------------------------------------
struct RayAccum {
    float T, c0, c1, c2, a;
};

template <int W>
NTX_DEV float wave_sum(float v) {
#pragma unroll
    for (int d = W / 2; d >= 1; d >>= 1) v += __shfl_xor(v, d, W);
    return v;
}

// the scan itself: `a` = this sample's alpha in [0,1], `c` = its mapped colour
template <int W>
NTX_DEV void composite_core(RayAccum &ra, float a, const float (&c)[3], bool valid, int j, float *w_out) {
    const float trans = (1.0f - a) + 1e-10f;                                          // renderer.py:198 / 342
    float P = trans;   // inclusive product scan over the W lanes of the batch
#pragma unroll
    for (int d = 1; d < W; d <<= 1) {
        const float v = __shfl_up(P, d, W);
        if (j >= d) P = v * P;
    }
    float E = __shfl_up(P, 1, W);
    if (j == 0) E = 1.0f;
    const float w = a * (ra.T * E);
    if (w_out && valid) *w_out = w;
    ra.c0 += wave_sum<W>(valid ? w * c[0] : 0.0f);
    ra.c1 += wave_sum<W>(valid ? w * c[1] : 0.0f);
    ra.c2 += wave_sum<W>(valid ? w * c[2] : 0.0f);
    ra.a += wave_sum<W>(valid ? w : 0.0f);
    ra.T *= __shfl(P, W - 1, W);
}

// `noise` = this sample's N(0, raw_noise_std) draw (renderer.py:190-192), 0 when the regulariser is off
template <int W>
NTX_DEV void composite_step(RayAccum &ra, float sigma, const float (&raw)[3], float dist, bool valid,
                            uint32_t flags, int j, float *w_out, float noise = 0.0f) {
    float c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) c[k] = (flags & NTX_FLAG_MAP_EXR) ? elu1f_(raw[k]) : sigmoidf_(raw[k]);   // :182-187
    const float a = valid ? 1.0f - expf(-__builtin_fmaxf(sigma + noise, 0.0f) * dist) : 0.0f;          // :195
    composite_core<W>(ra, a, c, valid, j, w_out);
}

// ---------------------------------------------------------------------------------------------
// fused render kernel: one wave per ray, S/32 batches per ray (renderer.py:47-213)
// ---------------------------------------------------------------------------------------------
// mip-NeRF cone-segment gaussian (renderer.py:416-424 / 575-578): moments along the ray and across it
NTX_DEV void cone_moments(float mu, float hw, float radii, float &t_mean, float &t_var, float &r_var) {
    const float mu2 = mu * mu, hw2 = hw * hw, hw4 = hw2 * hw2, den = 3.0f * mu2 + hw2;
    t_mean = mu + (2.0f * mu * hw2) / den;
    t_var = hw2 / 3.0f - (4.0f / 15.0f) * ((hw4 * (12.0f * mu2 - hw2)) / (den * den));
    r_var = (radii * radii) * (mu2 / 4.0f + (5.0f / 12.0f) * hw2 - 4.0f / 15.0f * hw4 / den);
}
// diagonal covariance in world space (renderer.py:429-435 / 581-586)
NTX_DEV void cone_cov(float t_var, float r_var, const float (&d)[3], float (&cov)[3]) {
    const float mag = __builtin_fmaxf(1e-10f, d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float dd = d[c] * d[c];
        cov[c] = t_var * dd + r_var * (1.0f - dd / mag);
    }
}

struct RenderArgs {
    const f32x4 *wstream;
    uint32_t stream_bytes;   // stream + wrap-around tail
    const float *aux;
    const float *rays_o, *rays_d, *t, *params, *cone, *z_vals;
    float *color_out, *alpha_out, *weights_out;   // weights_out: NULL or [N,S] per-sample compositing weights
    int32_t *status;
    int64_t n_rays, rays_per_row;
    int n_samples, blur_idx;
    uint32_t flags;
    float delta;   // float32(1 / (S - 1)): the step of tf.linspace(0., 1., S)
    float bkgd[3];
    // the compacted indices of the rays with t0 != inf and their number (compact_hits_kernel, which has then already
    // written the culled rays): every wave gets the same number of rays to march, however the misses are distributed
    // over the image
    const int32_t *hit_list, *hit_count;
    uint32_t seed_lo, seed_hi;   // NTX_FLAG_PERTURB: key of the counter-based generator behind the stratified jitter
    // fp16x3 kernels only: the float32 stream, whose records of C1's direction segment dir_block multiplies
    const f32x4 *dir_wstream;
    uint32_t dir_stream_bytes;
    int np_in;                        // generic family: width of the caller's parameter rows, and the column of every slot (-1: absent)
    int8_t pmap[MAX_PARAM_SLOTS];
    // ABI v3.  The counter of the jitter / noise generator is the GLOBAL index of a ray,
    //   idx0 + (k / idx_run) * idx_stride + k % idx_run   for local ray k (ntx_render_opts; identity: 0, 0xffffffff, 0),
    // so a sharded or chunked image draws what the whole image draws.
    float raw_noise_std;              // NTX_FLAG_RAW_NOISE: sigma += raw_noise_std * N(0,1) per sample (renderer.py:190-192)
    uint32_t idx_run;
    int64_t idx0, idx_stride;
};

// this lane's index within its wave64, recomputed where it is called (v_mbcnt on an opaque zero: not hoistable, not CSE-able
// with the kernel's own `lane`), for code that runs rarely between long register-starved stretches
NTX_DEV int fresh_lane_id() {
    uint32_t zero = 0;
    asm volatile("" : "+v"(zero));
    return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, zero));
}

// the by-value kernel argument struct, addressed in the kernarg segment (device pass only)
template <class T>
NTX_DEV const T *kernargs() {
#if defined(__HIP_DEVICE_COMPILE__)
    return (const T *)__builtin_amdgcn_kernarg_segment_ptr();
#else
    return nullptr;
#endif
}

// ---------------------------------------------------------------------------------------------
// sample depths (renderer.py:101-111)
// ---------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11; the generator behind tf.random.uniform), word 0 of the block at `ctr` under
// `key`.  Counter-based: the draw for (ray, sample) is a pure function of (seed, ray, sample), so the jitter needs no
// state, no [N,S] tensor and is independent of how rays are split over launches, waves or GPUs.
NTX_DEV uint32_t philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t *word1 = nullptr) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    if (word1) *word1 = c1;
    return c0;
}
// uniform float32 in [0,1) from 23 random bits, as tf.random.uniform makes it (random_distributions.h Uint32ToFloat)
NTX_DEV float uniform01(uint32_t x) { return __builtin_bit_cast(float, (x & 0x7fffffu) | 0x3f800000u) - 1.0f; }
// N(0,1) from words 0 and 1 of the Philox block at counter (sample, ray lo, ray hi, 1): the first output of the Box-Muller
// transform tf.random.normal applies to two uniforms (random_distributions.h BoxMullerFloat: u1 clamped to 1e-7,
// sqrt(-2 ln u1) sin(2 pi u2)).  Counter word 3 = 1 keeps the stream apart from the jitter's (word 3 = 0).
NTX_DEV float normal01(int64_t gray, int i, uint32_t seed_lo, uint32_t seed_hi) {
    uint32_t x1;
    const uint32_t x0 = philox4x32_10((uint32_t)i, (uint32_t)gray, (uint32_t)((uint64_t)gray >> 32), 1u, seed_lo, seed_hi, &x1);
    const float u1 = __builtin_fmaxf(uniform01(x0), 1.0e-7f);
    const float v1 = 6.28318530717958647692f * uniform01(x1);
    return sinf(v1) * __builtin_sqrtf(-2.0f * logf(u1));
}

// depth i of the npts points of tf.linspace between t0 and t1 (npts = S samples, or S+1 segment edges for the mip
// renderer, renderer.py:374-376); delta = float32(1 / (npts - 1))
NTX_DEV float z_lin(float delta, int i, float t0, float t1, int npts) {
    const float tv = i == 0 ? 0.0f : (i == npts - 1 ? 1.0f : delta * (float)i);
    return t0 * (1.0f - tv) + t1 * tv;   // renderer.py:102
}
// the same with the stratified jitter of renderer.py:106-111 / 379-383: uniform in [lower_i, upper_i), the midpoints to
// the neighbouring depths (the end points themselves at both ends); z_rand = the Philox draw at counter (i, ray)
NTX_DEV float z_jittered(float delta, int64_t ray, int i, float t0, float t1, int npts, uint32_t seed_lo, uint32_t seed_hi) {
    const float zc = z_lin(delta, i, t0, t1, npts);
    const float lower = i == 0 ? zc : 0.5f * (zc + z_lin(delta, i - 1, t0, t1, npts));
    const float upper = i == npts - 1 ? zc : 0.5f * (z_lin(delta, i + 1, t0, t1, npts) + zc);
    const float u = uniform01(philox4x32_10((uint32_t)i, (uint32_t)ray, (uint32_t)((uint64_t)ray >> 32), 0u, seed_lo, seed_hi));
    return lower + (upper - lower) * u;
}
// global index of local ray k under the index map (idx0, idx_run, idx_stride); k < 2^31 (ntx_reserve)
NTX_DEV int64_t global_index(int64_t idx0, uint32_t idx_run, int64_t idx_stride, int64_t k) {
    const uint32_t r = (uint32_t)k, q = r / idx_run;
    return idx0 + (int64_t)q * idx_stride + (int64_t)(r - q * idx_run);
}
// `ray` indexes the call's arrays, `gray` = its global index keys the generator
NTX_DEV float z_of(const RenderArgs &a, int64_t ray, int64_t gray, int i, float t0, float t1, int npts) {
    if (a.z_vals) return a.z_vals[ray * npts + i];
    if (a.flags & NTX_FLAG_PERTURB) return z_jittered(a.delta, gray, i, t0, t1, npts, a.seed_lo, a.seed_hi);
    return z_lin(a.delta, i, t0, t1, npts);
}

// ---------------------------------------------------------------------------------------------
// per-ray-constant direction features hoisted out of the per-sample network.  In Renderer.evaluate_model the view
// direction and the appearance parameters are repeated for every sample of a ray (renderer.py:152-154), so the
// direction segment of the colour layer, W_C1[:dir_map]^T dir_map, is one [256] vector per ray: 328 of 10 638 MFMAs and
// 41 of 113 encoder k-steps per 32-sample batch that need not be repeated.  dir_block computes bias_C1 + that vector for
// the next 32 rays of a WORKGROUP (8 per wave) straight into LDS: lane = ray slot, each of the 4 waves takes 2 of the 8
// output tiles, same bias initialisation, same k-order, same generator as the per-sample evaluation -- an output column
// of the MFMA depends only on its own B column, so starting the C1 accumulators from the stored row is bit-identical
// to evaluating the segment per sample.  No global scratch, no extra launch, no HBM traffic; one pair of workgroup
// barriers per 8 rays of each wave.  Not applicable when blur_idx scales an APPEARANCE parameter per sample
// (renderer.py:155-158); the host then launches the HOIST = 0 kernel.
// ---------------------------------------------------------------------------------------------
// one family of rows: acc = bias[BIAS_LAYER] + sum over k-steps 0 .. NSTEPS-1 of the segment whose first record is REC0, B
// operand from FEAT(step); this wave's 2 output tiles of the 32 ray slots -> rows[slot][half][16 t + r]
template <int NSTEPS, int REC0, int BIAS_LAYER, class Feat>
NTX_DEV void ray_rows(__amdgpu_buffer_rsrc_t rsrc, const float *aux, float *rows, int j, int h, int wv, int lane, Feat &&feat) {
    // tiles 2 wv and 2 wv + 1: elements (2 wv) % 4, +1 of record 2 s + wv / 2 of k-step s
    const int t0 = 2 * wv;
    f32x16 acc0, acc1;
    {
        const f32x4 *b = reinterpret_cast<const f32x4 *>(aux + BIAS_LAYER * AUX_BIAS_STRIDE + h * 128) + t0 * 4;
        const f32x4 v0 = b[0], v1 = b[1], v2 = b[2], v3 = b[3], v4 = b[4], v5 = b[5], v6 = b[6], v7 = b[7];
        acc0 = f32x16{v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
        acc1 = f32x16{v4.x, v4.y, v4.z, v4.w, v5.x, v5.y, v5.z, v5.w, v6.x, v6.y, v6.z, v6.w, v7.x, v7.y, v7.z, v7.w};
    }
    const uint32_t voff = (uint32_t)lane * 16u;
    const uint32_t rec_base = (uint32_t)(REC0 + (wv >> 1)) * 1024u;
    const bool odd = wv & 1;
    static_for<NSTEPS>([&](auto S) {
        constexpr int s = S;
        const i32x4 wi = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, rec_base + (uint32_t)s * 2048u, 0);
        const f32x4 w = __builtin_bit_cast(f32x4, wi);
        const float b = feat(S);
        acc0 = mfma32(odd ? w.z : w.x, b, acc0);
        acc1 = mfma32(odd ? w.w : w.y, b, acc1);
    });
    // lane (slot j, half h), register r of tile t = feature hidden_row(16 t + r, h) of ray slot j -> rows[j][h][16 t + r]
    f32x4 *o = reinterpret_cast<f32x4 *>(rows + j * DIR_ROW_STRIDE + h * 128 + t0 * 16);
    static_for<4>([&](auto Q) {
        constexpr int q = Q;
        o[q] = f32x4{acc0[4 * q], acc0[4 * q + 1], acc0[4 * q + 2], acc0[4 * q + 3]};
        o[4 + q] = f32x4{acc1[4 * q], acc1[4 * q + 1], acc1[4 * q + 2], acc1[4 * q + 3]};
    });
}

// CONSEC (render_kernel): a workgroup takes DIR_BLOCK_RAYS CONSECUTIVE hits of a block, wave w the DIR_BLOCK_ITERS from w *
//   DIR_BLOCK_ITERS on: rows[slot]  <->  hit number base + DIR_BLOCK_RAYS * workgroup + slot.
// else (render_kernel_x3): rows[slot] for slot = it * 4 + w  <->  hit number base + it * nwaves_total + 4 * workgroup + w.
// GS > 0: also the rows of L0 and L5 (bias + the first GS k-steps of their position segments: geometry blocks), at rows +
// DIR_BLOCK_FLOATS and rows + 2 DIR_BLOCK_FLOATS (render_kernel<CFG, 2 | 3>)
template <class CFG, int GS = 0, bool CONSEC = false>   // GS: leading geometry k-steps of the position segments to evaluate too
NTX_DEV void dir_block(const RenderArgs &a, __amdgpu_buffer_rsrc_t rsrc, const float *aux, float *rows, int base, int nwaves,
                       int wg, int wv, int lane, int n_work) {
    static_assert(CFG::CD != 0, "ParamNerf families");
    const int h = lane >> 5;
    static_assert(DIR_BLOCK_RAYS % 32 == 0, "whole MFMA column sets");
    for (int j = lane & 31; j < DIR_BLOCK_RAYS; j += 32) {   // 32 ray slots per MFMA column set
        // (wave-uniform) the rest of the block lies past the end of the list
        if (j >= 32 && (CONSEC ? base + DIR_BLOCK_RAYS * wg + (j & ~31) : base + (j >> 5) * 8 * nwaves) >= n_work) break;
        int idx = CONSEC ? base + DIR_BLOCK_RAYS * wg + j : base + (j >> 2) * nwaves + 4 * wg + (j & 3);
        idx = idx < n_work ? idx : n_work - 1;
        const int64_t ray = a.hit_list[idx];
        const float dx = a.rays_d[3 * ray], dy = a.rays_d[3 * ray + 1], dz = a.rays_d[3 * ray + 2];
        const float dnorm = __builtin_sqrtf(dx * dx + dy * dy + dz * dz);   // as the render kernel (renderer.py:98)
        SampleIn<CFG::NGEO, CFG::NAPP> in;
        in.pos[0] = in.pos[1] = in.pos[2] = 0.0f;
        in.cov[0] = in.cov[1] = in.cov[2] = 0.0f;
        in.dir[0] = dx / dnorm; in.dir[1] = dy / dnorm; in.dir[2] = dz / dnorm;
        const float *prow = a.params + (ray / a.rays_per_row) * param_stride<CFG>(a);
#pragma unroll
        for (int k = 0; k < CFG::NP; ++k) in.par[k] = param_at<CFG>(a, prow, (CFG::IPE != 0 && k >= a.blur_idx) ? k + 1 : k);
        ray_rows<CFG::DS, CFG::rec_pass(9), 9>(rsrc, aux, rows, j, h, wv, lane,
                                               [&](auto S) { return dir_feature<CFG::NGEO, CFG::NAPP, decltype(S)::value>(in, h); });
        if constexpr (GS > 0) {
            auto geo = [&](auto S) { return pos_feature<CFG::NGEO, CFG::NAPP, CFG::IPE, decltype(S)::value>(in, h); };
            ray_rows<GS, 0, 0>(rsrc, aux, rows + DIR_BLOCK_FLOATS, j, h, wv, lane, geo);
            ray_rows<GS, CFG::rec_pass(SKIP + 1), SKIP + 1>(rsrc, aux, rows + 2 * DIR_BLOCK_FLOATS, j, h, wv, lane, geo);
        }
    }
}

// Workgroups are dealt to the 8 XCDs round-robin (workgroup w runs on XCD w % 8), each XCD with its own L2.  Rays are
// handed out by a VIRTUAL workgroup number that is XCD-major, so that in every round the workgroups of one XCD take one
// contiguous run of the hit list: neighbouring rays share the cache lines of rays_o / rays_d / t / cone (12 + 12 + 8 + 4
// bytes per ray), and a line is then fetched into ONE L2 instead of up to eight.
NTX_DEV int xcd_major_workgroup(int wg, int n_wgs) {
    constexpr int XCDS = 8;
    return (n_wgs % XCDS) ? wg : (wg % XCDS) * (n_wgs / XCDS) + wg / XCDS;
}

// HOIST: 0 = everything per sample; 1 = the direction segment of C1 per ray; 2 = also the geometry-parameter blocks of the
// position segments of L0 and L5 (no blur_idx); 3 = all of them but parameter 0's (blur_idx = 0)
template <class CFG, int HOIST = 0>
__global__ __launch_bounds__(256) void render_kernel(RenderArgs a) {
    static_assert(HOIST == 0 || CFG::CD != 0, "hoisting is for the ParamNerf families");
    static_assert(HOIST < 2 || (CFG::IPE == 0 && CFG::NGEO > 0), "geometry hoisting: FourierFeatures families with geometry parameters");
    static_assert(HOIST != 3 || CFG::NGEO >= 2, "HOIST 3 keeps parameter 0's block per sample and hoists the others");
    __shared__ __attribute__((aligned(16))) float aux[aux_floats_of<CFG>() + 4 * pe_keep_floats<CFG>() + (HOIST >= 2 ? 3 : HOIST) * DIR_BLOCK_FLOATS];
    __shared__ __attribute__((aligned(16))) float out_all[4][DIR_BLOCK_ITERS][4];   // the RGBA of a wave's rays of the block ...
    load_aux(aux, a.aux, aux_floats_of<CFG>());
    float *dir_rows = aux + aux_floats_of<CFG>() + 4 * pe_keep_floats<CFG>();
    const int lane = threadIdx.x & 63, j = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int vwg = xcd_major_workgroup(blockIdx.x, gridDim.x);
    const int wave = __builtin_amdgcn_readfirstlane(vwg * 4 + (threadIdx.x >> 6));
    const int nwaves = gridDim.x * 4;
    const int S = a.n_samples;
    const int nb = (S + 31) >> 5;
    WStream ws;
    ws_prime<RecMap<CFG, HOIST>>(ws, a.wstream, a.stream_bytes, lane);

    // The compacted hit list is walked in blocks of DIR_BLOCK_ITERS rays per wave; `base` and the trip count of this loop are
    // uniform over the workgroup (the barriers of the HOIST variant sit in it).  Within a block a wave takes DIR_BLOCK_ITERS
    // CONSECUTIVE hits (the workgroup 32, its XCD 1024): their RGBA is collected in LDS and written by ONE store instruction
    // per block -- 24 + 8 lanes, 96 + 32 contiguous bytes where the hits are neighbouring pixels -- instead of 4 scalar stores of
    // lane 0 per ray scattered over the image (round 2: WRITE_SIZE 36 MB for 10 MB of RGBA on fur_sharded, partial lines
    // evicted one by one from an L2 the weight stream keeps turning over).
    const int n_work = *a.hit_count;
    for (int base = 0; base < n_work; base += DIR_BLOCK_ITERS * nwaves) {
        if constexpr (HOIST != 0) {
            __syncthreads();   // every wave is done with the previous block's rows
            // dir_block's lane-derived addresses and its view of the arguments are loop invariants that LICM would carry
            // across the ~10 000-MFMA body below, where every register is taken: 4 dwords of scratch per lane (r2 profiles:
            // Scratch_Size 20).  A lane index read afresh from the hardware and an opaque view of the arguments make it
            // recompute them here, once per 8 rays.
            const int lane_o = fresh_lane_id();
            const RenderArgs *apd = kernargs<RenderArgs>();
            asm volatile("" : "+s"(apd));
            dir_block<CFG, hoisted_geo_steps<CFG, HOIST>(), true>(*apd, ws.rsrc, aux, dir_rows, base, nwaves, vwg, wv, lane_o, n_work);
            __syncthreads();
        }
      for (int it = 0; it < DIR_BLOCK_ITERS; ++it) {
        const int idx = base + wave * DIR_BLOCK_ITERS + it;
        if (idx >= n_work) break;
        const int64_t ray = (int64_t)a.hit_list[idx];
        RayAccum ra{1.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        for (int b = 0; b < nb; ++b) {
            // The ray's own data is re-read (scalar loads, L2-resident) for every batch instead of staying live
            // across the ~10 600-MFMA body, where it cost ~10 spilled registers per ray (1.3 GB of scratch
            // stores per 800x800 launch).  The opaque copy of the index stops LICM from hoisting the loads back.
            // (readfirstlane: with the RGBA buffer below hipcc may keep `ray` in a vector register, and an "s" operand cannot be
            //  copied out of one -- "illegal VGPR to SGPR copy"; hit indices are < 2^31)
            int r32 = __builtin_amdgcn_readfirstlane((int)ray);
            asm volatile("" : "+s"(r32));
            int64_t r = r32;
            // same for the kernel arguments: read them from the kernarg segment through an opaque pointer at the
            // point of use, so a dozen argument pointers are not held in SGPRs (spilled into VGPR lanes) all along
            const RenderArgs *ap = kernargs<RenderArgs>();
            asm volatile("" : "+s"(ap));
            const RenderArgs &q = *ap;
            const float t0 = q.t[2 * r], t1 = q.t[2 * r + 1];
            const float ox = q.rays_o[3 * r], oy = q.rays_o[3 * r + 1], oz = q.rays_o[3 * r + 2];
            const float dx = q.rays_d[3 * r], dy = q.rays_d[3 * r + 1], dz = q.rays_d[3 * r + 2];
            const float dnorm = __builtin_sqrtf(dx * dx + dy * dy + dz * dz);   // renderer.py:98, 180
            const float cone = q.cone ? q.cone[r] : 0.0f;
            const float *prow = q.params + (r / q.rays_per_row) * param_stride<CFG>(q);
            const int i = 32 * b + j;
            const bool valid = i < S;
            const int ic = valid ? i : S - 1;
            const int blur_idx = q.blur_idx;
            // the generator's counter is the ray's global index (a scalar division, only when something is drawn)
            const int64_t gr = (q.flags & (NTX_FLAG_PERTURB | NTX_FLAG_RAW_NOISE)) ? global_index(q.idx0, q.idx_run, q.idx_stride, r) : r;
            SampleIn<CFG::NGEO, CFG::NAPP> in;
            in.dir[0] = dx / dnorm; in.dir[1] = dy / dnorm; in.dir[2] = dz / dnorm;      // rays_d_n
            float dist;
            if constexpr (CFG::IPE == 0) {
                const float z = z_of(q, r, gr, ic, t0, t1, S);
                // dists: z[i+1]-z[i], the last one a copy of the previous (renderer.py:174-177), times |d| (:180)
                const float zn = z_of(q, r, gr, ic < S - 1 ? ic + 1 : ic - 1, t0, t1, S);
                dist = (ic < S - 1 ? zn - z : z - zn) * dnorm;
                in.pos[0] = ox + dx * z; in.pos[1] = oy + dy * z; in.pos[2] = oz + dz * z;   // renderer.py:114
                in.cov[0] = in.cov[1] = in.cov[2] = 0.0f;
#pragma unroll
                for (int k = 0; k < CFG::NP; ++k) {
                    float p = param_at<CFG>(q, prow, k);
                    if (k == blur_idx) p = p * (cone * z);                                // renderer.py:155-158
                    in.par[k] = p;
                }
            } else {
                // MipRenderer.render_rays (renderer.py:365-409): sample i = the cone segment between edges i and i+1 of
                // S+1 depths, encoded by its gaussian (mean, diagonal covariance); the blur parameter times cone_scale
                // is the cone radius and is spliced out of the model's parameters; dists need no copy (:441-444)
                const float e0 = z_of(q, r, gr, ic, t0, t1, S + 1), e1 = z_of(q, r, gr, ic + 1, t0, t1, S + 1);
                dist = (e1 - e0) * dnorm;
                float t_mean, t_var, r_var;
                cone_moments((e0 + e1) / 2.0f, (e1 - e0) / 2.0f, prow[blur_idx] * cone, t_mean, t_var, r_var);
                in.pos[0] = ox + dx * t_mean; in.pos[1] = oy + dy * t_mean; in.pos[2] = oz + dz * t_mean;
                const float dd[3] = {dx, dy, dz};
                cone_cov(t_var, r_var, dd, in.cov);
#pragma unroll
                for (int k = 0; k < CFG::NP; ++k) in.par[k] = prow[k < blur_idx ? k : k + 1];
            }
            float sigma, raw[3];
            float *pe = pe_column<CFG>(aux, wv, lane);
            if constexpr (HOIST != 0) mlp_batch<CFG, HOIST>(in, ws, aux, lane, sigma, raw, dir_rows + (wv * DIR_BLOCK_ITERS + it) * DIR_ROW_STRIDE, pe);
            else mlp_batch<CFG>(in, ws, aux, lane, sigma, raw, nullptr, pe);
            const RenderArgs *ap2 = kernargs<RenderArgs>();
            asm volatile("" : "+s"(ap2));
            float noise = 0.0f;   // renderer.py:190-192; drawn after the network: nothing more lives across its ~10 000 MFMAs
            if (ap2->flags & NTX_FLAG_RAW_NOISE)
                noise = ap2->raw_noise_std * normal01(global_index(ap2->idx0, ap2->idx_run, ap2->idx_stride, ray), ic, ap2->seed_lo, ap2->seed_hi);
            composite_step<32>(ra, sigma, raw, dist, valid, ap2->flags, j,
                               ap2->weights_out ? ap2->weights_out + ray * S + ic : nullptr, noise);
        }
        float out[4] = {ra.c0, ra.c1, ra.c2, ra.a};
        if (a.flags & NTX_FLAG_COMPOSITE_BKGD) {   // renderer.py:210-211
#pragma unroll
            for (int k = 0; k < 3; ++k) out[k] = out[k] + (1.0f - ra.a) * a.bkgd[k];
        }
        if (lane == 0) {
            *reinterpret_cast<f32x4 *>(out_all[wv][it]) = f32x4{out[0], out[1], out[2], out[3]};   // (the arrays are indexed in place: a pointer
                                                                                                      //  variable decays to a flat address)
            if ((a.flags & NTX_FLAG_CHECK_NUMERICS) && a.status) {
                const float s = out[0] + out[1] + out[2] + out[3];
                if (!(__builtin_fabsf(s) <= 3.0e38f)) atomicOr(a.status, 1);
            }
        }
      }
        // flush the block's RGBA: lane l = (ray l / 4, component l % 4).  Everything it needs is recomputed here (the number of
        // rays done from the loop bounds, the lane index afresh, the arguments from the kernarg segment): a counter or a pointer
        // carried through the loop above is a spilled register in the families that have none to spare.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        static_assert(DIR_BLOCK_ITERS * 4 <= 64, "one lane per output float");
        {
            const RenderArgs *apo = kernargs<RenderArgs>();
            asm volatile("" : "+s"(apo));
            const int first = base + wave * DIR_BLOCK_ITERS, left = n_work - first;
            const int n_done = left < 0 ? 0 : (left < DIR_BLOCK_ITERS ? left : DIR_BLOCK_ITERS);
            const int lane_o = fresh_lane_id();
            if (lane_o < 4 * n_done) {
                const int64_t oray = apo->hit_list[first + (lane_o >> 2)];
                const int c = lane_o & 3;
                float *dst = c < 3 ? apo->color_out + 3 * oray + c : apo->alpha_out + oray;
                *dst = out_all[wv][lane_o >> 2][c];
            }
        }
        __builtin_amdgcn_wave_barrier();   // the buffer is rewritten in the next block
    }
}

// ---------------------------------------------------------------------------------------------
// InstanceRenderer tail (renderer.py:247-354) on the buffers of instancer.get_model_input
// (instancer.pyx:38-54).  One wave per hit ray: (1) wave-level compaction of the in-patch samples
// (dists > 0, renderer.py:284-288) into an index list in LDS, (2) the MLP on 32 compacted samples at a
// time with per-sample directions and parameters, (3) the scan, (4) the appended sample
// (color_last, alpha_last; renderer.py:331,339).  Skipped samples have alpha 0, i.e. a transmittance
// factor (1-0)+1e-10 == 1.0f in float32, so leaving them out of the scan is exact.
// ---------------------------------------------------------------------------------------------
constexpr int MAX_INSTANCE_SAMPLES = 4096;

struct InstanceArgs {
    const f32x4 *wstream;
    uint32_t stream_bytes;
    const float *aux;
    const float *rays_d_map, *pts, *t, *dists, *color_last, *alpha_last, *alpha_weight, *params_map, *cone;
    const float *instance_color;
    const int32_t *instance_id;
    const uint8_t *hit;
    float *color_out, *alpha_out;
    int32_t *status;
    int64_t n_rays;
    int n_samples, blur_idx;
    uint32_t flags;
    float patch_scale, density_scale;
    float bkgd[3];
    int32_t *work_counter;   // device scalar, zero at launch: rays are handed out dynamically (their cost varies 0..S/32 batches)
    const int32_t *order;    // the rays, costliest first (inst_*_kernel, ntx_small_kernels.h): claim k marches ray order[k]
    const int32_t *chunk_tab;   // {r1, q0, q1, p1}: the hand-out in chunks of 1 | 2 | 4 | 2 | 1 rays of the cost order (inst_order_kernel)
    const int32_t *count;    // in-patch samples of every ray (inst_count_kernel): the float32 kernel lays a ray's samples out before it compacts them
    int np_in;               // generic family, as RenderArgs
    int8_t pmap[MAX_PARAM_SLOTS];
    // ABI v3: NTX_FLAG_RAW_NOISE (renderer.py:335-337), counter = (marching-sample index, global ray index, 1) as RenderArgs
    float raw_noise_std;
    uint32_t seed_lo, seed_hi, idx_run;
    int64_t idx0, idx_stride;
    int run_hoist;           // 0: every sample is its own run (A/B knob: NERFTEX_NO_DIR_HOIST at ntx_create)
    uint16_t *sidx_scratch;  // [n_workgroups * 4][INST_EXEC_CAP]: every wave's execution list of the bundle of rays in flight (context scratch)
};

// Tail packing.  A ray's in-patch samples fill count / 32 whole batches and leave a TAIL of count % 32 samples; run as a
// batch of its own the tail wastes half a batch per ray on average (6 % of the carpet_instanced workload).  Tails of
// successive rays of a wave are therefore collected (up to 32 samples, up to PEND_MAX rays) and evaluated in ONE batch,
// each ray a SEGMENT of consecutive lanes.  The composite of a segment is written so that its result does not depend on
// where in the batch the segment sits or on what shares the batch: samples outside the segment enter as alpha = 0,
// i.e. as transmittance factors and weights that are exactly 1.0f / 0.0f, the scans associate relative to each lane,
// and every total is read at the segment's last lane.  Which rays meet in a batch depends on the dynamic ray hand-out;
// the image does not.
constexpr int PEND_MAX = 8;

struct InstancePending {          // per wave, in LDS
    uint16_t idx[32];             // marching-sample index of each packed lane
    uint8_t slot[32];             // its segment
    int32_t ray[PEND_MAX];        // per segment: ray, last lane, cone_scale, the accumulator after the ray's whole batches
    int32_t last[PEND_MAX];
    float cone[PEND_MAX];
    float acc[PEND_MAX][5];
};

// one segment of a packed batch: `a` is this sample's alpha for lanes of the segment and 0 elsewhere; e = its last lane
NTX_DEV void composite_segment(RayAccum &ra, float a, const float (&c)[3], int j, int e) {
    const float trans = (1.0f - a) + 1e-10f;                                          // renderer.py:342; 1.0f outside the segment
    float P = trans;
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) {
        const float v = __shfl_up(P, d, 32);
        if (j >= d) P = v * P;
    }
    float E = __shfl_up(P, 1, 32);
    if (j == 0) E = 1.0f;
    const float w = a * (ra.T * E);
    float q[4] = {w * c[0], w * c[1], w * c[2], w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float sacc = q[k];
#pragma unroll
        for (int d = 1; d < 32; d <<= 1) {
            const float v = __shfl_up(sacc, d, 32);
            if (j >= d) sacc = v + sacc;
        }
        q[k] = __shfl(sacc, e, 32);
    }
    ra.c0 += q[0]; ra.c1 += q[1]; ra.c2 += q[2]; ra.a += q[3];
    ra.T *= __shfl(P, e, 32);
}

// ---------------------------------------------------------------------------------------------
// Direction features once per instance RUN.  The instancer hands every marching sample its own direction and parameters
// (instancer.pyx:41-54), but it fills them per (ray, patch instance): rays_d_map = getDir(ray direction, instance) and the
// light direction likewise (instancer.cpp:943-960), the other appearance parameters are the view's constants -- so along the
// run of consecutive in-patch samples of one instance the inputs of C1's direction segment (model.py:96-101, 115) do not
// change; what varies per sample is the position and the texture-mapped GEOMETRY parameter (instancer.cpp:913-918).  As in
// render_kernel<CFG, 1>, the segment W_C1[:dir_map]^T dir_map (328 of 10 648 MFMAs and 41 sin() per lane and batch) is then a
// vector per run: when a ray's samples are compacted, the first sample of every run (inputs bit-different from its
// predecessor's) is flagged; the runs of the next batches of the ray -- as many batches as lead_slots() rows can serve -- are
// evaluated in ONE pass of the segment (leader_rows: lane = run, same bias initialisation, k-order and generator as the
// per-sample evaluation, so the same bits) into rows in LDS, and every batch of the group starts its C1 accumulators from its
// samples' rows (mlp_batch<CFG, 4>).  Nothing is assumed: a batch whose runs do not fit (e.g. per-sample directions), the
// packed tail batches and a blur_idx on an appearance parameter take the per-sample kernel, bit-identical either way.
// ---------------------------------------------------------------------------------------------
constexpr int LEAD_FLAG = 0x8000;       // bit 15 of a compacted sample index (marching indices are < 4096)
constexpr int LEAD_GROUP_MAX = 16;      // batches one group of rows may serve
constexpr int SIDX_WINDOW = 1024;       // entries of a ray's compacted index list kept in LDS (the list itself lives in global scratch)
template <class CFG>
constexpr int lead_slots() { return CFG::CD == 0 || CFG::FLEX != 0 ? 0 : 32; }   // (flex family: the per-sample kernel)   // rows per wave: one per sample of a batch, so EVERY batch can be served

// direction and appearance parameters of marching sample sm as the instancer delivered them (the run flags compare THESE; a
// blur_idx on an appearance parameter scales it per sample, renderer.py:259-262, and then every sample is its own run)
template <class CFG>
NTX_DEV void dir_inputs(const InstanceArgs &a, int64_t sm, SampleIn<CFG::NGEO, CFG::NAPP> &in) {
#pragma unroll
    for (int c = 0; c < 3; ++c) in.dir[c] = a.rays_d_map[3 * sm + c];
    if constexpr (CFG::IPE == 0) {
#pragma unroll
        for (int c = CFG::NGEO; c < CFG::NP; ++c) in.par[c] = param_at<CFG>(a, a.params_map + param_stride<CFG>(a) * sm, c);
    } else {
        const float *pr = a.params_map + CFG::NP_IN * sm;
#pragma unroll
        for (int c = CFG::NGEO; c < CFG::NP; ++c) in.par[c] = pr[c < a.blur_idx ? c : c + 1];
    }
}

// rows[slot] = bias_C1 + W_C1[:dir_map]^T dir_map(in of lane `slot`), all 8 output tiles by this wave: the direction segment
// of run_segment<.., CFG::DS, 8, rec_pass(9)> with the weights loaded straight from the stream
template <class CFG, int NSLOT>
NTX_DEV void leader_rows(__amdgpu_buffer_rsrc_t rsrc, const float *aux, float *rows, const SampleIn<CFG::NGEO, CFG::NAPP> &in, int lane) {
    const int h = lane >> 5, j = lane & 31;
    f32x16 acc[8];
    init_bias<8>(acc, aux, 9, h);
    const uint32_t voff = (uint32_t)lane * 16u;
    static_for<CFG::DS>([&](auto S) {
        constexpr int s = S;
        constexpr uint32_t rec = (uint32_t)(CFG::rec_pass(9) + 2 * s);
        const f32x4 w0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, rec * 1024u, 0));
        const f32x4 w1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, (rec + 1u) * 1024u, 0));
        const float b = dir_feature<CFG::NGEO, CFG::NAPP, s>(in, h);
        static_for<4>([&](auto E) { acc[decltype(E)::value] = mfma32(w0[decltype(E)::value], b, acc[decltype(E)::value]); });
        static_for<4>([&](auto E) { acc[4 + decltype(E)::value] = mfma32(w1[decltype(E)::value], b, acc[4 + decltype(E)::value]); });
    });
    if (j < NSLOT) {   // lane (slot j, half h), register r of tile t -> rows[j][h][16 t + r]  (the layout init_bias_tile_row reads)
        f32x4 *o = reinterpret_cast<f32x4 *>(rows + j * DIR_ROW_STRIDE + h * 128);
        static_for<8>([&](auto T) {
            constexpr int t = T;
            static_for<4>([&](auto Q) {
                constexpr int q = Q;
                o[4 * t + q] = f32x4{acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]};
            });
        });
    }
}

// ---------------------------------------------------------------------------------------------
// The work of a wave: BUNDLES of rays compiled into an EXECUTION LIST (round 3).  Until v14 a wave claimed one ray at a time
// and everything that is paid per ray was paid for ~4 batches of work: the claim -> order -> hit -> dists -> run flags chain of
// exposed memory latencies, one leader_rows pass (the direction segment for 32 columns, whatever the number of runs: ~27 k
// cycles for the 8 runs of an average ray of the bench workload), and a second such pass for every packed tail batch, which
// took per-sample rows and threw the group's rows away.  Now a wave claims up to BUNDLE_MAX rays with ONE atomic, reads their
// facts with one round of loads (lane r = ray r) and lays their samples out in the order they will be evaluated:
//   entry (16 bits) = marching index | slot << 12 | LEAD_FLAG;  slot = one of INST_SLOTS live rays of the wave, 7 = empty lane
//   whole batches of ray 0 | [a packed batch that closed] | whole batches of ray 1 | ...
// A ray's tail (count % 32 samples) joins the OPEN packed batch (kept in LDS across bundles; <= OPEN_MAX rays), which is
// emitted into the list -- padded with empty lanes if need be -- when the next tail does not fit.  The list lives in the
// context's scratch (L2) and is read through the window in LDS; the run flags are computed on the window.  The consumer then
// takes 32 entries at a time, whatever they are: a batch whose 32 lanes belong to one ray is one of its whole batches
// (sequential composite on the ray's accumulator in its slot), anything else is a packed batch of tails, each a segment of
// consecutive lanes that finishes its ray (composite_segment: position-independent, see above).  Groups of run rows span rays
// and packed batches alike.  Every ray still sees exactly the arithmetic it saw before -- its whole batches in order, then its
// tail as a segment -- so the image does not depend on the bundle size, the company or the hand-out
// (test_packed_tails_do_not_depend_on_the_company), and NERFTEX_DEBUG_RUNS bit 3 (bundles of one ray) gives the same bits.
// Bundle size: 4 rays while many are left, 2, then 1 near the end of the cost-ordered hand-out (the ragged end is one
// claim's worth of work per wave), and what the list can hold (rays arrive costliest first: the previous claim's count
// bounds the next ones).
// ---------------------------------------------------------------------------------------------
constexpr int INST_SLOTS = 7;               // live rays of a wave (slot 7 = empty lane of a packed batch)
constexpr int OPEN_MAX = 6;                 // rays in the open packed batch: one slot always stays free for a new ray
constexpr int BUNDLE_MAX = 4;
constexpr int ENTRY_IDX = 0x0fff, ENTRY_SLOT_SHIFT = 12, ENTRY_EMPTY = 7 << ENTRY_SLOT_SHIFT;
static_assert(MAX_INSTANCE_SAMPLES <= ENTRY_IDX + 1, "marching indices take 12 bits of a list entry");
// bundles are sized so that the list fits: 4 rays of <= 1024 + 7 samples (the order is by bins of 8) + 4 emitted packed batches
constexpr int INST_EXEC_CAP = MAX_INSTANCE_SAMPLES + 256;

struct InstanceWave {                       // per wave, in LDS
    uint16_t open_idx[32];                  // the open packed batch: entries of the tails collected so far
    int32_t ray[8];                         // per slot: ray, position behind its last whole batch when it has no tail (else -1),
    int32_t fin_pos[8];                     //           cone_scale, the accumulator (T, r, g, b, a), the appended sample
    float cone[8];
    float acc[8][5];
    float last[8][4];                       // color_last rgb, alpha_last (renderer.py:323-339)
    int32_t b_ray[BUNDLE_MAX], b_n[BUNDLE_MAX];   // the claimed rays not yet compiled: ray, in-patch samples (-1: not hit), cone_scale, appended sample
    float b_cone[BUNDLE_MAX], b_last[BUNDLE_MAX][4];
    int32_t st[8];                          // what only next_bundle needs (kept out of the registers that live across the network):
};                                          // open_n, open_k, open_mask, exhausted, q_i, q_n

template <class CFG>
__global__ __launch_bounds__(256) void instance_kernel(InstanceArgs a) {
    constexpr int NSLOT = lead_slots<CFG>();
    constexpr bool ROWS = NSLOT > 0;            // ParamNerf: C1 always starts from rows; plain Nerf has no C1 (per-sample kernel as before)
    __shared__ __attribute__((aligned(16))) float aux[aux_floats_of<CFG>() + (ROWS ? 4 * NSLOT * DIR_ROW_STRIDE : 4 * pe_keep_floats<CFG>())];
    __shared__ uint16_t win_all[4][SIDX_WINDOW];
    __shared__ InstanceWave tab_all[4];
    __shared__ uint8_t slot_all[4][32];         // row of each sample of the batch in flight
    __shared__ uint16_t lead_all[4][32];        // position (in the execution list) of the group's run leaders
    __shared__ float park_all[4][32][2];        // dists and density weight of the batch in flight (no register survives the network)
    load_aux(aux, a.aux, aux_floats_of<CFG>());
    int lane = threadIdx.x & 63, j = lane & 31;   // (re-read from the hardware at the top of the loop and behind the network: no lane-derived register lives across it)
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int S = a.n_samples;
    float (*park)[2] = park_all[wv];
    // the execution list of the bundle in flight: global scratch of the context (L2-resident, INST_EXEC_CAP entries per wave), read
    // through a window in LDS
    uint16_t *gs = a.sidx_scratch + ((size_t)blockIdx.x * 4 + wv) * INST_EXEC_CAP;
    uint16_t *win = win_all[wv];
    InstanceWave &tab = tab_all[wv];
    float *rows = aux + aux_floats_of<CFG>() + wv * NSLOT * DIR_ROW_STRIDE;
    float *pe_col = ROWS ? nullptr : pe_column<CFG>(aux, wv, lane);
    uint8_t *slots = slot_all[wv];
    uint16_t *lead_pos = lead_all[wv];
    WStream ws;
    ws_prime<RecMap<CFG, ROWS ? 4 : 0>>(ws, a.wstream, a.stream_bytes, lane);
    // runs are looked for unless switched off (A/B knob of the host) or blur_idx scales an appearance parameter per sample;
    // without them every sample is its own run (the rows of a batch are then evaluated for that batch alone: the work of the
    // per-sample kernel, bit-identical results)
    const bool runs_on = ROWS && (a.run_hoist & 1) != 0 && !(a.blur_idx >= CFG::NGEO && CFG::IPE == 0);
    const int group_max = (a.run_hoist & 4) ? 1 : LEAD_GROUP_MAX;   // (development knobs: bit 1 = flags only, bit 2 = one batch per group, bit 3 = bundles of one ray)
    const bool grouped = ROWS && runs_on && !(a.run_hoist & 2);      // (wave-uniform) batches take their rows from groups of runs

    auto lds_sync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    // the appended sample (colour taken as is, alpha_last is an alpha, not a density: renderer.py:323-339) and the store
    auto finish = [&](int64_t ray, const RayAccum &ra, const float *last) {
        const float wl = last[3] * ra.T;
        float out[4] = {ra.c0 + wl * last[0], ra.c1 + wl * last[1], ra.c2 + wl * last[2], ra.a + wl};
        if (a.flags & NTX_FLAG_COMPOSITE_BKGD) {   // renderer.py:351-352
            const float A = out[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) out[c] = out[c] + (1.0f - A) * a.bkgd[c];
        }
        if (lane == 0) {
            a.color_out[3 * ray + 0] = out[0]; a.color_out[3 * ray + 1] = out[1];
            a.color_out[3 * ray + 2] = out[2]; a.alpha_out[ray] = out[3];
            if ((a.flags & NTX_FLAG_CHECK_NUMERICS) && a.status) {
                const float sm_ = out[0] + out[1] + out[2] + out[3];
                if (!(__builtin_fabsf(sm_) <= 3.0e38f)) atomicOr(a.status, 1);
            }
        }
    };

    int p = 0, exec_len = 0;                       // the batch due is entries [p, p + 32) of a list of exec_len
    int win0 = 0;                                  // the window holds list entries [win0, win0 + SIDX_WINDOW)
    int grp_p0 = 0, grp_end = 0, slot_base = 0;    // rows in LDS serve the batches at positions [grp_p0, grp_end); next free row of the group
    if (lane < 8) tab.st[lane] = 0;
    lds_sync();

    auto put = [&](int pos, int e) {
        gs[pos] = (uint16_t)e;
        if (pos < SIDX_WINDOW) win[pos] = (uint16_t)e;
    };
    // run flags of window entries [0, nwin): the first sample of every run = direction / appearance inputs bit-different from the
    // previous entry's (or no previous entry in the window, or an empty lane before it); a flag too many costs a row, never a bit
    auto win_flag = [&](int nwin) {
        if (!runs_on) return;
        constexpr int NV = 3 + CFG::NAPP;
        uint32_t carry[NV] = {};
        bool carry_valid = false;
        for (int k0 = 0; k0 < nwin; k0 += 128) {   // two steps of 64 entries, their gathers in flight together
            uint32_t bits[2][NV];
            bool val[2];
            int ent[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int k = k0 + 64 * u + lane;
                const int e = k < nwin ? win[k] : ENTRY_EMPTY;
                const int sl = (e >> ENTRY_SLOT_SHIFT) & 7;
                val[u] = sl != 7; ent[u] = e;
                SampleIn<CFG::NGEO, CFG::NAPP> din;
                if (val[u]) {
                    dir_inputs<CFG>(a, (int64_t)tab.ray[sl] * S + (e & ENTRY_IDX), din);
#pragma unroll
                    for (int c = 0; c < NV; ++c) bits[u][c] = __builtin_bit_cast(uint32_t, c < 3 ? din.dir[c < 3 ? c : 0] : din.par[CFG::NGEO + (c < 3 ? 0 : c - 3)]);
                } else {
#pragma unroll
                    for (int c = 0; c < NV; ++c) bits[u][c] = 0u;
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int k = k0 + 64 * u + lane;
                int pvld = __shfl_up((int)val[u], 1, 64);
                if (lane == 0) pvld = (int)carry_valid;
                bool diff = !pvld;
#pragma unroll
                for (int c = 0; c < NV; ++c) {
                    uint32_t pv = (uint32_t)__shfl_up((int)bits[u][c], 1, 64);
                    if (lane == 0) pv = carry[c];
                    diff = diff || pv != bits[u][c];
                    carry[c] = (uint32_t)__shfl((int)bits[u][c], 63, 64);
                }
                carry_valid = __shfl((int)val[u], 63, 64) != 0;
                if (val[u] && diff && k < nwin) win[k] = (uint16_t)(ent[u] | LEAD_FLAG);
            }
        }
        lds_sync();
    };
    auto win_load = [&](int p0) {
        win0 = p0;
        const int nwin = exec_len - p0 < SIDX_WINDOW ? exec_len - p0 : SIDX_WINDOW;
        for (int k = lane; k < nwin; k += 64) win[k] = gs[p0 + k];
        lds_sync();
        win_flag(nwin);
    };
    // the open packed batch becomes entries [at, at + 32) of the list (empty lanes behind its samples)
    auto emit_open = [&](int at, int &open_n, int &open_k, uint32_t &open_mask) {
        lds_sync();
        if (lane < 32) put(at + lane, lane < open_n ? (int)tab.open_idx[lane] : ENTRY_EMPTY);
        open_n = 0; open_k = 0; open_mask = 0;
    };

    // ---- compile the next bundle of rays into the list (claiming a chunk when the queue is empty); false = nothing is left at all
    auto next_bundle = [&]() -> bool {
        p = 0; exec_len = 0; grp_p0 = grp_end = 0;
        int open_n = __builtin_amdgcn_readfirstlane(tab.st[0]), open_k = __builtin_amdgcn_readfirstlane(tab.st[1]);   // the open packed batch: samples, rays,
        uint32_t open_mask = (uint32_t)__builtin_amdgcn_readfirstlane(tab.st[2]);                                     // the slots of those rays
        bool exhausted = __builtin_amdgcn_readfirstlane(tab.st[3]) != 0;
        int q_i = __builtin_amdgcn_readfirstlane(tab.st[4]), q_n = __builtin_amdgcn_readfirstlane(tab.st[5]);         // the claimed rays still to compile
        const bool fast = (S & 3) == 0 && (reinterpret_cast<uintptr_t>(a.dists) & 15) == 0;   // rows of dists as 16-byte loads
        const unsigned long long lt = (1ull << lane) - 1ull;
        bool any = true;
        while (exec_len == 0) {
            if (q_i == q_n) {
                if (exhausted) {
                    if (open_n == 0) { any = false; break; }
                    emit_open(0, open_n, open_k, open_mask);
                    exec_len = 32;
                    break;
                }
                // The hand-out: claim c of the shared counter is a CHUNK of the cost order -- single rays, pairs, fours, pairs, single
                // rays, a closed form of c over the ranks inst_order_kernel chose (ntx_small_kernels.h: a chunk costs at most half a
                // wave's share, and the end of the hand-out is single cheap rays).  (A wave that sized its claim from what it saw at
                // its previous one overshot: +5 % at 16 384 rays.)
                const int64_t r1 = a.chunk_tab[0], q0 = a.chunk_tab[1], q1 = a.chunk_tab[2], p1 = a.chunk_tab[3];
                const int64_t c1 = r1, c2 = c1 + (q0 - r1) / 2, c3 = c2 + (q1 - q0) / 4, c4 = c3 + (p1 - q1) / 2, n_chunks = c4 + (a.n_rays - p1);
                int r32 = 0;
                if (lane == 0) r32 = atomicAdd(a.work_counter, 1);
                const int64_t c = (int64_t)__builtin_amdgcn_readfirstlane(r32);
                if (c >= n_chunks) { exhausted = true; continue; }
                const int64_t first = c < c1 ? c : c < c2 ? r1 + 2 * (c - c1) : c < c3 ? q0 + 4 * (c - c2) : c < c4 ? q1 + 2 * (c - c3) : p1 + (c - c4);
                const int K = c < c1 ? 1 : c < c2 ? 2 : c < c3 ? 4 : c < c4 ? 2 : 1;
                // the rays' facts, lane r = ray r: one round of loads for the chunk.  The number of in-patch samples differs from ray
                // to ray (0 .. S), so a static ray -> wave map leaves waves idle at the end (19 % on the carpet_instanced bench workload)
                if (lane < K) {
                    const int64_t ray = a.order[first + lane];
                    tab.b_ray[lane] = (int32_t)ray;
                    tab.b_n[lane] = a.hit[ray] ? a.count[ray] : -1;
                    tab.b_cone[lane] = a.cone ? a.cone[ray] : 0.0f;
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) tab.b_last[lane][cc] = a.color_last[3 * ray + cc];
                    tab.b_last[lane][3] = a.alpha_last[ray];
                }
                lds_sync();
                q_i = 0; q_n = K;
            }
            int pos = 0;
            uint32_t used = open_mask;
            const int q0 = q_i;
            bool stop = false;
            // the rows of all queued rays in flight together when a row is one step of four 16-byte loads (S <= 1024)
            const bool pre = fast && S <= 1024;
            f32x4 dv[BUNDLE_MAX][4];
            static_for<BUNDLE_MAX>([&](auto R) {
                constexpr int r = R;
                const bool want = pre && q0 + r < q_n && __builtin_amdgcn_readfirstlane(tab.b_n[(q0 + r) & (BUNDLE_MAX - 1)]) > 0;
                const f32x4 *d4 = reinterpret_cast<const f32x4 *>(a.dists + (int64_t)__builtin_amdgcn_readfirstlane(tab.b_ray[(q0 + r) & (BUNDLE_MAX - 1)]) * S);
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int i = 64 * u + lane; dv[r][u] = want && i < (S >> 2) ? d4[i] : f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
            });
            static_for<BUNDLE_MAX>([&](auto R) {
                constexpr int r = R;
                if (stop || q0 + r >= q_n) return;
                const int qi = q0 + r;
                const int64_t ray = (int64_t)__builtin_amdgcn_readfirstlane(tab.b_ray[qi]);
                const int n = __builtin_amdgcn_readfirstlane(tab.b_n[qi]);
                if (n < 0) {   // renderer.py:265-272, 313-314: stays 0, also under composite_bkgd
                    if (lane < 3) a.color_out[3 * ray + lane] = 0.0f;
                    if (lane == 3) a.alpha_out[ray] = 0.0f;
                    q_i = qi + 1;
                    return;
                }
                if (n == 0) {   // hit, but no sample inside a patch: the appended sample alone
                    finish(ray, RayAccum{1.0f, 0.0f, 0.0f, 0.0f, 0.0f}, tab.b_last[qi]);
                    q_i = qi + 1;
                    return;
                }
                const int nfull = n >> 5, rem = n & 31;
                // no slot left, or the list full: the ray stays queued for the next bundle (the first ray of a bundle always fits)
                if (__popc(used) >= INST_SLOTS || pos + 32 * nfull + 64 > INST_EXEC_CAP) { stop = true; return; }
                if (rem > 0 && (open_n + rem > 32 || open_k == OPEN_MAX)) { emit_open(pos, open_n, open_k, open_mask); pos += 32; }
                const int sl = __builtin_ctz(~used);
                used |= 1u << sl;
                const int base = pos, tbase = open_n;
                pos += 32 * nfull;
                if (lane == 0) {
                    tab.ray[sl] = (int32_t)ray; tab.fin_pos[sl] = rem == 0 ? pos : -1; tab.cone[sl] = tab.b_cone[qi];
                    tab.acc[sl][0] = 1.0f; tab.acc[sl][1] = 0.0f; tab.acc[sl][2] = 0.0f; tab.acc[sl][3] = 0.0f; tab.acc[sl][4] = 0.0f;
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) tab.last[sl][cc] = tab.b_last[qi][cc];
                }
                // compaction: in-patch sample number k of the ray (marching index i) -> its place in the list / the open batch
                auto place = [&](int k, int i) {
                    if (k >= n) return;   // (n = what inst_count_kernel counted on the same row)
                    const int e = i | (sl << ENTRY_SLOT_SHIFT);
                    if (k < 32 * nfull) put(base + k, e);
                    else tab.open_idx[tbase + k - 32 * nfull] = (uint16_t)e;
                };
                auto place4 = [&](const f32x4 &v, int i, int &cnt) {   // marching indices i .. i + 3 of this lane, lanes in index order
                    unsigned long long m[4];
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) m[cc] = __ballot(v[cc] > 0.0f);
                    int k = cnt + __popcll(m[0] & lt) + __popcll(m[1] & lt) + __popcll(m[2] & lt) + __popcll(m[3] & lt);
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) if (v[cc] > 0.0f) { place(k, i + cc); ++k; }
                    cnt += __popcll(m[0]) + __popcll(m[1]) + __popcll(m[2]) + __popcll(m[3]);
                };
                const float *drow = a.dists + ray * S;
                int cnt = 0;
                if (pre) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) place4(dv[r][u], 4 * (64 * u + lane), cnt);
                } else if (fast) {
                    const f32x4 *d4 = reinterpret_cast<const f32x4 *>(drow);
                    const int q = S >> 2;
                    for (int i0 = 0; i0 < q; i0 += 256) {   // 4 loads of 16 bytes in flight, then their 16 ballots
                        f32x4 w[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) { const int i = i0 + 64 * u + lane; w[u] = i < q ? d4[i] : f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
#pragma unroll
                        for (int u = 0; u < 4; ++u) place4(w[u], 4 * (i0 + 64 * u + lane), cnt);
                    }
                } else {
                    for (int base0 = 0; base0 < S; base0 += 512) {   // 8 independent loads in flight, then their 8 ballots
                        float w[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) { const int i = base0 + 64 * u + lane; w[u] = i < S ? drow[i] : 0.0f; }
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const bool v = w[u] > 0.0f;
                            const unsigned long long m = __ballot(v);
                            if (v) place(cnt + __popcll(m & lt), base0 + 64 * u + lane);
                            cnt += __popcll(m);
                        }
                    }
                }
                if (rem > 0) { open_n += rem; ++open_k; open_mask |= 1u << sl; }
                q_i = qi + 1;
            });
            exec_len = pos;
        }
        if (lane == 0) {
            tab.st[0] = open_n; tab.st[1] = open_k; tab.st[2] = (int32_t)open_mask; tab.st[3] = exhausted ? 1 : 0; tab.st[4] = q_i; tab.st[5] = q_n;
        }
        lds_sync();
        if (!any) return false;
        win0 = 0;
        win_flag(exec_len < SIDX_WINDOW ? exec_len : SIDX_WINDOW);
        return true;
    };

    // what a sample brings from memory, as loaded: the gathers of batch b + 1 are issued in the middle of batch b's network (MID of
    // mlp_batch_tuned: behind the skip layer's position segment, where batch b's inputs die) and land under the rest of it -- at the
    // head of a batch they cost an exposed HBM latency per batch.  The shipped ParamNerf families only: the others have no
    // registers to spare.  (NERFTEX_DEBUG_RUNS bit 4 switches it off; same bits either way.)
    struct RawIn { float pos[3], dir[3], par[CFG::NP > 0 ? CFG::NP : 1], t, dist, wgt, blurp; };
    constexpr bool PREFETCH = ROWS && CFG::GEN == 0 && CFG::IPE == 0 && CFG::FLEX == 0;
    const bool prefetch_on = PREFETCH && !(a.run_hoist & 16);
    auto gather = [&](int at, RawIn &r) {   // lane j's sample of the batch at list position `at` (inside the window)
        int e = win[at - win0 + j];
        if (((e >> ENTRY_SLOT_SHIFT) & 7) == 7) e = win[at - win0];   // an empty lane of a packed batch shadows lane 0's, which is never empty
        const int64_t sm = (int64_t)tab.ray[(e >> ENTRY_SLOT_SHIFT) & 7] * S + (e & ENTRY_IDX);
#pragma unroll
        for (int c = 0; c < 3; ++c) { r.pos[c] = a.pts[3 * sm + c]; r.dir[c] = a.rays_d_map[3 * sm + c]; }
        r.t = 0.0f; r.blurp = 0.0f;
        if constexpr (CFG::IPE == 0) {
#pragma unroll
            for (int c = 0; c < CFG::NP; ++c) r.par[c] = param_at<CFG>(a, a.params_map + param_stride<CFG>(a) * sm, c);
            if (a.blur_idx >= 0) r.t = a.t[sm];
        } else {
            const float *pr = a.params_map + CFG::NP_IN * sm;
#pragma unroll
            for (int c = 0; c < CFG::NP; ++c) r.par[c] = pr[c < a.blur_idx ? c : c + 1];
            r.blurp = pr[a.blur_idx];
            r.t = a.t[sm];
        }
        r.dist = a.dists[sm];
        r.wgt = a.alpha_weight ? a.alpha_weight[sm] : 1.0f;
    };
    RawIn raw;
    bool have_next = false;

    for (;;) {
        lane = fresh_lane_id(); j = lane & 31;
        if (p >= exec_len) { have_next = false; if (!next_bundle()) break; }
        // a group never looks past the window: slide it when the batches a new group may cover would
        if ((grouped ? p >= grp_end && p + 32 * LEAD_GROUP_MAX > win0 + SIDX_WINDOW : p + 32 > win0 + SIDX_WINDOW) && exec_len > win0 + SIDX_WINDOW)
            win_load(p);

        // ---- this lane's sample: its inputs were gathered under the previous batch's network, or are now
        if (!have_next) gather(p, raw);
        SampleIn<CFG::NGEO, CFG::NAPP> in;
        {
            int e = win[p - win0 + j];
            if (((e >> ENTRY_SLOT_SHIFT) & 7) == 7) e = win[p - win0];
            const float cone_l = tab.cone[(e >> ENTRY_SLOT_SHIFT) & 7];
#pragma unroll
            for (int c = 0; c < 3; ++c) { in.pos[c] = raw.pos[c]; in.dir[c] = raw.dir[c]; }
            if constexpr (CFG::IPE == 0) {
                in.cov[0] = in.cov[1] = in.cov[2] = 0.0f;
#pragma unroll
                for (int c = 0; c < CFG::NP; ++c) {
                    float pv = raw.par[c];
                    if (c == a.blur_idx) pv = pv * (cone_l * raw.t / a.patch_scale);                 // renderer.py:259-262
                    in.par[c] = pv;
                }
            } else {
                // MipInstanceRenderer (renderer.py:510-540, 570-587): radius = blur parameter * cone_scale / patch_scale,
                // spliced out of the parameters; gaussian with mu = t and (sic) hw = dists; the mean is the sample point
                float t_mean, t_var, r_var;
                cone_moments(raw.t, raw.dist, raw.blurp * cone_l / a.patch_scale, t_mean, t_var, r_var);
                cone_cov(t_var, r_var, in.dir, in.cov);
#pragma unroll
                for (int c = 0; c < CFG::NP; ++c) in.par[c] = raw.par[c];
            }
            // what the composite needs of this sample is fetched with the rest (one exposed latency per batch instead of two) and parked
            if (lane < 32) {
                park[j][0] = raw.dist;
                park[j][1] = a.alpha_weight ? raw.wgt * a.density_scale : a.density_scale;   // renderer.py:300
            }
        }
        lds_sync();

        // ---- the rows this batch starts C1 from
        if constexpr (ROWS) {
            if (grouped) {
                if (p >= grp_end) {
                    // new group from this batch: row 0 = the run in progress at its first sample, then every flagged sample of as many
                    // batches as the 32 rows can serve (at least this one)
                    int nlead = 1, covered = 0;
                    if (lane == 0) lead_pos[0] = (uint16_t)p;
                    for (int pp = p; pp < exec_len && pp + 32 <= win0 + SIDX_WINDOW && covered < group_max; pp += 32) {
                        const bool f = (win[pp - win0 + j] & LEAD_FLAG) != 0 && !(pp == p && j == 0);
                        const uint32_t m = (uint32_t)__ballot(f);              // lanes j and j + 32 agree: the low word has it
                        const int d = __popc(m);
                        if (nlead + d > NSLOT) break;
                        if (f && lane < 32) lead_pos[nlead + __popc(m & ((1u << j) - 1u))] = (uint16_t)(pp + j);
                        nlead += d; ++covered;
                    }
                    grp_p0 = p; grp_end = p + 32 * covered; slot_base = 0;
                    lds_sync();
                    SampleIn<CFG::NGEO, CFG::NAPP> lin = in;   // (only dir and the appearance parameters are read)
                    const int le = win[lead_pos[j < nlead ? j : 0] - win0];
                    dir_inputs<CFG>(a, (int64_t)tab.ray[(le >> ENTRY_SLOT_SHIFT) & 7] * S + (le & ENTRY_IDX), lin);
                    leader_rows<CFG, NSLOT>(ws.rsrc, aux, rows, lin, lane);
                }
                const bool f = (win[p - win0 + j] & LEAD_FLAG) != 0 && !(p == grp_p0 && j == 0);
                const uint32_t m = (uint32_t)__ballot(f);
                if (lane < 32) slots[j] = (uint8_t)(slot_base + __popc(m & ((2u << j) - 1u)));
                slot_base += __popc(m);
            } else {
                // every sample its own row (runs switched off)
                if (lane < 32) slots[j] = (uint8_t)j;
                leader_rows<CFG, NSLOT>(ws.rsrc, aux, rows, in, lane);
            }
            lds_sync();
        }

        float sigma, rgb_raw[3];
        if constexpr (ROWS) {
            // (wave-uniform) the next batch is known and inside the window: its gathers go out under this batch's network
            const bool want_next = prefetch_on && p + 32 < exec_len && p + 64 <= win0 + SIDX_WINDOW;
            mlp_batch<CFG, 4, false>(in, ws, aux, lane, sigma, rgb_raw, rows, nullptr, slots, [&]() {
                if constexpr (PREFETCH) {
                    if (want_next) { lane = fresh_lane_id(); j = lane & 31; gather(p + 32, raw); }
                }
            });
            have_next = want_next;
        } else {
            mlp_batch<CFG>(in, ws, aux, lane, sigma, rgb_raw, nullptr, pe_col);
        }

        // ---- behind the network nothing of the above is alive: the lane finds its sample again
        lane = fresh_lane_id(); j = lane & 31;
        int e2 = win[p - win0 + j];
        const bool valid2 = ((e2 >> ENTRY_SLOT_SHIFT) & 7) != 7;
        if (!valid2) e2 = win[p - win0];
        const int sl2 = (e2 >> ENTRY_SLOT_SHIFT) & 7;
        const int64_t ray2 = tab.ray[sl2];
        const int idx2 = e2 & ENTRY_IDX;
        const int64_t sm2 = ray2 * S + idx2;
        const float wgt = park[j][1], dist_l = park[j][0];
        sigma = sigma * wgt;
        float col[3];
        if (a.instance_color) {                                                                // :306-307, 322-323
            const int id = a.instance_id[sm2];
#pragma unroll
            for (int c = 0; c < 3; ++c) col[c] = a.instance_color[3 * id + c];
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) col[c] = (a.flags & NTX_FLAG_MAP_EXR) ? elu1f_(rgb_raw[c]) : sigmoidf_(rgb_raw[c]);
        }
        if (a.flags & NTX_FLAG_RAW_NOISE)                                                         // :335-337
            sigma += a.raw_noise_std * normal01(global_index(a.idx0, a.idx_run, a.idx_stride, ray2), idx2, a.seed_lo, a.seed_hi);
        const float al = valid2 ? 1.0f - expf(-__builtin_fmaxf(sigma, 0.0f) * dist_l / a.patch_scale) : 0.0f;   // :339
        const int sl0 = __builtin_amdgcn_readfirstlane(sl2);
        const bool whole = (uint32_t)__ballot(valid2 && sl2 == sl0) == 0xffffffffu;   // 32 lanes of one ray: tails are shorter
        if (whole) {
            RayAccum ra{tab.acc[sl0][0], tab.acc[sl0][1], tab.acc[sl0][2], tab.acc[sl0][3], tab.acc[sl0][4]};
            composite_core<32>(ra, al, col, true, j, nullptr);
            if (p + 32 == tab.fin_pos[sl0]) {
                finish(tab.ray[sl0], ra, tab.last[sl0]);
            } else if (lane == 0) {
                tab.acc[sl0][0] = ra.T; tab.acc[sl0][1] = ra.c0; tab.acc[sl0][2] = ra.c1; tab.acc[sl0][3] = ra.c2; tab.acc[sl0][4] = ra.a;
            }
        } else {
            uint32_t todo = (uint32_t)__ballot(valid2);
            while (todo) {   // the segments, in lane order; each is the tail of its ray and finishes it
                const int l0 = __builtin_ctz(todo);
                const int sk = __shfl(sl2, l0, 64);
                const uint32_t seg = (uint32_t)__ballot(valid2 && sl2 == sk);
                RayAccum rk{tab.acc[sk][0], tab.acc[sk][1], tab.acc[sk][2], tab.acc[sk][3], tab.acc[sk][4]};
                composite_segment(rk, (valid2 && sl2 == sk) ? al : 0.0f, col, j, 31 - __builtin_clz(seg));
                finish(tab.ray[sk], rk, tab.last[sk]);
                todo &= ~seg;
            }
        }
        lds_sync();
        p += 32;
    }
}

// ---------------------------------------------------------------------------------------------
// stand-alone MLP: samples are independent, 32 per wave
// ---------------------------------------------------------------------------------------------
struct MlpArgs {
    const f32x4 *wstream;
    uint32_t stream_bytes;
    const float *aux;
    const float *pos, *dirs, *params;
    float *color_out, *sigma_out;
    int64_t m;
    int np_in;               // generic family, as RenderArgs
    int8_t pmap[MAX_PARAM_SLOTS];
};

template <class CFG>
__global__ __launch_bounds__(256) void mlp_kernel(MlpArgs a) {
    __shared__ __attribute__((aligned(16))) float aux[aux_floats_of<CFG>() + 4 * pe_keep_floats<CFG>()];
    load_aux(aux, a.aux, aux_floats_of<CFG>());
    const int lane = threadIdx.x & 63, j = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    const int nwaves = gridDim.x * 4;
    WStream ws;
    ws_prime<RecMap<CFG, 0>>(ws, a.wstream, a.stream_bytes, lane);
    const int64_t nbatch = (a.m + 31) >> 5;
    for (int64_t b = wave; b < nbatch; b += nwaves) {
        const int64_t m = b * 32 + j;
        const bool valid = m < a.m;
        const int64_t mc = valid ? m : a.m - 1;
        SampleIn<CFG::NGEO, CFG::NAPP> in;
#pragma unroll
        for (int k = 0; k < 3; ++k) {   // IPE models take pos[M,6] = (mean, diagonal covariance)
            in.pos[k] = a.pos[(CFG::IPE ? 6 : 3) * mc + k];
            in.cov[k] = CFG::IPE ? a.pos[6 * mc + 3 + k] : 0.0f;
            in.dir[k] = a.dirs[3 * mc + k];
        }
#pragma unroll
        for (int k = 0; k < CFG::NP; ++k)   // the MODEL's parameters, [M, NP] (an IPE model's caller has spliced the blur parameter out)
            in.par[k] = param_at<CFG>(a, a.params + (CFG::GEN != 0 ? a.np_in : CFG::NP) * mc, k);
        float sigma, raw[3];
        mlp_batch<CFG>(in, ws, aux, lane, sigma, raw, nullptr, pe_column<CFG>(aux, threadIdx.x >> 6, lane));
        if (valid && lane < 32) {
            a.color_out[3 * m + 0] = raw[0]; a.color_out[3 * m + 1] = raw[1]; a.color_out[3 * m + 2] = raw[2];
            a.sigma_out[m] = sigma;
        }
    }
}

}  // namespace ntx
