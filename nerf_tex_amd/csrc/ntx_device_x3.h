// ntx_device_x3.h -- opt-in "fp16x3" precision of the fused render kernels (gfx950).
//
// Every Dense layer is evaluated with the 16-bit matrix cores on a 3-term split of both operands into IEEE halves,
//     x*w ~= hi(x)*hi(w) + hi(x)*lo(w) + lo(x)*hi(w),      hi = fp16(v), lo = fp16(v - hi),   float32 accumulate,
// (v_mfma_f32_32x32x16_f16: 16x the MAC rate of the f32 MFMA, on a pipe the VALU does not share; it keeps subnormal
// inputs, checked on the hardware, so lo parts below 6e-5 are not flushed).  hi + lo carries 22 mantissa bits and each
// product of two halves is exact in float32, so what is lost is the lo*lo term and the rounding of lo: ~2^-22 relative
// per product.  Measured: 2.5e-6 rel-Linf from the float32 restatement of the reference on the plumbing image
// (tools/emulate_bf16_split.py fp16; the float32 kernel itself: 1.7e-6).  The same scheme with bfloat16 halves (what this
// file did first) is 10x less accurate at the same cost.  Range: |activation|, |weight| <= 65504, else inf -> flagged by
// NTX_FLAG_CHECK_NUMERICS.  The exact-f32 kernel of ntx_device.h stays the default.
//
// Structure: one wave64 = 32 samples with activations in registers, as in the f32 kernel -- but the 4 waves of a
// workgroup SHARE one copy of the weight stream through an LDS ring.  (Every wave streaming its own copy from L2, the
// f32 kernel's structure, tops out at the L1/TA's 64 B/clk/CU: 84% TA-busy at 62% MFMA utilisation, measured; the shared
// ring moves that traffic to the LDS -- tools/ubench/fp16x3_stream.hip: 6.44 vs 8.3 us per 256x256 layer.)
//   * the stream is cut into STAGES of 16 records (16 KiB = one k16-step of an 8-tile layer); the ring holds 4 stages;
//   * each wave fetches its quarter of a stage with 4 LDS-DMA loads (buffer_load_dwordx4 ... lds: L2 -> LDS, no VGPRs),
//     three stages ahead of the one being multiplied;
//   * one s_waitcnt vmcnt + s_barrier per stage: at the end of stage s every wave has its quarter of stage s+2 landed and
//     has finished reading stage s, whose slot the next stage's fetch overwrites;
//   * A operands come back with ds_read_b128, one MFMA pair-group (4 records) ahead.
// The waves of a workgroup therefore run in lockstep: the kernel walks a COMPACTED list of hit rays (built by
// compact_hits_kernel, ntx_small_kernels.h) so that every wave has the same trip count.
// A k16-step of the f16 MFMA is 8 consecutive k2-steps of the f32 layout (ntx_layout.h): element e of lane half h in
// k16-step u is the feature hidden_row(8u+e, h) / pos_row(8u+e, h) / dir_row(8u+e, h), so the same accumulator-register
// -> next-layer-B-operand identity holds, now with a bias+ReLU+split+pack between.
#pragma once

#include "ntx_device.h"

namespace ntx {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) char lds_char;

constexpr int vmcnt_imm(int n) { return (n & 0xF) | (7 << 4) | (0xF << 8) | ((n >> 4) << 14); }   // s_waitcnt vmcnt(n) only

struct WShared {
    i32x4 rsrc;           // buffer descriptor of the stream (SGPRs)
    uint32_t voff;        // lane * 16: global side of the DMA
    uint32_t woff;        // (wave in workgroup) * 4096: this wave's quarter of every stage
    uint32_t ring;        // LDS byte address of the ring: NSTAGE16 x 16 KiB
    const lds_char *lane16;   // ring + lane * 16: LDS side of the reads
    half8 a[4];          // A operands of the upcoming MFMA pair-group: tile 2g hi, lo, tile 2g+1 hi, lo
};

// this wave's quarter of stage ST (modulo the padded stream) -> its ring slot: four LDS-DMA loads, M0 = LDS address of
// lane 0's 16 bytes (the hardware adds lane * 16).  Inline asm on purpose: given the intrinsic, hipcc's waitcnt pass
// assumes every later ds_read may alias the DMA's destination and puts s_waitcnt vmcnt(0) in front of it, which
// serialises fetch and compute; the landing of a stage is ordered by stage_end() instead.
template <int ST, int NST>
NTX_DEV void stage_fetch(const WShared &ws) {
    constexpr int st = ST % NST, slot = st % NSTAGE16;
    static_assert(NST % NSTAGE16 == 0, "whole ring turns per batch");
    uint32_t soff;
    asm volatile("s_add_u32 m0, %[woff], %[lds]\n\t"
                 "s_add_u32 %[soff], %[woff], %[goff]\n\t"
                 "buffer_load_dwordx4 %[voff], %[rsrc], %[soff] offen lds\n\t"
                 "buffer_load_dwordx4 %[voff], %[rsrc], %[soff] offen offset:1024 lds\n\t"
                 "buffer_load_dwordx4 %[voff], %[rsrc], %[soff] offen offset:2048 lds\n\t"
                 "buffer_load_dwordx4 %[voff], %[rsrc], %[soff] offen offset:3072 lds"
                 : [soff] "=&s"(soff)
                 : [woff] "s"(ws.woff), [lds] "s"(ws.ring + slot * (STAGE16 * 1024)), [goff] "s"(st * (STAGE16 * 1024)),
                   [voff] "v"(ws.voff), [rsrc] "s"(ws.rsrc)
                 : "memory", "scc");
}

// end of a stage: my quarter of the stage after next has landed (only the newest fetch may still be in flight), and
// everybody is done reading this one
NTX_DEV void stage_end() {
#ifndef NTX_X3_EXPERIMENT_NO_VMWAIT     // timing experiments only (results are then wrong): DESIGN.md section 4.1b
    __builtin_amdgcn_s_waitcnt(vmcnt_imm(4));
#endif
#ifndef NTX_X3_EXPERIMENT_NO_BARRIER
    __builtin_amdgcn_s_barrier();
#endif
}

// records REC .. REC+3 (one pair-group; never straddles a stage) -> registers
template <int REC, int NST>
NTX_DEV void read_group(const WShared &ws, half8 (&n)[4]) {
    constexpr int rec = REC % (NST * STAGE16), slot = (rec / STAGE16) % NSTAGE16, r0 = rec % STAGE16;
    static_assert(r0 % 4 == 0, "pair-groups are 4 records");
    const lds_char *p = ws.lane16 + (slot * STAGE16 + r0) * 1024;
    static_for<4>([&](auto K) { n[K] = *reinterpret_cast<const __attribute__((address_space(3))) half8 *>(p + decltype(K)::value * 1024); });
}

template <int NST>
NTX_DEV void ws_prime(WShared &ws, const void *base, uint32_t stream_bytes, lds_char *ring, int lane, int wave_in_wg) {
    // raw buffer descriptor: base, stride 0, num_records = bytes, untyped dword format (as make_buffer_rsrc(.., 0x00020000))
    const uint64_t b = (uint64_t)base;
    ws.rsrc = i32x4{__builtin_amdgcn_readfirstlane((int)(uint32_t)b), __builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32) & 0xffff),
                    __builtin_amdgcn_readfirstlane((int)stream_bytes), 0x00020000};
    ws.voff = (uint32_t)lane * 16u;
    ws.woff = (uint32_t)wave_in_wg * 4096u;
    ws.ring = (uint32_t)(uintptr_t)ring;
    ws.lane16 = ring + lane * 16;
    stage_fetch<0, NST>(ws);
    stage_fetch<1, NST>(ws);
    stage_fetch<2, NST>(ws);
    stage_end();            // stages 0 and 1 are in
    read_group<0, NST>(ws, ws.a);
}

// B operand of one k16-step: 8 features per lane, split
struct B16 {
    half8 hi, lo;
};

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// two floats -> two IEEE halves in one dword, round to nearest even (v_cvt_f16_f32 x2 + v_pack_b32_f16; NOT v_cvt_pkrtz).
// Values beyond 65504 become inf and poison the sample, which NTX_FLAG_CHECK_NUMERICS reports.
NTX_DEV uint32_t pack2(float a, float b) {
    const half2_t v = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(uint32_t, v);
}
NTX_DEV float lo_f32(uint32_t w) { return (float)__builtin_bit_cast(half2_t, w)[0]; }
NTX_DEV float hi_f32(uint32_t w) { return (float)__builtin_bit_cast(half2_t, w)[1]; }

// relu as an integer max: one v_max_i32, no canonicalising pre-op (fmaxf on an MFMA result costs two instructions);
// +NaN stays NaN, -NaN and -0 become +0
NTX_DEV float relu1(float x) {
    const int i = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, i > 0 ? i : 0);
}

NTX_DEV f32x16 mfma16(half8 a, half8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

// geometry of the k16 stream (host packer: pack16 in nerftex.hip).  Within a pass the hidden segment comes FIRST and the
// encoder segment second (the order of the k-summation is free): the activations are converted just in time behind the
// hidden segment's own MFMAs and never need to be held as a whole.
template <class CFG, bool WD = false>   // WD: with the direction segment of C1 in the stream (instanced kernel)
struct Cfg16 {
    static constexpr int PS16 = steps16(CFG::PS), DS16 = steps16(CFG::DS), HS16 = HSTEPS / 8;
    static constexpr int rec_pass(int li) {   // first record of hidden pass li (1..8 = L1..L7, F; 9 = C1)
        return PS16 * 16 + (li - 1) * HS16 * 16 + (li > SKIP + 1 ? PS16 * 16 : 0) + (CFG::CD && WD && li > 9 ? DS16 * 16 : 0);
    }
    static constexpr int REC_C2 = rec_pass(9 + (CFG::CD ? 1 : 0));
    static constexpr int REC_END = REC_C2 + (CFG::CD ? 0 : DS16 * 8) + HS16 * 8;
    static constexpr int REC_PAD = stream16_padded(CFG::NGEO, CFG::NAPP, CFG::CD, WD, CFG::IPE);
    static constexpr int NST = REC_PAD / STAGE16;   // stages per batch
    static_assert(REC_END == stream16_records(CFG::NGEO, CFG::NAPP, CFG::CD, WD, CFG::IPE), "stream bookkeeping");
    static_assert(REC_END % 8 == 0 && REC_PAD % (STAGE16 * NSTAGE16) == 0, "pair-groups and ring turns");
};

// ---- B-operand generators.  The B operand of k16-step U is produced by 12 PIECES of VALU work, piece<U, Q>(), which
// run_segment16 places behind the MFMA pairs of step U-1 (the 16-bit MFMA pipe and the VALU are separate: a pair of MFMAs
// hides ~8 VALU instructions); value<U>() hands over the finished operand.
struct Words16 {
    uint32_t hw[4], lw[4];
    NTX_DEV B16 get() const {
        const u32x4 h = {hw[0], hw[1], hw[2], hw[3]}, l = {lw[0], lw[1], lw[2], lw[3]};
        return B16{__builtin_bit_cast(half8, h), __builtin_bit_cast(half8, l)};
    }
};

// activations: bias is already in the accumulators; act + split + pack.  The 8 values of k16-step U are registers
// 8(U&1)..+7 of tile U/2 = features hidden_row(8U+e, half).  Pair p = Q/3 in three stages Q%3: read + act (+ alpha head,
// model.py:111, on the float32 value), hi word + residuals, lo word.
template <bool RELU, bool ALPHA>
struct ConvGen {
    const f32x16 (&prev)[8];
    const float *aux;
    int h;
    float &sig;
    Words16 w;
    float x0, x1, r0, r1;
    template <int U, int Q>
    NTX_DEV void piece() {
        constexpr int p = Q / 3, t = Q % 3, reg = 8 * (U & 1) + 2 * p;
        if constexpr (t == 0) {
            // read the two accumulator registers HERE: left to the register allocator, all 128 values of the drained set
            // are copied to VGPRs at the top of the pass and stay live through it, which tips the kernel into spilling
            asm volatile("v_accvgpr_read_b32 %0, %2\n\tv_accvgpr_read_b32 %1, %3"
                         : "=&v"(x0), "=&v"(x1) : "a"(prev[U >> 1][reg]), "a"(prev[U >> 1][reg + 1]));
            if constexpr (RELU) { x0 = relu1(x0); x1 = relu1(x1); }
            if constexpr (ALPHA) {
                sig = __builtin_fmaf(x0, aux[aux_alpha_off() + h * 128 + 8 * U + 2 * p], sig);
                sig = __builtin_fmaf(x1, aux[aux_alpha_off() + h * 128 + 8 * U + 2 * p + 1], sig);
            }
        } else if constexpr (t == 1) {
            w.hw[p] = pack2(x0, x1);
            r0 = x0 - lo_f32(w.hw[p]);
            r1 = x1 - hi_f32(w.hw[p]);
        } else {
            w.lw[p] = pack2(r0, r1);
        }
    }
    template <int U>
    NTX_DEV B16 value() const { return w.get(); }
};

// encoders: pieces 0..7 evaluate one feature each, pieces 8..11 split and pack a pair
template <class CFG, bool DIR>
struct EncGen16 {
    const SampleIn<CFG::NGEO, CFG::NAPP> &in;
    int h;
    Words16 w;
    float v[8];
    template <int U, int Q>
    NTX_DEV void piece() {
        if constexpr (Q < 8) {
            constexpr int s = 8 * U + Q;
            if constexpr (DIR) {
                if constexpr (s < CFG::DS) v[Q] = dir_feature<CFG::NGEO, CFG::NAPP, s>(in, h);
                else v[Q] = 0.0f;
            } else {
                if constexpr (s < CFG::PS) v[Q] = pos_feature<CFG::NGEO, CFG::NAPP, CFG::IPE, s>(in, h);
                else v[Q] = 0.0f;
            }
        } else {
            constexpr int p = Q - 8;
            w.hw[p] = pack2(v[2 * p], v[2 * p + 1]);
            w.lw[p] = pack2(v[2 * p] - lo_f32(w.hw[p]), v[2 * p + 1] - hi_f32(w.hw[p]));
        }
    }
    template <int U>
    NTX_DEV B16 value() const { return w.get(); }
};

// one segment: acc[mt] += (W_hi + W_lo)^T * (B_hi + B_lo) without the lo*lo term, NSTEPS k16-steps.  Tiles are taken two
// at a time (a pair-group = 4 records = 6 MFMAs) so that consecutive MFMAs alternate accumulators: a SLOT is one pair of
// MFMAs followed by the VALU/LDS work placed in its shadow (pieces of the next step's B operand, extra(U, Q)), pinned by
// a sched_barrier.  The A operands of the NEXT pair-group are read from the LDS ring before this one's MFMAs; stage
// fetches and barriers fall where the record index says.
template <int NSTEPS, int NMT, int REC0, int NST, class Gen, class Extra>
NTX_DEV void run_segment16(f32x16 (&acc)[8], WShared &ws, Gen &gen, Extra &&extra) {
    constexpr int NSLOT = NMT / 2 * 3, PPS = 12 / NSLOT;
    static_for<12>([&](auto Q) { gen.template piece<0, decltype(Q)::value>(); });   // exposed
    B16 b = gen.template value<0>();
    __builtin_amdgcn_sched_barrier(0);
    static_for<NSTEPS>([&](auto U) {
        constexpr int u = U;
        static_for<NMT / 2>([&](auto G) {
            constexpr int g = G;
            constexpr int rec = REC0 + (u * NMT + 2 * g) * 2;   // records: tile 2g hi, lo, tile 2g+1 hi, lo
            if constexpr (rec % STAGE16 == 0) stage_fetch<rec / STAGE16 + NSTAGE16 - 1, NST>(ws);
            const half8 a0h = ws.a[0], a0l = ws.a[1], a1h = ws.a[2], a1l = ws.a[3];
            read_group<rec + 4, NST>(ws, ws.a);
            __builtin_amdgcn_sched_barrier(0);
            static_for<3>([&](auto T) {
                constexpr int t = T, q = 3 * g + t;
                if constexpr (t == 0) {
                    acc[2 * g] = mfma16(a0h, b.hi, acc[2 * g]);
                    acc[2 * g + 1] = mfma16(a1h, b.hi, acc[2 * g + 1]);
                } else if constexpr (t == 1) {
                    acc[2 * g] = mfma16(a0h, b.lo, acc[2 * g]);
                    acc[2 * g + 1] = mfma16(a1h, b.lo, acc[2 * g + 1]);
                } else {
                    acc[2 * g] = mfma16(a0l, b.hi, acc[2 * g]);
                    acc[2 * g + 1] = mfma16(a1l, b.hi, acc[2 * g + 1]);
                }
                if constexpr (u + 1 < NSTEPS)
                    static_for<PPS>([&](auto K) { gen.template piece<u + 1, q * PPS + decltype(K)::value>(); });
                extra(U, std::integral_constant<int, q>{});
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (rec % STAGE16 == STAGE16 - 4) { stage_end(); __builtin_amdgcn_sched_barrier(0); }
        });
        if constexpr (u + 1 < NSTEPS) b = gen.template value<u + 1>();
    });
}

// a stage of zero padding at the end of the stream: keep the fetch / barrier pipeline turning, multiply nothing
template <int ST, int NST>
NTX_DEV void skip_stage(WShared &ws) {
    stage_fetch<ST + NSTAGE16 - 1, NST>(ws);
    stage_end();
    __builtin_amdgcn_sched_barrier(0);
}

// WD = false (render kernel): the colour layer C1 starts from the per-ray vector that dir_block (ntx_device.h) left in
// LDS at aux[c1_off .. c1_off + 256) ([half][128], accumulator order) and has no direction segment; WD = true
// (instanced kernel, per-sample directions): static bias, direction segment evaluated.
template <class CFG, bool WD = false>
NTX_DEV void mlp_batch_x3(const SampleIn<CFG::NGEO, CFG::NAPP> &in, WShared &ws, const float *aux_in, int lane,
                            float &sigma, float (&rgb)[3], int c1_off) {
    using G16 = Cfg16<CFG, WD>;
    constexpr int NGEO = CFG::NGEO, NAPP = CFG::NAPP;
    const int h = lane >> 5;
    uint32_t opaque_zero = 0;
    asm volatile("" : "+v"(opaque_zero));
    const float *aux = aux_in + opaque_zero;

    f32x16 accA[8], accB[8];
    auto none = [](auto, auto) {};

    // ---- trunk layer 0 into set A; set B <- bias of layer 1
    init_bias<8>(accA, aux, 0, h);
    {
        EncGen16<CFG, false> gen{in, h, {}, {}};
        run_segment16<G16::PS16, 8, 0, G16::NST>(accA, ws, gen, [&](auto U, auto Q) {
            constexpr int u = decltype(U)::value, q = decltype(Q)::value;
            if constexpr (u < 4 && q < 2) init_bias_tile<2 * u + q>(accB, aux, 1, h);
        });
        static_assert(G16::PS16 >= 4, "layer-1 bias initialised behind layer 0");
    }

    constexpr int NPASS = 8 + (CFG::CD ? 1 : 0);
    float sig_part = 0.0f;
    auto hidden_pass = [&](auto LI, f32x16 (&cur)[8], f32x16 (&prev)[8]) {
        constexpr int li = decltype(LI)::value;
        constexpr bool relu_in = li != 9;
        constexpr int rec0 = G16::rec_pass(li);
        constexpr bool has_pos = li == SKIP + 1;
        constexpr bool init_next = li < NPASS;
        // ParamNerf's colour layer C1 (li = 9) starts from the per-ray vector bias_C1 + W_dir^T dir_map (dir_block,
        // float32): its direction segment is not evaluated per sample
        constexpr bool next_is_c1 = CFG::CD != 0 && !WD && li + 1 == 9;
        constexpr bool has_dir = CFG::CD != 0 && WD && li == 9;
        {
            ConvGen<relu_in, li == DEPTH> cg{prev, aux, h, sig_part, {}, 0.f, 0.f, 0.f, 0.f};
            run_segment16<G16::HS16, 8, rec0, G16::NST>(cur, ws, cg, [&](auto U, auto Q) {
                constexpr int u = decltype(U)::value, q = decltype(Q)::value;
                // tile T of the drained set is free once groups 2T and 2T+1 are converted (behind steps 2T-1 and 2T)
                if constexpr (init_next && (u & 1) == 1 && q == 6) {
                    if constexpr (next_is_c1) init_bias_tile<(u - 1) / 2>(prev, aux + c1_off, 0, h);
                    else init_bias_tile<(u - 1) / 2>(prev, aux, li + 1, h);
                }
            });
        }
        if constexpr (has_pos) {   // input = concat[pos_map, h]  (model.py:107-108)
            const SampleIn<NGEO, NAPP> in2 = launder(in);
            EncGen16<CFG, false> gen{in2, h, {}, {}};
            run_segment16<G16::PS16, 8, rec0 + G16::HS16 * 16, G16::NST>(cur, ws, gen, none);
        }
        if constexpr (has_dir) {   // input = concat[dir_map, feature]  (model.py:115), directions per sample
            const SampleIn<NGEO, NAPP> in2 = launder(in);
            EncGen16<CFG, true> gen{in2, h, {}, {}};
            run_segment16<G16::DS16, 8, rec0 + G16::HS16 * 16, G16::NST>(cur, ws, gen, none);
        }
    };
    static_for<NPASS>([&](auto I) {
        constexpr int li = decltype(I)::value + 1;
        if constexpr (li & 1) hidden_pass(std::integral_constant<int, li>{}, accB, accA);
        else hidden_pass(std::integral_constant<int, li>{}, accA, accB);
    });
    sigma = sig_part + __shfl_xor(sig_part, 32, 64) + aux[aux_alpha_off() + 256];

    // ---- colour half layer, 4 tiles
    auto color_half = [&](f32x16 (&cur)[8], f32x16 (&prev)[8]) {
        init_bias<4>(cur, aux, 10, h);
        float unused = 0.0f;
        if constexpr (CFG::CD == 0) {   // plain Nerf: input = concat[dir_map, feature]  (model.py:39-42)
            ConvGen<false, false> cg{prev, aux, h, unused, {}, 0.f, 0.f, 0.f, 0.f};
            run_segment16<G16::HS16, 4, G16::REC_C2, G16::NST>(cur, ws, cg, none);
            const SampleIn<NGEO, NAPP> in2 = launder(in);
            EncGen16<CFG, true> gen{in2, h, {}, {}};
            run_segment16<G16::DS16, 4, G16::REC_C2 + G16::HS16 * 8, G16::NST>(cur, ws, gen, none);
        } else {
            ConvGen<true, false> cg{prev, aux, h, unused, {}, 0.f, 0.f, 0.f, 0.f};
            run_segment16<G16::HS16, 4, G16::REC_C2, G16::NST>(cur, ws, cg, none);
        }
        // rgb head (model.py:123) on the float32 result
        static_for<3>([&](auto C) {
            constexpr int c = C;
            const f32x4 *wc = reinterpret_cast<const f32x4 *>(aux + aux_rgb_off() + (c * 2 + h) * 64);
            float p = 0.0f;
            static_for<16>([&](auto I) {
                constexpr int i = I;
                const f32x4 w = wc[i];
                static_for<4>([&](auto K) {
                    constexpr int v = 4 * i + decltype(K)::value;
                    p = __builtin_fmaf(relu1(cur[v >> 4][v & 15]), w[decltype(K)::value], p);
                });
            });
            rgb[c] = p + __shfl_xor(p, 32, 64) + aux[aux_rgb_off() + 384 + c];
        });
    };
    if constexpr (NPASS & 1) color_half(accA, accB);
    else color_half(accB, accA);
    // zero-pad stages, then the first pair-group of the next batch (the prefetch of the last real group read padding)
    static_for<(G16::REC_PAD - G16::REC_END) / STAGE16>([&](auto I) { skip_stage<G16::REC_END / STAGE16 + decltype(I)::value, G16::NST>(ws); });
    if constexpr (G16::REC_PAD != G16::REC_END) read_group<0, G16::NST>(ws, ws.a);
    static_assert(G16::REC_END % STAGE16 == 0, "the stream ends on a stage boundary");

    float chk = in.pos[0] + in.pos[1] + in.pos[2] + in.dir[0] + in.dir[1] + in.dir[2];
    if constexpr (CFG::IPE != 0) chk += in.cov[0] + in.cov[1] + in.cov[2];
#pragma unroll
    for (int k = 0; k < CFG::NP; ++k) chk += in.par[k];
    chk = chk - chk;
    sigma += chk;
#pragma unroll
    for (int c = 0; c < 3; ++c) rgb[c] += chk;
}

// the fused render kernel at fp16x3 precision.  Same per-ray work as render_kernel<CFG> around the MLP, but over the
// compacted hit list and in workgroup lockstep: iteration `it` gives wave w of (virtual, XCD-major) workgroup g the hit ray number
// it * (4 * gridDim) + 4 g + w; waves past the end of the list go through the motions on the last hit ray and store nothing.
template <class CFG>
__global__ __launch_bounds__(256) void render_kernel_x3(RenderArgs a) {
    using G16 = Cfg16<CFG>;
    __shared__ __attribute__((aligned(1024))) char ring[NSTAGE16 * STAGE16 * 1024];
    __shared__ __attribute__((aligned(16))) float aux[aux_total() + (CFG::CD ? DIR_BLOCK_FLOATS : 0)];   // + the block's per-ray C1 vectors
    load_aux(aux, a.aux, aux_total());
    const int lane = threadIdx.x & 63, j = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int S = a.n_samples;
    const int nb = (S + 31) >> 5;
    const int n_hit = *a.hit_count;
    const int per_it = gridDim.x * 4;
    const int iters = (n_hit + per_it - 1) / per_it;
    if (iters == 0) return;   // uniform over the grid
    WShared ws;
    ws_prime<G16::NST>(ws, a.wstream, a.stream_bytes, (lds_char *)ring, lane, wv);

    const int vwg = xcd_major_workgroup(blockIdx.x, gridDim.x);   // rays are handed out XCD-major (ntx_device.h)
    __amdgpu_buffer_rsrc_t dir_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<f32x4 *>(a.dir_wstream), 0, a.dir_stream_bytes, 0x00020000);
    for (int it = 0; it < iters; ++it) {
        if constexpr (CFG::CD != 0) {
            if (it % DIR_BLOCK_ITERS == 0) {   // the C1 start vectors of the next 8 rays of each wave (float32, as the f32 kernel)
                __syncthreads();
                const int lane_o = fresh_lane_id();   // dir_block's loop invariants are not carried across the MFMA body (ntx_device.h)
                const RenderArgs *apd = kernargs<RenderArgs>();
                asm volatile("" : "+s"(apd));
                dir_block<CFG>(*apd, dir_rsrc, aux, aux + aux_total(), it * per_it, per_it, vwg, wv, lane_o, n_hit);
                __syncthreads();
            }
        }
        const int idx = it * per_it + vwg * 4 + wv;
        const bool live = idx < n_hit;
        const int64_t ray = a.hit_list[live ? idx : n_hit - 1];
        RayAccum ra{1.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        for (int b = 0; b < nb; ++b) {
            int64_t r = ray;
            asm volatile("" : "+s"(r));
            const RenderArgs *ap = kernargs<RenderArgs>();
            asm volatile("" : "+s"(ap));
            const RenderArgs &q = *ap;
            const float t0 = q.t[2 * r], t1 = q.t[2 * r + 1];
            const float ox = q.rays_o[3 * r], oy = q.rays_o[3 * r + 1], oz = q.rays_o[3 * r + 2];
            const float dx = q.rays_d[3 * r], dy = q.rays_d[3 * r + 1], dz = q.rays_d[3 * r + 2];
            const float dnorm = __builtin_sqrtf(dx * dx + dy * dy + dz * dz);
            const float cone = q.cone ? q.cone[r] : 0.0f;
            const float *prow = q.params + (r / q.rays_per_row) * param_stride<CFG>(q);
            const int i = 32 * b + j;
            const bool valid = i < S;
            const int ic = valid ? i : S - 1;
            const int blur_idx = q.blur_idx;
            const int64_t gr = (q.flags & (NTX_FLAG_PERTURB | NTX_FLAG_RAW_NOISE)) ? global_index(q.idx0, q.idx_run, q.idx_stride, r) : r;
            SampleIn<CFG::NGEO, CFG::NAPP> in;
            in.dir[0] = dx / dnorm; in.dir[1] = dy / dnorm; in.dir[2] = dz / dnorm;
            float dist;
            if constexpr (CFG::IPE == 0) {
                const float z = z_of(q, r, gr, ic, t0, t1, S);
                const float zn = z_of(q, r, gr, ic < S - 1 ? ic + 1 : ic - 1, t0, t1, S);
                dist = (ic < S - 1 ? zn - z : z - zn) * dnorm;
                in.pos[0] = ox + dx * z; in.pos[1] = oy + dy * z; in.pos[2] = oz + dz * z;
                in.cov[0] = in.cov[1] = in.cov[2] = 0.0f;
#pragma unroll
                for (int k = 0; k < CFG::NP; ++k) {
                    float p = param_at<CFG>(q, prow, k);
                    if (k == blur_idx) p = p * (cone * z);
                    in.par[k] = p;
                }
            } else {   // MipRenderer.render_rays (renderer.py:365-409), as in render_kernel<CFG>
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
            mlp_batch_x3<CFG>(in, ws, aux, lane, sigma, raw, aux_total() + ((it % DIR_BLOCK_ITERS) * 4 + wv) * DIR_ROW_STRIDE);
            const RenderArgs *ap2 = kernargs<RenderArgs>();
            asm volatile("" : "+s"(ap2));
            float noise = 0.0f;   // renderer.py:190-192, as render_kernel<CFG>
            if (ap2->flags & NTX_FLAG_RAW_NOISE)
                noise = ap2->raw_noise_std * normal01(global_index(ap2->idx0, ap2->idx_run, ap2->idx_stride, ray), ic, ap2->seed_lo, ap2->seed_hi);
            composite_step<32>(ra, sigma, raw, dist, valid && live, ap2->flags, j,
                               ap2->weights_out ? ap2->weights_out + r * S + ic : nullptr, noise);
        }
        float out[4] = {ra.c0, ra.c1, ra.c2, ra.a};
        if (a.flags & NTX_FLAG_COMPOSITE_BKGD) {
#pragma unroll
            for (int k = 0; k < 3; ++k) out[k] = out[k] + (1.0f - ra.a) * a.bkgd[k];
        }
        if (lane == 0 && live) {
            a.color_out[3 * ray + 0] = out[0]; a.color_out[3 * ray + 1] = out[1];
            a.color_out[3 * ray + 2] = out[2]; a.alpha_out[ray] = out[3];
            if ((a.flags & NTX_FLAG_CHECK_NUMERICS) && a.status) {
                const float s = out[0] + out[1] + out[2] + out[3];
                if (!(__builtin_fabsf(s) <= 3.0e38f)) atomicOr(a.status, 1);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// InstanceRenderer tail at fp16x3 (instance_kernel<CFG> of ntx_device.h is the float32 one).  Rays cost a different
// number of 32-sample batches each, and the workgroup shares one weight stream in lockstep, so the workgroup proceeds in
// ROUNDS of one batch per wave: before each round every wave runs the scheduler of instance_kernel<CFG> (take the next
// unclaimed ray off the device work counter, compact its in-patch samples, collect tails of successive rays into one
// packed batch); a wave that finds nothing left keeps its place in the barriers with idle batches until its three
// neighbours are done (at most one ray's worth at the very end).
// ---------------------------------------------------------------------------------------------
template <class CFG>
__global__ __launch_bounds__(256) void instance_kernel_x3(InstanceArgs a) {
    using G16 = Cfg16<CFG, true>;
    __shared__ __attribute__((aligned(1024))) char ring[NSTAGE16 * STAGE16 * 1024];
    __shared__ __attribute__((aligned(16))) float aux[aux_total()];
    __shared__ uint16_t sidx_all[4][MAX_INSTANCE_SAMPLES];
    __shared__ InstancePending pend_all[4];
    __shared__ int busy[4];
    load_aux(aux, a.aux, aux_total());
    const int lane = threadIdx.x & 63, j = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int S = a.n_samples;
    uint16_t *sidx = sidx_all[wv];
    InstancePending &pend = pend_all[wv];
    WShared ws;
    ws_prime<G16::NST>(ws, a.wstream, a.stream_bytes, (lds_char *)ring, lane, wv);

    // the appended sample (colour taken as is, alpha_last is an alpha, not a density: renderer.py:323-339) and the store
    auto finish = [&](int64_t ray, const RayAccum &ra) {
        const float wl = a.alpha_last[ray] * ra.T;
        float out[4] = {ra.c0 + wl * a.color_last[3 * ray], ra.c1 + wl * a.color_last[3 * ray + 1],
                        ra.c2 + wl * a.color_last[3 * ray + 2], ra.a + wl};
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

    // The number of in-patch samples differs from ray to ray (0 .. S), so a static ray -> wave map leaves waves idle at the
    // end (19 % on the carpet_instanced bench workload): each wave takes the next unclaimed ray instead.
    int64_t cur = -1;            // the ray whose whole batches are being marched, -1 = none
    int count = 0, nfull = 0, b = 0, pend_n = 0, pend_k = 0;
    bool exhausted = false;
    float cone = 0.0f;
    RayAccum ra{1.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    for (;;) {
        // ---- scheduler (wave-uniform): advance until a batch is due.  mode 1 = a whole batch of `cur`, 2 = the packed tails
        int mode = 0;
        for (;;) {
            if (cur >= 0 && b < nfull) { mode = 1; break; }
            if (cur >= 0) {
                const int r = count - 32 * nfull;
                if (r == 0) { finish(cur, ra); cur = -1; continue; }
                if (pend_n + r <= 32 && pend_k < PEND_MAX) {   // the tail joins the pending batch as segment pend_k
                    if (lane < r) { pend.idx[pend_n + lane] = sidx[32 * nfull + lane]; pend.slot[pend_n + lane] = (uint8_t)pend_k; }
                    if (lane == 0) {
                        pend.ray[pend_k] = (int32_t)cur; pend.last[pend_k] = pend_n + r - 1; pend.cone[pend_k] = cone;
                        pend.acc[pend_k][0] = ra.T; pend.acc[pend_k][1] = ra.c0; pend.acc[pend_k][2] = ra.c1;
                        pend.acc[pend_k][3] = ra.c2; pend.acc[pend_k][4] = ra.a;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    pend_n += r; ++pend_k; cur = -1;
                    continue;
                }
                mode = 2; break;   // no room: flush the pending batch first, `cur` joins the next one
            }
            if (!exhausted) {
                int r32 = 0;
                if (lane == 0) r32 = atomicAdd(a.work_counter, 1);
                const int64_t claim = (int64_t)__builtin_amdgcn_readfirstlane(r32);
                if (claim >= a.n_rays) { exhausted = true; continue; }
                const int64_t ray = a.order[claim];
                if (!a.hit[ray]) {   // renderer.py:265-272, 313-314: stays 0, also under composite_bkgd
                    if (lane < 3) a.color_out[3 * ray + lane] = 0.0f;
                    if (lane == 3) a.alpha_out[ray] = 0.0f;
                    continue;
                }
                const float *drow = a.dists + ray * S;
                int n = 0;
                for (int base0 = 0; base0 < S; base0 += 512) {   // 8 independent loads in flight, then their 8 ballots
                    float dv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { const int i = base0 + 64 * u + lane; dv[u] = i < S ? drow[i] : 0.0f; }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int i = base0 + 64 * u + lane;
                        const bool v = dv[u] > 0.0f;
                        const unsigned long long m = __ballot(v);
                        if (v) sidx[n + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)i;
                        n += __popcll(m);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                cur = ray; count = n; nfull = n >> 5; b = 0;
                cone = a.cone ? a.cone[ray] : 0.0f;
                ra = RayAccum{1.0f, 0.0f, 0.0f, 0.0f, 0.0f};
                continue;
            }
            if (pend_n > 0) mode = 2;
            break;
        }
        // the four waves share the weight stream in lockstep ROUNDS of one batch each: a wave with nothing left keeps its
        // place in the stream's barriers with an idle batch on sample 0 until its three neighbours are done too
        if (lane == 0) busy[wv] = mode != 0;
        __syncthreads();
        const bool any = (busy[0] | busy[1] | busy[2] | busy[3]) != 0;   // (rewritten only after the batch's own barriers)
        if (!any) break;                               // the same for the four waves

        // ---- this lane's sample
        bool valid = mode != 0;
        int slot = 0;
        int64_t sm = 0;
        float cone_l = cone;
        if (mode == 1) {
            sm = cur * S + sidx[32 * b + j];
        } else if (mode == 2) {
            valid = j < pend_n;
            const int jc = valid ? j : 0;
            slot = pend.slot[jc];
            sm = (int64_t)pend.ray[slot] * S + pend.idx[jc];
            cone_l = pend.cone[slot];
        }
        SampleIn<CFG::NGEO, CFG::NAPP> in;
#pragma unroll
        for (int c = 0; c < 3; ++c) { in.pos[c] = a.pts[3 * sm + c]; in.dir[c] = a.rays_d_map[3 * sm + c]; }
        if constexpr (CFG::IPE == 0) {
            in.cov[0] = in.cov[1] = in.cov[2] = 0.0f;
#pragma unroll
            for (int c = 0; c < CFG::NP; ++c) {
                float p = param_at<CFG>(a, a.params_map + param_stride<CFG>(a) * sm, c);
                if (c == a.blur_idx) p = p * (cone_l * a.t[sm] / a.patch_scale);                 // renderer.py:259-262
                in.par[c] = p;
            }
        } else {
            // MipInstanceRenderer (renderer.py:510-540, 570-587): radius = blur parameter * cone_scale / patch_scale,
            // spliced out of the parameters; gaussian with mu = t and (sic) hw = dists; the mean is the sample point
            const float *pr = a.params_map + CFG::NP_IN * sm;
            float t_mean, t_var, r_var;
            cone_moments(a.t[sm], a.dists[sm], pr[a.blur_idx] * cone_l / a.patch_scale, t_mean, t_var, r_var);
            cone_cov(t_var, r_var, in.dir, in.cov);
#pragma unroll
            for (int c = 0; c < CFG::NP; ++c) in.par[c] = pr[c < a.blur_idx ? c : c + 1];
        }
        float sigma, raw[3];
        mlp_batch_x3<CFG, true>(in, ws, aux, lane, sigma, raw, 0);
        if (mode == 0) continue;
        const float wgt = a.alpha_weight ? a.alpha_weight[sm] * a.density_scale : a.density_scale;   // :300
        sigma = sigma * wgt;
        float col[3];
        if (a.instance_color) {                                                                // :306-307, 322-323
            const int id = a.instance_id[sm];
#pragma unroll
            for (int c = 0; c < 3; ++c) col[c] = a.instance_color[3 * id + c];
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) col[c] = (a.flags & NTX_FLAG_MAP_EXR) ? elu1f_(raw[c]) : sigmoidf_(raw[c]);
        }
        if (a.flags & NTX_FLAG_RAW_NOISE) {                                                       // :335-337
            const int64_t ray_l = sm / S;
            sigma += a.raw_noise_std * normal01(global_index(a.idx0, a.idx_run, a.idx_stride, ray_l), (int)(sm - ray_l * S), a.seed_lo, a.seed_hi);
        }
        const float al = valid ? 1.0f - expf(-__builtin_fmaxf(sigma, 0.0f) * a.dists[sm] / a.patch_scale) : 0.0f;   // :339
        if (mode == 1) {
            composite_core<32>(ra, al, col, true, j, nullptr);
            ++b;
        } else {
            for (int k = 0; k < pend_k; ++k) {
                RayAccum rk{pend.acc[k][0], pend.acc[k][1], pend.acc[k][2], pend.acc[k][3], pend.acc[k][4]};
                composite_segment(rk, (valid && slot == k) ? al : 0.0f, col, j, pend.last[k]);
                finish(pend.ray[k], rk);
            }
            __builtin_amdgcn_wave_barrier();   // the pending batch is rewritten from here on
            pend_n = 0; pend_k = 0;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// stand-alone MLP at fp16x3 (mlp_kernel<CFG> of ntx_device.h is the float32 one): samples are independent, 32 per wave;
// the workgroup's four waves take batches 4i .. 4i+3 of each round, waves past the end run an idle batch to keep their
// place in the stream's barriers.  Directions are per sample: the stream that keeps C1's direction segment.
// ---------------------------------------------------------------------------------------------
template <class CFG>
__global__ __launch_bounds__(256) void mlp_kernel_x3(MlpArgs a) {
    using G16 = Cfg16<CFG, true>;
    __shared__ __attribute__((aligned(1024))) char ring[NSTAGE16 * STAGE16 * 1024];
    __shared__ __attribute__((aligned(16))) float aux[aux_total()];
    load_aux(aux, a.aux, aux_total());
    const int lane = threadIdx.x & 63, j = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t nbatch = (a.m + 31) >> 5;
    const int64_t per_it = (int64_t)gridDim.x * 4;
    const int64_t iters = (nbatch + per_it - 1) / per_it;
    if (iters == 0) return;
    WShared ws;
    ws_prime<G16::NST>(ws, a.wstream, a.stream_bytes, (lds_char *)ring, lane, wv);
    for (int64_t it = 0; it < iters; ++it) {
        const int64_t b = it * per_it + (int64_t)blockIdx.x * 4 + wv;
        const int64_t m = b * 32 + j;
        const bool valid = b < nbatch && m < a.m;
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
        mlp_batch_x3<CFG, true>(in, ws, aux, lane, sigma, raw, 0);
        if (valid && lane < 32) {
            a.color_out[3 * m + 0] = raw[0]; a.color_out[3 * m + 1] = raw[1]; a.color_out[3 * m + 2] = raw[2];
            a.sigma_out[m] = sigma;
        }
    }
}

}  // namespace ntx
