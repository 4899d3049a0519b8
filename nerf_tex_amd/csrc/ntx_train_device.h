// ntx_train_device.h -- the three matrix-core kernels of a training step (included by ntx_train.hip only; gfx950).
//
//   fwd_chain_kernel   the whole network forward on one block of 32 samples per wave, the way the render kernel does it (ntx_device.h,
//                      ntx_layout.h): every Dense layer transposed on v_mfma_f32_32x32x2_f32, out^T[feature, sample] = W^T . h^T, so that an
//                      accumulator register, after ReLU, IS the next layer's B operand -- the activations of a block never leave the
//                      register file on their way through the network.  What training adds: every layer's activations are STORED once
//                      (the weight gradients need them) and one bit per output says whether its ReLU let it through.
//   dx_chain_kernel    the way back, same shape: dX^T[in feature, sample] = W[in, out] . dY^T[out feature, sample] with the weights packed
//                      transposed, masked by the forward pass's bits, every layer's gradient stored once.
//   dw_kernel          the weight gradients dW = X^T . dY (reduction over the samples) straight from the stored operands: both are kept in
//                      the matrix cores' own operand order (below), so a wave's 16-byte loads ARE its A and B operands -- no LDS, no
//                      barrier, no VALU work in the loop.
//
// O layout ("operand order") of a stored matrix with R rows (features; R a multiple of 32) over the samples: element (row 32 T + i,
// sample 32 blk + p) is float
//     ((blk * R/32 + T) * 4 + (p >> 3)) * 256 + (i + 32 * ((p >> 2) & 1)) * 4 + (p & 3)
// i.e. per block of 32 samples and tile of 32 rows four 1 KiB records q = p >> 3; lane l = (i, kh) of a wave reads its float4 at l * 16 and
// holds samples 8 q + 4 kh + (0..3) of row i: k-step 4 q + c of v_mfma_f32_32x32x2_f32 pairs sample 8 q + c (lower half-wave) with
// 8 q + 4 + c (upper) -- the order of the summation over the samples is free.
#pragma once

#include "ntx_device.h"   // sin_q, mfma32, static_for, load_aux; ntx_layout.h: RING, hidden_row

namespace ntx_train {

using ntx::f32x16;
using ntx::f32x4;
using ntx::hidden_row;
using ntx::mfma32;
using ntx::RING;
using ntx::static_for;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define TRN_DEV __device__ __forceinline__

TRN_DEV __amdgpu_buffer_rsrc_t make_rsrc(const void *base, long long bytes) {
    const long long b = bytes < 0 ? 0 : (bytes > 0x7ffffff0ll ? 0x7ffffff0ll : bytes);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)b, 0x00020000);
}

// where lane (n = l & 31, h = l >> 5) of the transposed layers' accumulator layout stores inside a block of an O-layout matrix, and where
// value V = 16 T + 4 g + c of a lane (feature 32 T + 8 g + 4 h + c: ntx_layout.h hidden_row) goes
TRN_DEV uint32_t o_lane_bytes(int lane) {
    const int n = lane & 31, h = lane >> 5;
    return (uint32_t)((n >> 3) * 1024 + ((n >> 2) & 1) * 512 + h * 64 + (n & 3) * 4);
}
// cache policy of the activations' stores
constexpr int STORE_NT = 0;   // (measured: with the nt bit L2 stops combining the 16-byte pieces of a line -- twice the HBM writes, chains 20 % slower)
constexpr uint32_t o_value_bytes(int V) { return (uint32_t)((V >> 4) * 4096 + ((V & 15) >> 2) * 128 + (V & 3) * 16); }

// ---------------------------------------------------------------------------------------------------------------------------
// the weight stream of a chain: 1 KiB records (64 lanes x float4 = the A operands of four tiles of one k-step), consumed strictly in order
// through a register ring RING records deep, straight from L2 (ntx_device.h WStream).  Every segment is a whole number of ring turns, so
// the ring's phase is a compile-time fact everywhere; the stream ends with a copy of its first RING records: the prefetch runs on into the
// next block of samples.
// ---------------------------------------------------------------------------------------------------------------------------
template <int R>
struct WRingT {
    static constexpr int DEPTH = R;
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t voff;
    f32x4 r[R];
};
typedef WRingT<RING> WRing;            // the forward chain: as the render kernels (its registers are spoken for)
constexpr int DX_RING = 16;            // the chain back has registers to spare: twice the depth rides out an L2 miss of the weight stream
// (development probes, tools/dev/r6_variant.sh + r6_steady_pmc.sh: NTX_X_WSMALL keeps the weight stream inside its first 64 KB -- always in
// L2 --, NTX_X_NOWLOAD never reloads the ring, NTX_X_NOSTORE drops the activations' stores, NTX_X_NOBITS the masks', NTX_X_DWSMALL / _DWNOLOAD
// the same for the weight gradients' operands: wrong results, the time of what is left)
template <class WR>
TRN_DEV f32x4 wr_load(const WR &w, uint32_t byte_off) {
#ifdef NTX_X_WSMALL
    byte_off &= 0xffffu;
#endif
#ifdef NTX_X_NOWLOAD
    if (byte_off >= (uint32_t)WR::DEPTH * 1024u) return w.r[(byte_off >> 10) % WR::DEPTH];
#endif
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w.rsrc, w.voff, byte_off, 0));
}

// NSTEPS k-steps of an NMT-tile layer whose B operands are this lane's `hin` (the previous layer's outputs); ZERO: the accumulators start
// from nothing (the first k-step takes the constant 0 as C).  extra(S, MT): what else goes into the slot behind MFMA (S, MT).
template <int NSTEPS, int NMT, bool ZERO, class WR, class Extra>
TRN_DEV void seg_hidden(f32x16 (&acc)[8], WR &ws, uint32_t &sbase, const float (&hin)[128], Extra &&extra) {
    constexpr int RPS = NMT / 4, RING = WR::DEPTH;
    static_assert((NSTEPS * RPS) % RING == 0, "whole ring turns");
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    static_for<NSTEPS>([&](auto S) {
        constexpr int s = S;
        f32x4 w;
        static_for<NMT>([&](auto MT) {
            constexpr int mt = MT;
            if constexpr (mt % 4 == 0) {
                constexpr int rec = s * RPS + mt / 4;
                w = ws.r[rec % RING];
                ws.r[rec % RING] = wr_load(ws, sbase + (uint32_t)(rec + RING) * 1024u);
            }
            if constexpr (ZERO && s == 0) acc[mt] = mfma32(w[mt % 4], hin[s], zero);
            else acc[mt] = mfma32(w[mt % 4], hin[s], acc[mt]);
            extra(S, MT);
            __builtin_amdgcn_sched_barrier(0);
        });
    });
    sbase += (uint32_t)(NSTEPS * RPS) * 1024u;
}

// NG groups of 4 k-steps of an 8-tile layer whose B operands come from memory (the encoded position / direction of the samples) through a
// rolling buffer of three groups: groups 0 and 1 were asked for long ago (fetch(G) fills pb[4 (G % 3) ..]), group g + 2 is asked for when
// group g starts.  One group = one ring turn.
template <int NG, class Fetch>
TRN_DEV void seg_mem(f32x16 (&acc)[8], WRing &ws, uint32_t &sbase, float (&pb)[12], Fetch &&fetch) {
    static_assert(RING == 8, "a group of 4 k-steps of 8 tiles is one ring turn");
    static_for<NG>([&](auto G) {
        constexpr int g = G;
        if constexpr (g + 2 < NG) fetch(std::integral_constant<int, g + 2>{});
        static_for<4>([&](auto K) {
            constexpr int k = K;
            f32x4 w;
            static_for<8>([&](auto MT) {
                constexpr int mt = MT;
                if constexpr (mt % 4 == 0) {
                    constexpr int slot = 2 * k + mt / 4;
                    w = ws.r[slot];
                    ws.r[slot] = wr_load(ws, sbase + (uint32_t)(8 * g + slot + RING) * 1024u);
                }
                acc[mt] = mfma32(w[mt % 4], pb[4 * (g % 3) + k], acc[mt]);
                __builtin_amdgcn_sched_barrier(0);
            });
        });
    });
    sbase += (uint32_t)NG * 8192u;
}

// one ring turn of which the first NREC records are k-steps with the given B values (NMT tiles; NREC = k-steps * NMT / 4), the rest padding
// that is fetched to keep the ring turning and never multiplied
template <int NKS, int NMT, bool ZERO, class WR>
TRN_DEV void seg_few(f32x16 (&acc)[8], WR &ws, uint32_t &sbase, const float (&b)[NKS]) {
    constexpr int RPS = NMT / 4, RING = WR::DEPTH;
    static_assert(NKS * RPS <= RING, "one ring turn");
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    static_for<RING>([&](auto I) {
        constexpr int rec = I;
        const f32x4 w = ws.r[rec];
        ws.r[rec] = wr_load(ws, sbase + (uint32_t)(rec + RING) * 1024u);
        if constexpr (rec < NKS * RPS) {
            constexpr int s = rec / RPS, t0 = 4 * (rec % RPS);
            static_for<4>([&](auto C) {
                constexpr int c = C;
                if constexpr (ZERO && s == 0) acc[t0 + c] = mfma32(w[c], b[s], zero);
                else acc[t0 + c] = mfma32(w[c], b[s], acc[t0 + c]);
                __builtin_amdgcn_sched_barrier(0);
            });
        }
    });
    sbase += (uint32_t)RING * 1024u;
}

// ---------------------------------------------------------------------------------------------------------------------------
// an accumulator set becomes the next layer's input.  VALU instructions do not hide under the f32 MFMA (it runs on the vector ALUs'
// lanes: DESIGN 4.1), so each conversion is ONE dense block between two layers.
// ---------------------------------------------------------------------------------------------------------------------------
// hin[V0 .. V0+8) <- relu(accumulators), and one bit per value -- (v > 0), exactly: the sign of 0 - v, which is +0 for v = +-0 -- shifted into
// w from below (value V0 + k ends up 31 - (32-value index) from the top: see keep_bit)
template <int V0>
TRN_DEV void convert8_relu_bits(float (&hin)[128], const f32x16 (&prev)[8], uint32_t &w) {
    constexpr int T = V0 >> 4, R = V0 & 15;
    float t0, t1, t2, t3;
    asm("v_accvgpr_read_b32 %0, %13\n\tv_accvgpr_read_b32 %1, %14\n\tv_accvgpr_read_b32 %2, %15\n\tv_accvgpr_read_b32 %3, %16\n\t"
        "v_accvgpr_read_b32 %4, %17\n\tv_accvgpr_read_b32 %5, %18\n\tv_accvgpr_read_b32 %6, %19\n\tv_accvgpr_read_b32 %7, %20\n\t"
        "v_sub_f32 %9, 0, %0\n\tv_sub_f32 %10, 0, %1\n\tv_sub_f32 %11, 0, %2\n\tv_sub_f32 %12, 0, %3\n\t"
        "v_alignbit_b32 %8, %8, %9, 31\n\tv_alignbit_b32 %8, %8, %10, 31\n\tv_alignbit_b32 %8, %8, %11, 31\n\tv_alignbit_b32 %8, %8, %12, 31\n\t"
        "v_sub_f32 %9, 0, %4\n\tv_sub_f32 %10, 0, %5\n\tv_sub_f32 %11, 0, %6\n\tv_sub_f32 %12, 0, %7\n\t"
        "v_alignbit_b32 %8, %8, %9, 31\n\tv_alignbit_b32 %8, %8, %10, 31\n\tv_alignbit_b32 %8, %8, %11, 31\n\tv_alignbit_b32 %8, %8, %12, 31\n\t"
        "v_max_f32 %0, 0, %0\n\tv_max_f32 %1, 0, %1\n\tv_max_f32 %2, 0, %2\n\tv_max_f32 %3, 0, %3\n\t"
        "v_max_f32 %4, 0, %4\n\tv_max_f32 %5, 0, %5\n\tv_max_f32 %6, 0, %6\n\tv_max_f32 %7, 0, %7"
        : "=&v"(hin[V0 + 0]), "=&v"(hin[V0 + 1]), "=&v"(hin[V0 + 2]), "=&v"(hin[V0 + 3]), "=&v"(hin[V0 + 4]), "=&v"(hin[V0 + 5]), "=&v"(hin[V0 + 6]),
          "=&v"(hin[V0 + 7]), "+v"(w), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "a"(prev[T][R + 0]), "a"(prev[T][R + 1]), "a"(prev[T][R + 2]), "a"(prev[T][R + 3]), "a"(prev[T][R + 4]), "a"(prev[T][R + 5]), "a"(prev[T][R + 6]),
          "a"(prev[T][R + 7]));
}
// the bit of value V = 16 T + R in word T >> 1 of a block's mask: values enter a word from below in the order (T & 1, R)
constexpr int keep_bit(int V) { return 31 - (V & 31); }

// NMT tiles -> hin, ReLU, the mask bits of the layer into `bits` (words T >> 1)
template <int NMT>
TRN_DEV void convert_relu_bits(float (&hin)[128], const f32x16 (&prev)[8], u32x4 &bits) {
    static_for<(NMT + 1) / 2>([&](auto W) {
        constexpr int wd = W;
        uint32_t w = 0;
        static_for<(NMT - 2 * wd >= 2 ? 4 : 2)>([&](auto Q) { convert8_relu_bits<32 * wd + 8 * decltype(Q)::value>(hin, prev, w); });
        bits[wd] = w;
    });
}
// a linear layer's outputs: as they are
template <int NMT>
TRN_DEV void convert_linear(float (&hin)[128], const f32x16 (&prev)[8]) {
    static_for<NMT * 16>([&](auto V) { constexpr int v = V; hin[v] = prev[v >> 4][v & 15]; });
}
// the way back: kept where the forward pass left a bit
template <int NMT>
TRN_DEV void convert_mask(float (&hin)[128], const f32x16 (&prev)[8], const u32x4 &bits) {
    static_for<NMT * 16>([&](auto V) {
        constexpr int v = V;
        const float x = prev[v >> 4][v & 15];
        const int keep = __builtin_amdgcn_sbfe((int)bits[v >> 5], keep_bit(v), 1);          // 0 or -1
        hin[v] = __builtin_bit_cast(float, __builtin_bit_cast(int, x) & keep);
    });
}

// accumulator tile <- bias from the LDS image of the aux block ([layer][half][128] in accumulator order); one 16-wide definition
// (ntx_device.h init_bias_tile)
constexpr int AUX_BIAS = 0, AUX_ALPHA_W = 11 * 256, AUX_ALPHA_B = AUX_ALPHA_W + 256, AUX_RGB_W = AUX_ALPHA_B + 4, AUX_RGB_B = AUX_RGB_W + 384,
              AUX_FLOATS = AUX_RGB_B + 4;
template <int MT>
TRN_DEV void bias_tile(f32x16 (&acc)[8], const float *aux, int layer, int h) {
    const f32x4 *b = reinterpret_cast<const f32x4 *>(aux + AUX_BIAS + layer * 256 + h * 128) + MT * 4;
    const f32x4 v0 = b[0], v1 = b[1], v2 = b[2], v3 = b[3];
    acc[MT] = f32x16{v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
}

// ---------------------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int MAX_PB_GROUPS = 12;                      // 48 k-steps = 96 encoded features at most
struct FwdArgs {
    const float *stream; uint32_t stream_bytes;        // the forward weight stream (pack_kernel), tail included
    const float *aux;                                  // biases and the two narrow heads, AUX_FLOATS
    long long M;
    int ptiles, dtiles;                                // rows / 32 of pos / dir
    const float *pos, *dir;                            // encoded inputs (O layout, what the weight gradients read): row 2 S + h is k-step S's B operand
    float *act; long long act_stride;                  // O layout, act + i * act_stride: h0 .. h7 (256 rows), feature (256), c1o (256), c2o (128)
    unsigned int *bits; long long bits_stride;         // h0 .. h7, c1o, c2o: [block][64 lanes][4 words]
    float *sigma, *raw_rgb;                            // [M], [M][3]
    // HOIST builds: the colour layer's direction segment once per RAY (dirrow_kernel): bias_C1 + W_C1[:dir_map]^T . dir_map in accumulator
    // order, [n_rays][half][128]; S samples a ray (a multiple of 32)
    const float *dirrow; int n_rays, S;
};

#ifdef NTX_TRAIN_FWD
// PSG / DSG: groups of 4 k-steps of the position / direction segment (the stream pads them with zero rows).
// HOIST: Renderer.evaluate_model repeats the view direction and the appearance parameters for every sample of a ray (renderer.py:152-154), so
// the direction segment of the colour layer is ONE vector per ray -- 8 DSG of the block's 10 656 MFMAs recompute a constant (the render
// kernels hoist it too).  The HOIST build starts the colour layer's accumulators from the ray's row (bias included; staged in
// the wave's LDS and read exactly like a bias) and steps over the segment's records in the stream (whole ring turns: the ring is primed again
// behind them).  The weight gradients still contract the per-sample dir_map rows (encode_kernel); nothing else changes.  Not for a blur_idx
// on an appearance parameter (:155-158) nor for S that is no multiple of 32 (a block then lies in two rays): the host picks the build.
template <int PSG, int DSG, bool HOIST>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void fwd_chain_kernel(FwdArgs a) {
    __shared__ __attribute__((aligned(16))) float aux_lds[AUX_FLOATS];
    __shared__ __attribute__((aligned(16))) float rows_lds[HOIST ? 4 * 256 : 4];      // HOIST: per wave the row of the ray its block lies in
    ntx::load_aux(aux_lds, a.aux, AUX_FLOATS);
    const int lane = threadIdx.x & 63, n = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)), nwaves = gridDim.x * 4;
    const int n_blocks = (int)((a.M + 31) >> 5);
    if (wave >= n_blocks) return;
    WRing ws;
    ws.rsrc = make_rsrc(a.stream, a.stream_bytes);
    ws.voff = (uint32_t)lane * 16u;
    static_for<RING>([&](auto I) { ws.r[I] = wr_load(ws, (uint32_t)decltype(I)::value * 1024u); });
    // (row 2 S + h, sample n) of an O-layout block: this lane's part of the offset, and S's: (S >> 4) * 4096 + (S & 15) * 32
    const uint32_t lane_o = o_lane_bytes(lane), lane_r = (uint32_t)((n >> 3) * 1024 + ((n >> 2) & 1) * 512 + (n & 3) * 4 + h * 16), lane16 = (uint32_t)lane * 16u;
    float pb[12];
    const __amdgpu_buffer_rsrc_t rs_row = make_rsrc(a.dirrow, HOIST ? (long long)a.n_rays * 1024 : 0);
    __amdgpu_buffer_rsrc_t rs_in = make_rsrc(a.pos + (size_t)wave * a.ptiles * 1024, (long long)a.ptiles * 4096);
    auto fetch = [&](auto G) {                          // group G of the segment rs_in points at
        constexpr int g = G;
        static_for<4>([&](auto K) {
            constexpr int k = K;
            constexpr int s = 4 * g + k;
            pb[4 * (g % 3) + k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_in, lane_r, (uint32_t)((s >> 4) * 4096 + (s & 15) * 32), 0));
        });
    };
    fetch(std::integral_constant<int, 0>{}); fetch(std::integral_constant<int, 1>{});
    for (int blk = wave; blk < n_blocks; blk += nwaves) {
        // (the aux block never changes: without an opaque offset the optimiser hoists every bias read out of the loop and spills: ntx_device.h)
        uint32_t opaque_zero = 0;
        asm volatile("" : "+v"(opaque_zero));
        const float *aux = aux_lds + opaque_zero;
        const float *rows_aux = rows_lds + opaque_zero + (HOIST ? (threadIdx.x >> 6) * 256 : 0) - AUX_BIAS - 9 * 256;     // bias_tile(.., rows_aux, 9, h) reads the ray's row
        uint32_t sbase = 0;
        f32x16 acc[8];                                     // ONE accumulator set (see the note at `layer` below)
        float hin[128];
        const long long m = (long long)blk * 32 + n;
        const bool valid = m < a.M;
        if constexpr (HOIST) {
            // the block's ray's row -- 1 KiB, 16 bytes a lane -- into the wave's LDS now, while no layer's operands are live
            const uint32_t ray = __builtin_amdgcn_readfirstlane((int)(((uint32_t)blk * 32u) / (uint32_t)a.S));
            const f32x4 x0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_row, lane16, (ray < (uint32_t)a.n_rays ? ray : (uint32_t)a.n_rays - 1u) * 1024u, 0));
            *reinterpret_cast<f32x4 *>(rows_lds + (threadIdx.x >> 6) * 256 + lane * 4) = x0;
            __builtin_amdgcn_wave_barrier();
        }
        // input `idx` of the chain (h0 .. h7, feature, c1o) leaves for memory one value per k-step while the layer that reads it runs
        __amdgpu_buffer_rsrc_t rs_out = make_rsrc(a.act, 0);
        auto leaves_to = [&](int idx, int tiles) { rs_out = make_rsrc(a.act + (size_t)idx * a.act_stride + (size_t)blk * tiles * 1024, (long long)tiles * 4096); };
        auto leave = [&](auto V) {
            constexpr int v = V;
#ifndef NTX_X_NOSTORE
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, hin[v]), rs_out, lane_o, o_value_bytes(v), STORE_NT);
#endif
        };
        auto bits_leave = [&](int idx, const u32x4 &bw) {
#ifndef NTX_X_NOBITS
            const __amdgpu_buffer_rsrc_t rb = make_rsrc(a.bits + (size_t)idx * a.bits_stride + (size_t)blk * 256, 1024);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, bw), rb, lane16, 0, 0);
#endif
        };
        // ---- trunk layer 0: pos_map -> 256 (model.py:104-106)
        // (the accumulators start from their layer's bias: LDS reads straight into them)
        static_for<8>([&](auto T) { bias_tile<decltype(T)::value>(acc, aux, 0, h); });
        seg_mem<PSG>(acc, ws, sbase, pb, fetch);
        float sig_part = 0.0f;
        // ---- layers LI = 1 .. 10: trunk 1 .. 7, the feature layer (8, linear), the colour layers (9: 256, 10: 128).
        // Round 5 alternated TWO accumulator sets so that a layer's bias could be read while the layer before ran: 16 tiles = all 256 AGPRs live
        // through every layer, an exact fit that hipcc's allocator met for some segment lengths and not for others (four bias tiles spilled in the
        // middle of the chain, each reload behind a full vmcnt(0): profiles/r06/train_probes.md).  One set: the layer's results are converted
        // out of it, its bias is read into it (32 LDS reads between two layers: ~300 of a layer's 65 000 cycles), and half the AGPRs are free.
        auto layer = [&](auto LIc) {
            f32x16 (&cur)[8] = acc; f32x16 (&prev)[8] = acc;
            constexpr int LI = decltype(LIc)::value, in_idx = LI - 1;
            constexpr bool relu_in = in_idx != 8;                    // the feature layer is linear (model.py:114)
            u32x4 bw = {0u, 0u, 0u, 0u};
            if constexpr (HOIST && LI == 9) {
                // step over the direction segment's records (DSG ring turns: the ring's phase stays) and prime the ring behind them, now --
                // the conversion below runs while they arrive
                sbase += (uint32_t)DSG * 8192u;
                static_for<RING>([&](auto I) { ws.r[I] = wr_load(ws, sbase + (uint32_t)decltype(I)::value * 1024u); });
            }
            // two tiles at a time: their 32 values out of the accumulators (ReLU, mask bits), then this layer's bias into the two tiles just
            // drained -- the LDS reads of a pair run under the conversion of the next
            static_for<4>([&](auto W) {
                constexpr int wd = W;
                if constexpr (relu_in) {
                    uint32_t w = 0;
                    static_for<4>([&](auto Q) { convert8_relu_bits<32 * wd + 8 * decltype(Q)::value>(hin, prev, w); });
                    bw[wd] = w;
                } else static_for<32>([&](auto V) { constexpr int v = 32 * wd + decltype(V)::value; hin[v] = prev[v >> 4][v & 15]; });
                const float *bsrc = (HOIST && LI == 9) ? rows_aux : aux;      // (HOIST: the colour layer starts from the ray's row)
                if constexpr (LI < 10 || wd < 2) { bias_tile<2 * wd>(cur, bsrc, LI, h); bias_tile<2 * wd + 1>(cur, bsrc, LI, h); }
            });
            if constexpr (relu_in) bits_leave(in_idx < 8 ? in_idx : 8, bw);
            if constexpr (LI == 8) {                                 // the density head rides on h7 (model.py:111): one dense block of FMAs
                const f32x4 *wa = reinterpret_cast<const f32x4 *>(aux + AUX_ALPHA_W + h * 128);
                static_for<32>([&](auto I) {
                    constexpr int i = I;
                    const f32x4 w = wa[i];
                    sig_part = __builtin_fmaf(hin[4 * i + 0], w.x, sig_part); sig_part = __builtin_fmaf(hin[4 * i + 1], w.y, sig_part);
                    sig_part = __builtin_fmaf(hin[4 * i + 2], w.z, sig_part); sig_part = __builtin_fmaf(hin[4 * i + 3], w.w, sig_part);
                });
            }
            __builtin_amdgcn_sched_barrier(0);
            leaves_to(in_idx, 8);
            if constexpr (LI == 5) seg_mem<PSG>(cur, ws, sbase, pb, fetch);        // concat[pos_map, h4] (model.py:107-108)
            if constexpr (LI == 9 && !HOIST) seg_mem<DSG>(cur, ws, sbase, pb, fetch);        // concat[dir_map, feature] (model.py:115)
            // the next memory-fed segment's first two groups are asked for a layer ahead (this block's position again for the skip, its
            // direction for the colour layer, the next block's position)
            if constexpr (LI == 4) rs_in = make_rsrc(a.pos + (size_t)blk * a.ptiles * 1024, (long long)a.ptiles * 4096);
            if constexpr (LI == 8 && !HOIST) rs_in = make_rsrc(a.dir + (size_t)blk * a.dtiles * 1024, (long long)a.dtiles * 4096);
            if constexpr (LI == 10) {
                const int nb = blk + nwaves < n_blocks ? blk + nwaves : blk;
                rs_in = make_rsrc(a.pos + (size_t)nb * a.ptiles * 1024, (long long)a.ptiles * 4096);
            }
            auto extra = [&](auto S, auto MT) {
                constexpr int s = decltype(S)::value, mt = decltype(MT)::value;
                if constexpr ((LI == 4 || (LI == 8 && !HOIST) || LI == 10) && mt == 2 && (s == 8 || s == 24)) fetch(std::integral_constant<int, (s == 8 ? 0 : 1)>{});
                if constexpr (mt == 5 || (LI == 10 && mt == 3)) leave(S);
            };
            if constexpr (LI == 10) seg_hidden<128, 4, false>(cur, ws, sbase, hin, extra);
            else seg_hidden<128, 8, false>(cur, ws, sbase, hin, extra);
        };
        static_for<10>([&](auto I) { layer(std::integral_constant<int, decltype(I)::value + 1>{}); });
        // ---- c2o (the 128-wide colour layer): ReLU, bits, out; the 3-wide colour head on the VALU (model.py:123)
        {
            u32x4 bw = {0u, 0u, 0u, 0u};
            convert_relu_bits<4>(hin, acc, bw);
            bits_leave(9, bw);
            leaves_to(10, 4);
            static_for<64>([&](auto V) { leave(V); });
            float rgb[3];
            static_for<3>([&](auto C) {
                constexpr int c = C;
                const f32x4 *wc = reinterpret_cast<const f32x4 *>(aux + AUX_RGB_W + (c * 2 + h) * 64);
                float p = 0.0f;
                static_for<16>([&](auto I) {
                    constexpr int i = I;
                    const f32x4 w = wc[i];
                    p = __builtin_fmaf(hin[4 * i + 0], w.x, p); p = __builtin_fmaf(hin[4 * i + 1], w.y, p);
                    p = __builtin_fmaf(hin[4 * i + 2], w.z, p); p = __builtin_fmaf(hin[4 * i + 3], w.w, p);
                });
                rgb[c] = p + __shfl_xor(p, 32, 64) + aux[AUX_RGB_B + c];
            });
            const float sigma = sig_part + __shfl_xor(sig_part, 32, 64) + aux[AUX_ALPHA_B];
            if (valid && h == 0) {
                a.sigma[m] = sigma;
                a.raw_rgb[3 * m + 0] = rgb[0]; a.raw_rgb[3 * m + 1] = rgb[1]; a.raw_rgb[3 * m + 2] = rgb[2];
            }
        }
    }
}

#endif   // NTX_TRAIN_FWD

// ---------------------------------------------------------------------------------------------------------------------------
// backward through the activations
// ---------------------------------------------------------------------------------------------------------------------------
struct DxArgs {
    const float *stream; uint32_t stream_bytes;        // the transposed weight stream, tail included
    long long M;
    const float *dgrad;                                // [M][4]: dL/d raw rgb (3), dL/d sigma, from the composite's adjoint
    float *out; long long out_stride;                  // O layout, out + i * out_stride: d c2o (128 rows), d c1o, d feature, dy7 .. dy0 (256 rows): what each layer's dW contracts with
    const unsigned int *bits; long long bits_stride;   // the forward pass's masks: h0 .. h7, c1o, c2o
};

#ifdef NTX_TRAIN_DX
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void dx_chain_kernel(DxArgs a) {
    const int lane = threadIdx.x & 63, n = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)), nwaves = gridDim.x * 4;
    const int n_blocks = (int)((a.M + 31) >> 5);
    if (wave >= n_blocks) return;
    WRingT<DX_RING> ws;
    ws.rsrc = make_rsrc(a.stream, a.stream_bytes);
    ws.voff = (uint32_t)lane * 16u;
    static_for<DX_RING>([&](auto I) { ws.r[I] = wr_load(ws, (uint32_t)decltype(I)::value * 1024u); });
    const uint32_t lane_o = o_lane_bytes(lane), lane16 = (uint32_t)lane * 16u;
    auto fetch_grad = [&](int blk) {
        const long long m = (long long)blk * 32 + n;
        const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
        return m < a.M ? *reinterpret_cast<const f32x4 *>(a.dgrad + 4 * m) : zero;
    };
    f32x4 g = fetch_grad(wave);
    for (int blk = wave; blk < n_blocks; blk += nwaves) {
        uint32_t sbase = 0;
        f32x16 acc[8];
        float hin[128];
        __amdgpu_buffer_rsrc_t rs_out = make_rsrc(a.out, 0);
        auto leaves_to = [&](int idx, int tiles) { rs_out = make_rsrc(a.out + (size_t)idx * a.out_stride + (size_t)blk * tiles * 1024, (long long)tiles * 4096); };
        auto leave = [&](auto V) {
            constexpr int v = V;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, hin[v]), rs_out, lane_o, o_value_bytes(v), STORE_NT);
        };
        auto fetch_bits = [&](int idx) {
            const __amdgpu_buffer_rsrc_t rb = make_rsrc(a.bits + (size_t)idx * a.bits_stride + (size_t)blk * 256, 1024);
            return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, lane16, 0, 0));
        };
        // ---- d c2o = (d raw . W_rgb^T) where c2o > 0: K = 3 as two k-steps (d_r, d_g), (d_b, 0)
        u32x4 bw = fetch_bits(9);
        const float gs = h ? 0.0f : g.w;
        {
            const float b[2] = {h ? g.y : g.x, h ? 0.0f : g.z};
            seg_few<2, 4, true>(acc, ws, sbase, b);
        }
        {
            const int nb = blk + nwaves < n_blocks ? blk + nwaves : blk;
            g = fetch_grad(nb);
        }
        convert_mask<4>(hin, acc, bw);
        // ---- d c1o = (d c2o . W_c2^T) where c1o > 0
        bw = fetch_bits(8);
        leaves_to(0, 4);
        seg_hidden<64, 8, true>(acc, ws, sbase, hin, [&](auto S, auto MT) { if constexpr (decltype(MT)::value == 5) leave(S); });
        convert_mask<8>(hin, acc, bw);
        // ---- d feature = d c1o . W_c1[dir_map rows skipped]^T (a linear layer: all of it)
        leaves_to(1, 8);
        seg_hidden<128, 8, true>(acc, ws, sbase, hin, [&](auto S, auto MT) { if constexpr (decltype(MT)::value == 5) leave(S); });
        convert_linear<8>(hin, acc);
        // ---- d h7 = (d feature . W_feature^T + d_sigma (x) W_alpha) where h7 > 0, then down the trunk
        static_for<8>([&](auto J) {
            constexpr int j = J;                               // writes d h(7 - j); reads out[2 + j] = d feature, dy7, dy6 ...
            bw = fetch_bits(7 - j);
            leaves_to(2 + j, 8);
            seg_hidden<128, 8, true>(acc, ws, sbase, hin, [&](auto S, auto MT) { if constexpr (decltype(MT)::value == 5) leave(S); });
            if constexpr (j == 0) {
                const float b[1] = {gs};
                seg_few<1, 8, false>(acc, ws, sbase, b);
            }
            convert_mask<8>(hin, acc, bw);
        });
        leaves_to(10, 8);
        static_for<128>([&](auto V) { leave(V); });
    }
}

#endif   // NTX_TRAIN_DX

// ---------------------------------------------------------------------------------------------------------------------------
// weight gradients
// ---------------------------------------------------------------------------------------------------------------------------
// dW = X^T . dY of every layer (reduction over the samples) in one launch of PERSISTENT workgroups, one per CU, each with an equal share of
// the launch's matrix work fixed before it starts: nothing is scheduled at run time, nobody waits for a last workgroup, and a layer's sum
// over the samples falls into about as many partial sums as the layer has CUs' worth of work (a fixed order: a step is bit-reproducible).
//
// A JOB is what one workgroup's four waves do side by side on the same blocks of 32 samples: the waves of a job share their operands (a
// 256 x 256 layer: wave w takes X tiles 4 (w >> 1) .. + 3 and dY tiles 4 (w & 1) .. + 3, so each of the layer's 16 tiles is read by two
// waves of ONE CU at the same moment and comes from HBM once).  A wave's share of a block: NA x NB tiles of 32 x 32 (rows: X's features,
// columns: dY's), 16 k-steps; both operands are read straight from their O-layout records, half a block (8 k-steps, 128 MFMAs) ahead.
//
// The split: the jobs' blocks are laid end to end, job j's block costing cost_j (the MFMAs of its slowest wave); workgroup g of G takes the
// cost interval [g W / G, (g + 1) W / G) of the total W, i.e. of job j the blocks [cut(g, j), cut(g + 1, j)) with
// cut(g, j) = clamp(round((g W / G - start_j) / cost_j), 0, n_blocks).  Its sums go to slot g - first_g(j) of the job's partial sums; every
// workgroup between the job's first and last writes its slot, were its range empty (zeros).
struct DwWave {
    const float *A; int rtA, a0;                       // X: tiles of 32 rows per block, the wave's first tile
    const float *B; int rtB, b0;                       // dY
    int shape;                                         // 0: 4 x 4 tiles, 1: 3 x 4, 2: 4 x 1; -1: the wave has nothing to do in this job
    long long out; int ldc;                            // the matrix within a slot of the job's partial sums (floats): out + row * ldc + (col - c_lo)
    int row0, rows_valid, col0, c_lo, c_hi;            // the wave's first row / column, and what of the tiles exists
    long long bias_out;                                // -1, or within the slot: [2][c_hi - c_lo], the column sums of dY (two halves of the samples)
};
struct DwJob { DwWave w[4]; int cost; long long slot_floats, first_float; };      // first_float: the job's slot 0 in the partial buffer
struct DwArgs { const DwJob *jobs; int n_jobs, n_blocks; long long total_cost; float *partial;      // total_cost: sum of the jobs' costs (per block)
                unsigned long long *clocks; };   // development builds (-DNTX_TRAIN_CLOCKS): [workgroup][2 + 3 n_jobs] wall clock at start / end and per piece (blocks, start, end)

// the split, on both sides (the reduction has to know how many slots a job filled)
__host__ __device__ inline long long dw_cut(long long g, long long G, long long W, long long start, long long cost, long long n_blocks) {
    const long long x = g * W / G - start;             // cost units into the job
    long long b = x <= 0 ? 0 : (x + cost / 2) / cost;
    return b > n_blocks ? n_blocks : b;
}
__host__ __device__ inline long long dw_first_g(long long G, long long W, long long start) { return start * G / W; }                 // the workgroup whose interval holds the job's first unit
__host__ __device__ inline long long dw_last_g(long long G, long long W, long long start, long long span) { return (start + span - 1) * G / W; }

#ifdef NTX_TRAIN_DW
#ifndef NTX_DW_SYNC
#define NTX_DW_SYNC 16
#endif
constexpr int DW_SYNC = NTX_DW_SYNC;               // blocks of 32 samples between two meetings of a job's waves (a power of two)
template <int NA, int NB>
TRN_DEV void dw_piece(const DwWave &tg, float *slot, int blk0, int blk1, int lane) {
    auto uniform_ptr = [](const float *p) {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)p), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uintptr_t)p >> 32));
        return (const float *)(uintptr_t)(((uint64_t)hi << 32) | (uint64_t)lo);
    };
    const float *A = uniform_ptr(tg.A), *B = uniform_ptr(tg.B);
    const int rtA = __builtin_amdgcn_readfirstlane(tg.rtA), a0 = __builtin_amdgcn_readfirstlane(tg.a0), rtB = __builtin_amdgcn_readfirstlane(tg.rtB), b0 = __builtin_amdgcn_readfirstlane(tg.b0);
    f32x16 acc[NA][NB];
    static_for<NA>([&](auto Ai) { static_for<NB>([&](auto Bi) {
        acc[Ai][Bi] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; }); });
    f32x4 bsum[NB];
    static_for<NB>([&](auto Bi) { bsum[Bi] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; });
    const bool want_bias = tg.bias_out >= 0;
    const int nblk = blk1 - blk0;
    if (nblk > 0) {
        const __amdgpu_buffer_rsrc_t ra = make_rsrc(A + (size_t)blk0 * rtA * 1024, (long long)nblk * rtA * 4096);
        const __amdgpu_buffer_rsrc_t rb = make_rsrc(B + (size_t)blk0 * rtB * 1024, (long long)nblk * rtB * 4096);
        const uint32_t voff = (uint32_t)lane * 16u;
        const uint32_t stepA = (uint32_t)rtA * 4096u, stepB = (uint32_t)rtB * 4096u, offA = (uint32_t)a0 * 4096u, offB = (uint32_t)b0 * 4096u;
        f32x4 xa[2][NA][2], xb[2][NB][2];                  // two halves of a block in flight: [half][tile][record of the half]
        auto fetch = [&](auto BUF, int i) {                // half BUF of block blk0 + i
            constexpr int buf = BUF;
#ifdef NTX_X_DWSMALL
            i &= 3;
#endif
#ifdef NTX_X_DWNOLOAD
            if (i > 0) return;
#endif
            const uint32_t oa = (uint32_t)i * stepA + offA + (uint32_t)buf * 2048u, ob = (uint32_t)i * stepB + offB + (uint32_t)buf * 2048u;
            static_for<NA>([&](auto Ai) { static_for<2>([&](auto Q) {
                xa[buf][Ai][Q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, voff, oa + (uint32_t)(decltype(Ai)::value * 4 + decltype(Q)::value) * 1024u, 0)); }); });
            static_for<NB>([&](auto Bi) { static_for<2>([&](auto Q) {
                xb[buf][Bi][Q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, voff, ob + (uint32_t)(decltype(Bi)::value * 4 + decltype(Q)::value) * 1024u, 0)); }); });
        };
        auto compute = [&](auto BUF) {
            constexpr int buf = BUF;
            static_for<2>([&](auto Q) { static_for<4>([&](auto C) {
                constexpr int q = Q, c = C;
                static_for<NA>([&](auto Ai) { static_for<NB>([&](auto Bi) {
                    acc[Ai][Bi] = mfma32(xa[buf][Ai][q][c], xb[buf][Bi][q][c], acc[Ai][Bi]);
                }); });
            }); });
            if (want_bias) static_for<NB>([&](auto Bi) { bsum[Bi] += xb[buf][Bi][0] + xb[buf][Bi][1]; });
        };
        // The fetches are UNCONDITIONAL (behind the range's end the last block is asked for again and not used): with a branch around a fetch
        // the compiler cannot count how many loads are younger than the ones it waits for, and waits for all of them
        fetch(std::integral_constant<int, 0>{}, 0);
        for (int i = 0; i < nblk; ++i) {
            // Every DW_SYNC blocks the job's four waves wait for each other (a wave without work in this job keeps the count: dw_kernel): they
            // drift apart otherwise, and what two of them share is then read from HBM twice.  Measured on one box, kernel time / HBM-side read a
            // step: never 2.69 ms / 8.4 GB; every block 2.85 ms / 6.7 GB (the waves wait for each other's loads); every 8 blocks 2.69 / 6.7;
            // every 16 or 64 blocks 2.67 / 6.7.
            if ((i & (DW_SYNC - 1)) == 0) __builtin_amdgcn_s_barrier();
            fetch(std::integral_constant<int, 1>{}, i);
            __builtin_amdgcn_sched_barrier(0);
            compute(std::integral_constant<int, 0>{});
            __builtin_amdgcn_sched_barrier(0);
            fetch(std::integral_constant<int, 0>{}, i + 1 < nblk ? i + 1 : i);
            __builtin_amdgcn_sched_barrier(0);
            compute(std::integral_constant<int, 1>{});
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // D of a tile: lane l, register r <-> row 8 (r >> 2) + (r & 3) + 4 (l >> 5), column l & 31.  What of a tile exists is a matter of
    // offsets, not of branches: a row or column beyond the matrix gets an offset beyond the buffer, and the hardware drops the store.
    const int j = lane & 31, hh = lane >> 5;
    const int ldc = __builtin_amdgcn_readfirstlane(tg.ldc), row0 = __builtin_amdgcn_readfirstlane(tg.row0), rows_valid = __builtin_amdgcn_readfirstlane(tg.rows_valid);
    const int col0 = __builtin_amdgcn_readfirstlane(tg.col0), c_lo = __builtin_amdgcn_readfirstlane(tg.c_lo), c_hi = __builtin_amdgcn_readfirstlane(tg.c_hi);
    const long long at_out = (long long)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)tg.out >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uint64_t)tg.out));
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(slot + at_out, (long long)rows_valid * ldc * 4);
    constexpr uint32_t NOWHERE = 0x7ffffff0u;
    static_for<NB>([&](auto Bi) {
        const int col = col0 + 32 * decltype(Bi)::value + j;
        const bool col_ok = col >= c_lo && col < c_hi;
        const uint32_t coff = (uint32_t)(col - c_lo) * 4u;
        static_for<NA>([&](auto Ai) {
            static_for<16>([&](auto R) {
                constexpr int r = R;
                const int row = row0 + 32 * decltype(Ai)::value + 8 * (r >> 2) + (r & 3) + 4 * hh;
                const uint32_t off = (col_ok && row < rows_valid) ? (uint32_t)(row * ldc) * 4u + coff : NOWHERE;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, (float)acc[Ai][Bi][r]), ro, off, 0, 0);
            });
        });
    });
    if (want_bias) {
        const long long at_bias = (long long)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)tg.bias_out >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uint64_t)tg.bias_out));
        float *bo = slot + at_bias + (size_t)hh * (c_hi - c_lo);
        static_for<NB>([&](auto Bi) {
            const int col = col0 + 32 * decltype(Bi)::value + j;
            const f32x4 sm = bsum[Bi];
            if (col >= c_lo && col < c_hi) bo[col - c_lo] = (sm.x + sm.y) + (sm.z + sm.w);
        });
    }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void dw_kernel(DwArgs a) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long g = blockIdx.x, G = gridDim.x, W = a.total_cost * a.n_blocks;
    long long start = 0;
#ifdef NTX_TRAIN_CLOCKS
    unsigned long long *ck = a.clocks ? a.clocks + (size_t)blockIdx.x * (2 + 3 * a.n_jobs) : nullptr;
    if (ck && threadIdx.x == 0) ck[0] = wall_clock64();
#endif
    for (int jn = 0; jn < a.n_jobs; ++jn) {
        const DwJob &job = a.jobs[jn];
        const long long cost = job.cost, span = cost * a.n_blocks;
        const long long g0 = dw_first_g(G, W, start), g1 = dw_last_g(G, W, start, span);
        if (__builtin_amdgcn_readfirstlane((int)(g >= g0 && g <= g1))) {
            // (64-bit divisions: the compiler no longer sees that their results are the same in every lane)
            const int blk0 = __builtin_amdgcn_readfirstlane((int)dw_cut(g, G, W, start, cost, a.n_blocks)), blk1 = __builtin_amdgcn_readfirstlane((int)dw_cut(g + 1, G, W, start, cost, a.n_blocks));
            const DwWave &wv = job.w[wave];
            const int shape = __builtin_amdgcn_readfirstlane(wv.shape);
            const int my_slot = __builtin_amdgcn_readfirstlane((int)(g - g0));
            float *slot = a.partial + job.first_float + (long long)my_slot * job.slot_floats;
#ifdef NTX_TRAIN_CLOCKS
            const unsigned long long c0 = wall_clock64();
#endif
            if (shape == 0) dw_piece<4, 4>(wv, slot, blk0, blk1, lane);
            else if (shape == 1) dw_piece<3, 4>(wv, slot, blk0, blk1, lane);
            else if (shape == 2) dw_piece<4, 1>(wv, slot, blk0, blk1, lane);
            else for (int i = 0; i < blk1 - blk0; i += DW_SYNC) __builtin_amdgcn_s_barrier();
#ifdef NTX_TRAIN_CLOCKS
            __syncthreads();
            if (ck && threadIdx.x == 0) { ck[2 + 3 * jn] = (unsigned long long)(blk1 - blk0); ck[3 + 3 * jn] = c0; ck[4 + 3 * jn] = wall_clock64(); }
#endif
        }
        start += span;
    }
#ifdef NTX_TRAIN_CLOCKS
    if (ck && threadIdx.x == 0) ck[1] = wall_clock64();
#endif
}
#endif   // NTX_TRAIN_DW

// the launchers of ntx_train_chain.hip (one object per kernel: each takes minutes to compile).  The forward chain exists for the segment
// lengths of the shipped families -- (9, 11) carpet, (9, 8) grass / fur / plush, (11, 8) grass_filtered -- and for the longest (12, 12); a model
// runs on the smallest build that holds its segments, its streams padded with zero rows.  (grass_filtered needs (11, 7): that build, like every
// one tried with a direction segment shorter than 8 groups, makes hipcc spill four bias tiles -- 64 scratch instructions a block, each reload
// behind a full s_waitcnt vmcnt(0); one group of zero rows more costs 32 MFMAs of 10 656 and none of that.  profiles/r06/train_probes.md)
constexpr int FWD_VARIANTS[4][2] = {{9, 11}, {9, 8}, {11, 8}, {MAX_PB_GROUPS, MAX_PB_GROUPS}};
void launch_fwd_chain(int variant, bool hoist, hipStream_t st, unsigned grid, const FwdArgs &a);
void launch_dx_chain(hipStream_t st, unsigned grid, const DxArgs &a);
void launch_dw(hipStream_t st, unsigned grid, const DwArgs &a);

}   // namespace ntx_train
