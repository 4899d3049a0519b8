// ntx_device.h -- gfx950 device code of the NeRF-Tex render path (included by nerftex.hip only).
//
// Kernels (all float32, one wave64 = one batch of 32 samples, activations in registers):
//   render_kernel<CFG>      rays -> premultiplied RGBA   (renderer.py:47-213 fused)
//   mlp_kernel<CFG>         (pos, dir, params) -> (raw rgb, raw sigma)   (model.py:58-125)
//   composite_kernel        map_model_output alone        (renderer.py:170-213)
//   raygen_kernel           rays_from_camera + Proxy/AABB (ray_sampler.py:23-48, proxy.py:13-35)
//   fourier_kernel          FourierFeatures alone         (layer.py:8-23)
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#include <utility>

#include "ntx_layout.h"

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
    uint32_t soff;   // wave-uniform byte offset of the next record to consume
    uint32_t voff;   // lane * 16
    f32x4 ring[RING];
};

NTX_DEV f32x4 ws_load(const WStream &ws, uint32_t rec_ahead) {
    const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ws.rsrc, ws.voff, ws.soff + rec_ahead * 1024u, 0);
    return __builtin_bit_cast(f32x4, v);
}

NTX_DEV void ws_prime(WStream &ws, const f32x4 *base, uint32_t stream_bytes, int lane) {
    ws.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<f32x4 *>(base), 0, stream_bytes, 0x00020000);
    ws.soff = 0;
    ws.voff = (uint32_t)lane * 16u;
    static_for<RING>([&](auto I) { ws.ring[I] = ws_load(ws, I); });
}

NTX_DEV f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// acc[mt] += W_segment^T * B over NSTEPS k-steps; bfn(integral_constant<s>) is this lane's B value.
template <int NSTEPS, int NMT, class BFn>
NTX_DEV void run_segment(f32x16 (&acc)[8], WStream &ws, BFn &&bfn) {
    constexpr int RPS = NMT / 4;
    static_assert((NSTEPS * RPS) % RING == 0, "segment must be a whole number of ring turns");
    float b = bfn(std::integral_constant<int, 0>{});
    static_for<NSTEPS>([&](auto S) {
        constexpr int s = S;
        // B value of the NEXT k-step is produced inside this step's scheduling region, so its VALU
        // work (positional encoding) can interleave with this step's MFMAs
        float bn = 0.0f;
        if constexpr (s + 1 < NSTEPS) bn = bfn(std::integral_constant<int, s + 1>{});
        static_for<RPS>([&](auto Q) {
            constexpr int q = Q;
            constexpr int rec = s * RPS + q;
            constexpr int slot = rec % RING;
            const f32x4 w = ws.ring[slot];
            ws.ring[slot] = ws_load(ws, rec + RING);
            acc[4 * q + 0] = mfma32(w.x, b, acc[4 * q + 0]);
            acc[4 * q + 1] = mfma32(w.y, b, acc[4 * q + 1]);
            acc[4 * q + 2] = mfma32(w.z, b, acc[4 * q + 2]);
            acc[4 * q + 3] = mfma32(w.w, b, acc[4 * q + 3]);
        });
        b = bn;
        // Nothing may be scheduled across a k-step boundary: hipcc otherwise sinks every prefetch
        // load down to its first use (load; s_waitcnt vmcnt(0); mfma) and the ring collapses.
        __builtin_amdgcn_sched_barrier(0);
    });
    ws.soff += NSTEPS * RPS * 1024u;
}

// accumulators <- bias, straight from the LDS copy of the aux block ([half][128] per layer)
template <int NMT>
NTX_DEV void init_bias(f32x16 (&acc)[8], const float *aux, int layer, int h) {
    const f32x4 *b = reinterpret_cast<const f32x4 *>(aux + layer * AUX_BIAS_STRIDE + h * 128);
    static_for<NMT>([&](auto MT) {
        constexpr int mt = MT;
        static_for<4>([&](auto Q) {
            constexpr int q = Q;
            const f32x4 v = b[mt * 4 + q];
            acc[mt][4 * q + 0] = v.x; acc[mt][4 * q + 1] = v.y;
            acc[mt][4 * q + 2] = v.z; acc[mt][4 * q + 3] = v.w;
        });
    });
}

template <int NMT, bool RELU>
NTX_DEV void store_act(float (&hin)[128], const f32x16 (&acc)[8]) {
    static_for<NMT>([&](auto MT) {
        constexpr int mt = MT;
        static_for<16>([&](auto R) {
            constexpr int r = R;
            const float v = acc[mt][r];
            hin[16 * mt + r] = RELU ? __builtin_fmaxf(v, 0.0f) : v;
        });
    });
}

// ---------------------------------------------------------------------------------------------
// per-lane inputs of one sample and the positional-encoding k-steps (layer.py:8-23)
// ---------------------------------------------------------------------------------------------
template <int NGEO, int NAPP>
struct SampleIn {
    float pos[3];
    float dir[3];
    float par[NGEO + NAPP > 0 ? NGEO + NAPP : 1];
};

// Copy of the inputs whose values the optimiser cannot relate to the original: without it, CSE keeps
// the 36-44 encoded position features alive from layer 0 to the skip layer and LICM hoists the
// direction features out of the layer loop (~80 VGPRs), pushing the kernel into spill/serialise
// mode.  Recomputing them in place costs VALU slots that sit in the shadow of the MFMAs.
template <int NGEO, int NAPP>
NTX_DEV SampleIn<NGEO, NAPP> launder(const SampleIn<NGEO, NAPP> &in) {
    SampleIn<NGEO, NAPP> o = in;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        asm volatile("" : "+v"(o.pos[k]));
        asm volatile("" : "+v"(o.dir[k]));
    }
#pragma unroll
    for (int k = 0; k < NGEO + NAPP; ++k) asm volatile("" : "+v"(o.par[k]));
    return o;
}

// identity value v of the position segment: [pos(3) | geometry params]
template <int NGEO, int NAPP, int V>
NTX_DEV float pos_id_value(const SampleIn<NGEO, NAPP> &in) {
    if constexpr (V < 3) return in.pos[V];
    else if constexpr (V - 3 < NGEO) return in.par[V - 3];
    else return 0.0f;
}
// identity value v of the direction segment: [dir(3) | appearance params]
template <int NGEO, int NAPP, int V>
NTX_DEV float dir_id_value(const SampleIn<NGEO, NAPP> &in) {
    if constexpr (V < 3) return in.dir[V];
    else if constexpr (V - 3 < NAPP) return in.par[NGEO + V - 3];
    else return 0.0f;
}

template <int NGEO, int NAPP, int S>
NTX_DEV float pos_feature(const SampleIn<NGEO, NAPP> &in, int h) {
    constexpr int nid = pos_id_steps(NGEO);
    if constexpr (S < nid) {
        const float lo = pos_id_value<NGEO, NAPP, 2 * S>(in), hi = pos_id_value<NGEO, NAPP, 2 * S + 1>(in);
        return h ? hi : lo;
    } else if constexpr (S - nid < 3 * POS_FREQ) {
        constexpr int q = S - nid, f = q / 3, c = q % 3;
        return sin_q(in.pos[c] * (float)(1 << f), h);
    } else if constexpr (S - nid - 3 * POS_FREQ < NGEO * PAR_FREQ) {
        constexpr int q = S - nid - 3 * POS_FREQ, f = q / NGEO, g = q % NGEO;
        return sin_q(in.par[g] * (float)(1 << f), h);
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
        constexpr int q = S - nid - 3 * DIR_FREQ, f = q / NAPP, a = q % NAPP;
        return sin_q(in.par[NGEO + a] * (float)(1 << f), h);
    } else {
        return 0.0f;
    }
}

// ---------------------------------------------------------------------------------------------
// the MLP on one batch of 32 samples (model.py:58-125 / 9-45); lanes l and l+32 hold sample l&31
// ---------------------------------------------------------------------------------------------
template <int NGEO_, int NAPP_, int CD_>
struct Cfg {
    static constexpr int NGEO = NGEO_, NAPP = NAPP_, CD = CD_;
    static constexpr int NP = NGEO_ + NAPP_;
    static constexpr int PS = pos_steps(NGEO_);
    static constexpr int DS = dir_steps(NAPP_, CD_ ? 4 : 8);
};

template <class CFG>
NTX_DEV void mlp_batch(const SampleIn<CFG::NGEO, CFG::NAPP> &in, WStream &ws,
                       const float *aux_in, int lane, float &sigma, float (&rgb)[3]) {
    constexpr int NGEO = CFG::NGEO, NAPP = CFG::NAPP;
    const int h = lane >> 5;
    // The aux block in LDS never changes, so the optimiser would hoist every bias / head-weight
    // read out of the batch loop and then spill ~600 values to scratch.  Laundering the pointer
    // keeps each ds_read next to its use.
    uint32_t opaque_zero = 0;
    asm volatile("" : "+v"(opaque_zero));   // an opaque OFFSET (not pointer) keeps the LDS address space
    const float *aux = aux_in + opaque_zero;
    f32x16 acc[8];
    float hin[128];

    // ---- trunk layer 0: pos_map -> 256 (model.py:104-106)
    init_bias<8>(acc, aux, 0, h);
    run_segment<CFG::PS, 8>(acc, ws, [&](auto S) { return pos_feature<NGEO, NAPP, decltype(S)::value>(in, h); });
    store_act<8, true>(hin, acc);

    // ---- hidden passes: L1..L7, then F (linear), then (ParamNerf) C1.  One 1024-MFMA body.
    constexpr int NPASS = 8 + (CFG::CD ? 1 : 0);
    for (int li = 1; li <= NPASS; ++li) {
        init_bias<8>(acc, aux, li, h);
        if (li == SKIP + 1) {   // input = concat[pos_map, h]  (model.py:107-108)
            const SampleIn<NGEO, NAPP> in2 = launder(in);
            run_segment<CFG::PS, 8>(acc, ws, [&](auto S) { return pos_feature<NGEO, NAPP, decltype(S)::value>(in2, h); });
        }
        if constexpr (CFG::CD != 0)
            if (li == 9) {   // input = concat[dir_map, feature]  (model.py:115)
                const SampleIn<NGEO, NAPP> in2 = launder(in);
                run_segment<CFG::DS, 8>(acc, ws, [&](auto S) { return dir_feature<NGEO, NAPP, decltype(S)::value>(in2, h); });
            }
        run_segment<HSTEPS, 8>(acc, ws, [&](auto S) { return hin[decltype(S)::value]; });
        if (li == 8) {
            store_act<8, false>(hin, acc);   // "feature" layer has no activation (model.py:114)
        } else {
            store_act<8, true>(hin, acc);
        }
        if (li == DEPTH - 1) {
            // alpha head on the output of trunk layer 7 (model.py:111), on the VALU:
            // each half-wave holds 128 of the 256 features of its sample
            const f32x4 *wa = reinterpret_cast<const f32x4 *>(aux + aux_alpha_off() + h * 128);
            float p = 0.0f;
            static_for<32>([&](auto I) {
                constexpr int i = I;
                const f32x4 w = wa[i];
                p = __builtin_fmaf(hin[4 * i + 0], w.x, p);
                p = __builtin_fmaf(hin[4 * i + 1], w.y, p);
                p = __builtin_fmaf(hin[4 * i + 2], w.z, p);
                p = __builtin_fmaf(hin[4 * i + 3], w.w, p);
            });
            sigma = p + __shfl_xor(p, 32, 64) + aux[aux_alpha_off() + 256];
        }
    }

    // ---- colour half layer (-> 128, relu; model.py:122 / 42), 4 M-tiles
    init_bias<4>(acc, aux, 10, h);
    if constexpr (CFG::CD == 0) {   // plain Nerf: input = concat[dir_map, feature]  (model.py:39-42)
        const SampleIn<NGEO, NAPP> in2 = launder(in);
        run_segment<CFG::DS, 4>(acc, ws, [&](auto S) { return dir_feature<NGEO, NAPP, decltype(S)::value>(in2, h); });
    }
    run_segment<HSTEPS, 4>(acc, ws, [&](auto S) { return hin[decltype(S)::value]; });
    store_act<4, true>(hin, acc);

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

    // the stream's tail replicates its first RING records, so the ring already holds the head
    ws.soff = 0;
}

NTX_DEV void load_aux(float *lds, const float *aux_g, int n) {
    for (int i = threadIdx.x; i < n / 4; i += blockDim.x)
        reinterpret_cast<f32x4 *>(lds)[i] = reinterpret_cast<const f32x4 *>(aux_g)[i];
    __syncthreads();
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

template <int W>
NTX_DEV void composite_step(RayAccum &ra, float sigma, const float (&raw)[3], float dist, bool valid,
                            uint32_t flags, int j, float *w_out) {
    float c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) c[k] = (flags & NTX_FLAG_MAP_EXR) ? elu1f_(raw[k]) : sigmoidf_(raw[k]);
    const float a = valid ? 1.0f - expf(-__builtin_fmaxf(sigma, 0.0f) * dist) : 0.0f;   // :195
    const float trans = (1.0f - a) + 1e-10f;                                          // :198
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

// ---------------------------------------------------------------------------------------------
// fused render kernel: one wave per ray, S/32 batches per ray (renderer.py:47-213)
// ---------------------------------------------------------------------------------------------
struct RenderArgs {
    const f32x4 *wstream;
    uint32_t stream_bytes;   // stream + wrap-around tail
    const float *aux;
    const float *rays_o, *rays_d, *t, *params, *cone, *z_vals;
    float *color_out, *alpha_out;
    int32_t *status;
    int64_t n_rays, rays_per_row;
    int n_samples, blur_idx;
    uint32_t flags;
    float delta;   // float32(1 / (S - 1)): the step of tf.linspace(0., 1., S)
    float bkgd[3];
};

NTX_DEV float z_of(const RenderArgs &a, int64_t ray, int i, float t0, float t1) {
    if (a.z_vals) return a.z_vals[ray * a.n_samples + i];
    const float tv = i == 0 ? 0.0f : (i == a.n_samples - 1 ? 1.0f : a.delta * (float)i);
    return t0 * (1.0f - tv) + t1 * tv;   // renderer.py:102
}

template <class CFG>
__global__ __launch_bounds__(256) void render_kernel(RenderArgs a) {
    __shared__ __attribute__((aligned(16))) float aux[aux_total()];
    load_aux(aux, a.aux, aux_total());
    const int lane = threadIdx.x & 63, j = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    const int nwaves = gridDim.x * 4;
    const int S = a.n_samples;
    const int nb = (S + 31) >> 5;
    WStream ws;
    ws_prime(ws, a.wstream, a.stream_bytes, lane);

    for (int64_t ray = wave; ray < a.n_rays; ray += nwaves) {
        const float t0 = a.t[2 * ray], t1 = a.t[2 * ray + 1];
        if (t0 == __builtin_inff()) {   // culled ray (renderer.py:58-67, 81-86); NaN counts as a hit
            if (lane < 3) a.color_out[3 * ray + lane] = (a.flags & NTX_FLAG_COMPOSITE_BKGD) ? a.bkgd[lane] : 0.0f;
            if (lane == 3) a.alpha_out[ray] = 0.0f;
            continue;
        }
        const float ox = a.rays_o[3 * ray], oy = a.rays_o[3 * ray + 1], oz = a.rays_o[3 * ray + 2];
        const float dx = a.rays_d[3 * ray], dy = a.rays_d[3 * ray + 1], dz = a.rays_d[3 * ray + 2];
        const float dnorm = __builtin_sqrtf(dx * dx + dy * dy + dz * dz);   // renderer.py:98, 180
        const float cone = a.cone ? a.cone[ray] : 0.0f;
        const float *prow = a.params + (ray / a.rays_per_row) * CFG::NP;

        RayAccum ra{1.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        for (int b = 0; b < nb; ++b) {
            const int i = 32 * b + j;
            const bool valid = i < S;
            const int ic = valid ? i : S - 1;
            const float z = z_of(a, ray, ic, t0, t1);
            // dists: z[i+1]-z[i], the last one a copy of the previous (renderer.py:174-177), times |d| (:180)
            const float zn = z_of(a, ray, ic < S - 1 ? ic + 1 : ic - 1, t0, t1);
            const float dist = (ic < S - 1 ? zn - z : z - zn) * dnorm;

            SampleIn<CFG::NGEO, CFG::NAPP> in;
            in.pos[0] = ox + dx * z; in.pos[1] = oy + dy * z; in.pos[2] = oz + dz * z;   // renderer.py:114
            in.dir[0] = dx / dnorm; in.dir[1] = dy / dnorm; in.dir[2] = dz / dnorm;      // rays_d_n
#pragma unroll
            for (int k = 0; k < CFG::NP; ++k) {
                float p = prow[k];
                if (k == a.blur_idx) p = p * (cone * z);                                  // renderer.py:155-158
                in.par[k] = p;
            }
            float sigma, raw[3];
            mlp_batch<CFG>(in, ws, aux, lane, sigma, raw);
            composite_step<32>(ra, sigma, raw, dist, valid, a.flags, j, nullptr);
        }
        float out[4] = {ra.c0, ra.c1, ra.c2, ra.a};
        if (a.flags & NTX_FLAG_COMPOSITE_BKGD) {   // renderer.py:210-211
#pragma unroll
            for (int k = 0; k < 3; ++k) out[k] = out[k] + (1.0f - ra.a) * a.bkgd[k];
        }
        if (lane == 0) {
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
// stand-alone MLP: samples are independent, 32 per wave
// ---------------------------------------------------------------------------------------------
struct MlpArgs {
    const f32x4 *wstream;
    uint32_t stream_bytes;
    const float *aux;
    const float *pos, *dirs, *params;
    float *color_out, *sigma_out;
    int64_t m;
};

template <class CFG>
__global__ __launch_bounds__(256) void mlp_kernel(MlpArgs a) {
    __shared__ __attribute__((aligned(16))) float aux[aux_total()];
    load_aux(aux, a.aux, aux_total());
    const int lane = threadIdx.x & 63, j = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    const int nwaves = gridDim.x * 4;
    WStream ws;
    ws_prime(ws, a.wstream, a.stream_bytes, lane);
    const int64_t nbatch = (a.m + 31) >> 5;
    for (int64_t b = wave; b < nbatch; b += nwaves) {
        const int64_t m = b * 32 + j;
        const bool valid = m < a.m;
        const int64_t mc = valid ? m : a.m - 1;
        SampleIn<CFG::NGEO, CFG::NAPP> in;
#pragma unroll
        for (int k = 0; k < 3; ++k) { in.pos[k] = a.pos[3 * mc + k]; in.dir[k] = a.dirs[3 * mc + k]; }
#pragma unroll
        for (int k = 0; k < CFG::NP; ++k) in.par[k] = a.params[CFG::NP * mc + k];
        float sigma, raw[3];
        mlp_batch<CFG>(in, ws, aux, lane, sigma, raw);
        if (valid && lane < 32) {
            a.color_out[3 * m + 0] = raw[0]; a.color_out[3 * m + 1] = raw[1]; a.color_out[3 * m + 2] = raw[2];
            a.sigma_out[m] = sigma;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// stand-alone composite: one wave64 per ray, lane = sample, chunks of 64 with carry
// ---------------------------------------------------------------------------------------------
struct CompositeArgs {
    const float *color, *sigma, *z, *rays_d;
    float *color_out, *alpha_out, *weights_out;
    int64_t n_rays;
    int n_samples;
    uint32_t flags;
    float bkgd[3];
};

__global__ __launch_bounds__(256) void composite_kernel(CompositeArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const int S = a.n_samples;
    for (int64_t ray = wave; ray < a.n_rays; ray += nwaves) {
        const float dx = a.rays_d[3 * ray], dy = a.rays_d[3 * ray + 1], dz = a.rays_d[3 * ray + 2];
        const float dnorm = __builtin_sqrtf(dx * dx + dy * dy + dz * dz);
        const float *zr = a.z + ray * S;
        RayAccum ra{1.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        for (int base = 0; base < S; base += 64) {
            const int i = base + lane;
            const bool valid = i < S;
            const int ic = valid ? i : S - 1;
            const float z = zr[ic];
            const float zn = zr[ic < S - 1 ? ic + 1 : ic - 1];
            const float dist = (ic < S - 1 ? zn - z : z - zn) * dnorm;
            const float sg = a.sigma[ray * S + ic];
            float raw[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) raw[k] = a.color[(ray * S + ic) * 3 + k];
            composite_step<64>(ra, sg, raw, dist, valid, a.flags, lane,
                               a.weights_out ? a.weights_out + ray * S + ic : nullptr);
        }
        float out[4] = {ra.c0, ra.c1, ra.c2, ra.a};
        if (a.flags & NTX_FLAG_COMPOSITE_BKGD) {
#pragma unroll
            for (int k = 0; k < 3; ++k) out[k] = out[k] + (1.0f - ra.a) * a.bkgd[k];
        }
        if (lane == 0) {
            a.color_out[3 * ray + 0] = out[0]; a.color_out[3 * ray + 1] = out[1];
            a.color_out[3 * ray + 2] = out[2]; a.alpha_out[ray] = out[3];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// ray generation (pixel_sampler.py:14-15, ray_sampler.py:23-48, proxy.py:13-35)
// ---------------------------------------------------------------------------------------------
struct RaygenArgs {
    float c2w[16];
    float b0[3], b1[3];
    float focal, half_w, half_h, near_t, far_t;
    int width, mode;
    int64_t pixel0, n;
    float *rays_o, *rays_d, *t, *cone;
};

__global__ __launch_bounds__(256) void raygen_kernel(RaygenArgs a) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.n) return;
    const int64_t pix = a.pixel0 + k;
    const float li = (float)(pix / a.width), lj = (float)(pix % a.width);   // (row, col), Full sampler
    const float d0 = (lj + 0.5f - a.half_w) / a.focal;                       // ray_sampler.py:41
    const float d1 = -(li + 0.5f - a.half_h) / a.focal;
    const float d2 = -1.0f;
    float rd[3], ro[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        rd[r] = (d0 * a.c2w[4 * r + 0] + d1 * a.c2w[4 * r + 1]) + d2 * a.c2w[4 * r + 2];   // :42
        ro[r] = a.c2w[4 * r + 3];                                                            // :43
    }
    const float nxy = __builtin_sqrtf(d0 * d0 + d1 * d1);
    const float nrm = __builtin_sqrtf((d0 * d0 + d1 * d1) + d2 * d2);
    const float cone = cosf(atanf(nxy)) / nrm / a.focal;                                    // :46
    float t0, t1;
    if (a.mode == 0) {
        const float n = __builtin_sqrtf((rd[0] * rd[0] + rd[1] * rd[1]) + rd[2] * rd[2]);  // :34
#pragma unroll
        for (int r = 0; r < 3; ++r) rd[r] = rd[r] / n;
        // proxy.py:16-33; comparisons written exactly as tf.where does them so NaNs fall the same way
        float tmax = 0.0f, tmin = 0.0f;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float inv = 1.0f / rd[r];
            const float ta = (a.b0[r] - ro[r]) * inv, tb = (a.b1[r] - ro[r]) * inv;
            const float lo = ta < tb ? ta : tb;
            const float hi = ta > tb ? ta : tb;
            if (r == 0) { tmax = lo; tmin = hi; }
            else {
                // reduce_max / reduce_min propagate NaN
                tmax = (lo != lo || tmax != tmax) ? __builtin_nanf("") : (lo > tmax ? lo : tmax);
                tmin = (hi != hi || tmin != tmin) ? __builtin_nanf("") : (hi < tmin ? hi : tmin);
            }
        }
        const bool hit = tmax < tmin;
        t0 = hit ? tmax : __builtin_inff();
        t1 = hit ? tmin : __builtin_inff();
    } else {
        t0 = a.near_t; t1 = a.far_t;
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) { a.rays_o[3 * k + r] = ro[r]; a.rays_d[3 * k + r] = rd[r]; }
    a.t[2 * k] = t0; a.t[2 * k + 1] = t1;
    a.cone[k] = cone;
}

// ---------------------------------------------------------------------------------------------
// FourierFeatures alone (layer.py:8-23): thread per (row, component)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fourier_kernel(const float *x, int64_t m, int d, int nf, float *out) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= m * d) return;
    const int64_t row = k / d;
    const int c = (int)(k % d);
    const float v = x[k];
    float *o = out + row * (int64_t)(d * (1 + 2 * nf));
    o[c] = v;
    float f = 1.0f;
    for (int i = 0; i < nf; ++i) {
        o[d + 2 * i * d + c] = sin_q(f * v, 0);
        o[d + (2 * i + 1) * d + c] = sin_q(f * v, 1);
        f *= 2.0f;
    }
}

}  // namespace ntx
