// Go/no-go microbenchmark for a 3-term bf16 split of the MLP: how fast can ONE wave per SIMD
// run  acc[t] += Ahi*Bhi + Ahi*Blo + Alo*Bhi  (v_mfma_f32_32x32x16_bf16) when every wave streams its own copy of the
// packed weights (hi and lo records, 1 KiB each) from L2 -- the structure of the f32 kernel -- with RING records in flight.
// Reports the equivalent "f32 layers per second" against the f32-MFMA kernel's 65.5k cycles per 256x256 layer.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int RING>
__global__ __launch_bounds__(256) void k(const int *w, unsigned bytes, float *out, int layers) {
    const int lane = threadIdx.x & 63;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<int *>(w), 0, bytes, 0x00020000);
    const unsigned voff = lane * 16u;
    f32x16 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
    i32x4 ring[RING];
    // one layer = 16 k16-steps x 8 tiles x (hi, lo) = 256 records = 256 KiB
#pragma unroll
    for (int i = 0; i < RING; ++i) ring[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, i * 1024u, 0);
    bf16x8 bhi, blo;
    for (int j = 0; j < 8; ++j) { bhi[j] = (__bf16)(0.001f * lane + j); blo[j] = (__bf16)(1e-5f * j); }
    for (int L = 0; L < layers; ++L) {
        const unsigned base = (unsigned)(L & 7) * 262144u;   // 8 layers of weights = 2 MiB, L2-resident
#pragma unroll
        for (int rec = 0; rec < 256; rec += 2) {
            const int t = (rec >> 1) & 7;
            const bf16x8 ahi = __builtin_bit_cast(bf16x8, ring[rec % RING]);
            const bf16x8 alo = __builtin_bit_cast(bf16x8, ring[(rec + 1) % RING]);
            const unsigned nxt = base + (unsigned)(rec + RING) * 1024u;
            ring[rec % RING] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, nxt, 0);
            ring[(rec + 1) % RING] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, nxt + 1024u, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, bhi, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, blo, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, bhi, acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int t = 0; t < 8; ++t) s += acc[t][0];
    if (s == 123.456f) out[lane] = s;
}

// Same arithmetic, but the 4 waves of a workgroup share ONE copy of the stream through an LDS ring (NSTAGE stages of 16
// records = one k16-step of an 8-tile layer): every wave fetches a quarter of each stage with LDS-DMA loads
// (buffer_load_dwordx4 ... lds, no VGPRs), one s_barrier per stage, A operands read back with ds_read_b128.  Cuts the
// L2->L1 traffic (the per-wave structure's limit, ~61 of 64 B/clk/CU) by 4 and moves it to the LDS (128 B/clk/CU).
constexpr int vmcnt_imm(int n) { return (n & 0xF) | (7 << 4) | (0xF << 8) | ((n >> 4) << 14); }

// UL = layers per loop body: 1 -> ~7 KB of code (I-cache resident); 12 / 24 -> ~80 / ~160 KB of straight-line code, to see
// what instruction fetch costs once the body no longer fits the 64 KB instruction cache
template <int NSTAGE, int UL>
__global__ __launch_bounds__(256) void k_lds(const int *w, unsigned bytes, float *out, int layers) {
    __shared__ __attribute__((aligned(1024))) char ring[NSTAGE * 16384];
    const int lane = threadIdx.x & 63;
    const unsigned wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<int *>(w), 0, bytes, 0x00020000);
    const unsigned voff = lane * 16u;
    f32x16 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
    bf16x8 bhi, blo;
    for (int j = 0; j < 8; ++j) { bhi[j] = (__bf16)(0.001f * lane + j); blo[j] = (__bf16)(1e-5f * j); }
    // this wave's quarter (records 4 wv .. 4 wv + 3) of the stage at byte offset `goff`, into ring slot `slot`
    auto issue = [&](unsigned goff, int slot) {
        auto *dst = (__attribute__((address_space(3))) void *)(ring + slot * 16384 + wv * 4096);   // lane 0's slot; + lane*16 by hardware
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, voff, goff + wv * 4096u, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, voff, goff + wv * 4096u, 1024, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, voff, goff + wv * 4096u, 2048, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, voff, goff + wv * 4096u, 3072, 0);
    };
    auto rd = [&](int slot, int rec) {
        return *reinterpret_cast<const bf16x8 *>(ring + slot * 16384 + rec * 1024 + lane * 16);
    };
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s) issue(s * 16384u, s);
    __builtin_amdgcn_s_waitcnt(vmcnt_imm(4 * (NSTAGE - 3)));   // stages 0 and 1 have landed
    __builtin_amdgcn_s_barrier();
    bf16x8 a[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = rd(0, k);
    for (int L0 = 0; L0 < layers; L0 += UL) {
#pragma unroll
      for (int LL = 0; LL < UL; ++LL) {
        const int L = L0 + LL;
        const unsigned base = (unsigned)(L & 7) * 262144u;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            issue(base + (unsigned)(s + NSTAGE - 1) * 16384u, (s + NSTAGE - 1) % NSTAGE);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x8 n[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) n[k] = g < 3 ? rd(s % NSTAGE, 4 * (g + 1) + k) : rd((s + 1) % NSTAGE, k);
                __builtin_amdgcn_sched_barrier(0);   // the reads of the NEXT pair go first
                acc[2 * g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bhi, acc[2 * g], 0, 0, 0);
                acc[2 * g + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], bhi, acc[2 * g + 1], 0, 0, 0);
                acc[2 * g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], blo, acc[2 * g], 0, 0, 0);
                acc[2 * g + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], blo, acc[2 * g + 1], 0, 0, 0);
                acc[2 * g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], bhi, acc[2 * g], 0, 0, 0);
                acc[2 * g + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[3], bhi, acc[2 * g + 1], 0, 0, 0);
#pragma unroll
                for (int k = 0; k < 4; ++k) a[k] = n[k];
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_waitcnt(vmcnt_imm(4 * (NSTAGE - 3)));   // this wave's quarter of stage s+2 has landed
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    float sum = 0;
    for (int t = 0; t < 8; ++t) sum += acc[t][0];
    if (sum == 123.456f) out[lane] = sum;
}

template <int NSTAGE, int UL>
void run_lds() {
    const size_t bytes = 9 * 262144 + NSTAGE * 16384 + 4096;
    int *w; float *out;
    hipMalloc(&w, bytes); hipMemset(w, 0x11, bytes); hipMalloc(&out, 1024);
    const int layers = 1920;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_lds<NSTAGE, UL><<<256, 256>>>(w, (unsigned)bytes, out, 48);
    hipEventRecord(e0);
    k_lds<NSTAGE, UL><<<256, 256>>>(w, (unsigned)bytes, out, layers);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("LDS ring, %d stages, %2d layers per loop body: %.3f ms = %.3f us per 256x256 layer per wave (%s)\n", NSTAGE, UL, ms, ms * 1e3 / layers,
           hipGetErrorString(hipGetLastError()));
    hipFree(w); hipFree(out);
}

template <int RING>
void run() {
    const size_t bytes = 9 * 262144 + RING * 1024 + 4096;
    int *w; float *out;
    hipMalloc(&w, bytes); hipMemset(w, 0x11, bytes); hipMalloc(&out, 1024);
    const int layers = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<RING><<<256, 256>>>(w, (unsigned)bytes, out, 50);
    hipEventRecord(e0);
    k<RING><<<256, 256>>>(w, (unsigned)bytes, out, layers);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double cyc_per_layer = ms * 1e-3 * 2.4e9 / layers;
    printf("RING %2d: %.3f ms = %.3f us per layer per wave; at a nominal 2.4 GHz %.0f cycles per 256x256 layer per wave (ideal 3-term bf16 12288, f32 MFMA 65536) -> %.2fx the f32 MFMA rate, L2->wave %.1f B/clk/CU\n",
           RING, ms, ms * 1e3 / layers, cyc_per_layer, 65536.0 / cyc_per_layer, 4 * 262144.0 / cyc_per_layer);
    hipFree(w); hipFree(out);
}
int main() { run<16>(); run<24>(); run_lds<4, 1>(); run_lds<4, 4>(); run_lds<4, 8>(); run_lds<4, 12>(); run_lds<4, 24>(); return 0; }
