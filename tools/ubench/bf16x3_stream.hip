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
    printf("RING %2d: %.3f ms, %.0f cycles per 256x256 layer per wave (ideal 3-term bf16 12288, f32 MFMA 65536) -> %.2fx the f32 MFMA rate, L2->wave %.1f B/clk/CU\n",
           RING, ms, cyc_per_layer, 65536.0 / cyc_per_layer, 4 * 262144.0 / cyc_per_layer);
    hipFree(w); hipFree(out);
}
int main() { run<8>(); run<16>(); run<24>(); run<32>(); return 0; }
