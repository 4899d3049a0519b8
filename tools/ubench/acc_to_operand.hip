// Microbenchmark: a layer's results go from the accumulators (AGPRs) to the next layer's B operands (VGPRs) -- through the vector ALU
// (v_accvgpr_read_b32, one VALU instruction a value: what the render and training chains do today) or through LDS (ds_write_b128 reads
// AGPRs directly, ds_read_b128 fills VGPRs: no VALU issue slot).  One wave per SIMD (4 waves a workgroup, 256 workgroups), 128 values a
// lane, 1024 x v_mfma_f32_32x32x2_f32 a "layer" (8 tiles x 128 k-steps, the B operand of k-step k is value k), then the conversion
// (scale + ReLU: 2 VALU a value in every mode).
//   MODE 0: the conversion on 2 values only (the floor)      MODE 1: v_accvgpr_read_b32 a value      MODE 2: ds_write_b128 / ds_read_b128 a quad
// hipcc --offload-arch=gfx950 -O3 -o acc_to_operand tools/ubench/acc_to_operand.hip && ./acc_to_operand
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <utility>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// (plain templates instead of lambdas: inline asm operands inside a lambda do not capture)
template <int V> __device__ __forceinline__ void read1(float (&hin)[128], const f32x16 (&acc)[8]) {
    float x;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(acc[V >> 4][V & 15]));
    hin[V] = __builtin_fmaxf(x * 1e-3f, 0.0f);
}
template <int... V> __device__ __forceinline__ void read_all(float (&hin)[128], const f32x16 (&acc)[8], std::integer_sequence<int, V...>) { (read1<V>(hin, acc), ...); }
template <int Q> __device__ __forceinline__ void write1(uint32_t base, const f32x16 (&acc)[8]) {
    const f32x4 quad = {acc[Q >> 2][4 * (Q & 3) + 0], acc[Q >> 2][4 * (Q & 3) + 1], acc[Q >> 2][4 * (Q & 3) + 2], acc[Q >> 2][4 * (Q & 3) + 3]};
    asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(base), "a"(quad), "n"(Q * 1024) : "memory");
}
template <int... Q> __device__ __forceinline__ void write_all(uint32_t base, const f32x16 (&acc)[8], std::integer_sequence<int, Q...>) { (write1<Q>(base, acc), ...); }
template <int Q> __device__ __forceinline__ void back1(uint32_t base, f32x4 (&back)[32]) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(back[Q]) : "v"(base), "n"(Q * 1024) : "memory");
}
template <int... Q> __device__ __forceinline__ void back_all(uint32_t base, f32x4 (&back)[32], std::integer_sequence<int, Q...>) { (back1<Q>(base, back), ...); }

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a0) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float hin[128];
#pragma unroll
    for (int v = 0; v < 128; ++v) hin[v] = (float)(v + lane) * 1e-6f;
    f32x16 acc[8];
    float a[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) a[t] = a0 * (float)(t + 1 + lane);          // eight different tiles (identical ones would be merged)
    uint32_t base = (uint32_t)(wave * 32768 + lane * 16);       // a wave's 32 KB: quad q of lane L at q * 1024 + L * 16
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[t][j] = 0.0f;
#pragma unroll
        for (int kk = 0; kk < 128; ++kk)
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], hin[kk], acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MODE == 0) {
            float x0, x1;
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x0) : "a"(acc[0][0]));
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x1) : "a"(acc[7][15]));
            hin[0] = __builtin_fmaxf(x0 * 1e-3f, 0.0f); hin[127] = __builtin_fmaxf(x1 * 1e-3f, 0.0f);
            asm volatile("" :: "a"(acc[1]), "a"(acc[2]), "a"(acc[3]), "a"(acc[4]), "a"(acc[5]), "a"(acc[6]));   // (the other tiles are computed all the same)
        } else if constexpr (MODE == 1) {
            read_all(hin, acc, std::make_integer_sequence<int, 128>{});
        } else {
            write_all(base, acc, std::make_integer_sequence<int, 32>{});
            f32x4 back[32];
            back_all(base, back, std::make_integer_sequence<int, 32>{});
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int v = 0; v < 128; ++v) hin[v] = __builtin_fmaxf(back[v >> 2][v & 3] * 1e-3f, 0.0f);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0;
#pragma unroll
    for (int v = 0; v < 128; ++v) s += hin[v];
    if (s == 12345.678f) out[threadIdx.x] = s + lds[threadIdx.x];
}

template <int MODE>
double run(const char *name) {
    float *out; hipMalloc(&out, 4096);
    const int iters = 4000;
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256, 256, 131072>>>(out, iters / 4, 1.0f);            // warm: the clock settles under the MFMA load
    hipEventRecord(e0);
    k<MODE><<<256, 256, 131072>>>(out, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us_per_layer = ms * 1e3 / iters;
    printf("%-44s %8.3f us a layer (1024 MFMAs = %.3f us at 2.4 GHz)\n", name, us_per_layer, 1024 * 64 / 2.4e3);
    hipFree(out);
    return us_per_layer;
}

int main() {
    double f = 1e9, v = 1e9, l = 1e9;
    for (int rep = 0; rep < 3; ++rep) {                               // interleaved, the best of three: the boxes' clocks wander
        f = fmin(f, run<0>("floor (2 values converted)"));
        v = fmin(v, run<1>("v_accvgpr_read_b32 + scale + ReLU, 128 values"));
        l = fmin(l, run<2>("ds_write_b128 / ds_read_b128 + scale + ReLU"));
    }
    printf("conversion through the vector ALU: +%.3f us a layer (%.2f %%); through LDS: +%.3f us (%.2f %%)\n", v - f, 100 * (v - f) / f, l - f, 100 * (l - f) / f);
    return 0;
}
