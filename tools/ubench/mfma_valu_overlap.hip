// Microbenchmark: does VALU work hide under v_mfma_f32_16x16x4_f32 / 32x32x2_f32 when it comes from
// (a) the same wave, in blocks; (b) a second wave resident on the same SIMD?
// Each wave repeats: NM MFMAs (independent accumulators) then NV dependent v_fma.  waves/SIMD = 1 or 2.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE 0: identical waves; 1: waves 4-7 of the block (the second wave of each SIMD) run at s_setprio 1;
//      2: waves 4-7 start with a VALU block of half an iteration's MFMA time (phase offset)
template <int SHAPE, int NV, int MODE>
__global__ __launch_bounds__(512) void k(float *out, int iters, float a, float b) {
    float x = threadIdx.x * 1e-3f;
    const bool second = __builtin_amdgcn_readfirstlane(threadIdx.x) >= 256;
    if (MODE == 1 && second) __builtin_amdgcn_s_setprio(1);
    if (MODE == 2 && second) {
        for (int v = 0; v < 200; ++v) x = __builtin_fmaf(x, a, b);   // ~1000 cycles
    }
    f32x4 c4[16];
    f32x16 c16[4];
#pragma unroll
    for (int i = 0; i < 16; ++i) c4[i] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) c16[i][j] = 0;
    for (int it = 0; it < iters; ++it) {
        if constexpr (SHAPE == 16) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) c4[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c4[i], 0, 0, 0);   // 64 x 32 cyc
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) c16[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c16[i], 0, 0, 0);   // 32 x 64 cyc
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int v = 0; v < NV; ++v) x = __builtin_fmaf(x, a, b);
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = x;
    for (int i = 0; i < 16; ++i) s += c4[i][0];
    for (int i = 0; i < 4; ++i) s += c16[i][0];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int SHAPE, int NV, int MODE = 0>
void run(int waves_per_simd) {
    float *out; hipMalloc(&out, 4096);
    const int iters = 20000;
    dim3 grid(256), block(256 * waves_per_simd);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<SHAPE, NV, MODE><<<grid, block>>>(out, 100, 1.0f, 0.5f);
    hipEventRecord(e0);
    k<SHAPE, NV, MODE><<<grid, block>>>(out, iters, 1.0f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_cycles = 2048.0 * iters * waves_per_simd;   // per SIMD: 64x32 or 32x64 cycles per iteration per wave
    const double cycles = ms * 1e-3 * 2.4e9;
    printf("mode %d shape %2d  NV %4d  waves/SIMD %d : %8.3f ms   MFMA-only bound %8.3f ms   pipe util %.3f\n", MODE, SHAPE, NV, waves_per_simd, ms,
           mfma_cycles / 2.4e9 * 1e3, mfma_cycles / cycles);
    hipFree(out);
}

int main() {
    run<16, 64>(1); run<16, 256>(1);
    run<16, 64, 0>(2); run<16, 256, 0>(2);
    run<16, 64, 1>(2); run<16, 256, 1>(2);
    run<16, 64, 2>(2); run<16, 256, 2>(2);
    return 0;
}
