// ntx_train.hip -- one training step of the reference on the GPU: network/train.py:49-70 (GradientTape over Renderer.__call__, a loss of
// network/loss.py:6-59, Adam under ExponentialDecay) for the ParamNerf architecture of the shipped training configs (8 x 256, skip 4,
// color_depth 1; configs/config_carpet_train.py: 4 images x 256 rays x 256 samples = 262 144 samples a step).  gfx950 only.
//
// Inference fuses the whole network into one kernel because nothing of it has to survive (ntx_device.h).  A training step has to keep
// every layer's activations for the backward pass; with 288 GB of HBM they are simply stored -- 13 layers x 262 144 x 256 floats = 3.5 GB
// -- and the step is a sequence of dense contractions on the f32 matrix cores:
//
//   encode_kernel          sample points, positional encodings of position / direction / parameters (layer.py:8-23), written into the two
//                          concat buffers the network reads them from ([pos_map | pad | h4] for the skip, [dir_map | pad | feature])
//   rows_kernel<NT, MODE>  forward Y = act(X . W + b) and dX = (dY . W^T) masked: a wave owns 32 samples and all of a layer's outputs on
//                          v_mfma_f32_32x32x2_f32, X straight from HBM, the weights from a packed image through an LDS ring
//                          (pack_records_kernel makes the images once a step)
//   gemm_kernel<A, TN, TK> dW = X^T . dY: 128 x 128 x 16 tiles through LDS (double buffered), split along the 262 144 samples into
//                          partial sums that one pass adds up in a fixed order (a step is bit-reproducible), the bias gradient riding along;
//                          also C = A . B on caller buffers (ntx_gemm_f32)
//   head kernels           the 1-wide density head and the 3-wide colour head, forward and backward, on the vector ALUs
//   composite_forward / _backward   renderer.py:170-213 per ray (wave per ray) and its adjoint: suffix sums of the weights' gradients
//   loss_kernel            NerfLoss / AlphaLoss with mse / smape (loss.py), value and gradient
//   adam_kernel            tf.keras.optimizers.Adam under ExponentialDecay (train.py:49-52), one fused pass over the 2.7 MB of weights
//
// Everything is float32 with float32 accumulation, like the reference's TensorFlow graph.
#include "nerftex.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

extern "C" int ntx_set_error(int code, const char *fmt, ...);   // nerftex.hip

#define TRAIN_TRY(expr)                                                                                  \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) return ntx_set_error(NTX_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

namespace ntx_train {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------------------------------------------
// the general contraction.  C[i][j] = sum_p A'(i, p) B(p, j) for i < M, j < N, p < K, with B[p * ldb + j] and
//   A'(i, p) = A_KCONTIG ? A[i * lda + p] : A[p * lda + i]
// In a training step it takes the weight gradients, dW = X^T . dY (A' = X^T: the reduction runs over the samples, split into ranges whose
// partial sums are added in a fixed order); the row-major form (A_KCONTIG) is what ntx_gemm_f32 offers on caller buffers -- the step's
// forward and dX contractions were here until rows_kernel (below) took them.  A workgroup (4 waves) owns a 128 x 128 tile of C, a wave a
// 64 x 64 quarter of it = 2 x 2 MFMA tiles of 32 x 32 (64 accumulator registers).  K advances a panel (16) at a time: the next
// 128 x 16 / 16 x 128 panels are fetched into registers (16-byte loads when the panel lies inside the matrices and the rows are 16-byte
// aligned, element by element with bounds otherwise) while the current ones, already in LDS as As[p][i] / Bs[p][j], feed the MFMAs; one
// barrier per panel (double buffered).  The f32 MFMA shares the vector ALUs' lanes, so every VALU instruction of the loop
// costs MFMA time: hence the vector loads and the branch-free interior path.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int TM = 128;
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct GemmArgs {
    const float *A; int lda; const float *B; int ldb; float *C; int ldc;
    int M, N, K;
    const float *bias;               // NULL or [N]: added to every row
    const float *mask; int ldmask;   // NULL, or C[i][j] is kept only where mask[i * ldmask + j] > 0 (the ReLU of a stored activation)
    int relu, accumulate;            // C = max(C, 0);  C += what was there
    int k_chunk; long long split_stride;   // blockIdx.z = z covers p in [z * k_chunk, (z + 1) * k_chunk) and writes to C + z * split_stride
    float *colsum;                   // NULL, or [n_split][N]: the column sums of B over each split's rows (the bias gradient rides along with dW)
    int aligned;                     // every row of A and B starts on a 16-byte boundary
};

// n consecutive floats of a row into registers: 16-byte loads, or one by one under a bound
template <int n, bool FAST>
__device__ __forceinline__ void fetch_run(const float *g, bool row_ok, int first, int bound, float *r, bool aligned_ok = false) {
    if (FAST) {
        const f32x4 *v = reinterpret_cast<const f32x4 *>(g);
#pragma unroll
        for (int q = 0; q < n / 4; ++q) { const f32x4 x = v[q]; r[4 * q] = x.x; r[4 * q + 1] = x.y; r[4 * q + 2] = x.z; r[4 * q + 3] = x.w; }
    } else if (row_ok && first + n <= bound && aligned_ok) {         // the run lies inside: vector loads here too
        const f32x4 *v = reinterpret_cast<const f32x4 *>(g);
#pragma unroll
        for (int q = 0; q < n / 4; ++q) { const f32x4 x = v[q]; r[4 * q] = x.x; r[4 * q + 1] = x.y; r[4 * q + 2] = x.z; r[4 * q + 3] = x.w; }
    } else {
#pragma unroll
        for (int q = 0; q < n; ++q) r[q] = (row_ok && first + q < bound) ? g[q] : 0.0f;
    }
}

// TN_: columns of the workgroup's tile (128 or 256: two or four 64-wide waves across), TK_: depth of a panel; a wave always owns 64 x 64
template <bool A_KCONTIG, int TN_, int TK_>
__device__ __forceinline__ void gemm_body(const GemmArgs &g, int bx, int by, int bz) {
    constexpr int THREADS = TN_ * 2, WCOLS = TN_ / 64, LROWA = TM + 4, LROWB_ = TN_ + 4;
    constexpr int FA = TM * TK_ / THREADS;          // floats of the A panel a thread carries
    constexpr int FB = TN_ * TK_ / THREADS;         // ... of the B panel
    constexpr int TPR = THREADS / TK_;              // threads along a panel row (B, and A when its rows run along i)
    static_assert(FA % 4 == 0 && FB % 4 == 0 && FA * (THREADS / TM) == TK_ && TPR * FB == TN_ && TPR * FA == TM, "panel split");
    __shared__ __attribute__((aligned(16))) float As[2][TK_][LROWA], Bs[2][TK_][LROWB_];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j0 = bx * TN_, i0 = by * TM;
    const int k_begin = bz * g.k_chunk;
    const int k_end = k_begin + g.k_chunk < g.K ? k_begin + g.k_chunk : g.K;
    float *C = g.C + (size_t)bz * (size_t)g.split_stride;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    // two panels are in flight from memory at any time: the registers of panel kt + 2 are being filled while panel kt + 1 goes from its
    // registers into LDS and panel kt feeds the MFMAs
    float ra[2][FA], rb[2][FB], cs[FB];
#pragma unroll
    for (int q = 0; q < FB; ++q) cs[q] = 0.0f;
    const bool want_colsum = !A_KCONTIG && g.colsum != nullptr && by == 0;
    const bool inner = g.aligned && i0 + TM <= g.M && j0 + TN_ <= g.N;          // the tile lies inside A' and B: only the K end of a panel can stick out
    // thread -> its run of the A panel (k-contiguous rows A[i][p]: row t % 128, FA elements from (t / 128) * FA; rows along i, A[p][i]: panel row
    // t / TPR, FA elements from (t % TPR) * FA) and of the B panel (B[p][j]: panel row t / TPR, FB elements from (t % TPR) * FB)
    const int a_row = A_KCONTIG ? (int)(threadIdx.x % TM) : (int)(threadIdx.x / TPR), a_off = A_KCONTIG ? (int)(threadIdx.x / TM) * FA : (int)(threadIdx.x % TPR) * FA;
    const int b_row = (int)(threadIdx.x / TPR), b_off = (int)(threadIdx.x % TPR) * FB;
    auto fetch = [&](int k0, float *fa, float *fb) {                  // with bounds: a run that lies inside still comes by 16-byte loads
        const bool al = g.aligned != 0;
        if (A_KCONTIG) fetch_run<FA, false>(g.A + (size_t)(i0 + a_row) * g.lda + k0 + a_off, i0 + a_row < g.M, k0 + a_off, k_end, fa, al);
        else fetch_run<FA, false>(g.A + (size_t)(k0 + a_row) * g.lda + i0 + a_off, k0 + a_row < k_end, i0 + a_off, g.M, fa, al);
        fetch_run<FB, false>(g.B + (size_t)(k0 + b_row) * g.ldb + j0 + b_off, k0 + b_row < k_end, j0 + b_off, g.N, fb, al);
    };
    auto stash = [&](int buf, const float *fa, const float *fb) {
        if (want_colsum) {
#pragma unroll
            for (int q = 0; q < FB; ++q) cs[q] += fb[q];
        }
        if (A_KCONTIG) {
#pragma unroll
            for (int q = 0; q < FA; ++q) As[buf][a_off + q][a_row] = fa[q];
        } else {
            f32x4 *d = reinterpret_cast<f32x4 *>(&As[buf][a_row][a_off]);
#pragma unroll
            for (int q = 0; q < FA / 4; ++q) d[q] = f32x4{fa[4 * q], fa[4 * q + 1], fa[4 * q + 2], fa[4 * q + 3]};
        }
        f32x4 *d = reinterpret_cast<f32x4 *>(&Bs[buf][b_row][b_off]);
#pragma unroll
        for (int q = 0; q < FB / 4; ++q) d[q] = f32x4{fb[4 * q], fb[4 * q + 1], fb[4 * q + 2], fb[4 * q + 3]};
    };
    const int n_panels = (k_end - k_begin + TK_ - 1) / TK_;
    const int n_full = inner ? (k_end - k_begin) / TK_ : 0;            // panels the bounds-free pipeline takes; the rest (a K tail, edge tiles) go one by one
    // a wave's 64 x 64 quarter as 2 x 2 MFMA tiles that INTERLEAVE: tile (a, b) = its rows 2 m + a, its columns 2 n + b -- a lane's two A
    // (two B) operands of a k-step then sit side by side in LDS (one 8-byte read each) and its results pair up into 8-byte stores
    const int wi = (wave / WCOLS) * 64 + 2 * (lane & 31), wj = (wave % WCOLS) * 64 + 2 * (lane & 31), kh = lane >> 5;
    const bool wave_live = j0 + (wave % WCOLS) * 64 < g.N;       // a narrow matrix leaves some of the tile's waves without columns
    // the MFMAs of one panel in LDS[buf].  The operands of k-step s + 1 are asked for BEFORE the four MFMAs of step s are issued (the
    // scheduling barriers keep the compiler from sinking the reads back down to their use, which leaves the matrix pipe idle for an LDS
    // round trip every step)
    auto compute = [&](int buf) {
        if (!wave_live) return;
        f32x2 av[2], bv[2];
        av[0] = *reinterpret_cast<const f32x2 *>(&As[buf][kh][wi]); bv[0] = *reinterpret_cast<const f32x2 *>(&Bs[buf][kh][wj]);
#pragma unroll
        for (int st = 0; st < TK_ / 2; ++st) {
            const int c = st & 1, n = c ^ 1;
            if (st + 1 < TK_ / 2) {
                const int kk = 2 * (st + 1) + kh;
                av[n] = *reinterpret_cast<const f32x2 *>(&As[buf][kk][wi]); bv[n] = *reinterpret_cast<const f32x2 *>(&Bs[buf][kk][wj]);
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c].x, bv[c].x, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c].x, bv[c].y, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c].y, bv[c].x, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c].y, bv[c].y, acc[1][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto fetch_fast = [&](int k0, float *fa, float *fb) {
        const float *ga = A_KCONTIG ? g.A + (size_t)(i0 + a_row) * g.lda + k0 + a_off : g.A + (size_t)(k0 + a_row) * g.lda + i0 + a_off;
        fetch_run<FA, true>(ga, true, 0, 0, fa);
        fetch_run<FB, true>(g.B + (size_t)(k0 + b_row) * g.ldb + j0 + b_off, true, 0, 0, fb);
    };
    // Two panels are in flight from memory at any time: panel kt feeds the MFMAs from LDS, panel kt + 1 waits in one register set for its
    // turn to go into LDS, panel kt + 2 is on its way into the other.  With the bounds-free loads the steady-state loop has no branch around
    // a load, so the wait in front of the LDS stores covers panel kt + 1 only (s_waitcnt vmcnt(loads of one panel)), not the panel just
    // asked for.  Panels [first, last) of this workgroup's K range.
    auto pipeline = [&](auto fast_tag, int first, int last) {
        constexpr bool FAST = decltype(fast_tag)::value;
        auto get = [&](int kt, float *fa, float *fb) { if (FAST) fetch_fast(k_begin + kt * TK_, fa, fb); else fetch(k_begin + kt * TK_, fa, fb); };
        if (first >= last) return;
        get(first, ra[0], rb[0]); stash(0, ra[0], rb[0]);
        if (first + 1 < last) get(first + 1, ra[1], rb[1]);
        __syncthreads();
        int kt = first;                                               // LDS buffer of panel kt = (kt - first) & 1
        for (; kt + 3 < last; kt += 2) {
            get(kt + 2, ra[0], rb[0]); compute(0); stash(1, ra[1], rb[1]); __syncthreads();
            get(kt + 3, ra[1], rb[1]); compute(1); stash(0, ra[0], rb[0]); __syncthreads();
        }
        for (; kt < last; ++kt) {                                     // the last two or three panels: nothing left to ask for behind them
            const int buf = (kt - first) & 1;
            if (kt + 2 < last) { if (buf) get(kt + 2, ra[1], rb[1]); else get(kt + 2, ra[0], rb[0]); }
            compute(buf);
            if (kt + 1 < last) { if (buf) stash(0, ra[0], rb[0]); else stash(1, ra[1], rb[1]); }
            __syncthreads();
        }
    };
    pipeline(std::true_type{}, 0, n_full);
    pipeline(std::false_type{}, n_full, n_panels);                    // a K tail; every panel of a tile on the matrix's edge
    // D of a 32 x 32 tile: lane l, register r  <->  tile row m = 8 (r >> 2) + (r & 3) + 4 (l >> 5), tile column n = l & 31; with the
    // interleaved tiles that is row 2 m + a, columns 2 n and 2 n + 1 (b = 0, 1): one 8-byte access per (a, r)
    const bool whole = i0 + TM <= g.M && j0 + TN_ <= g.N && (g.ldc % 2 == 0) && (!g.mask || g.ldmask % 2 == 0) && ((uintptr_t)C % 8 == 0) && ((uintptr_t)g.mask % 8 == 0);
    const int j = j0 + (wave % WCOLS) * 64 + 2 * (lane & 31);
    const float bj0 = (g.bias && j < g.N) ? g.bias[j] : 0.0f, bj1 = (g.bias && j + 1 < g.N) ? g.bias[j + 1] : 0.0f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = i0 + (wave / WCOLS) * 64 + 2 * (8 * (r >> 2) + (r & 3) + 4 * kh) + a;
            float v0 = acc[a][0][r], v1 = acc[a][1][r];
            float *c = C + (size_t)i * g.ldc + j;
            if (whole) {
                f32x2 *c2 = reinterpret_cast<f32x2 *>(c);
                if (g.accumulate) { const f32x2 o = *c2; v0 += o.x; v1 += o.y; }
                v0 += bj0; v1 += bj1;
                if (g.relu) { v0 = v0 > 0.0f ? v0 : 0.0f; v1 = v1 > 0.0f ? v1 : 0.0f; }
                if (g.mask) { const f32x2 mk = *reinterpret_cast<const f32x2 *>(g.mask + (size_t)i * g.ldmask + j); if (!(mk.x > 0.0f)) v0 = 0.0f; if (!(mk.y > 0.0f)) v1 = 0.0f; }
                *c2 = f32x2{v0, v1};
            } else if (i < g.M) {
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    if (j + b >= g.N) continue;
                    float v = b ? v1 : v0;
                    if (g.accumulate) v = v + c[b];
                    v = v + (b ? bj1 : bj0);
                    if (g.relu) v = v > 0.0f ? v : 0.0f;
                    if (g.mask && !(g.mask[(size_t)i * g.ldmask + j + b] > 0.0f)) v = 0.0f;
                    c[b] = v;
                }
            }
        }
    if (want_colsum) {                                        // the panel rows a column was spread over, added up in a fixed order
        float (*red)[LROWB_] = Bs[0];
#pragma unroll
        for (int q = 0; q < FB; ++q) red[b_row][b_off + q] = cs[q];
        __syncthreads();
        if ((int)threadIdx.x < TN_ && j0 + (int)threadIdx.x < g.N) {
            float sum = 0.0f;
            for (int q = 0; q < TK_; ++q) sum += red[q][threadIdx.x];
            g.colsum[(size_t)bz * g.N + j0 + threadIdx.x] = sum;
        }
    }
}


// Which tile of which K range.  The tiles of ONE range read the same panels of A and B: workgroups are dealt to the 8 XCDs round-robin by
// their linear number, each XCD has its own L2 -- so a range's tiles are given numbers that land on one XCD, next to each other in time,
// and the panels come from HBM once instead of once per tile (dW of a 256 x 256 layer: 999 MB a launch at 3.8 TB/s before, for 537 MB of
// operands).  lin: the workgroup's number within its problem; nx x ny tiles, nz ranges (a multiple of 8, or the plain order is kept).
__device__ __forceinline__ void gemm_place(int lin, int nx, int ny, int nz, int &bx, int &by, int &bz) {
    const int tiles = nx * ny;
    if (nz % 8 == 0) {
        const int xcd = lin & 7, slot = lin >> 3, tile = slot % tiles;
        bz = (slot / tiles) * 8 + xcd; bx = tile % nx; by = tile / nx;
    } else { bx = lin % nx; by = (lin / nx) % ny; bz = lin / tiles; }
}
template <bool A_KCONTIG, int TN_, int TK_, int WAVES_PER_EU = 2>
__global__ __launch_bounds__(TN_ * 2) __attribute__((amdgpu_waves_per_eu(WAVES_PER_EU, 8))) void gemm_kernel(GemmArgs g) {
    int bx, by, bz;
    gemm_place(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gridDim.x, gridDim.y, gridDim.z, bx, by, bz);
    gemm_body<A_KCONTIG, TN_, TK_>(g, bx, by, bz);
}
// several contractions in one launch (the weight gradients of every layer of a step: 12 launches' ramps, tails and write-backs become one);
// a problem's workgroups are numbered consecutively from first[p] (multiples of 8, so that a workgroup's XCD is its local number's too)
constexpr int MAX_GEMM_BATCH = 14;
struct GemmBatch { GemmArgs g[MAX_GEMM_BATCH]; int first[MAX_GEMM_BATCH + 1]; int nx[MAX_GEMM_BATCH], ny[MAX_GEMM_BATCH], nz[MAX_GEMM_BATCH]; int n; };
template <bool A_KCONTIG, int TN_, int TK_, int WAVES_PER_EU = 2>
__global__ __launch_bounds__(TN_ * 2) __attribute__((amdgpu_waves_per_eu(WAVES_PER_EU, 8))) void gemm_batch_kernel(GemmBatch b) {
    int p = 0;
    while (p + 1 < b.n && (int)blockIdx.x >= b.first[p + 1]) ++p;
    const int lin = (int)blockIdx.x - b.first[p];
    if (lin >= b.nx[p] * b.ny[p] * b.nz[p]) return;           // (the numbers between two problems, rounded up to 8)
    int bx, by, bz;
    gemm_place(lin, b.nx[p], b.ny[p], b.nz[p], bx, by, bz);
    gemm_body<A_KCONTIG, TN_, TK_>(b.g[p], bx, by, bz);
}

// out[e] = sum_z partial[z][e], z ascending: the fixed order that makes a step reproducible
__global__ void reduce_partials_kernel(const float *__restrict__ partial, int n_split, long long stride, long long count, float *__restrict__ out) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count) return;
    // four running sums over z = 0, 4, 8 ... / 1, 5, ... / ..., combined at the end: a fixed order, four loads in flight
    float s4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    int z = 0;
    for (; z + 4 <= n_split; z += 4) {
#pragma unroll
        for (int q = 0; q < 4; ++q) s4[q] += partial[(size_t)(z + q) * stride + e];
    }
    for (int q = 0; z < n_split; ++z, ++q) s4[q] += partial[(size_t)z * stride + e];
    out[e] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
}
// ... of several results in one launch; a job's elements are numbered from first (multiples of 256: a block belongs to one job)
constexpr int MAX_REDUCE_BATCH = 2 * MAX_GEMM_BATCH;
struct ReduceJob { const float *partial; int n_split; long long stride, count; float *out; long long first; };
struct ReduceBatch { ReduceJob job[MAX_REDUCE_BATCH]; int n; };
__global__ void reduce_batch_kernel(ReduceBatch b) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int j = 0;
    while (j + 1 < b.n && g >= b.job[j + 1].first) ++j;
    const ReduceJob &r = b.job[j];
    const long long e = g - r.first;
    if (e >= r.count) return;
    float s4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    int z = 0;
    for (; z + 4 <= r.n_split; z += 4) {
#pragma unroll
        for (int q = 0; q < 4; ++q) s4[q] += r.partial[(size_t)(z + q) * r.stride + e];
    }
    for (int q = 0; z < r.n_split; ++z, ++q) s4[q] += r.partial[(size_t)z * r.stride + e];
    r.out[e] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
}
// the same for FEW elements and MANY partial sums (the narrow heads: a thousand row blocks of 257 or 771 numbers): 32 elements a block,
// 8 threads each taking every 8th partial sum, their results added in order -- as fixed an order as the above, an eighth of the chain
__global__ __launch_bounds__(256) void reduce_partials_wide_kernel(const float *__restrict__ partial, int n_split, long long stride, long long count, float *__restrict__ out) {
    __shared__ float part[8][32];
    const int slice = threadIdx.x >> 5, lane = threadIdx.x & 31;
    const long long e = (long long)blockIdx.x * 32 + lane;
    float sum = 0.0f;
    if (e < count)
        for (int z = slice; z < n_split; z += 8) sum += partial[(size_t)z * stride + e];
    part[slice][lane] = sum;
    __syncthreads();
    if (slice == 0 && e < count) {
        float total = part[0][lane];
        for (int q = 1; q < 8; ++q) total += part[q][lane];
        out[e] = total;
    }
}
// ---------------------------------------------------------------------------------------------------------------------------
// The two contractions of a step whose reduction is SHORT (the layer's width) and whose other side is the samples -- forward Y = X . W and
// dX = dY . W^T -- the way the render kernel does a layer (ntx_device.h): one wave owns 32 samples and ALL of a layer's
// outputs (8 tiles of 32 x 32 = 128 accumulator registers) on v_mfma_f32_32x32x2_f32.  The A operand is the wave's own 32 rows of X,
// read straight from HBM 16 bytes a lane, a body (32 k) ahead.  The B operand is the weights, from a PACKED image that the workgroup's
// four waves pull through a triple-buffered LDS ring with direct global -> LDS loads (one 1 KiB record per instruction; a body's 32 records
// in four quarters) and read back with ds_read_b128 two records ahead: the weights cost no registers, wait on the LDS counter and not
// behind the HBM loads on the in-order vector-memory counter (a register ring did: every X load held up the ring eight records later), and
// reach a CU once per workgroup instead of once per wave.  No VALU work in the k loop (the f32 MFMA shares the vector ALUs' lanes: every
// VALU instruction costs MFMA time).  Where the 128 x 128 LDS tiles of gemm_kernel reach 0.58-0.64 of the MFMA peak, this is bound by
// the matrix pipe and the clock the power limit leaves it (2.25-2.38 GHz while it runs).  dW = X^T . dY reduces over the samples: it
// stays with gemm_kernel.
//
// One wave per SIMD, one stream of instructions with nothing to wait for:
//  * a body = 128 MFMAs on one 32 KiB chunk of the image.  The chunk after the next one is asked for and the workgroup's one barrier per
//    body falls 8 MFMAs BEFORE a body's end: behind it the next chunk is known to be complete (every wave waited for its own quarter) and
//    the buffer two chunks back to be free, so the ds_reads run on into the next chunk without a gap -- no wave ever stands at a barrier
//    with an empty pipe behind it;
//  * two accumulator sets: while block n + 1 accumulates into one, block n's results leave the other: made final in ONE dense block of VALU
//    work (bias is already in; ReLU and its bits), then stored PIECE BY PIECE in the shadow of the MFMAs of block n + 1's first two bodies
//    (one output register every second MFMA).  The stores never come in bursts (all waves of the chip storing 32 KB each at the same
//    moment cost 3.5 us a block with the matrix pipe idle), and the wait in front of the barrier counts them out (vmcnt(stores of this
//    body): only what is older has to be there).  VALU instructions scattered between f32 MFMAs cost 18 cycles each here, in a dense block
//    8: hence the split.  (The mask of dX alone is applied on the way out, four outputs at a time: all 128 at once need more registers
//    than a wave has.)
//
// k order.  A lane (sample m = l & 31, half h = l >> 5) loads X[m][8 q + 4 h .. + 3] in one piece; k-step s = 4 q + c then pairs
// k = 8 q + c (lower half-wave) with k = 8 q + 4 + c (upper) -- the order of the summation over k is free.  The packed image follows:
// record (s, g) = 64 lanes x 4 floats, component c' of lane (f, h) = Wsrc[k(s, h)][32 (4 g + c') + f]: one 16-byte read feeds 4 MFMAs.
// Rows of the image with no counterpart (K rounded up to 32; the pad columns between the two halves of a concat buffer) are zero, the X
// values they meet are finite (pad columns are zeroed once, what lies behind a row is the next row; beyond the matrix the buffer
// descriptor returns 0).
//
// D of a tile (the MFMA's A operand is X, its B operand the weights): lane (f = l & 31, h), register r <-> sample 8 (r >> 2) + 4 h + (r & 3),
// feature 32 t + f: a register is 32 consecutive features of one sample per half-wave, so a row-major Y[sample][feature] is written in
// full 128-byte lines.  The ReLU mask dX needs is not read back as 268 MB of activations: the forward pass leaves ONE BIT per output,
// in the accumulators' own layout (lane, tile, register: 128 bits = 16 bytes a lane and block), and dX -- whose outputs lie in the same
// layout -- reads those 8 MB.
// ---------------------------------------------------------------------------------------------------------------------------
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// A launch walks every pair of row blocks through a CHAIN of layers (the whole trunk forward; the whole way back for dX): block A layer l,
// block B layer l, block A layer l + 1 ...  A block's layer-l outputs leave under the other block's MFMAs and are complete a body later,
// long before the same wave reads them back as layer l + 1's X (its own rows, through its own CU's L2: nothing is shared between waves
// but the weights) -- ten launches' ramps, first loads, last stores and L2 write-backs become one.
struct RowsLayer {
    const float *X; int ldx;
    const float *recs; int kblocks;                    // packed weights; blocks of 8 k (a multiple of 4, at least 12)
    float *Y; int ldy;
    const float *bias; int linear;                     // forward: bias [32 NT]; linear: no ReLU (and no bits)
    unsigned int *bits_out;                            // forward: one bit per output, set where it is > 0 (or NULL)
    const unsigned int *bits_in;                       // dX: keep where the bit is set (a layer without a ReLU behind it: a buffer of ones)
};
constexpr int MAX_ROWS_LAYERS = 10;
struct RowsArgs { RowsLayer layer[MAX_ROWS_LAYERS]; int n_layers; long long M; };
enum { ROWS_FORWARD = 0, ROWS_DX_MASK = 2 };           // what the epilogue does: a template parameter
template <int N, class F> __device__ __forceinline__ void static_for_(F &&f) {
    if constexpr (N > 0) { static_for_<N - 1>(f); f(std::integral_constant<int, N - 1>{}); }
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rows_rsrc(const void *base, long long bytes) {
    const long long b = bytes < 0 ? 0 : (bytes > 0x7ffffff0ll ? 0x7ffffff0ll : bytes);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)b, 0x00020000);
}
// s_waitcnt vmcnt(n) alone (gfx9 encoding: vmcnt in bits 3:0 and 15:14, expcnt 6:4 and lgkmcnt 11:8 left at their maxima) -- as an
// instruction the compiler's own counting sees, unlike inline assembly
template <int N> __device__ __forceinline__ void wait_vmcnt() { __builtin_amdgcn_s_waitcnt(((N >> 4) << 14) | 0x0F70 | (N & 15)); }

template <int NT, int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void rows_kernel(RowsArgs a) {
    constexpr int G = NT / 4;                          // records per k-step
    constexpr int CHUNK = 16 * G;                      // records per body (4 blocks of 8 k = 16 k-steps): 16 or 32 KiB
    constexpr int RING = 2;                            // records read ahead of the MFMAs (CHUNK is a multiple: the ring's phase is the same in every body)
    constexpr int TAIL = CHUNK - RING - 1;             // the barrier stands behind this record's MFMAs: the next one reads ahead into the next chunk
    __shared__ f32x4 lds[3 * CHUNK * 64 + MAX_ROWS_LAYERS * NT * 8];     // three chunks, then every layer's bias
    const int lane = threadIdx.x & 63, m = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long n_blocks = (a.M + 31) / 32, n_groups = (n_blocks + 3) / 4;
    if ((long long)blockIdx.x >= n_groups) return;
    const int L = a.n_layers;
    const int n_mine = (int)((n_groups - blockIdx.x + gridDim.x - 1) / gridDim.x), n_items = (n_mine + 1) / 2 * 2 * L;   // pairs of groups x layers x 2
    constexpr bool MASK = MODE == ROWS_DX_MASK, BIAS = MODE == ROWS_FORWARD;
    const uint32_t woff = (uint32_t)lane * 16u;
    if constexpr (BIAS) {
        float *bl = reinterpret_cast<float *>(&lds[3 * CHUNK * 64]);
        for (int e = threadIdx.x; e < L * NT * 32; e += 256) bl[e] = a.layer[e / (NT * 32)].bias[e % (NT * 32)];
    }
    // item w of this wave: which rows, which layer.  Behind the last one (and for the second half of an odd pair): no rows -- empty descriptors
    const long long no_rows = n_blocks * 32;           // "no block": behind every row AND on a block boundary (its bits, too, must fall outside: M / 32 is the last block when M is ragged)
    auto item_row0 = [&](int w) -> long long {
        if (w < 0 || w >= n_items) return no_rows;
        const int p = w / (2 * L), rem = w - p * 2 * L;
        const long long grp = (long long)blockIdx.x + (long long)(2 * p + (rem & 1)) * gridDim.x;
        return grp < n_groups ? (grp * 4 + wave) * 32 : no_rows;
    };
    auto item_layer = [&](int w) -> int { return (w < 0 || w >= n_items) ? 0 : (w % (2 * L)) >> 1; };
    // this wave's quarter of a chunk, straight into LDS: one record (64 lanes x 16 bytes, lane-linear) per instruction
    auto fill = [&](const __amdgpu_buffer_rsrc_t &rw, int chunk, int buf_byte) {
        static_for_<CHUNK / 4>([&](auto I) {
            const int r = wave * (CHUNK / 4) + I;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void *)(reinterpret_cast<char *>(lds) + buf_byte + r * 1024), 16, woff,
                                                     (uint32_t)(chunk * CHUNK + r) * 1024u, 0, 0);
        });
    };
    auto w_rsrc = [&](int l) { return rows_rsrc(a.layer[l].recs, (long long)a.layer[l].kblocks * NT * 1024); };
    auto x_rsrc = [&](int l, long long row0) { return rows_rsrc(a.layer[l].X + row0 * a.layer[l].ldx, (a.M - row0) * a.layer[l].ldx * 4); };
    auto xload4 = [&](const __amdgpu_buffer_rsrc_t &rx, uint32_t xoff, int q0, f32x4 (&x)[4]) {
        static_for_<4>([&](auto I) { x[I] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, xoff, (uint32_t)(q0 + I) * 32u, 0)); });
    };
    auto x_off = [&](int l) { return (uint32_t)(m * a.layer[l].ldx + 4 * h) * 4u; };
    f32x16 accs[2][NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)                       // set 1 "leaves" once before it has been filled (into an empty descriptor): keep even that read defined
#pragma unroll
        for (int r = 0; r < 16; ++r) accs[1][t][r] = 0.0f;
    f32x4 x[4], xn[4], ring[RING];
    u32x4 bits_set[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};   // dX: the masks of the blocks in the two accumulator sets, asked for in a block's last body (every body without an epilogue asks: no branch)
    // A block's outputs are made final in place in ONE dense block of VALU work (ReLU and its bits): VALU instructions
    // scattered between f32 MFMAs cost several times their own issue time (the f32 MFMA runs on the vector ALUs' lanes;
    // measured here, 18 cycles an instruction against 8) -- only the stores, which need no ALU, are spread over the next block's MFMAs.
    // The bit of (tile t, register r) lies in word t >> 1 at position 16 (t & 1) + r.
    auto finalize = [&](auto SETc, int l, long long blk) {
        constexpr int SET = SETc;
        if constexpr (MODE == ROWS_FORWARD) {
            if (a.layer[l].linear) return;
            u32x4 out_bits = {0u, 0u, 0u, 0u};
            static_for_<NT>([&](auto T) {
                static_for_<16>([&](auto R) {
                    constexpr int t = T, r = R;
                    const float v = fmaxf(accs[SET][t][r], 0.0f);
                    accs[SET][t][r] = v;
                    const int one = __builtin_bit_cast(int, v) < 1 ? __builtin_bit_cast(int, v) : 1;       // v >= 0: its bits as an integer are 0 or positive
                    out_bits[t >> 1] |= (unsigned)one << (16 * (t & 1) + r);
                });
                __builtin_amdgcn_sched_barrier(0);
            });
            const __amdgpu_buffer_rsrc_t rbo = rows_rsrc(a.layer[l].bits_out ? (const void *)a.layer[l].bits_out : (const void *)a.layer[l].recs, a.layer[l].bits_out ? n_blocks * 1024 : 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, out_bits), rbo, woff, (uint32_t)blk * 1024u, 0);
        }
    };
    // where the block in the OTHER accumulator set goes: its layer's Y at its rows (set by block())
    __amdgpu_buffer_rsrc_t ry = rows_rsrc(a.layer[0].Y, 0);
    uint32_t yoff = 0, ld4 = 0;
    // output register (tile t, register r) leaves: two full 128-byte lines
    // (dX under a mask: the mask goes on here, four outputs at a time -- 8 VALU instructions in one piece every 8 MFMAs; all 128 at once
    // need more registers than there are, one at a time between the MFMAs costs 18 us a launch)
    auto element = [&](auto SETc, auto Tc, auto Rc) {
        constexpr int SET = SETc, t = Tc, r = Rc;
        float v = accs[SET][t][r];                     // (a vector element is not an lvalue __builtin_bit_cast can take: it would read element 0)
        if constexpr (MASK) {
            const int keep = __builtin_amdgcn_sbfe((int)bits_set[SET][t >> 1], 16 * (t & 1) + r, 1);   // 0 or -1
            v = __builtin_bit_cast(float, __builtin_bit_cast(int, v) & keep);
        }
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), ry, yoff + (uint32_t)t * 128u, (uint32_t)(8 * (r >> 2) + (r & 3)) * ld4, 0);
    };
    auto leaves_to = [&](int l, long long row0) {
        ry = rows_rsrc(a.layer[l].Y + row0 * a.layer[l].ldy, (a.M - row0) * a.layer[l].ldy * 4);
        yoff = (uint32_t)(4 * h * a.layer[l].ldy + m) * 4u; ld4 = (uint32_t)a.layer[l].ldy * 4u;
    };
    auto acc_init = [&](auto SETc, int l) {
        constexpr int SET = SETc;
        if constexpr (BIAS) {                          // the accumulators start from the bias: a lane's feature is the same in all of a tile's registers
            const float *bl = reinterpret_cast<const float *>(&lds[3 * CHUNK * 64]) + l * NT * 32;
            static_for_<NT>([&](auto T) {
                const float b = bl[32 * T + m];
#pragma unroll
                for (int r = 0; r < 16; ++r) accs[SET][T][r] = b;
            });
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) accs[SET][t][r] = 0.0f;
        }
    };
    // LDS byte offsets of the chunk being read, the next one, the one after (being filled): they rotate
    int buf0 = 0, buf1 = CHUNK * 1024, buf2 = 2 * CHUNK * 1024;
    {
        const int l0 = item_layer(0);
        const __amdgpu_buffer_rsrc_t rw = w_rsrc(l0);
        fill(rw, 0, buf0); fill(rw, 1, buf1);
        xload4(x_rsrc(l0, item_row0(0)), x_off(l0), 0, x);
        wait_vmcnt<0>();
        __syncthreads();
        const f32x4 *Lp = reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(lds) + buf0) + lane;
        static_for_<RING>([&](auto I) { ring[I] = Lp[I * 64]; });
    }
    // the block under way and the one after it
    __amdgpu_buffer_rsrc_t rx = x_rsrc(0, no_rows), rxn = rx, rw = w_rsrc(0), rwn = rw, rbits = rw;
    uint32_t xoff = 0, xoffn = 0, bits_at = 0;
    int nb = 3;
    // one body: EP = 0 nothing else, 1 / 2 the first / second half of the previous block's outputs leave from the other accumulator set
    auto body = [&](auto SETc, auto EPc, int b) {
        constexpr int SET = SETc, EP = EPc;
        const bool last = b == nb - 1;
        if (nb == 3 && last) wait_vmcnt<0>();          // (a three-body block: the outputs that left in its first two bodies are what the next block's X may be)
        xload4(last ? rxn : rx, last ? xoffn : xoff, last ? 0 : 4 * (b + 1), xn);
        if constexpr (MASK && EP == 0) bits_set[SET] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rbits, woff, bits_at, 0));
        const f32x4 *Lp = reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(lds) + buf0) + lane;
        const f32x4 *Ln = reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(lds) + buf1) + lane;
        static_for_<CHUNK>([&](auto IDX) {
            constexpr int idx = IDX, ks = idx / G, gi = idx % G, qi = ks / 4, c = ks % 4;
            const f32x4 w = ring[idx % RING];
            if constexpr (idx + RING < CHUNK) ring[idx % RING] = Lp[(idx + RING) * 64];
            else ring[idx % RING] = Ln[(idx + RING - CHUNK) * 64];      // behind the barrier: the next chunk's first records
            static_for_<4>([&](auto T) {
                constexpr int tt = T, tile = 4 * gi + T;
                accs[SET][tile] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[qi][c], w[tt], accs[SET][tile], 0, 0, 0);
                // output e = 0 .. 8 NT - 1 of this half of the block: register-major, so that a sample's stores follow each other
                auto leave = [&](auto Ec) {
                    constexpr int e = Ec, r = e / (NT / 2), t = (EP - 1) * (NT / 2) + e % (NT / 2);
                    element(std::integral_constant<int, SET ^ 1>{}, std::integral_constant<int, t>{}, std::integral_constant<int, r>{});
                };
                if constexpr (EP != 0 && !MASK && (tt == 0 || tt == 2)) leave(std::integral_constant<int, 2 * idx + tt / 2>{});
                if constexpr (EP != 0 && MASK && idx % 2 == 0 && tt == 0) static_for_<4>([&](auto I) { leave(std::integral_constant<int, 2 * idx + I>{}); });
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (idx == TAIL) {
                // what is older than this body's stores -- the next chunk's quarter, the next body's X -- has to be there; then everyone's is, and
                // nobody reads the chunk before this one any more: its buffer takes the chunk after the next (of this block's image or the next's)
                wait_vmcnt<(EP == 0 ? 0 : MASK ? 4 * (TAIL / 2 + 1) : 2 * (TAIL + 1))>();
                __builtin_amdgcn_s_barrier();
                const bool over = b + 2 >= nb;
                fill(over ? rwn : rw, over ? b + 2 - nb : b + 2, buf2);
                __builtin_amdgcn_sched_barrier(0);
            }
        });
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = xn[i];
        const int tmp = buf0; buf0 = buf1; buf1 = buf2; buf2 = tmp;
    };
    // item w into accumulator set SET; the item before it (in the other set) leaves meanwhile
    auto block = [&](auto SETc, int w) {
        const int l = item_layer(w), ln = item_layer(w + 1);
        const long long row0 = item_row0(w), row0n = item_row0(w + 1);
        rx = x_rsrc(l, row0); xoff = x_off(l); rxn = x_rsrc(ln, row0n); xoffn = x_off(ln);
        rw = w_rsrc(l); rwn = w_rsrc(ln); nb = a.layer[l].kblocks / 4;
        if constexpr (MASK) { rbits = rows_rsrc(a.layer[l].bits_in, n_blocks * 1024); bits_at = (uint32_t)(row0 / 32) * 1024u; }
        leaves_to(item_layer(w - 1), item_row0(w - 1));
        acc_init(SETc, l);
        body(SETc, std::integral_constant<int, 1>{}, 0);
        body(SETc, std::integral_constant<int, 2>{}, 1);
        for (int b = 2; b < nb; ++b) body(SETc, std::integral_constant<int, 0>{}, b);
        finalize(SETc, l, row0 / 32);
    };
    for (int w = 0; w < n_items; w += 2) {
        block(std::integral_constant<int, 0>{}, w);
        block(std::integral_constant<int, 1>{}, w + 1);
    }
    // the last block's outputs (always in set 1)
    leaves_to(item_layer(n_items - 1), item_row0(n_items - 1));
    static_for_<16>([&](auto R) { static_for_<NT>([&](auto T) { element(std::integral_constant<int, 1>{}, T, R); }); });
}

// the packed images of every layer, both directions, in one launch (the weights move every step)
struct PackJob {
    const float *src; long long sk, sc;               // k < K1: Wsrc[k][col] = src[k * sk + col * sc]
    const float *src2; long long sk2, sc2;            // K1p <= k < K1p + K2: src2[(k - K1p) * sk2 + col * sc2]
    int K1, K1p, K2;                                  // K1 <= k < K1p: the pad of a concat buffer, zero; zero behind K1p + K2
    int nt, kblocks;
    float *dst; long long first;                      // where the image lies; the job's first float in the launch's index space
};
constexpr int MAX_PACK_JOBS = 24;
struct PackArgs { PackJob job[MAX_PACK_JOBS]; int n_jobs; long long total; };
__global__ void pack_records_kernel(PackArgs a) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= a.total) return;
    int j = 0;
    while (j + 1 < a.n_jobs && e >= a.job[j + 1].first) ++j;
    const PackJob &p = a.job[j];
    const long long o = e - p.first;
    const int c = (int)(o & 3), lane = (int)((o >> 2) & 63);
    const long long rec = o >> 8;
    const int G = p.nt / 4, g = (int)(rec % G), s = (int)(rec / G);
    const int f = lane & 31, h = lane >> 5;
    const int k = 8 * (s >> 2) + 4 * h + (s & 3), col = 32 * (4 * g + c) + f;
    float v = 0.0f;
    if (k < p.K1) v = p.src[k * p.sk + col * p.sc];
    else if (k >= p.K1p && k < p.K1p + p.K2) v = p.src2[(k - p.K1p) * p.sk2 + col * p.sc2];
    p.dst[o] = v;
}

// ---------------------------------------------------------------------------------------------------------------------------
// encoder: layer.FourierFeatures (layer.py:8-23) of position [+ geometry parameters] and of direction [+ appearance parameters]
// (model.py:77-101), the sample points of renderer.py:98-114 and the blur product of :155-158; thread per sample
// ---------------------------------------------------------------------------------------------------------------------------
struct EncodeArgs {
    const float *rays_o, *rays_d, *z, *params, *cone;
    long long rays_per_param_row;
    int n_rays, S, n_geo, n_app, pos_freq, dir_freq, param_freq, blur_idx;
    float *pos_out; int ld_pos;      // [M][ld_pos]: pos_map in columns 0 .. Kp
    float *dir_out; int ld_dir;      // [M][ld_dir]: dir_map in columns 0 .. Kd
    float *dists;                    // [N][S]: z[i+1] - z[i], the last one repeated, times |rays_d| (renderer.py:174-180)
};
// feature j of layer.FourierFeatures over x[0 .. d): [x | sin(x), cos(x) | sin(2 x), cos(2 x) | ...] (layer.py:14-23)
__device__ __forceinline__ float fourier_feature(const float *x, int d, int j) {
    if (j < d) return x[j];
    const int jj = j - d, k = jj / (2 * d), r = jj - k * 2 * d;
    const float arg = ldexpf(1.0f, k) * (r < d ? x[r] : x[r - d]);
    return r < d ? sinf(arg) : cosf(arg);
}
// 64 samples a block: every thread works out a quarter of its sample's features into LDS, then the block writes the rows out in runs of
// whole feature vectors (a thread writing its own sample's 153 numbers one by one, 1.3 KB from its neighbour's, ran at 0.8 TB/s)
constexpr int ENC_SAMPLES = 64, ENC_BASE = 24;           // per sample in LDS: its features, and in front of them pos (3), dir (3), parameters (16)
__global__ __launch_bounds__(256) void encode_kernel(EncodeArgs a) {
    extern __shared__ float enc_lds[];                     // [ENC_SAMPLES][ENC_BASE], then [ENC_SAMPLES][Kp + Kd]
    float *enc_tile = enc_lds + ENC_SAMPLES * ENC_BASE;
    const long long M = (long long)a.n_rays * a.S;
    const int ls = threadIdx.x & (ENC_SAMPLES - 1), part = threadIdx.x / ENC_SAMPLES;
    const long long m = (long long)blockIdx.x * ENC_SAMPLES + ls;
    const int P = a.n_geo + a.n_app;
    const int Kp3 = 3 * (1 + 2 * a.pos_freq), Kpg = a.n_geo * (1 + 2 * a.param_freq), Kp = Kp3 + Kpg;
    const int Kd3 = 3 * (1 + 2 * a.dir_freq), Kda = a.n_app * (1 + 2 * a.param_freq), Kd = Kd3 + Kda, KF = Kp + Kd;
    float *base = enc_lds + ls * ENC_BASE;                  // (registers cannot be indexed by a feature's number: the sample's inputs go through LDS)
    if (m < M && part == 0) {
        const int ray = (int)(m / a.S), s = (int)(m - (long long)ray * a.S);
        const float o[3] = {a.rays_o[3 * ray], a.rays_o[3 * ray + 1], a.rays_o[3 * ray + 2]};
        const float d[3] = {a.rays_d[3 * ray], a.rays_d[3 * ray + 1], a.rays_d[3 * ray + 2]};
        const float dn = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
        // A ray that misses the proxy (t = inf: the reference's Renderer.__call__ filters it out and scatters 0 / the background back,
        // renderer.py:58-86) stays in the batch with depth 0 and distances 0: every alpha of it is 1 - exp(-sigma 0) = 0, so it composites to
        // exactly 0 / the background, no gradient flows into or out of its rows, and the loss still counts it among its rays
        const float zr = a.z[(size_t)ray * a.S + s];
        const bool hit = isfinite(zr);
        const float z = hit ? zr : 0.0f;
        for (int c = 0; c < 3; ++c) { base[c] = o[c] + d[c] * z; base[3 + c] = d[c] / dn; }   // renderer.py:114 (un-normalised rays_d), :98
        const float *pr = a.params + (size_t)(ray / a.rays_per_param_row) * P;
        for (int c = 0; c < P; ++c) base[6 + c] = c == a.blur_idx ? (hit ? pr[c] * (a.cone[ray] * z) : 0.0f) : pr[c];   // :155-158 (a missing ray's cone scale may be anything)
        const float zn = s + 1 < a.S ? a.z[(size_t)ray * a.S + s + 1] : 0.0f;
        const float dist = s + 1 < a.S ? zn - z : (a.S > 1 ? z - a.z[(size_t)ray * a.S + s - 1] : 0.0f);
        a.dists[(size_t)ray * a.S + s] = hit ? dist * dn : 0.0f;
    }
    __syncthreads();
    if (m < M) {
        float *row = enc_tile + ls * KF;
        for (int j = part; j < KF; j += 4) {                                                  // model.py:88-101
            float v;
            if (j < Kp3) v = fourier_feature(base, 3, j);
            else if (j < Kp) v = fourier_feature(base + 6, a.n_geo, j - Kp3);
            else if (j < Kp + Kd3) v = fourier_feature(base + 3, 3, j - Kp);
            else v = fourier_feature(base + 6 + a.n_geo, a.n_app, j - Kp - Kd3);
            row[j] = v;
        }
    }
    __syncthreads();
    const long long m0 = (long long)blockIdx.x * ENC_SAMPLES;
    for (int e = threadIdx.x; e < ENC_SAMPLES * Kp; e += 256) {
        const int r = e / Kp, j = e - r * Kp;
        if (m0 + r < M) a.pos_out[(size_t)(m0 + r) * a.ld_pos + j] = enc_tile[r * KF + j];
    }
    for (int e = threadIdx.x; e < ENC_SAMPLES * Kd; e += 256) {
        const int r = e / Kd, j = e - r * Kd;
        if (m0 + r < M) a.dir_out[(size_t)(m0 + r) * a.ld_dir + j] = enc_tile[r * KF + Kp + j];
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// the narrow heads on the vector ALUs: y[m][c] = x[m] . W[:, c] + b[c] for n_out = 1 (alpha) or 3 (color)
// ---------------------------------------------------------------------------------------------------------------------------
// A row is read by K / 4 neighbouring lanes, 16 bytes each (a wave's load is one or two whole rows); their partial sums meet by butterfly.
// A thread takes four rows, a block's worth of rows apart: four loads in flight.  K / 4 = 64 or 32.
__global__ __launch_bounds__(256) void head_forward_kernel(const float *__restrict__ X, int ldx, int K, const float *__restrict__ W, const float *__restrict__ b, int n_out,
                                                           long long M, float *__restrict__ Y) {
    const int q = K / 4, kq = threadIdx.x % q, rows = 256 / q;
    const long long row0 = (long long)blockIdx.x * (4 * rows) + threadIdx.x / q;
    f32x4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long long m = row0 + i * rows;
        v[i] = m < M ? *reinterpret_cast<const f32x4 *>(X + (size_t)m * ldx + 4 * kq) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
    float w[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        for (int c = 0; c < 3; ++c) w[j][c] = c < n_out ? W[(4 * kq + j) * n_out + c] : 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float e[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
        float acc[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[c] += e[j] * w[j][c];
        for (int o = q / 2; o > 0; o >>= 1)
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[c] += __shfl_xor(acc[c], o);
        const long long m = row0 + i * rows;
        if (kq == 0 && m < M)
            for (int c = 0; c < n_out; ++c) Y[(size_t)m * n_out + c] = acc[c] + b[c];
    }
}
// dX[m][k] (+)= sum_c dY[m][c] W[k][c], kept where mask > 0 (mask NULL: everywhere); thread per four neighbouring k of a row (K, lddx and
// ldmask are multiples of 4)
__global__ void head_backward_dx_kernel(const float *__restrict__ dY, int n_out, const float *__restrict__ W, int K, long long M, const float *__restrict__ mask,
                                        int ldmask, int accumulate, float *__restrict__ dX, int lddx) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int q = K / 4;
    if (e >= M * q) return;
    const long long m = e / q; const int k = 4 * (int)(e - m * q);
    float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int c = 0; c < n_out; ++c) {
        const float d = dY[(size_t)m * n_out + c];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += d * W[(k + j) * n_out + c];
    }
    f32x4 *out = reinterpret_cast<f32x4 *>(dX + (size_t)m * lddx + k);
    if (accumulate) { const f32x4 o = *out; v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w; }
    if (mask) {
        const f32x4 k4 = *reinterpret_cast<const f32x4 *>(mask + (size_t)m * ldmask + k);
        if (!(k4.x > 0.0f)) v[0] = 0.0f;
        if (!(k4.y > 0.0f)) v[1] = 0.0f;
        if (!(k4.z > 0.0f)) v[2] = 0.0f;
        if (!(k4.w > 0.0f)) v[3] = 0.0f;
    }
    *out = f32x4{v[0], v[1], v[2], v[3]};
}
// dst[m * ld] = src[m]: a column of a row-major matrix
__global__ void column_kernel(const float *__restrict__ src, long long M, float *__restrict__ dst, int ld) {
    const long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (m < M) dst[m * ld] = src[m];
}
// (dW | db) partial over a block of `rows` rows: partial[b][K + 1][n_out], row k < K = sum_m X[m][k] dY[m][c], row K = sum_m dY[m][c] (the bias
// row: kernel and bias are neighbours in the blob).  K / 4 neighbouring threads read a row 16 bytes each; 256 / (K / 4) such groups take every
// group-count-th row of the block, eight rows' loads in flight each, and their sums meet in LDS in group order (a fixed order of additions).
__global__ __launch_bounds__(256) void head_backward_dw_partial_kernel(const float *__restrict__ X, int ldx, int K, const float *__restrict__ dY, int n_out, long long M, int rows,
                                                                       float *__restrict__ partial) {
    __shared__ float part[8][(256 + 1) * 3];
    const int q = K / 4, groups = 256 / q, grp = threadIdx.x / q, c4 = threadIdx.x % q;
    const long long m0 = (long long)blockIdx.x * rows, m1 = m0 + rows < M ? m0 + rows : M;
    float acc[4][3], bias[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[j][c] = 0.0f;
    for (long long m = m0 + grp; m < m1; m += 8 * groups) {
        f32x4 x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const long long r = m + (long long)i * groups;
            x[i] = r < m1 ? *reinterpret_cast<const f32x4 *>(X + (size_t)r * ldx + 4 * c4) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const long long r = m + (long long)i * groups;
            if (r < m1) {
                const float e[4] = {x[i].x, x[i].y, x[i].z, x[i].w};
                for (int c = 0; c < n_out; ++c) {
                    const float d = dY[(size_t)r * n_out + c];
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j][c] += e[j] * d;
                    if (c4 == 0) bias[c] += d;
                }
            }
        }
    }
    for (int c = 0; c < n_out; ++c) {
#pragma unroll
        for (int j = 0; j < 4; ++j) part[grp][(4 * c4 + j) * n_out + c] = acc[j][c];
        if (c4 == 0) part[grp][K * n_out + c] = bias[c];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < (K + 1) * n_out; e += 256) {
        float total = part[0][e];
        for (int g = 1; g < groups; ++g) total += part[g][e];
        partial[(size_t)blockIdx.x * (K + 1) * n_out + e] = total;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// map_model_output (renderer.py:170-213) per ray and its adjoint; wave per ray, lane l holds samples l, l + 64, ...
// ---------------------------------------------------------------------------------------------------------------------------
struct CompositeArgs {
    const float *raw_rgb, *sigma, *dists;      // [N][S][3], [N][S], [N][S]
    const float *noise;                        // NULL, or [N][S]: raw_noise_std * N(0,1), added to the density before its ReLU (renderer.py:190-195)
    int n_rays, S, map_exr, composite_bkgd; float bkgd[3];
    float *color, *alpha;                      // forward outputs [N][3], [N]
    const float *d_color, *d_alpha;            // backward inputs
    float *d_raw_rgb, *d_sigma;                // backward outputs
};
constexpr int MAX_TRAIN_SAMPLES = 1024;
__device__ __forceinline__ float wave_sumf(float v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }
__device__ __forceinline__ float rgb_of(float raw, int map_exr) {
    if (map_exr) return raw > 0.0f ? raw + 1.0f : expf(raw);                                   // elu + 1 (:184-185)
    return 1.0f / (1.0f + expf(-raw));                                                         // sigmoid (:187)
}
template <bool BACKWARD>
__global__ __launch_bounds__(256) void composite_kernel(CompositeArgs a) {
    __shared__ float sh_a[4][MAX_TRAIN_SAMPLES], sh_T[4][MAX_TRAIN_SAMPLES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ray = blockIdx.x * 4 + wave;
    if (ray >= a.n_rays) return;
    const int S = a.S;
    const float *sg = a.sigma + (size_t)ray * S, *ds = a.dists + (size_t)ray * S, *rr = a.raw_rgb + (size_t)ray * S * 3;
    const float *nz = a.noise ? a.noise + (size_t)ray * S : nullptr;
    float *al = sh_a[wave], *T = sh_T[wave];
    for (int s = lane; s < S; s += 64) { const float v = nz ? sg[s] + nz[s] : sg[s]; const float r = v > 0.0f ? v : 0.0f; al[s] = 1.0f - expf(-r * ds[s]); }    // :190-195
    __builtin_amdgcn_wave_barrier();
    // exclusive running product of (1 - a) + 1e-10, sequential like tf.math.cumprod (:198): chunks of 64 with a carry
    float carry = 1.0f;
    for (int s0 = 0; s0 < S; s0 += 64) {
        const int s = s0 + lane;
        float f = s < S ? (1.0f - al[s]) + 1e-10f : 1.0f, incl = f;
        for (int o = 1; o < 64; o <<= 1) { const float w = __shfl_up(incl, o); if (lane >= o) incl *= w; }
        float excl = __shfl_up(incl, 1);
        if (lane == 0) excl = 1.0f;
        if (s < S) T[s] = carry * excl;
        carry *= __shfl(incl, 63);
    }
    __builtin_amdgcn_wave_barrier();
    float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f, A = 0.0f;
    for (int s = lane; s < S; s += 64) {
        const float w = al[s] * T[s];
        c0 += w * rgb_of(rr[3 * s], a.map_exr); c1 += w * rgb_of(rr[3 * s + 1], a.map_exr); c2 += w * rgb_of(rr[3 * s + 2], a.map_exr);
        A += w;
    }
    c0 = wave_sumf(c0); c1 = wave_sumf(c1); c2 = wave_sumf(c2); A = wave_sumf(A);
    if (!BACKWARD) {
        if (a.composite_bkgd) { c0 += (1.0f - A) * a.bkgd[0]; c1 += (1.0f - A) * a.bkgd[1]; c2 += (1.0f - A) * a.bkgd[2]; }   // :210-211
        if (lane == 0) { a.color[3 * ray] = c0; a.color[3 * ray + 1] = c1; a.color[3 * ray + 2] = c2; a.alpha[ray] = A; }
        return;
    }
    // adjoint.  C = sum w rgb (+ (1 - A) bkgd), A = sum w, w_i = a_i T_i, T_i = prod_{j<i} ((1 - a_j) + 1e-10):
    //   g_i = dL/dw_i = dC . rgb_i + dA';   dL/da_i = T_i g_i - (sum_{k>i} g_k w_k) / ((1 - a_i) + 1e-10)
    const float dC[3] = {a.d_color[3 * ray], a.d_color[3 * ray + 1], a.d_color[3 * ray + 2]};
    float dA = a.d_alpha[ray];
    if (a.composite_bkgd) dA -= (dC[0] * a.bkgd[0] + dC[1] * a.bkgd[1]) + dC[2] * a.bkgd[2];
    float suffix = 0.0f;                                   // sum of g_k w_k over the samples behind the current chunk
    for (int s0 = ((S - 1) / 64) * 64; s0 >= 0; s0 -= 64) {
        const int s = s0 + lane;
        float gw = 0.0f, g = 0.0f, rgb[3] = {0.0f, 0.0f, 0.0f};
        if (s < S) {
            for (int c = 0; c < 3; ++c) rgb[c] = rgb_of(rr[3 * s + c], a.map_exr);
            g = ((dC[0] * rgb[0] + dC[1] * rgb[1]) + dC[2] * rgb[2]) + dA;
            gw = g * (al[s] * T[s]);
        }
        float incl = gw;                                   // inclusive suffix sum inside the chunk
        for (int o = 1; o < 64; o <<= 1) { const float w = __shfl_down(incl, o); if (lane + o < 64) incl += w; }
        const float behind = (incl - gw) + suffix;         // strictly behind s
        if (s < S) {
            const float w = al[s] * T[s];
            const float d_a = T[s] * g - behind / ((1.0f - al[s]) + 1e-10f);
            const float sig = nz ? sg[s] + nz[s] : sg[s];
            a.d_sigma[(size_t)ray * S + s] = sig > 0.0f ? d_a * ds[s] * expf(-sig * ds[s]) : 0.0f;
            for (int c = 0; c < 3; ++c) {
                const float raw = rr[3 * s + c];
                const float drgb = a.map_exr ? (raw > 0.0f ? 1.0f : expf(raw)) : rgb[c] * (1.0f - rgb[c]);
                a.d_raw_rgb[((size_t)ray * S + s) * 3 + c] = w * dC[c] * drgb;
            }
        }
        suffix += __shfl(incl, 0);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// loss.py: NerfLoss / AlphaLoss over mse / smape; one workgroup (a training batch is a thousand rays).  Value and gradient.
// ---------------------------------------------------------------------------------------------------------------------------
struct LossArgs {
    const float *color_true, *alpha_true, *color_pred, *alpha_pred;
    int n_rays, kind, loss_fn, alpha_loss_fn, filter_color_loss, use_hard_mask; float gamma;
    float *loss, *d_color, *d_alpha;
};
__device__ __forceinline__ void loss_term(int fn, float t, float p, float inv_n, float &value, float &grad) {
    if (fn == NTX_LOSS_MSE) { const float e = t - p; value = e * e * inv_n; grad = -2.0f * e * inv_n; }          // loss.py:51-54
    else {                                                                                                        // smape, eps 1e-2 (:56-59)
        const float e = t - p, den = (t + p) + 1e-2f, ae = fabsf(e);
        const float sgn = e > 0.0f ? 1.0f : (e < 0.0f ? -1.0f : 0.0f);
        value = ae / den * inv_n; grad = (-sgn / den - ae / (den * den)) * inv_n;
    }
}
__global__ __launch_bounds__(1024) void loss_kernel(LossArgs a) {
    __shared__ float red[1024];
    float total = 0.0f;
    const float inv_c = 1.0f / (float)(a.n_rays * 3), inv_a = 1.0f / (float)a.n_rays;
    for (int r = threadIdx.x; r < a.n_rays; r += blockDim.x) {
        float mask = 1.0f;
        if (a.kind == NTX_LOSS_ALPHA && a.filter_color_loss) mask = a.use_hard_mask ? (a.alpha_true[r] > 0.0f ? 1.0f : 0.0f) : a.alpha_true[r];   // :29-35
        for (int c = 0; c < 3; ++c) {
            float v, gr;
            loss_term(a.loss_fn, a.color_true[3 * r + c] * mask, a.color_pred[3 * r + c] * mask, inv_c, v, gr);
            total += v; a.d_color[3 * r + c] = gr * mask;
        }
        float ga = 0.0f;
        if (a.kind == NTX_LOSS_ALPHA) { float v; loss_term(a.alpha_loss_fn, a.alpha_true[r], a.alpha_pred[r], inv_a, v, ga); total += a.gamma * v; ga *= a.gamma; }   // :38
        a.d_alpha[r] = ga;
    }
    red[threadIdx.x] = total;
    __syncthreads();
    for (int o = blockDim.x / 2; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) *a.loss = red[0];
}

// tf.keras.optimizers.Adam (TF 2.4, non-amsgrad): m = b1 m + (1 - b1) g; v = b2 v + (1 - b2) g^2;
// w -= lr sqrt(1 - b2^t) / (1 - b1^t) * m / (sqrt(v) + eps), t = iterations + 1; lr from ExponentialDecay (train.py:49-52) on the host
__global__ void adam_kernel(float *__restrict__ w, const float *__restrict__ g, float *__restrict__ m, float *__restrict__ v, long long n, float lr_t, float b1,
                            float b2, float eps) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const float ge = g[e];
    const float me = m[e] + (ge - m[e]) * (1.0f - b1);
    const float ve = v[e] + (ge * ge - v[e]) * (1.0f - b2);
    m[e] = me; v[e] = ve;
    w[e] = w[e] - (me * lr_t) / (sqrtf(ve) + eps);
}

}   // namespace ntx_train

// ---------------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------------
struct TLayer { int in, out; size_t w, b; };      // offsets into the Keras-order blob (kernel [in][out], then bias)

struct ntx_trainer {
    int device = 0;
    ntx_model_desc desc{};
    int Kp = 0, Kd = 0, P = 0, Kp4 = 0, Kd4 = 0, ldp = 0, ldd = 0;   // the two concat buffers: [pos_map | pad to 16 bytes | h4], row stride ldp = Kp4 + 256; [dir_map | pad | feature], ldd
    TLayer trunk[8], feature, c1, c2, rgb, alpha;
    size_t n_weights = 0;
    long long cap = 0;                         // samples the buffers hold
    float *w = nullptr, *wp = nullptr, *grad = nullptr, *adam_m = nullptr, *adam_v = nullptr;   // wp: the packed images rows_kernel streams (made once a step)
    ntx_train::PackArgs pack{};
    const float *fwd_recs[11] = {}; int fwd_kblocks[11] = {};   // trunk 0-7, feature, c1, c2
    const float *bwd_recs[10] = {}; int bwd_kblocks[10] = {};   // trunk 1-7 (index i - 1), feature, c1, c2
    // activations (per sample): h[i] = output of trunk layer i (h[4] lives inside h4c), c1o, c2o; concat buffers; heads' raw outputs
    float *h[8] = {}, *h4c = nullptr, *fc = nullptr, *c1o = nullptr, *c2o = nullptr, *raw_rgb = nullptr, *sigma = nullptr;
    unsigned int *bits_ones = nullptr;         // all set: the "mask" of a layer without a ReLU behind it
    unsigned int *bits[9] = {};                // where h[0..7] and c1o are > 0, one bit per output in rows_kernel's layout (1 KiB per 32 samples)
    float *gf = nullptr;                       // [d feature (256) | d_sigma | 3 zeros] per sample, row stride LDGF
    float *dyt[8] = {};                        // the gradient at every trunk layer's output (what its dW contracts with): kept, so that all dW run in one launch
    float *dw_partial = nullptr; size_t dw_partial_floats = 0;
    float *noise = nullptr;                    // [M]: the density regulariser's draws of a step (raw_noise_std)
    float *z = nullptr, *dists = nullptr, *g0 = nullptr, *g1 = nullptr, *d_raw = nullptr, *d_sigma = nullptr, *partial = nullptr;
    float *color = nullptr, *alpha_out = nullptr, *d_color = nullptr, *d_alpha = nullptr, *loss = nullptr;
    long long cap_rays = 0;
    size_t partial_floats = 0;
    long long adam_iterations = 0;
};

namespace {

using namespace ntx_train;

constexpr int SPLIT = 128;        // partial sums of a weight gradient along the samples
constexpr int HEAD_ROWS = 512;    // rows per block of the narrow reductions
constexpr int LDGF = 260;

void free_all(ntx_trainer *t) {
    if (!t) return;
    (void)hipSetDevice(t->device);
    void *ptrs[] = {t->w, t->wp, t->grad, t->adam_m, t->adam_v, t->h4c, t->fc, t->c1o, t->c2o, t->raw_rgb, t->sigma, t->z, t->dists, t->noise, t->g0, t->g1, t->gf, t->d_raw, t->d_sigma,
                    t->partial, t->color, t->alpha_out, t->d_color, t->d_alpha, t->loss};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    for (int i = 0; i < 8; ++i) if (i != 4 && t->h[i]) (void)hipFree(t->h[i]);
    for (int i = 0; i < 9; ++i) if (t->bits[i]) (void)hipFree(t->bits[i]);
    if (t->bits_ones) (void)hipFree(t->bits_ones);
    for (int i = 0; i < 8; ++i) if (t->dyt[i]) (void)hipFree(t->dyt[i]);
    if (t->dw_partial) (void)hipFree(t->dw_partial);
    delete t;
}

// one contraction on caller buffers (ntx_gemm_f32): 128 x 128 tiles, panels of 16 -- the shape every measurement of the round ended on
// (deeper panels or 256-wide tiles cost occupancy: 0.552 / 0.467 of the peak in the step against 0.58)
template <bool AK>
void launch_gemm(hipStream_t st, GemmArgs g) {
    g.k_chunk = g.K;
    g.aligned = (g.lda % 4 == 0) && (g.ldb % 4 == 0) && (((uintptr_t)g.A | (uintptr_t)g.B) % 16 == 0);
    hipLaunchKernelGGL((gemm_kernel<AK, 128, 16>), dim3((g.N + 127) / 128, (g.M + TM - 1) / TM, 1), dim3(256), 0, st, g);
}

// a chain of layers in one launch: forward Y = act(X . W + b) through each layer's packed image, or dX = dY . W^T kept where the forward
// pass left a bit
void launch_rows(hipStream_t st, const RowsArgs &a, int N, bool forward) {
    static const int cus = [] { int dev = 0, n = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
    const long long n_groups = ((a.M + 31) / 32 + 3) / 4;
    const dim3 grid((unsigned)(n_groups < cus ? n_groups : cus)), wg(256);     // persistent: a workgroup of four waves per CU
    if (N != 256) hipLaunchKernelGGL((rows_kernel<4, ROWS_FORWARD>), grid, wg, 0, st, a);
    else if (forward) hipLaunchKernelGGL((rows_kernel<8, ROWS_FORWARD>), grid, wg, 0, st, a);
    else hipLaunchKernelGGL((rows_kernel<8, ROWS_DX_MASK>), grid, wg, 0, st, a);
}
RowsLayer forward_layer(const float *X, int ldx, const float *recs, int kblocks, const float *b, float *Y, int ldy, int relu, unsigned int *bits_out) {
    RowsLayer r{}; r.X = X; r.ldx = ldx; r.recs = recs; r.kblocks = kblocks; r.Y = Y; r.ldy = ldy; r.bias = b; r.linear = !relu; r.bits_out = bits_out;
    return r;
}
// dY [M][..] (row stride lddy) -> dX [M][256]
RowsLayer dx_layer(const float *dY, int lddy, const float *recs, int kblocks, const unsigned int *bits, float *dX, int lddx) {
    RowsLayer r{}; r.X = dY; r.ldx = lddy; r.recs = recs; r.kblocks = kblocks; r.Y = dX; r.ldy = lddx; r.bits_in = bits;
    return r;
}

}   // namespace

extern "C" {

int ntx_sample_depths(const float *t, int64_t n_rays, int n_points, uint32_t flags, uint64_t perturb_seed, const ntx_render_opts *opts, float *z_out,
                      ntx_stream stream);
int ntx_sample_noise(int64_t n_rays, int n_points, uint64_t seed, const ntx_render_opts *opts, float *noise_out, ntx_stream stream);

int ntx_trainer_create(const ntx_model_desc *desc, const float *weights, size_t n_floats, int device, int64_t max_rays, int max_samples_per_ray, ntx_trainer **out) {
    if (!out) return ntx_set_error(NTX_E_INVALID, "out is NULL");
    *out = nullptr;
    if (!desc || !weights) return ntx_set_error(NTX_E_INVALID, "desc / weights is NULL");
    if (desc->kind != NTX_MODEL_PARAMNERF || desc->depth != 8 || desc->width != 256 || desc->skip != 4 || desc->color_depth != 1 || desc->n_pos != 3 ||
        desc->pos_encoding != NTX_POS_FOURIER)
        return ntx_set_error(NTX_E_UNSUPPORTED, "training is built for the ParamNerf architecture of the shipped training configs (depth 8, width 256, skips [4], color_depth 1, "
                                                "Fourier features); others render but do not train");
    if (desc->n_geo < 0 || desc->n_app < 0 || desc->n_geo + desc->n_app > 16) return ntx_set_error(NTX_E_INVALID, "n_parameters out of range");
    if (max_rays < 1 || max_samples_per_ray < 2 || max_samples_per_ray > MAX_TRAIN_SAMPLES || max_rays * (int64_t)max_samples_per_ray > (int64_t)1 << 30)
        return ntx_set_error(NTX_E_INVALID, "max_rays / max_samples_per_ray out of range (samples per ray <= %d)", MAX_TRAIN_SAMPLES);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ntx_set_error(NTX_E_NODEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return ntx_set_error(NTX_E_INVALID, "device %d out of range [0,%d)", device, ndev);
    ntx_trainer *t = new ntx_trainer();
    t->device = device; t->desc = *desc;
    t->P = desc->n_geo + desc->n_app;
    t->Kp = 3 * (1 + 2 * desc->pos_freq) + desc->n_geo * (1 + 2 * desc->param_freq);
    t->Kd = 3 * (1 + 2 * desc->dir_freq) + desc->n_app * (1 + 2 * desc->param_freq);
    size_t p = 0;
    auto take = [&](int in, int o) { TLayer l{in, o, p, p + (size_t)in * o}; p += (size_t)in * o + o; return l; };
    int k = t->Kp;
    for (int i = 0; i < 8; ++i) { t->trunk[i] = take(k, 256); k = 256 + (i == 4 ? t->Kp : 0); }       // model.py:104-108
    t->feature = take(256, 256); t->c1 = take(256 + t->Kd, 256); t->c2 = take(256, 128); t->rgb = take(128, 3); t->alpha = take(256, 1);   // Keras order: alpha last
    t->n_weights = p;
    t->Kp4 = (t->Kp + 3) / 4 * 4; t->Kd4 = (t->Kd + 3) / 4 * 4;
    t->ldp = t->Kp4 + 256; t->ldd = t->Kd4 + 256;
    if (n_floats != p) { delete t; return ntx_set_error(NTX_E_INVALID, "weights: %zu floats, the model has %zu", n_floats, p); }
    const long long M = (long long)max_rays * max_samples_per_ray;
    t->cap = M; t->cap_rays = max_rays;
    auto alloc = [&](float **d, size_t n) -> int { TRAIN_TRY(hipMalloc((void **)d, (n ? n : 1) * sizeof(float))); return NTX_OK; };
    int rc = hipSetDevice(device) == hipSuccess ? NTX_OK : ntx_set_error(NTX_E_HIP, "hipSetDevice(%d) failed", device);
    if (rc == NTX_OK) rc = alloc(&t->w, p);
    if (rc == NTX_OK) rc = alloc(&t->grad, p);
    if (rc == NTX_OK) rc = alloc(&t->adam_m, p);
    if (rc == NTX_OK) rc = alloc(&t->adam_v, p);
    if (rc == NTX_OK) rc = alloc(&t->h4c, (size_t)M * t->ldp);
    if (rc == NTX_OK) rc = alloc(&t->fc, (size_t)M * t->ldd);
    for (int i = 0; i < 8 && rc == NTX_OK; ++i)
        if (i != 4) rc = alloc(&t->h[i], (size_t)M * 256);
    for (int i = 0; i < 9 && rc == NTX_OK; ++i) rc = alloc((float **)&t->bits[i], (size_t)((M + 31) / 32) * 256);
    if (rc == NTX_OK) rc = alloc((float **)&t->bits_ones, (size_t)((M + 31) / 32) * 256);
    if (rc == NTX_OK && hipMemset(t->bits_ones, 0xFF, (size_t)((M + 31) / 32) * 1024) != hipSuccess) rc = ntx_set_error(NTX_E_HIP, "hipMemset failed");
    if (rc == NTX_OK) rc = alloc(&t->c1o, (size_t)M * 256);
    if (rc == NTX_OK) rc = alloc(&t->c2o, (size_t)M * 128);
    if (rc == NTX_OK) rc = alloc(&t->raw_rgb, (size_t)M * 3);
    if (rc == NTX_OK) rc = alloc(&t->sigma, (size_t)M);
    if (rc == NTX_OK) rc = alloc(&t->z, (size_t)M);
    if (rc == NTX_OK) rc = alloc(&t->dists, (size_t)M);
    if (rc == NTX_OK) rc = alloc(&t->noise, (size_t)M);
    if (rc == NTX_OK) rc = alloc(&t->g0, (size_t)M * 256);
    if (rc == NTX_OK) rc = alloc(&t->g1, (size_t)M * 256);
    if (rc == NTX_OK) rc = alloc(&t->gf, (size_t)M * LDGF);
    for (int i = 0; i < 8 && rc == NTX_OK; ++i) rc = alloc(&t->dyt[i], (size_t)M * 256);
    {
        size_t per_split = 0;
        for (int i = 0; i < 8; ++i) per_split += (size_t)(t->trunk[i].in + 1) * 256;
        per_split += (size_t)(t->feature.in + 1) * 256 + (size_t)(t->c1.in + 2) * 256 + (size_t)(t->c2.in + 1) * 128 + 256;
        t->dw_partial_floats = (size_t)SPLIT * per_split;
        if (rc == NTX_OK) rc = alloc(&t->dw_partial, t->dw_partial_floats);
    }
    if (rc == NTX_OK) rc = alloc(&t->d_raw, (size_t)M * 3);
    if (rc == NTX_OK) rc = alloc(&t->d_sigma, (size_t)M);
    t->partial_floats = (size_t)SPLIT * (256 + (t->Kd > t->Kp ? t->Kd : t->Kp)) * 256 + (size_t)SPLIT * 256;
    {
        const size_t head = (size_t)((M + HEAD_ROWS - 1) / HEAD_ROWS) * 257 * 3;
        if (head > t->partial_floats) t->partial_floats = head;
    }
    if (rc == NTX_OK) rc = alloc(&t->partial, t->partial_floats);
    if (rc == NTX_OK) rc = alloc(&t->color, (size_t)max_rays * 3);
    if (rc == NTX_OK) rc = alloc(&t->alpha_out, (size_t)max_rays);
    if (rc == NTX_OK) rc = alloc(&t->d_color, (size_t)max_rays * 3);
    if (rc == NTX_OK) rc = alloc(&t->d_alpha, (size_t)max_rays);
    if (rc == NTX_OK) rc = alloc(&t->loss, 1);
    if (rc == NTX_OK && hipMemcpy(t->w, weights, p * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) rc = ntx_set_error(NTX_E_HIP, "weight upload failed");
    if (rc == NTX_OK && (hipMemset(t->adam_m, 0, p * sizeof(float)) != hipSuccess || hipMemset(t->adam_v, 0, p * sizeof(float)) != hipSuccess || hipMemset(t->grad, 0, p * sizeof(float)) != hipSuccess))
        rc = ntx_set_error(NTX_E_HIP, "hipMemset failed");
    if (rc != NTX_OK) { free_all(t); return rc; }
    // the pad columns of the concat buffers meet zero weights: they only have to be finite
    if (hipMemset(t->h4c, 0, (size_t)M * t->ldp * sizeof(float)) != hipSuccess || hipMemset(t->fc, 0, (size_t)M * t->ldd * sizeof(float)) != hipSuccess ||
        hipMemset(t->gf, 0, (size_t)M * LDGF * sizeof(float)) != hipSuccess) {
        free_all(t); return ntx_set_error(NTX_E_HIP, "hipMemset failed");
    }
    t->h[4] = t->h4c + t->Kp4;                    // trunk layer 4 writes behind the position features: [pos_map | h4] is the skip's concat (model.py:108)
    {   // the packed images: where each lies and what it is gathered from
        PackArgs &pa = t->pack;
        long long first = 0;
        auto job = [&](const float *src, long long sk, long long sc, int K1, int K1p, int K2, const float *src2, long long sk2, long long sc2, int n_out, const float **recs,
                       int *kblocks) {
            PackJob &j = pa.job[pa.n_jobs++];
            j.src = src; j.sk = sk; j.sc = sc; j.src2 = src2; j.sk2 = sk2; j.sc2 = sc2; j.K1 = K1; j.K1p = K1p; j.K2 = K2; j.nt = n_out / 32;
            j.kblocks = ((K1p + K2 + 7) / 8 + 3) / 4 * 4;
            j.first = first; j.dst = nullptr;
            *kblocks = j.kblocks;
            *recs = (const float *)(uintptr_t)first;          // an offset until the buffer exists
            first += (long long)j.kblocks * 8 * n_out;
        };
        auto fwd = [&](const TLayer &l, int K1, int K1p, int K2, int slot) {
            job(t->w + l.w, l.out, 1, K1, K1p, K2, t->w + l.w + (size_t)K1 * l.out, l.out, 1, l.out, &t->fwd_recs[slot], &t->fwd_kblocks[slot]);
        };
        auto bwd = [&](const TLayer &l, int row0, int slot) {
            job(t->w + l.w + (size_t)row0 * l.out, 1, l.out, l.out, l.out, 0, nullptr, 0, 0, 256, &t->bwd_recs[slot], &t->bwd_kblocks[slot]);
        };
        for (int i = 0; i < 8; ++i) {
            if (i == 5) fwd(t->trunk[i], t->Kp, t->Kp4, 256, i); else fwd(t->trunk[i], t->trunk[i].in, t->trunk[i].in, 0, i);
        }
        fwd(t->feature, 256, 256, 0, 8); fwd(t->c1, t->Kd, t->Kd4, 256, 9); fwd(t->c2, 256, 256, 0, 10);
        for (int i = 1; i < 8; ++i) bwd(t->trunk[i], i == 5 ? t->Kp : 0, i - 1);      // the skip's position rows take no gradient further
        bwd(t->c1, t->Kd, 8); bwd(t->c2, 0, 9);
        // d h7 takes two gradients on: d feature . W_feature^T and d_sigma (x) W_alpha (model.py:111-115).  One contraction: the 1-wide head's
        // weights are row 256 of the image, d_sigma column 256 of the buffer d feature is written to (row stride LDGF)
        job(t->w + t->feature.w, 1, 256, 256, 256, 1, t->w + t->alpha.w, 0, 1, 256, &t->bwd_recs[7], &t->bwd_kblocks[7]);
        pa.total = first;
        if (alloc(&t->wp, (size_t)first) != NTX_OK) { free_all(t); return NTX_E_HIP; }
        for (int j = 0; j < pa.n_jobs; ++j) pa.job[j].dst = t->wp + pa.job[j].first;
        for (int i = 0; i < 11; ++i) t->fwd_recs[i] = t->wp + (uintptr_t)t->fwd_recs[i];
        for (int i = 0; i < 10; ++i) t->bwd_recs[i] = t->wp + (uintptr_t)t->bwd_recs[i];
    }
    *out = t;
    return NTX_OK;
}

int ntx_trainer_destroy(ntx_trainer *t) { free_all(t); return NTX_OK; }

size_t ntx_trainer_weight_count(const ntx_trainer *t) { return t ? t->n_weights : 0; }

int ntx_trainer_get(ntx_trainer *t, int what, float *out_host, size_t n_floats) {
    if (!t || !out_host) return ntx_set_error(NTX_E_INVALID, "NULL argument");
    if (n_floats != t->n_weights) return ntx_set_error(NTX_E_INVALID, "%zu floats asked, the model has %zu", n_floats, t->n_weights);
    const float *src = what == NTX_TRAINER_WEIGHTS ? t->w : what == NTX_TRAINER_GRADIENTS ? t->grad : what == NTX_TRAINER_ADAM_M ? t->adam_m : what == NTX_TRAINER_ADAM_V ? t->adam_v : nullptr;
    if (!src) return ntx_set_error(NTX_E_INVALID, "what = %d", what);
    TRAIN_TRY(hipSetDevice(t->device));
    TRAIN_TRY(hipDeviceSynchronize());
    TRAIN_TRY(hipMemcpy(out_host, src, n_floats * sizeof(float), hipMemcpyDeviceToHost));
    return NTX_OK;
}

int ntx_trainer_activation(ntx_trainer *t, int layer, int64_t n_samples_total, float *out_host) {
    if (!t || !out_host) return ntx_set_error(NTX_E_INVALID, "NULL argument");
    if (n_samples_total < 1 || n_samples_total > t->cap) return ntx_set_error(NTX_E_INVALID, "n_samples_total out of range");
    const float *src = nullptr; int ld = 256, width = 256;
    if (layer >= 0 && layer < 8) { src = t->h[layer]; ld = layer == 4 ? t->ldp : 256; }
    else if (layer == 8) src = t->c1o;
    else if (layer == 9) { src = t->c2o; ld = 128; width = 128; }
    else if (layer == 10) { src = t->sigma; ld = 1; width = 1; }
    else if (layer >= 20 && layer < 28) src = t->dyt[layer - 20];
    else if (layer == 28) src = t->g1;
    else if (layer == 29) { src = t->gf; ld = LDGF; }
    else return ntx_set_error(NTX_E_INVALID, "layer %d (0-7 trunk, 8 / 9 the colour layers, 10 the density; 20-29 the kept gradients)", layer);
    TRAIN_TRY(hipSetDevice(t->device));
    TRAIN_TRY(hipDeviceSynchronize());
    TRAIN_TRY(hipMemcpy2D(out_host, (size_t)width * sizeof(float), src, (size_t)ld * sizeof(float), (size_t)width * sizeof(float), (size_t)n_samples_total, hipMemcpyDeviceToHost));
    return NTX_OK;
}

int ntx_trainer_set_weights(ntx_trainer *t, const float *weights_host, size_t n_floats) {
    if (!t || !weights_host) return ntx_set_error(NTX_E_INVALID, "NULL argument");
    if (n_floats != t->n_weights) return ntx_set_error(NTX_E_INVALID, "%zu floats given, the model has %zu", n_floats, t->n_weights);
    TRAIN_TRY(hipSetDevice(t->device));
    TRAIN_TRY(hipMemcpy(t->w, weights_host, n_floats * sizeof(float), hipMemcpyHostToDevice));
    return NTX_OK;
}

int ntx_trainer_set(ntx_trainer *t, int what, const float *values_host, size_t n_floats) {
    if (!t || !values_host) return ntx_set_error(NTX_E_INVALID, "NULL argument");
    if (n_floats != t->n_weights) return ntx_set_error(NTX_E_INVALID, "%zu floats given, the model has %zu", n_floats, t->n_weights);
    float *dst = what == NTX_TRAINER_WEIGHTS ? t->w : what == NTX_TRAINER_GRADIENTS ? t->grad : what == NTX_TRAINER_ADAM_M ? t->adam_m : what == NTX_TRAINER_ADAM_V ? t->adam_v : nullptr;
    if (!dst) return ntx_set_error(NTX_E_INVALID, "what = %d", what);
    TRAIN_TRY(hipSetDevice(t->device));
    TRAIN_TRY(hipDeviceSynchronize());
    TRAIN_TRY(hipMemcpy(dst, values_host, n_floats * sizeof(float), hipMemcpyHostToDevice));
    return NTX_OK;
}

int ntx_trainer_allreduce_gradients(ntx_trainer *t, ntx_comm *comm, ntx_stream stream) {
    if (!t || !comm) return ntx_set_error(NTX_E_INVALID, "NULL argument");
    return ntx_allreduce_mean_f32(comm, t->grad, t->n_weights, stream);
}

int ntx_train_step_gradients(ntx_trainer *t, const float *rays_o, const float *rays_d, const float *tnear_far, const float *params, int64_t rays_per_param_row,
                             const float *cone_scale, int64_t n_rays, int n_samples, int blur_idx, uint32_t flags, const float *bkgd, uint64_t perturb_seed,
                             const ntx_render_opts *opts, const float *z_vals, const float *color_true, const float *alpha_true, const ntx_loss_desc *loss,
                             float *color_pred, float *alpha_pred, float *loss_out, ntx_stream stream) {
    if (!t) return ntx_set_error(NTX_E_INVALID, "trainer is NULL");
    if (!rays_o || !rays_d || (!tnear_far && !z_vals) || !color_true || !loss || (t->P > 0 && !params)) return ntx_set_error(NTX_E_INVALID, "NULL buffer");
    if (n_rays < 1 || n_rays > t->cap_rays || n_samples < 2 || (long long)n_rays * n_samples > t->cap) return ntx_set_error(NTX_E_INVALID, "n_rays x n_samples beyond what the trainer was created for");
    if (n_samples > MAX_TRAIN_SAMPLES) return ntx_set_error(NTX_E_INVALID, "n_samples > %d", MAX_TRAIN_SAMPLES);
    if (blur_idx >= t->P || (blur_idx >= 0 && !cone_scale)) return ntx_set_error(NTX_E_INVALID, "bad blur_idx / cone_scale");
    if (loss->size < sizeof(ntx_loss_desc) || (loss->kind != NTX_LOSS_NERF && loss->kind != NTX_LOSS_ALPHA) || (loss->loss_fn != NTX_LOSS_MSE && loss->loss_fn != NTX_LOSS_SMAPE) ||
        (loss->alpha_loss_fn != NTX_LOSS_MSE && loss->alpha_loss_fn != NTX_LOSS_SMAPE))
        return ntx_set_error(NTX_E_INVALID, "bad ntx_loss_desc");
    if (loss->kind == NTX_LOSS_ALPHA && !alpha_true) return ntx_set_error(NTX_E_INVALID, "AlphaLoss needs alpha_true");
    if (rays_per_param_row < 1) rays_per_param_row = 1;
    TRAIN_TRY(hipSetDevice(t->device));
    hipStream_t st = (hipStream_t)stream;
    const long long M = (long long)n_rays * n_samples;
    const int S = n_samples, Kp = t->Kp, Kd = t->Kd, ldp = t->ldp, ldd = t->ldd;
    const float *W = t->w;
    hipLaunchKernelGGL(pack_records_kernel, dim3((unsigned)((t->pack.total + 255) / 256)), dim3(256), 0, st, t->pack);
    // ---- forward, every activation kept ----------------------------------------------------------------------------------------
    const float *z = z_vals;
    if (!z) {
        int rc = ntx_sample_depths(tnear_far, n_rays, S, flags & NTX_FLAG_PERTURB, perturb_seed, opts, t->z, stream);     // renderer.py:101-111
        if (rc != NTX_OK) return rc;
        z = t->z;
    }
    const float *noise = nullptr;
    if (flags & NTX_FLAG_RAW_NOISE) {                                                           // renderer.py:190-192
        int rc = ntx_sample_noise(n_rays, S, perturb_seed, opts, t->noise, stream);
        if (rc != NTX_OK) return rc;
        noise = t->noise;
    }
    {
        EncodeArgs e{}; e.rays_o = rays_o; e.rays_d = rays_d; e.z = z; e.params = params; e.cone = cone_scale; e.rays_per_param_row = rays_per_param_row;
        e.n_rays = (int)n_rays; e.S = S; e.n_geo = t->desc.n_geo; e.n_app = t->desc.n_app; e.pos_freq = t->desc.pos_freq; e.dir_freq = t->desc.dir_freq;
        e.param_freq = t->desc.param_freq; e.blur_idx = blur_idx; e.pos_out = t->h4c; e.ld_pos = ldp; e.dir_out = t->fc; e.ld_dir = ldd; e.dists = t->dists;
        hipLaunchKernelGGL(encode_kernel, dim3((unsigned)((M + ENC_SAMPLES - 1) / ENC_SAMPLES)), dim3(256), (size_t)ENC_SAMPLES * (ENC_BASE + Kp + Kd) * sizeof(float), st, e);
    }
    {   // the trunk, the feature layer and the first colour layer as ONE chain (model.py:104-119)
        RowsArgs fa{}; fa.M = M;
        for (int i = 0; i < 8; ++i) {
            const TLayer &l = t->trunk[i];
            const float *X = (i == 0 || i == 5) ? t->h4c : t->h[i - 1];
            const int ldx = (i == 0 || i == 5) ? ldp : (i - 1 == 4 ? ldp : 256);
            fa.layer[fa.n_layers++] = forward_layer(X, ldx, t->fwd_recs[i], t->fwd_kblocks[i], W + l.b, t->h[i], i == 4 ? ldp : 256, 1, t->bits[i]);
        }
        fa.layer[fa.n_layers++] = forward_layer(t->h[7], 256, t->fwd_recs[8], t->fwd_kblocks[8], W + t->feature.b, t->fc + t->Kd4, ldd, 0, nullptr);    // :114-115
        fa.layer[fa.n_layers++] = forward_layer(t->fc, ldd, t->fwd_recs[9], t->fwd_kblocks[9], W + t->c1.b, t->c1o, 256, 1, t->bits[8]);               // :118-119
        launch_rows(st, fa, 256, true);
    }
    hipLaunchKernelGGL(head_forward_kernel, dim3((unsigned)((M + 15) / 16)), dim3(256), 0, st, t->h[7], 256, 256, W + t->alpha.w, W + t->alpha.b, 1, M, t->sigma);   // :111
    {
        RowsArgs ca{}; ca.M = M; ca.n_layers = 1;
        ca.layer[0] = forward_layer(t->c1o, 256, t->fwd_recs[10], t->fwd_kblocks[10], W + t->c2.b, t->c2o, 128, 1, nullptr);                         // :122
        launch_rows(st, ca, 128, true);
    }
    hipLaunchKernelGGL(head_forward_kernel, dim3((unsigned)((M + 31) / 32)), dim3(256), 0, st, t->c2o, 128, 128, W + t->rgb.w, W + t->rgb.b, 3, M, t->raw_rgb);   // :123
    CompositeArgs c{};
    c.raw_rgb = t->raw_rgb; c.sigma = t->sigma; c.dists = t->dists; c.noise = noise; c.n_rays = (int)n_rays; c.S = S; c.map_exr = (flags & NTX_FLAG_MAP_EXR) ? 1 : 0;
    c.composite_bkgd = (flags & NTX_FLAG_COMPOSITE_BKGD) ? 1 : 0;
    for (int k = 0; k < 3; ++k) c.bkgd[k] = bkgd ? bkgd[k] : 1.0f;
    c.color = t->color; c.alpha = t->alpha_out; c.d_color = t->d_color; c.d_alpha = t->d_alpha; c.d_raw_rgb = t->d_raw; c.d_sigma = t->d_sigma;
    hipLaunchKernelGGL(composite_kernel<false>, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, st, c);
    // ---- loss (loss.py) -----------------------------------------------------------------------------------------------------------
    {
        LossArgs a{}; a.color_true = color_true; a.alpha_true = alpha_true; a.color_pred = t->color; a.alpha_pred = t->alpha_out; a.n_rays = (int)n_rays;
        a.kind = loss->kind; a.loss_fn = loss->loss_fn; a.alpha_loss_fn = loss->alpha_loss_fn; a.filter_color_loss = loss->filter_color_loss; a.use_hard_mask = loss->use_hard_mask;
        a.gamma = loss->gamma; a.loss = t->loss; a.d_color = t->d_color; a.d_alpha = t->d_alpha;
        hipLaunchKernelGGL(loss_kernel, dim3(1), dim3(1024), 0, st, a);
    }
    // ---- backward -----------------------------------------------------------------------------------------------------------------
    hipLaunchKernelGGL(composite_kernel<true>, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, st, c);
    float *G = t->grad;
    // The gradient runs down the network first (every layer's dY kept), then ALL weight gradients are taken in one launch: dW = X^T . dY and
    // db = the column sums of dY (riding along in the same kernel), both through partial sums added up in a fixed order by one more launch;
    // kernel [K][N] and bias [N] are neighbours in the blob
    GemmBatch gb{}; ReduceBatch rb{};
    size_t partial_used = 0; long long reduce_first = 0; int wg_first = 0;
    const int tk = 16;
    const int k_chunk = (((int)M + SPLIT - 1) / SPLIT + tk - 1) / tk * tk, parts = ((int)M + k_chunk - 1) / k_chunk;
    auto dw = [&](const float *X, int ldx, int K, const float *dY, int lddy, int N, float *dW, float *db) -> int {
        if (gb.n >= MAX_GEMM_BATCH || rb.n + 2 > MAX_REDUCE_BATCH) return ntx_set_error(NTX_E_INVALID, "trainer: too many weight gradients for one launch");
        const size_t need = (size_t)parts * K * N + (db ? (size_t)parts * N : 0);
        if (partial_used + need > t->dw_partial_floats) return ntx_set_error(NTX_E_INVALID, "trainer: partial buffer too small");
        float *pw = t->dw_partial + partial_used, *pb = pw + (size_t)parts * K * N;
        partial_used += need;
        GemmArgs &g = gb.g[gb.n];
        g = GemmArgs{}; g.A = X; g.lda = ldx; g.B = dY; g.ldb = lddy; g.C = pw; g.ldc = N; g.M = K; g.N = N; g.K = (int)M; g.split_stride = (long long)K * N;
        g.colsum = db ? pb : nullptr; g.k_chunk = k_chunk;
        g.aligned = (g.lda % 4 == 0) && (g.ldb % 4 == 0) && (((uintptr_t)g.A | (uintptr_t)g.B) % 16 == 0);
        gb.nx[gb.n] = (N + 127) / 128; gb.ny[gb.n] = (K + TM - 1) / TM; gb.nz[gb.n] = parts; gb.first[gb.n] = wg_first;
        wg_first += (gb.nx[gb.n] * gb.ny[gb.n] * parts + 7) / 8 * 8;
        gb.n += 1; gb.first[gb.n] = wg_first;
        auto red = [&](const float *partial, long long count, float *out) {
            ReduceJob &r = rb.job[rb.n++];
            r.partial = partial; r.n_split = parts; r.stride = count; r.count = count; r.out = out; r.first = reduce_first;
            reduce_first += (count + 255) / 256 * 256;
        };
        red(pw, (long long)K * N, dW);
        if (db) red(pb, N, db);
        return NTX_OK;
    };
    // a layer that reads a concat buffer [first | pad | 256 more]: two contractions when there is a pad, the bias gradient rides with the first
    auto concat_dw = [&](const float *X, int ldx, int K1, int K1p, const float *dY, const TLayer &l) -> int {
        if (K1 == K1p) return dw(X, ldx, K1 + 256, dY, 256, 256, G + l.w, G + l.b);
        const int rc1 = dw(X, ldx, K1, dY, 256, 256, G + l.w, G + l.b);
        if (rc1 != NTX_OK) return rc1;
        return dw(X + K1p, ldx, 256, dY, 256, 256, G + l.w + (size_t)K1 * 256, nullptr);
    };
    const int hb = (int)((M + HEAD_ROWS - 1) / HEAD_ROWS);
    auto head_dw = [&](const float *X, int ldx, int K, const float *dY, int n_out, const TLayer &l) {       // (kernel | bias) of a narrow head
        hipLaunchKernelGGL(head_backward_dw_partial_kernel, dim3(hb), dim3(256), 0, st, X, ldx, K, dY, n_out, M, HEAD_ROWS, t->partial);
        const long long count = (long long)(K + 1) * n_out;
        hipLaunchKernelGGL(reduce_partials_wide_kernel, dim3((unsigned)((count + 31) / 32)), dim3(256), 0, st, t->partial, hb, count, count, G + l.w);
    };
    // color head (128 -> 3): dW, db; d c2o = (d_raw . W^T) where c2o > 0
    head_dw(t->c2o, 128, 128, t->d_raw, 3, t->rgb);
    hipLaunchKernelGGL(head_backward_dx_kernel, dim3((unsigned)((M * 32 + 255) / 256)), dim3(256), 0, st, t->d_raw, 3, W + t->rgb.w, 128, M, t->c2o, 128, 0, t->g0, 128);
    head_dw(t->h[7], 256, 256, t->d_sigma, 1, t->alpha);
    // d h7 = (d feature . W_feature^T + d_sigma (x) W_alpha) where h7 > 0, as ONE contraction over 257: d_sigma goes beside d feature
    hipLaunchKernelGGL(column_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, t->d_sigma, M, t->gf + 256, LDGF);
    {   // the way back, one chain: d c1o (masked by its ReLU), d feature (a linear layer: all kept), d h7, ... d h0
        RowsArgs ba{}; ba.M = M;
        ba.layer[ba.n_layers++] = dx_layer(t->g0, 128, t->bwd_recs[9], t->bwd_kblocks[9], t->bits[8], t->g1, 256);
        ba.layer[ba.n_layers++] = dx_layer(t->g1, 256, t->bwd_recs[8], t->bwd_kblocks[8], t->bits_ones, t->gf, LDGF);
        ba.layer[ba.n_layers++] = dx_layer(t->gf, LDGF, t->bwd_recs[7], t->bwd_kblocks[7], t->bits[7], t->dyt[7], 256);
        for (int i = 7; i >= 1; --i) ba.layer[ba.n_layers++] = dx_layer(t->dyt[i], 256, t->bwd_recs[i - 1], t->bwd_kblocks[i - 1], t->bits[i - 1], t->dyt[i - 1], 256);
        launch_rows(st, ba, 256, false);
    }
    int rc = dw(t->c1o, 256, 256, t->g0, 128, 128, G + t->c2.w, G + t->c2.b);
    if (rc == NTX_OK) rc = concat_dw(t->fc, ldd, Kd, t->Kd4, t->g1, t->c1);
    if (rc == NTX_OK) rc = dw(t->h[7], 256, 256, t->gf, LDGF, 256, G + t->feature.w, G + t->feature.b);
    for (int i = 7; i >= 0 && rc == NTX_OK; --i) {
        const TLayer &l = t->trunk[i];
        const float *X = (i == 0 || i == 5) ? t->h4c : t->h[i - 1];
        const int ldx = (i == 0 || i == 5) ? ldp : 256;
        rc = i == 5 ? concat_dw(X, ldx, Kp, t->Kp4, t->dyt[i], l) : dw(X, ldx, l.in, t->dyt[i], 256, 256, G + l.w, G + l.b);
    }
    if (rc != NTX_OK) return rc;
    hipLaunchKernelGGL((gemm_batch_kernel<false, 128, 16>), dim3((unsigned)wg_first), dim3(256), 0, st, gb);
    hipLaunchKernelGGL(reduce_batch_kernel, dim3((unsigned)(reduce_first / 256)), dim3(256), 0, st, rb);
    if (color_pred) TRAIN_TRY(hipMemcpyAsync(color_pred, t->color, (size_t)n_rays * 3 * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (alpha_pred) TRAIN_TRY(hipMemcpyAsync(alpha_pred, t->alpha_out, (size_t)n_rays * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (loss_out) TRAIN_TRY(hipMemcpyAsync(loss_out, t->loss, sizeof(float), hipMemcpyDeviceToDevice, st));
    TRAIN_TRY(hipGetLastError());
    return NTX_OK;
}

int ntx_trainer_adam_step(ntx_trainer *t, float lrate, float lrate_decay_steps, float lrate_decay_rate, float beta_1, float beta_2, float epsilon, ntx_stream stream) {
    if (!t) return ntx_set_error(NTX_E_INVALID, "trainer is NULL");
    TRAIN_TRY(hipSetDevice(t->device));
    const double step = (double)t->adam_iterations;
    double lr = lrate;
    if (lrate_decay_steps > 0) lr = (double)lrate * std::pow((double)lrate_decay_rate, step / (double)lrate_decay_steps);      // ExponentialDecay, staircase off
    const double tt = step + 1.0;
    const float lr_t = (float)((double)(float)lr * std::sqrt(1.0 - std::pow((double)beta_2, tt)) / (1.0 - std::pow((double)beta_1, tt)));
    hipLaunchKernelGGL(ntx_train::adam_kernel, dim3((unsigned)((t->n_weights + 255) / 256)), dim3(256), 0, (hipStream_t)stream, t->w, t->grad, t->adam_m, t->adam_v,
                       (long long)t->n_weights, lr_t, beta_1, beta_2, epsilon);
    t->adam_iterations += 1;
    TRAIN_TRY(hipGetLastError());
    return NTX_OK;
}

int64_t ntx_trainer_iterations(const ntx_trainer *t) { return t ? t->adam_iterations : -1; }

/* The contraction the trainer is made of, on caller buffers (DEVICE): C[M][N] = op(A) . op(B) (+ bias) (ReLU), op = identity or transpose as
 * a_kcontig / b_kcontig say (see gemm_kernel).  For tests and benches of the kernel itself. */
int ntx_gemm_f32(const float *A, int lda, int a_kcontig, const float *B, int ldb, int b_kcontig, float *C, int ldc, int M, int N, int K, const float *bias, int relu,
                 ntx_stream stream) {
    if (!A || !B || !C || M < 1 || N < 1 || K < 1) return ntx_set_error(NTX_E_INVALID, "bad GEMM arguments");
    ntx_train::GemmArgs g{}; g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.bias = bias; g.relu = relu;
    hipStream_t st = (hipStream_t)stream;
    if (b_kcontig) return ntx_set_error(NTX_E_UNSUPPORTED, "B must be [K][N] (the trainer transposes its weights once a step instead)");
    if (a_kcontig) launch_gemm<true>(st, g); else launch_gemm<false>(st, g);
    TRAIN_TRY(hipGetLastError());
    return NTX_OK;
}

}   // extern "C"
