// ntx_train.hip -- one training step of the reference on the GPU: network/train.py:49-70 (GradientTape over Renderer.__call__, a loss of
// network/loss.py:6-59, Adam under ExponentialDecay) for the ParamNerf architecture of the shipped training configs (8 x 256, skip 4,
// color_depth 1; configs/config_carpet_train.py: 4 images x 256 rays x 256 samples = 262 144 samples a step).  gfx950 only.
//
// Inference fuses the whole network into one kernel because nothing of it has to survive (ntx_device.h).  A training step has to keep
// every layer's activations for the weight gradients; with 288 GB of HBM they are simply stored, once each, and the step is three passes on
// the f32 matrix cores (ntx_train_device.h) between a handful of small kernels:
//
//   pack_kernel            the weights move every step: the forward and the transposed weight streams and the aux block (biases, narrow heads)
//   encode_kernel          sample points, positional encodings of position / direction / parameters (layer.py:8-23): the first layer's and the
//                          two concatenations' inputs, in the row order the chain reads and in the operand order the weight gradients read
//   fwd_chain_kernel       the network forward with the activations of a block of 32 samples in registers from layer to layer, stored once
//   composite_loss_kernel  renderer.py:170-213 per ray, the ray's term of the loss (loss.py) and the adjoint of both (wave per ray)
//   dx_chain_kernel        the gradient back through the layers, masked by the forward pass's ReLU bits, every layer's stored once
//   dw_kernel              dW = X^T . dY of every layer in one launch, partial sums over ranges of samples; reduce_batch_kernel adds them in a
//                          fixed order (a step is bit-reproducible)
//   adam_kernel            tf.keras.optimizers.Adam under ExponentialDecay (train.py:49-52), one fused pass over the 2.7 MB of weights
//
// Everything is float32 with float32 accumulation, like the reference's TensorFlow graph.
#include "nerftex.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "ntx_train_device.h"

extern "C" int ntx_set_error(int code, const char *fmt, ...);   // nerftex.hip

#define TRAIN_TRY(expr)                                                                                  \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) return ntx_set_error(NTX_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

namespace ntx_train {

// ---------------------------------------------------------------------------------------------------------------------------
// the general contraction on caller buffers (ntx_gemm_f32; the step itself runs on the kernels of ntx_train_device.h).
//   C[i][j] = sum_p A'(i, p) B(p, j) for i < M, j < N, p < K, with B[p * ldb + j] and
//   A'(i, p) = A_KCONTIG ? A[i * lda + p] : A[p * lda + i]
// A workgroup (4 waves) owns a 128 x 128 tile of C, a wave a
// 64 x 64 quarter of it = 2 x 2 MFMA tiles of 32 x 32 (64 accumulator registers).  K advances a panel (16) at a time: the next
// 128 x 16 / 16 x 128 panels are fetched into registers (16-byte loads when the panel lies inside the matrices and the rows are 16-byte
// aligned, element by element with bounds otherwise) while the current ones, already in LDS as As[p][i] / Bs[p][j], feed the MFMAs; one
// barrier per panel (double buffered).  The f32 MFMA shares the vector ALUs' lanes, so every VALU instruction of the loop
// costs MFMA time: hence the vector loads and the branch-free interior path.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int TM = 128;

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

// out[e] = sum_z partial[z][e] (+ the second half of a bias's pair), z ascending: the fixed order that makes a step reproducible.  Several
// results in one launch; a job's elements are numbered from first (multiples of 256: a block belongs to one job)
constexpr int MAX_REDUCE_BATCH = 32;
struct ReduceJob { const float *partial; int n_split; long long stride, count, pair; float *out; long long first; };   // pair > 0: element e of a split is partial[e] + partial[pair + e]
struct ReduceBatch { ReduceJob job[MAX_REDUCE_BATCH]; int n; };
__global__ void reduce_batch_kernel(ReduceBatch b) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int j = 0;
    while (j + 1 < b.n && g >= b.job[j + 1].first) ++j;
    const ReduceJob &r = b.job[j];
    const long long e = g - r.first;
    if (e >= r.count) return;
    // four running sums over z = 0, 4, 8 ... / 1, 5, ... / ..., combined at the end: a fixed order, four loads in flight
    float s4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    auto at = [&](int z) { const float *p = r.partial + (size_t)z * r.stride + e; return r.pair > 0 ? p[0] + p[r.pair] : p[0]; };
    int z = 0;
    for (; z + 4 <= r.n_split; z += 4) {
#pragma unroll
        for (int q = 0; q < 4; ++q) s4[q] += at(z + q);
    }
    for (int q = 0; z < r.n_split; ++z, ++q) s4[q] += at(z);
    r.out[e] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
}

// ---------------------------------------------------------------------------------------------------------------------------
// the weights as the chains stream them (ntx_train_device.h), made once a step.  A segment of a stream: records (k-step s, tile group g) of
// 64 lanes x 4 floats, component c of lane (f, kh) = Wsrc[row(s, kh)][32 (4 g + c) + f] with Wsrc[k][col] = src[k * sk + col * sc]; rows
// beyond K (padding k-steps, the odd half of a last k-step) and columns beyond ncols are zero.  The aux block's pieces ride along.
// ---------------------------------------------------------------------------------------------------------------------------
enum { PACK_HIDDEN = 0, PACK_LINEAR = 1, PACK_AUX_ROW = 2, PACK_AUX_RGB = 3, PACK_COPY = 4 };
struct PackSeg {
    const float *src; long long sk, sc;
    int mode;                 // PACK_HIDDEN: row(s, kh) = hidden_row(s, kh) (the k-steps of a layer whose input is a lane's registers); PACK_LINEAR: 2 s + kh
    int K, ncols, nt;         // rows / columns that exist; tiles of the layer (4 or 8)
    float *dst; long long first, count;   // where it goes; the segment's first float in the launch's index space and how many
};
struct PackArgs { const PackSeg *seg; int n_seg; long long total; };
__global__ void pack_kernel(PackArgs a) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= a.total) return;
    int lo = 0, hi = a.n_seg - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (a.seg[mid].first <= e) lo = mid; else hi = mid - 1; }
    const PackSeg &p = a.seg[lo];
    const long long o = e - p.first;
    float v = 0.0f;
    if (p.mode == PACK_COPY) v = p.src[o];
    else if (p.mode == PACK_AUX_ROW) {                   // [half][128]: value V of half h <-> feature hidden_row(V, h)
        const int h = (int)(o >> 7) & 1, V = (int)(o & 127), k = hidden_row(V, h);
        v = k < p.K ? p.src[k * p.sk] : 0.0f;
    } else if (p.mode == PACK_AUX_RGB) {                 // [3][half][64]
        const int c = (int)(o >> 7), h = (int)(o >> 6) & 1, V = (int)(o & 63);
        v = p.src[hidden_row(V, h) * 3 + c];
    } else {
        const int c = (int)(o & 3), lane = (int)((o >> 2) & 63);
        const long long rec = o >> 8;
        const int G = p.nt / 4, g = (int)(rec % G), s = (int)(rec / G);
        const int f = lane & 31, kh = lane >> 5;
        const int k = p.mode == PACK_HIDDEN ? hidden_row(s, kh) : 2 * s + kh, col = 32 * (4 * g + c) + f;
        if (k < p.K && col < p.ncols) v = p.src[k * p.sk + col * p.sc];
    }
    p.dst[o] = v;
}

// ---------------------------------------------------------------------------------------------------------------------------
// encoder: layer.FourierFeatures (layer.py:8-23) of position [+ geometry parameters] and of direction [+ appearance parameters]
// (model.py:77-101), the sample points of renderer.py:98-114 and the blur product of :155-158.  One wave per block of 32 samples and map:
// lane (n, h) evaluates sin (h = 0) or cos (h = 1) of its sample with ONE function (ntx_device.h sin_q, the render kernels' own) and writes
// rows of 32 samples in O layout: the weight gradients' A operands, and what the chain gathers its B operands from.
// ---------------------------------------------------------------------------------------------------------------------------
struct EncodeArgs {
    const float *rays_o, *rays_d, *z, *params, *cone;
    long long rays_per_param_row, M;
    int n_rays, S, n_geo, n_app, pos_freq, dir_freq, param_freq, blur_idx;
    float *posO; int ptiles;             // rows 0 .. Kp: pos_map; the rest of the ptiles * 32 rows stays zero
    float *dirO; int dtiles;
    float *dists;                        // [N][S]: z[i+1] - z[i], the last one repeated, times |rays_d| (renderer.py:174-180)
};
__global__ __launch_bounds__(64) void encode_kernel(EncodeArgs a) {
    const int lane = threadIdx.x, n = lane & 31, h = lane >> 5, blk = blockIdx.x, part = blockIdx.y;
    const long long m = (long long)blk * 32 + n;
    const bool valid = m < a.M;
    const int ray = valid ? (int)(m / a.S) : 0, s = valid ? (int)(m - (long long)ray * a.S) : 0;
    const float d[3] = {a.rays_d[3 * ray], a.rays_d[3 * ray + 1], a.rays_d[3 * ray + 2]};
    const float dn = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
    // A ray that misses the proxy (t = inf: the reference's Renderer.__call__ filters it out and scatters 0 / the background back,
    // renderer.py:58-86) stays in the batch with depth 0 and distances 0: every alpha of it is 1 - exp(-sigma 0) = 0, so it composites to
    // exactly 0 / the background, no gradient flows into or out of its rows, and the loss still counts it among its rays
    const float zr = a.z[(size_t)ray * a.S + s];
    const bool hit = isfinite(zr);
    const float z = hit ? zr : 0.0f;
    const int P = a.n_geo + a.n_app;
    const float *pr = a.params + (size_t)(ray / a.rays_per_param_row) * (P > 0 ? P : 1);
    auto param = [&](int c) { return c == a.blur_idx ? (hit ? pr[c] * (a.cone[ray] * z) : 0.0f) : pr[c]; };   // :155-158 (a missing ray's cone scale may be anything)
    float *O = part == 0 ? a.posO : a.dirO;
    const int tiles = part == 0 ? a.ptiles : a.dtiles;
    auto put = [&](int row, float v) {
        if (!valid) v = 0.0f;                                                            // the tail of the last block: finite, and no gradient comes back
        O[(((size_t)blk * tiles + (row >> 5)) * 4 + (n >> 3)) * 256 + ((row & 31) + 32 * ((n >> 2) & 1)) * 4 + (n & 3)] = v;
    };
    // FourierFeatures of x[0 .. D) with L bands from row r0 on: [x | sin(2^0 x) | cos(2^0 x) | sin(2^1 x) | ...], every block D wide (layer.py:14-23)
    auto fourier = [&](int r0, int D, int L, auto x) {
        if (h == 0) for (int c = 0; c < D; ++c) put(r0 + c, x(c));
        for (int f = 0; f < L; ++f)
            for (int c = 0; c < D; ++c) put(r0 + D + 2 * D * f + h * D + c, ntx::sin_q(ldexpf(1.0f, f) * x(c), h));
    };
    if (part == 0) {
        const float o[3] = {a.rays_o[3 * ray], a.rays_o[3 * ray + 1], a.rays_o[3 * ray + 2]};
        fourier(0, 3, a.pos_freq, [&](int c) { return o[c] + d[c] * z; });                  // renderer.py:114 (un-normalised rays_d)
        if (a.n_geo > 0) fourier(3 * (1 + 2 * a.pos_freq), a.n_geo, a.param_freq, [&](int c) { return param(c); });      // model.py:88-93
    } else {
        fourier(0, 3, a.dir_freq, [&](int c) { return d[c] / dn; });                        // renderer.py:98
        if (a.n_app > 0) fourier(3 * (1 + 2 * a.dir_freq), a.n_app, a.param_freq, [&](int c) { return param(a.n_geo + c); });   // model.py:96-101
        if (valid && h == 0) {
            const float zn = s + 1 < a.S ? a.z[(size_t)ray * a.S + s + 1] : 0.0f;
            const float dist = s + 1 < a.S ? zn - z : (a.S > 1 ? z - a.z[(size_t)ray * a.S + s - 1] : 0.0f);
            a.dists[(size_t)ray * a.S + s] = hit ? dist * dn : 0.0f;
        }
    }
}

// The colour layer's direction segment once per ray (fwd_chain_kernel's HOIST builds): row[f] = bias_C1[f] + sum_k dir_map[k] W_C1[k][f] over
// the Kd rows of dir_map = FourierFeatures(direction) | FourierFeatures(appearance parameters) (model.py:96-101, 115), written in the
// accumulators' order [ray][half h][16 T + 4 g + c] for feature 32 T + 8 g + 4 h + c.  Workgroup per ray, thread per output feature.
struct DirRowArgs {
    const float *rays_d, *params; long long rays_per_param_row;
    int n_rays, n_geo, n_app, dir_freq, param_freq, Kd;
    const float *w, *bias;                     // W_C1 [Kd + 256][256] (its first Kd rows), bias_C1 [256]
    float *rows;
};
__global__ __launch_bounds__(256) void dirrow_kernel(DirRowArgs a) {
    __shared__ float feat[8 * MAX_PB_GROUPS];
    const int ray = blockIdx.x, f = threadIdx.x;
    const float d[3] = {a.rays_d[3 * ray], a.rays_d[3 * ray + 1], a.rays_d[3 * ray + 2]};
    const float dn = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
    const int P = a.n_geo + a.n_app;
    const float *pr = a.params + (size_t)(ray / a.rays_per_param_row) * (P > 0 ? P : 1);
    if (f < a.Kd) {                                                                             // row f of dir_map, as encode_kernel lays it out
        const int K3 = 3 * (1 + 2 * a.dir_freq);
        const int r = f < K3 ? f : f - K3, D = f < K3 ? 3 : a.n_app;
        auto x = [&](int c) { return f < K3 ? d[c] / dn : pr[a.n_geo + c]; };                  // renderer.py:98; model.py:96-101
        float v;
        if (r < D) v = x(r);
        else { const int q = r - D, band = q / (2 * D), hc = q - band * 2 * D, hh = hc / D, c = hc - hh * D; v = ntx::sin_q(ldexpf(1.0f, band) * x(c), hh); }
        feat[f] = v;
    }
    __syncthreads();
    float acc = a.bias[f];
    for (int k = 0; k < a.Kd; ++k) acc = fmaf(feat[k], a.w[(size_t)k * 256 + f], acc);
    const int T = f >> 5, g = (f >> 3) & 3, h = (f >> 2) & 1, c = f & 3;
    a.rows[(size_t)ray * 256 + h * 128 + 16 * T + 4 * g + c] = acc;
}

// ---------------------------------------------------------------------------------------------------------------------------
// map_model_output (renderer.py:170-213) per ray, the ray's terms of the loss (loss.py: both losses are means over the rays, so a ray's
// gradient needs nothing of the others) and the adjoint of both; wave per ray, lane l holds samples l, l + 64, ...
// ---------------------------------------------------------------------------------------------------------------------------
struct CompositeArgs {
    const float *raw_rgb, *sigma, *dists;      // [N][S][3], [N][S], [N][S]
    const float *noise;                        // NULL, or [N][S]: raw_noise_std * N(0,1), added to the density before its ReLU (renderer.py:190-195)
    int n_rays, S, map_exr, composite_bkgd; float bkgd[3];
    const float *color_true, *alpha_true;
    int kind, loss_fn, alpha_loss_fn, filter_color_loss, use_hard_mask; float gamma;
    float *color, *alpha, *ray_loss;           // [N][3], [N], [N]: the predictions and each ray's share of the loss
    float *weights;                            // NULL, or [N][S]: the composite's weights a_i T_i (what the importance sampler takes, renderer.py:127-128)
    float *dgrad;                              // [M][4]: dL/d raw rgb, dL/d sigma per sample (the way back starts from these)
    float *dhead;                              // the same as one O-layout tile per block of 32 samples (rows 0-2, 3): the narrow heads' dY
    long long M;
};
constexpr int MAX_TRAIN_SAMPLES = 1024;
__device__ __forceinline__ float wave_sumf(float v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }
__device__ __forceinline__ float rgb_of(float raw, int map_exr) {
    if (map_exr) return raw > 0.0f ? raw + 1.0f : expf(raw);                                   // elu + 1 (:184-185)
    return 1.0f / (1.0f + expf(-raw));                                                         // sigmoid (:187)
}
__device__ __forceinline__ void loss_term(int fn, float t, float p, float inv_n, float &value, float &grad) {
    if (fn == NTX_LOSS_MSE) { const float e = t - p; value = e * e * inv_n; grad = -2.0f * e * inv_n; }          // loss.py:51-54
    else {                                                                                                        // smape, eps 1e-2 (:56-59)
        const float e = t - p, den = (t + p) + 1e-2f, ae = fabsf(e);
        const float sgn = e > 0.0f ? 1.0f : (e < 0.0f ? -1.0f : 0.0f);
        value = ae / den * inv_n; grad = (-sgn / den - ae / (den * den)) * inv_n;
    }
}
__device__ __forceinline__ size_t dhead_at(long long m, int row) {
    const long long blk = m >> 5; const int p = (int)(m & 31);
    return (size_t)((blk * 4 + (p >> 3)) * 256 + (row + 32 * ((p >> 2) & 1)) * 4 + (p & 3));
}
__global__ __launch_bounds__(256) void composite_loss_kernel(CompositeArgs a) {
    __shared__ float sh_a[4][MAX_TRAIN_SAMPLES], sh_T[4][MAX_TRAIN_SAMPLES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (blockIdx.x == 0) {                                 // the tail of the last block of 32 samples: no gradient
        const long long end = (a.M + 31) / 32 * 32;
        for (long long m = a.M + threadIdx.x; m < end; m += 256)
            for (int r = 0; r < 4; ++r) a.dhead[dhead_at(m, r)] = 0.0f;
    }
    const int ray = blockIdx.x * 4 + wave;
    if (ray >= a.n_rays) return;
    const int S = a.S;
    const float *sg = a.sigma + (size_t)ray * S, *ds = a.dists + (size_t)ray * S, *rr = a.raw_rgb + (size_t)ray * S * 3;
    const float *nz = a.noise ? a.noise + (size_t)ray * S : nullptr;
    // el[s] = exp(-relu(sigma) dist): a_s = 1 - el[s] (:195), and the factor of the running product, (1 - a_s) + 1e-10 (:198), is taken as
    // el[s] + 1e-10 -- the value of the reference's expression without the float32 round trip through 1 - (1 - e), which on a saturated sample
    // (e ~ 1e-6) leaves 1 - a with two digits: the ray's transmittance, and with it every gradient behind the sample, would carry that error
    float *el = sh_a[wave], *T = sh_T[wave];
    for (int s = lane; s < S; s += 64) { const float v = nz ? sg[s] + nz[s] : sg[s]; const float r = v > 0.0f ? v : 0.0f; el[s] = expf(-r * ds[s]); }    // :190-195
    __builtin_amdgcn_wave_barrier();
    // exclusive running product of (1 - a) + 1e-10, sequential like tf.math.cumprod (:198): chunks of 64 with a carry
    float carry = 1.0f;
    for (int s0 = 0; s0 < S; s0 += 64) {
        const int s = s0 + lane;
        float f = s < S ? el[s] + 1e-10f : 1.0f, incl = f;
        for (int o = 1; o < 64; o <<= 1) { const float w = __shfl_up(incl, o); if (lane >= o) incl *= w; }
        float excl = __shfl_up(incl, 1);
        if (lane == 0) excl = 1.0f;
        if (s < S) T[s] = carry * excl;
        carry *= __shfl(incl, 63);
    }
    __builtin_amdgcn_wave_barrier();
    float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f, A = 0.0f;
    for (int s = lane; s < S; s += 64) {
        const float w = (1.0f - el[s]) * T[s];
        c0 += w * rgb_of(rr[3 * s], a.map_exr); c1 += w * rgb_of(rr[3 * s + 1], a.map_exr); c2 += w * rgb_of(rr[3 * s + 2], a.map_exr);
        A += w;
        if (a.weights) a.weights[(size_t)ray * S + s] = w;
    }
    c0 = wave_sumf(c0); c1 = wave_sumf(c1); c2 = wave_sumf(c2); A = wave_sumf(A);
    if (a.composite_bkgd) { c0 += (1.0f - A) * a.bkgd[0]; c1 += (1.0f - A) * a.bkgd[1]; c2 += (1.0f - A) * a.bkgd[2]; }   // :210-211
    // the ray's terms of the loss and their derivatives (loss.py:6-49)
    const float cp[3] = {c0, c1, c2};
    float dC[3], dA = 0.0f, total = 0.0f;
    {
        const float inv_c = 1.0f / (float)(a.n_rays * 3), inv_a = 1.0f / (float)a.n_rays;
        float mask = 1.0f;
        if (a.kind == NTX_LOSS_ALPHA && a.filter_color_loss) mask = a.use_hard_mask ? (a.alpha_true[ray] > 0.0f ? 1.0f : 0.0f) : a.alpha_true[ray];   // :29-35
        for (int c = 0; c < 3; ++c) {
            float v, gr;
            loss_term(a.loss_fn, a.color_true[3 * ray + c] * mask, cp[c] * mask, inv_c, v, gr);
            total += v; dC[c] = gr * mask;
        }
        if (a.kind == NTX_LOSS_ALPHA) { float v; loss_term(a.alpha_loss_fn, a.alpha_true[ray], A, inv_a, v, dA); total += a.gamma * v; dA *= a.gamma; }   // :38
    }
    if (lane == 0) { a.color[3 * ray] = c0; a.color[3 * ray + 1] = c1; a.color[3 * ray + 2] = c2; a.alpha[ray] = A; a.ray_loss[ray] = total; }
    // adjoint.  C = sum w rgb (+ (1 - A) bkgd), A = sum w, w_i = a_i T_i, T_i = prod_{j<i} f_j, f_j = (1 - a_j) + 1e-10.  With
    // dL/dw_k = c_k + dA' (c_k = dC . rgb_k, dA' = dA - dC . bkgd):
    //     dL/da_i = T_i (dA' Z_i + (c_i - V_i)),
    //     V_i = sum_{k>i} c_k a_k prod_{i<j<k} f_j   (the colour composited behind sample i, along dC):   V_{i-1} = c_i a_i + f_i V_i,  V_{S-1} = 0
    //     Z_i = 1 - sum_{k>i} a_k prod_{i<j<k} f_j   (what is left of the ray behind sample i):            Z_{i-1} = f_i Z_i - 1e-10,   Z_{S-1} = 1
    // Nothing is divided and the opacity term is never formed as a difference of two sums.  tf.math.cumprod's own gradient
    // (TF 2.4 math_grad.py _CumprodGrad: cumsum(out * grad, reverse) / x) is the same derivative as a quotient by f_i, which on a saturated
    // sample (f_i -> 1e-10) loses the digits the forward product kept, and "dA' (1 - sum)" loses them again when the ray saturates BEHIND
    // sample i (round 5's kernel did both: profiles/r05/soak_train_seed3.txt, case 297).
    // A chunk of 64 samples is a suffix scan of the affine maps X -> b_i + f_i X (composed pairwise), carried from chunk to chunk back to front.
    if (a.composite_bkgd) dA -= (dC[0] * a.bkgd[0] + dC[1] * a.bkgd[1]) + dC[2] * a.bkgd[2];
    float Z_carry = 1.0f, V_carry = 0.0f;                  // Z, V of the last sample of the current chunk (nothing lies behind the ray's end)
    for (int s0 = ((S - 1) / 64) * 64; s0 >= 0; s0 -= 64) {
        const int s = s0 + lane;
        float cs = 0.0f, rgb[3] = {0.0f, 0.0f, 0.0f}, e = 1.0f;
        float F = 1.0f, Bz = 0.0f, Bv = 0.0f;              // the identity for the lanes past the ray's end
        if (s < S) {
            e = el[s];
            for (int c = 0; c < 3; ++c) rgb[c] = rgb_of(rr[3 * s + c], a.map_exr);
            cs = (dC[0] * rgb[0] + dC[1] * rgb[1]) + dC[2] * rgb[2];
            F = e + 1e-10f; Bz = -1e-10f; Bv = cs * (1.0f - e);
        }
        // inclusive suffix composition: lane l ends with the maps of samples l .. 63 of the chunk composed, X_{l-1} = B + F X_63
        for (int o = 1; o < 64; o <<= 1) {
            const float Fo = __shfl_down(F, o), Bzo = __shfl_down(Bz, o), Bvo = __shfl_down(Bv, o);
            if (lane + o < 64) { Bz = fmaf(F, Bzo, Bz); Bv = fmaf(F, Bvo, Bv); F *= Fo; }
        }
        float Fn = __shfl_down(F, 1), Bzn = __shfl_down(Bz, 1), Bvn = __shfl_down(Bv, 1);   // the lanes strictly behind this one
        if (lane == 63) { Fn = 1.0f; Bzn = 0.0f; Bvn = 0.0f; }
        const float Z = fmaf(Fn, Z_carry, Bzn), V = fmaf(Fn, V_carry, Bvn);
        if (s < S) {
            const float w = (1.0f - e) * T[s];
            const float d_a = T[s] * fmaf(dA, Z, cs - V);
            const float sig = nz ? sg[s] + nz[s] : sg[s];
            float gr[4];
            for (int c = 0; c < 3; ++c) {
                const float raw = rr[3 * s + c];
                const float drgb = a.map_exr ? (raw > 0.0f ? 1.0f : expf(raw)) : rgb[c] * (1.0f - rgb[c]);
                gr[c] = w * dC[c] * drgb;
            }
            gr[3] = sig > 0.0f ? d_a * ds[s] * e : 0.0f;                                       // da/dsigma = dist exp(-sigma dist)
            const long long m = (long long)ray * S + s;
            *reinterpret_cast<f32x4 *>(a.dgrad + 4 * m) = f32x4{gr[0], gr[1], gr[2], gr[3]};
            for (int r = 0; r < 4; ++r) a.dhead[dhead_at(m, r)] = gr[r];
        }
        const float F0 = __shfl(F, 0);
        Z_carry = fmaf(F0, Z_carry, __shfl(Bz, 0)); V_carry = fmaf(F0, V_carry, __shfl(Bv, 0));
    }
}
// the loss: the rays' terms added up by one workgroup in a fixed order
__global__ __launch_bounds__(1024) void loss_sum_kernel(const float *__restrict__ ray_loss, int n_rays, float *__restrict__ loss) {
    __shared__ float red[1024];
    float total = 0.0f;
    for (int r = threadIdx.x; r < n_rays; r += blockDim.x) total += ray_loss[r];
    red[threadIdx.x] = total;
    __syncthreads();
    for (int o = blockDim.x / 2; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) *loss = red[0];
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
    int device = 0, cus = 256;
    ntx_model_desc desc{};
    int Kp = 0, Kd = 0, P = 0, ptiles = 0, dtiles = 0, PS = 0, DS = 0;   // pos_map / dir_map: features, tiles of 32 rows of their buffers, k-steps of their segments
    TLayer trunk[8], feature, c1, c2, rgb, alpha;
    size_t n_weights = 0;
    long long cap = 0, cap_blocks = 0, cap_rays = 0;   // samples (blocks of 32 samples, rays) the buffers hold
    float *w = nullptr, *grad = nullptr, *adam_m = nullptr, *adam_v = nullptr;
    // what pack_kernel makes of the weights once a step: the two streams and the aux block
    float *wfwd = nullptr, *wdx = nullptr, *aux = nullptr; size_t fwd_floats = 0, dx_floats = 0;
    ntx_train::PackSeg *pack_seg = nullptr; int n_pack = 0; long long pack_total = 0;
    // forward: the encoded inputs and every layer's output (O layout), the ReLU bits, the heads' raw outputs
    float *posO = nullptr, *dirO = nullptr;
    float *act = nullptr; long long act_stride = 0;       // eleven matrices act + i * act_stride: h0 .. h7, feature, c1o, c2o
    unsigned int *bits = nullptr; long long bits_stride = 0;   // ten: h0 .. h7, c1o, c2o
    int fwd_variant = 0;
    float *sigma = nullptr, *raw_rgb = nullptr, *z = nullptr, *dists = nullptr, *noise = nullptr;
    float *dirrow = nullptr;                   // [max_rays][256]: the colour layer's direction segment per ray (dirrow_kernel)
    // backward: the composite's adjoint, the gradient at every layer's output (O layout: d c2o, d c1o, d feature, dy7 .. dy0)
    float *dgrad = nullptr, *dhead = nullptr, *gout = nullptr; long long gout_stride = 0;
    ntx_train::DwJob *jobs = nullptr; int n_jobs = 0; long long total_cost = 0;      // the weight gradients' jobs (ntx_train_device.h), their costs per block added up
    std::vector<ntx_train::DwJob> jobs_host; std::vector<int> reduce_job;      // reduce_job[i]: the job whose slots reduction i adds up
    float *dw_partial = nullptr;
    ntx_train::ReduceBatch reduce{};           // (n_split per step)
    float *color = nullptr, *alpha_out = nullptr, *ray_loss = nullptr, *loss = nullptr;
    float *weights_out = nullptr;              // caller's [N][S] buffer for the composite's weights of the next steps, or NULL
    float *stash = nullptr;                    // a second gradient (ntx_trainer_stash_gradients)
    long long adam_iterations = 0;
};

namespace {

using namespace ntx_train;


void free_all(ntx_trainer *t) {
    if (!t) return;
    (void)hipSetDevice(t->device);
    void *ptrs[] = {t->w, t->grad, t->adam_m, t->adam_v, t->wfwd, t->wdx, t->aux, t->pack_seg, t->posO, t->dirO, t->sigma, t->raw_rgb, t->z, t->dists, t->noise,
                    t->dgrad, t->dhead, t->jobs, t->dw_partial, t->stash, t->dirrow, t->color, t->alpha_out, t->ray_loss, t->loss, t->act, t->bits, t->gout};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    delete t;
}

// one contraction on caller buffers (ntx_gemm_f32): 128 x 128 tiles, panels of 16
template <bool AK>
void launch_gemm(hipStream_t st, GemmArgs g) {
    g.k_chunk = g.K;
    g.aligned = (g.lda % 4 == 0) && (g.ldb % 4 == 0) && (((uintptr_t)g.A | (uintptr_t)g.B) % 16 == 0);
    hipLaunchKernelGGL((gemm_kernel<AK, 128, 16>), dim3((g.N + 127) / 128, (g.M + TM - 1) / TM, 1), dim3(256), 0, st, g);
}

}   // namespace

namespace ntx_train {
template <int K, bool HOIST> void launch_fwd_variant(hipStream_t st, unsigned grid, const FwdArgs &a);      // ntx_train_chain.hip, one object each
void launch_fwd_chain(int variant, bool hoist, hipStream_t st, unsigned grid, const FwdArgs &a) {
    switch (variant * 2 + (hoist ? 1 : 0)) {
        case 0: launch_fwd_variant<0, false>(st, grid, a); break;
        case 1: launch_fwd_variant<0, true>(st, grid, a); break;
        case 2: launch_fwd_variant<1, false>(st, grid, a); break;
        case 3: launch_fwd_variant<1, true>(st, grid, a); break;
        case 4: launch_fwd_variant<2, false>(st, grid, a); break;
        case 5: launch_fwd_variant<2, true>(st, grid, a); break;
        case 6: launch_fwd_variant<3, false>(st, grid, a); break;
        default: launch_fwd_variant<3, true>(st, grid, a); break;
    }
}
}   // namespace ntx_train

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
    if (desc->pos_freq < 0 || desc->dir_freq < 0 || desc->param_freq < 0) return ntx_set_error(NTX_E_INVALID, "negative band count");
    const int Kp = 3 * (1 + 2 * desc->pos_freq) + desc->n_geo * (1 + 2 * desc->param_freq), Kd = 3 * (1 + 2 * desc->dir_freq) + desc->n_app * (1 + 2 * desc->param_freq);
    if (Kp > 8 * MAX_PB_GROUPS || Kd > 8 * MAX_PB_GROUPS)
        return ntx_set_error(NTX_E_UNSUPPORTED, "training: pos_map (%d) / dir_map (%d) wider than %d features (the chain holds a block's encoded inputs in registers)", Kp, Kd,
                             8 * MAX_PB_GROUPS);
    if (max_rays < 1 || max_samples_per_ray < 2 || max_samples_per_ray > MAX_TRAIN_SAMPLES || max_rays * (int64_t)max_samples_per_ray > (int64_t)1 << 30)
        return ntx_set_error(NTX_E_INVALID, "max_rays / max_samples_per_ray out of range (samples per ray <= %d)", MAX_TRAIN_SAMPLES);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ntx_set_error(NTX_E_NODEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return ntx_set_error(NTX_E_INVALID, "device %d out of range [0,%d)", device, ndev);
    ntx_trainer *t = new ntx_trainer();
    t->device = device; t->desc = *desc;
    t->P = desc->n_geo + desc->n_app; t->Kp = Kp; t->Kd = Kd;
    t->ptiles = (Kp + 31) / 32; t->dtiles = (Kd + 31) / 32;
    {   // the forward chain's build for these segment lengths, or the longest one (the streams are then padded with zero rows)
        const int psg = ((Kp + 1) / 2 + 3) / 4, dsg = ((Kd + 1) / 2 + 3) / 4;
        t->fwd_variant = 3;                                  // the smallest build that holds both segments
        for (int v = 2; v >= 0; --v)
            if (FWD_VARIANTS[v][0] >= psg && FWD_VARIANTS[v][1] >= dsg &&
                FWD_VARIANTS[v][0] + FWD_VARIANTS[v][1] <= FWD_VARIANTS[t->fwd_variant][0] + FWD_VARIANTS[t->fwd_variant][1]) t->fwd_variant = v;
        t->PS = 4 * FWD_VARIANTS[t->fwd_variant][0]; t->DS = 4 * FWD_VARIANTS[t->fwd_variant][1];
    }
    size_t p = 0;
    auto take = [&](int in, int o) { TLayer l{in, o, p, p + (size_t)in * o}; p += (size_t)in * o + o; return l; };
    int k = Kp;
    for (int i = 0; i < 8; ++i) { t->trunk[i] = take(k, 256); k = 256 + (i == 4 ? Kp : 0); }       // model.py:104-108
    t->feature = take(256, 256); t->c1 = take(256 + Kd, 256); t->c2 = take(256, 128); t->rgb = take(128, 3); t->alpha = take(256, 1);   // Keras order: alpha last
    t->n_weights = p;
    if (n_floats != p) { delete t; return ntx_set_error(NTX_E_INVALID, "weights: %zu floats, the model has %zu", n_floats, p); }
    const long long M = (long long)max_rays * max_samples_per_ray, NB = (M + 31) / 32;
    t->cap = M; t->cap_rays = max_rays; t->cap_blocks = NB;
    int rc = hipSetDevice(device) == hipSuccess ? NTX_OK : ntx_set_error(NTX_E_HIP, "hipSetDevice(%d) failed", device);
    if (rc == NTX_OK) { int n = 0; if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && n > 0) t->cus = n; }
    auto alloc = [&](float **d, size_t n, bool zero = false) -> int {
        if (rc != NTX_OK) return rc;
        if (hipMalloc((void **)d, (n ? n : 1) * sizeof(float)) != hipSuccess) return rc = ntx_set_error(NTX_E_HIP, "hipMalloc of %zu floats failed", n);
        if (zero && hipMemset(*d, 0, (n ? n : 1) * sizeof(float)) != hipSuccess) return rc = ntx_set_error(NTX_E_HIP, "hipMemset failed");
        return rc;
    };
    alloc(&t->w, p); alloc(&t->grad, p, true); alloc(&t->adam_m, p, true); alloc(&t->adam_v, p, true);
    // rows of the encoded inputs beyond Kp / Kd meet zero weights and are never written: they have to be finite
    alloc(&t->posO, (size_t)NB * t->ptiles * 1024, true); alloc(&t->dirO, (size_t)NB * t->dtiles * 1024, true);
    t->act_stride = t->gout_stride = NB * 8 * 1024; t->bits_stride = NB * 256;
    alloc(&t->act, (size_t)t->act_stride * 11); alloc((float **)&t->bits, (size_t)t->bits_stride * 10); alloc(&t->gout, (size_t)t->gout_stride * 11);
    auto act = [&](int i) { return t->act + (size_t)i * t->act_stride; };
    auto gout = [&](int i) { return t->gout + (size_t)i * t->gout_stride; };
    alloc(&t->sigma, (size_t)M); alloc(&t->raw_rgb, (size_t)M * 3); alloc(&t->z, (size_t)M); alloc(&t->dists, (size_t)M); alloc(&t->noise, (size_t)M);
    alloc(&t->dirrow, (size_t)max_rays * 256);
    alloc(&t->dgrad, (size_t)NB * 32 * 4); alloc(&t->dhead, (size_t)NB * 1024, true);       // rows 4 .. 31 of the heads' dY tile stay zero
    alloc(&t->color, (size_t)max_rays * 3); alloc(&t->alpha_out, (size_t)max_rays); alloc(&t->ray_loss, (size_t)max_rays); alloc(&t->loss, 1);
    if (rc == NTX_OK && hipMemcpy(t->w, weights, p * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) rc = ntx_set_error(NTX_E_HIP, "weight upload failed");
    // ---- the two weight streams and the aux block: what lies where, and what pack_kernel gathers it from
    std::vector<PackSeg> segs;
    long long first = 0;
    auto seg = [&](const float *src, long long sk, long long sc, int mode, int K, int ncols, int nt, size_t *stream_floats, int nsteps) {
        PackSeg s{}; s.src = src; s.sk = sk; s.sc = sc; s.mode = mode; s.K = K; s.ncols = ncols; s.nt = nt;
        const int ring = stream_floats == &t->dx_floats ? DX_RING : RING;                 // a segment is whole turns of its chain's ring
        const long long recs = ((long long)nsteps * (nt / 4) + ring - 1) / ring * ring;
        s.count = recs * 256; s.first = first; s.dst = (float *)(uintptr_t)(*stream_floats * sizeof(float));      // an offset until the buffer exists
        first += s.count; *stream_floats += (size_t)s.count;
        segs.push_back(s);
    };
    const float *W = t->w;
    // forward (model.py:104-123): W_l[k][col] row-major, k in the order of the layer's input
    auto fwd_hidden = [&](const TLayer &l, int row0, int nt) { seg(W + l.w + (size_t)row0 * l.out, l.out, 1, PACK_HIDDEN, 256, l.out, nt, &t->fwd_floats, 128); };
    auto fwd_linear = [&](const TLayer &l, int K, int nsteps) { seg(W + l.w, l.out, 1, PACK_LINEAR, K, l.out, 8, &t->fwd_floats, nsteps); };
    fwd_linear(t->trunk[0], Kp, t->PS);
    for (int i = 1; i < 8; ++i) {
        if (i == 5) { fwd_linear(t->trunk[5], Kp, t->PS); fwd_hidden(t->trunk[5], Kp, 8); }
        else fwd_hidden(t->trunk[i], 0, 8);
    }
    fwd_hidden(t->feature, 0, 8);
    fwd_linear(t->c1, Kd, t->DS); fwd_hidden(t->c1, Kd, 8);
    fwd_hidden(t->c2, 0, 4);
    const size_t n_fwd_segs = segs.size();
    // backward: row(s, kh) runs over the layer's OUTPUTS (what the lane holds of dY), the columns over its inputs: Wsrc[k][col] = W_l[col][k]
    auto dx_hidden = [&](const TLayer &l, int row0, int K, int nsteps) { seg(W + l.w + (size_t)row0 * l.out, 1, l.out, PACK_HIDDEN, K, 256, 8, &t->dx_floats, nsteps); };
    seg(W + t->rgb.w, 1, 3, PACK_LINEAR, 3, 128, 4, &t->dx_floats, 2);             // d c2o = d raw . W_rgb^T
    dx_hidden(t->c2, 0, 128, 64);                                                   // d c1o = d c2o . W_c2^T
    dx_hidden(t->c1, Kd, 256, 128);                                                 // d feature = d c1o . W_c1[the feature rows]^T
    dx_hidden(t->feature, 0, 256, 128);                                             // d h7 = d feature . W_feature^T
    seg(W + t->alpha.w, 1, 1, PACK_LINEAR, 1, 256, 8, &t->dx_floats, 1);            //        + d_sigma (x) W_alpha (model.py:111)
    for (int i = 7; i >= 1; --i) dx_hidden(t->trunk[i], i == 5 ? Kp : 0, 256, 128);  // d h(i-1) = dy_i . W_i^T (the skip's position rows take no gradient further)
    const size_t n_stream_segs = segs.size();
    // a stream ends with its first RING records again
    auto tail = [&](size_t of, size_t *stream_floats) { PackSeg s = segs[of]; s.count = (long long)(stream_floats == &t->dx_floats ? DX_RING : RING) * 256; s.first = first; s.dst = (float *)(uintptr_t)(*stream_floats * sizeof(float));
                                                        first += s.count; *stream_floats += (size_t)s.count; segs.push_back(s); };
    tail(0, &t->fwd_floats); tail(n_fwd_segs, &t->dx_floats);
    // aux: biases of the eleven layers in accumulator order, the density head's weights and bias, the colour head's
    auto aux_seg = [&](const float *src, int mode, int K, long long count, size_t at) {
        PackSeg s{}; s.src = src; s.sk = 1; s.mode = mode; s.K = K; s.count = count; s.first = first; s.dst = (float *)(uintptr_t)(at * sizeof(float));
        first += count; segs.push_back(s);
    };
    for (int i = 0; i < 8; ++i) aux_seg(W + t->trunk[i].b, PACK_AUX_ROW, 256, 256, AUX_BIAS + (size_t)i * 256);
    aux_seg(W + t->feature.b, PACK_AUX_ROW, 256, 256, AUX_BIAS + 8 * 256); aux_seg(W + t->c1.b, PACK_AUX_ROW, 256, 256, AUX_BIAS + 9 * 256);
    aux_seg(W + t->c2.b, PACK_AUX_ROW, 128, 256, AUX_BIAS + 10 * 256);
    aux_seg(W + t->alpha.w, PACK_AUX_ROW, 256, 256, AUX_ALPHA_W); aux_seg(W + t->alpha.b, PACK_COPY, 1, 1, AUX_ALPHA_B);
    aux_seg(W + t->rgb.w, PACK_AUX_RGB, 128, 384, AUX_RGB_W); aux_seg(W + t->rgb.b, PACK_COPY, 3, 3, AUX_RGB_B);
    t->pack_total = first; t->n_pack = (int)segs.size();
    alloc(&t->wfwd, t->fwd_floats); alloc(&t->wdx, t->dx_floats); alloc(&t->aux, AUX_FLOATS, true);
    if (rc == NTX_OK) {
        for (size_t i = 0; i < segs.size(); ++i) {
            const bool is_aux = i >= n_stream_segs + 2, is_fwd = i < n_fwd_segs || i == n_stream_segs;
            float *base = is_aux ? t->aux : is_fwd ? t->wfwd : t->wdx;
            segs[i].dst = (float *)((char *)base + (uintptr_t)segs[i].dst);
        }
        if (hipMalloc((void **)&t->pack_seg, segs.size() * sizeof(PackSeg)) != hipSuccess ||
            hipMemcpy(t->pack_seg, segs.data(), segs.size() * sizeof(PackSeg), hipMemcpyHostToDevice) != hipSuccess)
            rc = ntx_set_error(NTX_E_HIP, "segment table upload failed");
    }
    // ---- the weight gradients' jobs: dW_l = X_l^T . dY_l, X_l = the O-layout input of layer l, dY_l = the gradient at its output.  A job =
    // four waves side by side on the same blocks of samples (ntx_train_device.h); a slot of its partial sums holds the matrices its waves write
    std::vector<DwJob> &jobs = t->jobs_host;
    struct RJ { int job; long long at, count, pair; size_t out; };
    std::vector<RJ> rjobs;
    auto new_job = [&]() { DwJob j{}; for (DwWave &w : j.w) w.shape = -1; jobs.push_back(j); return (int)jobs.size() - 1; };
    // a matrix of the job's slot: the [K][N] kernel gradient, behind it (g_bias >= 0) the [2][N] halves of the bias gradient
    auto matrix = [&](int j, int K, int N, size_t g_kernel, long long g_bias, long long *at_bias) {
        const long long at = jobs[j].slot_floats;
        jobs[j].slot_floats += (long long)K * N;
        rjobs.push_back(RJ{j, at, (long long)K * N, 0, g_kernel});
        *at_bias = -1;
        if (g_bias >= 0) { *at_bias = jobs[j].slot_floats; jobs[j].slot_floats += 2 * N; rjobs.push_back(RJ{j, *at_bias, N, N, (size_t)g_bias}); }
        return at;
    };
    auto wave = [&](int j, int w, int shape, const float *A, int rtA, int a0, const float *B, int rtB, int b0, long long at, int K, int N, int c_lo, long long at_bias) {
        DwWave &d = jobs[j].w[w];
        d.A = A; d.rtA = rtA; d.a0 = a0; d.B = B; d.rtB = rtB; d.b0 = b0; d.shape = shape; d.out = at; d.ldc = N; d.row0 = a0 * 32; d.rows_valid = K;
        d.col0 = b0 * 32; d.c_lo = c_lo; d.c_hi = c_lo + N; d.bias_out = a0 == 0 ? at_bias : -1;
        // a block's cost in MFMAs of the full shape: 16 k-steps x 16 tiles = 256; the narrower shapes load more per MFMA (7 tiles for 12, 5 for 4) and
        // run 3 % / 6 % behind their MFMA counts (192, 64): measured per block with the kernel's clock probe (-DNTX_TRAIN_CLOCKS)
        const int cost = shape == 0 ? 256 : shape == 1 ? 198 : 68;
        if (cost > jobs[j].cost) jobs[j].cost = cost;
    };
    // a 256 x 256 layer: wave w takes X tiles 4 (w >> 1) .., dY tiles 4 (w & 1) ..
    auto layer_job = [&](const float *X, const float *dY, size_t g_kernel, size_t g_bias) {
        const int j = new_job(); long long ab; const long long at = matrix(j, 256, 256, g_kernel, (long long)g_bias, &ab);
        for (int w = 0; w < 4; ++w) wave(j, w, 0, X, 8, 4 * (w >> 1), dY, 8, 4 * (w & 1), at, 256, 256, 0, ab);
    };
    {
        const TLayer &r = t->rgb, &al = t->alpha, &c2 = t->c2, &c1 = t->c1, &f = t->feature;
        for (int i = 7; i >= 1; --i) layer_job(act(i - 1), gout(10 - i), t->trunk[i].w + (size_t)(i == 5 ? Kp : 0) * 256, t->trunk[i].b);    // trunk 7 .. 1 (the skip: its h4 rows)
        layer_job(act(7), gout(2), f.w, f.b);                                       // feature layer: X = h7
        layer_job(act(8), gout(1), c1.w + (size_t)Kd * 256, c1.b);                  // C1: the feature rows of X = [dir_map | feature]
        {   // the position rows: trunk 0 (X = pos_map) and the skip (X = [pos_map | h4]); up to three tiles of rows x two halves of the columns
            const int j = new_job(); long long ab0, ab5;
            const long long at0 = matrix(j, Kp, 256, t->trunk[0].w, (long long)t->trunk[0].b, &ab0), at5 = matrix(j, Kp, 256, t->trunk[5].w, -1, &ab5);
            for (int w = 0; w < 2; ++w) { wave(j, w, 1, t->posO, t->ptiles, 0, gout(10), 8, 4 * w, at0, Kp, 256, 0, ab0); wave(j, 2 + w, 1, t->posO, t->ptiles, 0, gout(5), 8, 4 * w, at5, Kp, 256, 0, ab5); }
        }
        {   // C1's direction rows (X = dir_map) beside C2 (X = c1o, dY 128 wide)
            const int j = new_job(); long long abd, ab2;
            const long long atd = matrix(j, Kd, 256, c1.w, -1, &abd), at2 = matrix(j, 256, 128, c2.w, (long long)c2.b, &ab2);
            for (int w = 0; w < 2; ++w) { wave(j, w, 1, t->dirO, t->dtiles, 0, gout(1), 8, 4 * w, atd, Kd, 256, 0, abd); wave(j, 2 + w, 0, act(9), 8, 4 * w, gout(0), 4, 0, at2, 256, 128, 0, ab2); }
        }
        {   // the narrow heads: X = c2o against d raw (columns 0-2 of the heads' tile), X = h7 against d sigma (column 3)
            const int j = new_job(); long long abr, aba;
            const long long atr = matrix(j, 128, 3, r.w, (long long)r.b, &abr), ata = matrix(j, 256, 1, al.w, (long long)al.b, &aba);
            wave(j, 0, 2, act(10), 4, 0, t->dhead, 1, 0, atr, 128, 3, 0, abr);
            for (int w = 0; w < 2; ++w) wave(j, 1 + w, 2, act(7), 8, 4 * w, t->dhead, 1, 0, ata, 256, 1, 3, aba);
        }
    }
    t->n_jobs = (int)jobs.size();
    size_t partial_floats = 0;
    for (DwJob &j : jobs) {
        t->total_cost += j.cost;
    }
    for (DwJob &j : jobs) {                                  // a job fills at most its share of the workgroups' slots (+ the two it may share with its neighbours)
        const long long slots = ((long long)j.cost * t->cus + t->total_cost - 1) / t->total_cost + 2;
        j.first_float = (long long)partial_floats; partial_floats += (size_t)(slots * j.slot_floats);
    }
    alloc(&t->dw_partial, partial_floats);
    if (rc == NTX_OK) {
        if (hipMalloc((void **)&t->jobs, jobs.size() * sizeof(DwJob)) != hipSuccess ||
            hipMemcpy(t->jobs, jobs.data(), jobs.size() * sizeof(DwJob), hipMemcpyHostToDevice) != hipSuccess)
            rc = ntx_set_error(NTX_E_HIP, "job table upload failed");
        if ((int)rjobs.size() > MAX_REDUCE_BATCH) rc = ntx_set_error(NTX_E_INVALID, "trainer: too many weight gradients for one launch");
        long long rfirst = 0;
        for (size_t i = 0; i < rjobs.size() && rc == NTX_OK; ++i) {
            ReduceJob &r = t->reduce.job[t->reduce.n++];
            const DwJob &j = jobs[rjobs[i].job];
            r.partial = t->dw_partial + j.first_float + rjobs[i].at; r.n_split = 0; r.stride = j.slot_floats; r.count = rjobs[i].count; r.pair = rjobs[i].pair;
            r.out = t->grad + rjobs[i].out; r.first = rfirst;
            rfirst += (rjobs[i].count + 255) / 256 * 256;
            t->reduce_job.push_back(rjobs[i].job);
        }
    }
    if (rc != NTX_OK) { free_all(t); return rc; }
    *out = t;
    return NTX_OK;
}

int ntx_trainer_destroy(ntx_trainer *t) { free_all(t); return NTX_OK; }

size_t ntx_trainer_weight_count(const ntx_trainer *t) { return t ? t->n_weights : 0; }

static float *trainer_vector(ntx_trainer *t, int what) {
    return what == NTX_TRAINER_WEIGHTS ? t->w : what == NTX_TRAINER_GRADIENTS ? t->grad : what == NTX_TRAINER_ADAM_M ? t->adam_m : what == NTX_TRAINER_ADAM_V ? t->adam_v : nullptr;
}

int ntx_trainer_get(ntx_trainer *t, int what, float *out_host, size_t n_floats) {
    if (!t || !out_host) return ntx_set_error(NTX_E_INVALID, "NULL argument");
    if (n_floats != t->n_weights) return ntx_set_error(NTX_E_INVALID, "%zu floats asked, the model has %zu", n_floats, t->n_weights);
    const float *src = trainer_vector(t, what);
    if (!src) return ntx_set_error(NTX_E_INVALID, "what = %d", what);
    TRAIN_TRY(hipSetDevice(t->device));
    TRAIN_TRY(hipDeviceSynchronize());
    TRAIN_TRY(hipMemcpy(out_host, src, n_floats * sizeof(float), hipMemcpyDeviceToHost));
    return NTX_OK;
}

int ntx_trainer_activation(ntx_trainer *t, int layer, int64_t n_samples_total, float *out_host) {
    if (!t || !out_host) return ntx_set_error(NTX_E_INVALID, "NULL argument");
    if (n_samples_total < 1 || n_samples_total > t->cap) return ntx_set_error(NTX_E_INVALID, "n_samples_total out of range");
    const float *src = nullptr; int tiles = 8;
    auto act = [&](int i) { return t->act + (size_t)i * t->act_stride; };
    auto gout = [&](int i) { return t->gout + (size_t)i * t->gout_stride; };
    if (layer >= 0 && layer < 8) src = act(layer);
    else if (layer == 8) src = act(9);
    else if (layer == 9) { src = act(10); tiles = 4; }
    else if (layer == 10) src = t->sigma;
    else if (layer >= 20 && layer < 28) src = gout(10 - (layer - 20));
    else if (layer == 28) src = gout(1);
    else if (layer == 29) src = gout(2);
    else if (layer == 11) src = t->raw_rgb;
    else if (layer == 30) src = t->dgrad;
    else return ntx_set_error(NTX_E_INVALID, "layer %d (0-7 trunk, 8 / 9 the colour layers, 10 the density, 11 the raw colour; 20-29 the kept gradients, 30 the composite's adjoint)", layer);
    TRAIN_TRY(hipSetDevice(t->device));
    TRAIN_TRY(hipDeviceSynchronize());
    if (layer == 10 || layer == 11 || layer == 30) {
        const size_t width = layer == 10 ? 1 : (layer == 11 ? 3 : 4);
        TRAIN_TRY(hipMemcpy(out_host, src, (size_t)n_samples_total * width * sizeof(float), hipMemcpyDeviceToHost)); return NTX_OK;
    }
    // O layout -> [sample][feature]
    const long long nb = (n_samples_total + 31) / 32;
    std::vector<float> tmp((size_t)nb * tiles * 1024);
    TRAIN_TRY(hipMemcpy(tmp.data(), src, tmp.size() * sizeof(float), hipMemcpyDeviceToHost));
    const int width = tiles * 32;
    for (long long m = 0; m < n_samples_total; ++m) {
        const long long blk = m >> 5; const int p = (int)(m & 31);
        const float *rec = tmp.data() + ((size_t)blk * tiles * 4 + (p >> 3)) * 256 + 32 * ((p >> 2) & 1) * 4 + (p & 3);
        float *o = out_host + (size_t)m * width;
        for (int T = 0; T < tiles; ++T)
            for (int i = 0; i < 32; ++i) o[32 * T + i] = rec[(size_t)T * 1024 + i * 4];
    }
    return NTX_OK;
}

int ntx_trainer_set_weights(ntx_trainer *t, const float *weights_host, size_t n_floats) { return ntx_trainer_set(t, NTX_TRAINER_WEIGHTS, weights_host, n_floats); }

int ntx_trainer_set(ntx_trainer *t, int what, const float *values_host, size_t n_floats) {
    if (!t || !values_host) return ntx_set_error(NTX_E_INVALID, "NULL argument");
    if (n_floats != t->n_weights) return ntx_set_error(NTX_E_INVALID, "%zu floats given, the model has %zu", n_floats, t->n_weights);
    float *dst = trainer_vector(t, what);
    if (!dst) return ntx_set_error(NTX_E_INVALID, "what = %d", what);
    TRAIN_TRY(hipSetDevice(t->device));
    TRAIN_TRY(hipDeviceSynchronize());
    TRAIN_TRY(hipMemcpy(dst, values_host, n_floats * sizeof(float), hipMemcpyHostToDevice));
    return NTX_OK;
}

int ntx_trainer_set_iterations(ntx_trainer *t, int64_t iterations) {
    if (!t || iterations < 0) return ntx_set_error(NTX_E_INVALID, "trainer is NULL or iterations < 0");
    t->adam_iterations = iterations;
    return NTX_OK;
}

int ntx_trainer_composite_weights(ntx_trainer *t, float *weights_dev) {
    if (!t) return ntx_set_error(NTX_E_INVALID, "trainer is NULL");
    t->weights_out = weights_dev;
    return NTX_OK;
}

namespace ntx_train {
__global__ void add_kernel(float *__restrict__ dst, const float *__restrict__ src, long long n) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) dst[e] += src[e];
}
}   // namespace ntx_train

int ntx_trainer_stash_gradients(ntx_trainer *t, int op, ntx_stream stream) {
    if (!t || (op != 0 && op != 1)) return ntx_set_error(NTX_E_INVALID, "trainer is NULL or op is not 0 (keep) / 1 (add back)");
    TRAIN_TRY(hipSetDevice(t->device));
    if (!t->stash) {
        if (op == 1) return ntx_set_error(NTX_E_INVALID, "no gradient was kept");
        TRAIN_TRY(hipMalloc((void **)&t->stash, t->n_weights * sizeof(float)));
    }
    if (op == 0) TRAIN_TRY(hipMemcpyAsync(t->stash, t->grad, t->n_weights * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    else hipLaunchKernelGGL(ntx_train::add_kernel, dim3((unsigned)((t->n_weights + 255) / 256)), dim3(256), 0, (hipStream_t)stream, t->grad, t->stash, (long long)t->n_weights);
    TRAIN_TRY(hipGetLastError());
    return NTX_OK;
}

int ntx_trainer_device_weights(ntx_trainer *t, const float **weights_dev) {
    if (!t || !weights_dev) return ntx_set_error(NTX_E_INVALID, "NULL argument");
    *weights_dev = t->w;
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
    const int S = n_samples, n_blocks = (int)((M + 31) / 32);
    {
        PackArgs pa{t->pack_seg, t->n_pack, t->pack_total};
        hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((t->pack_total + 255) / 256)), dim3(256), 0, st, pa);
    }
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
        EncodeArgs e{}; e.rays_o = rays_o; e.rays_d = rays_d; e.z = z; e.params = params; e.cone = cone_scale; e.rays_per_param_row = rays_per_param_row; e.M = M;
        e.n_rays = (int)n_rays; e.S = S; e.n_geo = t->desc.n_geo; e.n_app = t->desc.n_app; e.pos_freq = t->desc.pos_freq; e.dir_freq = t->desc.dir_freq;
        e.param_freq = t->desc.param_freq; e.blur_idx = blur_idx; e.posO = t->posO; e.ptiles = t->ptiles; e.dirO = t->dirO;
        e.dtiles = t->dtiles; e.dists = t->dists;
        hipLaunchKernelGGL(encode_kernel, dim3((unsigned)n_blocks, 2), dim3(64), 0, st, e);
    }
    const unsigned chain_grid = (unsigned)std::min<long long>(t->cus, (n_blocks + 3) / 4);      // persistent: a workgroup of four waves per CU
    {
        FwdArgs f{}; f.stream = t->wfwd; f.stream_bytes = (uint32_t)(t->fwd_floats * sizeof(float)); f.aux = t->aux; f.M = M;
        f.ptiles = t->ptiles; f.dtiles = t->dtiles; f.pos = t->posO; f.dir = t->dirO;
        f.act = t->act; f.act_stride = t->act_stride; f.bits = t->bits; f.bits_stride = t->bits_stride;
        f.sigma = t->sigma; f.raw_rgb = t->raw_rgb;
        // the direction segment of the colour layer per ray instead of per sample -- unless blur_idx scales an APPEARANCE parameter per sample
        // (renderer.py:155-158) or a block of 32 samples can lie in two rays (S no multiple of 32)
        const bool hoist = (blur_idx < 0 || blur_idx < t->desc.n_geo) && S % 32 == 0 && getenv("NERFTEX_TRAIN_NO_DIR_HOIST") == nullptr;
        if (hoist) {
            DirRowArgs dr{}; dr.rays_d = rays_d; dr.params = params; dr.rays_per_param_row = rays_per_param_row; dr.n_rays = (int)n_rays; dr.n_geo = t->desc.n_geo;
            dr.n_app = t->desc.n_app; dr.dir_freq = t->desc.dir_freq; dr.param_freq = t->desc.param_freq; dr.Kd = t->Kd; dr.w = t->w + t->c1.w; dr.bias = t->w + t->c1.b;
            dr.rows = t->dirrow;
            hipLaunchKernelGGL(dirrow_kernel, dim3((unsigned)n_rays), dim3(256), 0, st, dr);
        }
        f.dirrow = t->dirrow; f.n_rays = (int)n_rays; f.S = S;
        launch_fwd_chain(t->fwd_variant, hoist, st, chain_grid, f);
    }
    // ---- the composite, the loss (loss.py) and their adjoint ------------------------------------------------------------------------
    {
        CompositeArgs c{};
        c.raw_rgb = t->raw_rgb; c.sigma = t->sigma; c.dists = t->dists; c.noise = noise; c.n_rays = (int)n_rays; c.S = S; c.map_exr = (flags & NTX_FLAG_MAP_EXR) ? 1 : 0;
        c.composite_bkgd = (flags & NTX_FLAG_COMPOSITE_BKGD) ? 1 : 0;
        for (int k = 0; k < 3; ++k) c.bkgd[k] = bkgd ? bkgd[k] : 1.0f;
        c.color_true = color_true; c.alpha_true = alpha_true; c.kind = loss->kind; c.loss_fn = loss->loss_fn; c.alpha_loss_fn = loss->alpha_loss_fn;
        c.filter_color_loss = loss->filter_color_loss; c.use_hard_mask = loss->use_hard_mask; c.gamma = loss->gamma;
        c.weights = t->weights_out;
        c.color = color_pred ? color_pred : t->color; c.alpha = alpha_pred ? alpha_pred : t->alpha_out; c.ray_loss = t->ray_loss; c.dgrad = t->dgrad; c.dhead = t->dhead; c.M = M;
        hipLaunchKernelGGL(composite_loss_kernel, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, st, c);
        hipLaunchKernelGGL(loss_sum_kernel, dim3(1), dim3(1024), 0, st, t->ray_loss, (int)n_rays, loss_out ? loss_out : t->loss);
    }
    // ---- backward -----------------------------------------------------------------------------------------------------------------
    {
        DxArgs d{}; d.stream = t->wdx; d.stream_bytes = (uint32_t)(t->dx_floats * sizeof(float)); d.M = M; d.dgrad = t->dgrad;
        d.out = t->gout; d.out_stride = t->gout_stride; d.bits = t->bits; d.bits_stride = t->bits_stride;
        launch_dx_chain(st, chain_grid, d);
    }
    {   // every layer's dW = X^T . dY and db = the column sums of dY in one launch of one workgroup per CU, each with an equal share of the work ...
        DwArgs d{}; d.jobs = t->jobs; d.n_jobs = t->n_jobs; d.n_blocks = n_blocks; d.total_cost = t->total_cost; d.partial = t->dw_partial;
        const long long G = t->cus, W = t->total_cost * n_blocks;
#ifdef NTX_TRAIN_CLOCKS
        static unsigned long long *clocks = nullptr;
        const size_t nck = (size_t)G * (2 + 3 * t->n_jobs);
        if (getenv("NERFTEX_DW_CLOCKS")) {
            if (!clocks) (void)hipMalloc((void **)&clocks, nck * sizeof(unsigned long long));
            (void)hipMemsetAsync(clocks, 0, nck * sizeof(unsigned long long), st);
            d.clocks = clocks;
        }
#endif
        launch_dw(st, (unsigned)G, d);
#ifdef NTX_TRAIN_CLOCKS
        if (d.clocks) {     // development: every workgroup's pieces, in 100 MHz ticks from the earliest start
            std::vector<unsigned long long> h(nck);
            (void)hipStreamSynchronize(st);
            (void)hipMemcpy(h.data(), clocks, nck * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            unsigned long long t0 = ~0ull;
            for (long long g = 0; g < G; ++g) t0 = std::min(t0, h[g * (2 + 3 * t->n_jobs)]);
            FILE *f = fopen(getenv("NERFTEX_DW_CLOCKS"), "w");
            if (f) {
                for (long long g = 0; g < G; ++g) {
                    const unsigned long long *c = &h[g * (2 + 3 * t->n_jobs)];
                    fprintf(f, "%lld %llu %llu", g, c[0] - t0, c[1] - t0);
                    for (int j = 0; j < t->n_jobs; ++j) if (c[4 + 3 * j]) fprintf(f, "  j%d cost %d blocks %llu %llu-%llu", j, t->jobs_host[j].cost, c[2 + 3 * j], c[3 + 3 * j] - t0, c[4 + 3 * j] - t0);
                    fprintf(f, "\n");
                }
                fclose(f);
            }
        }
#endif
        // ... and the slots every job filled added up in a fixed order
        ReduceBatch rb = t->reduce;
        std::vector<int> slots(t->n_jobs);
        long long start = 0;
        for (int j = 0; j < t->n_jobs; ++j) {
            const long long span = (long long)t->jobs_host[j].cost * n_blocks;
            slots[j] = (int)(dw_last_g(G, W, start, span) - dw_first_g(G, W, start) + 1);
            start += span;
        }
        for (int i = 0; i < rb.n; ++i) rb.job[i].n_split = slots[t->reduce_job[i]];
        const long long total = rb.job[rb.n - 1].first + (rb.job[rb.n - 1].count + 255) / 256 * 256;
        hipLaunchKernelGGL(reduce_batch_kernel, dim3((unsigned)(total / 256)), dim3(256), 0, st, rb);
    }
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

/* The contraction on caller buffers (DEVICE): C[M][N] = op(A) . op(B) (+ bias) (ReLU), op = identity or transpose as
 * a_kcontig / b_kcontig say (see gemm_kernel).  For tests and benches of the kernel itself. */
int ntx_gemm_f32(const float *A, int lda, int a_kcontig, const float *B, int ldb, int b_kcontig, float *C, int ldc, int M, int N, int K, const float *bias, int relu,
                 ntx_stream stream) {
    if (!A || !B || !C || M < 1 || N < 1 || K < 1) return ntx_set_error(NTX_E_INVALID, "bad GEMM arguments");
    ntx_train::GemmArgs g{}; g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.bias = bias; g.relu = relu;
    hipStream_t st = (hipStream_t)stream;
    if (b_kcontig) return ntx_set_error(NTX_E_UNSUPPORTED, "B must be [K][N]");
    if (a_kcontig) launch_gemm<true>(st, g); else launch_gemm<false>(st, g);
    TRAIN_TRY(hipGetLastError());
    return NTX_OK;
}

}   // extern "C"
