// ntx_comm.hip -- the multi-GPU side of the C ABI (include/nerftex.h): shard map, RCCL communicator, image gather.
// RCCL is dlopen-ed on first use -- the copy PyTorch has already loaded when there is one, so that a process never
// holds two RCCL instances -- and only its public C API (rccl.h) is used.  gfx950 only.
#include "nerftex.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// root side of the gather: staging[rank][local pixel][4] -> image[pixel][4] for the shard map of include/nerftex.h (run q
// of `run_length` pixels belongs to rank q % n_ranks, local run q / n_ranks).  Thread per pixel, 16-byte accesses.
__global__ __launch_bounds__(256) void ntx_unshard_kernel(const f32x4 *staging, int64_t n_pixels, int64_t run_length, int n_ranks,
                                                          int64_t rank_stride, f32x4 *image) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pixels) return;
    const int64_t q = p / run_length;
    image[p] = staging[(q % n_ranks) * rank_stride + (q / n_ranks) * run_length + p % run_length];
}

extern "C" int ntx_set_error(int code, const char *fmt, ...);   // nerftex.hip: per-thread message behind ntx_last_error()

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Gather)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

// NULL + message when librccl cannot be had
const Rccl *rccl() {
    static Rccl r;
    static bool tried = false;
    if (tried) return r.handle ? &r : nullptr;
    tried = true;
    const char *names[] = {"librccl.so.1", "librccl.so"};
    for (const char *n : names)   // already in the process (PyTorch links its own copy)?
        if ((r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    if (!r.handle)
        for (const char *n : names)
            if ((r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!r.handle) return nullptr;
    bool ok = true;
    auto sym = [&](auto &fn, const char *name) {
        fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(r.handle, name));
        ok = ok && fn != nullptr;
    };
    sym(r.GetUniqueId, "ncclGetUniqueId"); sym(r.CommInitRank, "ncclCommInitRank"); sym(r.CommDestroy, "ncclCommDestroy");
    sym(r.Gather, "ncclGather"); sym(r.Send, "ncclSend"); sym(r.Recv, "ncclRecv");
    sym(r.GroupStart, "ncclGroupStart"); sym(r.GroupEnd, "ncclGroupEnd"); sym(r.GetErrorString, "ncclGetErrorString");
    if (!ok) r.handle = nullptr;
    return r.handle ? &r : nullptr;
}

int64_t shard_count(int64_t n, int64_t L, int R, int rank) {
    const int64_t runs = (n + L - 1) / L;                    // run q -> rank q % R
    if (runs <= rank) return 0;
    const int64_t mine = (runs - 1 - rank) / R + 1;          // runs rank, rank + R, ...
    const int64_t last = rank + (mine - 1) * R;              // only the very last run of the image can be short
    return mine * L - (last == runs - 1 ? runs * L - n : 0);
}

}  // namespace

struct ntx_comm {
    ncclComm_t comm;
    int n_ranks, rank, device;
};

#define RCCL_TRY(expr)                                                                              \
    do {                                                                                            \
        ncclResult_t r_ = (expr);                                                                   \
        if (r_ != ncclSuccess) return ntx_set_error(NTX_E_HIP, "%s: %s", #expr, R->GetErrorString(r_)); \
    } while (0)
#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return ntx_set_error(NTX_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

extern "C" {

int64_t ntx_shard_count(int64_t n_pixels, int64_t run_length, int n_ranks, int rank) {
    if (n_pixels < 0 || run_length < 1 || n_ranks < 1 || rank < 0 || rank >= n_ranks) {
        ntx_set_error(NTX_E_INVALID, "bad shard map: n_pixels %lld run_length %lld n_ranks %d rank %d", (long long)n_pixels,
                      (long long)run_length, n_ranks, rank);
        return -1;
    }
    return shard_count(n_pixels, run_length, n_ranks, rank);
}

int ntx_comm_unique_id(uint8_t *id_out) {
    static_assert(NTX_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    if (!id_out) return ntx_set_error(NTX_E_INVALID, "id_out is NULL");
    const Rccl *R = rccl();
    if (!R) return ntx_set_error(NTX_E_UNSUPPORTED, "librccl.so.1 could not be loaded (or lacks a symbol of rccl.h)");
    ncclUniqueId id;
    RCCL_TRY(R->GetUniqueId(&id));
    memcpy(id_out, id.internal, NTX_COMM_ID_BYTES);
    return NTX_OK;
}

int ntx_comm_create(const uint8_t *id, int n_ranks, int rank, int device, ntx_comm **out) {
    if (!out) return ntx_set_error(NTX_E_INVALID, "out is NULL");
    *out = nullptr;
    if (!id || n_ranks < 1 || rank < 0 || rank >= n_ranks) return ntx_set_error(NTX_E_INVALID, "bad communicator arguments");
    const Rccl *R = rccl();
    if (!R) return ntx_set_error(NTX_E_UNSUPPORTED, "librccl.so.1 could not be loaded (or lacks a symbol of rccl.h)");
    HIP_TRY(hipSetDevice(device));
    ncclUniqueId uid;
    memcpy(uid.internal, id, NTX_COMM_ID_BYTES);
    ncclComm_t c;
    RCCL_TRY(R->CommInitRank(&c, n_ranks, uid, rank));
    *out = new ntx_comm{c, n_ranks, rank, device};
    return NTX_OK;
}

int ntx_comm_destroy(ntx_comm *comm) {
    if (!comm) return NTX_OK;
    const Rccl *R = rccl();
    if (R) (void)R->CommDestroy(comm->comm);
    delete comm;
    return NTX_OK;
}

int ntx_gather_image(ntx_comm *comm, const float *local_rgba, int64_t n_pixels, int64_t run_length, float *image_out,
                     float *staging, int root, ntx_stream stream) {
    if (!comm) return ntx_set_error(NTX_E_INVALID, "comm is NULL");
    const int R_ = comm->n_ranks, me = comm->rank;
    if (n_pixels < 0 || run_length < 1 || root < 0 || root >= R_) return ntx_set_error(NTX_E_INVALID, "bad gather arguments");
    if (n_pixels == 0) return NTX_OK;
    const Rccl *R = rccl();
    if (!R) return ntx_set_error(NTX_E_UNSUPPORTED, "librccl.so.1 could not be loaded");
    const int64_t mine = shard_count(n_pixels, run_length, R_, me);
    const int64_t cap = shard_count(n_pixels, run_length, R_, 0);                 // rank 0 always holds the most
    bool equal = true;
    for (int r = 1; r < R_; ++r) equal = equal && shard_count(n_pixels, run_length, R_, r) == cap;
    const bool contiguous = run_length * R_ >= n_pixels;                          // at most one run per rank: bands
    if (!local_rgba && mine > 0) return ntx_set_error(NTX_E_INVALID, "local_rgba is NULL");
    const bool direct = equal && contiguous;                                      // the gather lands in pixel order
    if (me == root && (!image_out || (!direct && !staging)))
        return ntx_set_error(NTX_E_INVALID, "root needs image_out%s", direct ? "" : " and staging (uneven or interleaved shards)");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(comm->device));
    float *dst = direct ? image_out : staging;
    if (equal) {
        // the one collective of the render path: every peer sends its shard straight to the root
        RCCL_TRY(R->Gather(local_rgba, dst, (size_t)cap * 4, ncclFloat, root, comm->comm, st));
    } else {
        // same exchange with the exact per-rank counts; the group is always closed, also on an error inside it
        if (me == root && mine > 0)
            HIP_TRY(hipMemcpyAsync(dst + (size_t)me * cap * 4, local_rgba, (size_t)mine * 16, hipMemcpyDeviceToDevice, st));
        RCCL_TRY(R->GroupStart());
        ncclResult_t rc = ncclSuccess;
        if (me == root) {
            for (int r = 0; r < R_ && rc == ncclSuccess; ++r) {
                const int64_t cnt = shard_count(n_pixels, run_length, R_, r);
                if (cnt > 0 && r != me) rc = R->Recv(dst + (size_t)r * cap * 4, (size_t)cnt * 4, ncclFloat, r, comm->comm, st);
            }
        } else if (mine > 0) {
            rc = R->Send(local_rgba, (size_t)mine * 4, ncclFloat, root, comm->comm, st);
        }
        const ncclResult_t rc_end = R->GroupEnd();
        RCCL_TRY(rc);
        RCCL_TRY(rc_end);
    }
    if (me == root && !direct) {
        ntx_unshard_kernel<<<dim3((unsigned)((n_pixels + 255) / 256)), dim3(256), 0, st>>>(
            reinterpret_cast<const f32x4 *>(staging), n_pixels, run_length, R_, cap, reinterpret_cast<f32x4 *>(image_out));
        HIP_TRY(hipGetLastError());
    }
    return NTX_OK;
}

}  // extern "C"
