// ntx_comm.hip -- the multi-GPU side of the C ABI (include/nerftex.h): shard map, RCCL communicator, image gather.
// RCCL is dlopen-ed on first use -- the copy PyTorch has already loaded when there is one, so that a process never
// holds two RCCL instances -- and only its public C API (rccl.h) is used.  gfx950 only.
#include "nerftex.h"
#include "ntx_shard.h"

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
    image[p] = staging[ntx_shard::staging_index(p, run_length, n_ranks, rank_stride)];
}

__global__ void ntx_scale_kernel(float *v, size_t n, float f) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) v[e] *= f;
}

extern "C" int ntx_set_error(int code, const char *fmt, ...);   // nerftex.hip: per-thread message behind ntx_last_error()

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Gather)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    char path[512] = "";
    char why[512] = "";           // dlerror() of the dlopen / dlsym that failed, taken where it failed (a later call would overwrite or clear it)
};

// librccl is bound once per process (function-local static: initialisation is thread-safe).  The copy PyTorch has already
// loaded is preferred (RTLD_NOLOAD matches its soname librccl.so.1), so that a process never holds two RCCL instances.
Rccl load_rccl() {
    Rccl r;
    const char *names[] = {"librccl.so.1", "librccl.so"};
    for (const char *n : names)   // already in the process (PyTorch links its own copy)?
        if ((r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    if (!r.handle)
        for (const char *n : names)
            if ((r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!r.handle) {
        const char *e = dlerror();
        snprintf(r.why, sizeof(r.why), "%s", e ? e : "dlopen failed without a message");
        return r;
    }
    bool ok = true;
    auto sym = [&](auto &fn, const char *name) {
        fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(r.handle, name));
        if (fn == nullptr && ok) { const char *e = dlerror(); snprintf(r.why, sizeof(r.why), "%s: %s", name, e ? e : "symbol not found"); }
        ok = ok && fn != nullptr;
    };
    sym(r.GetUniqueId, "ncclGetUniqueId"); sym(r.CommInitRank, "ncclCommInitRank"); sym(r.CommDestroy, "ncclCommDestroy");
    sym(r.Gather, "ncclGather"); sym(r.AllReduce, "ncclAllReduce"); sym(r.Send, "ncclSend"); sym(r.Recv, "ncclRecv");
    sym(r.GroupStart, "ncclGroupStart"); sym(r.GroupEnd, "ncclGroupEnd"); sym(r.GetErrorString, "ncclGetErrorString"); sym(r.GetVersion, "ncclGetVersion");
    if (!ok) { r.handle = nullptr; return r; }
    Dl_info info;   // which file the symbols really come from (diagnostics: ntx_comm_library)
    if (dladdr(reinterpret_cast<void *>(r.Gather), &info) && info.dli_fname) snprintf(r.path, sizeof(r.path), "%s", info.dli_fname);
    return r;
}

// NULL when librccl cannot be had (rccl_why() then says why)
const Rccl &rccl_state() {
    static const Rccl r = load_rccl();
    return r;
}
const Rccl *rccl() { return rccl_state().handle ? &rccl_state() : nullptr; }
const char *rccl_why() { return rccl_state().why[0] ? rccl_state().why : "no message"; }

using ntx_shard::shard_count;

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

int ntx_unshard_map(int64_t n_pixels, int64_t run_length, int n_ranks, int64_t *src_out) {
    if (n_pixels < 0 || run_length < 1 || n_ranks < 1 || (!src_out && n_pixels > 0))
        return ntx_set_error(NTX_E_INVALID, "bad shard map: n_pixels %lld run_length %lld n_ranks %d", (long long)n_pixels, (long long)run_length, n_ranks);
    const int64_t cap = shard_count(n_pixels, run_length, n_ranks, 0);
    for (int64_t p = 0; p < n_pixels; ++p) src_out[p] = ntx_shard::staging_index(p, run_length, n_ranks, cap);
    return NTX_OK;
}

const char *ntx_comm_library(void) {
    const Rccl *R = rccl();
    return R ? R->path : "";
}

int ntx_comm_version(void) {
    const Rccl *R = rccl();
    int v = 0;
    return (R && R->GetVersion(&v) == ncclSuccess) ? v : 0;
}

int ntx_comm_preflight(int device) {
    if (!rccl()) return ntx_set_error(NTX_E_UNSUPPORTED, "librccl.so.1 could not be loaded (or lacks a symbol of rccl.h): %s", rccl_why());
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return ntx_set_error(NTX_E_INVALID, "device %d out of range [0,%d)", device, ndev);
    return NTX_OK;
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

int ntx_gather_plan(int64_t n_pixels, int64_t run_length, int n_ranks, int64_t *counts_out, int64_t *offsets_out, int *equal_out,
                    int *direct_out) {
    if (n_pixels < 0 || run_length < 1 || n_ranks < 1)
        return ntx_set_error(NTX_E_INVALID, "bad shard map: n_pixels %lld run_length %lld n_ranks %d", (long long)n_pixels, (long long)run_length, n_ranks);
    const ntx_shard::Plan pl = ntx_shard::plan(n_pixels, run_length, n_ranks);
    for (int r = 0; r < n_ranks; ++r) {
        if (counts_out) counts_out[r] = shard_count(n_pixels, run_length, n_ranks, r);
        if (offsets_out) offsets_out[r] = ntx_shard::rank_block(r, pl.cap);
    }
    if (equal_out) *equal_out = pl.equal;
    if (direct_out) *direct_out = pl.direct;
    return NTX_OK;
}

int ntx_gather_image(ntx_comm *comm, const float *local_rgba, int64_t n_pixels, int64_t run_length, float *image_out,
                     float *staging, int root, ntx_stream stream) {
    return ntx_gather_image_ex(comm, local_rgba, n_pixels, run_length, image_out, staging, root, 0, stream);
}

int ntx_gather_image_ex(ntx_comm *comm, const float *local_rgba, int64_t n_pixels, int64_t run_length, float *image_out,
                        float *staging, int root, uint32_t flags, ntx_stream stream) {
    if (!comm) return ntx_set_error(NTX_E_INVALID, "comm is NULL");
    if (flags & ~(uint32_t)NTX_GATHER_FORCE_EXCHANGE) return ntx_set_error(NTX_E_INVALID, "unknown gather flags 0x%x", flags);
    const int R_ = comm->n_ranks, me = comm->rank;
    if (n_pixels < 0 || run_length < 1 || root < 0 || root >= R_) return ntx_set_error(NTX_E_INVALID, "bad gather arguments");
    if (n_pixels == 0) return NTX_OK;
    const Rccl *R = rccl();
    if (!R) return ntx_set_error(NTX_E_UNSUPPORTED, "librccl.so.1 could not be loaded");
    // the plan (ntx_shard.h; ntx_gather_plan hands the same numbers to host code): rank r's count, its block at pixel slot
    // rank_block(r, cap) of the destination, and whether the blocks already are the image
    ntx_shard::Plan pl = ntx_shard::plan(n_pixels, run_length, R_);
    // NTX_GATHER_FORCE_EXCHANGE: the exact-count branch whatever the counts -- grouped ncclSend / ncclRecv into `staging` and the un-shard
    // pass -- with the root's own block going through a send to itself as well, so that ONE rank alone runs every call of that branch
    const bool forced = (flags & NTX_GATHER_FORCE_EXCHANGE) != 0;
    if (forced) { pl.equal = false; pl.direct = false; }
    const int64_t mine = shard_count(n_pixels, run_length, R_, me), cap = pl.cap;
    if (!local_rgba && mine > 0) return ntx_set_error(NTX_E_INVALID, "local_rgba is NULL");
    const bool direct = pl.direct;                                                // the gather lands in pixel order
    if (me == root && (!image_out || (!direct && !staging)))
        return ntx_set_error(NTX_E_INVALID, "root needs image_out%s", direct ? "" : " and staging (uneven or interleaved shards)");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(comm->device));
    float *dst = direct ? image_out : staging;
    if (pl.equal) {
        // the one collective of the render path: every peer sends its shard straight to the root (block r at r * cap)
        RCCL_TRY(R->Gather(local_rgba, dst, (size_t)cap * 4, ncclFloat, root, comm->comm, st));
    } else {
        // same exchange with the exact per-rank counts; the group is always closed, also on an error inside it
        if (me == root && mine > 0 && !forced)
            HIP_TRY(hipMemcpyAsync(dst + (size_t)ntx_shard::rank_block(me, cap) * 4, local_rgba, (size_t)mine * 16, hipMemcpyDeviceToDevice, st));
        RCCL_TRY(R->GroupStart());
        ncclResult_t rc = ncclSuccess;
        if (me == root) {
            for (int r = 0; r < R_ && rc == ncclSuccess; ++r) {
                const int64_t cnt = shard_count(n_pixels, run_length, R_, r);
                if (cnt > 0 && (r != me || forced))
                    rc = R->Recv(dst + (size_t)ntx_shard::rank_block(r, cap) * 4, (size_t)cnt * 4, ncclFloat, r, comm->comm, st);
            }
            if (forced && mine > 0 && rc == ncclSuccess) rc = R->Send(local_rgba, (size_t)mine * 4, ncclFloat, root, comm->comm, st);
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

int ntx_comm_size(const ntx_comm *comm) { return comm ? comm->n_ranks : 0; }

int ntx_allreduce_mean_f32(ntx_comm *comm, float *values, size_t n, ntx_stream stream) {
    if (!comm || (!values && n > 0)) return ntx_set_error(NTX_E_INVALID, "NULL argument");
    if (n == 0) return NTX_OK;
    const Rccl *R = rccl();                                   // (a communicator of one rank runs the collective too: the call is exercised wherever there is a GPU)
    if (!R) return ntx_set_error(NTX_E_UNSUPPORTED, "librccl is not loaded: %s", rccl_why());
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(comm->device));
    RCCL_TRY(R->AllReduce(values, values, n, ncclFloat, ncclSum, comm->comm, st));
    if (comm->n_ranks > 1) hipLaunchKernelGGL(ntx_scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, values, n, 1.0f / (float)comm->n_ranks);
    HIP_TRY(hipGetLastError());
    return NTX_OK;
}

}  // extern "C"
