// ntx_shard.h -- the shard map of include/nerftex.h (ntx_shard_count) as plain integer arithmetic, shared by the host side of
// ntx_gather_image, its un-shard kernel and the host-only ntx_unshard_map (which lets CPU tests drive exactly this code).
//
// The row-major pixel sequence [0, n) is cut into runs of L pixels (the last may be short); run q belongs to rank q % R and is
// that rank's local run q / R.  Rank r's shard lands in the root's staging buffer at element offset r * cap (cap = the largest
// shard = rank 0's), in local ray order.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define NTX_SHARD_HD __host__ __device__
#else
#define NTX_SHARD_HD
#endif

namespace ntx_shard {

NTX_SHARD_HD inline int64_t shard_count(int64_t n, int64_t L, int R, int rank) {
    const int64_t runs = (n + L - 1) / L;                    // run q -> rank q % R
    if (runs <= rank) return 0;
    const int64_t mine = (runs - 1 - rank) / R + 1;          // runs rank, rank + R, ...
    const int64_t last = rank + (mine - 1) * R;              // only the very last run of the image can be short
    return mine * L - (last == runs - 1 ? runs * L - n : 0);
}

// first pixel slot of rank r's block in the gather destination
NTX_SHARD_HD inline int64_t rank_block(int r, int64_t cap) { return (int64_t)r * cap; }

// pixel slot of the staging buffer that holds pixel p: rank's block + local run + offset in the run
NTX_SHARD_HD inline int64_t staging_index(int64_t p, int64_t L, int R, int64_t cap) {
    const int64_t q = p / L;
    return rank_block((int)(q % R), cap) + (q / R) * L + p % L;
}

// what ntx_gather_image does for a map: `equal` = every rank holds `cap` pixels (one ncclGather; else grouped Send/Recv with
// the exact counts); `direct` = the blocks in rank order ARE the image (contiguous bands of equal size), no staging pass
struct Plan {
    int64_t cap;
    bool equal, direct;
};
inline Plan plan(int64_t n, int64_t L, int R) {
    Plan p{shard_count(n, L, R, 0), true, false};            // rank 0 always holds the most
    for (int r = 1; r < R; ++r) p.equal = p.equal && shard_count(n, L, R, r) == p.cap;
    p.direct = p.equal && L * R >= n;                        // at most one run per rank: bands
    return p;
}

}  // namespace ntx_shard
