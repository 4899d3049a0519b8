// ntx_train_chain.hip -- one of the matrix-core kernels of a training step (ntx_train_device.h) per object: -DNTX_TRAIN_KERNEL=0..3 the forward
// chain for the segment lengths FWD_VARIANTS[k], 4 the chain back, 5 the weight gradients, 6-9 the forward chain again with the colour layer's
// direction segment hoisted per ray.  gfx950 only.
#if NTX_TRAIN_KERNEL < 4 || NTX_TRAIN_KERNEL >= 6
#define NTX_TRAIN_FWD 1
#elif NTX_TRAIN_KERNEL == 4
#define NTX_TRAIN_DX 1
#else
#define NTX_TRAIN_DW 1
#endif
#include "ntx_train_device.h"

namespace ntx_train {

#if NTX_TRAIN_KERNEL < 4 || NTX_TRAIN_KERNEL >= 6
template <int K, bool HOIST>
void launch_fwd_variant(hipStream_t st, unsigned grid, const FwdArgs &a) {
    hipLaunchKernelGGL((fwd_chain_kernel<FWD_VARIANTS[K][0], FWD_VARIANTS[K][1], HOIST>), dim3(grid), dim3(256), 0, st, a);
}
template void launch_fwd_variant<(NTX_TRAIN_KERNEL < 4 ? NTX_TRAIN_KERNEL : NTX_TRAIN_KERNEL - 6), (NTX_TRAIN_KERNEL >= 6)>(hipStream_t, unsigned, const FwdArgs &);
#elif NTX_TRAIN_KERNEL == 4
void launch_dx_chain(hipStream_t st, unsigned grid, const DxArgs &a) { hipLaunchKernelGGL(dx_chain_kernel, dim3(grid), dim3(256), 0, st, a); }
#else
void launch_dw(hipStream_t st, unsigned grid, const DwArgs &a) { hipLaunchKernelGGL(dw_kernel, dim3(grid), dim3(256), 0, st, a); }
#endif

}   // namespace ntx_train
