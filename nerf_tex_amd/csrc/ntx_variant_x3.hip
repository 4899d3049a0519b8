// ntx_variant_x3.hip -- the three MFMA kernels of ONE model family at fp16x3 precision (ntx_device_x3.h).
// Compiled once per family with -DNTX_VARIANT=k (k as in kVariants[] of nerftex.hip).
#include <hip/hip_runtime.h>

#include "ntx_device_x3.h"

#ifndef NTX_VARIANT
#error "compile with -DNTX_VARIANT=0..5"
#endif

namespace ntx {

#if NTX_VARIANT == 0
using VCfg = Cfg<1, 6, 1>;   // carpet
#define NTX_FN(name) name##_v0
#elif NTX_VARIANT == 1
using VCfg = Cfg<1, 4, 1>;   // grass, fur, plush
#define NTX_FN(name) name##_v1
#elif NTX_VARIANT == 2
using VCfg = Cfg<2, 3, 1>;   // grass_filtered
#define NTX_FN(name) name##_v2
#elif NTX_VARIANT == 3
using VCfg = Cfg<0, 0, 0>;   // plain Nerf
#define NTX_FN(name) name##_v3
#elif NTX_VARIANT == 4
using VCfg = Cfg<1, 3, 1, 1>;   // mip: IPE position encoding, grass_filtered with the blur parameter spliced out
#define NTX_FN(name) name##_v4
#else
using VCfg = Cfg<GEN_NGEO, GEN_NAPP, 1, 0, 1>;   // generic: any ParamNerf n_parameters = [g <= 4, a <= 8] (absent parameters = zero rows)
#define NTX_FN(name) name##_v5
#endif

hipError_t NTX_FN(launch_render_x3)(int n_wgs, RenderArgs &a, hipStream_t st) {
    render_kernel_x3<VCfg><<<dim3(n_wgs), dim3(256), 0, st>>>(a);
    return hipGetLastError();
}

hipError_t NTX_FN(launch_mlp_x3)(int n_wgs, MlpArgs &a, hipStream_t st) {
    mlp_kernel_x3<VCfg><<<dim3(n_wgs), dim3(256), 0, st>>>(a);
    return hipGetLastError();
}

hipError_t NTX_FN(launch_instance_x3)(int n_wgs, InstanceArgs &a, hipStream_t st) {
    instance_kernel_x3<VCfg><<<dim3(n_wgs), dim3(256), 0, st>>>(a);
    return hipGetLastError();
}

}  // namespace ntx
