// ntx_variant.hip -- the fused render kernel and the stand-alone MLP kernel of ONE model family.
// Compiled once per family with -DNTX_VARIANT=k (k as in kVariants[] of nerftex.hip).
#include <hip/hip_runtime.h>

#include "ntx_device.h"

#ifndef NTX_VARIANT
#error "compile with -DNTX_VARIANT=0..7"
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
#elif NTX_VARIANT == 5
using VCfg = Cfg<GEN_NGEO, GEN_NAPP, 1, 0, 1>;   // generic: any ParamNerf n_parameters = [g <= 4, a <= 8] (absent parameters = zero rows)
#define NTX_FN(name) name##_v5
#elif NTX_VARIANT == 6
using VCfg = Cfg<GEN_NGEO, GEN_NAPP, 1, 0, 1, 1>;   // flex: depth, width <= 256, skips, color_depth as the model has them (a layer loop)
#define NTX_FN(name) name##_v6
#else
using VCfg = Cfg<GEN_NGEO, GEN_NAPP, 1, 0, 1, 2>;   // flex with param_depth > 0: Dense layers on the parameter features (model.py:88-101)
#define NTX_FN(name) name##_v7
#endif

#ifdef NTX_HOIST
// further translation units of the family: -DNTX_HOIST=1 the render kernel with the direction segment hoisted per ray,
// -DNTX_HOIST=2 with the geometry-parameter blocks of the position segments hoisted as well, =3 all but parameter 0's
#if NTX_HOIST == 2
hipError_t NTX_FN(launch_render_hoist2)(int n_wgs, RenderArgs &a, hipStream_t st) {
    render_kernel<VCfg, 2><<<dim3(n_wgs), dim3(256), 0, st>>>(a);
    return hipGetLastError();
}
#elif NTX_HOIST == 3
hipError_t NTX_FN(launch_render_hoist3)(int n_wgs, RenderArgs &a, hipStream_t st) {
    render_kernel<VCfg, 3><<<dim3(n_wgs), dim3(256), 0, st>>>(a);
    return hipGetLastError();
}
#else
hipError_t NTX_FN(launch_render_hoist)(int n_wgs, RenderArgs &a, hipStream_t st) {
    render_kernel<VCfg, 1><<<dim3(n_wgs), dim3(256), 0, st>>>(a);
    return hipGetLastError();
}
#endif
#else
hipError_t NTX_FN(launch_render)(int n_wgs, RenderArgs &a, hipStream_t st) {
    render_kernel<VCfg><<<dim3(n_wgs), dim3(256), 0, st>>>(a);
    return hipGetLastError();
}

hipError_t NTX_FN(launch_instance)(int n_wgs, InstanceArgs &a, hipStream_t st) {
    instance_kernel<VCfg><<<dim3(n_wgs), dim3(256), 0, st>>>(a);
    return hipGetLastError();
}

hipError_t NTX_FN(launch_mlp)(int n_wgs, MlpArgs &a, hipStream_t st) {
    mlp_kernel<VCfg><<<dim3(n_wgs), dim3(256), 0, st>>>(a);
    return hipGetLastError();
}
#endif

}  // namespace ntx
