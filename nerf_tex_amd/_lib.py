"""ctypes binding of libnerftex_hip.so (C ABI: include/nerftex.h).

The product path has NO fallback: if the shared library is missing or fails to load, importing
this module raises.  Build it with `python -c "import __graft_entry__ as g; g.build()"` or
`make -C nerf_tex_amd/csrc`.
"""

from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# NERFTEX_LIB selects another build of the same library (kernel A/B experiments); there is still no fallback.
LIB_PATH = os.environ.get("NERFTEX_LIB") or os.path.join(_HERE, "libnerftex_hip.so")

NTX_OK, NTX_E_INVALID, NTX_E_UNSUPPORTED, NTX_E_HIP, NTX_E_NODEVICE = 0, -1, -2, -3, -4
FLAG_MAP_EXR, FLAG_COMPOSITE_BKGD, FLAG_CHECK_NUMERICS, FLAG_FP16X3, FLAG_PERTURB, FLAG_RAW_NOISE = 1, 2, 4, 8, 16, 32
ABI_VERSION = 7
KIND_PARAMNERF_EX = 2           # NTX_MODEL_PARAMNERF_EX: the descriptor's param_depth / param_width count
SKIP_MASK = 0x40000000          # NTX_SKIP_MASK: ntx_model_desc.skip carries a mask of skip-layer indices
COMM_ID_BYTES = 128
GATHER_FORCE_EXCHANGE = 1      # NTX_GATHER_FORCE_EXCHANGE
DEFAULT_MAX_RAYS = 1 << 20


class ModelDesc(C.Structure):
    """struct ntx_model_desc_ex: struct ntx_model_desc (the first twelve fields, what every entry point reads unless kind is
    NTX_MODEL_PARAMNERF_EX = 2) followed by param_depth, param_width and six reserved words"""
    _fields_ = [(n, C.c_int32) for n in ("kind", "n_geo", "n_app", "n_pos", "pos_freq", "dir_freq",
                                          "param_freq", "depth", "width", "skip", "color_depth", "pos_encoding",
                                          "param_depth", "param_width")] + [("reserved", C.c_int32 * 6)]


class RenderOpts(C.Structure):
    """struct ntx_render_opts (ABI v3): raw_noise_std / noise_seed and the global ray index map of the generators"""
    _fields_ = [("size", C.c_uint32), ("raw_noise_std", C.c_float), ("noise_seed", C.c_uint64),
                ("ray_index0", C.c_int64), ("ray_run_length", C.c_int64), ("ray_run_stride", C.c_int64), ("flags", C.c_uint32), ("reserved", C.c_uint32)]


OPT_INSTANCER_SPARSE = 1


def render_opts(raw_noise_std: float = 0.0, noise_seed: int = 0, ray_index=None, flags: int = 0) -> "RenderOpts":
    """`ray_index` = (index0, run_length, run_stride) of the call's rays (ShardMap.pixel_set(rank)[0, 2, 3]; a chunk that starts at
    ray k0 of a larger call: (k0, n, n)), or None = the index within the call."""
    i0, run, stride = (0, 0, 0) if ray_index is None else (int(v) for v in ray_index)
    return RenderOpts(C.sizeof(RenderOpts), float(raw_noise_std), int(noise_seed) & (2 ** 64 - 1), i0, run, stride, int(flags), 0)


class InstancerDesc(C.Structure):
    """struct ntx_instancer_desc (ABI v4): what the constructor of the reference's instancer keeps (instancer.cpp:53-93)"""
    _fields_ = [("size", C.c_uint32), ("b_0", C.c_float * 3), ("b_1", C.c_float * 3), ("n_parameters", C.c_int32),
                ("light_dir_parameter_idx", C.c_int32), ("light_strength_parameter_idx", C.c_int32),
                ("instance_sample_method", C.c_int32), ("use_mean_distance", C.c_int32), ("cast_shadow_rays", C.c_int32),
                ("patch_scale", C.c_float), ("min_shadow_samples", C.c_int32), ("n_shadow_samples", C.c_int32)]


class Texture(C.Structure):
    """struct ntx_texture (ABI v5): one channel matrix as loadTexture builds it (instancer.cpp:34-50), texels[r * cols + c]"""
    _fields_ = [("texels", C.POINTER(C.c_float)), ("rows", C.c_int32), ("cols", C.c_int32)]


class LossDesc(C.Structure):
    """struct ntx_loss_desc (ABI v5): network.loss.NerfLoss / AlphaLoss over mse / smape (loss.py:6-59)"""
    _fields_ = [("size", C.c_uint32), ("kind", C.c_int32), ("loss_fn", C.c_int32), ("alpha_loss_fn", C.c_int32), ("gamma", C.c_float),
                ("filter_color_loss", C.c_int32), ("use_hard_mask", C.c_int32)]


LOSS_NERF, LOSS_ALPHA, LOSS_MSE, LOSS_SMAPE = 0, 1, 0, 1
TRAINER_WEIGHTS, TRAINER_GRADIENTS, TRAINER_ADAM_M, TRAINER_ADAM_V = 0, 1, 2, 3


class NtxError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"nerftex error {code}: {message}")
        self.code = code


PRECISIONS = {"float32": 0, "fp16x3": FLAG_FP16X3}   # precision name -> its bit in `flags` (ntx_precision is chosen per call)

_fp = C.POINTER(C.c_float)
_vp = C.c_void_p

# every symbol include/nerftex.h declares: (restype, argtypes)
_u8p = C.POINTER(C.c_uint8)
_op = C.POINTER(RenderOpts)
_i64p = C.POINTER(C.c_int64)
SYMBOLS = {
    "ntx_abi_version": (C.c_int, []),
    "ntx_last_error": (C.c_char_p, []),
    "ntx_weight_count": (C.c_size_t, [C.POINTER(ModelDesc)]),
    "ntx_create": (C.c_int, [C.POINTER(ModelDesc), _fp, C.c_size_t, C.c_int, C.POINTER(_vp)]),
    "ntx_set_weights": (C.c_int, [_vp, _fp, C.c_size_t]),
    "ntx_destroy": (C.c_int, [_vp]),
    "ntx_reserve": (C.c_int, [_vp, C.c_int64]),
    "ntx_generate_rays": (C.c_int, [_fp, C.c_int, C.c_int, C.c_float, C.c_int64, C.c_int64, C.c_int, _fp, _fp,
                                    C.c_float, C.c_float, _vp, _vp, _vp, _vp, _vp]),
    "ntx_generate_rays_strided": (C.c_int, [_fp, C.c_int, C.c_int, C.c_float, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int,
                                            _fp, _fp, C.c_float, C.c_float, _vp, _vp, _vp, _vp, _vp]),
    "ntx_generate_rays_at": (C.c_int, [_fp, C.c_int, C.c_int, C.c_float, _vp, C.c_int64, C.c_int, _fp, _fp, C.c_float, C.c_float,
                                       _vp, _vp, _vp, _vp, _vp]),
    "ntx_aabb_intersect": (C.c_int, [_vp, _vp, C.c_int64, _fp, _fp, _vp, _vp]),
    "ntx_fourier_features": (C.c_int, [_vp, C.c_int64, C.c_int, C.c_int, _vp, _vp]),
    "ntx_mlp_forward": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int64, C.c_uint32, _vp, _vp, _vp]),
    "ntx_composite": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int64, C.c_int, C.c_uint32, _fp, _vp, _vp, _vp, _vp]),
    "ntx_sample_depths": (C.c_int, [_vp, C.c_int64, C.c_int, C.c_uint32, C.c_uint64, _op, _vp, _vp]),
    "ntx_render_rays": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int64, _vp, C.c_int64, C.c_int, C.c_int, C.c_uint32,
                                  _fp, _vp, C.c_uint64, _op, _vp, _vp, _vp, _vp, _vp]),
    "ntx_sample_pdf": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int64, C.c_int, C.c_int, C.c_uint32, C.c_uint64, _op, _vp, _vp]),
    "ntx_render_instanced": (C.c_int, [_vp] * 12 + [C.c_int64, C.c_int, C.c_int, C.c_float, C.c_float, C.c_uint32, _fp,
                                        _vp, _op, _vp, _vp, _vp, _vp]),
    "ntx_image_epilogue": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp]),
    "ntx_shard_count": (C.c_int64, [C.c_int64, C.c_int64, C.c_int, C.c_int]),
    "ntx_comm_preflight": (C.c_int, [C.c_int]),
    "ntx_comm_library": (C.c_char_p, []),
    "ntx_gather_plan": (C.c_int, [C.c_int64, C.c_int64, C.c_int, _i64p, _i64p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ntx_unshard_map": (C.c_int, [C.c_int64, C.c_int64, C.c_int, _i64p]),
    "ntx_comm_unique_id": (C.c_int, [_u8p]),
    "ntx_comm_create": (C.c_int, [_u8p, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    "ntx_comm_destroy": (C.c_int, [_vp]),
    "ntx_gather_image": (C.c_int, [_vp, _vp, C.c_int64, C.c_int64, _vp, _vp, C.c_int, _vp]),
    "ntx_gather_image_ex": (C.c_int, [_vp, _vp, C.c_int64, C.c_int64, _vp, _vp, C.c_int, C.c_uint32, _vp]),
    "ntx_comm_version": (C.c_int, []),
    "ntx_kernel_info": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ntx_packed_count": (C.c_size_t, [C.POINTER(ModelDesc)]),
    "ntx_pack_weights": (C.c_int, [C.POINTER(ModelDesc), _fp, C.c_size_t, _fp, C.c_size_t]),
    "ntx_packed_fp16x3_bytes": (C.c_size_t, [C.POINTER(ModelDesc)]),
    "ntx_pack_weights_fp16x3": (C.c_int, [C.POINTER(ModelDesc), _fp, C.c_size_t, C.POINTER(C.c_uint16), C.c_size_t]),
    "ntx_instancer_create": (C.c_int, [C.POINTER(InstancerDesc), _fp, C.c_int64, C.c_int, C.POINTER(_vp)]),
    "ntx_instancer_destroy": (C.c_int, [_vp]),
    "ntx_instancer_reserve": (C.c_int, [_vp, C.c_int64]),
    "ntx_instancer_count": (C.c_int64, [_vp]),
    "ntx_instancer_matrices": (C.c_int, [_vp, _fp, _fp, _fp]),
    "ntx_instancer_set_mesh": (C.c_int, [_vp, _fp, C.c_int64, C.POINTER(C.c_int32), C.c_int64]),
    "ntx_instancer_set_meshes": (C.c_int, [_vp, _fp, _fp, C.c_int64, C.POINTER(C.c_int32), _u8p, C.c_int64]),
    "ntx_instancer_set_parameter_textures": (C.c_int, [_vp, _fp, _fp, C.c_int64, C.POINTER(C.c_int32), C.c_int64, C.c_float, C.c_int,
                                                       C.POINTER(C.c_int32), C.POINTER(Texture), C.c_int, C.c_int]),
    "ntx_instancer_set_mesh_textures": (C.c_int, [_vp, _fp, C.c_int64, C.POINTER(C.c_int32), C.c_int64, C.c_int, C.POINTER(Texture)]),
    "ntx_trainer_create": (C.c_int, [C.POINTER(ModelDesc), _fp, C.c_size_t, C.c_int, C.c_int64, C.c_int, C.POINTER(_vp)]),
    "ntx_trainer_destroy": (C.c_int, [_vp]),
    "ntx_trainer_weight_count": (C.c_size_t, [_vp]),
    "ntx_trainer_get": (C.c_int, [_vp, C.c_int, _fp, C.c_size_t]),
    "ntx_trainer_activation": (C.c_int, [_vp, C.c_int, C.c_int64, _fp]),
    "ntx_trainer_set_weights": (C.c_int, [_vp, _fp, C.c_size_t]),
    "ntx_sample_noise": (C.c_int, [C.c_int64, C.c_int, C.c_uint64, _op, _vp, _vp]),
    "ntx_trainer_set": (C.c_int, [_vp, C.c_int, _fp, C.c_size_t]),
    "ntx_trainer_allreduce_gradients": (C.c_int, [_vp, _vp, _vp]),
    "ntx_comm_size": (C.c_int, [_vp]),
    "ntx_allreduce_mean_f32": (C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    "ntx_train_step_gradients": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int64, _vp, C.c_int64, C.c_int, C.c_int, C.c_uint32, _fp, C.c_uint64, _op, _vp,
                                           _vp, _vp, C.POINTER(LossDesc), _vp, _vp, _vp, _vp]),
    "ntx_trainer_adam_step": (C.c_int, [_vp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _vp]),
    "ntx_trainer_iterations": (C.c_int64, [_vp]),
    "ntx_trainer_set_iterations": (C.c_int, [_vp, C.c_int64]),
    "ntx_trainer_device_weights": (C.c_int, [_vp, C.POINTER(_vp)]),
    "ntx_set_weights_device": (C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    "ntx_trainer_composite_weights": (C.c_int, [_vp, _vp]),
    "ntx_trainer_stash_gradients": (C.c_int, [_vp, C.c_int, _vp]),
    "ntx_gemm_f32": (C.c_int, [_vp, C.c_int, C.c_int, _vp, C.c_int, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp]),
    "ntx_instancer_model_input": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int64, C.c_int, C.c_float, C.c_uint64, _op] + [_vp] * 12),
}


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found: the HIP library is not built "
                          f"(run `make -C {os.path.join(_HERE, 'csrc')}`); there is no CPU fallback")
    try:
        # PyTorch-ROCm ships its own libamdhip64; this library names /opt/rocm's.  Both have one SONAME, so whichever is mapped first serves the
        # whole process -- and torch finds "No HIP GPUs" on a runtime it was not built with.  torch first, always (measured: a fresh
        # `python -m nerf_tex_amd.main <training config>` made the trainer before torch had been imported).
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError here = ABI mismatch, also fatal
        fn.restype, fn.argtypes = res, args
    if lib.ntx_abi_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH}: ABI version {lib.ntx_abi_version()} != {ABI_VERSION}")
    return lib


lib = _load()


def check(rc: int) -> None:
    if rc != NTX_OK:
        raise NtxError(rc, lib.ntx_last_error().decode("utf-8", "replace"))


def f3(values) -> "C.Array":
    return (C.c_float * 3)(*[float(v) for v in values])
