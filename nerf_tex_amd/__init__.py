"""nerf_tex_amd -- MI355X-native (gfx950) implementation of NeRF-Tex's volumetric render path.

Python here is the host-side mirror of the reference's operator interface for this path
(`network/renderer.py`, `model.py`, `layer.py`, `ray_sampler.py`, `proxy.py`, `pixel_sampler.py`,
`render.py` of hbaatz/nerf-tex): same class names, constructor kwargs, call signatures and output
keys, reachable by dotted path through the same `{'module': ...}` config convention.  All arithmetic
runs in hand-written HIP kernels behind the C ABI of `include/nerftex.h`, loaded with ctypes
(`_lib.py`); PyTorch only owns device buffers, streams and `torch.distributed`.

There is no CPU fallback: modules that compute import `_lib`, which raises if
`libnerftex_hip.so` is not built.
"""

__version__ = "0.1.0"
