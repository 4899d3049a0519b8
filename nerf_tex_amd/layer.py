"""Embedding descriptions (reference: network/layer.py).

In the reference `FourierFeatures` is a Keras layer that computes; here the positional encoding is
fused into the MLP kernel, so the class only carries its configuration.  `encode` runs the
stand-alone HIP kernel (`ntx_fourier_features`) for parity checks of layer.py:22-23.
"""

from __future__ import annotations


class FourierFeatures:
    """layer.FourierFeatures (layer.py:8-23): [x | sin(2^0 x) | cos(2^0 x) | sin(2^1 x) | ...]."""

    def __init__(self, n_freq_bands: int) -> None:
        self.n_freq_bands = int(n_freq_bands)

    def out_dim(self, d: int) -> int:
        return d * (1 + 2 * self.n_freq_bands)

    def __call__(self, inputs):
        return self.encode(inputs)

    def encode(self, inputs):
        import torch
        from . import _lib
        x = inputs.contiguous().float()
        m, d = x.shape
        out = torch.empty((m, self.out_dim(d)), device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib.ntx_fourier_features(x.data_ptr(), m, d, self.n_freq_bands, out.data_ptr(),
                                                     torch.cuda.current_stream(x.device).cuda_stream))
        return out


class IntegratedPositionalEncoding:
    """layer.IntegratedPositionalEncoding (layer.py:25-41), mip-NeRF's encoding of (mean, diagonal covariance).
    Configuration only: the encoding runs inside the MLP kernel of an IPE model (`ntx_mlp_forward` on pos[M,6])."""

    def __init__(self, n_freq_bands: int) -> None:
        self.n_freq_bands = int(n_freq_bands)

    def out_dim(self, d: int = 6) -> int:
        return 6 * self.n_freq_bands


def embedding_kind(embedding) -> str:
    """'fourier' or 'ipe' for an embedding object or its reference config dict."""
    if isinstance(embedding, IntegratedPositionalEncoding):
        return "ipe"
    if isinstance(embedding, dict) and str(embedding.get("module", "")).endswith("IntegratedPositionalEncoding"):
        return "ipe"
    return "fourier"


def n_freq_bands_of(embedding) -> int:
    """Accept an instantiated FourierFeatures or the reference's config dict for one
    (`{'module': 'network.model.FourierFeatures', 'n_freq_bands': 10}`)."""
    if isinstance(embedding, (FourierFeatures, IntegratedPositionalEncoding)):
        return embedding.n_freq_bands
    if isinstance(embedding, dict):
        module = str(embedding.get("module", ""))
        if not (module.endswith("FourierFeatures") or module.endswith("IntegratedPositionalEncoding")):
            raise NotImplementedError(f"embedding '{module}' has no HIP kernel (only FourierFeatures is on the render path)")
        return int(embedding["n_freq_bands"])
    raise TypeError(f"cannot read an embedding from {type(embedding)}")
