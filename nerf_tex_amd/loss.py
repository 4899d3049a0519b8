"""network/loss.py of the reference, as descriptions the training step evaluates on the GPU (`ntx_train_step_gradients`, include/nerftex.h).

Same names, same constructor keywords: a training config's `loss_config` ({'module': 'network.loss.AlphaLoss', 'loss_fn':
'network.loss.smape', 'alpha_loss_fn': 'network.loss.mse'}, configs/config_carpet_train.py:95-99) instantiates these through
`util.remap_reference_config`.  A loss object is not called on tensors here -- value and gradient come out of the fused step -- it carries
`desc()`, the `ntx_loss_desc` of its settings."""

from __future__ import annotations

import ctypes as C

from . import _lib

_FNS = {"mse": _lib.LOSS_MSE, "smape": _lib.LOSS_SMAPE}


def _fn(path: str) -> int:
    name = path.rsplit(".", 1)[-1]
    if name not in _FNS:
        raise _lib.NtxError(_lib.NTX_E_UNSUPPORTED, f"loss function {path!r}: mse and smape (loss.py:51-59) are built")
    return _FNS[name]


def mse(*_a, **_k):
    """network.loss.mse (loss.py:51-54): evaluated inside the training step; here only a name to point a config at."""
    raise TypeError("nerf_tex_amd.loss.mse is evaluated inside ntx_train_step_gradients")


def smape(*_a, **_k):
    """network.loss.smape (loss.py:56-59)."""
    raise TypeError("nerf_tex_amd.loss.smape is evaluated inside ntx_train_step_gradients")


class NerfLoss:
    """network.loss.NerfLoss (loss.py:6-19): loss_fn(color_true, color_pred)."""

    def __init__(self, loss_fn: str = "network.loss.mse") -> None:
        self.kind, self.loss_fn, self.alpha_loss_fn = _lib.LOSS_NERF, _fn(loss_fn), _fn(loss_fn)
        self.gamma, self.filter_color_loss, self.use_hard_mask = 1.0, False, False

    def desc(self) -> "_lib.LossDesc":
        return _lib.LossDesc(C.sizeof(_lib.LossDesc), self.kind, self.loss_fn, self.alpha_loss_fn, float(self.gamma), int(self.filter_color_loss), int(self.use_hard_mask))


class AlphaLoss(NerfLoss):
    """network.loss.AlphaLoss (loss.py:21-49): colours masked by alpha_true (hard: alpha_true > 0), + gamma * alpha_loss_fn(alpha_true, alpha_pred)."""

    def __init__(self, loss_fn: str = "network.loss.mse", alpha_loss_fn: str = None, gamma: float = 1, filter_color_loss: bool = True,
                 use_hard_mask: bool = True) -> None:
        self.kind, self.loss_fn = _lib.LOSS_ALPHA, _fn(loss_fn)
        self.alpha_loss_fn = self.loss_fn if alpha_loss_fn is None else _fn(alpha_loss_fn)
        self.gamma, self.filter_color_loss, self.use_hard_mask = gamma, filter_color_loss, use_hard_mask
