"""Normalize: global scalar affine normalisation (reference transforms/norm.py:12-99).

`scale_data` is one fused device reduction (min / max / sum / sum of squares in
fp64, pointwise.hip) -- no host synchronisation; `forward` / `invert` are one
elementwise kernel, and `Magnitude` / `MFCC` fold them into their GEMM epilogue.
"""
from typing import Union

import torch

from .. import ops
from .base import AudioTransform

__all__ = ["Normalize"]

MagnitudeModeType = Union[None, str]


def stats_to_affine(st: torch.Tensor, n: int, mode: str):
    """[min, max, sum, sumsq] (fp64, on device) -> (offset, scale) 0-dim fp32 tensors
    with the reference's definitions (norm.py:25-38)."""
    mn, mx = st[0].float(), st[1].float()
    if mode == "unipolar":
        return mn, mx - mn                       # (x - min).max() == fl(max - min)
    if mode == "bipolar":
        offset = (mx + mn) / 2
        return offset, mx - offset
    if mode == "gaussian":
        mean = st[2] / n
        var = (st[3] - st[2] * mean) / max(n - 1, 1)   # unbiased, like Tensor.std()
        return mean.float(), var.clamp_min(0).sqrt().float()
    raise ValueError("unknown normalisation mode %s" % mode)


class Normalize(AudioTransform):
    scriptable = False

    def __repr__(self):
        return "Normalize(mode=%s)" % self.mode

    def __init__(self, mode: MagnitudeModeType = "gaussian"):
        super().__init__()
        self.mode = mode
        self.needs_scaling = True
        self.register_buffer("offset", torch.zeros(0))
        self.register_buffer("scale", torch.ones(1))

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        for k in ("offset", "scale"):     # shapes change once scaled ((0,)/(1,) -> 0-dim)
            if prefix + k in state_dict:
                self._buffers[k] = torch.zeros_like(state_dict[prefix + k])
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def set_affine(self, offset: torch.Tensor, scale: torch.Tensor) -> None:
        self.offset, self.scale = offset, scale
        self.needs_scaling = False

    def scale_data(self, x: torch.Tensor) -> None:
        if self.mode in ("unipolar", "bipolar", "gaussian"):
            st = ops.stats(x, take_abs=False)
            self.set_affine(*stats_to_affine(st, x.numel(), self.mode))
        self.needs_scaling = False

    def _params(self, x):
        if self.offset.numel() != 1:
            raise RuntimeError("Normalize used before scale_data(): offset has shape %s" % (tuple(self.offset.shape),))
        if self.offset.device != x.device:
            self.to(x.device)
        return self.offset, self.scale

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        off, sc = self._params(x)
        return ops.affine(x, off, sc, inverse=False)

    def invert(self, x: torch.Tensor, inversion_mode=None, **kwargs) -> torch.Tensor:
        off, sc = self._params(x)
        return ops.affine(x, off, sc, inverse=True)

    # -- self-test hooks (reference norm.py:49-97): every mode on 256-sample frames of the audio ------------
    def _framed(self, x: torch.Tensor) -> torch.Tensor:
        from ..utils.misc import frame
        return frame(x, min(256, x.shape[-1]), min(64, x.shape[-1]), -1).contiguous()

    def test_forward(self, x: torch.Tensor, time: torch.Tensor = None):
        x = self._framed(x)
        tolerance = torch.finfo(x.dtype).eps
        x_norm = x
        for mode in self.get_normalization_modes():
            self.mode = mode
            self.scale_data(x)
            x_norm = self(x)
            if mode == "unipolar":
                assert float(x_norm.min()) == 0.0 and float(x_norm.max()) == 1.0
            elif mode == "bipolar":
                assert float(x_norm.min()) == -1.0 and float(x_norm.max()) == 1.0
            else:
                assert float(x_norm.mean()) < tolerance and float((x_norm.std() - 1).pow(2)) < tolerance
        return x_norm if time is None else (x_norm, time)

    def test_inversion(self, x: torch.Tensor, tolerance: float = None):
        x = self._framed(x)
        if tolerance is None:
            tolerance = torch.finfo(x.dtype).eps
        for mode in self.get_normalization_modes():
            self.mode = mode
            self.scale_data(x)
            back = self.invert(self(x))
            assert float((x.min() - back.min()).pow(2)) < tolerance and float((x.max() - back.max()).pow(2)) < tolerance
        return {}

    @classmethod
    def test_scripted_transform(cls, transform, invert: bool = True):
        x = torch.rand((5, 256), device="cuda")
        transform.scale_data(x)
        y = transform(x)
        if invert:
            transform.invert(y)

    def get_normalization_modes(self):
        return ["unipolar", "bipolar", "gaussian"]
