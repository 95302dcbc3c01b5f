"""OverlapAdd: streaming framer and overlap-add with carried state
(reference transforms/oadd.py:6-104).  `forward` writes [history | chunk] once
and returns the frames as a zero-copy strided view (as the reference's `frame`
does); `invert` is one gather-form overlap-add kernel.  The reference's own
streaming state is only well formed for chunks of at least (n_fft/hop - 1)*hop
samples (oadd.py:41: shorter ones raise on the following call).  Extension:
shorter chunks are accepted when they are a whole number of hops (down to one
hop per call, BASELINE config 5's per-hop step); the frame sequence and the
overlap-added output stream are then, sample for sample, those of any other
chunking of the same input (the sums are taken in the same order)."""
from typing import Union

import torch

from .. import ops
from ..utils.misc import frame
from .base import AudioTransform

__all__ = ["OverlapAdd"]


def _gain_compensation(n_fft: int, hop: int) -> torch.Tensor:
    """Peak of the overlap-add of all-ones frames scaled by 2/overlap (reference oadd.py:31, 57-67)."""
    frames_out = n_fft // hop - 1
    ones = frame(torch.ones(1, (frames_out + 1) * n_fft), n_fft, hop, -1)
    overlap = int(n_fft / hop)
    out = torch.zeros(1, ones.size(-2) * hop + n_fft)
    for i in range(ones.size(-2)):
        out[..., i * hop:i * hop + n_fft] += ones[..., i, :] / (overlap / 2)
    return out.max()


class OverlapAdd(AudioTransform):
    invertible = True
    scriptable = False
    needs_scaling = False

    def __repr__(self):
        return "OverlapAdd(n_fft=%s, hop_length=%s)" % (self._n_fft, self._hop)

    def __init__(self, n_fft: int = 1024, hop_length: int = 128, dim: int = -1) -> None:
        super().__init__()
        self._n_fft, self._hop = int(n_fft), int(hop_length)
        self.register_buffer("n_fft", torch.tensor(n_fft))
        self.register_buffer("hop_length", torch.tensor(hop_length))
        self.frames_out = self._n_fft // self._hop - 1
        self._keep = self.frames_out * self._hop
        self.register_buffer("input_buffer", torch.zeros(self._keep))
        self.register_buffer("output_buffer", torch.zeros(self._keep))
        self.register_buffer("gain_compensation", _gain_compensation(self._n_fft, self._hop))

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        for k in ("input_buffer", "output_buffer"):
            if prefix + k in state_dict:
                self._buffers[k] = torch.zeros_like(state_dict[prefix + k])
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def _follow(self, x):
        if self.gain_compensation.device != x.device:
            self.to(x.device)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        self._follow(x)
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1])
        hist = self.input_buffer if self.input_buffer.shape[:-1] == lead else None   # zeros on a new batch shape
        hist2 = hist.reshape(-1, self._keep) if hist is not None else None
        buf, new_hist, nw = ops.oadd_forward(x2, hist2, self._keep, self._n_fft, self._hop)
        self.input_buffer = new_hist.reshape(tuple(lead) + (self._keep,))
        frames = torch.as_strided(buf, (buf.shape[0], nw, self._n_fft), (buf.stride(0), self._hop, 1))
        if len(lead) == 1:
            return frames
        # extra leading dims: keep the zero-copy view by splitting dim 0
        return torch.as_strided(buf, tuple(lead) + (nw, self._n_fft),
                                tuple(buf.stride(0) * s for s in _row_strides(lead)) + (self._hop, 1))

    def forward_with_time(self, x: torch.Tensor, time: torch.Tensor):
        transform = self.forward(x)
        shifts = torch.arange(transform.size(-2), device=time.device) * self._hop / self.sr
        return transform, shifts + time.unsqueeze(-1)

    def invert(self, x: torch.Tensor, inversion_mode: Union[str, None] = None,
               tolerance: Union[float, None] = None) -> torch.Tensor:
        self._follow(x)
        lead = x.shape[:-2]
        x3 = x.reshape((-1,) + tuple(x.shape[-2:]))
        tail = self.output_buffer if self.output_buffer.shape[:-1] == lead else None
        tail2 = tail.reshape(-1, self._keep) if tail is not None else None
        out, new_tail = ops.oadd_invert(x3, tail2, self._n_fft, self._hop, self._keep, self.gain_compensation)
        self.output_buffer = new_tail.reshape(tuple(lead) + (self._keep,))
        return out.reshape(tuple(lead) + (out.shape[-1],))


    # -- the reference's state helpers (oadd.py:33-67), for callers that drive the buffers themselves ------------
    def get_input_buffer(self, x: torch.Tensor) -> torch.Tensor:
        """History to put in front of chunk `x` (zeros on a new batch shape); the last (n_fft/hop - 1) hops of `x`
        become the next history."""
        self._follow(x)
        lead = x.shape[:-1]
        hist = self.input_buffer.clone() if self.input_buffer.shape[:-1] == lead else \
            torch.zeros(tuple(lead) + (self._keep,), device=x.device, dtype=x.dtype)
        self.input_buffer = x[..., -self._keep:]
        return hist

    def get_output_buffer(self, x: torch.Tensor) -> torch.Tensor:
        """Carried overlap-add tail for frames `x` (..., n, n_fft): zeros on a new batch shape."""
        self._follow(x)
        lead = x.shape[:-2]
        if self.output_buffer.shape[:-1] == lead:
            return self.output_buffer.clone()
        return torch.zeros(tuple(lead) + (self._keep,), device=x.device, dtype=x.dtype)

    def _forward_without_update(self, x: torch.Tensor) -> torch.Tensor:
        return frame(x, self._n_fft, self._hop, dim=-1)

    def _invert_without_update(self, x: torch.Tensor, inversion_mode: Union[str, None] = None,
                               tolerance: Union[float, None] = None) -> torch.Tensor:
        """Overlap-add of frames (..., n, n_fft) with no carried state: n * hop + n_fft samples, each frame scaled by
        2 / overlap and the sum divided by the gain compensation."""
        self._follow(x)
        lead = x.shape[:-2]
        x3 = x.reshape((-1,) + tuple(x.shape[-2:])).contiguous()
        out, tail = ops.oadd_invert(x3, None, self._n_fft, self._hop, self._keep, self.gain_compensation)
        full = torch.cat([out, tail / self.gain_compensation, torch.zeros(out.shape[0], self._hop, device=x.device)], -1)
        # oadd.py:65: every frame enters as x / (overlap / 2).  The kernel sums unscaled frames; for the usual
        # power-of-two overlaps the factor commutes with the sum and the division bit for bit (golden G19).
        overlap = int(self._n_fft / self._hop)
        if overlap != 2:
            full = full / (overlap / 2)
        return full.reshape(tuple(lead) + (full.shape[-1],))


def _row_strides(lead):
    """Contiguous strides (in rows) of the leading batch dims."""
    out, acc = [], 1
    for d in reversed(lead):
        out.append(acc)
        acc *= d
    return tuple(reversed(out))
