"""Channel / layout stage in front of the spectral path: Mono, Stereo, MidSide, Window, and the shape
helpers Squeeze, Unsqueeze, Transpose (reference transforms/raw.py:11-262, transforms/misc.py:8-152).

These are tensor-layout transforms (select / concatenate / average two channels, strided framing): no kernels
of their own -- they hand views or small elementwise results to the HIP stages that follow, on whatever
device the input lives.  Semantics follow the reference, including:
  * `Mono.invert` looks at the module's own `inversion_mode`, not at the argument (raw.py:70-71);
  * `MidSide` scales the mid signal by 1/sqrt(2) when `pad_mid` (raw.py:157-158) and its `invert` of a
    single channel just duplicates it;
  * `Window.invert` ("crop") keeps the first `hop` samples of every chunk plus the tail of the last one.
"""
import math
from typing import List, Union

import torch

from ..utils.misc import frame
from .base import AudioTransform, InversionEnumType, NotInvertibleError

__all__ = ["Mono", "Stereo", "MidSide", "Window", "Squeeze", "Unsqueeze", "Transpose"]


# ---------------------------------------------------------------------------------------------------------
# Two-channel stage.  Written from the behaviour table that tests/golden/g12_channels.npz pins (reference
# transforms/raw.py:11-180), not from the reference's code:
#
#   transform        input channels   forward output (channel axis = -2)                 invert
#   Mono(mix)        2                one channel, (L + R) / 2                          re-insert the axis; duplicate
#   Mono(left|right) 2                one channel, L | R                                it when the MODULE's own
#   Mono(any)        != 2             unchanged                                          inversion_mode == "stereo"
#   Stereo           1 (or 1-D)       [x, x]                                             same rule; > 2 channels: keep
#   Stereo           2                unchanged;  > 2: error                             the first two
#   MidSide          1 (or 1-D)       [x, 0]                                             1 channel: [x, x]
#   MidSide          2                [(L+R)/2 (/sqrt2 if pad_mid), (L-R)/2]             [M' + S, M' - S], M' = M*sqrt2
#   all              --               normalize=True divides by the global max
#
# A 1-D signal counts as "one channel" whose channel axis is created in front (dim 0).
# ---------------------------------------------------------------------------------------------------------
def _n_channels(x: torch.Tensor) -> int:
    return 1 if x.ndim == 1 else x.shape[-2]


def _pair(first: torch.Tensor, second: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
    """Two signals -> one tensor with a 2-long channel axis; `like` says where that axis goes: a 1-D signal
    gets it in front, a (..., 1, n) signal keeps its own."""
    if like.ndim == 1:
        return torch.stack([first, second], dim=0)
    return torch.cat([first, second], dim=-2)


def _peak_normalised(x: torch.Tensor, enabled: bool) -> torch.Tensor:
    return x / x.max() if enabled else x


class _ChannelStage(AudioTransform):
    scriptable = False
    invertible = True
    needs_scaling = False


class Mono(_ChannelStage):
    _PICK = {"left": 0, "right": 1}

    def __init__(self, mode: str = "mix", normalize: bool = False, squeeze: bool = True, inversion_mode="mono"):
        super().__init__()
        self.mode, self.normalize, self.squeeze, self.inversion_mode = mode, normalize, squeeze, inversion_mode

    def __repr__(self):
        return "Mono(mode=%s, normalize=%s squeeze=%s, inversion_mode=%s)" % (self.mode, self.normalize, self.squeeze,
                                                                             self.inversion_mode)

    def get_inversion_modes(self):
        return ["mono", "stereo"]

    def _downmix(self, x: torch.Tensor) -> torch.Tensor:
        if x.shape[-2] != 2:
            return x                                            # only a stereo pair is touched
        if self.mode in self._PICK:
            c = self._PICK[self.mode]
            return x.narrow(-2, c, 1)
        if self.mode == "mix":
            return (x.sum(-2) / 2).unsqueeze(-2)
        return x

    def forward(self, x: Union[torch.Tensor, List[torch.Tensor]]):
        if isinstance(x, list):
            return [self.forward(item) for item in x]
        y = _peak_normalised(self._downmix(x), self.normalize)
        return y.squeeze(-2) if self.squeeze else y

    def forward_with_time(self, x: torch.Tensor, time: torch.Tensor):
        t0 = time[..., 0]
        return self(x), (t0 if self.squeeze else t0.unsqueeze(-1))

    def invert(self, x, inversion_mode: InversionEnumType = None, tolerance: float = 0.0):
        y = x.unsqueeze(-2) if self.squeeze else x
        # the argument is ignored: the module's own setting decides (behaviour pinned by g12)
        if self.inversion_mode == "stereo" and y.shape[-2] == 1:
            y = torch.cat([y, y], dim=-2)
        return y

    def test_inversion(self, x: torch.Tensor):
        y = self.forward(x)
        return {mode: self.invert(y, inversion_mode=mode) for mode in self.get_inversion_modes()}


class Stereo(_ChannelStage):
    def __init__(self, normalize=False, sr=44100):
        super().__init__()
        self.normalize = normalize

    def __repr__(self):
        return "Stereo(normalize=%s)" % self.normalize

    @staticmethod
    def _widen(x: torch.Tensor, too_many) -> torch.Tensor:
        n = _n_channels(x)
        if n == 1:
            return _pair(x, x, x)
        return too_many(x) if n > 2 else x

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        def refuse(_):
            raise Exception("Stereo only works with 1/2 channels")
        return _peak_normalised(self._widen(x, refuse), self.normalize)

    def invert(self, x: torch.Tensor, inversion_mode: InversionEnumType = None, tolerance: float = 1.e-4) -> torch.Tensor:
        return self._widen(x, lambda t: t[..., :2, :])


class MidSide(_ChannelStage):
    def __init__(self, sr=44100, normalize=False, pad_mid=True):
        super().__init__(sr=sr)
        self.pad_mid = pad_mid
        self.normalize = normalize

    def __repr__(self):
        return "MidSide(normalize=%s)" % self.normalize

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        n = _n_channels(x)
        if n > 2:
            raise Exception("MidSide only works with 1 or 2 channels")
        if n == 1:
            y = _pair(x, torch.zeros_like(x), x)                # mono: it is all mid, the side channel is silence
        else:
            left, right = x.unbind(-2)
            mid, side = (left + right) / 2, (left - right) / 2
            if self.pad_mid:
                mid = mid / math.sqrt(2)
            y = torch.stack([mid, side], -2)
        return _peak_normalised(y, self.normalize)

    def invert(self, x: torch.Tensor, inversion_mode: InversionEnumType = None, tolerance: float = 1.e-4) -> torch.Tensor:
        if _n_channels(x) == 1:
            return _pair(x, x, x)
        mid, side = x[..., 0, :], x[..., 1, :]                  # channels beyond the second are ignored
        if self.pad_mid:
            mid = mid * math.sqrt(2)
        return torch.stack([mid + side, mid - side], dim=-2)


class Window(AudioTransform):
    scriptable = False
    invertible = True
    needs_scaling = False

    def __init__(self, sr: int = 44100, window_size: int = 1024, hop_size: int = 256, dim: int = -1, batch_dim: int = 0,
                 inversion_mode: str = "crop"):
        super().__init__()
        self.sr = sr
        self.window_size = window_size
        self.hop_size = hop_size or self.window_size
        assert self.window_size >= self.hop_size
        self.dim = dim
        self.batch_dim = batch_dim
        self.inversion_mode = inversion_mode

    def __repr__(self):
        return "Window(ws=%s, hs=%s, dim=%s, inversion=%s)" % (self.window_size, self.hop_size, self.dim, self.inversion_mode)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return frame(x, self.window_size, self.hop_size, self.dim)

    @property
    def ratio(self):
        return self.hop_size

    def forward_with_time(self, x: torch.Tensor, time: torch.Tensor):
        chunks = self.forward(x)
        n_chunks = chunks.size(-2)
        shifts = (torch.arange(n_chunks) * self.hop_size / self.sr).to(time.device)
        return chunks, shifts.expand(tuple(chunks.shape[:-2]) + (n_chunks,)) + time.unsqueeze(-1)

    def invert(self, x: torch.Tensor, inversion_mode: InversionEnumType = None, tolerance: float = 1.e-4) -> torch.Tensor:
        dim = self.dim if self.dim >= 0 else x.ndim + self.dim
        if self.window_size == self.hop_size:                  # no overlap: chunks are simply laid end to end
            shape = list(x.shape)
            return x.reshape(shape[:dim - 1] + [shape[dim - 1] * shape[dim]] + shape[dim + 1:])
        if self.inversion_mode == "crop":
            heads = x.narrow(dim, 0, self.hop_size)             # first hop of every chunk ...
            n = x.size(dim - 1)
            shape = list(heads.shape)
            body = heads.reshape(shape[:dim - 1] + [n * self.hop_size] + shape[dim + 1:])
            tail = x.select(dim - 1, n - 1).narrow(dim - 1, self.hop_size, x.size(dim) - self.hop_size)
            return torch.cat([body, tail], dim - 1)             # ... plus the rest of the last one
        return x


class Unsqueeze(AudioTransform):
    scriptable = False
    needs_scaling = False

    @property
    def invertible(self):
        return self.dim is not None

    def __repr__(self):
        return "Unsqueeze(dim=%s)" % self.dim

    def __init__(self, sr=44100, dim=1):
        super().__init__(sr)
        self.dim = dim

    def forward(self, x: torch.Tensor):
        return x.unsqueeze(self.dim)

    def invert(self, x: torch.Tensor, inversion_mode: InversionEnumType = None, tolerance: float = 1.e-4) -> torch.Tensor:
        return x.squeeze(self.dim)


    # self-test hooks: shape checks on a stand-in tensor (reference misc.py:37-52)
    def test_forward(self, x: torch.Tensor, time: torch.Tensor = None):
        stand_in = torch.zeros(2, 512, device=x.device)
        assert self(stand_in).shape == (2, 1, 512)
        return stand_in if time is None else (stand_in, time)

    def test_inversion(self, x: torch.Tensor):
        assert self.invert(self.forward(torch.zeros(2, 512, device=x.device))).shape == (2, 512)
        return {}


class Squeeze(AudioTransform):
    scriptable = False
    needs_scaling = False

    @property
    def invertible(self):
        return self.dim is not None

    def __init__(self, sr=44100, dim=None):
        super().__init__(sr)
        self.dim = dim

    def __repr__(self):
        return "Squeeze(dim=%s)" % self.dim

    def forward(self, x: torch.Tensor):
        return x.squeeze() if self.dim is None else x.squeeze(self.dim)

    def invert(self, x: torch.Tensor, inversion_mode: InversionEnumType = None, tolerance: float = 1.e-4) -> torch.Tensor:
        if self.dim is None:
            raise NotInvertibleError
        return x.unsqueeze(self.dim)


    # self-test hooks: full and partial squeeze of a stand-in tensor (reference misc.py:89-111)
    def test_forward(self, x: torch.Tensor, time: torch.Tensor = None):
        stand_in = torch.zeros(2, 1, 512, 1, device=x.device)
        self.dim = None
        assert self(stand_in).shape == (2, 512)
        self.dim = 1
        assert self(stand_in).shape == (2, 512, 1)
        return stand_in if time is None else (stand_in, time)

    def test_inversion(self, x: torch.Tensor):
        self.dim = 1
        stand_in = torch.zeros(2, 1, 512, 1, device=x.device)
        assert self.invert(self.forward(stand_in)).shape == stand_in.shape
        return {}


class Transpose(AudioTransform):
    scriptable = False
    invertible = True
    needs_scaling = False

    def __repr__(self):
        return "Transpose(dims=%s, contiguous=%s)" % (self.dims, self.contiguous)

    def __init__(self, dims=(-2, -1), contiguous=True):
        super().__init__()
        self.dims = list(dims)
        self.contiguous = bool(contiguous)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        y = x.transpose(self.dims[0], self.dims[1])
        return y.contiguous() if self.contiguous else y

    def invert(self, x: torch.Tensor, inversion_mode: InversionEnumType = None, tolerance: float = 1.e-4):
        return self(x)

    # self-test hooks (reference misc.py:140-154)
    def test_forward(self, x: torch.Tensor, time: torch.Tensor = None):
        y = self(torch.zeros(2, 128, 512, device=x.device))
        assert y.shape == (2, 512, 128)
        return y if time is None else (y, time)

    def test_inversion(self, x: torch.Tensor):
        assert self.invert(self.test_forward(x)).shape == (2, 128, 512)
        return {}
