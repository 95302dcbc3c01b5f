"""Phase-side and stacked spectral representations: Real, Imaginary, Phase, IF, Cartesian, Polar, PolarIF.

Drop-ins for the reference classes of the same names (transforms/spectral_repr.py:21-140, 261-537): same
constructors, `scale_data / forward / invert` semantics and quirks --

* `keep_nyquist=False` drops output bin 0 and `invert` pads a zero bin at the END (:49-53, 276-277);
* `IF` divides all rows but the last ("forward"), all but the first ("backward") or the interior rows
  ("central") by pi / -pi / 2 pi, row 0 (resp. the last row) being the raw unwrapped phase (:318-328), and
  `invert` mirrors that (:360-370); the frame weighting of `weighted=True` is not undone by `invert`;
* `fint_central` leaves the odd rows at zero for an odd frame count and overwrites the last row for an even
  one (utils/misc.py:96-104).

Deliberate differences: `invert` never modifies its argument (the reference integrates in place when the
representation has no normalisation), and `IF(weighted=True)` keeps working after its first call (the
reference caches a 1-D window and then asks it for `size(-2)`, which raises; fixture key
`ifw_second_call_raises` in tests/golden/g11_phase_repr.npz records that).

The arithmetic is phase_repr.hip: one scan kernel per call reads the complex spectrum once and writes the
normalised representation (angle, unwrap, finite difference, row scaling, weighting and Normalize fused).
"""
from typing import Tuple, Union

import torch

from .. import ops
from .base import AudioTransform, InversionEnumType
from .norm import Normalize
from .spectral_repr import Magnitude, _Identity

__all__ = ["Real", "Imaginary", "Phase", "IF", "SpectralRepresentation", "Cartesian", "Polar", "PolarIF"]


def _as_complex(x: torch.Tensor) -> torch.Tensor:
    """Tensor.angle() of a real tensor is 0 / pi: give the scan kernels a complex view of real input."""
    return x if x.is_complex() else torch.complex(ops._f32c(x), torch.zeros_like(x, dtype=torch.float32))


def _pad_last_bin(x: torch.Tensor) -> torch.Tensor:
    return torch.cat([x, torch.zeros(x.shape[:-1] + (1,), device=x.device, dtype=x.dtype)], -1)


class _Representation(AudioTransform):
    scriptable = False
    invertible = True
    needs_scaling = True

    def __init__(self, sr: int = 44100, mode: Union[str, None] = None, keep_nyquist: bool = True):
        super().__init__(sr=sr)
        self.norm = _Identity() if (mode is None or mode == "none") else Normalize(mode)
        self.keep_nyquist = keep_nyquist

    def _affine(self, x):
        if isinstance(self.norm, Normalize):
            return self.norm._params(x)
        return None, None

    def _raw(self, x: torch.Tensor) -> torch.Tensor:
        """The representation before Normalize (what scale_data measures)."""
        raise NotImplementedError

    def scale_data(self, x: torch.Tensor) -> None:
        if isinstance(self.norm, Normalize):
            self.norm.scale_data(self._raw(x))

    def invert(self, x, inversion_mode: InversionEnumType = None, tolerance: float = 1.e-4) -> torch.Tensor:
        off, sc = self._affine(x)
        out = ops.affine(x, off, sc, inverse=True) if off is not None else x
        return out if self.keep_nyquist else _pad_last_bin(out)

    # -- self-test hooks (the reference's test file drives every class through them) ------------------
    def test_forward(self, x: torch.Tensor, time: torch.Tensor = None):
        from ._selftest import stft_then
        return stft_then(self, x, time)

    @classmethod
    def test_scripted_transform(cls, transform, invert: bool = True):
        from ._selftest import random_spectrum
        X = random_spectrum("cuda")
        transform.scale_data(X)
        y = transform(X)
        if invert:
            transform.invert(y)

    def _round_trip(self, x: torch.Tensor, rebuild, n_fft: int = 1024, hop: int = 256):
        """window-less STFT -> representation -> invert -> `rebuild(X, inverted)` -> window-less ISTFT."""
        from ._selftest import rect_stft, rect_istft
        X, batch_shape = rect_stft(x, n_fft, hop)
        self.scale_data(X)
        inv = self.invert(self(X))
        return rect_istft(rebuild(X, inv), batch_shape, n_fft, hop)


class Real(_Representation):
    def __repr__(self):
        return "Real(norm=%s)" % self.norm.mode

    def _raw(self, x):
        return x.real

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not self.keep_nyquist:
            x = x[..., 1:]
        off, sc = self._affine(x)
        re = x.real
        return ops.affine(re, off, sc) if off is not None else re

    def test_inversion(self, x: torch.Tensor):
        # the reference's scenario uses n_fft 512 / hop 128 for this class
        return {"direct": self._round_trip(x, lambda X, re: torch.complex(re.contiguous(), X.imag.contiguous()), 512, 128)}


class Imaginary(_Representation):
    def __repr__(self):
        return "Imaginary(norm=%s)" % self.norm.mode

    def _raw(self, x):
        return x.imag

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if torch.is_complex(x):
            off, sc = self._affine(x)
            x = ops.affine(x.imag, off, sc) if off is not None else x.imag
        else:
            x = torch.zeros_like(x)
        return x if self.keep_nyquist else x[..., 1:]

    def test_inversion(self, x: torch.Tensor):
        return {"direct": self._round_trip(x, lambda X, im: torch.complex(X.real.contiguous(), im.contiguous()))}


class Phase(_Representation):
    def __init__(self, sr: int = 44100, mode: Union[str, None] = None, keep_nyquist: bool = True, unwrap: bool = False):
        super().__init__(sr=sr, mode=mode, keep_nyquist=keep_nyquist)
        self.unwrap = unwrap

    def __repr__(self):
        return "Phase(norm=%s, unwrap=%s)" % (self.norm.mode, self.unwrap)

    def _raw(self, x):
        return ops.phase_scan(_as_complex(x), "unwrap" if self.unwrap else "angle")

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        off, sc = self._affine(x)
        y = ops.phase_scan(_as_complex(x), "unwrap" if self.unwrap else "angle", offset=off, scale=sc)
        return y if self.keep_nyquist else y[..., 1:]

    def test_inversion(self, x: torch.Tensor):
        return {"direct": self._round_trip(x, lambda X, ph: ops.polar_to_complex(X.abs(), ph))}


class IF(_Representation):
    def __repr__(self):
        return "IF(method=%s, norm=%s)" % (self.method, self.norm.mode)

    def __init__(self, sr: int = 44100, mode: Union[str, None] = "gaussian", method: Union[str, None] = "forward",
                 weighted=False, keep_nyquist: bool = True):
        super().__init__(sr=sr, mode=mode)
        self.method = method
        self.weighted = weighted
        self.weighted_window = torch.zeros(0)
        self.keep_nyquist = keep_nyquist
        self.register_buffer("eps", torch.tensor(torch.finfo(torch.float32).eps))

    def get_if_methods(self):
        return ["backward", "forward", "central"]

    def _get_weighted_window(self, x: torch.Tensor) -> torch.Tensor:
        """Parabolic frame weight (1.5 N)/(N^2-1) (1 - ((n-(N/2-1))/(N/2))^2), N = frames (:337-345)."""
        N = x.size(-2)
        if self.weighted_window.numel() != N or self.weighted_window.device != x.device:
            n = torch.arange(N)
            self.weighted_window = ((1.5 * N) / (N ** 2 - 1) * (1 - ((n - (N / 2 - 1)) / (N / 2)) ** 2)).to(x.device)
        return self.weighted_window

    def _scan(self, data, off=None, sc=None):
        if self.method not in ("backward", "forward", "central"):
            raise AttributeError("method %s not known" % self.method)
        data = _as_complex(data)
        if self.method == "central" and data.size(-2) == 1:
            # fdiff_central of a single frame concatenates that frame twice (utils/misc.py:77-80)
            ph = ops.phase_scan(data, "angle")
            y = torch.cat([ph, ph], -2)
            if self.weighted:
                y = self._get_weighted_window(y).view(-1, 1) * y
            return ops.affine(y, off, sc) if off is not None else y
        window = self._get_weighted_window(data) if self.weighted else None
        return ops.phase_scan(data, self.method, frame_window=window, offset=off, scale=sc)

    def get_if(self, data: torch.Tensor) -> torch.Tensor:
        return self._scan(data)

    _raw = get_if

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        off, sc = self._affine(x)
        y = self._scan(x, off, sc)
        return y if self.keep_nyquist else y[..., 1:]

    def invert(self, x, inversion_mode: InversionEnumType = None, tolerance: float = 1.e-4) -> torch.Tensor:
        off, sc = self._affine(x)
        if self.method in ("backward", "forward", "central"):
            out = ops.phase_integrate(x, self.method, off, sc)
        else:                                   # the reference integrates nothing for an unknown method
            out = ops.affine(x, off, sc, inverse=True) if off is not None else x
        return out if self.keep_nyquist else _pad_last_bin(out)

    def test_inversion(self, x: torch.Tensor):
        outs = {}
        for method in self.get_if_methods():
            self.method = method
            outs[method] = self._round_trip(x, lambda X, ph: ops.polar_to_complex(X.abs(), ph))
        return outs


SpectralRepresentationType = Union[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]]


class SpectralRepresentation(AudioTransform):
    scriptable = False
    invertible = True
    needs_scaling = True

    def __init__(self, sr: int = 44100, magnitude_transform=None, phase_transform=None, magnitude_args={},
                 phase_args={}, stack=-2, keep_nyquist: bool = True):
        super().__init__(sr=sr)
        if type(self) == SpectralRepresentation:
            raise RuntimeError("SpectralRepresentation should not be called directly.")
        self.keep_nyquist = keep_nyquist
        self.magnitude = magnitude_transform(sr=sr, **magnitude_args, keep_nyquist=keep_nyquist)
        self.phase = phase_transform(sr=sr, **phase_args, keep_nyquist=keep_nyquist)
        self.stack = stack

    def scale_data(self, x: torch.Tensor) -> None:
        self.magnitude.scale_data(x)
        self.phase.scale_data(x)

    def _one_pass(self, x):
        """Polar with its default parts -- Magnitude over a banded bank of F filters next to a plain Phase,
        stacked on dim -2 -- is one pass over the spectrum that writes the stacked tensor in place."""
        mag, ph = self.magnitude, self.phase
        if not (type(mag) is Magnitude and type(ph) is Phase and self.stack == -2 and x.is_complex() and x.is_cuda
                and x.ndim >= 2 and mag.mel and mag.keep_nyquist and ph.keep_nyquist and not ph.unwrap):
            return None
        band = mag._band_of("mel_bank")
        if band is None or band.N != x.shape[-1] or mag.mel_bank.shape[-2] != x.shape[-1]:
            return None
        mag._follow(x)
        m_off, m_sc = mag._affine()
        p_off, p_sc = ph._affine(x)
        return ops.polar_forward(x, band, mag.contrast_mode, m_off, m_sc, mag._eps, p_off, p_sc)

    # -- fusion with the preceding STFT / DGT stage (ComposeAudioTransform.forward) -----------------------
    def can_fuse_with(self, stage, x: torch.Tensor) -> bool:
        """True when `stage` (offline STFT/DGT, n_fft=1024, hop=256) followed by this Polar can run as one kernel
        on the audio x: the complex spectrum is then never written."""
        mag, ph = self.magnitude, self.phase
        if not (type(mag) is Magnitude and type(ph) is Phase and self.stack == -2 and mag.keep_nyquist
                and ph.keep_nyquist and not ph.unwrap):
            return False
        if not mag.can_fuse_with(stage, x) or stage._hop != 256:      # the STFT+Polar kernel is built for hop 256
            return False
        band = mag._banded()
        return band is not None and band.N == 513

    def forward_fused(self, stage, x: torch.Tensor):
        from ..utils.misc import reshape_batches
        mag, ph = self.magnitude, self.phase
        stage._follow(x)
        mag._follow(x)
        m_off, m_sc = mag._affine()
        p_off, p_sc = ph._affine(x)
        xb, batch_shape = reshape_batches(x, -1)
        stage._release_phase_source()
        y = ops.stft_polar_forward(xb, stage.window[:1024], mag._banded(), mag.contrast_mode, m_off, m_sc, mag._eps,
                                   p_off, p_sc)
        # the stage's phase buffer (keep_input inversion) is the phase half of the result, de-normalised on demand
        half = y[..., 1, :]
        stage._replace_phase_buffer(
            (lambda: ops.affine(half, p_off, p_sc, inverse=True)) if p_off is not None else (lambda: half.contiguous()))
        return y.reshape(batch_shape + y.shape[-3:])

    def forward(self, x: torch.Tensor) -> SpectralRepresentationType:
        fused = self._one_pass(x)
        if fused is not None:
            return fused
        magnitude = self.magnitude(x)
        phase = self.phase(x)
        if self.stack is not None:
            return torch.stack([magnitude, phase], dim=self.stack)
        return (magnitude, phase)

    def _split(self, x):
        if self.stack is None:
            return x[0], x[1]
        return x.select(self.stack, 0), x.select(self.stack, 1)

    # -- self-test hooks (the reference's test file drives every class through them) ------------------
    def test_forward(self, x: torch.Tensor, time: torch.Tensor = None):
        from ._selftest import stft_then
        return stft_then(self, x, time)

    def _round_trip(self, x: torch.Tensor):
        from ._selftest import rect_stft, rect_istft
        X, batch_shape = rect_stft(x)
        self.scale_data(X)
        return rect_istft(self.invert(self(X)), batch_shape)

    def test_inversion(self, x: torch.Tensor):
        return {"direct": self._round_trip(x)}

    @classmethod
    def test_scripted_transform(cls, transform, invert: bool = True):
        from ._selftest import random_spectrum
        X = random_spectrum("cuda")
        transform.scale_data(X)
        y = transform(X)
        if invert:
            transform.invert(y)

    def _one_pass_invert(self, x):
        """The mirror image of `_one_pass`: de-normalise, invert the contrast, project through the banded inverse
        bank and attach exp(i phase), reading the stacked tensor once."""
        mag, ph = self.magnitude, self.phase
        if not (type(mag) is Magnitude and type(ph) is Phase and self.stack == -2 and isinstance(x, torch.Tensor)
                and x.is_cuda and x.ndim >= 3 and x.shape[-2] == 2 and mag.mel and mag.keep_nyquist and ph.keep_nyquist):
            return None
        band = mag._band_of("inverse_mel_bank")
        if band is None or band.K != x.shape[-1] or band.N != x.shape[-1]:
            return None
        mag._follow(x)
        m_off, m_sc = mag._affine()
        p_off, p_sc = ph._affine(x)
        return ops.polar_inverse(x, band, mag.contrast_mode, m_off, m_sc, mag._eps, p_off, p_sc)

    def invert(self, x, inversion_mode: InversionEnumType = None, tolerance: float = 1.e-4) -> torch.Tensor:
        if type(self) is Polar:
            fused = self._one_pass_invert(x)
            if fused is not None:
                return fused
        mag, phase = self._split(x)
        mag = self.magnitude.invert(mag)
        phase = self.phase.invert(phase)
        return ops.polar_to_complex(mag, phase)


class Cartesian(SpectralRepresentation):
    def __repr__(self):
        return "Cartesian(real_norm=%s, imag_norm=%s)" % (self.magnitude.norm.mode, self.phase.norm.mode)

    def __init__(self, sr: int = 44100, real_args={"mode": "gaussian"}, imag_args={"mode": "gaussian"}, stack=-2,
                 keep_nyquist: bool = True):
        super().__init__(sr, Real, Imaginary, real_args, imag_args, stack=stack, keep_nyquist=keep_nyquist)

    def _one_pass_ok(self, x, stacked: bool) -> bool:
        """Both halves as they come (Real / Imaginary with or without Normalize), stacked on dim -2: one kernel reads
        the spectrum once and writes the stacked tensor (and the reverse)."""
        if not (type(self.magnitude) is Real and type(self.phase) is Imaginary and self.stack == -2
                and self.keep_nyquist and isinstance(x, torch.Tensor) and x.is_cuda):
            return False
        return (x.dtype == torch.float32 and x.ndim >= 2 and x.shape[-2] == 2) if stacked else (x.is_complex() and x.ndim >= 1)

    def forward(self, x: torch.Tensor) -> SpectralRepresentationType:
        if self._one_pass_ok(x, False):
            return ops.cartesian_forward(x, *self.magnitude._affine(x), *self.phase._affine(x))
        return super().forward(x)

    def invert(self, x, inversion_mode: InversionEnumType = None, tolerance: float = 1.e-4) -> torch.Tensor:
        if self._one_pass_ok(x, True):
            return ops.cartesian_inverse(x, *self.magnitude._affine(x), *self.phase._affine(x))
        real, imag = self._split(x)
        return torch.complex(self.magnitude.invert(real).contiguous(), self.phase.invert(imag).contiguous())


class Polar(SpectralRepresentation):
    def __repr__(self):
        return "Polar(real_norm=%s, imag_norm=%s)" % (self.magnitude.norm.mode, self.phase.norm.mode)

    def __init__(self, sr: int = 44100, magnitude_args={"mode": "bipolar"}, phase_args={"mode": "bipolar"}, stack=-2,
                 keep_nyquist: bool = True):
        super().__init__(sr, Magnitude, Phase, magnitude_args, phase_args, stack=stack, keep_nyquist=keep_nyquist)


class PolarIF(SpectralRepresentation):
    def __repr__(self):
        return "PolarIF(real_norm=%s, imag_norm=%s)" % (self.magnitude.norm.mode, self.phase.norm.mode)

    def __init__(self, sr: int = 44100, magnitude_args={"mode": "bipolar"}, phase_args={"mode": "bipolar"}, stack=-2,
                 keep_nyquist: bool = True):
        super().__init__(sr, Magnitude, IF, magnitude_args, phase_args, stack=stack, keep_nyquist=keep_nyquist)

    def test_inversion(self, x: torch.Tensor):
        outs = {}
        for method in self.phase.get_if_methods():
            self.phase.method = method
            outs[method] = self._round_trip(x)
        return outs

    def _in_place_parts(self, F: int, inverse: bool):
        """Banded bank of the magnitude half when both halves can work inside the stacked tensor: Magnitude over a
        banded bank of F filters, IF with one of the three methods, stacked on dim -2."""
        mag, ph = self.magnitude, self.phase
        if not (type(mag) is Magnitude and type(ph) is IF and self.stack == -2 and mag.mel and mag.keep_nyquist
                and ph.keep_nyquist and ph.method in ("forward", "backward", "central")):
            return None
        band = mag._band_of("inverse_mel_bank" if inverse else "mel_bank")
        if band is None or band.K != F or band.N != F:
            return None
        return band

    def forward(self, x: torch.Tensor) -> SpectralRepresentationType:
        if isinstance(x, torch.Tensor) and x.is_cuda and x.is_complex() and x.ndim >= 2 \
                and not (self.phase.method == "central" and x.size(-2) == 1):
            band = self._in_place_parts(x.shape[-1], False)
            if band is not None:
                mag, ph = self.magnitude, self.phase
                mag._follow(x)
                m_off, m_sc = mag._affine()
                p_off, p_sc = ph._affine(x)
                window = ph._get_weighted_window(x) if ph.weighted else None
                return ops.polarif_forward(x, band, mag.contrast_mode, m_off, m_sc, mag._eps, ph.method, window, p_off, p_sc)
        return super().forward(x)

    def invert(self, x, inversion_mode: InversionEnumType = None, tolerance: float = 1.e-4) -> torch.Tensor:
        if isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and x.ndim >= 3 and x.shape[-2] == 2:
            band = self._in_place_parts(x.shape[-1], True)
            if band is not None:
                mag, ph = self.magnitude, self.phase
                mag._follow(x)
                m_off, m_sc = mag._affine()
                p_off, p_sc = ph._affine(x)
                return ops.polarif_inverse(x, band, mag.contrast_mode, m_off, m_sc, mag._eps, ph.method, p_off, p_sc)
        return super().invert(x, inversion_mode, tolerance)
