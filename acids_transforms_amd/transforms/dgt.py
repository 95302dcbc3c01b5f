"""DGT / RealtimeDGT on MI355X: Gaussian-window Gabor transform with PGHI
(phase-gradient heap integration) magnitude-to-phase reconstruction.

Drop-in for the reference's transforms/dgt.py (DGT :24-236, RealtimeDGT
:239-519): same constructors, buffers (`tolerance`, `hgi_mag_buffer`,
`hgi_phase_buffer`, ...), inversion modes and errors.  Forward / inverse reuse
the STFT kernels with the Gaussian analysis window (:108-112) and its
canonical dual (:114-123); PGHI runs in pghi.hip with the reference's exact
binary-heap order (utils/heapq.py).
"""
import math
from enum import Enum
from typing import Dict, List, Union

import torch

from .. import ops
from ..utils.misc import frame, reshape_batches
from .base import AudioTransform, InversionEnumType
from .stft import RealtimeSTFT, STFT

__all__ = ["DGT", "RealtimeDGT", "DGT_INVERSION_MODES"]


class DGT_INVERSION_MODES(Enum):
    KEEP_INPUT = 0
    GRIFFIN_LIM = 1
    PGHI = 2
    RANDOM = 3


class DGT(STFT):
    def __repr__(self):
        return "DGT(n_fft=%d, hop_length=%d, inversion_mode = %s)" % (self._n_fft, self._hop, self.inversion_mode)

    def __init__(self, sr: int = 44100, n_fft: int = 1024, hop_length: int = 256, dtype: torch.dtype = None,
                 inversion_mode: InversionEnumType = "pghi", tolerance: float = 1.e-2):
        AudioTransform.__init__(self, sr)
        self._register_common_buffers(dtype)
        self.register_buffer("tolerance", torch.tensor(tolerance))
        self._init_params(n_fft, hop_length, inversion_mode)

    @staticmethod
    def get_inversion_modes():
        return ["pghi", "griffin_lim", "random", "keep_input", "sinebank"]

    def realtime(self):
        mode = self.inversion_mode if self.inversion_mode in RealtimeDGT.get_inversion_modes() else "pghi"
        return RealtimeDGT(sr=self.sr, n_fft=self._n_fft, hop_length=self._hop, inversion_mode=mode)

    def _lambda(self):
        # Gaussian width such that the window falls to 1 % at its edges
        return (-self._nfft_tensor() ** 2 / (8 * math.log(0.01))) ** .5

    def _get_window(self) -> torch.Tensor:
        n = self._n_fft
        # sample exp(-t^2 / (2 (2 lambda)^2)) on the 2n+1 half-steps around the centre, keep the odd ones
        t = torch.arange(0, 2 * n + 1) - (2 * n) / 2
        w = torch.exp(-t ** 2 / (2 * (self._lambda() * 2) ** 2))
        return w[1:2 * n + 1:2]

    def _get_dual_window(self) -> torch.Tensor:
        # canonical dual for a painless frame: g[l] / sum_n g[l - n*hop]^2 (terms added for ascending n)
        n, h = self._n_fft, self._hop
        g = self.window[:n].detach().cpu()
        g2 = g * g
        denom = torch.zeros(n)
        for k in range(-(n // h), n // h + 1):
            shift = k * h                      # term g2[l - shift], valid where 0 <= l - shift < n
            lo, hi = max(0, shift), min(n, n + shift)
            if lo < hi:
                denom[lo:hi] = denom[lo:hi] + g2[lo - shift:hi - shift]
        return g / denom

    # -- magnitude-only inversion ------------------------------------------------
    def invert_without_phase(self, x: torch.Tensor, inversion_mode: InversionEnumType = None) -> torch.Tensor:
        if inversion_mode is None:
            inversion_mode = self.inversion_mode
        if inversion_mode == "keep_input":
            phase = self._get_phase_buffer(x)
            if phase.shape[0] == 0:
                phase = torch.pi * 2 * torch.rand_like(x)
        elif inversion_mode == "pghi":
            phase = self.pghi(x, self.tolerance)
        elif inversion_mode == "random":
            phase = torch.pi * 2 * torch.rand_like(x)
        elif inversion_mode == "griffin_lim":
            return self.griffin_lim(x)
        elif inversion_mode == "sinebank":
            return self.get_sinebank_inversion(x)
        else:
            raise ValueError("inversion mode %s not valid." % self.inversion_mode)
        return self._istft(mag=x, phase=phase)

    def pghi(self, mag: torch.Tensor, tolerance=1.e-4) -> torch.Tensor:
        """Phase for a (T, F) or (B, T, F) magnitude array (reference dgt.py:156-162,
        applied clip by clip as at :137-141; `invert` passes the module's own tolerance, a direct call defaults to
        the reference's 1e-4).  The caller's tensor is not modified."""
        self._follow(mag)
        tol = self._hostf("tolerance") if tolerance is self.tolerance else float(tolerance)
        squeeze = mag.dim() == 2
        m = mag.unsqueeze(0) if squeeze else mag
        phase = ops.pghi_offline(m, self._hostf("gamma"), self._n_fft, self._hop, tol, self._hostf("eps"))
        return phase[0] if squeeze else phase

    def perform_hgi(self, X: torch.Tensor, tgradw: torch.Tensor, fgradw: torch.Tensor, abstol: float = 1e-7,
                    tol: float = 1.e-2) -> torch.Tensor:
        """The heap integration on its own (reference dgt.py:168-220): phase of a (T, F) -- or (B, T, F) -- magnitude
        array along the given gradients, `pghi(mag)` being `perform_hgi(clamp(mag, eps), *modgabphasegrad(mag), eps,
        tolerance)`.  Unlike the reference's method this one does not overwrite X."""
        self._follow(X)
        squeeze = X.dim() == 2
        args = [t.unsqueeze(0) if squeeze else t for t in (X, tgradw, fgradw)]
        phase = ops.pghi_integrate(args[0], args[1], args[2], float(tol), float(abstol))
        return phase[0] if squeeze else phase

    def modgabphasegrad(self, mag: torch.Tensor):
        """(tgradw, fgradw) of a clamped (T, F) magnitude array (reference dgt.py:222-236)."""
        self._follow(mag)
        m = mag.unsqueeze(0) if mag.dim() == 2 else mag
        tg, fg = ops.pghi_gradients(m, self._hostf("gamma"), self._n_fft, self._hop, self._hostf("eps"))
        return (tg[0], fg[0]) if mag.dim() == 2 else (tg, fg)

    def test_inversion(self, x: torch.Tensor) -> Dict[str, torch.Tensor]:
        outs = {}
        x_dgt = self.forward(x)
        outs["direct"] = self.invert(x_dgt)
        for inv_type in ("pghi", "keep_input", "random"):
            outs[inv_type] = self.invert(x_dgt.abs(), inversion_mode=inv_type)
        return outs


class RealtimeDGT(DGT):
    def __init__(self, sr: int = 44100, n_fft=1024, hop_length=256, dtype=None,
                 batch_size: Union[int, List[int]] = 2, inversion_mode: InversionEnumType = "pghi"):
        super().__init__(sr=sr, n_fft=n_fft, hop_length=hop_length, dtype=dtype, inversion_mode=inversion_mode)
        self.batch_size = [batch_size] if isinstance(batch_size, int) else batch_size
        self.register_buffer("hgi_mag_buffer", torch.zeros(*self.batch_size, 2, n_fft // 2 + 1))
        self.register_buffer("hgi_phase_buffer", torch.zeros(*self.batch_size, n_fft // 2 + 1))
        self.register_buffer("random_phase", 2 * torch.pi * torch.rand(int(self._n_fft / 2 + 1)))
        self.register_buffer("time_index", torch.tensor(0.))

    def __repr__(self):
        return "RealtimeDGT(n_fft=%d, hop_length=%d, inversion_mode = %s)" % (self._n_fft, self._hop,
                                                                             self.inversion_mode)

    @staticmethod
    def get_inversion_modes():
        return ["random", "pghi", "keep_input", "sinebank"]

    def get_batch_size(self) -> List[int]:
        return [int(b) for b in self.batch_size]

    def set_batch_size(self, batch_size: Union[int, List[int]]):
        self.reset(batch_size)

    def batch_size(self) -> List[int]:
        """The reference declares this method (dgt.py:271-273) and then shadows it on every instance with the attribute
        of the same name set in __init__ / reset(); kept for the class's surface."""
        return self.hgi_mag_buffer.shape[:-2]

    def reset(self, batch_size: Union[int, List[int]]) -> None:
        self.batch_size = [batch_size] if isinstance(batch_size, int) else torch.Size(batch_size)
        dev = self.window.device
        F = self._n_fft // 2 + 1
        self.hgi_mag_buffer = torch.zeros(torch.Size(self.batch_size) + torch.Size([2, F]), device=dev)
        self.hgi_phase_buffer = torch.zeros(torch.Size(self.batch_size) + torch.Size([F]), device=dev)

    def update_buffers(self, x: torch.Tensor) -> None:
        """Carry the last two magnitude rows and the last phase row of a complex chunk (..., n, F) into the PGHI state
        (reference dgt.py:330-336)."""
        self._follow(x)
        if x.shape[-2] > 1:
            self.hgi_mag_buffer = x[..., -2:, :].abs()
        else:
            self.hgi_mag_buffer = torch.stack([self.hgi_mag_buffer[..., 1, :], x[..., -1, :].abs()], -2)
        self.hgi_phase_buffer = ops.angle(x[..., -1, :].contiguous())

    def _get_gamma(self) -> torch.Tensor:
        # the streaming variant keeps lambda itself (reference dgt.py:373-374)
        return self._lambda()

    _rt_forward = RealtimeSTFT._rt_forward

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        self._follow(x)
        self._release_phase_source()
        x_dgt = self._rt_forward(x)
        self._replace_phase_buffer(x_dgt)
        return x_dgt

    def forward_with_time(self, x: torch.Tensor, time: torch.Tensor):
        return self(x), time

    def invert(self, x: torch.Tensor, inversion_mode: InversionEnumType = None, **kwargs) -> torch.Tensor:
        self._follow(x)
        if not torch.is_complex(x):
            return self.invert_without_phase(x, inversion_mode)
        return ops.irfft_frames(x, self.inv_window[:self._n_fft], self._n_fft)

    def invert_without_phase(self, x: torch.Tensor, inversion_mode: InversionEnumType = None) -> torch.Tensor:
        batch_size = x.shape[:-2]
        if batch_size != self.batch_size:   # a list never equals a torch.Size: first call always resets (dgt.py:305-307)
            self.reset(batch_size)
        if inversion_mode is None:
            inversion_mode = self.inversion_mode
        if inversion_mode == "keep_input":
            phase = self._get_phase_buffer(x)
            if phase.shape[0] == 0:
                phase = torch.pi * 2 * torch.rand_like(x)
        elif inversion_mode == "pghi":
            phase = self.pghi(x, self.tolerance)
        elif inversion_mode == "random":
            phase = torch.pi * 2 * torch.rand_like(x)
        elif inversion_mode == "sinebank":
            return self.get_sinebank_inversion(x, windowed=True)
        else:
            raise ValueError("inversion mode %s not valid." % self.inversion_mode)
        n = self._n_fft
        if phase.shape != x.shape:
            phase = phase.expand_as(x)
        # one fused kernel: polar->complex, irfft, window; also refreshes the PGHI history
        # (|x[-2:]| and angle(x[-1]), reference dgt.py:330-336)
        frames, self.hgi_mag_buffer, self.hgi_phase_buffer = ops.rt_polar_irfft_update(
            x, phase, self.inv_window[:n], n, self.hgi_mag_buffer)
        return frames

    def get_sinebank_inversion(self, x_fft: torch.Tensor, windowed: bool = False) -> torch.Tensor:
        """Per-chunk oscillator bank (reference dgt.py:356-371): (..., n, F) -> (..., n, n_fft) frames; windowed=True
        multiplies by the dual window as `invert(mode="sinebank")` does (dgt.py:321-322)."""
        from .sinebank import sinebank_realtime
        return sinebank_realtime(self, x_fft, self.inv_window[:self._n_fft] if windowed else None)

    def pghi(self, mag: torch.Tensor, tolerance=1e-6, noise: torch.Tensor = None):
        """Streaming PGHI (reference dgt.py:338-354, 378-466) for (..., n, F) magnitudes using the
        two-frame magnitude history and the previous phase.  `noise` (same shape as mag)
        overrides the standard-normal draws used for bins at or below the tolerance."""
        self._follow(mag)
        tol = self._hostf("tolerance") if tolerance is self.tolerance else float(tolerance)   # direct calls: 1e-6 (dgt.py:338)
        m, batch_shape = reshape_batches(mag, -2)
        hist, _ = reshape_batches(self.hgi_mag_buffer, -2)
        prev, _ = reshape_batches(self.hgi_phase_buffer, -1)
        if noise is None:
            noise = torch.randn_like(m)
        else:
            noise, _ = reshape_batches(noise, -2)
        phase = ops.pghi_realtime(hist, m, prev, noise, self._hostf("gamma"), self._n_fft, self._hop, tol,
                                  self._hostf("eps"))
        return phase.reshape(batch_shape + phase.shape[1:])

    def test_forward(self, x: torch.Tensor, time: torch.Tensor = None):
        out = self(frame(x, self._n_fft, self._hop, -1))
        return out if time is None else (out, None)

    def _stream_round_trip(self, x: torch.Tensor, modes):
        """The reference's streaming self-test (stft.py:324-351, dgt.py:480-509): the audio in chunks of 4 n_fft samples
        through OverlapAdd -> this transform -> invert -> OverlapAdd.invert, once with the complex frames ("direct") and
        once per spectrogram inversion mode from the magnitudes alone."""
        from .oadd import OverlapAdd
        n, h = self._n_fft, self._hop
        outs = {}
        for mode in [None] + list(modes):
            self.reset(list(x.shape[:-1]))
            oadd = OverlapAdd(n, h).to(x.device)
            if mode is not None:
                self.inversion_mode = mode
            pieces = []
            for chunk in x.split(4 * n, -1):
                X = self(oadd(chunk))
                frames = self.invert(X) if mode is None else self.invert(X.abs(), inversion_mode=mode)
                pieces.append(oadd.invert(frames))
            outs["direct" if mode is None else mode] = torch.cat(pieces, -1)
        return outs

    def test_inversion(self, x: torch.Tensor) -> Dict[str, torch.Tensor]:
        return self._stream_round_trip(x, self.get_inversion_modes())

    @classmethod
    def test_scripted_transform(cls, transform, invert: bool = True):
        x = torch.zeros(2, 1, transform._n_fft, device="cuda")
        transform.reset(list(x.shape[:-1]))
        X = transform(x)
        if invert:
            transform.invert(X)
            for mode in cls.get_inversion_modes():
                transform.invert(X.abs(), inversion_mode=mode)
