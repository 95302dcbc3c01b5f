"""STFT / RealtimeSTFT on MI355X.

Drop-in for the reference's transforms/stft.py: same constructor signatures
(:32-38, :217), registered buffers (:41-48), `forward`/`invert`/
`forward_with_time`/`set_params`/`set_inversion_mode`/`realtime`/`ratio`, same
error types.  The arithmetic (framing, window, rFFT, irFFT, overlap-add,
polar->complex) runs in hand-written HIP kernels through the C ABI
(`acids_transforms_amd.ops`); Python only owns shapes and state.
"""
import math
from typing import Dict, Optional

import torch

from .. import ops
from ..utils.misc import frame, reshape_batches
from .base import AudioTransform, InversionEnumType

__all__ = ["STFT", "RealtimeSTFT"]

MAX_NFFT = 16384

def _as_clip_layout(x: torch.Tensor, n_fft: int):
    """Describe pre-framed input (..., n, n_fft) to the forward kernel without copying.

    Returns (tensor, B, T, clip_stride, hop, L).  A `frame()` view (strides
    (..., Lbuf, hop, 1)) is passed through as overlapping clips; anything else is
    made contiguous and treated as B*n one-frame clips.
    """
    if x.dim() >= 2 and x.stride(-1) == 1 and x.dim() >= 3:
        n, hop = x.shape[-2], x.stride(-2)
        lead_ok = True
        for i in range(x.dim() - 3):
            if x.stride(i) != x.stride(i + 1) * x.shape[i + 1]:
                lead_ok = False
        if lead_ok and 0 < hop < n_fft and x.stride(-3) >= (n - 1) * hop + n_fft:
            B = 1
            for d in x.shape[:-2]:
                B *= d
            return x, B, n, x.stride(-3), hop, (n - 1) * hop + n_fft
    xc = x.contiguous()
    B = xc.numel() // n_fft
    return xc, B, 1, n_fft, n_fft, n_fft


class STFT(AudioTransform):
    scriptable = False   # ctypes-backed (reference advertises True, stft.py:15-17)
    invertible = True
    needs_scaling = False

    def __repr__(self):
        return "STFT(n_fft=%d, hop_length=%d, inversion_mode = %s)" % (self._n_fft, self._hop, self.inversion_mode)

    def __init__(self, sr: int = 44100, n_fft: int = 1024, hop_length: int = 256, dtype: torch.dtype = None,
                 inversion_mode: str = "griffin_lim", window: str = "hann"):
        super().__init__(sr=sr)
        self._register_common_buffers(dtype)
        if hasattr(torch, "%s_window" % window):
            self.window_type = getattr(torch, "%s_window" % window)
        else:
            raise ValueError("Window %s is not known" % window)
        self._init_params(n_fft, hop_length, inversion_mode)

    # -- construction helpers shared with DGT ---------------------------------
    def _register_common_buffers(self, dtype):
        dtype = dtype or torch.get_default_dtype()
        if dtype != torch.float32:
            # The reference's `dtype` types its eps buffer and is meant for double-precision use (stft.py:36-47).  There
            # is no fp64 kernel behind this class: say so at construction rather than compute in fp32 under that name.
            raise ops.AcidsHipError("dtype=%s: the MI355X kernels compute in float32 / complex64 only (construct with "
                                    "dtype=None or torch.float32)" % str(dtype).replace("torch.", ""))
        self.register_buffer("n_fft", torch.zeros(1).long())
        self.register_buffer("hop_length", torch.zeros(1).long())
        self.register_buffer("window", torch.zeros(MAX_NFFT))
        self.register_buffer("inv_window", torch.zeros(MAX_NFFT))
        self.register_buffer("gamma", torch.zeros(1))
        self.register_buffer("eps", torch.tensor(torch.finfo(dtype).eps, dtype=dtype))
        self.register_buffer("phase_buffer", torch.zeros(0))
        self.register_buffer("_env16", torch.zeros(0), persistent=False)
        self._n_fft = 0
        self._hop = 0
        self._phase_src = None      # complex spectrum of the last forward (lazy phase_buffer)
        self.eager_phase = False    # True: compute angle() inside the forward kernel, like the reference

    def _init_params(self, n_fft, hop_length, inversion_mode):
        if n_fft is not None:
            assert hop_length is not None, "n_fft and hop_length must be given together"
        if hop_length is not None:
            assert n_fft is not None, "n_fft and hop_length must be given together"
        if (n_fft is not None) and (hop_length is not None):
            self.set_params(n_fft, hop_length)
        if inversion_mode in type(self).get_inversion_modes():
            self.inversion_mode = inversion_mode
        else:
            raise ValueError("Inversion mode %s not known" % inversion_mode)

    def set_params(self, n_fft: int, hop_length: int) -> None:
        n_fft, hop_length = int(n_fft), int(hop_length)
        self._n_fft, self._hop = n_fft, hop_length
        self.n_fft.fill_(n_fft)
        self.hop_length.fill_(hop_length)
        self.window.zero_()
        self.inv_window.zero_()
        dev = self.window.device
        self.window[:n_fft] = self._get_window().to(dev)
        self.inv_window[:n_fft] = self._get_dual_window().to(dev)
        self.gamma = self._get_gamma().to(dev)
        self._env16 = self._make_env16().to(dev)

    def _nfft_tensor(self):
        return torch.zeros(1).long().fill_(self._n_fft)

    def _get_gamma(self) -> torch.Tensor:
        # 2*pi*lambda^2 with lambda^2 = -N^2 / (8 ln 0.01), evaluated on an int64 tensor (stft.py:77-78)
        n = self._nfft_tensor()
        return 2 * torch.pi * ((-n ** 2 / (8 * math.log(0.01))) ** .5) ** 2

    def _get_window(self) -> torch.Tensor:
        return self.window_type(self._n_fft)

    def _get_dual_window(self) -> torch.Tensor:
        return self._get_window()

    def _make_env16(self) -> torch.Tensor:
        """Window-envelope table of the fused overlap-add kernel (n_fft = R * hop, R = 2, 4 or 8):
        env[mask][r] = sum_{q in mask} w[hop*(R-1-q)+r]^2 over the R frames that overlap a hop, oldest first.
        (The name dates from R = 4, the reference's default n_fft = 4 * hop: 16 masks.)"""
        n, h = self._n_fft, self._hop
        if n % h or n // h not in (2, 4, 8):
            return torch.zeros(0)
        R = n // h
        w2 = self.inv_window[:n].detach().cpu() ** 2
        env = torch.zeros(1 << R, h)
        for mask in range(1 << R):
            for q in range(R):
                if mask & (1 << q):
                    env[mask] = env[mask] + w2[h * (R - 1 - q):h * (R - q)]
        return env

    @property
    def ratio(self):
        return self._hop

    def set_inversion_mode(self, inversion_mode: str) -> None:
        if inversion_mode in self.get_inversion_modes():
            self.inversion_mode = inversion_mode
        else:
            raise AttributeError("inversion mode %s not valid" % inversion_mode)

    @staticmethod
    def get_inversion_modes():
        return ["griffin_lim", "keep_input", "random", "sinebank"]

    # -- device / state plumbing ----------------------------------------------
    def _hostf(self, name: str) -> float:
        """Python float of a scalar buffer (gamma, eps, tolerance) without a device sync per call:
        the value is re-read only when the buffer object or its version counter changed."""
        t = getattr(self, name)
        key = (t.data_ptr(), t._version)
        cache = self.__dict__.setdefault("_hostf_cache", {})
        hit = cache.get(name)
        if hit is None or hit[0] != key:
            hit = (key, float(t))
            cache[name] = hit
        return hit[1]

    def _follow(self, x: torch.Tensor):
        if self.window.device != x.device:
            self.to(x.device)

    def __getattr__(self, name):
        if name == "phase_buffer":
            self._materialise_phase()
        return super().__getattr__(name)

    def _materialise_phase(self):
        src = self.__dict__.get("_phase_src")
        if src is not None:
            self.__dict__["_phase_src"] = None
            # a complex spectrum, or a deferred computation (fused paths that never wrote the spectrum)
            self._buffers["phase_buffer"] = src() if callable(src) else ops.angle(src)

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        self._materialise_phase()
        return super()._save_to_state_dict(destination, prefix, keep_vars)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # phase_buffer changes shape with the data: accept whatever was saved
        key = prefix + "phase_buffer"
        if key in state_dict:
            self._buffers["phase_buffer"] = torch.zeros_like(state_dict[key])
        self.__dict__["_phase_src"] = None
        out = super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
        self._n_fft = int(self._buffers["n_fft"].item())
        self._hop = int(self._buffers["hop_length"].item())
        self._buffers["_env16"] = self._make_env16().to(self._buffers["window"].device)
        return out

    def _release_phase_source(self) -> None:
        """Called by every forward right before it allocates its output: the spectrum the LAST forward parked for the lazy
        phase_buffer is about to be superseded, so let go of it first.  Otherwise the new spectrum is allocated while the
        old one is still referenced: 2.9 GB held for nothing at the bench size, and the caching allocator hands out two
        blocks in turn -- the period-2 alternation of step times (the blocks are not equally fast: +2.4 % on the forward,
        tools/step_probe.py `ptrs` against `hold`, profiles/r05_power_clock.md)."""
        self.__dict__["_phase_src"] = None

    def _replace_phase_buffer(self, spectrum: Optional[torch.Tensor], phase: Optional[torch.Tensor] = None) -> None:
        if phase is not None:
            self.__dict__["_phase_src"] = None
            self._buffers["phase_buffer"] = phase
        else:
            self.__dict__["_phase_src"] = spectrum

    def _get_phase_buffer(self, mag: torch.Tensor) -> torch.Tensor:
        pb = self.phase_buffer
        if mag.shape[:-2] != pb.shape[:-2]:
            self._replace_phase_buffer(None, torch.tensor(0))
            return self._buffers["phase_buffer"]
        return pb

    # -- transform -------------------------------------------------------------
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        self._follow(x)
        x, batch_shape = reshape_batches(x, -1)
        window = self.window[:self._n_fft]
        self._release_phase_source()
        if self.eager_phase:
            x_fft, phase = ops.stft_forward(x, window, self._n_fft, self._hop, center=True, want_phase=True)
            self._replace_phase_buffer(None, phase)
        else:
            x_fft = ops.stft_forward(x, window, self._n_fft, self._hop, center=True)
            self._replace_phase_buffer(x_fft)
        return x_fft.reshape(batch_shape + x_fft.shape[-2:])

    def forward_with_time(self, x: torch.Tensor, time: torch.Tensor):
        transform = self.forward(x)
        n_chunks = transform.size(-2)
        shifts = torch.arange(n_chunks, device=time.device) * self._hop / self.sr
        return transform, shifts + time.unsqueeze(-1)

    def invert(self, x: torch.Tensor, inversion_mode: InversionEnumType = None, **kwargs) -> torch.Tensor:
        self._follow(x)
        x, batch_shape = reshape_batches(x, -2)
        if not torch.is_complex(x):
            x_inv = self.invert_without_phase(x, inversion_mode)
        else:
            x_inv = self._istft(x)
        return x_inv.reshape(batch_shape + x_inv.shape[-1:])

    def _check_nola(self, T: int) -> None:
        """torch.istft's window-envelope check (the reference calls torch.istft, stft.py:120-128): the overlap-added
        squared window over the T frames, n_fft/2 trimmed at both ends, must stay above 1e-11 -- otherwise the
        division by the envelope yields inf/NaN samples.  Host-side, cached per (window, T)."""
        n, h = self._n_fft, self._hop
        w = self.inv_window
        key = (w.data_ptr(), w._version, n, h, T)
        cache = self.__dict__.setdefault("_nola_cache", {})
        if key not in cache:
            if len(cache) > 64:
                cache.clear()
            w2 = (w[:n].detach().cpu().double()) ** 2
            expected = n + h * (T - 1)
            env = torch.zeros(expected, dtype=torch.float64)
            idx = (torch.arange(T).unsqueeze(1) * h + torch.arange(n).unsqueeze(0)).reshape(-1)
            env.index_add_(0, idx, w2.repeat(T))
            core = env[n // 2:n // 2 + h * (T - 1) + (n & 1)]      # torch.istft keeps one more sample when n_fft is odd
            cache[key] = float(core.abs().min()) if core.numel() else 1.0
        if cache[key] < 1e-11:
            raise RuntimeError("istft(n_fft=%d, hop_length=%d): window overlap add min: %g -- the window envelope "
                               "vanishes somewhere (NOLA violated), as torch.istft would report" % (n, h, cache[key]))

    def _istft(self, X=None, mag=None, phase=None):
        src = X if X is not None else mag
        self._check_nola(int(src.shape[-2]))
        env = self._env16 if self._env16.numel() else None
        return ops.istft(X, self.inv_window[:self._n_fft], self._n_fft, self._hop, env16=env, mag=mag, phase=phase)

    def realtime(self):
        mode = self.inversion_mode if self.inversion_mode in RealtimeSTFT.get_inversion_modes() else "random"
        return RealtimeSTFT(sr=self.sr, n_fft=self._n_fft, hop_length=self._hop, inversion_mode=mode)

    def invert_without_phase(self, x: torch.Tensor, inversion_mode: InversionEnumType = None) -> torch.Tensor:
        if inversion_mode is None:
            inversion_mode = self.inversion_mode
        if inversion_mode == "keep_input":
            phase = self._get_phase_buffer(x)
            if phase.shape[0] == 0:   # IndexError on a 0-dim buffer, exactly like the reference (stft.py:155)
                phase = torch.pi * 2 * torch.rand_like(x)
            return self._istft(mag=x, phase=phase)
        if inversion_mode == "random":
            phase = torch.pi * 2 * torch.rand_like(x)
            return self._istft(mag=x, phase=phase)
        if inversion_mode == "griffin_lim":
            return self.griffin_lim(x)
        if inversion_mode == "sinebank":
            return self.get_sinebank_inversion(x)
        raise ValueError("inversion mode %s not valid." % inversion_mode)

    def get_sinebank_inversion(self, x_fft: torch.Tensor, random_phase: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Oscillator-bank resynthesis (reference stft.py:180-191): one sinusoid per bin at the bin frequency,
        amplitude = the magnitude track linearly interpolated to the sample rate, random start phases; output
        length hop*T + n_fft, peak-normalised.  `random_phase` (F, 1) replaces the random draw (added, for tests)."""
        from .sinebank import sinebank_offline
        return sinebank_offline(x_fft, self.sr, self._n_fft, self._hop, random_phase)

    def griffin_lim(self, x: torch.Tensor, n_iter: int = 30, momentum: float = 0.99,
                    angles0: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Griffin-Lim inversion of a (B, T, F) magnitude spectrogram with the synthesis window, 30 iterations,
        momentum 0.99 and random initialisation -- the arguments the reference passes to
        torchaudio.functional.griffinlim (stft.py:174-178).  A composition of the ISTFT and STFT kernels plus
        one fused phase-update kernel per iteration.  `angles0` (complex, same shape) overrides the random start."""
        n, h = self._n_fft, self._hop
        window = self.inv_window[:n]
        m = momentum / (1 + momentum)
        if angles0 is None:
            angles0 = torch.rand(x.shape, dtype=torch.complex64, device=x.device)
        env = self._env16 if self._env16.numel() else None
        X = ops.scale_complex(x, angles0)
        tprev = None
        if n == 1024 and h in (128, 256, 512) and env is not None and x.dim() == 3:
            # every inverse after the first is istft(update(...)): one kernel, the updated spectrum is never written
            inverse = ops.istft(X, window, n, h, env16=env)
            del X
            for _ in range(n_iter):
                rebuilt = ops.stft_forward(inverse, window, n, h, center=True)
                inverse = ops.istft_griffinlim(x, rebuilt, tprev, m, window, n, h, env)
                tprev = rebuilt
            return inverse
        for _ in range(n_iter):
            inverse = ops.istft(X, window, n, h, env16=env)
            rebuilt = ops.stft_forward(inverse, window, n, h, center=True)
            X = ops.griffinlim_update(x, rebuilt, tprev, m)
            tprev = rebuilt
        return ops.istft(X, window, n, h, env16=env)

    # -- self tests (same hooks as the reference) -------------------------------
    def test_inversion(self, x: torch.Tensor) -> Dict[str, torch.Tensor]:
        outs = {}
        x_stft = self.forward(x)
        outs["direct"] = self.invert(x_stft)
        for inv_type in ("keep_input", "random"):
            outs[inv_type] = self.invert(x_stft.abs(), inversion_mode=inv_type)
        return outs


class RealtimeSTFT(STFT):
    """Per-frame transform on pre-framed input (..., n_fft) / (..., n, n_fft):
    rfft(x * window) and irfft(X) * inv_window (reference stft.py:215-310)."""

    def __init__(self, sr: int = 44100, n_fft: int = 1024, hop_length: int = 256, dtype: torch.dtype = None,
                 inversion_mode: InversionEnumType = "random", window: str = "hann", batch_size: int = 2):
        super().__init__(sr=sr, n_fft=n_fft, hop_length=hop_length, dtype=dtype, inversion_mode=inversion_mode,
                         window=window)
        self.batch_size = batch_size
        self.register_buffer("random_phase", 2 * torch.pi * torch.rand(int(self._n_fft / 2 + 1)))
        self.register_buffer("time_index", torch.tensor(0.))

    def __repr__(self):
        return "RealtimeSTFT(n_fft=%d, hop_length=%d, inversion_mode = %s)" % (self._n_fft, self._hop,
                                                                              self.inversion_mode)

    @staticmethod
    def get_inversion_modes():
        return ["keep_input", "random", "sinebank"]

    def reset(self, x=None):
        self.time_index = torch.tensor(0., device=self.window.device)
        self.__dict__.pop("_time_index_host", None)

    def get_batch_size(self, batch_size: int = None):
        return self.batch_size if batch_size is None else batch_size

    def set_batch_size(self, batch_size: int):
        self.batch_size = batch_size

    def _rt_forward(self, x: torch.Tensor) -> torch.Tensor:
        n = self._n_fft
        if x.shape[-1] != n:
            raise RuntimeError("RealtimeSTFT expects frames of n_fft=%d samples, got %d" % (n, x.shape[-1]))
        x = ops._f32c(x)          # raises on float64 (no silent narrowing); widens integer / half inputs
        out_shape = x.shape[:-1] + (n // 2 + 1,)
        xt, B, T, clip_stride, hop, L = _as_clip_layout(x, n)
        X = ops.stft_forward(xt, self.window[:n], n, hop, center=False, T=T, clip_stride=clip_stride, L=L, B=B)
        return X.reshape(out_shape)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        self._follow(x)
        self._release_phase_source()
        x_fft = self._rt_forward(x)
        self._replace_phase_buffer(x_fft)
        return x_fft

    def forward_with_time(self, x: torch.Tensor, time: torch.Tensor):
        return self(x), time

    def invert(self, x: torch.Tensor, inversion_mode: InversionEnumType = None, **kwargs) -> torch.Tensor:
        self._follow(x)
        if not torch.is_complex(x):
            return self.invert_without_phase(x, inversion_mode)
        return ops.irfft_frames(x, self.inv_window[:self._n_fft], self._n_fft)

    def invert_without_phase(self, x: torch.Tensor, inversion_mode: InversionEnumType = None) -> torch.Tensor:
        if inversion_mode is None:
            inversion_mode = self.inversion_mode
        if inversion_mode == "keep_input":
            phase = self._get_phase_buffer(x)
            if phase.shape[0] == 0:
                phase = torch.pi * 2 * torch.rand_like(x)
        elif inversion_mode == "random":
            phase = torch.pi * 2 * torch.rand_like(x)
        elif inversion_mode == "sinebank":
            return self.get_sinebank_inversion(x, windowed=True)
        else:
            raise ValueError("inversion mode %s not valid." % self.inversion_mode)
        return ops.irfft_frames(None, self.inv_window[:self._n_fft], self._n_fft, mag=x, phase=phase)

    def get_sinebank_inversion(self, x_fft: torch.Tensor, windowed: bool = False) -> torch.Tensor:
        """Per-chunk oscillator bank with a running clock and per-stream phases (reference stft.py:276-291):
        (..., n, F) magnitudes -> (..., n, n_fft) frames.  windowed=True (what `invert(mode="sinebank")` returns,
        stft.py:303-304): the frames times the synthesis window, in the same kernel."""
        from .sinebank import sinebank_realtime
        return sinebank_realtime(self, x_fft, self.inv_window[:self._n_fft] if windowed else None)

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
            self.reset()
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
        return self._stream_round_trip(x, ["sinebank"])      # the one mode the reference's own self-test runs

    @classmethod
    def test_scripted_transform(cls, transform, invert: bool = True):
        x = torch.zeros(2, transform._n_fft, device="cuda")
        X = transform(x)
        if invert:
            transform.invert(X)
            for mode in cls.get_inversion_modes():
                transform.invert(X.abs(), inversion_mode=mode)
