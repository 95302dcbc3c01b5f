"""`MFCC`: the reference class of that name is a mel *power spectrogram*
(torchaudio MelSpectrogram(sr, n_fft, hop_length, n_mels=128, power=2), no DCT,
not invertible -- reference transforms/mel.py:10-77).  Same here: periodic-Hann
STFT (center/reflect) -> |X|^power -> HTK filterbank -> (..., n_mels, T), by the
STFT kernel plus the MFMA projection with a channel-major store.

`n_mfcc` (build extension, default None = reference behaviour): when set, a
DCT-II (ortho) of 10*log10(mel power) is applied on top, giving (..., n_mfcc, T)
-- the "MFCC(40)" BASELINE config 4 names; it has no counterpart in the
reference and its parity is checked against scipy.fft.dct only.
"""
import math
from typing import Union

import torch

from .. import ops
from ..utils.banded import BandedBank
from ..utils.melbank import melscale_fbanks
from ..utils.misc import reshape_batches
from .base import AudioTransform, NotInvertibleError
from .norm import Normalize

__all__ = ["MFCC"]


class MFCC(AudioTransform):
    invertible = False
    scriptable = False

    @property
    def needs_scaling(self):
        return self.norm is not None

    def __repr__(self):
        s = "MFCC(n_fft=%s, hop_length=%spower=%s, n_mels=%s" % (self.n_fft, self.hop_length, self.power, self.n_mels)
        if self.norm is not None:
            s += ", f%s" % self.norm
        return s + ")"

    def __init__(self, n_fft: int = 1024, hop_length=256, power: float = 2., n_mels: int = 128, sr=44100,
                 norm_mode: str = None, n_mfcc: int = None):
        super().__init__(sr=sr)
        self.norm: Union[None, Normalize] = None
        if norm_mode is not None:
            self.norm = Normalize(mode=norm_mode)
        self.n_mfcc = n_mfcc
        self.set_transform(n_fft, n_mels, hop_length, power)

    def set_transform(self, n_fft, n_mels, hop_length, power):
        if power not in (1, 1.0, 2, 2.0):
            raise NotImplementedError("MFCC: only power=1 and power=2 are implemented on the HIP path")
        self.n_fft, self.hop_length, self.power, self.n_mels = n_fft, hop_length, power, n_mels
        dev = self._buffers["fbank"].device if "fbank" in self._buffers else None
        window = torch.zeros(n_fft)
        window[:] = torch.hann_window(n_fft)
        fb = melscale_fbanks(n_fft // 2 + 1, 0.0, float(self.sr // 2), n_mels, self.sr)
        self.register_buffer("window", window.to(dev) if dev else window, persistent=False)
        self.register_buffer("fbank", fb.to(dev) if dev else fb, persistent=False)
        self._band = BandedBank(fb)
        if self.n_mfcc is not None:
            n = torch.arange(float(n_mels))
            k = torch.arange(float(self.n_mfcc)).unsqueeze(1)
            dct = torch.cos(math.pi / float(n_mels) * (n + 0.5) * k)
            dct[0] *= 1.0 / math.sqrt(2.0)
            dct *= math.sqrt(2.0 / float(n_mels))
            self.register_buffer("dct", (dct.t().contiguous() * (10.0 / math.log(10.0))).to(dev) if dev
                                 else dct.t().contiguous() * (10.0 / math.log(10.0)), persistent=False)

    def _follow(self, x):
        if self.fbank.device != x.device:
            self.to(x.device)

    @property
    def ratio(self):
        return self.hop_length

    def forward_with_time(self, x: torch.Tensor, time: torch.Tensor):
        transform = self.forward(x)
        n_chunks = transform.size(-2)     # = n_mels: the reference's own accounting (mel.py:49), kept as is
        shifts = torch.arange(n_chunks, device=time.device) * self.hop_length / self.sr
        return transform, shifts + time.unsqueeze(-1)

    def scale_data(self, x: torch.Tensor):
        if self.norm is not None:
            self.norm.scale_data(x)       # statistics of the *input*, as the reference does (mel.py:60-62)

    def forward(self, x: torch.Tensor):
        self._follow(x)
        xb, batch_shape = reshape_batches(x, -1)
        fusable = (self.n_fft == 1024 and self.hop_length == 256 and self._band.fusable
                   and xb.dtype == torch.float32 and xb.shape[-1] > 512 and not (xb.shape[-1] & 1))
        if fusable and self.n_mfcc is None:
            # one kernel, audio -> mel power: the spectrum never goes to HBM
            off = sc = None
            if self.norm is not None:
                off, sc = self.norm._params(x)
            _, _, mel = ops.stft_mel_forward(xb, self.window, self._band, None, off, sc, power=int(self.power),
                                             want_spectrum=False, channel_major=True)
            return mel.reshape(batch_shape + mel.shape[-2:])
        if fusable:
            # extension (n_mfcc): the same kernel emits log(mel power), the DCT is a small second contraction
            _, _, logmel = ops.stft_mel_forward(xb, self.window, self._band, "log", None, None, eps=1e-10,
                                                power=int(self.power), want_spectrum=False)
            off = sc = None
            if self.norm is not None:
                off, sc = self.norm._params(x)
            out = ops.mel_forward_real(logmel, self.dct, off, sc, channel_major_T=logmel.shape[-2])
            return out.reshape(batch_shape + out.shape[-2:])
        if (((self.n_fft == 2048 and self._band.fusable2048) or (self.n_fft == 512 and self._band.fusable512))
                and xb.dtype == torch.float32 and xb.shape[-1] > self.n_fft // 2):
            # the same single kernel for torchaudio's / librosa's usual n_fft = 2048 and for 512 (any hop): features only
            if self.n_mfcc is None:
                off = sc = None
                if self.norm is not None:
                    off, sc = self.norm._params(x)
                _, _, mel = ops.stft_mel_forward(xb, self.window, self._band, None, off, sc, power=int(self.power),
                                                 want_spectrum=False, channel_major=True, hop=self.hop_length, n_fft=self.n_fft)
                return mel.reshape(batch_shape + mel.shape[-2:])
            _, _, logmel = ops.stft_mel_forward(xb, self.window, self._band, "log", None, None, eps=1e-10,
                                                power=int(self.power), want_spectrum=False, hop=self.hop_length, n_fft=self.n_fft)
            off = sc = None
            if self.norm is not None:
                off, sc = self.norm._params(x)
            out = ops.mel_forward_real(logmel, self.dct, off, sc, channel_major_T=logmel.shape[-2])
            return out.reshape(batch_shape + out.shape[-2:])
        X = ops.stft_forward(xb, self.window, self.n_fft, self.hop_length, center=True)     # (B, T, F)
        T = X.shape[-2]
        off = sc = None
        if self.norm is not None and self.n_mfcc is None:
            off, sc = self.norm._params(x)
        if self.n_mfcc is None:
            mel = ops.mel_forward(X, self.fbank, None, off, sc, power=int(self.power), channel_major_T=T,
                                  band=self._band)
            return mel.reshape(batch_shape + mel.shape[-2:])
        # extension: MFCC = DCT-II(10 log10(mel power)); ln -> dB factor is folded into the DCT matrix
        logmel = ops.mel_forward(X, self.fbank, "log", None, None, eps=1e-10, power=int(self.power),
                                 band=self._band)                                               # (B, T, n_mels)
        if self.norm is not None:
            off, sc = self.norm._params(x)
        out = ops.mel_forward_real(logmel, self.dct, off, sc, channel_major_T=T)
        return out.reshape(batch_shape + out.shape[-2:])

    def invert(self, x: torch.Tensor, inversion_mode=None, **kwargs):
        raise NotInvertibleError
