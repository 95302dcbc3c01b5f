"""Magnitude representation: |X| -> (mel filterbank) -> contrast -> Normalize.

Drop-in for the reference's `Magnitude` (transforms/spectral_repr.py:143-258):
same constructor, buffers (`eps`, `mel_bank` (1,F,N), `inverse_mel_bank`
(1,N,F), `norm.offset`, `norm.scale`) and quirks (scale_data ignores the mel
projection, :242-245; inputs of any rank; keep_nyquist=False drops output bin 0
and pads at the end).  `n_mels` is an added optional argument: the reference
always builds an n_bins x n_bins bank (:173-178), which stays the default.
forward / invert are one fused MFMA kernel each (mel.hip).
"""
from typing import Union

import torch

from .. import ops
from ..utils.banded import BandedBank
from ..utils.melbank import melscale_fbanks
from .base import AudioTransform, InversionEnumType
from .norm import Normalize, stats_to_affine

__all__ = ["Magnitude"]

ContrastModeType = Union[None, str]


class Dummy(AudioTransform):
    """The no-op stage the reference puts where a representation has no normalisation (spectral_repr.py:17-18,
    25-26); `mode` is what the `__repr__`s of the representations print for it."""
    mode = None


_Identity = Dummy


class Magnitude(AudioTransform):
    scriptable = False
    invertible = True
    needs_scaling = True

    def __repr__(self):
        if self.mel:
            return "Magnitude(mel=%s, n_fft=%s, norm=%s)" % (self.mel, self.n_fft, self.norm.mode)
        return "Magnitude(norm=%s)" % self.norm.mode

    def __init__(self, sr: int = 44100, mode: Union[str, None] = "unipolar", contrast: ContrastModeType = "log1p",
                 mel: bool = True, n_fft: int = 1024, dtype: torch.dtype = None, eps: float = None,
                 keep_nyquist: bool = True, n_mels: int = None, bank_dtype: str = "fp32"):
        super().__init__(sr=sr)
        if bank_dtype not in ("fp32", "bf16"):
            raise ValueError("bank_dtype must be 'fp32' or 'bf16', got %r" % (bank_dtype,))
        # "bf16" (added option; BASELINE config 5): forward projection as a dense bf16 MFMA GEMM with fp32
        # accumulation -- ~4e-3 relative to the fp32 chain, so never the default; invert stays fp32
        self.bank_dtype = bank_dtype
        self.norm = _Identity() if (mode is None or mode == "none") else Normalize(mode)
        self.contrast_mode = contrast
        self.mel = mel
        self.n_fft = n_fft
        if dtype is None:
            dtype = torch.get_default_dtype()
        if dtype != torch.float32:
            raise ops.AcidsHipError("dtype=%s: the MI355X kernels compute in float32 / complex64 only (construct with "
                                    "dtype=None or torch.float32)" % str(dtype).replace("torch.", ""))
        if eps is None:
            eps = torch.finfo(dtype).eps
        self.register_buffer("eps", torch.tensor(eps))
        self._eps = float(eps)
        self.keep_nyquist = keep_nyquist
        assert sr is not None
        assert n_fft is not None
        n_bins = n_fft // 2 + 1
        fft_scale = torch.arange(n_bins) / n_fft * sr
        if not self.keep_nyquist:
            fft_scale = fft_scale[..., 1:]
        bank = melscale_fbanks(n_bins, fft_scale[0], fft_scale[-1], n_bins if n_mels is None else int(n_mels), sr)
        self._set_bank(bank)

    def _set_bank(self, bank: torch.Tensor) -> None:
        """Forward bank: every filter (column) scaled to unit sum; inverse bank: every
        frequency row scaled to unit sum, transposed.  All-zero columns/rows are left alone."""
        col = bank.sum(0)
        fwd = bank / torch.where(col != 0, col, torch.ones_like(col)).unsqueeze(0)
        row = bank.sum(1)
        inv = bank / torch.where(row != 0, row, torch.ones_like(row)).unsqueeze(1)
        self.register_buffer("mel_bank", fwd.unsqueeze(0))
        self.register_buffer("inverse_mel_bank", inv.transpose(-2, -1).unsqueeze(0))

    # ------------------------------------------------------------------
    def _follow(self, x):
        if self.eps.device != x.device:
            self.to(x.device)

    def _affine(self):
        if isinstance(self.norm, Normalize):
            if self.norm.offset.numel() != 1:
                raise RuntimeError("Magnitude used before scale_data(): norm.offset has shape %s"
                                   % (tuple(self.norm.offset.shape),))
            return self.norm.offset, self.norm.scale
        return None, None

    # -- fusion with the preceding STFT / DGT stage ---------------------------------------------
    def _band_of(self, name):
        """Banded walk tables of the buffer `name` (mel_bank / inverse_mel_bank), or None when that bank is not
        banded enough; rebuilt when the buffer changes."""
        bank = getattr(self, name)
        key = (bank.data_ptr(), bank._version)
        cache = self.__dict__.setdefault("_band_cache", {})
        if name not in cache or cache[name][0] != key:
            cache[name] = (key, BandedBank(bank))
        return cache[name][1] if cache[name][1].eligible else None

    def _bf16_image(self):
        """bf16 operand image of `mel_bank` for the MFMA projection; rebuilt when the buffer changes."""
        bank = self.mel_bank
        key = (bank.data_ptr(), bank._version, bank.device)
        hit = self.__dict__.get("_bf16_cache")
        if hit is None or hit[0] != key:
            hit = (key, ops.mel_bf16_pack_bank(bank))
            self.__dict__["_bf16_cache"] = hit
        return hit[1]

    def _banded(self):
        """Banded form of `mel_bank` for the fused forward kernel (None when the bank is not banded
        enough, or when this module's options rule the fusion out)."""
        if not self.mel or not self.keep_nyquist or self.bank_dtype != "fp32":
            return None
        band = self._band_of("mel_bank")
        return band if band is not None and band.fusable else None

    def can_fuse_with(self, stage, x: torch.Tensor) -> bool:
        """True when `stage` (an offline STFT/DGT with n_fft=1024 and hop 256, 128 or 512) followed by this module can
        run as the single fused kernel on input x."""
        from .stft import STFT, RealtimeSTFT
        from .dgt import RealtimeDGT
        if not isinstance(stage, STFT) or isinstance(stage, (RealtimeSTFT, RealtimeDGT)):
            return False
        if stage._n_fft != 1024 or stage._hop not in (128, 256, 512) or not x.is_cuda or x.dtype != torch.float32:
            return False
        if x.shape[-1] <= 512 or (x.shape[-1] & 1):
            return False
        ops.contrast_code(self.contrast_mode)
        return self._banded() is not None

    def forward_fused(self, stage, x: torch.Tensor, return_spectrum: bool = False):
        """self(stage(x)) in one kernel; `stage` keeps its phase-buffer side effect.
        return_spectrum=True also hands back stage(x) (it is written anyway): (X, features)."""
        from ..utils.misc import reshape_batches
        stage._follow(x)
        self._follow(x)
        off, sc = self._affine()
        xb, batch_shape = reshape_batches(x, -1)
        stage._release_phase_source()
        X, phase, feat = ops.stft_mel_forward(xb, stage.window[:1024], self._banded(), self.contrast_mode, off, sc,
                                              self._eps, want_phase=stage.eager_phase, hop=stage._hop)
        stage._replace_phase_buffer(X, phase)
        feat = feat.reshape(batch_shape + feat.shape[-2:])
        if return_spectrum:
            return X.reshape(batch_shape + X.shape[-2:]), feat
        return feat

    # -- self-test hooks (the reference's test file drives every class through them) ------------------
    def test_forward(self, x: torch.Tensor, time: torch.Tensor = None):
        from ._selftest import stft_then
        return stft_then(self, x, time)

    def test_inversion(self, x: torch.Tensor):
        """|X| through the representation and back, the phase kept aside, audio by the window-less ISTFT."""
        from ._selftest import rect_stft, rect_istft
        X, batch_shape = rect_stft(x)
        self.scale_data(X)
        mag = self.invert(self(X))
        return {"direct": rect_istft(ops.polar_to_complex(mag, X.angle()), batch_shape)}

    @classmethod
    def test_scripted_transform(cls, transform, invert: bool = True):
        from ._selftest import random_spectrum
        X = random_spectrum("cuda")
        transform.scale_data(X)
        y = transform(X)
        if invert:
            transform.invert(y)

    def contrast(self, mag: torch.Tensor) -> torch.Tensor:
        ops.contrast_code(self.contrast_mode)   # TypeError on unknown modes, like the reference
        return ops.mag_pointwise(mag, self.contrast_mode, eps=self._eps)

    def invert_contrast(self, mag: torch.Tensor) -> torch.Tensor:
        ops.contrast_code(self.contrast_mode)
        return ops.mag_pointwise(mag, self.contrast_mode, eps=self._eps, inverse=True)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        self._follow(x)
        off, sc = self._affine()
        if self.mel and self.bank_dtype == "bf16":
            K, N = self.mel_bank.shape[-2], self.mel_bank.shape[-1]
            mag = ops.mel_forward_bf16(x, self._bf16_image(), K, N, self.contrast_mode, off, sc, self._eps)
        elif self.mel:
            mag = ops.mel_forward(x, self.mel_bank, self.contrast_mode, off, sc, self._eps,
                                  band=self._band_of("mel_bank"))
        else:
            mag = ops.mag_pointwise(x, self.contrast_mode, off, sc, self._eps)
        if not self.keep_nyquist:
            mag = mag[..., 1:]
        return mag

    def invert(self, x: torch.Tensor, inversion_mode: InversionEnumType = None, tolerance: float = 1.e-4) -> torch.Tensor:
        self._follow(x)
        off, sc = self._affine()
        if not self.keep_nyquist:
            # reference order: Normalize.invert, zero-pad the LAST bin, then invert_contrast (+ bank)
            if off is not None:
                x = ops.affine(x, off, sc, inverse=True)
            x = torch.cat([x, torch.zeros(x.shape[:-1] + (1,), device=x.device, dtype=x.dtype)], -1)
            off = sc = None
        if self.mel:
            return ops.mel_inverse(x, self.inverse_mel_bank, self.contrast_mode, off, sc, self._eps,
                                   band=self._band_of("inverse_mel_bank"))
        return ops.mag_pointwise(x, self.contrast_mode, off, sc, self._eps, inverse=True)

    def scale_data(self, x: torch.Tensor) -> None:
        """Statistics of contrast(|x|) -- the mel projection is NOT applied (reference :242-245)."""
        self._follow(x)
        if isinstance(self.norm, Normalize):
            st = ops.stats(x, self.contrast_mode, self._eps, take_abs=True)
            self.norm.set_affine(*stats_to_affine(st, x.numel(), self.norm.mode))
