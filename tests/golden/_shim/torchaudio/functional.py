"""See package docstring.  `_INJECTED_BANK` is set by make_golden.py."""
import torch

_INJECTED_BANK = None


def melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate, norm=None, mel_scale="htk"):
    if _INJECTED_BANK is None:
        # shape-correct placeholder: identity-like bank (never used for a golden)
        return torch.eye(n_freqs, n_mels)
    assert _INJECTED_BANK.shape == (n_freqs, n_mels), (_INJECTED_BANK.shape, n_freqs, n_mels)
    return _INJECTED_BANK.clone()


def griffinlim(*a, **k):  # stft.py:2,178
    raise NotImplementedError("torchaudio.functional.griffinlim is not available")
