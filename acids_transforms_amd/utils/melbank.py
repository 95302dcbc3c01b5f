"""HTK mel filterbank (host-side constant construction).

The reference obtains its bank from `torchaudio.functional.melscale_fbanks`
(spectral_repr.py:177-178; torchaudio MelSpectrogram inside mel.py:43-44).
torchaudio is not a dependency here; the bank is rebuilt from the published
definition: triangular filters whose corner frequencies are equally spaced on
the HTK mel scale m = 2595 log10(1 + f/700), evaluated on n_freqs linearly
spaced bins between 0 and sr // 2, no area normalisation.
"""
import math

import torch


def hz_to_mel_htk(f: float) -> float:
    return 2595.0 * math.log10(1.0 + f / 700.0)


def melscale_fbanks(n_freqs: int, f_min: float, f_max: float, n_mels: int, sample_rate: int) -> torch.Tensor:
    """(n_freqs, n_mels) triangular HTK filterbank."""
    bins = torch.linspace(0, sample_rate // 2, n_freqs)
    mel_pts = torch.linspace(hz_to_mel_htk(float(f_min)), hz_to_mel_htk(float(f_max)), n_mels + 2)
    corner = 700.0 * (10 ** (mel_pts / 2595.0) - 1.0)          # Hz of the n_mels + 2 triangle corners
    width = corner[1:] - corner[:-1]
    dist = corner.unsqueeze(0) - bins.unsqueeze(1)              # (n_freqs, n_mels + 2)
    rising = (-1.0 * dist[:, :-2]) / width[:-1]
    falling = dist[:, 2:] / width[1:]
    return torch.clamp(torch.min(rising, falling), min=0.0)
