"""Audio front end of the reference's examples and tests: `import_data` (utils/misc.py:29-59) and the sample-rate
conversion it applies (`torchaudio.transforms.Resample`, default arguments).

torchaudio is not part of the reference tree (requirements.txt:3, unpinned) and is absent from this image, so
  * wav files are decoded with scipy.io.wavfile and normalised the way `torchaudio.load(normalize=True)` does
    (integer PCM divided by its full scale, float left alone), channels first;
  * Resample restates torchaudio's published algorithm (`_get_sinc_resample_kernel` /
    `_apply_sinc_resample_kernel`: Hann-windowed sinc polyphase bank built in float64, zero padding by `width`
    in front and `width + orig` behind, output cropped to ceil(new * L / orig)) -- **parity unpinned**;
    the convolution itself is resample.hip.
`import_data` keeps the reference's quirks: files inside a directory are always brought to 44100 Hz (the
recursive call drops `sr`), a folder with any stereo file turns every clip stereo (mono is duplicated), clips are
zero-padded to the longest one, and files that fail to load are skipped silently.
"""
import math
import os

import numpy as np
import torch

from .. import ops

__all__ = ["Resample", "resample", "import_data", "load_wav"]


def sinc_filter_bank(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """(filters (new, 2*width + orig) float32, width, orig, new) for rates already divided by their gcd."""
    base_freq = min(orig_freq, new_freq) * rolloff
    width = math.ceil(lowpass_filter_width * orig_freq / base_freq)
    idx = torch.arange(-width, width + orig_freq, dtype=torch.float64)[None, :] / orig_freq
    t = torch.arange(0, -new_freq, -1, dtype=torch.float64)[:, None] / new_freq + idx
    t = (t * base_freq).clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    scale = base_freq / orig_freq
    kernels = torch.where(t == 0, torch.tensor(1.0, dtype=torch.float64), t.sin() / t)
    kernels = kernels * window * scale
    return kernels.to(torch.float32).contiguous(), width


class Resample(torch.nn.Module):
    """torchaudio.transforms.Resample(orig_freq, new_freq) with its default interpolation."""

    def __init__(self, orig_freq: int = 16000, new_freq: int = 16000, lowpass_filter_width: int = 6,
                 rolloff: float = 0.99):
        super().__init__()
        self.orig_freq, self.new_freq = int(orig_freq), int(new_freq)
        g = math.gcd(self.orig_freq, self.new_freq)
        self.orig, self.new = self.orig_freq // g, self.new_freq // g
        if self.orig != self.new:
            bank, self.width = sinc_filter_bank(self.orig, self.new, lowpass_filter_width, rolloff)
            self.register_buffer("kernel", bank, persistent=False)

    def forward(self, waveform: torch.Tensor) -> torch.Tensor:
        if self.orig == self.new:
            return waveform
        lead = waveform.shape[:-1]
        x = waveform.reshape(-1, waveform.shape[-1])
        y = ops.resample_sinc(x, self.orig, self.new, self.width, self.kernel.to(x.device))
        return y.reshape(tuple(lead) + (y.shape[-1],))


def resample(waveform: torch.Tensor, orig_freq: int, new_freq: int) -> torch.Tensor:
    return Resample(orig_freq, new_freq)(waveform)


def load_wav(path: str):
    """(waveform (channels, samples) float32 in [-1, 1), sample rate)."""
    from scipy.io import wavfile
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sr, a = wavfile.read(path)
    a = np.asarray(a)
    if a.ndim == 1:
        a = a[:, None]
    if a.dtype == np.uint8:
        x = (a.astype(np.float32) - 128.0) / 128.0
    elif np.issubdtype(a.dtype, np.integer):
        x = a.astype(np.float32) / float(2 ** (8 * a.dtype.itemsize - 1))
    else:
        x = a.astype(np.float32)
    return torch.from_numpy(np.ascontiguousarray(x.T)), int(sr)


def import_data(path: str, sr=44100, device=None):
    """File: (waveform (C, L) at `sr`, file name).  Directory: (stacked (N, C, Lmax) clips, names).
    Resampling runs on the GPU (`device`, default the current ROCm device); results come back on the CPU like
    the reference's unless `device` is given."""
    if os.path.isfile(path):
        x, sr_file = load_wav(path)
        if sr_file != sr:
            dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
            x = Resample(sr_file, sr)(x.to(dev))
            if device is None:
                x = x.cpu()
        elif device is not None:
            x = x.to(device)
        return x, os.path.basename(path)
    if os.path.isdir(path):
        data, names = [], []
        for f in os.listdir(path):
            try:
                x, n = import_data("%s/%s" % (path, f), device=device)      # sr not forwarded: always 44100
                data.append(x)
                names.append(os.path.splitext(os.path.basename(n))[0])
            except Exception:
                pass
        max_size = max(d.shape[1] for d in data)
        stereo = 2 in [d.shape[0] for d in data]
        for i, d in enumerate(data):
            if d.shape[0] > 1:
                d = d if stereo else d[0].unsqueeze(0)
            else:
                d = torch.cat([d, d]) if stereo else d
            if d.shape[1] <= max_size:
                d = torch.cat([d, torch.zeros(d.shape[0], max_size - d.shape[1], dtype=d.dtype, device=d.device)], 1)
            data[i] = d
        return torch.stack(data), names
    raise FileNotFoundError(path)
