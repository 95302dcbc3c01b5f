"""Helpers behind the per-class self-test hooks (`test_forward`, `test_inversion`, `test_scripted_transform`).

The reference's classes carry these hooks and its test file drives every transform through them
(test/test_transforms.py:28-60): spectral representations are fed an STFT of the audio, their inversions are
completed with the part of the spectrum they do not model and brought back to audio with a *window-less*
torch.stft / torch.istft pair (n_fft 1024, hop 256; `Real` uses 512 / 128).  Here the same scenarios run on the
device, through this package's own kernels -- a window of ones is what "no window" means to torch.
"""
import torch

from .. import ops
from ..utils.misc import reshape_batches


def rect_stft(x: torch.Tensor, n_fft: int = 1024, hop: int = 256):
    """torch.stft(x, n_fft, hop, return_complex=True).transpose(-2, -1) on flattened batches -> (X, batch_shape)."""
    xb, batch_shape = reshape_batches(x, -1)
    xb = xb.float().contiguous()
    window = torch.ones(n_fft, device=xb.device)
    return ops.stft_forward(xb, window, n_fft, hop, center=True), tuple(batch_shape)


def rect_istft(X: torch.Tensor, batch_shape, n_fft: int = 1024, hop: int = 256) -> torch.Tensor:
    """torch.istft(X.transpose(-2, -1), n_fft, hop) -> audio of shape batch_shape + (samples,)."""
    window = torch.ones(n_fft, device=X.device)
    y = ops.istft(X.to(torch.complex64).contiguous(), window, n_fft, hop)
    return y.reshape(tuple(batch_shape) + (y.shape[-1],))


def stft_then(transform, x: torch.Tensor, time=None):
    """The forward self-test of every spectral representation: STFT() of the audio, scale_data, forward."""
    from .stft import STFT
    stage = STFT().to(x.device)
    if time is None:
        X = stage(x)
        transform.scale_data(X)
        return transform(X)
    X, time = stage.forward_with_time(x, time)
    transform.scale_data(X)
    return transform.forward_with_time(X, time)


def random_spectrum(device, shape=(2, 10, 513)) -> torch.Tensor:
    mag = torch.randn(*shape, device=device)
    ang = 2 * torch.pi * torch.rand(*shape, device=device)
    return torch.polar(mag.abs(), ang) * torch.sign(mag)
