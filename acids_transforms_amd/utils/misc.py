"""Shape helpers shared by the transforms (reference utils/misc.py:138-178)."""
__all__ = ["format_input_data", "pad", "frame", "n_frames", "reshape_batches", "get_fft_idx", "deriv", "unwrap", "fdiff_forward", "fdiff_backward", "fdiff_central",
           "fint_forward", "fint_backward", "fint_central"]
from typing import Tuple

import torch


def format_input_data(x: torch.Tensor, dim=-1):
    """Present because the reference exports it (utils/misc.py:61-63): there it splits `x.shape` at `dim` into a batch
    part and a data part and then returns NOTHING (no return statement).  Same here, on purpose: anything else would
    be a different function under the same name; `reshape_batches` is the one that does the job."""
    batch_size = x.shape[:dim]          # noqa: F841 -- as in the reference: computed, not returned
    data_size = x.shape[dim:]           # noqa: F841
    return None


def pad(tensor: torch.Tensor, target_size: int, dim: int):
    """Zero-pad `dim` up to target_size (no-op when already longer)."""
    if tensor.size(dim) > target_size:
        return tensor
    shape = list(tensor.shape)
    shape[dim] = target_size - tensor.shape[dim]
    return torch.cat([tensor, torch.zeros(shape, dtype=tensor.dtype, device=tensor.device)], dim=dim)


def n_frames(length: int, wsize: int, hsize: int) -> int:
    """Number of windows frame() produces (reference utils/misc.py:153-155)."""
    n = (length - wsize) // hsize
    if length >= n * hsize + wsize:
        n += 1
    return n


def frame(tensor: torch.Tensor, wsize: int, hsize: int, dim: int):
    """Zero-copy overlapping frames along `dim` (as_strided view, like the reference)."""
    if dim < 0:
        dim = tensor.ndim + dim
    if not tensor.is_contiguous():
        tensor = tensor.contiguous()
    nw = n_frames(tensor.shape[dim], wsize, hsize)
    tensor = pad(tensor, nw * hsize + wsize, dim)
    shape = list(tensor.shape)
    shape[dim] = nw
    shape.insert(dim + 1, wsize)
    strides = [tensor.stride(i) for i in range(tensor.ndim)]
    strides.insert(dim, hsize * tensor.stride(dim))
    return torch.as_strided(tensor, shape, strides)


def reshape_batches(x: torch.Tensor, dim: int, allow_clone: bool = True) -> Tuple[torch.Tensor, torch.Size]:
    """Flatten the leading batch dims: (..., event) -> (B, event)."""
    batch_size = x.shape[:dim]
    event_size = x.shape[dim:]
    if x.is_contiguous():
        x = x.view(torch.Size([-1]) + event_size)
    elif allow_clone:
        x = x.reshape(torch.Size([-1]) + event_size)
    else:
        raise ValueError("found non contiguous tensor of size : %s" % (x.shape,))
    return x, batch_size


# ---- scans along the frame axis (reference utils/misc.py:12-26, 65-104); arithmetic in phase_repr.hip ----
def unwrap(tensor: torch.Tensor) -> torch.Tensor:
    """Phase unwrapping along dim -2 (frames), torch.cumsum's CPU arithmetic reproduced on the device."""
    from .. import ops
    return ops.phase_scan(tensor, "unwrap")


def _fdiff(x, method):
    from .. import ops
    return ops.phase_scan(x, method, bare=True)


def fdiff_forward(x):
    """Row 0 kept, row t = (x[t] - x[t-1]) / 2."""
    return _fdiff(x, "forward")


def fdiff_backward(x):
    """Last row kept, row t = (x[t] - x[t+1]) / 2."""
    return _fdiff(x, "backward")


def fdiff_central(x):
    """First and last rows kept, row t = (x[t+1] - x[t-1]) / 4 (a single frame comes back twice)."""
    if x.shape[-2] == 1:
        return torch.cat([x, x], -2)
    return _fdiff(x, "central")


def _fint(x, method):
    from .. import ops
    return ops.phase_integrate(x, method, rescale=False)


def fint_forward(x):
    """Inverse of fdiff_forward: rows >= 1 doubled, cumulative sum.  Returns a new tensor."""
    return _fint(x, "forward")


def fint_backward(x):
    return _fint(x, "backward")


def fint_central(x):
    """The reference's two interleaved recurrences, quirks included (see phase_repr.hip)."""
    return _fint(x, "central")


def get_fft_idx(L: int) -> torch.Tensor:
    """Signed bin indices of an L-point FFT in storage order: 0 .. ceil(L/2) (the Nyquist bin counted as positive), then
    the negative ones (reference utils/misc.py:130-135)."""
    if L % 2 == 0:
        return torch.cat([torch.arange(0, L // 2 + 1), torch.arange(-L // 2 + 1, 0)])
    return torch.cat([torch.arange(0, (L + 1) // 2), torch.arange(-(L - 1) // 2, 0)])


def deriv(mag: torch.Tensor, order=2) -> torch.Tensor:
    """Derivative of a periodic signal sampled on [0, 1) along dim 0 (ltfat's pderiv; reference utils/misc.py:107-127):
    centred differences of order 2 or 4, or the spectral derivative for order = inf.  (The reference's spectral branch
    cannot run -- it calls Tensor.transpose() without arguments and never applies the bin index; this is the operation
    it documents.  No transform of the reference calls `deriv`.)"""
    assert order in (2, 4, float("inf")), "order must be 2, 4 or inf"
    L = mag.shape[0]
    if order == 2:
        return L * (mag.roll(-1, 0) - mag.roll(1, 0)) / 2
    if order == 4:
        return L * (-mag.roll(-2, 0) + 8 * mag.roll(-1, 0) - 8 * mag.roll(1, 0) + mag.roll(2, 0)) / 12
    n = get_fft_idx(L).to(mag.device)
    if L % 2 == 0:
        n = n.clone()
        n[L // 2] = 0                     # the Nyquist bin of a real signal has no derivative of its own
    n = n.reshape((L,) + (1,) * (mag.dim() - 1))
    out = 2 * torch.pi * torch.fft.ifft(1j * n * torch.fft.fft(mag, dim=0), dim=0)
    return out if mag.is_complex() else out.real
