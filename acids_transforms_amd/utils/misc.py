"""Shape helpers shared by the transforms (reference utils/misc.py:138-178)."""
__all__ = ["pad", "frame", "n_frames", "reshape_batches", "unwrap", "fdiff_forward", "fdiff_backward", "fdiff_central",
           "fint_forward", "fint_backward", "fint_central"]
from typing import Tuple

import torch


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
