"""Tensor-level wrappers over the C ABI (include/acids_hip.h).

Each function takes ROCm-device tensors, allocates the output with torch
(device memory + stream plumbing only) and launches the HIP kernels on the
current torch stream.  No arithmetic happens in Python.
"""
import torch

from . import _lib
from ._lib import check, lib, ptr, require_device, stream_ptr


def _f32c(t):
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def stft_forward(x, window, n_fft, hop, center=True, want_phase=False, T=None, clip_stride=None, L=None, B=None):
    """x: (B, L) float32 -> (B, T, F) complex64 [, phase (B, T, F) float32].

    With center=False and explicit T / clip_stride / L the kernel frames an
    overlapping strided view directly (OverlapAdd.forward output) without a copy.
    """
    require_device(x, window)
    if B is None:
        B = x.shape[0]
    if L is None:
        x = _f32c(x)
        L = x.shape[1]
        clip_stride = L
    if T is None:
        T = 1 + L // hop if center else None
    F = n_fft // 2 + 1
    out = torch.empty((B, T, F), dtype=torch.complex64, device=x.device)
    phase = torch.empty((B, T, F), dtype=torch.float32, device=x.device) if want_phase else None
    check(lib().at_stft_forward(ptr(x), B, L, clip_stride, T, n_fft, hop, int(bool(center)), ptr(window),
                                ptr(out), ptr(phase), stream_ptr()), "at_stft_forward")
    return (out, phase) if want_phase else out


def istft_envelope_table(inv_window, n_fft, hop):
    require_device(inv_window)
    env = torch.empty((16, hop), dtype=torch.float32, device=inv_window.device)
    check(lib().at_istft_envelope_table(ptr(inv_window), n_fft, hop, ptr(env), stream_ptr()), "at_istft_envelope_table")
    return env


def istft(X, inv_window, n_fft, hop, env16=None, mag=None, phase=None):
    """(B, T, F) complex64 -- or mag & phase float32 -- -> (B, hop*(T-1)) float32."""
    src = X if X is not None else mag
    require_device(src, inv_window)
    if X is not None:
        X = X if X.is_contiguous() else X.contiguous()
        if X.dtype != torch.complex64:
            X = X.to(torch.complex64)
    else:
        mag, phase = _f32c(mag), _f32c(phase)
        if phase.shape != mag.shape:
            phase = phase.expand_as(mag).contiguous()
    B, T, F = src.shape
    assert F == n_fft // 2 + 1, "last dim must be n_fft/2+1"
    y = torch.empty((B, hop * max(T - 1, 0)), dtype=torch.float32, device=src.device)
    wsb = lib().at_istft_workspace_bytes(B, T, n_fft, hop)
    ws = torch.empty((wsb // 4,), dtype=torch.float32, device=src.device) if wsb else None
    check(lib().at_istft(ptr(X), ptr(mag), ptr(phase), B, T, n_fft, hop, ptr(inv_window), ptr(env16), ptr(y),
                         ptr(ws), wsb, stream_ptr()), "at_istft")
    return y


def irfft_frames(X, inv_window, n_fft, mag=None, phase=None):
    """(..., F) spectra -> (..., n_fft) windowed frames (no overlap-add)."""
    src = X if X is not None else mag
    require_device(src, inv_window)
    if X is not None:
        X = X if X.is_contiguous() else X.contiguous()
        if X.dtype != torch.complex64:
            X = X.to(torch.complex64)
    else:
        mag, phase = _f32c(mag), _f32c(phase)
        if phase.shape != mag.shape:
            phase = phase.expand_as(mag).contiguous()
    lead = src.shape[:-1]
    n = 1
    for d in lead:
        n *= d
    out = torch.empty(tuple(lead) + (n_fft,), dtype=torch.float32, device=src.device)
    check(lib().at_irfft_frames(ptr(X), ptr(mag), ptr(phase), n, n_fft, ptr(inv_window), ptr(out), stream_ptr()),
          "at_irfft_frames")
    return out


def angle(X):
    """complex64 -> float32 atan2(im, re), same shape."""
    require_device(X)
    X = X if X.is_contiguous() else X.contiguous()
    out = torch.empty(X.shape, dtype=torch.float32, device=X.device)
    check(lib().at_angle(ptr(X), X.numel(), ptr(out), stream_ptr()), "at_angle")
    return out
