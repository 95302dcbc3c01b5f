"""Tensor-level wrappers over the C ABI (include/acids_hip.h).

Each function takes ROCm-device tensors, allocates the output with torch
(device memory + stream plumbing only) and launches the HIP kernels on the
current torch stream.  No arithmetic happens in Python.
"""
import torch

from ._lib import AT_EUNSUPPORTED, AcidsHipError, check, device_scoped, lib, ptr, require_device, stream_ptr


_FP64_NARROWING = False


class allow_fp64_narrowing:
    """Opt-in for call sites written against the reference that hand over float64 audio (numpy / soundfile default) or
    complex128 spectra: `acids_transforms_amd.allow_fp64_narrowing(True)` (process-wide), or as a context manager
    `with allow_fp64_narrowing(): ...`.  Operands are then narrowed to float32 / complex64 on entry and RESULTS ARE
    FLOAT32 -- the reference would have stayed in double (stft.py:36-47, torch.stft promotes).  Off by default: a silent
    loss of digits is an error here (documented divergence with an explicit way in)."""

    def __init__(self, enabled: bool = True):
        global _FP64_NARROWING
        self.prev = _FP64_NARROWING
        _FP64_NARROWING = bool(enabled)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        global _FP64_NARROWING
        _FP64_NARROWING = self.prev
        return False


def _no_fp64(t, what="input"):
    """The kernels compute in fp32 / complex64.  The reference run on float64 data stays in double precision
    (torch.stft promotes: complex128 out, reference stft.py:98-104); narrowing that silently would hand back fewer
    digits than the caller asked for, so it is an error with the way out in the message."""
    if t.dtype in (torch.float64, torch.complex128) and not _FP64_NARROWING:
        raise AcidsHipError("%s is %s: the MI355X kernels compute in float32 / complex64 and do not narrow silently -- "
                            "convert explicitly (x.float() / X.to(torch.complex64)), or opt in once with "
                            "acids_transforms_amd.allow_fp64_narrowing(True), if single precision is acceptable"
                            % (what, str(t.dtype).replace("torch.", "")))
    return t


def _f32c(t):
    _no_fp64(t)
    if t.dtype != torch.float32:
        t = t.float()         # integer / half inputs: widening only
    return t if t.is_contiguous() else t.contiguous()


def _c64(X):
    _no_fp64(X, "spectrum")
    return X if X.dtype == torch.complex64 else X.to(torch.complex64)


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
        T = 1 + (L - (n_fft & 1)) // hop if center else None      # torch.stft pads n_fft // 2 (floor) on both sides
    F = n_fft // 2 + 1
    out = torch.empty((B, T, F), dtype=torch.complex64, device=x.device)
    phase = torch.empty((B, T, F), dtype=torch.float32, device=x.device) if want_phase else None
    check(lib().at_stft_forward(ptr(x), B, L, clip_stride, T, n_fft, hop, int(bool(center)), ptr(window),
                                ptr(out), ptr(phase), stream_ptr()), "at_stft_forward")
    return (out, phase) if want_phase else out


def istft_envelope_table(inv_window, n_fft, hop):
    require_device(inv_window)
    env = torch.empty((1 << (n_fft // hop), hop), dtype=torch.float32, device=inv_window.device)
    check(lib().at_istft_envelope_table(ptr(inv_window), n_fft, hop, ptr(env), stream_ptr()), "at_istft_envelope_table")
    return env


def istft(X, inv_window, n_fft, hop, env16=None, mag=None, phase=None):
    """(B, T, F) complex64 -- or mag & phase float32 -- -> (B, hop*(T-1) + (n_fft & 1)) float32 (torch.istft's length)."""
    src = X if X is not None else mag
    require_device(src, inv_window)
    if X is not None:
        X = X if X.is_contiguous() else X.contiguous()
        X = _c64(X)
    else:
        mag, phase = _f32c(mag), _f32c(phase)
        if phase.shape != mag.shape:
            phase = phase.expand_as(mag).contiguous()
    B, T, F = src.shape
    assert F == n_fft // 2 + 1, "last dim must be n_fft/2+1"
    # torch.istft trims n_fft // 2 at both ends of the n_fft + hop (T - 1) overlap-added samples
    y = torch.empty((B, hop * (T - 1) + (n_fft & 1) if T > 0 else 0), dtype=torch.float32, device=src.device)
    if env16 is None and n_fft in (512, 1024, 2048, 4096) and hop in (n_fft // 8, n_fft // 4, n_fft // 2):
        env16 = istft_envelope_table(inv_window, n_fft, hop)      # the fused kernel's table (modules cache theirs)
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
        X = _c64(X)
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


_CONTRAST = {None: 0, "none": 0, "log1p": 1, "log": 2, "log10": 3}


def contrast_code(mode):
    if mode not in _CONTRAST:
        raise TypeError("unknown contrast type %s" % mode)
    return _CONTRAST[mode]


def _a_kind(x, power=1):
    if torch.is_complex(x):
        return 1 if power == 2 else 0
    return 3


def _prep_in(x):
    if torch.is_complex(x):
        x = _c64(x)
    else:
        x = _f32c(x)
    return x if x.is_contiguous() else x.contiguous()


def _project_banded(x, a_kind, band, contrast, inverse, offset, scale, eps, out, N, channel_major_T, ld_out=None,
                    phase_out=None, ld_phase=0, phase_offset=None, phase_scale=None, phase_in=None, rows=None, lda=None):
    lane_filter, lane_start, weights = band.on(x.device)
    K = band.K
    check(lib().at_mel_project_banded(ptr(x), a_kind, x.numel() // K if rows is None else rows, K if lda is None else lda,
                                      K, ptr(lane_filter), ptr(lane_start),
                                      ptr(weights), N, band.n_passes, band.pass_len.ctypes.data,
                                      contrast_code(contrast), int(inverse), ptr(offset), ptr(scale), eps, ptr(out),
                                      N if ld_out is None else ld_out, channel_major_T, phase_out, ld_phase,
                                      ptr(phase_offset), ptr(phase_scale), phase_in, stream_ptr()),
          "at_mel_project_banded")
    return out


def polar_inverse(y, inv_band, contrast=None, mag_offset=None, mag_scale=None, eps=1.1920929e-07, phase_offset=None,
                  phase_scale=None):
    """Polar.invert in one pass: y (..., T, 2, F) stacked -> (..., T, F) complex64 =
    (invert_contrast(y[.., 0, :] * s + o) @ inverse_bank) * exp(i * (y[.., 1, :] * ps + po))."""
    import ctypes
    require_device(y)
    y = _f32c(y)
    F = y.shape[-1]
    assert inv_band.K == F and inv_band.N == F and inv_band.eligible
    rows = y.numel() // (2 * F)
    out = torch.empty(y.shape[:-2] + (F,), dtype=torch.complex64, device=y.device)
    phase_ptr = ctypes.c_void_p(y.data_ptr() + 4 * F)
    _project_banded(y, 2, inv_band, contrast, True, mag_offset, mag_scale, eps, out, F, 0, ld_out=F, ld_phase=2 * F,
                    phase_offset=phase_offset, phase_scale=phase_scale, phase_in=phase_ptr, rows=rows, lda=2 * F)
    return out


def polar_forward(x, band, contrast=None, mag_offset=None, mag_scale=None, eps=1.1920929e-07, phase_offset=None,
                  phase_scale=None):
    """Polar.forward in one pass: x (..., T, F) complex64 -> (..., T, 2, F) with [..., 0, :] =
    normalise(contrast(|x| @ bank)) (banded bank with F filters) and [..., 1, :] = normalise(angle(x))."""
    import ctypes
    require_device(x)
    x = _prep_in(x)
    F = x.shape[-1]
    assert band.N == F and band.eligible
    out = torch.empty(x.shape[:-1] + (2, F), dtype=torch.float32, device=x.device)
    phase_ptr = ctypes.c_void_p(out.data_ptr() + 4 * F)
    _project_banded(x, 0, band, contrast, False, mag_offset, mag_scale, eps, out, F, 0, ld_out=2 * F,
                    phase_out=phase_ptr, ld_phase=2 * F, phase_offset=phase_offset, phase_scale=phase_scale)
    return out


def mel_forward(x, bank, contrast=None, offset=None, scale=None, eps=1.1920929e-07, power=1, channel_major_T=0,
                band=None, out=None):
    """normalise(contrast(|x|^power @ bank)); x: (..., K) complex64/float32, bank: (K, N).
    channel_major_T = T > 0 stores (..., N, T) for x of shape (..., T, K) (MelSpectrogram layout).
    band: utils.banded.BandedBank of `bank` (eligible) -> the HBM-bound banded walk instead of the dense MFMA
    contraction."""
    require_device(x, bank)
    x = _prep_in(x)
    K, N = bank.shape[-2], bank.shape[-1]
    assert x.shape[-1] == K, "last dim of the input (%d) must match the bank (%d)" % (x.shape[-1], K)
    if band is not None and band.eligible:
        shape = (x.shape[:-2] + (N, channel_major_T)) if channel_major_T else (x.shape[:-1] + (N,))
        out = _out_buffer(out, shape, x.device)
        return _project_banded(x, _a_kind(x, power), band, contrast, False, offset, scale, eps, out, N, channel_major_T)
    bank2 = bank.reshape(K, N)
    bank2 = bank2 if bank2.is_contiguous() else bank2.contiguous()
    rows = x.numel() // K
    if channel_major_T:
        out = _out_buffer(out, x.shape[:-2] + (N, channel_major_T), x.device)
    else:
        out = _out_buffer(out, x.shape[:-1] + (N,), x.device)
    check(lib().at_mel_project(ptr(x), _a_kind(x, power), rows, K, K, ptr(bank2), N, N, contrast_code(contrast), 0,
                               ptr(offset), ptr(scale), eps, ptr(out), N, channel_major_T, stream_ptr()),
          "at_mel_project")
    return out


def mel_bf16_pack_bank(bank):
    """(K, N) fp32 bank -> the bf16 operand image of at_mel_project_bf16 (a uint8 tensor, opaque)."""
    require_device(bank)
    K, N = bank.shape[-2], bank.shape[-1]
    b2 = _f32c(bank.reshape(K, N))
    img = torch.empty(lib().at_mel_bf16_bank_bytes(K, N), dtype=torch.uint8, device=bank.device)
    check(lib().at_mel_bf16_pack_bank(ptr(b2), K, N, N, ptr(img), stream_ptr()), "at_mel_bf16_pack_bank")
    return img


def mel_forward_bf16(x, bank_image, K, N, contrast=None, offset=None, scale=None, eps=1.1920929e-07, power=1,
                     out=None):
    """normalise(contrast(bf16(|x|^power) @ bf16(bank))), fp32 accumulation on the bf16 matrix cores.
    x: (..., K) complex64 / float32; bank_image: mel_bf16_pack_bank(bank) of a (K, N) bank."""
    require_device(x, bank_image)
    x = _prep_in(x)
    assert x.shape[-1] == K, "last dim of the input (%d) must match the bank (%d)" % (x.shape[-1], K)
    out = _out_buffer(out, x.shape[:-1] + (N,), x.device)
    check(lib().at_mel_project_bf16(ptr(x), _a_kind(x, power), x.numel() // K, K, K, ptr(bank_image), N,
                                    contrast_code(contrast), ptr(offset), ptr(scale), eps, ptr(out), N, stream_ptr()),
          "at_mel_project_bf16")
    return out


def mel_inverse(y, inv_bank, contrast=None, offset=None, scale=None, eps=1.1920929e-07, band=None):
    """invert_contrast(y*scale+offset) @ inv_bank; y: (..., K) float32, inv_bank: (K, N).
    band: BandedBank of `inv_bank` (eligible) -> banded walk."""
    require_device(y, inv_bank)
    y = _prep_in(y)
    K, N = inv_bank.shape[-2], inv_bank.shape[-1]
    assert y.shape[-1] == K
    if band is not None and band.eligible:
        out = torch.empty(y.shape[:-1] + (N,), dtype=torch.float32, device=y.device)
        return _project_banded(y, 2, band, contrast, True, offset, scale, eps, out, N, 0)
    b2 = inv_bank.reshape(K, N)
    b2 = b2 if b2.is_contiguous() else b2.contiguous()
    rows = y.numel() // K
    out = torch.empty(y.shape[:-1] + (N,), dtype=torch.float32, device=y.device)
    check(lib().at_mel_project(ptr(y), 2, rows, K, K, ptr(b2), N, N, contrast_code(contrast), 1, ptr(offset),
                               ptr(scale), eps, ptr(out), N, 0, stream_ptr()), "at_mel_project")
    return out


def _out_buffer(out, shape, device):
    """Caller-supplied result buffer (persistent outputs of a captured streaming step) or a fresh one."""
    if out is None:
        return torch.empty(shape, dtype=torch.float32, device=device)
    if tuple(out.shape) != tuple(shape) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != device:
        raise ValueError("out= must be a contiguous float32 tensor of shape %s on %s" % (tuple(shape), device))
    return out


def mag_pointwise(x, contrast=None, offset=None, scale=None, eps=1.1920929e-07, inverse=False, out=None):
    """mel=False chains: normalise(contrast(|x|)) or invert_contrast(x*scale+offset)."""
    require_device(x)
    x = _prep_in(x)
    kind = 2 if inverse else _a_kind(x)
    out = _out_buffer(out, x.shape, x.device)
    check(lib().at_mag_pointwise(ptr(x), kind, x.numel(), contrast_code(contrast), int(inverse), ptr(offset),
                                 ptr(scale), eps, ptr(out), stream_ptr()), "at_mag_pointwise")
    return out


def stats(x, contrast=None, eps=1.1920929e-07, take_abs=True):
    """float64 tensor [min, max, sum, sumsq] of contrast(|x|) (take_abs) or of real x."""
    require_device(x)
    x = _prep_in(x)
    kind = _a_kind(x) if (take_abs or torch.is_complex(x)) else 2
    out = torch.empty(4, dtype=torch.float64, device=x.device)
    wsb = lib().at_stats_workspace_bytes()
    ws = torch.empty(wsb // 8, dtype=torch.float64, device=x.device)
    check(lib().at_stats(ptr(x), kind, x.numel(), contrast_code(contrast), eps, ptr(out), ptr(ws), wsb, stream_ptr()),
          "at_stats")
    return out


def affine(x, offset, scale, inverse=False):
    require_device(x, offset, scale)
    x = _prep_in(x)
    out = torch.empty_like(x)
    check(lib().at_affine(ptr(x), x.numel(), ptr(offset), ptr(scale), int(inverse), ptr(out), stream_ptr()), "at_affine")
    return out


def mel_forward_real(x, mat, offset=None, scale=None, channel_major_T=0):
    """normalise(x @ mat) for real x (no |.|): second projection of the MFCC extension."""
    require_device(x, mat)
    x = _prep_in(x)
    K, N = mat.shape
    rows = x.numel() // K
    if channel_major_T:
        out = torch.empty(x.shape[:-2] + (N, channel_major_T), dtype=torch.float32, device=x.device)
    else:
        out = torch.empty(x.shape[:-1] + (N,), dtype=torch.float32, device=x.device)
    mat = _f32c(mat)
    if K <= 128 and N <= 64:      # narrow matrix: columns in registers, one wavefront per row
        check(lib().at_project_small(ptr(x), rows, K, ptr(mat), N, ptr(offset), ptr(scale), ptr(out), channel_major_T,
                                     stream_ptr()), "at_project_small")
        return out
    check(lib().at_mel_project(ptr(x), 2, rows, K, K, ptr(mat), N, N, 0, 0, ptr(offset), ptr(scale), 0.0, ptr(out), N,
                               channel_major_T, stream_ptr()), "at_mel_project")
    return out


def _workspace(nbytes, device):
    return torch.empty(((nbytes + 7) // 8,), dtype=torch.int64, device=device)


def pghi_gradients(mag, gamma, n_fft, hop, eps=1.1920929e-07):
    """(B, T, F) magnitudes -> (tgradw, fgradw), each (B, T, F)."""
    require_device(mag)
    mag = _f32c(mag)
    B, T, F = mag.shape
    tg = torch.empty_like(mag)
    fg = torch.empty_like(mag)
    check(lib().at_pghi_gradients(ptr(mag), B, T, F, gamma, n_fft, hop, eps, ptr(tg), ptr(fg), ptr(None),
                                  stream_ptr()), "at_pghi_gradients")
    return tg, fg


def pghi_offline(mag, gamma, n_fft, hop, tol, eps=1.1920929e-07, debug=False):
    """(B, T, F) magnitudes -> phase (B, T, F); debug=True also returns (npops, order)."""
    require_device(mag)
    mag = _f32c(mag)
    B, T, F = mag.shape
    phase = torch.empty_like(mag)
    wsb = lib().at_pghi_offline_workspace_bytes(B, T, F)
    ws = _workspace(wsb, mag.device)
    npops = torch.zeros(B, dtype=torch.int64, device=mag.device) if debug else None
    order = torch.full((B, T * F), -1, dtype=torch.int32, device=mag.device) if debug else None
    check(lib().at_pghi_offline(ptr(mag), B, T, F, gamma, n_fft, hop, tol, eps, ptr(phase), ptr(ws), wsb, ptr(npops),
                                ptr(order), stream_ptr()), "at_pghi_offline")
    return (phase, npops, order) if debug else phase


def pghi_integrate(mag, tgradw, fgradw, tol, abstol=1.1920929e-07, debug=False):
    """DGT.perform_hgi: heap integration of (B, T, F) magnitudes along caller-supplied gradients -> phase (B, T, F);
    debug=True also returns (npops, order).  `mag` is left untouched."""
    require_device(mag, tgradw, fgradw)
    mag, tgradw, fgradw = _f32c(mag), _f32c(tgradw), _f32c(fgradw)
    assert tgradw.shape == mag.shape and fgradw.shape == mag.shape
    B, T, F = mag.shape
    phase = torch.empty_like(mag)
    wsb = lib().at_pghi_offline_workspace_bytes(B, T, F)
    ws = _workspace(wsb, mag.device)
    npops = torch.zeros(B, dtype=torch.int64, device=mag.device) if debug else None
    order = torch.full((B, T * F), -1, dtype=torch.int32, device=mag.device) if debug else None
    check(lib().at_pghi_integrate(ptr(mag), ptr(tgradw), ptr(fgradw), B, T, F, tol, abstol, ptr(phase), ptr(ws), wsb,
                                  ptr(npops), ptr(order), stream_ptr()), "at_pghi_integrate")
    return (phase, npops, order) if debug else phase


def pghi_realtime(mag_hist, mag, prev_phase, noise, gamma, n_fft, hop, tol, eps=1.1920929e-07, debug=False):
    """streaming PGHI for S streams: (S,2,F), (S,n,F), (S,F), (S,n,F) -> phase (S,n,F)."""
    require_device(mag, mag_hist, prev_phase, noise)
    mag_hist, mag, prev_phase, noise = _f32c(mag_hist), _f32c(mag), _f32c(prev_phase), _f32c(noise)
    S, n, F = mag.shape
    phase = torch.empty_like(mag)
    wsb = lib().at_pghi_rt_workspace_bytes(S, n, F)
    ws = _workspace(wsb, mag.device)
    tg = torch.empty((S, n + 2, F), dtype=torch.float32, device=mag.device) if debug else None
    fg = torch.empty((S, n + 2, F), dtype=torch.float32, device=mag.device) if debug else None
    check(lib().at_pghi_realtime(ptr(mag_hist), ptr(mag), ptr(prev_phase), ptr(noise), S, n, F, gamma, n_fft, hop, tol,
                                 eps, ptr(phase), ptr(tg), ptr(fg), ptr(ws), wsb, stream_ptr()), "at_pghi_realtime")
    return (phase, tg, fg) if debug else phase


def pghi_realtime_seeded(mag_hist, mag, prev_phase, rng_state, gamma, n_fft, hop, tol, eps=1.1920929e-07):
    """pghi_realtime with the below-tolerance draws made on the device (Philox keyed by `rng_state`, a 4-element int32
    tensor {seed lo, seed hi, step counter, 0} that the call advances): nothing but the kernels in a captured step."""
    require_device(mag, mag_hist, prev_phase, rng_state)
    mag_hist, mag, prev_phase = _f32c(mag_hist), _f32c(mag), _f32c(prev_phase)
    assert rng_state.dtype == torch.int32 and rng_state.numel() == 4 and rng_state.is_contiguous()
    S, n, F = mag.shape
    phase = torch.empty_like(mag)
    wsb = lib().at_pghi_rt_workspace_bytes(S, n, F)
    ws = _workspace(wsb, mag.device)
    check(lib().at_pghi_realtime_seeded(ptr(mag_hist), ptr(mag), ptr(prev_phase), ptr(rng_state), S, n, F, gamma, n_fft, hop,
                                        tol, eps, ptr(phase), ptr(ws), wsb, stream_ptr()), "at_pghi_realtime_seeded")
    return phase


def rt_update_buffers_(mag, phase, mag_hist, prev_phase):
    """RealtimeDGT.update_buffers (dgt.py:330-336) IN PLACE on the state tensors mag_hist (S, 2, F) and
    prev_phase (S, F), for x = mag * exp(i * phase) with mag / phase (S, n, F)."""
    require_device(mag, phase, mag_hist, prev_phase)
    mag, phase = _f32c(mag), _f32c(phase)
    S, n, F = mag.shape
    assert mag_hist.is_contiguous() and prev_phase.is_contiguous() and mag_hist.dtype == torch.float32
    assert tuple(mag_hist.shape) == (S, 2, F) and tuple(prev_phase.shape) == (S, F)
    check(lib().at_rt_update_buffers(ptr(mag), ptr(phase), S, n, F, ptr(mag_hist), ptr(mag_hist), ptr(prev_phase),
                                     stream_ptr()), "at_rt_update_buffers")


def rt_polar_irfft_update(mag, phase, inv_window, n_fft, mag_hist):
    """x = mag*exp(i*phase): windowed irfft frames (..., n, n_fft) plus the refreshed PGHI
    history buffers (|x[-2:]|, angle(x[-1]))."""
    require_device(mag, phase, inv_window, mag_hist)
    mag, phase = _f32c(mag), _f32c(phase)
    frames = irfft_frames(None, inv_window, n_fft, mag=mag, phase=phase)
    lead = mag.shape[:-2]
    n, F = mag.shape[-2], mag.shape[-1]
    S = 1
    for d in lead:
        S *= d
    hist_in = _f32c(mag_hist)
    hist_out = torch.empty(tuple(lead) + (2, F), dtype=torch.float32, device=mag.device)
    ph_out = torch.empty(tuple(lead) + (F,), dtype=torch.float32, device=mag.device)
    check(lib().at_rt_update_buffers(ptr(mag), ptr(phase), S, n, F, ptr(hist_in), ptr(hist_out), ptr(ph_out),
                                     stream_ptr()), "at_rt_update_buffers")
    return frames, hist_out, ph_out


def oadd_forward(x2d, hist, keep, n_fft, hop):
    """x2d (S, C), hist (S, keep) or None -> (buf (S, buf_len), new_hist (S, keep), n_frames)."""
    from .utils.misc import n_frames
    require_device(x2d)
    x2d = _f32c(x2d)
    S, C = x2d.shape
    if C < keep and C % hop:
        # reference granularity is keep samples (oadd.py:41); shorter chunks are an extension for whole hops only
        raise ValueError("OverlapAdd needs chunks of at least %d samples, or a whole number of hops (got %d)" % (keep, C))
    total = keep + C
    nw = n_frames(total, n_fft, hop)
    buf_len = max(total, nw * hop + n_fft)
    buf = torch.empty((S, buf_len), dtype=torch.float32, device=x2d.device)
    new_hist = torch.empty((S, keep), dtype=torch.float32, device=x2d.device)
    check(lib().at_oadd_forward(ptr(x2d), ptr(hist), S, C, keep, buf_len, ptr(buf), ptr(new_hist), stream_ptr()),
          "at_oadd_forward")
    return buf, new_hist, nw


def oadd_push_(buf, x2d, keep):
    """Streaming input state in place: buf (S, >= keep + C) = [history | chunk | pad] <- [last keep samples of
    (history | old chunk) | x2d | pad]."""
    require_device(buf, x2d)
    x2d = _f32c(x2d)
    S, C = x2d.shape
    assert buf.is_contiguous() and buf.dtype == torch.float32 and buf.shape[0] == S and buf.shape[1] >= keep + C
    check(lib().at_oadd_push(ptr(x2d), S, C, keep, buf.shape[1], ptr(buf), stream_ptr()), "at_oadd_push")


def oadd_invert(frames3d, tail, n_fft, hop, keep, gain, out=None, in_place=False):
    """frames (S, n, n_fft), tail (S, keep) or None -> (out (S, (n-1)*hop+n_fft-keep), new_tail (S, keep)).
    in_place=True: `tail` itself is updated (and returned)."""
    require_device(frames3d, gain)
    frames3d = _f32c(frames3d)
    S, n, _ = frames3d.shape
    out_len = (n - 1) * hop + n_fft - keep
    out = _out_buffer(out, (S, out_len), frames3d.device)
    if in_place:
        assert tail is not None and tail.is_contiguous() and tuple(tail.shape) == (S, keep)
        new_tail = tail
    else:
        new_tail = torch.empty((S, keep), dtype=torch.float32, device=frames3d.device)
    check(lib().at_oadd_invert(ptr(frames3d), ptr(tail), S, n, n_fft, hop, keep, ptr(gain), ptr(out), ptr(new_tail),
                               stream_ptr()), "at_oadd_invert")
    return out, new_tail


def mulaw_encode(x, channels):
    require_device(x)
    if not x.is_floating_point():
        raise TypeError("The input Tensor must be of floating type.")
    x = _f32c(x)
    out = torch.empty(x.shape, dtype=torch.int64, device=x.device)
    check(lib().at_mulaw_encode(ptr(x), x.numel(), channels, ptr(out), stream_ptr()), "at_mulaw_encode")
    return out


def mulaw_decode(codes, channels):
    require_device(codes)
    codes = codes if codes.is_contiguous() else codes.contiguous()
    ci = cf = None
    if codes.is_floating_point():
        cf = _f32c(codes)
    else:
        ci = codes.long()
    out = torch.empty(codes.shape, dtype=torch.float32, device=codes.device)
    check(lib().at_mulaw_decode(ptr(ci), ptr(cf), codes.numel(), channels, ptr(out), stream_ptr()), "at_mulaw_decode")
    return out


def onehot(x, classes, channel_major=False):
    require_device(x)
    if x.dtype != torch.int64:
        raise RuntimeError("one_hot is only applicable to index tensor of type LongTensor.")
    x = x if x.is_contiguous() else x.contiguous()
    if channel_major:
        inner = x.shape[-1]
        out = torch.empty(x.shape[:-1] + (classes, inner), dtype=torch.int64, device=x.device)
    else:
        inner = 0
        out = torch.empty(x.shape + (classes,), dtype=torch.int64, device=x.device)
    check(lib().at_onehot(ptr(x), x.numel(), classes, inner, ptr(out), stream_ptr()), "at_onehot")
    return out


def argmax_last(x):
    require_device(x)
    x = x if x.is_contiguous() else x.contiguous()
    xi = xf = None
    if x.is_floating_point():
        xf = _f32c(x)
    else:
        xi = x.long()
    cols = x.shape[-1]
    out = torch.empty(x.shape[:-1], dtype=torch.int64, device=x.device)
    check(lib().at_argmax_last(ptr(xi), ptr(xf), x.numel() // cols, cols, ptr(out), stream_ptr()), "at_argmax_last")
    return out


def scale_complex(mag, z):
    """mag (real) * z (complex64), same shape -> complex64."""
    require_device(mag, z)
    mag = _f32c(mag)
    z = z if z.is_contiguous() else z.contiguous()
    out = torch.empty(mag.shape, dtype=torch.complex64, device=mag.device)
    check(lib().at_scale_complex(ptr(mag), ptr(z), mag.numel(), ptr(out), stream_ptr()), "at_scale_complex")
    return out


def griffinlim_update(mag, rebuilt, tprev, momentum_over_1p):
    """One Griffin-Lim phase update: mag * normalise(rebuilt - m * tprev) -> complex64."""
    require_device(mag, rebuilt)
    mag = _f32c(mag)
    out = torch.empty(mag.shape, dtype=torch.complex64, device=mag.device)
    check(lib().at_griffinlim_update(ptr(mag), ptr(rebuilt), ptr(tprev), momentum_over_1p, mag.numel(), ptr(out),
                                     stream_ptr()), "at_griffinlim_update")
    return out


def istft_griffinlim(mag, rebuilt, tprev, momentum_over_1p, inv_window, n_fft, hop, env16):
    """istft(griffinlim_update(mag, rebuilt, tprev, m)) in one kernel (n_fft 1024, hop 128 / 256 / 512)."""
    require_device(mag, rebuilt, inv_window, env16)
    mag = _f32c(mag)
    B, T, F = mag.shape
    y = torch.empty((B, hop * max(T - 1, 0)), dtype=torch.float32, device=mag.device)
    check(lib().at_istft_griffinlim(ptr(mag), ptr(rebuilt), ptr(tprev), momentum_over_1p, B, T, n_fft, hop,
                                    ptr(inv_window), ptr(env16), ptr(y), stream_ptr()), "at_istft_griffinlim")
    return y


def stft_mel_forward(x, window, band, contrast=None, offset=None, scale=None, eps=1.1920929e-07, power=1,
                     want_spectrum=True, want_phase=False, channel_major=False, hop=256, n_fft=1024):
    """Fused n_fft=1024 forward (hop 256; 128 / 512 without channel_major): x (B, L) -> (X (B,T,513) complex64 or
    None, phase or None, features).  `band` is a utils.banded.BandedBank (fusable).  features: (B, T, N), or
    (B, N, T) when channel_major.  n_fft=2048 / 512 (any hop): features only (want_spectrum=False)."""
    require_device(x, window)
    x = _f32c(x)
    B, L = x.shape
    T = 1 + L // hop
    F = n_fft // 2 + 1
    assert n_fft == 1024 or (n_fft in (2048, 512) and not want_spectrum)
    lane_filter, lane_start, weights = band.on(x.device)
    N = band.N
    X = torch.empty((B, T, F), dtype=torch.complex64, device=x.device) if want_spectrum else None
    phase = torch.empty((B, T, F), dtype=torch.float32, device=x.device) if (want_phase and want_spectrum) else None
    feat = torch.empty((B, N, T) if channel_major else (B, T, N), dtype=torch.float32, device=x.device)
    check(lib().at_stft_mel_forward(ptr(x), B, L, L, T, n_fft, hop, ptr(window), ptr(lane_filter), ptr(lane_start),
                                    ptr(weights), N, band.n_passes, band.pass_len.ctypes.data,
                                    contrast_code(contrast), int(power == 2),
                                    ptr(offset), ptr(scale), eps, ptr(X), ptr(phase), ptr(feat), int(channel_major),
                                    stream_ptr()), "at_stft_mel_forward")
    return X, phase, feat


# ----------------------------------------------------------------------------------------------
# phase-side representations (phase_repr.hip)
# ----------------------------------------------------------------------------------------------
SCAN_MODES = {"unwrap": 0, "forward": 1, "backward": 2, "central": 3, "angle": 4}


def _btf(x):
    if x.ndim < 2:
        raise IndexError("expected (..., frames, bins), got shape %s" % (tuple(x.shape),))
    T, F = x.shape[-2], x.shape[-1]
    return x.numel() // max(T * F, 1) if T * F else 0, T, F


def phase_scan(x, mode, frame_window=None, offset=None, scale=None, bare=False):
    """Scan along dim -2 of x (..., T, F): complex64 spectrum (the angle is taken inside) or float32 phase.
    mode: "angle" | "unwrap" | "forward" | "backward" | "central" (IF of the unwrapped phase, reference row
    scaling included); bare=True: the plain fdiff_* of a real signal.  Optional per-frame weight (T,) and
    Normalize affine are applied last.  Returns float32 of x's shape."""
    if mode not in SCAN_MODES:
        raise AttributeError("method %s not known" % mode)
    require_device(x)
    cplx = x.is_complex()
    x = _c64(x) if cplx else _f32c(x)
    x = x if x.is_contiguous() else x.contiguous()
    B, T, F = _btf(x)
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    if frame_window is not None:
        frame_window = _f32c(frame_window.to(x.device))
        assert frame_window.numel() == T
    check(lib().at_phase_scan(ptr(x) if cplx else None, None if cplx else ptr(x), B, T, F, SCAN_MODES[mode], int(bare),
                              ptr(frame_window), ptr(offset), ptr(scale), ptr(out), stream_ptr()), "at_phase_scan")
    return out


def phase_integrate(y, method, offset=None, scale=None, rescale=True):
    """IF.invert's tail: (de-normalise,) undo the row scaling of `method` (rescale) and integrate along dim -2."""
    if method not in ("forward", "backward", "central"):
        raise AttributeError("method %s not known" % method)
    require_device(y)
    y = _f32c(y)
    B, T, F = _btf(y)
    out = torch.empty_like(y)
    check(lib().at_phase_integrate(ptr(y), B, T, F, SCAN_MODES[method], int(rescale), ptr(offset), ptr(scale), ptr(out),
                                   stream_ptr()), "at_phase_integrate")
    return out


def polarif_forward(x, band, contrast, mag_offset, mag_scale, eps, method, frame_window=None, if_offset=None,
                    if_scale=None):
    """PolarIF.forward written straight into the stacked tensor: x (..., T, F) complex64 -> (..., T, 2, F) with
    [..., 0, :] = normalise(contrast(|x| @ bank)) (banded bank with F filters) and [..., 1, :] = normalise(IF(x))."""
    import ctypes
    require_device(x)
    x = _prep_in(x)
    B, T, F = _btf(x)
    assert band.N == F and band.K == F and band.eligible and method in ("forward", "backward", "central")
    out = torch.empty(x.shape[:-1] + (2, F), dtype=torch.float32, device=x.device)
    if frame_window is not None:
        frame_window = _f32c(frame_window.to(x.device))
        assert frame_window.numel() == T
    # one pass over the spectrum where the clip-per-block scan applies (>= 64 clips, 256..2048 bins; the library answers
    # AT_EUNSUPPORTED otherwise, and under variant("scan_layout", 1)); same bits as the two kernels below
    start, length, woff, w = band.by_filter(x.device)
    err = lib().at_polarif_forward(ptr(x), B, T, F, SCAN_MODES[method], ptr(frame_window), ptr(if_offset), ptr(if_scale),
                                   ptr(start), ptr(length), ptr(woff), ptr(w), w.numel(), contrast_code(contrast),
                                   ptr(mag_offset), ptr(mag_scale), float(eps), ptr(out), stream_ptr())
    if err != AT_EUNSUPPORTED:
        check(err, "at_polarif_forward")
        return out
    _project_banded(x, 0, band, contrast, False, mag_offset, mag_scale, eps, out, F, 0, ld_out=2 * F)
    check(lib().at_phase_scan_strided(ptr(x), None, B, T, F, SCAN_MODES[method], 0, ptr(frame_window), ptr(if_offset),
                                      ptr(if_scale), ctypes.c_void_p(out.data_ptr() + 4 * F), 2 * F, stream_ptr()),
          "at_phase_scan_strided")
    return out


def polarif_inverse(y, inv_band, contrast, mag_offset, mag_scale, eps, method, if_offset=None, if_scale=None):
    """PolarIF.invert reading the stacked tensor in place: y (..., T, 2, F) -> (..., T, F) complex64 =
    Magnitude.invert(y[.., 0, :]) * exp(i * IF.invert(y[.., 1, :]))."""
    import ctypes
    require_device(y)
    y = _f32c(y)
    F = y.shape[-1]
    T = y.shape[-3]
    rows = y.numel() // (2 * F)
    assert inv_band.K == F and inv_band.N == F and inv_band.eligible and method in ("forward", "backward", "central")
    mag = torch.empty(y.shape[:-2] + (F,), dtype=torch.float32, device=y.device)
    _project_banded(y, 2, inv_band, contrast, True, mag_offset, mag_scale, eps, mag, F, 0, rows=rows, lda=2 * F)
    out = torch.empty(y.shape[:-2] + (F,), dtype=torch.complex64, device=y.device)
    check(lib().at_phase_integrate_polar(ctypes.c_void_p(y.data_ptr() + 4 * F), 2 * F, rows // T, T, F, SCAN_MODES[method],
                                         ptr(if_offset), ptr(if_scale), ptr(mag), ptr(out), stream_ptr()),
          "at_phase_integrate_polar")
    return out


def cartesian_forward(x, re_offset=None, re_scale=None, im_offset=None, im_scale=None):
    """Cartesian.forward in one pass: x (..., F) complex64 -> (..., 2, F) = [normalise(x.real), normalise(x.imag)]."""
    require_device(x)
    x = _prep_in(x)
    F = x.shape[-1]
    out = torch.empty(x.shape[:-1] + (2, F), dtype=torch.float32, device=x.device)
    check(lib().at_cartesian_pack(ptr(x), x.numel() // F, F, ptr(re_offset), ptr(re_scale), ptr(im_offset), ptr(im_scale),
                                  ptr(out), stream_ptr()), "at_cartesian_pack")
    return out


def cartesian_inverse(y, re_offset=None, re_scale=None, im_offset=None, im_scale=None):
    """Cartesian.invert in one pass: y (..., 2, F) float32 -> (..., F) complex64."""
    require_device(y)
    y = _f32c(y)
    F = y.shape[-1]
    out = torch.empty(y.shape[:-2] + (F,), dtype=torch.complex64, device=y.device)
    check(lib().at_cartesian_unpack(ptr(y), y.numel() // (2 * F), F, ptr(re_offset), ptr(re_scale), ptr(im_offset),
                                    ptr(im_scale), ptr(out), stream_ptr()), "at_cartesian_unpack")
    return out


def polar_to_complex(mag, phase):
    """mag * exp(1j * phase) as complex64."""
    require_device(mag, phase)
    mag, phase = torch.broadcast_tensors(mag, phase)
    mag, phase = _f32c(mag), _f32c(phase)
    out = torch.empty(mag.shape, dtype=torch.complex64, device=mag.device)
    check(lib().at_polar_to_complex(ptr(mag), ptr(phase), mag.numel(), ptr(out), stream_ptr()), "at_polar_to_complex")
    return out


# ----------------------------------------------------------------------------------------------
# sinebank inversion (sinebank.hip)
# ----------------------------------------------------------------------------------------------
def sinebank_offline(x, c, t, phi, block_frame_offset, W3, max_abs, n_pass):
    """x (B, T, F) magnitudes -> (B, L) oscillator-bank resynthesis before the final max-normalisation.
    c (F,), t (L,), phi (F,), W3 (n_pass, L) float32 and block_frame_offset (n_pass, ceil(L/128)) int64 on x's
    device."""
    require_device(x, c, t, phi, block_frame_offset, W3, max_abs)
    c, t, phi, W3 = _f32c(c), _f32c(t), _f32c(phi), _f32c(W3)
    block_frame_offset = block_frame_offset.contiguous()
    x = _f32c(x)
    B, T, F = x.shape
    L = t.numel()
    out = torch.empty((B, L), dtype=torch.float32, device=x.device)
    wsb = lib().at_sinebank_workspace_bytes(B, F, L, n_pass)
    ws = _workspace(wsb, x.device)
    check(lib().at_sinebank_offline(ptr(x), B, T, F, ptr(c), ptr(t), ptr(phi), L, n_pass, ptr(block_frame_offset), ptr(W3),
                                    ptr(max_abs), ptr(out), ptr(ws), wsb, stream_ptr()), "at_sinebank_offline")
    return out


def sinebank_realtime(x, c, tau, phi, window=None):
    """x (S, T, F), tau (T, N), phi (S, F) -> (S, T, N) frames (times window (N,) when given)."""
    require_device(x, c, tau, phi, window)
    window = _f32c(window) if window is not None else None
    x, c, tau, phi = _f32c(x), _f32c(c), _f32c(tau), _f32c(phi)
    S, T, F = x.shape
    N = tau.shape[-1]
    out = torch.empty((S, T, N), dtype=torch.float32, device=x.device)
    check(lib().at_sinebank_realtime(ptr(x), S, T, F, N, ptr(c), ptr(tau), ptr(phi), ptr(window), ptr(out),
                                     stream_ptr()), "at_sinebank_realtime")
    return out


def resample_sinc(x, orig, new, width, filters):
    """x (rows, L) -> (rows, ceil(new * L / orig)) through the (new, 2*width + orig) polyphase bank."""
    require_device(x, filters)
    x, filters = _f32c(x), _f32c(filters)
    rows, L = x.shape
    out_len = -(-new * L // orig)
    out = torch.empty((rows, out_len), dtype=torch.float32, device=x.device)
    check(lib().at_resample_sinc(ptr(x), rows, L, orig, new, width, ptr(filters), out_len, ptr(out), stream_ptr()),
          "at_resample_sinc")
    return out


def stft_polar_forward(x, window, band, contrast=None, mag_offset=None, mag_scale=None, eps=1.1920929e-07,
                       phase_offset=None, phase_scale=None):
    """Compose(STFT -> Polar) in one kernel: x (B, L) -> (B, T, 2, F): [.., 0, :] = normalise(contrast(|X| @ bank)),
    [.., 1, :] = normalise(angle X).  n_fft = 1024, hop = 256, bank with F = 513 banded filters."""
    require_device(x, window)
    x = _f32c(x)
    B, L = x.shape
    T = 1 + L // 256
    lane_filter, lane_start, weights = band.on(x.device)
    out = torch.empty((B, T, 2, 513), dtype=torch.float32, device=x.device)
    check(lib().at_stft_polar_forward(ptr(x), B, L, L, T, 1024, 256, ptr(window), ptr(lane_filter), ptr(lane_start),
                                      ptr(weights), band.N, band.n_passes, band.pass_len.ctypes.data,
                                      contrast_code(contrast), ptr(mag_offset), ptr(mag_scale), eps, ptr(phase_offset),
                                      ptr(phase_scale), ptr(out), stream_ptr()), "at_stft_polar_forward")
    return out


# every public op enters the device of its operands (see _lib.device_scoped)
for _name, _fn in list(globals().items()):
    if callable(_fn) and not _name.startswith("_") and getattr(_fn, "__module__", None) == __name__ \
            and _name not in ("contrast_code",):
        globals()[_name] = device_scoped(_fn)
del _name, _fn
