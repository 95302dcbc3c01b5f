"""Host side of the "sinebank" inversion mode (oscillator-bank resynthesis of a magnitude spectrogram).

Offline: STFT/DGT.get_sinebank_inversion (reference transforms/stft.py:180-191); per chunk:
RealtimeSTFT/RealtimeDGT.get_sinebank_inversion (stft.py:276-291, dgt.py:356-371).  The kernels are in
csrc/sinebank.hip; this module prepares what they need with the very torch calls the reference makes, so
that the fp32 rounding of frequencies, time stamps, random phases and interpolation weights is the same:

  * f_k = torch.linspace(0, sr/2, F), c_k = 2 pi f_k, t_n = torch.linspace(0, L/sr, L)  (CPU, fp32);
  * phi = 2 pi torch.rand(...) from the CPU default generator, like the reference;
  * the linear-interpolation weights come out of F.interpolate itself: it is run on two indicator rows
    (even frames / odd frames), whose outputs are exactly the weights torch gives the even and the odd frame
    next to every output sample -- no re-derivation of ATen's index arithmetic.
"""
import numpy as np
import torch

from .. import ops

_TABLES = {}


def _offline_tables(T, F, n_fft, hop, sr):
    key = (T, F, n_fft, hop, sr)
    if key in _TABLES:
        return _TABLES[key]
    L = hop * T + n_fft
    freqs = torch.linspace(0, sr / 2, F)
    c = 2 * torch.pi * freqs
    t = torch.linspace(0, L / sr, L)
    # frame below every output sample (align_corners=False source index, clamped at 0)
    src = np.float32(T) / np.float32(L) * (np.arange(L, dtype=np.float32) + np.float32(0.5)) - np.float32(0.5)
    j0 = np.clip(np.floor(np.maximum(src, 0)).astype(np.int64), 0, T - 1)
    j1 = np.minimum(j0 + 1, T - 1)
    probe = torch.zeros(1, 2, T)
    probe[0, 0, 0::2] = 1.0
    probe[0, 1, 1::2] = 1.0
    w_par = torch.nn.functional.interpolate(probe, L, mode="linear")[0].numpy()     # (2, L): even / odd frame weight
    n_blocks = (L + 127) // 128
    first = j0[np.arange(n_blocks) * 128]                                            # j0 is non-decreasing
    last = j1[np.minimum(np.arange(n_blocks) * 128 + 127, L - 1)]
    n_pass = int((last - first).max()) + 1
    frames = np.minimum(first[None, :] + np.arange(n_pass)[:, None], T - 1)         # (P, blocks)
    W = np.zeros((n_pass, L), np.float32)
    blk = np.arange(L) // 128
    idx = np.arange(L)
    for f in (j0, j1):
        p = f - first[blk]                                                           # pass that reads frame f
        W[p, idx] = w_par[f % 2, idx]       # j0 == j1 (last frame): same cell written twice with the same value
    tabs = (L, n_pass, c, t, torch.from_numpy(frames * F), torch.from_numpy(W))
    _TABLES[key] = tabs
    if len(_TABLES) > 8:
        _TABLES.pop(next(iter(_TABLES)))
    return tabs


def sinebank_offline(x_fft: torch.Tensor, sr, n_fft: int, hop: int, random_phase: torch.Tensor = None) -> torch.Tensor:
    """(..., T, F) magnitudes -> (..., hop*T + n_fft) audio.  random_phase (F, 1) overrides the draw."""
    lead = x_fft.shape[:-2]
    T, F = x_fft.shape[-2], x_fft.shape[-1]
    x = x_fft.reshape((-1, T, F))
    dev = x.device
    L, n_pass, c, t, frame_off, W = _offline_tables(T, F, n_fft, hop, sr)
    if random_phase is None:
        random_phase = 2 * torch.pi * torch.rand(F, 1)
    phi = random_phase.reshape(F).float()
    max_abs = ops.stats(x, take_abs=True)[1:2].float()                 # max |x| over the whole batch, device scalar
    y = ops.sinebank_offline(x, c.to(dev), t.to(dev), phi.to(dev), frame_off.to(dev), W.to(dev), max_abs, n_pass)
    peak = ops.stats(y, take_abs=False)[1:2].float()                   # x / x.max()
    y = ops.affine(y, torch.zeros(1, device=dev), peak)
    return y.reshape(tuple(lead) + (L,))


def sinebank_realtime(mod, x_fft: torch.Tensor, window: torch.Tensor = None) -> torch.Tensor:
    """Per-chunk resynthesis with the module's running `time_index` and per-stream `random_phase`; `window`
    (n_fft,) multiplies every frame (the realtime classes' invert(mode="sinebank"))."""
    T, F = x_fft.shape[-2], x_fft.shape[-1]
    batch_shape = tuple(x_fft.shape[:-2])
    n_fft, hop = mod._n_fft, mod._hop
    dev = x_fft.device
    if batch_shape != tuple(mod.random_phase.shape[:-2]):
        mod.random_phase = (2 * torch.pi * torch.rand(batch_shape + (1, F))).to(dev)
    # running clock: the reference keeps `time_index` as an fp32 tensor on the CPU; a host shadow of the buffer
    # (same fp32 additions) spares the streaming loop a device->host read per chunk
    now = mod.__dict__.get("_time_index_host")
    if now is None:
        now = np.float32(float(mod.time_index))
    tidx = torch.arange(n_fft).unsqueeze(0) + torch.arange(T).unsqueeze(1) * hop
    tau = tidx / mod.sr + torch.tensor(now)                            # (T, n_fft), fp32 like the reference
    c = 2 * torch.pi * torch.linspace(0, mod.sr / 2, int(n_fft / 2 + 1))
    S = 1
    for d in batch_shape:
        S *= d
    phi = mod.random_phase.to(dev).reshape(-1, F) if batch_shape else mod.random_phase.to(dev).reshape(1, F)
    y = ops.sinebank_realtime(x_fft.reshape(S, T, F), c.to(dev), tau.to(dev), phi.expand(S, F).contiguous(), window)
    step = (T * hop + n_fft) / mod.sr
    mod.__dict__["_time_index_host"] = np.float32(now + np.float32(step))
    mod.time_index = mod.time_index + step
    return y.reshape(batch_shape + (T, n_fft))
