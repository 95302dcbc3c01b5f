"""Streaming round trip as one hipGraph: chunk -> OverlapAdd frames -> RealtimeDGT -> |X|
(-> mel features) (-> model) -> RTPGHI phase -> irfft -> overlap-add -> chunk.

The reference drives this chain from Python, module by module
(RealtimeDGT.test_inversion, transforms/dgt.py:480-508; OverlapAdd, oadd.py:69-104).  Here
the same kernels are launched once under HIP stream capture; every later step is a single
`hipGraphLaunch` replay: shapes static, all streaming state in persistent device buffers that
the kernels update IN PLACE (no state copies), no host synchronisation inside the step.

Step size.  A step takes `chunk` samples per stream, any whole number of hops -- down to ONE
hop (256 samples: BASELINE config 5's per-hop step, 5.8 ms of audio).  What that means for
each stage (SURVEY.md hard part 9):
  * framing, DGT analysis, mel features, irfft and overlap-add are chunk-invariant: the per-hop
    run produces, sample for sample, the frames / features / audio of any chunked run
    (tests/test_stream_quant_gpu.py checks it against the reference's own chunked goldens);
  * RTPGHI is NOT chunk-invariant in the reference itself: its time derivative looks one
    frame ahead inside a chunk and takes 0 for the frame after the chunk's last
    (dgt.py:388-394), and its tolerance is relative to the chunk's maximum (dgt.py:400).  A
    step of n frames therefore computes exactly what `RealtimeDGT.invert(mag, "pghi")`
    computes when called with n frames -- for the per-hop step, the reference's own n = 1 path
    (`update_buffers`' single-frame branch, dgt.py:333-335) -- and is pinned to reference
    outputs at that n (tests/golden/g15_rtpghi_per_hop.npz).
"""
from typing import Callable, Optional

import torch

from . import ops
from .transforms.dgt import RealtimeDGT
from .transforms.oadd import OverlapAdd


class StreamingDGTSession:
    def __init__(self, streams: int, chunk: int, n_fft: int = 1024, hop_length: int = 256, sr: int = 44100,
                 device="cuda", magnitude_fn: Optional[Callable[[torch.Tensor], torch.Tensor]] = None,
                 random_phase_below_tolerance: bool = True, use_graph: bool = True, mel_bands: int = 0,
                 mel_dtype: str = "fp32"):
        """streams: concurrent streams S; chunk: samples per step, a whole number of hops (>= 1 hop).
        `magnitude_fn`, if given, maps the (S, n, F) magnitudes to the magnitudes to resynthesise (a model
        working on |X|); it must be capturable (device ops only).  random_phase_below_tolerance: True = standard-normal
        phases for the bins at or below the tolerance (dgt.py:404-405), drawn inside the RTPGHI kernels; False = zeros;
        "external" = read from the persistent buffer `noise_in` (S, n, F), which the caller fills before each step
        (parity tests feed the reference's recorded draws).  mel_bands > 0 also emits log1p mel features of
        every analysed frame (`mel_out`, (S, n, mel_bands)); mel_dtype "bf16" = the dense bf16 MFMA projection
        (config 5), "fp32" = the banded fp32 one (1e-5 parity)."""
        self.S, self.C, self.n_fft, self.hop = int(streams), int(chunk), int(n_fft), int(hop_length)
        if self.C <= 0 or self.C % self.hop:
            raise ValueError("a step takes a whole number of hops (%d samples each), got %d" % (self.hop, self.C))
        if mel_dtype not in ("fp32", "bf16"):
            raise ValueError("mel_dtype must be 'fp32' or 'bf16'")
        dev = torch.device(device)
        self.device = dev
        self.n = self.C // self.hop                                   # frames per step
        self.dgt = RealtimeDGT(sr=sr, n_fft=n_fft, hop_length=hop_length, batch_size=[self.S]).to(dev)
        oa = OverlapAdd(n_fft, hop_length)
        self.keep = oa._keep
        self.gain = oa.gain_compensation.to(dev)
        self.magnitude_fn = magnitude_fn
        self.random_phase = random_phase_below_tolerance
        self.mel = None
        F = n_fft // 2 + 1
        # persistent streaming state (what OverlapAdd / RealtimeDGT keep as module buffers in the reference)
        self.x_in = torch.zeros(self.S, self.C, device=dev)
        self.buf = torch.zeros(self.S, self.keep + self.C, device=dev)   # [OverlapAdd.input_buffer | current chunk]
        self.tail = torch.zeros(self.S, self.keep, device=dev)           # OverlapAdd.output_buffer
        self.mag_hist = torch.zeros(self.S, 2, F, device=dev)            # RealtimeDGT.hgi_mag_buffer
        self.prev_phase = torch.zeros(self.S, F, device=dev)             # RealtimeDGT.hgi_phase_buffer
        # persistent outputs
        self.y_out = torch.zeros(self.S, self.C, device=dev)
        self.mag_out = torch.zeros(self.S, self.n, F, device=dev)
        self.noise_in = torch.zeros(self.S, self.n, F, device=dev) if self.random_phase == "external" else None
        # True: draws made inside the RTPGHI kernels (Philox; seed taken from torch's generator, so torch.manual_seed
        # governs it), counter advanced by the step itself -- no generator launch in the captured step
        self.rng_state = None
        if self.random_phase is True:
            seed = torch.randint(-2 ** 31, 2 ** 31 - 1, (2,), dtype=torch.int64)
            self.rng_state = torch.tensor([int(seed[0]), int(seed[1]), 0, 0], dtype=torch.int32, device=dev)
        self.mel_out = None
        if mel_bands:
            from .transforms.spectral_repr import Magnitude
            self.mel = Magnitude(sr=sr, n_fft=n_fft, n_mels=int(mel_bands), mode=None, contrast="log1p",
                                 bank_dtype=mel_dtype).to(dev)
            self.mel_out = torch.zeros(self.S, self.n, int(mel_bands), device=dev)
            if mel_dtype == "bf16":
                self.mel._bf16_image()                                   # packed before capture
        self._gamma, self._tol, self._eps = (self.dgt._hostf("gamma"), self.dgt._hostf("tolerance"),
                                             self.dgt._hostf("eps"))
        self._state = (self.buf, self.tail, self.mag_hist, self.prev_phase)
        self.graph = None
        # warm-up outside capture (one-time library init, allocator pools), then reset the state
        for _ in range(2):
            self._body()
        self.reset()
        if use_graph:
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self._body()
            self.reset()

    def _body(self):
        n_fft, h, n, F = self.n_fft, self.hop, self.n, self.n_fft // 2 + 1
        ops.oadd_push_(self.buf, self.x_in, self.keep)                   # history + chunk, in place
        X = ops.stft_forward(self.buf, self.dgt.window[:n_fft], n_fft, h, center=False, T=n,
                             clip_stride=self.buf.stride(0), L=(n - 1) * h + n_fft, B=self.S)
        # |X| of the analysis.  `mag_out` holds exactly this -- the magnitudes BEFORE `magnitude_fn` (until round 2 it
        # was a copy taken after it); a magnitude_fn that edits its argument in place edits this buffer too.  What
        # was resynthesised is `mag_out` itself when there is no magnitude_fn, otherwise that function's result.
        mag = ops.mag_pointwise(X, out=self.mag_out)
        if self.mel is not None:
            m = self.mel
            if m.bank_dtype == "bf16":
                ops.mel_forward_bf16(X, m._bf16_image(), F, m.mel_bank.shape[-1], m.contrast_mode, None, None, m._eps,
                                     out=self.mel_out)
            else:
                ops.mel_forward(X, m.mel_bank, m.contrast_mode, None, None, m._eps, band=m._band_of("mel_bank"),
                                out=self.mel_out)
        if self.magnitude_fn is not None:
            mag = self.magnitude_fn(mag)
        if self.rng_state is not None:
            phase = ops.pghi_realtime_seeded(self.mag_hist, mag, self.prev_phase, self.rng_state, self._gamma, n_fft, h,
                                             self._tol, self._eps)
        else:
            noise = self.noise_in if self.random_phase == "external" else torch.zeros_like(mag)
            phase = ops.pghi_realtime(self.mag_hist, mag, self.prev_phase, noise, self._gamma, n_fft, h, self._tol,
                                      self._eps)
        frames = ops.irfft_frames(None, self.dgt.inv_window[:n_fft], n_fft, mag=mag, phase=phase)
        ops.rt_update_buffers_(mag, phase, self.mag_hist, self.prev_phase)      # PGHI history, in place
        ops.oadd_invert(frames, self.tail, n_fft, h, self.keep, self.gain, out=self.y_out, in_place=True)

    def step(self, chunk: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One chunk (S, C) in, one resynthesised chunk (S, C) out (delayed by n_fft - hop samples).  The returned
        tensor (and `mag_out` / `mel_out`) is a persistent buffer, overwritten by the next step.  chunk=None: the
        caller has written the samples into `x_in` itself (a producer kernel on the same stream): the step is then the
        graph replay alone."""
        if chunk is not None:
            self.x_in.copy_(chunk, non_blocking=True)
        if self.graph is not None:
            self.graph.replay()
        else:
            self._body()
        return self.y_out

    def reset(self):
        for t in self._state:
            t.zero_()
