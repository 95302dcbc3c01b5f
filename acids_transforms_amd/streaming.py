"""Streaming round trip as one hipGraph: chunk -> OverlapAdd frames -> RealtimeDGT -> |X|
(-> model) -> RTPGHI phase -> irfft -> overlap-add -> chunk.

The reference drives this chain from Python, module by module
(RealtimeDGT.test_inversion, transforms/dgt.py:480-508; OverlapAdd, oadd.py).  Here the
same kernels are launched once under HIP stream capture; every later step is a single
`hipGraphLaunch` replay (shapes static, all streaming state in persistent device buffers
that the graph updates in place, no host synchronisation inside the step).
"""
from typing import Callable, Optional

import torch

from . import ops
from .transforms.dgt import RealtimeDGT
from .transforms.oadd import OverlapAdd


class StreamingDGTSession:
    def __init__(self, streams: int, chunk: int, n_fft: int = 1024, hop_length: int = 256, sr: int = 44100,
                 device="cuda", magnitude_fn: Optional[Callable[[torch.Tensor], torch.Tensor]] = None,
                 random_phase_below_tolerance: bool = True, use_graph: bool = True, mel_bands: int = 0):
        """streams: concurrent streams S; chunk: samples per step (>= n_fft - hop).  `magnitude_fn`, if given,
        maps the (S, n, F) magnitudes to the magnitudes to resynthesise (a model working on |X|); it must
        be capturable (device ops only).  mel_bands > 0 also emits log1p mel features of every analysed frame
        (`mel_out`, (S, n, mel_bands); banded projection of the spectrum, part of the captured graph)."""
        self.S, self.C, self.n_fft, self.hop = int(streams), int(chunk), int(n_fft), int(hop_length)
        dev = torch.device(device)
        self.device = dev
        self.dgt = RealtimeDGT(sr=sr, n_fft=n_fft, hop_length=hop_length, batch_size=[self.S]).to(dev)
        oa = OverlapAdd(n_fft, hop_length)
        self.keep = oa._keep
        if self.C < self.keep:
            raise ValueError("chunks must hold at least %d samples" % self.keep)
        self.gain = oa.gain_compensation.to(dev)
        self.magnitude_fn = magnitude_fn
        self.random_phase = random_phase_below_tolerance
        self.mel = None
        self.mel_out = None
        if mel_bands:
            from .transforms.spectral_repr import Magnitude
            self.mel = Magnitude(sr=sr, n_fft=n_fft, n_mels=int(mel_bands), mode=None, contrast="log1p").to(dev)
        F = n_fft // 2 + 1
        # persistent streaming state (what OverlapAdd / RealtimeDGT keep as module buffers in the reference)
        self.x_in = torch.zeros(self.S, self.C, device=dev)
        self.hist = torch.zeros(self.S, self.keep, device=dev)        # OverlapAdd.input_buffer
        self.tail = torch.zeros(self.S, self.keep, device=dev)        # OverlapAdd.output_buffer
        self.mag_hist = torch.zeros(self.S, 2, F, device=dev)         # RealtimeDGT.hgi_mag_buffer
        self.prev_phase = torch.zeros(self.S, F, device=dev)          # RealtimeDGT.hgi_phase_buffer
        self.y_out = None
        self.mag_out = None
        self._gamma, self._tol, self._eps = (self.dgt._hostf("gamma"), self.dgt._hostf("tolerance"),
                                             self.dgt._hostf("eps"))
        self.graph = None
        # warm-up outside capture (one-time library init, allocator pools), then reset the state
        for _ in range(2):
            self._body()
        for t in (self.hist, self.tail, self.mag_hist, self.prev_phase):
            t.zero_()
        if use_graph:
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self._body()
            for t in (self.hist, self.tail, self.mag_hist, self.prev_phase):
                t.zero_()

    def _body(self):
        n, h = self.n_fft, self.hop
        buf, new_hist, nw = ops.oadd_forward(self.x_in, self.hist, self.keep, n, h)
        self.hist.copy_(new_hist)
        X = ops.stft_forward(buf, self.dgt.window[:n], n, h, center=False, T=nw, clip_stride=buf.stride(0),
                             L=(nw - 1) * h + n, B=self.S)
        mag = ops.mag_pointwise(X)                                     # |X|
        if self.mel is not None:
            feat = self.mel(X)
            if self.mel_out is None:
                self.mel_out = torch.empty_like(feat)
            self.mel_out.copy_(feat)
        if self.magnitude_fn is not None:
            mag = self.magnitude_fn(mag)
        noise = torch.randn_like(mag) if self.random_phase else torch.zeros_like(mag)
        phase = ops.pghi_realtime(self.mag_hist, mag, self.prev_phase, noise, self._gamma, n, h, self._tol, self._eps)
        frames, mh, pp = ops.rt_polar_irfft_update(mag, phase, self.dgt.inv_window[:n], n, self.mag_hist)
        self.mag_hist.copy_(mh)
        self.prev_phase.copy_(pp)
        y, new_tail = ops.oadd_invert(frames, self.tail, n, h, self.keep, self.gain)
        self.tail.copy_(new_tail)
        if self.y_out is None:
            self.y_out = torch.empty_like(y)
            self.mag_out = torch.empty_like(mag)
        self.y_out.copy_(y)
        self.mag_out.copy_(mag)

    def step(self, chunk: torch.Tensor) -> torch.Tensor:
        """One chunk (S, C) in, one resynthesised chunk (S, C) out (delayed by n_fft - hop samples)."""
        self.x_in.copy_(chunk, non_blocking=True)
        if self.graph is not None:
            self.graph.replay()
        else:
            self._body()
        return self.y_out

    def reset(self):
        for t in (self.hist, self.tail, self.mag_hist, self.prev_phase):
            t.zero_()
