"""Banded view of a (K x N) filterbank for the fused STFT -> mel kernel.

Mel filterbanks are banded: column n is zero outside a short run of frequency rows.  The fused
forward kernel exploits exactly that (and nothing else): per filter it needs [start, start+len) and
the weights inside.  Built on the host from the dense bank (one device->host copy per bank version),
cached by the caller.  A bank that is not banded enough simply reports `eligible = False` and the
dense MFMA projection (`at_mel_project`) is used instead.
"""
import numpy as np
import torch

MAX_SLOTS = 4        # filters per lane (N <= 256)
MAX_BAND = 128       # longest band a lane will walk


class BandedBank:
    def __init__(self, bank: torch.Tensor):
        """bank: (K, N) or (1, K, N) dense filterbank (any device)."""
        b = bank.detach().reshape(bank.shape[-2], bank.shape[-1]).float().cpu().numpy()
        K, N = b.shape
        self.K, self.N = K, N
        nz = b != 0
        has = nz.any(0)
        first = np.where(has, nz.argmax(0), 0)
        last = np.where(has, K - 1 - nz[::-1].argmax(0), -1)
        first = (first // 4) * 4                     # bands start on a 16-byte boundary: ds_read_b128 walks
        length = np.where(has, last - first + 1, 0).astype(np.int32)
        self.lmax = int(length.max()) if N else 0
        self.n_slots = (N + 63) // 64
        # row length of the weight table: the longest band rounded up to whole 16-byte quads, and an odd number
        # of quads -- lane l reads quad (l' * lpad/4 + j/4), an odd quad stride spreads 16 lanes over the
        # 16 quad slots of the 256-byte LDS bank row (conflict-free ds_read_b128)
        lpad = max(4, (self.lmax + 3) // 4 * 4)
        if (lpad // 4) % 2 == 0 and N * (lpad + 4) <= 4096:
            lpad += 4
        self.eligible = bool(0 < N and self.n_slots <= MAX_SLOTS and self.lmax <= MAX_BAND and N * lpad <= 4096)
        if not self.eligible:
            return
        wT = np.zeros((N, lpad), np.float32)
        for n in range(N):
            if length[n] > 0:
                wT[n, :length[n]] = b[first[n]:first[n] + length[n], n]
        # lane assignment: filters sorted by band length; every other pass is reversed so that a lane
        # that walks a long band in one pass gets a short one in the next
        order = np.argsort(-length, kind="stable")
        slot = np.full(self.n_slots * 64, -1, np.int32)
        slot_len = np.zeros(4, np.int32)
        for q in range(self.n_slots):
            chunk = order[q * 64:(q + 1) * 64]
            slot_len[q] = (int(length[chunk].max()) + 3) // 4 * 4 if len(chunk) else 0
            if q & 1:
                chunk = chunk[::-1]
            slot[q * 64:q * 64 + len(chunk)] = chunk
        self.slot_len = slot_len                         # host array handed to the C ABI
        self.lpad = lpad
        self.executed_macs = int(length.sum())          # multiply-adds per frame (vs K*N dense)
        self._host = (first.astype(np.int32), length, slot, wT)
        self._dev = {}

    def on(self, device):
        """(start, len, slot, wT) tensors on `device`."""
        key = str(device)
        if key not in self._dev:
            self._dev[key] = tuple(torch.from_numpy(np.ascontiguousarray(a)).to(device) for a in self._host)
        return self._dev[key]
