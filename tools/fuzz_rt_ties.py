#!/usr/bin/env python3
"""Realtime PGHI: the scan path (default, round 5) AND the rank fast path (at_set_variant pghi_kernel = 4) against the
cooperative heap kernel (pghi_kernel = 3) on spectra with
INJECTED ties -- a handful of magnitudes per stream copied to other bins of the same or the neighbouring frame, at random
distances (far apart: the pops commute and the frame stays on the fast path; one or two bins apart, or onto the frame
maximum: the pre-pass must report them and the frame must take the heap).  Phases must be the same bits."""
import os
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import acids_transforms_amd as A  # noqa: E402
from acids_transforms_amd import ops  # noqa: E402
from acids_transforms_amd._lib import variant  # noqa: E402

dev = torch.device("cuda")
rng = np.random.RandomState(int(os.environ.get("FUZZ_SEED", "0")))
n_cases = int(os.environ.get("FUZZ_CASES", "120"))
frames = 0
for case in range(n_cases):
    n_fft = int(rng.choice([256, 400, 512, 1024]))
    hop = n_fft // 4
    F = n_fft // 2 + 1
    S, n = int(rng.randint(1, 9)), int(rng.randint(1, 6))
    m = np.abs(rng.randn(S, n + 2, F) + 1j * rng.randn(S, n + 2, F)).astype(np.float32)
    kind = rng.choice(["plain", "sparse", "decay"])
    if kind == "sparse":
        m = (m * (rng.rand(S, n + 2, F) < 0.2) + 1e-6).astype(np.float32)
    elif kind == "decay":
        m = (m * np.exp(-np.arange(F) / (F / 8.0))[None, None]).astype(np.float32)
    for s in range(S):
        for _ in range(int(rng.randint(0, 12))):
            r0, k0 = int(rng.randint(0, n + 2)), int(rng.randint(0, F))
            r1 = min(n + 1, max(0, r0 + int(rng.randint(-1, 2))))
            mode = rng.randint(0, 4)
            k1 = int(rng.randint(0, F)) if mode == 0 else min(F - 1, max(0, k0 + int(rng.randint(-3, 4))))
            if mode == 3:                                   # onto the maximum of a row
                r1 = int(rng.randint(0, n + 2))
                k1 = int(m[s, r1].argmax())
            m[s, r0, k0] = m[s, r1, k1]
    rt = A.RealtimeDGT(n_fft=n_fft, hop_length=hop, batch_size=[S]).to(dev)
    mt = torch.from_numpy(m)
    hist, mag = mt[:, :2].contiguous().to(dev), mt[:, 2:].contiguous().to(dev)
    prev = torch.from_numpy((rng.rand(S, F) * 6.28).astype(np.float32)).to(dev)
    noise = torch.from_numpy(rng.randn(S, n, F).astype(np.float32)).to(dev)
    args = (float(rt.gamma), n_fft, hop, float(rt.tolerance), float(rt.eps))
    got = ops.pghi_realtime(hist, mag, prev, noise, *args)
    with variant("pghi_kernel", 4):
        ranked = ops.pghi_realtime(hist, mag, prev, noise, *args)
    with variant("pghi_kernel", 3):
        ref = ops.pghi_realtime(hist, mag, prev, noise, *args)
    for name, out in (("scan", got), ("rank", ranked)):
        if not torch.equal(out, ref):
            bad = (out != ref).nonzero()
            print("MISMATCH (%s path) case %d n_fft %d S %d n %d kind %s: %d bins, first %s" % (
                name, case, n_fft, S, n, kind, len(bad), bad[0].tolist()))
            sys.exit(1)
    frames += S * n
print("%d cases ok (%d stream-frames), scan path == rank fast path == heap kernel bit for bit" % (n_cases, frames))
