#!/usr/bin/env python3
"""Banded walk against the dense MFMA contraction of the same module (the dense path is the one the goldens pin):
Magnitude over random FFT sizes up to 4096, mel counts, contrasts, normalisations, keep_nyquist, extra batch dims."""
import os
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import acids_transforms_amd as A  # noqa: E402

dev = torch.device("cuda")
rng = np.random.RandomState(int(os.environ.get("FUZZ_SEED", "0")))
n_cases = int(os.environ.get("FUZZ_CASES", "120"))
worst = 0.0
n_banded = 0
for i in range(n_cases):
    n = int(rng.choice([rng.randint(16, 4300), 2 ** rng.randint(4, 13), 2048, 4096, 1536, 3000]))
    F = n // 2 + 1
    keep = bool(rng.rand() < 0.7)
    n_mels = rng.choice([None, 40, 80, 128, 256, int(rng.randint(1, F + 1))])
    contrast = rng.choice(["log1p", "log", "log10", None])
    mode = rng.choice(["unipolar", "bipolar", "gaussian", None])
    kw = dict(n_fft=n, contrast=contrast, mode=mode, keep_nyquist=keep)
    if n_mels is not None:
        kw["n_mels"] = int(min(n_mels, F))
    shape = tuple(int(v) for v in rng.randint(1, 4, size=rng.randint(1, 3))) + (int(rng.randint(1, 40)), F)
    X = torch.from_numpy((rng.randn(*shape) + 1j * rng.randn(*shape)).astype(np.complex64)).to(dev)
    a, b = A.Magnitude(**kw).to(dev), A.Magnitude(**kw).to(dev)
    b._band_of = lambda name: None                      # dense contraction
    a.scale_data(X)
    b.scale_data(X)
    if a._band_of("mel_bank") is not None:
        n_banded += 1
    ya, yb = a(X), b(X)
    assert ya.shape == yb.shape, (kw, shape)
    d = float(yb.abs().max())
    e1 = float((ya - yb).abs().max()) / d if d > 0 else 0.0
    xa, xb = a.invert(yb), b.invert(yb)
    d = float(xb.abs().max())
    e2 = float((xa - xb).abs().max()) / d if d > 0 else 0.0
    worst = max(worst, e1, e2)
    assert max(e1, e2) < 2e-5, (kw, shape, e1, e2)
print("%d cases ok (%d on the banded walk), worst relative difference %.2e" % (n_cases, n_banded, worst), flush=True)
