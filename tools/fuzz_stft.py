#!/usr/bin/env python3
"""Random shapes against the oracle: STFT / DGT at n_fft 1024, hop 128 / 256 / 512 -- forward, inverse, fused mel."""
import os
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import acids_transforms_amd as A  # noqa: E402
from oracle import oracle as O  # noqa: E402

dev = torch.device("cuda")
rng = np.random.RandomState(int(os.environ.get("FUZZ_SEED", "0")))
n_cases = int(os.environ.get("FUZZ_CASES", "120"))
worst = 0.0
mods = {}
for i in range(n_cases):
    hop = int(rng.choice([128, 256, 512]))
    cls = A.STFT if rng.rand() < 0.5 else A.DGT
    B = int(rng.randint(1, 10))
    L = int(rng.choice([rng.randint(513, 3000), rng.randint(3000, 30000), 2 * rng.randint(300, 9000)]))
    key = (cls.__name__, hop)
    if key not in mods:
        mods[key] = cls(n_fft=1024, hop_length=hop).to(dev)
    t = mods[key]
    x = torch.from_numpy(rng.randn(B, L).astype(np.float32) * 0.1)
    w, wi = t.window[:1024].cpu(), t.inv_window[:1024].cpu()
    X = t(x.to(dev))
    Xr = O.stft_forward(x, w, 1024, hop)
    e1 = float((X.cpu() - Xr).abs().max() / Xr.abs().max())
    y, yr = t.invert(X).cpu(), O.istft(Xr, wi, 1024, hop)
    assert y.shape == yr.shape, (key, B, L)
    e2 = float((y - yr).abs().max() / yr.abs().max()) if yr.numel() else 0.0
    e3 = 0.0
    if L > 512 and L % 2 == 0:
        mg = A.Magnitude(n_mels=128, mode=None).to(dev)
        if mg.can_fuse_with(t, x.to(dev)):
            f = (t + mg)(x.to(dev))
            s = mg(X)
            e3 = float((f - s).abs().max() / s.abs().max())
    worst = max(worst, e1, e2, e3)
    assert max(e1, e2, e3) < 1e-5, (key, B, L, e1, e2, e3)
print("%d cases ok, worst relative error %.2e" % (n_cases, worst))
