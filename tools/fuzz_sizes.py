#!/usr/bin/env python3
"""Random FFT sizes against the oracle: any n_fft in [2, 5000] (and a few larger ones), any hop, odd / even / prime
sizes, STFT and DGT windows -- forward, complex and polar inverse, Magnitude (banded and dense banks),
and DGT.invert(|X|, "pghi") pop for pop on small cases.  FUZZ_SEED / FUZZ_CASES select the run."""
import os
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import acids_transforms_amd as A  # noqa: E402
from acids_transforms_amd import ops  # noqa: E402
from oracle import oracle as O  # noqa: E402

dev = torch.device("cuda")
rng = np.random.RandomState(int(os.environ.get("FUZZ_SEED", "0")))
n_cases = int(os.environ.get("FUZZ_CASES", "150"))
TOL = 1e-5


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    d = float(np.abs(b).max()) if b.size else 0.0
    return float(np.abs(a - b).max() / d) if d > 0 else float(np.abs(a - b).max() if a.size else 0.0)


worst = 0.0
kinds = {}
for i in range(n_cases):
    pick = rng.rand()
    if pick < 0.15:
        n = int(2 ** rng.randint(1, 14))
    elif pick < 0.3:
        n = int(rng.choice([400, 441, 480, 600, 882, 960, 1000, 1200, 1764, 1920, 2000, 2400, 3000, 3072, 4410, 4800]))
    elif pick < 0.9:
        n = int(rng.randint(2, 5001))
    else:
        n = int(rng.randint(5001, 12001)) // 2 * 2
    if n > 8191 and n % 2:
        n -= 1
    # hops up to n / 2: with less overlap the window envelope torch.istft divides by comes arbitrarily close to zero at
    # the frame edges and both sides amplify their rounding noise there (ill-conditioned, not a parity question)
    hop = int(rng.choice([max(1, n // 4), max(1, n // 2), max(1, n // 8), rng.randint(1, max(2, n // 2 + 1))]))
    cls = A.STFT if rng.rand() < 0.5 else A.DGT
    B = int(rng.randint(1, 4))
    L = int(rng.randint(n // 2 + 2, max(n // 2 + 3, 6 * n + 50)))
    if L // hop > 3000:                      # keep the frame count (and the oracle's time) bounded
        hop = max(hop, L // 3000)
    x = torch.from_numpy(rng.randn(B, L).astype(np.float32) * 0.1)
    tag = (cls.__name__, n, hop, B, L)
    try:
        t = cls(n_fft=n, hop_length=hop).to(dev)
    except Exception as exc:             # sizes the host class itself refuses (the reference does, too)
        kinds["ctor:" + type(exc).__name__] = kinds.get("ctor:" + type(exc).__name__, 0) + 1
        continue
    w, wi = t.window[:n].cpu(), t.inv_window[:n].cpu()
    X = t(x.to(dev))
    Xr = O.stft_forward(x, w, n, hop)
    assert X.shape == Xr.shape, (tag, X.shape, Xr.shape)
    e1 = rel(X.cpu().numpy(), Xr.numpy())
    e2 = e3 = e4 = e5 = 0.0
    if Xr.shape[-2] > 1 and hop <= n:
        try:
            yr = O.istft(Xr, wi, n, hop)
        except RuntimeError:                 # torch.istft's NOLA check: the module raises the same way
            yr = None
            try:
                t.invert(X)
                raise AssertionError(("NOLA accepted", tag))
            except RuntimeError:
                kinds["nola"] = kinds.get("nola", 0) + 1
        if yr is not None:
            y = t.invert(X).cpu()
            assert y.shape == yr.shape, (tag, y.shape, yr.shape)
            e2 = rel(y.numpy(), yr.numpy())
            yp = t._istft(mag=X.abs(), phase=X.angle()).cpu()
            e3 = rel(yp.numpy(), yr.numpy()) / 2
    # Magnitude over the spectrum (default bank of F filters and a mel-40 one)
    F = n // 2 + 1
    if 8 <= F <= 2600 and rng.rand() < 0.5:
        n_mels = int(rng.choice([F, min(40, F), min(128, F)]))
        mg = A.Magnitude(n_fft=n, n_mels=n_mels).to(dev)
        mg.scale_data(X)
        f_max = float((torch.arange(F) / n * 44100)[-1])     # spectral_repr.py:174-178: the last bin's frequency
        fwd, inv = O.magnitude_banks(O.melscale_fbanks(F, 0.0, f_max, n_mels, 44100))
        off, sc = O.magnitude_scale_stats(X.cpu(), "log1p", "unipolar")
        mr = O.magnitude_forward(X.cpu(), fwd, "log1p", off, sc)
        m = mg(X)
        e4 = rel(m.cpu().numpy(), mr.numpy())
        e5 = rel(mg.invert(m).cpu().numpy(), O.magnitude_invert(mr, inv, "log1p", off, sc).numpy()) / 2
        kinds["mag_banded" if mg._band_of("mel_bank") is not None else "mag_dense"] = \
            kinds.get("mag_banded" if mg._band_of("mel_bank") is not None else "mag_dense", 0) + 1
    # PGHI pop order on small spectra
    if cls is A.DGT and X.shape[-2] * F <= 40000 and X.shape[-2] >= 3 and rng.rand() < 0.5:
        mags = X.abs()
        ph, npops, order = ops.pghi_offline(mags, float(t.gamma), n, hop, float(t.tolerance), float(t.eps), debug=True)
        for b in range(B):
            r = O.pghi_offline(mags[b].cpu(), n, hop, want_order=True)
            k = len(r["order"])
            assert int(npops[b]) == k, (tag, "pops", int(npops[b]), k)
            assert np.array_equal(order[b][:k].cpu().numpy(), r["order"][:, 0] * F + r["order"][:, 1]), (tag, "order")
        kinds["pghi"] = kinds.get("pghi", 0) + 1
    kinds["pow2" if n & (n - 1) == 0 else ("odd" if n % 2 else "even")] = kinds.get(
        "pow2" if n & (n - 1) == 0 else ("odd" if n % 2 else "even"), 0) + 1
    worst = max(worst, e1, e2, e3, e4, e5)
    assert max(e1, e2, e3, e4, e5) < TOL, (tag, e1, e2, e3, e4, e5)
print("%d cases ok, worst relative error %.2e; %s" % (n_cases, worst, kinds), flush=True)
