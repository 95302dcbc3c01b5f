#!/usr/bin/env python3
"""Polar / PolarIF in their one-pass forms against their parts run one after the other (the reference's own structure:
spectral_repr.py:441-451, 505-537), on random sizes: FFT sizes with banded default banks, clip counts on both sides of the
one-block-per-clip threshold (64), frame counts around the batches of eight, every contrast / normalisation / IF method.
PolarIF: equal bit for bit (both routes run the same arithmetic); Polar: the magnitude half bit for bit, the phase half and
the inverse within 2e-6 (reciprocal against division in the normalisation, sin / cos of the same angle)."""
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
n_cases = int(os.environ.get("FUZZ_CASES", "60"))
done = 0
for i in range(n_cases):
    n_fft = int(rng.choice([256, 512, 598, 1024, 2048]))
    F = n_fft // 2 + 1
    B = int(rng.choice([1, 3, 17, 64, 65, 96]))
    T = int(rng.choice([1, 2, 7, 8, 9, 16, 23, 40, 41]))
    contrast = [None, "log1p", "log", "log10"][int(rng.randint(4))]
    mode = [None, "unipolar", "bipolar", "gaussian"][int(rng.randint(4))]
    method = ["forward", "backward", "central"][int(rng.randint(3))]
    weighted = bool(rng.randint(2))
    g = torch.Generator().manual_seed(int(rng.randint(1 << 30)))
    X = (torch.randn(B, T, F, generator=g) * torch.exp(2j * np.pi * torch.rand(B, T, F, generator=g))).to(torch.complex64).to(dev)
    tag = (n_fft, B, T, contrast, mode, method, weighted)
    # Polar
    p = A.Polar(magnitude_args={"mode": mode, "n_fft": n_fft, "contrast": contrast}, phase_args={"mode": mode}).to(dev)
    p.scale_data(X)
    y = p(X)
    # (the one-pass Polar normalises its phases by the reciprocal of the scale, Phase.forward divides: <= 1.5 ulp apart)
    assert torch.equal(y[..., 0, :], p.magnitude(X)), ("polar fwd, magnitude half",) + tag
    ph = p.phase(X)
    assert float((y[..., 1, :] - ph).abs().max()) <= 1e-6 * max(1.0, float(ph.abs().max())), ("polar fwd, phase half",) + tag
    back = p.invert(y)
    parts = ops.polar_to_complex(p.magnitude.invert(y[..., 0, :]), p.phase.invert(y[..., 1, :]))
    assert float((back - parts).abs().max()) <= 2e-6 * max(1e-30, float(parts.abs().max())), ("polar inv",) + tag
    # PolarIF
    if T == 1 and (method == "central" or weighted):
        continue
    q = A.PolarIF(magnitude_args={"mode": mode, "n_fft": n_fft, "contrast": contrast},
                  phase_args={"mode": mode, "method": method, "weighted": weighted}).to(dev)
    q.scale_data(X)
    y = q(X)
    with variant("scan_layout", 1):
        y2 = q(X)
    assert torch.equal(y, y2), ("polarif fwd, one pass vs two kernels",) + tag
    assert torch.equal(y[..., 0, :], q.magnitude(X)) and torch.equal(y[..., 1, :], q.phase(X)), ("polarif fwd",) + tag
    back = q.invert(y)
    parts = ops.polar_to_complex(q.magnitude.invert(y[..., 0, :]), q.phase.invert(y[..., 1, :]))
    assert torch.equal(back, parts), ("polarif inv",) + tag
    done += 1
print("fuzz_polar: %d cases (%d with PolarIF), all equal" % (n_cases, done))
