#!/usr/bin/env python3
"""Offline PGHI pop order against the exact-heap C oracle on random shapes, with quantised magnitudes (many exact ties:
the heap's tie-breaking is part of the contract), sparse spectra (reseeds) and batches of unequal clips."""
import os
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acids_transforms_amd import ops  # noqa: E402
from oracle import oracle as O  # noqa: E402

dev = torch.device("cuda")
rng = np.random.RandomState(int(os.environ.get("FUZZ_SEED", "0")))
n_cases = int(os.environ.get("FUZZ_CASES", "60"))
if os.environ.get("FUZZ_PGHI_KERNEL"):      # 1: winner-bit kernel, 2: single-lane kernel (C ABI at_set_variant)
    from acids_transforms_amd._lib import lib, check, VARIANTS  # noqa: E402
    check(lib().at_set_variant(VARIANTS["pghi_kernel"], int(os.environ["FUZZ_PGHI_KERNEL"])), "at_set_variant")
pops = 0
for i in range(n_cases):
    n_fft = int(rng.choice([32, 128, 512, 1024]))
    hop = n_fft // int(rng.choice([2, 4, 8]))
    F = n_fft // 2 + 1
    T = int(rng.randint(1, 70 if n_fft >= 512 else 200))
    B = int(rng.randint(1, 6))
    kind = rng.choice(["noise", "ties", "sparse", "smooth"])
    m = np.abs(rng.randn(B, T, F) + 1j * rng.randn(B, T, F)).astype(np.float32)
    if kind == "ties":
        m = np.round(m * 4) / 4 + 0.25
    elif kind == "sparse":
        m = m * (rng.rand(B, T, F) < 0.05) + 1e-6
    elif kind == "smooth":
        m = (np.exp(-((np.arange(F)[None, None] - F * rng.rand(B, 1, 1)) / (F / 6)) ** 2) *
             (1 + 0.1 * np.sin(np.arange(T)[None, :, None] / 3.0))).astype(np.float32)
    m = m.astype(np.float32)
    gamma = float(O.gamma_offline(n_fft))
    ph, npops, order = ops.pghi_offline(torch.from_numpy(m).to(dev), gamma, n_fft, hop, 1e-2, debug=True)
    torch.cuda.synchronize()
    for b in range(B):
        r = O.pghi_offline(m[b], n_fft, hop, tol=np.float32(1e-2), want_order=True)
        n = int(npops[b])
        o = order[b, :n].cpu().numpy()
        got = np.stack([o // F, o % F], 1)
        assert n == len(r["order"]) and np.array_equal(got, r["order"]), (i, b, n_fft, hop, T, kind, n, len(r["order"]))
        assert np.array_equal(ph[b].cpu().numpy() == 0, r["phase"] == 0)
        pops += n
print("%d cases ok, %d pops in identical order" % (n_cases, pops))
