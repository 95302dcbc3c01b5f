#!/usr/bin/env python3
"""Realtime PGHI against the C oracle on random sizes (streams, frames per chunk, n_fft), with tied and sparse
magnitudes.  The serial single-lane kernel (FUZZ_PGHI_KERNEL=2 in a second process: at_set_variant) must agree bit for bit, the
oracle within the tests' phase tolerance."""
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
if os.environ.get("FUZZ_PGHI_KERNEL"):      # 2: single-lane kernels (C ABI at_set_variant)
    from acids_transforms_amd._lib import lib, check, VARIANTS  # noqa: E402
    check(lib().at_set_variant(VARIANTS["pghi_kernel"], int(os.environ["FUZZ_PGHI_KERNEL"])), "at_set_variant")
out = []
for i in range(n_cases):
    n_fft = int(rng.choice([32, 128, 1024]))
    hop = n_fft // 4
    F = n_fft // 2 + 1
    S, n = int(rng.randint(1, 5)), int(rng.randint(1, 9))
    kind = rng.choice(["noise", "ties", "sparse"])
    mk = lambda *shape: np.abs(rng.randn(*shape) + 1j * rng.randn(*shape)).astype(np.float32)
    mh, m = mk(S, 2, F), mk(S, n, F)
    if kind == "ties":
        mh, m = np.round(mh * 4) / 4 + 0.25, np.round(m * 4) / 4 + 0.25
    elif kind == "sparse":
        mh, m = mh * (rng.rand(S, 2, F) < 0.1) + 1e-6, m * (rng.rand(S, n, F) < 0.1) + 1e-6
    mh, m = mh.astype(np.float32), m.astype(np.float32)
    pp = (rng.rand(S, F).astype(np.float32) - 0.5) * 6
    nz = rng.randn(S, n, F).astype(np.float32)
    gamma = float(O.gamma_realtime(n_fft))
    T_ = lambda a: torch.from_numpy(a).to(dev)
    ph = ops.pghi_realtime(T_(mh), T_(m), T_(pp), T_(nz), gamma, n_fft, hop, 1e-2).cpu().numpy()
    r = O.pghi_realtime(mh, m, pp, nz, n_fft, hop, tol=1e-2)["phase"]
    tol = 2e-3 + 16 * np.spacing(np.abs(r).astype(np.float32)) + 2e-6 * np.abs(r)
    assert np.all(np.abs(ph - r) <= tol), (i, n_fft, S, n, kind, float(np.abs(ph - r).max()))
    out.append(ph)
np.save(os.environ.get("FUZZ_OUT", "/tmp/rt_fuzz.npy"), np.concatenate([o.ravel() for o in out]))
print("%d cases ok" % n_cases)
