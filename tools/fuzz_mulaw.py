#!/usr/bin/env python3
"""Mu-law codes at the decision boundaries: for every code boundary of the 256- and 64-level quantisers, the inputs
within +-8 ulp of it (where the last bit of log1p decides), plus 4 M random inputs, against the oracle (torch CPU).

The kernel takes log1p correctly rounded (fp64, rounded once); torch's CPU log1p (SLEEF, 1.0 ulp) is occasionally one
ulp off that, and (x_mu + 1) cancels five digits, so such an input can land on the other side of a boundary.  Every
mismatch is checked to be of that kind: the oracle's own log1p differs from the correctly rounded value there."""
import os
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import acids_transforms_amd as A  # noqa: E402
from oracle import oracle as O  # noqa: E402

dev = torch.device("cuda")
for ch in (256, 64):
    mu = ch - 1.0
    # boundaries in the companded domain: (y + 1) / 2 * mu + 0.5 = k  ->  y_k; x_k = sign(y) (exp(|y| log1p(mu)) - 1) / mu
    k = np.arange(1, ch, dtype=np.float64)
    y = (k - 0.5) / mu * 2 - 1
    xk = np.sign(y) * np.expm1(np.abs(y) * np.log1p(mu)) / mu
    xs = []
    for x0 in xk.astype(np.float32):
        v = np.float32(x0)
        lo = v
        for _ in range(8):
            lo = np.nextafter(lo, np.float32(-2))
        cur = lo
        for _ in range(17):
            xs.append(cur)
            cur = np.nextafter(cur, np.float32(2))
    xs = np.array(xs, np.float32)
    rnd = (np.random.RandomState(ch).rand(4_000_000).astype(np.float32) * 2 - 1)
    x = torch.from_numpy(np.concatenate([xs, rnd]))
    m = A.MuLaw(channels=ch).to(dev)
    got = m(x.to(dev)).cpu().numpy()
    ref = O.mulaw_encode(x, ch).numpy()
    bad = np.nonzero(got != ref)[0]
    print("channels %d: %d boundary-neighbourhood inputs + %d random: %d mismatches" % (ch, len(xs), len(rnd), len(bad)))
    for i in bad:
        a = np.float32(np.float32(mu) * np.abs(np.float32(x[i])))
        l_cr = np.float32(np.log1p(np.float64(a)))
        l_torch = np.float32(torch.log1p(torch.full((16,), float(a)))[0].item())
        print("   x=%r got %d ref %d: log1p correctly rounded %r, torch %r" % (float(x[i]), got[i], ref[i], float(l_cr), float(l_torch)))
        assert l_cr != l_torch and abs(int(got[i]) - int(ref[i])) == 1, "unexplained mismatch"
