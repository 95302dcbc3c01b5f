#!/usr/bin/env python3
"""Coefficients of csrc/fastmath.h's atan polynomial: atan(t) = t + t s P(s), s = t^2, t in [0, 1].

Weighted least squares on Chebyshev nodes (a few Lawson re-weightings push it towards the minimax solution), then
the result is checked in emulated fp32 against numpy's float64 atan2 over the four quadrants.
    python tools/fit_atan.py
"""
import numpy as np


def fit(deg):
    n = 4000
    t = 0.5 - 0.5 * np.cos(np.pi * (np.arange(n) + 0.5) / n)          # Chebyshev nodes on [0, 1]
    t = t[t > 1e-4]
    s = t * t
    target = (np.arctan(t) - t) / (t * s)                                # P(s)
    A = np.vander(s, deg + 1, increasing=True)
    w = np.ones_like(t)
    scale = t * s                                                       # error in atan = scale * error in P
    for _ in range(60):
        c, *_ = np.linalg.lstsq(A * (w * scale)[:, None], target * w * scale, rcond=None)
        err = np.abs((A @ c - target) * scale)
        w = w * (0.5 + err / err.max())
    return c, err.max()


def atan2_f32(y, x, c):
    f = np.float32
    ax, ay = np.abs(x), np.abs(y)
    mx, mn = np.maximum(ax, ay), np.minimum(ax, ay)
    t = np.where(mx == 0, f(0), mn * (f(1) / mx)).astype(f)
    s = (t * t).astype(f)
    u = np.full_like(s, f(c[-1]))
    for k in range(len(c) - 2, -1, -1):
        u = (u.astype(np.float64) * s + f(c[k])).astype(f)             # fma: one rounding
    r = ((u * s).astype(f).astype(np.float64) * t + t).astype(f)        # t + t s P: mul, then fma
    r = np.where(ay > ax, f(np.pi / 2) - r, r).astype(f)
    r = np.where(np.signbit(x), f(np.pi) - r, r).astype(f)
    return np.copysign(r, y).astype(f)


if __name__ == "__main__":
    for deg in (6, 7, 8):
        c, e = fit(deg)
        rng = np.random.RandomState(0)
        x = rng.randn(2_000_000).astype(np.float32) * np.float32(10) ** rng.uniform(-3, 3, 2_000_000).astype(np.float32)
        y = rng.randn(2_000_000).astype(np.float32) * np.float32(10) ** rng.uniform(-3, 3, 2_000_000).astype(np.float32)
        got = atan2_f32(y, x, c.astype(np.float32))
        ref = np.arctan2(y.astype(np.float64), x.astype(np.float64))
        print("degree %d in s: fit error %.2e, fp32 max abs error %.3e rad" % (deg, e, np.abs(got - ref).max()))
        print("   ", ", ".join("%.9ef" % v for v in c.astype(np.float32)))
