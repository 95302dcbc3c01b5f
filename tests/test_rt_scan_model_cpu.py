"""The DERIVATION behind the realtime scan path (csrc/pghi.hip rt_scan_frame), checked on the CPU against
the exact-heap C restatement of the reference (oracle/pghi_ref.c, dgt.py:396-466): a plain numpy statement of "levels by
two directional recurrences, parents by the largest level, phases along the parent chains" must give the reference's
phases BIT FOR BIT on every frame it accepts, over tens of thousands of small random strips -- every sparsity pattern,
the frame maximum anywhere, dead sources, exact ties (small-integer magnitudes) -- and must decline, not guess, where
tied levels compete or several unreached bins form an island.  (The HIP implementation is pinned to the heap kernels on
the GPU; this pins the mathematics to the reference.)"""
import numpy as np

from oracle import oracle as O

f32 = np.float32
NEG = f32(-np.inf)


def scan_frame(a, b, ph0, ph1_init, tg0, tg1, fg, abstol):
    """One frame of the strip: a = row f-1 (untouched), b = row f, both clamped.  Returns the phase row, or None where
    the scan path declines."""
    F = len(b)
    kmax = int(np.argmax(b))                       # first index of the maximum
    if not b[kmax] > abstol:
        return ph1_init.copy()
    live = b > abstol
    c = np.where(a > abstol, a, NEG).astype(f32)

    def out(k, x):                                 # level at which (f, k) pops, given the best level arriving from behind
        if k == kmax:
            return b[k]                            # the unmarked seed pops at its own key
        if not live[k]:
            return NEG
        return min(b[k], max(c[k], x))

    X = np.full(F, NEG, f32)
    for k in range(1, F):
        X[k] = out(k - 1, X[k - 1])
    Y = np.full(F, NEG, f32)
    for k in range(F - 2, 0, -1):                  # bin 0 is never reached downward
        Y[k] = out(k + 1, Y[k + 1])
    bmax = b[kmax]
    par = np.zeros(F, np.int32)
    seed_tie = False
    for k in range(F):
        if not live[k]:
            continue
        x, y = X[k], Y[k]
        if k == kmax:                              # its neighbours are S's children whatever else reaches them
            if k - 1 >= 1 and live[k - 1]:
                x = b[k - 1]
            if k >= 1 and k + 1 < F and live[k + 1]:
                y = b[k + 1]
        m = max(c[k], x, y)
        if m != NEG and int(c[k] == m) + int(x == m) + int(y == m) > 1:
            return None                            # competing tie
        if k == kmax and m == bmax:
            seed_tie = True                        # which phase S hands on is the heap's to say: matters only if S has a child
        par[k] = 4 if m == NEG else (1 if c[k] == m else (2 if x == m else 3))
    # ... and only if some source is larger than the frame maximum: S is pushed first and nothing passes an equal key, so
    # with no larger source S is the flood's very first pop and hands on the initial phase
    if seed_tie and ((kmax + 1 < F and par[kmax + 1] == 2) or (kmax - 1 >= 1 and par[kmax - 1] == 3)) and c.max() > bmax:
        return None
    for k in range(F):
        if par[k] == 4 and ((k >= 1 and par[k - 1] == 4) or (k + 1 < F and par[k + 1] == 4)):
            return None                            # an island of several unreached bins: the reference reseeds
    cmax = c[kmax]
    phi_s = f32(ph0[kmax] + f32(0.5) * f32(tg0[kmax] + tg1[kmax])) if cmax > bmax else ph1_init[kmax]
    ph = ph1_init.copy()
    done = np.zeros(F, bool)
    for _ in range(F + 2):
        for k in range(F):
            if par[k] == 1 and not done[k]:
                ph[k] = f32(ph0[k] + f32(0.5) * f32(tg0[k] + tg1[k]))
                done[k] = True
            elif par[k] == 2 and not done[k]:
                j = k - 1
                if j == kmax or done[j]:
                    pp = phi_s if j == kmax else ph[j]
                    ph[k] = f32(pp + f32(0.5) * f32(fg[j] + fg[k]))
                    done[k] = True
        for k in range(F - 1, -1, -1):
            if par[k] == 3 and not done[k]:
                j = k + 1
                if j == kmax or done[j]:
                    pp = phi_s if j == kmax else ph[j]
                    ph[k] = f32(pp - f32(0.5) * f32(fg[j] + fg[k]))
                    done[k] = True
    assert all(done[k] for k in range(F) if 1 <= par[k] <= 3)
    return ph


def run_cases(rng, n_cases, make):
    accepted = declined = 0
    eps = f32(1.1920929e-07)
    for _ in range(n_cases):
        F = int(rng.integers(2, 13))
        n = int(rng.integers(1, 4))
        spec = make(rng, n + 2, F).astype(f32)
        tol = f32(rng.choice([1e-6, 0.05, 0.3, 0.6]))
        prev = rng.uniform(0, 6.28, F).astype(f32)
        noise = rng.standard_normal((n, F)).astype(f32)
        r = O.pghi_realtime(spec[None, :2], spec[None, 2:], prev[None], noise[None], 16, 4, tol=float(tol), gamma=1.0, eps=float(eps))
        want = r["phase"][0]
        tg, fg = r["tgradw"][0], r["fgradw"][0]                     # row r of the reference's padded arrays = row r - 2 here
        s = np.maximum(spec, eps)
        abstol = max(f32(tol * s.max()), eps)
        for f in range(2, n + 2):
            ph0 = prev if f == 2 else want[f - 3]
            init = np.where(s[f] > abstol, f32(0), noise[f - 2]).astype(f32)
            tg0 = tg[f - 3] if f - 1 >= 2 else np.zeros(F, f32)
            got = scan_frame(s[f - 1], s[f], ph0, init, tg0, tg[f - 2], fg[f - 2], abstol)
            if got is None:
                declined += 1
                continue
            accepted += 1
            assert np.array_equal(got, want[f - 2]), (spec.tolist(), float(tol), f, got.tolist(), want[f - 2].tolist())
    return accepted, declined


def test_scan_model_equals_the_reference_flood_on_random_float_strips():
    rng = np.random.default_rng(1)

    def make(rng, R, F):
        m = np.abs(rng.standard_normal((R, F)) + 1j * rng.standard_normal((R, F)))
        return m * (rng.random((R, F)) < rng.choice([1.0, 0.9, 0.6, 0.3]))
    acc, dec = run_cases(rng, 15000, make)
    assert acc > 20000 and dec < 0.25 * (acc + dec), (acc, dec)


def test_scan_model_declines_rather_than_guesses_on_tied_strips():
    """Small-integer magnitudes: ties everywhere.  What the model accepts must still be the reference's bits."""
    rng = np.random.default_rng(2)

    def make(rng, R, F):
        return rng.integers(0, int(rng.choice([3, 5, 9, 40])), (R, F)).astype(np.float64)
    acc, dec = run_cases(rng, 15000, make)
    assert acc > 4000 and dec > 1500, (acc, dec)


def test_scan_model_with_the_maximum_pinned_to_the_edges():
    rng = np.random.default_rng(3)

    def make(rng, R, F):
        m = np.abs(rng.standard_normal((R, F))) + 0.01
        m *= rng.random((R, F)) < 0.8
        for r in range(2, R):
            m[r, int(rng.choice([0, 1, F - 2, F - 1]) % F)] = 5.0 + r
        if rng.random() < 0.5:
            m[:, 0] *= rng.random() < 0.5                               # bin 0 dead in every row, or alive
        return m
    acc, dec = run_cases(rng, 10000, make)
    assert acc > 10000, (acc, dec)


def test_scan_model_on_held_frames():
    """Row f equal to row f-1 (a stationary signal): every source ties with its own bin.  Harmless except at the seed with
    a child: the model must accept most of these frames and still match the reference bit for bit."""
    rng = np.random.default_rng(4)

    def make(rng, R, F):
        row = np.abs(rng.standard_normal(F)) * (rng.random(F) < rng.choice([1.0, 0.7, 0.4]))
        return np.tile(row, (R, 1))
    acc, dec = run_cases(rng, 10000, make)
    assert acc > 15000 and dec < 0.1 * acc, (acc, dec)
