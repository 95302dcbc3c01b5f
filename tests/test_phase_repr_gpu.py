"""GPU parity: phase-side representations (unwrap, IF, Phase, Real/Imaginary, Cartesian/Polar/PolarIF) through
the C ABI (at_phase_scan / at_phase_integrate / at_polar_to_complex) against the reference's own outputs
(tests/golden/g11_phase_repr.npz) and the oracle.

Bars: everything downstream of a *real* input (unwrap / fdiff / fint of given numbers, IF.invert) is plain
IEEE fp32 + a double cumsum accumulator, reproduced operation by operation -> bit-exact.  Paths that start from
a complex spectrum go through atan2f, whose last bit differs between libms -> 1e-5 of the output's largest
magnitude (the north_star fp32 tolerance)."""
import numpy as np
import pytest
import torch

import acids_transforms_amd as A
from acids_transforms_amd import ops
from acids_transforms_amd.utils import misc as M
from conftest import rel_max
from acids_transforms_amd._lib import variant
from oracle import oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-5
T_ = torch.from_numpy
METHODS = ("forward", "backward", "central")


def cpu(t):
    return t.detach().cpu()


def test_scan_helpers_bit_exact(golden, dev):
    g = golden("g11_phase_repr")
    for T in (10, 11):
        r = T_(g["r%d" % T]).to(dev)
        assert torch.equal(cpu(M.unwrap(r * 3.0)), T_(g["unwrap%d" % T]))
        for m in METHODS:
            assert torch.equal(cpu(getattr(M, "fdiff_" + m)(r)), T_(g["fdiff_%s%d" % (m, T)])), (m, T)
            keep = r.clone()
            assert torch.equal(cpu(getattr(M, "fint_" + m)(r)), T_(g["fint_%s%d" % (m, T)])), (m, T)
            assert torch.equal(r, keep)                       # never in place
    one = torch.randn(3, 1, 5, device=dev)
    assert torch.equal(M.fdiff_central(one), torch.cat([one, one], -2))        # single-frame quirk
    assert torch.equal(M.fdiff_forward(one), one) and torch.equal(M.fint_central(one), one)
    # unwrap of the CPU angle (real input): exact
    ang = T_(g["X"]).angle()
    assert torch.equal(cpu(M.unwrap(ang.to(dev))), T_(g["unwrap_angle_X"]))


def test_phase_and_if_golden(golden, dev):
    g = golden("g11_phase_repr")
    for tag in ("X", "Xr"):
        X = T_(g[tag]).to(dev)
        modes = ("none", "bipolar", "gaussian") if tag == "X" else ("gaussian",)
        for mode in modes:
            for unwrap in (0, 1):
                for keep in ((1, 0) if mode == "none" else (1,)):
                    key = "phase_%s_%s_%d_%d" % (tag, mode, unwrap, keep)
                    ph = A.Phase(mode=mode, unwrap=bool(unwrap), keep_nyquist=bool(keep))
                    ph.scale_data(X)
                    if mode != "none":
                        assert abs(float(ph.norm.offset) - float(g[key + "_offset"])) <= 1e-5 * max(1, abs(float(g[key + "_offset"])))
                        assert abs(float(ph.norm.scale) - float(g[key + "_scale"])) <= 1e-5 * abs(float(g[key + "_scale"]))
                    y = ph(X)
                    assert y.shape == g[key].shape
                    assert rel_max(cpu(y).numpy(), g[key]) < TOL, key
                    inv = ph.invert(T_(g[key]).to(dev))
                    assert rel_max(cpu(inv).numpy(), g[key + "_inv"]) < TOL, key
            for m in METHODS:
                for keep in ((1, 0) if mode == "none" else (1,)):
                    key = "if_%s_%s_%s_%d" % (tag, mode, m, keep)
                    f = A.IF(mode=mode, method=m, keep_nyquist=bool(keep))
                    f.scale_data(X)
                    y = f(X)
                    assert y.shape == g[key].shape
                    assert rel_max(cpu(y).numpy(), g[key]) < TOL, key
                    yin = T_(g[key]).to(dev)
                    before = yin.clone()
                    if mode != "none":     # use the reference's statistics: inversion is then exact arithmetic
                        f.norm.set_affine(T_(g[key + "_offset"]).to(dev), T_(g[key + "_scale"]).to(dev))
                    inv = f.invert(yin)
                    assert torch.equal(yin, before)
                    assert torch.equal(cpu(inv), T_(g[key + "_inv"])), key
        for m in METHODS:                 # weighted: first call as the reference, and it keeps working
            f = A.IF(mode="none", method=m, weighted=True)
            for _ in range(2):
                assert rel_max(cpu(f.get_if(X)).numpy(), g["ifw_%s_%s" % (tag, m)]) < TOL
    with pytest.raises(AttributeError):
        A.IF(method="sideways").get_if(X)


def test_real_imag_and_stacked_golden(golden, dev):
    g = golden("g11_phase_repr")
    X = T_(g["X"]).to(dev)
    for mode in ("none", "gaussian"):
        for cls, name in ((A.Real, "real"), (A.Imaginary, "imag")):
            for keep in (1, 0):
                key = "%s_X_%s_%d" % (name, mode, keep)
                r = cls(mode=mode, keep_nyquist=bool(keep))
                r.scale_data(X)
                y = r(X)
                assert rel_max(cpu(y).numpy(), g[key]) < TOL, key
                assert rel_max(cpu(r.invert(T_(g[key]).to(dev))).numpy(), g[key + "_inv"]) < TOL, key
    assert torch.equal(A.Imaginary(mode=None)(X.real.contiguous()), torch.zeros_like(X.real))
    bank = T_(g["bank65"])

    def build(name, stack, keep):
        if name == "cartesian":
            return A.Cartesian(stack=stack, keep_nyquist=keep).to(dev)
        cls = A.Polar if name == "polar" else A.PolarIF
        t = cls(magnitude_args={"mode": "bipolar", "n_fft": 128}, stack=stack, keep_nyquist=keep)
        t.magnitude._set_bank(bank)          # the golden run injected this bank into the torchaudio shim
        return t.to(dev)

    for name in ("cartesian", "polar", "polarif"):
        for stack in (-2, None):
            for keep in ((True, False) if name == "cartesian" else (True,)):
                t = build(name, stack, keep)
                t.scale_data(X)
                y = t(X)
                key = "%s_%s_%d" % (name, "none" if stack is None else "m2", int(keep))
                if stack is None:
                    assert rel_max(cpu(y[0]).numpy(), g[key + "_a"]) < TOL and rel_max(cpu(y[1]).numpy(), g[key + "_b"]) < TOL
                    yin = (T_(g[key + "_a"]).to(dev), T_(g[key + "_b"]).to(dev))
                else:
                    assert y.shape == g[key].shape
                    assert rel_max(cpu(y).numpy(), g[key]) < TOL, key
                    yin = T_(g[key]).to(dev)
                inv = t.invert(yin)
                assert inv.dtype == torch.complex64
                # inversion is judged against the input spectrum's scale (the phases are O(100 rad) unwrapped)
                assert rel_max(cpu(inv).numpy(), g[key + "_inv"]) < 2e-5, key
    with pytest.raises(RuntimeError):
        A.SpectralRepresentation(magnitude_transform=A.Magnitude, phase_transform=A.Phase)


def test_against_oracle_random_shapes(dev):
    """Ragged / tiny shapes and extra batch dims, complex input, all scan modes vs the oracle."""
    gen = torch.Generator().manual_seed(5)
    for shape in [(1, 1, 1), (2, 2, 3), (3, 7, 65), (2, 3, 4, 33), (5, 513), (1, 64, 513), (3, 40, 65), (2, 690, 33), (2, 18, 7)]:
        X = (torch.randn(*shape, generator=gen) * torch.exp(2j * np.pi * torch.rand(*shape, generator=gen))).to(torch.complex64)
        Xd = X.to(dev)
        assert rel_max(cpu(ops.phase_scan(Xd, "angle")).numpy(), X.angle().numpy()) < TOL
        assert rel_max(cpu(ops.phase_scan(Xd, "unwrap")).numpy(), O.unwrap(X.angle()).numpy()) < TOL
        for m in METHODS:
            if m == "central" and shape[-2] == 1:
                continue
            for w in (False, True):
                if w and shape[-2] == 1:
                    continue            # the window's (N^2 - 1) denominator vanishes
                want = O.inst_freq(X, m, weighted=w)
                f = A.IF(mode=None, method=m, weighted=w)
                assert rel_max(cpu(f(Xd)).numpy(), want.numpy()) < TOL, (shape, m, w)
            y = torch.randn(*shape, generator=gen)
            assert torch.equal(cpu(ops.phase_integrate(y.to(dev), m)), O.inst_freq_invert(y, m)), (shape, m)
    mag, ph = torch.rand(4, 9, 17, generator=gen), (torch.rand(4, 9, 17, generator=gen) - 0.5) * 2e4
    want = O.polar_to_complex(mag, ph)
    got = cpu(ops.polar_to_complex(mag.to(dev), ph.to(dev)))
    assert rel_max(torch.view_as_real(got).numpy(), torch.view_as_real(want).numpy()) < TOL
    empty = torch.zeros(0, 4, 5, dtype=torch.complex64, device=dev)
    assert ops.phase_scan(empty, "forward").shape == (0, 4, 5)
    with pytest.raises(IndexError):
        ops.phase_scan(torch.zeros(5, dtype=torch.complex64, device=dev), "unwrap")


@pytest.mark.parametrize("shape", [(64, 19, 513), (70, 8, 257), (65, 1, 513), (64, 17, 1025), (64, 9, 2049), (66, 23, 300)])
def test_clip_per_block_scans_equal_flat_layout_and_oracle(dev, shape):
    """>= 64 clips with rows of >= 256 bins that are not whole 64-byte segments take the one-block-per-clip layout
    (2 or 4 columns per thread, wavefronts in lockstep): bit-identical to the flattened-column layout
    (`variant("scan_layout", 1)`) for every mode, bit-identical to the oracle on real input, 1e-5 from it through atan2."""
    import os
    gen = torch.Generator().manual_seed(sum(shape))
    X = (torch.randn(*shape, generator=gen) * torch.exp(2j * np.pi * torch.rand(*shape, generator=gen))).to(torch.complex64)
    y = torch.randn(*shape, generator=gen) * 3.0
    Xd, yd = X.to(dev), y.to(dev)
    win = torch.rand(shape[-2], generator=gen).to(dev)
    off, sc = torch.tensor(0.3, device=dev), torch.tensor(1.7, device=dev)

    def run():
        out = {"unwrap_c": ops.phase_scan(Xd, "unwrap"), "unwrap_r": M.unwrap(yd)}
        for m in METHODS:
            out["if_" + m] = ops.phase_scan(Xd, m, frame_window=win, offset=off, scale=sc)
            out["fdiff_" + m] = getattr(M, "fdiff_" + m)(yd)
            out["fint_" + m] = getattr(M, "fint_" + m)(yd)
            out["int_" + m] = ops.phase_integrate(yd, m, offset=off, scale=sc)
        return {k: cpu(v) for k, v in out.items()}

    got = run()
    with variant("scan_layout", 1):
        flat = run()
    for k in got:
        assert torch.equal(got[k], flat[k]), (k, shape)
    assert torch.equal(got["unwrap_r"], O.unwrap(y))
    assert rel_max(got["unwrap_c"].numpy(), O.unwrap(X.angle()).numpy()) < TOL
    for m in METHODS:
        assert torch.equal(got["fdiff_" + m], O.fdiff(y, m)), m
        assert torch.equal(got["fint_" + m], O.fint(y, m)), m


def test_full_size_round_trip_properties(dev):
    """BASELINE config-2 size (1024 clips would be 1.4 GB per tensor; 256 clips x 690 frames x 513 bins here):
    size-independent properties instead of an oracle run --
      * fint_forward(fdiff_forward(u)) == u up to rounding, same for backward (exact inverses in exact arithmetic);
      * IF("forward") -> invert reproduces unwrap(angle X);  exp(i unwrap) == exp(i angle);
      * Polar -> invert reproduces X."""
    B, T, F = 256, 690, 513
    gen = torch.Generator(device=dev).manual_seed(11)
    X = torch.view_as_complex(torch.randn(B, T, F, 2, device=dev, generator=gen))
    u = ops.phase_scan(X, "unwrap")
    ang = ops.phase_scan(X, "angle")
    assert float((torch.cos(u) - torch.cos(ang)).abs().max()) < 2e-3      # |u| reaches ~1e3 rad: fp32 ulp ~6e-5
    for m in ("forward", "backward"):
        f = A.IF(mode=None, method=m)
        back = f.invert(f(X))
        assert float((back - u).abs().max()) <= 2e-6 * float(u.abs().max()) + 1e-4, m
        d = getattr(M, "fdiff_" + m)(u)
        assert float((getattr(M, "fint_" + m)(d) - u).abs().max()) <= 2e-6 * float(u.abs().max()) + 1e-4
    del u, ang, back, d
    pol = A.Polar(magnitude_args={"mode": None, "contrast": None, "mel": False}, phase_args={"mode": None}).to(dev)
    Y = pol(X)
    assert Y.shape == (B, T, 2, F)
    Xr = pol.invert(Y)
    assert float((Xr - X).abs().max()) < 1e-5 * float(X.abs().max())


def test_reference_combination_chains(dev):
    """The four chains of the reference's own combination test (test/test_transforms.py:72-78) run through
    ComposeAudioTransform exactly as that test drives them: realtime(), scale_data, forward_with_time, invert."""
    gen = torch.Generator().manual_seed(21)
    raw = (torch.randn(2, 2, 8192, generator=gen) * 0.3).clamp(-0.99, 0.99).to(dev)    # audio range: codes stay < 256
    chains = {
        "stft+magnitude": A.STFT() + A.Magnitude(),
        "stereo+mulaw+onehot": A.Stereo() + A.MuLaw(channels=256) + A.OneHot(n_classes=256),
        "stft+polar": A.STFT() + A.Polar(),
        "overlap+stft": A.OverlapAdd() + A.RealtimeSTFT(),
    }
    for name, t in chains.items():
        t = t.to(dev)
        t.realtime()
        if t.needs_scaling:
            t.scale_data(raw)
        time = torch.zeros(*raw.shape[:-1], device=dev)
        y, tm = t.forward_with_time(raw, time)
        x_inv = t.invert(y)
        assert torch.isfinite(torch.view_as_real(x_inv) if x_inv.is_complex() else x_inv.float()).all(), name
        if name == "stft+polar":
            assert y.shape == (2, 2, 33, 2, 513)
            X = A.STFT().to(dev)(raw)
            pol = t[1]
            assert rel_max(cpu(pol(X)).numpy(), cpu(y).numpy()) < TOL
            # Polar.invert rebuilds the complex spectrum, STFT.invert the audio: round trip within fp32 noise
            # of the 513x513 mel / inverse-mel pair being only approximately inverse -> compare phases instead
            ph = pol.phase.invert(y.select(-2, 1))
            assert rel_max(cpu(torch.cos(ph)).numpy(), cpu(torch.cos(X.angle())).numpy()) < 1e-4
            assert x_inv.shape[-1] == 8192
        if name == "stereo+mulaw+onehot":
            assert y.shape == (2, 2, 8192, 256) and x_inv.shape == raw.shape
            assert float((x_inv - raw).abs().max()) < 0.05            # 8-bit companding error
        if name == "overlap+stft":
            assert y.shape[-1] == 513 and x_inv.shape[:-1] == raw.shape[:-1]


def test_polar_one_pass_equals_parts(dev):
    """Polar.forward with default parts runs as one kernel (banded magnitude + angle written into the stacked
    tensor): same values as Magnitude and Phase run on their own, and as the oracle."""
    gen = torch.Generator().manual_seed(17)
    X = (torch.randn(3, 2, 21, 513, generator=gen) * torch.exp(2j * np.pi * torch.rand(3, 2, 21, 513, generator=gen))).to(torch.complex64)
    Xd = X.to(dev)
    pol = A.Polar().to(dev)
    pol.scale_data(Xd)
    assert pol._one_pass(Xd) is not None
    y = pol(Xd)
    assert y.shape == (3, 2, 21, 2, 513)
    assert rel_max(cpu(y[..., 0, :]).numpy(), cpu(pol.magnitude(Xd)).numpy()) < TOL
    assert rel_max(cpu(y[..., 1, :]).numpy(), cpu(pol.phase(Xd)).numpy()) < TOL
    fwd, _ = O.magnitude_banks(O.melscale_fbanks(513, 0.0, 22050.0, 513, 44100))
    off, sc = O.magnitude_scale_stats(X, "log1p", "bipolar")
    assert rel_max(cpu(y[..., 0, :]).numpy(), O.magnitude_forward(X, fwd, "log1p", off, sc).numpy()) < TOL
    po, ps = O.normalize_stats(X.angle(), "bipolar")
    assert rel_max(cpu(y[..., 1, :]).numpy(), O.affine(X.angle(), po, ps).numpy()) < TOL
    # invert mirrors it: one pass over the stacked tensor == Magnitude.invert, Phase.invert, mag * exp(i phase)
    assert pol._one_pass_invert(y) is not None
    Xi = pol.invert(y)
    parts = ops.polar_to_complex(pol.magnitude.invert(y[..., 0, :]), pol.phase.invert(y[..., 1, :]))
    assert Xi.dtype == torch.complex64 and Xi.shape == X.shape
    assert rel_max(cpu(Xi).numpy(), cpu(parts).numpy()) < TOL
    # variants that do not qualify fall back to the generic path with the same values
    for kw in ({"stack": None}, {"phase_args": {"mode": "bipolar", "unwrap": True}}, {"keep_nyquist": False}):
        p2 = A.Polar(**kw).to(dev)
        p2.scale_data(Xd)
        assert p2._one_pass(Xd) is None
        p2(Xd)


@pytest.mark.parametrize("shape", [(3, 2, 21, 513), (2, 1, 513), (1, 2, 129), (4, 33, 1025), (7, 257), (9, 67, 513), (70, 300),
                                   (3, 40, 513), (2, 690, 129)])
def test_cartesian_and_polarif_work_inside_the_stacked_tensor(dev, shape):
    """Cartesian is one pack / unpack kernel, PolarIF's halves are written into / read from the stacked tensor in place
    (banded magnitude with a row stride, IF scan with a row stride, integration fused with mag * exp(i phase)): the
    same values as the parts run on their own and torch.stack / polar_to_complex."""
    gen = torch.Generator().manual_seed(sum(shape))
    X = (torch.randn(*shape, generator=gen) * torch.exp(2j * np.pi * torch.rand(*shape, generator=gen))).to(torch.complex64)
    Xd = X.to(dev)
    F = shape[-1]
    for kw in ({}, {"real_args": {"mode": None}}, {"imag_args": {"mode": "unipolar"}, "real_args": {"mode": "bipolar"}}):
        c = A.Cartesian(**kw).to(dev)
        c.scale_data(Xd)
        assert c._one_pass_ok(Xd, False)
        y = c(Xd)
        want = torch.stack([c.magnitude(Xd), c.phase(Xd)], -2)
        assert y.shape == shape[:-1] + (2, F) and torch.equal(y, want)
        back = c.invert(y)
        assert back.dtype == torch.complex64
        assert torch.equal(back, torch.complex(c.magnitude.invert(y[..., 0, :]).contiguous(), c.phase.invert(y[..., 1, :]).contiguous()))
        assert rel_max(cpu(back).numpy(), X.numpy()) < TOL
    if len(shape) < 3:
        return
    for method in ("forward", "backward", "central"):
        for weighted in (False, True):
            if shape[-2] == 1 and (method == "central" or weighted):     # one frame: the weight divides by N^2 - 1 = 0
                continue
            p = A.PolarIF(magnitude_args={"mode": "bipolar", "n_fft": 2 * (F - 1)},
                          phase_args={"mode": "gaussian", "method": method, "weighted": weighted}).to(dev)
            p.scale_data(Xd)
            assert p._in_place_parts(F, False) is not None and p._in_place_parts(F, True) is not None
            y = p(Xd)
            assert y.shape == shape[:-1] + (2, F)
            assert torch.equal(y[..., 0, :], p.magnitude(Xd)) and torch.equal(y[..., 1, :], p.phase(Xd)), (method, weighted)
            back = p.invert(y)
            parts = ops.polar_to_complex(p.magnitude.invert(y[..., 0, :]), p.phase.invert(y[..., 1, :]))
            assert back.dtype == torch.complex64 and torch.equal(back, parts), (method, weighted)
    # a variant that does not qualify takes the generic path
    p2 = A.PolarIF(stack=None).to(dev)
    assert p2._in_place_parts(F, False) is None


@pytest.mark.parametrize("shape", [(64, 19, 513), (70, 8, 257), (65, 1, 513), (64, 17, 1025), (66, 23, 300), (64, 3, 2049)])
def test_polarif_forward_one_pass_equals_the_two_kernels(dev, shape):
    """>= 64 clips of 256..4096 bins: PolarIF.forward is ONE kernel (at_polarif_forward: the clip-per-block IF scan with
    the banded magnitude of the same rows summed from LDS).  Both halves bit-identical to the stand-alone kernels
    (`variant("scan_layout", 1)` sends the call down the two-kernel path), every method, weighted or not, every
    contrast, with and without normalisation; the tail batch (T not a multiple of 8) and single-frame clips included."""
    from acids_transforms_amd._lib import variant
    gen = torch.Generator().manual_seed(sum(shape) + 1)
    X = (torch.randn(*shape, generator=gen) * torch.exp(2j * np.pi * torch.rand(*shape, generator=gen))).to(torch.complex64)
    Xd = X.to(dev)
    F = shape[-1]
    cases = [("forward", False, "log1p", "bipolar"), ("backward", True, "log", "unipolar"), ("central", False, "log10", None),
             ("forward", True, None, "gaussian"), ("central", True, "log1p", "unipolar")]
    for method, weighted, contrast, mode in cases:
        if shape[-2] == 1 and (method == "central" or weighted):
            continue
        p = A.PolarIF(magnitude_args={"mode": mode, "n_fft": 2 * (F - 1), "contrast": contrast},
                      phase_args={"mode": "gaussian" if mode else None, "method": method, "weighted": weighted}).to(dev)
        p.scale_data(Xd[:8])
        assert p._in_place_parts(F, False) is not None
        y = p(Xd)
        with variant("scan_layout", 1):
            want = p(Xd)
        assert y.shape == shape[:-1] + (2, F)
        assert torch.equal(y[..., 1, :], want[..., 1, :]), (method, weighted, contrast, mode, "IF half")
        assert torch.equal(y[..., 0, :], want[..., 0, :]), (method, weighted, contrast, mode, "magnitude half")
        assert torch.equal(y[..., 0, :], p.magnitude(Xd)) and torch.equal(y[..., 1, :], p.phase(Xd))
    # the raw entry point refuses what the layout does not cover
    from acids_transforms_amd._lib import lib, ptr
    small = Xd[:8].contiguous()
    st, ln, wo, w = p._in_place_parts(F, False).by_filter(dev)
    out = torch.empty(small.shape[:-1] + (2, F), device=dev)
    rc = lib().at_polarif_forward(ptr(small), 8, shape[-2], F, 1, None, None, None, ptr(st), ptr(ln), ptr(wo), ptr(w),
                                  w.numel(), 1, None, None, 1e-7, ptr(out), None)
    assert rc == -2
    # ... and takes what it does (so the comparisons above were between two different routes)
    out = torch.empty(shape[:-1] + (2, F), device=dev)
    rc = lib().at_polarif_forward(ptr(Xd), shape[0], shape[-2], F, 1, None, None, None, ptr(st), ptr(ln), ptr(wo), ptr(w),
                                  w.numel(), 1, None, None, 1e-7, ptr(out), None)
    assert rc == (0 if F <= 1025 else -2), (F, rc)


def test_compose_stft_polar_is_one_kernel(dev):
    """ComposeAudioTransform(STFT|DGT + Polar) with default parts: framing, FFT, banded magnitude and phase in a
    single kernel that never writes the complex spectrum.  Same values as stage by stage; the STFT stage's
    phase buffer (keep_input inversion) is recovered from the phase half of the result."""
    gen = torch.Generator().manual_seed(23)
    x = torch.randn(2, 3, 10240, generator=gen) * 0.1
    xd = x.to(dev)
    for cls in (A.STFT, A.DGT):
        st, pol = cls().to(dev), A.Polar().to(dev)
        comp = st + pol
        comp.scale_data(xd)
        assert pol.can_fuse_with(st, xd)
        y = comp(xd)                                   # fused
        X = cls().to(dev)(xd)
        y2 = pol(X)                                    # stage by stage (one-pass Polar over the stored spectrum)
        assert y.shape == y2.shape == (2, 3, 41, 2, 513)
        assert rel_max(cpu(y[..., 0, :]).numpy(), cpu(y2[..., 0, :]).numpy()) < TOL
        # phases: +pi and -pi are the same angle (DC / Nyquist are real, the sign of their zero imaginary part is
        # not significant) -> compare the de-normalised halves on the circle
        a = cpu(pol.phase.invert(y[..., 1, :].contiguous()))
        b = cpu(pol.phase.invert(y2[..., 1, :].contiguous()))
        big = cpu(X.abs()) > 1e-2          # the angle of a near-zero bin is ill-conditioned
        assert float(torch.angle(torch.exp(1j * (a - b)))[big].abs().max()) < 1e-4
        # side effect of the STFT stage: phase_buffer == angle(X) where the magnitude is significant
        d = torch.angle(torch.exp(1j * (cpu(st.phase_buffer).reshape(X.shape) - cpu(X).angle())))
        assert float(d[big].abs().max()) < 1e-3
        # keep_input inversion right after the fused forward uses that phase
        mag = X.abs()
        ya = st.invert(mag, inversion_mode="keep_input")
        yb = cls().to(dev).invert(X)
        assert rel_max(cpu(ya).numpy(), cpu(yb).numpy()) < 1e-4
    # not fusable: odd length, stack=None, unwrapped phase
    st = A.STFT().to(dev)
    assert not A.Polar().to(dev).can_fuse_with(st, torch.zeros(2, 9001, device=dev))
    assert not A.Polar(stack=None).to(dev).can_fuse_with(st, xd)
    assert not A.Polar(phase_args={"mode": "bipolar", "unwrap": True}).to(dev).can_fuse_with(st, xd)


@pytest.mark.gpu
def test_fast_atan2_accuracy_and_edge_cases(dev):
    """csrc/fastmath.h: every angle in the library.  <= 5e-7 rad against float64 atan2 over all quadrants and twelve
    decades of magnitude (the parity bar is 1e-5); libm's answers, bit for bit, for zeros, signed zeros, denormal,
    huge and infinite arguments (those take libm's atan2f)."""
    rng = np.random.RandomState(5)
    n = 1 << 20
    re = (rng.randn(n) * 10.0 ** rng.uniform(-6, 6, n)).astype(np.float32)
    im = (rng.randn(n) * 10.0 ** rng.uniform(-6, 6, n)).astype(np.float32)
    got = ops.angle(torch.complex(torch.from_numpy(re), torch.from_numpy(im)).to(dev)).cpu().numpy()
    want = np.arctan2(im.astype(np.float64), re.astype(np.float64))
    assert np.abs(got - want).max() < 5e-7
    # axes and diagonals: exact multiples of pi/4 up to the rounding of the constants
    ax = np.array([1, 1, 0, -1, -1, -1, 0, 1], np.float32)
    ay = np.array([0, 1, 1, 1, 0, -1, -1, -1], np.float32)
    g = ops.angle(torch.complex(torch.from_numpy(ax), torch.from_numpy(ay)).to(dev)).cpu().numpy()
    assert np.abs(g - np.arctan2(ay.astype(np.float64), ax.astype(np.float64))).max() < 3e-7
    # special arguments: same bits as libm (signed zeros, denormals, huge, inf)
    sx = np.array([0.0, -0.0, 0.0, -0.0, 1e-41, -1e-41, 3e38, -3e38, np.inf, -np.inf, 1.0, -1.0], np.float32)
    sy = np.array([0.0, 0.0, -0.0, -0.0, 1e-42, 1e-40, 3e38, 1e38, 1.0, -1.0, np.inf, -np.inf], np.float32)
    gs = ops.angle(torch.complex(torch.from_numpy(sx), torch.from_numpy(sy)).to(dev)).cpu().numpy()
    ws = np.arctan2(sy, sx)                                  # float32 libm on the host
    assert np.allclose(gs, ws, rtol=0, atol=3e-7)
    assert np.array_equal(np.signbit(gs), np.signbit(ws))
