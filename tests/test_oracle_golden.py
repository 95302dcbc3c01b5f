"""Pin the oracle (oracle/oracle.py + oracle/pghi_ref.c) against outputs of the
reference itself (tests/golden/*.npz, made by tests/golden/make_golden.py).
CPU only."""
import numpy as np
import pytest
import torch

from conftest import rel_max
from oracle import oracle as O

T_ = torch.from_numpy

T = torch.from_numpy


def test_windows_and_constants(golden):
    g = golden("g1_constants")
    for (n, h) in [(1024, 256), (512, 128), (128, 32), (32, 8), (2048, 512), (64, 16)]:
        k = "%d_%d" % (n, h)
        assert np.array_equal(O.hann_window(n).numpy(), g["hann_" + k])
        gw = O.gauss_window(n)
        assert np.allclose(gw.numpy(), g["gauss_" + k], rtol=3e-7, atol=0)     # exp(): <= 1 ulp across host CPUs
        assert np.allclose(O.dual_window(gw, n, h).numpy(), g["dual_" + k], rtol=1e-6, atol=0)
        assert np.array_equal(O.gamma_offline(n).numpy(), g["gamma_dgt_" + k])
        assert np.array_equal(O.gamma_offline(n).numpy(), g["gamma_stft_" + k])
        assert np.array_equal(O.gamma_realtime(n).numpy(), g["gamma_rt_" + k])
        assert float(O.oadd_gain(n, h)) == float(g["oadd_gain_" + k])
    assert np.float32(O.EPS) == g["eps"]


@pytest.mark.parametrize("name,n,h", [("stft", 1024, 256), ("stft", 128, 32), ("dgt", 1024, 256), ("dgt", 128, 32)])
def test_stft_istft(golden, name, n, h):
    g = golden("g2_stft")
    x = T(g["x"])
    if name == "stft":
        w = O.hann_window(n)
        iw = w
    else:
        w = O.gauss_window(n)
        iw = O.dual_window(w, n, h)
    k = "%s_%d_%d" % (name, n, h)
    X = O.stft_forward(x, w, n, h)
    assert rel_max(X.numpy(), g["X_" + k]) < 3e-6       # same torch.stft; last-bit differences between host CPUs
    y = O.istft(X, iw, n, h)
    assert rel_max(y.numpy(), g["y_" + k]) < 3e-6


@pytest.mark.parametrize("name,h", [("stft", 128), ("stft", 512), ("dgt", 128), ("dgt", 512)])
def test_stft_istft_other_hops(golden, name, h):
    """G14: the reference itself at n_fft = 1024, hop 128 / 512 (the hop-dependent DGT dual window included)."""
    g = golden("g14_other_hops")
    x = T(g["x"])
    w = O.hann_window(1024) if name == "stft" else O.gauss_window(1024)
    iw = w if name == "stft" else O.dual_window(w, 1024, h)
    k = "%s_%d" % (name, h)
    assert rel_max(iw.numpy(), g["inv_window_" + k]) < 1e-6
    X = O.stft_forward(x, w, 1024, h)
    assert X.shape == g["X_" + k].shape and rel_max(X.numpy(), g["X_" + k]) < 3e-6
    y = O.istft(X, iw, 1024, h)
    assert y.shape == g["y_" + k].shape and rel_max(y.numpy(), g["y_" + k]) < 3e-6


@pytest.mark.parametrize("n,h", [(2048, 512), (512, 128), (4096, 1024), (256, 64), (400, 160), (1000, 250), (441, 147)])
def test_stft_istft_other_sizes(golden, n, h):
    """G16: the reference itself at other FFT sizes -- powers of two, 400 / 1000, the odd size 441 -- windows, dual
    windows, forward and inverse of STFT and DGT."""
    g = golden("g16_other_sizes")
    x = T(g["x_%d" % n])
    for name in ("stft", "dgt"):
        w = O.hann_window(n) if name == "stft" else O.gauss_window(n)
        iw = w if name == "stft" else O.dual_window(w, n, h)
        k = "%s_%d" % (name, n)
        assert rel_max(w.numpy(), g["window_" + k]) < 1e-6 and rel_max(iw.numpy(), g["inv_window_" + k]) < 1e-6
        X = O.stft_forward(x, w, n, h)
        assert X.shape == g["X_" + k].shape and rel_max(X.numpy(), g["X_" + k]) < 3e-6
        y = O.istft(X, iw, n, h)
        assert y.shape == g["y_" + k].shape and rel_max(y.numpy(), g["y_" + k]) < 3e-6


def test_pghi_and_magnitude_other_sizes(golden):
    """G16: the reference's PGHI at n_fft 400 (phase and reconstruction) and Magnitude at n_fft 2048 (its own bank)."""
    g = golden("g16_other_sizes")
    mag = g["pghi_mag_400"][0]
    r = O.pghi_offline(mag, 400, 100)
    ref = g["pghi_phase_400"]
    assert np.array_equal(r["phase"] == 0, ref == 0)
    assert np.all(np.abs(r["phase"] - ref) <= 1e-3 + 1e-5 * np.abs(ref))
    bank = torch.zeros(1025, 1025)              # the bank the reference built (through the shim: one bin per filter)
    idx = torch.from_numpy(g["mag2048_bank_idx"]).long()
    bank[idx[:, 0], idx[:, 1]] = T(g["mag2048_bank_val"])
    fwd, inv = O.magnitude_banks(bank)
    X = T(g["mag2048_X"])
    off, sc = float(g["mag2048_offset"]), float(g["mag2048_scale"])
    o2, s2 = O.magnitude_scale_stats(X, "log1p", "unipolar")
    assert abs(float(o2) - off) < 1e-6 and abs(float(s2) - sc) < 1e-5
    y = O.magnitude_forward(X, fwd, "log1p", off, sc)
    assert rel_max(y.numpy(), g["mag2048_y"]) < 3e-6
    yi = O.magnitude_invert(T(g["mag2048_y"]), inv, "log1p", off, sc)
    assert rel_max(yi.numpy(), g["mag2048_inv"]) < 1e-5


def test_stft_multidim_and_time(golden):
    g = golden("g2_stft")
    x = T(g["x_md"])
    X = O.stft_forward(x, O.hann_window(1024), 1024, 256)
    assert X.shape == g["X_md"].shape and rel_max(X.numpy(), g["X_md"]) < 3e-6
    assert rel_max(O.istft(X, O.hann_window(1024), 1024, 256).numpy(), g["y_md"]) < 3e-6
    tt = O.forward_with_time(17, 256, 44100, T(g["fwt_time_in"]))
    assert np.allclose(tt.numpy(), g["fwt_time_out"], rtol=1e-6)


def test_keep_input(golden):
    g = golden("g2_stft")
    X = T(g["X_stft_1024_256"])
    y = O.polar_istft(X.abs(), T(g["phase_buffer_stft_1024_256"]), O.hann_window(1024), 1024, 256)
    assert rel_max(y.numpy(), g["y_keep_input"]) < 3e-6


PGHI_CASES = ["n12x17", "t12x17", "s12x17", "n40x65", "t40x65", "d40x65", "s40x65", "n64x257", "d64x257",
              "const6x17", "one1x17", "zero5x17"]


@pytest.mark.parametrize("case", PGHI_CASES)
def test_pghi_offline_exact_order(golden, case):
    g = golden("g4_pghi_offline")
    n_fft, hop, tol = g[case + "_params"]
    r = O.pghi_offline(g[case + "_mag"], int(n_fft), int(hop), tol=np.float32(tol), want_order=True)
    # (ii) pop order is bit exact (depends only on magnitude compares + heap tie-breaks)
    assert np.array_equal(r["order"], g[case + "_order"])
    # visited mask: phase exactly 0 where the reference left it at 0
    assert np.array_equal(r["phase"] == 0, g[case + "_phase"] == 0)
    # gradients: glibc logf vs torch's vectorised log differ by <= ~1 ulp of log|s|
    assert np.allclose(r["tgradw"], g[case + "_tgradw"], rtol=0, atol=2e-5)
    assert np.allclose(r["fgradw"], g[case + "_fgradw"], rtol=1e-5, atol=2e-4)
    # (iii) phase within 1e-3 + 8 ulp(|phase_ref|)
    ref = g[case + "_phase"]
    tol_arr = 1e-3 + 8 * np.spacing(np.abs(ref).astype(np.float32)) + 2e-6 * np.abs(ref)
    assert np.all(np.abs(r["phase"] - ref) <= tol_arr)


def test_pghi_real_audio(golden):
    g = golden("g10_agogo")
    r = O.pghi_offline(g["mag_dgt"], 1024, 256)
    ref = g["phase_pghi"]
    assert np.array_equal(r["phase"] == 0, ref == 0)
    tol_arr = 5e-3 + 16 * np.spacing(np.abs(ref).astype(np.float32)) + 1e-5 * np.abs(ref)
    assert np.all(np.abs(r["phase"] - ref) <= tol_arr)


@pytest.mark.parametrize("tag", ["k1", "k2", "k3"])
def test_rtpghi_kernel(golden, tag):
    g = golden("g5_rtpghi_kernel")
    n, h = [int(v) for v in g[tag + "_params"]]
    r = O.pghi_realtime(g[tag + "_magbuf"], g[tag + "_mag"], g[tag + "_phasebuf"], g[tag + "_noise"], n, h)
    assert np.allclose(r["tgradw"], g[tag + "_tgradw"], rtol=1e-5, atol=3e-5)
    assert np.allclose(r["fgradw"], g[tag + "_fgradw"], rtol=1e-5, atol=2e-3)
    ref = g[tag + "_phase"]
    tol_arr = 2e-3 + 16 * np.spacing(np.abs(ref).astype(np.float32)) + 1e-5 * np.abs(ref)
    assert np.all(np.abs(r["phase"] - ref) <= tol_arr)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_rtpghi_per_hop(golden, tag):
    """The reference's RealtimeDGT called one frame at a time (G15): the oracle at n = 1, state carried by the
    oracle's own update_buffers restatement but re-seated on the reference's after every hop."""
    g = golden("g15_rtpghi_per_hop")
    n, h, nsteps, chunk = [int(v) for v in g[tag + "_params"]]
    F = n // 2 + 1
    magbuf, phasebuf = np.zeros((2, 2, F), np.float32), np.zeros((2, F), np.float32)
    dw = O.dual_window(O.gauss_window(n), n, h)
    fa = O.OverlapAddState(n, h)
    x = T(g[tag + "_x"])
    frames = torch.cat([fa.forward(x[:, c:c + chunk]) for c in range(0, x.shape[-1], chunk)], -2)
    assert np.array_equal(frames.numpy(), g[tag + "_frames"])
    for j in range(nsteps):
        X = O.rt_forward(frames[:, j:j + 1], O.gauss_window(n))
        assert rel_max(X.abs().numpy(), g["%s_mag_%d" % (tag, j)]) < 3e-6
        mag = g["%s_mag_%d" % (tag, j)]
        r = O.pghi_realtime(magbuf, mag, phasebuf, g["%s_noise_%d" % (tag, j)], n, h)
        yf = O.rt_invert(O.polar_to_complex(T(mag), T(r["phase"])), dw).numpy()
        ref = g["%s_yframes_%d" % (tag, j)]
        snr = 10 * np.log10((ref ** 2).sum() / max(((yf - ref) ** 2).sum(), 1e-30))
        assert snr > 40.0, (j, snr)
        _, mh, pp = O.rt_update_buffers(mag, r["phase"], magbuf)
        assert np.allclose(np.asarray(mh), g["%s_magbuf_%d" % (tag, j)], rtol=1e-5, atol=1e-6)
        magbuf, phasebuf = g["%s_magbuf_%d" % (tag, j)], g["%s_phasebuf_%d" % (tag, j)]


@pytest.mark.parametrize("key", ["1024_256_4096", "1024_256_1024", "64_16_128"])
def test_overlap_add_stream(golden, key):
    g = golden("g6_overlap_add")
    n, h, chunk = [int(v) for v in key.split("_")]
    x = T(g["x_" + key])
    fa, fi = O.OverlapAddState(n, h), O.OverlapAddState(n, h)
    w = O.hann_window(n)
    gw = O.gauss_window(n)
    dw = O.dual_window(gw, n, h)
    for c in range(3):
        fr = fa.forward(x[:, c * chunk:(c + 1) * chunk])
        X = O.rt_forward(fr, w)
        yf = O.rt_invert(X, w)
        y = fi.invert(yf)
        assert rel_max(y.numpy(), g["y_%s_%d" % (key, c)]) < 3e-6
        assert rel_max(fi.outbuf.numpy(), g["outbuf_%s_%d" % (key, c)]) < 3e-6
        assert np.array_equal(fa.inbuf.numpy(), g["inbuf_%s_%d" % (key, c)])
        if ("frames_%s_%d" % (key, c)) in g:
            assert np.array_equal(fr.numpy(), g["frames_%s_%d" % (key, c)])
            assert rel_max(X.numpy(), g["X_%s_%d" % (key, c)]) < 3e-6
            assert rel_max(yf.numpy(), g["yframes_%s_%d" % (key, c)]) < 3e-6
            Xd = O.rt_forward(fr, gw)
            assert rel_max(Xd.numpy(), g["Xd_%s_%d" % (key, c)]) < 3e-6
            assert np.allclose(O.rt_invert(Xd, dw).numpy(), g["ydframes_%s_%d" % (key, c)], rtol=3e-6, atol=3e-7)


def test_magnitude_all_modes(golden):
    g = golden("g7_magnitude")
    fwd, inv = O.magnitude_banks(T(g["bank"]))
    assert np.array_equal(fwd.numpy(), g["mel_bank"])
    assert np.array_equal(inv.numpy(), g["inverse_mel_bank"])
    X = T(g["X"])
    for c in ["log1p", "log", "log10", "none"]:
        for mode in ["unipolar", "bipolar", "gaussian", "none"]:
            for mel in [1, 0]:
                k = "%s_%s_%d" % (c, mode, mel)
                off = sc = None
                if mode != "none":
                    off, sc = O.magnitude_scale_stats(X, c, mode)
                    assert np.allclose(off.numpy(), g["offset_" + k], rtol=3e-6, atol=3e-7)
                    assert np.allclose(sc.numpy(), g["scale_" + k], rtol=3e-6, atol=3e-7)
                y = O.magnitude_forward(X, fwd, c, off, sc, mel=bool(mel))
                assert rel_max(y.numpy(), g["y_" + k]) < 2e-6, k
                xi = O.magnitude_invert(y, inv, c, off, sc, mel=bool(mel))
                assert rel_max(xi.numpy(), g["inv_" + k]) < 5e-6, k
    assert rel_max(O.magnitude_forward(X[0], fwd, "log1p").numpy(), g["y_2d"]) < 2e-6
    assert rel_max(O.magnitude_forward(X[0, 0], fwd, "log1p").numpy(), g["y_1d"]) < 2e-6
    fwd, inv = O.magnitude_banks(T(g["bank513"]))
    assert np.array_equal(fwd.numpy(), g["mel_bank513"])
    X = T(g["X513"])
    off, sc = O.magnitude_scale_stats(X, "log1p", "unipolar")
    y = O.magnitude_forward(X, fwd, "log1p", off, sc)
    assert rel_max(y.numpy(), g["y513"]) < 2e-6
    assert rel_max(O.magnitude_invert(y, inv, "log1p", off, sc).numpy(), g["inv513"]) < 5e-6


def test_compose_stft_magnitude(golden):
    g = golden("g7_compose")
    x = T(g["x"])
    fwd, inv = O.magnitude_banks(T(g["bank"]))
    X = O.stft_forward(x, O.hann_window(1024), 1024, 256)
    off, sc = O.magnitude_scale_stats(X, "log1p", "unipolar")
    assert np.allclose(off.numpy(), g["offset"], rtol=1e-6, atol=1e-7) and np.allclose(sc.numpy(), g["scale"], rtol=1e-6)
    y = O.magnitude_forward(X, fwd, "log1p", off, sc)
    assert rel_max(y.numpy(), g["y"]) < 2e-6
    assert rel_max(O.magnitude_invert(y, inv, "log1p", off, sc).numpy(), g["mag_inv"]) < 5e-6


def test_normalize(golden):
    g = golden("g9_normalize")
    x = T(g["x"])
    for mode in ["unipolar", "bipolar", "gaussian"]:
        off, sc = O.normalize_stats(x, mode)
        assert np.allclose(off.numpy(), g["offset_" + mode], rtol=3e-6, atol=3e-7)
        assert np.allclose(sc.numpy(), g["scale_" + mode], rtol=1e-6)
        y = (x - off) / sc
        assert rel_max(y.numpy(), g["y_" + mode]) < 2e-6
        assert rel_max((y * sc + off).numpy(), g["inv_" + mode]) < 2e-6


def test_pghi_invert_end_to_end(golden):
    g = golden("g4_pghi_invert")
    mag = g["mag"]
    gw = O.gauss_window(128)
    dw = O.dual_window(gw, 128, 32)
    ph = O.pghi_offline_batch(mag, 128, 32)
    y = O.polar_istft(T(mag), T(ph), dw, 128, 32).numpy()
    ref = g["y"]
    # (iv) resynthesised audio compared by SNR vs the reference's own PGHI audio
    snr = 10 * np.log10((ref ** 2).sum() / max(((y - ref) ** 2).sum(), 1e-30))
    assert snr > 60.0, snr


def test_unpinned_restatements_are_sane():
    """torchaudio-backed pieces are parity-UNPINNED; check documented properties only."""
    fb = O.melscale_fbanks(513, 0.0, 22050.0, 128, 44100)
    assert fb.shape == (513, 128) and float(fb.min()) >= 0 and float(fb.max()) <= 1.0
    assert int(((fb > 0).sum(0) == 0).sum()) <= 1     # at most the first (narrowest) filter is empty
    fb513 = O.magnitude_default_bank(44100, 1024)
    assert fb513.shape == (513, 513)
    assert int((fb513.sum(0) == 0).sum()) == 109 and int((fb513 != 0).sum()) == 1019
    x = torch.linspace(-1, 1, 4001)
    c = O.mulaw_encode(x)
    assert c.dtype == torch.int64 and int(c.min()) == 0 and int(c.max()) == 255
    assert float((O.mulaw_decode(c) - x).abs().max()) < 0.04
    import scipy.fft
    m = torch.rand(5, 128)
    assert np.allclose(O.mfcc_dct(m, 40).numpy(), scipy.fft.dct(m.numpy(), type=2, norm="ortho")[:, :40], atol=1e-5)


def test_phase_representations(golden):
    """oracle unwrap / fdiff / fint / IF / Phase restatements against the reference's own outputs (G11).
    Same torch CPU ops in the same order: equality up to libm differences of atan2 between hosts."""
    g = golden("g11_phase_repr")
    t = lambda k: torch.from_numpy(g[k])  # noqa: E731
    for T in (10, 11):
        r = t("r%d" % T)
        assert torch.equal(O.unwrap(r * 3.0), t("unwrap%d" % T))
        for m in ("forward", "backward", "central"):
            assert torch.equal(O.fdiff(r, m), t("fdiff_%s%d" % (m, T)))
            assert torch.equal(O.fint(r, m), t("fint_%s%d" % (m, T)))
    assert int(g["ifw_second_call_raises"]) == 1      # the reference's weighted IF only survives one call
    for tag in ("X", "Xr"):
        X = t(tag)
        close = lambda a, b: torch.allclose(a, b, rtol=2e-6, atol=2e-5)  # noqa: E731
        if tag == "X":
            assert close(O.unwrap(X.angle()), t("unwrap_angle_X"))
        for m in ("forward", "backward", "central"):
            assert close(O.inst_freq(X, m, True), t("ifw_%s_%s" % (tag, m)))
            for mode in (("none", "gaussian", "bipolar") if tag == "X" else ("gaussian",)):
                key = "if_%s_%s_%s_1" % (tag, mode, m)
                raw = O.inst_freq(X, m)
                off = sc = None
                if mode != "none":
                    off, sc = t(key + "_offset"), t(key + "_scale")
                    o2, s2 = O.normalize_stats(raw, mode)
                    assert torch.allclose(o2, off, rtol=1e-5, atol=1e-6) and torch.allclose(s2, sc, rtol=1e-5)
                y = O.affine(raw, off, sc)
                assert close(y, t(key))
                # inversion from the golden forward output: plain IEEE arithmetic, exact
                inv = O.inst_freq_invert(O.affine(t(key), off, sc, inverse=True), m)
                assert torch.equal(inv, t(key + "_inv"))
            if tag == "X":
                y0 = O.drop_first_bin(O.inst_freq(X, m), False)
                assert close(y0, t("if_X_none_%s_0" % m))
                inv0 = O.pad_last_bin(O.inst_freq_invert(t("if_X_none_%s_0" % m), m), False)
                assert torch.equal(inv0, t("if_X_none_%s_0_inv" % m))
        for unwrap in (0, 1):
            for mode in (("none", "gaussian", "bipolar") if tag == "X" else ("gaussian",)):
                key = "phase_%s_%s_%d_1" % (tag, mode, unwrap)
                off = sc = None
                if mode != "none":
                    off, sc = t(key + "_offset"), t(key + "_scale")
                assert close(O.affine(O.phase_raw(X, bool(unwrap)), off, sc), t(key))
                assert torch.equal(O.affine(t(key), off, sc, inverse=True), t(key + "_inv"))
    X = t("X")
    assert torch.equal(O.pad_last_bin(t("phase_X_none_1_0"), False), t("phase_X_none_1_0_inv"))
    assert torch.equal(X.real, t("real_X_none_1")) and torch.equal(X.imag[..., 1:], t("imag_X_none_0"))
    # stacked: Polar = (Magnitude, Phase) on dim -2; inversion = mag * exp(i phase)
    P = t("polar_m2_1")
    assert P.shape == X.shape[:-1] + (2, X.shape[-1])
    assert torch.allclose(t("polar_none_1_a"), P.select(-2, 0)) and torch.allclose(t("polar_none_1_b"), P.select(-2, 1))


def test_sinebank(golden):
    """oracle sinebank (offline + per-chunk) against the reference's outputs with the recorded random phases."""
    g = golden("g13_sinebank")
    t = lambda k: torch.from_numpy(g[k])  # noqa: E731
    y = O.sinebank_offline(t("mag"), 44100, 128, 32, t("offline_phase"))
    assert y.shape == (2, 32 * 9 + 128) and torch.allclose(y, t("offline"), rtol=1e-5, atol=2e-6)
    y = O.sinebank_offline(t("mag"), 44100, 128, 32, t("offline_via_invert_phase"))
    assert torch.allclose(y, t("offline_via_invert"), rtol=1e-5, atol=2e-6)
    y = O.sinebank_offline(t("mag1k"), 44100, 1024, 256, t("offline1k_phase"))
    assert torch.allclose(y, t("offline1k"), rtol=1e-5, atol=2e-6)
    for name in ("rtstft", "rtdgt"):
        now = torch.tensor(0.)
        for i in range(3):
            y, now = O.sinebank_realtime(t("chunks")[i], 44100, 128, 32, t(name + "_phase"), now)
            assert torch.allclose(y, t(name)[i], rtol=1e-5, atol=2e-6), (name, i)
        assert torch.equal(now, t(name + "_time"))
    now = torch.tensor(0.)
    for i in range(2):
        y, now = O.sinebank_realtime(t("chunks")[i, 0], 44100, 128, 32, t("rt_unbatched_phase"), now)
        assert torch.allclose(y, t("rt_unbatched")[i], rtol=1e-5, atol=2e-6)
    # invert(mode="sinebank") of the realtime classes = the frames times the synthesis window (stft.py:303-304)
    for name in ("rtstft_invert", "rtdgt_invert"):
        now = torch.tensor(0.)
        for i in range(2):
            y, now = O.sinebank_realtime(t("chunks")[i], 44100, 128, 32, t(name + "_phase"), now)
            assert torch.allclose(y * t(name + "_inv_window"), t(name)[i], rtol=1e-5, atol=2e-6), (name, i)


@pytest.mark.parametrize("n", [512, 2048, 400])
def test_streaming_classes_other_sizes(golden, n):
    """G17: the reference's streaming chain at n_fft 512 / 2048 / 400 against the oracle's restatement."""
    g = golden("g17_streaming_sizes")
    n_, h, chunk = [int(v) for v in g["params_%d" % n]]
    x = T(g["x_%d" % n])
    fa, fi, fd = O.OverlapAddState(n, h), O.OverlapAddState(n, h), O.OverlapAddState(n, h)
    w, gw = O.hann_window(n), O.gauss_window(n)
    dw = O.dual_window(gw, n, h)
    for c in range(2):
        fr = fa.forward(x[:, c * chunk:(c + 1) * chunk])
        assert np.array_equal(fr.numpy(), g["frames_%d_%d" % (n, c)])
        X = O.rt_forward(fr, w)
        assert rel_max(X.numpy(), g["X_%d_%d" % (n, c)]) < 3e-6
        yf = O.rt_invert(X, w)
        assert rel_max(yf.numpy(), g["yframes_%d_%d" % (n, c)]) < 3e-6
        assert rel_max(fi.invert(yf).numpy(), g["y_%d_%d" % (n, c)]) < 3e-6
        Xd = O.rt_forward(fr, gw)
        assert rel_max(Xd.numpy(), g["Xd_%d_%d" % (n, c)]) < 3e-6
        ydf = O.rt_invert(Xd, dw)
        assert np.allclose(ydf.numpy(), g["ydframes_%d_%d" % (n, c)], rtol=3e-6, atol=3e-7)
        assert rel_max(fd.invert(ydf).numpy(), g["yd_%d_%d" % (n, c)]) < 3e-6


@pytest.mark.parametrize("tag", ["k512", "k400"])
def test_rtpghi_kernel_other_sizes(golden, tag):
    g = golden("g17_streaming_sizes")
    n, h = [int(v) for v in g[tag + "_params"]]
    r = O.pghi_realtime(g[tag + "_magbuf"], g[tag + "_mag"], g[tag + "_phasebuf"], g[tag + "_noise"], n, h)
    ref = g[tag + "_phase"]
    tol_arr = 2e-3 + 16 * np.spacing(np.abs(ref).astype(np.float32)) + 1e-5 * np.abs(ref)
    assert np.all(np.abs(r["phase"] - ref) <= tol_arr)


def test_readme_chain(golden):
    """G18: Mono() + DGT(pghi) + Magnitude(mel, unipolar, log1p) (reference README.md:48-61) stage by stage and end to
    end: the oracle's chain against the reference's own outputs."""
    g = golden("g18_readme_chain")
    x = T_(g["x"])
    mono = O.mono_mix(x)
    assert torch.equal(mono, T_(g["mono"]))
    win = O.gauss_window(1024)
    X = O.stft_forward(mono, win, 1024, 256)
    assert rel_max(X.numpy(), g["spec"]) < 1e-6
    fwd, inv = O.magnitude_banks(T_(g["bank"]))
    off, sc = O.magnitude_scale_stats(X, "log1p", "unipolar")
    assert abs(float(off) - float(g["offset"])) < 1e-6 and abs(float(sc) - float(g["scale"])) < 1e-6 * float(g["scale"])
    y = O.magnitude_forward(X, fwd, "log1p", off, sc)
    assert rel_max(y.numpy(), g["y"]) < 1e-6
    mag = O.magnitude_invert(T_(g["y"]), inv, "log1p", off, sc)
    assert rel_max(mag.numpy(), g["mag_inv"]) < 1e-5
    # PGHI + polar ISTFT on the reference's own inverse-mel magnitudes: the reference's audio
    mag_ref = T_(g["mag_inv"])
    phase = np.stack([O.pghi_offline(mag_ref[b], 1024, 256)["phase"] for b in range(mag_ref.shape[0])])
    audio = O.polar_istft(mag_ref, T_(phase), O.dual_window(win, 1024, 256), 1024, 256)
    ref = g["x_inv"][:, 0]
    err = audio.numpy().astype(np.float64) - ref
    assert 10 * np.log10((ref.astype(np.float64) ** 2).sum() / max((err ** 2).sum(), 1e-300)) > 60.0


@pytest.mark.parametrize("n,h", [(1024, 256), (1024, 128), (64, 16), (512, 256)])
def test_overlap_add_helpers(golden, n, h):
    """G19: the stateless framing / overlap-add pair and the buffer helpers (oadd.py:33-67)."""
    g = golden("g19_oadd_helpers")
    key = "%d_%d" % (n, h)
    st = O.OverlapAddState(n, h)
    assert torch.equal(O.frame(T_(g["x_" + key]), n, h), T_(g["frames_" + key]))
    assert rel_max(st.invert_without_update(T_(g["in_" + key])).numpy(), g["inv_" + key]) < 1e-7
    c0, c1 = T_(g["c0_" + key]), T_(g["c1_" + key])
    st.forward(c0)
    assert torch.equal(st.inbuf, T_(g["inbuf1_" + key]))         # what the second get_input_buffer call hands back
    assert float(np.abs(g["inbuf0_" + key]).max()) == 0.0 and float(np.abs(g["outbuf0_" + key]).max()) == 0.0
