"""GPU parity: Magnitude (mel projection + contrast + Normalize), Normalize, MFCC.
fp32 tolerance 1e-5 normwise (north_star); the projection runs on exact-fp32 MFMA."""
import numpy as np
import pytest
import os
import torch

import acids_transforms_amd as A
from conftest import rel_max
from acids_transforms_amd._lib import variant
from oracle import oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-5
T_ = torch.from_numpy


def cpu(t):
    return t.detach().cpu().numpy()


def test_magnitude_all_modes_golden(golden, dev):
    g = golden("g7_magnitude")
    X = T_(g["X"]).to(dev)
    for c in ["log1p", "log", "log10", "none"]:
        for mode in ["unipolar", "bipolar", "gaussian", "none"]:
            for mel in [1, 0]:
                k = "%s_%s_%d" % (c, mode, mel)
                m = A.Magnitude(mode=mode, contrast=c, mel=bool(mel), n_fft=128)
                m._set_bank(T_(g["bank"]))
                m = m.to(dev)
                assert np.array_equal(cpu(m.mel_bank), g["mel_bank"])
                assert np.array_equal(cpu(m.inverse_mel_bank), g["inverse_mel_bank"])
                m.scale_data(X)
                if mode != "none":
                    assert abs(float(m.norm.offset) - float(g["offset_" + k])) <= 2e-6 * max(1, abs(float(g["offset_" + k]))), k
                    assert abs(float(m.norm.scale) - float(g["scale_" + k])) <= 2e-6 * abs(float(g["scale_" + k])), k
                    assert m.norm.offset.dim() == 0
                y = m(X)
                assert y.shape == g["y_" + k].shape
                assert rel_max(cpu(y), g["y_" + k]) < TOL, k
                xi = m.invert(T_(g["y_" + k]).to(dev))
                assert rel_max(cpu(xi), g["inv_" + k]) < 2e-5, k     # exp() amplifies 1 ulp of its argument
    m = A.Magnitude(mode="none", contrast="log1p", mel=True, n_fft=128)
    m._set_bank(T_(g["bank"]))
    m = m.to(dev)
    y2, y1 = m(X[0]), m(X[0, 0])
    assert y2.shape == g["y_2d"].shape and rel_max(cpu(y2), g["y_2d"]) < TOL
    assert y1.shape == g["y_1d"].shape and rel_max(cpu(y1), g["y_1d"]) < TOL


def test_magnitude_513_golden(golden, dev):
    g = golden("g7_magnitude")
    m = A.Magnitude(mode="unipolar", contrast="log1p", mel=True)
    m._set_bank(T_(g["bank513"]))
    m = m.to(dev)
    X = T_(g["X513"]).to(dev)
    m.scale_data(X)
    assert abs(float(m.norm.offset) - float(g["offset513"])) < 1e-6
    y = m(X)
    assert rel_max(cpu(y), g["y513"]) < TOL
    assert rel_max(cpu(m.invert(T_(g["y513"]).to(dev))), g["inv513"]) < 2e-5


def test_compose_stft_magnitude_golden(golden, dev):
    g = golden("g7_compose")
    mag = A.Magnitude(mode="unipolar", contrast="log1p", mel=True)
    mag._set_bank(T_(g["bank"]))
    comp = (A.STFT() + mag).to(dev)
    assert isinstance(comp, A.ComposeAudioTransform) and comp.needs_scaling
    x = T_(g["x"]).to(dev)
    comp.scale_data(x)
    assert abs(float(comp[1].norm.scale) - float(g["scale"])) < 1e-5 * float(g["scale"])
    y = comp(x)
    assert rel_max(cpu(y), g["y"]) < TOL
    assert rel_max(cpu(comp[1].invert(T_(g["y"]).to(dev))), g["mag_inv"]) < 2e-5
    with pytest.raises(TypeError):
        comp + 3


def test_normalize_golden(golden, dev):
    g = golden("g9_normalize")
    x = T_(g["x"]).to(dev)
    for mode in ["unipolar", "bipolar", "gaussian"]:
        nm = A.Normalize(mode).to(dev)
        with pytest.raises(RuntimeError):
            nm(x)                       # unscaled module raises, like the reference
        nm.scale_data(x)
        assert abs(float(nm.offset) - float(g["offset_" + mode])) < 1e-6
        assert abs(float(nm.scale) - float(g["scale_" + mode])) < 1e-6
        y = nm(x)
        assert rel_max(cpu(y), g["y_" + mode]) < TOL
        assert rel_max(cpu(nm.invert(T_(g["y_" + mode]).to(dev))), g["inv_" + mode]) < TOL
    # the reference's own known-answer asserts (norm.py:59-67)
    nm = A.Normalize("unipolar").to(dev)
    nm.scale_data(x)
    y = nm(x)
    assert float(y.min()) == 0.0 and float(y.max()) == 1.0
    nm = A.Normalize("bipolar").to(dev)
    nm.scale_data(x)
    y = nm(x)
    assert float(y.min()) == -1.0 and float(y.max()) == 1.0


@pytest.mark.parametrize("rows,K,N", [(1000, 513, 128), (77, 513, 513), (33, 257, 80), (5, 129, 129), (64, 65, 40),
                                       (3, 33, 16), (200, 17, 17), (9, 1025, 64), (1, 513, 1), (4097, 513, 130),
                                       (40, 128, 513), (31, 100, 37)])
def test_projection_vs_oracle_shapes(dev, rows, K, N):
    g = torch.Generator().manual_seed(rows * 7 + K)
    X = (torch.randn(rows, K, generator=g) * torch.exp(2j * np.pi * torch.rand(rows, K, generator=g))).to(torch.complex64)
    bank = torch.rand(K, N, generator=g) * (torch.rand(K, N, generator=g) < 0.3)
    from acids_transforms_amd import ops
    y = ops.mel_forward(X.to(dev), bank.to(dev), "log1p")
    yr = torch.log(1 + torch.matmul(X.abs().double(), bank.double())).float()
    assert rel_max(cpu(y), yr.numpy()) < TOL
    yi = ops.mel_inverse(yr.to(dev), bank.t().contiguous().to(dev), "log1p")
    yir = torch.matmul((torch.exp(yr) - 1).double(), bank.t().double()).float()
    assert rel_max(cpu(yi), yir.numpy()) < TOL
    yp = ops.mel_forward(X.to(dev), bank.to(dev), None, power=2)
    ypr = torch.matmul((X.abs().double() ** 2), bank.double()).float()
    assert rel_max(cpu(yp), ypr.numpy()) < TOL


def test_default_bank_and_nmels(dev):
    m = A.Magnitude().to(dev)            # reference default: 513 x 513 HTK bank, 1019 non-zeros
    fb = O.magnitude_default_bank(44100, 1024)
    fwd, inv = O.magnitude_banks(fb)
    assert np.array_equal(cpu(m.mel_bank), fwd.numpy()) and np.array_equal(cpu(m.inverse_mel_bank), inv.numpy())
    torch.manual_seed(1)
    x = torch.randn(4, 8192) * 0.1
    s = A.STFT().to(dev)
    X = s(x.to(dev))
    m.scale_data(X)
    Xr = O.stft_forward(x, O.hann_window(1024), 1024, 256)
    off, sc = O.magnitude_scale_stats(Xr, "log1p", "unipolar")
    yr = O.magnitude_forward(Xr, fwd, "log1p", off, sc)
    assert rel_max(cpu(m(X)), yr.numpy()) < TOL
    m128 = A.Magnitude(n_mels=128, mode=None).to(dev)   # BASELINE config 2's "mel=128"
    y = m128(X)
    assert y.shape == (4, 33, 128)
    fb128 = O.melscale_fbanks(513, 0.0, 22050.0, 128, 44100)
    fwd128, _ = O.magnitude_banks(fb128)
    assert rel_max(cpu(y), O.magnitude_forward(Xr, fwd128, "log1p").numpy()) < TOL
    # keep_nyquist=False: drops output bin 0, pads the last one on the way back
    mk = A.Magnitude(mode=None, keep_nyquist=False).to(dev)
    yk = mk(X)
    assert yk.shape[-1] == 512 and mk.invert(yk).shape[-1] == 513
    fwd_k, inv_k = O.magnitude_banks(O.magnitude_default_bank(44100, 1024, keep_nyquist=False))
    assert rel_max(cpu(yk), O.magnitude_forward(Xr, fwd_k, "log1p", keep_nyquist=False).numpy()) < TOL
    assert rel_max(cpu(mk.invert(yk)), O.magnitude_invert(torch.from_numpy(cpu(yk)), inv_k, "log1p", keep_nyquist=False).numpy()) < TOL


def test_mfcc_melspectrogram(dev):
    torch.manual_seed(2)
    x = torch.randn(3, 2, 6000) * 0.1
    f = A.MFCC().to(dev)
    y = f(x.to(dev))
    yr = O.melspectrogram(x, 44100, 1024, 256, 128, 2.0)
    assert y.shape == yr.shape == (3, 2, 128, 24)
    assert rel_max(cpu(y), yr.numpy()) < TOL
    with pytest.raises(A.NotInvertibleError):
        f.invert(y)
    _, tt = f.forward_with_time(x.to(dev), torch.zeros(3, 2, device=dev))
    assert tt.shape == (3, 2, 128)       # the reference counts n_mels as chunks (mel.py:49)
    fn = A.MFCC(norm_mode="gaussian").to(dev)
    fn.scale_data(x.to(dev))             # statistics of the raw input, as the reference does
    off, sc = O.normalize_stats(x, "gaussian")
    assert rel_max(cpu(fn(x.to(dev))), ((yr - off) / sc).numpy()) < TOL
    # build extension: true MFCC(40) = DCT-II(ortho) of 10 log10(mel power)
    f40 = A.MFCC(n_mfcc=40).to(dev)
    c = f40(x.to(dev))
    db = 10.0 * torch.log10(torch.clamp(yr, min=1e-10)).transpose(-1, -2)
    cr = O.mfcc_dct(db, 40).transpose(-1, -2)
    assert c.shape == cr.shape == (3, 2, 40, 24)
    assert rel_max(cpu(c), cr.numpy()) < 2e-5


def test_magnitude_n_fft_2048_golden(golden, dev):
    """G16: the reference's Magnitude(n_fft=2048) -- 1025 filters, its own bank (built through the shim: one bin per
    filter), log1p + unipolar statistics -- forward and invert through the long-row banded walk (17 passes)."""
    g = golden("g16_other_sizes")
    X = torch.from_numpy(g["mag2048_X"]).to(dev)
    bank = torch.zeros(1025, 1025)
    idx = torch.from_numpy(g["mag2048_bank_idx"]).long()
    bank[idx[:, 0], idx[:, 1]] = torch.from_numpy(g["mag2048_bank_val"])
    mg = A.Magnitude(n_fft=2048)
    mg._set_bank(bank)
    mg = mg.to(dev)
    mg.scale_data(X)
    assert abs(float(mg.norm.offset) - float(g["mag2048_offset"])) < 1e-6
    assert abs(float(mg.norm.scale) - float(g["mag2048_scale"])) < 1e-5
    assert mg._band_of("mel_bank") is not None and mg._band_of("mel_bank").n_passes == 17
    y = mg(X)
    assert rel_max(cpu(y), g["mag2048_y"]) < TOL
    assert rel_max(cpu(mg.invert(torch.from_numpy(g["mag2048_y"]).to(dev))), g["mag2048_inv"]) < 2e-5


@pytest.mark.parametrize("n,h,n_mels,L", [(2048, 512, 128, 30000), (512, 128, 64, 9001), (4096, 1024, 128, 50000),
                                          (256, 64, 40, 5000), (2048, 512, 80, 2048 * 3 + 17), (400, 160, 40, 16000),
                                          (1024, 100, 128, 7000), (2048, 300, 200, 20001), (2048, 2048, 40, 9000),
                                          (512, 160, 80, 16001), (512, 100, 200, 7001), (512, 256, 40, 3000)])
def test_melspectrogram_at_other_sizes(dev, n, h, n_mels, L):
    """MFCC (= MelSpectrogram, mel.py:31-73) away from the fused n_fft 1024 / hop 256 kernel: STFT + the banded walk
    with its channel-major register window (one- and two-pass banks, runs of rows that start and end inside a clip,
    frame counts that are not a multiple of eight), and the DCT behind n_mfcc."""
    torch.manual_seed(n + h)
    x = torch.randn(5, 3, L) * 0.1
    f = A.MFCC(n_fft=n, hop_length=h, n_mels=n_mels).to(dev)
    # n_fft 2048: one kernel, the spectrum is never written (40 filters over 1025 bins have bands of 168 bins, beyond
    # the fused walk's 128: that bank takes the two-kernel path)
    assert f._band.fusable2048 == (n == 2048 and n_mels != 40)
    assert f._band.fusable512 == (n == 512)            # likewise (two frames per FFT, one walk for both)
    y = f(x.to(dev))
    yr = O.melspectrogram(x, 44100, n, h, n_mels, 2.0)
    assert y.shape == yr.shape and rel_max(cpu(y), yr.numpy()) < TOL
    fn = A.MFCC(n_fft=n, hop_length=h, n_mels=n_mels, norm_mode="gaussian").to(dev)
    fn.scale_data(x.to(dev))
    off, sc = O.normalize_stats(x, "gaussian")
    assert rel_max(cpu(fn(x.to(dev))), ((yr - off) / sc).numpy()) < TOL
    n_mfcc = min(40, n_mels)
    c = A.MFCC(n_fft=n, hop_length=h, n_mels=n_mels, n_mfcc=n_mfcc).to(dev)(x.to(dev))
    db = 10.0 * torch.log10(torch.clamp(yr, min=1e-10)).transpose(-1, -2)
    cr = O.mfcc_dct(db, n_mfcc).transpose(-1, -2)
    assert c.shape == cr.shape and rel_max(cpu(c), cr.numpy()) < 2e-5


def test_zero_block_skipping_keeps_dense_semantics(dev):
    """Banded banks skip all-zero bank blocks; results must equal the dense contraction, including
    inf/NaN propagation (0 * NaN = NaN in a dense matmul), and a dense bank must be unaffected."""
    from acids_transforms_amd import ops
    g = torch.Generator().manual_seed(11)
    rows, K, N = 200, 513, 128
    X = (torch.randn(rows, K, generator=g) * torch.exp(2j * np.pi * torch.rand(rows, K, generator=g))).to(torch.complex64)
    band = O.melscale_fbanks(513, 0.0, 22050.0, 128, 44100)
    dense = torch.rand(K, N, generator=g) + 0.1
    for bank in (band, dense):
        y = ops.mel_forward(X.to(dev), bank.to(dev), None)
        yr = torch.matmul(X.abs().double(), bank.double()).float()
        assert rel_max(cpu(y), yr.numpy()) < TOL
    Xn = X.clone()
    Xn[37, 400] = complex(float("nan"), 0.0)        # one poisoned bin in tile 1 (rows 32..63)
    Xn[150, 3] = complex(float("inf"), 1.0)         # and an inf in tile 4
    y = cpu(ops.mel_forward(Xn.to(dev), band.to(dev), None))
    yr = torch.matmul(Xn.abs(), band).numpy()
    assert np.array_equal(np.isnan(y), np.isnan(yr))
    assert np.array_equal(np.isinf(y), np.isinf(yr))
    ok = np.isfinite(yr)
    assert np.abs(y[ok] - yr[ok]).max() / np.abs(yr[ok]).max() < TOL


def test_fused_stft_mel_equals_separate_stages(dev):
    """Compose(STFT|DGT + Magnitude(banded bank)) runs as one kernel: same features as the two-stage path and
    as the oracle, the spectrum side effects (phase buffer, keep_input) intact."""
    g = torch.Generator().manual_seed(31)
    x = torch.randn(5, 2, 9000, generator=g) * 0.1
    for cls, w in [(A.STFT, O.hann_window(1024)), (A.DGT, O.gauss_window(1024))]:
        for contrast, mode in [("log1p", "unipolar"), ("log", "gaussian"), (None, None)]:
            st = cls().to(dev)
            mg = A.Magnitude(n_mels=128, mode=mode, contrast=contrast).to(dev)
            comp = st + mg
            xd = x.to(dev)
            comp.scale_data(xd)
            assert mg.can_fuse_with(st, xd)
            y = comp(xd)                                  # fused
            y2 = mg(st(xd))                               # stage by stage (MFMA projection)
            assert y.shape == y2.shape == (5, 2, 36, 128)
            assert rel_max(cpu(y), cpu(y2)) < TOL
            Xr = O.stft_forward(x, w, 1024, 256)
            fwd, _ = O.magnitude_banks(O.melscale_fbanks(513, 0.0, 22050.0, 128, 44100))
            off = sc = None
            if mode is not None:
                off, sc = O.magnitude_scale_stats(Xr, contrast, mode)
            yr = O.magnitude_forward(Xr, fwd, contrast, off, sc)
            assert rel_max(cpu(y), yr.numpy()) < TOL
            # spectrum side effects of the STFT stage survive the fusion
            Xn = Xr.reshape(10, 36, 513).numpy()
            big = np.abs(Xn) > 1e-2
            dphi = np.angle(np.exp(1j * (cpu(st.phase_buffer.reshape(10, 36, 513)) - np.angle(Xn))))
            assert np.abs(dphi[big]).max() < 1e-3
    # not fusable: odd clip length, realtime stage, a dense (non-banded) bank -> plain two-stage path
    st, mg = A.STFT().to(dev), A.Magnitude(n_mels=128, mode=None).to(dev)
    assert not mg.can_fuse_with(st, torch.zeros(2, 9001, device=dev))
    assert not mg.can_fuse_with(A.RealtimeSTFT().to(dev), torch.zeros(2, 9000, device=dev))
    dense = A.Magnitude(n_mels=32, mode=None)
    dense._set_bank(torch.rand(513, 32))
    assert not dense.to(dev).can_fuse_with(st, torch.zeros(2, 9000, device=dev))
    yo = (st + mg)(x[..., :8999].to(dev))
    assert yo.shape == (5, 2, 36, 128)
    # editing the bank buffer invalidates the cached band table
    mg.mel_bank[0, 100, 5] = 0.5
    y_edit = (st + mg)(x.to(dev))
    assert rel_max(cpu(y_edit), cpu(mg(st(x.to(dev))))) < TOL


def test_fused_reference_default_chain(dev):
    """`STFT() + Magnitude()` with every default -- the reference's own "stft+magnitude" chain: a 513-filter
    bank (109 of them empty), nine passes of the fused epilogue.  Same features as stage by stage and the oracle."""
    g = torch.Generator().manual_seed(33)
    x = torch.randn(3, 12000, generator=g) * 0.1
    st, mg = A.STFT().to(dev), A.Magnitude().to(dev)
    comp = st + mg
    xd = x.to(dev)
    comp.scale_data(xd)
    assert mg.can_fuse_with(st, xd) and mg._banded().n_passes == 9
    y = comp(xd)
    y2 = mg(st(xd))
    assert y.shape == y2.shape == (3, 47, 513)
    assert rel_max(cpu(y), cpu(y2)) < TOL
    Xr = O.stft_forward(x, O.hann_window(1024), 1024, 256)
    fwd, _ = O.magnitude_banks(O.melscale_fbanks(513, 0.0, 22050.0, 513, 44100))
    off, sc = O.magnitude_scale_stats(Xr, "log1p", "unipolar")
    assert rel_max(cpu(y), O.magnitude_forward(Xr, fwd, "log1p", off, sc).numpy()) < TOL
    # empty filters: contrast(0), normalised (the fused epilogue multiplies by 1/scale where the stand-alone kernel divides)
    assert abs(float(y[..., 0].abs().max()) - float(y2[..., 0].abs().max())) <= 1e-6 * float(y2[..., 0].abs().max())


def test_fused_random_banded_bank(dev):
    g = torch.Generator().manual_seed(32)
    K, N = 513, 96
    bank = torch.zeros(K, N)
    for n in range(N):
        s0 = int(torch.randint(0, K - 70, (1,), generator=g))
        ln = int(torch.randint(1, 60, (1,), generator=g))
        bank[s0:s0 + ln, n] = torch.rand(ln, generator=g)
    bank[:, 7] = 0.0                                        # an empty filter
    x = torch.randn(3, 5000, generator=g) * 0.1
    st = A.STFT().to(dev)
    mg = A.Magnitude(n_mels=N, mode=None, contrast="log1p")
    mg._set_bank(bank)
    mg = mg.to(dev)
    y = (st + mg)(x.to(dev))
    Xr = O.stft_forward(x, O.hann_window(1024), 1024, 256)
    fwd, _ = O.magnitude_banks(bank)
    assert rel_max(cpu(y), O.magnitude_forward(Xr, fwd, "log1p").numpy()) < TOL


def test_banded_projection_equals_dense_contraction(dev):
    """at_mel_project_banded (what Magnitude.forward / invert run for a banded bank) against the dense MFMA
    contraction of the same bank and the oracle: forward |x| and |x|^2, every contrast, with and without
    normalisation, channel-major output; inverse chain; banks of 513 (reference default), 128 and 65 filters."""
    from acids_transforms_amd import ops
    from acids_transforms_amd.utils.banded import BandedBank
    gen = torch.Generator().manual_seed(41)
    for n_fft, n_mels, rows in [(1024, 513, (3, 50)), (1024, 128, (2, 2, 33)), (128, 65, (5, 17)), (1024, 40, (1, 1))]:
        F = n_fft // 2 + 1
        m = A.Magnitude(n_fft=n_fft, n_mels=n_mels).to(dev)
        X = (torch.randn(*rows, F, generator=gen) * torch.exp(2j * np.pi * torch.rand(*rows, F, generator=gen))).to(torch.complex64)
        Xd = X.to(dev)
        fb = BandedBank(m.mel_bank)
        ib = BandedBank(m.inverse_mel_bank)
        assert fb.eligible and ib.eligible
        off = torch.tensor(0.3, device=dev)
        sc = torch.tensor(1.7, device=dev)
        for contrast in ("log1p", "log", "log10", None):
            for (o, s_) in ((None, None), (off, sc)):
                for power in (1, 2):
                    dense = ops.mel_forward(Xd, m.mel_bank, contrast, o, s_, m._eps, power=power)
                    band = ops.mel_forward(Xd, m.mel_bank, contrast, o, s_, m._eps, power=power, band=fb)
                    assert band.shape == dense.shape
                    assert rel_max(cpu(band), cpu(dense)) < TOL, (n_mels, contrast, power)
                y = ops.mel_forward(Xd, m.mel_bank, contrast, o, s_, m._eps, band=fb)
                inv_d = ops.mel_inverse(y, m.inverse_mel_bank, contrast, o, s_, m._eps)
                inv_b = ops.mel_inverse(y, m.inverse_mel_bank, contrast, o, s_, m._eps, band=ib)
                assert rel_max(cpu(inv_b), cpu(inv_d)) < TOL, (n_mels, contrast)
        T = rows[-1]
        cm_d = ops.mel_forward(Xd, m.mel_bank, None, None, None, power=2, channel_major_T=T)
        cm_b = ops.mel_forward(Xd, m.mel_bank, None, None, None, power=2, channel_major_T=T, band=fb)
        assert cm_b.shape == rows[:-1] + (n_mels, T) and rel_max(cpu(cm_b), cpu(cm_d)) < TOL
        fwd, inv = O.magnitude_banks(O.melscale_fbanks(F, 0.0, 22050.0, n_mels, 44100))
        want = O.magnitude_forward(X, fwd, "log1p")
        assert rel_max(cpu(m.__class__(n_fft=n_fft, n_mels=n_mels, mode=None).to(dev)(Xd)), want.numpy()) < TOL


def test_small_projection_shapes(dev):
    """at_project_small (the MFCC DCT): every K <= 128 / N <= 64 shape class, row-major and channel-major output
    with frame counts that are not multiples of the 8-frame store window, with and without normalisation."""
    from acids_transforms_amd import ops
    gen = torch.Generator().manual_seed(51)
    off = torch.tensor(0.25, device=dev)
    sc = torch.tensor(3.0, device=dev)
    # K = 128 / 80 / 64 take the matrix-core form (tile pairs of 32 rows: row counts around 32 and 64, clips shorter
    # than a tile, every count of 16-channel tiles), the other K the one-wave-per-row form
    for (B, T, K, N) in [(3, 17, 128, 40), (1, 1, 128, 40), (5, 8, 64, 13), (2, 23, 20, 64), (7, 3, 128, 1), (2, 1000, 33, 7),
                         (3, 17, 80, 20), (2, 690, 128, 40), (9, 7, 128, 64), (1, 100, 64, 17), (4, 33, 80, 40),
                         (1, 31, 128, 33), (1, 32, 128, 16), (1, 65, 128, 48), (64, 1, 128, 40)]:
        x = torch.randn(B, T, K, generator=gen)
        W = torch.randn(K, N, generator=gen)
        want = (x.double() @ W.double()).float()
        for (o, s_) in ((None, None), (off, sc)):
            ref = want if o is None else (want - 0.25) / 3.0
            y = ops.mel_forward_real(x.to(dev), W.to(dev), o, s_)
            assert y.shape == (B, T, N) and rel_max(cpu(y), ref.numpy()) < TOL, (B, T, K, N)
            yc = ops.mel_forward_real(x.to(dev), W.to(dev), o, s_, channel_major_T=T)
            assert yc.shape == (B, N, T) and rel_max(cpu(yc), ref.transpose(-2, -1).numpy()) < TOL, (B, T, K, N)
    # both forms of the same shape agree to rounding (different summation orders)
    x = torch.randn(6, 50, 128, generator=gen).to(dev)
    W = torch.randn(128, 40, generator=gen).to(dev)
    y_mfma = ops.mel_forward_real(x, W, off, sc, channel_major_T=50)
    with variant("small_projection", 1):
        y_valu = ops.mel_forward_real(x, W, off, sc, channel_major_T=50)
    assert rel_max(cpu(y_mfma), cpu(y_valu)) < 1e-5
    big = torch.randn(4, 10, 200, generator=gen)           # K > 128: the MFMA contraction takes over
    Wb = torch.randn(200, 30, generator=gen)
    assert rel_max(cpu(ops.mel_forward_real(big.to(dev), Wb.to(dev))), (big @ Wb).numpy()) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("hop", [128, 512])
def test_fused_stft_mel_at_other_hops(dev, hop):
    """Compose(STFT|DGT(hop=128 | 512) + Magnitude) is still one kernel (sliding window of 1 or 4 register slots per
    frame + the banded epilogue): same features as stage by stage and as the oracle; Polar falls back to stages."""
    g = torch.Generator().manual_seed(hop)
    x = torch.randn(5, 30000, generator=g) * 0.1
    xd = x.to(dev)
    for cls in (A.STFT, A.DGT):
        st, mg = cls(hop_length=hop).to(dev), A.Magnitude(n_mels=128).to(dev)
        comp = st + mg
        comp.scale_data(xd)
        assert mg.can_fuse_with(st, xd)
        y = comp(xd)
        X = st(xd)
        assert y.shape == (5, 1 + 30000 // hop, 128)
        assert rel_max(cpu(y), cpu(mg(X))) < TOL
        Xr = O.stft_forward(x, st.window[:1024].cpu(), 1024, hop)
        fwd, _ = O.magnitude_banks(O.melscale_fbanks(513, 0.0, 22050.0, 128, 44100))
        off, sc = O.magnitude_scale_stats(Xr, "log1p", "unipolar")
        assert rel_max(cpu(y), O.magnitude_forward(Xr, fwd, "log1p", off, sc).numpy()) < TOL
        Xf, feat = mg.forward_fused(st, xd, return_spectrum=True)
        assert rel_max(cpu(Xf), Xr.numpy()) < TOL and torch.equal(feat, y)
    pol = A.Polar().to(dev)
    pol.scale_data(A.STFT(hop_length=hop).to(dev)(xd))
    assert not pol.can_fuse_with(A.STFT(hop_length=hop).to(dev), xd)


@pytest.mark.gpu
@pytest.mark.parametrize("n_fft", [256, 512, 1536, 2048, 3000, 4096, 8192])
def test_magnitude_at_other_fft_sizes(dev, n_fft):
    """Magnitude(n_fft != 1024): the reference-default F x F bank and an 80-mel one, forward and invert against the
    oracle -- banded walk while a frame fits the LDS row (F <= 2112: the 17- and 33-segment kernels, exact at 2048 /
    4096 and clamped for the sizes in between), the dense MFMA projection above that."""
    from acids_transforms_amd.utils.banded import BandedBank
    F = n_fft // 2 + 1
    torch.manual_seed(n_fft)
    X = (torch.randn(3, 37, F) + 1j * torch.randn(3, 37, F)).to(torch.complex64)
    for kw in ({}, {"n_mels": 80}):
        mg = A.Magnitude(n_fft=n_fft, **kw).to(dev)
        mg.scale_data(X.to(dev))
        assert (mg._band_of("mel_bank") is not None) == (F <= 2112)
        assert (mg._band_of("inverse_mel_bank") is not None) == (F <= 2112)
        # the fused n_fft = 1024 epilogue keeps its tighter limits AND walks a 513-bin row only: a bank built for another
        # n_fft must never pass for it
        assert BandedBank(mg.mel_bank).fusable == (F == 513)
        fwd, inv = O.magnitude_banks(O.melscale_fbanks(F, 0.0, 22050.0, kw.get("n_mels", F), 44100))
        off, sc = O.magnitude_scale_stats(X, "log1p", "unipolar")
        yr = O.magnitude_forward(X, fwd, "log1p", off, sc)
        y = mg(X.to(dev))
        assert y.shape == yr.shape and rel_max(cpu(y), yr.numpy()) < TOL
        assert rel_max(cpu(mg.invert(y)), O.magnitude_invert(yr, inv, "log1p", off, sc).numpy()) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("n_fft", [1536, 2048, 4096])
def test_polar_one_pass_at_long_rows(dev, n_fft):
    """Polar over the spectra of the larger FFTs: still one banded pass each way (magnitude + angle into the stacked
    tensor; de-normalise, project, attach the phase), same values as the parts."""
    from acids_transforms_amd import ops
    F = n_fft // 2 + 1
    gen = torch.Generator().manual_seed(n_fft)
    X = (torch.randn(2, 29, F, generator=gen) * torch.exp(2j * np.pi * torch.rand(2, 29, F, generator=gen))).to(torch.complex64)
    Xd = X.to(dev)
    pol = A.Polar(magnitude_args={"mode": "bipolar", "n_fft": n_fft}).to(dev)
    pol.scale_data(Xd)
    assert pol._one_pass(Xd) is not None
    y = pol(Xd)
    assert y.shape == (2, 29, 2, F)
    assert rel_max(cpu(y[..., 0, :]), cpu(pol.magnitude(Xd))) < TOL
    po, ps = O.normalize_stats(X.angle(), "bipolar")
    assert rel_max(cpu(y[..., 1, :]), O.affine(X.angle(), po, ps).numpy()) < TOL
    assert pol._one_pass_invert(y) is not None
    Xi = pol.invert(y)
    parts = ops.polar_to_complex(pol.magnitude.invert(y[..., 0, :]), pol.phase.invert(y[..., 1, :]))
    assert rel_max(cpu(Xi), cpu(parts)) < TOL


# ----------------------------------------------------------------------------------------------------------
# bf16 MFMA projection (BASELINE config 5; opt-in `Magnitude(bank_dtype="bf16")`)
# ----------------------------------------------------------------------------------------------------------
BF16_TOL = 4e-3          # vs the fp32 chain: both operands keep 8 significant bits (SURVEY hard part 5)


def _bf16(t):
    return t.to(torch.bfloat16).float()


@pytest.mark.parametrize("rows,K,N", [(1000, 513, 128), (1, 513, 128), (129, 513, 40), (300, 513, 513), (77, 257, 64),
                                      (260, 129, 128), (33, 1025, 96), (5, 33, 7)])
def test_bf16_projection_exact_operands(dev, rows, K, N):
    """Operands that ARE bf16 numbers: the products are exact in fp32, only the summation order differs from a
    CPU matmul -- the MFMA contraction itself is checked at the fp32 bar (1e-5), every tile shape / edge."""
    from acids_transforms_amd import ops
    g = torch.Generator().manual_seed(rows * 7 + K + N)
    a = _bf16(torch.rand(rows, K, generator=g) * 3.0)          # real input goes through |.| like the fp32 projection
    bank = _bf16(torch.rand(K, N, generator=g) * (torch.rand(K, N, generator=g) < 0.3))      # asymmetric, sparse-ish
    img = ops.mel_bf16_pack_bank(bank.to(dev))
    y = ops.mel_forward_bf16(a.to(dev), img, K, N)
    want = a.double() @ bank.double()
    assert y.shape == (rows, N)
    assert rel_max(cpu(y), want.numpy()) < TOL
    # |.| of real input, contrast and normalisation in the epilogue
    off, sc = torch.tensor(0.25, device=dev), torch.tensor(1.75, device=dev)
    y2 = ops.mel_forward_bf16((-a).to(dev), img, K, N, "log1p", off, sc)
    want2 = (torch.log1p(a.abs().double() @ bank.double()) - 0.25) / 1.75
    assert rel_max(cpu(y2), want2.numpy()) < TOL


@pytest.mark.parametrize("contrast", ["log1p", "none", "log"])
def test_bf16_magnitude_vs_oracle(dev, contrast):
    """Magnitude(bank_dtype="bf16") on a complex spectrum: (a) against the oracle chain evaluated on bf16-rounded
    operands -- tight (a magnitude that lands within an fp32 ulp of a bf16 rounding boundary may round the other
    way: 1e-3); (b) against the fp32 oracle at the stated bf16 tolerance 4e-3."""
    g = torch.Generator().manual_seed(17)
    x = torch.randn(6, 9000, generator=g) * 0.1
    Xr = O.stft_forward(x, O.hann_window(1024), 1024, 256)
    fwd, _ = O.magnitude_banks(O.melscale_fbanks(513, 0.0, 22050.0, 128, 44100))
    off, sc = O.magnitude_scale_stats(Xr, contrast, "unipolar")
    m = A.Magnitude(n_mels=128, mode="unipolar", contrast=contrast, bank_dtype="bf16").to(dev)
    Xd = Xr.to(dev)
    m.scale_data(Xd)
    y = m(Xd)
    assert y.shape == (6, Xr.shape[1], 128) and y.dtype == torch.float32
    want_fp32 = O.magnitude_forward(Xr, fwd, contrast, off, sc)
    want_bf16 = O.affine(O.contrast(_bf16(Xr.abs()) @ _bf16(fwd[0] if fwd.dim() == 3 else fwd), contrast), off, sc)
    assert rel_max(cpu(y), want_bf16.numpy()) < 1e-3
    assert rel_max(cpu(y), want_fp32.numpy()) < BF16_TOL
    # and it IS the bf16 path: the fp32 module differs from it by more than fp32 noise
    m32 = A.Magnitude(n_mels=128, mode="unipolar", contrast=contrast).to(dev)
    m32.scale_data(Xd)
    assert rel_max(cpu(m32(Xd)), want_fp32.numpy()) < TOL
    assert rel_max(cpu(y), cpu(m32(Xd))) > 1e-5
    # invert stays on the fp32 chain
    assert rel_max(cpu(m.invert(m32(Xd))), cpu(m32.invert(m32(Xd)))) == 0.0
    with pytest.raises(ValueError):
        A.Magnitude(bank_dtype="fp8")


@pytest.mark.gpu
def test_fixed_length_epilogue_forms_match_the_generic_one(dev, monkeypatch):
    """The fixed-length epilogue of the fused n_fft-1024 forward (round 3: compile-time walk lengths, contrast and
    power) in its three instantiations -- log1p / |X| with and without the spectrum (the headline), log / |X|^2
    features only (the log-mel of BASELINE configs[3]) -- against the generic epilogue on the same input and against
    the oracle; a bank with other walk lengths must still take the generic one."""
    from acids_transforms_amd import ops
    g = torch.Generator().manual_seed(4242)
    x = (torch.randn(5, 30000, generator=g) * 0.2).to(dev)
    st = A.STFT().to(dev)
    mag = A.Magnitude(n_mels=128, mode="unipolar", contrast="log1p").to(dev)
    mag.scale_data(st(x))
    off, sc = mag._affine()
    band = mag._banded()
    assert [int(v) for v in band.pass_len[:2]] == [32, 8]          # what the fixed-length forms are instantiated for
    cases = [("log1p", 1, True), ("log1p", 1, False), ("log", 2, False)]
    for contrast, power, want_X in cases:
        eps = 1e-10 if contrast == "log" else mag._eps
        args = dict(contrast=contrast, offset=off, scale=sc, eps=eps, power=power, want_spectrum=want_X)
        Xa, _, fa = ops.stft_mel_forward(x, st.window[:1024], band, **args)
        with variant("epilogue", 1):
            Xb, _, fb = ops.stft_mel_forward(x, st.window[:1024], band, **args)
        assert rel_max(cpu(fa), cpu(fb)) < 2e-6, (contrast, power, want_X)
        if want_X:
            assert rel_max(cpu(torch.view_as_real(Xa)), cpu(torch.view_as_real(Xb))) < 2e-6
        Xr = O.stft_forward(x.cpu(), O.hann_window(1024), 1024, 256)
        fwd, _ = O.magnitude_banks(O.melscale_fbanks(513, 0.0, 22050.0, 128, 44100))
        m = Xr.abs() ** power
        want = O.affine(O.contrast(torch.matmul(m, fwd), contrast, eps), float(off), float(sc))
        assert rel_max(cpu(fa), want.numpy()) < TOL, (contrast, power, want_X)


@pytest.mark.gpu
def test_fixed_length_epilogue_with_huge_finite_samples(dev):
    """the fixed-length epilogue does not clear the slab behind bin 512 and relies on zero weight x finite
    leftover = 0.  Samples near the top of the range that keeps |X|^2 finite (1e15: |X| ~ 1e17, |X|^2 ~ 1e35) must give
    finite features, equal to the generic epilogue's (which writes explicit zeros) to rounding."""
    from acids_transforms_amd import ops
    g = torch.Generator().manual_seed(99)
    x = (torch.randn(3, 20000, generator=g) * 1e15).to(dev)
    st = A.STFT().to(dev)
    mag = A.Magnitude(n_mels=128, mode=None, contrast="log1p").to(dev)
    band = mag._banded()
    _, _, fa = ops.stft_mel_forward(x, st.window[:1024], band, contrast="log1p", offset=None, scale=None, eps=mag._eps)
    with variant("epilogue", 1):
        _, _, fb = ops.stft_mel_forward(x, st.window[:1024], band, contrast="log1p", offset=None, scale=None, eps=mag._eps)
    assert bool(torch.isfinite(fa).all()) and bool(torch.isfinite(fb).all())
    assert rel_max(cpu(fa), cpu(fb)) < 2e-6


@pytest.mark.gpu
def test_fixed_form_projection_equals_the_fused_epilogue_bit_for_bit(dev, monkeypatch):
    """`Magnitude.forward` on a stored spectrum (the fixed-form stand-alone projection, round 3) and the fused
    `STFT + Magnitude` kernel share their arithmetic: for the headline bank the two feature tensors are the SAME bits
    (ADVICE r2 had found them 1.5 ulp apart); both are within 1e-5 of the oracle, and the general projection kernel
    (`variant("epilogue", 1)`, C ABI at_set_variant) agrees to 2e-6."""
    g = torch.Generator().manual_seed(777)
    x = (torch.randn(7, 40000, generator=g) * 0.2).to(dev)
    st = A.STFT().to(dev)
    mag = A.Magnitude(n_mels=128, mode="unipolar", contrast="log1p").to(dev)
    X = st(x)
    mag.scale_data(X)
    Xf, fused = mag.forward_fused(st, x, return_spectrum=True)
    assert torch.equal(torch.view_as_real(Xf), torch.view_as_real(X))
    alone = mag(X)
    assert torch.equal(alone, fused)
    with variant("epilogue", 1):
        general = mag(X)
    assert rel_max(cpu(alone), cpu(general)) < 2e-6
    Xr = O.stft_forward(x.cpu(), O.hann_window(1024), 1024, 256)
    fwd, _ = O.magnitude_banks(O.melscale_fbanks(513, 0.0, 22050.0, 128, 44100))
    want = O.magnitude_forward(Xr, fwd, "log1p", float(mag.norm.offset), float(mag.norm.scale))
    assert rel_max(cpu(alone), want.numpy()) < TOL
    # rows that are not a multiple of anything, a single row
    for rows in (1, 5, 1027):
        Xs = X.reshape(-1, 513)[:rows].contiguous()
        with variant("epilogue", 1):
            ref = mag(Xs)
        assert rel_max(cpu(mag(Xs)), cpu(ref)) < 2e-6, rows


@pytest.mark.gpu
def test_packed_epilogue_is_only_taken_by_513_filter_banks(dev):
    """the packed fixed-length epilogue hard-codes a 513-float feature row, but was selected by the nine pass
    lengths alone.  A 520-filter bank with exactly the default bank's pass lengths (12, 8, 8, 4, 4, 4, 4, 0, 0 bins: 64
    filters of three quads, 128 of two, 256 of one, 72 empty) must take the generic epilogue: same features as with
    `variant("epilogue", 1)`, bit for bit, and as the dense contraction."""
    from acids_transforms_amd import ops
    from acids_transforms_amd.utils.banded import BandedBank
    g = torch.Generator().manual_seed(520)
    K, N = 513, 520
    bank = torch.zeros(K, N)
    quads = [3] * 64 + [2] * 128 + [1] * 256 + [0] * 72
    perm = torch.randperm(N, generator=g).tolist()
    for f, nq in zip(perm, quads):
        if nq:
            q0 = int(torch.randint(0, (K - 4 * nq) // 4, (1,), generator=g))
            bank[4 * q0:4 * q0 + 4 * nq, f] = torch.rand(4 * nq, generator=g) + 0.1
    band = BandedBank(bank)
    assert band.n_passes == 9 and [int(v) for v in band.pass_len[:9]] == [12, 8, 8, 4, 4, 4, 4, 0, 0] and band.fusable
    x = (torch.randn(3, 20000, generator=g) * 0.2).to(dev)
    st = A.STFT().to(dev)
    eps = float(torch.finfo(torch.float32).eps)
    _, _, fa = ops.stft_mel_forward(x, st.window[:1024], band, contrast="log1p", eps=eps, want_spectrum=False)
    with variant("epilogue", 1):
        _, _, fb = ops.stft_mel_forward(x, st.window[:1024], band, contrast="log1p", eps=eps, want_spectrum=False)
    assert fa.shape == (3, 79, N)
    assert torch.equal(fa, fb)
    want = torch.log1p(torch.matmul(st(x).abs().cpu(), bank))
    assert rel_max(cpu(fa), want.numpy()) < TOL
