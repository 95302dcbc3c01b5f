"""GPU parity: STFT / DGT forward, ISTFT, realtime frames -- HIP kernels through the
C ABI vs (a) goldens from the reference, (b) the CPU oracle on seeded inputs,
(c) size-independent properties at BASELINE sizes.

Tolerance (north_star): <= 1e-5 relative for fp32 STFT, measured normwise as
max|d| / max|ref| per tensor (SURVEY.md hard part 4)."""
import numpy as np
import pytest
import torch

import acids_transforms_amd as A
from conftest import rel_max
from acids_transforms_amd._lib import variant
from oracle import oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-5


def cpu(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("name,n,h", [("stft", 1024, 256), ("stft", 128, 32), ("dgt", 1024, 256), ("dgt", 128, 32)])
def test_forward_inverse_golden(golden, dev, name, n, h):
    g = golden("g2_stft")
    cls = A.STFT if name == "stft" else A.DGT
    m = cls(n_fft=n, hop_length=h).to(dev)
    x = torch.from_numpy(g["x"]).to(dev)
    X = m(x)
    k = "%s_%d_%d" % (name, n, h)
    assert X.shape == g["X_" + k].shape and X.dtype == torch.complex64
    assert rel_max(cpu(X), g["X_" + k]) < TOL
    y = m.invert(X)
    assert y.shape == g["y_" + k].shape
    assert rel_max(cpu(y), g["y_" + k]) < TOL
    # inverse fed with the reference's own spectrum
    y2 = m.invert(torch.from_numpy(g["X_" + k]).to(dev))
    assert rel_max(cpu(y2), g["y_" + k]) < TOL
    # phase buffer side effect (lazy by default, eager on request)
    pb = cpu(m.phase_buffer)
    big = np.abs(g["X_" + k]) > 1e-3 * np.abs(g["X_" + k]).max()
    d = np.angle(np.exp(1j * (pb - g["phase_buffer_" + k])))
    assert pb.shape == g["phase_buffer_" + k].shape and np.abs(d[big]).max() < 1e-3
    m.eager_phase = True
    m(x)
    assert np.abs(np.angle(np.exp(1j * (cpu(m.phase_buffer) - g["phase_buffer_" + k])))[big]).max() < 1e-3


def test_multidim_odd_length_golden(golden, dev):
    g = golden("g2_stft")
    m = A.STFT().to(dev)
    X = m(torch.from_numpy(g["x_md"]).to(dev))          # (3, 2, 3001): odd length -> unaligned clips
    assert X.shape == g["X_md"].shape
    assert rel_max(cpu(X), g["X_md"]) < TOL
    y = m.invert(X)
    assert y.shape == g["y_md"].shape and rel_max(cpu(y), g["y_md"]) < TOL


def test_keep_input_and_time_golden(golden, dev):
    g = golden("g2_stft")
    x = torch.from_numpy(g["x"]).to(dev)
    m = A.STFT().to(dev)
    X = m(x)
    y = m.invert(X.abs(), inversion_mode="keep_input")
    assert rel_max(cpu(y), g["y_keep_input"]) < TOL
    d = A.DGT().to(dev)
    Xd = d(x)
    assert rel_max(cpu(d.invert(Xd.abs(), inversion_mode="keep_input")), g["y_dgt_keep_input"]) < TOL
    _, tt = m.forward_with_time(x, torch.from_numpy(g["fwt_time_in"]).to(dev))
    assert np.allclose(cpu(tt), g["fwt_time_out"], rtol=1e-6)


@pytest.mark.parametrize("B,L,n,h", [(3, 5000, 1024, 256), (1, 1024, 1024, 256), (5, 2049, 1024, 256),
                                      (2, 7777, 1024, 128), (2, 7777, 1024, 512), (4, 4100, 256, 64),
                                      (2, 3000, 2048, 512), (3, 700, 64, 16), (1, 40000, 8192, 2048),
                                      (2, 600, 32, 8), (1, 513, 1024, 256),
                                      # hops that do not divide n_fft (frame-at-a-time kernels, gather overlap-add)
                                      (2, 9001, 1024, 100), (3, 6000, 1024, 300), (2, 5000, 1024, 1000),
                                      (2, 8000, 1024, 1024), (2, 3001, 256, 100), (1, 30000, 2048, 441)])
def test_vs_oracle_seeded(dev, B, L, n, h):
    g = torch.Generator().manual_seed(B * 1000 + L)
    x = torch.randn(B, L, generator=g)
    for cls, w, iw in [(A.STFT, O.hann_window(n), O.hann_window(n)),
                       (A.DGT, O.gauss_window(n), None)]:
        if iw is None:
            iw = O.dual_window(w, n, h)
        m = cls(n_fft=n, hop_length=h).to(dev)
        X = m(x.to(dev))
        Xr = O.stft_forward(x, w, n, h)
        assert X.shape == Xr.shape
        assert rel_max(cpu(X), Xr.numpy()) < TOL
        if Xr.shape[-2] > 1 and n // h >= 2:
            y = m.invert(X)
            yr = O.istft(Xr, iw, n, h)
            assert y.shape == yr.shape
            assert rel_max(cpu(y), yr.numpy()) < TOL


def test_polar_inverse_vs_oracle(dev):
    g = torch.Generator().manual_seed(5)
    mag = torch.rand(3, 9, 513, generator=g)
    ph = (torch.rand(3, 9, 513, generator=g) - 0.5) * 2e5      # PGHI-sized unwrapped phases (~1e5 rad)
    m = A.DGT().to(dev)
    y = m._istft(mag=mag.to(dev), phase=ph.to(dev))
    yr = O.polar_istft(mag, ph, O.dual_window(O.gauss_window(1024), 1024, 256), 1024, 256)
    assert rel_max(cpu(y), yr.numpy()) < 2e-5    # sincos of 1e5 rad: one fp32 ulp of the argument is 8e-3 rad


def test_realtime_frames_golden(golden, dev):
    g = golden("g6_overlap_add")
    for key in ["1024_256_4096", "64_16_128"]:
        n, h, chunk = [int(v) for v in key.split("_")]
        rs = A.RealtimeSTFT(n_fft=n, hop_length=h).to(dev)
        rd = A.RealtimeDGT(n_fft=n, hop_length=h).to(dev)
        fr = torch.from_numpy(g["frames_%s_1" % key]).to(dev)
        X = rs(fr)
        assert rel_max(cpu(X), g["X_%s_1" % key]) < TOL
        assert rel_max(cpu(rs.invert(X)), g["yframes_%s_1" % key]) < TOL
        Xd = rd(fr)
        assert rel_max(cpu(Xd), g["Xd_%s_1" % key]) < TOL
        assert rel_max(cpu(rd.invert(Xd)), g["ydframes_%s_1" % key]) < TOL
        # overlapping strided view (what OverlapAdd.forward returns) is consumed without a copy
        from acids_transforms_amd.utils.misc import frame
        xs = torch.from_numpy(g["x_" + key]).to(dev)
        view = frame(xs, n, h, -1)
        Xv = rs(view)
        Xr = O.rt_forward(O.frame(torch.from_numpy(g["x_" + key]), n, h), O.hann_window(n))
        assert Xv.shape == Xr.shape and rel_max(cpu(Xv), Xr.numpy()) < TOL


def test_full_size_properties(dev):
    """BASELINE config sizes (batch of 4-s clips): reconstruction, linearity, Parseval."""
    B, L = 64, 176400
    torch.manual_seed(0)
    x = torch.randn(B, L, device=dev) * 0.1
    m = A.STFT().to(dev)
    X = m(x)
    assert X.shape == (B, 690, 513)
    y = m.invert(X)
    assert y.shape == (B, 176384)
    assert float((y - x[:, :176384]).abs().max()) < 2e-6          # perfect reconstruction (reference: 7.2e-7)
    x2 = torch.randn(B, L, device=dev) * 0.1
    lin = m(x + 2.0 * x2) - (X + 2.0 * m(x2))
    assert float(lin.abs().max()) / float(X.abs().max()) < 2e-6   # linearity
    # Parseval per interior frame: sum |X_k|^2 (two-sided) == N * sum (w x)^2
    w = m.window[:1024]
    t = 100
    seg = x[:, t * 256 - 512:t * 256 + 512] * w
    e_time = (seg ** 2).sum(-1) * 1024
    Xt = X[:, t]
    e_freq = (Xt.abs() ** 2).sum(-1) * 2 - Xt[:, 0].abs() ** 2 - Xt[:, 512].abs() ** 2
    assert float(((e_time - e_freq).abs() / e_time).max()) < 1e-5
    # DGT: dual window + istft envelope -> gain ~1.17 (reference quirk), stable across the clip
    d = A.DGT().to(dev)
    yd = d.invert(d(x))
    ratio = (yd[:, 4096:-4096] * x[:, 4096:176384 - 4096]).sum() / (x[:, 4096:176384 - 4096] ** 2).sum()
    assert 1.13 < float(ratio) < 1.21


def test_fails_loudly_on_cpu_tensor():
    with pytest.raises(A.AcidsHipError):
        A.STFT()(torch.randn(2, 4096))


def test_griffin_lim(dev):
    """STFT's default inversion mode: same algorithm as the oracle restatement from identical starting angles;
    and with the random start the resynthesis' magnitudes converge to the target (spectral convergence)."""
    g = torch.Generator().manual_seed(12)
    x = torch.randn(2, 6000, generator=g) * 0.1
    m = A.STFT().to(dev)
    assert m.inversion_mode == "griffin_lim"
    mag = m(x.to(dev)).abs()
    a0 = torch.rand(mag.shape, dtype=torch.complex64, generator=g)
    y = m.griffin_lim(mag, n_iter=5, angles0=a0.to(dev))
    yr = O.griffinlim(mag.cpu(), O.hann_window(1024), 1024, 256, a0, n_iter=5)
    assert y.shape == yr.shape
    assert rel_max(cpu(y), yr.numpy()) < 2e-4          # 5 nonlinear iterations amplify fp32 round-off
    y30 = m.invert(mag)                                 # default mode, 30 iterations, random start
    assert y30.shape == (2, 5888) and bool(torch.isfinite(y30).all())
    mag30 = m(torch.nn.functional.pad(y30, (0, 6000 - 5888))).abs()
    sc = float((mag30[:, 2:-3] - mag[:, 2:-3]).norm() / mag[:, 2:-3].norm())
    assert sc < 0.35, sc
    d = A.DGT(inversion_mode="griffin_lim").to(dev)
    assert d.invert(d(x.to(dev)).abs()).shape == (2, 5888)


@pytest.mark.gpu
def test_results_do_not_depend_on_the_batch(dev):
    """A clip's spectrum, features, inverse and PGHI reconstruction are the same bits whether it runs alone or
    inside a 700-clip batch: the launchers cut clips into runs by occupancy, and PGHI sizes the LDS share of each
    heap by the batch, so neither may leak into the arithmetic."""
    torch.manual_seed(11)
    x0 = torch.randn(1, 40000, device=dev) * 0.1
    xb = torch.cat([x0, torch.randn(699, 40000, device=dev) * 0.1])
    for make in (A.STFT, A.DGT):
        t = make().to(dev)
        X1, Xb = t(x0), t(xb)
        assert torch.equal(X1[0], Xb[0])
        assert torch.equal(t.invert(X1)[0], t.invert(Xb)[0])
    mg = A.Magnitude(n_mels=128).to(dev)
    st = A.STFT().to(dev)
    comp = st + mg
    comp.scale_data(xb[:4])
    assert torch.equal(comp(x0)[0], comp(xb)[0])
    d = A.DGT().to(dev)
    m1 = d(x0).abs()
    mb = d(torch.cat([x0, torch.randn(4199, 40000, device=dev) * 0.1])).abs()
    alone = d.invert(m1, inversion_mode="pghi")[0]          # 16383-entry LDS heap top (checked against the oracle
    for B in (300, 700, 1100, 2100, 4200):                   # elsewhere); 8191, 4095, 2047, 1023 and 511 entries
        assert torch.equal(alone, d.invert(mb[:B], inversion_mode="pghi")[0]), B
    # a decaying clip reseeds hundreds of times: the segment bounds behind the reseeds (1024, 512 or 192 segments per
    # clip, by batch size) must not show either
    xd = x0 * torch.exp(-8.0 * torch.arange(40000, device=dev) / 44100.0)
    md = d(xd).abs()
    alone = d.pghi(md, d.tolerance)[0]
    assert int((alone == 0).sum()) > 1000                     # many bins under the tolerance: a sparse flood
    for B in (700, 1100, 2100):
        assert torch.equal(alone, d.pghi(torch.cat([md, mb[1:B]]), d.tolerance)[0]), B


@pytest.mark.gpu
@pytest.mark.parametrize("hop", [128, 256, 512])
def test_other_hops_take_the_sliding_kernels(dev, hop):
    """n_fft = 1024 with hop = n_fft/8, /4 (reference default), /2: the sliding-window forward and the fused
    overlap-add inverse (8, 4 or 2 overlapping frames), STFT and DGT windows, clip lengths down to the shortest torch
    accepts, against the oracle; plus the phase side output and batch independence."""
    for cls in (A.STFT, A.DGT):
        t = cls(n_fft=1024, hop_length=hop).to(dev)
        w, wi = t.window[:1024].cpu(), t.inv_window[:1024].cpu()
        for L in (40000, 4096, 5000, 2047, 1536, 1024, 513):
            torch.manual_seed(hop + L)
            x = torch.randn(3, L) * 0.1
            X = t(x.to(dev))
            Xr = O.stft_forward(x, w, 1024, hop)
            assert X.shape == Xr.shape
            assert rel_max(X.cpu().numpy(), Xr.numpy()) < TOL, (cls.__name__, L)
            y, yr = t.invert(X).cpu(), O.istft(Xr, wi, 1024, hop)
            assert y.shape == yr.shape
            if yr.numel():
                assert rel_max(y.numpy(), yr.numpy()) < TOL, (cls.__name__, L)
    from acids_transforms_amd import ops
    st = A.STFT(n_fft=1024, hop_length=hop).to(dev)
    st.eager_phase = True                       # angle() inside the forward kernel, like the reference's forward
    torch.manual_seed(3)
    x0 = torch.randn(1, 30000, device=dev) * 0.1
    xb = torch.cat([x0, torch.randn(299, 30000, device=dev) * 0.1])
    X0, Xb = st(x0), st(xb)
    ph = st.phase_buffer
    assert torch.equal(X0[0], Xb[0]) and torch.equal(st.invert(X0)[0], st.invert(Xb)[0])
    big = Xb.abs().reshape(ph.shape) > 1e-2
    d = (ph - Xb.angle().reshape(ph.shape) + np.pi) % (2 * np.pi) - np.pi
    assert float(d[big].abs().max()) < 1e-5
    # polar input to the fused inverse
    yp = ops.istft(None, st.inv_window[:1024], 1024, hop, env16=st._env16, mag=Xb.abs(), phase=Xb.angle())
    assert rel_max(yp.cpu().numpy(), st.invert(Xb).cpu().numpy()) < TOL


@pytest.mark.gpu
@pytest.mark.parametrize("name,h", [("stft", 128), ("stft", 512), ("dgt", 128), ("dgt", 512)])
def test_other_hops_golden(golden, dev, name, h):
    """G14: outputs of the reference itself at n_fft = 1024, hop 128 / 512 -- forward, complex inverse, the DGT dual
    window, and a PGHI reconstruction at hop 128."""
    g = golden("g14_other_hops")
    m = (A.STFT if name == "stft" else A.DGT)(n_fft=1024, hop_length=h).to(dev)
    k = "%s_%d" % (name, h)
    x = torch.from_numpy(g["x"]).to(dev)
    assert rel_max(cpu(m.inv_window[:1024]), g["inv_window_" + k]) < 1e-6
    X = m(x)
    assert X.shape == g["X_" + k].shape and rel_max(cpu(X), g["X_" + k]) < TOL
    y = m.invert(torch.from_numpy(g["X_" + k]).to(dev))
    assert y.shape == g["y_" + k].shape and rel_max(cpu(y), g["y_" + k]) < TOL
    if name == "dgt" and h == 128:
        yp = m.invert(torch.from_numpy(g["pghi_mag"]).to(dev), inversion_mode="pghi")
        ref = g["pghi_y"]
        assert yp.shape == ref.shape
        err = cpu(yp) - ref
        assert 10 * np.log10((ref ** 2).sum() / max((err ** 2).sum(), 1e-30)) >= 40.0     # same bar as the hop-256 cases


@pytest.mark.gpu
@pytest.mark.parametrize("n,h", [(2048, 512), (512, 128), (4096, 1024), (256, 64), (400, 160), (1000, 250), (441, 147)])
def test_other_sizes_golden(golden, dev, n, h):
    """G16: outputs of the reference itself at other FFT sizes (register-core kernels at 2048 / 512 / 4096 / 256, mixed-radix
    kernels at 400 / 1000 and the odd 441): windows, forward and complex inverse of STFT and DGT; at n_fft 400 a PGHI
    phase (same visited set, same values) and its reconstruction."""
    g = golden("g16_other_sizes")
    x = torch.from_numpy(g["x_%d" % n]).to(dev)
    for name, cls in (("stft", A.STFT), ("dgt", A.DGT)):
        m = cls(n_fft=n, hop_length=h).to(dev)
        k = "%s_%d" % (name, n)
        assert rel_max(cpu(m.window[:n]), g["window_" + k]) < 1e-6 and rel_max(cpu(m.inv_window[:n]), g["inv_window_" + k]) < 1e-6
        X = m(x)
        assert X.shape == g["X_" + k].shape and rel_max(cpu(X), g["X_" + k]) < TOL
        y = m.invert(torch.from_numpy(g["X_" + k]).to(dev))
        assert y.shape == g["y_" + k].shape and rel_max(cpu(y), g["y_" + k]) < TOL
    if n == 400:
        d = A.DGT(n_fft=400, hop_length=100).to(dev)
        mag = torch.from_numpy(g["pghi_mag_400"]).to(dev)
        ph, ref = cpu(d.pghi(mag[0], d.tolerance)), g["pghi_phase_400"]
        assert np.array_equal(ph == 0, ref == 0)
        assert np.all(np.abs(ph - ref) <= 1e-3 + 8 * np.spacing(np.abs(ref).astype(np.float32)) + 1e-5 * np.abs(ref))
        yp, yref = cpu(d.invert(mag, inversion_mode="pghi")), g["pghi_y_400"]
        assert yp.shape == yref.shape
        assert 10 * np.log10((yref ** 2).sum() / max(((yp - yref) ** 2).sum(), 1e-30)) >= 40.0


@pytest.mark.gpu
@pytest.mark.parametrize("hop", [128, 256, 512])
def test_griffinlim_update_fused_into_the_inverse(dev, hop):
    """at_istft_griffinlim == at_istft(at_griffinlim_update(...)): the phase update taken while the inverse kernel
    loads its frames (first iteration without, later ones with a previous spectrum), and the whole 30-iteration
    inversion against the unfused composition from the same random start."""
    from acids_transforms_amd import ops
    torch.manual_seed(hop)
    st = A.STFT(n_fft=1024, hop_length=hop).to(dev)
    x = torch.randn(3, 20000, device=dev) * 0.1
    mag = st(x).abs()
    rebuilt = st(torch.randn(3, 20000, device=dev) * 0.1)
    tprev = st(torch.randn(3, 20000, device=dev) * 0.1)
    w, env = st.inv_window[:1024], st._env16
    for tp in (None, tprev):
        want = ops.istft(ops.griffinlim_update(mag, rebuilt, tp, 0.99 / 1.99), w, 1024, hop, env16=env)
        got = ops.istft_griffinlim(mag, rebuilt, tp, 0.99 / 1.99, w, 1024, hop, env)
        assert got.shape == want.shape and rel_max(cpu(got), cpu(want)) < TOL
    angles0 = torch.rand(mag.shape, dtype=torch.complex64, device=dev)
    y = st.griffin_lim(mag, n_iter=8, angles0=angles0)
    X = ops.scale_complex(mag, angles0)
    tp = None
    for _ in range(8):
        rb = ops.stft_forward(ops.istft(X, w, 1024, hop, env16=env), w, 1024, hop, center=True)
        X = ops.griffinlim_update(mag, rb, tp, 0.99 / 1.99)
        tp = rb
    # eight iterations amplify the 1e-7 differences of a single step; at 50 % overlap (Hann envelope close to zero at
    # the frame edges) the iteration is far less contractive
    assert rel_max(cpu(y), cpu(ops.istft(X, w, 1024, hop, env16=env))) < (1e-4 if hop < 512 else 5e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [8, 16, 32, 64, 256, 512, 2048, 4096, 16384])
def test_every_power_of_two_size(dev, n):
    """The generic kernels (radix-4 / radix-2 Stockham in LDS) over the whole range the reference's window buffers
    allow (stft.py:10 MAX_NFFT = 16384), hops n/8, n/4, n/2, STFT and DGT, forward and inverse against the oracle."""
    for hop in (max(1, n // 8), n // 4, n // 2):
        torch.manual_seed(n + hop)
        x = torch.randn(2, max(3 * n, 2000)) * 0.1
        for cls in (A.STFT, A.DGT):
            t = cls(n_fft=n, hop_length=hop).to(dev)
            X = t(x.to(dev))
            Xr = O.stft_forward(x, t.window[:n].cpu(), n, hop)
            assert X.shape == Xr.shape and rel_max(cpu(X), Xr.numpy()) < TOL, (hop, cls.__name__)
            y, yr = t.invert(X), O.istft(Xr, t.inv_window[:n].cpu(), n, hop)
            assert y.shape == yr.shape and rel_max(cpu(y), yr.numpy()) < TOL, (hop, cls.__name__)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [6, 12, 77, 254, 400, 441, 1000, 1200, 1536, 1920, 2000, 2401, 6000, 8191, 12000])
def test_sizes_that_are_not_powers_of_two(dev, n):
    """torch.stft takes any n_fft (stft.py:67-75), so do the mixed-radix kernels: radix 4 / 2 / 3 / 5 / 7 stages,
    a direct DFT for other prime factors (254 = 2 x 127, 8191 prime), odd sizes at full length (77, 441, 2401, 8191),
    sizes whose twiddle table does not fit LDS (12000).  STFT and DGT, forward and inverse, against the oracle."""
    from acids_transforms_amd import ops
    hops = (max(1, n // 4), max(1, n // 3)) if n < 8000 else (n // 4,)
    for hop in hops:
        torch.manual_seed(n + hop)
        x = torch.randn(2, max(3 * n, 2000)) * 0.1
        for cls in (A.STFT, A.DGT):
            t = cls(n_fft=n, hop_length=hop).to(dev)
            X = t(x.to(dev))
            Xr = O.stft_forward(x, t.window[:n].cpu(), n, hop)
            assert X.shape == Xr.shape and rel_max(cpu(X), Xr.numpy()) < TOL, (hop, cls.__name__)
            y, yr = t.invert(X), O.istft(Xr, t.inv_window[:n].cpu(), n, hop)
            assert y.shape == yr.shape and rel_max(cpu(y), yr.numpy()) < TOL, (hop, cls.__name__)
            # polar input of the inverse (magnitude + phase) through the same kernels
            yp = ops.istft(None, t.inv_window[:n], n, hop, mag=X.abs(), phase=X.angle())
            assert rel_max(cpu(yp), yr.numpy()) < 2 * TOL, (hop, cls.__name__)


def test_bench_step_at_full_batch_vs_oracle(dev):
    """What bench.py times, at its size: B = 1024 clips x 4 s through the fused STFT+mel forward and the ISTFT; the
    launchers cut 1024 clips into runs by occupancy, so clips from both ends and the middle are checked against the
    oracle (the headline number must be a number on verified output)."""
    B, L = 1024, 176400
    gen = torch.Generator(device=dev).manual_seed(1234)
    x = torch.randn(B, L, device=dev, generator=gen) * 0.1
    stft = A.STFT().to(dev)
    mag = A.Magnitude(n_mels=128, mode="unipolar", contrast="log1p").to(dev)
    mag.scale_data(stft(x[:8]))
    assert mag.can_fuse_with(stft, x)
    X, feat = mag.forward_fused(stft, x, return_spectrum=True)
    y = stft.invert(X)
    assert X.shape == (B, 690, 513) and feat.shape == (B, 690, 128) and y.shape == (B, 176384)
    ids = [0, 511, 1023]
    xs = x[ids].cpu()
    w = O.hann_window(1024)
    Xr = O.stft_forward(xs, w, 1024, 256)
    fwd, _ = O.magnitude_banks(O.melscale_fbanks(513, 0.0, 22050.0, 128, 44100))
    fr = O.magnitude_forward(Xr, fwd, "log1p", float(mag.norm.offset), float(mag.norm.scale))
    yr = O.istft(Xr, w, 1024, 256)
    assert rel_max(cpu(X[ids]), Xr.numpy()) < TOL
    assert rel_max(cpu(feat[ids]), fr.numpy()) < TOL
    assert rel_max(cpu(y[ids]), yr.numpy()) < TOL
    # and the stage-by-stage path produces the same spectrum bit for bit / the same features within the bar
    assert torch.equal(stft(x[1000:1008]), X[1000:1008])
    assert rel_max(cpu(mag(X[1000:1008])), cpu(feat[1000:1008])) < TOL


@pytest.mark.parametrize("n", [128, 256, 512, 2048, 4096])
def test_register_core_sizes_512_and_2048(dev, n):
    """n_fft = 128 / 256 / 512 (eight / four / two frames per wave-level FFT), 2048 (two FFTs + a radix-2 stage per
    frame) and 4096 (four FFTs + a radix-4 stage) on the register core:
    forward, complex and polar inverse, phase side output, realtime frames, odd frame counts, unaligned clip
    lengths and hops, STFT and DGT windows -- against the oracle."""
    from acids_transforms_amd import ops
    for (B, L, h) in [(3, 9 * n + 7, n // 4), (2, 5 * n + 1, 100), (1, n // 2 + 3, n // 4), (5, 4 * n, n // 2),
                      (2, 7 * n + 2, n // 8), (1, 20 * n, n // 4)]:
        g = torch.Generator().manual_seed(B * 131 + L + h)
        x = torch.randn(B, L, generator=g)
        for cls, w in [(A.STFT, O.hann_window(n)), (A.DGT, O.gauss_window(n))]:
            iw = w if cls is A.STFT else O.dual_window(w, n, h)
            m = cls(n_fft=n, hop_length=h).to(dev)
            X = m(x.to(dev))
            Xr = O.stft_forward(x, w, n, h)
            assert X.shape == Xr.shape, (B, L, h)
            assert rel_max(cpu(X), Xr.numpy()) < TOL, (cls.__name__, B, L, h)
            if Xr.shape[-2] > 1 and n // h >= 2 and n % h == 0:
                y = m.invert(X)
                yr = O.istft(Xr, iw, n, h)
                assert y.shape == yr.shape and rel_max(cpu(y), yr.numpy()) < TOL, (cls.__name__, B, L, h)
                yp = m._istft(mag=X.abs(), phase=X.angle())                 # polar input path
                assert rel_max(cpu(yp), yr.numpy()) < 2e-5
    # eager phase output, realtime (pre-framed) forward / inverse
    g = torch.Generator().manual_seed(n)
    x = torch.randn(2, 6 * n, generator=g)
    st = A.STFT(n_fft=n, hop_length=n // 4).to(dev)
    st.eager_phase = True
    X = st(x.to(dev))
    assert rel_max(cpu(st.phase_buffer), cpu(X.angle())) < 1e-5
    rs = A.RealtimeSTFT(n_fft=n, hop_length=n // 4).to(dev)
    fr = torch.randn(3, 5, n, generator=g)
    Xf = rs(fr.to(dev))
    Xfr = O.rt_forward(fr, O.hann_window(n))
    assert rel_max(cpu(Xf), Xfr.numpy()) < TOL
    assert rel_max(cpu(rs.invert(Xf)), O.rt_invert(Xfr, O.hann_window(n)).numpy()) < TOL
    assert rel_max(cpu(ops.irfft_frames(None, rs.inv_window[:n], n, mag=Xf.abs(), phase=Xf.angle())),
                   O.rt_invert(Xfr, O.hann_window(n)).numpy()) < 2e-5


@pytest.mark.parametrize("n", [2048, 4096])
def test_sliding_aligned_forward_2048_4096_full_batch(dev, n, monkeypatch):
    """The hop = n/4 forward of n_fft 2048 / 4096 (sliding window in registers + aligned stream stores, round 3) at the
    size where a wave's run is long enough for the column rotation to wrap inside it (1024 clips x 4 s: 87 / 44 frames
    per run): clips {0, 511, 1023} against the oracle, and the whole tensor against the frame-at-a-time kernel."""
    B, L, h = 1024, 176400, n // 4
    g = torch.Generator().manual_seed(n + 5)
    x = torch.randn(B, L, generator=g) * 0.1
    st = A.STFT(n_fft=n, hop_length=h).to(dev)
    xd = x.to(dev)
    X = st(xd)
    ids = [0, 511, 1023]
    Xr = O.stft_forward(x[ids], O.hann_window(n), n, h)
    assert X.shape[1:] == Xr.shape[1:]
    assert rel_max(cpu(X[ids]), Xr.numpy()) < TOL
    with variant("frame_kernels", 1):
        Xf = st(xd)
    scale = float(torch.view_as_real(Xf).abs().max())
    assert float((torch.view_as_real(X) - torch.view_as_real(Xf)).abs().max()) < 3e-6 * scale
    del X, Xf
    # a clip length that is not a multiple of the hop, few clips (short runs: heads and carried tails everywhere), DGT window
    for (Bs, Ls) in [(3, 40 * n + 4 * 37), (2, 9 * n), (1, 300 * h + 8)]:
        xs = torch.randn(Bs, Ls, generator=g)
        d = A.DGT(n_fft=n, hop_length=h).to(dev)
        assert rel_max(cpu(d(xs.to(dev))), O.stft_forward(xs, O.gauss_window(n), n, h).numpy()) < TOL, (Bs, Ls)


def test_sliding_aligned_forward_512_full_batch(dev, monkeypatch):
    """n_fft 512 at hop 128: frame PAIRS per wave FFT, sliding window, aligned stream stores with the second frame of
    a pair moved up one lane (round 3).  Full batch (345-pair runs: the rotation wraps inside them) against the oracle
    and the frame-pair-at-a-time kernel; then odd / even frame counts, short clips, few clips (short runs), DGT."""
    n, h = 512, 128
    B, L = 1024, 176400
    g = torch.Generator().manual_seed(517)
    x = torch.randn(B, L, generator=g) * 0.1
    st = A.STFT(n_fft=n, hop_length=h).to(dev)
    xd = x.to(dev)
    X = st(xd)
    ids = [0, 511, 1023]
    Xr = O.stft_forward(x[ids], O.hann_window(n), n, h)
    assert X.shape[1:] == Xr.shape[1:]
    assert rel_max(cpu(X[ids]), Xr.numpy()) < TOL
    with variant("frame_kernels", 1):
        Xf = st(xd)
    scale = float(torch.view_as_real(Xf).abs().max())
    assert float((torch.view_as_real(X) - torch.view_as_real(Xf)).abs().max()) < 3e-6 * scale
    del X, Xf
    for (Bs, Ls) in [(3, 40 * n + 2 * 37), (2, 9 * n), (1, 300 * h + 8), (5, 512), (4, 640), (2, 33 * h), (1, 34 * h + 6)]:
        xs = torch.randn(Bs, Ls, generator=g)
        for cls, w in ((A.DGT, O.gauss_window(n)), (A.STFT, O.hann_window(n))):
            m = cls(n_fft=n, hop_length=h).to(dev)
            assert rel_max(cpu(m(xs.to(dev))), O.stft_forward(xs, w, n, h).numpy()) < TOL, (Bs, Ls)


@pytest.mark.parametrize("n", [512, 1024, 2048, 4096])
def test_c_abi_output_that_is_not_512_byte_aligned(dev, n):
    """The aligned-stream forward kernels need a 512-byte aligned output; torch allocations are, a C-ABI caller's
    pointer need not be.  An output that starts 8 bytes into a buffer must take the row-store kernels and give the
    same spectrum (through the C ABI directly, as such a caller would)."""
    from acids_transforms_amd import _lib
    from acids_transforms_amd._lib import ptr, stream_ptr, check
    h = n // 4
    g = torch.Generator().manual_seed(n + 9)
    x = (torch.randn(3, 24 * n, generator=g) * 0.1).to(dev)
    st = A.STFT(n_fft=n, hop_length=h).to(dev)
    want = st(x)
    B, T, F = want.shape
    buf = torch.zeros(B * T * F + 1, dtype=torch.complex64, device=dev)
    out = buf[1:]                                             # 8 bytes past the allocation's start
    assert out.data_ptr() % 512 == 8
    check(_lib.lib().at_stft_forward(ptr(x), B, x.shape[1], x.shape[1], T, n, h, 1, ptr(st.window[:n].contiguous()),
                                     ptr(out), None, stream_ptr()), "at_stft_forward")
    torch.cuda.synchronize()
    got = out.reshape(B, T, F)
    assert rel_max(cpu(torch.view_as_real(got)), cpu(torch.view_as_real(want))) < 2e-6
    assert complex(buf[0]) == 0j                              # nothing written in front of the output


def test_float64_inputs_raise_instead_of_being_narrowed(dev):
    """a float64 / complex128 operand is an error that names the way out, never a silent cast to fp32
    (round 3 cast: `x.float()`); half and integer inputs are still widened."""
    x = torch.randn(2, 8192, device=dev)
    st = A.STFT().to(dev)
    X = st(x)
    for call in (lambda: st(x.double()), lambda: st.invert(X.to(torch.complex128)), lambda: st.invert(X.abs().double()),
                 lambda: A.Magnitude(mode=None).to(dev)(X.to(torch.complex128)), lambda: A.DGT().to(dev)(x.double()),
                 lambda: A.MFCC().to(dev)(x.double()), lambda: A.Phase(mode=None).to(dev)(X.to(torch.complex128)),
                 lambda: A.RealtimeSTFT().to(dev)(x.double().reshape(2, 8, 1024))):
        with pytest.raises(A.AcidsHipError, match="float64|complex128"):
            call()
    assert rel_max(cpu(st(x.half())), cpu(st(x.half().float()))) == 0.0
    # the explicit way in for reference call sites that hand over float64 (numpy / soundfile audio): narrowed on
    # entry, float32 results, only while the switch is on
    with A.allow_fp64_narrowing():
        Xd = st(x.double())
        assert Xd.dtype == torch.complex64 and torch.equal(Xd, X)
        assert torch.equal(st.invert(X.to(torch.complex128)), st.invert(X))
    with pytest.raises(A.AcidsHipError, match="allow_fp64_narrowing"):
        st(x.double())


def test_tiled_inverse_is_the_long_run_inverse_bit_for_bit(dev):
    """Round 5: full batches take istft1024_tile_kernel (a workgroup owns a tile of consecutive frames, the overlap state
    crosses the cuts between its waves through LDS, additions in frame order); `variant("istft_runs", 1)` forces the
    long-run kernel.  Every frame count from 64 to 300 -- every remainder of the tile and of a wave's share, clip ends
    inside the first, middle and last wave of a tile, last waves holding 0, 1, 2 frames -- complex and polar input, STFT
    and DGT windows: identical bits, and the long-run kernel is the one the goldens and the oracle pin."""
    from acids_transforms_amd import ops
    g = torch.Generator(device=dev).manual_seed(77)
    st = A.STFT().to(dev)
    dg = A.DGT().to(dev)
    B = 600                                   # >= two tiles per workgroup the chip holds, at every T below
    Xall = torch.randn(B, 300, 513, 2, device=dev, generator=g)
    for T in list(range(64, 300)) + [300]:
        X = torch.view_as_complex(Xall[:, :T].contiguous())
        mod = st if T % 2 else dg
        y_tile = mod.invert(X)
        with variant("istft_runs", 1):
            y_runs = mod.invert(X)
        assert y_tile.shape == (B, 256 * (T - 1))
        assert torch.equal(y_tile, y_runs), T
    for T in (64, 117, 118, 119, 120, 233, 234, 235):
        mag = Xall[:, :T, :, 0].abs().contiguous()
        ph = (Xall[:, :T, :, 1] * 3e4).contiguous()
        w = dg.inv_window[:1024]
        y_tile = ops.istft(None, w, 1024, 256, env16=dg._env16, mag=mag, phase=ph)
        with variant("istft_runs", 1):
            y_runs = ops.istft(None, w, 1024, 256, env16=dg._env16, mag=mag, phase=ph)
        assert torch.equal(y_tile, y_runs), T
    # and the tiled kernel itself against the oracle at the bench size's frame count
    x = torch.randn(600, 176400, device=dev, generator=g) * 0.1
    X = st(x)
    y = st.invert(X)
    ids = [0, 299, 599]
    yr = O.istft(X[ids].cpu(), O.hann_window(1024), 1024, 256)
    assert rel_max(cpu(y[ids]), yr.numpy()) < TOL
