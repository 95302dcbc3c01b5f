"""GPU parity: OverlapAdd streaming (K6/K7), the realtime round trip, MuLaw / OneHot (bit exact)."""
import numpy as np
import pytest
import torch

import acids_transforms_amd as A
from conftest import rel_max
from oracle import oracle as O

pytestmark = pytest.mark.gpu
T_ = torch.from_numpy


def cpu(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("key", ["1024_256_4096", "1024_256_1024", "64_16_128"])
def test_overlap_add_stream_golden(golden, dev, key):
    g = golden("g6_overlap_add")
    n, h, chunk = [int(v) for v in key.split("_")]
    x = T_(g["x_" + key]).to(dev)
    fa, fi = A.OverlapAdd(n, h).to(dev), A.OverlapAdd(n, h).to(dev)
    rs = A.RealtimeSTFT(n_fft=n, hop_length=h).to(dev)
    for c in range(3):
        fr = fa(x[:, c * chunk:(c + 1) * chunk])
        assert fr.shape == (2, chunk // h, n)
        assert fr.stride(-2) == h and fr.stride(-1) == 1           # zero-copy overlapping view, like the reference
        assert np.array_equal(cpu(fa.input_buffer), g["inbuf_%s_%d" % (key, c)])
        if ("frames_%s_%d" % (key, c)) in g:
            assert np.array_equal(cpu(fr), g["frames_%s_%d" % (key, c)])   # pure data movement: bit exact
        X = rs(fr)
        y = fi.invert(rs.invert(X))
        assert rel_max(cpu(y), g["y_%s_%d" % (key, c)]) < 1e-5
        assert rel_max(cpu(fi.output_buffer), g["outbuf_%s_%d" % (key, c)]) < 1e-5
    # streaming round trip: gain 0.75 with 768 samples of latency (Hann x Hann at 75 % overlap)
    if n == 1024:
        fa2, fi2 = A.OverlapAdd(n, h).to(dev), A.OverlapAdd(n, h).to(dev)
        ys = torch.cat([fi2.invert(rs.invert(rs(fa2(x[:, c * chunk:(c + 1) * chunk])))) for c in range(3)], -1)
        lat = n - h
        ratio = ys[:, lat + n:] / x[:, n:ys.shape[-1] - lat]
        assert abs(float(ratio.median()) - 0.75) < 1e-4


def test_overlap_add_vs_oracle_multidim_and_errors(dev):
    g = torch.Generator().manual_seed(8)
    x = torch.randn(2, 3, 3 * 512, generator=g)
    oa, oi = A.OverlapAdd(256, 64).to(dev), A.OverlapAdd(256, 64).to(dev)
    ra, ri = O.OverlapAddState(256, 64), O.OverlapAddState(256, 64)
    for c in range(3):
        xc = x[..., c * 512:(c + 1) * 512]
        fr = oa(xc.to(dev))
        frr = ra.forward(xc)
        assert fr.shape == frr.shape and np.array_equal(cpu(fr), frr.numpy())
        y = oi.invert(fr * 0.5)
        yr = ri.invert(frr * 0.5)
        assert y.shape == yr.shape and rel_max(cpu(y), yr.numpy()) < 1e-6
    with pytest.raises(ValueError):
        A.OverlapAdd(1024, 256).to(dev)(torch.zeros(2, 300, device=dev))    # short AND not a whole number of hops


def test_mulaw_onehot_bit_exact(dev):
    g = torch.Generator().manual_seed(9)
    x = torch.cat([torch.rand(2, 100000, generator=g) * 2 - 1,
                   torch.tensor([[-1.0, 1.0, 0.0, -0.0, 1e-8, -1e-8, 0.5, -0.5] + [0.0] * 99992] * 2)], 0)
    for ch in [256, 64]:
        m = A.MuLaw(channels=ch).to(dev)
        codes = m(x.to(dev))
        ref = O.mulaw_encode(x, ch)
        assert codes.dtype == torch.int64
        assert np.array_equal(cpu(codes), ref.numpy())                       # integer output: bit exact
        dec = m.invert(codes)
        assert np.allclose(cpu(dec), O.mulaw_decode(ref, ch).numpy(), rtol=2e-6, atol=1e-7)
    for mode in ["categorical", "channel"]:
        m = A.MuLaw(one_hot=mode).to(dev)
        oh = m(x[:, :500].to(dev))
        ref = torch.nn.functional.one_hot(O.mulaw_encode(x[:, :500]), 256)
        if mode == "channel":
            ref = ref.transpose(-1, -2).contiguous()
        assert oh.shape == ref.shape and np.array_equal(cpu(oh), ref.numpy())
        assert np.array_equal(cpu(m.decode(oh)), cpu(A.MuLaw().to(dev).invert(O.mulaw_encode(x[:, :500]).to(dev))))
    oh = A.OneHot()
    c = torch.randint(0, 256, (2, 4410), generator=g)
    oh.scale_data(c)
    assert oh.n_classes == int(c.max()) + 1 and not oh.needs_scaling
    y = oh(c.to(dev))
    assert np.array_equal(cpu(y), O.onehot(c, oh.n_classes).numpy())
    assert np.array_equal(cpu(oh.invert(y)), c.numpy())
    # argmax tie-breaking: first index of the maximum
    t = torch.tensor([[0, 3, 3, 1], [5, 5, 5, 5], [0, 0, 0, 1]])
    assert np.array_equal(cpu(oh.invert(t.to(dev))), O.onehot_invert(t).numpy())
    # README-style chain: stereo audio -> mulaw -> onehot and back
    comp = (A.MuLaw() + A.OneHot(n_classes=256)).to(dev)
    z = comp(x[:, :1000].to(dev))
    assert z.shape == (4, 1000, 256)
    back = comp.invert(z)
    assert float((back - x[:, :1000].to(dev)).abs().max()) < 0.04


def test_graph_captured_streaming_session_matches_modules(dev):
    """One hipGraph replay per chunk == the eager module chain (OverlapAdd -> RealtimeDGT -> RTPGHI -> OverlapAdd)."""
    from acids_transforms_amd.streaming import StreamingDGTSession
    S, C, n, h = 3, 1024, 1024, 256
    g = torch.Generator().manual_seed(21)
    x = torch.randn(S, 5 * C, generator=g) * 0.1
    sess = StreamingDGTSession(S, C, n, h, device=dev, random_phase_below_tolerance=False, use_graph=True, mel_bands=128)
    assert sess.graph is not None
    mel = A.Magnitude(n_fft=n, n_mels=128, mode=None, contrast="log1p").to(dev)
    oa, oi = A.OverlapAdd(n, h).to(dev), A.OverlapAdd(n, h).to(dev)
    rt = A.RealtimeDGT(n_fft=n, hop_length=h, batch_size=[S]).to(dev)
    for c in range(5):
        xc = x[:, c * C:(c + 1) * C].to(dev)
        y = sess.step(xc).clone()
        fr = oa(xc)
        mag = rt(fr).abs()
        ph = rt.pghi(mag, rt.tolerance, noise=torch.zeros_like(mag))
        from acids_transforms_amd import ops
        frames, rt.hgi_mag_buffer, rt.hgi_phase_buffer = ops.rt_polar_irfft_update(mag, ph, rt.inv_window[:n], n,
                                                                                    rt.hgi_mag_buffer)
        yr = oi.invert(frames)
        assert y.shape == yr.shape == (S, C)
        assert rel_max(cpu(y), cpu(yr)) < 1e-5, c
        assert sess.mel_out.shape == (S, C // h, 128)               # per-frame mel features ride in the same graph
        assert rel_max(cpu(sess.mel_out), cpu(mel(rt(fr)))) < 1e-5, c
    # the resynthesis follows the input (PGHI keeps the magnitudes, re-estimates the phase)
    assert bool(torch.isfinite(y).all()) and float(y.abs().max()) > 1e-3


# ----------------------------------------------------------------------------------------------------------
# hop-sized steps (BASELINE config 5; SURVEY hard part 9: "same output stream as the chunked run")
# ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("key,step", [("1024_256_4096", 256), ("1024_256_1024", 256), ("1024_256_4096", 512),
                                      ("64_16_128", 16)])
def test_overlap_add_per_hop_equals_chunked_golden(golden, dev, key, step):
    """The reference's chunked OverlapAdd goldens (G6), reproduced by feeding the same samples one hop (or two) per
    call: frames bit exact (data movement), overlap-added audio bit-identical to our own chunked run (same
    summation order) and within 1e-5 of the reference's."""
    g = golden("g6_overlap_add")
    n, h, chunk = [int(v) for v in key.split("_")]
    x = T_(g["x_" + key]).to(dev)
    fa, fi = A.OverlapAdd(n, h).to(dev), A.OverlapAdd(n, h).to(dev)          # per-hop stream
    ca, ci = A.OverlapAdd(n, h).to(dev), A.OverlapAdd(n, h).to(dev)          # chunked stream (reference granularity)
    rs = A.RealtimeSTFT(n_fft=n, hop_length=h).to(dev)
    for c in range(3):
        frs, ys = [], []
        for j in range(c * chunk, (c + 1) * chunk, step):
            fr = fa(x[:, j:j + step])
            assert fr.shape == (2, step // h, n)
            frs.append(fr.clone())
            ys.append(fi.invert(rs.invert(rs(fr))))
        fr_all, y_all = torch.cat(frs, -2), torch.cat(ys, -1)
        frc = ca(x[:, c * chunk:(c + 1) * chunk])
        yc = ci.invert(rs.invert(rs(frc)))
        assert np.array_equal(cpu(fr_all), cpu(frc))
        assert np.array_equal(cpu(fa.input_buffer), g["inbuf_%s_%d" % (key, c)])
        if ("frames_%s_%d" % (key, c)) in g:
            assert np.array_equal(cpu(fr_all), g["frames_%s_%d" % (key, c)])
        assert np.array_equal(cpu(y_all), cpu(yc))                            # chunk-invariant, bit for bit
        assert np.array_equal(cpu(fi.output_buffer), cpu(ci.output_buffer))
        assert rel_max(cpu(y_all), g["y_%s_%d" % (key, c)]) < 1e-5
        assert rel_max(cpu(fi.output_buffer), g["outbuf_%s_%d" % (key, c)]) < 1e-5


def _snr_db(y, ref):
    return 10 * np.log10((ref ** 2).sum() / max(((y - ref) ** 2).sum(), 1e-30))


@pytest.mark.parametrize("tag", ["a", "b"])
def test_per_hop_rtpghi_golden(golden, dev, tag):
    """The reference's RealtimeDGT driven one frame per call (G15): the module at n = 1, and the graph-captured
    per-hop session fed the reference's recorded noise, both kept on the reference's trajectory."""
    from acids_transforms_amd import ops
    from acids_transforms_amd.streaming import StreamingDGTSession
    g = golden("g15_rtpghi_per_hop")
    n, h, nsteps, chunk = [int(v) for v in g[tag + "_params"]]
    S, F = 2, n // 2 + 1
    x = T_(g[tag + "_x"]).to(dev)
    rt = A.RealtimeDGT(n_fft=n, hop_length=h, batch_size=[S]).to(dev)
    rt.reset(torch.Size([S]))
    sess = StreamingDGTSession(S, h, n, h, device=dev, random_phase_below_tolerance="external", use_graph=True)
    ys = []
    for j in range(nsteps):
        mag_ref = g["%s_mag_%d" % (tag, j)]
        noise = T_(g["%s_noise_%d" % (tag, j)]).to(dev)
        ref = g["%s_yframes_%d" % (tag, j)]
        # (1) module, one frame per call, on the golden magnitudes
        mag = T_(mag_ref).to(dev)
        ph = rt.pghi(mag, rt.tolerance, noise=noise)
        frames = ops.irfft_frames(None, rt.inv_window[:n], n, mag=mag, phase=ph)
        ops.rt_update_buffers_(mag, ph, rt.hgi_mag_buffer, rt.hgi_phase_buffer)
        assert _snr_db(cpu(frames), ref) > 40.0, (j, _snr_db(cpu(frames), ref))
        assert np.allclose(cpu(rt.hgi_mag_buffer), g["%s_magbuf_%d" % (tag, j)], rtol=1e-5, atol=1e-6)
        # (2) session: audio in, audio out, one hipGraph replay per hop
        sess.noise_in.copy_(noise)
        ys.append(sess.step(x[:, j * h:(j + 1) * h]).clone())
        assert rel_max(cpu(sess.mag_out), mag_ref) < 1e-5
        assert np.allclose(cpu(sess.mag_hist), g["%s_magbuf_%d" % (tag, j)], rtol=1e-5, atol=1e-5 * float(mag_ref.max()))
        dphi = np.angle(np.exp(1j * (cpu(sess.prev_phase) - g["%s_phasebuf_%d" % (tag, j)])))
        big = g["%s_magbuf_%d" % (tag, j)][:, 1] > 1e-2 * g["%s_magbuf_%d" % (tag, j)].max()
        assert np.abs(dphi[big]).max() < 5e-2, j
        # keep both on the reference's trajectory for the next hop
        for hist, prev in ((rt.hgi_mag_buffer, rt.hgi_phase_buffer), (sess.mag_hist, sess.prev_phase)):
            hist.copy_(T_(g["%s_magbuf_%d" % (tag, j)]).to(dev))
            prev.copy_(T_(g["%s_phasebuf_%d" % (tag, j)]).to(dev))
    y = cpu(torch.cat(ys, -1))
    assert y.shape == g[tag + "_y"].shape
    assert _snr_db(y, g[tag + "_y"]) > 40.0, _snr_db(y, g[tag + "_y"])


@pytest.mark.parametrize("C", [256, 1024, 4096])
def test_streaming_session_256_streams(dev, C):
    """BASELINE config 5 as written: 256 concurrent streams, hop-sized / 1024 / 4096-sample steps, bf16 MFMA mel
    features, one hipGraph replay per step -- against the eager module chain on the same input."""
    from acids_transforms_amd import ops
    from acids_transforms_amd.streaming import StreamingDGTSession
    S, n, h = 256, 1024, 256
    steps = 3 if C == 4096 else 6
    g = torch.Generator().manual_seed(77)
    x = (torch.randn(S, steps * C, generator=g) * 0.1).to(dev)
    sess = StreamingDGTSession(S, C, n, h, device=dev, random_phase_below_tolerance=False, use_graph=True, mel_bands=128,
                               mel_dtype="bf16")
    assert sess.graph is not None and sess.n == C // h
    mel32 = A.Magnitude(n_fft=n, n_mels=128, mode=None, contrast="log1p").to(dev)
    oa, oi = A.OverlapAdd(n, h).to(dev), A.OverlapAdd(n, h).to(dev)
    rt = A.RealtimeDGT(n_fft=n, hop_length=h, batch_size=[S]).to(dev)
    rt.reset(torch.Size([S]))
    for c in range(steps):
        xc = x[:, c * C:(c + 1) * C]
        y = sess.step(xc)
        fr = oa(xc)
        X = rt(fr)
        mag = X.abs()
        ph = rt.pghi(mag, rt.tolerance, noise=torch.zeros_like(mag))
        frames = ops.irfft_frames(None, rt.inv_window[:n], n, mag=mag, phase=ph)
        ops.rt_update_buffers_(mag, ph, rt.hgi_mag_buffer, rt.hgi_phase_buffer)
        yr = oi.invert(frames)
        assert y.shape == yr.shape == (S, C)
        assert rel_max(cpu(y), cpu(yr)) < 1e-5, c
        assert np.array_equal(cpu(sess.mag_hist), cpu(rt.hgi_mag_buffer))
        assert sess.mel_out.shape == (S, C // h, 128)
        assert rel_max(cpu(sess.mel_out), cpu(mel32(X))) < 4e-3, c          # bf16 operands: SURVEY hard part 5
    assert bool(torch.isfinite(y).all()) and float(y.abs().max()) > 1e-3


def test_streaming_per_hop_analysis_equals_chunked(dev):
    """Everything chunk-invariant in the step -- magnitudes and mel features of every analysed frame -- is the same
    whether the session is fed one hop or a 1024-sample chunk at a time (fp32 mel: 1e-5 bar, bit-identical here)."""
    from acids_transforms_amd.streaming import StreamingDGTSession
    S, n, h = 5, 1024, 256
    g = torch.Generator().manual_seed(78)
    x = (torch.randn(S, 8 * 1024, generator=g) * 0.1).to(dev)
    a = StreamingDGTSession(S, 256, n, h, device=dev, random_phase_below_tolerance=False, mel_bands=128)
    b = StreamingDGTSession(S, 1024, n, h, device=dev, random_phase_below_tolerance=False, mel_bands=128)
    for c in range(8):
        b.step(x[:, c * 1024:(c + 1) * 1024])
        mags, mels = [], []
        for j in range(4):
            a.step(x[:, c * 1024 + j * 256:c * 1024 + (j + 1) * 256])
            mags.append(a.mag_out.clone())
            mels.append(a.mel_out.clone())
        assert np.array_equal(cpu(torch.cat(mags, 1)), cpu(b.mag_out))
        assert np.array_equal(cpu(torch.cat(mels, 1)), cpu(b.mel_out))
    with pytest.raises(ValueError):
        StreamingDGTSession(S, 300, n, h, device=dev)


def test_streaming_session_draws_its_noise_in_the_kernels(dev):
    """random_phase_below_tolerance=True: the captured step holds no generator launch -- the RTPGHI kernels draw, the
    step advances the counter -- and `step(None)` replays on samples the caller wrote into `x_in`."""
    from acids_transforms_amd.streaming import StreamingDGTSession
    torch.manual_seed(5)
    S, C = 4, 256
    sess = StreamingDGTSession(S, C, device=dev, random_phase_below_tolerance=True, use_graph=True)
    assert sess.rng_state is not None and sess.graph is not None
    c0 = int(sess.rng_state[2])
    tone = 0.5 * torch.sin(2 * torch.pi * 440.0 * torch.arange(8 * C, device=dev) / 44100.0).repeat(S, 1)
    outs = []
    for j in range(8):
        sess.x_in.copy_(tone[:, j * C:(j + 1) * C])
        outs.append(sess.step(None).clone())
    assert int(sess.rng_state[2]) == c0 + 8                 # one counter per replay
    y = torch.cat(outs, -1)
    assert bool(torch.isfinite(y).all()) and float(y.abs().max()) > 1e-3
    # a pure tone leaves most bins under the tolerance: their phases are the draws, different every step
    torch.manual_seed(5)
    again = StreamingDGTSession(S, C, device=dev, random_phase_below_tolerance=True, use_graph=False)
    assert torch.equal(again.rng_state[:2], sess.rng_state[:2])    # torch.manual_seed governs the session's seed


@pytest.mark.parametrize("n", [512, 2048, 400])
def test_streaming_classes_at_other_sizes_golden(golden, dev, n):
    """G17: the reference's OverlapAdd -> RealtimeSTFT / RealtimeDGT -> invert -> OverlapAdd.invert over two chunks at
    n_fft 512 and 2048 (register-core frame kernels) and 400 (mixed-radix kernels): frames bit for bit, spectra,
    synthesised frames and audio at 1e-5."""
    g = golden("g17_streaming_sizes")
    n_, h, chunk = [int(v) for v in g["params_%d" % n]]
    x = torch.from_numpy(g["x_%d" % n]).to(dev)
    oa, oi, od = A.OverlapAdd(n, h).to(dev), A.OverlapAdd(n, h).to(dev), A.OverlapAdd(n, h).to(dev)
    rs, rd = A.RealtimeSTFT(n_fft=n, hop_length=h).to(dev), A.RealtimeDGT(n_fft=n, hop_length=h).to(dev)
    for c in range(2):
        fr = oa(x[:, c * chunk:(c + 1) * chunk])
        assert np.array_equal(fr.cpu().numpy(), g["frames_%d_%d" % (n, c)])
        X = rs(fr)
        assert rel_max(X.cpu().numpy(), g["X_%d_%d" % (n, c)]) < 1e-5
        yf = rs.invert(torch.from_numpy(g["X_%d_%d" % (n, c)]).to(dev))
        assert rel_max(yf.cpu().numpy(), g["yframes_%d_%d" % (n, c)]) < 1e-5
        assert rel_max(oi.invert(torch.from_numpy(g["yframes_%d_%d" % (n, c)]).to(dev)).cpu().numpy(), g["y_%d_%d" % (n, c)]) < 1e-5
        Xd = rd(fr)
        assert rel_max(Xd.cpu().numpy(), g["Xd_%d_%d" % (n, c)]) < 1e-5
        ydf = rd.invert(torch.from_numpy(g["Xd_%d_%d" % (n, c)]).to(dev))
        assert rel_max(ydf.cpu().numpy(), g["ydframes_%d_%d" % (n, c)]) < 1e-5
        assert rel_max(od.invert(torch.from_numpy(g["ydframes_%d_%d" % (n, c)]).to(dev)).cpu().numpy(), g["yd_%d_%d" % (n, c)]) < 1e-5


@pytest.mark.parametrize("tag", ["k512", "k400"])
def test_realtime_pghi_at_other_sizes_golden(golden, dev, tag):
    """G17: RealtimeDGT.pghi on a fixed state at n_fft 512 and 400 (257 / 201 bins per frame), the reference's recorded
    noise fed back in."""
    g = golden("g17_streaming_sizes")
    n, h = [int(v) for v in g[tag + "_params"]]
    rt = A.RealtimeDGT(n_fft=n, hop_length=h, batch_size=[2]).to(dev)
    rt.hgi_mag_buffer = torch.from_numpy(g[tag + "_magbuf"]).to(dev)
    rt.hgi_phase_buffer = torch.from_numpy(g[tag + "_phasebuf"]).to(dev)
    mag = torch.from_numpy(g[tag + "_mag"]).to(dev)
    ph = rt.pghi(mag, rt.tolerance, noise=torch.from_numpy(g[tag + "_noise"]).to(dev)).cpu().numpy()
    ref = g[tag + "_phase"]
    tol = 2e-3 + 16 * np.spacing(np.abs(ref).astype(np.float32)) + 1e-5 * np.abs(ref)
    assert np.all(np.abs(ph - ref) <= tol)
