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
        A.OverlapAdd(1024, 256).to(dev)(torch.zeros(2, 256, device=dev))    # hop-sized chunks are not a valid stream


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
        ph = rt.pghi(mag, noise=torch.zeros_like(mag))
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
