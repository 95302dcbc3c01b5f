"""GPU parity: "sinebank" inversion (at_sinebank_offline / at_sinebank_realtime) against the reference's outputs
(tests/golden/g13_sinebank.npz, random phases recorded) and the oracle.  The oscillator phase is formed in fp32
exactly as the reference does; what differs is the summation order over bins (MFMA contraction vs torch.sum)
and sin()'s last bit -> 1e-5 of the output's largest magnitude (the output is peak-normalised to 1)."""
import numpy as np
import pytest
import torch

import acids_transforms_amd as A
from conftest import rel_max
from oracle import oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-5
T_ = torch.from_numpy


def cpu(t):
    return t.detach().cpu()


def test_offline_golden(golden, dev):
    g = golden("g13_sinebank")
    s = A.STFT(n_fft=128, hop_length=32).to(dev)
    y = s.get_sinebank_inversion(T_(g["mag"]).to(dev), random_phase=T_(g["offline_phase"]))
    assert y.shape == g["offline"].shape
    assert rel_max(cpu(y).numpy(), g["offline"]) < TOL
    s1k = A.STFT().to(dev)
    y = s1k.get_sinebank_inversion(T_(g["mag1k"]).to(dev), random_phase=T_(g["offline1k_phase"]))
    assert rel_max(cpu(y).numpy(), g["offline1k"]) < TOL
    # through invert(): same CPU generator stream as the reference -> same phases after the same seed
    torch.manual_seed(77)
    want_phase = 2 * torch.pi * torch.rand(65, 1)
    torch.manual_seed(77)
    d = A.DGT(n_fft=128, hop_length=32).to(dev)
    y = d.invert(T_(g["mag"]).to(dev), inversion_mode="sinebank")
    want = O.sinebank_offline(T_(g["mag"]), 44100, 128, 32, want_phase)
    assert rel_max(cpu(y).numpy(), want.numpy()) < TOL
    assert "sinebank" in s.get_inversion_modes() and "sinebank" in A.DGT.get_inversion_modes()


def test_realtime_golden(golden, dev):
    g = golden("g13_sinebank")
    chunks = T_(g["chunks"]).to(dev)
    for name, cls in (("rtstft", A.RealtimeSTFT), ("rtdgt", A.RealtimeDGT)):
        r = cls(n_fft=128, hop_length=32).to(dev)
        r.random_phase = T_(g[name + "_phase"]).to(dev)        # batch-shaped already: no re-draw
        for i in range(3):
            y = r.get_sinebank_inversion(chunks[i])
            assert y.shape == (3, 4, 128)
            assert rel_max(cpu(y).numpy(), g[name][i]) < TOL, (name, i)
        assert abs(float(r.time_index) - float(g[name + "_time"])) < 1e-6
    r = A.RealtimeSTFT(n_fft=128, hop_length=32).to(dev)
    r.random_phase = T_(g["rt_unbatched_phase"]).to(dev)
    for i in range(2):
        y = r.get_sinebank_inversion(chunks[i, 0])
        assert rel_max(cpu(y).numpy(), g["rt_unbatched"][i]) < TOL
    # invert(mode="sinebank") hands OverlapAdd the frames TIMES the synthesis window (stft.py:303-304, dgt.py:321-322)
    for name, cls in (("rtstft_invert", A.RealtimeSTFT), ("rtdgt_invert", A.RealtimeDGT)):
        ri = cls(n_fft=128, hop_length=32).to(dev)
        ri.random_phase = T_(g[name + "_phase"]).to(dev)
        assert np.allclose(cpu(ri.inv_window[:128]).numpy(), g[name + "_inv_window"], rtol=1e-6, atol=1e-7)
        for i in range(2):
            y = ri.invert(chunks[i], inversion_mode="sinebank")
            assert rel_max(cpu(y).numpy(), g[name][i]) < TOL, (name, i)
    r.reset()
    y0 = r.get_sinebank_inversion(chunks[0, 0])
    assert rel_max(cpu(y0).numpy(), g["rt_unbatched"][0]) < TOL          # the clock restarts


def test_offline_against_oracle_shapes(dev):
    """Default geometry at a size the oracle still handles, ragged batch shapes, hop < 128 (more frames per block)."""
    gen = torch.Generator().manual_seed(9)
    for (shape, n_fft, hop) in [((3, 40, 513), 1024, 256), ((2, 2, 7, 513), 1024, 256), ((1, 1, 513), 1024, 256),
                                ((4, 33, 129), 256, 64), ((2, 50, 33), 64, 16)]:
        mag = torch.rand(*shape, generator=gen) ** 2
        F = shape[-1]
        phase = 2 * torch.pi * torch.rand(F, 1, generator=gen)
        s = A.STFT(n_fft=n_fft, hop_length=hop).to(dev)
        y = s.get_sinebank_inversion(mag.to(dev), random_phase=phase)
        want = O.sinebank_offline(mag.reshape((-1,) + shape[-2:]), 44100, n_fft, hop, phase)
        assert y.shape == shape[:-2] + (hop * shape[-2] + n_fft,)
        assert rel_max(cpu(y).reshape(want.shape).numpy(), want.numpy()) < TOL, (shape, n_fft, hop)
        assert abs(float(y.max()) - 1.0) < 1e-6                       # peak-normalised


def test_full_size_properties(dev):
    """BASELINE clip length (4 s -> 690 frames), 64 clips: properties that do not need the oracle at this size --
    linear in the magnitudes up to the two normalisations, and the clip of a constant single-bin spectrum is that
    bin's sinusoid with a constant envelope."""
    T, F = 690, 513
    s = A.STFT().to(dev)
    gen = torch.Generator().manual_seed(10)
    phase = 2 * torch.pi * torch.rand(F, 1, generator=gen)
    mag = torch.zeros(64, T, F)
    mag[:, :, 40] = 1.0                     # 40 * 44100 / 1024 = 1722.66 Hz
    mag[1:] += 0.0
    y = cpu(s.get_sinebank_inversion(mag.to(dev), random_phase=phase))
    L = 256 * T + 1024
    assert y.shape == (64, L)
    t = torch.linspace(0, L / 44100, L)
    arg = (2 * torch.pi * torch.linspace(0, 22050, F)[40]) * t + phase[40, 0]
    want = torch.sin(arg)
    want = want / want.max()
    assert float((y[0] - want).abs().max()) < 1e-5 and torch.equal(y[0], y[63])
    # scaling the input leaves the output unchanged (x / max|x| first)
    m2 = torch.rand(8, T, F, generator=gen)
    a = s.get_sinebank_inversion(m2.to(dev), random_phase=phase)
    b = s.get_sinebank_inversion((m2 * 3.0).to(dev), random_phase=phase)
    assert float((a - b).abs().max()) < 2e-6
