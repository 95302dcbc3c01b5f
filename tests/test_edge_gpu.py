"""GPU edge cases: empty / ragged / tiny inputs, odd hops, strided and non-fp32 inputs, state round trips."""
import numpy as np
import pytest
import torch

import acids_transforms_amd as A
from acids_transforms_amd import ops
from conftest import rel_max
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def cpu(t):
    return t.detach().cpu().numpy()


def test_empty_batch(dev):
    m = A.STFT().to(dev)
    X = m(torch.zeros(0, 4096, device=dev))
    assert X.shape == (0, 17, 513) and X.dtype == torch.complex64
    assert m.invert(X).shape == (0, 4096)
    mg = A.Magnitude(mode=None).to(dev)
    assert mg(X).shape == (0, 17, 513)
    d = A.DGT().to(dev)
    assert d.invert(torch.zeros(0, 5, 513, device=dev), inversion_mode="pghi").shape == (0, 1024)


@pytest.mark.parametrize("L", [513, 600, 767, 768, 1023, 1025])
def test_shortest_clips(dev, L):
    g = torch.Generator().manual_seed(L)
    x = torch.randn(3, L, generator=g)
    m = A.STFT().to(dev)
    X = m(x.to(dev))
    Xr = O.stft_forward(x, O.hann_window(1024), 1024, 256)
    assert X.shape == Xr.shape and rel_max(cpu(X), Xr.numpy()) < 1e-5
    y = m.invert(X)
    yr = O.istft(Xr, O.hann_window(1024), 1024, 256)
    assert y.shape == yr.shape and rel_max(cpu(y), yr.numpy()) < 1e-5


@pytest.mark.parametrize("n,h", [(1024, 100), (1024, 333), (1024, 64), (512, 100), (1024, 768)])
def test_odd_hops(dev, n, h):
    g = torch.Generator().manual_seed(n + h)
    x = torch.randn(2, 9001, generator=g)
    m = A.STFT(n_fft=n, hop_length=h).to(dev)
    X = m(x.to(dev))
    Xr = O.stft_forward(x, O.hann_window(n), n, h)
    assert X.shape == Xr.shape and rel_max(cpu(X), Xr.numpy()) < 1e-5
    if h <= n // 2:     # torch.istft needs a non-zero window envelope (NOLA)
        y = m.invert(X)
        yr = O.istft(Xr, O.hann_window(n), n, h)
        assert y.shape == yr.shape and rel_max(cpu(y), yr.numpy()) < 1e-5


def test_strided_and_non_fp32_inputs(dev):
    g = torch.Generator().manual_seed(3)
    base = torch.randn(4, 2, 6000, generator=g)
    m = A.STFT().to(dev)
    xs = base.to(dev)[:, 1, ::2]                       # non-contiguous view
    assert not xs.is_contiguous()
    Xr = O.stft_forward(base[:, 1, ::2].contiguous(), O.hann_window(1024), 1024, 256)
    assert rel_max(cpu(m(xs)), Xr.numpy()) < 1e-5
    x16 = base[:, 0].half()                            # widened to fp32 (float64 is refused, not narrowed: test_stft_gpu.py)
    assert rel_max(cpu(m(x16.to(dev))), O.stft_forward(x16.float(), O.hann_window(1024), 1024, 256).numpy()) < 1e-5
    with pytest.raises(A.AcidsHipError, match="float64"):
        m(base[:, 0].double().to(dev))
    Xn = m(base[:, 0].to(dev)).transpose(0, 1)         # non-contiguous complex input to the inverse / magnitude
    y = m.invert(Xn.transpose(0, 1))
    assert y.shape == (4, 5888)
    mg = A.Magnitude(mode=None, mel=False, contrast=None).to(dev)
    real_in = torch.randn(3, 5, 513, generator=g)
    assert np.array_equal(cpu(mg(real_in.to(dev))), real_in.abs().numpy())     # |.| of a real input, like x.abs()


def test_state_dict_round_trip_on_device(dev):
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 4096, generator=g).to(dev)
    comp = (A.DGT() + A.Magnitude(n_mels=64)).to(dev)
    comp.scale_data(x)
    y = comp(x)
    sd = comp.state_dict()
    assert "transforms.0.phase_buffer" in sd and sd["transforms.0.phase_buffer"].shape == (2, 17, 513)
    comp2 = (A.DGT() + A.Magnitude(n_mels=64))
    comp2.load_state_dict({k: v.cpu() for k, v in sd.items()})
    comp2 = comp2.to(dev)
    assert torch.equal(comp2(x), y)
    # keep_input after a reload uses the stored phase
    y1 = comp[0].invert(comp[0](x).abs(), inversion_mode="keep_input")
    assert float((y1 - x).abs().max()) < 0.3 * float(x.abs().max())   # DGT dual-window gain ~1.17, not unity


def test_module_follows_input_device_and_streams(dev):
    m = A.STFT()                      # constructed on the CPU, first device tensor moves it
    x = torch.randn(2, 4096, device=dev)
    X = m(x)
    assert m.window.device == x.device
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):        # kernels follow torch's current stream
        X2 = m(x)
    s.synchronize()
    assert torch.equal(X, X2)


def test_pghi_tiny_and_ragged_batches(dev):
    d = A.DGT(n_fft=128, hop_length=32).to(dev)
    g = torch.Generator().manual_seed(5)
    mag = torch.rand(5, 1, 65, generator=g)               # single-frame clips
    ph = ops.pghi_offline(mag.to(dev), d._hostf("gamma"), 128, 32, d._hostf("tolerance"), d._hostf("eps"))
    for b in range(5):
        r = O.pghi_offline(mag[b], 128, 32)
        assert np.allclose(cpu(ph[b]), r["phase"], atol=1e-3)
    big = torch.rand(1, 300, 65, generator=g)              # one long clip alone in the launch
    r = O.pghi_offline(big[0], 128, 32, want_order=True)
    ph, npops, order = ops.pghi_offline(big.to(dev), d._hostf("gamma"), 128, 32, d._hostf("tolerance"), d._hostf("eps"),
                                        debug=True)
    k = len(r["order"])
    assert int(npops[0]) == k and np.array_equal(cpu(order[0][:k]), r["order"][:, 0] * 65 + r["order"][:, 1])


def test_operands_on_mixed_devices_are_rejected_and_other_device_works(dev):
    """every op runs with its operands' device current (stream + per-device tables), and operands on
    different devices are refused.  The second half needs two devices."""
    from acids_transforms_amd import AcidsHipError, ops
    import acids_transforms_amd as A
    x = torch.randn(2, 4096, device=dev)
    with pytest.raises(AcidsHipError):
        ops.stft_forward(x, torch.hann_window(1024), 1024, 256)          # CPU window: no silent CPU route
    if torch.cuda.device_count() < 2:
        pytest.skip("one device: the cross-device half needs two")
    d1 = torch.device("cuda:1")
    with pytest.raises(AcidsHipError):
        ops.stft_forward(x, torch.hann_window(1024, device=d1), 1024, 256)
    s0, s1 = A.STFT().to(dev), A.STFT().to(d1)
    assert torch.cuda.current_device() == 0
    X1 = s1(x.to(d1))                                                     # current device is cuda:0
    assert X1.device == d1 and torch.cuda.current_device() == 0
    assert torch.equal(X1.cpu(), s0(x).cpu())
    assert torch.equal(s1.invert(X1).cpu(), s0.invert(s0(x)).cpu())


def test_empty_and_single_row_inputs_of_the_round_two_paths(dev):
    """Empty batches and single frames through the entry points added in round 2: the one-kernel MelSpectrogram at
    n_fft 2048 / 512, Cartesian / PolarIF in place, the long-row banded walk, the mixed-radix sizes, perform_hgi."""
    for n, h in ((2048, 512), (512, 128)):
        mf = A.MFCC(n_fft=n, hop_length=h, n_mels=64).to(dev)
        y = mf(torch.zeros(0, 4 * n, device=dev))
        assert y.shape == (0, 64, 1 + 4 * n // h)
        one = mf(torch.randn(1, n // 2 + 1, device=dev))          # shortest clip torch.stft accepts: three frames at hop n/4
        assert one.shape[0] == 1 and one.shape[1] == 64 and bool(torch.isfinite(one).all())
    X0 = torch.zeros(0, 7, 1025, dtype=torch.complex64, device=dev)
    X1 = torch.randn(1, 1, 1025, dtype=torch.complex64, device=dev)
    car = A.Cartesian(real_args={"mode": None}, imag_args={"mode": None}).to(dev)
    assert car(X0).shape == (0, 7, 2, 1025) and car.invert(car(X0)).shape == (0, 7, 1025)
    assert torch.equal(car.invert(car(X1)), X1)
    mg = A.Magnitude(n_fft=2048, mode=None).to(dev)
    assert mg(X0).shape == (0, 7, 1025) and mg.invert(mg(X0)).shape == (0, 7, 1025)
    assert mg(X1).shape == (1, 1, 1025)
    pif = A.PolarIF(magnitude_args={"mode": None, "n_fft": 2048}, phase_args={"mode": None}).to(dev)
    assert pif(X0).shape == (0, 7, 2, 1025)
    y1 = pif(X1)                                                   # a single frame: forward differences of nothing
    assert y1.shape == (1, 1, 2, 1025) and pif.invert(y1).shape == (1, 1, 1025)
    st = A.STFT(n_fft=400, hop_length=160).to(dev)
    Xe = st(torch.zeros(0, 1600, device=dev))
    assert Xe.shape == (0, 11, 201) and st.invert(Xe).shape == (0, 1600)
    d = A.DGT(n_fft=128, hop_length=32).to(dev)
    z = torch.zeros(0, 5, 65, device=dev)
    assert d.perform_hgi(z, z, z).shape == (0, 5, 65)
    m1 = torch.rand(1, 65, device=dev) + 0.1
    assert d.perform_hgi(m1, torch.zeros_like(m1), torch.zeros_like(m1)).shape == (1, 65)
