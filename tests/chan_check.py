"""Shared body of the channel-stage parity check (golden G12, outputs of the reference's raw.py:11-262 / misc.py
classes): run on CPU tensors by tests/test_host_cpu.py and on device tensors by tests/test_chain_gpu.py."""
import pytest
import torch

import acids_transforms_amd as A


def check_channel_stage(g, device):
    dev = torch.device(device)
    T_ = lambda k: torch.from_numpy(g[k]).to(dev)  # noqa: E731

    def same(a, b):
        """Bit-equal on the CPU.  On the device torch itself divides by a Python scalar as a multiplication by its
        reciprocal (`x / sqrt(2)` in MidSide, `x / x.max()`: one ulp from the CPU quotient) -- the reference's own code
        moved to the device would differ from its CPU output in the same way, so two ulps are allowed there."""
        if a.device.type != dev.type or a.shape != b.shape:
            return False
        if dev.type == "cpu" or not a.is_floating_point():
            return torch.equal(a, b)
        # sums of such values (MidSide.invert: mid * sqrt(2) +- side) cancel: a few ulps of the LARGEST value
        return torch.allclose(a, b, rtol=2.4e-7, atol=4e-7 * float(b.abs().max()) if b.numel() else 0.0)
    st, mo, one = T_("st"), T_("mo"), T_("one")
    times = torch.arange(6., device=dev).reshape(3, 2)
    for mode in ("mix", "left", "right"):
        for squeeze in (True, False):
            for inv in ("mono", "stereo"):
                m = A.Mono(mode=mode, squeeze=squeeze, inversion_mode=inv, normalize=(mode == "left"))
                key = "mono_%s_%d_%s" % (mode, int(squeeze), inv)
                y = m(st)
                assert same(y, T_(key)), key
                assert same(m.invert(y), T_(key + "_inv")), key
                assert same(m(mo), T_(key + "_m")), key
    assert [t.shape for t in A.Mono()([st, mo])] == [(3, 100), (3, 100)]
    _, tm = A.Mono().forward_with_time(st, times)
    assert same(tm, T_("mono_time"))
    for name, t in (("stereo", A.Stereo()), ("stereo_n", A.Stereo(normalize=True)), ("midside", A.MidSide()),
                    ("midside_np", A.MidSide(pad_mid=False, normalize=True))):
        for tag, x in (("st", st), ("mo", mo), ("one", one)):
            y = t(x)
            assert same(y, T_("%s_%s" % (name, tag))), (name, tag)
            assert same(t.invert(y), T_("%s_%s_inv" % (name, tag))), (name, tag)
    g3 = torch.Generator().manual_seed(123)
    x3 = torch.randn(2, 3, 10, generator=g3).to(dev)
    assert same(A.Stereo().invert(x3), T_("stereo_inv3"))
    with pytest.raises(Exception):
        A.Stereo()(x3)
    with pytest.raises(Exception):
        A.MidSide()(x3)
    for ws, hs in ((16, 4), (8, 8), (10, 5)):
        w = A.Window(window_size=ws, hop_size=hs)
        key = "window_%d_%d" % (ws, hs)
        y = w(st)
        assert same(y, T_(key)) and w.ratio == hs
        assert same(w.invert(y), T_(key + "_inv")), key
        _, tm = w.forward_with_time(st, times)
        assert torch.allclose(tm, T_(key + "_time"))
    z = lambda *shape: torch.zeros(*shape, device=dev)  # noqa: E731
    assert A.Squeeze()(z(2, 1, 5, 1)).shape == g["squeeze"].shape
    assert A.Squeeze(dim=1)(z(2, 1, 5, 1)).shape == g["squeeze1"].shape
    assert A.Unsqueeze()(z(2, 5)).shape == g["unsqueeze"].shape
    assert A.Unsqueeze().invert(A.Unsqueeze()(z(2, 5))).shape == (2, 5)
    assert same(A.Transpose()(st), T_("transpose")) and A.Transpose()(st).is_contiguous()
    with pytest.raises(A.NotInvertibleError):
        A.Squeeze().invert(st)
    assert not A.Squeeze().invertible and A.Squeeze(dim=1).invertible
