"""CPU-only checks: host-side logic of the drop-in modules against the reference's
constants (goldens), the C ABI surface, and the product/oracle separation."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import acids_transforms_amd as A
from acids_transforms_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_windows_gammas_gain_match_reference(golden):
    g = golden("g1_constants")
    for (n, h) in [(1024, 256), (512, 128), (128, 32), (32, 8), (2048, 512), (64, 16)]:
        k = "%d_%d" % (n, h)
        s, d, r = A.STFT(n_fft=n, hop_length=h), A.DGT(n_fft=n, hop_length=h), A.RealtimeDGT(n_fft=n, hop_length=h)
        assert np.array_equal(s.window[:n].numpy(), g["hann_" + k]) and float(s.window[n:].abs().sum()) == 0
        assert np.array_equal(s.inv_window[:n].numpy(), g["hann_" + k])
        # exp() may differ by an ulp between host CPUs (vectorised libm variants)
        assert np.allclose(d.window[:n].numpy(), g["gauss_" + k], rtol=3e-7, atol=0)
        assert np.allclose(d.inv_window[:n].numpy(), g["dual_" + k], rtol=1e-6, atol=0)
        assert np.array_equal(s.gamma.numpy(), g["gamma_stft_" + k])
        assert np.array_equal(d.gamma.numpy(), g["gamma_dgt_" + k])
        assert np.array_equal(r.gamma.numpy(), g["gamma_rt_" + k])
        assert float(A.OverlapAdd(n, h).gain_compensation) == float(g["oadd_gain_" + k])
        assert s.n_fft.shape == (1,) and s.n_fft.dtype == torch.int64 and int(s.hop_length) == h
    assert np.float32(A.DGT().eps) == g["eps"] and np.float32(A.DGT().tolerance) == g["tolerance"]


def test_state_dict_keys_match_reference(golden):
    g = golden("g1_constants")
    assert sorted(A.STFT().state_dict().keys()) == list(g["stft_state_keys"])
    assert sorted(A.DGT().state_dict().keys()) == list(g["dgt_state_keys"])
    assert sorted(A.RealtimeDGT().state_dict().keys()) == list(g["rtdgt_state_keys"])
    m = A.Magnitude()
    assert sorted(m.state_dict().keys()) == ["eps", "inverse_mel_bank", "mel_bank", "norm.offset", "norm.scale"]
    assert m.mel_bank.shape == (1, 513, 513) and m.inverse_mel_bank.shape == (1, 513, 513)
    d2 = A.DGT(n_fft=512, hop_length=128)
    d2.load_state_dict(A.DGT().state_dict())             # a saved state re-targets n_fft / hop
    assert d2.ratio == 256 and d2._n_fft == 1024
    rt = A.RealtimeDGT(batch_size=[3, 2])
    assert rt.hgi_mag_buffer.shape == (3, 2, 2, 513) and rt.hgi_phase_buffer.shape == (3, 2, 513)
    rt.reset([5])
    assert rt.hgi_mag_buffer.shape == (5, 2, 513) and rt.get_batch_size() == [5]


def test_api_surface_and_errors():
    s = A.STFT()
    assert s.get_inversion_modes() == ["griffin_lim", "keep_input", "random", "sinebank"]
    assert A.DGT.get_inversion_modes() == ["pghi", "griffin_lim", "random", "keep_input", "sinebank"]
    assert A.RealtimeDGT.get_inversion_modes() == ["random", "pghi", "keep_input", "sinebank"]
    assert A.RealtimeSTFT.get_inversion_modes() == ["keep_input", "random", "sinebank"]
    assert s.ratio == 256 and s.invertible and not s.needs_scaling and s.inversion_mode == "griffin_lim"
    assert A.DGT().inversion_mode == "pghi" and isinstance(A.DGT().realtime(), A.RealtimeDGT)
    assert isinstance(s.realtime(), A.RealtimeSTFT) and s.realtime().inversion_mode == "random"
    with pytest.raises(ValueError):
        A.STFT(window="nope")
    with pytest.raises(ValueError):
        A.STFT(inversion_mode="nope")
    with pytest.raises(AssertionError):
        A.STFT(n_fft=1024, hop_length=None)
    with pytest.raises(AttributeError):
        s.set_inversion_mode("nope")
    s.set_inversion_mode("random")
    s.set_params(512, 128)
    assert s.ratio == 128 and float(s.window[512:].abs().sum()) == 0
    with pytest.raises(TypeError):
        s + 1
    c = A.STFT() + A.Magnitude()
    assert isinstance(c, A.ComposeAudioTransform) and len(c) == 2 and c.ratio == 256 and c.needs_scaling and c.invertible
    c3 = A.OverlapAdd() + c
    assert len(c3) == 3 and isinstance(c3[0], A.OverlapAdd)
    assert not (A.STFT() + A.MFCC()).invertible
    with pytest.raises(A.NotInvertibleError):
        A.MFCC().invert(torch.zeros(1))
    with pytest.raises(TypeError):
        A.Magnitude(contrast="nope").contrast(torch.zeros(1))
    rc = c.realtime()
    assert isinstance(rc[0], A.RealtimeSTFT)
    assert A.OverlapAdd().ratio == 1 and int(A.OverlapAdd().hop_length) == 128     # default hop is 128 (oadd.py:23)


def test_no_cpu_fallback():
    for mod, x in [(A.STFT(), torch.zeros(2, 4096)), (A.DGT(), torch.zeros(2, 4096)),
                   (A.Magnitude(mode=None), torch.zeros(2, 5, 513, dtype=torch.complex64)),
                   (A.MFCC(), torch.zeros(2, 4096)), (A.OverlapAdd(1024, 256), torch.zeros(2, 4096)),
                   (A.MuLaw(), torch.zeros(2, 100)), (A.RealtimeDGT(), torch.zeros(2, 3, 1024))]:
        with pytest.raises(A.AcidsHipError):
            mod(x)
    with pytest.raises(A.AcidsHipError):
        A.DGT().invert(torch.zeros(2, 5, 513), inversion_mode="pghi")


def test_c_abi_exports_every_declared_symbol():
    """The .so loads and exports exactly what include/acids_hip.h declares (no compute without a GPU)."""
    hdr = open(os.path.join(ROOT, "include", "acids_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(at_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 25
    so = os.path.join(ROOT, "acids_transforms_amd", "libacids_hip.so")
    assert os.path.exists(so), "build the library first (python -c 'import __graft_entry__ as g; g.build()')"
    L = ctypes.CDLL(so)
    for name in declared:
        assert hasattr(L, name), "missing export: " + name
    assert sorted(_lib.exported_symbols()) == declared            # the Python binding covers the whole ABI
    lib = _lib.lib()
    assert lib.at_abi_version() == _lib.ABI_VERSION == 4
    assert lib.at_error_string(-2).decode() == "unsupported configuration"
    assert lib.at_istft_workspace_bytes(4, 10, 1024, 256) == 0
    assert lib.at_istft_workspace_bytes(4, 10, 512, 128) == 0 and lib.at_istft_workspace_bytes(4, 10, 2048, 512) == 0
    assert lib.at_istft_workspace_bytes(4, 10, 256, 64) == 4 * 10 * 256 * 4
    assert lib.at_pghi_offline_workspace_bytes(2, 10, 513) >= 2 * (3 * 5130 * 4 + 5132 * 8)


def test_library_reads_no_environment_and_variants_are_explicit():
    """the shipped library takes no decisions from environment variables -- no getenv in its sources
    outside the -DAT_DEV_SWITCHES accessor, none among its dynamic imports -- and the kernel variants the parity tests
    need are an explicit, validated C-ABI table."""
    import subprocess
    csrc = os.path.join(ROOT, "acids_transforms_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h")) and f != "variants.h":
            assert "getenv" not in open(os.path.join(csrc, f)).read(), f
    v = open(os.path.join(csrc, "variants.h")).read()
    assert v.count("getenv(") == 2 and "#ifdef AT_DEV_SWITCHES" in v      # one use, one mention, both in the dev branch
    so = os.path.join(ROOT, "acids_transforms_amd", "libacids_hip.so")
    undefined = subprocess.run(["nm", "-D", "--undefined-only", so], capture_output=True, text=True).stdout
    assert "getenv" not in undefined
    lib = _lib.lib()
    for which in _lib.VARIANTS.values():
        assert lib.at_get_variant(which) == 0
    assert lib.at_set_variant(99, 1) == -1 and lib.at_set_variant(0, 7) == -1 and lib.at_get_variant(-1) == -1
    with _lib.variant("pghi_kernel", 2):
        assert lib.at_get_variant(_lib.VARIANTS["pghi_kernel"]) == 2
        with _lib.variant("epilogue", 1):
            assert lib.at_get_variant(_lib.VARIANTS["epilogue"]) == 1
    assert all(lib.at_get_variant(w) == 0 for w in _lib.VARIANTS.values())


def test_product_never_touches_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use oracle/."""
    pkg = os.path.join(ROOT, "acids_transforms_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in src.lower().replace("# oracle", ""), os.path.join(dirpath, f)
    # and at run time: importing the whole package (every transform, ops, streaming, dist) in a fresh interpreter
    # must not pull in anything from oracle/
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import acids_transforms_amd, acids_transforms_amd.streaming, "
            "acids_transforms_amd.dist, acids_transforms_amd.ops; "
            "bad = [m for m in sys.modules if m == 'oracle' or m.startswith('oracle.')]; "
            "print('LOADED', bad); sys.exit(1 if bad else 0)" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr


def test_melbank_properties():
    from acids_transforms_amd.utils.melbank import melscale_fbanks
    fb = melscale_fbanks(513, 0.0, 22050.0, 513, 44100)
    assert int((fb.sum(0) == 0).sum()) == 109 and int((fb != 0).sum()) == 1019
    fb128 = melscale_fbanks(513, 0.0, 22050.0, 128, 44100)
    assert fb128.shape == (513, 128) and float(fb128.min()) >= 0 and float(fb128.max()) <= 1.0
    peaks = fb128[:, 5:].argmax(0)
    assert bool((peaks[1:] >= peaks[:-1]).all())          # centre frequencies increase


def test_banded_bank_tables_reproduce_dense_bank_and_are_conflict_free():
    """utils/banded.py: the walk tables handed to at_stft_mel_forward must (a) reproduce the dense contraction
    exactly when replayed on the CPU and (b) keep the ds_read_b128 magnitude reads of the standard mel banks
    free of LDS bank conflicts under the documented banking model."""
    from acids_transforms_amd.utils.banded import BandedBank
    from acids_transforms_amd.utils.melbank import melscale_fbanks
    rng = np.random.default_rng(0)
    for n_mels in (40, 64, 128, 256):
        fb = melscale_fbanks(513, 0., 22050., n_mels, 44100)
        band = BandedBank(fb)
        assert band.eligible and band.n_passes == (n_mels + 63) // 64
        lane_filter, lane_start, weights = band._host
        assert weights.size == 64 * int(band.pass_len.sum()) <= 8192
        assert (lane_start % 4 == 0).all() and int(lane_start.max()) + int(band.pass_len.max()) <= 640
        assert sorted(int(f) for f in lane_filter if f >= 0) == list(range(n_mels))
        mag = np.zeros(640, np.float32)
        mag[:513] = rng.random(513).astype(np.float32)
        feat = np.zeros(n_mels, np.float64)
        base = 0
        for q in range(band.n_passes):
            steps = int(band.pass_len[q]) // 4
            w = weights[base * 256:(base + steps) * 256].reshape(steps, 64, 4)
            for lane in range(64):
                f = lane_filter[q * 64 + lane]
                if f >= 0:
                    s0 = lane_start[q * 64 + lane]
                    feat[f] = float((mag[s0:s0 + 4 * steps].astype(np.float64) * w[:, lane, :].reshape(-1)).sum())
            base += steps
        want = mag[:513].astype(np.float64) @ fb.double().numpy()
        assert np.allclose(feat, want, rtol=1e-12, atol=1e-12)
        cycles, ideal = band.lds_read_cycles()
        assert cycles <= ideal + 2, (n_mels, cycles, ideal)
    dense = BandedBank(torch.rand(513, 32))
    assert not dense.eligible                                  # not banded: the MFMA projection is used instead


def test_channel_stage_matches_reference(golden):
    """Mono / Stereo / MidSide / Window / Squeeze / Unsqueeze / Transpose against the reference's outputs (G12).
    Layout-only transforms: the same check runs on device tensors in tests/test_chain_gpu.py."""
    from chan_check import check_channel_stage
    check_channel_stage(golden("g12_channels"), "cpu")


def test_periodic_derivative_helpers():
    """utils.misc.deriv / get_fft_idx (reference utils/misc.py:107-135): derivative of sin(2 pi k t) on [0, 1)."""
    import math
    from acids_transforms_amd.utils.misc import deriv, get_fft_idx
    assert get_fft_idx(6).tolist() == [0, 1, 2, 3, -2, -1] and get_fft_idx(5).tolist() == [0, 1, 2, -2, -1]
    L, k = 256, 3
    t = torch.arange(L, dtype=torch.float64) / L
    x = torch.sin(2 * math.pi * k * t).unsqueeze(1)
    want = 2 * math.pi * k * torch.cos(2 * math.pi * k * t).unsqueeze(1)
    assert float((deriv(x, float("inf")) - want).abs().max()) < 1e-9
    assert float((deriv(x, 4) - want).abs().max()) < 2e-4 * float(want.abs().max())
    assert float((deriv(x, 2) - want).abs().max()) < 2e-3 * float(want.abs().max())


def test_package_surface_follows_the_reference():
    """what `from acids_transforms import *` gives a user (the reference re-exports utils and
    transforms at package level, __init__.py:1-2, utils/__init__.py:1-2) exists here under the same names -- the
    helpers included: `heappush` / `heappop` (utils/heapq.py), `format_input_data` (utils/misc.py:61-63, which returns
    None there and here), and `Dummy` where the reference keeps it (spectral_repr.py:17)."""
    names = ["unwrap", "import_data", "format_input_data", "fdiff_forward", "fdiff_backward", "fdiff_central", "fint_forward",
             "fint_backward", "fint_central", "deriv", "get_fft_idx", "pad", "frame", "reshape_batches", "heappush",
             "heappop", "AudioTransform", "ComposeAudioTransform", "NotInvertibleError", "Mono", "Stereo", "MidSide",
             "Window", "MuLaw", "STFT", "RealtimeSTFT", "DGT", "RealtimeDGT", "Normalize", "Real", "Imaginary", "Magnitude",
             "Phase", "IF", "Cartesian", "Polar", "PolarIF", "MFCC", "OneHot", "Squeeze", "Unsqueeze", "Transpose", "OverlapAdd"]
    missing = [n for n in names if not hasattr(A, n)]
    assert not missing, missing
    from acids_transforms_amd.transforms import spectral_repr
    assert issubclass(spectral_repr.Dummy, A.AudioTransform)
    assert isinstance(A.Magnitude(mode=None).norm, spectral_repr.Dummy)
    assert A.format_input_data(torch.zeros(2, 3, 4), dim=-1) is None


def test_host_heap_reproduces_the_recorded_pop_order(golden):
    """`heappush` / `heappop` with the reference's ordering rule (strict `<` on the key, right child on ties): the flood of
    dgt.py:168-220 written out on the host with these two functions pops the bins of the golden cases -- noise, the
    piecewise-constant TIE case, a sparse one with reseeds -- in exactly the order recorded from the reference."""
    g = golden("g4_pghi_offline")
    eps = np.float32(torch.finfo(torch.float32).eps)
    for case in ("n12x17", "t12x17", "s12x17", "t40x65", "d40x65"):
        mag = np.maximum(g[case + "_mag"].astype(np.float32), eps)
        tol = np.float32(g[case + "_params"][2])
        T, F = mag.shape
        work = mag.copy()
        order = []
        heap = []
        first = True
        while True:
            mx = work.max()
            if first:
                work[work < mx * tol] = eps                 # dgt.py:177-178, once, relative to the global maximum
                first = False
            if not (mx > eps):
                break
            t, f = np.argwhere(work == mx)[0]
            work[t, f] = eps
            A.heappush(heap, (-mx, (int(t), int(f))))
            while heap:
                _, (t, f) = A.heappop(heap)
                order.append((t, f))
                for (tt, ff) in ((t + 1, f), (t - 1, f), (t, f + 1), (t, f - 1)):
                    if 0 <= tt < T and 0 <= ff < F and work[tt, ff] > eps:
                        A.heappush(heap, (-work[tt, ff], (tt, ff)))
                        work[tt, ff] = eps
        assert np.array_equal(np.array(order).reshape(-1, 2), g[case + "_order"]), case
    h = []
    with pytest.raises(IndexError):
        A.heappop(h)


def test_double_precision_is_refused_not_narrowed():
    """the reference's `dtype=torch.float64` keeps double buffers (stft.py:36-47); there is no fp64
    kernel here, and computing in fp32 under that name would be a silent narrowing -- the constructors say so."""
    for ctor in (A.STFT, A.DGT, A.RealtimeSTFT, A.RealtimeDGT, A.Magnitude):
        with pytest.raises(A.AcidsHipError, match="float32"):
            ctor(dtype=torch.float64)
        ctor(dtype=torch.float32)
