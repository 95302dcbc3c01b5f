"""GPU: audio front end -- sinc resampling (at_resample_sinc) against the oracle's restatement of torchaudio's
published algorithm (parity unpinned: torchaudio is not in the reference tree), plus properties, and
`import_data` on wav files written by the test itself."""
import math
import os

import numpy as np
import pytest
import torch

from acids_transforms_amd.utils.audio_io import Resample, import_data, load_wav
from conftest import rel_max
from oracle import oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-5


def test_resample_against_oracle(dev):
    gen = torch.Generator().manual_seed(3)
    for (orig, new, shape) in [(22050, 44100, (2, 5000)), (44100, 22050, (3, 2, 4001)), (48000, 44100, (1, 9000)),
                               (44100, 48000, (7777,)), (8000, 44100, (2, 1000)), (44100, 44100, (2, 100)),
                               (3, 2, (1, 1))]:
        x = torch.randn(*shape, generator=gen)
        want = O.resample(x, orig, new)
        got = Resample(orig, new)(x.to(dev)).cpu()
        assert got.shape == want.shape == x.shape[:-1] + (math.ceil(new // math.gcd(orig, new) * shape[-1]
                                                                     / (orig // math.gcd(orig, new))),)
        if want.numel():
            assert rel_max(got.numpy(), want.numpy()) < TOL, (orig, new)


def test_resample_preserves_a_band_limited_tone(dev):
    sr0, sr1, f = 22050, 44100, 1000.0
    n = torch.arange(22050, dtype=torch.float64)
    x = torch.sin(2 * math.pi * f * n / sr0).float()
    y = Resample(sr0, sr1)(x.to(dev)[None])[0].cpu()
    m = torch.arange(y.numel(), dtype=torch.float64)
    want = torch.sin(2 * math.pi * f * m / sr1).float()
    assert y.numel() == 44100
    assert float((y - want)[200:-200].abs().max()) < 2e-3          # away from the zero-padded edges


def test_import_data_files_and_folder(tmp_path, dev):
    from scipy.io import wavfile
    rng = np.random.default_rng(0)
    mono = (rng.standard_normal(3000) * 0.2).astype(np.float32)
    stereo16 = (rng.standard_normal((5000, 2)) * 8000).astype(np.int16)
    low = (rng.standard_normal((1000, 2)) * 0.1).astype(np.float32)
    wavfile.write(str(tmp_path / "a_mono.wav"), 44100, mono)
    wavfile.write(str(tmp_path / "b_stereo.wav"), 44100, stereo16)
    wavfile.write(str(tmp_path / "c_low.wav"), 22050, low)
    (tmp_path / "notes.txt").write_text("not audio")
    x, name = import_data(str(tmp_path / "b_stereo.wav"))
    assert name == "b_stereo.wav" and x.shape == (2, 5000) and x.device.type == "cpu"
    assert torch.equal(x, torch.from_numpy(stereo16.T.astype(np.float32) / 32768.0))
    w, sr = load_wav(str(tmp_path / "a_mono.wav"))
    assert sr == 44100 and torch.equal(w, torch.from_numpy(mono)[None])
    x, _ = import_data(str(tmp_path / "c_low.wav"), sr=44100)      # resampled on the GPU, returned on the CPU
    assert x.shape == (2, 2000) and x.device.type == "cpu"
    assert rel_max(x.numpy(), O.resample(torch.from_numpy(low.T.copy()), 22050, 44100).numpy()) < TOL
    data, names = import_data(str(tmp_path))
    assert sorted(names) == ["a_mono", "b_stereo", "c_low"] and data.shape == (3, 2, 5000)   # txt skipped, all stereo
    i = names.index("a_mono")
    assert torch.equal(data[i, 0, :3000], torch.from_numpy(mono)) and torch.equal(data[i, 0], data[i, 1])
    assert float(data[i, :, 3000:].abs().max()) == 0.0
    with pytest.raises(FileNotFoundError):
        import_data(str(tmp_path / "missing"))
    os.remove(str(tmp_path / "b_stereo.wav"))
    os.remove(str(tmp_path / "c_low.wav"))
    data, names = import_data(str(tmp_path))
    assert data.shape == (1, 1, 3000)                                # mono-only folder stays mono
