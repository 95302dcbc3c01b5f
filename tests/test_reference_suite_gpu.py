"""GPU: the reference's own test file (test/test_transforms.py) replayed against this package.

The reference instantiates EVERY transform class with its defaults and drives it through the class's self-test hooks:
`test_forward(raw)`, `test_forward(raw, time)` (:28-35), `realtime().test_forward(raw, time)` (:37-42),
`test_inversion(raw)` for the invertible ones (:44-60), the scripted-module scenario (:62-68: the HIP modules are not
scriptable, the scenario itself runs), and four composed chains (:71-104, covered in test_phase_repr_gpu.py).  The
audio is synthetic (three stereo clips) instead of the reference's three wav files; nothing is written to disk.
"""
import inspect

import pytest
import torch

import acids_transforms_amd as A
from acids_transforms_amd.transforms.base import AudioTransform

pytestmark = pytest.mark.gpu


def all_transforms():
    out = []
    for name in sorted(dir(A.transforms)):
        obj = getattr(A.transforms, name)
        if inspect.isclass(obj) and issubclass(obj, AudioTransform) and name not in (
                "AudioTransform", "ComposeAudioTransform", "SpectralRepresentation"):
            out.append(obj)
    return out


ALL = all_transforms()
INVERTIBLE = [c for c in ALL if c().invertible]


@pytest.fixture(scope="module")
def raw(dev):
    g = torch.Generator().manual_seed(1234)
    t = torch.arange(22050) / 44100.0
    tones = 0.4 * torch.sin(2 * torch.pi * torch.tensor([220.0, 330.0, 1000.0]).view(3, 1, 1) * t)
    noise = 0.05 * torch.randn(3, 2, 22050, generator=g)
    return (tones + noise).clamp(-0.99, 0.99).to(dev)


def finite(y):
    if isinstance(y, (tuple, list)):
        return all(finite(v) for v in y)
    if isinstance(y, dict):
        return all(finite(v) for v in y.values())
    if not isinstance(y, torch.Tensor) or not (y.is_floating_point() or y.is_complex()):
        return True
    return bool(torch.isfinite(torch.view_as_real(y) if y.is_complex() else y).all())


def test_every_class_of_the_reference_exists():
    names = {c.__name__ for c in ALL}
    assert names >= {"Mono", "Stereo", "MidSide", "Window", "MuLaw", "STFT", "RealtimeSTFT", "DGT", "RealtimeDGT", "Normalize",
                     "Real", "Imaginary", "Magnitude", "Phase", "IF", "Cartesian", "Polar", "PolarIF", "MFCC", "Unsqueeze",
                     "Squeeze", "Transpose", "OneHot", "OverlapAdd"}


@pytest.mark.parametrize("cls", ALL, ids=lambda c: c.__name__)
def test_forward(raw, cls):
    t = cls().to(raw.device)
    time = torch.zeros(raw.shape[:-1], device=raw.device)
    y = t.test_forward(raw)
    assert finite(y)
    y, tm = t.test_forward(raw, time)
    assert finite(y)


@pytest.mark.parametrize("cls", ALL, ids=lambda c: c.__name__)
def test_realtime(raw, cls):
    t = cls().to(raw.device)
    time = torch.zeros(raw.shape[:-1], device=raw.device)
    rt = t.realtime()
    if isinstance(rt, torch.nn.Module):
        rt = rt.to(raw.device)
    out = rt.test_forward(raw, time)
    assert finite(out)


@pytest.mark.parametrize("cls", INVERTIBLE, ids=lambda c: c.__name__)
def test_inversion(raw, cls):
    t = cls().to(raw.device)
    outs = t.test_inversion(raw)
    assert isinstance(outs, dict) and finite(outs)
    for k, v in outs.items():
        if isinstance(v, torch.Tensor) and v.is_floating_point() and v.ndim >= 2 and v.shape[:2] == raw.shape[:2]:
            assert v.shape[-1] > 0, (cls.__name__, k)


@pytest.mark.parametrize("cls", [c for c in ALL if c.__name__ not in ("Window",)], ids=lambda c: c.__name__)
def test_scripted_scenario(dev, cls):
    t = cls().to(dev)
    cls.test_scripted_transform(t, invert=t.invertible)
