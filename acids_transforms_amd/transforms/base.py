"""Composition layer: the nn.Module contract every transform honours.

Mirrors the public surface of the reference's transforms/base.py
(AudioTransform :13-80, ComposeAudioTransform :83-180, NotInvertibleError :6,
apply_transform_to_list :183-200): same method names, argument meaning and
error types, so user code written against acids_transforms keeps working.
HIP-backed modules report ``scriptable = False`` (a ctypes call cannot be
TorchScript-compiled) -- the one documented deviation.
"""
from typing import Optional, Union

import torch
import torch.nn as nn


class NotInvertibleError(Exception):
    pass


InversionEnumType = Union[str, None]


class AudioTransform(nn.Module):
    invertible = True
    scriptable = False
    needs_scaling = False

    def __init__(self, sr=44100):
        super().__init__()
        self.sr = sr

    def __repr__(self):
        return "AudioTransform()"

    def __add__(self, other):
        if isinstance(other, ComposeAudioTransform):
            return ComposeAudioTransform(transforms=[self] + list(other.transforms))
        if isinstance(other, AudioTransform):
            return ComposeAudioTransform(transforms=[self, other])
        raise TypeError("AudioTransform cannot be added to type: %s" % type(other))

    def scale_data(self, x: torch.Tensor) -> None:
        return None

    def forward(self, x):
        return x

    def get_inversion_modes(self):
        return None

    def invert(self, x: torch.Tensor, inversion_mode: InversionEnumType = None, **kwargs) -> torch.Tensor:
        return x

    def forward_with_time(self, x: torch.Tensor, time: torch.Tensor):
        return self.forward(x), time

    def realtime(self):
        return self

    @property
    def ratio(self):
        return 1

    # same self-test hooks as the reference classes expose
    def test_forward(self, x: torch.Tensor, time: Optional[torch.Tensor] = None):
        return self.forward(x) if time is None else self.forward_with_time(x, time)

    def test_inversion(self, x: torch.Tensor):
        if not self.invertible:
            raise NotImplementedError
        return {"inverted": self.invert(self.forward(x))}

    @classmethod
    def test_scripted_transform(cls, transform, batch_size=(2, 2), invert=True):
        """The reference runs this on a TorchScript copy of the module; the HIP modules are not scriptable
        (`scriptable = False`), the scenario itself -- silence in, forward, forward_with_time, invert -- is kept."""
        x = torch.zeros(*batch_size, 44100, device="cuda")
        time = torch.zeros(*batch_size, device="cuda")
        x_t = transform.forward(x)
        x_t, _ = transform.forward_with_time(x, time)
        if invert:
            transform.invert(x_t)


class ComposeAudioTransform(AudioTransform):
    def __init__(self, transforms=(), sr=44100):
        super().__init__(sr=sr)
        self.transforms = nn.ModuleList(list(transforms))

    @property
    def invertible(self):
        return all(t.invertible for t in self.transforms)

    @property
    def needs_scaling(self):
        return any(t.needs_scaling for t in self.transforms)

    @property
    def scriptable(self):
        return all(t.scriptable for t in self.transforms)

    def __getitem__(self, item):
        return self.transforms[item]

    def __len__(self):
        return len(self.transforms)

    def __repr__(self) -> str:
        return "ComposeAudioTransform(%s)" % [repr(t) + "\n" for t in self.transforms]

    def __add__(self, other):
        if not isinstance(other, AudioTransform):
            raise TypeError("ComposeAudioTransform can only be added to other AudioTransforms")
        if isinstance(other, ComposeAudioTransform):
            return ComposeAudioTransform(list(self.transforms) + list(other.transforms))
        return ComposeAudioTransform(list(self.transforms) + [other])

    def __radd__(self, other):
        if not isinstance(other, AudioTransform):
            raise TypeError("ComposeAudioTransform can only be added to other AudioTransforms")
        if isinstance(other, ComposeAudioTransform):
            return ComposeAudioTransform(list(other.transforms) + list(self.transforms))
        return ComposeAudioTransform([other] + list(self.transforms))

    def realtime(self):
        return ComposeAudioTransform(transforms=[t.realtime() for t in self.transforms], sr=self.sr)

    @property
    def ratio(self):
        r = 1
        for t in self.transforms:
            r = r * t.ratio
        return r

    def scale_data(self, x):
        # sequential: stage i is scaled on the output of stages < i (reference base.py:144-148)
        for t in self.transforms:
            t.scale_data(x)
            x = t(x)

    def forward(self, x: torch.Tensor):
        stages = list(self.transforms)
        i = 0
        while i < len(stages):
            t = stages[i]
            nxt = stages[i + 1] if i + 1 < len(stages) else None
            # STFT/DGT directly followed by a Magnitude with a banded (mel) bank runs as one fused kernel:
            # same results, the spectrum is not read back from HBM
            if nxt is not None and hasattr(nxt, "can_fuse_with") and nxt.can_fuse_with(t, x):
                x = nxt.forward_fused(t, x)
                i += 2
                continue
            x = t(x)
            i += 1
        return x

    def forward_with_time(self, x: torch.Tensor, time: torch.Tensor):
        for t in self.transforms:
            x, time = t.forward_with_time(x, time)
        return x, time

    def invert(self, x, inversion_mode: InversionEnumType = None):
        for t in reversed(list(self.transforms)):
            x = t.invert(x, inversion_mode=inversion_mode)
        return x

    def get_inversion_modes(self, idx):
        return type(self.transforms[idx]).get_inversion_modes()


def apply_transform_to_list(transform, data, time=None, **kwargs):
    if time is None:
        return [transform(d, **kwargs) for d in data]
    outs = [transform(d, time=t, **kwargs) for d, t in zip(data, time)]
    return [o[0] for o in outs], [o[1] for o in outs]


def apply_invert_transform_to_list(transform, data, time=None, **kwargs):
    if time is None:
        return [transform.invert(d, **kwargs) for d in data]
    outs = [transform.invert(d, time=t, **kwargs) for d, t in zip(data, time)]
    return [o[0] for o in outs], [o[1] for o in outs]
