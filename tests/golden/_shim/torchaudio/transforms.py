"""See package docstring: constructors exist so the reference modules import and
instantiate; calling them raises."""
import torch


class _Unavailable(torch.nn.Module):
    def __init__(self, *a, **k):
        super().__init__()

    def forward(self, *a, **k):
        raise NotImplementedError("%s is not available in the build container" % type(self).__name__)


class MelSpectrogram(_Unavailable):
    pass


class MuLawEncoding(_Unavailable):
    pass


class MuLawDecoding(_Unavailable):
    pass


class Resample(_Unavailable):
    pass
