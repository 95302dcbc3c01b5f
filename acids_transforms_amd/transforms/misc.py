"""OneHot (reference transforms/misc.py:156-213): F.one_hot forward, argmax inverse."""
import torch

from .. import ops
from .base import AudioTransform, InversionEnumType

__all__ = ["OneHot"]


class OneHot(AudioTransform):
    scriptable = False
    invertible = True

    @property
    def needs_scaling(self):
        return self.n_classes == -1

    def __init__(self, sr=44100, dtype=torch.long, n_classes: int = -1):
        super().__init__(sr)
        self.dtype = dtype
        self.n_classes = n_classes

    def __repr__(self):
        return "OneHot(n_classes=%s)" % self.n_classes

    def scale_data(self, x: torch.Tensor) -> None:
        self.n_classes = int(x.max()) + 1

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        n = self.n_classes
        if n == -1:     # F.one_hot infers the class count from the data
            n = int(x.max()) + 1
        return ops.onehot(x, n)

    def invert(self, x_onehot: torch.Tensor, inversion_mode: InversionEnumType = None,
               tolerance: float = 1.e-4) -> torch.Tensor:
        return ops.argmax_last(x_onehot)

    # -- self-test hooks (reference misc.py:191-215): a random mu-law code vector stands in for the audio -------
    def test_forward(self, x: torch.Tensor, time: torch.Tensor = None):
        codes = torch.randint(0, 256, (2, 44100), device=x.device)
        self.scale_data(codes)
        return self(codes) if time is None else self.forward_with_time(codes, time)

    def test_inversion(self, x: torch.Tensor):
        self.invert(torch.randint(0, 256, tuple(x.shape), device=x.device).to(x.dtype))
        return {}

    @classmethod
    def test_scripted_transform(cls, transform, invert: bool = True):
        codes = torch.randint(0, 256, (2, 44100), device="cuda")
        transform.scale_data(codes)
        y = transform(codes)
        if invert:
            transform.invert(y)
