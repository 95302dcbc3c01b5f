"""MuLaw companding to integer codes (reference transforms/raw.py:265-316, which
wraps torchaudio MuLawEncoding/MuLawDecoding).  Codes are int64 and must match
bit for bit: see quant.hip."""

from .. import ops
from .base import AudioTransform, InversionEnumType

__all__ = ["MuLaw"]


class MuLaw(AudioTransform):
    scriptable = False
    invertible = True
    needs_scaling = False

    def __init__(self, channels=256, one_hot="none", **kwargs):
        super().__init__()
        self.channels = channels
        self.one_hot = one_hot

    def __repr__(self):
        return "MuLaw(channels=%s, one_hot=%s)" % (self.channels, self.one_hot)

    def encode(self, x):
        out = ops.mulaw_encode(x, self.channels)
        if self.one_hot == "channel":
            out = ops.onehot(out, self.channels, channel_major=True)
        elif self.one_hot == "categorical":
            out = ops.onehot(out, self.channels)
        return out

    def decode(self, x):
        x = x.long()
        if self.one_hot == "channel":
            x = ops.argmax_last(x.transpose(-2, -1))
        elif self.one_hot == "categorical":
            x = ops.argmax_last(x)
        return ops.mulaw_decode(x, self.channels)

    def forward(self, x):
        return self.encode(x)

    def invert(self, x, inversion_mode: InversionEnumType = None, tolerance: float = 1.e-4):
        # the reference decodes the codes directly, skipping the one-hot step (raw.py:316)
        return ops.mulaw_decode(x, self.channels)
