"""acids_transforms_amd -- the spectral hot path of acids_transforms on MI355X.

Same class names and nn.Module API as the reference package; the arithmetic
is hand-written HIP for gfx950 behind a C ABI (include/acids_hip.h).
"""
from .utils import *  # noqa: F401,F403  (the reference re-exports its utils at package level: __init__.py:1)
from .transforms import *  # noqa: F401,F403
from ._lib import AcidsHipError, build  # noqa: F401
from .ops import allow_fp64_narrowing  # noqa: F401
# the reference's star-imports also bind its submodules at package level (acids_transforms.stft, .dgt, .norm, ...,
# .heapq; `misc` is transforms.misc there, imported last): same names here, plus the modules of the rows SURVEY 8f added
from .utils import heapq  # noqa: F401,E402
from .transforms import (base, raw, stft, dgt, norm, spectral_repr, mel, misc, oadd, phase_repr, channels,  # noqa: F401,E402
                         sinebank)

__version__ = "0.1.0"
