"""Import shim used ONLY by tests/golden/make_golden.py inside the build container.

The reference imports `torchaudio` at module scope (utils/misc.py:6, stft.py:2,
raw.py:2, mel.py:2, spectral_repr.py via star-import) but torchaudio is not
installed in this image.  This package makes the *import* succeed; none of the
arithmetic that really lives in torchaudio is used to pin a golden vector
(DESIGN.md "parity unpinned" list).  The one function that must return data is
`functional.melscale_fbanks` (spectral_repr.py:177 calls it unconditionally in
`Magnitude.__init__`): it returns whatever bank make_golden.py injected.
"""
from . import functional, transforms  # noqa: F401


def load(*a, **k):  # utils/misc.py:31
    raise NotImplementedError("torchaudio.load is not available in the build container")


def save(*a, **k):
    raise NotImplementedError("torchaudio.save is not available in the build container")
