from .base import *  # noqa: F401,F403
from .base import (AudioTransform, ComposeAudioTransform, NotInvertibleError, InversionEnumType,
                   apply_transform_to_list, apply_invert_transform_to_list)
from .stft import STFT, RealtimeSTFT
from .dgt import DGT, RealtimeDGT
