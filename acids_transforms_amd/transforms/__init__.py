from .base import (AudioTransform, ComposeAudioTransform, NotInvertibleError, InversionEnumType,
                   apply_transform_to_list, apply_invert_transform_to_list)
from .stft import STFT, RealtimeSTFT
from .dgt import DGT, RealtimeDGT, DGT_INVERSION_MODES
from .norm import Normalize
from .spectral_repr import Magnitude, Dummy
from .phase_repr import Real, Imaginary, Phase, IF, SpectralRepresentation, Cartesian, Polar, PolarIF
from .mel import MFCC
from .oadd import OverlapAdd
from .raw import MuLaw
from .misc import OneHot
from .channels import Mono, Stereo, MidSide, Window, Squeeze, Unsqueeze, Transpose

__all__ = ["AudioTransform", "ComposeAudioTransform", "NotInvertibleError", "InversionEnumType",
           "apply_transform_to_list", "apply_invert_transform_to_list", "STFT", "RealtimeSTFT", "DGT", "RealtimeDGT", "DGT_INVERSION_MODES",
           "Normalize", "Magnitude", "Real", "Imaginary", "Phase", "IF", "SpectralRepresentation", "Cartesian", "Polar",
           "PolarIF", "MFCC", "OverlapAdd", "MuLaw", "OneHot", "Mono", "Stereo", "MidSide", "Window", "Squeeze", "Unsqueeze", "Transpose"]
