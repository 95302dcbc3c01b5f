from .misc import *  # noqa
from .heapq import heappush, heappop  # noqa
from .audio_io import Resample, resample, import_data, load_wav  # noqa
