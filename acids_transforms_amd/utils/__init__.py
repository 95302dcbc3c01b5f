from .misc import *  # noqa
