"""`epg` namespace: operators + StateMatrix + functions (mirrors epgpy/core.py:80-83)"""
from .utils import *  # noqa: F401,F403
from .utils import Axes, gamma_1H, get_wavenumber
from .statematrix import StateMatrix
from .operators import *  # noqa: F401,F403
from .functions import (simulate, modify, get_adc_times, getshape, getnshift, getkdim, flatten_sequence,
                        compile_sequence)
