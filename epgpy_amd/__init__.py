"""epgpy_amd -- MI355X-native engine for epgpy's hot path.

Drop-in for the data-parallel path of py-baudin/epgpy: `epg.T / E / P / S / ADC`,
`epg.StateMatrix`, `epg.simulate` with the reference's signatures, executed by hand-written
HIP kernels (csrc/) through the C ABI of include/epgx.h.  Usage mirrors the reference:

    from epgpy_amd import epg
    seq = [epg.T(90, 90)] + [[epg.S(1), epg.E(5, 150, [30, 40, 50]), epg.T(120, 0),
                              epg.S(1), epg.E(5, 150, [30, 40, 50]), epg.ADC]] * 20
    signal = epg.simulate(seq)          # (20, 3) complex128

There is no CPU execution path: without libepgx.so and a GPU every computation raises
`epgpy_amd.EpgxError`.
"""
from .core import *  # noqa: F401,F403
from . import core as epg
from . import operators, functions, statematrix, common, utils
from ._lib import EpgxError


def set_array_module(name=None):
    """accepted for source compatibility with `epgpy.set_array_module("numpy" | "cupy")`
    (common.py:21-74): this package has exactly one execution path, the HIP library, and host-side
    arrays are always NumPy -- the call changes nothing"""
    if name not in (None, "numpy", "cupy"):
        raise ValueError(f"Unknown array module: {name}")


def get_array_module(*args):
    """host-side array module (always NumPy; device data never surfaces as an array object)"""
    import numpy
    return numpy

__version__ = "0.1.0"
