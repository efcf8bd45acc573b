"""Relaxation / precession operators E, P, R (mirrors epgpy/evolution.py:9-256)."""
import numpy as np

from . import common, opscalar


def evolution_operator(rT, rL, r0=None):
    """arr = [conj(e^-rT), e^-rT, e^-rL], arr0 = [0, 0, 1 - e^-r0] (evolution.py:220-242)"""
    rT, rL, r0 = common.expand_arrays(rT, rL, r0, append=True)
    shape = common.broadcast_shapes(common.get_shape(rT), common.get_shape(rL),
                                    common.get_shape(r0), [1])
    arr = np.zeros(shape + (3,), dtype=np.complex128)
    arr[..., 1] = np.exp(-rT)
    arr[..., 0] = arr[..., 1].conj()
    arr[..., 2] = np.exp(-rL)
    arr0 = None
    if r0 is not None:
        arr0 = np.zeros(shape + (3,), dtype=np.complex128)
        arr0[..., 2] = 1 - np.exp(-r0)
    return arr, arr0


def precession_operator(tau, g):
    """evolution.py:245-248"""
    tau, g = common.expand_arrays(tau, g, append=True)
    return evolution_operator(2j * np.pi * g * tau, rL=0, r0=None)


def relaxation_operator(tau, T1, T2, g):
    """evolution.py:251-256: tau, T1, T2 in ms, g in kHz"""
    tau, T1, T2, g = common.expand_arrays(tau, T1, T2, g, append=True)
    rT = tau * (1 / T2 + 2j * np.pi * g)
    rL = tau / T1
    return evolution_operator(rT, rL, rL)


def _no_derivatives(kwargs):
    if kwargs.get("order1") or kwargs.get("order2"):
        raise NotImplementedError("derivatives (order1/order2) are outside the device hot path")
    kwargs.pop("order1", None), kwargs.pop("order2", None)


class R(opscalar.ScalarOp):
    """evolution with explicit rates rT, rL, r0 (evolution.py:9-66)"""

    def __init__(self, rT=0, rL=0, *, r0=None, axes=None, name=None, duration=None, **kwargs):
        _no_derivatives(kwargs)
        rT, rL, r0 = common.map_arrays([rT, rL, r0])
        if not name:
            name = common.repr_operator("R", ["rT", "rL", "r0"], [rT, rL, r0], [".1f"] * 3)
        self.rT, self.rL, self.r0 = rT, rL, r0
        opscalar.operator.Operator.__init__(self, name=name, duration=duration, **kwargs)
        self._init(*evolution_operator(rT, rL, r0), axes=axes)


class E(opscalar.ScalarOp):
    """relaxation + precession during tau (evolution.py:69-153)"""

    def __init__(self, tau, T1, T2, g=0, *, axes=None, name=None, duration=None, **kwargs):
        _no_derivatives(kwargs)
        tau, T1, T2, g = common.map_arrays([tau, T1, T2, g])
        if not name:
            name = common.repr_operator("E", ["tau", "T1", "T2", "g"], [tau, T1, T2, g],
                                        [".1f", ".1f", ".1f", ".3f"])
        self.tau, self.T1, self.T2, self.g = tau, T1, T2, g
        self._duration = duration
        duration = self.tau if duration is True else duration
        opscalar.operator.Operator.__init__(self, name=name, duration=duration, **kwargs)
        self._init(*relaxation_operator(tau, T1, T2, g), axes=axes)


class P(opscalar.ScalarOp):
    """precession only (evolution.py:156-213)"""

    def __init__(self, tau, g, *, axes=None, name=None, duration=None, **kwargs):
        _no_derivatives(kwargs)
        tau, g = common.map_arrays([tau, g])
        if not name:
            name = common.repr_operator("P", ["tau", "g"], [tau, g], [".1f", ".3f"])
        self.tau, self.g = tau, g
        self._duration = duration
        duration = self.tau if duration is True else duration
        opscalar.operator.Operator.__init__(self, name=name, duration=duration, **kwargs)
        self._init(*precession_operator(tau, g), axes=axes)
