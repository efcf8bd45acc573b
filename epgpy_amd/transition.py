"""RF transition operators T, Tx, Ty, Phi (mirrors epgpy/transition.py:7-151)."""
import numpy as np

from . import common, opmatrix


def rotation_alpha(alpha):
    """rotation about x by alpha degrees, EPG basis (transition.py:120-137)"""
    a = np.pi / 180.0 * np.atleast_1d(alpha)
    mat = np.empty(a.shape + (3, 3), dtype=np.complex128)
    half_c, half_s, s = np.cos(a / 2) ** 2, np.sin(a / 2) ** 2, np.sin(a)
    mat[..., 0, 0], mat[..., 0, 1], mat[..., 0, 2] = half_c, half_s, -1j * s
    mat[..., 1, 0], mat[..., 1, 1], mat[..., 1, 2] = half_s, half_c, 1j * s
    mat[..., 2, 0], mat[..., 2, 1], mat[..., 2, 2] = -1j / 2 * s, 1j / 2 * s, np.cos(a)
    return mat


def rotation_phi(phi):
    """rotation about z by phi degrees (transition.py:140-151)"""
    p = np.atleast_1d(phi) * np.pi / 180.0
    mat = np.zeros(p.shape + (3, 3), dtype=np.complex128)
    mat[..., 0, 0] = np.exp(1j * p)
    mat[..., 1, 1] = np.exp(-1j * p)
    mat[..., 2, 2] = 1
    return mat


def rotation_operator(alpha, phi):
    """Rz(phi) Rx(alpha) Rz(-phi), same evaluation order as transition.py:114-117"""
    alpha, phi = common.expand_arrays(alpha, phi, append=True)
    return rotation_phi(phi) @ rotation_alpha(alpha) @ rotation_phi(-phi)


class T(opmatrix.MatrixOp):
    """instantaneous RF pulse: flip angle alpha, phase phi, degrees (transition.py:13-65)"""

    def __init__(self, alpha, phi, *, axes=None, name=None, duration=None, **kwargs):
        if kwargs.get("order1") or kwargs.get("order2"):
            raise NotImplementedError("derivatives (order1/order2) are outside the device hot path")
        kwargs.pop("order1", None), kwargs.pop("order2", None)
        params = common.map_arrays(alpha=alpha, phi=phi)
        if not name:
            name = common.repr_operator("T", ["alpha", "phi"], [alpha, phi], [".1f", "1f"])
        self.alpha, self.phi = params["alpha"], params["phi"]
        opmatrix.operator.Operator.__init__(self, name=name, duration=duration, **kwargs)
        self._init(rotation_operator(self.alpha, self.phi), None, axes=axes)


class Tx(T):
    def __init__(self, alpha, **kwargs):
        T.__init__(self, alpha, 0, **kwargs)


class Ty(T):
    def __init__(self, alpha, **kwargs):
        T.__init__(self, alpha, 90, **kwargs)


class Phi(opmatrix.MatrixOp):
    """phase offset (transition.py:79-108)"""

    def __init__(self, phi, *, axes=None, name=None, duration=0, **kwargs):
        if not name:
            name = common.repr_operator("Phi", ["phi"], [phi], [".1f"])
        self.phi = common.map_arrays(phi=phi)["phi"]
        opmatrix.operator.Operator.__init__(self, name=name, duration=duration, **kwargs)
        self._init(rotation_phi(self.phi), None, axes=axes)
