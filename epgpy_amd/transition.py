"""RF transition operators T, Tx, Ty, Phi (mirrors epgpy/transition.py:7-151)."""
import functools

import numpy as np

from . import common, opmatrix, diff


def rotation_alpha(alpha):
    """rotation about x by alpha degrees, EPG basis (transition.py:120-137)"""
    a = np.pi / 180.0 * np.atleast_1d(alpha)
    mat = np.empty(a.shape + (3, 3), dtype=np.complex128)
    half_c, half_s, s = np.cos(a / 2) ** 2, np.sin(a / 2) ** 2, np.sin(a)
    mat[..., 0, 0], mat[..., 0, 1], mat[..., 0, 2] = half_c, half_s, -1j * s
    mat[..., 1, 0], mat[..., 1, 1], mat[..., 1, 2] = half_s, half_c, 1j * s
    mat[..., 2, 0], mat[..., 2, 1], mat[..., 2, 2] = -1j / 2 * s, 1j / 2 * s, np.cos(a)
    return mat


@functools.lru_cache(maxsize=64)
def _rotation_phi_scalar(phi):
    """rotation_phi of ONE angle, remembered: a train has hundreds of pulses and a handful of phases (read-only array)"""
    mat = rotation_phi(np.float64(phi))
    mat.setflags(write=False)
    return mat


def rotation_phi(phi):
    """rotation about z by phi degrees (transition.py:140-151)"""
    if type(phi) in (int, float):
        return _rotation_phi_scalar(float(phi))
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


def rotation_alpha_d(alpha):
    """d rotation_alpha / d alpha, per degree (transition.py:175-190)"""
    a = np.pi / 180.0 * np.atleast_1d(alpha)
    mat = np.empty(a.shape + (3, 3), dtype=np.complex128)
    s, c = np.sin(a), np.cos(a)
    mat[..., 0, 0], mat[..., 0, 1], mat[..., 0, 2] = -0.5 * s, 0.5 * s, -1j * c
    mat[..., 1, 0], mat[..., 1, 1], mat[..., 1, 2] = 0.5 * s, -0.5 * s, 1j * c
    mat[..., 2, 0], mat[..., 2, 1], mat[..., 2, 2] = -1j / 2 * c, 1j / 2 * c, -s
    return mat * np.pi / 180


def rotation_phi_d(phi):
    """d rotation_phi / d phi, per degree (transition.py:193-200)"""
    p = np.atleast_1d(phi) * np.pi / 180.0
    mat = np.zeros(p.shape + (3, 3), dtype=np.complex128)
    mat[..., 0, 0] = 1j * np.exp(1j * p)
    mat[..., 1, 1] = -1j * np.exp(-1j * p)
    return mat * np.pi / 180


def rotation_d_alpha(alpha, phi):
    """transition.py:160-162"""
    alpha, phi = common.expand_arrays(alpha, phi, append=True)
    return rotation_phi(phi) @ rotation_alpha_d(alpha) @ rotation_phi(-phi)


def rotation_d_phi(alpha, phi):
    """transition.py:165-169"""
    alpha, phi = common.expand_arrays(alpha, phi, append=True)
    rx = rotation_alpha(alpha)
    return rotation_phi_d(phi) @ rx @ rotation_phi(-phi) - rotation_phi(phi) @ rx @ rotation_phi_d(-phi)


# -- second derivatives (transition.py:203-247) --------------------------------------------------------

def rotation_alpha_d2(alpha):
    a = np.pi / 180.0 * np.atleast_1d(alpha)
    mat = np.empty(a.shape + (3, 3), dtype=np.complex128)
    s, c = np.sin(a), np.cos(a)
    mat[..., 0, 0], mat[..., 0, 1], mat[..., 0, 2] = -0.5 * c, 0.5 * c, 1j * s
    mat[..., 1, 0], mat[..., 1, 1], mat[..., 1, 2] = 0.5 * c, -0.5 * c, -1j * s
    mat[..., 2, 0], mat[..., 2, 1], mat[..., 2, 2] = 1j / 2 * s, -1j / 2 * s, -c
    return mat * (np.pi / 180) ** 2


def rotation_phi_d2(phi):
    p = np.atleast_1d(phi) * np.pi / 180.0
    mat = np.zeros(p.shape + (3, 3), dtype=np.complex128)
    mat[..., 0, 0] = -np.exp(1j * p)
    mat[..., 1, 1] = -np.exp(-1j * p)
    return mat * (np.pi / 180) ** 2


def rotation_partials2(alpha, phi):
    """{('alpha','alpha'), ('alpha','phi'), ('phi','phi')} -> d2 R  (transition.py:203-221)"""
    alpha, phi = common.expand_arrays(alpha, phi, append=True)
    rx, rxd = rotation_alpha(alpha), rotation_alpha_d(alpha)
    zp, zm, dzp, dzm = rotation_phi(phi), rotation_phi(-phi), rotation_phi_d(phi), rotation_phi_d(-phi)
    return {
        ("alpha", "alpha"): zp @ rotation_alpha_d2(alpha) @ zm,
        ("alpha", "phi"): dzp @ rxd @ zm - zp @ rxd @ dzm,
        ("phi", "phi"): rotation_phi_d2(phi) @ rx @ zm + zp @ rx @ rotation_phi_d2(-phi) - 2 * dzp @ rx @ dzm,
    }


class T(diff.DiffMixin, opmatrix.MatrixOp):
    """instantaneous RF pulse: flip angle alpha, phase phi, degrees (transition.py:13-65)

    order1: first-order derivatives w.r.t. "alpha" / "phi" (see diff.py)"""

    PARAMETERS_ORDER1 = {"alpha", "phi"}
    PARAMETERS_ORDER2 = {("alpha", "alpha"), ("alpha", "phi"), ("phi", "phi")}

    def _raw_partials1(self):
        return {"alpha": (rotation_d_alpha(self.alpha, self.phi), None), "phi": (rotation_d_phi(self.alpha, self.phi), None)}

    def _raw_partials2(self):
        return {pair: (mat, None) for pair, mat in rotation_partials2(self.alpha, self.phi).items()}

    def _partial_tables(self, params):
        fun = {"alpha": rotation_d_alpha, "phi": rotation_d_phi}
        return {p: diff.pack_matrix_partial(fun[p](self.alpha, self.phi)) for p in params}

    def __init__(self, alpha, phi, *, axes=None, name=None, duration=None, **kwargs):
        self._init_partials(kwargs)
        self._daxes = axes
        params = common.map_arrays(alpha=alpha, phi=phi)
        if not name:
            name = common.repr_operator("T", ["alpha", "phi"], [alpha, phi], [".1f", "1f"])
        self.alpha, self.phi = params["alpha"], params["phi"]
        opmatrix.operator.Operator.__init__(self, name=name, duration=duration, **kwargs)
        self._init(rotation_operator(self.alpha, self.phi), None, axes=axes, check=False)  # symmetric by construction


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
        self._init(rotation_phi(self.phi), None, axes=axes, check=False)
