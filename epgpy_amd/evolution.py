"""Relaxation / precession operators E, P, R (mirrors epgpy/evolution.py:9-256)."""
import numpy as np

from . import common, opscalar, diff


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


# -- first derivatives of (arr, arr0): evolution.py:263-399 -----------------------------------------

def evolution_partials(rT, rL, r0):
    """{'rT', 'rL', 'r0'} -> (d arr, d arr0)  (evolution.py:263-285)"""
    zero = 0 * np.real(np.asarray(rL, dtype=complex))
    arr, _ = evolution_operator(rT, zero)
    arr[..., 2] = 0
    out = {"rT": (-arr, None)}
    arr, _ = evolution_operator(0 * np.asarray(rT), rL)
    arr[..., :2] = 0
    out["rL"] = (-arr, None)
    if r0 is not None:
        arr, arr0 = evolution_operator(0 * np.asarray(rT), zero, r0)
        arr0[..., 2] -= 1
        out["r0"] = (0 * arr, -arr0)
    return out


def precession_partials(tau, g):
    """{'tau', 'g'} -> (d arr, None)  (evolution.py:314-329)"""
    tau, g = common.expand_arrays(tau, g, append=True)
    out = {}
    for name, factor in (("tau", -2j * np.pi * g), ("g", -2j * np.pi * tau)):
        arr, _ = evolution_operator(2j * np.pi * g * tau, rL=0, r0=None)
        arr[..., 1] *= factor
        arr[..., 0] = arr[..., 1].conj()
        arr[..., 2] = 0
        out[name] = (arr, None)
    return out


def relaxation_partials(tau, T1, T2, g, only=None):
    """{'tau', 'T1', 'T2', 'g'} -> (d arr, d arr0)  (evolution.py:360-399); `only`: the parameters wanted (a plan asks for
    the ones its variables name: four exp() evaluations per operator are the cost of building a differentiated train)"""
    tau, T1, T2, g = common.expand_arrays(tau, T1, T2, g, append=True)
    rT = tau * (1 / T2 + 2j * np.pi * g)
    rL = tau / T1
    want = (lambda name: True) if only is None else (lambda name: name in only)
    out = {}
    if want("tau"):
        arr, arr0 = evolution_operator(rT, rL, rL)
        arr[..., 1] *= -rT / tau
        arr[..., 0] = arr[..., 1].conj()
        arr[..., 2] *= -1 / T1
        arr0[..., 2] = -arr[..., 2]
        out["tau"] = (arr, arr0)
    if want("T1"):
        arr, arr0 = evolution_operator(0 * rT, rL, rL)
        arr[..., :2] = 0
        arr[..., 2] *= tau / T1 ** 2
        arr0[..., 2] = -arr[..., 2]
        out["T1"] = (arr, arr0)
    if want("T2"):
        arr, _ = evolution_operator(rT, 0 * rL)
        arr[..., :2] *= np.asarray(tau / T2 ** 2)[..., None]
        arr[..., 2] = 0
        out["T2"] = (arr, None)
    if want("g"):
        arr, _ = evolution_operator(rT, 0 * rL)
        arr[..., 1] *= -2j * np.pi * tau
        arr[..., 0] = arr[..., 1].conj()
        arr[..., 2] = 0
        out["g"] = (arr, None)
    return out


# -- second derivatives of (arr, arr0): evolution.py:288-303, :331-355, :402-487 ------------------------

def evolution_partials2(rT, rL, r0):
    zero = 0 * np.real(np.asarray(rL, dtype=complex))
    arr, _ = evolution_operator(rT, zero)
    arr[..., 2] = 0
    out = {("rT", "rT"): (arr, None)}
    arr, _ = evolution_operator(0 * np.asarray(rT), rL)
    arr[..., :2] = 0
    out[("rL", "rL")] = (arr, None)
    if r0 is not None:
        arr, arr0 = evolution_operator(0 * np.asarray(rT), zero, r0)
        arr0[..., 2] -= 1
        out[("r0", "r0")] = (0 * arr, arr0)
    return out


def precession_partials2(tau, g):
    tau, g = common.expand_arrays(tau, g, append=True)
    out = {}
    for pair, factor in ((("tau", "tau"), (-2j * np.pi * g) ** 2), (("g", "g"), (-2j * np.pi * tau) ** 2),
                         (("g", "tau"), -2j * np.pi * (1 - 2j * np.pi * g * tau))):
        arr, _ = evolution_operator(2j * np.pi * g * tau, rL=0, r0=None)
        arr[..., 1] *= factor
        arr[..., 0] = arr[..., 1].conj()
        arr[..., 2] = 0
        out[pair] = (arr, None)
    return out


def relaxation_partials2(tau, T1, T2, g):
    tau, T1, T2, g = common.expand_arrays(tau, T1, T2, g, append=True)
    rT = tau * (1 / T2 + 2j * np.pi * g)
    rL = tau / T1
    out = {}

    def transverse(factor, recover=True):
        arr, _ = evolution_operator(rT, rL, rL) if recover else evolution_operator(rT, 0 * rL)
        arr[..., 1] *= factor
        arr[..., 0] = arr[..., 1].conj()
        arr[..., 2] = 0
        return arr, None

    arr, arr0 = evolution_operator(rT, rL, rL)
    arr[..., 1] *= (rT / tau) ** 2
    arr[..., 0] = arr[..., 1].conj()
    arr[..., 2] *= 1 / T1 ** 2
    arr0[..., 2] = -arr[..., 2]
    out[("tau", "tau")] = (arr, arr0)
    arr, arr0 = evolution_operator(0 * rT, rL, rL)
    arr[..., :2] = 0
    arr[..., 2] *= tau ** 2 / T1 ** 4 - 2 * tau / T1 ** 3
    arr0[..., 2] = -arr[..., 2]
    out[("T1", "T1")] = (arr, arr0)
    arr, _ = evolution_operator(rT, 0 * rL)
    arr[..., :2] *= np.asarray(tau ** 2 / T2 ** 4 - 2 * tau / T2 ** 3)[..., None]
    arr[..., 2] = 0
    out[("T2", "T2")] = (arr, None)
    out[("g", "g")] = transverse((-2j * np.pi * tau) ** 2, recover=False)
    arr, arr0 = evolution_operator(rT, rL, rL)
    arr[..., :2] = 0
    arr[..., 2] *= (1 - rL) / T1 ** 2
    arr0[..., 2] = -arr[..., 2]
    out[("T1", "tau")] = (arr, arr0)
    out[("T2", "tau")] = transverse((1 - rT) / T2 ** 2)
    out[("g", "tau")] = transverse(-2j * np.pi * (1 - rT))
    out[("T2", "g")] = transverse(-2j * np.pi * (tau / T2) ** 2)
    return out


class _DiffScalar(diff.DiffMixin):
    def _raw_partials1(self):
        return self._partials()

    def _partials(self, only=None):
        raise NotImplementedError

    def _partial_tables(self, params):
        partials = self._partials(only=set(params))
        return {p: diff.pack_scalar_partial(*partials[p]) for p in params}

    def _partial_column_groups(self, var):
        """column groups of d(table)/d(var) for Encoder.assembled_table, or None.  Differentiation does not widen what a
        column depends on: the F factor's partial varies along the axes of the transverse parameters only, the Z factor's
        and the recovery's along those of the longitudinal ones -- the groups of the value table (`_column_groups`) with
        the partial's values sliced out of the (memoised) partial table.  Not with array-valued coefficients in order1
        (they add axes of their own) or a custom `axes=` placement"""
        cache = self.__dict__.setdefault("_partial_colgroups", {})
        if var not in cache:
            cols = None
            scalar_coeffs = all(np.ndim(c) == 0 for c in self.order1[var].values())
            if scalar_coeffs and self._daxes is None and self._column_groups() is not None:
                table = self._variable_tables()[var]
                value_groups, _ = self._column_groups()
                lead = table.shape[:-1]
                if len(lead) == value_groups[0].ndim - 1:
                    sel = [tuple(slice(None) if n > 1 else slice(0, 1) for n in g.shape[:-1]) for g in value_groups]
                    groups = [np.ascontiguousarray(table[sel[0] + (slice(0, 2),)]), np.ascontiguousarray(table[sel[1] + (slice(2, 4),)])]
                    cols = (groups, [(0, 0), (0, 1), (1, 0), (1, 1)])
            cache[var] = cols
        return cache[var]


class R(_DiffScalar, opscalar.ScalarOp):
    """evolution with explicit rates rT, rL, r0 (evolution.py:9-66)"""

    PARAMETERS_ORDER1 = {"rT", "rL", "r0"}
    PARAMETERS_ORDER2 = {("rT", "rT"), ("rL", "rL"), ("r0", "r0")}

    def _partials(self, only=None):
        return evolution_partials(self.rT, self.rL, self.r0)

    def _raw_partials2(self):
        return evolution_partials2(self.rT, self.rL, self.r0)

    def __init__(self, rT=0, rL=0, *, r0=None, axes=None, name=None, duration=None, **kwargs):
        self._init_partials(kwargs)
        self._daxes = axes
        rT, rL, r0 = common.map_arrays([rT, rL, r0])
        if not name:
            name = common.repr_operator("R", ["rT", "rL", "r0"], [rT, rL, r0], [".1f"] * 3)
        self.rT, self.rL, self.r0 = rT, rL, r0
        opscalar.operator.Operator.__init__(self, name=name, duration=duration, **kwargs)
        self._init(*evolution_operator(rT, rL, r0), axes=axes, check=False)  # arr[0] = conj(arr[1]) by construction

    def _column_groups(self):
        if self._daxes is not None:
            return None
        rT, rL, r0 = common.expand_arrays(self.rT, self.rL, self.r0, append=True)
        return self._dependent_columns([rT], [rL] + ([] if r0 is None else [r0]))


class E(_DiffScalar, opscalar.ScalarOp):
    """relaxation + precession during tau (evolution.py:69-153)"""

    PARAMETERS_ORDER1 = {"tau", "T1", "T2", "g"}
    PARAMETERS_ORDER2 = {("tau", "tau"), ("T1", "T1"), ("T2", "T2"), ("g", "g"), ("T1", "tau"), ("T2", "tau"),
                         ("g", "tau"), ("T2", "g")}

    def _partials(self, only=None):
        return relaxation_partials(self.tau, self.T1, self.T2, self.g, only)

    def _partial_column_groups(self, var):
        """the column groups of d(table)/d(var) evaluated DIRECTLY on the parameters they depend on -- the F factor's partial on
        (tau, T2, g) with T1 held at one value, the Z factor's and the recovery's on (tau, T1) -- instead of evaluating the
        full [*grid, 4] partial table and slicing it (0.8 ms per relaxation of a 100 x 100 grid: half the compile time of a
        differentiated MRF train).  The same elementwise arithmetic on the same inputs: the same bits
        (`test_relaxation_partial_columns_equal_the_sliced_table`)"""
        cache = self.__dict__.setdefault("_partial_colgroups", {})
        if var not in cache:
            cols = None
            coeffs = self.order1[var]
            if all(np.ndim(c) == 0 for c in coeffs.values()) and self._daxes is None and self._column_groups() is not None:
                value_groups, columns = self._column_groups()
                tau, T1, T2, g = common.expand_arrays(self.tau, self.T1, self.T2, self.g, append=True)

                def one(x):
                    x = np.asarray(x)
                    return x.reshape(-1)[:1].reshape((1,) * x.ndim)

                def table(parts):
                    total = None
                    for param, coeff in coeffs.items():
                        term = diff.pack_scalar_partial(*parts[param]) * np.asarray(coeff, dtype=np.float64)
                        total = term if total is None else total + term
                    return np.ascontiguousarray(total, dtype=np.float64)

                want = set(coeffs)
                transverse = table(relaxation_partials(tau, one(T1), T2, g, want))[..., 0:2]
                longitudinal = table(relaxation_partials(tau, T1, one(T2), one(g), want))[..., 2:4]
                if transverse.shape == value_groups[0].shape and longitudinal.shape == value_groups[1].shape:
                    cols = ([np.ascontiguousarray(transverse), np.ascontiguousarray(longitudinal)], columns)
                else:
                    cols = super()._partial_column_groups(var)
                    self.__dict__["_partial_colgroups"].pop(var, None)
            cache[var] = cols
        return cache[var]

    def _partial_shape_facts(self, var):
        """(leading shape of d(table)/d(var), whether it has an imaginary part) without evaluating it, or None"""
        coeffs = self.order1.get(var)
        if not coeffs or self._daxes is not None or not all(np.ndim(c) == 0 for c in coeffs.values()):
            return None
        return tuple(self.arr.shape[:-1]), bool("g" in coeffs or np.any(np.asarray(self.g) != 0))

    def _raw_partials2(self):
        return relaxation_partials2(self.tau, self.T1, self.T2, self.g)

    def __init__(self, tau, T1, T2, g=0, *, axes=None, name=None, duration=None, **kwargs):
        self._init_partials(kwargs)
        self._daxes = axes
        tau, T1, T2, g = common.map_arrays([tau, T1, T2, g])
        if not name:
            name = common.repr_operator("E", ["tau", "T1", "T2", "g"], [tau, T1, T2, g],
                                        [".1f", ".1f", ".1f", ".3f"])
        self.tau, self.T1, self.T2, self.g = tau, T1, T2, g
        self._duration = duration
        duration = self.tau if duration is True else duration
        opscalar.operator.Operator.__init__(self, name=name, duration=duration, **kwargs)
        self._init(*relaxation_operator(tau, T1, T2, g), axes=axes, check=False)

    def _column_groups(self):
        if self._daxes is not None:
            return None
        tau, T1, T2, g = common.expand_arrays(self.tau, self.T1, self.T2, self.g, append=True)
        return self._dependent_columns([tau, T2, g], [tau, T1])


class P(_DiffScalar, opscalar.ScalarOp):
    """precession only (evolution.py:156-213)"""

    PARAMETERS_ORDER1 = {"tau", "g"}
    PARAMETERS_ORDER2 = {("tau", "tau"), ("g", "g"), ("g", "tau")}

    def _partials(self, only=None):
        return precession_partials(self.tau, self.g)

    def _raw_partials2(self):
        return precession_partials2(self.tau, self.g)

    def __init__(self, tau, g, *, axes=None, name=None, duration=None, **kwargs):
        self._init_partials(kwargs)
        self._daxes = axes
        tau, g = common.map_arrays([tau, g])
        if not name:
            name = common.repr_operator("P", ["tau", "g"], [tau, g], [".1f", ".3f"])
        self.tau, self.g = tau, g
        self._duration = duration
        duration = self.tau if duration is True else duration
        opscalar.operator.Operator.__init__(self, name=name, duration=duration, **kwargs)
        self._init(*precession_operator(tau, g), axes=axes, check=False)

    def _column_groups(self):
        if self._daxes is not None:
            return None
        tau, g = common.expand_arrays(self.tau, self.g, append=True)
        return self._dependent_columns([tau, g], [])
