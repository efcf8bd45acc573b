"""First-order derivatives (mirrors the order-1 part of epgpy/diff.py:20-288, :384-416).

The reference carries, next to the state matrix, one derivative state matrix per variable and
updates it operator by operator in NumPy:

    dS_v <- Op(dS_v, equilibrium removed) + sum_p coeff[v][p] * (dOp/dp)(S)          (diff.py:264-288)

Here the recurrence runs inside the device kernel (csrc/epgx_deriv_kernels.hip.h): a wavefront
keeps S and up to three dS_v of its voxel in registers, and every operator that was built with
`order1=...` ships, per variable, ONE table "sum_p coeff[v][p] * dOp/dp" to the plan (the
combination over parameters is linear in the tables, so it is done once on the host).  The
`Jacobian` probe reads F0 / Z0 of S ("magnitude") and of every dS_v at each ADC.

Second-order derivatives (order2, Hessian) are outside the device path.
"""
import numpy as np

from . import common, probe as _probe


def parse_order1(order1, order2, parameters):
    """normalise `order1` to {variable: {parameter: coefficient}}  (diff.py:153-198)"""
    if order2:
        raise NotImplementedError("second-order derivatives (order2) are outside the device hot path")
    parameters = set(parameters)
    if isinstance(order1, str):
        order1 = [order1]
    if not order1:
        order1 = {}
    elif order1 is True:
        order1 = {param: {param: 1} for param in parameters}
    elif isinstance(order1, (list, tuple, set)):
        order1 = {param: {param: 1} for param in order1}
    elif isinstance(order1, dict) and all(isinstance(value, str) for value in order1.values()):
        order1 = {var: {order1[var]: 1} for var in order1}
    elif isinstance(order1, dict) and all(isinstance(value, dict) for value in order1.values()):
        order1 = {var: dict(order1[var]) for var in order1}
    else:
        raise ValueError(f"Invalid parameter 'order1' value: {order1}")
    invalid = {param for var in order1 for param in set(order1[var]) - parameters}
    if invalid:
        raise ValueError(f"Unknown parameter(s): {invalid}")
    return order1


class DiffMixin:
    """operators that know their partial derivatives: T (alpha, phi), E (tau, T1, T2, g),
    P (tau, g), R (rT, rL, r0)"""

    PARAMETERS_ORDER1 = set()
    order1 = {}
    order2 = set()

    def _init_partials(self, kwargs):
        """pops order1 / order2 from the constructor keywords"""
        self.order1 = parse_order1(kwargs.pop("order1", False), kwargs.pop("order2", False),
                                   self.PARAMETERS_ORDER1)
        self.order2 = set()
        self._dtables = None
        self._daxes = None

    @property
    def parameters_order1(self):
        return {param for var in self.order1 for param in self.order1[var]}

    def _partial_tables(self, params):
        """{param: float64 [*opshape, ncoef] device table of dOp/dparam} -- per operator class"""
        raise NotImplementedError

    def _variable_tables(self):
        """{variable: table of sum_p coeff * dOp/dp}; a non-scalar coefficient broadcasts against
        the operator's own shape under the append-axes rule (diff.py:535-551 combine_partials)"""
        if self._dtables is None:
            partials = self._partial_tables(self.parameters_order1)
            tables = {}
            for var, coeffs in self.order1.items():
                total = None
                for param, coeff in coeffs.items():
                    tab = partials[param]
                    coeff = np.asarray(coeff, dtype=np.float64)
                    if coeff.ndim:
                        ndim = max(tab.ndim - 1, coeff.ndim)
                        tab = tab.reshape(tab.shape[:-1] + (1,) * (ndim - (tab.ndim - 1)) + tab.shape[-1:])
                        coeff = coeff.reshape(coeff.shape + (1,) * (ndim - coeff.ndim))[..., None]
                    term = tab * coeff
                    total = term if total is None else total + term
                if self._daxes is not None:
                    total = common.set_axes(1, total, self._daxes)
                tables[var] = np.ascontiguousarray(total, dtype=np.float64)
            self._dtables = tables
        return self._dtables

    def _encode(self, enc):
        super()._encode(enc)
        if self.order1:
            enc.add_partials({var: (tab, ("D1", id(self), var)) for var, tab in self._variable_tables().items()})

    # -- operator-by-operator use: op(sm) keeps sm.order1 up to date (diff.py:119-139, :264-288) ----
    def _partial_ops(self):
        """{variable: plain operator that applies sum_p coeff * dOp/dp}, built from the device tables"""
        if getattr(self, "_dops", None) is None:
            from . import opmatrix, opscalar
            ops = {}
            for var, tab in self._variable_tables().items():
                if tab.shape[-1] == 10:      # general symmetric 3x3
                    u, p_, q = tab[..., 0] + 1j * tab[..., 1], tab[..., 2] + 1j * tab[..., 3], tab[..., 4] + 1j * tab[..., 5]
                    t_, c22 = tab[..., 6] + 1j * tab[..., 7], tab[..., 8]
                    mat = np.empty(tab.shape[:-1] + (3, 3), dtype=np.complex128)
                    mat[..., 0, 0], mat[..., 0, 1], mat[..., 0, 2] = u, p_, q
                    mat[..., 1, 0], mat[..., 1, 1], mat[..., 1, 2] = p_.conj(), u.conj(), q.conj()
                    mat[..., 2, 0], mat[..., 2, 1], mat[..., 2, 2] = t_, t_.conj(), c22
                    ops[var] = opmatrix.MatrixOp(mat, check=False, name=f"d{self.name}/d{var}")
                else:                        # diagonal + recovery
                    e0 = tab[..., 0] + 1j * tab[..., 1]
                    arr = np.stack([e0, e0.conj(), tab[..., 2] + 0j], axis=-1)
                    arr0 = np.stack([np.zeros_like(e0), np.zeros_like(e0), tab[..., 3] + 0j], axis=-1)
                    ops[var] = opscalar.ScalarOp(arr, arr0, check=False, name=f"d{self.name}/d{var}")
            self._dops = ops
        return self._dops

    def __call__(self, sm, *, inplace=False):
        """sm <- Op(sm), and for every derivative state  dS_v <- Op(dS_v) + (dOp/dv)(S)"""
        from .plan import apply_operators
        previous = getattr(sm, "order1", None) or {}
        if not previous and not self.order1:
            return super().__call__(sm, inplace=inplace)
        sm = self.prepare(sm, inplace=inplace)
        order1 = {}
        for var, dsm in previous.items():        # derivative states carry no equilibrium: plain apply
            dsm = self.prepare(dsm, inplace=inplace)
            order1[var] = apply_operators(dsm, [self])
        for var, dop in self._partial_ops().items() if self.order1 else ():
            part = apply_operators(dop.prepare(sm, inplace=False), [dop])
            if var in order1:
                _accumulate(order1[var], part)
            else:
                part._state.zero_density()
                order1[var] = part
        sm = self._apply(sm)
        sm.order1 = order1
        return sm

    def combine(self, other, **kwargs):
        if self.order1 or getattr(other, "order1", None):
            raise NotImplementedError("combining (@) operators that carry order1 derivatives")
        return super().combine(other, **kwargs)


def _accumulate(dsm, part):
    """dsm += part on the device (same grid and capacity first)"""
    grid = common.broadcast_shapes(dsm.shape, part.shape, append=True)
    dsm._broadcast_to(grid)
    part._broadcast_to(grid)
    K = max(dsm._state.K, part._state.K)
    dsm._reserve(K)
    part._reserve(K)
    dsm._nstate = max(dsm._nstate, part._nstate)
    dsm._state.axpy(part._state, 1.0, zero_density=True)


def propagate_plain(op, sm, order1, inplace):
    """operators without parameters of their own that still act on derivative states (S)"""
    from .plan import apply_operators
    return {var: apply_operators(op.prepare(dsm, inplace=inplace), [op]) for var, dsm in order1.items()}


def pack_matrix_partial(mat):
    """d(mat)/dp [..., 3, 3] -> the 10 coefficients of the general symmetric device form"""
    m00, m01, m02, m20, m22 = mat[..., 0, 0], mat[..., 0, 1], mat[..., 0, 2], mat[..., 2, 0], mat[..., 2, 2]
    cols = [m00.real, m00.imag, m01.real, m01.imag, m02.real, m02.imag, m20.real, m20.imag,
            m22.real, np.zeros_like(m22.real)]
    return np.ascontiguousarray(np.stack(cols, axis=-1), dtype=np.float64)


def pack_scalar_partial(arr, arr0=None):
    """d(arr, arr0)/dp -> Re/Im d arr[0], d arr[2], d arr0[2]"""
    e0, e2 = arr[..., 0], arr[..., 2]
    r0 = np.zeros(e2.shape) if arr0 is None else arr0[..., 2].real
    cols = [e0.real, e0.imag, e2.real, np.broadcast_to(r0, e2.shape)]
    return np.ascontiguousarray(np.stack(cols, axis=-1), dtype=np.float64)


class Jacobian(_probe.Probe):
    """probe of the signal's first derivatives: [..., len(variables)]  (diff.py:384-416)

    variables: names given to `order1=` in the sequence, plus "magnitude" for the signal itself;
    a name no operator differentiates against gives zeros.
    """

    def __init__(self, variables, *, probe="F0"):
        if probe not in _probe.DEVICE_KINDS:
            raise NotImplementedError(f"Jacobian probe {probe!r}: the device records F0 or Z0")
        self.probe = probe
        if not isinstance(variables, list):
            variables = [variables]
        self.variables = variables
        _probe.operator.Operator.__init__(self, name=None)
        self._post = None

    def __repr__(self):
        return f"Jacobian({self.probe})"

    def _device_kind(self):
        return _probe.DEVICE_KINDS[self.probe]

    def _device_variables(self):
        return [var for var in self.variables if var != "magnitude"]

    def _assemble(self, base, partials):
        """base: probe of the state; partials: {variable: probe of the derivative state}"""
        zeros = None
        cols = []
        for var in self.variables:
            if var == "magnitude":
                cols.append(base)
            elif var in partials:
                cols.append(partials[var])
            else:
                zeros = np.zeros(base.shape) if zeros is None else zeros   # real zeros, as the reference
                cols.append(zeros)
        return np.stack(cols, axis=-1)

    def _acquire(self, sm):
        """operator-by-operator path (callbacks, op(sm) chains): read the probe off sm.order1"""
        order1 = getattr(sm, "order1", None) or {}
        return self._assemble(np.asarray(getattr(sm, self.probe)),
                              {var: np.asarray(getattr(dsm, self.probe)) for var, dsm in order1.items()})


class Hessian(_probe.Probe):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("second-order derivatives (Hessian) are outside the device hot path")
