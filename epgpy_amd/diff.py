"""First-order derivatives (mirrors the order-1 part of epgpy/diff.py:20-288, :384-416).

The reference carries, next to the state matrix, one derivative state matrix per variable and
updates it operator by operator in NumPy:

    dS_v <- Op(dS_v, equilibrium removed) + sum_p coeff[v][p] * (dOp/dp)(S)          (diff.py:264-288)

Here the recurrence runs inside the device kernel (csrc/epgx_deriv_kernels.hip.h): a wavefront
keeps S and up to three dS_v of its voxel in registers, and every operator that was built with
`order1=...` ships, per variable, ONE table "sum_p coeff[v][p] * dOp/dp" to the plan (the
combination over parameters is linear in the tables, so it is done once on the host).  The
`Jacobian` probe reads F0 / Z0 of S ("magnitude") and of every dS_v at each ADC.

Second-order derivatives (order2, Hessian) follow the reference's recurrence operator by operator
(`_apply_order2`): every term is a device state matrix, updated with the same kernels and
`epgx_state_axpy`; there is no fused kernel for them.
"""
import numpy as np

from . import common, probe as _probe


def Pair(p1, p2=None):
    """sorted pair (diff.py:533-539)"""
    if p2 is None:
        p1, p2 = p1
    return (p2, p1) if p1 > p2 else (p1, p2)


def parse_partials(order1, order2, parameters1, parameters2):
    """normalise `order1` to {variable: {parameter: coefficient}} and `order2` to
    {(variable, variable): {parameter: second-order coefficient}}  (diff.py:153-262);
    returns (order1, order2, auto_cross_derivatives)"""
    parameters = set(parameters1)
    pairs_allowed = {Pair(p) for p in parameters2}
    auto_cross = isinstance(order2, (bool, str)) or all(isinstance(item, str) for item in order2)
    if (not order1) and isinstance(order2, (bool, str)):
        order1 = order2
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
    if not order2:
        return order1, {}, auto_cross
    if not order1:
        raise ValueError("order1 must be set.")
    if order2 is True:
        order2 = {pair: {} for pair in pairs_allowed}
    elif isinstance(order2, str):
        order2 = {(order2, order2): {}}
    elif all(isinstance(param, str) for param in order2):
        order2 = {Pair(a, b): {} for a in order2 for b in order2}
    elif not isinstance(order2, dict) and all(isinstance(pair, tuple) for pair in order2):
        order2 = {Pair(pair): {} for pair in order2}
    elif isinstance(order2, dict) and all(isinstance(pair, tuple) and isinstance(order2[pair], dict) for pair in order2):
        order2 = {Pair(pair): dict(order2[pair]) for pair in order2}
    else:
        raise ValueError(f"Invalid parameter 'order2' value: {order2}")
    invalid = {pair for pair in order2 if not (set(pair) & set(order1))}
    if invalid:
        raise ValueError(f"Invalid variable pair(s), no match in order1 variables: {invalid}")
    invalid = {pair for pair in order2 if (set(pair) - set(order1)) and order2[pair]}
    if invalid:
        raise ValueError(f"Invalid variable pair(s), expecting no coefficient: {invalid}")
    invalid = {param for pair in order2 for param in (set(order2[pair]) - parameters)}
    if invalid:
        raise ValueError(f"Unknown parameter(s) in order2: {invalid}")
    return order1, order2, auto_cross


def parse_order1(order1, order2, parameters):
    """first-order part only (kept for callers that never take order2)"""
    if order2:
        raise NotImplementedError("second-order derivatives (order2) need an operator with PARAMETERS_ORDER2")
    return parse_partials(order1, False, parameters, ())[0]


class DiffMixin:
    """operators that know their partial derivatives: T (alpha, phi), E (tau, T1, T2, g),
    P (tau, g), R (rT, rL, r0)"""

    PARAMETERS_ORDER1 = set()
    PARAMETERS_ORDER2 = set()
    order1 = {}
    order2 = {}
    auto_cross_derivatives = True

    def _init_partials(self, kwargs):
        """pops order1 / order2 from the constructor keywords"""
        self.order1, self.order2, self.auto_cross_derivatives = parse_partials(
            kwargs.pop("order1", False), kwargs.pop("order2", False), self.PARAMETERS_ORDER1, self.PARAMETERS_ORDER2)
        self._dtables = None
        self._daxes = None

    @property
    def parameters_order2(self):
        """pairs of parameters whose second derivatives this operator needs (diff.py:87-97)"""
        allowed = {Pair(p) for p in self.PARAMETERS_ORDER2}
        return {Pair(p1, p2) for v1, v2 in self.order2 for p1 in self.order1.get(v1, []) for p2 in self.order1.get(v2, [])
                if Pair(p1, p2) in allowed}

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
            enc.add_partials({var: ("entry", enc.partial_table(self, var)) for var in self.order1})

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

    def _param_ops(self, second=False):
        """{parameter (pair): plain operator applying dOp/dparam (d2Op/dparam pair)} from the raw arrays"""
        cache = "_pops2" if second else "_pops1"
        if getattr(self, cache, None) is None:
            from . import opmatrix, opscalar
            raw = self._raw_partials2() if second else self._raw_partials1()
            ops = {}
            for key, (arr, arr0) in raw.items():
                name = f"d{self.name}/d{key}"
                if isinstance(self, opmatrix.MatrixOp):    # (a [2, 3, 3] array is a grid of diagonals for E)
                    ops[Pair(key) if second else key] = opmatrix.MatrixOp(arr, arr0, axes=self._daxes, check=False, name=name)
                else:
                    ops[Pair(key) if second else key] = opscalar.ScalarOp(arr, arr0, axes=self._daxes, check=False, name=name)
            setattr(self, cache, ops)
        return getattr(self, cache)

    def _derive(self, state, key, second=False):
        """(dOp/dparam)(state) as a new derivative state (no equilibrium)  (diff.py:103-117)"""
        from .plan import apply_operators
        op = self._param_ops(second)[key]
        out = apply_operators(op.prepare(state, inplace=False), [op])
        out._state.zero_density()
        return out

    def _apply_order2(self, sm, order1, order2, inplace):
        """second-order recurrence, operator by operator (diff.py:290-379)"""
        from .plan import apply_operators
        new = {}
        for pair, dsm in order2.items():             # (a, b) and (b, a) are the same object: once
            if Pair(pair) not in new:
                new[Pair(pair)] = apply_operators(self.prepare(dsm, inplace=inplace), [self])

        def add(pair, state, coeff):
            if np.ndim(coeff):
                raise NotImplementedError("array-valued coefficients in order2")
            if pair in new:
                _accumulate(new[pair], state, float(coeff))
            else:
                if float(coeff) != 1.0:
                    state._state.axpy(state._state, float(coeff) - 1.0)
                new[pair] = state

        # second-order coefficients of the parameters w.r.t. the variable pairs
        for pair, coeffs in self.order2.items():
            for param, coeff in coeffs.items():
                add(pair, self._derive(sm, param), coeff)
        # second partials of the operator (a dict, as in the reference: a parameter pair counts once)
        available = self.parameters_order2
        for v1, v2 in self.order2:
            coeffs = {Pair(p1, p2): c1 * c2 for p1, c1 in self.order1.get(v1, {}).items()
                      for p2, c2 in self.order1.get(v2, {}).items()}
            for pp, coeff in coeffs.items():
                if pp in available:
                    add(Pair(v1, v2), self._derive(sm, pp, second=True), coeff)
        # cross terms: first partials of the operator applied to the first-order states
        if self.auto_cross_derivatives:
            vars_cross = {Pair(v1, v2) for v1 in self.order1 for v2 in order1}
        else:
            vars_cross = set(self.order2)
        for keep in (lambda a, b: a >= b, lambda a, b: a <= b):
            for v1 in order1:
                for v2 in self.order1:
                    if Pair(v1, v2) in vars_cross and keep(v1, v2):
                        for p1, c1 in self.order1[v2].items():
                            add(Pair(v1, v2), self._derive(order1[v1], p1), c1)
        for pair in list(new):
            if pair[0] != pair[1]:
                new[pair[::-1]] = new[pair]
        return new

    def __call__(self, sm, *, inplace=False):
        """sm <- Op(sm), and for every derivative state  dS_v <- Op(dS_v) + (dOp/dv)(S)"""
        from .plan import apply_operators
        previous = getattr(sm, "order1", None) or {}
        previous2 = getattr(sm, "order2", None) or {}
        if not previous and not self.order1 and not previous2 and not self.order2:
            return super().__call__(sm, inplace=inplace)
        sm = self.prepare(sm, inplace=inplace)
        order2 = self._apply_order2(sm, previous, previous2, inplace) if (previous2 or self.order2) else {}
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
        sm.order2 = order2
        return sm

    def combine(self, other, **kwargs):
        if self.order1 or getattr(other, "order1", None):
            raise NotImplementedError("combining (@) operators that carry order1 derivatives")
        return super().combine(other, **kwargs)


def _accumulate(dsm, part, alpha=1.0):
    """dsm += alpha * part on the device (same grid and capacity first)"""
    grid = common.broadcast_shapes(dsm.shape, part.shape, append=True)
    dsm._broadcast_to(grid)
    part._broadcast_to(grid)
    K = max(dsm._state.K, part._state.K)
    dsm._reserve(K)
    part._reserve(K)
    dsm._nstate = max(dsm._nstate, part._nstate)
    dsm._state.axpy(part._state, alpha, zero_density=True)


def propagate_plain(op, sm, order1, inplace):
    """operators without parameters of their own that still act on derivative states (S);
    keys that share one state (the symmetric entries of order2) are applied once"""
    from .plan import apply_operators
    done, out = {}, {}
    for var, dsm in order1.items():
        if id(dsm) not in done:
            done[id(dsm)] = apply_operators(op.prepare(dsm, inplace=inplace), [op])
        out[var] = done[id(dsm)]
    return out


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
    """probe of the signal's second derivatives: [..., len(variables1), len(variables2)]  (diff.py:419-472);
    evaluated operator by operator (simulate() takes the stepwise path), "magnitude" rows / columns
    give the first derivatives"""

    def __init__(self, variables1, variables2=None, *, probe="F0"):
        self.probe = probe
        if not isinstance(variables1, list):
            variables1 = [variables1]
        if not variables2:
            variables2 = variables1
        elif not isinstance(variables2, list):
            variables2 = [variables2]
        self.variables1, self.variables2 = variables1, variables2
        _probe.operator.Operator.__init__(self, name=None)
        self._post = None

    def __repr__(self):
        return f"Hessian({self.probe})"

    def _device_kind(self):
        return None

    def _acquire(self, sm):
        order1 = getattr(sm, "order1", None) or {}
        order2 = getattr(sm, "order2", None) or {}
        missing = np.zeros(sm.shape)
        rows = []
        for v1 in self.variables1:
            row = []
            for v2 in self.variables2:
                if v1 == "magnitude":
                    src = order1.get(v2)
                elif v2 == "magnitude":
                    src = order1.get(v1)
                else:
                    src = order2.get(Pair(v1, v2))
                row.append(np.asarray(getattr(src, self.probe)) if src is not None else missing)
            rows.append(np.stack(row, axis=-1))
        return np.stack(rows, axis=-2)


class PartialsPruner:
    """callback for `simulate(..., callback=PartialsPruner(...))`: once a derivative state has decayed it is dropped from
    `sm.order1` / `sm.order2`, and the operators that follow stop propagating it (the reference's functor of the same name,
    diff.py:478-528; out of the device hot path: it works on the stepwise path, one operator per call).

    condition: a threshold on the state matrix norm (default 1e-5), or a predicate `condition(state_matrix) -> bool array`
    variables: restrict the pruning to these variables (second-order pairs qualify when either member is listed)"""

    def __init__(self, *, condition=1e-5, variables=None):
        self.variables = frozenset(variables) if variables else None
        if callable(condition):
            self._negligible = condition
        elif common.isscalar(condition):
            limit = condition
            self._negligible = lambda dsm: dsm.norm < limit
        else:
            raise TypeError(condition)

    def _wanted(self, key):
        """is this entry (a variable name, or a pair of names) subject to pruning?"""
        if self.variables is None:
            return True
        names = key if isinstance(key, tuple) else (key,)
        return any(name in self.variables for name in names)

    def _prune(self, states):
        for key in [key for key in states if self._wanted(key)]:
            if np.all(self._negligible(states[key])):
                del states[key]

    def __call__(self, sm):
        order1 = getattr(sm, "order1", None)
        if not order1:
            return
        self._prune(order1)
        order2 = getattr(sm, "order2", None)
        if order2:
            self._prune(order2)

    def __repr__(self):
        return f"PartialsPruner({len(self.variables)} variables)" if self.variables else "PartialsPruner(all variables)"
