"""CPU oracle (NumPy) for the epgpy hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import
this module; the product package `epgpy_amd` never does (it fails loudly when the HIP
library is missing).

This is an independent restatement of the algorithm the reference (py-baudin/epgpy,
mounted at /root/reference in the build container) executes on its NumPy path for
`simulate()` over T / E / P / S / ADC (+ SPOILER, PD, RESET):

  * state matrix  : dense complex128 `states[*grid, 2n+1, 3]`, row r <-> order k = r - n,
                    col 0 = F_k, col 1 = conj(F_-k), col 2 = Z_k      (statematrix.py:55, :392)
  * equilibrium   : zero except `[n, 2] = density`                     (statematrix.py:379-385)
  * T(alpha, phi) : 3x3 matrix Rz(phi) Rx(alpha) Rz(-phi), degrees     (transition.py:114-151)
                    applied to every row                               (opmatrix.py:199-221)
  * E(tau,T1,T2,g): diagonal [conj(e^-rT), e^-rT, e^-rL] + recovery
                    [0,0,1-e^-rL]*equilibrium, rT = tau(1/T2 + 2 pi i g), rL = tau/T1
                                                                       (evolution.py:220-256,
                                                                        opscalar.py:213-232)
  * S(k) int      : grow to min(n+|k|, nmax) by symmetric zero padding, then shift col 0 up /
                    col 1 down by k rows with zero fill                 (shift.py:82-101, :271-294,
                                                                        statematrix.py:793-804)
  * ADC           : F0 = states[..., n, 0]                              (statematrix.py:148-151)
  * driver        : flatten, broadcast shapes with *appended* axes, loop (functions.py:50-192,
                                                                        common.py:273-303)

Sequences are described by plain tuples so that neither the reference nor the product
package is needed to run the oracle:

    ("T", alpha, phi) ("E", tau, T1, T2, g) ("P", tau, g) ("S", k) ("ADC",) ("ADC", "Z0")
    ("ADC", "F0", phase_deg) ("SPOILER",) ("RESET",) ("PD", pd) ("WAIT",)

Array-valued parameters follow the reference's convention: their axes are the *leading*
grid axes, missing axes are appended.

Pinned against the golden vectors in tests/golden/*.npz, which were produced by the
reference itself (tests/golden/make_golden.py), see tests/test_oracle.py.
"""
import numpy as np

__all__ = [
    "rotation_matrix", "relaxation_coeffs", "precession_coeffs", "shift_rows",
    "broadcast_append", "seq_shape", "simulate", "expand_half", "fold_half",
]


# ----------------------------------------------------------------------------- helpers
def _shape_of(x):
    return np.shape(x)


def broadcast_append(*shapes):
    """common shape with missing axes APPENDED (common.py:273-303)"""
    ndim = max([len(s) for s in shapes] + [1])
    out = [1] * ndim
    for s in shapes:
        s = tuple(s) + (1,) * (ndim - len(s))
        for i, d in enumerate(s):
            if d == 1:
                continue
            if out[i] not in (1, d):
                raise ValueError(f"Incompatible shapes: {shapes}")
            out[i] = d
    return tuple(out)


def _append_axes(arr, ndim):
    arr = np.asarray(arr)
    if arr.ndim == 0:
        return arr
    return arr.reshape(arr.shape + (1,) * (ndim - arr.ndim))


def _expand_params(*params):
    """give all array parameters the same ndim by appending axes (common.py:306-334)"""
    ndim = max([np.ndim(p) for p in params] + [0])
    return [_append_axes(p, ndim) for p in params]


# ----------------------------------------------------------------------------- coefficients
def rotation_matrix(alpha, phi):
    """RF rotation, angles in degrees  (transition.py:114-151)

    R = Rz(phi) @ Rx(alpha) @ Rz(-phi); evaluated as that triple product (same order of
    floating-point operations as the reference) so coefficients agree to the last bit.
    returns [..., 3, 3] complex128 with leading shape broadcast(alpha, phi) or (1,)
    """
    alpha, phi = _expand_params(alpha, phi)
    a = np.pi / 180.0 * np.atleast_1d(alpha)
    rx = np.empty(a.shape + (3, 3), dtype=np.complex128)
    c2, s2, s = np.cos(a / 2) ** 2, np.sin(a / 2) ** 2, np.sin(a)
    rx[..., 0, 0], rx[..., 0, 1], rx[..., 0, 2] = c2, s2, -1j * s
    rx[..., 1, 0], rx[..., 1, 1], rx[..., 1, 2] = s2, c2, 1j * s
    rx[..., 2, 0], rx[..., 2, 1], rx[..., 2, 2] = -1j / 2 * s, 1j / 2 * s, np.cos(a)

    def rz(p_deg):
        p = np.atleast_1d(p_deg) * np.pi / 180.0
        m = np.zeros(p.shape + (3, 3), dtype=np.complex128)
        m[..., 0, 0] = np.exp(1j * p)
        m[..., 1, 1] = np.exp(-1j * p)
        m[..., 2, 2] = 1
        return m

    return rz(phi) @ rx @ rz(-phi)


def _evolution(rT, rL, r0):
    rT, rL, r0 = _expand_params(rT, rL, r0)
    shape = np.broadcast_shapes(np.shape(rT), np.shape(rL), np.shape(r0), (1,))
    arr = np.zeros(shape + (3,), dtype=np.complex128)
    arr[..., 1] = np.exp(-rT)
    arr[..., 0] = arr[..., 1].conj()
    arr[..., 2] = np.exp(-rL)
    arr0 = np.zeros(shape + (3,), dtype=np.complex128)
    arr0[..., 2] = 1 - np.exp(-r0)
    return arr, arr0


def relaxation_coeffs(tau, T1, T2, g=0):
    """E(tau, T1, T2, g): (arr, arr0) of shape broadcast(...)+(3,)  (evolution.py:220-256)"""
    tau, T1, T2, g = _expand_params(tau, T1, T2, g)
    rT = tau * (1 / T2 + 2j * np.pi * g)
    rL = tau / T1
    return _evolution(rT, rL, rL)


def precession_coeffs(tau, g):
    """P(tau, g): precession only, no relaxation, no recovery  (evolution.py:245-248)"""
    tau, g = _expand_params(tau, g)
    arr, _ = _evolution(2j * np.pi * g * tau, 0 * np.real(tau), 0 * np.real(tau))
    return arr, None


# ----------------------------------------------------------------------------- state ops
def _pad_rows(states, n_new):
    """symmetric zero pad / crop of the order axis to 2*n_new+1 rows (statematrix.py:793-804)"""
    n = (states.shape[-2] - 1) // 2
    d = n_new - n
    if d > 0:
        pad = [(0, 0)] * (states.ndim - 2) + [(d, d), (0, 0)]
        return np.pad(states, pad)
    if d < 0:
        return states[..., -d:d, :].copy()
    return states


def shift_rows(states, k):
    """in-place integer shift of an already-sized state matrix  (shift.py:283-292)"""
    if k > 0:
        states[..., k:, 0] = states[..., :-k, 0].copy()
        states[..., :-k, 1] = states[..., k:, 1].copy()
        states[..., :k, 0] = 0
        states[..., -k:, 1] = 0
    elif k < 0:
        k = -k
        states[..., :-k, 0] = states[..., k:, 0].copy()
        states[..., k:, 1] = states[..., :-k, 1].copy()
        states[..., -k:, 0] = 0
        states[..., :k, 1] = 0
    return states


def apply_matrix(states, mat):
    """rows <- mat @ rows for every row (opmatrix.py:208-221)"""
    gnd = states.ndim - 2
    lead = mat.shape[:-2] + (1,) * (gnd - (mat.ndim - 2))
    m = mat.reshape(lead + (1, 3, 3))
    return np.matmul(m, states[..., None])[..., 0]


def apply_scalar(states, arr, arr0, equilibrium):
    """states*arr + arr0*equilibrium  (opscalar.py:213-232)"""
    gnd = states.ndim - 2
    lead = arr.shape[:-1] + (1,) * (gnd - (arr.ndim - 1))
    out = states * arr.reshape(lead + (1, 3))
    if arr0 is not None:
        out = out + arr0.reshape(lead + (1, 3)) * equilibrium
    return out


# ----------------------------------------------------------------------------- driver
def op_shape(op):
    kind = op[0]
    if kind == "T":
        return np.broadcast_shapes(*[np.shape(p) for p in _expand_params(op[1], op[2])], (1,))
    if kind == "E":
        return np.broadcast_shapes(*[np.shape(p) for p in _expand_params(*op[1:5])], (1,))
    if kind == "P":
        return np.broadcast_shapes(*[np.shape(p) for p in _expand_params(op[1], op[2])], (1,))
    if kind == "PD":
        return np.shape(np.atleast_1d(op[1]))
    return (1,)


def seq_shape(ops):
    """functions.py:14-17"""
    return broadcast_append(*[op_shape(op) for op in ops])


def simulate(ops, *, shape=None, max_nstate=None, init=None, density=1.0, return_states=False):
    """Run a tuple-described sequence; returns signal [n_adc, *grid] (and final states).

    Follows functions.py:50-192: start from equilibrium [0,0,density] with n = 0 (or from
    `init`, a full [*grid, 2n+1, 3] array), apply the operators in order, record at every ADC.
    """
    grid = broadcast_append(seq_shape(ops), tuple(shape) if shape else (1,))
    gnd = len(grid)
    density = np.broadcast_to(_append_axes(np.asarray(density, float), gnd), grid).astype(float)
    if init is None:
        states = np.zeros(grid + (1, 3), dtype=np.complex128)
        states[..., 0, 2] = density
    else:
        init = np.asarray(init, dtype=np.complex128)
        lead = init.shape[:-2] + (1,) * (gnd - (init.ndim - 2))
        states = np.broadcast_to(init.reshape(lead + init.shape[-2:]), grid + init.shape[-2:]).copy()

    def equilibrium(n):
        eq = np.zeros(grid + (2 * n + 1, 3), dtype=np.complex128)
        eq[..., n, 2] = density
        return eq

    signal = []
    for op in ops:
        kind = op[0]
        n = (states.shape[-2] - 1) // 2
        if kind == "T":
            states = apply_matrix(states, rotation_matrix(op[1], op[2]))
        elif kind == "E":
            arr, arr0 = relaxation_coeffs(*op[1:5]) if len(op) >= 5 else relaxation_coeffs(*op[1:4])
            states = apply_scalar(states, arr, arr0, equilibrium(n))
        elif kind == "P":
            arr, _ = precession_coeffs(op[1], op[2])
            states = apply_scalar(states, arr, None, None)
        elif kind == "S":
            k = int(op[1])
            nmax = max_nstate if max_nstate else (op[2] if len(op) > 2 and op[2] else None)
            n_new = n + abs(k) if nmax is None else min(n + abs(k), nmax)
            states = shift_rows(_pad_rows(states, n_new), k)
        elif kind == "ADC":
            what = op[1] if len(op) > 1 else "F0"
            val = states[..., n, 0] if what == "F0" else states[..., n, 2]
            val = np.array(val)
            if len(op) > 2 and op[2] is not None:
                ph = np.exp(1j * np.asarray(op[2]) / 180 * np.pi)
                val = val * _append_axes(ph, val.ndim)
            signal.append(val)
        elif kind == "SPOILER":
            states = states.copy()
            states[..., 0:2] = 0
        elif kind == "RESET":
            states = equilibrium(0)
        elif kind == "PD":
            pd = np.atleast_1d(np.asarray(op[1], float))
            density = np.broadcast_to(_append_axes(pd, gnd), grid).astype(float)
            if len(op) < 3 or op[2]:
                states = equilibrium(n)
        elif kind in ("WAIT", "NULL"):
            pass
        else:
            raise ValueError(f"unknown op {kind}")
    sig = np.asarray(signal) if signal else np.zeros((0,) + grid, complex)
    if return_states:
        return sig, states
    return sig


# ----------------------------------------------------------------------------- representations
def fold_half(states):
    """[..., 2n+1, 3] reference layout -> half representation [..., 3, n+1] (k >= 0):
    comp 0 = F_k, comp 1 = conj(F_-k), comp 2 = Z_k"""
    n = (states.shape[-2] - 1) // 2
    return np.ascontiguousarray(np.moveaxis(states[..., n:, :], -1, -2))


def expand_half(half, nstate=None):
    """inverse of fold_half: mirror k<0 rows  (statematrix.py:416-421 symmetry)
    row(-k) = conj(row(k)[[1,0,2]])"""
    half = np.asarray(half)
    K = half.shape[-1]
    n = K - 1 if nstate is None else nstate
    pos = np.moveaxis(half[..., : n + 1], -2, -1)  # [..., n+1, 3]
    if n + 1 > K:
        pad = [(0, 0)] * (pos.ndim - 2) + [(0, n + 1 - K), (0, 0)]
        pos = np.pad(pos, pad)
    neg = pos[..., :0:-1, :][..., [1, 0, 2]].conj()
    return np.concatenate([neg, pos], axis=-2)


# ----------------------------------------------------------------------------- first-order derivatives
# Restatement of the reference's derivative operators and of the order-1 recurrence of
# DiffOperator.__call__ / _apply_order1 (epgpy/diff.py:119-139, :264-288):
#     dS_var <- Op(dS_var, no equilibrium term)  +  sum_param coeff[var][param] * (dOp/dparam)(S)
#     S      <- Op(S)
# An operator tuple may carry a trailing dict {"order1": {var: {param: coeff}}} (the normalised
# form `_parse_partials` produces, diff.py:153-198).

def _rx_d(alpha):
    """d Rx / d alpha (per degree)  (transition.py:175-190)"""
    a = np.atleast_1d(alpha) * np.pi / 180
    rot = np.empty(a.shape + (3, 3), dtype=np.complex128)
    s, c = np.sin(a), np.cos(a)
    rot[..., 0, 0], rot[..., 0, 1], rot[..., 0, 2] = -0.5 * s, 0.5 * s, -1j * c
    rot[..., 1, 0], rot[..., 1, 1], rot[..., 1, 2] = 0.5 * s, -0.5 * s, 1j * c
    rot[..., 2, 0], rot[..., 2, 1], rot[..., 2, 2] = -1j / 2 * c, 1j / 2 * c, -s
    return rot * np.pi / 180


def _rz(p_deg):
    p = np.atleast_1d(p_deg) * np.pi / 180.0
    m = np.zeros(p.shape + (3, 3), dtype=np.complex128)
    m[..., 0, 0], m[..., 1, 1], m[..., 2, 2] = np.exp(1j * p), np.exp(-1j * p), 1
    return m


def _rz_d(p_deg):
    """d Rz / d phi (per degree)  (transition.py:193-200)"""
    p = np.atleast_1d(p_deg) * np.pi / 180.0
    m = np.zeros(p.shape + (3, 3), dtype=np.complex128)
    m[..., 0, 0], m[..., 1, 1] = 1j * np.exp(1j * p), -1j * np.exp(-1j * p)
    return m * np.pi / 180


def _rx(alpha):
    return rotation_matrix(alpha, 0 * np.asarray(alpha, float))


def rotation_partials(alpha, phi):
    """{'alpha': dR/dalpha, 'phi': dR/dphi}  (transition.py:160-172)"""
    alpha, phi = _expand_params(alpha, phi)
    rx = _rx(alpha)
    return {"alpha": _rz(phi) @ _rx_d(alpha) @ _rz(-phi),
            "phi": _rz_d(phi) @ rx @ _rz(-phi) - _rz(phi) @ rx @ _rz_d(-phi)}


def relaxation_partials(tau, T1, T2, g=0):
    """{'tau','T1','T2','g'} -> (darr, darr0)  (evolution.py:360-399)"""
    tau, T1, T2, g = _expand_params(tau, T1, T2, g)
    rT = tau * (1 / T2 + 2j * np.pi * g)
    rL = tau / T1
    out = {}
    mat, mat0 = _evolution(rT, rL, rL)
    mat[..., 1] *= -rT / tau
    mat[..., 0] = mat[..., 1].conj()
    mat[..., 2] *= -1 / T1
    mat0[..., 2] = -mat[..., 2]
    out["tau"] = (mat, mat0)
    mat, mat0 = _evolution(0 * rT, rL, rL)
    mat[..., :2] = 0
    mat[..., 2] *= tau / T1 ** 2
    mat0[..., 2] = -mat[..., 2]
    out["T1"] = (mat, mat0)
    mat, _ = _evolution(rT, 0 * rL, 0 * rL)
    mat[..., 0] *= tau / T2 ** 2
    mat[..., 1] *= tau / T2 ** 2
    mat[..., 2] = 0
    out["T2"] = (mat, None)
    mat, _ = _evolution(rT, 0 * rL, 0 * rL)
    mat[..., 1] *= -2j * np.pi * tau
    mat[..., 0] = mat[..., 1].conj()
    mat[..., 2] = 0
    out["g"] = (mat, None)
    return out


def precession_partials(tau, g):
    """evolution.py:314-329"""
    tau, g = _expand_params(tau, g)
    rT = 2j * np.pi * g * tau
    out = {}
    for name, factor in (("tau", -2j * np.pi * g), ("g", -2j * np.pi * tau)):
        mat, _ = _evolution(rT, 0 * np.real(rT), 0 * np.real(rT))
        mat[..., 1] *= factor
        mat[..., 0] = mat[..., 1].conj()
        mat[..., 2] = 0
        out[name] = (mat, None)
    return out


def evolution_partials(rT, rL, r0):
    """R(rT, rL, r0): evolution.py:263-285"""
    mat, _ = _evolution(rT, 0 * np.real(rL), 0 * np.real(rL))
    mat[..., 2] = 0
    out = {"rT": (-mat, None)}
    mat, _ = _evolution(0 * np.asarray(rT), rL, 0 * np.real(rL))
    mat[..., :-1] = 0
    out["rL"] = (-mat, None)
    mat, mat0 = _evolution(0 * np.asarray(rT), 0 * np.real(rL), r0)
    mat[:] = 0
    mat0[..., -1] -= 1
    out["r0"] = (mat, -mat0)
    return out


def simulate_jacobian(ops, variables, *, probe="F0", shape=None, max_nstate=None, through_plain=False):
    """Jacobian probe at every ADC: array [n_adc, *grid, len(variables)], 'magnitude' = the probe
    itself, unknown variables = 0  (diff.py:384-416).

    SPOILER (and D, PD without reset) are plain Operators in the reference (operator.py:95-104,
    :281-341): they update the state and leave the derivative states alone.  through_plain=True
    applies them to the derivative states as well (the build's `exact_partials` option).  A reset
    (RESET, PD(reset=True)) always clears the derivative states, see below."""
    plain = [op[:-1] if isinstance(op[-1], dict) else op for op in ops]
    grid = broadcast_append(seq_shape(plain), tuple(shape) if shape else (1,))
    gnd = len(grid)
    states = np.zeros(grid + (1, 3), dtype=np.complex128)
    states[..., 0, 2] = 1.0
    dstates = {}
    density = np.ones(grid)

    def equilibrium(n):
        eq = np.zeros(grid + (2 * n + 1, 3), dtype=np.complex128)
        eq[..., n, 2] = density
        return eq

    col = 0 if probe == "F0" else 2
    out = []
    for op, base in zip(ops, plain):
        order1 = op[-1].get("order1", {}) if isinstance(op[-1], dict) else {}
        kind = base[0]
        n = (states.shape[-2] - 1) // 2
        if kind == "T":
            mat = rotation_matrix(base[1], base[2])
            partials = {p: apply_matrix(states, m) for p, m in rotation_partials(base[1], base[2]).items()}
            apply = lambda st: apply_matrix(st, mat)
        elif kind in ("E", "P", "R"):
            if kind == "E":
                arr, arr0 = relaxation_coeffs(*base[1:])
                pp = relaxation_partials(*base[1:])
            elif kind == "P":
                arr, arr0 = precession_coeffs(base[1], base[2])
                pp = precession_partials(base[1], base[2])
            else:
                arr, arr0 = _evolution(base[1], base[2], base[3])
                pp = evolution_partials(base[1], base[2], base[3])
            eq = equilibrium(n)
            partials = {p: apply_scalar(states, d, d0, eq) for p, (d, d0) in pp.items()}
            apply = lambda st, arr=arr: apply_scalar(st, arr, None, None)
        elif kind == "S":
            k = int(base[1])
            n_new = n + abs(k) if not max_nstate else min(n + abs(k), max_nstate)
            states = shift_rows(_pad_rows(states, n_new), k)
            dstates = {v: shift_rows(_pad_rows(d, n_new), k) for v, d in dstates.items()}
            continue
        elif kind == "ADC":
            cols = []
            for var in variables:
                if var == "magnitude":
                    cols.append(states[..., n, col])
                elif var in dstates:
                    cols.append(dstates[var][..., n, col])
                else:
                    cols.append(np.zeros(grid, complex))
            out.append(np.stack([np.broadcast_to(c, grid) for c in cols], axis=-1))
            continue
        elif kind == "SPOILER":
            states = states.copy()
            states[..., 0:2] = 0
            if through_plain:
                for d in dstates.values():
                    d[..., 0:2] = 0
            continue
        elif kind in ("RESET", "PD"):
            if kind == "PD":
                density = np.broadcast_to(_append_axes(np.atleast_1d(np.asarray(base[1], float)), gnd), grid).astype(float)
            if kind == "RESET" or len(base) < 3 or base[2]:
                # the reference keeps the stale derivative states here and later broadcasts a 1-row
                # partial over all their rows (statematrix.py:257-259): undefined behaviour, not
                # restated -- the derivative states restart from zero
                states = equilibrium(0) if kind == "RESET" else equilibrium(n)
                dstates = {}
            continue
        else:
            raise ValueError(f"unsupported op {kind} in simulate_jacobian")
        new_d = {v: apply(d) for v, d in dstates.items()}
        for var, params in order1.items():
            for param, coeff in params.items():
                term = partials[param] * _append_axes(np.asarray(coeff), gnd)[..., None, None] \
                    if np.ndim(coeff) else partials[param] * coeff
                new_d[var] = new_d[var] + term if var in new_d else term
        dstates = {v: np.broadcast_to(d, grid + d.shape[-2:]).copy() for v, d in new_d.items()}
        states = apply(states) if kind == "T" else apply_scalar(states, arr, arr0, equilibrium(n))
    return np.asarray(out)


# ----------------------------------------------------------------------------- config 5: n-D integer shifts + diffusion
# Restatement of shiftnd / unique_1d (epgpy/shift.py:297-364, :461-475), StateMatrix.setup_coords / .k
# (statematrix.py:314-329, :177-186) and of the diffusion operator (diffusion.py:60-147).
# State = (states [*grid, R, 3], coords [R, kdim] or [*grid-like, R, kdim] int, or None): the coordinate set is sorted
# lexicographically (first component most significant) and symmetric about the centre row.  It is shared by all
# voxels until a shift comes with one vector per voxel (k of shape [*lead, kdim], shift.py:38-41): from then on
# every voxel has its own coordinates, while rows are still merged / sorted as whole [*grid-like, kdim] slices
# (np.unique along the row axis, shift.py:461-465).

def _mirror(states):
    states[..., 1] = states[..., ::-1, 0].conj()
    return states


def shift_nd(states, coords, delta, *, nmax=None, prune=True, tol=1e-8):
    """S(k) with an integer vector k on a coordinate-indexed state matrix  (shift.py:297-364).
    coords [R, kdim] and delta [kdim]: shared coordinates; coords [*lead, R, kdim] and / or delta [*lead, 1, kdim]
    (leading axes aligned with the grid, size-1 where shared): per-voxel coordinates"""
    coords = np.asarray(coords, dtype=np.int64)
    delta = np.asarray(delta, dtype=np.int64)
    if delta.ndim == 1:
        delta = delta.reshape(1, -1)
    n1 = coords.shape[-2]
    k_l = coords + 0 * delta
    cand = np.concatenate([k_l, k_l + delta, k_l - delta], axis=-2)
    uniq, inverse = np.unique(cand, axis=-2, return_inverse=True)
    inverse = inverse.reshape(-1)
    idx_l, idx_t = inverse[:n1], inverse[n1:2 * n1]
    keep_l = keep_t = np.ones(n1, dtype=bool)
    if nmax is not None:
        keep = np.any(np.all(np.abs(uniq) <= nmax, axis=-1), axis=tuple(range(uniq.ndim - 2)))
        if not keep.all():
            uniq = uniq[..., keep, :]
            remap = -np.ones(keep.size, dtype=np.int64)
            remap[keep] = np.arange(uniq.shape[-2])
            idx_l, idx_t = remap[idx_l], remap[idx_t]
            keep_l, keep_t = idx_l >= 0, idx_t >= 0
    new = np.zeros(states.shape[:-2] + (uniq.shape[-2], 3), dtype=np.complex128)
    new[..., idx_l[keep_l], 2] = states[..., keep_l, 2]
    new[..., idx_t[keep_t], 0] = states[..., keep_t, 0]
    _mirror(new)
    if prune:
        lead = tuple(range(new.ndim - 2))
        nonzero = ~np.all(np.isclose(new, 0, atol=tol), axis=lead + (new.ndim - 1,))
        nonzero[(uniq.shape[-2] - 1) // 2] = True
        new, uniq = new[..., nonzero, :], uniq[..., nonzero, :]
    if uniq.shape[-2] % 2 == 0:
        raise ValueError("asymmetrical state matrix")
    return new, uniq


def bmatrix(tau, k1, k2=None):
    """b-matrix of a linear change of wavenumber k1 -> k2 during tau  (diffusion.py:86-123);
    tau ms, k rad/m -> s/mm^2"""
    tau = tau * 1e-3
    k1 = np.atleast_2d(k1) * 1e-3
    b = k1[..., :, None] * k1[..., None, :] * tau
    if k2 is None:
        return b
    kd = np.atleast_2d(k2) * 1e-3 - k1
    if np.allclose(kd, 0):
        return b
    outer = lambda a, c: a[..., :, None] * c[..., None, :]
    return b + tau * (outer(k1, kd) / 2 + outer(kd, k1) / 2 + outer(kd, kd) / 3)


def diffusion_factors(tau, D, k, shift=None):
    """(DL, DT) per coordinate row: exp(-tr(bL D)), exp(-tr(bT D))  (diffusion.py:60-79, :126-147).
    D: scalar, [d, d] tensor, or an array over the grid given as ("field", values) (per-voxel
    isotropic diffusivity -- the build's extension; the reference takes scalars / tensors only)"""
    bL = bmatrix(tau, k)
    bT = bL if shift is None else bmatrix(tau, k - shift, k)
    if isinstance(D, tuple) and D[0] == "field":
        trL = np.trace(bL, axis1=-2, axis2=-1)
        trT = np.trace(bT, axis1=-2, axis2=-1)
        Dv = np.asarray(D[1], dtype=float)[..., None]
        return np.exp(-trL * Dv), np.exp(-trT * Dv)
    D = np.asarray(D, dtype=float)
    if D.ndim == 0:
        return (np.exp(-np.trace(bL, axis1=-2, axis2=-1) * D), np.exp(-np.trace(bT, axis1=-2, axis2=-1) * D))
    return np.exp(-np.sum(bL * D, axis=(-2, -1))), np.exp(-np.sum(bT * D, axis=(-2, -1)))


def simulate_nd(ops, *, kvalue=1.0, shape=None, max_nstate=None, prune=True, return_states=False):
    """sequences with integer n-D shifts ("S", [kx, ky, ...]) and diffusion ("D", tau, D[, k]);
    everything else as `simulate`.  Returns signal [n_adc, *grid] (and final (states, coords))"""
    def plain(op):
        if op[0] == "S" and not np.isscalar(op[1]):
            lead = np.shape(op[1])[:-1]
            return ("PD", np.ones(lead), False) if lead and lead != (1,) else ("S", 1)   # only its grid shape matters
        if op[0] == "D":
            D = op[2]
            return ("PD", D[1], False) if isinstance(D, tuple) else ("WAIT",)   # only its grid shape matters
        return op
    grid = broadcast_append(seq_shape([plain(op) for op in ops]), tuple(shape) if shape else (1,))
    gnd = len(grid)
    states = np.zeros(grid + (1, 3), dtype=np.complex128)
    states[..., 0, 2] = 1.0
    coords = None
    kvalue = np.atleast_1d(np.asarray(kvalue, dtype=float))
    tol = 1e-8 if prune in (True, False) else float(prune)

    def centre():
        return (states.shape[-2] - 1) // 2

    def wavenumbers():
        c = coords if coords is not None else np.arange(-centre(), centre() + 1)[:, None]
        kv = kvalue if kvalue.size == 1 else kvalue[: c.shape[-1]]
        return c[..., :3] * kv

    signal = []
    for op in ops:
        kind = op[0]
        n = centre()
        eq = np.zeros(grid + (states.shape[-2], 3), dtype=np.complex128)
        eq[..., n, 2] = 1.0
        if kind == "T":
            states = apply_matrix(states, rotation_matrix(op[1], op[2]))
        elif kind == "E":
            arr, arr0 = relaxation_coeffs(*op[1:])
            states = apply_scalar(states, arr, arr0, eq)
        elif kind == "S":
            k = op[1]
            nmax = max_nstate if max_nstate else (op[2] if len(op) > 2 and op[2] else None)
            if np.isscalar(k) and coords is None:
                n_new = n + abs(int(k)) if nmax is None else min(n + abs(int(k)), nmax)
                states = shift_rows(_pad_rows(states, n_new), int(k))
                continue
            kvec = np.atleast_1d(np.asarray([int(k)] if np.isscalar(k) else k, dtype=np.int64))   # [kdim'] or [*lead, kdim']
            if kvec.ndim == 2 and kvec.shape[0] == 1:
                kvec = kvec[0]
            if coords is None:      # setup_coords: the 1-D orders become the first component
                coords = np.zeros((2 * n + 1, kvec.shape[-1]), dtype=np.int64)
                coords[:, 0] = np.arange(-n, n + 1)
            kdim = max(coords.shape[-1], kvec.shape[-1])
            if kdim > coords.shape[-1]:
                coords = np.concatenate([coords, np.zeros(coords.shape[:-1] + (kdim - coords.shape[-1],), np.int64)], -1)
            delta = np.zeros(kvec.shape[:-1] + (kdim,), dtype=np.int64)
            delta[..., : kvec.shape[-1]] = kvec
            if delta.ndim > 1:      # one vector per point of the leading grid axes: align with the grid, keep a row axis
                delta = delta.reshape(delta.shape[:-1] + (1,) * (gnd - (delta.ndim - 1)) + (1, kdim))
                if coords.ndim == 2:
                    coords = coords.reshape((1,) * gnd + coords.shape)
            elif coords.ndim > 2:
                delta = delta.reshape((1,) * (gnd + 1) + (kdim,))
            states, coords = shift_nd(states, coords, delta, nmax=nmax, prune=bool(prune), tol=tol)
        elif kind == "D":
            tau, D = op[1], op[2]
            kvec = op[3] if len(op) > 3 and op[3] is not None else None
            k = wavenumbers()
            shift = None if kvec is None else np.atleast_1d(np.asarray(kvec, float)) * (kvalue if kvalue.size == 1 else kvalue[: len(np.atleast_1d(kvec))])
            if isinstance(D, tuple) and k.ndim > 2:      # per-voxel coordinates AND per-voxel diffusivity: align both with the grid
                vals = np.asarray(D[1], dtype=float)
                D = ("field", vals.reshape(vals.shape + (1,) * (gnd - vals.ndim)))
            DL, DT = diffusion_factors(tau, D, k, shift)
            if isinstance(D, tuple):      # per-voxel factors [*opshape, R] -> grid
                lead = DL.shape[:-1] + (1,) * (gnd - (DL.ndim - 1))
                DL, DT = DL.reshape(lead + DL.shape[-1:]), DT.reshape(lead + DT.shape[-1:])
            states = states.copy()
            states[..., 0] = DT * states[..., 0]
            states[..., 2] = DL * states[..., 2]
            _mirror(states)
        elif kind == "ADC":
            what = op[1] if len(op) > 1 else "F0"
            signal.append(np.array(states[..., n, 0] if what == "F0" else states[..., n, 2]))
        elif kind == "SPOILER":
            states = states.copy()
            states[..., 0:2] = 0
        else:
            raise ValueError(f"unknown op {kind} in simulate_nd")
    sig = np.asarray(signal)
    if return_states:
        return sig, (states, coords)
    return sig
