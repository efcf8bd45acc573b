"""Device-resident StateMatrix (replaces the storage of epgpy/statematrix.py:9-373).

The reference keeps `states[*grid, 2n+1, 3]` complex128 in host (or cupy) memory and
re-allocates it with np.pad at every shift (statematrix.py:293-297, :654-676).  Here the
state lives in HBM as the half representation `[nvox][3][K]` (k >= 0 only, K = fixed
capacity, component-major so each component of a voxel is one coalesced line); the logical
number of states `nstate` is host bookkeeping, so growing costs nothing until K is exceeded.
The reference layout is rebuilt on demand by `.states` (mirror: row(-k) = conj(row(k)[[1,0,2]]),
statematrix.py:416-421).
"""
import math

import numpy as np

from . import common, _lib


def _format_states(states, check=True):
    """same shape rules and symmetry checks as statematrix.py:388-422"""
    states = np.asarray(states).astype(np.complex128)
    if states.ndim == 1:
        if check and states.size != 3:
            raise ValueError("The number of state dimensions must be 3")
        states = states.reshape((1, 1, 3))
    elif states.ndim == 2:
        if check and states.shape[1] != 3:
            raise ValueError("The number of state dimensions must be 3")
        elif check and states.shape[0] % 2 != 1:
            raise ValueError("The number of states must be odd")
        states = states.reshape((1,) + states.shape)
    else:
        if check and states.shape[-1] != 3:
            raise ValueError("The number of state dimensions must be 3")
        elif check and states.shape[-2] % 2 != 1:
            raise ValueError("The number of states must be odd")
    if check:
        if not np.allclose(states[..., 1], states[..., ::-1, 0].conj()):
            raise ValueError("The F-state columns do no match.")
        if not np.allclose(states[..., 2], states[..., ::-1, 2].conj()):
            raise ValueError("The Z-state columns is not symmetrical.")
    return states


def _to_grid(arr, grid, tail):
    """broadcast [*lead, *tail_axes] to [*grid, *tail_axes], appending missing grid axes"""
    lead = arr.shape[: arr.ndim - tail]
    tshape = arr.shape[arr.ndim - tail:]
    arr = arr.reshape(lead + (1,) * (len(grid) - len(lead)) + tshape)
    return np.broadcast_to(arr, tuple(grid) + tshape)


def _capacity_for(nstate):
    for K in _lib.SUPPORTED_K:
        if K >= nstate + 1:
            return K
    raise NotImplementedError(
        f"nstate={nstate} exceeds the device capacity of {_lib.SUPPORTED_K[-1] - 1} orders per voxel")


def fold(states, K):
    """[*grid, 2n+1, 3] -> [nvox, 3, K] (rows k >= 0, zero padded)"""
    n = (states.shape[-2] - 1) // 2
    grid = states.shape[:-2]
    half = np.zeros((int(np.prod(grid)), 3, K), dtype=np.complex128)
    pos = states[..., n:, :].reshape(-1, n + 1, 3)
    half[:, :, : n + 1] = np.moveaxis(pos, -1, -2)
    return half


def unfold(half, grid, nstate):
    """[nvox, 3, K] -> [*grid, 2n+1, 3] by mirroring the k < 0 rows"""
    K = half.shape[-1]
    m = min(nstate + 1, K)
    pos = np.zeros((half.shape[0], nstate + 1, 3), dtype=np.complex128)
    pos[:, :m, :] = np.moveaxis(half[:, :, :m], -2, -1)
    neg = pos[:, :0:-1, :][..., [1, 0, 2]].conj()
    return np.concatenate([neg, pos], axis=-2).reshape(tuple(grid) + (2 * nstate + 1, 3))


def _is_plain_equilibrium(equilibrium):
    """True for a formatted equilibrium [.., 2 n + 1, 3] of the form every kernel carries by itself: Z_0 = density (real),
    every other coefficient zero (the reference's own tests and examples use no other)"""
    neq = (equilibrium.shape[-2] - 1) // 2
    return (not np.any(equilibrium[..., :2] != 0) and not np.any(np.delete(equilibrium[..., 2], neq, axis=-1) != 0)
            and not np.any(equilibrium[..., neq, 2].imag != 0))


def _equilibrium_density(equilibrium):
    """the density of an equilibrium of the plain form (real part of Z_0, statematrix.py:93-95)"""
    neq = (equilibrium.shape[-2] - 1) // 2
    if not _is_plain_equilibrium(equilibrium):
        raise NotImplementedError("only equilibria of the form [0, 0, density] are supported here")
    return equilibrium[..., neq, 2].real


class StateMatrix:
    """n-dimensional phase-state matrix stored on the GPU (statematrix.py:9-80)"""

    def __init__(self, init=None, *, density=1, equilibrium=None, coords=None, kvalue=1.0,
                 tvalue=1.0, nstate=None, shape=None, check=True, device=None, **options):
        if equilibrium is None:
            dens = np.atleast_1d(np.asarray(density, dtype=np.float64))
            equilibrium = np.zeros(dens.shape + (1, 3), dtype=np.complex128)
            equilibrium[..., 0, 2] = dens
        equilibrium = _format_states(equilibrium, check=check)
        # A GENERAL equilibrium (transverse or k != 0 coefficients, statematrix.py:56-59) lives in a second device-resident state
        # matrix; the state itself then carries density 0, so no kernel adds a recovery term, and `plan.apply_operators` adds
        # `arr0 * equilibrium` from the second matrix (operator-by-operator runs: stepwise mode of simulate, op(sm))
        general = None if _is_plain_equilibrium(equilibrium) else equilibrium
        dens = np.zeros(equilibrium.shape[:-2]) if general is not None else _equilibrium_density(equilibrium)
        init = equilibrium if init is None else _format_states(init, check=check)

        n = (init.shape[-2] - 1) // 2
        if nstate:
            n_new = int(nstate)
            if n_new >= n:
                pad = [(0, 0)] * (init.ndim - 2) + [(n_new - n, n_new - n), (0, 0)]
                init = np.pad(init, pad)
            else:
                init = init[..., n - n_new: n + n_new + 1, :]
            n = n_new
        grid = common.broadcast_shapes(init.shape[:-2], dens.shape, append=True)
        if shape:
            grid = common.broadcast_shapes(grid, tuple(shape), append=True)
        init = _to_grid(init, grid, 2)
        dens = _to_grid(dens, grid, 0)

        self._ctx = _lib.get_context(device)
        self._shape = tuple(int(d) for d in grid)
        self._nstate = n
        self._state = _lib.DeviceState(self._ctx, int(np.prod(grid)), _capacity_for(max(n, options.get("max_nstate") or 0)))
        self._state.upload(fold(init, self._state.K), np.ascontiguousarray(dens, dtype=np.float64).reshape(-1))
        self._eq = self._eq_host = None
        if general is not None:
            self._set_equilibrium(_to_grid(general, grid, 2))
        self.kvalue, self.tvalue = kvalue, tvalue
        self.options = options
        # k-space coordinate set once an n-D shift has been applied (kspace.py), or handed over with the states
        self._kspace = None if coords is None else self._planned_coords(coords)

    def _planned_coords(self, coords):
        """`coords=` [.., 2 nstate + 1, kdim] as a planner state: integer, symmetric, sorted rows (what `sm.coords` of an
        earlier run returns; anything else cannot have come from a shift and is rejected), all rows taken as populated"""
        from . import kspace
        ks = kspace.KSpace.from_coords(coords)
        if ks.nstate != self._nstate:
            raise ValueError(f"coords: {ks.nrow} rows for a state matrix with nstate={self._nstate}")
        if not common.broadcastable(self._shape, ks.lead or (1,), append=True):
            raise ValueError(f"coords: leading shape {ks.lead} does not fit the state matrix {self._shape}")
        return ks

    # -- device plumbing ---------------------------------------------------------------
    @classmethod
    def _wrap(cls, ctx, state, shape, nstate, options=None, kvalue=1.0, tvalue=1.0):
        sm = cls.__new__(cls)
        sm._ctx, sm._state, sm._shape, sm._nstate = ctx, state, tuple(shape), int(nstate)
        sm.options, sm.kvalue, sm.tvalue = dict(options or {}), kvalue, tvalue
        sm._kspace = None
        sm._eq = sm._eq_host = None
        return sm

    def _set_equilibrium(self, eq):
        """a general equilibrium [*grid, 2 m + 1, 3] on this matrix's grid: kept on the host (what `.equilibrium` returns) and
        as a device-resident state matrix of the state's capacity (density 0)"""
        self._eq_host = np.array(eq, dtype=np.complex128)
        m = (eq.shape[-2] - 1) // 2
        self._reserve(_capacity_for(m))
        self._eq = _lib.DeviceState(self._ctx, self.size, self._state.K)
        self._eq.upload(fold(self._eq_host, self._state.K), np.zeros(self.size))

    def _equilibrium_matrix(self):
        """a copy of the general equilibrium as a state matrix on this grid (density 0: relaxing it recovers nothing)"""
        return StateMatrix._wrap(self._ctx, self._eq.copy(), self._shape, (self._eq_host.shape[-2] - 1) // 2, {}, self.kvalue, self.tvalue)

    def _reset_to_equilibrium(self):
        """RESET with a general equilibrium (operator.py:297-304: states <- equilibrium, then every array cropped to order 0)"""
        m = (self._eq_host.shape[-2] - 1) // 2
        centre = self._eq_host[..., m:m + 1, :]
        self._set_equilibrium(centre)
        self._state = self._eq.copy()
        self._nstate = 0
        self._kspace = None

    def _reserve(self, K):
        if K > self._state.K:
            self._state = self._state.copy(K)
        if getattr(self, "_eq", None) is not None and K > self._eq.K:
            self._eq = self._eq.copy(K)

    def _broadcast_to(self, grid):
        grid = tuple(int(d) for d in grid)
        if grid == self._shape:
            return
        src = np.arange(self.size, dtype=np.int64).reshape(self._shape + (1,) * (len(grid) - self.ndim))
        index = np.broadcast_to(src, grid).reshape(-1)
        if index.size != self.size or len(grid) != self.ndim:
            if index.size != self.size:
                self._state = self._state.broadcast(index.astype(np.int32))
                if self._eq is not None:
                    self._eq = self._eq.broadcast(index.astype(np.int32))
                    self._eq_host = np.broadcast_to(self._eq_host.reshape(self._shape + (1,) * (len(grid) - self.ndim) + self._eq_host.shape[-2:]),
                                                    grid + self._eq_host.shape[-2:]).copy()
        if self._eq_host is not None and self._eq_host.shape[:-2] != grid:
            self._eq_host = self._eq_host.reshape(grid + self._eq_host.shape[-2:])
        self._shape = grid

    def _download(self):
        half, dens = self._state.download()
        return half, dens

    # -- public attributes (statematrix.py:84-230) -----------------------------------------
    @property
    def states(self):
        half, _ = self._download()
        return unfold(half, self._shape, self._nstate)

    @states.setter
    def states(self, value):
        value = _format_states(value, check=False)
        while value.ndim - 2 > self.ndim and value.shape[0] == 1:  # leading singleton added by _format_states
            value = value.reshape(value.shape[1:])
        lead = value.shape[:-2]
        if not common.broadcastable(self._shape, lead, append=True):
            raise ValueError(f"states: leading shape {lead} does not fit the state matrix {self._shape}")
        grid = tuple(common.broadcast_shapes(self._shape, lead, append=True))
        if grid != self._shape:          # the new states widen the grid (statematrix.py:88-90: `arrays.set`): density follows
            if len(grid) > self.ndim:
                self.expand(len(grid))
            self._broadcast_to(grid)
        value = _to_grid(value, self._shape, 2)
        n = (value.shape[-2] - 1) // 2
        self._reserve(_capacity_for(n))
        _, dens = self._download()
        self._state.upload(fold(value, self._state.K), dens)
        self._nstate = n

    @property
    def density(self):
        if self._eq_host is not None:      # (statematrix.py:93-95: real part of the equilibrium's Z_0)
            return self._eq_host[..., (self._eq_host.shape[-2] - 1) // 2, 2].real.reshape(self._shape)
        _, dens = self._download()
        return dens.reshape(self._shape)

    @property
    def equilibrium(self):
        """equilibrium state matrix with this matrix's number of orders (statematrix.py:97-100: resized with the states)"""
        eq = np.zeros(self._shape + (2 * self._nstate + 1, 3), dtype=np.complex128)
        if self._eq_host is not None:
            m, n = (self._eq_host.shape[-2] - 1) // 2, self._nstate
            k = min(m, n)
            eq[..., n - k:n + k + 1, :] = self._eq_host[..., m - k:m + k + 1, :]
            return eq
        eq[..., self._nstate, 2] = self.density
        return eq

    @property
    def coords(self):
        """integer k-space coordinates after an n-D shift, else None: [1.., 2n+1, kdim], or [*lead, 1.., 2n+1, kdim]
        when a vectorised shift made them differ along the leading grid axes `lead`"""
        if self._kspace is None:
            return None
        ks = self._kspace
        return ks.coords.reshape(ks.lead + (1,) * (self.ndim - len(ks.lead)) + (ks.nrow, ks.kdim))

    @property
    def ndim(self):
        return len(self._shape)

    @property
    def shape(self):
        return self._shape

    @property
    def size(self):
        return math.prod(self._shape)

    @property
    def nstate(self):
        return self._nstate

    @property
    def kdim(self):
        return 1 if self._kspace is None else self._kspace.kdim

    @property
    def i0(self):
        return self._nstate

    def _component(self, c):
        half, _ = self._download()
        n, K = self._nstate, half.shape[-1]
        m = min(n + 1, K)
        pos = np.zeros((half.shape[0], n + 1), dtype=np.complex128)
        pos[:, :m] = half[:, c, :m]
        if c == 2:
            neg = pos[:, :0:-1].conj()
        else:
            other = np.zeros_like(pos)
            other[:, :m] = half[:, 1 - c, :m]
            neg = other[:, :0:-1].conj()
        return np.concatenate([neg, pos], axis=-1).reshape(self._shape + (2 * n + 1,))

    @property
    def F(self):
        return self._component(0)

    @property
    def Z(self):
        return self._component(2)

    @property
    def F0(self):
        half, _ = self._download()
        return half[:, 0, 0].reshape(self._shape).copy()

    F0t = F0

    @property
    def Z0(self):
        half, _ = self._download()
        return half[:, 2, 0].reshape(self._shape).copy()

    @property
    def k(self):
        coords = self.coords
        if coords is None:
            n = self._nstate
            coords = np.arange(-n, n + 1).reshape((1,) * self.ndim + (2 * n + 1, 1))
        kvalue = self.kvalue
        if not common.isscalar(kvalue):
            kvalue = np.asarray(kvalue)[: coords.shape[-1]]
        return coords[..., :3] * kvalue

    @property
    def t(self):
        return 0

    t0 = t

    @property
    def norm(self):
        """sqrt(sum |states[..., 1:]|^2) over orders and columns (utils.py:152-154)"""
        states = self.states
        return np.sqrt(np.sum(np.abs(states[..., 1:]) ** 2, axis=(-2, -1)))

    @property
    def zeros(self):
        return self.copy(np.zeros(self._shape + (2 * self._nstate + 1, 3)))

    @property
    def writeable(self):
        return True

    @property
    def array_module(self):
        return np

    def __repr__(self):
        return f"StateMatrix({self.shape}, nstate={self.nstate})"

    def __array__(self, dtype=None, copy=None):
        arr = self.states
        return arr if dtype is None else arr.astype(dtype)

    # -- arithmetic (host round trip; not a hot path) --------------------------------------
    def _cmp(self, other):
        if isinstance(other, StateMatrix):
            return other.states
        if np.isscalar(other):
            return other
        return np.asarray(other)[..., np.newaxis, np.newaxis]

    def __add__(self, other):
        return self.copy(self.states + self._cmp(other))

    __radd__ = __add__

    def __iadd__(self, other):
        self.states = self.states + self._cmp(other)
        return self

    def __mul__(self, other):
        return self.copy(self.states * self._cmp(other))

    __rmul__ = __mul__

    def __imul__(self, other):
        self.states = self.states * self._cmp(other)
        return self

    def __eq__(self, other):
        return self.states == self._cmp(other)

    __hash__ = object.__hash__

    # -- public functions (statematrix.py:276-312) -----------------------------------------
    def copy(self, states=None, **kwargs):
        equilibrium = kwargs.pop("equilibrium", None)
        coords = kwargs.pop("coords", None)
        kvalue = kwargs.pop("kvalue", self.kvalue)
        tvalue = kwargs.pop("tvalue", self.tvalue)
        new = StateMatrix._wrap(self._ctx, self._state.copy(), self._shape, self._nstate,
                                {**self.options, **kwargs}, kvalue, tvalue)
        new._kspace = self._kspace
        if self._eq is not None and equilibrium is None:
            new._eq, new._eq_host = self._eq.copy(), self._eq_host.copy()
        if equilibrium is not None and not _is_plain_equilibrium(_format_states(equilibrium, check=True)):
            eq = _format_states(equilibrium, check=True)
            if tuple(common.broadcast_shapes(self._shape, eq.shape[:-2] or (1,), append=True)) != self._shape:
                raise ValueError(f"equilibrium: leading shape {eq.shape[:-2]} does not fit the state matrix {self._shape}")
            half, _ = new._download()
            new._state.upload(half, np.zeros(self.size))
            new._set_equilibrium(_to_grid(eq, self._shape, 2))
        elif equilibrium is not None:      # (statematrix.py:282-283; the copy keeps this matrix's grid)
            dens = _equilibrium_density(_format_states(equilibrium, check=True))
            if (not common.broadcastable(self._shape, dens.shape or (1,), append=True)
                    or tuple(common.broadcast_shapes(self._shape, dens.shape or (1,), append=True)) != self._shape):
                raise ValueError(f"equilibrium: leading shape {dens.shape} does not fit the state matrix {self._shape}")
            half, _ = new._download()
            new._state.upload(half, np.ascontiguousarray(_to_grid(dens, self._shape, 0), dtype=np.float64).reshape(-1))
        if states is not None:
            new.states = states
        if coords is not None:
            new._kspace = new._planned_coords(coords)
        return new

    def resize(self, nstate):
        """symmetric zero pad / crop of the order axis (statematrix.py:293-297)"""
        nstate = int(nstate)
        if nstate == self._nstate:
            return
        if nstate > self._nstate:
            self._reserve(_capacity_for(nstate))
        else:
            # cropping must really drop the orders above nstate
            half, dens = self._download()
            half[:, :, nstate + 1:] = 0
            self._state.upload(half, dens)
        self._nstate = nstate

    def expand(self, ndim):
        diff = ndim - self.ndim
        if diff > 0:
            self._shape = self._shape + (1,) * diff

    def reduce(self, ndim):
        diff = self.ndim - ndim
        if diff > 0:
            if any(d != 1 for d in self._shape[ndim:]):
                raise ValueError("cannot remove axes of extent > 1")
            self._shape = self._shape[:ndim]

    def check(self):
        st = self.states
        return np.allclose(st, st[..., ::-1, [1, 0, 2]].conj())
