"""Plan compiler: a flat list of operators -> the arrays `epgx_plan_create` wants.

This replaces the Python `for op in sequence` loop of the reference
(epgpy/functions.py:173-192) by a one-off compilation:

  * every operator contributes one record (opcode, shift / slot, table reference);
  * an operator's coefficient table `[*opshape, ncoef]` is stored ONCE in a float64 pool
    however often the operator object appears (the reference's "reuse operators" idiom,
    docs/basics.md:111), and however large the grid is: the kernel indexes it through an
    *index space* -- strides over the grid axes with stride 0 on axes the operator does not
    depend on.  This is the reference's append-trailing-axes broadcasting
    (common.py:273-303) evaluated lazily on the device instead of materialised by NumPy;
  * the logical number of states n is tracked exactly as `S._apply` does
    (shift.py:86, :98): n <- min(n + |k|, nmax), which fixes the device capacity K.
"""
import numpy as np

from . import common, _lib

NO_TRUNCATION = 1 << 30


class Encoder:
    def __init__(self, grid_shape, options=None, nstate0=0):
        self.grid = tuple(int(d) for d in grid_shape)
        self.options = dict(options or {})
        self.records = []          # (opcode, space, ia, ib, coef_off, ncoef)
        self.spaces = []           # strides tuples
        self.tables = {}           # key -> (space, offset, ncoef)
        self.pool = []
        self.pool_size = 0
        self.nstate = int(nstate0)
        self.peak = int(nstate0)
        self.n_adc = 0

    # -- tables ------------------------------------------------------------------------
    def _space_of(self, opshape):
        if len(opshape) > len(self.grid):
            raise ValueError(f"Operator shape {opshape} has more axes than the grid {self.grid}")
        strides, acc = [0] * len(self.grid), 1
        for d in reversed(range(len(opshape))):
            if opshape[d] == 1:
                continue
            if opshape[d] != self.grid[d]:
                raise ValueError(f"Incompatible shapes: operator {opshape}, grid {self.grid}")
            strides[d] = acc
            acc *= opshape[d]
        if not any(strides):
            return -1
        strides = tuple(strides)
        if strides not in self.spaces:
            if len(self.spaces) == _lib.MAX_SPACES:
                raise NotImplementedError(
                    f"more than {_lib.MAX_SPACES} distinct operator broadcast patterns in one plan")
            self.spaces.append(strides)
        return self.spaces.index(strides)

    def _table(self, table, key):
        if key is not None and key in self.tables:
            return self.tables[key]
        table = np.ascontiguousarray(table, dtype=np.float64)
        opshape, ncoef = table.shape[:-1], table.shape[-1]
        space = self._space_of(opshape)
        entry = (space, self.pool_size, ncoef)
        self.pool.append(table.reshape(-1))
        self.pool_size += table.size
        if key is not None:
            self.tables[key] = entry
        return entry

    # -- records -----------------------------------------------------------------------
    def add(self, opcode, *, table=None, key=None, ia=0, ib=0):
        space, off, ncoef = (-1, 0, 0) if table is None else self._table(table, key)
        if ncoef != _lib.NCOEF.get(opcode, 0):
            raise ValueError(f"opcode {opcode}: table has {ncoef} coefficients")
        self.records.append((opcode, space, int(ia), int(ib), off, ncoef))

    def add_shift(self, k, nmax):
        new = self.nstate + abs(int(k))
        if nmax:
            new = min(new, int(nmax))
        self.nstate = new
        self.peak = max(self.peak, new)
        self.add(_lib.OP_S, ia=int(k), ib=int(nmax) if nmax else NO_TRUNCATION)

    def add_adc(self, kind=0):
        slot = self.n_adc
        self.n_adc += 1
        self.add(_lib.OP_ADC, ia=slot, ib=int(kind))
        return slot

    # -- output ------------------------------------------------------------------------
    @property
    def nvox(self):
        return int(np.prod(self.grid))

    def capacity(self, at_least=0):
        need = max(self.peak + 1, int(at_least), 1)
        for K in _lib.SUPPORTED_K:
            if K >= need:
                return K
        raise NotImplementedError(
            f"{need} phase states per voxel exceed the device capacity {_lib.SUPPORTED_K[-1]}; "
            "bound the state matrix with max_nstate=...")

    def arrays(self):
        ops = np.zeros(max(len(self.records), 1), dtype=_lib.OP_DTYPE)
        for i, (opcode, space, ia, ib, off, ncoef) in enumerate(self.records):
            ops[i] = (opcode, space, ia, ib, off, ncoef, 0)
        if not self.records:
            ops[0] = (_lib.OP_NOP, -1, 0, 0, 0, 0, 0)
        coef = np.concatenate(self.pool) if self.pool else np.zeros(0)
        return ops, np.asarray(self.grid, dtype=np.int64), list(self.spaces), coef

    def device_plan(self, ctx):
        ops, grid, spaces, coef = self.arrays()
        return _lib.DevicePlan(ctx, ops, grid, spaces, coef, self.n_adc)


def apply_operators(sm, ops):
    """op(sm) for one or several operators: one launch of the fused kernel, state streamed
    HBM -> registers -> HBM once (the per-timestep mode of DESIGN.md)."""
    grid = common.broadcast_shapes(sm.shape, *[op.shape for op in ops], append=True)
    enc = Encoder(grid, options=sm.options, nstate0=sm.nstate)
    for op in ops:
        op._encode(enc)
    if not enc.records:
        return sm
    sm._broadcast_to(grid)
    sm._reserve(enc.capacity(at_least=sm.nstate + 1))
    plan = enc.device_plan(sm._ctx)
    _lib.run(sm._ctx, plan, 0, plan.n_ops, 0, plan.nvox, sm._state, sm._state, sm._state.K, None, 0, 0)
    sm._nstate = enc.nstate
    return sm
