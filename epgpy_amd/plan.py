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
import os
import threading

import numpy as np

from . import common, _lib, kspace

NO_TRUNCATION = 1 << 30


class Encoder:
    def __init__(self, grid_shape, options=None, nstate0=0, kspace0=None):
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
        # n-D integer shifts: the coordinate set is planned on the host (kspace.py); None while
        # the sequence only uses the 1-D integer shift
        self.kspace = kspace0
        self.deferred = []         # (record index, builder(K) -> table) for tables whose size is 3*K
        # first-order derivatives: record index -> {variable: (table, key)}; `variables` = the (at
        # most MAX_VARS) names this plan propagates, every ADC then owns 1 + len(variables) rows
        self.partials = {}
        self.variables = []
        self.deriv_flags = 0
        # tables the library generates on the device (epgx_fuse): offsets in the generated part of
        # the pool are kept as -(offset + 1) until the size of the host part is final
        self.generated = {}        # key -> (space, tagged offset, ncoef)
        self.generated_size = 0
        self.fuses = []
        self.fuse_partials = []    # partials of generated tables (epgx_fuse_partial), same tagged offsets
        # tables the library assembles on the device from per-axis columns (epgx_assemble)
        self.assembles = []
        self._arrays_lock = threading.RLock()     # plan_arrays() mutates the encoder: one thread at a time

    # -- tables ------------------------------------------------------------------------
    def _strides_of(self, opshape):
        """index-space strides of a table of shape `opshape` (stride 0 = broadcast axis)"""
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
        return tuple(strides)

    def _dense_strides(self):
        return self._strides_of(self.grid)

    def _space_of(self, opshape):
        """(index space id, strides actually used).  The kernel knows MAX_SPACES broadcast patterns
        per plan; a further pattern is served by an existing space that varies along a superset of
        its axes (ultimately the dense grid, for which the last slot is kept), at the price of a
        materialised table"""
        strides = self._strides_of(opshape)
        if not any(strides):
            return -1, strides
        if strides in self.spaces:
            return self.spaces.index(strides), strides
        dense = self._dense_strides()
        free = _lib.MAX_SPACES - len(self.spaces)
        if strides == dense or free > (0 if dense in self.spaces else 1):
            self.spaces.append(strides)
            return len(self.spaces) - 1, strides
        supersets = [sp for sp in self.spaces if all(b != 0 for a, b in zip(strides, sp) if a != 0)]
        if supersets:
            best = min(supersets, key=lambda sp: int(np.prod([g for g, st in zip(self.grid, sp) if st])))
            return self.spaces.index(best), best
        self.spaces.append(dense)
        return len(self.spaces) - 1, dense

    def _table(self, table, key):
        if key is not None and key in self.tables:
            return self.tables[key]
        table = np.ascontiguousarray(table, dtype=np.float64)
        opshape, ncoef = table.shape[:-1], table.shape[-1]
        space, strides = self._space_of(opshape)
        if space >= 0 and strides != self._strides_of(opshape):
            # borrowed space: materialise the broadcast over the extra axes
            lead = tuple(opshape) + (1,) * (len(self.grid) - len(opshape))
            target = tuple(g if st else 1 for g, st in zip(self.grid, strides))
            table = np.ascontiguousarray(np.broadcast_to(table.reshape(lead + (ncoef,)), target + (ncoef,)))
        entry = (space, self.pool_size, ncoef)
        self.pool.append(table.reshape(-1))
        self.pool_size += table.size
        if key is not None:
            self.tables[key] = entry
        return entry

    def _generated(self, shape, ncoef, key, sources=()):
        """reserve a device-generated table [*shape, ncoef]; returns ((space, tagged offset, ncoef), new).
        `sources`: index spaces of the tables it is generated from -- when the plan ran out of index
        spaces a source may sit in a borrowed, larger space; the destination must then vary along
        every axis its sources vary along, i.e. move to the dense grid as well"""
        if key in self.generated:
            return self.generated[key], False
        space, strides = self._space_of(tuple(shape))
        for src in sources:
            if src >= 0 and any(a != 0 and b == 0 and g > 1 for a, b, g in zip(self.spaces[src], strides, self.grid)):
                dense = self._dense_strides()
                if dense not in self.spaces:
                    self.spaces.append(dense)      # _space_of keeps the last slot free for it
                space, strides = self.spaces.index(dense), dense
                break
        entries = 1 if space < 0 else int(np.prod([g for g, st in zip(self.grid, strides) if st]))
        entry = (space, -(self.generated_size + 1), ncoef)
        self.generated_size += entries * ncoef
        self.generated[key] = entry
        return entry, True

    ASSEMBLE_MIN_ENTRIES = 4096     # below this a table is simply uploaded

    def assembled_table(self, key, shape, groups, columns):
        """pool entry of a table [*shape, len(columns)] given as an outer combination of small column groups:
        groups[g] = array [*gshape, ncol] with gshape broadcastable to `shape` (leading axes, 1 = does not vary);
        columns[c] = (group, column in the group).  The host part of the pool only receives the groups; the
        library writes the full table on the device when the plan is created (epgx_assemble).  Returns None when
        that does not pay (small table, or a group as large as the table): the caller then ships the table"""
        if key in self.generated:
            return self.generated[key]
        entries = int(np.prod(shape))
        if (entries < self.ASSEMBLE_MIN_ENTRIES or len(groups) > _lib.MAX_ASM_SRC or len(columns) > _lib.MAX_ASM_COLS
                or max(int(np.prod(g.shape[:-1])) for g in groups) * 2 > entries):
            return None
        entry, _ = self._generated(tuple(shape), len(columns), key)
        if entry[0] < 0:
            return None
        sources = []
        for g in groups:
            g = np.ascontiguousarray(g, dtype=np.float64)
            gshape = tuple(g.shape[:-1])
            while len(gshape) > 1 and gshape[-1] == 1:      # trailing broadcast axes carry no stride
                gshape = gshape[:-1]
            sources.append((self.pool_size, g.shape[-1], self._strides_of(gshape)))
            self.pool.append(g.reshape(-1))
            self.pool_size += g.size
        self.assembles.append((entry[1], entry[0], len(columns), sources, [c[0] for c in columns], [c[1] for c in columns]))
        return entry

    def partial_table(self, op, var):
        """pool entry of d(op)/d(var), the table `op._variable_tables()[var]`: assembled on the device from per-axis column
        groups when the operator can name them and that pays (a relaxation partial over a (T1, T2) grid: KBs shipped
        instead of 34 MB per variable and simulate() call), otherwise uploaded -- once per operator object, variable and
        plan either way"""
        key = ("D1", id(op), var)
        if key in self.tables:
            return self.tables[key]
        if key in self.generated:
            return self.generated[key]
        if hasattr(op, "_partial_column_groups") and getattr(op, "_daxes", None) is None:
            lead = op.arr.shape[:-1]         # (the table of a scalar operator's partial has the operator's own shape, unless a
            if int(np.prod(lead)) >= self.ASSEMBLE_MIN_ENTRIES:    # coefficient array widens it: no column groups then)
                cols = op._partial_column_groups(var)
                if cols is not None:
                    entry = self.assembled_table(key, lead, *cols)
                    if entry is not None:
                        return entry
        return self._table(op._variable_tables()[var], key)

    def add_fuse(self, dst, src, e, after):
        """dst <- rotation `src` combined with relaxation `e` (entries as returned by _table / _generated)"""
        self.fuses.append((dst[1], src[1], e[1], dst[0], src[0], e[0], src[2], 1 if after else 0, 0))

    def add_fuse_partial(self, dst, src, dsrc, e, de, after):
        """dst <- partial of the table that `add_fuse(.., src, e, after)` generates, with respect to one variable:
        dsrc / de = entries of the partials of the rotation / the relaxation (None: does not depend on the variable)"""
        none = (-1, None, 0)
        dsrc, de = dsrc or none, de or none
        self.fuse_partials.append((dst[1], src[1], dsrc[1], e[1], de[1], dst[0], src[0], dsrc[0], e[0], de[0], src[2], dsrc[2],
                                   1 if after else 0))

    # -- records -----------------------------------------------------------------------
    def add(self, opcode, *, table=None, key=None, ia=0, ib=0, entry=None):
        space, off, ncoef = entry if entry is not None else ((-1, 0, 0) if table is None else self._table(table, key))
        if ncoef != _lib.NCOEF.get(opcode, 0):
            raise ValueError(f"opcode {opcode}: table has {ncoef} coefficients")
        self.records.append((opcode, space, int(ia), int(ib), off, ncoef))

    def add_partials(self, tables):
        """partial-derivative tables of the operator just added (diff.py DiffMixin._encode): {variable: (table, key)}, or
        {variable: ("entry", pool entry)} for a partial the library generates (add_fuse_partial)"""
        self.partials[len(self.records) - 1] = tables

    def add_deferred(self, opcode, builder):
        """operator whose table is laid out [*opshape, 3, K]: built once the capacity K is known"""
        self.deferred.append((len(self.records), builder))
        self.records.append((opcode, -1, 0, 0, 0, 0))

    # -- structural bookkeeping of the value-only operators (only needed in n-D mode) -------------
    def note(self, what, **kw):
        if self.kspace is None:
            return
        if what == "mix":
            self.kspace = self.kspace.after_mixing()
        elif what == "relax":
            self.kspace = self.kspace.after_relaxation(kw.get("recovery", True))
        elif what == "spoil":
            self.kspace = self.kspace.after_spoiler()
        elif what == "reset":
            self.kspace = kspace.KSpace.equilibrium(self.kspace.kdim)

    def kspace_now(self, kdim=1):
        """coordinate set at this point of the sequence (virtual 1-D orders if no n-D shift yet)"""
        if self.kspace is not None:
            return self.kspace.with_kdim(max(kdim, self.kspace.kdim))
        return kspace.KSpace.from_orders(self.nstate, kdim)

    def add_gather_shift(self, delta, nmax):
        """S(k) with an integer vector k  (shift.py:103-118 'shift-nd')"""
        delta = np.atleast_1d(np.asarray(delta))             # [kdim] or [*lead, kdim]
        if self.kspace is None:
            # first n-D shift on a state that so far only knew 1-D orders (statematrix.py:314-329);
            # nothing is known about which orders are populated, so all of them are assumed to be
            self.kspace = kspace.KSpace.from_orders(self.nstate, delta.shape[-1])
        new, tab = self.kspace.shifted(delta, nmax)
        self.kspace = new
        self.nstate = new.nstate
        self.peak = max(self.peak, new.nstate)

        def build(K, tab=tab):
            full = np.full((3, K), _lib.GS_ZERO, dtype=np.int32)
            full[:, : tab.shape[1]] = tab
            return full.reshape(-1).view(np.float64)[None, :]   # raw int32 [3][K] in the float64 pool

        self.add_deferred(_lib.OP_GS, build)

    def add_shift(self, k, nmax):
        if self.kspace is not None:   # coordinates exist: an int k means [k, 0, ...] (shift.py:224-229)
            return self.add_gather_shift([int(k)] + [0] * (self.kspace.kdim - 1), nmax)
        new = self.nstate + abs(int(k))
        if nmax:
            new = min(new, int(nmax))
        self.nstate = new
        self.peak = max(self.peak, new)
        self.add(_lib.OP_S, ia=int(k), ib=int(nmax) if nmax else NO_TRUNCATION)

    def add_adc(self, kind=0):
        slot = self.n_adc
        self.n_adc += 1 + len(self.variables)
        self.add(_lib.OP_ADC, ia=slot, ib=int(kind))
        return slot

    # -- output ------------------------------------------------------------------------
    @property
    def nvox(self):
        return int(np.prod(self.grid))

    def packable(self, derivatives=False):
        """capacity (16 or 32 orders per voxel: 4 or 2 voxels per wavefront) if a state-resident run can
        use the packed kernels, else 0: at most 31 orders, shifts by +-1 only, no n-D shifts /
        diffusion / general 3x3 matrices; `derivatives` selects the derivative variant"""
        if self.peak + 1 > _lib.PACKED_K[-1] or self.kspace is not None or self.deferred:
            return 0
        if bool(self.variables) != bool(derivatives):
            return 0
        ok = all(rec[0] not in (_lib.OP_D, _lib.OP_GS, _lib.OP_MAT, _lib.OP_MAT0) and (rec[0] != _lib.OP_S or abs(rec[2]) == 1)
                 for rec in self.records)
        return next(K for K in _lib.PACKED_K if K >= self.peak + 1) if ok else 0

    def packable_nd(self):
        """16 if a state-resident run from equilibrium of this plan -- one WITH integer n-D shifts / diffusion (their tables
        are laid out for the capacity they run at) -- can use the four-voxels-per-wavefront kernel: at most 16 orders, no
        general matrices, no derivative states; else 0"""
        if (self.kspace is None and not self.deferred) or self.peak + 1 > _lib.PACKED_K[0] or self.variables:
            return 0
        ok = all(rec[0] not in (_lib.OP_MAT, _lib.OP_MAT0) and (rec[0] != _lib.OP_S or abs(rec[2]) == 1) for rec in self.records)
        return _lib.PACKED_K[0] if ok else 0

    def capacity(self, at_least=0, resident=False):
        """device capacity (orders per voxel) for this plan; `resident`: the caller runs it state-resident from equilibrium,
        where 2048 orders are available too (plain operators and shifts by +-1 only)"""
        need = max(self.peak + 1, int(at_least), 1)
        for K in _lib.SUPPORTED_K:
            if K >= need:
                return K
        plain = (self.kspace is None and not self.deferred and not self.variables
                 and all(rec[0] != _lib.OP_S or abs(rec[2]) == 1 for rec in self.records))
        if resident and plain and need <= _lib.RESIDENT_ONLY_K:
            return _lib.RESIDENT_ONLY_K
        raise NotImplementedError(
            f"{need} phase states per voxel exceed the device capacity {_lib.SUPPORTED_K[-1]} "
            f"({_lib.RESIDENT_ONLY_K} for simulate() of rotations / relaxation / shifts by one from equilibrium); "
            "bound the state matrix with max_nstate=...")

    def arrays(self, K=None):
        if self.deferred:
            if K is None:
                K = self.capacity()
            for index, builder in self.deferred:
                opcode = self.records[index][0]
                space, off, ncoef = self._table(builder(K), None)
                self.records[index] = (opcode, space, 0, 0, off, ncoef)
            self.deferred = []
        partial_entries = {}                  # (host tables join the pool BEFORE its size is read off)
        if self.variables:
            for index, tables in self.partials.items():
                for v, var in enumerate(self.variables):
                    if var in tables:
                        given = tables[var]
                        generated = isinstance(given[0], str) and given[0] == "entry"
                        partial_entries[index, v] = given[1] if generated else self._table(*given)
        host_size = self.pool_size            # generated tables follow the host part of the pool
        fix = lambda off: off if off >= 0 else host_size + (-off - 1)
        ops = np.zeros(max(len(self.records), 1), dtype=_lib.OP_DTYPE)
        for i, (opcode, space, ia, ib, off, ncoef) in enumerate(self.records):
            ops[i] = (opcode, space, ia, ib, fix(off), ncoef, 0)
        if not self.records:
            ops[0] = (_lib.OP_NOP, -1, 0, 0, 0, 0, 0)
        dops = None
        if self.variables:
            dops = np.zeros(len(ops), dtype=_lib.DOP_DTYPE)
            dops["space"], dops["coef_off"] = -1, -1
            for (index, v), (space, off, _) in partial_entries.items():
                dops[index]["space"][v], dops[index]["coef_off"][v] = space, fix(off)
        coef = np.concatenate(self.pool) if self.pool else np.zeros(0)
        return ops, np.asarray(self.grid, dtype=np.int64), list(self.spaces), coef, dops

    def fuse_array(self):
        """the epgx_fuse list with final offsets (call after arrays())"""
        host_size = self.pool_size
        fix = lambda off: off if off >= 0 else host_size + (-off - 1)
        out = np.zeros(len(self.fuses), dtype=_lib.FUSE_DTYPE)
        for i, (dst, src, e, ds, ss, es, nc, after, _) in enumerate(self.fuses):
            out[i] = (fix(dst), fix(src), fix(e), ds, ss, es, nc, after, 0)
        return out

    def fuse_partial_array(self):
        """the epgx_fuse_partial list with final offsets (call after arrays())"""
        host_size = self.pool_size
        fix = lambda off: -1 if off is None else (off if off >= 0 else host_size + (-off - 1))
        out = np.zeros(len(self.fuse_partials), dtype=_lib.FUSE_PARTIAL_DTYPE)
        for i, (dst, src, dsrc, e, de, s0, s1, s2, s3, s4, nc, dnc, after) in enumerate(self.fuse_partials):
            out[i] = (fix(dst), fix(src), fix(dsrc), fix(e), fix(de), s0, s1, s2, s3, s4, nc, dnc, after)
        return out

    def assemble_array(self):
        """the epgx_assemble list with final offsets (call after arrays())"""
        host_size = self.pool_size
        out = np.zeros(len(self.assembles), dtype=_lib.ASSEMBLE_DTYPE)
        for i, (dst, space, ncoef, sources, col_src, col_idx) in enumerate(self.assembles):
            rec = out[i]
            rec["dst_off"], rec["dst_space"], rec["ncoef"], rec["n_src"] = host_size + (-dst - 1), space, ncoef, len(sources)
            for k, (off, ncol, strides) in enumerate(sources):
                rec["src"][k]["off"], rec["src"][k]["ncol"] = off, ncol
                rec["src"][k]["strides"][: len(strides)] = strides
            rec["col_src"][:ncoef], rec["col_idx"][:ncoef] = col_src, col_idx
        return out

    def plan_arrays(self, K=None):
        """the host-side description of the plan at capacity K (keyword arguments of _lib.DevicePlan), built once and kept.
        Building MUTATES the encoder (deferred tables join the pool, records are rewritten): one thread at a time, and
        multi-GPU callers build here, on their own thread, before they fan out over the devices (_Fleet)"""
        with self._arrays_lock:
            return self._plan_arrays_locked(K)

    def _plan_arrays_locked(self, K):
        cached = getattr(self, "_plan_arrays", None)
        if cached is None or cached[0] != K:
            ops, grid, spaces, coef, dops = self.arrays(K)
            if (self.packable() == _lib.PACKED_K[0] and not any(rec[0] == _lib.OP_SPOIL for rec in self.records)
                    and not self.variables and not os.environ.get("EPGX_FOLD16")):
                # the state-resident run of this plan takes the 16-orders-per-voxel kernel (one order per lane): there the
                # library's run-time fold of relaxations into rotations costs more than it saves (include/epgx.h,
                # EPGX_PLAN_NO_FOLD) -- unless the train is spoiled: the fold absorbs the spoilers, which otherwise send
                # every repetition through the flag-tested record body (500 spoiled repetitions over 10^6 voxels: 24.4 ms
                # unfolded, 14.7 ms folded).  Set on the PLAN, so that its per-timestep launches (K = 64) fold alike (the same chains; rotations about x at
                # 64 orders state-resident excepted: sum / difference form, last bits).
                # Plans WITH derivative states keep the fold: a relaxation stage there acts on every state and brings a partial
                # stage along, and the folded repetition replaces both by 4 + 2 multiply-adds per variable
                # (packed_dfold_kernel: 1000-TR MRF over 10^6 voxels, max_nstate = 10, three variables 166 -> 127 ms).
                self.deriv_flags |= _lib.PLAN_NO_FOLD
            cached = (K, dict(ops=ops, grid_shape=grid, space_strides=spaces, coef=coef, n_adc=self.n_adc, dops=dops,
                              n_vars=len(self.variables), deriv_flags=self.deriv_flags,
                              fuse=self.fuse_array() if self.fuses else None, n_coef_generated=self.generated_size,
                              assemble=self.assemble_array() if self.assembles else None,
                              fuse_partial=self.fuse_partial_array() if self.fuse_partials else None))
            self._plan_arrays = cached
        return cached[1]

    def device_plan(self, ctx, K=None):
        """the epgx_plan of this sequence on `ctx`; called once per GPU by multi-GPU runs (the host arrays are built once)"""
        return _lib.DevicePlan(ctx, **self.plan_arrays(K))


def apply_operators(sm, ops, _plain=False):
    """op(sm) for one or several operators: one launch of the fused kernel, state streamed
    HBM -> registers -> HBM once (the per-timestep mode of DESIGN.md).  User-written operators (Operator._on_host) split
    the list: what stands before one is launched, then its own `_apply` works on the state matrix."""
    if any(op._on_host() for op in ops):
        batch = []
        for op in list(ops) + [None]:
            if op is not None and not op._on_host():
                batch.append(op)
                continue
            if batch:
                sm = apply_operators(sm, batch)
                batch = []
            if op is not None:
                result = op._apply(op.prepare(sm, inplace=True))
                sm = sm if result is None else result
        return sm
    if not _plain and getattr(sm, "_eq", None) is not None:
        return _apply_with_equilibrium(sm, ops)
    grid = common.broadcast_shapes(sm.shape, *[op.shape for op in ops], append=True)
    opts = dict(sm.options)
    opts.setdefault("kvalue", sm.kvalue)
    enc = Encoder(grid, options=opts, nstate0=sm.nstate, kspace0=sm._kspace)
    enc.deriv_flags |= _lib.PLAN_NO_FOLD     # op(sm): the reference's operator-by-operator arithmetic
    for op in ops:
        op._encode(enc)
    if not enc.records:
        return sm
    sm._broadcast_to(grid)
    sm._reserve(enc.capacity(at_least=sm.nstate + 1))
    plan = enc.device_plan(sm._ctx, sm._state.K)
    _lib.run(sm._ctx, plan, 0, plan.n_ops, 0, plan.nvox, sm._state, sm._state, sm._state.K, None, 0, 0)
    sm._nstate = enc.nstate
    sm._kspace = enc.kspace
    return sm


def _apply_with_equilibrium(sm, ops):
    """op(sm) on a state matrix with a GENERAL equilibrium (statematrix.py:56-59; `states += arr0 * equilibrium`,
    opscalar.py:213-232).  The kernels know the equilibrium [0, 0, density]; here the state carries density 0 (no kernel adds
    anything) and the term comes from the second device-resident matrix: a copy of the equilibrium is multiplied by arr0 -- one
    launch of the same scalar stage -- and added to the state on the device (epgx_state_axpy).  Operator by operator: three
    launches per relaxation, the price of the rare case (stepwise mode of simulate, op(sm))."""
    from . import opscalar, operator as _operator

    for op in ops:
        for part in op._parts():
            if isinstance(part, _operator.Reset):
                sm._reset_to_equilibrium()
                continue
            if isinstance(part, _operator.PD) or getattr(part, "mat0", None) is not None and not isinstance(part, opscalar.ScalarOp):
                raise NotImplementedError(f"{type(part).__name__} on a state matrix with a general equilibrium")
            sm = apply_operators(sm, [part], _plain=True)      # (the state carries density 0: nothing recovers here)
            arr0 = getattr(part, "arr0", None) if isinstance(part, opscalar.ScalarOp) else None
            if arr0 is not None and np.any(arr0 != 0):
                term = sm._equilibrium_matrix()
                term = apply_operators(term, [opscalar.ScalarOp(np.asarray(arr0, dtype=np.complex128))])     # arr0 * equilibrium
                term._broadcast_to(sm.shape)
                sm._reserve(term._state.K)
                term._reserve(sm._state.K)
                sm._state.axpy(term._state, 1.0)
    return sm
