"""Plan-level peephole: collapse  E2 . T . E1  into one operator.

A relaxation without precession is a REAL diagonal diag(e, e, e2) plus a recovery r on Z_0
(evolution.py:220-242).  Sandwiching an RF rotation T between two of them keeps everything the
device's `apply_T` relies on -- the EPG symmetry and a real m00 -- and only rescales its
coefficients, while the recoveries turn into a constant term on the k = 0 order:

    E2 (T (E1 x + r1 z) ) + r2 z  =  (D2 T D1) x  +  (D2 T z) r1 + r2 z            z = density * e_Z0

so one EPGX_OP_T0 record (8 + 3 coefficients) replaces three operators and saves the two diagonal
multiplications per k-state (14 of 56 fp64 instructions per echo of the MSE benchmark).  The
reference offers the same algebra by hand as `E @ T` (opmatrix.py:85-135); here it is a compiler
pass over the flattened sequence:

  * a real E commutes with integer shifts (it is k-independent and S does not move Z_0), so an E
    that follows `T S ... S` is pulled in front of the shifts and absorbed into the T;
  * an E in front of a T (again across shifts) is absorbed from the right;
  * only when the fused table is no larger than the largest table it replaces (a per-voxel E next
    to a uniform T: yes; an E over (T1, T2) next to a T over a B1 axis: no, that would be the full
    3-D grid per pulse);
  * never across probes, spoilers, n-D shifts or anything else.

Results differ from the unfused sequence by rounding only (a different but equivalent order of
the same multiplications).
"""
import numpy as np

from . import common, operator, opmatrix, opscalar, shift as _shift, _lib


class FusedTE(operator.Operator):
    """T sandwiched between precession-free relaxations: device table [*shape, 12]
    (the 8 coefficients of EPGX_OP_T, then Re o0, Im o0, o2, 0)"""

    def __init__(self, table, parts):
        self.table = np.ascontiguousarray(table, dtype=np.float64)
        self.parts = tuple(parts)
        super().__init__(name="(" + " . ".join(p.name for p in reversed(self.parts)) + ")",
                         duration=sum((p.duration for p in self.parts), 0))

    @property
    def shape(self):
        return self.table.shape[:-1]

    def _encode(self, enc):
        enc.add(_lib.OP_T0, table=self.table, key=("T0", id(self)))
        enc.note("mix")
        enc.note("relax")


def _t_table(op):
    """[*shape, 12] table of a fusable T-like operator, or None"""
    if isinstance(op, FusedTE):
        return op.table
    if not isinstance(op, opmatrix.MatrixOp) or op.mat0 is not None or getattr(op, "order1", None):
        return None
    if op._packed is None:
        op._packed = opmatrix.pack_matrix(op.mat, op.mat0)
    opcode, table = op._packed
    if opcode != _lib.OP_T:
        return None
    return np.concatenate([table, np.zeros(table.shape[:-1] + (4,))], axis=-1)


def _e_table(op):
    """[*shape, 4] table (e, 0, e2, r) of a precession-free ScalarOp, or None"""
    if not isinstance(op, opscalar.ScalarOp) or getattr(op, "order1", None):
        return None
    if op._packed is None:
        op._packed = opscalar.pack_scalar(op.arr, op.arr0)
    table = op._packed[1]
    if np.any(table[..., 1] != 0.0):
        return None
    return table


def _align(a, b):
    """append-axes broadcasting of two tables' leading shapes; None if the result would be larger
    than both inputs"""
    sa, sb = a.shape[:-1], b.shape[:-1]
    if not common.broadcastable(sa, sb, append=True):
        return None
    shape = common.broadcast_shapes(sa, sb, append=True)
    if int(np.prod(shape)) > max(int(np.prod(sa)), int(np.prod(sb))):
        return None
    nd = len(shape)
    a = a.reshape(sa + (1,) * (nd - len(sa)) + a.shape[-1:])
    b = b.reshape(sb + (1,) * (nd - len(sb)) + b.shape[-1:])
    return a, b, shape


def _combine(t, e, after):
    """table of (E after T) or (T after E); None if the index spaces do not nest"""
    aligned = _align(t, e)
    if aligned is None:
        return None
    t, e, shape = aligned
    out = np.empty(shape + (12,))
    out[...] = t
    er, e2, r = e[..., 0], e[..., 2], e[..., 3]
    if after:    # rows of T scaled, constant term scaled and recovered
        out[..., 0:5] *= er[..., None]          # m00, m01, m02
        out[..., 5:8] *= e2[..., None]          # m20, m22
        out[..., 8:10] *= er[..., None]
        out[..., 10] = out[..., 10] * e2 + r
    else:        # columns of T scaled; the recovery in front of T passes through T's third column
        out[..., 8] += t[..., 3] * r
        out[..., 9] += t[..., 4] * r
        out[..., 10] += t[..., 7] * r
        out[..., 0:3] *= er[..., None]          # m00, m01
        out[..., 5:7] *= er[..., None]          # m20
        out[..., 3:5] *= e2[..., None]          # m02
        out[..., 7] *= e2                       # m22
    return out


def fusable(sequence):
    """the pass only runs on plain 1-D sequences (no n-D shifts / diffusion: their planner tracks
    structural zeros operator by operator)"""
    from . import diffusion
    return not any(isinstance(op, diffusion.D) or (isinstance(op, _shift.S) and not isinstance(op.k, int))
                   for op in sequence)


def fuse_sequence(sequence):
    """flat operator list -> equivalent flat list with E . T . E collapsed (see module docstring)"""
    out, cache = [], {}

    def merged(first, second, table):
        # the same operator objects are reused echo after echo: so are the fused ones (one table)
        key = (id(first), id(second))
        if key not in cache:
            parts = ((first.parts if isinstance(first, FusedTE) else (first,))
                     + (second.parts if isinstance(second, FusedTE) else (second,)))
            cache[key] = (FusedTE(table, parts), first, second)   # keep the inputs alive: ids stay unique
        return cache[key][0]

    for op in sequence:
        e_tab = _e_table(op)
        if e_tab is not None:
            j = len(out) - 1
            while j >= 0 and isinstance(out[j], _shift.S) and isinstance(out[j].k, int):
                j -= 1
            t_tab = _t_table(out[j]) if j >= 0 else None
            if t_tab is not None:
                key = (id(out[j]), id(op))
                table = None if key in cache else _combine(t_tab, e_tab, after=True)
                if key in cache or table is not None:
                    out[j] = merged(out[j], op, table)
                    continue
            out.append(op)
            continue
        t_tab = _t_table(op) if not isinstance(op, FusedTE) else None
        if t_tab is not None and out:
            # "E S ... S T": the E commutes with the shifts, so it is absorbed from the right and
            # the shifts stay where they are
            j = len(out) - 1
            while j >= 0 and isinstance(out[j], _shift.S) and isinstance(out[j].k, int):
                j -= 1
            e_prev = _e_table(out[j]) if j >= 0 else None
            if e_prev is not None:
                key = (id(out[j]), id(op))
                table = None if key in cache else _combine(t_tab, e_prev, after=False)
                if key in cache or table is not None:
                    fused = merged(out[j], op, table)
                    del out[j]
                    out.append(fused)
                    continue
        out.append(op)
    return out
