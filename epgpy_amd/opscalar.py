"""State-wise diagonal operators (mirrors epgpy/opscalar.py).

`arr[*opshape, 3]` multiplies the three columns, `arr0[*opshape, 3]` multiplies the
equilibrium ([0, 0, density] on the k = 0 row only, statematrix.py:379-385), i.e. the only
effective recovery term is Z_0 += arr0[2] * density.  Device form: 4 doubles
(Re/Im arr[0], arr[2], arr0[2]); arr[1] = conj(arr[0]) is the symmetry `scalar_format` checks
(opscalar.py:178-192).
"""
import math

import numpy as np

from . import common, operator, _lib

NAX = np.newaxis


def scalar_format(arr, check=True):
    arr = np.asarray(arr, dtype=np.complex128)
    if arr.ndim == 1:
        arr = arr[NAX]
    if arr.ndim < 2 or arr.shape[-1] != 3:
        raise ValueError(f"Expected ...x3 array shape, found: {arr.shape}")
    if check and not np.allclose(arr, arr[..., (1, 0, 2)].conj()):
        raise ValueError(f"Invalid coefficients: {arr}")
    return arr


def scalar_setup(arr, arr0=None, *, axes=None, check=True):
    arr = scalar_format(arr, check=check)
    if arr0 is not None:
        arr0 = scalar_format(arr0, check=check)
        arr, arr0 = np.broadcast_arrays(arr, arr0)
    if axes is not None:
        arr = common.set_axes(1, arr, axes)
        arr0 = None if arr0 is None else common.set_axes(1, arr0, axes)
    return arr, arr0


def as_matrix(arr):
    """[..., 3] diagonal -> [..., 3, 3] (opscalar.py:154-158)"""
    return None if arr is None else arr[..., NAX] * np.eye(3)


def scalar_combine(arr_1, arr_2, arr0_1=None, arr0_2=None):
    """coefficients of (op1 then op2) (opscalar.py:195-210)"""
    arr_1, arr_2, arr0_1, arr0_2 = common.extend_operators(1, arr_1, arr_2, arr0_1, arr0_2)
    arr = arr_2 * arr_1
    if arr0_1 is None and arr0_2 is None:
        arr0 = None
    elif arr0_1 is None:
        arr0 = arr0_2.copy()
    else:
        arr0 = arr_2 * arr0_1
        if arr0_2 is not None:
            arr0 = arr0 + arr0_2
    return arr, arr0


def pack_scalar(arr, arr0):
    e0, e2 = arr[..., 0], arr[..., 2]
    # arr0 multiplies the equilibrium, whose only non-zero entry is Z_0 = density
    # (statematrix.py:379-385): arr0[0] and arr0[1] never contribute
    r0 = np.zeros(e2.shape) if arr0 is None else arr0[..., 2].real
    cols = [e0.real, e0.imag, e2.real, np.broadcast_to(r0, e2.shape)]
    return _lib.OP_E, np.ascontiguousarray(np.stack(cols, axis=-1), dtype=np.float64)


class ScalarOp(operator.CombinableOperator):
    """state-wise scalar multiplication (opscalar.py:11-78)"""

    def __init__(self, arr, arr0=None, *, axes=None, check=True, **kwargs):
        super().__init__(**kwargs)
        self._init(arr, arr0, axes=axes, check=check)

    def _init(self, arr, arr0=None, *, axes=None, check=True):
        self.arr, self.arr0 = scalar_setup(arr, arr0, axes=axes, check=check)
        self._packed = None

    @property
    def shape(self):
        return self.arr.shape[:-1]

    @property
    def mat(self):
        return as_matrix(self.arr)

    @property
    def mat0(self):
        return as_matrix(self.arr0)

    def combinable(self, other):
        return isinstance(other, type(self))   # opscalar.py:90-91

    @classmethod
    def _combine(cls, op1, op2, **kwargs):
        arr, arr0 = scalar_combine(op1.arr, op2.arr, op1.arr0, op2.arr0)
        return ScalarOp(arr, arr0, **kwargs)

    def _column_groups(self):
        """(groups, columns) if the 4-coefficient table is an outer combination of small per-axis columns
        (Encoder.assembled_table), else None.  Subclasses that know which parameter feeds which coefficient
        override this (evolution.py); a general ScalarOp ships its table"""
        return None

    def _dependent_columns(self, transverse, longitudinal):
        """column groups of a relaxation-type table: the F factor (arr[..., 0]) only varies along the axes the
        `transverse` parameters vary along, the Z factor and the recovery (arr[..., 2], arr0[..., 2]) along those of
        the `longitudinal` ones -- the reference evaluates exp() on exactly those small arrays and broadcasts the
        results into `arr` (evolution.py:220-242), so slicing them back out gives its values bit for bit"""
        lead = self.arr.shape[:-1]

        def pick(params):
            varies = [False] * len(lead)
            for par in params:
                shape = np.shape(par)
                if len(shape) > len(lead):
                    return None
                for d, n in enumerate(shape):
                    varies[d] = varies[d] or n > 1
            return tuple(slice(None) if v else slice(0, 1) for v in varies)

        sel_t, sel_l = pick(transverse), pick(longitudinal)
        if sel_t is None or sel_l is None:
            return None
        e0 = self.arr[sel_t + (0,)]
        e2 = self.arr[sel_l + (2,)]
        r0 = np.zeros(e2.shape) if self.arr0 is None else self.arr0[sel_l + (2,)].real
        groups = [np.stack([e0.real, e0.imag], axis=-1), np.stack([e2.real, r0], axis=-1)]
        return groups, [(0, 0), (0, 1), (1, 0), (1, 1)]

    def _pool_entry(self, enc):
        """(space, offset, 4) of this operator's table in the plan's pool: assembled on the device from its column
        groups when that pays, otherwise uploaded (once per operator object and plan either way)"""
        key = ("SCAL", id(self))
        if key in enc.tables:
            return enc.tables[key]
        if key in enc.generated:
            return enc.generated[key]
        lead = self.arr.shape[:-1]
        if math.prod(lead) >= enc.ASSEMBLE_MIN_ENTRIES:      # (small tables are simply uploaded: do not even slice them)
            cols = self.__dict__.get("_colgroups", False)
            if cols is False:
                cols = self.__dict__["_colgroups"] = self._column_groups()
            if cols is not None:
                entry = enc.assembled_table(key, lead, *cols)
                if entry is not None:
                    return entry
        if self._packed is None:
            self._packed = pack_scalar(self.arr, self.arr0)
        return enc._table(self._packed[1], key)

    def _encode(self, enc):
        enc.add(_lib.OP_E, entry=self._pool_entry(enc))
        enc.note("relax", recovery=self.arr0 is not None)
