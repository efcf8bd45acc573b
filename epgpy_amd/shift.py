"""Shift operator S (mirrors epgpy/shift.py:14-158, integer 1-D branch).

`S(k)` with a Python int k moves F_j -> F_{j+k}: the number of states grows by |k| up to
`max_nstate` (option on the state matrix, shift.py:86) or this operator's `nmax`; beyond
that the highest order is dropped (shift.py:98, :283-287).  On the device this is a DPP
wave shift (|k| = 1) or an LDS-staged permutation (|k| > 1) inside the fused kernel.
Integer vectors -- one for all voxels, or one per point of the leading grid axes -- take the host-planned
gather shift (kspace.py); float wavenumbers (shift-merge / shift-prune) are not on the device path.
"""
import numpy as np

from . import common, operator, _lib


class S(operator.Operator):
    def __init__(self, k, *, nmax=None, kgrid=None, prune=1e-8, name=None, duration=None):
        if np.allclose(k, 0):
            raise TypeError("Cannot have k == 0")
        if isinstance(k, (int, np.integer)) and not isinstance(k, bool):
            k = int(k)
        else:
            k = np.atleast_2d(k)
            if k.shape[-1] not in [1, 2, 3, 4]:
                raise ValueError("k.shape[-1] must belong to [1, 2, 3, 4]")
        self.k, self.nmax, self.prune, self.kgrid = k, nmax, prune, kgrid
        if not name:
            name = common.repr_operator("S", ["k"], [k], ["" if isinstance(k, int) else ".2f"])
        super().__init__(name=name, duration=duration)

    @property
    def nshift(self):
        if common.isscalar(self.k):
            return abs(self.k)
        return np.round(np.max(abs(self.k))).astype(int)

    @property
    def shape(self):
        return (1,) if common.isscalar(self.k) else self.k.shape[:-1]

    @property
    def kdim(self):
        return 1 if common.isscalar(self.k) else self.k.shape[-1]

    def __call__(self, sm, *, inplace=False):
        """S has no parameters but is a DiffOperator in the reference (shift.py:14): derivative states
        attached to `sm` are shifted with it"""
        order1, order2 = getattr(sm, "order1", None), getattr(sm, "order2", None)
        sm = super().__call__(sm, inplace=inplace)
        if order1 or order2:
            from . import diff
            sm.order1 = diff.propagate_plain(self, sm, order1 or {}, inplace)
            sm.order2 = diff.propagate_plain(self, sm, order2 or {}, inplace)
        return sm

    def _encode(self, enc):
        # shift.py:86: the state-matrix option wins over the operator's own nmax
        nmax = enc.options.get("max_nstate") or self.nmax or None
        if isinstance(self.k, int):
            enc.add_shift(self.k, nmax)                      # 'shift-1d' (or [k,0,..] once coords exist)
            return
        if not np.issubdtype(self.k.dtype, np.integer):
            if (enc.options.get("kgrid") or self.kgrid) is None:
                raise AttributeError("kgrid not set")        # the reference's own error (shift.py:131-132)
            raise NotImplementedError("float wavenumbers (shift-merge / shift-prune, shift.py:367-542) are "
                                      "not on the device path")
        # 'shift-nd', shift.py:103-118; a vectorised k (one vector per point of the leading grid axes) keeps ONE row
        # structure for all voxels, only the coordinates of the rows then differ per voxel (kspace.py)
        enc.add_gather_shift(self.k[0] if self.k.shape[:-1] == (1,) else self.k, nmax)
