"""Host-side k-space bookkeeping for integer n-D shifts (the planner behind EPGX_OP_GS / EPGX_OP_D).

The reference's `shiftnd` (epgpy/shift.py:297-364) keeps, next to `states[..., R, 3]`, an integer
coordinate per row, `coords[..., R, kdim]`, lexicographically sorted and symmetric (row r <-> row
R-1-r has the negated coordinate; the centre row is k = 0).  A shift by `delta` re-derives the
coordinate set as unique({k, k + delta, k - delta}), scatters Z to its old coordinate and column 0
to coordinate + delta, rebuilds column 1 as the mirror image and prunes rows that are zero in
every voxel.

As long as the shift is the same for all voxels the coordinate set is voxel-independent, so
all of that is *planning*: this module evolves the coordinate set on the host and emits, per
shift, a gather table for the device (which old order each new order takes its F, conj(F-) and Z
from).  Instead of the reference's numerical pruning (all-voxel |value| < tol) rows are dropped
when they are *structurally* empty (no operator can have put anything there); this never drops
a row the reference keeps, so signals are identical -- only `sm.states` may carry extra all-zero
rows where the reference pruned a numerically cancelled state.
"""
import numpy as np

from . import _lib


def _lexsort_rows(rows):
    """unique rows in lexicographic order (first column most significant), as shift.py:461-475"""
    rows = np.unique(np.asarray(rows, dtype=np.int64), axis=0)
    order = np.lexsort(rows.T[::-1])
    return rows[order]


class KSpace:
    """coordinate set of a state matrix + which entries can be non-zero"""

    def __init__(self, coords, nz_f, nz_z):
        self.coords = np.asarray(coords, dtype=np.int64)      # [R, kdim], symmetric, lexsorted
        self.nz_f = np.asarray(nz_f, dtype=bool)               # column 0 may be non-zero at row r
        self.nz_z = np.asarray(nz_z, dtype=bool)               # column 2 may be non-zero at row r
        assert len(self.coords) % 2 == 1

    # -- constructors --------------------------------------------------------------------------
    @classmethod
    def equilibrium(cls, kdim):
        return cls(np.zeros((1, kdim), np.int64), [False], [True])

    @classmethod
    def from_orders(cls, nstate, kdim, dense=True):
        """1-D orders -n..n on the first axis (statematrix.py:425-432 `_setup_coords`)"""
        coords = np.zeros((2 * nstate + 1, kdim), np.int64)
        coords[:, 0] = np.arange(-nstate, nstate + 1)
        full = np.ones(2 * nstate + 1, bool)
        return cls(coords, full if dense else ~full, full)

    # -- geometry ------------------------------------------------------------------------------
    @property
    def kdim(self):
        return self.coords.shape[1]

    @property
    def nrow(self):
        return len(self.coords)

    @property
    def centre(self):
        return (self.nrow - 1) // 2

    @property
    def nstate(self):
        return self.centre

    @property
    def half(self):
        """coordinates of the stored orders (k >= 0 half, centre first)"""
        return self.coords[self.centre:]

    def with_kdim(self, kdim):
        if kdim == self.kdim:
            return self
        if kdim < self.kdim:
            raise RuntimeError("Cannot remove existing k-dimension")
        pad = np.zeros((self.nrow, kdim - self.kdim), np.int64)
        return KSpace(np.concatenate([self.coords, pad], axis=1), self.nz_f, self.nz_z)

    def _lookup(self):
        return {tuple(c): r for r, c in enumerate(self.coords.tolist())}

    # -- structural effect of the value-only operators -----------------------------------------
    def after_mixing(self):
        """T / MAT: a row's three columns mix"""
        any_ = self.nz_f | self.nz_f[::-1] | self.nz_z
        return KSpace(self.coords, any_, any_)

    def after_relaxation(self, recovery=True):
        nz_z = self.nz_z.copy()
        if recovery:
            nz_z[self.centre] = True
        return KSpace(self.coords, self.nz_f, nz_z)

    def after_spoiler(self):
        return KSpace(self.coords, np.zeros(self.nrow, bool), self.nz_z)

    # -- the shift -----------------------------------------------------------------------------
    def shifted(self, delta, nmax=None):
        """(new KSpace, gather table int32 [3, n_new_half]) for S(delta)  (shift.py:297-364)"""
        delta = np.asarray(delta, dtype=np.int64).reshape(-1)
        old = self.with_kdim(max(self.kdim, delta.size))
        delta = np.pad(delta, (0, old.kdim - delta.size))
        c0 = old.centre
        moved = old.coords[old.nz_f] + delta                      # where column 0 goes
        keep_z = old.coords[old.nz_z]
        cand = np.concatenate([np.zeros((1, old.kdim), np.int64), keep_z, -keep_z, moved, -moved])
        if nmax is not None:                                      # shift.py:330-341
            cand = cand[np.all(np.abs(cand) <= nmax, axis=1)]
        new_coords = _lexsort_rows(cand)
        look_old = old._lookup()
        n_new = len(new_coords)
        cn = (n_new - 1) // 2
        nz_f = np.zeros(n_new, bool)
        nz_z = np.zeros(n_new, bool)
        for r, c in enumerate(new_coords):
            src = look_old.get(tuple(c - delta))
            nz_f[r] = src is not None and old.nz_f[src]
            srz = look_old.get(tuple(c))
            nz_z[r] = srz is not None and old.nz_z[srz]
        new = KSpace(new_coords, nz_f, nz_z)

        nh = n_new - cn
        tab = np.full((3, nh), _lib.GS_ZERO, dtype=np.int32)
        for j in range(nh):
            c = new_coords[cn + j]
            # A_j = column 0 at c  <- old column 0 at c - delta
            r = look_old.get(tuple(c - delta))
            if r is not None and old.nz_f[r]:
                tab[0, j] = (r - c0) if r >= c0 else ((c0 - r) | _lib.GS_CONJ)   # conj(B_i) below the centre
            # B_j = conj(column 0 at -c)  <- conj(old column 0 at -c - delta)
            r = look_old.get(tuple(-c - delta))
            if r is not None and old.nz_f[r]:
                tab[1, j] = ((r - c0) | _lib.GS_CONJ) if r >= c0 else (c0 - r)   # conj(A_i) above the centre
            # Z_j stays at its coordinate
            r = look_old.get(tuple(c))
            if r is not None and old.nz_z[r]:
                tab[2, j] = r - c0
        return new, tab

    # -- diffusion geometry --------------------------------------------------------------------
    def bmatrices(self, kvalue, shift=None):
        """per stored order j: (bL, bT, bT_mirror) for tau = 1 ms  (diffusion.py:86-123)

        bL = tau k k^T;  bT = tau (k1 k1^T + 1/2 k1 kd^T + 1/2 kd k1^T + 1/3 kd kd^T) with
        k1 = k - shift, kd = shift;  k in rad/mm, tau in s.  bT_mirror is bT of the order -k
        (column 1 is the mirror image of column 0, diffusion.py:77).
        """
        kv = np.broadcast_to(np.asarray(kvalue, float).reshape(-1)[: self.kdim] if np.ndim(kvalue) else
                             np.asarray(kvalue, float), (self.kdim,))
        k = self.half * kv * 1e-3                                       # rad/mm
        tau = 1e-3                                                      # 1 ms in s

        def outer(a, b):
            return a[:, :, None] * b[:, None, :]

        b_l = outer(k, k) * tau
        if shift is None:
            return b_l, b_l, b_l
        sh = np.zeros(self.kdim)
        sh[: np.size(shift)] = np.asarray(shift, float).reshape(-1)
        sh = sh * kv * 1e-3

        def ramp(k2):
            k1 = k2 - sh
            kd = np.broadcast_to(sh, k1.shape)
            if np.allclose(kd, 0):
                return outer(k1, k1) * tau
            return tau * (outer(k1, k1) + 0.5 * outer(k1, kd) + 0.5 * outer(kd, k1) + outer(kd, kd) / 3)

        return b_l, ramp(k), ramp(-k)
