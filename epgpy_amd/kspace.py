"""Host-side k-space bookkeeping for integer n-D shifts (the planner behind EPGX_OP_GS / EPGX_OP_D).

The reference's `shiftnd` (epgpy/shift.py:297-364) keeps, next to `states[..., R, 3]`, an integer
coordinate per row, `coords[..., R, kdim]`, lexicographically sorted and symmetric (row r <-> row
R-1-r has the negated coordinate; the centre row is k = 0).  A shift by `delta` re-derives the
coordinate set as unique({k, k + delta, k - delta}), scatters Z to its old coordinate and column 0
to coordinate + delta, rebuilds column 1 as the mirror image and prunes rows that are zero in
every voxel.

The coordinate of a row may differ from voxel to voxel (a vectorised `k`, e.g. one gradient
direction or amplitude per point of a grid axis: shift.py:38-41, test_shift.py:196-203), but the ROW
STRUCTURE never does: `unique` runs over whole rows `[*lead, kdim]` (shift.py:461-475), so two rows
merge only if they coincide in every voxel.  All of that is therefore *planning*: this module evolves
the coordinate set on the host -- a row is the vector of its coordinates in all `lead` voxels -- and
emits, per shift, ONE gather table for the device (which old order each new order takes its F,
conj(F-) and Z from), shared by all voxels.  Instead of the reference's numerical pruning (all-voxel
|value| < tol) rows are dropped when they are *structurally* empty (no operator can have put anything
there); this never drops a row the reference keeps, so signals are identical -- only `sm.states` may
carry extra all-zero rows where the reference pruned a numerically cancelled state.
"""
import numpy as np

from . import common, _lib


def _row_ids(*blocks):
    """integer id per row of several [n_i, w] blocks: equal rows <-> equal ids, ids ordered like the
    rows' lexicographic order (first column most significant, shift.py:461-475)"""
    rows = np.concatenate(blocks, axis=0)
    _, inverse = np.unique(rows, axis=0, return_inverse=True)
    inverse = inverse.reshape(-1)
    out, at = [], 0
    for blk in blocks:
        out.append(inverse[at: at + len(blk)])
        at += len(blk)
    return out


class KSpace:
    """coordinate set of a state matrix + which entries can be non-zero.

    `points [R, L, kdim]`: coordinate of row r in each of the L = prod(lead) voxel classes; `lead` is a leading part
    of the parameter grid (append rule), () while the coordinates are the same for all voxels."""

    def __init__(self, coords, nz_f, nz_z, lead=()):
        pts = np.asarray(coords, dtype=np.int64)
        self.lead = tuple(int(d) for d in lead)
        self.points = pts.reshape(pts.shape[0], -1, pts.shape[-1])   # [R, L, kdim], symmetric, lexsorted
        assert self.points.shape[1] == int(np.prod(self.lead, dtype=np.int64))
        self.nz_f = np.asarray(nz_f, dtype=bool)               # column 0 may be non-zero at row r
        self.nz_z = np.asarray(nz_z, dtype=bool)               # column 2 may be non-zero at row r
        assert len(self.points) % 2 == 1

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

    @classmethod
    def from_coords(cls, coords, nz_f=None, nz_z=None):
        """coordinates as `StateMatrix.coords` returns them, [*lead, R, kdim] (checked: integer, symmetric, sorted)"""
        coords = np.asarray(coords)
        if coords.ndim < 2 or coords.shape[-2] % 2 != 1:
            raise ValueError("coords: expected [..., 2n+1, kdim]")
        if not np.all(coords == np.round(coords)):
            raise NotImplementedError("float wavenumbers (shift-merge / shift-prune) are not on the device path")
        lead = coords.shape[:-2]
        while lead and lead[-1] == 1:
            lead = lead[:-1]
        pts = np.moveaxis(coords.astype(np.int64).reshape(lead + coords.shape[-2:]), -2, 0)   # [R, *lead, kdim]
        nrow = pts.shape[0]
        flat = pts.reshape(nrow, -1)
        if not np.array_equal(flat, -flat[::-1]):
            raise ValueError("coords: rows must be symmetric about the centre row")
        ids, = _row_ids(flat)
        if not np.array_equal(ids, np.arange(nrow)):
            raise ValueError("coords: rows must be distinct and sorted lexicographically")
        full = np.ones(nrow, bool)
        return cls(pts, full if nz_f is None else nz_f, full if nz_z is None else nz_z, lead)

    # -- geometry ------------------------------------------------------------------------------
    @property
    def coords(self):
        """[R, kdim] while voxel-independent, else [*lead, R, kdim]"""
        if not self.lead:
            return self.points[:, 0, :]
        return np.moveaxis(self.points.reshape((self.nrow,) + self.lead + (self.kdim,)), 0, -2)

    @property
    def kdim(self):
        return self.points.shape[2]

    @property
    def nrow(self):
        return len(self.points)

    @property
    def centre(self):
        return (self.nrow - 1) // 2

    @property
    def nstate(self):
        return self.centre

    @property
    def half(self):
        """coordinates of the stored orders (k >= 0 half, centre first): [n_half, kdim] or [*lead, n_half, kdim]"""
        return self.coords[..., self.centre:, :]

    def with_kdim(self, kdim):
        if kdim == self.kdim:
            return self
        if kdim < self.kdim:
            raise RuntimeError("Cannot remove existing k-dimension")
        pad = np.zeros(self.points.shape[:2] + (kdim - self.kdim,), np.int64)
        return KSpace(np.concatenate([self.points, pad], axis=2), self.nz_f, self.nz_z, self.lead)

    def with_lead(self, lead):
        """the same rows seen from a larger leading grid (append rule): every new voxel class repeats its source"""
        lead = tuple(int(d) for d in lead)
        if lead == self.lead:
            return self
        pts = self.points.reshape((self.nrow,) + self.lead + (1,) * (len(lead) - len(self.lead)) + (self.kdim,))
        pts = np.broadcast_to(pts, (self.nrow,) + lead + (self.kdim,))
        return KSpace(pts, self.nz_f, self.nz_z, lead)

    def _like(self, nz_f, nz_z):
        return KSpace(self.points, nz_f, nz_z, self.lead)

    # -- structural effect of the value-only operators -----------------------------------------
    def after_mixing(self):
        """T / MAT: a row's three columns mix"""
        any_ = self.nz_f | self.nz_f[::-1] | self.nz_z
        return self._like(any_, any_)

    def after_relaxation(self, recovery=True):
        nz_z = self.nz_z.copy()
        if recovery:
            nz_z[self.centre] = True
        return self._like(self.nz_f, nz_z)

    def after_spoiler(self):
        return self._like(np.zeros(self.nrow, bool), self.nz_z)

    # -- the shift -----------------------------------------------------------------------------
    def shifted(self, delta, nmax=None):
        """(new KSpace, gather table int32 [3, n_new_half]) for S(delta)  (shift.py:297-364);
        delta: [kdim'] or [*dlead, kdim'] (one vector per voxel class of a leading part of the grid)"""
        delta = np.atleast_1d(np.asarray(delta, dtype=np.int64))
        dlead = delta.shape[:-1]
        while dlead and dlead[-1] == 1:
            dlead = dlead[:-1]
        delta = delta.reshape(dlead + delta.shape[-1:])
        lead = tuple(common.broadcast_shapes(self.lead or (1,), dlead or (1,), append=True)) if (self.lead or dlead) else ()
        if lead == (1,):
            lead = ()
        old = self.with_kdim(max(self.kdim, delta.shape[-1])).with_lead(lead)
        kdim, nl = old.kdim, old.points.shape[1]
        dl = np.zeros(dlead + (kdim,), np.int64)
        dl[..., : delta.shape[-1]] = delta
        dl = np.broadcast_to(dl.reshape(dlead + (1,) * (len(lead) - len(dlead)) + (kdim,)), lead + (kdim,)).reshape(1, nl * kdim)

        c0 = old.centre
        flat = old.points.reshape(old.nrow, nl * kdim)
        moved = flat[old.nz_f] + dl                               # where column 0 goes
        keep_z = flat[old.nz_z]
        cand = np.concatenate([np.zeros((1, nl * kdim), np.int64), keep_z, -keep_z, moved, -moved])
        if nmax is not None:                                      # shift.py:330-341: kept if inside the box in ANY voxel
            inside = np.all(np.abs(cand.reshape(-1, nl, kdim)) <= nmax, axis=2).any(axis=1)
            cand = cand[inside]
        new_flat = np.unique(cand, axis=0)                        # lexicographic, first column most significant
        n_new = len(new_flat)
        cn = (n_new - 1) // 2

        # which old row (if any) sits at: c - delta (source of column 0), -c - delta (source of the mirrored column), c
        id_old, id_from, id_mirror, id_same = _row_ids(flat, new_flat - dl, -new_flat - dl, new_flat)
        look_old = {int(i): r for r, i in enumerate(id_old)}
        src_f = [look_old.get(int(i)) for i in id_from]
        src_m = [look_old.get(int(i)) for i in id_mirror]
        src_z = [look_old.get(int(i)) for i in id_same]
        nz_f = np.array([r is not None and bool(old.nz_f[r]) for r in src_f])
        nz_z = np.array([r is not None and bool(old.nz_z[r]) for r in src_z])
        new = KSpace(new_flat.reshape(n_new, nl, kdim), nz_f, nz_z, lead)

        nh = n_new - cn
        tab = np.full((3, nh), _lib.GS_ZERO, dtype=np.int32)
        for j in range(nh):
            # A_j = column 0 at c  <- old column 0 at c - delta
            r = src_f[cn + j]
            if r is not None and old.nz_f[r]:
                tab[0, j] = (r - c0) if r >= c0 else ((c0 - r) | _lib.GS_CONJ)   # conj(B_i) below the centre
            # B_j = conj(column 0 at -c)  <- conj(old column 0 at -c - delta)
            r = src_m[cn + j]
            if r is not None and old.nz_f[r]:
                tab[1, j] = ((r - c0) | _lib.GS_CONJ) if r >= c0 else (c0 - r)   # conj(A_i) above the centre
            # Z_j stays at its coordinate
            r = src_z[cn + j]
            if r is not None and old.nz_z[r]:
                tab[2, j] = r - c0
        return new, tab

    # -- diffusion geometry --------------------------------------------------------------------
    def bmatrices(self, kvalue, shift=None):
        """per stored order j: (bL, bT, bT_mirror) for tau = 1 ms  (diffusion.py:86-123), each [n_half, kdim, kdim]
        (or [*lead, n_half, kdim, kdim] when the coordinates differ between voxels)

        bL = tau k k^T;  bT = tau (k1 k1^T + 1/2 k1 kd^T + 1/2 kd k1^T + 1/3 kd kd^T) with
        k1 = k - shift, kd = shift;  k in rad/mm, tau in s.  bT_mirror is bT of the order -k
        (column 1 is the mirror image of column 0, diffusion.py:77).
        """
        kv = np.broadcast_to(np.asarray(kvalue, float).reshape(-1)[: self.kdim] if np.ndim(kvalue) else
                             np.asarray(kvalue, float), (self.kdim,))
        k = self.half * kv * 1e-3                                       # rad/mm
        tau = 1e-3                                                      # 1 ms in s

        def outer(a, b):
            return a[..., :, None] * b[..., None, :]

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
