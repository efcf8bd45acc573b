"""Diffusion operator D (mirrors epgpy/diffusion.py:14-176).

`D(tau, D, k=None)` attenuates every phase state by exp(-tr(b D)): longitudinal states with
b_L = tau k k^T, transverse states with the b-matrix of a wavenumber ramping linearly from
k - shift to k during tau (`k=` names the shift of the gradient lobe that ends at this operator).
The factors depend on the state's k-space coordinate, not on the voxel, so they are tabulated on
the host -- `[3][K]` real numbers per table entry (F, mirrored F, Z) -- and applied by the fused
kernel as one per-order multiply (EPGX_OP_D).

Extension over the reference: `field=True` lets `D` be an ARRAY of scalar diffusivities laid out
on the parameter grid (one table entry per value), which the reference cannot express
(`diffusion.py:166-169` rejects 1-D arrays; BASELINE config 5 sweeps ADC over a grid axis).
"""
import numpy as np

from . import common, operator, _lib


def get_shape(tau, D, k, field=False):
    """operator shape and k dimension, same checks as diffusion.py:150-176"""
    tau_shape, k_shape, d_shape = common.get_shape(tau), common.get_shape(k), common.get_shape(D)
    if not k_shape:
        k_shape = ()
    elif len(k_shape) == 1:
        k_shape = (1,) + k_shape
    if field:
        lead = d_shape
    else:
        if len(d_shape) == 1:
            raise ValueError("D can only be a scalar or a 2d matrix")
        if len(set(d_shape[-2:])) == 2:
            raise ValueError("D must be a square 2d matrix")
        if len(d_shape) and len(k_shape) and d_shape[-1] != k_shape[-1]:
            raise ValueError("Incompatible D and k dimensions")
        lead = d_shape[:-2]
    shape = common.broadcast_shapes(tau_shape, lead, k_shape[:-1], [1], append=True)
    return shape, (k_shape[-1] if k_shape else 1)


class D(operator.Operator):
    def __init__(self, tau, D, k=None, *, method=None, field=False, name=None, duration=None):
        tau, D, k = common.map_arrays((tau, D, k))
        self._shape, self._kdim = get_shape(tau, D, k, field)
        if name is None:
            name = common.repr_operator("D", ["tau", "D", "k"], [tau, D, k], [".1f", "", ""])
        self._duration = duration
        if duration is True:
            duration = tau
        self.tau, self.D, self.k, self.field = tau, D, k, bool(field)
        if k is not None and np.ndim(k) > 1 and np.shape(k)[:-1] != (1,):
            # (the reference accepts the argument but cannot apply it: `sm.k - shift` does not broadcast, diffusion.py:69)
            raise NotImplementedError("a different `k=` per voxel is not supported (S takes one: the coordinates then differ per voxel)")
        super().__init__(name=name, duration=duration)

    @property
    def shape(self):
        return self._shape

    @property
    def kdim(self):
        return self._kdim

    def _encode(self, enc):
        kdim = max(self._kdim, 1 if np.isscalar(self.D) or self.field else np.shape(self.D)[-1])
        ks = enc.kspace_now(kdim)
        kvalue = enc.options.get("kvalue", 1.0)
        shift = None if self.k is None else np.asarray(self.k).reshape(-1)
        b_l, b_t, b_m = ks.bmatrices(kvalue, shift)          # [n_half, kdim, kdim], tau = 1 ms
        tau = np.asarray(self.tau, dtype=float)
        isotropic = self.field or np.ndim(self.D) == 0
        dval = np.asarray(self.D, dtype=float)
        if not isotropic and dval.shape[-1] != b_l.shape[-1]:
            raise ValueError("Incompatible D and k dimensions")

        def along(lead_arr, rows):
            """lead_arr [*la] x rows [*lr, n] -> [*broadcast(la, lr), n]  (append rule on the leading axes)"""
            lead_arr = np.asarray(lead_arr, dtype=float)
            nd = max(lead_arr.ndim, rows.ndim - 1)
            return (lead_arr.reshape(lead_arr.shape + (1,) * (nd - lead_arr.ndim) + (1,))
                    * rows.reshape(rows.shape[:-1] + (1,) * (nd - rows.ndim + 1) + rows.shape[-1:]))

        def build(K):
            cols = []
            for b in (b_t, b_m, b_l):                          # F, mirrored F, Z;  b is for tau = 1 ms: [*klead, n_half, d, d]
                if isotropic:                                  # exp(-tr(b) D), diffusion.py:133-139
                    geo = np.trace(b, axis1=-2, axis2=-1)      # [*klead, n_half]
                    t_e, d_e = common.expand_arrays(tau, dval, append=True)
                    expo = along(np.asarray(t_e) * np.asarray(d_e), geo)
                else:                                          # exp(-tr(b D)), diffusion.py:140-145
                    nd = max(b.ndim - 3, dval.ndim - 2)
                    bb = b.reshape(b.shape[:-3] + (1,) * (nd - (b.ndim - 3)) + b.shape[-3:])
                    dd = dval.reshape(dval.shape[:-2] + (1,) * (nd - (dval.ndim - 2)) + (1,) + dval.shape[-2:])
                    geo = np.sum(bb * dd, axis=(-2, -1))       # [*broadcast(klead, dlead), n_half]
                    expo = along(tau, geo)
                expo = np.atleast_2d(expo)
                col = np.ones(expo.shape[:-1] + (K,))
                col[..., : expo.shape[-1]] = np.exp(-expo)
                cols.append(col)
            nd = max(c.ndim for c in cols)
            cols = [c.reshape(c.shape[:-1] + (1,) * (nd - c.ndim) + (K,)) for c in cols]
            lead = np.broadcast_shapes(*[c.shape[:-1] for c in cols])
            table = np.stack([np.broadcast_to(c, lead + (K,)) for c in cols], axis=-2)   # [*opshape, 3, K]
            return table.reshape(lead + (3 * K,))

        enc.add_deferred(_lib.OP_D, build)
