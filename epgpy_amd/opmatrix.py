"""State-wise 3x3 matrix operators (mirrors epgpy/opmatrix.py).

`mat[*opshape, 3, 3]` is built on the host; the device applies it to every k-state of every
voxel inside the fused kernel (csrc/epgx_kernels.hip.h: apply_T / apply_MAT), exploiting the
symmetry the reference checks in `matrix_format` (opmatrix.py:157-170):
    mat == conj(mat[[1,0,2]][:, [1,0,2]])
so only m00, m01, m02, m20, m22 are shipped.
"""
import numpy as np

from . import common, operator, _lib

NAX = np.newaxis


def matrix_format(mat, check=True):
    """[..., 3, 3] complex128 with the EPG symmetry (opmatrix.py:157-170)"""
    mat = np.asarray(mat, dtype=np.complex128)
    if mat.ndim == 2:
        mat = mat[NAX]
    if mat.ndim < 3 or mat.shape[-2:] != (3, 3):
        raise ValueError(f"Expected ...x3x3 array shape, found: {mat.shape}")
    if check and not np.allclose(mat, mat[..., (1, 0, 2), :][..., (1, 0, 2)].conj()):
        raise ValueError(f"Invalid matrix coefficients: {mat}")
    return mat


def matrix_setup(mat, mat0=None, axes=None, check=True):
    mat = matrix_format(mat, check=check)
    if mat0 is not None:
        raise NotImplementedError("MatrixOp with an equilibrium term (mat0) is not on the device path")
    if axes is not None:
        mat = common.set_axes(2, mat, axes)
    return mat, None


def pack_matrix(mat):
    """device tables: (opcode, [*opshape, ncoef] float64)

    T-like matrices have a real m00 (up to rounding of Rz Rx Rz^-1): 8 coefficients and 30
    fp64 ops per k-state; anything else uses the general symmetric form (10 doubles, 9 used).
    """
    m00, m01, m02, m20, m22 = mat[..., 0, 0], mat[..., 0, 1], mat[..., 0, 2], mat[..., 2, 0], mat[..., 2, 2]
    real00 = np.all(np.abs(m00.imag) <= 4e-16 * np.maximum(np.abs(m00.real), 1e-300)) or np.all(m00.imag == 0)
    if real00:
        cols = [m00.real, m01.real, m01.imag, m02.real, m02.imag, m20.real, m20.imag, m22.real]
        return _lib.OP_T, np.ascontiguousarray(np.stack(cols, axis=-1), dtype=np.float64)
    cols = [m00.real, m00.imag, m01.real, m01.imag, m02.real, m02.imag, m20.real, m20.imag,
            m22.real, np.zeros_like(m22.real)]
    return _lib.OP_MAT, np.ascontiguousarray(np.stack(cols, axis=-1), dtype=np.float64)


class MatrixOp(operator.Operator):
    """state-wise matrix multiplication (opmatrix.py:10-63)"""

    def __init__(self, mat, mat0=None, *, axes=None, check=True, **kwargs):
        super().__init__(**kwargs)
        self._init(mat, mat0, axes=axes, check=check)

    def _init(self, mat, mat0=None, *, axes=None, check=True):
        self.mat, self.mat0 = matrix_setup(mat, mat0, axes=axes, check=check)
        self._packed = None

    @property
    def shape(self):
        return self.mat.shape[:-2]

    def _encode(self, enc):
        if self._packed is None:
            self._packed = pack_matrix(self.mat)
        opcode, table = self._packed
        enc.add(opcode, table=table, key=("MAT", id(self)))
        enc.note("mix")
