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
        mat0 = matrix_format(mat0, check=check)
        mat, mat0 = np.broadcast_arrays(mat, mat0)
    if axes is not None:
        mat = common.set_axes(2, mat, axes)
        mat0 = None if mat0 is None else common.set_axes(2, mat0, axes)
    return mat, mat0


def matrix_combine(mat1, mat2, mat01=None, mat02=None):
    """matrices of (op1 then op2) (opmatrix.py:173-187)"""
    mat1, mat2, mat01, mat02 = common.extend_operators(2, mat1, mat2, mat01, mat02)
    mat = mat2 @ mat1
    if mat01 is None and mat02 is None:
        mat0 = None
    elif mat01 is None:
        mat0 = mat02.copy()
    else:
        mat0 = mat2 @ mat01
        if mat02 is not None:
            mat0 = mat0 + mat02
    return mat, mat0


def pack_matrix(mat, mat0=None):
    """device tables: (opcode, [*opshape, ncoef] float64)

    T-like matrices have a real m00 (up to rounding of Rz Rx Rz^-1): 8 coefficients and 30
    fp64 ops per k-state; anything else uses the general symmetric form (10 doubles, 9 used).
    """
    m00, m01, m02, m20, m22 = mat[..., 0, 0], mat[..., 0, 1], mat[..., 0, 2], mat[..., 2, 0], mat[..., 2, 2]
    real00 = np.all(np.abs(m00.imag) <= 4e-16 * np.maximum(np.abs(m00.real), 1e-300)) or np.all(m00.imag == 0)
    if mat0 is not None:
        # mat0 multiplies the equilibrium [0, 0, density] of the k = 0 row (opmatrix.py:199-205):
        # only its third column matters: (o0, conj(o0), o2) * density is added to (F_0, F_0^*, Z_0)
        o0, o2 = mat0[..., 0, 2], mat0[..., 2, 2]
        cols = [m00.real, m00.imag, m01.real, m01.imag, m02.real, m02.imag, m20.real, m20.imag,
                m22.real, np.zeros_like(m22.real), o0.real, o0.imag, o2.real, np.zeros_like(o2.real)]
        return _lib.OP_MAT0, np.ascontiguousarray(np.stack(np.broadcast_arrays(*cols), axis=-1), dtype=np.float64)
    if real00:
        cols = [m00.real, m01.real, m01.imag, m02.real, m02.imag, m20.real, m20.imag, m22.real]
        return _lib.OP_T, np.ascontiguousarray(np.stack(cols, axis=-1), dtype=np.float64)
    cols = [m00.real, m00.imag, m01.real, m01.imag, m02.real, m02.imag, m20.real, m20.imag,
            m22.real, np.zeros_like(m22.real)]
    return _lib.OP_MAT, np.ascontiguousarray(np.stack(cols, axis=-1), dtype=np.float64)


class MatrixOp(operator.CombinableOperator):
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

    @classmethod
    def combinable(cls, other):
        from . import opscalar
        return isinstance(other, (MatrixOp, opscalar.ScalarOp))

    @classmethod
    def _combine(cls, op1, op2, **kwargs):
        mat, mat0 = matrix_combine(op1.mat, op2.mat, op1.mat0, op2.mat0)
        return MatrixOp(mat, mat0, **kwargs)

    def _encode(self, enc):
        if self._packed is None:
            self._packed = pack_matrix(self.mat, self.mat0)
        opcode, table = self._packed
        enc.add(opcode, table=table, key=("MAT", id(self)))
        enc.note("mix")
        if self.mat0 is not None:
            enc.note("relax")   # the constant term re-populates the k = 0 row
