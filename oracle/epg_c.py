"""ctypes front-end of the C oracle (oracle/epg_oracle.c) -- TEST INFRASTRUCTURE ONLY.

Takes the same tuple-described sequences as oracle/epg_numpy.py, builds the per-operator
coefficient tables with that module's restatement of the reference's host-side builders
(transition.py:114-151, evolution.py:220-256), expands them per voxel and calls
`epgo_simulate`.  Used (a) to cross-check the NumPy oracle, (b) as the `cpu_baseline`
("port") timed by bench.py on the GPU box's host cores.
"""
import ctypes
import os
import subprocess

import numpy as np

from . import epg_numpy as onp

HERE = os.path.dirname(os.path.abspath(__file__))
LIBPATH = os.path.join(HERE, "libepgoracle.so")

KIND = {"NOP": 0, "MAT": 1, "SCAL": 2, "SHIFT": 3, "ADC_F0": 4, "ADC_Z0": 5, "SPOIL": 6,
        "RESET": 7, "PD": 8}


class _Op(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("k", ctypes.c_int32),
                ("coef", ctypes.c_void_p), ("stride", ctypes.c_int64)]


def build(force=False):
    src = os.path.join(HERE, "epg_oracle.c")
    if force or not os.path.exists(LIBPATH) or os.path.getmtime(LIBPATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-B", "libepgoracle.so"],
                              stdout=subprocess.DEVNULL)
    return LIBPATH


_lib = None
_path = None      # library in use: EPGORACLE_LIBRARY (e.g. the sanitizer build), the native build, or the portable one


def use_native_build():
    """switch to a -O3 -march=native build made on THIS host (oracle/Makefile `native`); returns True if it is in
    use.  Same arithmetic as the portable build (contraction stays off under -std=c99)."""
    global _lib, _path
    if os.environ.get("EPGORACLE_LIBRARY"):
        return False
    native = os.path.join(HERE, "libepgoracle_native.so")
    try:
        subprocess.check_call(["make", "-C", HERE, "-B", "native"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        ctypes.CDLL(native)
    except (OSError, subprocess.CalledProcessError):
        return False
    _lib, _path = None, native
    return True


def lib():
    global _lib
    if _lib is None:
        path = os.environ.get("EPGORACLE_LIBRARY") or _path
        if not path:
            build()
            path = LIBPATH
        _lib = ctypes.CDLL(path)
        _lib.epgo_simulate.restype = ctypes.c_int
        _lib.epgo_simulate.argtypes = [
            ctypes.POINTER(_Op), ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        _lib.epgo_final_nstate.restype = ctypes.c_int
        _lib.epgo_final_nstate.argtypes = [ctypes.POINTER(_Op), ctypes.c_int, ctypes.c_int,
                                           ctypes.c_int]
        _lib.epgo_max_threads.restype = ctypes.c_int
    return _lib


def max_threads():
    return lib().epgo_max_threads()


def _per_voxel(coef, grid, tail):
    """broadcast an operator table [*opshape, *tail] over the grid -> ([nvox, prod(tail)], stride)"""
    coef = np.asarray(coef)
    lead = coef.shape[: coef.ndim - tail]
    tshape = coef.shape[coef.ndim - tail:]
    ncoef = int(np.prod(tshape))
    if int(np.prod(lead)) == 1:
        flat = np.ascontiguousarray(coef.reshape(ncoef))
        return flat, 0
    lead = lead + (1,) * (len(grid) - len(lead))
    full = np.broadcast_to(coef.reshape(lead + tshape), grid + tshape)
    flat = np.ascontiguousarray(full.reshape(-1, ncoef))
    return flat, ncoef


def compile_ops(ops, grid):
    """tuple ops -> (ctypes array of _Op, keep-alive list, n_adc)"""
    keep, cops, n_adc = [], [], 0
    seen = {}   # a tuple object that occurs many times in the list (blk * necho) is tabulated once
    for op in ops:
        kind = op[0]
        if id(op) in seen:
            cops.append(seen[id(op)])
            n_adc += kind == "ADC"
            continue
        mark = len(cops)
        if kind == "T":
            mat = onp.rotation_matrix(op[1], op[2])
            flat, stride = _per_voxel(mat, grid, 2)
            flat = flat.view(np.float64)
            keep.append(flat)
            cops.append((KIND["MAT"], 0, flat, stride * 2))
        elif kind in ("E", "P"):
            if kind == "E":
                arr, arr0 = onp.relaxation_coeffs(*op[1:])
            else:
                arr, arr0 = onp.precession_coeffs(op[1], op[2])
                arr0 = np.zeros_like(arr)
            both = np.concatenate([arr, arr0], axis=-1)  # [..., 6]
            flat, stride = _per_voxel(both, grid, 1)
            flat = flat.view(np.float64)
            keep.append(flat)
            cops.append((KIND["SCAL"], 0, flat, stride * 2))
        elif kind == "S":
            cops.append((KIND["SHIFT"], int(op[1]), None, 0))
        elif kind == "ADC":
            what = op[1] if len(op) > 1 else "F0"
            cops.append((KIND["ADC_F0"] if what == "F0" else KIND["ADC_Z0"], 0, None, 0))
            n_adc += 1
        elif kind == "SPOILER":
            cops.append((KIND["SPOIL"], 0, None, 0))
        elif kind == "RESET":
            cops.append((KIND["RESET"], 0, None, 0))
        elif kind == "PD":
            pd = np.atleast_1d(np.asarray(op[1], float))
            flat, stride = _per_voxel(pd[..., None], grid, 1)
            flat = np.ascontiguousarray(flat, dtype=np.float64)
            keep.append(flat)
            cops.append((KIND["PD"], 1 if (len(op) < 3 or op[2]) else 0, flat, stride))
        elif kind in ("WAIT", "NULL"):
            continue
        else:
            raise ValueError(f"unknown op {kind}")
        if len(cops) == mark + 1:
            seen[id(op)] = cops[-1]
    arr = (_Op * len(cops))()
    for i, (k, ik, flat, stride) in enumerate(cops):
        arr[i].kind, arr[i].k = k, ik
        arr[i].coef = flat.ctypes.data if flat is not None else None
        arr[i].stride = stride
    return arr, keep, n_adc


def simulate(ops, *, shape=None, max_nstate=None, density=None, nthreads=1,
             return_states=False, compiled=None):
    """same contract as epg_numpy.simulate (ADC phase compensation not supported here)"""
    L = lib()
    grid = onp.broadcast_append(onp.seq_shape(ops), tuple(shape) if shape else (1,))
    nvox = int(np.prod(grid))
    cops, keep, n_adc = compiled if compiled is not None else compile_ops(ops, grid)
    nmax = int(max_nstate) if max_nstate else 0
    nfinal = L.epgo_final_nstate(cops, len(cops), 0, nmax)
    signal = np.empty((n_adc,) + grid, dtype=np.complex128)
    states = np.empty(grid + (2 * nfinal + 1, 3), dtype=np.complex128) if return_states else None
    dens = None
    if density is not None:
        dens = np.ascontiguousarray(
            np.broadcast_to(onp._append_axes(np.asarray(density, float), len(grid)), grid), float)
    rc = L.epgo_simulate(cops, len(cops), nvox, nmax, 0, None,
                         dens.ctypes.data if dens is not None else None,
                         signal.ctypes.data,
                         states.ctypes.data if states is not None else None, int(nthreads))
    if rc != 0:
        raise MemoryError("epgo_simulate failed")
    if return_states:
        return signal, states
    return signal
