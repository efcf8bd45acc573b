"""Host-side array helpers: the reference's broadcasting rules, restated.

epgpy aligns operator axes with the *leading* axes of the parameter grid and *appends*
missing axes (epgpy/common.py:273-303); `axes=` inserts size-1 axes in front of an
operator's own axes (common.py:337-347).  Everything here is NumPy-on-host bookkeeping:
no device work happens in this module.
"""
import logging
import os

import numpy as np

logging.basicConfig(level=os.environ.get("LOG_LEVEL", "WARN").upper())
LOGGER = logging.getLogger("epgpy_amd")


def isscalar(value):
    """True for anything without a length -- np.array(1) counts as scalar (common.py:236-242)"""
    try:
        len(value)
    except TypeError:
        return True
    return False


def get_shape(obj):
    """shape of an array or nested sequence (common.py:257-270)"""
    if hasattr(obj, "shape"):
        return tuple(obj.shape)
    if hasattr(obj, "__len__"):
        return (len(obj),) + get_shape(obj[0])
    return ()


def map_arrays(arrays=None, func=np.asarray, **kwargs):
    """convert the non-scalar members of a list/tuple/dict with `func` (common.py:137-153)"""
    def conv(value):
        return value if isscalar(value) else func(value)

    arrays = kwargs if arrays is None else arrays
    if isinstance(arrays, (list, tuple)):
        return type(arrays)(conv(a) for a in arrays)
    if isinstance(arrays, dict):
        return {key: conv(val) for key, val in arrays.items()}
    return conv(arrays)


def expand_shapes(*shapes, append=False):
    ndim = max(len(s) for s in shapes)
    if append:
        return [tuple(s) + (1,) * (ndim - len(s)) for s in shapes]
    return [(1,) * (ndim - len(s)) + tuple(s) for s in shapes]


def broadcastable(*shapes, append=False):
    """can the shapes be expanded (prepend / append) and broadcast? (common.py:284-287)"""
    shapes = expand_shapes(*shapes, append=append)
    return all(len(set(dims) - {1}) <= 1 for dims in zip(*shapes))


def broadcast_shapes(*shapes, append=False):
    """shape after broadcasting, new axes prepended or appended (common.py:290-303)"""
    shapes = expand_shapes(*shapes, append=append)
    out = []
    for dims in zip(*shapes):
        sizes = set(dims) - {1}
        if len(sizes) > 1:
            raise ValueError(f"Incompatible shapes: {shapes}")
        out.append(sizes.pop() if sizes else 1)
    return tuple(out)


def expand_arrays(*objs, append=False):
    """give every array argument the same ndim (scalars pass through) (common.py:306-334)"""
    if not objs:
        return objs
    shapes = [get_shape(o) for o in objs]
    if not broadcastable(*shapes, append=append):
        raise ValueError("ArrayTuple cannot be broadcast to a single shape")
    ndim = max(len(s) for s in shapes)
    out = []
    for obj, shape in zip(objs, shapes):
        if not shape:
            out.append(obj)
            continue
        arr = np.asarray(obj)
        if len(shape) != ndim:          # (a reshape: np.expand_dims costs several microseconds per call, and operator
            arr = arr.reshape(shape + (1,) * (ndim - len(shape)) if append else (1,) * (ndim - len(shape)) + shape)   # construction calls this per parameter)
        out.append(arr)
    return tuple(out)


def set_axes(ntail, arr, axes):
    """place an operator's own axes at grid axes `axes` by inserting size-1 axes
    (common.py:337-347); `ntail` = trailing coefficient axes of `arr`"""
    nown = arr.ndim - ntail
    if isinstance(axes, (int, np.integer)):
        axes = tuple(range(int(axes), int(axes) + nown))
    elif not isinstance(axes, tuple) or not all(isinstance(ax, (int, np.integer)) for ax in axes):
        raise ValueError(f"Invalid axes: {axes}")
    newdims = tuple(i for i in range(max(axes)) if i not in axes)
    return np.expand_dims(arr, newdims)


def extend_operators(ntail, *ops):
    """give coefficient arrays [*opshape, *tail] the same number of operator axes by appending
    size-1 axes before the `ntail` trailing coefficient axes (common.py:350-364)"""
    shapes = [get_shape(op)[:-ntail] for op in ops if op is not None]
    ndim = len(broadcast_shapes(*shapes, append=True))
    out = []
    for op in ops:
        if op is None:
            out.append(None)
            continue
        op = np.asarray(op)
        lead = op.shape[:-ntail]
        out.append(op.reshape(lead + (1,) * (ndim - len(lead)) + op.shape[op.ndim - ntail:]))
    return out


def repr_value(value, fmt):
    if isscalar(value):
        return f"{value:{fmt}}"
    return "(" + "x".join(map(str, get_shape(value))) + ")"


def repr_operator(cls, names, values, fmts=None):
    """'T(alpha=.., phi=..)'-style default operator names (common.py:371-384)"""
    fmts = fmts or [""] * len(names)
    args = []
    for name, value, fmt in zip(names, values, fmts):
        if value is None:
            continue
        text = repr_value(value, fmt)
        args.append(f"{name}={text}" if name else text)
    return f"{cls}({', '.join(args)})"


def asnumpy(arr, copy=False):
    """host copy of a (nested) array-like -- device objects expose __array__"""
    if isinstance(arr, (list, tuple)):
        return type(arr)(asnumpy(a, copy=copy) for a in arr)
    return np.array(arr, copy=True) if copy else np.asarray(arr)
