"""Probe operators (mirrors epgpy/probe.py:7-165, :223).

`ADC` (= `Adc("F0")`) is the hot-path read-out: inside the fused kernel lane 0 of each
wavefront stores F_0 (or Z_0) of its voxel into the signal buffer, so nothing is copied
per ADC.  Weights / reduction / phase compensation (probe.py:141-165) are applied on the
host to the downloaded signal.  General probes (eval'd expressions, callables, "F", "Z",
...) need the whole state on the host: `simulate` then falls back to segment-wise execution
and evaluates them on a downloaded StateMatrix view.
"""
import numpy as np

from . import common, operator

SM_LOCALS = ["nstate", "ndim", "kdim", "states", "coords", "F", "F0", "F0t", "Z", "Z0", "k", "t", "t0"]
DEVICE_KINDS = {"F0": 0, "Z0": 1}


class _LazyAttrs(dict):
    """names resolved from a StateMatrix only when the expression touches them"""

    def __init__(self, sm, extra):
        super().__init__(extra)
        self._sm = sm

    def __missing__(self, key):
        if key in SM_LOCALS:
            return getattr(self._sm, key)
        raise KeyError(key)


class Probe(operator.EmptyOperator):
    """records data; does not modify the state (probe.py:7-79)"""

    SM_LOCALS = SM_LOCALS

    def __init__(self, obj, *args, post=None, **kwargs):
        if isinstance(obj, str):
            self._expr = obj
            self._acquire = self._acquire_expr
        elif callable(obj):
            self._callable = obj
            self._acquire = self._acquire_callable
        else:
            raise TypeError(f"Invalid probe: {obj}")
        self._args, self._kwargs = args, kwargs
        self._post = post
        self._repr = f"'{obj}'"
        super().__init__()

    def _device_kind(self):
        """0 / 1 if the kernel can record this probe itself (F0 / Z0), else None"""
        expr = getattr(self, "_expr", None)
        if expr is not None and expr.strip() in DEVICE_KINDS and not self._kwargs:
            return DEVICE_KINDS[expr.strip()]
        return None

    def _finish(self, raw):
        """turn the raw device record into what `_acquire` would have returned"""
        return raw

    def _is_plain(self):
        """True if neither `_finish` nor `post` changes a device record"""
        return (type(self)._finish is Probe._finish and type(self).post is Probe.post
                and not getattr(self, "_post", None) and not hasattr(self, "_assemble"))

    def _acquire_expr(self, sm):
        return eval(self._expr, vars(np), _LazyAttrs(sm, self._kwargs))  # noqa: S307 (reference semantics)

    def _acquire_callable(self, sm):
        return self._callable(sm, *self._args, **self._kwargs)

    def acquire(self, sm, post=None):
        post = post if post else self.post
        return post(common.asnumpy(self._acquire(sm), copy=True))

    def post(self, obj):
        if not getattr(self, "_post", None):
            return obj
        return self._post(obj)

    def __call__(self, sm, **kwargs):
        return sm

    def __repr__(self):
        return self.name or f"Probe({self._repr})"


def _trailing(arr, ndim):
    """`arr` with size-1 axes appended up to `ndim` axes (parameters align with the LEADING grid axes)"""
    arr = np.asarray(arr)
    return arr.reshape(arr.shape + (1,) * (ndim - arr.ndim)) if (arr.size > 1 and arr.ndim < ndim) else arr


def _reduction_axes(reduce, weights):
    """normalised `reduce` argument of Adc: None / False (keep everything), True (sum everything) or a tuple of axes.
    With weights and no explicit `reduce`, the sum runs over the axes of the weights (probe.py:110-131)."""
    if reduce is not None and reduce is not True and reduce:
        axes = (reduce,) if isinstance(reduce, int) else tuple(reduce)
        if any(not isinstance(ax, int) for ax in axes):
            raise ValueError(f"Expected (tuple of) int, got: {axes}")
        reduce = axes
    if weights is not None:
        span = max(weights.ndim, 1)
        if reduce is None:
            reduce = tuple(range(span))
        elif reduce is not True and reduce and not set(reduce) <= set(range(span)):
            raise ValueError(f"Invalid reduce dimension(s): {reduce}")
    return reduce


class Adc(Probe):
    """probe of one StateMatrix attribute, optionally weighted, summed over grid axes and phase-compensated
    (probe.py:82-165).  Three independent post-processing steps, applied in this order:
        record * weights  ->  sum over `reduce`  ->  * exp(i phase)          (phase in degrees, applied by `post`)
    F0 / Z0 records come straight out of the kernel; weights + reduce then run on the device (epgx_signal_reduce)."""

    def __init__(self, attr="F0", *, phase=None, reduce=None, weights=None, name="ADC"):
        if attr not in self.SM_LOCALS:
            raise ValueError(f"Invalid StateMatrix attribute: {attr}")
        self.attr = attr
        self.weights = None if weights is None else np.asarray(weights)
        self.reduce = _reduction_axes(reduce, self.weights)
        self.phase = None if phase is None else np.asarray(phase)
        if self.phase is not None:
            self.phasor = np.exp(1j * self.phase / 180 * np.pi)
        self._repr = attr if phase is None else f"'{attr}', {common.repr_value(phase, '.1f')}"
        operator.Operator.__init__(self, name=name)

    def _device_kind(self):
        return DEVICE_KINDS.get(self.attr)

    def _is_plain(self):
        return self.weights is None and (self.reduce is None or self.reduce is False) and self.phase is None

    def _device_reduction(self, grid):
        """(reduce mask over the grid axes, weights or None) if the device can do `_finish`
        (epgx_signal_reduce), else None: then the record is downloaded and finished on the host"""
        if self.reduce is None or self.reduce is False or self.reduce == ():
            return None
        ndim = len(grid)
        if self.reduce is True:
            mask = [1] * ndim
        else:
            axes = (self.reduce,) if isinstance(self.reduce, int) else self.reduce   # reduce=0 stays an int (probe.py:113-118)
            axes = [ax + ndim if ax < 0 else ax for ax in axes]
            if any(ax < 0 or ax >= ndim for ax in axes) or len(set(axes)) != len(axes):
                return None          # let NumPy raise its own error on the host path
            mask = [1 if d in axes else 0 for d in range(ndim)]
        weights = self.weights
        if weights is not None:
            if weights.ndim > ndim or weights.dtype.kind not in "fciub" or any(
                    w not in (1, g) for w, g in zip(weights.shape, grid)):
                return None
        return mask, weights

    def _finish(self, arr):
        if self.weights is not None:
            arr = arr * _trailing(self.weights, arr.ndim)
        if self.reduce is None or self.reduce is False:
            return arr
        return arr.sum() if self.reduce is True else arr.sum(axis=self.reduce)

    def _acquire(self, sm):
        return self._finish(getattr(sm, self.attr))

    def _post(self, obj):
        arr = np.asarray(obj)
        return arr if self.phase is None else arr * _trailing(self.phasor, arr.ndim)


ADC = Adc(attr="F0", name="ADC")
