"""Operator protocol of the hot path (mirrors epgpy/operator.py:13-341).

An operator is a small immutable host object: it owns its coefficient table (NumPy, built
once in the constructor exactly like the reference builds `mat` / `arr`), knows its `shape`,
`nshift`, `duration` and `name`, and can be (a) called on a device-resident StateMatrix
(`op(sm, inplace=False)`), which launches the fused HIP kernel on a one-operator plan, or
(b) put in a (nested) list handed to `simulate()`, which compiles the whole list into one
plan and runs it state-resident on the GPU.
"""
import abc

import numpy as np

from . import common, _lib


class Operator(abc.ABC):
    """Base operator (epgpy/operator.py:13-113)"""

    def __init__(self, *, name=None, duration=None):
        if duration is None:
            duration = 0
        elif np.any(np.asarray(duration) < 0):
            raise ValueError("Cannot have duration < 0")
        self.duration = duration
        self.name = name if name else type(self).__name__

    # -- protocol --------------------------------------------------------------------
    @property
    def shape(self):
        return (1,)

    @property
    def ndim(self):
        return len(self.shape)

    @property
    def size(self):
        return int(np.prod(self.shape))

    @property
    def nshift(self):
        return 0

    def __repr__(self):
        return self.name

    def __mul__(self, other):
        return MultiOperator([self, other])

    @classmethod
    def from_list(cls, sequence):
        return MultiOperator(sequence)

    def _encode(self, enc):
        """append this operator's device record(s) to a plan encoder (see plan.py)"""
        raise NotImplementedError(f"{type(self).__name__} has no device encoding")

    def prepare(self, sm, inplace=False):
        """type / shape checks, copy unless inplace, expand ndim (operator.py:73-94)"""
        from .statematrix import StateMatrix

        if not isinstance(sm, StateMatrix):
            raise TypeError(f"Not a StateMatrix: {sm}")
        if not common.broadcastable(sm.shape, self.shape, append=True):
            raise ValueError(
                f"Incompatible StateMatrix and operator shapes: {sm.shape}, {self.shape}")
        if not inplace:
            sm = sm.copy()
        if sm.ndim < self.ndim:
            sm.expand(self.ndim)
        return sm

    def _apply(self, sm):
        """run this operator on the device state of `sm` (in place)"""
        from .plan import apply_operators

        return apply_operators(sm, [self])

    def __call__(self, sm, *, inplace=False):
        sm = self.prepare(sm, inplace=inplace)
        return self._apply(sm)

    def copy(self, name=None, duration=None):
        new = self.__new__(type(self))
        new.__dict__.update(self.__dict__)
        new.name = name or self.name
        new.duration = duration or self.duration
        return new


class MultiOperator(Operator):
    """A sequence of operators seen as one (operator.py:118-203)"""

    def __init__(self, operators=None, *, name=None, duration=None):
        operators = [] if not operators else list(operators)
        self._nshift = 0
        self._shape = (1,)
        self.operators = []
        self.duration = 0
        for op in operators:
            self.append(op)
        if not name:
            name = " | ".join(op.name for op in operators)
        if duration is None:
            duration = self.duration
        super().__init__(name=name, duration=duration)

    @property
    def shape(self):
        return self._shape

    @property
    def nshift(self):
        return self._nshift

    def __iter__(self):
        return iter(self.operators)

    def __len__(self):
        return len(self.operators)

    def __getitem__(self, i):
        return self.operators[i]

    def __mul__(self, other):
        self.append(other)
        return self

    def append(self, op):
        if not isinstance(op, Operator):
            raise TypeError("Invalid operator: %s" % str(op))
        shape = common.broadcast_shapes(self.shape, op.shape, append=True)
        if isinstance(op, MultiOperator):
            self.operators.extend(op.operators)
        else:
            self.operators.append(op)
        self._shape = shape
        self._nshift += op.nshift
        self.duration += op.duration

    def _encode(self, enc):
        for op in self.operators:
            op._encode(enc)

    def _apply(self, sm):
        from .plan import apply_operators

        return apply_operators(sm, list(self.operators))


class CombinableOperator(Operator, abc.ABC):
    """operators that can be merged with `@` (operator.py:206-241): `op1 @ op2` is ONE operator
    equivalent to applying op1 then op2"""

    @abc.abstractmethod
    def combinable(self, other):
        pass

    @classmethod
    @abc.abstractmethod
    def _combine(cls, op1, op2, **kwargs):
        pass

    def combine(self, other, *, right=False, name=None, duration=None, **kwargs):
        if not isinstance(other, CombinableOperator):
            raise TypeError(f"Non-combinable operator: {other}")
        if not self.combinable(other):
            return NotImplemented
        op1, op2 = (other, self) if right else (self, other)
        if name is None:
            name = f"{op1.name}|{op2.name}"
        if duration is None:
            duration = op1.duration + op2.duration
        return self._combine(op1, op2, name=name, duration=duration, **kwargs)

    def __matmul__(self, other):
        return self.combine(other)

    def __rmatmul__(self, other):
        return self.combine(other, right=True)


class EmptyOperator(Operator):
    """does nothing (operator.py:248-252)"""

    def _encode(self, enc):
        pass

    def _apply(self, sm):
        return sm


NULL = EmptyOperator(name="NULL")


class Wait(EmptyOperator):
    """empty operator with a duration (operator.py:259-265)"""

    def __init__(self, duration, name=None):
        name = name if name is not None else f"Wait({duration})"
        super().__init__(duration=duration, name=name)


class Offset(EmptyOperator):
    """empty operator with a possibly negative duration (operator.py:268-274)"""

    def __init__(self, duration, name=None):
        name = name if name is not None else f"Offset({duration})"
        super().__init__(duration=abs(duration), name=name)
        self.duration = duration


class Spoiler(Operator):
    """perfect spoiler: transverse magnetisation <- 0 (operator.py:281-286)"""

    def _encode(self, enc):
        enc.add(_lib.OP_SPOIL)
        enc.note("spoil")


SPOILER = Spoiler(name="Spoiler")


class Reset(Operator):
    """back to equilibrium, nstate <- 0 (operator.py:297-304)"""

    def _encode(self, enc):
        enc.add(_lib.OP_RESET)
        enc.nstate = 0
        enc.note("reset")


RESET = Reset(name="Reset")


class PD(Operator):
    """set the proton density, optionally reset to the new equilibrium (operator.py:315-341)"""

    def __init__(self, pd, *, reset=True, name=None, **kwargs):
        self.pd = common.map_arrays(pd=pd)["pd"]
        self.reset = reset
        if name is None:
            name = common.repr_operator("PD", ["pd"], [self.pd], [".1f"])
        super().__init__(name=name, **kwargs)

    @property
    def shape(self):
        return getattr(self.pd, "shape", (1,))

    def _encode(self, enc):
        table = np.atleast_1d(np.asarray(self.pd, dtype=np.float64))[..., None]
        enc.add(_lib.OP_PD, table=table, key=("PD", id(self)), ia=1 if self.reset else 0)
        if self.reset and enc.kspace is not None:   # states <- new equilibrium, coordinates kept
            ks = enc.kspace
            enc.kspace = type(ks)(ks.coords, [False] * ks.nrow, [i == ks.centre for i in range(ks.nrow)])
