"""Operator protocol of the hot path (interface of epgpy/operator.py:13-341, re-designed around the plan compiler).

An operator here is a *description*: a named node that knows

  * its footprint on the parameter grid (`shape`), the number of phase-state shifts it causes (`nshift`) and how
    long it lasts (`duration`) -- the three facts `simulate()` needs before anything runs on the device;
  * `_parts()`: the primitive operators it stands for (itself, its members, or nothing at all), and
  * `_encode(enc)`: how a primitive appends its record(s) and coefficient table to a plan (`plan.Encoder`).

Execution is not an operator method: `op(sm)` hands `op._parts()` to `plan.apply_operators`, which compiles them into
ONE launch of the fused HIP kernel over the device-resident state matrix (the per-timestep path), and `simulate()`
compiles a whole (nested) list the same way for a state-resident run.  The reference's names, argument meaning,
return conventions and exceptions are kept so that its scripts and tests read unchanged:
`op.shape / ndim / size / nshift / duration / name`, `op(sm, inplace=False)`, `op1 * op2`, `op1 @ op2`,
`NULL`, `Wait`, `Offset`, `SPOILER`, `RESET`, `PD`.
"""
import numpy as np

from . import common, _lib


def _lasting(duration, allow_negative=False):
    """validated duration of an operator (None: instantaneous)"""
    if duration is None:
        return 0
    if not allow_negative and np.any(np.asarray(duration) < 0):
        raise ValueError("Cannot have duration < 0")
    return duration


class Operator:
    """node of a sequence; subclasses say what they are through `shape`, `nshift`, `_parts` and `_encode`"""

    PASSIVE = False        # True: stands for no device work at all (probes, delays)

    def __init__(self, *, name=None, duration=None):
        self.duration = _lasting(duration)
        self.name = name or type(self).__name__

    # ---- facts the plan compiler asks for -------------------------------------------------
    shape = (1,)           # footprint on the parameter grid (leading axes; overridden as a property where it varies)
    nshift = 0             # |k| summed over the shifts this operator stands for

    @property
    def ndim(self):
        return len(self.shape)

    @property
    def size(self):
        return int(np.prod(self.shape))

    def _parts(self):
        """the primitive operators this node stands for, in order"""
        return [] if self.PASSIVE else [self]

    def _encode(self, enc):
        """append this primitive's record(s) to a plan encoder (plan.py)"""
        if not self.PASSIVE:
            raise NotImplementedError(f"{type(self).__name__} has no device encoding")

    def _on_host(self):
        """a user-written operator in the reference's style (operator.py:13-113, test/test_operator.py:10-41): a subclass
        that brings its own `_apply(sm)` -- Python working on `sm.states` -- and no device encoding.  Plans cannot hold it:
        `plan.apply_operators` launches what stands before it, hands it the state matrix, and goes on; `simulate()` takes
        the operator-by-operator loop for sequences that contain one."""
        cls = type(self)
        return not self.PASSIVE and cls._apply is not Operator._apply and cls._encode is Operator._encode

    # ---- the reference's calling conventions ----------------------------------------------
    def prepare(self, sm, inplace=False):
        """the state matrix this operator may write to: type and shape checks, a copy unless `inplace`, trailing grid
        axes added when the operator has more of them (operator.py:73-94)"""
        from .statematrix import StateMatrix

        if not isinstance(sm, StateMatrix):
            raise TypeError(f"Not a StateMatrix: {sm}")
        if not common.broadcastable(sm.shape, self.shape, append=True):
            raise ValueError(f"Incompatible StateMatrix and operator shapes: {sm.shape}, {self.shape}")
        target = sm if inplace else sm.copy()
        if target.ndim < self.ndim:
            target.expand(self.ndim)
        return target

    def _apply(self, sm):
        """run the primitives of this node on the device state of `sm`, in place: one fused launch"""
        parts = self._parts()
        if not parts:
            return sm
        from .plan import apply_operators

        return apply_operators(sm, parts)

    def __call__(self, sm, *, inplace=False):
        return self._apply(self.prepare(sm, inplace=inplace))

    def __mul__(self, other):
        return MultiOperator([self, other])

    @classmethod
    def from_list(cls, sequence):
        return MultiOperator(sequence)

    def copy(self, name=None, duration=None):
        twin = self.__new__(type(self))
        twin.__dict__.update(self.__dict__)
        twin.name, twin.duration = name or self.name, duration or self.duration
        return twin

    def __repr__(self):
        return self.name


class MultiOperator(Operator):
    """several operators used as one (operator.py:118-203): its footprint, shift count and duration are those of its
    members, which a plan sees flattened"""

    def __init__(self, operators=None, *, name=None, duration=None):
        self.operators = []
        self._shape, self._nshift, total = (1,), 0, 0
        for op in (operators or ()):
            total = total + self._take(op)
        label = name or " | ".join(op.name for op in (operators or ()))
        super().__init__(name=label, duration=total if duration is None else duration)

    def _take(self, op):
        """add one member (a MultiOperator is dissolved into its members); returns its duration"""
        if not isinstance(op, Operator):
            raise TypeError("Invalid operator: %s" % str(op))
        self._shape = common.broadcast_shapes(self._shape, op.shape, append=True)   # raises on incompatible shapes, before anything changes
        self.operators.extend(op.operators if isinstance(op, MultiOperator) else [op])
        self._nshift += op.nshift
        return op.duration

    def append(self, op):
        self.duration = self.duration + self._take(op)

    shape = property(lambda self: self._shape)
    nshift = property(lambda self: self._nshift)

    def __mul__(self, other):          # `seq * op` grows the sequence in place (operator.py:176-178)
        self.append(other)
        return self

    def __iter__(self):
        return iter(self.operators)

    def __len__(self):
        return len(self.operators)

    def __getitem__(self, index):
        return self.operators[index]

    def _parts(self):
        return [part for op in self.operators for part in op._parts()]

    def _encode(self, enc):
        for op in self.operators:
            op._encode(enc)


class CombinableOperator(Operator):
    """operators with an algebra: `op1 @ op2` is ONE operator equivalent to op1 followed by op2 (operator.py:206-241).
    Subclasses provide `combinable(other)` and the class method `_combine(op1, op2, **kwargs)`."""

    def combinable(self, other):
        raise NotImplementedError

    @classmethod
    def _combine(cls, op1, op2, **kwargs):
        raise NotImplementedError

    def combine(self, other, *, right=False, name=None, duration=None, **kwargs):
        if not isinstance(other, CombinableOperator):
            raise TypeError(f"Non-combinable operator: {other}")
        if not self.combinable(other):
            return NotImplemented
        first, second = (other, self) if right else (self, other)
        return self._combine(first, second, name=name if name is not None else f"{first.name}|{second.name}",
                             duration=first.duration + second.duration if duration is None else duration, **kwargs)

    def __matmul__(self, other):
        return self.combine(other)

    def __rmatmul__(self, other):
        return self.combine(other, right=True)


class EmptyOperator(Operator):
    """takes part in the timing of a sequence, never in its arithmetic (operator.py:248-252)"""
    PASSIVE = True


class Wait(EmptyOperator):
    """a delay (operator.py:259-265)"""

    def __init__(self, duration, name=None):
        super().__init__(duration=duration, name=f"Wait({duration})" if name is None else name)


class Offset(EmptyOperator):
    """a delay that may be negative: moves the clock of `get_adc_times` back (operator.py:268-274)"""

    def __init__(self, duration, name=None):
        super().__init__(name=f"Offset({duration})" if name is None else name)
        self.duration = _lasting(duration, allow_negative=True)


class _Marker(Operator):
    """a primitive without parameters: one record with an opcode, one note to the structural planner"""
    OPCODE, NOTE = None, None

    def _encode(self, enc):
        enc.add(self.OPCODE)
        enc.note(self.NOTE)


class Spoiler(_Marker):
    """perfect spoiler: transverse magnetisation <- 0 (operator.py:281-286)"""
    OPCODE, NOTE = _lib.OP_SPOIL, "spoil"


class Reset(_Marker):
    """back to equilibrium, nstate <- 0 (operator.py:297-304)"""
    OPCODE, NOTE = _lib.OP_RESET, "reset"

    def _encode(self, enc):
        super()._encode(enc)
        enc.nstate = 0


NULL = EmptyOperator(name="NULL")
SPOILER = Spoiler(name="Spoiler")
RESET = Reset(name="Reset")


class PD(Operator):
    """set the proton density, optionally reset to the new equilibrium (operator.py:315-341)"""

    def __init__(self, pd, *, reset=True, name=None, **kwargs):
        self.pd = common.map_arrays(pd=pd)["pd"]
        self.reset = reset
        super().__init__(name=name if name is not None else common.repr_operator("PD", ["pd"], [self.pd], [".1f"]), **kwargs)

    @property
    def shape(self):
        return getattr(self.pd, "shape", (1,))

    def _encode(self, enc):
        table = np.atleast_1d(np.asarray(self.pd, dtype=np.float64))[..., None]
        enc.add(_lib.OP_PD, table=table, key=("PD", id(self)), ia=1 if self.reset else 0)
        if self.reset and enc.kspace is not None:   # states <- new equilibrium, coordinates kept
            ks = enc.kspace
            enc.kspace = ks._like([False] * ks.nrow, [i == ks.centre for i in range(ks.nrow)])
