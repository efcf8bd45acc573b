"""GPU: the drop-in boundary as a user of the reference meets it -- the operator protocol with USER-WRITTEN
operators (subclasses bringing their own `_apply(sm)` on `sm.states`: operator.py:13-113, the shape of
test/test_operator.py's custom operator), operator algebra, probes given as callables / expressions / `Adc` options
(probe.py:7-165, known answers of test/test_probe.py), and such operators and probes inside `simulate()`."""
import numpy as np
import pytest

from epgpy_amd import epg, operator as eop

pytestmark = pytest.mark.gpu


class Grow(eop.Operator):
    """user-written: broadcasts the state matrix to the operator's own footprint, touches no value"""

    def __init__(self, nshift=0, shape=(1,), **kw):
        self._nshift, self._shape = nshift, tuple(shape)
        super().__init__(**kw)

    def _apply(self, sm):
        shape = self.shape + (1,) * max(0, sm.ndim - self.ndim)
        shape = np.broadcast_shapes(shape, sm.shape)
        if sm.shape != shape:
            sm.states = np.broadcast_to(sm.states, shape + sm.states.shape[-2:])
        return sm

    nshift = property(lambda self: self._nshift)
    shape = property(lambda self: self._shape)


class Damp(eop.Operator):
    """user-written: scales the transverse states by a per-voxel factor (NumPy on the downloaded state matrix)"""

    def __init__(self, factor, **kw):
        self.factor = np.asarray(factor, dtype=float)
        super().__init__(**kw)

    shape = property(lambda self: self.factor.shape or (1,))

    def _apply(self, sm):
        states = sm.states
        states[..., :2] *= np.reshape(self.factor, self.shape + (1,) * (sm.ndim - len(self.shape)) + (1, 1))
        sm.states = states
        return sm


def test_operator_attributes_and_products():
    op = eop.EmptyOperator()
    assert (op.name, op.duration, op.ndim, op.shape, op.nshift) == ("EmptyOperator", 0, 1, (1,), 0)
    op = eop.EmptyOperator(name="gap", duration=1)
    assert op.name == "gap" and op.duration == 1
    a, b = eop.EmptyOperator(name="a", duration=1), eop.EmptyOperator(name="b", duration=2)
    ab = a * b
    assert isinstance(ab, eop.MultiOperator) and ab.operators == [a, b]
    twin = Grow(name="g", duration=1).copy()
    assert isinstance(twin, Grow) and twin.name == "g" and twin.duration == 1


def test_user_written_operator_broadcasts_the_state_matrix():
    sm = epg.StateMatrix(shape=(3, 1))
    assert Grow(shape=[1])(sm, inplace=True) is sm and sm.shape == (3, 1)
    assert Grow(shape=[1, 2])(sm, inplace=False).shape == (3, 2) and sm.shape == (3, 1)      # (a copy grew)
    Grow(shape=[1, 2])(sm, inplace=True)
    assert sm.shape == (3, 2) and np.allclose(sm.Z0, 1) and sm.density.shape == (3, 2)
    assert Grow(shape=[3, 2, 2])(sm, inplace=False).shape == (3, 2, 2) and sm.shape == (3, 2)
    Grow(shape=[3, 2, 2])(sm, inplace=True)
    assert sm.shape == (3, 2, 2)
    sm = epg.StateMatrix(shape=(3, 1))
    for bad in ([2, 1], [4, 4], [2, 3, 2]):
        with pytest.raises(ValueError):
            Grow(shape=bad)(sm, inplace=False)
    # the states setter itself: new states may widen the grid, never contradict it
    sm = epg.StateMatrix([1, 1, 0.5], nstate=1, shape=(3, 1))
    sm.states = np.broadcast_to(sm.states, (3, 4, 3, 3))
    assert sm.shape == (3, 4) and np.allclose(sm.F0, 1) and np.allclose(sm.Z0, 0.5)
    with pytest.raises(ValueError):
        sm.states = np.zeros((2, 4, 3, 3))


def test_user_written_operators_in_a_multioperator():
    a, b = Grow(name="a", duration=1, shape=(1, 3), nshift=2), Grow(name="b", duration=2, shape=(2, 1), nshift=1)
    ab = eop.MultiOperator([a, b], name="both")
    assert ab.operators == [a, b] and ab.duration == 3 and ab.nshift == 3 and ab.shape == (2, 3)
    assert (a * b).operators == ab.operators
    for x, y in (((2, 1), (3,)), ((2, 1), (3, 1))):
        with pytest.raises(ValueError):
            Grow(shape=x) * Grow(shape=y)
    sm = epg.StateMatrix(shape=(2, 3))
    assert ab(sm, inplace=True) is sm
    assert ab(epg.StateMatrix(shape=(2,))).shape == (2, 3) and ab(epg.StateMatrix(shape=(2, 3, 4))).shape == (2, 3, 4)
    with pytest.raises(ValueError):
        ab(epg.StateMatrix(shape=(3,)))
    # library operators around a user-written one: launched before / after it, same result as without the (value-neutral) one
    T2 = np.array([[50.0, 80.0]])
    mixed = eop.MultiOperator([epg.T(40, 10), epg.E(5, 1000, 60), Grow(shape=(1, 2)), epg.E(5, 1000, T2), epg.S(1), epg.T(70, 0)])
    plain = eop.MultiOperator([epg.T(40, 10), epg.E(5, 1000, 60), epg.E(5, 1000, T2), epg.S(1), epg.T(70, 0)])
    got, ref = mixed(epg.StateMatrix(shape=(3,))), plain(epg.StateMatrix(shape=(3, 1)))
    assert got.shape == ref.shape == (3, 2) and np.array_equal(got.states, ref.states)


def test_combinable_operator_algebra():
    class Plain(eop.Operator):
        def _apply(self, sm):
            return sm

    class Fusable(eop.CombinableOperator):
        def _apply(self, sm):
            return sm

        @classmethod
        def combinable(cls, other):
            return True

        @classmethod
        def _combine(cls, op1, op2, **kwargs):
            return Fusable(**kwargs)

    a, b = Fusable(name="a", duration=1), Fusable(name="b", duration=2)
    ab = a @ b
    assert ab.name == "a|b" and ab.duration == 3
    ab = b.combine(a, right=True, duration=2)
    assert ab.name == "a|b" and ab.duration == 2
    with pytest.raises(TypeError):
        a @ Plain(name="p")


def test_spoiler_reset_and_density_operators():
    sm0 = epg.StateMatrix(0.5 * np.ones((3, 3)))
    sm1 = epg.SPOILER(sm0)
    assert np.allclose(sm0.F, 0.5) and np.allclose(sm1.F, 0) and np.allclose(sm1.Z, 0.5)
    sm1 = epg.RESET(sm0)
    assert np.allclose(sm0.F, 0.5) and np.allclose(sm1.states, sm1.equilibrium)
    sm = epg.PD(2, reset=False)(epg.StateMatrix([1, 1, 0]))
    assert np.allclose(sm.density, 2) and np.allclose(sm.equilibrium, [0, 0, 2]) and np.allclose(sm.states, [[1, 1, 0]])
    sm0 = epg.StateMatrix(shape=(2,))
    sm = epg.PD([2, 3])(sm0)
    assert np.allclose(sm.density, [2, 3]) and np.allclose(sm.equilibrium, [[[0, 0, 2]], [[0, 0, 3]]])
    assert np.allclose(sm.states, [[[0, 0, 2]], [[0, 0, 3]]])          # (reset=True is the default)
    with pytest.raises(ValueError):
        epg.PD([2, 3, 4])(sm0)


def test_probes_given_as_callables_and_expressions():
    sm = epg.StateMatrix([1, 1, 0.5], nstate=1)
    probe = epg.Probe(lambda sm: sm.F0)
    assert probe(sm, inplace=True) is sm and np.allclose(probe.acquire(sm), [1])
    probe = epg.Probe(lambda sm: sm.F)
    first = probe.acquire(sm)
    assert np.allclose(first, [0, 1, 0])
    states = sm.states
    states[:, 0], states[:, 2] = [0.5, 0.5, 0], [0.5, 0.5, 0]
    sm.states = states
    assert np.allclose(first, [0, 1, 0]) and np.allclose(probe.acquire(sm), [0.5, 1, 0.5])    # (acquisitions are copies)
    sm = epg.StateMatrix([1, 1, 0.5], nstate=1, shape=(3, 2))
    for name in ("F", "F0", "Z", "Z0"):
        assert np.allclose(epg.Probe(name).acquire(sm), getattr(sm, name)), name
    probe = epg.Probe("F0.mean(axes.T2)", axes=epg.Axes("T2", "B1"))
    sm = epg.StateMatrix(shape=(3, 4))
    assert sm.F0.shape == (3, 4) and probe.acquire(sm).shape == (4,)


def test_adc_options_known_answers():
    sm = epg.StateMatrix([[[1j, -1j, 0.5]], [[-1j, 1j, 0.5]]])
    assert np.allclose(epg.Adc().acquire(sm), [1j, -1j])
    adc = epg.Adc(phase=90)
    assert np.allclose(adc.acquire(sm), [-1, 1]) and np.allclose(adc.post(epg.ADC.acquire(sm)), [-1, 1])
    assert np.allclose(epg.Adc(phase=90, weights=[2, 0.5], reduce=0).acquire(sm), -1.5)
    assert np.allclose(epg.Adc(phase=90, weights=1).acquire(sm), 0)
    assert np.allclose(epg.Adc(phase=90, weights=[2, 0.5], reduce=False).acquire(sm), [-2, 0.5])
    with pytest.raises(ValueError):
        epg.Adc(weights=[2, 0.5], reduce=1)


def test_user_written_operators_and_callable_probes_inside_simulate():
    """a sequence with a user-written operator runs operator by operator (the library's operators between two of them as
    one launch); the result equals the all-library sequence where the user's arithmetic has a library equivalent"""
    T2 = np.array([40.0, 70.0, 110.0])
    factor = np.array([0.5, 0.8, 1.0])
    train = [epg.S(1), epg.E(5, 900, T2), epg.T(150, 0), epg.S(1), epg.E(5, 900, T2), epg.ADC]
    seq = [epg.T(90, 90)] + train * 2 + [Damp(factor)] + train * 3
    got = epg.simulate(seq)
    assert got.shape == (5, 3)
    # Damp == a relaxation that only shrinks the transverse states: E(tau, inf, T2') with exp(-tau / T2') = factor
    with np.errstate(divide="ignore"):
        twin = epg.E(1.0, 1e300, -1.0 / np.log(np.minimum(factor, 1 - 1e-16)))
    ref = epg.simulate([epg.T(90, 90)] + train * 2 + [twin] + train * 3)
    assert np.allclose(got, ref, rtol=0, atol=1e-14)
    with pytest.raises(ValueError):
        epg.simulate(seq, mode="resident")                 # (plans hold library operators only)
    # grid-widening operator + a callable probe next to ADC
    seq = [epg.T(30, 0), Grow(shape=(1, 2)), epg.E(5, 1000, [[50, 80]]), epg.ADC, epg.Probe(lambda sm: sm.Z0.real)]
    sig = epg.simulate(seq, asarray=False)
    ref = epg.simulate([epg.T(30, 0), epg.E(5, 1000, [[50, 80]]), epg.ADC, epg.Probe("Z0")], asarray=False)
    assert len(sig) == 2 and np.shape(sig[0]) == (1, 2)
    assert np.allclose(sig[0], ref[0]) and np.allclose(sig[1], np.real(ref[1]))
