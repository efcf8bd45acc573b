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


# ------------------------------------------------------------------ which kernel a configuration takes
def test_kernel_of_every_baseline_config():
    """epgx_kernel_for: the ONE decision function of the library (choose_kernel, epgx_api.hip) pinned for every
    BASELINE.json configuration, their Jacobian variants and the capacity classes -- on small grids: the decision depends on
    the plan, the range, the capacity and on whether states are given, never on the number of voxels"""
    from epgpy_amd import _lib, functions, workloads as wl

    ctx = _lib.get_context()

    def compiled(seq, options=None, variables=()):
        return functions.compile_sequence(seq, None, options=options or {}, variables=variables)

    # configs[0] README MSE, configs[1] MSE grid: the echo train from equilibrium grows through 16 / 32 / 64 orders; per-timestep
    # mode: the first launch starts from equilibrium and writes the state, the others stream it through HBM
    for T1, T2 in ((150.0, [30.0, 40.0, 50.0]), (np.linspace(200, 3000, 8)[:, None], np.linspace(20, 300, 8)[None, :])):
        enc, _, bounds = compiled(wl.mse_sequence(epg, T1, T2), {"max_nstate": 63})
        plan = enc.device_plan(ctx, 64)
        assert enc.capacity() == 64 and _lib.kernel_for(ctx, plan, 64) == "rows_grow_kernel<1>"
        st = _lib.DeviceState(ctx, enc.nvox, 64)
        assert _lib.kernel_for(ctx, plan, 64, 0, bounds[0], None, st) == "run_kernel<1, 1, false>"
        assert _lib.kernel_for(ctx, plan, 64, bounds[0], bounds[1], st, st) == "run_kernel<1, 1, true>"
    # configs[2] / [3]: 1000-TR MRF (30 of 1000 repetitions below 64 orders: no phases), repetitions folded at run time; its
    # short-state-matrix variants (max_nstate = 10 / 20) at 16 / 32 orders
    T1, T2, B1 = np.linspace(300, 3000, 3)[:, None, None], np.linspace(20, 300, 3)[None, :, None], np.linspace(0.7, 1.3, 3)[None, None, :]
    alpha, TR = wl.mrf_trains(1000)
    enc, _, _ = compiled(wl.mrf_sequence(epg, T1, T2, B1, alpha, TR), {"max_nstate": 63})
    assert _lib.kernel_for(ctx, enc.device_plan(ctx, 64), 64) == "rows_kernel<2, 4, true>"
    for cap, KP in ((10, 16), (20, 32)):
        enc, _, _ = compiled(wl.mrf_sequence(epg, T1, T2, B1, alpha[:50], TR[:50]), {"max_nstate": cap})
        assert enc.packable() == KP and _lib.kernel_for(ctx, enc.device_plan(ctx, enc.capacity()), KP) == f"rows_kernel<2, {KP // 16}, true>"
    # configs[4]: PGSE with 3-D shifts and diffusion, at most 7 orders: 16 lanes per voxel
    seq5, _, _, opts5 = wl.build(epg, "pgse_512")
    enc5, _, _ = compiled(seq5, opts5)
    assert enc5.packable_nd() == 16 and _lib.kernel_for(ctx, enc5.device_plan(ctx, 16), 16) == "rows_kernel<2, 1, false>"
    # Jacobians of the MSE train (fused echoes with logarithmic relaxation partials: shape 309) and of the MRF train (repetitions
    # folded at run time: shape 154; three derivative states of folded runs: the last variable alone, then the first two)
    T1j, T2j = np.linspace(200, 3000, 6)[:, None], np.linspace(20, 300, 6)[None, :]
    rlx = epg.E(5.0, T1j, T2j, order1=["T1", "T2"])
    seqj = [epg.T(90, 90, order1={"B1": {"alpha": 90}})] + [epg.S(1), rlx, epg.T(120, 0, order1={"B1": {"alpha": 120}}), epg.S(1), rlx, epg.ADC] * 20
    seqm = [epg.T(180 * B1, 90, order1={"B1": {"alpha": 180.0}}), epg.E(20.0, T1, T2, order1=["T1", "T2"])]
    for a_, tr_ in zip(alpha[:40], TR[:40]):
        seqm += [epg.T(a_ * B1, 90, order1={"B1": {"alpha": float(a_)}}), epg.E(3.0, T1, T2, order1=["T1", "T2"]), epg.ADC,
                 epg.E(tr_ - 3.0, T1, T2, order1=["T1", "T2"]), epg.S(1)]
    for names in (["T2"], ["T2", "T1"], ["T2", "T1", "B1"]):
        V = len(names)
        enc, _, _ = compiled(seqj, {"max_nstate": 63}, names)
        assert _lib.kernel_for(ctx, enc.device_plan(ctx, 64), 64) == f"drun_kernel<4, {V}, 309, 0>"
        enc, _, _ = compiled(seqm, {"max_nstate": 63}, names)
        want = f"drun_kernel<4, {V}, 154, 0>" if V < 3 else "drun_kernel<4, 1, 154, 2> + drun_kernel<4, 2, 154, 0>"
        assert _lib.kernel_for(ctx, enc.device_plan(ctx, 64), 64) == want
    # capacity classes: long state matrices from equilibrium (trains that grow all the way: the growing kernels), from a state
    # buffer, and with a state output
    T1c, T2c = np.linspace(200, 3000, 4)[:, None], np.linspace(20, 300, 4)[None, :]
    expected = {128: ("rows_kernel<1, 8, false>", "run_contig_kernel<2, 1, true>", "run_kernel<2, 1, true>"),
                256: ("run_contig_grow_kernel<4, 1>", "run_contig_kernel<4, 1, true>", "run_kernel<4, 1, true>"),
                512: ("run_contig_grow_kernel<8, 1>", "run_contig_kernel<8, 1, true>", "run_kernel<8, 1, true>"),
                1024: ("run_contig_grow_kernel<16, 1>", "run_contig_kernel<16, 1, true>", "run_kernel<16, 1, true>"),
                2048: ("run_kernel<8, 1, false> + run_split_kernel<4, 1, true>", None, None)}
    for K, (resident, streamed, in_out) in expected.items():
        enc, _, _ = compiled(wl.mse_sequence(epg, T1c, T2c, necho=K // 2 - 4))
        assert enc.capacity(resident=True) == K
        plan = enc.device_plan(ctx, K)
        assert _lib.kernel_for(ctx, plan, K) == resident
        if streamed:
            st = _lib.DeviceState(ctx, enc.nvox, K)
            assert _lib.kernel_for(ctx, plan, K, 0, plan.n_ops, st, None) == streamed
            assert _lib.kernel_for(ctx, plan, K, 0, plan.n_ops, st, st) == in_out
