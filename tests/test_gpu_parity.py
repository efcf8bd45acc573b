"""GPU: parity of the HIP path (through the C ABI) with the reference.

Three kinds of checks:
  * against golden vectors produced by the reference itself (tests/golden/*.npz);
  * against the CPU oracle on seeded inputs (bit-for-bit-close: fp64, tolerance below);
  * the reference's own known-answer / contract tests, values copied from its test files.

Tolerance: north_star asks <= 1e-6 relative in complex magnitude; fp64 state + fp64
arithmetic deliver ~1e-15, so the tests assert TOL = 1e-12 (absolute, signals are O(1)).
"""
import os

import numpy as np
import pytest

from epgpy_amd import epg, _lib, functions
from oracle import epg_numpy as onp, epg_c
from tests import sequences as sq

pytestmark = pytest.mark.gpu
TOL = 1e-12


def close(a, b, tol=TOL):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = float(np.max(np.abs(a - b))) if a.size else 0.0
    assert err <= tol * max(1.0, float(np.max(np.abs(b))) if b.size else 1.0), err


def resident_k(ops, **options):
    """orders per voxel a state-resident simulate() of `ops` runs at (64: the rows kernel whose x rotations take the
    sum / difference form -- sq.same_bits(..., x64=True) when it is compared with another kernel)"""
    enc, _, _ = functions.compile_sequence(ops, options=options)
    return enc.packable() or enc.capacity()


def run_ops(seq, **options):
    """apply the operators one by one to a device StateMatrix (per-timestep path)"""
    sm = epg.StateMatrix(shape=epg.getshape(seq), **options)
    for op in epg.flatten_sequence(seq):
        sm = op(sm, inplace=True)
    return sm


# ------------------------------------------------------------------ golden vectors
@pytest.mark.parametrize("mode", ["resident", "stream", "stepwise"])
def test_g1_readme(golden, mode):
    g = golden("g1_readme_mse")
    seq = sq.mse_ops(epg, float(g["T1"]), g["T2"], FA=float(g["FA"]), ESP=float(g["ESP"]), necho=int(g["necho"]))
    times, sig = epg.simulate(seq, adc_time=True, mode=mode)
    assert sig.shape == (20, 3) and sig.dtype == np.complex128
    close(sig, g["signal"])
    close(times, g["times"])
    if mode == "resident":
        sm = run_ops(seq)
        assert sm.nstate == 40 and sm.shape == (3,)
        close(sm.states, g["states"])


@pytest.mark.parametrize("cap,tag", [(None, "unbounded"), (63, "cap63"), (10, "cap10")])
def test_g2_random_mse(golden, cap, tag):
    g = golden("g2_random_mse")
    opts = {} if cap is None else {"max_nstate": cap}
    seq = sq.mse_ops(epg, g["T1"], g["T2"], g["B1"])
    for mode in ("resident", "stream"):
        close(epg.simulate(seq, mode=mode, **opts), g["signal_" + tag])
    sm = run_ops(seq, **opts)
    close(sm.states, g["states_" + tag])


def test_g3_mrf_three_axis_grid(golden):
    g = golden("g3_mrf")
    seq = sq.mrf_ops(epg, g["T1"], g["T2"], g["B1"], g["alpha"], g["TR"], float(g["TE"]))
    assert epg.getshape(seq) == (2, 2, 4)
    sig = epg.simulate(seq, max_nstate=63)
    assert sig.shape == (1000, 2, 2, 4)
    close(sig, g["signal"])
    close(epg.simulate(seq, max_nstate=63, mode="stream"), g["signal"])
    close(run_ops(seq, max_nstate=63).states, g["states"])


def test_g5_spgr_phase_cycling_and_offresonance(golden):
    g = golden("g5_spgr")
    relax = epg.E(float(g["tau"]), float(g["T1"]), g["T2"], g["g"])
    shift = epg.S(1)
    spgr = [[epg.T(float(g["alpha"]), ph), relax, epg.Adc(phase=-ph), relax, shift] for ph in g["phases"]]
    close(epg.simulate(spgr, max_nstate=63), g["signal"])
    raw = [[epg.T(float(g["alpha"]), ph), relax, epg.ADC, relax, shift] for ph in g["phases"]]
    close(epg.simulate(raw, max_nstate=63), g["signal_raw"])
    close(run_ops(raw, max_nstate=63).states, g["states"])


def test_g6_negative_and_double_shifts(golden):
    g = golden("g6_ssfp")
    TR = float(g["TR"])
    rf = epg.T(float(g["FA"]), 0)
    s1, rx1 = epg.S(-1, duration=TR / 3), epg.E(TR / 3, 1e3, 1e2)
    s2, rx2 = epg.S(2, duration=TR * 2 / 3), epg.E(TR * 2 / 3, 1e3, 1e2)
    seq = [[rf, s1, rx1, epg.ADC, s2, rx2, epg.ADC]] * int(g["nrf"])
    times, sig = epg.simulate(seq, adc_time=True)
    close(sig, g["signal"])
    close(times, g["times"])
    close(run_ops(seq).states, g["states"])
    close(epg.simulate(seq, max_nstate=5), g["signal_cap5"])
    close(run_ops(seq, max_nstate=5).states, g["states_cap5"])


def test_g8_hyperecho_many_states(golden):
    """test/test_core.py:9-32: 2x201 pulses, 804 shifts -> K = 1024 (16 states per lane)"""
    g = golden("g8_hyperecho")
    n = int(g["npulse"])
    excit, grad, adc = epg.T(90, 90), epg.S(1), epg.ADC
    se1, se2 = [grad, epg.T(10, 0), grad, adc], [grad, epg.T(-10, 0), grad, adc]
    seq = [excit] + se1 * n + [grad, epg.T(180, 0), grad] + se2 * n
    F0, Z0 = epg.simulate(seq, probe=["F0", "Z0"])
    close(F0, g["F0"], 1e-11)
    close(Z0, g["Z0"], 1e-11)
    sim = epg.simulate(seq, probe="(F0, Z0)")  # eval'd probe -> stepwise path
    assert not np.allclose(sim, [[1], [0]])
    assert np.allclose(sim[-1], [[1], [0]])


def test_hyperecho_beyond_1024_orders():
    """the reference grows its state matrix without bound (shift.py:86,98): 2 x 401 pulses and 1606 shifts need 1607 orders --
    K = 2048: one wavefront per voxel while at most 512 orders hold anything, then up to four (run_split_kernel<4, ..>, state-resident
    from equilibrium), against the oracle and
    the known answer of test/test_core.py:9-32 (the hyper-echo refocuses completely: F0 = 1, Z0 = 0)"""
    n = 401
    T2 = np.array([40.0, 1e9])                       # with and without relaxation (the second refocuses exactly)
    excit, grad, rlx = epg.T(90, 90), epg.S(1), epg.E(0.05, 1e9, T2)
    se1, se2 = [grad, epg.T(10, 0), grad, rlx, epg.ADC], [grad, epg.T(-10, 0), grad, rlx, epg.ADC]
    seq = [excit] + se1 * n + [grad, epg.T(180, 0), grad] + se2 * n
    enc, _, _ = epg.compile_sequence(seq)
    assert enc.peak == 4 * n + 2 and enc.capacity(resident=True) == 2048
    with pytest.raises(NotImplementedError):
        enc.capacity()
    tuples = [("T", 90, 90)] + [("S", 1), ("T", 10, 0), ("S", 1), ("E", 0.05, 1e9, T2, 0), ("ADC",)] * n + \
             [("S", 1), ("T", 180, 0), ("S", 1)] + [("S", 1), ("T", -10, 0), ("S", 1), ("E", 0.05, 1e9, T2, 0), ("ADC",)] * n
    ref = epg_c.simulate(tuples)
    F0, Z0 = epg.simulate(seq, probe=["F0", "Z0"])
    assert F0.shape == (2 * n, 2)
    close(F0, ref, 1e-11)
    assert abs(F0[-1, 1] - 1) < 1e-6 and abs(Z0[-1, 1]) < 1e-6      # the hyper-echo (T2 = 1e9 ms: 1 - 4e-8)
    assert abs(F0[n, 1]) < 0.5                                         # (and not before)
    with pytest.raises((NotImplementedError, _lib.EpgxError)):
        epg.simulate(seq, mode="stream")                                # a 2048-order state matrix has no HBM form
    # truncation inside the four-wavefront layout (max_nstate around the seams) and S(-1)
    for cap in (1023, 1024, 1025, 1100, 1535, 1536, 1600):
        got = epg.simulate(seq, max_nstate=cap)
        close(got, epg_c.simulate(tuples, max_nstate=cap), 1e-11)


def test_g9_backend_parity_sequence(golden):
    """the reference's cupy<->numpy parity check (test/test_common.py:123-162)"""
    g = golden("g9_parity_mse")
    relax = epg.E(5, 1e3, g["T2"], g=g["g"])
    seq = [epg.T(90, 90)] + [epg.S(1), relax, epg.T(150, 0), epg.S(1), relax, epg.ADC] * 10
    sig = epg.simulate(seq)
    assert sig.shape == (10, 3, 2)
    close(sig, g["signal"])
    sm = epg.StateMatrix()
    for op in seq:
        sm = op(sm)
    close(sm.states, g["states"])


def test_g10_direct_operator_calls(golden):
    g = golden("g10_direct_ops")
    sm0 = epg.StateMatrix(g["init"])
    assert sm0.shape == (3, 2) and sm0.nstate == 4
    close(sm0.states, g["init"], 0)
    close(epg.T(g["T_alpha"], g["T_phi"])(sm0).states, g["T_states"])
    close(epg.E(7.5, 900.0, g["E_T2"], g["E_g"])(sm0).states, g["E_states"])
    close(sm0.states, g["init"], 0)  # op(sm) does not touch its argument (operator.py:87-88)
    for k in (1, -1, 2, -3):
        out = epg.S(k)(sm0)
        assert out.nstate == 4 + abs(k)
        close(out.states, g[f"S{k}_states"], 0)
        smc = epg.StateMatrix(g["init"], max_nstate=4)
        out = epg.S(k)(smc)
        assert out.nstate == 4
        close(out.states, g[f"S{k}_cap_states"], 0)
    sm = epg.StateMatrix(density=[1.0, 3.0])
    for op in [epg.T(60, 20), epg.S(1), epg.E(10, 200, 50)] * 2:
        sm = op(sm)
    close(sm.states, g["density_states"])
    close(sm.F0, g["F0"])
    close(sm.Z0, g["Z0"])
    close(sm.norm, g["norm"])
    sp = epg.SPOILER(sm)
    close(sp.states, g["spoiler_states"])
    close(epg.E(10, 200, 50)(sp).states, g["spoiler_E_states"])


# ------------------------------------------------------------------ reference known answers
def test_T_known_answers():
    """test/test_transition.py:8-56"""
    sm0 = epg.StateMatrix()
    assert np.allclose(epg.T(90, 90)(sm0).states, [[[1, 1, 0]]])
    assert np.allclose(epg.T(90, 0)(sm0).states, [[[-1j, 1j, 0]]])
    sm = epg.T(90, [90, 0])(sm0)
    assert sm.shape == (2,)
    assert np.allclose(sm.states, [[[1, 1, 0]], [[-1j, 1j, 0]]])
    sm = epg.T(90, [[90, 0]])(sm0)
    assert sm.shape == (1, 2)
    assert np.allclose(sm.states, [[[[1, 1, 0]], [[-1j, 1j, 0]]]])
    sm0 = epg.StateMatrix(shape=(1, 2))
    assert np.allclose(epg.T(90, 90)(sm0).states, [[[1, 1, 0]], [[1, 1, 0]]])
    op = epg.T([[90, 90]], [0, 90])
    sm = op(sm0)
    assert sm.shape == (2, 2)
    assert np.allclose(sm.states[0], [[[-1j, 1j, 0]], [[-1j, 1j, 0]]])
    assert np.allclose(sm.states[1], [[[1, 1, 0]], [[1, 1, 0]]])
    assert epg.T(90, [[0], [90]])(epg.StateMatrix(shape=(1, 2))).shape == (2, 2)
    with pytest.raises(ValueError):
        epg.T(90, [90] * 3)(epg.StateMatrix(shape=(4,)))
    with pytest.raises(TypeError):
        epg.T(90, 0)(np.zeros((1, 3)))
    sm0 = epg.StateMatrix([1, 1, 0])
    assert np.allclose(sm0.states, epg.T(0, 90)(sm0).states)


def test_E_known_answers():
    """test/test_evolution.py:8-58"""
    sm1 = epg.StateMatrix([1, 1, 0])
    assert np.allclose(epg.E(10, 1e10, 1e10)(sm1).states, [[[1, 1, 0]]])
    assert np.allclose(epg.E(10, 1e10, 1e-10)(sm1).states, [[[0, 0, 0]]])
    assert np.allclose(epg.E(10, 1e-10, 1e-10)(sm1).states, [[[0, 0, 1]]])
    assert np.allclose(epg.E(10, 1e10, 1e10, 0.025)(sm1).states, [[[1j, -1j, 0]]])
    assert np.allclose(epg.E(10, 1e10, [1e10, 1e-10])(sm1).states, [[[1, 1, 0]], [[0, 0, 0]]])
    assert np.allclose(epg.E(10, [1e10, 1e-10], 1e10)(sm1).states, [[[1, 1, 0]], [[1, 1, 1]]])
    assert np.allclose(epg.E(10, 1e10, 1e10, [0.025, 0.05])(sm1).states, [[[1j, -1j, 0]], [[-1, -1, 0]]])
    sm = epg.E(10, [[1e10, 1e-10]], [[1e10], [1e-10]])(sm1)
    assert sm.shape == (2, 2)
    assert np.allclose(sm.states[..., 0, :], [[[1, 1, 0], [1, 1, 1]], [[0, 0, 0], [0, 0, 1]]])
    assert np.allclose(epg.E([0, 10], 1e-10, 1e-10)(sm1).states, [[[1, 1, 0]], [[0, 0, 1]]])
    sm = epg.E([[0, 10]], 1e-10, [1e-10] * 3)(sm1)
    assert sm.shape == (3, 2)
    assert np.allclose(sm.states, [[[1, 1, 0]], [[0, 0, 1]]])
    # P: precession only (test/test_evolution.py test_P_class)
    assert np.allclose(epg.P(10, 0.025)(sm1).states, [[[1j, -1j, 0]]])


def test_S_known_answers():
    """test/test_shift.py:169-184, :249-270"""
    sm0 = epg.StateMatrix([1, 1, 0], max_nstate=1)
    sm1 = epg.S(1)(sm0)
    assert np.allclose(sm1.states, [[[0, 1, 0], [0, 0, 0], [1, 0, 0]]])
    sm2 = epg.S(-1)(sm1)
    assert np.allclose(sm2.states, [[[0, 0, 0], [1, 1, 0], [0, 0, 0]]])
    # hyper-echo through direct calls, growing state
    alphas = np.linspace(10, 80, 30)
    grad = epg.S(1)
    seq = [epg.T(90, 90)] + sum([[grad, epg.T(a, 0)] for a in alphas], start=[])
    seq += [grad, epg.T(180, 0)] + sum([[grad, epg.T(-a, 0)] for a in alphas[::-1]], start=[]) + [grad]
    sm = epg.StateMatrix()
    for op in seq:
        sm = op(sm)
    assert sm.nstate == 62
    assert np.allclose(sm.states[:, sm.nstate], [1, 1, 0])
    assert np.allclose(sm.states[:, : sm.nstate], 0)


def test_statematrix_contract():
    """test/test_statematrix.py:118-266 (the parts on the path)"""
    sm = epg.StateMatrix(init=[0, 0, 1])
    assert sm.ndim == 1 and sm.shape == (1,) and sm.nstate == 0 and sm.coords is None
    assert np.allclose(sm.states, [[[0, 0, 1]]]) and np.allclose(sm.density, [1])
    sm = epg.StateMatrix(init=[[0, 0, 0], [0, 0, 1], [0, 0, 0]])
    assert sm.shape == (1,) and sm.nstate == 1
    sm = epg.StateMatrix(init=[[[0, 0, 1]]] * 5)
    assert sm.shape == (5,) and sm.nstate == 0
    sm = epg.StateMatrix(init=[[[[0, 0, 1]]], [[[0, 0, 2]]]])
    assert sm.shape == (2, 1)
    sm = epg.StateMatrix(density=[1, 3])
    assert sm.shape == (2,) and np.allclose(sm.density, [1, 3])
    assert np.allclose(sm.states, sm.equilibrium)
    for bad in ([0, 1], [[0, 0, 0, 1]], [[[0, 1]]], [[0, 0, 1], [0, 0, 1]], [[[0, 0, 1], [0, 0, 1]]]):
        with pytest.raises(ValueError):
            epg.StateMatrix(init=bad)
    sm.expand(3)
    assert sm.shape == (2, 1, 1) and sm.states.shape == (2, 1, 1, 1, 3)
    sm.reduce(1)
    assert sm.shape == (2,)
    sm2 = sm.copy()
    assert np.allclose(sm2.density, [1, 3])
    # copy(equilibrium=...) (statematrix.py:282-283): the copy relaxes towards the new equilibrium, the original keeps its own
    sm3 = sm.copy(equilibrium=[[[0, 0, 5]], [[0, 0, 7]]])
    assert np.allclose(sm3.density, [5, 7]) and np.allclose(sm.density, [1, 3]) and np.allclose(sm3.states, sm.states)
    assert np.allclose(epg.E(1e9, 1.0, 1.0)(sm3).Z0, [5, 7]) and np.allclose(epg.E(1e9, 1.0, 1.0)(sm).Z0, [1, 3])
    assert np.allclose(sm.copy(equilibrium=[0, 0, 4]).density, [4, 4])
    sm4 = sm.copy(equilibrium=[1, 1, 0.5])          # a general equilibrium: a second device-resident matrix (golden G15)
    assert sm4._eq is not None and np.allclose(sm4.density, [0.5, 0.5]) and np.allclose(sm4.equilibrium[:, 0], [1, 1, 0.5])
    with pytest.raises(ValueError):
        sm.copy(equilibrium=np.zeros((3, 1, 3)))
    sm = epg.StateMatrix(nstate=3)
    assert sm.nstate == 3
    sm.resize(5)
    assert sm.nstate == 5 and sm.equilibrium.shape[-2:] == sm.states.shape[-2:]
    sm.resize(0)
    assert sm.nstate == 0 and np.allclose(sm, [[[0, 0, 1]]])
    assert epg.StateMatrix(shape=(3,)).shape == (3,)
    sm = epg.StateMatrix(init=[[1 / 2, 1 / 2, 0], [1, 1, 0.5], [1 / 2, 1 / 2, 0]])
    assert np.allclose(sm.F, [1 / 2, 1, 1 / 2]) and np.allclose(sm.F0, [1])
    assert np.allclose(sm.Z, [0, 0.5, 0]) and np.allclose(sm.Z0, [0.5])
    sm1, sm2 = epg.StateMatrix([0, 0, 1]), epg.StateMatrix([1, 1, 0])
    assert np.allclose((sm1 + sm2), [1, 1, 1])
    sm1 += sm2
    assert np.allclose(sm1, [1, 1, 1])
    assert epg.StateMatrix(max_nstate=3, kgrid=2).options == {"max_nstate": 3, "kgrid": 2}
    # norm conservation (test/test_statematrix.py:251-266)
    sm = epg.StateMatrix()
    for op in [epg.T(30, 30), epg.S(1)] * 10:
        sm = op(sm)
    assert np.isclose(sm.norm, 1) and sm.check()
    sm = epg.StateMatrix(equilibrium=[0, 0, 10])
    for op in [epg.T(30, 30), epg.S(1)] * 10:
        sm = op(sm)
    assert np.isclose(sm.norm, 10)


def test_simulate_contract():
    """test/test_functions.py:6-107"""
    excit, refoc = epg.T(90, 90), epg.T(180, 0)
    grad, relax = epg.S(1, duration=10), epg.E(10, 1000, 30)
    seq1 = [excit, grad, relax, refoc, grad, relax, epg.ADC]
    seq2 = excit * grad * relax * refoc * grad * relax * epg.ADC
    signal_1 = epg.simulate(seq1)
    assert np.allclose(signal_1, epg.simulate(seq2))
    assert np.allclose(signal_1, onp.simulate([("T", 90, 90), ("S", 1), ("E", 10, 1000, 30, 0), ("T", 180, 0),
                                               ("S", 1), ("E", 10, 1000, 30, 0), ("ADC",)]))
    seq3 = list(seq1)
    seq3[-1] = epg.Probe("(real(F0), imag(F0))")
    res = epg.simulate(seq3)
    assert np.allclose(res[0], [np.real(signal_1[0]), np.imag(signal_1[0])])
    assert np.allclose(epg.simulate(seq3, probe="abs(F0)"), np.abs(signal_1))
    f0, z0 = epg.simulate(seq3, probe="F0"), epg.simulate(seq3, probe="Z0")
    res = epg.simulate(seq3, probe=["F0", "Z0"])
    assert np.allclose(res[0], f0) and np.allclose(res[1], z0)
    seq4 = [excit, grad, relax, refoc, grad, relax, epg.Adc(phase=15)]
    assert np.isclose(epg.simulate(seq4), f0 * np.exp(1j * 15 / 180 * np.pi))
    assert np.isclose(epg.simulate(seq4, probe="Z0"), z0 * np.exp(1j * 15 / 180 * np.pi))
    adcn = epg.Adc(reduce=1, weights=[[1, 2, 3, 4, 5]])
    relaxn = epg.E(10, 1000, [[30], [40], [50]], g=[[-0.1, -0.05, 0, 0.05, 1]])
    res_ = epg.simulate([excit, grad, relaxn, refoc, grad, relaxn, epg.ADC])
    resn = epg.simulate([excit, grad, relaxn, refoc, grad, relaxn, adcn])
    assert np.allclose(np.dot(res_, [1, 2, 3, 4, 5]), resn)
    # callback sees the state after each non-probe operator (functions.py:190-191)
    seen = []
    epg.simulate(seq1, callback=lambda sm: seen.append(sm.nstate))
    assert seen == [0, 1, 1, 1, 2, 2]


def test_simulate_ndim_and_init():
    """test/test_functions.py:79-107"""
    ax = epg.Axes("FA", "T2")
    refoc = epg.T([180, 150], 0, axes=ax.FA)
    relax = epg.E(10, 1e3, [30, 40, 50], axes=ax.T2)
    seq = [epg.T(90, 90)] + [epg.S(1), relax, refoc, epg.S(1), relax, epg.ADC] * 2
    signal = epg.simulate(seq)
    assert all(sig.shape == (2, 3) for sig in signal)
    ref = onp.simulate([("T", 90, 90)] + [("S", 1), ("E", 10, 1e3, [[30, 40, 50]], 0), ("T", [[180], [150]], 0),
                                          ("S", 1), ("E", 10, 1e3, [[30, 40, 50]], 0), ("ADC",)] * 2)
    close(signal, ref)
    signal = epg.simulate(seq, init=epg.StateMatrix(shape=(1, 1, 4)))
    assert all(sig.shape == (2, 3, 4) for sig in signal)
    with pytest.raises(ValueError):
        epg.simulate(seq + [epg.T([90] * 3, 180)])
    with pytest.raises(ValueError):
        epg.simulate(seq, init=epg.StateMatrix(shape=(3, 3)))
    # pre-sized init with max_nstate (examples/basics/hyperecho.py:29-31) is not modified
    init = epg.StateMatrix(nstate=8, max_nstate=8)
    s1 = epg.simulate(seq, init=init)
    assert init.nstate == 8 and np.allclose(init.states[:, 8], [0, 0, 1])
    close(s1, ref)
    # init from a non-equilibrium state
    close(epg.simulate([epg.S(1), epg.T(45, 10), epg.S(-1), epg.ADC], init=[1, 1, 0]),
          onp.simulate([("S", 1), ("T", 45, 10), ("S", -1), ("ADC",)], init=np.array([[1, 1, 0]], complex)))


# ------------------------------------------------------------------ oracle parity on seeded inputs
@pytest.mark.parametrize("nshift,K", [(70, 128), (150, 256), (300, 512), (600, 1024)])
def test_large_state_capacities(nshift, K):
    """states spread over several registers per lane (M = K/64 > 1), unbounded growth"""
    rng = np.random.default_rng(nshift)
    T2 = rng.uniform(30, 200, 5)
    tuples = [("T", 70, 20)]
    for i in range(nshift):
        tuples += [("S", 1), ("E", 2.0, 800.0, T2, 0.003), ("T", float(rng.uniform(5, 60)), float(rng.uniform(0, 360)))]
        if i % 10 == 9:
            tuples += [("ADC",)]
    ops = sq.to_ops(epg, tuples)
    enc, _, _ = epg.compile_sequence(ops)
    assert enc.capacity() == K
    ref, ref_states = epg_c.simulate(tuples, return_states=True)
    close(epg.simulate(ops), ref)
    close(epg.simulate(ops, mode="stream"), ref)
    sm = run_ops(ops)
    assert sm.nstate == nshift
    close(sm.states, ref_states)


@pytest.mark.parametrize("cap", [None, 1000, 650, 520, 511])
def test_two_wavefronts_per_voxel_at_1024_orders(cap):
    """K = 1024 state-resident launches (rounds 1-3: a voxel on TWO wavefronts, orders 0..511 / 512..1023; now one wavefront with
    16 consecutive orders per lane, run_contig_kernel<16, ..> / run_contig_grow_kernel<16, ..>): growth across order 512,
    truncation on either side of it and exactly at it, S(-1), Z0 probes, SPOILER / RESET / PD in the middle, an odd voxel
    count, a given initial state -- against the oracle and the per-timestep mode"""
    rng = np.random.default_rng(1024)
    T2 = rng.uniform(30, 200, 3)
    tuples = [("T", 70, 20)]
    for i in range(560):
        tuples += [("S", 1), ("E", 2.0, 800.0, T2, 0.003), ("T", float(rng.uniform(5, 60)), float(rng.uniform(0, 360)))]
        if i % 40 == 39:
            tuples += [("ADC",), ("ADC", "Z0")]
        if i == 300:
            tuples += [("S", -1), ("S", -1), ("ADC",)]
    tuples += [("SPOILER",), ("E", 5.0, 800.0, T2, 0), ("ADC", "Z0"), ("PD", 0.7, False), ("T", 30, 0), ("S", 1), ("ADC",),
               ("RESET",), ("T", 50, 10), ("S", 1), ("E", 3.0, 800.0, T2, 0), ("T", 20, 60), ("S", -1), ("ADC",)]
    ops = sq.to_ops(epg, tuples)
    kw = {"max_nstate": cap} if cap else {}
    enc, _, _ = epg.compile_sequence(ops, options=kw)
    assert enc.capacity() == (1024 if cap != 511 else 512)
    ref = epg_c.simulate(tuples, **kw)
    res = epg.simulate(ops, **kw)
    close(res, ref)
    close(epg.simulate(ops, mode="stream", **kw), ref)
    # from a prepared state (HAS_IN variant): 530 orders populated, then more shifts across the seam
    sm = epg.StateMatrix(shape=(3,), **kw)
    head = 3 * 530 + 1
    for op in ops[:head]:
        sm = op(sm, inplace=True)
    rest = [op for op in ops[head:]]
    got = epg.simulate(rest, init=sm, **kw)
    n_head = sum(1 for t in tuples[:head] if t[0] == "ADC")
    close(got, ref[n_head:])


def test_big_shifts_at_1024_orders():
    """S(+-n) with |n| > 1 at K = 1024: the LDS staging of a general shift is two arrays of K complex per wavefront (128 KiB for
    the four wavefronts of a block; three arrays -- what a gather shift stages -- would not fit the 160 KiB of a CU), in every
    mode and with a derivative state"""
    rng = np.random.default_rng(1024)
    T2 = rng.uniform(30, 200, 3)
    tuples = [("T", 80, 45)]
    for i in range(330):
        k = int(rng.choice([1, 1, 2, 3, -1, -2, 5]))
        tuples += [("S", k), ("E", 3.0, 700.0, T2, 0.01), ("T", float(rng.uniform(10, 90)), float(rng.uniform(0, 360)))]
        if i % 10 == 9:
            tuples.append(("ADC",))
    ops = sq.to_ops(epg, tuples)
    enc, _, _ = epg.compile_sequence(ops)
    assert enc.capacity() == 1024
    ref = epg_c.simulate(tuples)
    for mode in ("resident", "stream"):
        close(epg.simulate(ops, mode=mode), ref)
    tuples_j = [(t + ({"order1": {"T2": {"T2": 1}}},)) if t[0] == "E" else t for t in tuples]
    ops_j = [epg.E(*t[1:5], order1=["T2"]) if t[0] == "E" else op for t, op in zip(tuples, ops)]
    got = epg.simulate(ops_j, probe=epg.Jacobian(["magnitude", "T2"]))
    close(got, onp.simulate_jacobian(tuples_j, ["magnitude", "T2"]), 1e-10)


@pytest.mark.parametrize("cap", [None, 40, 100])
def test_mixed_shifts_multi_register(cap):
    """S(+-n) with |n| > 1 (LDS path) and +-1 (DPP path) on K = 128/256, with truncation"""
    rng = np.random.default_rng(7)
    T2 = rng.uniform(30, 200, 3)
    tuples = [("T", 80, 45)]
    for i in range(60):
        k = int(rng.choice([1, 1, 2, 3, -1, -2, 5]))
        tuples += [("S", k), ("E", 3.0, 700.0, T2, 0.01), ("T", float(rng.uniform(10, 90)), float(rng.uniform(0, 360))), ("ADC",)]
    ops = sq.to_ops(epg, tuples)
    opts = {} if cap is None else {"max_nstate": cap}
    ref, ref_states = onp.simulate(tuples, max_nstate=cap, return_states=True)
    close(epg.simulate(ops, **opts), ref)
    close(epg.simulate(ops, mode="stream", **opts), ref)
    close(run_ops(ops, **opts).states, ref_states)


def test_resident_and_stream_are_bit_identical():
    rng = np.random.default_rng(3)
    T1, T2, B1 = rng.uniform(200, 3000, 1000), rng.uniform(20, 300, 1000), rng.uniform(0.7, 1.2, 1000)
    seq = sq.mse_ops(epg, T1, T2, B1)
    a = epg.simulate(seq, max_nstate=63, mode="resident")
    b = epg.simulate(seq, max_nstate=63, mode="stream")
    assert sq.same_bits(a, b, x64=True)       # (64 orders, refocusing about x: see same_bits)
    close(a, epg_c.simulate(sq.mse_tuples(T1, T2, B1), max_nstate=63))


def test_spoiler_reset_pd_in_sequence():
    pd = np.array([0.5, 1.0, 2.0])
    tuples = [("PD", pd), ("T", 40, 0), ("S", 1), ("E", 5, 300, 40, 0), ("ADC",), ("SPOILER",), ("T", 40, 0), ("S", 1),
              ("E", 5, 300, 40, 0), ("ADC",), ("ADC", "Z0"), ("RESET",), ("T", 90, 90), ("ADC",)]
    ref = onp.simulate(tuples)
    sig = epg.simulate(sq.to_ops(epg, tuples))
    close(sig, ref)
    assert np.allclose(sig[-1], pd)


# ------------------------------------------------------------------ C ABI, direct
def test_c_abi_host_entry_point(golden):
    """epgx_simulate_f64 with plain host buffers, as a ctypes binding inside the reference
    would call it"""
    import ctypes
    g = golden("g2_random_mse")
    seq = sq.mse_ops(epg, g["T1"], g["T2"], g["B1"])
    ctx = _lib.get_context()
    for fuse in (True, False):     # with device-generated E.T.E tables (epgx_fuse) / primitive operators only
        enc, _, _ = epg.compile_sequence(seq, options={"max_nstate": 63}, fuse=fuse)
        ops, grid, spaces, coef, _ = enc.arrays()
        fuses = enc.fuse_array()
        assert bool(len(fuses)) == fuse
        strides = np.zeros((max(len(spaces), 1), _lib.MAX_DIMS), dtype=np.int64)
        for s, st in enumerate(spaces):
            strides[s, : len(st)] = st
        desc = _lib.PlanDesc(ctypes.sizeof(_lib.PlanDesc), len(ops), ops.ctypes.data, len(grid), grid.ctypes.data, len(spaces), strides.ctypes.data,
                             coef.size, coef.ctypes.data, enc.n_adc, 0, None, 0, len(fuses),
                             fuses.ctypes.data if len(fuses) else None, enc.generated_size)
        signal = np.zeros((20, 64), dtype=np.complex128)
        half = np.zeros((64, 3, 64), dtype=np.complex128)
        rc = ctx.lib.epgx_simulate_f64(ctx.handle, ctypes.byref(desc), 64, None, None, signal.ctypes.data, half.ctypes.data, 0)
        assert rc == 0, ctx.lib.epgx_last_error()
        close(signal, g["signal_cap63"])
        if fuse:   # a T0 operator that points into the generated part without a recipe is rejected
            norecipe = _lib.PlanDesc(ctypes.sizeof(_lib.PlanDesc), len(ops), ops.ctypes.data, len(grid), grid.ctypes.data, len(spaces),
                                     strides.ctypes.data, coef.size, coef.ctypes.data, enc.n_adc)
            assert ctx.lib.epgx_simulate_f64(ctx.handle, ctypes.byref(norecipe), 64, None, None, signal.ctypes.data, None, 0) == -1
    close(onp.expand_half(half[:, :, :41]), g["states_cap63"])
    # sharded entry point over 1 GPU gives the same bits
    signal2 = np.zeros_like(signal)
    rc = ctx.lib.epgx_simulate_sharded_f64(ctypes.byref(desc), 64, 1, None, signal2.ctypes.data, 0)
    assert rc == 0, ctx.lib.epgx_last_error()
    assert sq.same_bits(signal, signal2, x64=True)      # (state output: per-timestep kernel; none: the 64-order rows kernel)
    # errors are reported, not thrown
    bad = ops.copy()
    bad["opcode"][0] = 99
    desc_bad = _lib.PlanDesc(ctypes.sizeof(_lib.PlanDesc), len(bad), bad.ctypes.data, len(grid), grid.ctypes.data, len(spaces), strides.ctypes.data,
                             coef.size, coef.ctypes.data, enc.n_adc)
    rc = ctx.lib.epgx_simulate_f64(ctx.handle, ctypes.byref(desc_bad), 64, None, None, signal.ctypes.data, None, 0)
    assert rc == -1 and b"unknown opcode" in ctx.lib.epgx_last_error()
    bad = ops.copy()
    bad["coef_off"][2] = coef.size
    desc_bad.ops = bad.ctypes.data
    rc = ctx.lib.epgx_simulate_f64(ctx.handle, ctypes.byref(desc_bad), 64, None, None, signal.ctypes.data, None, 0)
    assert rc == -1 and b"exceeds the pool" in ctx.lib.epgx_last_error()
    rc = ctx.lib.epgx_simulate_f64(ctx.handle, ctypes.byref(desc), 96, None, None, signal.ctypes.data, None, 0)
    assert rc == -4


def test_voxel_ranges_are_bit_identical_to_full_run():
    """sharding must not change arithmetic"""
    T1 = np.linspace(200, 3000, 37)[:, None]
    T2 = np.linspace(20, 300, 11)[None, :]
    seq = sq.mse_ops(epg, T1, T2, necho=6)
    full = epg.simulate(seq, max_nstate=63)
    from epgpy_amd.distributed import ShardedPlan
    parts = []
    for r in range(3):
        sp = ShardedPlan(seq, rank=r, world_size=3, max_nstate=63).bind()
        buf = _lib.DeviceBuffer(sp._ctx, 16 * sp.n_adc * sp.slab)
        sp._ctx.lib.epgx_memset(sp._ctx.handle, buf.ptr, 0, buf.nbytes)
        sp.run(buf.ptr.value)
        parts.append(buf.download(np.complex128, (sp.n_adc, sp.slab)))
    assert np.array_equal(sp.assemble(np.stack(parts)), full)
    # stream mode on a slab
    sp = ShardedPlan(seq, rank=1, world_size=3, max_nstate=63).bind()
    buf = _lib.DeviceBuffer(sp._ctx, 16 * sp.n_adc * sp.slab)
    sp.run(buf.ptr.value, mode="stream", state=sp.new_state())
    assert np.array_equal(buf.download(np.complex128, (sp.n_adc, sp.slab)), parts[1])
    # short state matrices (four / two voxels per wavefront): slabs that are no multiple of the packing
    for cap in (10, 25):
        alpha, TR = sq.mrf_trains(25)
        B1 = np.linspace(0.8, 1.2, 5)[None, None, :]
        seq = sq.mrf_ops(epg, T1[:, :, None], T2[:, :, None], B1, alpha, TR)
        full = epg.simulate(seq, max_nstate=cap)
        parts = []
        for r in range(7):
            sp = ShardedPlan(seq, rank=r, world_size=7, max_nstate=cap).bind()
            assert sp.K_resident == (16 if cap == 10 else 32)
            buf = _lib.DeviceBuffer(sp._ctx, 16 * sp.n_adc * sp.slab)
            sp._ctx.lib.epgx_memset(sp._ctx.handle, buf.ptr, 0, buf.nbytes)
            sp.run(buf.ptr.value)
            parts.append(buf.download(np.complex128, (sp.n_adc, sp.slab)))
        assert np.array_equal(sp.assemble(np.stack(parts)), full)


# ------------------------------------------------------------------ full size
@pytest.mark.parametrize("side", [256, 1024])
def test_full_size_mse(side):
    """BASELINE configs[1] (256 x 256) and the >= 10^6-voxel target C2-L (1024 x 1024): 20-echo MSE over a (T1, T2) grid,
    64 k-states.  Checked (a) on EVERY voxel and echo against the C oracle run over the whole grid on the host cores
    (1 048 576 voxels x 20 echoes: a mis-indexed table row in one wave group cannot hide between samples), (b) through
    size-independent properties: stream == resident, signal scales linearly with density, and the first echo of every voxel
    equals the closed form sin^2(FA/2) * exp(-ESP/T2).  (The cupy <-> numpy parity pattern of the reference's
    test/test_common.py:123-162: the same sequence on both back-ends, whole arrays compared.)"""
    T1 = np.linspace(200, 3000, side)[:, None]
    T2 = np.linspace(20, 300, side)[None, :]
    seq = sq.mse_ops(epg, T1, T2)
    sig = epg.simulate(seq, max_nstate=63)
    assert sig.shape == (20, side, side)
    ref = epg_c.simulate(sq.mse_tuples(T1, T2), max_nstate=63, nthreads=max(1, len(os.sched_getaffinity(0))))
    assert ref.shape == sig.shape
    close(sig, ref)
    del ref
    assert np.allclose(sig[0].real, np.sin(np.pi / 3) ** 2 * np.exp(-10.0 / T2) * np.ones_like(T1), rtol=0, atol=1e-13)
    assert sq.same_bits(sig, epg.simulate(seq, max_nstate=63, mode="stream"), x64=True)
    scaled = epg.simulate([epg.PD(2.5)] + seq, max_nstate=63)
    assert np.allclose(scaled, 2.5 * sig, rtol=1e-14, atol=1e-15)


# ------------------------------------------------------------------ config 5: n-D shifts + diffusion
def test_g7_pgse_diffusion_grid(golden):
    """BASELINE config 5 (scaled down): PGSE with a 3-D gradient, (T2, ADC) grid.  The golden
    signal was produced by looping the reference over ADC (its D rejects arrays,
    diffusion.py:166-169); here ADC is a grid axis (`field=True`)."""
    g = golden("g7_pgse")
    T2, adc, T1 = g["T2"], g["ADC"], float(g["T1"])
    kvalue, k1 = list(g["kvalue"]), [int(v) for v in g["k"]]
    T2g, ADCg = T2[:, None], adc[None, :]
    seq = [epg.T(90, 90), epg.S(k1), epg.D(10, ADCg, k=k1, field=True), epg.E(10, T1, T2g),
           epg.D(20, ADCg, field=True), epg.E(20, T1, T2g), epg.T(180, 0),
           epg.D(20, ADCg, field=True), epg.E(20, T1, T2g), epg.S(k1), epg.D(10, ADCg, k=k1, field=True),
           epg.E(10, T1, T2g), epg.ADC]
    assert epg.getshape(seq) == (8, 8)
    for mode in ("resident", "stream"):
        sig = epg.simulate(seq, kvalue=kvalue, mode=mode)
        assert sig.shape == (1, 8, 8)
        close(sig[0], g["signal"])
    # closed form exp(-b D) * exp(-TE/T2): b = (2/3 delta^3 ... ) checked through the reference value
    # quoted in SURVEY.md section 8d: attenuation 0.97579769 at D = 1e-3 (T2 -> infinity)
    seq_inf = [epg.T(90, 90), epg.S(k1), epg.D(10, 1e-3, k=k1), epg.D(20, 1e-3), epg.T(180, 0), epg.D(20, 1e-3),
               epg.S(k1), epg.D(10, 1e-3, k=k1), epg.ADC]
    att = epg.simulate(seq_inf, kvalue=kvalue)
    assert np.isclose(abs(att[0, 0]), 0.97579769, atol=1e-8)


def test_D_known_answers():
    """test/test_diffusion.py:143-199 (test_D_class), values and closed forms from there"""
    sm0 = epg.StateMatrix([1, 1, 0], kvalue=1e5)
    d1 = epg.D(1, 1e-3)
    assert np.allclose(d1(sm0).states, sm0.states)
    shift1, shift2 = epg.S(1), epg.S(-1)
    sm1 = shift2(d1(shift1(sm0)))
    assert np.isclose(sm1.F0, np.exp(-sm1.kvalue ** 2 * d1.tau * d1.D * 1e-9))
    d2 = epg.D(1, 1e-3, k=1)
    sm1 = shift2(d2(shift1(sm0)))
    assert np.isclose(sm1.F0, np.exp(-sm1.kvalue ** 2 * (1 / 4 + 1 / 12) * d1.tau * d1.D * 1e-9))
    # spin echo
    exc, ref, shift = epg.T(90, 90), epg.T(180, 0), epg.S(1)
    d1, d2 = epg.D(1, 1e-3, k=1), epg.D(2e-1, 1e-3)
    sm = epg.StateMatrix(kvalue=1e4)
    for op in [exc, shift, d1, d2, ref, d2, shift, d1]:
        sm = op(sm)
    Dm, k = d1.D * 1e-9, sm.kvalue
    assert np.isclose(sm.F0, np.exp(-2 / 3 * k ** 2 * d1.tau * Dm) * np.exp(-2 * k ** 2 * d2.tau * Dm))
    # 2-D isotropic tensor, 2-D discrete gradient
    Dt = np.diag([1, 1])
    shift = epg.S([1, 0])
    d1, d2 = epg.D(1, Dt, k=[1, 0]), epg.D(2e-1, Dt)
    sm = epg.StateMatrix(kvalue=1e4)
    for op in [exc, shift, d1, d2, ref, d2, shift, d1]:
        sm = op(sm)
    assert sm.kdim == 2 and sm.coords.shape[-1] == 2
    assert np.isclose(sm.F0, np.exp(-2 / 3 * k ** 2 * d1.tau * 1e-9) * np.exp(-2 * k ** 2 * d2.tau * 1e-9))
    # anisotropic
    Dt = np.diag([1, 2])
    shift = epg.S([1, 1])
    d1, d2 = epg.D(1, Dt, k=[1, 1]), epg.D(2e-1, Dt)
    sm = epg.StateMatrix(kvalue=1e4)
    for op in [exc, shift, d1, d2, ref, d2, shift, d1]:
        sm = op(sm)

    def bmat(tau, k1, k2=None):  # diffusion.py:86-123 in rad/mm, s
        k1 = np.asarray(k1, float) * 1e-3
        b = np.outer(k1, k1) * tau * 1e-3
        if k2 is not None:
            kd = np.asarray(k2, float) * 1e-3 - k1
            b = b + tau * 1e-3 * (np.outer(k1, kd) / 2 + np.outer(kd, k1) / 2 + np.outer(kd, kd) / 3)
        return b
    bT = bmat(1, [0, 0], [k, k]) + bmat(2e-1, [k, k]) + bmat(2e-1, [-k, -k]) + bmat(1, [-k, -k], [0, 0])
    assert np.isclose(sm.F0, np.exp(-np.trace(bT @ Dt)))
    with pytest.raises(ValueError):
        epg.D(1, [1.0, 2.0])            # 1-D D without field=True (diffusion.py:166-167)
    with pytest.raises(ValueError):
        epg.D(1, np.ones((2, 3)))


def test_nd_shift_known_answers():
    """test/test_shift.py:34-72 (coordinates), :270-284 (hyper-echo with a 3-D gradient)"""
    sm0 = epg.StateMatrix([1, 1, 0])
    sm1 = epg.S([1, 0, 0])(sm0)
    assert np.allclose(sm1.states, [[[0, 1, 0], [0, 0, 0], [1, 0, 0]]])
    assert np.array_equal(sm1.coords[0], [[-1, 0, 0], [0, 0, 0], [1, 0, 0]])
    sm2 = epg.S([-1, 0, 0])(sm1)
    assert np.allclose(sm2.F0, 1) and np.allclose(sm2.states[:, sm2.nstate], [1, 1, 0])
    assert np.allclose(np.delete(sm2.states, sm2.nstate, axis=-2), 0)
    # 3-D round trip
    shifts = [[i, j, k] for i in [-1, 2] for j in [-2, 1] for k in [2, -2]]
    shifts += [[-i, -j, -k] for i, j, k in shifts]
    sm = sm0
    for dk in shifts:
        sm = epg.S(dk)(sm)
    assert np.allclose(sm.states[:, sm.nstate], [1, 1, 0])
    assert np.allclose(np.delete(sm.states, sm.nstate, axis=-2), 0)
    # hyper-echo
    alphas = np.linspace(10, 80, 12)
    grad = epg.S([1, -2, 0])
    seq = [epg.T(90, 90)] + sum([[grad, epg.T(a, 0)] for a in alphas], start=[])
    seq += [grad, epg.T(180, 0)] + sum([[grad, epg.T(-a, 0)] for a in alphas[::-1]], start=[]) + [grad]
    sm = epg.StateMatrix()
    for op in seq:
        sm = op(sm)
    assert np.allclose(sm.states[:, sm.nstate], [1, 1, 0])
    assert np.allclose(sm.states[:, : sm.nstate], 0)
    f0 = epg.simulate(seq + [epg.ADC])
    assert np.allclose(f0, 1)
    # a 1-D int shift after coordinates exist means [k, 0, 0] (shift.py:224-229)
    a = epg.simulate([epg.T(60, 10), epg.S([1, 0]), epg.T(40, 0), epg.S(1), epg.T(30, 0), epg.S(-2), epg.ADC])
    b = onp.simulate([("T", 60, 10), ("S", 1), ("T", 40, 0), ("S", 1), ("T", 30, 0), ("S", -2), ("ADC",)])
    close(a, b)


def test_full_size_pgse_512x512(golden):
    """BASELINE config 5 at full size: PGSE over a 512 x 512 (T2, ADC) grid, 3-D k-space shift.
    Checked against the golden corners (same end points as the 8 x 8 fixture) and through the
    size-independent property that relaxation and diffusion attenuations are separable:
    S(T2, ADC) * S(T2_0, ADC_0) == S(T2, ADC_0) * S(T2_0, ADC)."""
    g = golden("g7_pgse")
    T1, kvalue, k1 = float(g["T1"]), list(g["kvalue"]), [int(v) for v in g["k"]]
    T2g = np.linspace(20, 300, 512)[:, None]
    ADCg = np.linspace(1e-4, 3e-3, 512)[None, :]
    seq = [epg.T(90, 90), epg.S(k1), epg.D(10, ADCg, k=k1, field=True), epg.E(10, T1, T2g),
           epg.D(20, ADCg, field=True), epg.E(20, T1, T2g), epg.T(180, 0),
           epg.D(20, ADCg, field=True), epg.E(20, T1, T2g), epg.S(k1), epg.D(10, ADCg, k=k1, field=True),
           epg.E(10, T1, T2g), epg.ADC]
    sig = epg.simulate(seq, kvalue=kvalue)[0]
    assert sig.shape == (512, 512)
    for (i, gi) in ((0, 0), (511, 7)):
        for (j, gj) in ((0, 0), (511, 7)):
            assert abs(sig[i, j] - g["signal"][gi, gj]) < 1e-13
    lhs = sig * sig[0, 0]
    rhs = sig[:, :1] * sig[:1, :]
    assert np.max(np.abs(lhs - rhs)) < 1e-14
    assert np.all(np.diff(np.abs(sig), axis=0) > 0) and np.all(np.diff(np.abs(sig), axis=1) < 0)


def test_modify_on_device():
    """test/test_functions.py:146-166: modify() attaches precession / B1 attenuation"""
    seq = [epg.T(90, 90), epg.Wait(1), epg.T(90, 90), epg.ADC]
    newseq = epg.modify(seq, g=[[0, 0.25, 0.5]], att=[1, 0.5])
    signal = epg.simulate(newseq)[0]
    assert signal.shape == (2, 3)
    assert np.isclose(signal[0, 0], 0) and np.isclose(signal[0, 1], 1j) and np.isclose(signal[0, 2], 0)
    assert np.isclose(signal[1, 0], 1)
    assert np.isclose(signal[1, 1], epg.simulate([epg.T(45, 180), epg.T(45, 90), epg.ADC]))
    # relaxation attached to the durations of a timing-only spin echo
    se = [epg.T(90, 90), epg.S(1, duration=7), epg.T(180, 0), epg.S(1, duration=7), epg.ADC]
    T2 = np.array([40.0, 80.0])
    got = epg.simulate(epg.modify(se, T1=900.0, T2=T2))
    ref = onp.simulate([("T", 90, 90), ("S", 1), ("E", 7, 900.0, T2, 0), ("T", 180, 0), ("S", 1),
                        ("E", 7, 900.0, T2, 0), ("ADC",)])
    close(got, ref)


# ------------------------------------------------------------------ edge cases
@pytest.mark.parametrize("nvox", [1, 2, 3, 5, 63, 64, 65, 257])
def test_ragged_voxel_counts(nvox):
    """grids that do not fill a workgroup (4 voxels) or a 16-block placement group"""
    rng = np.random.default_rng(nvox)
    T1, T2 = rng.uniform(300, 2000, nvox), rng.uniform(30, 200, nvox)
    seq = sq.mse_ops(epg, T1, T2, necho=4)
    ref = onp.simulate(sq.mse_tuples(T1, T2, necho=4), max_nstate=63)
    for mode in ("resident", "stream"):
        got = epg.simulate(seq, max_nstate=63, mode=mode)
        assert got.shape == (4, nvox)
        close(got, ref)


def test_sequence_edge_cases():
    # ADC before anything else, several ADCs in a row, no state-changing operator at all
    assert np.allclose(epg.simulate([epg.ADC]), 0)
    assert np.allclose(epg.simulate([epg.ADC, epg.ADC], probe="Z0"), 1)
    out = epg.simulate([epg.T(90, 90), epg.ADC, epg.Adc("Z0"), epg.ADC])
    assert np.allclose(out[0], 1) and np.allclose(out[1], 0) and np.allclose(out[2], 1)
    # Wait / NULL do nothing but advance time
    t, s = epg.simulate([epg.T(90, 90), epg.Wait(3), epg.NULL, epg.ADC], adc_time=True)
    assert np.allclose(t, [3]) and np.allclose(s, 1)
    # truncation to a single order: every shift empties the transverse states
    s = epg.simulate([epg.T(90, 90), epg.S(1), epg.ADC, epg.T(90, 0), epg.S(-1), epg.ADC], max_nstate=0)
    r = onp.simulate([("T", 90, 90), ("S", 1), ("ADC",), ("T", 90, 0), ("S", -1), ("ADC",)], max_nstate=None)
    assert s.shape == (2, 1)
    # shift larger than the number of populated orders, in both directions
    tup = [("T", 70, 30), ("S", 5), ("T", 50, 0), ("S", -7), ("ADC",), ("S", 3), ("T", 20, 10), ("ADC",)]
    close(epg.simulate(sq.to_ops(epg, tup)), onp.simulate(tup))
    close(epg.simulate(sq.to_ops(epg, tup), max_nstate=4), onp.simulate(tup, max_nstate=4))
    # Phi (general symmetric matrix op) and R
    sm = epg.StateMatrix([1, 1, 0])
    assert np.allclose(epg.Phi(90)(sm).states, [[[1j, -1j, 0]]])
    assert np.allclose(epg.R(0.1 + 0.5j, 0.2, r0=0.2)(epg.StateMatrix([[0.3, 0.3, 0.6]])).states,
                       [[[0.3 * np.exp(-0.1 + 0.5j), 0.3 * np.exp(-0.1 - 0.5j), 0.6 * np.exp(-0.2) + 1 - np.exp(-0.2)]]])


def test_library_rejects_bad_requests():
    ctx = _lib.get_context()
    with pytest.raises(_lib.EpgxError, match="supported capacities"):
        _lib.DeviceState(ctx, 4, 96)
    with pytest.raises(_lib.EpgxError, match="nvox < 1"):
        _lib.DeviceState(ctx, 0, 64)
    enc, _, _ = epg.compile_sequence([epg.T(30, 0), epg.S(1), epg.ADC])
    plan = enc.device_plan(ctx)
    st = _lib.DeviceState(ctx, 2, 64)
    with pytest.raises(_lib.EpgxError, match="holds 2 voxels"):
        _lib.run(ctx, plan, 0, 3, 0, 1, st, st, 64, None, 0, 0)
    with pytest.raises(_lib.EpgxError, match="signal is NULL"):
        _lib.run(ctx, plan, 0, 3, 0, 1, None, None, 64, None, 0, 0)
    with pytest.raises(_lib.EpgxError, match="outside the grid"):
        _lib.run(ctx, plan, 0, 3, 1, 1, None, None, 64, None, 0, 0)
    with pytest.raises(_lib.EpgxError, match="operator range"):
        _lib.run(ctx, plan, 2, 9, 0, 1, None, None, 64, None, 0, 0)
    with pytest.raises(AttributeError, match="kgrid not set"):
        epg.simulate([epg.S([1.5, 0.2]), epg.ADC])      # the reference's own error (shift.py:131-132)
    with pytest.raises(NotImplementedError):
        epg.simulate([epg.S([1.5, 0.2], kgrid=0.1), epg.ADC])   # float wavenumbers: shift-merge, out of scope


def test_combined_operators_on_device():
    """`E @ T` etc.: one combined operator == the two in sequence;
    MatrixOp with a constant term (test/test_opmatrix.py:27-49)"""
    from epgpy_amd.opmatrix import MatrixOp
    from epgpy_amd.opscalar import ScalarOp
    mat = [[0, 1j, 0], [-1j, 0, 0], [0, 0, 1]]
    sm0 = epg.StateMatrix([1, 1, 1])
    assert np.allclose(MatrixOp(mat)(sm0).states, [1j, -1j, 1])
    const = np.diag([0, 0, 0.5])
    op = MatrixOp([mat, mat], [const, const])
    assert op.shape == (2,)
    assert np.allclose(op(sm0).states, [[[1j, -1j, 1.5]], [[1j, -1j, 1.5]]])
    rs = np.random.RandomState(0)
    m = rs.uniform(-1, 1, (3, 3, 2)).dot([1, 1j])
    m += m[..., (1, 0, 2), :][..., (1, 0, 2)].conj()
    m0 = rs.uniform(-1, 1, (3, 3, 2)).dot([1, 1j])
    m0 += m0[..., (1, 0, 2), :][..., (1, 0, 2)].conj()
    assert np.allclose(MatrixOp(m, m0)(sm0).states, m @ [1, 1, 1] + m0 @ [0, 0, 1])
    # E @ T == E then T, on a populated multi-order state with per-voxel parameters and density
    T2 = np.array([35.0, 70.0, 140.0])
    e, t = epg.E(7.5, 900.0, T2, 0.02), epg.T([[30.0, 80.0]], 25.0)
    sm = epg.StateMatrix(density=[1.0, 2.0, 0.5])
    for op in [epg.T(50, 10), epg.S(1), epg.E(3, 500, 60), epg.T(40, 70), epg.S(1)]:
        sm = op(sm)
    seq_states = t(e(sm)).states
    close((e @ t)(sm).states, seq_states)
    close((e @ e)(sm).states, e(e(sm)).states)
    close((t @ e @ t)(sm).states, t(e(t(sm))).states)
    # inside simulate
    blk = [epg.S(1), e @ t, epg.S(1), e, epg.ADC]
    ref = [epg.S(1), e, t, epg.S(1), e, epg.ADC]
    close(epg.simulate([epg.T(90, 90)] + blk * 5), epg.simulate([epg.T(90, 90)] + ref * 5))


# ------------------------------------------------------------------ first-order derivatives
def test_g11_jacobian_golden(golden):
    """Jacobian probes vs the reference's own output (tests/golden/make_golden.py g11)"""
    g = golden("g11_jacobian")
    _, ops, variables = sq.jac_mse(g["T1"], g["T2"], g["B1"])
    close(epg.simulate(ops(epg), probe=epg.Jacobian(variables)), g["jac_mse"])
    _, ops, variables = sq.jac_spgr(g["phases"], g["g"], g["T2b"])
    close(epg.simulate(ops(epg), probe=epg.Jacobian(variables), max_nstate=63), g["jac_spgr"])
    close(epg.simulate(ops(epg), probe=epg.Jacobian(["T2", "fa"], probe="Z0"), max_nstate=63), g["jac_spgr_z"])
    _, ops, variables = sq.jac_params()
    close(epg.simulate(ops(epg), probe=epg.Jacobian(variables)), g["jac3"])


def test_g16_long_jacobian_golden(golden, capfd):
    """derivative plans above 256 orders per voxel (the reference has no limit: diff.py:119-139): 321 orders -> K = 512, three
    derivative states per pass (two passes for the four variables); 601 orders -> K = 1024, one per pass -- against the
    reference's own output; the state column is the plain simulation"""
    g = golden("g16_long_jacobian")
    for necho, K, kernels in ((160, 512, ("deriv_kernel<8, 1, 3>", "deriv_kernel<8, 1, 1, true>")), (300, 1024, ("deriv_kernel<16, 1, 1, true>",))):
        _, ops, variables = sq.jac_long(g["T1"], g["T2"], g["B1"], necho)
        enc, _, _ = epg.compile_sequence(ops(epg))
        assert enc.capacity() == K
        os.environ["EPGX_TRACE"] = "1"
        try:
            capfd.readouterr()
            got = epg.simulate(ops(epg), probe=epg.Jacobian(variables))
            seen = capfd.readouterr().err
        finally:
            del os.environ["EPGX_TRACE"]
        for name in kernels:
            assert name in seen, seen
        assert got.shape == (necho, 3, 5)
        close(got[19::20], g[f"jac_{necho}"], 1e-11)
        plain = epg.simulate([op if not hasattr(op, "order1") else op for op in ops(epg)])
        close(got[..., 0], plain, 1e-12)
    # a ragged grid and B1 on its own axis at K = 512: against the oracle
    rng = np.random.default_rng(16)
    T1, T2, B1 = rng.uniform(300, 2000, (7, 1)), rng.uniform(60, 200, (7, 1)), np.linspace(0.8, 1.2, 3)[None, :]
    tuples, ops, variables = sq.jac_long(T1, T2, B1, 140)
    close(epg.simulate(ops(epg), probe=epg.Jacobian(variables)), onp.simulate_jacobian(tuples, variables), 1e-11)
    # from a given state matrix (the state column is loaded in the layout the kernel keeps it in: consecutive orders per lane)
    sm = epg.StateMatrix(shape=(7, 3))
    for op in [epg.T(30, 10), epg.S(1), epg.E(3.0, 800.0, 90.0), epg.T(50, 0), epg.S(1)]:
        sm = op(sm, inplace=True)
    head = [("T", 30, 10), ("S", 1), ("E", 3.0, 800.0, 90.0, 0), ("T", 50, 0), ("S", 1)]
    ref = onp.simulate_jacobian(head + tuples, variables, shape=(7, 3))
    close(epg.simulate(ops(epg), probe=epg.Jacobian(variables), init=sm), ref, 1e-11)
    # beyond 1024 orders the derivative states have no device form
    _, ops, variables = sq.jac_long(g["T1"], g["T2"], g["B1"], 520)
    with pytest.raises(NotImplementedError):
        epg.simulate(ops(epg), probe=epg.Jacobian(variables))


@pytest.mark.parametrize("nvox,necho", [(1, 3), (777, 12), (4096, 70)])
def test_jacobian_vs_oracle(nvox, necho):
    """seeded grids, K = 64 / 64 / 256 (70 echoes -> 141 orders), ragged voxel counts"""
    rng = np.random.default_rng(nvox)
    T1, T2, B1 = rng.uniform(300, 2500, nvox), rng.uniform(20, 300, nvox), rng.uniform(0.7, 1.3, nvox)
    tuples, ops, variables = sq.jac_mse(T1, T2, B1, necho=necho)
    got = epg.simulate(ops(epg), probe=epg.Jacobian(variables))
    n = min(nvox, 64)   # the NumPy oracle carries 4 full state matrices: check a slice
    tuples, _, _ = sq.jac_mse(T1[:n], T2[:n], B1[:n], necho=necho)
    close(got[:, :n], onp.simulate_jacobian(tuples, variables))
    # the undifferentiated signal is the plain simulation (bit for bit without the E.T.E fusion)
    plain = epg.simulate(sq.mse_ops(epg, T1, T2, B1, necho=necho))
    close(got[..., 0], plain)
    unfused = epg.simulate(ops(epg), probe=epg.Jacobian(variables), fuse=False)
    assert np.array_equal(unfused[..., 0], epg.simulate(sq.mse_ops(epg, T1, T2, B1, necho=necho), fuse=False))
    close(unfused[:, :n], onp.simulate_jacobian(tuples, variables))
    # one variable per plan: the E . T . E runs of the differentiated train are fused too, their partials generated by
    # the library (epgx_fuse_partial) -- the same derivative to rounding, the state column that of the fused plain plan
    for v, var in enumerate(variables):
        if var == "magnitude":                   # (d/d magnitude is the signal itself: column 0 below)
            continue
        one = epg.simulate(ops(epg), probe=epg.Jacobian(["magnitude", var]))
        close(one[..., 1], unfused[..., v])
        close(one[:, :n, 1], onp.simulate_jacobian(tuples, [var])[..., 0])
        if necho <= 31:
            assert np.array_equal(one[..., 0], plain)


def test_jacobian_multi_axis_grid_and_mixed_probes():
    """2-D grid (T2 x off-resonance), several probes per ADC, unknown variable -> zeros"""
    g = np.linspace(-0.03, 0.03, 5)[None, :]
    T2b = np.linspace(40, 120, 7)
    phases = 58.5 * np.arange(12) ** 2
    tuples, ops, _ = sq.jac_spgr(phases, g, T2b)
    jac, sig, jz = epg.simulate(ops(epg), probe=[epg.Jacobian(["T2", "zzz", "g", "phi0", "fa"]), "F0",
                                                 epg.Jacobian(["magnitude", "fa"], probe="Z0")], max_nstate=20)
    ref = onp.simulate_jacobian(tuples, ["T2", "zzz", "g", "phi0", "fa", "magnitude"], max_nstate=20)
    close(jac, ref[..., :5])
    close(sig, ref[..., 5])
    close(jz, onp.simulate_jacobian(tuples, ["magnitude", "fa"], probe="Z0", max_nstate=20))
    assert not jac[..., 1].any()


def test_jacobian_limits():
    seq = [epg.T(30, 0, order1=True), epg.ADC]
    with pytest.raises(NotImplementedError):
        epg.simulate(seq, probe=epg.Jacobian("alpha"), mode="stream")
    with pytest.raises(NotImplementedError):
        epg.simulate([epg.T(30, 0, order1=True)] + [epg.S(1)] * 1100 + [epg.ADC], probe=epg.Jacobian("alpha"))
    sm = epg.T(30, 0, order1=True)(epg.StateMatrix())           # operator-by-operator: sm.order1 (tested below)
    assert set(sm.order1) == {"alpha", "phi"}
    # a Jacobian nobody feeds: plain kernel, zeros
    out = epg.simulate([epg.T(30, 0), epg.ADC], probe=epg.Jacobian(["magnitude", "alpha"]))
    assert out.shape == (1, 1, 2) and out[0, 0, 1] == 0


def test_reference_diff_tests_through_simulate():
    """test/test_diff.py:282-331 (test_diff_chain_mse) and :515-548 (test_jacobian_class), the parts
    that go through simulate(); the op-by-op values they compare with are finite differences here"""
    exc = epg.T(90, 90, name="exc")
    ref = epg.T(150, 0, order1="alpha", name="ref")
    relax = epg.E(5, 1e3, 35, order1="T2", name="relax")
    grad = epg.S(1, name="grad")
    necho = 5
    assert ref.parameters_order1 == {"alpha"} and relax.parameters_order1 == {"T2"}
    spinecho = [exc] + [grad, relax, ref, grad, relax, epg.ADC] * necho
    probe = ["F0", epg.Jacobian("T2"), epg.Jacobian("alpha")]
    signal, gradT2, gradalpha = epg.simulate(spinecho, init=[0, 0, 1], probe=probe)
    assert signal.shape == (5, 1) and gradT2.shape == (5, 1, 1) and gradalpha.shape == (5, 1, 1)

    def mse(alpha, T2):
        return epg.simulate([exc] + [grad, epg.E(5, 1e3, T2), epg.T(alpha, 0), grad, epg.E(5, 1e3, T2), epg.ADC] * necho)

    close(signal, mse(150, 35))
    h = 1e-5
    close(gradT2[..., 0], (mse(150, 35 + h) - mse(150, 35 - h)) / (2 * h), tol=1e-8)
    close(gradalpha[..., 0], (mse(150 + h, 35) - mse(150 - h, 35)) / (2 * h), tol=1e-8)

    rf = epg.T(15, 90, order1=["alpha"])
    rlx = epg.E(5, 1e3, 30, order1=["T2"])
    seq = [rf, rlx, epg.S(1), epg.ADC] * necho
    jac1, jac2, jac3 = epg.simulate(seq, probe=[epg.Jacobian(["alpha"]), epg.Jacobian(["alpha", "T2"]),
                                                epg.Jacobian(["magnitude", "alpha"])])
    assert jac1.shape == (5, 1, 1) and jac2.shape == (5, 1, 2) and jac3.shape == (5, 1, 2)
    tuples = [("T", 15, 90, {"order1": {"alpha": {"alpha": 1}}}), ("E", 5, 1e3, 30, 0, {"order1": {"T2": {"T2": 1}}}),
              ("S", 1), ("ADC",)] * necho
    ref_jac = onp.simulate_jacobian(tuples, ["magnitude", "alpha", "T2"])
    close(jac1[..., 0], ref_jac[..., 1])
    close(jac2, ref_jac[..., 1:])
    close(jac3, ref_jac[..., :2])


def test_jacobian_from_initial_state():
    """derivatives from a prepared (non-equilibrium, pre-sized) state: the plain signal is unchanged"""
    rng = np.random.default_rng(5)
    T2 = rng.uniform(30, 120, 33)
    prep = [epg.T(35, 10), epg.S(1), epg.E(3, 900, T2)]
    sm = epg.StateMatrix(shape=(33,), max_nstate=20)
    for op in prep:
        sm = op(sm)
    seq = [epg.T(20, 0, order1="alpha"), epg.E(4, 900, T2, order1="T2"), epg.ADC, epg.S(1)] * 8
    jac = epg.simulate(seq, init=sm, probe=epg.Jacobian(["magnitude", "alpha", "T2"]))
    plain = epg.simulate([epg.T(20, 0), epg.E(4, 900, T2), epg.ADC, epg.S(1)] * 8, init=sm, fuse=False)
    assert np.array_equal(jac[..., 0], plain)
    h = 1e-5
    up = epg.simulate([epg.T(20 + h, 0), epg.E(4, 900, T2), epg.ADC, epg.S(1)] * 8, init=sm)
    dn = epg.simulate([epg.T(20 - h, 0), epg.E(4, 900, T2), epg.ADC, epg.S(1)] * 8, init=sm)
    close(jac[..., 1], (up - dn) / (2 * h), tol=1e-8)


# ------------------------------------------------------------------ randomized differential test
@pytest.mark.parametrize("seed", range(40))
def test_random_sequences_vs_oracle(seed):
    """random operator sequences, grids and broadcast patterns: device (all three modes) vs the
    NumPy oracle -- signals at every probe and the final state matrix"""
    rng = np.random.default_rng(1000 + seed)
    grid = tuple(int(x) for x in rng.integers(1, 6, rng.integers(1, 4)))
    cap = [None, None, 3, 7, 20][int(rng.integers(0, 5))]
    tuples = sq.random_sequence(rng, grid, nops=int(rng.integers(10, 60)))
    ref, ref_states = onp.simulate(tuples, shape=grid, max_nstate=cap, return_states=True)
    ops = sq.to_ops(epg, tuples)
    opts = {"max_nstate": cap} if cap else {}
    init = epg.StateMatrix(shape=grid, **opts)
    for mode in ("resident", "stream", "stepwise"):
        got = epg.simulate(ops, init=init, mode=mode, **opts)
        close(np.asarray(got), ref)
    sm = epg.StateMatrix(shape=grid, **opts)
    for op in ops:
        sm = op(sm, inplace=True)
    n = (ref_states.shape[-2] - 1) // 2
    assert sm.nstate == n
    close(sm.states, ref_states)


# ------------------------------------------------------------------ config 5 vs golden G12 and the oracle
def _coord_rows(coords, kdim):
    """[.., R, d] -> R hashable rows: a row is its coordinate in every voxel (padded to kdim components)"""
    coords = np.asarray(coords)
    coords = np.concatenate([coords, np.zeros(coords.shape[:-1] + (kdim - coords.shape[-1],), coords.dtype)], axis=-1)
    rows = np.moveaxis(coords, -2, 0).reshape(coords.shape[-2], -1)
    return [tuple(int(v) for v in r) for r in rows]


def _match_states(sm, ref_states, ref_coords):
    """the device keeps every structurally reachable coordinate (no value-based pruning): rows the
    reference kept must agree, rows it pruned must be ~0 (its tolerance: 1e-8)"""
    kdim = np.asarray(sm.coords).shape[-1]
    mine = _coord_rows(sm.coords, kdim)
    states = np.asarray(sm.states)
    lookup = {c: i for i, c in enumerate(mine)}
    per_voxel = len(mine[0]) > kdim
    seen = set()
    for r, key in enumerate(_coord_rows(ref_coords, kdim)):
        if per_voxel and len(key) == kdim:          # the oracle still had shared coordinates
            key = key * (len(mine[0]) // kdim)
        if key in lookup:
            close(states[..., lookup[key], :], ref_states[..., r, :])
            seen.add(lookup[key])
        else:
            assert np.max(np.abs(ref_states[..., r, :])) < 1e-8
    rest = [i for i in range(len(mine)) if i not in seen]
    if rest:
        assert np.max(np.abs(states[..., rest, :])) < 1e-8


def test_g12_nd_golden(golden):
    g = golden("g12_nd")
    for name, tuples, opts in sq.nd_cases():
        ops = sq.nd_to_ops(epg, tuples)
        for mode in ("resident", "stream", "stepwise"):
            close(epg.simulate(ops, mode=mode, **opts), g[name + "_signal"])
        sm = epg.StateMatrix(shape=epg.getshape(ops), **opts)
        for op in ops:
            sm = op(sm, inplace=True)
        _match_states(sm, g[name + "_states"], g[name + "_coords"])


def test_g14_vectorised_nd_shifts_golden(golden):
    """one shift vector per voxel (shift.py:38-41): signals in every mode, final states and per-voxel coordinates of the
    reference; `coords=` takes them back (statematrix.py:58-64)"""
    g = golden("g14_nd_vector")
    for name, tuples, opts in sq.nd_vector_cases():
        ops = sq.nd_to_ops(epg, tuples)
        for mode in ("resident", "stream", "stepwise"):
            close(epg.simulate(ops, mode=mode, **opts), g[name + "_signal"])
        sm = epg.StateMatrix(shape=epg.getshape(ops), **opts)
        for op in ops:
            sm = op(sm, inplace=True)
        assert sm.coords.shape[:-2] == g[name + "_coords"].shape[:-2] and sm.kdim == g[name + "_coords"].shape[-1]
        _match_states(sm, g[name + "_states"], g[name + "_coords"])
        # hand the coordinates over to a new state matrix and carry on from there: same as carrying on directly
        again = epg.StateMatrix(sm.states, coords=sm.coords, **opts)
        assert np.array_equal(again.coords, sm.coords) and again.nstate == sm.nstate
        more = sq.nd_to_ops(epg, [("T", 75, 10), ("S", tuples[1][1] if name == "dwi_dirs" else [1, -1]), ("T", 40, -60), ("S", -1)])
        a, b = sm, again
        for op in more:
            a, b = op(a), op(b)
        assert np.array_equal(a.coords, b.coords)
        close(b.F0, a.F0)
        twin = sm.copy(coords=sm.coords)
        assert np.array_equal(twin.coords, sm.coords)
    with pytest.raises(ValueError):
        epg.StateMatrix([[0, 0, 0], [0, 0, 1], [0, 0, 0]], coords=[[1, 0], [0, 0], [1, 0]])      # not symmetric
    sm1 = epg.S([[1, 0, 0], [2, 0, 0]])(epg.StateMatrix([1, 1, 0]))                              # test_shift.py:196-203
    assert sm1.shape == (2,) and sm1.kdim == 3 and sm1.nstate == 1
    close(sm1.states, np.broadcast_to([[0, 1, 0], [0, 0, 0], [1, 0, 0]], (2, 3, 3)))
    assert np.allclose(sm1.k[0] * 2, sm1.k[1])


@pytest.mark.parametrize("seed", range(12))
def test_random_vectorised_nd_sequences_vs_oracle(seed):
    """random sequences in which some shifts carry one vector per voxel of the leading grid axes; diffusion (scalar,
    per-voxel field, tensor) then sees a different wavenumber in every voxel"""
    rng = np.random.default_rng(7000 + seed)
    grid = tuple(int(x) for x in rng.integers(2, 5, rng.integers(1, 3)))
    kdim = int(rng.integers(1, 4))
    cap = [None, None, 3, 5][int(rng.integers(0, 4))]
    kvalue = [float(v) for v in rng.uniform(5e3, 4e4, 3)]
    tuples = sq.random_nd_sequence(rng, grid, kdim, nops=int(rng.integers(8, 26)), vector=True)
    opts = {"kvalue": kvalue}
    if cap:
        opts["max_nstate"] = cap
    ref, (ref_states, ref_coords) = onp.simulate_nd(tuples, shape=grid, return_states=True, **opts)
    ops = sq.nd_to_ops(epg, tuples)
    for mode in ("resident", "stream"):
        close(np.asarray(epg.simulate(ops, init=epg.StateMatrix(shape=grid, **opts), mode=mode, **opts)), ref)
    if ref_coords is not None:
        sm = epg.StateMatrix(shape=grid, **opts)
        for op in ops:
            sm = op(sm, inplace=True)
        _match_states(sm, ref_states, ref_coords)


@pytest.mark.parametrize("seed", range(24))
def test_random_nd_sequences_vs_oracle(seed):
    rng = np.random.default_rng(5000 + seed)
    grid = tuple(int(x) for x in rng.integers(1, 5, rng.integers(1, 3)))
    kdim = int(rng.integers(1, 4))
    cap = [None, None, 2, 4][int(rng.integers(0, 4))]
    kvalue = [float(v) for v in rng.uniform(5e3, 4e4, 3)]
    tuples = sq.random_nd_sequence(rng, grid, kdim, nops=int(rng.integers(8, 30)))
    opts = {"kvalue": kvalue}
    if cap:
        opts["max_nstate"] = cap
    ref, (ref_states, ref_coords) = onp.simulate_nd(tuples, shape=grid, return_states=True, **opts)
    ops = sq.nd_to_ops(epg, tuples)
    for mode in ("resident", "stream"):
        close(np.asarray(epg.simulate(ops, init=epg.StateMatrix(shape=grid, **opts), mode=mode, **opts)), ref)
    if ref_coords is not None:
        sm = epg.StateMatrix(shape=grid, **opts)
        for op in ops:
            sm = op(sm, inplace=True)
        _match_states(sm, ref_states, ref_coords)


def test_gather_shifts_beyond_512_coordinates():
    """an n-D shift sequence whose coordinate set outgrows 512 entries (found by the fuzz run: seed 50479 of the test above): at
    K = 1024 the gather stages F and conj(F-) first, then Z through the same LDS area (three arrays of 1024 complex for the four
    wavefronts of a block would not fit a CU's 160 KiB) -- both modes and operator by operator"""
    rng = np.random.default_rng(5000 + 50479)
    grid = tuple(int(x) for x in rng.integers(1, 5, rng.integers(1, 3)))
    kdim = int(rng.integers(1, 4))
    rng.integers(0, 4)
    kvalue = [float(v) for v in rng.uniform(5e3, 4e4, 3)]
    tuples = sq.random_nd_sequence(rng, grid, kdim, nops=int(rng.integers(8, 30)))
    ops = sq.nd_to_ops(epg, tuples)
    enc, _, _ = epg.compile_sequence(ops, shape=grid, options={"kvalue": kvalue})
    assert enc.capacity() == 1024
    test_random_nd_sequences_vs_oracle(50479)


@pytest.mark.parametrize("seed", range(16))
def test_random_jacobians_vs_oracle(seed):
    """random differentiated sequences: 5 shared variables (two device passes), random coefficients,
    per-voxel and broadcast parameters, shifts by +-1 / +-2, optional truncation"""
    rng = np.random.default_rng(9000 + seed)
    grid = tuple(int(x) for x in rng.integers(1, 5, rng.integers(1, 3)))
    cap = [None, 3, 10][int(rng.integers(0, 3))]
    tuples, ops, variables = sq.random_jacobian_sequence(rng, grid, nops=int(rng.integers(8, 40)))
    ref = onp.simulate_jacobian(tuples, variables, shape=grid, max_nstate=cap)
    got = epg.simulate(ops(epg), probe=epg.Jacobian(variables), init=epg.StateMatrix(shape=grid), **({"max_nstate": cap} if cap else {}))
    close(got, ref, tol=1e-11)
    ref_z = onp.simulate_jacobian(tuples, variables[1:3], probe="Z0", shape=grid, max_nstate=cap)
    got_z = epg.simulate(ops(epg), probe=epg.Jacobian(variables[1:3], probe="Z0"), init=epg.StateMatrix(shape=grid),
                         **({"max_nstate": cap} if cap else {}))
    close(got_z, ref_z, tol=1e-11)


# ------------------------------------------------------------------ Adc(weights, reduce) on the device
def test_adc_weights_and_reduce_on_device():
    """probe.py:141-165 semantics; the sums run in epgx_signal_reduce (both kernel variants: innermost
    axis reduced / kept), compared with NumPy on the full signal"""
    rng = np.random.default_rng(3)
    T2 = rng.uniform(30, 200, 6)[:, None, None]
    g = rng.uniform(-0.05, 0.05, 5)[None, :, None]
    B1 = rng.uniform(0.7, 1.2, 70)[None, None, :]
    body = [epg.T(90 * B1, 90), epg.S(1), epg.E(8, 1000, T2, g), epg.T(160 * B1, 0), epg.S(1), epg.E(8, 1000, T2, g)]
    full = epg.simulate(body + [epg.ADC, epg.E(3, 1000, T2, g), epg.ADC])          # (2, 6, 5, 70)
    assert full.shape == (2, 6, 5, 70)
    w1 = rng.uniform(0, 1, 6)
    w3 = rng.uniform(0, 1, (1, 1, 70)) + 1j * rng.uniform(0, 1, (1, 1, 70))
    w23 = rng.uniform(0, 1, (1, 5, 70))
    cases = [
        (dict(reduce=True), lambda s: s.sum(axis=(1, 2, 3))),
        (dict(reduce=0), lambda s: s.sum(axis=1)),
        (dict(reduce=2), lambda s: s.sum(axis=3)),
        (dict(reduce=(0, 2)), lambda s: s.sum(axis=(1, 3))),
        (dict(weights=w1), lambda s: (s * w1[None, :, None, None]).sum(axis=1)),
        (dict(weights=w3, reduce=2), lambda s: (s * w3[None]).sum(axis=3)),
        (dict(weights=w23, reduce=(1, 2)), lambda s: (s * w23[None]).sum(axis=(2, 3))),
        (dict(weights=w23, reduce=True), lambda s: (s * w23[None]).sum(axis=(1, 2, 3))),
        (dict(weights=w3, reduce=(0, 2), phase=30.0), lambda s: (s * w3[None]).sum(axis=(1, 3)) * np.exp(1j * np.pi / 6)),
    ]
    for kwargs, ref in cases:
        adc = epg.Adc(**kwargs)
        got = epg.simulate(body + [adc, epg.E(3, 1000, T2, g), adc])
        close(got, ref(full), tol=1e-11)
    # the reference's own case (test/test_functions.py:66-74): reduce=1 with row weights
    exc, refoc, grad = epg.T(90, 90), epg.T(180, 0), epg.S(1, duration=10)
    relaxn = epg.E(10, 1000, [[30], [40], [50]], g=[[-0.1, -0.05, 0, 0.05, 1]])
    res_ = epg.simulate([exc, grad, relaxn, refoc, grad, relaxn, epg.ADC])
    resn = epg.simulate([exc, grad, relaxn, refoc, grad, relaxn, epg.Adc(reduce=1, weights=[[1, 2, 3, 4, 5]])])
    close(resn, np.dot(res_, [1, 2, 3, 4, 5]))
    # several probes per ADC, only one of them reducing; replaced probe keeps the ADC's phase
    sig, summed, z = epg.simulate(body + [epg.Adc(phase=10.0)], probe=["F0", epg.Adc(reduce=(1, 2)), epg.Adc("Z0", reduce=True)])
    ph = np.exp(1j * np.pi / 18)
    close(sig, full[:1] * ph)
    close(summed, full[:1].sum(axis=(2, 3)) * ph, tol=1e-11)
    zfull = epg.simulate(body + [epg.Adc("Z0")])
    close(z, zfull.sum(axis=(1, 2, 3)) * ph, tol=1e-11)
    # weights that do not fit the grid: NumPy's own error from the host path
    with pytest.raises(ValueError):
        epg.simulate(body + [epg.Adc(weights=np.ones(4), reduce=0)])


def test_jacobian_across_plain_operators(golden):
    """default: SPOILER / D act on the state only, as in the reference (golden up to the first RESET);
    exact_partials=True: they act on the derivative states too = finite differences of the signal"""
    g = golden("g11_jacobian")
    T2b = g["T2b"]
    tuples, ops, variables = sq.jac_plain_ops(T2b)
    got = epg.simulate(ops(epg), probe=epg.Jacobian(variables))
    close(got[:5], g["jac_plain"][:5])
    close(got, onp.simulate_jacobian(tuples, variables))
    exact = epg.simulate(ops(epg), probe=epg.Jacobian(variables), exact_partials=True)
    close(exact, onp.simulate_jacobian(tuples, variables, through_plain=True))
    h = 1e-5

    def signal(da, dT2):
        e = epg.E(5, 1000, T2b + dT2)
        return epg.simulate([epg.T(30 + da, 0), e, epg.ADC, epg.SPOILER, epg.ADC, epg.T(20 + da, 0), epg.S(1), e, epg.ADC,
                             epg.ADC, epg.RESET, epg.ADC, epg.T(50 + da, 90), epg.ADC, epg.PD(0.7), epg.T(40, 0), epg.ADC])

    close(exact[..., 1], (signal(h, 0) - signal(-h, 0)) / (2 * h), tol=1e-8)
    close(exact[..., 2], (signal(0, h) - signal(0, -h)) / (2 * h), tol=1e-8)
    # diffusion between the pulses: attenuates the derivative only with exact_partials
    seq = [epg.T(30, 0, order1="alpha"), epg.S(1), epg.D(5, 1e-3), epg.S(-1), epg.ADC]
    ref_like = epg.simulate(seq, probe=epg.Jacobian(["magnitude", "alpha"]), kvalue=1e5)
    exact = epg.simulate(seq, probe=epg.Jacobian(["magnitude", "alpha"]), kvalue=1e5, exact_partials=True)
    att = abs(ref_like[0, 0, 0]) / np.sin(np.pi / 6)
    assert np.isclose(abs(ref_like[0, 0, 1]), np.cos(np.pi / 6) * np.pi / 180, atol=1e-12)          # reference: 0.01511499
    assert np.isclose(abs(exact[0, 0, 1]), att * np.cos(np.pi / 6) * np.pi / 180, atol=1e-12)


@pytest.mark.parametrize("seed", range(32))
def test_random_fused_sequences_vs_oracle(seed):
    """precession-free random sequences: E.T.E runs are fused (device-generated T0 tables), echoes become
    single records with a leading shift; all modes, fused and unfused, K = 64 ... 512, vs the oracle"""
    from epgpy_amd import functions
    rng = np.random.default_rng(7000 + seed)
    grid = tuple(int(x) for x in rng.integers(1, 6, rng.integers(1, 4)))
    cap = [None, None, 5, 63, 100][int(rng.integers(0, 5))]
    tuples = sq.random_sequence(rng, grid, nops=int(rng.integers(20, 400 if seed % 4 == 0 else 80)), precession=False)
    ref, ref_states = onp.simulate(tuples, shape=grid, max_nstate=cap, return_states=True)
    ops = sq.to_ops(epg, tuples)
    opts = {"max_nstate": cap} if cap else {}
    init = epg.StateMatrix(shape=grid, **opts)
    for mode in ("resident", "stream"):
        close(np.asarray(epg.simulate(ops, init=init, mode=mode, **opts)), ref, tol=1e-11)
    close(np.asarray(epg.simulate(ops, init=init, fuse=False, **opts)), ref, tol=1e-11)
    if seed < 8:   # the pass really fires on these sequences
        enc, _, _ = functions.compile_sequence(ops, shape=grid, options=opts)
        assert any(r[0] == _lib.OP_T0 for r in enc.records) or not any(t[0] == "E" for t in tuples)


def test_order1_operator_by_operator():
    """op(sm) keeps sm.order1 up to date (DiffOperator.__call__, diff.py:119-139): the op-by-op part of
    test/test_diff.py:282-331 (test_diff_chain_mse) and :515-548, against finite differences, the
    oracle, and the in-kernel Jacobian of simulate()"""
    exc = epg.T(90, 90, name="exc")
    ref = epg.T(150, 0, order1="alpha", name="ref")
    relax = epg.E(5, 1e3, 35, order1="T2", name="relax")
    grad = epg.S(1, name="grad")
    necho = 5
    seq = [exc] + [grad, relax, ref, grad, relax] * necho
    rlx_T2, ref_alpha = epg.E(5, 1e3, 35 + 1e-6), epg.T(150 + 1e-6, 0)
    sm = epg.StateMatrix([0, 0, 1])
    sm_T2, sm_alpha = sm.copy(), sm.copy()
    plain = {"exc": epg.T(90, 90), "ref": epg.T(150, 0), "relax": epg.E(5, 1e3, 35), "grad": epg.S(1)}
    for op in seq:
        sm = op(sm)
        sm_T2 = (rlx_T2 if op.name == "relax" else plain[op.name])(sm_T2)
        sm_alpha = (ref_alpha if op.name == "ref" else plain[op.name])(sm_alpha)
    assert set(sm.order1) == {"T2", "alpha"}
    assert sm.order1["T2"].nstate == sm.nstate == 10
    assert np.allclose((sm_T2.states - sm.states) * 1e6, sm.order1["T2"].states, atol=1e-5)
    assert np.allclose((sm_alpha.states - sm.states) * 1e6, sm.order1["alpha"].states, atol=1e-5)
    assert not sm.order1["T2"].density.any()          # derivative states carry no equilibrium (diff.py:103-109)
    # same numbers as the fused in-kernel propagation
    spinecho = [exc] + [grad, relax, ref, grad, relax, epg.ADC] * necho
    signal, gT2, galpha = epg.simulate(spinecho, probe=["F0", epg.Jacobian("T2"), epg.Jacobian("alpha")])
    close(signal[-1], sm.F0)
    close(gT2[-1, ..., 0], sm.order1["T2"].F0)
    close(galpha[-1, ..., 0], sm.order1["alpha"].F0)
    # non-inplace calls leave the input's derivative states alone; plain operators drop them on a copy
    before = sm.order1["T2"].states
    sm2 = relax(sm)
    assert np.array_equal(sm.order1["T2"].states, before) and sm2.order1["T2"] is not sm.order1["T2"]
    assert not getattr(epg.SPOILER(sm), "order1", None)
    # grids: per-voxel parameters, the derivative states follow the broadcast
    T2 = np.array([30.0, 60.0, 90.0])
    rl = epg.E(5, 1e3, T2, order1="T2")
    sm = epg.StateMatrix()
    for op in [epg.T(20, 0, order1="alpha"), rl, epg.S(1), epg.T(35, 90, order1={"alpha": {"alpha": 2.0}}), rl]:
        sm = op(sm)
    tuples = [("T", 20, 0, {"order1": {"alpha": {"alpha": 1}}}), ("E", 5, 1e3, T2, 0, {"order1": {"T2": {"T2": 1}}}), ("S", 1),
              ("T", 35, 90, {"order1": {"alpha": {"alpha": 2.0}}}), ("E", 5, 1e3, T2, 0, {"order1": {"T2": {"T2": 1}}}), ("ADC",)]
    jac = onp.simulate_jacobian(tuples, ["magnitude", "alpha", "T2"])
    close(sm.F0, jac[0, :, 0])
    close(sm.order1["alpha"].F0, jac[0, :, 1])
    close(sm.order1["T2"].F0, jac[0, :, 2])


def test_jacobian_with_callback_runs_stepwise():
    """a callback forces the operator-by-operator path: Jacobian probes then read sm.order1"""
    T2 = np.array([40.0, 80.0])
    seq = [epg.T(90, 90)] + [epg.S(1), epg.E(5, 900, T2, order1="T2"), epg.T(160, 0, order1={"fa": "alpha"}), epg.S(1),
                             epg.E(5, 900, T2, order1="T2"), epg.ADC] * 4
    seen = []
    stepwise = epg.simulate(seq, probe=epg.Jacobian(["magnitude", "T2", "fa"]), callback=lambda sm: seen.append(sm.nstate))
    resident = epg.simulate(seq, probe=epg.Jacobian(["magnitude", "T2", "fa"]))
    assert len(seen) == 1 + 5 * 4 and seen[-1] == 8      # probes excluded, as in the reference loop
    close(stepwise, resident)


# ------------------------------------------------------------------ 16 orders per voxel: four voxels per wavefront
@pytest.mark.parametrize("seed", range(24))
def test_packed_kernel_is_bit_identical(seed):
    """max_nstate <= 15 (the reference's usual MRF setting is 10): state-resident runs go four voxels per
    wavefront; same instruction sequence per k-state, so the signals equal the one-voxel-per-wave
    kernel's bit for bit, and the oracle to tolerance.  Ragged voxel counts, all record kinds."""
    from epgpy_amd import functions
    rng = np.random.default_rng(11000 + seed)
    grid = tuple(int(x) for x in rng.integers(1, 7, rng.integers(1, 4)))
    cap = int(rng.choice([1, 3, 10, 15, 16, 20, 31]))
    tuples = [t for t in sq.random_sequence(rng, grid, nops=int(rng.integers(10, 120)), precession=bool(seed % 2))
              if not (t[0] == "S" and abs(t[1]) > 1)]
    ops = sq.to_ops(epg, tuples)
    enc, _, _ = functions.compile_sequence(ops, options={"max_nstate": cap})
    assert enc.packable()
    ref = onp.simulate(tuples, max_nstate=cap)              # the grid the operators span themselves
    packed = np.asarray(epg.simulate(ops, max_nstate=cap))
    plain = np.asarray(epg.simulate(ops, max_nstate=cap, packed=False))
    assert packed.shape == ref.shape and sq.same_bits(packed, plain, x64=True)     # (packed=False: the 64-order kernel)
    close(packed, ref, tol=1e-11)


def test_packed_kernel_large_grid_and_limits():
    from epgpy_amd import functions
    rng = np.random.default_rng(4)
    n = 100003                                  # not a multiple of 4 or 16
    T1, T2, B1 = rng.uniform(300, 2500, n), rng.uniform(20, 300, n), rng.uniform(0.7, 1.3, n)
    alpha, TR = sq.mrf_trains(60)
    ops = sq.mrf_ops(epg, T1, T2, B1, alpha, TR)
    a = epg.simulate(ops, max_nstate=10)
    b = epg.simulate(ops, max_nstate=10, packed=False)
    assert a.shape == (60, n) and np.array_equal(a, b)
    idx = rng.integers(0, n, 32)
    close(a[:, idx], onp.simulate(sq.mrf_tuples(T1[idx], T2[idx], B1[idx], alpha, TR), max_nstate=10), tol=1e-11)
    # more than 15 orders, shifts by 2, diffusion: the one-voxel-per-wave kernel
    assert functions.compile_sequence(ops, options={"max_nstate": 16})[0].packable() == 32
    assert not functions.compile_sequence(ops, options={"max_nstate": 32})[0].packable()
    c = epg.simulate(ops[:300], max_nstate=25)
    assert np.array_equal(c, epg.simulate(ops[:300], max_nstate=25, packed=False))
    assert not functions.compile_sequence([epg.T(30, 0), epg.S(2), epg.ADC], options={"max_nstate": 8})[0].packable()
    ctx = _lib.get_context()
    enc, _, _ = functions.compile_sequence([epg.T(30, 0), epg.S(2), epg.ADC], options={"max_nstate": 8})
    plan = enc.device_plan(ctx, 64)
    sig = _lib.DeviceBuffer(ctx, 16)
    with pytest.raises(_lib.EpgxError):
        _lib.run(ctx, plan, 0, plan.n_ops, 0, 1, None, None, 16, sig.ptr.value, 1, 0)


@pytest.mark.parametrize("seed", range(12))
def test_packed_jacobians_vs_oracle(seed):
    """derivative plans with at most 16 orders: four voxels per wavefront (packed_deriv_kernel) vs the
    one-voxel kernel and the oracle; 5 variables (two passes), ragged voxel counts"""
    rng = np.random.default_rng(13000 + seed)
    grid = tuple(int(x) for x in rng.integers(1, 6, rng.integers(1, 3)))
    cap = int(rng.choice([2, 10, 15, 24, 31]))
    tuples, ops, variables = sq.random_jacobian_sequence(rng, grid, nops=int(rng.integers(8, 60)))
    tuples = [t for t in tuples if not (t[0] == "S" and abs(t[1]) > 1)]
    seq = [o for o in ops(epg) if not (isinstance(o, epg.S) and abs(o.k) > 1)]
    ref = onp.simulate_jacobian(tuples, variables, max_nstate=cap)
    got = epg.simulate(seq, probe=epg.Jacobian(variables), max_nstate=cap)
    one = epg.simulate(seq, probe=epg.Jacobian(variables), max_nstate=cap, packed=False)
    assert got.shape == ref.shape
    close(got, ref, tol=1e-11)
    close(got, one, tol=1e-12)
    assert np.array_equal(got[..., 0], one[..., 0])          # the state itself: same instruction chains


# ------------------------------------------------------------------ second-order derivatives (operator by operator)
def test_g13_hessian_golden(golden):
    """order2= / Hessian (diff.py:290-379, :419-472) against the reference's own output: the tutorial's
    2 x 2 Hessian with its Jacobian, "magnitude" rows, a (T2, g) grid with selected cross derivatives,
    coefficients on shared variables, Z0"""
    g = golden("g13_hessian")
    for name, seq, probes, opts in sq.hessian_cases(epg):
        res = epg.simulate(seq, probe=probes, **opts)
        for i, arr in enumerate(res):
            assert np.asarray(arr).shape == g[f"{name}_{i}"].shape
            close(arr, g[f"{name}_{i}"], tol=1e-11)
    # second derivatives = finite differences of the in-kernel first derivatives
    T2, h = 30.0, 1e-4
    def jac(alpha, t2):
        rlx = epg.E(4.5, 1400, t2, order1="T2")
        inv = epg.T(alpha, 0, order1="alpha")
        seq = [epg.T(90, 90)] + [epg.S(1), rlx, inv, epg.S(1), rlx, epg.ADC] * 6
        return epg.simulate(seq, probe=epg.Jacobian(["alpha", "T2"]))
    hes = g["tutorial_0"]                                        # (6, 1, 2, 2): rows / columns alpha, T2
    d_alpha = (jac(150 + h, T2) - jac(150 - h, T2)) / (2 * h)    # d/dalpha of (d/dalpha, d/dT2)
    d_T2 = (jac(150, T2 + h) - jac(150, T2 - h)) / (2 * h)
    close(hes[..., 0, :], d_alpha, tol=1e-7)
    close(hes[..., 1, :], d_T2, tol=1e-7)


def test_partials_pruner_callback():
    """test/test_diff.py:615-650 (test_partials_pruner_class): pruning negligible partials changes nothing
    measurable and removes the pruned variable from the state's order1"""
    necho = 5
    rf = epg.T(15, 90, order1=["alpha"], order2="alpha")
    rlx = epg.E(5, 1e3, 30, order1=["T2"])
    seq = [rf, rlx, epg.S(1), epg.ADC] * necho
    probe = [epg.Jacobian(["T2", "alpha"]), epg.Hessian("alpha")]
    jac1, hes1 = epg.simulate(seq, probe=probe)
    seen = {}
    pruner = epg.PartialsPruner(condition=1e-2, variables=["T2"])

    def callback(sm):
        pruner(sm)
        seen.update({k: True for k in sm.order1})
    jac2, hes2 = epg.simulate(seq, probe=probe, callback=callback)
    assert jac2.shape == jac1.shape and hes2.shape == hes1.shape
    close(hes2, hes1)                                     # alpha is never pruned
    close(jac2[..., 1], jac1[..., 1])
    assert np.max(np.abs(jac2[..., 0] - jac1[..., 0])) < 1e-2   # T2 partials below the threshold are dropped
    assert repr(pruner) == "PartialsPruner(1 variables)"


@pytest.mark.parametrize("max_nstate", [5, 12, 25, 63, 100])
@pytest.mark.parametrize("fuse", [True, False])
def test_runs_of_identical_records(max_nstate, fuse):
    """state-resident launches fold runs of identical records (an echo train) into one record with a repeat
    count (rows_run): odd / even run lengths, runs with and without a rotation or an ADC, truncating shifts,
    records that cannot be folded (Z0 probe, S(-1)) in between -- same bits as the per-timestep kernel"""
    rng = np.random.default_rng(max_nstate)
    T1, T2, B1 = rng.uniform(200, 3000, 37), rng.uniform(20, 300, 37), rng.uniform(0.7, 1.2, 37)
    blocks = [
        ([("S", 1), ("E", 5.0, T1, T2, 0), ("T", 120 * B1, 0), ("S", 1), ("E", 5.0, T1, T2, 0), ("ADC",)], 7),
        ([("T", 30 * B1, 40.0), ("E", 3.0, T1, T2, 0.01), ("ADC",), ("E", 7.0, T1, T2, 0.01), ("S", 1)], 4),
        ([("ADC", "Z0")], 1),
        ([("E", 2.0, T1, T2, 0), ("S", 1)], 5),
        ([("S", -1)], 2),
        ([("T", 15.0, 0), ("S", 1), ("ADC",)], 1),
        ([("ADC",)], 3),
        ([("S", 1)], 3),
        ([("S", 1), ("E", 5.0, T1, T2, 0.02), ("T", 150 * B1, 10.0), ("S", 1), ("E", 5.0, T1, T2, 0.02), ("ADC",)], 20),
    ]
    tuples, ops = [("T", 90 * B1, 90)], [epg.T(90 * B1, 90)]
    for blk, rep in blocks:
        blk_ops = sq.to_ops(epg, blk)          # the same operator objects in every repetition
        tuples += blk * rep
        ops += blk_ops * rep
    a = epg.simulate(ops, max_nstate=max_nstate, mode="resident", fuse=fuse)
    b = epg.simulate(ops, max_nstate=max_nstate, mode="stream", fuse=fuse)
    assert sq.same_bits(a, b, x64=resident_k(ops, max_nstate=max_nstate) == 64)
    close(a, epg_c.simulate(tuples, max_nstate=max_nstate))


@pytest.mark.parametrize("phi", [0.0, 90.0, 180.0, 270.0, -90.0, 45.0, 90.0 + 1e-9])
def test_rotation_zero_patterns(phi):
    """rotations about x (phi = 0, 180) and y (phi = +-90) have matrices with exactly-zero components up to the
    rounding of cos / sin at multiples of pi/2 (cos(pi/2) = 6e-17): epgx_plan_create clears those residues and the
    kernels run shorter chains (TX / TY).  Same bits in both modes, oracle within the usual 1e-12; an angle that
    is merely close (90 + 1e-9 deg) keeps the general chains."""
    rng = np.random.default_rng(5)
    T1, T2, B1 = rng.uniform(200, 3000, 29), rng.uniform(20, 300, 29), rng.uniform(0.7, 1.2, 29)
    blk = [("T", 35 * B1, phi), ("E", 3.0, T1, T2, 0), ("ADC",), ("E", 6.0, T1, T2, 0), ("S", 1)]
    tuples = [("T", 180 * B1, phi)] + blk * 12 + [("T", 20.0, phi), ("ADC", "Z0")]
    ops = [epg.T(180 * B1, phi)] + sq.to_ops(epg, blk) * 12 + [epg.T(20.0, phi), epg.Adc("Z0")]
    for max_nstate in (8, 63):
        ref = epg_c.simulate(tuples, max_nstate=max_nstate)
        for fuse in (True, False):
            a = epg.simulate(ops, max_nstate=max_nstate, mode="resident", fuse=fuse)
            b = epg.simulate(ops, max_nstate=max_nstate, mode="stream", fuse=fuse)
            assert np.array_equal(a, b)
            close(a, ref)


@pytest.mark.parametrize("seed", range(24))
def test_random_trains_vs_oracle(seed):
    """random repeated blocks (echo trains built from the same operator objects), from equilibrium, at capacities
    K = 16 ... 128: the state-resident kernels (four voxels per wavefront, folded runs, x / y rotation chains) give
    the bits of the per-timestep kernel and agree with the oracle"""
    rng = np.random.default_rng(7000 + seed)
    grid = tuple(int(x) for x in rng.integers(1, 7, rng.integers(1, 3)))
    cap = [5, 12, 25, 40, 63, 100][int(rng.integers(0, 6))]
    tuples, ops = [("T", 90.0, 90.0)], [epg.T(90.0, 90.0)]
    for blk, rep in sq.random_train_blocks(rng, grid):
        blk_ops = sq.to_ops(epg, blk)
        tuples += blk * rep
        ops += blk_ops * rep
    tuples.append(("ADC",))
    ops.append(epg.ADC)
    ref = onp.simulate(tuples, max_nstate=cap)
    for fuse in (True, False):
        a = np.asarray(epg.simulate(ops, max_nstate=cap, mode="resident", fuse=fuse))
        b = np.asarray(epg.simulate(ops, max_nstate=cap, mode="stream", fuse=fuse))
        assert sq.same_bits(a, b, x64=resident_k(ops, max_nstate=cap) == 64)
        close(a, ref)


@pytest.mark.parametrize("seed", range(12))
def test_random_long_trains_vs_oracle(seed):
    """the same random blocks repeated until the state matrix has grown to 100 ... 1400 orders (no max_nstate, or one above 64):
    the growing kernels at K = 256 ... 1024 (phases of 1, 2, 4 .. orders per lane), the two legs at 2048 -- the bits of the
    per-timestep kernel where a state of that size has an HBM form (K <= 1024), and the C oracle"""
    rng = np.random.default_rng(12000 + seed)
    grid = tuple(int(x) for x in rng.integers(1, 5, rng.integers(1, 3)))
    want = [100, 200, 300, 500, 700, 1100, 1400][int(rng.integers(0, 7))]
    blocks = sq.random_train_blocks(rng, grid, nblocks=5)
    per_round = sum(rep * sum(1 for t in blk if t[0] == "S") for blk, rep in blocks)
    rounds = max(1, -(-want // max(per_round, 1)))
    tuples, ops = [("T", 90.0, 90.0)], [epg.T(90.0, 90.0)]
    for _ in range(min(rounds, 400)):
        for blk, rep in blocks:
            tuples += blk * rep
            ops += sq.to_ops(epg, blk) * rep
    tuples.append(("ADC",))
    ops.append(epg.ADC)
    kw = {} if rng.random() < 0.7 else {"max_nstate": int(rng.integers(70, max(71, want)))}
    enc, _, _ = epg.compile_sequence(ops, options=kw)
    if enc.n_adc > 6000 or enc.peak + 1 > 2048:
        return      # (too many probes / orders for one case)
    ref = epg_c.simulate(tuples, **kw)
    got = np.asarray(epg.simulate(ops, **kw))
    close(got, ref, 1e-11)
    if enc.capacity(resident=True) <= 1024 and enc.capacity(resident=True) >= 128:
        assert np.array_equal(got, np.asarray(epg.simulate(ops, mode="stream", **kw)))


@pytest.mark.parametrize("seed", range(16))
def test_random_repetition_trains_vs_oracle(seed):
    """SSFP / MRF-type trains: repetitions [T, E, ADC, E, S(+1)] with a new flip angle and delay (new tables) in every
    repetition -- the state-resident kernel loops over such record pairs without dispatch (rows_pair_run); trains of
    different shapes (x / y / general rotations, with and without precession), interruptions between them, all capacities"""
    rng = np.random.default_rng(9000 + seed)
    n = int(rng.integers(3, 40))
    T1, T2, B1 = rng.uniform(200, 3000, n), rng.uniform(20, 300, n), rng.uniform(0.7, 1.2, n)
    cap = [6, 12, 25, 63, 100][int(rng.integers(0, 5))]
    tuples = [("T", 180.0 * B1, 90.0), ("E", 15.0, T1, T2, 0)]
    for _ in range(int(rng.integers(1, 4))):
        phi = float(rng.choice([0.0, 90.0, 180.0, -90.0])) if rng.random() < 0.7 else float(rng.uniform(-180, 180))
        g1 = 0 if rng.random() < 0.6 else float(rng.uniform(-0.02, 0.02))
        g2 = 0 if rng.random() < 0.6 else float(rng.uniform(-0.02, 0.02))
        te = float(rng.uniform(2, 5))
        for _ in range(int(rng.integers(1, 30))):
            alpha = float(rng.uniform(5, 70))
            tuples += [("T", alpha * B1 if rng.random() < 0.9 else alpha, phi), ("E", te, T1, T2, g1), ("ADC",),
                       ("E", float(rng.uniform(6, 12)), T1, T2, g2), ("S", 1)]
        tuples += [[("ADC", "Z0")], [("SPOILER",)], [("S", -1)], [("T", 30.0, 10.0), ("ADC",)], []][int(rng.integers(0, 5))]
    tuples.append(("ADC",))
    ops = sq.to_ops(epg, tuples)
    ref = onp.simulate(tuples, max_nstate=cap)
    for fuse in (True, False):
        a = np.asarray(epg.simulate(ops, max_nstate=cap, mode="resident", fuse=fuse))
        b = np.asarray(epg.simulate(ops, max_nstate=cap, mode="stream", fuse=fuse))
        assert sq.same_bits(a, b, x64=resident_k(ops, max_nstate=cap) == 64)
        close(a, ref)


# ------------------------------------------------------------------ run-time fold of E . T . E (library, F_FOLD)
@pytest.mark.parametrize("phi", [90.0, 0.0, 37.0])
@pytest.mark.parametrize("cap", [63, 5, 12, 100, 200])
def test_runtime_fold_of_relaxations_into_rotations(phi, cap):
    """rotations over a B1 axis between relaxations over (T1, T2): the host cannot multiply the tables ahead of time
    (the product would be the whole grid per pulse), so the library folds E_after . T . E_before per voxel at run
    time.  Trains of 1 .. 9 repetitions (odd / even counts, runs shorter than the loop threshold), truncation on the
    leading shift (cap 5, 12), K = 16 / 64 / 128 / 256 kernels; checked against the oracle, per-timestep ==
    state-resident bit for bit, and against the operator-by-operator arithmetic (fuse=False)"""
    rng = np.random.default_rng(int(phi) * 1000 + cap)
    T1 = rng.uniform(300, 3000, 5)[:, None, None]
    T2 = rng.uniform(20, 300, 3)[None, :, None]
    B1 = rng.uniform(0.7, 1.3, 7)[None, None, :]
    for ntr in (1, 2, 3, 4, 5, 8, 9):
        alpha, TR = sq.mrf_trains(ntr, seed=ntr)
        tuples = [("T", 180 * B1, phi), ("E", 20, T1, T2, 0)]
        for a, tr in zip(alpha, TR):
            tuples += [("T", a * B1, phi), ("E", 3.0, T1, T2, 0), ("ADC",), ("E", tr - 3.0, T1, T2, 0), ("S", 1)]
        tuples += [("T", 30 * B1, phi), ("ADC",), ("E", 4.0, T1, T2, 0), ("ADC", "Z0")]     # E_b only / a dangling E
        seq = sq.to_ops(epg, tuples)
        ref = onp.simulate(tuples, max_nstate=cap)
        res = epg.simulate(seq, max_nstate=cap)
        close(res, ref)
        assert np.array_equal(res, epg.simulate(seq, max_nstate=cap, mode="stream"))
        plain = epg.simulate(seq, max_nstate=cap, fuse=False)
        close(plain, ref)
        assert np.array_equal(plain, epg.simulate(seq, max_nstate=cap, mode="stream", fuse=False))


def test_runtime_fold_identical_records_and_mixed_tables():
    """an echo train whose rotation varies along a third axis: every echo is the SAME folded record (a repeat count),
    next to a variable-flip-angle train (a run of folded records with different tables), a spoiler in front of a
    rotation (no fold across it), a density change, precession (no fold) and a negative shift"""
    rng = np.random.default_rng(5)
    T1 = rng.uniform(300, 3000, 4)[:, None, None]
    T2 = rng.uniform(20, 300, 6)[None, :, None]
    B1 = rng.uniform(0.7, 1.3, 5)[None, None, :]
    blk = [("S", 1), ("E", 5, T1, T2, 0), ("T", 120 * B1, 0), ("S", 1), ("E", 5, T1, T2, 0), ("ADC",)]
    vfa = []
    for a in rng.uniform(20, 160, 7):
        vfa += [("S", 1), ("E", 4, T1, T2, 0), ("T", a * B1, 0), ("S", 1), ("E", 4, T1, T2, 0), ("ADC",)]
    tuples = ([("T", 90 * B1, 90)] + blk * 11 + vfa + [("E", 3, T1, T2, 0), ("SPOILER",), ("T", 40 * B1, 90), ("ADC",)]
              + [("E", 3, T1, T2, 0), ("PD", 0.8, False), ("T", 25 * B1, 90), ("E", 2, T1, T2, 0.01), ("ADC",)]
              + [("E", 3, T1, T2, 0), ("S", -1), ("T", 35 * B1, 45), ("E", 6, T1, T2, 0), ("S", 2), ("ADC",), ("ADC", "Z0")])
    seq = sq.to_ops(epg, tuples)
    for cap in (63, 9):
        ref = onp.simulate(tuples, max_nstate=cap)
        res = epg.simulate(seq, max_nstate=cap)
        close(res, ref)
        assert np.array_equal(res, epg.simulate(seq, max_nstate=cap, mode="stream"))


# ------------------------------------------------------------------ one derivative state in the rows layout
@pytest.mark.parametrize("var", ["T1", "T2", "B1"])
@pytest.mark.parametrize("nvox", [1, 6, 777])
def test_single_variable_jacobian_rows_kernel(var, nvox):
    """plans with ONE variable at K = 64 run rows_deriv_kernel (four voxels per wavefront, straight-line record bodies):
    against the oracle, against the same column of the three-variable run (deriv_kernel), and the undifferentiated
    signal bit for bit against the plain operator-by-operator simulation; ragged voxel counts"""
    rng = np.random.default_rng(nvox)
    T1, T2, B1 = rng.uniform(300, 2500, nvox), rng.uniform(20, 300, nvox), rng.uniform(0.7, 1.3, nvox)
    tuples, ops, variables = sq.jac_mse(T1, T2, B1, necho=14)
    one = epg.simulate(ops(epg), probe=epg.Jacobian(["magnitude", var]), max_nstate=63)
    assert one.shape == (14, nvox, 2)
    n = min(nvox, 48)
    ref = onp.simulate_jacobian(sq.jac_mse(T1[:n], T2[:n], B1[:n], necho=14)[0], ["magnitude", var], max_nstate=63)
    close(one[:, :n], ref)
    three = epg.simulate(ops(epg), probe=epg.Jacobian(variables), max_nstate=63)
    close(one[..., 1], three[..., variables.index(var)], tol=1e-11)
    # one variable: the E . T . E runs are fused like those of the plain plan (tables AND their partials generated by the
    # library), so the state column is the plain fused simulation bit for bit; without the fusion, the unfused one
    assert np.array_equal(one[..., 0], epg.simulate(sq.mse_ops(epg, T1, T2, B1, necho=14), max_nstate=63))
    stage = epg.simulate(ops(epg), probe=epg.Jacobian(["magnitude", var]), max_nstate=63, fuse=False)
    close(stage[:, :n], ref)
    close(stage[..., 1], one[..., 1], tol=1e-11)
    assert np.array_equal(stage[..., 0], epg.simulate(sq.mse_ops(epg, T1, T2, B1, necho=14), fuse=False, max_nstate=63))
    # short state matrices (packed kernels: 16 / 32 orders per voxel), one and two variables per plan
    for cap in (9, 20):
        ref_c = onp.simulate_jacobian(sq.jac_mse(T1[:n], T2[:n], B1[:n], necho=14)[0], ["magnitude", var, "T2"], max_nstate=cap)
        for fuse in (True, False):
            got = epg.simulate(ops(epg), probe=epg.Jacobian(["magnitude", var, "T2"]), max_nstate=cap, fuse=fuse)
            close(got[:, :n], ref_c)


@pytest.mark.parametrize("seed", range(24))
def test_random_fused_jacobians_vs_oracle(seed):
    """differentiated trains whose E . T . E runs the planner fuses (generated partials, nested before / after fusions,
    x / y / general rotation axes, per-voxel coefficients): one variable at up to 64 orders (rows_deriv_kernel), one or two
    below 32 (packed_deriv_kernel) -- against the oracle's recurrence and against the three-stage plan"""
    from epgpy_amd import functions
    rng = np.random.default_rng(31000 + seed)
    grid = tuple(int(x) for x in rng.integers(1, 6, rng.integers(1, 3)))
    nvars = 1 + seed % 2
    cap = [None, 63, 20, 9][int(rng.integers(0, 4))] if nvars == 1 else [20, 9, 14][int(rng.integers(0, 3))]
    tuples, ops, variables = sq.random_fusable_jacobian_sequence(rng, grid, nvars, nblocks=int(rng.integers(3, 12)))
    kw = {"max_nstate": cap} if cap else {}
    ref = onp.simulate_jacobian(tuples, variables, shape=grid, **kw)
    fused = epg.simulate(ops(epg), probe=epg.Jacobian(variables), shape=grid, **kw)
    stage = epg.simulate(ops(epg), probe=epg.Jacobian(variables), shape=grid, fuse=False, **kw)
    close(stage, ref)
    close(fused, ref, tol=1e-11)
    enc, _, _ = functions.compile_sequence(ops(epg), [epg.Jacobian(variables)], options=kw, variables=variables[1:], shape=grid)
    if seed in (0, 1, 2, 3):        # (these seeds do fuse: the test would be empty if the planner stopped doing it)
        assert enc.fuse_partials


@pytest.mark.parametrize("form", ["mse", "gre", "shift_after", "no_shift"])
@pytest.mark.parametrize("phi", [0.0, 90.0, 37.0])
def test_echo_trains_on_rotating_slots(form, phi, monkeypatch, capfd):
    """drun_kernel (epgx_drun_kernels.hip.h): runs of fused-echo records with 1 - 3 derivative states -- every run shape
    (leading / trailing shift or none: four sequence forms; rotation about x, about y, about a general axis), trains that are
    and are not a multiple of four records long, trains that repeat ONE record (lines loaded once) and trains with a new
    relaxation table per echo; against the oracle's recurrence, against the three-stage plan, and the state column bit for bit
    against the plain (undifferentiated) fused simulation"""
    monkeypatch.setenv("EPGX_TRACE", "1")
    rng = np.random.default_rng(int(phi) + len(form))
    nvox = 37
    T1, T2, B1 = rng.uniform(300, 2500, nvox), rng.uniform(20, 300, nvox), rng.uniform(0.7, 1.3, nvox)
    rl_o1 = {"T1": {"T1": 1}, "T2": {"T2": 1}}
    for necho, varying in ((4, False), (9, False), (13, True), (6, True), (33, False), (34, True)):    # (the last two: 64 orders in every form)
        taus = [5.0 + (0.37 * n if varying else 0.0) for n in range(necho)]
        alpha = 120.0 if form == "mse" else 35.0
        rf_o1 = {"B1": {"alpha": alpha}}
        tuples, plain = [], []

        def rot(out, diff):
            out.append(("T", alpha * B1, phi, {"order1": rf_o1}) if diff else ("T", alpha * B1, phi))

        def rlx(out, tau, diff):
            out.append(("E", tau, T1, T2, 0, {"order1": rl_o1}) if diff else ("E", tau, T1, T2, 0))

        for out, diff in ((tuples, True), (plain, False)):
            out.append(("T", 90 * B1, 90, {"order1": {"B1": {"alpha": 90}}}) if diff else ("T", 90 * B1, 90))
            for tau in taus:
                if form == "mse":              # S E T S E ADC  ->  [S(+1)  E.T.E  S(+1)  ADC]
                    out.append(("S", 1)); rlx(out, tau, diff); rot(out, diff); out.append(("S", 1)); rlx(out, tau, diff); out.append(("ADC",))
                elif form == "gre":            # T E ADC E S  ->  [S(+1)  E.T.E  ADC] from the second repetition on
                    rot(out, diff); rlx(out, 0.4 * tau, diff); out.append(("ADC",)); rlx(out, tau, diff); out.append(("S", 1))
                elif form == "shift_after":    # E T E S ADC  ->  [E.T.E  S(+1)  ADC]
                    rlx(out, tau, diff); rot(out, diff); rlx(out, 0.5 * tau, diff); out.append(("S", 1)); out.append(("ADC",))
                else:                          # E T E ADC  ->  [E.T.E  ADC]
                    rlx(out, tau, diff); rot(out, diff); rlx(out, 0.5 * tau, diff); out.append(("ADC",))

        def ops_of(tups):
            ops = []
            for t in tups:
                if t[0] == "T":
                    ops.append(epg.T(t[1], t[2], order1=t[3]["order1"]) if len(t) > 3 else epg.T(t[1], t[2]))
                elif t[0] == "E":
                    ops.append(epg.E(t[1], t[2], t[3], t[4], order1=["T1", "T2"]) if len(t) > 5 else epg.E(t[1], t[2], t[3], t[4]))
                elif t[0] == "S":
                    ops.append(epg.S(t[1]))
                else:
                    ops.append(epg.ADC)
            return ops

        state = {fused: epg.simulate(ops_of(plain), max_nstate=63, fuse=fused) for fused in (True, False)}
        for variables in (["magnitude", "T2"], ["magnitude", "B1", "T1"], ["magnitude", "T1", "T2", "B1"]):
            ref = onp.simulate_jacobian(tuples, variables, max_nstate=63)
            capfd.readouterr()
            got = epg.simulate(ops_of(tuples), probe=epg.Jacobian(variables), max_nstate=63)
            if necho >= 33 and form != "no_shift":   # 64 orders: runs of fused echoes, their relaxation partials in logarithmic form
                assert "fused echoes with logarithmic relaxation partials" in capfd.readouterr().err
            close(got, ref, tol=1e-11)
            # (three variables are fused in the 64-order class only: below it the plan keeps its three stages)
            fused = functions._fusion_pays(ops_of(tuples), variables[1:], 0, {"max_nstate": 63})
            if fused:
                assert sq.same_bits(got[..., 0], state[True], x64=True), (form, phi, necho, variables)
            else:       # (the library folds such a train at run time -- packed_dfold_kernel --: the plain fold's arithmetic, to rounding)
                close(got[..., 0], state[False], tol=1e-13)
            stage = epg.simulate(ops_of(tuples), probe=epg.Jacobian(variables), max_nstate=63, fuse=False)
            close(stage, got, tol=1e-11)


@pytest.mark.parametrize("form", ["mrf", "shift_before_adc", "no_shift", "ssfp_x", "general_axis"])
def test_repetition_trains_folded_at_run_time_with_derivatives(form, monkeypatch, capfd):
    """drun_kernel's folded runs (DRUN_FOLD): repetitions  [T(a_n B1)  E(TE)  ADC  E(TR_n - TE)  S]  over a (T1, T2) x B1 grid --
    a rotation over one index space between relaxations over another, new tables every repetition -- become one stage
    E_a . T . E_b per repetition whose line the wavefront computes; relaxation partials enter through their logarithmic
    form (logtab_kernel), the rotation's partial is folded like the rotation.  Every run shape (leading / trailing / no
    shift; rotation about y, about x, about a general axis), 1 - 3 variables in every combination of kinds, trains that are
    and are not whole fours; against the oracle's recurrence and against the unfolded three-stage plan (fuse=False)."""
    monkeypatch.setattr(functions, "FUSED_TABLE_BUDGET", 0.0)   # (as on a large grid: no host-side E . T . E tables)
    monkeypatch.setenv("EPGX_TRACE", "1")                       # (the library says which kernel a derivative launch takes)
    rng = np.random.default_rng(len(form))
    n1, n2, n3 = 5, 4, 3
    T1 = rng.uniform(300, 2500, n1)[:, None, None]
    T2 = rng.uniform(20, 300, n2)[None, :, None]
    B1 = rng.uniform(0.7, 1.3, n3)[None, None, :]
    grid = (n1, n2, n3)
    flat = [np.broadcast_to(x, grid).reshape(-1) for x in (T1, T2, B1)]
    phi = {"ssfp_x": 0.0, "general_axis": 35.0}.get(form, 90.0)
    rl_o1 = {"T1": {"T1": 1}, "T2": {"T2": 1}}
    for ntr in (9, 14):
        alpha, TR = rng.uniform(10, 60, ntr), rng.uniform(11, 16, ntr)

        def build(t1, t2, b1, as_ops):
            T = (lambda a, ph, o1: epg.T(a, ph, order1=o1)) if as_ops else (lambda a, ph, o1: ("T", a, ph, {"order1": o1}))
            E = (lambda tau: epg.E(tau, t1, t2, order1=["T1", "T2"])) if as_ops else (lambda tau: ("E", tau, t1, t2, 0, {"order1": rl_o1}))
            S, ADC = (epg.S(1), epg.ADC) if as_ops else (("S", 1), ("ADC",))
            seq = [T(180 * b1, 90, {"B1": {"alpha": 180}}), E(20.0)]
            e_te = E(3.0)                                   # (one operator object for every repetition's first relaxation)
            for a, tr in zip(alpha, TR):
                rot = T(a * b1, phi, {"B1": {"alpha": float(a)}})
                if form == "shift_before_adc":
                    seq += [rot, e_te, S, ADC, E(tr - 3.0)]
                elif form == "no_shift":
                    seq += [rot, e_te, ADC, E(tr - 3.0)]
                else:
                    seq += [rot, e_te, ADC, E(tr - 3.0), S]
            return seq

        ops, tuples = build(T1, T2, B1, True), build(*flat, False)
        for variables in (["magnitude", "T2"], ["magnitude", "B1"], ["magnitude", "T1", "T2"], ["magnitude", "B1", "T1"],
                          ["magnitude", "T1", "T2", "B1"]):
            ref = onp.simulate_jacobian(tuples, variables, max_nstate=63).reshape((ntr,) + grid + (len(variables),))
            capfd.readouterr()
            got = epg.simulate(ops, probe=epg.Jacobian(variables), max_nstate=63, packed=False)      # K = 64: folded runs
            assert "folded at run time" in capfd.readouterr().err
            close(got, ref, tol=1e-11)
            staged = epg.simulate(ops, probe=epg.Jacobian(variables), max_nstate=63, packed=False, fuse=False)   # (EPGX_PLAN_NO_FOLD)
            assert "folded at run time" not in capfd.readouterr().err
            close(staged, ref, tol=1e-11)
            assert not np.array_equal(got[..., 1:], staged[..., 1:])        # (two different arithmetic paths did run)
            # 16 / 32 orders per voxel (four / two voxels per wavefront): the same fold in packed_dfold_kernel, truncation included
            for cap in (10, 25):
                refc = onp.simulate_jacobian(tuples, variables, max_nstate=cap).reshape((ntr,) + grid + (len(variables),))
                capfd.readouterr()
                gotc = epg.simulate(ops, probe=epg.Jacobian(variables), max_nstate=cap)
                assert "packed_dfold_kernel" in capfd.readouterr().err
                close(gotc, refc, tol=1e-11)


@pytest.mark.parametrize("seed", range(10))
def test_random_repetition_trains_with_derivatives(seed, monkeypatch):
    """random SSFP / MRF-type trains with first-order derivatives at 64 orders: trains of different shapes (x / y / general
    rotations; the shift in front of or behind the ADC, or none; with and without precession -- the latter cannot fold),
    truncation below the number of shifts, interruptions between the trains (Z0 probes, spoilers, negative shifts, lone
    pulses), random variable sets; the host's fold pass has to cut runs, emit the leftovers unfolded and keep both in step"""
    monkeypatch.setattr(functions, "FUSED_TABLE_BUDGET", 0.0)
    rng = np.random.default_rng(31000 + seed)
    n = int(rng.integers(3, 30))
    T1, T2, B1 = rng.uniform(200, 3000, n), rng.uniform(20, 300, n), rng.uniform(0.7, 1.2, n)
    cap = [20, 40, 63][int(rng.integers(0, 3))]
    rl_o1 = {"T1": {"T1": 1}, "T2": {"T2": 1}}

    def Tt(a, phi, with_b1):
        return ("T", a * B1, phi, {"order1": {"B1": {"alpha": float(a)}}}) if with_b1 else ("T", a * B1, phi)

    def Et(tau, g):
        return ("E", tau, T1, T2, g, {"order1": rl_o1})

    tuples = [Tt(180.0, 90.0, True), Et(15.0, 0)]
    for _ in range(int(rng.integers(1, 4))):
        phi = float(rng.choice([0.0, 90.0, 180.0, -90.0])) if rng.random() < 0.7 else float(rng.uniform(-180, 180))
        g1 = 0 if rng.random() < 0.75 else float(rng.uniform(-0.02, 0.02))
        g2 = 0 if rng.random() < 0.75 else float(rng.uniform(-0.02, 0.02))
        te = float(rng.uniform(2, 5))
        form = int(rng.integers(0, 3))
        same_te = Et(te, g1)
        for _ in range(int(rng.integers(1, 24))):
            alpha = float(rng.uniform(5, 70))
            rot, e_b = Tt(alpha, phi, rng.random() < 0.9), Et(float(rng.uniform(6, 12)), g2)
            if form == 0:
                tuples += [rot, same_te, ("ADC",), e_b, ("S", 1)]
            elif form == 1:
                tuples += [rot, same_te, ("S", 1), ("ADC",), e_b]
            else:
                tuples += [rot, same_te, ("ADC",), e_b]
        tuples += [[("ADC", "Z0")], [("SPOILER",)], [("S", -1)], [("T", 30.0, 10.0), ("ADC",)], []][int(rng.integers(0, 5))]
    tuples.append(("ADC",))

    def to_ops(tups):
        ops = []
        for t in tups:
            if t[0] == "T":
                ops.append(epg.T(t[1], t[2], order1=t[3]["order1"]) if len(t) > 3 else epg.T(t[1], t[2]))
            elif t[0] == "E":
                ops.append(epg.E(t[1], t[2], t[3], t[4], order1=["T1", "T2"]))
            elif t[0] == "S":
                ops.append(epg.S(t[1]))
            elif t[0] == "SPOILER":
                ops.append(epg.SPOILER)
            else:
                ops.append(epg.ADC if len(t) == 1 else epg.Adc(t[1]))
        return ops

    ops = to_ops(tuples)
    pool = [["magnitude", "T2"], ["magnitude", "T1", "T2"], ["magnitude", "B1", "T2"], ["magnitude", "T1", "T2", "B1"], ["magnitude", "B1"]]
    for variables in (pool[int(rng.integers(0, 5))], pool[int(rng.integers(0, 5))]):
        ref = onp.simulate_jacobian(tuples, variables, max_nstate=cap)
        got = epg.simulate(ops, probe=epg.Jacobian(variables), max_nstate=cap, packed=False)
        close(got, ref, tol=1e-11)
        small = [8, 14, 27][int(rng.integers(0, 3))]          # 16 / 32 orders per voxel: the packed kernels
        close(epg.simulate(ops, probe=epg.Jacobian(variables), max_nstate=small), onp.simulate_jacobian(tuples, variables, max_nstate=small),
              tol=1e-11)


@pytest.mark.parametrize("phase_step", [0.0, 58.5])
def test_spoiled_repetition_trains_with_derivatives(phase_step, monkeypatch, capfd):
    """RF-spoiled gradient echo with a perfect spoiler per repetition and first-order derivatives (16 orders per voxel): the
    spoiler joins the run-time fold of packed_dfold_kernel -- for the STATE the transverse columns of the relaxation in front of
    the rotation count as zero; the derivative states follow the reference (its SPOILER leaves them alone) unless
    exact_partials=True spoils them too.  Both against the oracle's recurrence"""
    monkeypatch.setattr(functions, "FUSED_TABLE_BUDGET", 0.0)
    monkeypatch.setenv("EPGX_TRACE", "1")
    rng = np.random.default_rng(int(phase_step))
    n1, n2, n3 = 4, 5, 3
    T1 = rng.uniform(300, 2500, n1)[:, None, None]
    T2 = rng.uniform(20, 300, n2)[None, :, None]
    B1 = rng.uniform(0.7, 1.3, n3)[None, None, :]
    grid = (n1, n2, n3)
    flat = [np.broadcast_to(x, grid).reshape(-1) for x in (T1, T2, B1)]
    rl_o1 = {"T1": {"T1": 1}, "T2": {"T2": 1}}
    ntr = 11

    def build(t1, t2, b1, as_ops):
        seq = []
        e1 = epg.E(3.0, t1, t2, order1=["T1", "T2"]) if as_ops else ("E", 3.0, t1, t2, 0, {"order1": rl_o1})
        e2 = epg.E(7.0, t1, t2, order1=["T1", "T2"]) if as_ops else ("E", 7.0, t1, t2, 0, {"order1": rl_o1})
        for n in range(ntr):
            phi = float(phase_step * n * n % 360)
            seq.append(epg.T(14.8 * b1, phi, order1={"B1": {"alpha": 14.8}}) if as_ops else ("T", 14.8 * b1, phi, {"order1": {"B1": {"alpha": 14.8}}}))
            seq += [e1, epg.ADC if as_ops else ("ADC",), e2, epg.SPOILER if as_ops else ("SPOILER",)]
        return seq

    ops, tuples = build(T1, T2, B1, True), build(*flat, False)
    for variables in (["magnitude", "T1"], ["magnitude", "T1", "T2", "B1"]):
        for exact in (False, True):
            ref = onp.simulate_jacobian(tuples, variables, through_plain=exact).reshape((ntr,) + grid + (len(variables),))
            capfd.readouterr()
            got = epg.simulate(ops, probe=epg.Jacobian(variables), exact_partials=exact)
            assert "packed_dfold_kernel" in capfd.readouterr().err
            close(got, ref, tol=1e-11)
            close(epg.simulate(ops, probe=epg.Jacobian(variables), exact_partials=exact, fuse=False), ref, tol=1e-11)


def test_generated_partials_abi_checks():
    """epgx_fuse_partial (include/epgx.h): what epgx_plan_create refuses, and that a T0 operator may only point at a
    generated partial some entry writes"""
    from epgpy_amd import functions
    T2 = np.linspace(40.0, 90.0, 6)
    seq = [epg.T(30.0, 0, order1={"fa": "alpha"}), epg.E(5, 1000, T2, order1=["T2"]), epg.ADC, epg.S(1)] * 3
    ctx = _lib.get_context()

    def plan_with(change=None, drop=False, variables=("T2", "fa")):
        enc, _, _ = functions.compile_sequence(seq, [epg.Jacobian(list(variables))], variables=list(variables))
        ops, grid, spaces, coef, dops = enc.arrays()
        fpart = enc.fuse_partial_array()
        assert len(fpart) == 2 and (ops["opcode"] == _lib.OP_T0).sum() == 3
        if change:
            change(fpart, coef.size)
        return _lib.DevicePlan(ctx, ops, grid, spaces, coef, enc.n_adc, dops=dops, n_vars=len(variables), fuse=enc.fuse_array(),
                               n_coef_generated=enc.generated_size, fuse_partial=None if drop else fpart)

    plan_with()                                           # (the recipe as the planner writes it is accepted)

    def both_missing(fp, n):
        fp["dsrc_off"], fp["de_off"] = -1, -1

    def dst_in_host_part(fp, n):
        fp["dst_off"][0] = 0

    def bad_space(fp, n):
        fp["e_space"][0] = 7

    def bad_ncoef(fp, n):
        fp["dsrc_ncoef"] = 11
        fp["dsrc_off"] = 0

    def partial_beyond_host(fp, n):
        fp["de_off"][fp["de_off"] >= 0] = n - 2

    def rotation_source_not_generated(fp, n):
        fp["src_off"][0] = n + 1
        fp["src_ncoef"][0] = 12

    for change, text in ((both_missing, "neither the rotation nor the relaxation"), (dst_in_host_part, "destination outside"),
                         (bad_space, "index space"), (bad_ncoef, "10 or 14"), (partial_beyond_host, "E partial outside"),
                         (rotation_source_not_generated, "generated rotation source")):
        with pytest.raises(_lib.EpgxError, match=text):
            plan_with(change)
    with pytest.raises(_lib.EpgxError, match="no entry of `fuse_partial` writes there"):
        plan_with(drop=True)


@pytest.mark.parametrize("seed", range(12))
def test_random_single_variable_jacobians(seed):
    """random differentiated sequences with ONE shared variable (rows_deriv_kernel when the plan qualifies: K = 64,
    shifts by +-1; otherwise deriv_kernel): complex partials, truncation, spoilers / resets / density changes with
    and without `exact_partials`"""
    rng = np.random.default_rng(31000 + seed)
    grid = tuple(int(x) for x in rng.integers(1, 6, rng.integers(1, 3)))
    cap = [None, 3, 10, 40][int(rng.integers(0, 4))]
    tuples, ops, variables = sq.random_jacobian_sequence(rng, grid, nops=int(rng.integers(8, 40)))
    opts = {"max_nstate": cap} if cap else {}
    for var in variables[1:3]:
        ref = onp.simulate_jacobian(tuples, ["magnitude", var], shape=grid, max_nstate=cap)
        got = np.asarray(epg.simulate(ops(epg), probe=epg.Jacobian(["magnitude", var]), **opts))
        own = got.shape[1:-1]      # (the operators may not span the trailing axes of `grid`: the oracle was told the shape)
        got = np.broadcast_to(got.reshape(got.shape[:1] + own + (1,) * (len(grid) - len(own)) + got.shape[-1:]), ref.shape)
        close(got, ref, tol=1e-11)


def test_single_variable_jacobian_across_plain_operators():
    """SPOILER / RESET / PD inside a one-variable plan (the generic record of rows_deriv_kernel), with the reference's
    convention and with `exact_partials`; F0 and Z0 Jacobians"""
    T2 = np.array([40.0, 90.0, 150.0])
    tuples, ops, _ = sq.jac_plain_ops(T2)
    for var in ("alpha", "T2"):
        for exact in (False, True):
            for what in ("F0", "Z0"):
                ref = onp.simulate_jacobian(tuples, ["magnitude", var], probe=what, through_plain=exact, max_nstate=63)
                got = epg.simulate(ops(epg), probe=epg.Jacobian(["magnitude", var], probe=what), exact_partials=exact, max_nstate=63)
                close(got, ref, tol=1e-11)


@pytest.mark.parametrize("cap", [63, 7])
def test_spoiled_trains_fold_the_spoiler(cap):
    """RF-spoiled gradient echo with a perfect spoiler per repetition: `T(a, phi_n) E(TE) ADC E(TR - TE) SPOILER`.  The
    library folds the spoiler and both relaxations into the next rotation (F_FOLD_SPOIL: the F columns of the folded
    matrix vanish), so a repetition is one straight-line record; also with a shift behind the spoiler, a spoiler without
    relaxations around it, and phases 0 / 90 / general.  Oracle parity, per-timestep == resident, fuse=False"""
    rng = np.random.default_rng(cap)
    T1 = rng.uniform(300, 3000, 4)[:, None, None]
    T2 = rng.uniform(20, 300, 5)[None, :, None]
    B1 = rng.uniform(0.7, 1.3, 3)[None, None, :]
    for phases in ([0.0] * 9, [90.0] * 6, list(58.5 * np.arange(11) ** 2 % 360)):
        tuples = []
        for n, ph in enumerate(phases):
            tuples += [("T", 14.8 * B1, ph), ("E", 3.0, T1, T2, 0), ("ADC",), ("E", 7.0 + 0.1 * n, T1, T2, 0), ("SPOILER",)]
        tuples += [("T", 30 * B1, 0), ("ADC",), ("SPOILER",), ("T", 45 * B1, 90), ("ADC",), ("ADC", "Z0"),
                   ("E", 5.0, T1, T2, 0), ("S", 1), ("SPOILER",), ("T", 20 * B1, 0), ("S", 1), ("ADC",), ("ADC", "Z0")]
        seq = sq.to_ops(epg, tuples)
        ref = onp.simulate(tuples, max_nstate=cap)
        res = epg.simulate(seq, max_nstate=cap)
        close(res, ref)
        assert np.array_equal(res, epg.simulate(seq, max_nstate=cap, mode="stream"))
        close(epg.simulate(seq, max_nstate=cap, fuse=False), ref)


# ------------------------------------------------------------------ the state matrix grows: phases of 16 / 32 / 64 orders
def test_growing_state_matrix_phases(tmp_path, capfd):
    """State-resident launches from equilibrium at 64 orders walk their records in phases of 1 / 2 / 4 orders per lane while
    the populated orders fit 16 / 32 (rows_grow_kernel; the reference grows its state matrix the same way: functions.py:135,
    shift.py:86).  (a) against the oracle; (b) bit for bit the results of rows_kernel<., 4, .> (a child process with
    EPGX_GROW=0 runs the same sequences); (c) which launches take the kernel: the 20-echo train does, the 1000-TR train of
    config 3 (30 of 1000 repetitions below 64 orders) does not"""
    import subprocess
    import sys

    from tests import grow_cases

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "plain.npz")
    env = dict(os.environ, EPGX_GROW="0")
    subprocess.run([sys.executable, os.path.join(root, "tests", "grow_cases.py"), out], check=True, env=env, cwd=root, timeout=600)
    plain = np.load(out)
    for name, (seq, kw) in grow_cases.cases(epg).items():
        got = epg.simulate(seq, **kw)
        assert np.array_equal(got, plain[name]), name
        close(got, epg.simulate(seq, mode="stream", **kw), tol=1e-12)
    # (d) "T | S T S ADC ...": the shift behind the excitation opens the first echo's record (EPGX_LEAD_FORWARD=0: it closes the
    # excitation's, the packing of earlier rounds) -- the same chains in the same order either way
    out = str(tmp_path / "lead.npz")
    subprocess.run([sys.executable, os.path.join(root, "tests", "grow_cases.py"), out], check=True,
                   env=dict(os.environ, EPGX_LEAD_FORWARD="0"), cwd=root, timeout=600)
    lead = np.load(out)
    for name, (seq, kw) in grow_cases.cases(epg).items():
        assert np.array_equal(epg.simulate(seq, **kw), lead[name]), name
    T1, T2 = np.linspace(200, 3000, 16)[:, None], np.linspace(20, 300, 16)[None, :]
    close(epg.simulate(sq.mse_ops(epg, T1, T2), max_nstate=63), onp.simulate(sq.mse_tuples(T1, T2), max_nstate=63))
    os.environ["EPGX_TRACE"] = "1"
    try:
        capfd.readouterr()
        epg.simulate(sq.mse_ops(epg, T1[:15], T2), max_nstate=63)      # (a grid no plan exists for yet: the list is printed when it is cut)
        seen = capfd.readouterr().err
        assert "rows_grow_kernel" in seen and "at 16 orders per voxel" in seen, seen
        # the 20 echoes are ONE shape behind the excitation: 7 + 8 + 5 repetitions at 16 / 32 / 64 orders, four records in all
        listed = [ln for ln in seen.splitlines() if "grow list" in ln]
        assert len(listed) == 4 and [ln.split(" x ")[1].split()[0] for ln in listed] == ["1", "7", "8", "5"], listed
        alpha, TR = sq.mrf_trains(400)
        B1 = np.linspace(0.8, 1.2, 4)[None, None, :]
        epg.simulate(sq.mrf_ops(epg, T1[:4, :, None], T2[:, :4, None], B1, alpha, TR), max_nstate=63)
        assert "rows_grow_kernel" not in capfd.readouterr().err
    finally:
        del os.environ["EPGX_TRACE"]


# ------------------------------------------------------------------ a general equilibrium state matrix
def test_g15_general_equilibrium(golden):
    """StateMatrix(equilibrium=<state matrix with transverse and k != 0 coefficients>) (statematrix.py:56-59): golden G15 from
    the reference -- states and equilibrium after every operator of a sequence with relaxation + precession over a (T1, T2)
    grid, shifts both ways and a RESET, and the F0 / Z0 records of simulate(init=...).  The equilibrium lives in a second
    device-resident state matrix; `arr0 * equilibrium` is one more launch of the scalar stage plus epgx_state_axpy"""
    g = golden("g15_equilibrium")
    eq, T1, T2 = g["equilibrium"], g["T1"], g["T2"]
    seq = [epg.T(70, 25), epg.E(8.0, T1, T2, 0.01), epg.S(1), epg.T(120, 0), epg.E(5.0, T1, T2), epg.S(1), epg.E(12.0, T1, T2, -0.02),
           epg.S(-1), epg.T(40, 90), epg.RESET, epg.T(60, 10), epg.E(9.0, T1, T2), epg.S(1), epg.E(3.0, T1, T2)]
    sm = epg.StateMatrix(equilibrium=eq, shape=(3, 2))
    close(sm.states, g["init_states"])
    close(sm.density, g["init_density"])
    for i, op in enumerate(seq):
        sm = op(sm)
        close(sm.states, g[f"states_{i}"])
        close(sm.equilibrium, g[f"equilibrium_{i}"])
    again = sm.copy()
    close(again.equilibrium, sm.equilibrium)
    train = [epg.T(70, 25)] + [epg.E(8.0, T1, T2, 0.01), epg.S(1), epg.T(120, 0), epg.E(5.0, T1, T2), epg.ADC] * 4
    init = epg.StateMatrix(equilibrium=eq, shape=(3, 2))
    f0, z0 = epg.simulate(train, init=init, probe=["F0", "Z0"])
    close(f0, g["sim_F0"])
    close(z0, g["sim_Z0"])
    close(init.states, g["init_states"])                        # simulate never mutates the caller's init (functions.py:149)
    with pytest.raises(NotImplementedError):
        epg.simulate(train, init=init, mode="resident")
    with pytest.raises(NotImplementedError):
        epg.PD(2.0)(epg.StateMatrix(equilibrium=eq))
    # the plain form is still the kernels' own: a [0, 0, density] equilibrium creates no second matrix
    assert epg.StateMatrix(equilibrium=[0, 0, 0.7])._eq is None
    # random sequences (rotations, relaxation with precession, shifts by +-1..3, probes with phases, spoilers, resets) on other equilibria
    for i in range(4):
        rng = np.random.default_rng(1500 + i)
        tuples = [t for t in sq.random_sequence(rng, (4, 3), nops=30) if t[0] != "PD"]
        seq_i = sq.to_ops(epg, tuples)
        sm = epg.StateMatrix(equilibrium=g[f"rand{i}_equilibrium"], shape=(4, 3))
        for op in seq_i:
            sm = op(sm)
        close(sm.states, g[f"rand{i}_states"])
        f0, z0 = epg.simulate(seq_i, init=epg.StateMatrix(equilibrium=g[f"rand{i}_equilibrium"], shape=(4, 3)), probe=["F0", "Z0"])
        close(f0, g[f"rand{i}_F0"])
        close(z0, g[f"rand{i}_Z0"])


def test_growing_long_state_matrices(tmp_path, capfd):
    """Launches from equilibrium at 256 .. 1024 orders per voxel walk their records in phases of 1, 2, 4, 8 (, 16) orders per lane
    while the populated orders fit 64, 128, 256, 512 (run_contig_grow_kernel; the reference grows its state matrix the same way:
    functions.py:135, shift.py:86,98); at 2048 one wavefront per voxel runs the records while at most 512 orders hold anything (run_kernel<8, ..>), then
    four wavefronts per voxel take its state over (run_split_kernel<4, .., true>).  (a) bit for bit the results of the fixed-capacity kernels (a
    child process with EPGX_CGROW=0 EPGX_SPLIT_GROW=0 runs the same sequences) and of the same phases taken at 128 orders as
    well (EPGX_CGROW=2); (b) against the per-timestep kernel and the oracle; (c) which launches take the kernels"""
    import subprocess
    import sys

    from tests import cgrow_cases

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mine = {name: epg.simulate(seq, **kw) for name, (seq, kw) in cgrow_cases.cases(epg).items()}
    for setting in ("0", "2"):
        out = str(tmp_path / f"cgrow{setting}.npz")
        subprocess.run([sys.executable, os.path.join(root, "tests", "cgrow_cases.py"), out], check=True,
                       env=dict(os.environ, EPGX_CGROW=setting, EPGX_SPLIT_GROW="0" if setting == "0" else "1"), cwd=root, timeout=900)
        other = np.load(out)
        for name, got in mine.items():
            assert np.array_equal(got, other[name]), (setting, name)
    # the two legs at 2048 orders take the voxels in slabs (8 GiB of scratch at most): forced to slabs of 4 voxels, the same bits
    out = str(tmp_path / "slabs.npz")
    subprocess.run([sys.executable, os.path.join(root, "tests", "cgrow_cases.py"), out, "mse_900", "mse_900_cap1100", "mrf_1100"], check=True,
                   env=dict(os.environ, EPGX_SLAB_VOXELS="4"), cwd=root, timeout=900)
    other = np.load(out)
    for name in ("mse_900", "mse_900_cap1100", "mrf_1100"):
        assert np.array_equal(mine[name], other[name]), ("slabs", name)
    alpha_l, TR_l = sq.mrf_trains(1100)
    B1 = np.linspace(0.8, 1.2, 5)
    T1s, T2s = np.linspace(200, 3000, 9)[:3], np.linspace(20, 300, 7)[:3]
    close(mine["mrf_1100"], epg_c.simulate(sq.mrf_tuples(T1s[:, None, None], T2s[None, :, None], B1[None, None, :], alpha_l, TR_l)), 1e-11)
    for name in ("mse_65", "mse_150", "mrf_200_cap130", "train_0", "train_4"):
        seq, kw = cgrow_cases.cases(epg)[name]
        close(mine[name], epg.simulate(seq, mode="stream", **kw), tol=1e-12)
    T1, T2 = np.linspace(200, 3000, 9)[:, None], np.linspace(20, 300, 7)[None, :]
    close(mine["mse_150"], epg_c.simulate(sq.mse_tuples(T1, T2, necho=150)))
    close(mine["mse_200_cap150"], epg_c.simulate(sq.mse_tuples(T1, T2, necho=200), max_nstate=150))
    close(mine["mse_900_cap1100"], epg_c.simulate(sq.mse_tuples(T1[:2], T2[:, :5], necho=900), max_nstate=1100), 1e-11)
    os.environ["EPGX_TRACE"] = "1"
    try:
        capfd.readouterr()
        epg.simulate(sq.mse_ops(epg, T1[:8], T2, necho=100))
        seen = capfd.readouterr().err
        assert "run_contig_grow_kernel<4, " in seen and "[0, 32) at 64 orders per voxel, [32, 64) at 128" in seen, seen
        epg.simulate(sq.mse_ops(epg, T1[:2], T2, necho=800))
        seen = capfd.readouterr().err
        assert "run_kernel<8, 1, false> + run_split_kernel<4, 1, true>" in seen, seen
        assert "[0, 256) on one wavefront per voxel, the rest on four (parts 2 and 3 join at 512 and 768)" in seen, seen
        # 128 orders: the phases take the launch while the train mostly runs below 64 orders (else four voxels per wavefront, 8 orders per lane)
        epg.simulate(sq.mse_ops(epg, T1[:8], T2, necho=40))
        seen = capfd.readouterr().err
        assert "run_contig_grow_kernel<2, " in seen, seen
        epg.simulate(sq.mse_ops(epg, T1[:8], T2, necho=62))
        seen = capfd.readouterr().err
        assert "rows_kernel<1, 8, false>" in seen and "run_contig_grow_kernel" not in seen, seen
        # a train that spends its time at the capacity keeps the fixed-capacity kernel
        epg.simulate(sq.mse_ops(epg, T1[:8], T2, necho=1500), max_nstate=140)
        seen = capfd.readouterr().err
        assert "run_contig_grow_kernel" not in seen and "run_contig_kernel<4, " in seen, seen
    finally:
        del os.environ["EPGX_TRACE"]
