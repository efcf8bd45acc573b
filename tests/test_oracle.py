"""CPU: pin both oracles (NumPy restatement, C restatement) against the golden vectors that
tests/golden/make_golden.py produced from the reference itself."""
import numpy as np
import pytest

from oracle import epg_numpy as onp, epg_c
from tests import sequences as sq

TOL = 1e-13


def close(a, b, tol=TOL):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape
    scale = max(1.0, float(np.max(np.abs(b)))) if b.size else 1.0
    assert float(np.max(np.abs(a - b))) <= tol * scale if b.size else True


@pytest.mark.parametrize("sim", [onp.simulate, epg_c.simulate], ids=["numpy", "c"])
def test_g1_readme(golden, sim):
    g = golden("g1_readme_mse")
    seq = sq.mse_tuples(float(g["T1"]), g["T2"], FA=float(g["FA"]), ESP=float(g["ESP"]), necho=int(g["necho"]))
    sig, st = sim(seq, return_states=True)
    close(sig, g["signal"])
    close(st, g["states"])
    # spot values quoted in SURVEY.md section 8c
    assert np.allclose(sig[0].real, [0.537398482930342, 0.5841005873035536, 0.6140480648084863], rtol=0, atol=1e-15)
    assert np.allclose(sig[19].real, [0.00333804473123021, 0.01090963468479831, 0.02405611783842116], rtol=0, atol=1e-15)


@pytest.mark.parametrize("sim", [onp.simulate, epg_c.simulate], ids=["numpy", "c"])
@pytest.mark.parametrize("cap,tag", [(None, "unbounded"), (63, "cap63"), (10, "cap10")])
def test_g2_random_mse(golden, sim, cap, tag):
    g = golden("g2_random_mse")
    seq = sq.mse_tuples(g["T1"], g["T2"], g["B1"])
    sig, st = sim(seq, max_nstate=cap, return_states=True)
    close(sig, g["signal_" + tag])
    close(st, g["states_" + tag])


@pytest.mark.parametrize("sim", [onp.simulate, epg_c.simulate], ids=["numpy", "c"])
def test_g3_mrf(golden, sim):
    g = golden("g3_mrf")
    seq = sq.mrf_tuples(g["T1"], g["T2"], g["B1"], g["alpha"], g["TR"], float(g["TE"]))
    sig, st = sim(seq, max_nstate=63, return_states=True)
    close(sig, g["signal"])
    close(st, g["states"])


def test_g4_operator_tables(golden):
    g = golden("g4_operators")
    close(onp.rotation_matrix(120, 0)[0], g["T_120_0_mat"][0], 1e-16)
    close(onp.rotation_matrix(90, 90)[0], g["T_90_90_mat"][0], 1e-16)
    close(onp.rotation_matrix(g["T_alpha"], g["T_phi"]), g["T_mat"], 1e-16)
    arr, arr0 = onp.relaxation_coeffs(5, 150, 30, 0.01)
    close(arr, g["E_5_150_30_001_arr"], 1e-16)
    close(arr0, g["E_5_150_30_001_arr0"], 1e-16)
    assert np.isclose(arr[0, 0], 0.80505196038198 + 0.2615772384190187j, rtol=0, atol=1e-15)
    assert np.isclose(arr0[0, 2], 0.0327838995179941, rtol=0, atol=1e-15)
    arr, arr0 = onp.relaxation_coeffs(g["E_tau"], g["E_T1"], g["E_T2"], g["E_g"])
    close(arr, g["E_arr"], 1e-16)
    close(arr0, g["E_arr0"], 1e-16)
    parr, _ = onp.precession_coeffs(g["E_tau"], g["E_g"])
    close(parr, g["P_arr"], 1e-16)
    for k in (1, 2, -1, -3):
        grown = onp._pad_rows(g["shift_in"].copy(), 3 + abs(k))
        close(onp.shift_rows(grown, k), g[f"shift_out_k{k}"], 0)
        close(onp.shift_rows(g["shift_in"].copy(), k), g[f"shift_out_k{k}_nmax3"], 0)
    close(onp.shift_rows(onp._pad_rows(np.array([[1, 1, 0]], complex), 1), 1), g["shift_110_k1"], 0)


@pytest.mark.parametrize("sim", [onp.simulate, epg_c.simulate], ids=["numpy", "c"])
def test_g5_spgr(golden, sim):
    g = golden("g5_spgr")
    T1, T2, gg = float(g["T1"]), g["T2"], g["g"]
    seq = []
    for ph in g["phases"]:
        seq += [("T", float(g["alpha"]), ph), ("E", 5.0, T1, T2, gg), ("ADC",), ("E", 5.0, T1, T2, gg), ("S", 1)]
    sig, st = sim(seq, max_nstate=63, return_states=True)
    close(sig, g["signal_raw"])
    close(st, g["states"])
    if sim is onp.simulate:  # phase-compensated read-out (probe.py:155-165)
        seq2 = [op if op[0] != "ADC" else ("ADC", "F0", -ph) for ph in g["phases"] for op in
                [("T", float(g["alpha"]), ph), ("E", 5.0, T1, T2, gg), ("ADC",), ("E", 5.0, T1, T2, gg), ("S", 1)]]
        close(onp.simulate(seq2, max_nstate=63), g["signal"])


@pytest.mark.parametrize("sim", [onp.simulate, epg_c.simulate], ids=["numpy", "c"])
def test_g6_ssfp_negative_and_double_shifts(golden, sim):
    g = golden("g6_ssfp")
    TR = float(g["TR"])
    blk = [("T", float(g["FA"]), 0), ("S", -1), ("E", TR / 3, 1e3, 1e2, 0), ("ADC",), ("S", 2), ("E", TR * 2 / 3, 1e3, 1e2, 0), ("ADC",)]
    seq = blk * int(g["nrf"])
    sig, st = sim(seq, return_states=True)
    close(sig, g["signal"])
    close(st, g["states"])
    sig, st = sim(seq, max_nstate=5, return_states=True)
    close(sig, g["signal_cap5"])
    close(st, g["states_cap5"])


@pytest.mark.parametrize("sim", [onp.simulate, epg_c.simulate], ids=["numpy", "c"])
def test_g8_hyperecho(golden, sim):
    g = golden("g8_hyperecho")
    n = int(g["npulse"])
    se1 = [("S", 1), ("T", 10, 0), ("S", 1), ("ADC",), ("ADC", "Z0")]
    se2 = [("S", 1), ("T", -10, 0), ("S", 1), ("ADC",), ("ADC", "Z0")]
    seq = [("T", 90, 90)] + se1 * n + [("S", 1), ("T", 180, 0), ("S", 1)] + se2 * n
    sig = sim(seq)
    close(sig[0::2], g["F0"], 1e-12)
    close(sig[1::2], g["Z0"], 1e-12)
    assert np.allclose(sig[-2], 1) and np.allclose(sig[-1], 0)  # test/test_core.py:9-32


@pytest.mark.parametrize("sim", [onp.simulate, epg_c.simulate], ids=["numpy", "c"])
def test_g9_parity_sequence(golden, sim):
    g = golden("g9_parity_mse")
    blk = [("S", 1), ("E", 5, 1e3, g["T2"], g["g"]), ("T", 150, 0), ("S", 1), ("E", 5, 1e3, g["T2"], g["g"]), ("ADC",)]
    sig, st = sim([("T", 90, 90)] + blk * 10, return_states=True)
    close(sig, g["signal"])
    close(st, g["states"])


def test_known_answers_from_reference_tests():
    """values copied from test/test_transition.py:12-16, test/test_evolution.py:11-21,
    test/test_shift.py:10-11 of the reference"""
    eq = np.array([[0, 0, 1]], complex)
    assert np.allclose(onp.apply_matrix(eq[None], onp.rotation_matrix(90, 90))[0], [[1, 1, 0]])
    assert np.allclose(onp.apply_matrix(eq[None], onp.rotation_matrix(90, 0))[0], [[-1j, 1j, 0]])
    s110 = np.array([[[1, 1, 0]]], complex)
    eq0 = np.array([[[0, 0, 1]]], complex)
    arr, arr0 = onp.relaxation_coeffs(10, 1e10, 1e10, 0.025)
    assert np.allclose(onp.apply_scalar(s110, arr, arr0, eq0), [[[1j, -1j, 0]]])
    arr, arr0 = onp.relaxation_coeffs(10, 1e-10, 1e-10)
    assert np.allclose(onp.apply_scalar(s110, arr, arr0, eq0), [[[0, 0, 1]]])
    out = onp.shift_rows(onp._pad_rows(np.array([[1, 1, 0]], complex), 1), 1)
    assert np.allclose(out, [[0, 1, 0], [0, 0, 0], [1, 0, 0]])


def test_half_representation_roundtrip(golden):
    st = golden("g2_random_mse")["states_cap63"]
    half = onp.fold_half(st)
    assert half.shape == (64, 3, 41)
    close(onp.expand_half(half), st, 0)


def test_c_oracle_threads_agree():
    T1 = np.linspace(200, 3000, 16)[:, None]
    T2 = np.linspace(20, 300, 16)[None, :]
    seq = sq.mse_tuples(T1, T2)
    a = epg_c.simulate(seq, max_nstate=63, nthreads=1)
    b = epg_c.simulate(seq, max_nstate=63, nthreads=4)
    assert np.array_equal(a, b)
    close(a, onp.simulate(seq, max_nstate=63))


# ------------------------------------------------------------------ first-order derivatives
def test_g11_jacobian(golden):
    """the oracle's restatement of the order-1 recurrence (diff.py:264-288) vs the reference's Jacobian"""
    from tests import sequences as sq
    g = golden("g11_jacobian")
    tuples, _, variables = sq.jac_mse(g["T1"], g["T2"], g["B1"])
    np.testing.assert_allclose(onp.simulate_jacobian(tuples, variables), g["jac_mse"], rtol=0, atol=1e-13)
    tuples, _, variables = sq.jac_spgr(g["phases"], g["g"], g["T2b"])
    np.testing.assert_allclose(onp.simulate_jacobian(tuples, variables, max_nstate=63), g["jac_spgr"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(onp.simulate_jacobian(tuples, ["T2", "fa"], probe="Z0", max_nstate=63),
                               g["jac_spgr_z"], rtol=0, atol=1e-13)
    tuples, _, variables = sq.jac_params()
    np.testing.assert_allclose(onp.simulate_jacobian(tuples, variables), g["jac3"], rtol=0, atol=1e-13)


def test_g16_long_jacobian(golden):
    """derivatives of trains whose state matrix is never bounded (321 / 601 orders): the oracle vs the reference"""
    from tests import sequences as sq
    g = golden("g16_long_jacobian")
    for necho in (160, 300):
        tuples, _, variables = sq.jac_long(g["T1"], g["T2"], g["B1"], necho)
        np.testing.assert_allclose(onp.simulate_jacobian(tuples, variables)[19::20], g[f"jac_{necho}"], rtol=0, atol=1e-12)


def test_jacobian_finite_differences():
    """sanity of the restated partials: central differences of the plain simulation"""
    T1, T2 = np.array([800.0]), np.array([70.0])
    from tests import sequences as sq
    tuples, _, _ = sq.jac_mse(T1, T2, np.array([0.9]))
    jac = onp.simulate_jacobian(tuples, ["T2", "T1"])
    h = 1e-4
    for col, (d1, d2) in enumerate([(0.0, h), (h, 0.0)]):
        up = onp.simulate(sq.mse_tuples(T1 + d1, T2 + d2, 0.9, ESP=10.0, necho=6))
        dn = onp.simulate(sq.mse_tuples(T1 - d1, T2 - d2, 0.9, ESP=10.0, necho=6))
        np.testing.assert_allclose(jac[..., col], (up - dn) / (2 * h), rtol=0, atol=1e-9)


# ------------------------------------------------------------------ config 5: n-D shifts + diffusion
def test_g7_g12_nd_shift_and_diffusion(golden):
    """oracle restatement of shiftnd / D (shift.py:297-364, diffusion.py:60-147) vs the reference:
    signals, final state matrix and coordinate set"""
    from tests import sequences as sq
    g = golden("g12_nd")
    for name, tuples, opts in sq.nd_cases():
        sig, (states, coords) = onp.simulate_nd(tuples, return_states=True, **opts)
        np.testing.assert_allclose(sig, g[name + "_signal"], rtol=0, atol=1e-15)
        assert np.array_equal(coords, g[name + "_coords"])
        np.testing.assert_allclose(states, g[name + "_states"], rtol=0, atol=1e-15)
    g = golden("g7_pgse")
    T1, k1, F = float(g["T1"]), [int(v) for v in g["k"]], ("field", g["ADC"][None, :])
    T2 = g["T2"][:, None]
    seq = [("T", 90, 90), ("S", k1), ("D", 10, F, k1), ("E", 10, T1, T2), ("D", 20, F), ("E", 20, T1, T2), ("T", 180, 0),
           ("D", 20, F), ("E", 20, T1, T2), ("S", k1), ("D", 10, F, k1), ("E", 10, T1, T2), ("ADC",)]
    np.testing.assert_allclose(onp.simulate_nd(seq, kvalue=list(g["kvalue"]))[0], g["signal"], rtol=0, atol=1e-15)


def test_g14_vectorised_nd_shifts(golden):
    """one shift vector per voxel (shift.py:38-41): per-voxel coordinates, rows merged / sorted / cropped as whole slices
    (shift.py:330-341, :461-465) -- signals, final state matrix and coordinates of the reference"""
    from tests import sequences as sq
    g = golden("g14_nd_vector")
    for name, tuples, opts in sq.nd_vector_cases():
        sig, (states, coords) = onp.simulate_nd(tuples, return_states=True, **opts)
        np.testing.assert_allclose(sig, g[name + "_signal"], rtol=0, atol=1e-15)
        assert np.array_equal(coords, g[name + "_coords"])
        np.testing.assert_allclose(states, g[name + "_states"], rtol=0, atol=1e-15)
    assert np.ptp(np.abs(g["dwi_dirs_signal"][0]), axis=0).min() > 1e-3      # the directions really differ


def test_g11_jacobian_across_plain_operators(golden):
    """SPOILER leaves the derivative states alone in the reference (plain Operator): reproduced up
    to the first RESET, after which the reference's values come from a broadcasting accident"""
    from tests import sequences as sq
    g = golden("g11_jacobian")
    tuples, _, variables = sq.jac_plain_ops(g["T2b"])
    got = onp.simulate_jacobian(tuples, variables)
    np.testing.assert_allclose(got[:5], g["jac_plain"][:5], rtol=0, atol=1e-15)
    assert abs(got[1, 0, 0]) == 0 and abs(got[1, 0, 1]) > 1e-3          # spoiled signal, unspoiled "derivative"
    exact = onp.simulate_jacobian(tuples, variables, through_plain=True)
    assert not exact[1, :, 1:].any()                                   # d(0)/dv = 0
    np.testing.assert_allclose(exact[..., 0], got[..., 0], rtol=0, atol=0)
