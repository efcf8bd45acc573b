#!/usr/bin/env python3
"""Generate the golden input/output vectors under tests/golden/ from the reference.

Run ONLY in the build container, where the upstream reference (py-baudin/epgpy) is
mounted read-only at /root/reference:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference is imported as a black box (`import epgpy`) and driven through its public
API exactly as its own README / docs / tests do; only the resulting *data* (inputs and
expected outputs) are written, as small compressed .npz files.  Nothing of the reference
travels with the repository, and nothing on the GPU box needs /root/reference.

Cases:
  G1  README multi-spin-echo            (README.md:52-76)
  G2  random 64-voxel MSE-20            (T1, T2, B1 random; max_nstate 63/10)
  G3  MRF 1000-TR variable-FA SSFP      (examples/differentiation/optim_mrf.py:78-82 shape)
  G4  operator-level known answers      (transition.py:114, evolution.py:251, shift.py:271)
  G5  phase-cycled SPGR                 (docs/basics.md:124-127), g != 0
  G6  SSFP with S(-1), S(2)             (examples/basics/ssfp.py:9-15)
  G8  hyper-echo                        (test/test_core.py:9-32)
  G9  cupy<->numpy parity sequence      (test/test_common.py:123-162), 10 echoes, g array
  G10 direct operator calls on a StateMatrix (test/test_transition.py, test_evolution.py,
      test_shift.py known answers + random states)
  G7  PGSE diffusion, 3-D shift         (config 5; test/test_diffusion.py)
  G11 Jacobian probes (first-order derivatives, epgpy/diff.py)
  G14 vectorised integer n-D shifts: one vector per voxel (shift.py:38-41, test_shift.py:196-203)
  G15 a GENERAL equilibrium state matrix (statematrix.py:56-59): transverse and k != 0 coefficients, RESET
  G16 Jacobian probes of LONG unbounded trains (diff.py:119-139 has no limit on the orders): 160 and 300 echoes -- 321 / 601 orders
"""
import os
import sys

sys.dont_write_bytecode = True
REFERENCE = os.environ.get("EPGPY_REFERENCE", "/root/reference")
sys.path.insert(0, REFERENCE)

import numpy as np  # noqa: E402
import epgpy as epg  # noqa: E402  (the reference)
from epgpy import shift as ref_shift, transition as ref_transition, evolution as ref_evolution  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    size = os.path.getsize(path)
    print(f"{name}.npz: {size/1024:.1f} KiB  keys={sorted(arrays)}")


def run_with_states(seq, **options):
    """simulate + final state matrix (by applying the ops one by one, as test_common does)"""
    signal = epg.simulate(seq, **options)
    sm = epg.StateMatrix(shape=epg.getshape(seq), **options)
    for op in epg.functions.flatten_sequence(seq):
        sm = op(sm, inplace=True)
    return np.asarray(signal), np.asarray(sm.states)


# ---------------------------------------------------------------- G1
def g1():
    FA, ESP, necho = 120, 10, 20
    T1, T2 = 150, [30, 40, 50]
    exc, rfc = epg.T(90, 90), epg.T(FA, 0)
    rlx = epg.E(ESP / 2, T1, T2)
    shift = epg.S(1, duration=ESP / 2)
    seq = [exc] + [[shift, rlx, rfc, shift, rlx, epg.ADC]] * necho
    signal, states = run_with_states(seq)
    times = np.asarray(epg.get_adc_times(seq))
    save("g1_readme_mse", FA=FA, ESP=ESP, necho=necho, T1=T1, T2=np.asarray(T2, float),
         signal=signal, states=states, times=times)


# ---------------------------------------------------------------- G2
def g2():
    rng = np.random.default_rng(1)
    nvox, necho, FA, ESP = 64, 20, 120, 10
    T1 = rng.uniform(200, 3000, nvox)
    T2 = rng.uniform(20, 300, nvox)
    B1 = rng.uniform(0.7, 1.2, nvox)
    exc, rfc = epg.T(90 * B1, 90), epg.T(FA * B1, 0)
    rlx = epg.E(ESP / 2, T1, T2)
    shift = epg.S(1)
    seq = [exc] + [[shift, rlx, rfc, shift, rlx, epg.ADC]] * necho
    out = dict(T1=T1, T2=T2, B1=B1, FA=FA, ESP=ESP, necho=necho)
    for cap in (None, 63, 10):
        opts = {} if cap is None else {"max_nstate": cap}
        signal, states = run_with_states(seq, **opts)
        tag = "unbounded" if cap is None else f"cap{cap}"
        out[f"signal_{tag}"] = signal
        out[f"states_{tag}"] = states
    save("g2_random_mse", **out)


# ---------------------------------------------------------------- G3
def mrf_trains(ntr):
    rng = np.random.default_rng(0)
    u, v = rng.random(ntr), rng.random(ntr)
    i = np.arange(ntr)
    alpha = 10 + 50 * np.abs(np.sin(np.pi * i / 250)) * (0.6 + 0.4 * u)
    TR = 11 + 5 * v
    return alpha, TR


def g3():
    ntr, TE = 1000, 3.0
    alpha, TR = mrf_trains(ntr)
    # 16 voxels on a 2 x 2 x 4 (T1, T2, B1) grid
    T1 = np.array([500.0, 1500.0])[:, None, None]
    T2 = np.array([40.0, 120.0])[None, :, None]
    B1 = np.linspace(0.7, 1.3, 4)[None, None, :]
    seq = [epg.T(180 * B1, 90), epg.E(20, T1, T2)]
    for i in range(ntr):
        seq += [epg.T(alpha[i] * B1, 90), epg.E(TE, T1, T2), epg.ADC,
                epg.E(TR[i] - TE, T1, T2), epg.S(1)]
    signal, states = run_with_states(seq, max_nstate=63)
    save("g3_mrf", ntr=ntr, TE=TE, alpha=alpha, TR=TR, T1=T1, T2=T2, B1=B1,
         signal=signal, states=states)


# ---------------------------------------------------------------- G4
def g4():
    out = {}
    out["T_120_0_mat"] = epg.T(120, 0).mat
    out["T_90_90_mat"] = epg.T(90, 90).mat
    al = np.array([10.0, 33.3, 90.0, 120.0, 180.0, -45.0])
    ph = np.array([0.0, 15.0, 90.0, 117.0, -58.5, 270.0])
    out["T_alpha"], out["T_phi"] = al, ph
    out["T_mat"] = ref_transition.rotation_operator(al, ph)
    op = epg.E(5, 150, 30, 0.01)
    out["E_5_150_30_001_arr"], out["E_5_150_30_001_arr0"] = op.arr, op.arr0
    tau = np.array([1.0, 5.0, 12.5, 0.0])
    T1 = np.array([150.0, 1000.0, 3000.0, 800.0])
    T2 = np.array([30.0, 80.0, 300.0, 50.0])
    g = np.array([0.0, 0.01, -0.025, 0.2])
    arr, arr0 = ref_evolution.relaxation_operator(tau, T1, T2, g)
    out.update(E_tau=tau, E_T1=T1, E_T2=T2, E_g=g, E_arr=arr, E_arr0=arr0)
    parr, _ = ref_evolution.precession_operator(tau, g)
    out["P_arr"] = parr
    # shift1d
    rng = np.random.default_rng(4)
    n = 3
    half = rng.standard_normal((n + 1, 3)) + 1j * rng.standard_normal((n + 1, 3))
    full = np.zeros((2 * n + 1, 3), complex)
    full[n:] = half
    full[n, 1] = full[n, 0].conj()
    full[n, 2] = full[n, 2].real
    full[:n, 0] = full[:n:-1, 1].conj()
    full[:n, 1] = full[:n:-1, 0].conj()
    full[:n, 2] = full[:n:-1, 2].conj()
    out["shift_in"] = full
    for k in (1, 2, -1, -3):
        out[f"shift_out_k{k}"] = ref_shift.shift1d(full.copy(), k, inplace=False)
        out[f"shift_out_k{k}_nmax3"] = ref_shift.shift1d(full.copy(), k, inplace=False, nmax=3)
    out["shift_110_k1"] = ref_shift.shift1d(np.array([[1, 1, 0]], complex), 1, inplace=False)
    save("g4_operators", **out)


# ---------------------------------------------------------------- G5
def g5():
    necho = 400
    phases = 58.5 * np.arange(necho) ** 2
    T1, T2 = 1000.0, np.array([50.0, 100.0])
    g = np.array([[0.0, 0.013]])  # off-resonance axis (kHz)
    relax = epg.E(5, T1, T2, g)
    shift = epg.S(1)
    spgr = [[epg.T(14.8, ph), relax, epg.Adc(phase=-ph), relax, shift] for ph in phases]
    signal, states = run_with_states(spgr, max_nstate=63)
    # raw F0 without the ADC phase compensation
    spgr_raw = [[epg.T(14.8, ph), relax, epg.ADC, relax, shift] for ph in phases]
    signal_raw = np.asarray(epg.simulate(spgr_raw, max_nstate=63))
    save("g5_spgr", necho=necho, phases=phases, T1=T1, T2=T2, g=g, alpha=14.8, tau=5.0,
         signal=signal, signal_raw=signal_raw, states=states)


# ---------------------------------------------------------------- G6
def g6():
    FA, TR, nrf = 30, 10, 15
    rf = epg.T(FA, 0)
    s1, rx1 = epg.S(-1, duration=TR / 3), epg.E(TR / 3, 1e3, 1e2)
    s2, rx2 = epg.S(2, duration=TR * 2 / 3), epg.E(TR * 2 / 3, 1e3, 1e2)
    seq = [[rf, s1, rx1, epg.ADC, s2, rx2, epg.ADC]] * nrf
    signal, states = run_with_states(seq)
    sig_cap, st_cap = run_with_states(seq, max_nstate=5)
    save("g6_ssfp", FA=FA, TR=TR, nrf=nrf, signal=signal, states=states,
         signal_cap5=sig_cap, states_cap5=st_cap, times=np.asarray(epg.get_adc_times(seq)))


# ---------------------------------------------------------------- G8
def g8():
    npulse = 201
    excit, grad, adc = epg.T(90, 90), epg.S(1), epg.ADC
    p1, p2, inv = epg.T(10, 0), epg.T(-10, 0), epg.T(180, 0)
    se1, se2 = [grad, p1, grad, adc], [grad, p2, grad, adc]
    seq = [excit] + se1 * npulse + [grad, inv, grad] + se2 * npulse
    F0 = np.asarray(epg.simulate(seq, probe="F0"))
    Z0 = np.asarray(epg.simulate(seq, probe="Z0"))
    save("g8_hyperecho", npulse=npulse, F0=F0, Z0=Z0)


# ---------------------------------------------------------------- G9
def g9():
    exc, ref = epg.T(90, 90), epg.T(150, 0)
    shift = epg.S(1)
    T2, g = np.array([10.0, 20.0, 30.0]), np.array([[0.0, 0.1]])
    relax = epg.E(5, 1e3, T2, g=g)
    seq = [exc] + [shift, relax, ref, shift, relax, epg.ADC] * 10
    signal, states = run_with_states(seq)
    save("g9_parity_mse", T2=T2, g=g, signal=signal, states=states)


# ---------------------------------------------------------------- G10
def g10():
    """direct operator calls op(sm) on random (valid) state matrices"""
    rng = np.random.default_rng(10)
    n, shape = 4, (3, 2)
    half = rng.standard_normal(shape + (n + 1, 3)) + 1j * rng.standard_normal(shape + (n + 1, 3))
    full = np.zeros(shape + (2 * n + 1, 3), complex)
    full[..., n:, :] = half
    full[..., n, 1] = full[..., n, 0].conj()
    full[..., n, 2] = full[..., n, 2].real
    full[..., :n, 0] = full[..., :n:-1, 1].conj()
    full[..., :n, 1] = full[..., :n:-1, 0].conj()
    full[..., :n, 2] = full[..., :n:-1, 2].conj()
    sm0 = epg.StateMatrix(full)
    out = {"init": full}
    alpha = np.array([30.0, 90.0, 155.0])[:, None]
    phi = np.array([[10.0, -75.0]])
    out["T_alpha"], out["T_phi"] = alpha, phi
    out["T_states"] = np.asarray(epg.T(alpha, phi)(sm0).states)
    T2 = np.array([35.0, 70.0, 140.0])
    gg = np.array([[0.0, 0.02]])
    out["E_T2"], out["E_g"] = T2, gg
    out["E_states"] = np.asarray(epg.E(7.5, 900.0, T2, gg)(sm0).states)
    for k in (1, -1, 2, -3):
        out[f"S{k}_states"] = np.asarray(epg.S(k)(sm0).states)
        smc = epg.StateMatrix(full, max_nstate=n)
        out[f"S{k}_cap_states"] = np.asarray(epg.S(k)(smc).states)
    # density / PD / spoiler / reset
    smd = epg.StateMatrix(density=[1.0, 3.0])
    seq = [epg.T(60, 20), epg.S(1), epg.E(10, 200, 50), epg.T(60, 20), epg.S(1), epg.E(10, 200, 50)]
    sm = smd
    for op in seq:
        sm = op(sm)
    out["density_states"] = np.asarray(sm.states)
    out["spoiler_states"] = np.asarray(epg.SPOILER(sm).states)
    sm2 = epg.E(10, 200, 50)(epg.SPOILER(sm))
    out["spoiler_E_states"] = np.asarray(sm2.states)
    out["F0"], out["Z0"] = np.asarray(sm.F0), np.asarray(sm.Z0)
    out["norm"] = np.asarray(sm.norm)
    save("g10_direct_ops", **out)


# ---------------------------------------------------------------- G7 (config 5)
def g7():
    T2 = np.linspace(20, 300, 8)
    ADCs = np.linspace(1e-4, 3e-3, 8)
    T1 = 1000.0
    kvalue = [2e4, 1e4, 5e3]
    k1 = [1, 1, 1]
    sig = np.zeros((8, 8), complex)
    nstates = []
    for j, adc in enumerate(ADCs):
        seq = [
            epg.T(90, 90), epg.S(k1), epg.D(10, adc, k=k1), epg.E(10, T1, T2),
            epg.D(20, adc), epg.E(20, T1, T2),
            epg.T(180, 0),
            epg.D(20, adc), epg.E(20, T1, T2), epg.S(k1), epg.D(10, adc, k=k1), epg.E(10, T1, T2),
            epg.ADC,
        ]
        s = np.asarray(epg.simulate(seq, kvalue=kvalue))
        sig[:, j] = s[0]
    save("g7_pgse", T2=T2, ADC=ADCs, T1=T1, kvalue=np.asarray(kvalue), k=np.asarray(k1), signal=sig)


# ---------------------------------------------------------------- G11 (first-order derivatives)
def g11():
    """Jacobian probes (epgpy/diff.py:384-416): derivatives w.r.t. tissue / system parameters"""
    rng = np.random.default_rng(11)
    nvox = 8
    T1, T2, B1 = rng.uniform(300, 2000, nvox), rng.uniform(30, 200, nvox), rng.uniform(0.8, 1.2, nvox)
    exc = epg.T(90 * B1, 90, order1={"B1": {"alpha": 90}})
    rfc = epg.T(120 * B1, 0, order1={"B1": {"alpha": 120}})
    rlx = epg.E(5, T1, T2, order1=["T1", "T2"])
    sh = epg.S(1)
    seq = [exc] + [sh, rlx, rfc, sh, rlx, epg.ADC] * 6
    jac_mse = np.asarray(epg.simulate(seq, probe=epg.Jacobian(["magnitude", "T1", "T2", "B1"])))
    # SPGR-like train: off-resonance and RF-phase derivatives (complex), aliases, Z0 Jacobian
    ntr = 30
    phases = 58.5 * np.arange(ntr) ** 2
    g = np.array([[0.0, 0.013, -0.021]])
    T2b = np.array([50.0, 100.0])
    rl = epg.E(5, 1000.0, T2b, g, order1=["g", "T2"])
    spgr = []
    for ph in phases:
        spgr += [epg.T(14.8, ph, order1={"phi0": "phi", "fa": "alpha"}), rl, epg.ADC, rl, epg.S(1)]
    jac_spgr = np.asarray(epg.simulate(spgr, probe=epg.Jacobian(["g", "phi0", "T2", "fa", "magnitude"]), max_nstate=63))
    jac_spgr_z = np.asarray(epg.simulate(spgr, probe=epg.Jacobian(["T2", "fa"], probe="Z0"), max_nstate=63))
    # tau / P / R parameters
    seq3 = [epg.T(60, 20), epg.P(3.0, 0.02, order1=["g"]), epg.S(1), epg.E(4.0, 700.0, 60.0, order1={"tau": "tau"}),
            epg.T(70, -30, order1=True), epg.S(-1), epg.R(0.1 + 0.3j, 0.2, r0=0.2, order1=["rT", "rL", "r0"]), epg.ADC]
    jac3 = np.asarray(epg.simulate(seq3, probe=epg.Jacobian(["magnitude", "g", "tau", "alpha", "phi", "rT", "rL", "r0"])))
    # SPOILER / RESET / PD are plain Operators in the reference: they do not touch sm.order1
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from tests import sequences as sq
    _, ops, variables = sq.jac_plain_ops(T2b)
    seq_plain = ops(epg)
    jac_plain = [np.asarray(v) for v in epg.simulate(seq_plain, probe=epg.Jacobian(variables), asarray=False)]
    save("g11_jacobian", T1=T1, T2=T2, B1=B1, jac_mse=jac_mse, phases=phases, g=g, T2b=T2b,
         jac_spgr=jac_spgr, jac_spgr_z=jac_spgr_z, jac3=jac3, jac_plain=np.asarray(jac_plain))


# ---------------------------------------------------------------- G12 (n-D integer shifts, diffusion: more cases)
def g12_cases():
    """(name, oracle tuples, simulate options): shared with the tests through tests/sequences.py"""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from tests import sequences as sq
    return sq.nd_cases()


def g12(cases=None, fname="g12_nd"):
    out = {}
    for name, tuples, opts in (cases or g12_cases()):
        seq = []
        for t in tuples:
            if t[0] == "T":
                seq.append(epg.T(t[1], t[2]))
            elif t[0] == "E":
                seq.append(epg.E(*t[1:]))
            elif t[0] == "S":
                seq.append(epg.S(t[1] if np.isscalar(t[1]) else np.asarray(t[1], dtype=int).tolist()))
            elif t[0] == "D":
                seq.append(epg.D(t[1], t[2], k=(list(t[3]) if len(t) > 3 and t[3] is not None else None)))
            elif t[0] == "ADC":
                seq.append(epg.ADC if len(t) == 1 else epg.Adc(t[1]))
            elif t[0] == "SPOILER":
                seq.append(epg.SPOILER)
        final = {}

        def grab(sm):
            final["states"], final["coords"] = np.array(sm.states), (None if sm.coords is None else np.array(sm.coords))

        sig = np.asarray(epg.simulate(seq, callback=grab, **opts))
        out[name + "_signal"] = sig
        out[name + "_states"] = final["states"]
        out[name + "_coords"] = (final["coords"].reshape(final["coords"].shape[-2:]) if cases is None else final["coords"])
    save(fname, **out)


# ---------------------------------------------------------------- G14 (vectorised n-D shifts: per-voxel coordinates)
def g14():
    g12_cases()       # (puts the repository root on sys.path)
    from tests import sequences as sq
    g12(sq.nd_vector_cases(), "g14_nd_vector")


# ---------------------------------------------------------------- G13 (second-order derivatives)
def g13():
    """Hessian probes (epgpy/diff.py:419-472) -- evaluated by the reference itself"""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from tests import sequences as sq
    out = {}
    for name, seq, probes, opts in sq.hessian_cases(epg):
        res = epg.simulate(seq, probe=probes, **opts)
        for i, arr in enumerate(res):
            out[f"{name}_{i}"] = np.asarray(arr)
    save("g13_hessian", **out)


# ---------------------------------------------------------------- G15 (general equilibrium)
def g15():
    """StateMatrix(equilibrium=<a state matrix with transverse and k != 0 coefficients>) (statematrix.py:56-59): the recovery
    term of every relaxation is `arr0 * equilibrium` over ALL rows (opscalar.py:213-232), RESET returns to it and crops
    (operator.py:297-304).  The equilibrium itself is a valid state matrix made by the reference (two pulses and two shifts);
    operators over a (T1, T2) grid; state, equilibrium and F0 / Z0 after every operator"""
    eq = epg.S(1)(epg.T(50, 10)(epg.S(1)(epg.T(30, 40)(epg.StateMatrix([0, 0, 1]))))).states      # [1, 5, 3]
    eq = eq * 0.8
    T1 = np.array([300.0, 900.0, 2000.0])[:, None]
    T2 = np.array([40.0, 120.0])[None, :]
    seq = [epg.T(70, 25), epg.E(8.0, T1, T2, 0.01), epg.S(1), epg.T(120, 0), epg.E(5.0, T1, T2), epg.S(1), epg.E(12.0, T1, T2, -0.02),
           epg.S(-1), epg.T(40, 90), epg.RESET, epg.T(60, 10), epg.E(9.0, T1, T2), epg.S(1), epg.E(3.0, T1, T2)]
    sm = epg.StateMatrix(equilibrium=eq, shape=(3, 2))
    out = {"equilibrium": eq, "T1": T1, "T2": T2, "init_states": np.asarray(sm.states), "init_density": np.asarray(sm.density)}
    for i, op in enumerate(seq):
        sm = op(sm)
        out[f"states_{i}"] = np.asarray(sm.states)
        out[f"equilibrium_{i}"] = np.asarray(sm.equilibrium)
    # and through simulate(): init = a state matrix with that equilibrium, probes F0 and Z0
    train = [epg.T(70, 25)] + [epg.E(8.0, T1, T2, 0.01), epg.S(1), epg.T(120, 0), epg.E(5.0, T1, T2), epg.ADC] * 4
    f0, z0 = epg.simulate(train, init=epg.StateMatrix(equilibrium=eq, shape=(3, 2)), probe=["F0", "Z0"])
    out["sim_F0"], out["sim_Z0"] = np.asarray(f0), np.asarray(z0)
    # random sequences (tests/sequences.py::random_sequence without PD: T / E / P / S(+-1..3) / probes / SPOILER / RESET over a
    # 4 x 3 grid) on state matrices with other general equilibria: final states and the F0 / Z0 records of simulate(init=...)
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from tests import sequences as sq
    for i in range(4):
        rng = np.random.default_rng(1500 + i)
        tuples = [t for t in sq.random_sequence(rng, (4, 3), nops=30) if t[0] != "PD"]
        eq_i = epg.S(int(rng.choice([1, -1])))(epg.T(float(rng.uniform(20, 160)), float(rng.uniform(-180, 180)))(
            epg.S(1)(epg.T(float(rng.uniform(20, 160)), float(rng.uniform(-180, 180)))(epg.StateMatrix([0, 0, float(rng.uniform(0.5, 1.5))]))))).states
        seq_i = sq.to_ops(epg, tuples)
        sm = epg.StateMatrix(equilibrium=eq_i, shape=(4, 3))
        for op in seq_i:
            sm = op(sm)
        out[f"rand{i}_equilibrium"] = eq_i
        out[f"rand{i}_states"] = np.asarray(sm.states)
        f0, z0 = epg.simulate(seq_i, init=epg.StateMatrix(equilibrium=eq_i, shape=(4, 3)), probe=["F0", "Z0"])
        out[f"rand{i}_F0"], out[f"rand{i}_Z0"] = np.asarray(f0), np.asarray(z0)
    save("g15_equilibrium", **out)


# ---------------------------------------------------------------- G16 (derivatives of long state matrices)
def g16():
    """echo trains whose state matrix is never bounded, with four derivative variables (two passes at 512 orders per voxel, four
    at 1024 on the device); only every 20th echo is kept (the files stay small)"""
    rng = np.random.default_rng(16)
    nvox = 3
    T1, T2, B1 = rng.uniform(300, 2000, nvox), rng.uniform(60, 200, nvox), rng.uniform(0.8, 1.2, nvox)
    out = {"T1": T1, "T2": T2, "B1": B1}
    for necho in (160, 300):
        exc = epg.T(90 * B1, 90, order1={"B1": {"alpha": 90}})
        rfc = epg.T(150 * B1, 0, order1={"B1": {"alpha": 150}, "fa": {"alpha": 1.0}})
        rlx = epg.E(2.5, T1, T2, order1=["T1", "T2"])
        sh = epg.S(1)
        seq = [exc] + [sh, rlx, rfc, sh, rlx, epg.ADC] * necho
        jac = np.asarray(epg.simulate(seq, probe=epg.Jacobian(["magnitude", "T1", "T2", "B1", "fa"])))
        out[f"jac_{necho}"] = jac[19::20]
    save("g16_long_jacobian", **out)


if __name__ == "__main__":
    print("reference:", epg.__file__)
    only = sys.argv[1:]
    for fn in (g1, g2, g3, g4, g5, g6, g8, g9, g10, g7, g11, g12, g13, g14, g15, g16):
        if not only or fn.__name__ in only:
            fn()
