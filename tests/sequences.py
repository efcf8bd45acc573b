"""Sequence definitions shared by the tests: each builder returns BOTH the oracle's tuple
description and (lazily) the product's operator list, from the same parameters, so the
GPU path and the oracle always see identical inputs."""
import numpy as np


from epgpy_amd.workloads import mse_sequence as mse_ops, mrf_sequence as mrf_ops  # noqa: E402,F401  (product side)
from oracle.workloads import mse_tuples, mrf_tuples, mrf_trains  # noqa: E402,F401  (oracle side)


def to_ops(epg, tuples):
    """generic converter tuple description -> product operators"""
    ops = []
    for t in tuples:
        k = t[0]
        if k == "T":
            ops.append(epg.T(t[1], t[2]))
        elif k == "E":
            ops.append(epg.E(*t[1:]))
        elif k == "P":
            ops.append(epg.P(t[1], t[2]))
        elif k == "S":
            ops.append(epg.S(t[1], nmax=t[2] if len(t) > 2 else None))
        elif k == "ADC":
            what = t[1] if len(t) > 1 else "F0"
            phase = t[2] if len(t) > 2 else None
            ops.append(epg.ADC if (what == "F0" and phase is None) else epg.Adc(what, phase=phase))
        elif k == "SPOILER":
            ops.append(epg.SPOILER)
        elif k == "RESET":
            ops.append(epg.RESET)
        elif k == "PD":
            ops.append(epg.PD(t[1], reset=(len(t) < 3 or t[2])))
        elif k == "WAIT":
            ops.append(epg.NULL)
        else:
            raise ValueError(k)
    return ops


# ------------------------------------------------------------------ first-order derivatives (g11)
def jac_mse(T1, T2, B1, necho=6):
    """(oracle tuples, builder of product operators, Jacobian variables): MSE with T1/T2/B1 derivatives"""
    exc_o1, rfc_o1 = {"B1": {"alpha": 90}}, {"B1": {"alpha": 120}}
    rl_o1 = {"T1": {"T1": 1}, "T2": {"T2": 1}}
    tuples = [("T", 90 * B1, 90, {"order1": exc_o1})] + [
        ("S", 1), ("E", 5, T1, T2, 0, {"order1": rl_o1}), ("T", 120 * B1, 0, {"order1": rfc_o1}),
        ("S", 1), ("E", 5, T1, T2, 0, {"order1": rl_o1}), ("ADC",)] * necho

    def ops(epg):
        exc = epg.T(90 * B1, 90, order1=exc_o1)
        rfc = epg.T(120 * B1, 0, order1=rfc_o1)
        rlx = epg.E(5, T1, T2, order1=["T1", "T2"])
        sh = epg.S(1)
        return [exc] + [sh, rlx, rfc, sh, rlx, epg.ADC] * necho

    return tuples, ops, ["magnitude", "T1", "T2", "B1"]


def jac_long(T1, T2, B1, necho):
    """g16: an echo train whose state matrix is never bounded (2 necho + 1 orders), derivatives w.r.t. T1 / T2 / B1 / the
    refocusing angle -- (oracle tuples, builder of product operators, Jacobian variables)"""
    exc_o1, rfc_o1 = {"B1": {"alpha": 90}}, {"B1": {"alpha": 150}, "fa": {"alpha": 1.0}}
    rl_o1 = {"T1": {"T1": 1}, "T2": {"T2": 1}}
    tuples = [("T", 90 * B1, 90, {"order1": exc_o1})] + [
        ("S", 1), ("E", 2.5, T1, T2, 0, {"order1": rl_o1}), ("T", 150 * B1, 0, {"order1": rfc_o1}),
        ("S", 1), ("E", 2.5, T1, T2, 0, {"order1": rl_o1}), ("ADC",)] * necho

    def ops(epg):
        exc = epg.T(90 * B1, 90, order1=exc_o1)
        rfc = epg.T(150 * B1, 0, order1=rfc_o1)
        rlx = epg.E(2.5, T1, T2, order1=["T1", "T2"])
        sh = epg.S(1)
        return [exc] + [sh, rlx, rfc, sh, rlx, epg.ADC] * necho

    return tuples, ops, ["magnitude", "T1", "T2", "B1", "fa"]


def jac_spgr(phases, g, T2b, T1=1000.0):
    """RF-spoiled gradient echo: complex derivatives (off-resonance, RF phase), aliases"""
    rl_o1 = {"g": {"g": 1}, "T2": {"T2": 1}}
    t_o1 = {"phi0": {"phi": 1}, "fa": {"alpha": 1}}
    tuples = []
    for ph in phases:
        tuples += [("T", 14.8, ph, {"order1": t_o1}), ("E", 5, T1, T2b, g, {"order1": rl_o1}), ("ADC",),
                   ("E", 5, T1, T2b, g, {"order1": rl_o1}), ("S", 1)]

    def ops(epg):
        rl = epg.E(5, T1, T2b, g, order1=["g", "T2"])
        seq = []
        for ph in phases:
            seq += [epg.T(14.8, ph, order1={"phi0": "phi", "fa": "alpha"}), rl, epg.ADC, rl, epg.S(1)]
        return seq

    return tuples, ops, ["g", "phi0", "T2", "fa", "magnitude"]


def jac_params():
    """every differentiable parameter of T / P / E / R once"""
    all_t = {"alpha": {"alpha": 1}, "phi": {"phi": 1}}
    all_r = {"rT": {"rT": 1}, "rL": {"rL": 1}, "r0": {"r0": 1}}
    tuples = [("T", 60, 20), ("P", 3.0, 0.02, {"order1": {"g": {"g": 1}}}), ("S", 1),
              ("E", 4.0, 700.0, 60.0, 0, {"order1": {"tau": {"tau": 1}}}), ("T", 70, -30, {"order1": all_t}),
              ("S", -1), ("R", 0.1 + 0.3j, 0.2, 0.2, {"order1": all_r}), ("ADC",)]

    def ops(epg):
        return [epg.T(60, 20), epg.P(3.0, 0.02, order1=["g"]), epg.S(1),
                epg.E(4.0, 700.0, 60.0, order1={"tau": "tau"}), epg.T(70, -30, order1=True), epg.S(-1),
                epg.R(0.1 + 0.3j, 0.2, r0=0.2, order1=["rT", "rL", "r0"]), epg.ADC]

    return tuples, ops, ["magnitude", "g", "tau", "alpha", "phi", "rT", "rL", "r0"]


# ------------------------------------------------------------------ randomized sequences
def random_sequence(rng, grid, nops=40, precession=True):
    """random operator tuples over `grid`: parameters broadcast over random subsets of the grid
    axes (scalar / leading axes / inner axis only), shifts of +-1..3, all probe kinds,
    SPOILER / RESET / PD sprinkled in.  Always ends with an ADC.
    precession=False: relaxation only (g = 0, no P) and mostly shifts by +1 -- the shapes the E.T.E
    fusion and the leading-shift records are made for"""
    nd = len(grid)

    def param(lo, hi):
        pattern = rng.integers(0, 4)
        if pattern == 0:
            return float(rng.uniform(lo, hi))
        if pattern == 1:        # leading axes up to a random depth
            depth = int(rng.integers(1, nd + 1))
            return rng.uniform(lo, hi, grid[:depth])
        if pattern == 2:        # one inner axis only
            ax = int(rng.integers(0, nd))
            shape = [1] * (ax + 1)
            shape[ax] = grid[ax]
            return rng.uniform(lo, hi, shape)
        return rng.uniform(lo, hi, grid)

    ops = [("T", param(20, 160), param(-180, 180))]
    for _ in range(nops):
        r = rng.random()
        if r < 0.25:
            ops.append(("T", param(5, 175), param(-180, 180)))
        elif r < 0.50:
            ops.append(("E", param(1, 20), param(200, 3000), param(20, 300), param(-0.05, 0.05) if precession else 0))
        elif r < 0.55 and precession:
            ops.append(("P", param(1, 10), param(-0.1, 0.1)))
        elif r < 0.80:
            k = int(rng.choice([1, 1, 1, -1, -1, 2, -2, 3, -3] if precession else [1, 1, 1, 1, 1, 1, -1, 2]))
            ops.append(("S", k))
        elif r < 0.92:
            ops.append(("ADC", str(rng.choice(["F0", "Z0"])), None if rng.random() < 0.7 else float(rng.uniform(0, 360))))
        elif r < 0.95:
            ops.append(("SPOILER",))
        elif r < 0.97:
            ops.append(("RESET",))
        else:
            ops.append(("PD", param(0.2, 1.5), bool(rng.random() < 0.5)))
    ops.append(("ADC",))
    return ops


# ------------------------------------------------------------------ n-D integer shifts + diffusion (g12)
def nd_cases():
    """[(name, oracle tuples, simulate options)]: coordinate-indexed state matrices"""
    T2 = np.array([40.0, 80.0, 160.0])
    cases = []
    # 2-D gradients in several directions, an int shift once coordinates exist, Z0 probes
    a = [("T", 70, 30), ("S", [1, 0]), ("E", 5, 900, T2), ("T", 50, -40), ("S", [0, 1]), ("E", 5, 900, T2),
         ("ADC",), ("T", 110, 10), ("S", [-1, 1]), ("ADC", "Z0"), ("S", 1), ("E", 3, 900, T2), ("T", 35, 75),
         ("S", [1, 0]), ("ADC",), ("S", [0, -1]), ("T", 60, 0), ("S", [1, 1]), ("E", 2, 900, T2), ("ADC",)]
    cases.append(("grad2d", a, {}))
    # 1-D orders first, coordinates set up later (statematrix.py:314-329), cropping at max_nstate
    b = [("T", 60, 20), ("S", 1), ("E", 4, 700, 90), ("T", 45, 0), ("S", 1), ("ADC",), ("T", 80, 50),
         ("S", [1, 2, 0]), ("E", 4, 700, 90), ("ADC",), ("T", 30, -20), ("S", [0, -1, 1]), ("ADC",),
         ("T", 120, 0), ("S", [1, 2, 0]), ("E", 4, 700, 90), ("ADC",), ("ADC", "Z0")]
    cases.append(("setup_late", b, {"max_nstate": 3}))
    # anisotropic diffusion tensor, 3-D gradient, stimulated-echo style train
    Dt = np.array([[1.0, 0.2, 0.0], [0.2, 2.0, 0.1], [0.0, 0.1, 0.5]]) * 1e-3
    k = [2, -1, 1]
    c = [("T", 90, 90), ("S", k), ("D", 8, Dt, k), ("E", 8, 1000, T2), ("T", 90, 0), ("D", 30, Dt), ("E", 30, 1000, T2),
         ("ADC", "Z0"), ("T", 90, 0), ("S", k), ("D", 8, Dt, k), ("E", 8, 1000, T2), ("ADC",),
         ("SPOILER",), ("T", 40, 10), ("S", [1, 0, 0]), ("D", 5, 1.5e-3, [1, 0, 0]), ("ADC",)]
    cases.append(("tensor3d", c, {"kvalue": [3e4, 2e4, 1e4]}))
    return cases


def nd_vector_cases():
    """[(name, oracle tuples, simulate options)]: VECTORISED n-D shifts -- one integer vector per point of the leading
    grid axis (shift.py:38-41, test_shift.py:196-203): the rows' coordinates then differ from voxel to voxel"""
    T2 = np.array([[40.0, 80.0, 160.0]])                          # axis 1
    # axis 0: four gradient directions / amplitudes.  Given as [4, 1, kdim]: the reference aligns a vectorised k with the
    # grid NumPy-style (from the right), so on a 2-D grid the second axis has to be spelled out
    K = np.array([[1, 0, 0], [2, 1, 0], [1, -1, 2], [0, 2, -1]])[:, None, :]
    K2 = [[1, 1], [2, -1], [-1, 2], [3, 0]]
    Dt = np.array([[1.0, 0.2, 0.0], [0.2, 2.0, 0.1], [0.0, 0.1, 0.5]]) * 1e-3
    cases = []
    # diffusion-weighted spin echo, a direction per voxel; afterwards a shared gradient and a stimulated-echo pathway
    a = [("T", 90, 90), ("S", K), ("D", 8, Dt), ("E", 8, 1000, T2), ("T", 170, 0), ("S", K), ("D", 8, Dt), ("E", 8, 1000, T2),
         ("ADC",), ("T", 60, 20), ("S", [1, 0, 0]), ("D", 4, 1.2e-3, [1, 0, 0]), ("E", 4, 1000, T2), ("ADC", "Z0"),
         ("T", 100, -30), ("S", K), ("ADC",), ("T", 140, 10), ("S", K), ("S", [1, 0, 0]), ("E", 6, 1000, T2), ("ADC",)]
    cases.append(("dwi_dirs", a, {"kvalue": [3e4, 2e4, 1e4]}))
    # 1-D orders first, a vector per voxel later (statematrix.py:314-329); int shift on per-voxel coordinates;
    # spoiler; cropping at max_nstate (kept if inside the box in ANY voxel, shift.py:330-341)
    b = [("T", 60, 20), ("S", 1), ("E", 4, 700, 90), ("T", 45, 0), ("S", 1), ("ADC",), ("T", 80, 50), ("S", K2),
         ("E", 4, 700, 90), ("ADC",), ("T", 30, -20), ("S", [0, -1]), ("ADC",), ("T", 120, 0), ("S", K2), ("S", 1),
         ("E", 4, 700, 90), ("ADC",), ("SPOILER",), ("T", 50, 0), ("S", K2), ("T", 50, 90), ("S", -1), ("ADC",), ("ADC", "Z0")]
    cases.append(("vector_late", b, {"max_nstate": 3}))
    cases.append(("vector_free", b, {}))
    return cases


def nd_to_ops(epg, tuples):
    """oracle tuples of an n-D sequence -> product operators"""
    ops = []
    for t in tuples:
        if t[0] == "S":
            ops.append(epg.S(t[1] if np.isscalar(t[1]) else np.asarray(t[1], dtype=int).tolist()))
        elif t[0] == "D":
            D, k = t[2], (t[3] if len(t) > 3 else None)
            k = None if k is None else [int(v) for v in k]
            if isinstance(D, tuple):
                ops.append(epg.D(t[1], D[1], k=k, field=True))
            else:
                ops.append(epg.D(t[1], D, k=k))
        else:
            ops.extend(to_ops(epg, [t]))
    return ops


def random_nd_sequence(rng, grid, kdim, nops=25, vector=False):
    """random n-D gradient sequence: shifts with components in -2..2, T / E / D / SPOILER / probes;
    `vector`: some shifts come with one vector per point of the first grid axis (or of the first two)"""
    def param(lo, hi):
        return float(rng.uniform(lo, hi)) if rng.random() < 0.5 else rng.uniform(lo, hi, grid[: int(rng.integers(1, len(grid) + 1))])

    def delta():
        if vector and rng.random() < 0.45:
            lead = grid[: int(rng.integers(1, min(len(grid), 2) + 1))]
            while True:
                d = rng.integers(-2, 3, tuple(lead) + (kdim,))
                if d.any() and (np.prod(lead) > 1):
                    return d
                if np.prod(lead) == 1:
                    break
        while True:
            d = rng.integers(-2, 3, kdim)
            if d.any():
                return [int(v) for v in d]

    ops = [("T", param(30, 150), param(-180, 180))]
    last_shift = None
    have_coords = False   # a diffusion TENSOR needs a state whose coordinates have its dimension: before
    # the first n-D shift the reference would NumPy-broadcast a 1x1 b-matrix against it (diffusion.py:140-145)
    for _ in range(nops):
        r = rng.random()
        if r < 0.25:
            ops.append(("T", param(10, 170), param(-180, 180)))
            last_shift = None
        elif r < 0.45:
            ops.append(("E", param(1, 15), param(300, 2000), param(30, 200)))
        elif r < 0.70:
            last_shift = delta() if rng.random() < 0.85 else int(rng.choice([-1, 1]))
            have_coords = have_coords or not np.isscalar(last_shift)
            ops.append(("S", last_shift))
        elif r < 0.85:
            D = float(rng.uniform(0.2e-3, 3e-3))
            if rng.random() < 0.3:
                D = ("field", rng.uniform(0.2e-3, 3e-3, grid))
            elif rng.random() < 0.3 and have_coords:
                m = rng.uniform(-1, 1, (kdim, kdim))
                D = (m @ m.T + np.eye(kdim)) * 1e-3
            k = last_shift if (last_shift is not None and not np.isscalar(last_shift) and np.ndim(last_shift) == 1
                               and ops[-1][0] == "S") else None
            ops.append(("D", float(rng.uniform(1, 20)), D, k))
        elif r < 0.97:
            ops.append(("ADC", str(rng.choice(["F0", "Z0"]))))
        else:
            ops.append(("SPOILER",))
    ops.append(("ADC",))
    return ops


def random_jacobian_sequence(rng, grid, nops=25):
    """random T / E / P / S / ADC train whose operators differentiate against a pool of shared
    variable names with random coefficients; returns (oracle tuples, product-operator builder, variables)"""
    names = ["a", "b", "c", "d", "e"]

    def param(lo, hi):
        return float(rng.uniform(lo, hi)) if rng.random() < 0.5 else rng.uniform(lo, hi, grid[: int(rng.integers(1, len(grid) + 1))])

    def order1(params):
        o1 = {}
        for p in params:
            if rng.random() < 0.6:
                var = str(rng.choice(names))
                o1.setdefault(var, {})[p] = float(rng.uniform(-2, 2)) if rng.random() < 0.5 else 1
        return o1

    tuples, build = [], []
    for i in range(nops):
        r = rng.random()
        if r < 0.3 or i == 0:
            a, p, o1 = param(10, 170), param(-180, 180), order1(["alpha", "phi"])
            tuples.append(("T", a, p, {"order1": o1}))
            build.append(lambda epg, a=a, p=p, o1=o1: epg.T(a, p, order1=o1 or False))
        elif r < 0.6:
            args, o1 = (param(1, 15), param(300, 2000), param(30, 200), param(-0.05, 0.05)), order1(["tau", "T1", "T2", "g"])
            tuples.append(("E",) + args + ({"order1": o1},))
            build.append(lambda epg, args=args, o1=o1: epg.E(*args, order1=o1 or False))
        elif r < 0.65:
            args, o1 = (param(1, 10), param(-0.1, 0.1)), order1(["tau", "g"])
            tuples.append(("P",) + args + ({"order1": o1},))
            build.append(lambda epg, args=args, o1=o1: epg.P(*args, order1=o1 or False))
        elif r < 0.85:
            k = int(rng.choice([1, 1, -1, 2, -2]))
            tuples.append(("S", k))
            build.append(lambda epg, k=k: epg.S(k))
        else:
            tuples.append(("ADC",))
            build.append(lambda epg: epg.ADC)
    tuples.append(("ADC",))
    build.append(lambda epg: epg.ADC)
    return tuples, (lambda epg: [b(epg) for b in build]), ["magnitude"] + names


def random_fusable_jacobian_sequence(rng, grid, nvars, nblocks=8):
    """random spin-echo-like train whose relaxations are precession-free, so that the planner fuses E . T . E runs ALSO in
    differentiated plans (generated partials): rotations about x / y / a random axis, relaxations before and / or after
    them (nested fusions), shifts by +-1, partials of the rotation (alpha, phi) and of the relaxation (tau, T1, T2) against
    `nvars` shared variables with scalar or per-voxel coefficients.  Returns (tuples, builder, variables)"""
    names = ["u", "v"][:nvars]

    def param(lo, hi):
        return float(rng.uniform(lo, hi)) if rng.random() < 0.4 else rng.uniform(lo, hi, grid[: int(rng.integers(1, len(grid) + 1))])

    def coeff():
        r = rng.random()
        if r < 0.4:
            return 1
        if r < 0.8:
            return float(rng.uniform(-2, 2))
        return rng.uniform(-2, 2, grid[:1])

    def order1(params):
        o1 = {}
        for p in params:
            if rng.random() < 0.55:
                o1.setdefault(str(rng.choice(names)), {})[p] = coeff()
        return o1

    tuples, build = [], []

    def add_T():
        phi = [0.0, 90.0, 180.0, -90.0, float(rng.uniform(-180, 180))][int(rng.integers(0, 5))]
        a, o1 = param(20, 170), order1(["alpha"] if phi in (0.0, 90.0, 180.0, -90.0) and rng.random() < 0.7 else ["alpha", "phi"])
        tuples.append(("T", a, phi, {"order1": o1}))
        build.append(lambda epg, a=a, phi=phi, o1=o1: epg.T(a, phi, order1=o1 or False))

    def add_E():
        args, o1 = (param(1, 12), param(300, 2000), param(30, 200), 0), order1(["tau", "T1", "T2"])
        tuples.append(("E",) + args + ({"order1": o1},))
        build.append(lambda epg, args=args, o1=o1: epg.E(*args, order1=o1 or False))

    def add_S():
        k = int(rng.choice([1, 1, 1, -1]))
        tuples.append(("S", k))
        build.append(lambda epg, k=k: epg.S(k))

    add_T()
    for _ in range(nblocks):
        shape = rng.random()
        if rng.random() < 0.7:
            add_S()
        if shape < 0.75:
            add_E()
        add_T()
        if rng.random() < 0.7:
            add_S()
        if shape > 0.2:
            add_E()
        if rng.random() < 0.8:
            tuples.append(("ADC",))
            build.append(lambda epg: epg.ADC)
    tuples.append(("ADC",))
    build.append(lambda epg: epg.ADC)
    return tuples, (lambda epg: [b(epg) for b in build]), ["magnitude"] + names


def jac_plain_ops(T2):
    """derivatives across SPOILER / RESET / PD: the reference updates the state only (plain Operators)"""
    t_o1, e_o1 = {"alpha": {"alpha": 1}}, {"T2": {"T2": 1}}
    tuples = [("T", 30, 0, {"order1": t_o1}), ("E", 5, 1000, T2, 0, {"order1": e_o1}), ("ADC",), ("SPOILER",), ("ADC",),
              ("T", 20, 0, {"order1": t_o1}), ("S", 1), ("E", 5, 1000, T2, 0, {"order1": e_o1}), ("ADC",), ("ADC", "Z0"),
              ("RESET",), ("ADC", "Z0"), ("T", 50, 90, {"order1": t_o1}), ("ADC",), ("PD", 0.7, True), ("T", 40, 0), ("ADC",)]

    def ops(epg):
        e = epg.E(5, 1000, T2, order1=["T2"])
        return [epg.T(30, 0, order1="alpha"), e, epg.ADC, epg.SPOILER, epg.ADC, epg.T(20, 0, order1="alpha"), epg.S(1), e,
                epg.ADC, epg.Adc("Z0"), epg.RESET, epg.Adc("Z0"), epg.T(50, 90, order1="alpha"), epg.ADC,
                epg.PD(0.7), epg.T(40, 0), epg.ADC]

    return tuples, ops, ["magnitude", "alpha", "T2"]


# ------------------------------------------------------------------ second-order derivatives (g13)
def hessian_cases(epg):
    """[(name, sequence, probe list, simulate options)] built with `epg` (the reference or the product)"""
    cases = []
    # the reference's own tutorial shape (examples/differentiation/tutorial.py:84-93)
    exc = epg.T(90, 90)
    inv = epg.T(150, 0, order2="alpha")
    rlx = epg.E(4.5, 1400, 30, order2="T2")
    shift = epg.S(1)
    seq = [exc] + [shift, rlx, inv, shift, rlx, epg.ADC] * 6
    cases.append(("tutorial", seq, [epg.Hessian(["alpha", "T2"]), epg.Jacobian(["magnitude", "alpha", "T2"]),
                                    epg.Hessian(["magnitude", "alpha"], ["T2", "alpha"])], {}))
    # a (T2, g) grid, selected cross derivatives, Z0 probe, coefficients on a shared variable
    T2 = np.array([40.0, 90.0])
    g = np.array([[0.0, 0.02, -0.01]])
    rf = epg.T(35, 20, order1={"b1": {"alpha": 35.0}, "ph": {"phi": 1.0}}, order2=[("b1", "b1"), ("b1", "ph"), ("b1", "T2")])
    rl = epg.E(6, 900, T2, g, order1=["T2", "g"], order2=[("T2", "T2"), ("T2", "g"), ("b1", "T2")])
    seq2 = [rf, rl, epg.ADC, epg.S(1), rf, epg.S(1), rl, epg.ADC, epg.S(-1), rf, rl, epg.ADC]
    cases.append(("grid", seq2, [epg.Hessian(["b1", "ph", "T2", "g"]), epg.Hessian(["T2", "b1"], ["magnitude", "g", "T2"], probe="Z0")],
                  {"max_nstate": 5}))
    return cases


def random_train_blocks(rng, grid, nblocks=4):
    """[(block of operator tuples, repetitions)]: short random blocks of T / E / S(+-1) / probes that a sequence repeats
    with the SAME operator objects (echo trains: the state-resident kernels fold such runs); phases often multiples
    of 90 degrees (rotations about x / y: shorter chains), relaxation mostly without precession"""
    nd = len(grid)

    def param(lo, hi):
        pattern = rng.integers(0, 3)
        if pattern == 0:
            return float(rng.uniform(lo, hi))
        if pattern == 1:
            ax = int(rng.integers(0, nd))
            shape = [1] * (ax + 1)
            shape[ax] = grid[ax]
            return rng.uniform(lo, hi, shape)
        return rng.uniform(lo, hi, grid)

    def phase():
        return float(rng.choice([0.0, 90.0, 180.0, 270.0, -90.0])) if rng.random() < 0.7 else float(rng.uniform(-180, 180))

    blocks = []
    for _ in range(nblocks):
        blk = []
        for _ in range(int(rng.integers(1, 6))):
            r = rng.random()
            if r < 0.3:
                blk.append(("T", param(5, 175), phase()))
            elif r < 0.55:
                blk.append(("E", param(1, 20), param(200, 3000), param(20, 300), 0 if rng.random() < 0.8 else param(-0.05, 0.05)))
            elif r < 0.8:
                blk.append(("S", int(rng.choice([1, 1, 1, 1, -1]))))
            elif r < 0.97:
                blk.append(("ADC", "F0" if rng.random() < 0.85 else "Z0"))
            elif r < 0.98:
                blk.append(("SPOILER",))
            elif r < 0.99:
                blk.append(("RESET",))
            else:
                blk.append(("PD", param(0.2, 1.5), bool(rng.random() < 0.5)))
        blocks.append((blk, int(rng.choice([1, 2, 3, 4, 7, 12, 25]))))
    return blocks


X64_ATOL = 1e-13


def same_bits(a, b, x64=False):
    """Identity of two kernel paths on the same input: bit for bit.  x64=True marks the one place where the library
    gives that up: at 64 orders per voxel the state-resident rows kernel evaluates rotations about x in the sum /
    difference form (u = A + B keeps its direction, v = A - B mixes with Z: 16 instead of 18 instructions per order
    slot, epgx_rows_kernels.hip.h EPGX_SUMDIFF) -- the same products in another association order -- so against
    any OTHER kernel (per-timestep, 16 / 32 orders per voxel, the state column of a derivative kernel) its results
    differ in the last bits: X64_ATOL absolute there (signals are O(1)); against the oracle the bar stays 1e-12"""
    a, b = np.asarray(a), np.asarray(b)
    if not x64:
        return bool(np.array_equal(a, b))
    return a.shape == b.shape and bool(np.all(np.abs(a - b) <= X64_ATOL))
