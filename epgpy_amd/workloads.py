"""The synthetic workloads of BASELINE.json / SURVEY.md section 8(d), as product-side operator lists.

Used by `bench.py`, `__graft_entry__.smoke()`, the tools and the tests; the oracle-side (tuple)
descriptions of the same workloads live in `oracle/workloads.py`, built from the same numbers.

  C1 / C2  multi-spin-echo train   `mse_sequence`   (reference: README.md:52-76)
  C3 / C4  MRF-type SSFP train     `mrf_sequence`   (shape of examples/differentiation/optim_mrf.py:78-82)
  C5       PGSE, 3-D shift + D     `pgse_sequence`  
"""
import numpy as np

MRF_NTR = 1000
GRIDS = {
    # name: (kind, grid)
    "mse_1024": ("mse", (1024, 1024)),    # C2-L: the >= 1e6-voxel target of north_star
    "mse_256": ("mse", (256, 256)),       # C2: BASELINE.json configs[1]
    "mrf_100": ("mrf", (100, 100, 100)),  # C3 / C4: 1000-TR variable-FA SSFP over a (T1, T2, B1) grid
    "mrf_32": ("mrf", (32, 32, 32)),
    "pgse_512": ("pgse", (512, 512)),     # C5
}


def mse_sequence(epg, T1, T2, B1=1.0, FA=120.0, ESP=10.0, necho=20, g=0):
    """[T(90 B1, 90)] + [S(1), E(ESP/2), T(FA B1, 0), S(1), E(ESP/2), ADC] x necho"""
    exc, rfc = epg.T(90 * B1, 90), epg.T(FA * B1, 0)
    rlx = epg.E(ESP / 2, T1, T2, g)
    sh = epg.S(1, duration=ESP / 2)
    return [exc] + [[sh, rlx, rfc, sh, rlx, epg.ADC]] * necho


def mrf_trains(ntr=MRF_NTR, seed=0):
    """flip angles (degrees) and repetition times (ms) of the MRF train, SURVEY.md 8d:
    alpha_i = 10 + 50 |sin(pi i / 250)| (0.6 + 0.4 u_i),  TR_i = 11 + 5 v_i,  u, v ~ default_rng(seed)"""
    rng = np.random.default_rng(seed)
    u, v = rng.random(ntr), rng.random(ntr)
    i = np.arange(ntr)
    alpha = 10 + 50 * np.abs(np.sin(np.pi * i / 250)) * (0.6 + 0.4 * u)
    TR = 11 + 5 * v
    return alpha, TR


def mrf_sequence(epg, T1, T2, B1, alpha, TR, TE=3.0):
    """[T(180 B1, 90), E(20)] + [T(a_i B1, 90), E(TE), ADC, E(TR_i - TE), S(1)] for every repetition"""
    seq = [epg.T(180 * B1, 90), epg.E(20, T1, T2)]
    rlx1 = epg.E(TE, T1, T2)
    sh = epg.S(1)
    for a, tr in zip(alpha, TR):
        seq += [epg.T(a * B1, 90), rlx1, epg.ADC, epg.E(tr - TE, T1, T2), sh]
    return seq


PGSE_KVALUE = [2e4, 1e4, 5e3]   # rad/m per unit shift along x, y, z


def pgse_sequence(epg, T2, ADC, T1=1000.0, k=(1, 1, 1)):
    """pulsed-gradient spin echo with a 3-D k-space shift; `ADC` (apparent diffusion coefficient, mm^2/s)
    may be an array over a grid axis (the reference's D rejects that: diffusion.py:166-169)"""
    k = list(k)
    return [epg.T(90, 90), epg.S(k), epg.D(10, ADC, k=k, field=True), epg.E(10, T1, T2),
            epg.D(20, ADC, field=True), epg.E(20, T1, T2), epg.T(180, 0),
            epg.D(20, ADC, field=True), epg.E(20, T1, T2), epg.S(k), epg.D(10, ADC, k=k, field=True),
            epg.E(10, T1, T2), epg.ADC]


def grid_parameters(name, rows=None):
    """parameter arrays of workload `name` (see GRIDS); `rows` = (first, count, total) selects rows of the
    FIRST axis out of a `total`-row axis over the same range (weak scaling: rank r of N owns rows
    [r n1, (r+1) n1) of an N n1-row axis)"""
    kind, grid = GRIDS[name]
    first, count, total = rows if rows is not None else (0, grid[0], grid[0])
    sl = slice(first, first + count)
    if kind == "mse":
        T1 = np.linspace(200, 3000, total)[sl][:, None]
        T2 = np.linspace(20, 300, grid[1])[None, :]
        return T1, T2
    if kind == "mrf":
        T1 = np.linspace(300, 3000, total)[sl][:, None, None]
        T2 = np.linspace(20, 300, grid[1])[None, :, None]
        B1 = np.linspace(0.7, 1.3, grid[2])[None, None, :]
        return T1, T2, B1
    T2 = np.linspace(20, 300, total)[sl][:, None]
    ADC = np.linspace(1e-4, 3e-3, grid[1])[None, :]
    return T2, ADC


def build(epg, name, rows=None):
    """(sequence, parameter arrays, number of ADC rows, simulate options) of workload `name`"""
    kind, _ = GRIDS[name]
    params = grid_parameters(name, rows)
    if kind == "mse":
        return mse_sequence(epg, *params), params, 20, {"max_nstate": 63}
    if kind == "mrf":
        alpha, TR = mrf_trains()
        return mrf_sequence(epg, *params, alpha, TR), params, MRF_NTR, {"max_nstate": 63}
    return pgse_sequence(epg, *params), params, 1, {"kvalue": PGSE_KVALUE}
