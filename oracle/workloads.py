"""Tuple descriptions (oracle side) of the synthetic workloads of BASELINE.json / SURVEY.md 8(d) --
TEST INFRASTRUCTURE ONLY (imported by tests/, __graft_entry__.smoke() and bench.py's checker /
cpu_baseline legs; never by the product package).

The product-side operator lists of the same workloads are built by `epgpy_amd/workloads.py` from the same
numbers; `tests/test_host.py::test_workload_definitions_agree` pins the two against each other.
"""
import numpy as np

MRF_NTR = 1000
PGSE_KVALUE = [2e4, 1e4, 5e3]


def mse_tuples(T1, T2, B1=1.0, FA=120.0, ESP=10.0, necho=20, g=0):
    """README.md:52-76: [T(90, 90)] + [S(1), E(ESP/2), T(FA, 0), S(1), E(ESP/2), ADC] x necho"""
    blk = [("S", 1), ("E", ESP / 2, T1, T2, g), ("T", FA * B1, 0), ("S", 1), ("E", ESP / 2, T1, T2, g), ("ADC",)]
    return [("T", 90 * B1, 90)] + blk * necho


def mrf_trains(ntr=MRF_NTR, seed=0):
    """alpha_i = 10 + 50 |sin(pi i / 250)| (0.6 + 0.4 u_i), TR_i = 11 + 5 v_i"""
    rng = np.random.default_rng(seed)
    u, v = rng.random(ntr), rng.random(ntr)
    i = np.arange(ntr)
    return 10 + 50 * np.abs(np.sin(np.pi * i / 250)) * (0.6 + 0.4 * u), 11 + 5 * v


def mrf_tuples(T1, T2, B1, alpha, TR, TE=3.0):
    seq = [("T", 180 * B1, 90), ("E", 20, T1, T2, 0)]
    for a, tr in zip(alpha, TR):
        seq += [("T", a * B1, 90), ("E", TE, T1, T2, 0), ("ADC",), ("E", tr - TE, T1, T2, 0), ("S", 1)]
    return seq


def pgse_tuples(T2, D, T1=1000.0, k=(1, 1, 1)):
    """the PGSE train of SURVEY.md 8d for ONE diffusion coefficient D (the reference's D operator takes a
    scalar or a tensor only, diffusion.py:166-169: an ADC axis is a Python loop over this)"""
    k = list(k)
    return [("T", 90, 90), ("S", k), ("D", 10, D, k), ("E", 10, T1, T2, 0), ("D", 20, D), ("E", 20, T1, T2, 0),
            ("T", 180, 0), ("D", 20, D), ("E", 20, T1, T2, 0), ("S", k), ("D", 10, D, k), ("E", 10, T1, T2, 0), ("ADC",)]
