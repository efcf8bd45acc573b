"""Sequence definitions shared by the tests: each builder returns BOTH the oracle's tuple
description and (lazily) the product's operator list, from the same parameters, so the
GPU path and the oracle always see identical inputs."""
import numpy as np


def mse_tuples(T1, T2, B1=1.0, FA=120.0, ESP=10.0, necho=20, g=0):
    blk = [("S", 1), ("E", ESP / 2, T1, T2, g), ("T", FA * B1, 0), ("S", 1), ("E", ESP / 2, T1, T2, g), ("ADC",)]
    return [("T", 90 * B1, 90)] + blk * necho


def mse_ops(epg, T1, T2, B1=1.0, FA=120.0, ESP=10.0, necho=20, g=0):
    exc, rfc = epg.T(90 * B1, 90), epg.T(FA * B1, 0)
    rlx = epg.E(ESP / 2, T1, T2, g)
    sh = epg.S(1, duration=ESP / 2)
    return [exc] + [[sh, rlx, rfc, sh, rlx, epg.ADC]] * necho


def mrf_trains(ntr, seed=0):
    rng = np.random.default_rng(seed)
    u, v = rng.random(ntr), rng.random(ntr)
    i = np.arange(ntr)
    alpha = 10 + 50 * np.abs(np.sin(np.pi * i / 250)) * (0.6 + 0.4 * u)
    TR = 11 + 5 * v
    return alpha, TR


def mrf_tuples(T1, T2, B1, alpha, TR, TE=3.0):
    seq = [("T", 180 * B1, 90), ("E", 20, T1, T2, 0)]
    for a, tr in zip(alpha, TR):
        seq += [("T", a * B1, 90), ("E", TE, T1, T2, 0), ("ADC",), ("E", tr - TE, T1, T2, 0), ("S", 1)]
    return seq


def mrf_ops(epg, T1, T2, B1, alpha, TR, TE=3.0):
    seq = [epg.T(180 * B1, 90), epg.E(20, T1, T2)]
    rlx1 = epg.E(TE, T1, T2)
    sh = epg.S(1)
    for a, tr in zip(alpha, TR):
        seq += [epg.T(a * B1, 90), rlx1, epg.ADC, epg.E(tr - TE, T1, T2), sh]
    return seq


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
            ops.append(epg.S(t[1]))
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
