"""Latency of the operator-by-operator path (`sm = op(sm)`, what callbacks and the reference's own tests use): one plan
compilation + one launch per call.      python tools/op_latency_probe.py"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epgpy_amd import epg  # noqa: E402

T2 = np.linspace(20, 300, 4096)
ops = [epg.T(30, 90), epg.E(5, 1000, T2), epg.S(1)]
sm = epg.StateMatrix(shape=(4096,), max_nstate=20)
for op in ops * 5:
    sm = op(sm, inplace=True)
sm.F0
for name, op in zip("TES", ops):
    t0 = time.perf_counter()
    for _ in range(300):
        sm = op(sm, inplace=True)
    sm.F0
    print(name, "us per call", round(1e6 * (time.perf_counter() - t0) / 300, 1), flush=True)
pr = cProfile.Profile()
pr.enable()
for _ in range(100):
    for op in ops:
        sm = op(sm, inplace=True)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(16)
