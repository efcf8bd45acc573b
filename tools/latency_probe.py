"""small-problem latencies: README example through simulate(), and operator-by-operator calls"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epgpy_amd import epg

seq = [epg.T(90, 90)] + [[epg.S(1, duration=5), epg.E(5, 150, [30, 40, 50]), epg.T(120, 0),
                          epg.S(1, duration=5), epg.E(5, 150, [30, 40, 50]), epg.ADC]] * 20
epg.simulate(seq)
t0 = time.perf_counter()
for _ in range(20):
    epg.simulate(seq)
print(f"README MSE (3 voxels, 121 operators): simulate() {1e3 * (time.perf_counter() - t0) / 20:.2f} ms per call")
ops = epg.flatten_sequence(seq)
sm = epg.StateMatrix(shape=(3,))
for op in ops[:10]:
    sm = op(sm)
t0 = time.perf_counter()
sm = epg.StateMatrix(shape=(3,))
n = 0
for op in ops:
    if not isinstance(op, epg.Probe):
        sm = op(sm)
        n += 1
dt = time.perf_counter() - t0
print(f"operator by operator: {1e6 * dt / n:.0f} us per op(sm) call (copying), F0 = {np.asarray(sm.F0)[:1]}")
t0 = time.perf_counter()
sm = epg.StateMatrix(shape=(3,))
for op in ops:
    if not isinstance(op, epg.Probe):
        sm = op(sm, inplace=True)
dt = time.perf_counter() - t0
print(f"operator by operator, inplace: {1e6 * dt / n:.0f} us per call")
