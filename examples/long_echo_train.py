"""Long echo trains whose state matrix is never bounded -- the reference's default (shift.py:86,98: every S adds an order): a
CPMG train of N echoes ends with 2 N + 1 phase states.  The kernels walk such a train the way the state matrix grows (1, 2, 4 ...
orders per lane while it is short); derivative states travel with it up to 1024 orders.

    python examples/long_echo_train.py [n] [necho]        # n x n (T1, T2) grid, default 128 x 128, 250 echoes
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from a checkout
from epgpy_amd import epg  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
necho = int(sys.argv[2]) if len(sys.argv) > 2 else 250
T1 = np.linspace(200, 3000, n)[:, None]
T2 = np.linspace(40, 300, n)[None, :]
exc = epg.T(90, 90)
rfc = epg.T(150, 0, order1={"fa": {"alpha": 1.0}})          # d/d(refocusing angle), per degree
relax = epg.E(2.5, T1, T2, order1=["T2"])                     # d/dT2
shift = epg.S(1, duration=2.5)
seq = [exc] + [[shift, relax, rfc, shift, relax, epg.ADC]] * necho

enc, _, _ = epg.compile_sequence(seq)
print(f"{necho} echoes: {enc.peak + 1} phase states per voxel -> capacity {enc.capacity(resident=True)}")
epg.simulate(seq)
t0 = time.perf_counter()
signal = epg.simulate(seq)
dt = time.perf_counter() - t0
print(f"signal {signal.shape} in {1e3 * dt:.1f} ms = {signal.size / dt:.3e} echo*voxels/s")

# the same train with its Jacobian: [echo, *grid, (magnitude, dS/dT2, dS/dfa)]
t0 = time.perf_counter()
jac = epg.simulate(seq, probe=epg.Jacobian(["magnitude", "T2", "fa"]))
dt = time.perf_counter() - t0
c = n // 2
print(f"Jacobian {jac.shape} in {1e3 * dt:.1f} ms; centre voxel, last echo: |S| = {abs(jac[-1, c, c, 0]):.4f}, "
      f"dS/dT2 = {jac[-1, c, c, 1].real:+.3e} per ms, dS/dfa = {jac[-1, c, c, 2].real:+.3e} per degree")
h = 1e-3                                                        # (a central difference on T2 for the centre voxel)
up = epg.simulate([exc] + [[shift, epg.E(2.5, T1[c, 0], T2[0, c] + h), epg.T(150, 0), shift, epg.E(2.5, T1[c, 0], T2[0, c] + h), epg.ADC]] * necho)
dn = epg.simulate([exc] + [[shift, epg.E(2.5, T1[c, 0], T2[0, c] - h), epg.T(150, 0), shift, epg.E(2.5, T1[c, 0], T2[0, c] - h), epg.ADC]] * necho)
print(f"central difference: {((up[-1] - dn[-1]) / (2 * h)).real.item():+.3e}")
