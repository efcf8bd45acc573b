"""MRF-type variable-flip-angle train with first-order derivatives (the shape of the reference's
examples/differentiation/optim_mrf.py): signal and d(signal)/d(T1, T2, B1) for every voxel of a
(T1, T2, B1) grid, max_nstate = 10 -> four voxels per wavefront.

    python examples/mrf_jacobian.py [m] [ntr]     # m^3 voxels (default 32), ntr repetitions (default 400)
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from a checkout
from epgpy_amd import epg  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ntr = int(sys.argv[2]) if len(sys.argv) > 2 else 400
T1 = np.linspace(300, 3000, m)[:, None, None]
T2 = np.linspace(20, 300, m)[None, :, None]
B1 = np.linspace(0.7, 1.3, m)[None, None, :]
i = np.arange(ntr)
alpha = 10 + 50 * np.abs(np.sin(np.pi * i / 250))
TR, TE = 12.0, 3.0

def E(tau):
    return epg.E(tau, T1, T2, order1=["T1", "T2"])

seq = [epg.T(180 * B1, 90, order1={"B1": {"alpha": 180.0}}), E(20.0)]
e1, e2, shift = E(TE), E(TR - TE), epg.S(1)
for a in alpha:
    seq += [epg.T(a * B1, 90, order1={"B1": {"alpha": float(a)}}), e1, epg.ADC, e2, shift]

probe = epg.Jacobian(["magnitude", "T1", "T2", "B1"])
epg.simulate(seq[:12], probe=probe, max_nstate=10)
t0 = time.perf_counter()
jac = epg.simulate(seq, probe=probe, max_nstate=10)
dt = time.perf_counter() - t0
print(f"{m}^3 voxels x {ntr} TR: Jacobian {jac.shape} in {1e3 * dt:.0f} ms")
v = (m // 2,) * 3
print("centre voxel, last TR:  signal", np.round(jac[(-1,) + v + (0,)], 5), " d/dT1", f"{jac[(-1,) + v + (1,)]:.3e}",
      " d/dT2", f"{jac[(-1,) + v + (2,)]:.3e}", " d/dB1", np.round(jac[(-1,) + v + (3,)], 4))
