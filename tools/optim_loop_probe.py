"""Host cost of an optimisation loop in the style of the reference's examples/differentiation/optim_mrf.py: every iteration
builds a fresh 400-TR train (new flip angles -> new operator objects) and asks for the Jacobian on a handful of voxels.
The kernel is microseconds here; what the loop waits for is operator construction + plan compilation.

    python tools/optim_loop_probe.py [ntr] [nvox]
"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epgpy_amd import epg  # noqa: E402

ntr = int(sys.argv[1]) if len(sys.argv) > 1 else 400
nvox = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rng = np.random.default_rng(0)
T1, T2, B1 = rng.uniform(300, 3000, nvox), rng.uniform(20, 300, nvox), rng.uniform(0.7, 1.3, nvox)
probe = epg.Jacobian(["magnitude", "T1", "T2", "alpha"])


def iteration(alpha, TR):
    e1, shift = epg.E(3.0, T1, T2, order1=["T1", "T2"]), epg.S(1)
    seq = [epg.T(180 * B1, 90), epg.E(20.0, T1, T2, order1=["T1", "T2"])]
    for i, (a, tr) in enumerate(zip(alpha, TR)):
        seq += [epg.T(a * B1, 90, order1={"alpha": {"alpha": 1.0}}), e1, epg.ADC, epg.E(tr - 3.0, T1, T2, order1=["T1", "T2"]), shift]
    return epg.simulate(seq, probe=probe, max_nstate=10)


laps = []
for it in range(6):
    alpha, TR = rng.uniform(10, 60, ntr), rng.uniform(11, 16, ntr)
    t0 = time.perf_counter()
    jac = iteration(alpha, TR)
    laps.append(time.perf_counter() - t0)
print(f"{ntr} TR x {nvox} voxels, Jacobian {jac.shape}: per iteration ms", [round(1e3 * x, 1) for x in laps], flush=True)
pr = cProfile.Profile()
pr.enable()
iteration(rng.uniform(10, 60, ntr), rng.uniform(11, 16, ntr))
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
