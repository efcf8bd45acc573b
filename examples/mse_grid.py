"""20-echo multi-spin-echo train over a (T1, T2) grid: the workload of BASELINE.json configs[1].

    python examples/mse_grid.py [n]        # n x n grid, default 256
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from a checkout
from epgpy_amd import epg  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T1 = np.linspace(200, 3000, n)[:, None]          # ms, axis 0 of the grid
T2 = np.linspace(20, 300, n)[None, :]            # ms, axis 1
exc, rfc = epg.T(90, 90), epg.T(120, 0)
relax = epg.E(5.0, T1, T2)                       # ESP / 2
shift = epg.S(1, duration=5.0)
seq = [exc] + [[shift, relax, rfc, shift, relax, epg.ADC]] * 20

epg.simulate(seq)                                # first call loads the library and the kernels
t0 = time.perf_counter()
times, signal = epg.simulate(seq, adc_time=True)
dt = time.perf_counter() - t0
print(f"grid {epg.getshape(seq)}, {len(times)} echoes at {times[0]:.0f} .. {times[-1]:.0f} ms: "
      f"{signal.shape} {signal.dtype} in {1e3 * dt:.1f} ms = {signal.size / dt:.3e} echo*voxels/s")
print("echo amplitudes of the centre voxel:", np.round(np.abs(signal[:5, n // 2, n // 2]), 4), "...")

# single-precision RECORDS (the simulation itself stays float64): half the bytes cross PCIe -- what a caller of a large grid waits for
epg.simulate(seq, dtype=np.complex64)             # (first call: a page-locked result block of that size is pinned)
t0 = time.perf_counter()
signal32 = epg.simulate(seq, dtype=np.complex64)
dt32 = time.perf_counter() - t0
print(f"dtype=complex64: {signal32.dtype} in {1e3 * dt32:.1f} ms; max relative difference to complex128: "
      f"{np.max(np.abs(signal32 - signal)) / np.max(np.abs(signal)):.1e}")
