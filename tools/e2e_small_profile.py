"""cProfile of one epg.simulate() call on a SMALL grid with a LONG sequence (1000-TR MRF over 32^3 voxels): where the host
time of a dictionary-chunk / fitting loop goes.      python tools/e2e_small_profile.py"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epgpy_amd import epg, workloads as wl  # noqa: E402

seq, _, n_adc, opts = wl.build(epg, "mrf_32")
for _ in range(3):
    t0 = time.perf_counter()
    res = epg.simulate(seq, **opts)
    print("simulate", round(1e3 * (time.perf_counter() - t0), 2), "ms", res.shape, flush=True)
pr = cProfile.Profile()
pr.enable()
res = epg.simulate(seq, **opts)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
