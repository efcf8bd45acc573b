import sys, time, os, numpy as np
sys.path.insert(0, '.')
from epgpy_amd import epg, _lib, workloads as wl
import cProfile, pstats
seq, _, n, opts = wl.build(epg, "mse_1024")
for _ in range(4): res = epg.simulate(seq, **opts)
t=time.perf_counter(); res = epg.simulate(seq, **opts); print("before", round((time.perf_counter()-t)*1e3,2))
_lib.get_context(0).release_cache()
for _ in range(3): res = epg.simulate(seq, **opts)
pr=cProfile.Profile(); pr.enable()
t=time.perf_counter(); res = epg.simulate(seq, **opts); print("after", round((time.perf_counter()-t)*1e3,2))
pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(8)
