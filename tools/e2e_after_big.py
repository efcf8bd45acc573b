import sys, time, gc, numpy as np
sys.path.insert(0, '.')
from epgpy_amd import epg, _lib, workloads as wl
def laps(tag):
    seq, _, n, opts = wl.build(epg, "mse_1024")
    res = epg.simulate(seq, **opts); res = epg.simulate(seq, **opts)
    ts=[]
    for _ in range(6):
        t=time.perf_counter(); res = epg.simulate(seq, **opts); ts.append(time.perf_counter()-t)
    print(tag, [round(x*1e3,2) for x in ts], "live pinned", dict(_lib._PinnedBlock.live), flush=True)
    del res
laps("fresh")
seq3, _, n3, opts3 = wl.build(epg, "mrf_100")
r = epg.simulate(seq3, out="device", **opts3); del r
laps("after C3 device")
t=time.perf_counter(); r = epg.simulate(seq3, **opts3); print("C3 host", round(time.perf_counter()-t,3), flush=True)
del r; gc.collect()
laps("after C3 host result")
_lib.get_context(0).release_cache()
laps("after release_cache")
