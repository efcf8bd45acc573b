import sys, time, gc, numpy as np
sys.path.insert(0, '.')
from epgpy_amd import epg, _lib, workloads as wl
ctx = _lib.get_context(0)
def d2h(tag):
    buf = _lib.DeviceBuffer(ctx, 336 << 20)
    host = _lib.pinned_empty(ctx, (336 << 20) // 16, np.complex128)
    ts=[]
    for _ in range(4):
        t=time.perf_counter(); buf.download(np.complex128, host.shape, out=host); ts.append(time.perf_counter()-t)
    print(tag, "D2H 336MB pinned ms", [round(x*1e3,2) for x in ts], flush=True)
    buf.free(); del host
def laps(tag):
    seq, _, n, opts = wl.build(epg, "mse_1024")
    res = epg.simulate(seq, **opts); res = epg.simulate(seq, **opts)
    ts=[]
    for _ in range(4):
        t=time.perf_counter(); res = epg.simulate(seq, **opts); ts.append(time.perf_counter()-t)
    print(tag, "simulate ms", [round(x*1e3,2) for x in ts], flush=True)
d2h("fresh"); laps("fresh")
big = _lib.DeviceBuffer(ctx, 16 << 30); big.free()
d2h("after 16GB alloc+free (cached)"); laps("cached")
ctx.release_cache()
d2h("after release_cache"); laps("after release")
big = _lib.DeviceBuffer(ctx, 16 << 30); big.free()
d2h("after re-alloc 16GB"); laps("after re-alloc")
