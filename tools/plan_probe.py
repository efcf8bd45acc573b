"""scratch: where the 17 ms of epgx_plan_create go at 1024 x 1024"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epgpy_amd import epg, _lib, functions
n = 1024
T1 = np.linspace(200, 3000, n)[:, None]; T2 = np.linspace(20, 300, n)[None, :]
exc, rfc, rlx, sh = epg.T(90, 90), epg.T(120, 0), epg.E(5.0, T1, T2), epg.S(1)
seq = [exc] + [sh, rlx, rfc, sh, rlx, epg.ADC] * 20
ctx = _lib.get_context(None)
for i in range(3):
    t0 = time.perf_counter()
    enc, _, _ = functions.compile_sequence(seq, None, options={"max_nstate": 63})
    t1 = time.perf_counter()
    arrays = enc.arrays(64)
    t2 = time.perf_counter()
    plan = enc.device_plan(ctx, 64)
    t3 = time.perf_counter()
    buf = _lib.DeviceBuffer(ctx, arrays[3].nbytes + 1024)
    t4 = time.perf_counter()
    buf.upload(arrays[3])
    t5 = time.perf_counter()
    sig = _lib.DeviceBuffer(ctx, 336 << 20)
    t6 = time.perf_counter()
    sig.free(); buf.free()
    t7 = time.perf_counter()
    print(f"compile {1e3*(t1-t0):.2f}  arrays {1e3*(t2-t1):.2f}  device_plan {1e3*(t3-t2):.2f}  malloc32 {1e3*(t4-t3):.2f}  upload32 {1e3*(t5-t4):.2f} malloc336 {1e3*(t6-t5):.2f} free {1e3*(t7-t6):.2f}")
import cProfile, pstats
seq2 = seq
pr = cProfile.Profile(); pr.enable(); epg.simulate(seq2, max_nstate=63); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(8)
