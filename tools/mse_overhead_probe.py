"""Fixed cost per wave of the state-resident MSE launch and the cost per echo: time(necho) at 1024 x 1024 voxels, 64 orders of
capacity.  With the growing kernel (default; EPGX_GROW=0: all 64 orders at every echo) echoes 1 - 7 run with one order per lane,
8 - 15 with two, the rest with four: the differences between neighbouring points are the per-echo costs of the phases.
    python tools/mse_overhead_probe.py [necho ...]         (GPU box)
"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epgpy_amd import epg, _lib, functions, workloads as wl

T1, T2 = wl.grid_parameters("mse_1024")
ctx = _lib.get_context(None)
pts = []
for necho in ([int(a) for a in sys.argv[1:]] or (1, 2, 5, 10, 20, 30)):
    seq = wl.mse_sequence(epg, T1, T2, necho=necho)
    enc, _, _ = functions.compile_sequence(seq, None, options={"max_nstate": 63})
    plan = enc.device_plan(ctx, 64)
    sig = _lib.DeviceBuffer(ctx, 16 * enc.n_adc * enc.nvox)
    run = lambda: _lib.run(ctx, plan, 0, plan.n_ops, 0, enc.nvox, None, None, 64, sig.ptr.value, enc.nvox, 0)
    import time
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.025:      # (out of the idle clocks)
        run(); ctx.synchronize()
    ctx.timer_start()
    for _ in range(20): run()
    ms = ctx.timer_stop() / 20
    pts.append((necho, ms))
    print(json.dumps({"necho": necho, "n_records": plan.n_ops, "ms_per_launch": round(ms, 4)}), flush=True)
x, y = np.array(pts).T
b, a = np.polyfit(x[2:], y[2:], 1)
print(json.dumps({"fit_over": "necho >= 5", "ms_fixed": round(a, 4), "ms_per_echo": round(b, 5), "fixed_share_at_20": round(a / (a + 20 * b), 3)}))
