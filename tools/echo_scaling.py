"""scratch: resident kernel time vs number of echoes, fused / unfused (fixed cost per wave vs cost per echo)"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epgpy_amd import epg, _lib, functions
n = 1024
T1 = np.linspace(200, 3000, n)[:, None]; T2 = np.linspace(20, 300, n)[None, :]
exc, rfc, rlx, sh = epg.T(90, 90), epg.T(120, 0), epg.E(5.0, T1, T2), epg.S(1)
ctx = _lib.get_context(None)
for fuse in (False, True):
    for necho in (1, 5, 10, 20, 40):
        seq = [exc] + [sh, rlx, rfc, sh, rlx, epg.ADC] * necho
        enc, _, _ = functions.compile_sequence(seq, None, options={"max_nstate": 63}, fuse=fuse)
        plan = enc.device_plan(ctx, 64)
        sig = _lib.DeviceBuffer(ctx, 16 * enc.n_adc * enc.nvox)
        run = lambda: _lib.run(ctx, plan, 0, plan.n_ops, 0, enc.nvox, None, None, 64, sig.ptr.value, enc.nvox, 0)
        run(); ctx.synchronize(); ctx.timer_start()
        for _ in range(5): run()
        ms = ctx.timer_stop() / 5
        print(json.dumps({"fuse": fuse, "necho": necho, "ms": round(ms, 4)}), flush=True)
        sig.free()
