"""scratch: cost per fused record kind (resident kernel, 1024 x 1024 voxels): slope of time vs count"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epgpy_amd import epg, _lib, functions
n = 1024
T1 = np.linspace(200, 3000, n)[:, None]; T2 = np.linspace(20, 300, n)[None, :]
exc, rfc, rlx, sh = epg.T(90, 90), epg.T(120, 0), epg.E(5.0, T1, T2), epg.S(1)
rfp = epg.T(120, 30)
ctx = _lib.get_context(None)
kinds = {
    "TX0 (E.T.E fused)": (lambda N: [exc] + [rlx, rfc, rlx] * N + [epg.ADC], True),
    "T0 (phase)": (lambda N: [exc] + [rlx, rfp, rlx] * N + [epg.ADC], True),
    "TX": (lambda N: [exc] + [rfc, epg.NULL] * N + [epg.ADC], False),
    "TX+ER": (lambda N: [exc] + [rfc, rlx] * N + [epg.ADC], False),
    "S": (lambda N: [exc] + [sh] * N + [epg.ADC], False),
    "S+ADC": (lambda N: [exc] + [sh, epg.ADC] * N, False),
    "TX0+S+ADC": (lambda N: [exc] + [rlx, rfc, rlx, sh, epg.ADC] * N, True),
    "TX+ER+S+ADC": (lambda N: [exc] + [rfc, rlx, sh, epg.ADC] * N, False),
    "ER+S": (lambda N: [exc] + [rlx, sh] * N + [epg.ADC], False),
}
for name, (build, fuse) in kinds.items():
    ts = []
    for N in (20, 60):
        enc, _, _ = functions.compile_sequence(build(N), None, options={"max_nstate": 63}, fuse=fuse)
        plan = enc.device_plan(ctx, 64)
        sig = _lib.DeviceBuffer(ctx, 16 * enc.n_adc * enc.nvox)
        run = lambda: _lib.run(ctx, plan, 0, plan.n_ops, 0, enc.nvox, None, None, 64, sig.ptr.value, enc.nvox, 0)
        run(); ctx.synchronize(); ctx.timer_start()
        for _ in range(5): run()
        ts.append(ctx.timer_stop() / 5)
        sig.free()
    us = (ts[1] - ts[0]) / 40 * 1e3
    print(json.dumps({"record": name, "us_per_record": round(us, 2), "cycles_per_wave_record_at_2.3GHz": round(us * 1e-6 * 2.3e9 / 1024, 1),
                      "t20_ms": round(ts[0], 3), "t60_ms": round(ts[1], 3)}), flush=True)
