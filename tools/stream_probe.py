"""per-timestep (stream) kernel: time in-place launches of small plans over 1M voxels"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from epgpy_amd import epg, _lib, plan as _plan

n = 1024
T1 = np.linspace(200, 3000, n)[:, None]
T2 = np.linspace(20, 300, n)[None, :]
ctx = _lib.get_context()
cases = {
    "E uniform": [epg.E(5.0, 1000.0, 80.0)],
    "E per-voxel": [epg.E(5.0, T1, T2)],
    "E per-voxel + S": [epg.E(5.0, T1, T2), epg.S(1)],
    "echo (S E T S E ADC)": [epg.S(1), epg.E(5.0, T1, T2), epg.T(120, 0), epg.S(1), epg.E(5.0, T1, T2), epg.ADC],
}
state = _lib.DeviceState(ctx, n * n, 64)
sig = _lib.DeviceBuffer(ctx, 16 * n * n)
for name, ops in cases.items():
    enc = _plan.Encoder((n, n), options={"max_nstate": 63})
    for op in ops:
        if isinstance(op, epg.Probe):
            enc.add_adc(0)
        else:
            op._encode(enc)
    plan = enc.device_plan(ctx)
    for _ in range(3):
        _lib.run(ctx, plan, 0, plan.n_ops, 0, n * n, state, state, 64, sig.ptr.value, n * n, 0)
    ctx.timer_start()
    reps = 20
    for _ in range(reps):
        _lib.run(ctx, plan, 0, plan.n_ops, 0, n * n, state, state, 64, sig.ptr.value, n * n, 0)
    ms = ctx.timer_stop() / reps
    print(f"{name:24s} {ms:.3f} ms  {2 * n * n * 3072 / ms / 1e6:.0f} GB/s")
