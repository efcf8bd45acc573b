"""Where the difference between the headline's wall time per step and the launch's HIP-event duration comes from:
the same 20 state-resident launches of the 1024 x 1024 MSE train, timed by the host clock (synchronize on both sides, as
bench.py's `timed`) after 0 / 3 / 20 / 100 warm-up launches, next to the event time of the same loop, a 200-launch loop, and
the host's enqueue time alone.
    python tools/headline_gap_probe.py          (GPU box)
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epgpy_amd import epg, _lib, functions, workloads as wl

T1, T2 = wl.grid_parameters("mse_1024")
ctx = _lib.get_context(None)
seq = wl.mse_sequence(epg, T1, T2)
enc, _, _ = functions.compile_sequence(seq, None, options={"max_nstate": 63})
plan = enc.device_plan(ctx, 64)
sig = _lib.DeviceBuffer(ctx, 16 * enc.n_adc * enc.nvox)
run = lambda: _lib.run(ctx, plan, 0, plan.n_ops, 0, enc.nvox, None, None, 64, sig.ptr.value, enc.nvox, 0)


def wall(steps, warmup, idle_s=0.0):
    if idle_s:
        ctx.synchronize(); time.sleep(idle_s)
    for _ in range(warmup): run()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): run()
    t1 = time.perf_counter()
    ctx.synchronize()
    t2 = time.perf_counter()
    return 1e3 * (t2 - t0) / steps, 1e3 * (t1 - t0) / steps


def events(steps, warmup):
    for _ in range(warmup): run()
    ctx.synchronize(); ctx.timer_start()
    for _ in range(steps): run()
    return ctx.timer_stop() / steps


for idle in (0.0, 0.5):
    for warm in (0, 3, 20, 100):
        w, enq = wall(20, warm, idle)
        print(json.dumps({"clock": "host", "idle_before_s": idle, "warmup": warm, "steps": 20, "ms_per_step": round(w, 4),
                          "enqueue_ms_per_step": round(enq, 4)}), flush=True)
for steps in (1, 5, 20, 200):
    print(json.dumps({"clock": "host", "warmup": 20, "steps": steps, "ms_per_step": round(wall(steps, 20)[0], 4)}), flush=True)
    print(json.dumps({"clock": "hip events", "warmup": 20, "steps": steps, "ms_per_launch": round(events(steps, 20), 4)}), flush=True)
