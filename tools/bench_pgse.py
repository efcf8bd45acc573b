"""BASELINE config 5: PGSE (diffusion operator D, 3-D k-space shift) over a 512 x 512 (T2, ADC) grid.
    python tools/bench_pgse.py [--n 512]
"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epgpy_amd import epg, _lib, functions

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=512)
args = ap.parse_args()
n = args.n
T1, kvalue, k1 = 1000.0, [2e4, 1e4, 5e3], [1, 1, 1]
T2g = np.linspace(20, 300, n)[:, None]
ADCg = np.linspace(1e-4, 3e-3, n)[None, :]
seq = [epg.T(90, 90), epg.S(k1), epg.D(10, ADCg, k=k1, field=True), epg.E(10, T1, T2g),
       epg.D(20, ADCg, field=True), epg.E(20, T1, T2g), epg.T(180, 0),
       epg.D(20, ADCg, field=True), epg.E(20, T1, T2g), epg.S(k1), epg.D(10, ADCg, k=k1, field=True),
       epg.E(10, T1, T2g), epg.ADC]
for i in range(3):
    t0 = time.perf_counter(); sig = epg.simulate(seq, kvalue=kvalue); t1 = time.perf_counter()
    print(f"simulate #{i}: {1e3 * (t1 - t0):.2f} ms -> {sig.shape}", flush=True)
ctx = _lib.get_context(None)
for K in (16, 64):        # 16: four voxels per wavefront (rows_kernel, R = 1), 64: one wavefront per voxel (run_kernel)
    enc, _, _ = functions.compile_sequence(seq, None, options={"kvalue": kvalue})
    plan = enc.device_plan(ctx, K)
    buf = _lib.DeviceBuffer(ctx, 16 * enc.n_adc * enc.nvox)
    run = lambda: _lib.run(ctx, plan, 0, plan.n_ops, 0, enc.nvox, None, None, K, buf.ptr.value, enc.nvox, 0)
    run(); ctx.synchronize(); ctx.timer_start()
    for _ in range(20): run()
    ms = ctx.timer_stop() / 20
    print(json.dumps({"workload": f"PGSE {n}x{n} (T2, ADC), 3-D shift", "K": K, "kernel_ms": round(ms, 4),
                      "voxels_per_s": enc.nvox / ms * 1e3}))
