"""Latency of one epg.simulate() of BASELINE.json configs[0] (the README multi-spin-echo: 20 echoes, T2 = 30 / 40 / 50 ms), operators
prebuilt and rebuilt per call: what a caller of the smallest case waits for (the reference needs ~5 ms on one core).
    python tools/c1_latency.py          (GPU box)
"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epgpy_amd import epg, workloads as wl

seq = wl.mse_sequence(epg, 150.0, [30.0, 40.0, 50.0])
for _ in range(5):
    epg.simulate(seq)
laps = []
for _ in range(200):
    t0 = time.perf_counter(); epg.simulate(seq); laps.append(time.perf_counter() - t0)
laps2 = []
for _ in range(100):
    t0 = time.perf_counter(); epg.simulate(wl.mse_sequence(epg, 150.0, [30.0, 40.0, 50.0])); laps2.append(time.perf_counter() - t0)
print(json.dumps({"what": "README MSE, 3 voxels, 20 echoes", "simulate_ms_median_prebuilt_operators": round(1e3 * float(np.median(laps)), 3),
                  "simulate_ms_p95": round(1e3 * float(np.percentile(laps, 95)), 3),
                  "simulate_ms_median_operators_rebuilt_per_call": round(1e3 * float(np.median(laps2)), 3)}))
