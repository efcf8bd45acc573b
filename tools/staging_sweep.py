"""Staging ring of the download engine (results in ordinary host memory): sweep block size / ring length / copy threads.
Every configuration runs in its own process (the knobs are read once).     python tools/staging_sweep.py [--c3]"""
import json
import os
import subprocess
import sys

CHILD = r'''
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(sys.argv[0]))) if False else ".")
from epgpy_amd import epg, _lib, workloads as wl
_lib.PINNED_MAX_BYTES = 0                      # every result is an ordinary array: the staged route
seq, _, n, opts = wl.build(epg, "mse_1024")
res = epg.simulate(seq, **opts); res = epg.simulate(seq, **opts)
held, laps = [], []
for _ in range(6):
    t = time.perf_counter(); held.append(epg.simulate(seq, **opts)); laps.append(time.perf_counter() - t)
out = {"c2l_kept_ms": round(1e3 * sorted(laps)[len(laps) // 2], 2), "c2l_min_ms": round(1e3 * min(laps), 2)}
del held, res
if "--c3" in sys.argv:
    seq3, _, n3, opts3 = wl.build(epg, "mrf_100")
    r = epg.simulate(seq3, out="device", **opts3); del r
    for _ in range(2):
        t = time.perf_counter(); r = epg.simulate(seq3, **opts3); dt = time.perf_counter() - t
        out.setdefault("c3_s", []).append(round(dt, 3)); del r
print(json.dumps(out))
'''

configs = [(64, 4, None), (32, 4, None), (32, 8, None), (16, 8, None), (64, 6, None), (64, 4, 8), (64, 4, 16), (32, 8, 16), (128, 3, None)]
for mb, ring, threads in configs:
    env = dict(os.environ, EPGX_STAGE_MB=str(mb), EPGX_STAGES=str(ring))
    if threads:
        env["EPGX_COPY_THREADS"] = str(threads)
    res = subprocess.run([sys.executable, "-c", CHILD] + sys.argv[1:], env=env, capture_output=True, text=True)
    line = res.stdout.strip().splitlines()[-1] if res.stdout.strip() else res.stderr[-300:]
    print(json.dumps({"stage_mb": mb, "ring": ring, "threads": threads or "default"}), line, flush=True)
