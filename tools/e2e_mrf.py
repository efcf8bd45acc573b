"""What a caller waits for at BASELINE config 3: one epg.simulate() of the 1000-TR MRF train over 100^3 voxels -- with the
16 GB signal downloaded into a NumPy array, kept on the device (out="device"), or reduced on the device (Adc(reduce=...)).

    python tools/e2e_mrf.py [workload]        (GPU box; needs ~40 GB of host memory for the downloaded variant)
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epgpy_amd import epg, workloads as wl  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "mrf_100"
seq, _, n_adc, opts = wl.build(epg, name)
nvox = int(np.prod(wl.GRIDS[name][1]))


def lap(fn, n=2):
    global laps
    laps = []
    for _ in range(n):
        t0 = time.perf_counter()
        out = fn()
        laps.append(round(time.perf_counter() - t0, 4))
    return min(laps), out


t_dev, sig = lap(lambda: epg.simulate(seq, out="device", **opts))
print(json.dumps({"variant": 'out="device"', "s": round(t_dev, 4), "laps": laps, "TR_voxels_per_s": n_adc * nvox / t_dev}), flush=True)
del sig
t_host, res = lap(lambda: epg.simulate(seq, **opts))
print(json.dumps({"variant": "NumPy result", "s": round(t_host, 4), "laps": laps, "GB": round(res.nbytes / 1e9, 2), "GB_per_s": round(res.nbytes / 1e9 / t_host, 2),
                  "TR_voxels_per_s": n_adc * nvox / t_host}), flush=True)
