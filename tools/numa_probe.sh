#!/bin/bash
# Host topology of the GPU box and what it does to the staged download (results in ordinary host memory):
# the copy threads next to the staging blocks (default) or wherever the scheduler puts them (EPGX_COPY_AFFINITY=off).
lscpu | grep -i -E "numa|socket|model name|^CPU\(s\)"
for n in /sys/devices/system/node/node*; do echo "$n: $(cat $n/cpulist)"; done
for d in /sys/class/drm/card*/device; do echo "$d numa_node=$(cat $d/numa_node 2>/dev/null)"; done
cat /sys/fs/cgroup/cpu.max 2>/dev/null
cat /sys/kernel/mm/transparent_hugepage/enabled 2>/dev/null
python - <<'PY'
import os; print("affinity cpus:", len(os.sched_getaffinity(0)))
PY
for aff in auto off auto off auto off; do
  EPGX_COPY_AFFINITY=$aff EPGX_TRACE=1 python - <<'PY' 2>&1 | grep -E "staging ring|kept"
import os, sys, time
sys.path.insert(0, ".")
from epgpy_amd import epg, _lib, workloads as wl
_lib.PINNED_MAX_BYTES = 0
seq, _, n, opts = wl.build(epg, "mse_1024")
res = epg.simulate(seq, **opts); res = epg.simulate(seq, **opts)
held, laps = [], []
for _ in range(6):
    t = time.perf_counter(); held.append(epg.simulate(seq, **opts)); laps.append(time.perf_counter() - t)
print("kept", os.environ.get("EPGX_COPY_AFFINITY"), [round(1e3 * x, 2) for x in laps], flush=True)
PY
done
