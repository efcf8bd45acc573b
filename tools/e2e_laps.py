import os, sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from epgpy_amd import epg, _lib, functions, plan
n = 1024
T1 = np.linspace(200, 3000, n)[:, None]; T2 = np.linspace(20, 300, n)[None, :]
exc, rfc = epg.T(90, 90), epg.T(120, 0); rlx = epg.E(5.0, T1, T2); sh = epg.S(1, duration=5.0)
seq = [exc] + [[sh, rlx, rfc, sh, rlx, epg.ADC]] * 20
laps = []
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); laps.append((name, 1e3 * (time.perf_counter() - t0))); return r
    setattr(obj, name, g)
wrap(functions, "compile_sequence"); wrap(plan.Encoder, "device_plan"); wrap(plan.Encoder, "arrays")
wrap(_lib.DeviceBuffer, "download"); wrap(_lib, "host_empty"); wrap(_lib, "run")
orig_init = _lib.DeviceBuffer.__init__
def init(self, *a, **k):
    t0 = time.perf_counter(); orig_init(self, *a, **k); laps.append(("DeviceBuffer", 1e3 * (time.perf_counter() - t0)))
_lib.DeviceBuffer.__init__ = init
keep = []
for i in range(5):
    laps.clear()
    t0 = time.perf_counter(); sig = epg.simulate(seq, max_nstate=63); t1 = time.perf_counter()
    if i < 3: keep.append(sig)
    print(f"#{i}: {1e3*(t1-t0):.1f} ms :: " + ", ".join(f"{k} {v:.1f}" for k, v in laps), flush=True)
    t0 = time.perf_counter(); del sig; print(f"   del result: {1e3*(time.perf_counter()-t0):.1f} ms")
