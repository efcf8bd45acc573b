import os, sys, numpy as np, time
sys.path.insert(0, "/root/repo")
os.environ.setdefault("EPGX_ROWS", "1")
import epgpy_amd as epg
rng = np.random.default_rng(0)
T1 = rng.uniform(200, 3000, (50, 1)); T2 = rng.uniform(20, 300, (1, 37)); B1 = rng.uniform(0.7, 1.2, (50, 37))
def mse(g=0.0):
    exc = epg.T(90 * B1, 90); rfc = epg.T(120 * B1, 17.0 if g else 0); rlx = epg.E(5, T1, T2, g); shift = epg.S(1)
    return [exc] + [shift, rlx, rfc, shift, rlx, epg.ADC] * 20
for g in (0.0, 0.013):
    for fuse in (True, False):
        seq = mse(g)
        a = epg.simulate(seq, max_nstate=63, fuse=fuse)
        print("g", g, "fuse", fuse, a.shape, np.abs(a).max())
        np.save(f"gpurun_out/rows_{g}_{fuse}_{os.environ['EPGX_ROWS']}.npy", a)
