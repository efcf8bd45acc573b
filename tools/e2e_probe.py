"""End-to-end wall time of epg.simulate() (host call -> NumPy result) with a breakdown.
    python tools/e2e_probe.py [--n 1024]
"""
import argparse, os, sys, time, cProfile, pstats
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epgpy_amd import epg  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1024)
ap.add_argument("--profile", action="store_true")
args = ap.parse_args()
n = args.n
T1 = np.linspace(200, 3000, n)[:, None]
T2 = np.linspace(20, 300, n)[None, :]


def build():
    exc, rfc = epg.T(90, 90), epg.T(120, 0)
    rlx = epg.E(5.0, T1, T2)
    sh = epg.S(1, duration=5.0)
    return [exc] + [[sh, rlx, rfc, sh, rlx, epg.ADC]] * 20


t0 = time.perf_counter(); seq = build(); t1 = time.perf_counter()
print(f"build operators: {1e3*(t1-t0):.1f} ms")
for i in range(3):
    t0 = time.perf_counter(); sig = epg.simulate(seq, max_nstate=63); t1 = time.perf_counter()
    print(f"simulate #{i}: {1e3*(t1-t0):.1f} ms  -> {sig.shape} {sig.dtype}  {20*n*n/(t1-t0):.3e} echo*voxels/s")
if args.profile:
    pr = cProfile.Profile(); pr.enable(); epg.simulate(seq, max_nstate=63); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)

if os.environ.get("E2E_MRF"):
    from epgpy_amd import workloads as sq
    m = int(os.environ["E2E_MRF"])
    T1 = np.linspace(300, 3000, m)[:, None, None]
    T2 = np.linspace(20, 300, m)[None, :, None]
    B1 = np.linspace(0.7, 1.3, m)[None, None, :]
    alpha, TR = sq.mrf_trains(1000)
    t0 = time.perf_counter(); seq = sq.mrf_sequence(epg, T1, T2, B1, alpha, TR); t1 = time.perf_counter()
    print(f"MRF {m}^3 x 1000 TR: build operators {t1-t0:.2f} s", flush=True)
    for i in range(2):
        t0 = time.perf_counter(); sig = epg.simulate(seq, max_nstate=63); t1 = time.perf_counter()
        print(f"simulate #{i}: {t1-t0:.3f} s -> {sig.shape}, {sig.nbytes/1e9:.1f} GB, {1000*m**3/(t1-t0):.3e} echo*voxels/s", flush=True)
        del sig

print("fresh operators every call (a fitting loop that changes the tissue grid):")
for i in range(3):
    t0 = time.perf_counter(); seq2 = build(); t1 = time.perf_counter(); sig = epg.simulate(seq2, max_nstate=63); t2 = time.perf_counter()
    print(f"  build {1e3*(t1-t0):.1f} ms + simulate {1e3*(t2-t1):.1f} ms")
