import sys, time, os, numpy as np
sys.path.insert(0, '.')
from epgpy_amd import epg
n=1024
T1, T2 = np.linspace(200, 3000, n)[:, None], np.linspace(20, 300, n)[None, :]
exc = epg.T(90, 90, order1={"B1": {"alpha": 90}}); rfc = epg.T(120, 0, order1={"B1": {"alpha": 120}})
rlx = epg.E(5.0, T1, T2, order1=["T1", "T2"])
seq = [exc] + [epg.S(1), rlx, rfc, epg.S(1), rlx, epg.ADC] * 20
for _ in range(3):
    r=epg.simulate(seq, probe=epg.Jacobian(["magnitude","T2"]), max_nstate=63)
import cProfile, pstats
pr=cProfile.Profile(); pr.enable()
r=epg.simulate(seq, probe=epg.Jacobian(["magnitude","T2"]), max_nstate=63)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
