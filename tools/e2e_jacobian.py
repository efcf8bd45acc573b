import sys, time, numpy as np
sys.path.insert(0, '.')
from epgpy_amd import epg
n=1024
T1, T2 = np.linspace(200, 3000, n)[:, None], np.linspace(20, 300, n)[None, :]
exc = epg.T(90, 90, order1={"B1": {"alpha": 90}}); rfc = epg.T(120, 0, order1={"B1": {"alpha": 120}})
rlx = epg.E(5.0, T1, T2, order1=["T1", "T2"])
seq = [exc] + [epg.S(1), rlx, rfc, epg.S(1), rlx, epg.ADC] * 20
for names in (["T2"], ["T2","T1"], ["T2","T1","B1"]):
    laps=[]
    for _ in range(4):
        t=time.perf_counter(); r=epg.simulate(seq, probe=epg.Jacobian(["magnitude"]+names), max_nstate=63); laps.append(time.perf_counter()-t)
    print(names, [round(x*1e3,1) for x in laps], r.shape, round(r.nbytes/1e6), "MB", "pcie floor ms", round(r.nbytes/54e9*1e3,1), flush=True)
import os
os.environ["EPGX_TRACE"]="1"
r=epg.simulate(seq, probe=epg.Jacobian(["magnitude","T2"]), max_nstate=63)
