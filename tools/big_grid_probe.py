import sys, time, numpy as np
sys.path.insert(0, '.')
from epgpy_amd import epg, workloads as wl
from oracle import epg_c, workloads as ow
for n in (2048, 4096):
    T1=np.linspace(200,3000,n)[:,None]; T2=np.linspace(20,300,n)[None,:]
    seq=wl.mse_sequence(epg,T1,T2)
    for rep in range(2):
        t=time.perf_counter(); sig=epg.simulate(seq, max_nstate=63, out="device"); dt=time.perf_counter()-t
    print(n, "device", round(dt*1e3,2), "ms", 20*n*n/dt/1e9, "G echo*voxels/s", flush=True)
    t=time.perf_counter(); res=epg.simulate(seq, max_nstate=63); dt=time.perf_counter()-t
    print(n, "host", round(dt*1e3,1), "ms", res.shape, flush=True)
    rng=np.random.default_rng(0); i1=rng.integers(0,n,16); i2=rng.integers(0,n,16)
    ref=epg_c.simulate(ow.mse_tuples(T1[i1,0], T2[0,i2]), max_nstate=63)
    print("max err", np.abs(res[:, i1, i2]-ref).max())
    del res, sig
