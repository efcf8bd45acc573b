import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from epgpy_amd import epg
from oracle import epg_numpy as onp
from tests import sequences as sq
seed = int(sys.argv[1])
rng = np.random.default_rng(5000 + seed)
grid = tuple(int(x) for x in rng.integers(1, 5, rng.integers(1, 3)))
kdim = int(rng.integers(1, 4))
cap = [None, None, 2, 4][int(rng.integers(0, 4))]
kvalue = [float(v) for v in rng.uniform(5e3, 4e4, 3)]
tuples = sq.random_nd_sequence(rng, grid, kdim, nops=int(rng.integers(8, 30)))
opts = {"kvalue": kvalue}
if cap: opts["max_nstate"] = cap
print("grid", grid, "kdim", kdim, "cap", cap)
for t in tuples: print("  ", t[0], [np.shape(x) if hasattr(x, 'shape') else x for x in t[1:]])
ref, (ref_states, ref_coords) = onp.simulate_nd(tuples, shape=grid, return_states=True, **opts)
ops = sq.nd_to_ops(epg, tuples)
sm = epg.StateMatrix(shape=grid, **opts)
for n, op in enumerate(ops):
    sm = op(sm, inplace=True)
    # compare against the oracle after every operator
    r, (rs, rc) = onp.simulate_nd(tuples[:n + 1] + [("ADC",)], shape=grid, return_states=True, **opts)
    if rc is None: continue
    coords = np.asarray(sm.coords).reshape(-1, np.asarray(sm.coords).shape[-1])
    states = np.asarray(sm.states)
    lookup = {tuple(int(v) for v in c): i for i, c in enumerate(coords)}
    worst = 0
    for ri, c in enumerate(rc):
        key = tuple(int(v) for v in c) + (0,) * (coords.shape[-1] - len(c))
        if key in lookup:
            worst = max(worst, float(np.max(np.abs(states[..., lookup[key], :] - rs[..., ri, :]))))
        else:
            worst = max(worst, float(np.max(np.abs(rs[..., ri, :]))))
    print(n, tuples[n][0], tuples[n][1] if tuples[n][0] == "S" else "", "nrows dev", coords.shape[0], "ref", rc.shape[0], "worst", f"{worst:.2e}")
