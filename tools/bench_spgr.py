"""Spoiled gradient-echo train with a perfect SPOILER in every repetition: [T(a_i), E, ADC, E, S, SPOILER] x ntr over an
(T1, T2) grid -- every repetition is a record pair whose first record starts with the spoiler.
    python tools/bench_spgr.py [--n 256] [--ntr 200]
"""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epgpy_amd import epg, _lib, functions

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=256)
ap.add_argument("--ntr", type=int, default=200)
args = ap.parse_args()
n = args.n
T1 = np.linspace(300, 3000, n)[:, None]
T2 = np.linspace(20, 300, n)[None, :]
rng = np.random.default_rng(0)
rlx1, sh = epg.E(3.0, T1, T2), epg.S(1)
seq = []
for i in range(args.ntr):
    seq += [epg.T(float(rng.uniform(5, 25)), 0.0), rlx1, epg.ADC, epg.E(float(rng.uniform(6, 9)), T1, T2), sh, epg.SPOILER]
ctx = _lib.get_context(None)
for fuse in (True, False):
    enc, _, _ = functions.compile_sequence(seq, None, options={"max_nstate": 15}, fuse=fuse)
    K = enc.packable() or 64
    plan = enc.device_plan(ctx, K)
    sig = _lib.DeviceBuffer(ctx, 16 * enc.n_adc * enc.nvox)
    run = lambda: _lib.run(ctx, plan, 0, plan.n_ops, 0, enc.nvox, None, None, K, sig.ptr.value, enc.nvox, 0)
    run(); ctx.synchronize(); ctx.timer_start()
    for _ in range(5): run()
    ms = ctx.timer_stop() / 5
    print(json.dumps({"workload": f"spoiled GRE {args.ntr} TR, {n}x{n} voxels, max_nstate=15", "fuse": fuse, "K": K, "ms_per_pass": round(ms, 3),
                      "TR_voxels_per_s": args.ntr * enc.nvox / ms * 1e3}), flush=True)
