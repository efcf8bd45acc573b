"""RF-spoiled gradient echo with a perfect spoiler per repetition over a (T1, T2, B1) grid, state-resident:
    python tools/bench_spgr.py [--m 100] [--ntr 500]      (EPGX_FOLD=0: the unfolded records, for comparison)
"""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epgpy_amd import epg, _lib, functions

ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=100)
ap.add_argument("--ntr", type=int, default=500)
ap.add_argument("--derivatives", action="store_true", help="also time the train with 1 (T1) and 3 (T1, T2, B1) derivative states")
args = ap.parse_args()
m = args.m
T1 = np.linspace(300, 3000, m)[:, None, None]
T2 = np.linspace(20, 300, m)[None, :, None]
B1 = np.linspace(0.7, 1.3, m)[None, None, :]
rl1, rl2 = epg.E(3.0, T1, T2), epg.E(7.0, T1, T2)
seq = []
for n in range(args.ntr):
    seq += [epg.T(14.8 * B1, 58.5 * n * n % 360), rl1, epg.ADC, rl2, epg.SPOILER]
ctx = _lib.get_context(None)
enc, _, _ = functions.compile_sequence(seq, None, options={"max_nstate": 63})
K = enc.packable() or enc.capacity()
plan = enc.device_plan(ctx, 64)
sig = _lib.DeviceBuffer(ctx, 16 * enc.n_adc * enc.nvox)
run = lambda: _lib.run(ctx, plan, 0, plan.n_ops, 0, enc.nvox, None, None, K, sig.ptr.value, enc.nvox, 0)
run(); ctx.synchronize(); ctx.timer_start()
for _ in range(3): run()
ms = ctx.timer_stop() / 3
print(json.dumps({"workload": f"SPGR {args.ntr} TR with spoiler, {m}^3 voxels", "K": K, "ms_per_pass": round(ms, 3),
                  "TR_voxels_per_s": args.ntr * enc.nvox / ms * 1e3}))

if args.derivatives:     # the spoiler joins the run-time fold of packed_dfold_kernel
    rl1, rl2 = epg.E(3.0, T1, T2, order1=["T1", "T2"]), epg.E(7.0, T1, T2, order1=["T1", "T2"])
    seqd = []
    for n in range(args.ntr):
        seqd += [epg.T(14.8 * B1, 58.5 * n * n % 360, order1={"B1": {"alpha": 14.8}}), rl1, epg.ADC, rl2, epg.SPOILER]
    for variables in (["T1"], ["T1", "T2", "B1"]):
        encd, _, _ = functions.compile_sequence(seqd, None, options={"max_nstate": 63}, variables=variables)
        Kd = encd.packable(derivatives=True) or encd.capacity()
        pland = encd.device_plan(ctx, 64)
        sigd = _lib.DeviceBuffer(ctx, 16 * encd.n_adc * encd.nvox)
        rund = lambda: _lib.run(ctx, pland, 0, pland.n_ops, 0, encd.nvox, None, None, Kd, sigd.ptr.value, encd.nvox, 0)
        rund(); ctx.synchronize(); ctx.timer_start()
        for _ in range(3): rund()
        msd = ctx.timer_stop() / 3
        print(json.dumps({"workload": f"SPGR {args.ntr} TR with spoiler, {m}^3 voxels, state + {len(variables)} derivative states", "K": Kd,
                          "ms_per_pass": round(msd, 3), "TR_voxels_per_s": args.ntr * encd.nvox / msd * 1e3}))
        sigd.free()
