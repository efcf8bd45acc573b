"""How much of config 3's launch is the latency of its relaxation-table fetches?  The 1000-TR MRF train reads one 320 KB
E(TR_i - TE) table per repetition (320 MB in all: every line comes from HBM); the same train with the repetition times
drawn from a small set keeps those tables in L2 and runs the identical instruction stream.
    python tools/mrf_table_probe.py [--m 100] [--ntr 1000]          (GPU box)
"""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epgpy_amd import epg, _lib, functions, workloads as wl

ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=100)
ap.add_argument("--ntr", type=int, default=1000)
args = ap.parse_args()
m = args.m
T1 = np.linspace(200, 3000, m)[:, None, None]
T2 = np.linspace(20, 300, m)[None, :, None]
B1 = np.linspace(0.7, 1.2, m)[None, None, :]
alpha, TR = wl.mrf_trains(args.ntr)
ctx = _lib.get_context(None)
for label, trs in (("distinct TR per repetition", TR), ("8 distinct TRs", TR[np.arange(args.ntr) % 8]), ("one TR", np.full(args.ntr, 13.0)),
                   ("distinct TR per repetition (again)", TR)):
    seq = wl.mrf_sequence(epg, T1, T2, B1, alpha, trs)
    enc, _, _ = functions.compile_sequence(seq, None, options={"max_nstate": 63})
    K = enc.packable() or enc.capacity()
    plan = enc.device_plan(ctx, K)
    sig = _lib.DeviceBuffer(ctx, 16 * enc.n_adc * enc.nvox)
    run = lambda: _lib.run(ctx, plan, 0, plan.n_ops, 0, enc.nvox, None, None, K, sig.ptr.value, enc.nvox, 0)
    run(); ctx.synchronize(); ctx.timer_start()
    for _ in range(3): run()
    ms = ctx.timer_stop() / 3
    print(json.dumps({"trains": label, "K": K, "ms_per_pass": round(ms, 3), "TR_voxels_per_s": args.ntr * enc.nvox / ms * 1e3}), flush=True)
    del plan, sig
