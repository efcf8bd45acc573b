"""MRF-type train with max_nstate = 10 (the reference's usual setting), state-resident: capacity K = 16 (rows_kernel
with one order per lane) vs K = 64 (four orders per lane, truncating); then the same train with three derivative
states (K = 16: four voxels per wavefront, K = 64: one).

    python tools/bench_packed.py [--m 100] [--ntr 1000]
"""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epgpy_amd import epg, _lib, functions
from epgpy_amd import workloads as sq

ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=100)
ap.add_argument("--ntr", type=int, default=1000)
ap.add_argument("--nstate", type=int, default=10)
args = ap.parse_args()
m = args.m
T1 = np.linspace(300, 3000, m)[:, None, None]
T2 = np.linspace(20, 300, m)[None, :, None]
B1 = np.linspace(0.7, 1.3, m)[None, None, :]
alpha, TR = sq.mrf_trains(args.ntr)
seq = sq.mrf_sequence(epg, T1, T2, B1, alpha, TR)
ctx = _lib.get_context(None)
enc, _, _ = functions.compile_sequence(seq, None, options={"max_nstate": args.nstate})
plan = enc.device_plan(ctx, 64)
sig = _lib.DeviceBuffer(ctx, 16 * enc.n_adc * enc.nvox)
for K in [k for k in (64, enc.packable()) if k]:
    run = lambda: _lib.run(ctx, plan, 0, plan.n_ops, 0, enc.nvox, None, None, K, sig.ptr.value, enc.nvox, 0)
    run(); ctx.synchronize(); ctx.timer_start()
    for _ in range(3): run()
    ms = ctx.timer_stop() / 3
    print(json.dumps({"workload": f"MRF {args.ntr} TR, {m}^3 voxels, max_nstate={args.nstate}", "K": K, "orders_per_lane": max(1, K // 16),
                      "ms_per_pass": round(ms, 3), "echo_voxels_per_s": args.ntr * enc.nvox / ms * 1e3}), flush=True)

# ---- the same train with derivatives w.r.t. T2, T1 and B1 (Jacobian workloads of examples/differentiation)
rlx_cache = {}
def E(tau):
    return epg.E(tau, T1, T2, order1=["T1", "T2"])
seqd = [epg.T(180 * B1, 90, order1={"B1": {"alpha": 180.0}}), E(20)]
rlx1 = E(3.0)
for a_i, tr in zip(alpha, TR):
    seqd += [epg.T(a_i * B1, 90, order1={"B1": {"alpha": float(a_i)}}), rlx1, epg.ADC, E(tr - 3.0), epg.S(1)]
encd, _, _ = functions.compile_sequence(seqd, None, options={"max_nstate": args.nstate}, variables=["T2", "T1", "B1"])
pland = encd.device_plan(ctx, 64)
sigd = _lib.DeviceBuffer(ctx, 16 * encd.n_adc * encd.nvox)
for K in [k for k in (64, encd.packable(derivatives=True)) if k]:
    run = lambda: _lib.run(ctx, pland, 0, pland.n_ops, 0, encd.nvox, None, None, K, sigd.ptr.value, encd.nvox, 0)
    run(); ctx.synchronize(); ctx.timer_start()
    for _ in range(2): run()
    ms = ctx.timer_stop() / 2
    print(json.dumps({"workload": f"MRF {args.ntr} TR, {m}^3 voxels, max_nstate={args.nstate}, state + 3 derivative states", "K": K,
                      "voxels_per_wave": 64 // K, "ms_per_pass": round(ms, 3),
                      "echo_voxels_per_s": args.ntr * encd.nvox / ms * 1e3}), flush=True)
