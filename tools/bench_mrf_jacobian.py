"""Time the derivative kernels on the MRF train of BASELINE config 3 (variable-FA SSFP over a (T1, T2, B1) grid, max_nstate = 63)
with 0..3 derivative states (T1 / T2 / B1) -- the "dictionary with gradients" workload, whose
repetitions cannot be fused on the host (a rotation over the B1 axis between relaxations over (T1, T2)).

    python tools/bench_mrf_jacobian.py [--side 100] [--ntr 250] [--steps 3]
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epgpy_amd import epg, _lib, functions, workloads as wl  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--side", type=int, default=100)
    ap.add_argument("--ntr", type=int, default=250)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--max-nstate", type=int, default=63)
    ap.add_argument("--vars", type=int, nargs="*", default=[0, 1, 2, 3], help="numbers of derivative states to time")
    args = ap.parse_args()
    n = args.side
    T1 = np.linspace(300, 3000, n)[:, None, None]
    T2 = np.linspace(20, 300, n)[None, :, None]
    B1 = np.linspace(0.7, 1.3, n)[None, None, :]
    alpha, TR = wl.mrf_trains(args.ntr)
    o1 = ["T1", "T2"]
    seq = [epg.T(180 * B1, 90, order1={"B1": {"alpha": 180.0}}), epg.E(20, T1, T2, order1=o1)]
    rlx1, sh = epg.E(3.0, T1, T2, order1=o1), epg.S(1)
    for a, tr in zip(alpha, TR):
        seq += [epg.T(a * B1, 90, order1={"B1": {"alpha": float(a)}}), rlx1, epg.ADC, epg.E(tr - 3.0, T1, T2, order1=o1), sh]
    ctx = _lib.get_context(None)
    for variables in ([], ["T2"], ["T2", "T1"], ["T2", "T1", "B1"]):
        if len(variables) not in args.vars:
            continue
        enc, _, _ = functions.compile_sequence(seq, None, options={"max_nstate": args.max_nstate}, variables=variables)
        K = enc.packable(derivatives=bool(variables)) or enc.capacity()
        plan = enc.device_plan(ctx, K)
        sig = _lib.DeviceBuffer(ctx, 16 * enc.n_adc * enc.nvox)

        def run():
            _lib.run(ctx, plan, 0, plan.n_ops, 0, enc.nvox, None, None, K, sig.ptr.value, enc.nvox, 0)

        run()
        ctx.synchronize()
        ctx.timer_start()
        for _ in range(args.steps):
            run()
        ms = ctx.timer_stop() / args.steps
        units = args.ntr * enc.nvox
        print(json.dumps({"workload": f"mrf {n}^3, {args.ntr} TR, K={K}", "n_vars": len(variables), "ms_per_step": round(ms, 3),
                          "TR_voxels_per_s": units / ms * 1e3, "state_TR_voxels_per_s": units * (1 + len(variables)) / ms * 1e3}), flush=True)
        sig.free()


if __name__ == "__main__":
    main()
