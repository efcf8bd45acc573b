"""Time the derivative kernel: 20-echo MSE over an n x n (T1, T2) grid with 1..3 derivative states
(T2 / T1 / B1), state-resident.  Prints one JSON line per variable count.

    python tools/bench_jacobian.py [--n 1024] [--steps 5]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epgpy_amd import epg, _lib, functions  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--necho", type=int, default=20)
    ap.add_argument("--max-nstate", type=int, default=63, help="63: K = 64; 31: K = 32; 15: K = 16 (the packed kernels)")
    ap.add_argument("--force-fuse", action="store_true", help="fuse whatever the number of variables (functions.FUSE_DERIVATIVES)")
    ap.add_argument("--no-fuse", action="store_true", help="keep E, T, E three stages (no fused tables / generated partials)")
    args = ap.parse_args()
    n = args.n
    T1 = np.linspace(200, 3000, n)[:, None]
    T2 = np.linspace(20, 300, n)[None, :]
    exc = epg.T(90, 90, order1={"B1": {"alpha": 90}})
    rfc = epg.T(120, 0, order1={"B1": {"alpha": 120}})
    rlx = epg.E(5.0, T1, T2, order1=["T1", "T2"])
    sh = epg.S(1)
    seq = [exc] + [sh, rlx, rfc, sh, rlx, epg.ADC] * args.necho
    ctx = _lib.get_context(None)
    if args.force_fuse:
        functions.FUSE_DERIVATIVES = {64: 3}
    for variables in ([], ["T2"], ["T2", "T1"], ["T2", "T1", "B1"]):
        enc, _, _ = functions.compile_sequence(seq, None, options={"max_nstate": args.max_nstate}, variables=variables, fuse=not args.no_fuse)
        K = enc.packable(derivatives=bool(variables)) or enc.capacity()
        plan = enc.device_plan(ctx, K)
        sig = _lib.DeviceBuffer(ctx, 16 * enc.n_adc * enc.nvox)

        def run():
            _lib.run(ctx, plan, 0, plan.n_ops, 0, enc.nvox, None, None, K, sig.ptr.value, enc.nvox, 0)

        t0 = time.perf_counter()          # 25 ms under load first: plan compilation leaves the chip in its idle clocks
        try:
            run()
        except _lib.EpgxError as exc:     # (1024 orders: one derivative state per launch)
            print(json.dumps({"workload": f"mse {n}x{n}, {args.necho} echoes, K={K}", "n_vars": len(variables), "error": str(exc)[:120]}))
            sig.free()
            continue
        while time.perf_counter() - t0 < 0.025:
            run()
            ctx.synchronize()
        ctx.timer_start()
        for _ in range(args.steps):
            run()
        ms = ctx.timer_stop() / args.steps
        units = args.necho * enc.nvox
        print(json.dumps({"workload": f"mse {n}x{n}, {args.necho} echoes, K={K}", "n_vars": len(variables), "fused": not args.no_fuse,
                          "ms_per_step": round(ms, 3), "echo_voxels_per_s": units / ms * 1e3,
                          "state_echo_voxels_per_s": units * (1 + len(variables)) / ms * 1e3}))
        sig.free()


if __name__ == "__main__":
    main()
