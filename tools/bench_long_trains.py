"""Echo trains FROM EQUILIBRIUM whose state matrix is never bounded (2 n + 1 orders after n echoes: the reference's own growth,
functions.py:135 / shift.py:86): launch time per train length, state-resident.  One JSON line per train; run it once per setting
of EPGX_CGROW (the library reads the variable once per process):

    for s in 0 1 2; do EPGX_CGROW=$s python tools/bench_long_trains.py; done
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epgpy_amd import epg, _lib, functions, workloads  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nvox", type=int, default=262144)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--nechos", type=int, nargs="*", default=[40, 63, 100, 127, 180, 255, 400, 511, 800, 1023])
    ap.add_argument("--max-nstate", type=int, default=0, help="bound the state matrix (a train longer than that then runs at the capacity)")
    args = ap.parse_args()
    n = args.nvox
    side = int(round(n ** 0.5))
    T1, T2 = np.linspace(200, 3000, side)[:, None], np.linspace(20, 300, n // side)[None, :]
    ctx = _lib.get_context(None)
    for necho in args.nechos:
        seq = workloads.mse_sequence(epg, T1, T2, necho=necho)
        enc, _, _ = functions.compile_sequence(seq, None, options={"max_nstate": args.max_nstate} if args.max_nstate else {})
        K = enc.capacity(resident=True)
        plan = enc.device_plan(ctx, K)
        nv = enc.nvox
        sig = _lib.DeviceBuffer(ctx, 16 * enc.n_adc * nv)
        name = _lib.kernel_for(ctx, plan, K)
        _lib.run(ctx, plan, 0, plan.n_ops, 0, nv, None, None, K, sig.ptr.value, nv, 0)
        ctx.synchronize()
        ctx.timer_start()
        for _ in range(args.steps):
            _lib.run(ctx, plan, 0, plan.n_ops, 0, nv, None, None, K, sig.ptr.value, nv, 0)
        ms = ctx.timer_stop() / args.steps
        populated = sum(min(2 * e + 1, K) for e in range(1, necho + 1))      # orders that can hold anything, summed over the echoes
        print(json.dumps({"cgrow": os.environ.get("EPGX_CGROW", "default"), "necho": necho, "max_nstate": args.max_nstate or None, "K": K, "nvox": nv, "kernel": name,
                          "ms": round(ms, 3), "echo_voxels_per_s": necho * nv / ms * 1e3,
                          "capacity_order_echo_voxels_per_s": necho * nv * K / ms * 1e3,
                          "populated_order_echo_voxels_per_s": populated * nv / ms * 1e3}), flush=True)
        sig.free()
        del plan


if __name__ == "__main__":
    main()
