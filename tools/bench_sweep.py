"""Throughput of the fused kernel over the state capacity K (64 ... 1024 orders per voxel) in both
modes: MSE train with max_nstate = K - 1, necho echoes, n voxels.  One JSON line per (K, mode).

    python tools/bench_sweep.py [--nvox 262144] [--necho 20] [--steps 3]
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epgpy_amd import epg, _lib, functions  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nvox", type=int, default=262144)
    ap.add_argument("--necho", type=int, default=20)
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args()
    n = args.nvox
    T2 = np.linspace(20, 300, n)
    exc, rfc, rlx, sh = epg.T(90, 90), epg.T(120, 0), epg.E(5.0, 1000.0, T2), epg.S(1)
    seq = [exc] + [sh, rlx, rfc, sh, rlx, epg.ADC] * args.necho
    ctx = _lib.get_context(None)
    for K in _lib.SUPPORTED_K + (_lib.RESIDENT_ONLY_K,):
        wide = K == _lib.RESIDENT_ONLY_K     # 2048 orders: state-resident from equilibrium only (four wavefronts per voxel)
        enc, _, bounds = functions.compile_sequence(seq, None, options={"max_nstate": K - 1}, nstate0=0 if wide else K - 1)
        plan = enc.device_plan(ctx, K)
        sig = _lib.DeviceBuffer(ctx, 16 * enc.n_adc * n)
        state = None if wide else _lib.DeviceState(ctx, n, K)
        for mode in (("resident",) if wide else ("resident", "stream")):
            def run():
                if mode == "resident":
                    _lib.run(ctx, plan, 0, plan.n_ops, 0, n, state, None, K, sig.ptr.value, n, 0)
                    return 1
                begin = 0
                for end in bounds:
                    _lib.run(ctx, plan, begin, end, 0, n, state, state, K, sig.ptr.value, n, 0)
                    begin = end
                return len(bounds)
            run()
            ctx.synchronize()
            ctx.timer_start()
            for _ in range(args.steps):
                launches = run()
            ms = ctx.timer_stop() / args.steps
            units = args.necho * n
            bytes_per_launch = n * (2 * K * 48 + 16)
            line = {"K": K, "mode": mode, "nvox": n, "necho": args.necho, "ms_per_step": round(ms, 3),
                    "echo_voxels_per_s": units / ms * 1e3, "order_echo_voxels_per_s": units * K / ms * 1e3}
            if mode == "stream":
                line["GB_per_s"] = round(bytes_per_launch * launches / ms / 1e6, 1)
            print(json.dumps(line), flush=True)
        sig.free()
        del state


if __name__ == "__main__":
    main()
