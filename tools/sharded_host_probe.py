#!/usr/bin/env python3
"""Where the time of simulate_sharded(out="host") goes with N rank processes on ONE box (all on GPU 0 unless --spread):
per call the wall time on the destination, with EPGX_TRACE lines of the library (run_to_host) on stderr.

    python tools/sharded_host_probe.py [--ranks 2] [--workload mse_1024] [--calls 4] [--via pcie] [--c64] [--spread]
"""
import argparse
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, args):
    import numpy as np
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from epgpy_amd import epg, workloads as wl
    from epgpy_amd.distributed import simulate_sharded

    seq, _, n_adc, opts = wl.build(epg, args.workload)
    kw = dict(via=args.via, device=(rank if args.spread else 0), **opts)
    if args.c64:
        kw["dtype"] = np.complex64
    keep = []
    for call in range(args.calls):
        dist.barrier()
        t0 = time.perf_counter()
        res = simulate_sharded(seq, **kw)
        dt = time.perf_counter() - t0
        dist.barrier()
        if rank == 0:
            print(f"call {call}: {1e3 * dt:8.2f} ms  {res.nbytes / 1e9 / dt:6.1f} GB/s  ({res.nbytes / 1e6:.0f} MB, {res.dtype})", flush=True)
        if args.keep:
            keep.append(res)
        del res
    dist.destroy_process_group()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=2)
    ap.add_argument("--workload", default="mse_1024")
    ap.add_argument("--calls", type=int, default=4)
    ap.add_argument("--via", default="pcie")
    ap.add_argument("--c64", action="store_true")
    ap.add_argument("--keep", action="store_true", help="keep every result alive (no recycling of result blocks)")
    ap.add_argument("--spread", action="store_true", help="rank r on GPU r (a multi-GPU box)")
    args = ap.parse_args()
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(worker, args=(args.ranks, port, args), nprocs=args.ranks, join=True)
