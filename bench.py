#!/usr/bin/env python3
"""bench.py -- headline benchmark of the epgpy hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode resident|stream]
                    [--workload mse_1024|mse_256|mrf_100] [--no-cpu-baseline]

One "step" = one full pass of the hot path (the fused T/E/S/ADC kernel family of
libepgx.so) over one batch of synthetic input: the whole 20-echo multi-spin-echo sequence
over the (T1, T2) parameter grid of the workload, i.e. what one `epg.simulate(seq)` call
computes.  The plan (operator stream + coefficient tables) is uploaded once before the timed
region, so inputs are resident in HBM when timing starts; the signal stays in HBM.

Metric (BASELINE.json): echo-points x voxels / s.  Printed by rank 0 as ONE JSON line.

mode "resident" (default): one launch per step, every voxel's state matrix stays in VGPRs
  for the whole sequence (real HBM traffic ~ 16 B per echo.voxel).
mode "stream": one launch per echo, the state matrix [nvox][3][64] c128 is read and written
  once per launch -- the per-timestep kernel whose HBM roofline BASELINE.md quotes
  (B_alg = 2*64*3*16 + 16 = 6160 B per echo.voxel).
Both modes are measured in every run; `value` is the mode selected with --mode and the
other mode is reported in an extra object.  `roofline` always uses SURVEY.md section 8(d)'s
algorithmic bytes; for the resident kernel the fp64-VALU roofline is reported next to it
because that, not HBM, is what bounds it.

With N > 1 (launched by torch.distributed.run, one rank per GPU, backend nccl = RCCL) every
rank runs the same workload on its own grid slab of the same size (weak scaling: the grid
grows with N along T1), followed by ONE gather of the signal slabs to rank 0 inside the
timed region.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

K_STATES = 64
NECHO = 20
B_ALG = 2 * K_STATES * 3 * 16 + 16          # bytes per echo.voxel
FLOP_PER_UNIT = K_STATES * (66 + 2 * 14)    # fp64 flop per echo.voxel
HBM_PEAK_GBS = 8000.0                       # MI355X_MICROARCH.md: 8 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6                # 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz

WORKLOADS = {
    # name: (kind, grid)  -- SURVEY.md 8d synthetic inputs
    "mse_1024": ("mse", (1024, 1024)),   # C2-L: the >= 1e6-voxel target of north_star
    "mse_256": ("mse", (256, 256)),      # C2: BASELINE.json configs[1]
    "mrf_100": ("mrf", (100, 100, 100)), # C3: 1000-TR variable-FA SSFP over a (T1, T2, B1) grid
    "mrf_32": ("mrf", (32, 32, 32)),
}
MRF_NTR = 1000


def build_sequence(epg, kind, grid, rank=0, world=1):
    """this rank's sequence; weak scaling: the global grid is (world*n1) x ... and rank r owns
    rows [r*n1, (r+1)*n1) of the T1 axis.  Returns (sequence, params, n_adc, tuple_builder)"""
    from tests import sequences as sq

    n1 = grid[0]
    if kind == "mse":     # 20-echo MSE, README.md:52-76 shape
        T1 = np.linspace(200, 3000, n1 * world)[rank * n1:(rank + 1) * n1][:, None]
        T2 = np.linspace(20, 300, grid[1])[None, :]
        exc, rfc = epg.T(90, 90), epg.T(120, 0)
        rlx = epg.E(5.0, T1, T2)
        sh = epg.S(1, duration=5.0)
        seq = [exc] + [[sh, rlx, rfc, sh, rlx, epg.ADC]] * NECHO
        return seq, (T1, T2), NECHO, lambda i, j: sq.mse_tuples(T1[i, 0], T2[0, j])
    # MRF: [T(180 B1, 90), E(20)] + [T(a_i B1, 90), E(TE), ADC, E(TR_i - TE), S(1)] x 1000
    T1 = np.linspace(300, 3000, n1 * world)[rank * n1:(rank + 1) * n1][:, None, None]
    T2 = np.linspace(20, 300, grid[1])[None, :, None]
    B1 = np.linspace(0.7, 1.3, grid[2])[None, None, :]
    alpha, TR = sq.mrf_trains(MRF_NTR)
    seq = sq.mrf_ops(epg, T1, T2, B1, alpha, TR)
    return seq, (T1, T2, B1), MRF_NTR, lambda i, j, k: sq.mrf_tuples(T1[i, 0, 0], T2[0, j, 0], B1[0, 0, k], alpha, TR)


def cpu_baseline(n_side, threads):
    """time the C oracle (oracle/epg_oracle.c, a port of the reference algorithm) on a
    bounded sub-grid of the same workload on the host cores"""
    from oracle import epg_c
    from tests import sequences as sq

    T1 = np.linspace(200, 3000, n_side)[:, None]
    T2 = np.linspace(20, 300, n_side)[None, :]
    tuples = sq.mse_tuples(T1, T2)
    grid = (n_side, n_side)
    compiled = epg_c.compile_ops(tuples, grid)
    epg_c.simulate(tuples[:8], max_nstate=63, nthreads=threads)  # warm the library
    t0 = time.perf_counter()
    epg_c.simulate(tuples, max_nstate=63, nthreads=threads, compiled=compiled)
    dt = time.perf_counter() - t0
    return NECHO * n_side * n_side / dt, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", choices=["resident", "stream"], default="resident")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="mse_1024")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--only", action="store_true", help="measure only --mode (profiling runs)")
    ap.add_argument("--cpu-side", type=int, default=384, help="CPU baseline sub-grid side")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nnodes=1 "
                         "--nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")

    from epgpy_amd import epg, _lib
    from epgpy_amd.distributed import ShardedPlan

    dist = torch = None
    if "WORLD_SIZE" in os.environ:   # launched by torch.distributed.run (also with one rank)
        import torch
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    kind, grid = WORKLOADS[args.workload]
    seq, params, NADC, tuples_at = build_sequence(epg, kind, grid, rank, world)
    # every rank simulates its own full slab: a 1-rank ShardedPlan over the local grid
    sp = ShardedPlan(seq, rank=0, world_size=1, device=local_rank, max_nstate=K_STATES - 1)
    NCHUNK = 4   # multi-GPU: the slab is computed in NCHUNK pieces, each gathered asynchronously
    if torch is not None:
        sp.bind(torch.cuda.current_stream().cuda_stream)
        dev = torch.device("cuda", local_rank)
        csz = -(-sp.slab // NCHUNK)
        parts = [(c * csz, max(0, min(csz, sp.slab - c * csz))) for c in range(NCHUNK)]
        sig_parts = [torch.zeros((sp.n_adc, csz), dtype=torch.complex128, device=dev) for _ in parts]
        gather_bufs = [[torch.empty((sp.n_adc, csz, 2), dtype=torch.float64, device=dev) for _ in range(world)]
                       if rank == 0 else None for _ in parts]
        states = [sp.new_state(cnt) for _, cnt in parts]
    else:
        sp.bind()
        sig_buf = _lib.DeviceBuffer(sp._ctx, 16 * sp.n_adc * sp.slab)
        sig_ptr = sig_buf.ptr.value
        state = sp.new_state()
    ctx = sp._ctx
    nvox = sp.nvox
    units_per_step = NADC * nvox                       # echo.voxels per rank per step
    n_launch = {"resident": 1, "stream": len(sp.bounds)}
    if dist is not None:
        n_launch = {k: v * NCHUNK for k, v in n_launch.items()}

    def step(mode):
        if dist is None:
            sp.run(sig_ptr, mode=mode, state=state)
            return
        # ONE logical gather of the signal to rank 0, issued per chunk with async_op so that the
        # RCCL transfers of chunk c overlap the kernel of chunk c+1
        handles = []
        for part, sig_c, buf_c, st_c in zip(parts, sig_parts, gather_bufs, states):
            sp.run(sig_c.data_ptr(), mode=mode, state=st_c, part=part, signal_ld=sig_c.shape[1])
            handles.append(dist.gather(torch.view_as_real(sig_c), buf_c, dst=0, async_op=True))
        for h in handles:
            h.wait()

    def sync():
        if torch is not None:
            torch.cuda.synchronize()
        else:
            ctx.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()

    def timed(mode, steps, warmup):
        for _ in range(warmup):
            step(mode)
        sync(); barrier(); sync()
        ctx.timer_start()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(mode)
        kernel_ms = ctx.timer_stop() if dist is None else None   # HIP events on the launch stream
        sync(); barrier(); sync()
        wall = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([wall], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wall = float(t.item())
        return wall, kernel_ms

    results = {}
    for mode in ("resident", "stream"):
        if args.only and mode != args.mode:
            continue
        steps = args.steps if mode == args.mode else max(3, args.steps // 4)
        wall, kernel_ms = timed(mode, steps, args.warmup)
        per_launch_ms = (kernel_ms / (steps * n_launch[mode])) if kernel_ms is not None else None
        results[mode] = {"wall": wall, "steps": steps, "kernel_ms_per_launch": per_launch_ms,
                         "value": units_per_step * world * steps / wall}

    # parity spot check of what was just computed (rank 0, oracle as checker only)
    parity = None
    if rank == 0:
        from oracle import epg_c
        from tests import sequences as sq

        rng = np.random.default_rng(0)
        nsamp = 256 if kind == "mse" else 16
        coords = [rng.integers(0, g, nsamp) for g in grid]
        flat = np.ravel_multi_index(coords, grid)
        ref = epg_c.simulate(tuples_at(*coords), max_nstate=K_STATES - 1)
        if torch is not None:
            got = torch.cat(sig_parts, dim=1)[:, torch.as_tensor(flat, device=dev)].cpu().numpy()
            rows = np.arange(sp.n_adc)
        elif 16 * sp.n_adc * sp.slab <= (1 << 30):
            got = sig_buf.download(np.complex128, (sp.n_adc, sp.slab))[:, flat]
            rows = np.arange(sp.n_adc)
        else:   # the C3 signal is 16 GB: fetch single samples of the drawn voxels
            rows = np.unique(np.linspace(0, sp.n_adc - 1, 16).astype(int))
            got = np.zeros((sp.n_adc, nsamp), dtype=np.complex128)
            one = np.empty(1, dtype=np.complex128)
            for c, vx in enumerate(flat):
                for r in rows:
                    _lib.check(ctx.lib.epgx_memcpy_d2h(ctx.handle, one.ctypes.data,
                                                       sig_ptr + 16 * (int(r) * sp.slab + int(vx)), 16))
                    got[r, c] = one[0]
        parity = float(np.max(np.abs(got[rows] - ref[rows])))

    def pmc_traffic(mode):
        """HBM bytes per launch measured with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate
        passes, tools/prof.sh) on this workload; corrected as MI355X_MICROARCH.md prescribes"""
        try:
            t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))[args.workload][mode]
            return t["bytes"] if world == 1 else None
        except (OSError, KeyError, ValueError):
            return None

    def roofline(mode):
        r = results[mode]
        ms = r["kernel_ms_per_launch"]
        if ms is None:   # multi-GPU: derive from wall (includes the gather)
            ms = 1e3 * r["wall"] / (r["steps"] * n_launch[mode])
        units_per_launch = units_per_step / n_launch[mode]
        achieved = units_per_launch * B_ALG / (ms * 1e-3) / 1e9
        tflops = units_per_launch * FLOP_PER_UNIT / (ms * 1e-3) / 1e12
        return {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(mode),
                "kernel": "epgx::run_kernel<1>", "launch_ms": round(ms, 4),
                "units_per_launch": int(units_per_launch), "alg_bytes_per_unit": B_ALG,
                "fp64": {"achieved": round(tflops, 2), "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(tflops / FP64_VALU_PEAK_TFLOPS, 4), "flop_per_unit": FLOP_PER_UNIT}}

    if rank == 0:
        main_r = results[args.mode]
        other = "stream" if args.mode == "resident" else "resident"
        out = {
            "metric": "echo-points x voxels / sec (MSE, 20 echoes, 64 k-states)",
            "value": main_r["value"], "unit": "echo*voxels/s",
            "n_gpus": world, "steps": main_r["steps"], "warmup": args.warmup,
            "ms_per_step": 1e3 * main_r["wall"] / main_r["steps"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": (f"{args.workload}: 20-echo MSE (FA=120, ESP=10 ms), T1=linspace(200,3000,{grid[0]}*N) x "
                                    f"T2=linspace(20,300,{grid[1]}), max_nstate=63 (K=64), per-GPU grid {grid[0]}x{grid[1]}")
                       if kind == "mse" else
                       (f"{args.workload}: MRF {MRF_NTR}-TR variable-FA SSFP (SURVEY.md 8d C3), T1=linspace(300,3000,{grid[0]}*N) x "
                        f"T2=linspace(20,300,{grid[1]}) x B1=linspace(0.7,1.3,{grid[2]}), max_nstate=63 (K=64)"),
                       "mode": args.mode, "voxels_per_gpu": nvox, "echoes": NADC, "k_states": K_STATES,
                       "launches_per_step": n_launch[args.mode], "parallelism": f"voxel-slabs x{world}"},
            "roofline": roofline(args.mode),
        }
        if other in results:
            out[other] = {"value": results[other]["value"], "ms_per_step": 1e3 * results[other]["wall"] / results[other]["steps"],
                    "steps": results[other]["steps"], "launches_per_step": n_launch[other], "roofline": roofline(other)}
        out["parity_max_abs_err_vs_oracle"] = parity
        if not args.no_cpu_baseline and world == 1 and kind == "mse":
            threads = max(1, min(os.cpu_count() or 1, 16))
            v1, t1 = cpu_baseline(max(64, args.cpu_side // 3), 1)
            vn, tn = cpu_baseline(args.cpu_side, threads)
            out["cpu_baseline"] = {"value": vn, "unit": "echo*voxels/s", "cores": threads, "kind": "port",
                                   "sample": f"same MSE on a {args.cpu_side}x{args.cpu_side} sub-grid "
                                             f"({NECHO * args.cpu_side ** 2} echo*voxels, {tn:.1f} s), C oracle + OpenMP",
                                   "value_1core": v1}
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
