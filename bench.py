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
grows with N along T1).  Voxels never interact, so the timed region holds NO data-path
collective: like at N = 1, every GPU's signal slab stays resident in its own HBM.  After the
timed region the slabs are gathered ONCE to rank 0 over RCCL (what `simulate_sharded` does for
a caller that wants the whole array in one place); rank 0 checks slabs of several ranks against
the oracle and reports the gather time separately (`gather`), or inside the timed region with
--gather-in-step.
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


def cpu_baseline(n_side, threads, budget_s):
    """time the C oracle (oracle/epg_oracle.c, a port of the reference algorithm) on the same
    workload at n_side x n_side on the host cores: whole passes are repeated until `budget_s`
    seconds of wall time are spent.  Returns (echo.voxels/s, seconds, passes)"""
    from oracle import epg_c
    from tests import sequences as sq

    T1 = np.linspace(200, 3000, n_side)[:, None]
    T2 = np.linspace(20, 300, n_side)[None, :]
    tuples = sq.mse_tuples(T1, T2)
    grid = (n_side, n_side)
    compiled = epg_c.compile_ops(tuples, grid)
    epg_c.simulate(tuples[:8], max_nstate=63, nthreads=threads)  # warm the library
    passes, t0 = 0, time.perf_counter()
    while passes == 0 or (time.perf_counter() - t0 < budget_s and passes < 64):
        epg_c.simulate(tuples, max_nstate=63, nthreads=threads, compiled=compiled)
        passes += 1
    dt = time.perf_counter() - t0
    return passes * NECHO * n_side * n_side / dt, dt, passes


def cpu_baseline_mrf(m, threads):
    """the MRF workload is timed on an m^3 sub-grid of the same parameter ranges (C oracle, all
    host threads, one pass) and reported per TR.voxel.  Returns (TR.voxels/s, seconds)"""
    from oracle import epg_c
    from tests import sequences as sq

    T1 = np.linspace(300, 3000, m)[:, None, None]
    T2 = np.linspace(20, 300, m)[None, :, None]
    B1 = np.linspace(0.7, 1.3, m)[None, None, :]
    alpha, TR = sq.mrf_trains(MRF_NTR)
    tuples = sq.mrf_tuples(T1, T2, B1, alpha, TR)
    compiled = epg_c.compile_ops(tuples, (m, m, m))   # table preparation (Python) is not part of the timed pass
    epg_c.simulate(tuples[:8], max_nstate=63, nthreads=threads)  # warm the library
    t0 = time.perf_counter()
    epg_c.simulate(tuples, max_nstate=63, nthreads=threads, compiled=compiled)
    dt = time.perf_counter() - t0
    return MRF_NTR * m ** 3 / dt, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", choices=["resident", "stream"], default="resident")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="mse_1024")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--only", action="store_true", help="measure only --mode (profiling runs)")
    ap.add_argument("--cpu-side", type=int, default=1024, help="CPU baseline grid side (default: the workload's own)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline time budget (all-thread leg)")
    ap.add_argument("--no-fuse", action="store_true", help="keep E . T . E as three operators (A/B measurements)")
    ap.add_argument("--gather-in-step", action="store_true",
                    help="N > 1: gather the signal slabs to rank 0 inside every timed step")
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
        import datetime
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank),
                                timeout=datetime.timedelta(minutes=10))

    kind, grid = WORKLOADS[args.workload]
    seq, params, NADC, tuples_at = build_sequence(epg, kind, grid, rank, world)
    # every rank simulates its own full slab: a 1-rank ShardedPlan over the local grid
    sp = ShardedPlan(seq, rank=0, world_size=1, device=local_rank, max_nstate=K_STATES - 1, fuse=not args.no_fuse)
    if torch is not None:
        sp.bind(torch.cuda.current_stream().cuda_stream)
        dev = torch.device("cuda", local_rank)
        sig_t = torch.zeros((sp.n_adc, sp.slab), dtype=torch.complex128, device=dev)
        sig_ptr = sig_t.data_ptr()
    else:
        sp.bind()
        sig_buf = _lib.DeviceBuffer(sp._ctx, 16 * sp.n_adc * sp.slab)
        sig_ptr = sig_buf.ptr.value
    state = sp.new_state()
    ctx = sp._ctx
    nvox = sp.nvox
    units_per_step = NADC * nvox                       # echo.voxels per rank per step
    n_launch = {"resident": 1, "stream": len(sp.bounds)}
    sig_bytes = 16 * sp.n_adc * sp.slab
    can_gather = dist is not None and world * sig_bytes <= (96 << 30)
    gather_bufs = None

    def gather():
        """ONE gather of the signal slabs to rank 0: RCCL send/recv, every peer over its own xGMI link"""
        nonlocal gather_bufs
        if gather_bufs is None and rank == 0:
            gather_bufs = [torch.empty((sp.n_adc, sp.slab, 2), dtype=torch.float64, device=dev) for _ in range(world)]
        # the kernels may run on the library's own stream (torch's current stream is the null stream,
        # which epgx_ctx_set_stream reads as "use your own"): drain it before RCCL reads the signal
        ctx.synchronize()
        dist.gather(torch.view_as_real(sig_t), gather_bufs, dst=0)

    def step(mode):
        sp.run(sig_ptr, mode=mode, state=state)
        if args.gather_in_step and can_gather:
            gather()

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
        # HIP events on the launch stream (rank 0's own kernels)
        kernel_ms = ctx.timer_stop() if not (args.gather_in_step and can_gather) else None
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

    # BASELINE.json configs[1] itself (256 x 256 grid): 65 536 wavefronts do not fill the chip for long
    # enough to be a roofline measurement (a resident launch is ~0.2 ms), but its rate is reported too
    config1 = None
    if kind == "mse" and args.workload != "mse_256" and world == 1 and not args.only:
        seq1, _, _, _ = build_sequence(epg, "mse", WORKLOADS["mse_256"][1])
        sp1 = ShardedPlan(seq1, rank=0, world_size=1, device=local_rank, max_nstate=K_STATES - 1, fuse=not args.no_fuse)
        sp1.bind(torch.cuda.current_stream().cuda_stream if torch is not None else None)
        buf1 = _lib.DeviceBuffer(sp1._ctx, 16 * sp1.n_adc * sp1.slab)
        st1 = sp1.new_state()
        config1 = {"workload": "mse_256 (BASELINE.json configs[1]): the same sequence over 256x256 (T1, T2)"}
        for mode1 in ("resident", "stream"):
            for _ in range(5):
                sp1.run(buf1.ptr.value, mode=mode1, state=st1)
            sync()
            nrep = 50
            sp1._ctx.timer_start()
            for _ in range(nrep):
                sp1.run(buf1.ptr.value, mode=mode1, state=st1)
            ms1 = sp1._ctx.timer_stop() / nrep
            config1[mode1] = {"ms_per_step": round(ms1, 4), "value": NECHO * sp1.nvox / (ms1 * 1e-3)}
        buf1.free()

    # after the timed region: gather the slabs once (N > 1), timed on its own
    gather_info = None
    if dist is not None and world > 1 and can_gather:
        try:   # the measurement above is complete: a failing gather must not cost it its JSON line
            sync(); barrier()
            t0 = time.perf_counter()
            gather()
            sync(); barrier()
            gather_info = {"ms": round(1e3 * (time.perf_counter() - t0), 3), "GB_to_rank0": round((world - 1) * sig_bytes / 1e9, 3),
                           "in_timed_region": bool(args.gather_in_step)}
        except Exception as exc:   # noqa: BLE001
            gather_info = {"error": repr(exc)}
            gather_bufs = None

    # parity spot check of what was just computed (rank 0, oracle as checker only); with N > 1 the
    # gathered slabs of the first, a middle and the last rank are checked
    parity, parity_error = None, None
    if rank == 0:
        from oracle import epg_c

        try:   # a failing check must not cost the measurement its JSON line
            rng = np.random.default_rng(0)
            nsamp = 256 if kind == "mse" else 16
            coords = [rng.integers(0, g, nsamp) for g in grid]
            flat = np.ravel_multi_index(coords, grid)
            parity = 0.0
            for src in sorted({0, world // 2, world - 1}):
                tuples_src = build_sequence(epg, kind, grid, src, world)[3] if src else tuples_at
                ref = epg_c.simulate(tuples_src(*coords), max_nstate=K_STATES - 1)
                rows = np.arange(sp.n_adc)
                if torch is not None:
                    slab_t = sig_t if src == 0 else (torch.view_as_complex(gather_bufs[src]) if gather_bufs else None)
                    if slab_t is None:
                        continue
                    got = slab_t[:, torch.as_tensor(flat, device=dev)].cpu().numpy()
                elif sig_bytes <= (1 << 30):
                    got = sig_buf.download(np.complex128, (sp.n_adc, sp.slab))[:, flat]
                else:   # the C3 signal is 16 GB: fetch single samples of the drawn voxels
                    rows = np.unique(np.linspace(0, sp.n_adc - 1, 16).astype(int))
                    got = np.zeros((sp.n_adc, nsamp), dtype=np.complex128)
                    one = np.empty(1, dtype=np.complex128)
                    for c, vx in enumerate(flat):
                        for r in rows:
                            _lib.check(ctx.lib.epgx_memcpy_d2h(ctx.handle, one.ctypes.data,
                                                               sig_ptr + 16 * (int(r) * sp.slab + int(vx)), 16))
                            got[r, c] = one[0]
                parity = max(parity, float(np.max(np.abs(got[rows] - ref[rows]))))
        except Exception as exc:   # noqa: BLE001
            parity, parity_error = None, repr(exc)

    def pmc(mode, key):
        """per-launch figures measured with rocprofv3 --pmc on this workload (separate passes, tools/prof.sh ->
        tools/collect_profiles.py -> profiles/traffic.json): "bytes" = HBM traffic from FETCH_SIZE / WRITE_SIZE,
        corrected as MI355X_MICROARCH.md prescribes; "fp64_flop_executed" from SQ_INSTS_VALU_{FMA,MUL,ADD}_F64"""
        try:
            t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))[args.workload][mode]
            return t[key] if world == 1 else None
        except (OSError, KeyError, ValueError):
            return None

    # the kernel a launch of each mode runs at K = 64 (epgx_run): four voxels per wavefront and 4 orders per lane for
    # state-resident launches of plain T / E / S(+-1) / ADC sequences, one wavefront per voxel when the state streams
    kernel_name = {"resident": "epgx::rows_kernel<1, 4, true>", "stream": "epgx::run_kernel<1, 1, true>"}
    if kind != "mse":
        kernel_name["resident"] = "epgx::rows_kernel<NSP, 4, true>"

    def roofline(mode):
        r = results[mode]
        ms = r["kernel_ms_per_launch"]
        if ms is None:   # --gather-in-step: derive from wall (includes the gather)
            ms = 1e3 * r["wall"] / (r["steps"] * n_launch[mode])
        units_per_launch = units_per_step / n_launch[mode]
        achieved = units_per_launch * B_ALG / (ms * 1e-3) / 1e9
        tflops = units_per_launch * FLOP_PER_UNIT / (ms * 1e-3) / 1e12
        fp64 = {"achieved": round(tflops, 2), "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(tflops / FP64_VALU_PEAK_TFLOPS, 4), "flop_per_unit": FLOP_PER_UNIT,
                "note": "nominal operator-by-operator flop count of SURVEY.md 8d; the kernel executes fewer (fusion, zero patterns)"}
        executed = pmc(mode, "fp64_flop_executed")
        if executed:
            ex_tflops = executed / (ms * 1e-3) / 1e12
            fp64.update({"executed_flop_per_unit": round(executed / units_per_launch, 1), "executed_achieved": round(ex_tflops, 2),
                         "executed_frac": round(ex_tflops / FP64_VALU_PEAK_TFLOPS, 4)})
        return {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc(mode, "bytes"),
                "kernel": kernel_name[mode], "launch_ms": round(ms, 4),
                "units_per_launch": int(units_per_launch), "alg_bytes_per_unit": B_ALG, "fp64": fp64}

    if rank == 0:
        main_r = results[args.mode]
        other = "stream" if args.mode == "resident" else "resident"
        out = {
            "metric": ("echo-points x voxels / sec (MSE, 20 echoes, 64 k-states)" if kind == "mse" else
                       f"echo-points x voxels / sec (MRF, {MRF_NTR} TR, 64 k-states)"),
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
                       "launches_per_step": n_launch[args.mode], "parallelism": f"voxel-slabs x{world}",
                       "collective": ("gather of the signal slabs to rank 0 inside every step" if (args.gather_in_step and can_gather)
                                      else "none in the timed region (voxel slabs are independent; signal stays in each GPU's HBM)")},
            "roofline": roofline(args.mode),
        }
        if other in results:
            out[other] = {"value": results[other]["value"], "ms_per_step": 1e3 * results[other]["wall"] / results[other]["steps"],
                    "steps": results[other]["steps"], "launches_per_step": n_launch[other], "roofline": roofline(other)}
        if config1 is not None:
            out["configs1"] = config1
        out["parity_max_abs_err_vs_oracle"] = parity
        if parity_error:
            out["parity_error"] = parity_error
        if gather_info is not None:
            out["gather"] = gather_info
        if not args.no_cpu_baseline and world == 1 and kind == "mse":
            threads = max(1, min(os.cpu_count() or 1, 16))
            side1 = max(64, args.cpu_side // 4)
            v1, t1, p1 = cpu_baseline(side1, 1, args.cpu_seconds / 3)
            vn, tn, pn = cpu_baseline(args.cpu_side, threads, args.cpu_seconds)
            out["cpu_baseline"] = {"value": vn, "unit": "echo*voxels/s", "cores": threads, "kind": "port",
                                   "sample": f"the same 20-echo MSE on a {args.cpu_side}x{args.cpu_side} (T1, T2) grid, "
                                             f"{pn} passes = {pn * NECHO * args.cpu_side ** 2} echo*voxels in {tn:.1f} s, "
                                             f"C oracle + OpenMP ({threads} threads); 1-thread leg: {side1}x{side1}, "
                                             f"{p1} passes in {t1:.1f} s",
                                   "value_1core": v1}
        if not args.no_cpu_baseline and world == 1 and kind == "mrf":
            threads = max(1, min(os.cpu_count() or 1, 16))
            vm, tm = cpu_baseline_mrf(16, threads)
            out["cpu_baseline"] = {"value": vm, "unit": "echo*voxels/s", "cores": threads, "kind": "port",
                                   "sample": f"the same {MRF_NTR}-TR MRF train on a 16x16x16 (T1, T2, B1) sub-grid of the same ranges, "
                                             f"1 pass = {MRF_NTR * 4096} TR*voxels in {tm:.1f} s, C oracle + OpenMP ({threads} threads)"}
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
