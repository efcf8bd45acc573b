#!/usr/bin/env python3
"""bench.py -- headline benchmark of the epgpy hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode resident|stream]
                    [--workload mse_1024|mse_256|mrf_100|mrf_32] [--scaling weak|strong]
                    [--only] [--no-cpu-baseline] [--no-extra-legs]

One "step" = one full pass of the hot path (the fused T/E/S/ADC kernel family of libepgx.so) over one batch of
synthetic input: the whole sequence of the workload over its parameter grid, i.e. what one `epg.simulate(seq)`
call computes.  The plan (operator stream + coefficient tables) is uploaded once before the timed region, so
inputs are resident in HBM when timing starts; the signal stays in HBM.

Metric (BASELINE.json): echo-points x voxels / s.  Printed by rank 0 as ONE JSON line.

mode "resident" (default): one launch per step, every voxel's state matrix stays in VGPRs for the whole sequence
  (real HBM traffic ~ 16 B per echo.voxel).  The bound of that kernel is fp64 vector issue, so its `roofline`
  object is {"bound": "fp64_valu", ...}: EXECUTED fp64 flop (PMC pass, profiles/traffic.json) per launch divided
  by the launch duration measured here with HIP events, against the 78.6 TFLOP/s vector peak.  SURVEY.md 8(d)'s
  algorithmic-byte figure travels next to it as `hbm_equiv` (it exceeds the HBM peak by construction: the state
  never moves).
mode "stream": one launch per echo, the state matrix [nvox][3][64] c128 is read and written once per launch -- the
  per-timestep kernel whose HBM roofline BASELINE.md quotes (B_alg = 2*64*3*16 + 16 = 6160 B per echo.voxel);
  `roofline` = {"bound": "hbm", ...} over the read+write launches only (the first launch of a pass starts from
  equilibrium and only writes the state: it is timed separately and excluded from both bytes and time).
Both modes are measured in every run; `value` is the mode selected with --mode, the other mode is reported in
an extra object.  Order per mode: event-timed launches (`launch_ms_before_timed_region`), W warm-up steps, K steps
by the host clock, event-timed launches again (`launch_ms`).  The first set also takes the chip out of the idle clocks
that plan compilation leaves it in: 3 warm-up launches (3 ms) alone do not (tools/headline_gap_probe.py).  Further objects of the default run (rank 0, N = 1): `configs1` (256 x 256), `configs3` (the
1000-TR MRF over 100^3 voxels, with its own roofline and parity check), `configs5` (PGSE, latency-labelled),
`e2e` (one whole `epg.simulate()` call, host buffers out), `cpu_baseline`.

N > 1 (launched by torch.distributed.run, one rank per GPU; torch.distributed runs over gloo and only carries barriers,
scalars and the communicator id -- the one RCCL in the process is the one libepgx loads):
  --scaling weak (default): every rank runs the workload on its own grid slab of the same size (the grid grows
    with N along its first axis).  Voxels never interact, so the timed region holds NO data-path collective.
  --scaling strong: the SAME grid is cut into N contiguous voxel slabs (BASELINE.json configs[3]); `value` is the
    kernel-only rate of the whole grid; next to it the rate with the gather of the slabs to rank 0 inside every step --
    libepgx's own RCCL gather -- both ways: `gather_serial` (all kernels, then one gather) and `gather_overlapped`
    (the slab in 4 sub-slabs, sub-slab k on the wire while k + 1 computes), and the gather alone (`gather`).
  A weak run with N > 1 measures the strong splits of mse_1024 and mrf_100 as well (`strong_mse_1024`, `strong_mrf_100`,
  each with `kernel_only`, `gather_serial`, `gather_overlapped`), after its headline and behind a watchdog.
"""
import argparse
import hashlib
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

K_STATES = 64
B_ALG = 2 * K_STATES * 3 * 16 + 16          # bytes per echo.voxel
FLOP_PER_UNIT = K_STATES * (66 + 2 * 14)    # nominal fp64 flop per echo.voxel
HBM_PEAK_GBS = 8000.0                       # MI355X_MICROARCH.md: 8 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6                # 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz
REFERENCE_AS_SHIPPED = 9.8e4                # reference NumPy path, 1 core, 256x256 MSE, K = 64
FLOP_PER_UNIT_64_ORDERS = 1862.4            # executed fp64 flop per echo.voxel of the kernel that computes all 64 orders at every echo
                                            # (rows_kernel<1, 4, true>, profiles/r03_resident_pmc.csv: 39.06 GFLOP per C2-L launch)


def csrc_hash():
    """fingerprint of the device code the PMC figures of profiles/traffic.json were measured on"""
    h = hashlib.sha256()
    base = os.path.join(ROOT, "epgpy_amd", "csrc")
    for name in sorted(os.listdir(base)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(base, name), "rb").read())
    return h.hexdigest()[:16]


def pmc_entry(workload, mode):
    """per-launch figures measured with rocprofv3 --pmc (separate passes; tools/prof.sh -> tools/collect_profiles.py ->
    profiles/traffic.json): HBM bytes from FETCH_SIZE / WRITE_SIZE corrected as MI355X_MICROARCH.md prescribes,
    executed fp64 flop from SQ_INSTS_VALU_{FMA,MUL,ADD}_F64; with the kernel name and the hash of the device
    sources they were measured on"""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))[workload][mode]
    except (OSError, KeyError, ValueError):
        return None


def host_info():
    model = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"nproc": os.cpu_count(), "cpus_usable": usable_cpus(), "cpu_model": model}


def _column(buf, enc, v):
    """column v of the device signal [n_adc rows][nvox] as a host vector (a strided 2-D download: 16 bytes per row)"""
    out = np.empty((enc.n_adc, 1), dtype=np.complex128)
    buf.download_2d(out, 0, 1, enc.n_adc, enc.nvox, offset=v)
    return out[:, 0]


def usable_cpus():
    """host cores this process may actually use: affinity mask and cgroup CPU quota (a GPU box hands a
    1-GPU job a share of its cores; running one thread per LISTED core would only oversubscribe that share)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline_mse(n_side, threads, budget_s, necho):
    """time the C oracle (oracle/epg_oracle.c, a port of the reference algorithm) on the same workload at
    n_side x n_side on the host cores: whole passes until `budget_s` seconds of wall time are spent.
    Returns (echo.voxels/s, seconds, passes)"""
    from oracle import epg_c, workloads as ow

    T1 = np.linspace(200, 3000, n_side)[:, None]
    T2 = np.linspace(20, 300, n_side)[None, :]
    tuples = ow.mse_tuples(T1, T2)
    compiled = epg_c.compile_ops(tuples, (n_side, n_side))
    epg_c.simulate(tuples[:8], max_nstate=63, nthreads=threads)  # warm the library
    passes, t0 = 0, time.perf_counter()
    while passes == 0 or (time.perf_counter() - t0 < budget_s and passes < 64):
        epg_c.simulate(tuples, max_nstate=63, nthreads=threads, compiled=compiled)
        passes += 1
    dt = time.perf_counter() - t0
    return passes * necho * n_side * n_side / dt, dt, passes


def cpu_baseline_mrf(m, threads, ntr):
    """the MRF workload is timed on an m^3 sub-grid of the same parameter ranges (C oracle, one pass)
    and reported per TR.voxel.  Returns (TR.voxels/s, seconds)"""
    from oracle import epg_c, workloads as ow

    T1 = np.linspace(300, 3000, m)[:, None, None]
    T2 = np.linspace(20, 300, m)[None, :, None]
    B1 = np.linspace(0.7, 1.3, m)[None, None, :]
    alpha, TR = ow.mrf_trains(ntr)
    tuples = ow.mrf_tuples(T1, T2, B1, alpha, TR)
    compiled = epg_c.compile_ops(tuples, (m, m, m))   # table preparation (Python) is not part of the timed pass
    epg_c.simulate(tuples[:8], max_nstate=63, nthreads=threads)
    t0 = time.perf_counter()
    epg_c.simulate(tuples, max_nstate=63, nthreads=threads, compiled=compiled)
    dt = time.perf_counter() - t0
    return ntr * m ** 3 / dt, dt


def oracle_tuples(kind, params, coords):
    """oracle-side description of the workload at the drawn voxels"""
    from oracle import workloads as ow

    if kind == "mse":
        T1, T2 = params
        return ow.mse_tuples(T1[coords[0], 0], T2[0, coords[1]])
    T1, T2, B1 = params
    alpha, TR = ow.mrf_trains()
    return ow.mrf_tuples(T1[coords[0], 0, 0], T2[0, coords[1], 0], B1[0, 0, coords[2]], alpha, TR)


def kind_of(workload):
    from epgpy_amd import workloads as wl

    return wl.GRIDS[workload][0]


def wake(ctx, launch, ms=25.0):
    """keep the chip under load for `ms` milliseconds before a short event-timed loop: host-side preparation (plan
    compilation, table upload) leaves it in its idle clocks, and one warm-up launch of a sub-millisecond kernel does not
    bring it back (tools/headline_gap_probe.py)"""
    t0 = time.perf_counter()
    while 1e3 * (time.perf_counter() - t0) < ms:
        launch()
        ctx.synchronize()


class Leg:
    """one workload bound to this rank's device: plan, signal buffer, state, timing helpers"""

    def __init__(self, epg, lib, name, device, rank=0, world=1, scaling="weak", fuse=True, alloc_signal=True):
        from epgpy_amd import workloads as wl
        from epgpy_amd.distributed import ShardedPlan

        self.name, self.lib = name, lib
        self.kind, self.grid = wl.GRIDS[name]
        n1 = self.grid[0]
        rows = (rank * n1, n1, n1 * world) if scaling == "weak" else None
        self.rows = rows
        seq, self.params, self.n_adc, opts = wl.build(epg, name, rows)
        if scaling == "strong":    # rank r of N simulates slab r of the SAME grid
            self.sp = ShardedPlan(seq, rank=rank, world_size=world, device=device, fuse=fuse, **opts)
        else:                      # every rank simulates its own full grid
            self.sp = ShardedPlan(seq, rank=0, world_size=1, device=device, fuse=fuse, **opts)
        self.sp.bind()             # the library's own stream: buffers, kernels, gather and events are ordered on it
        self.ctx = self.sp._ctx
        self.nvox = self.sp.count
        self.units_per_step = self.n_adc * self.nvox
        self.sig_bytes = 16 * self.sp.n_adc * self.sp.slab
        self.own_sig, self.sig_ptr = None, None     # (alloc_signal=False: the caller points sig_ptr at its own buffer)
        if alloc_signal:
            self.own_sig = lib.DeviceBuffer(self.ctx, max(self.sig_bytes, 16))
            self.sig_ptr = self.own_sig.ptr.value
        self.state = None
        self.n_seg = len(self.sp.segments())

    def step(self, mode, segments=None):
        if mode == "stream" and self.state is None:
            self.state = self.sp.new_state()
        self.sp.run(self.sig_ptr, mode=mode, state=self.state, segments=segments)

    def kernel_ms(self, mode, steps):
        """average duration of ONE launch, HIP events on the launch stream.  stream: over the read+write launches
        only; the write-only first launch of every pass is timed on its own.  Returns (ms, first-launch ms or None)"""
        ctx = self.ctx
        ctx.synchronize()
        if mode == "resident":
            ctx.timer_start()
            for _ in range(steps):
                self.step(mode)
            return ctx.timer_stop() / steps, None
        rest, first = 0.0, 0.0
        for _ in range(steps):
            ctx.timer_start()
            self.step(mode, segments=(0, 1))
            first += ctx.timer_stop()
            ctx.timer_start()
            self.step(mode, segments=(1, self.n_seg))
            rest += ctx.timer_stop()
        return rest / (steps * max(self.n_seg - 1, 1)), first / steps

    def fetch(self, rows, flat):
        """signal[rows][:, flat] of this rank's buffer (single samples when the buffer is large)"""
        lib, ctx, ld = self.lib, self.ctx, self.sp.slab
        out = np.zeros((len(rows), len(flat)), dtype=np.complex128)
        if self.sig_bytes <= (1 << 30) and self.own_sig is not None:
            return self.own_sig.download(np.complex128, (self.sp.n_adc, ld))[np.ix_(rows, flat)]
        one = np.empty(1, dtype=np.complex128)
        for i, r in enumerate(rows):
            for c, vx in enumerate(flat):
                lib.check(ctx.lib.epgx_memcpy_d2h(ctx.handle, one.ctypes.data, self.sig_ptr + 16 * (int(r) * ld + int(vx)), 16))
                out[i, c] = one[0]
        return out

    def parity(self, nsamp=None):
        """max |GPU - oracle| over what the last step wrote (oracle as the checker only), and the number of voxels compared.
        MSE grids: EVERY voxel and echo of this rank's slab -- the C oracle runs the whole grid on the host cores in a fraction
        of a second.  MRF (1000 TR): four contiguous ranges of 4096 voxels (both ends of the launch and two in between: every
        offset inside a wave group and a workgroup), all repetitions.  `nsamp`: random voxels instead (small rehearsal runs)"""
        from oracle import epg_c, workloads as ow

        threads = usable_cpus()
        sp = self.sp
        if nsamp is None and self.kind == "mse" and self.own_sig is not None and self.sig_bytes <= (2 << 30):
            T1, T2 = self.params
            ref = epg_c.simulate(ow.mse_tuples(T1, T2), max_nstate=K_STATES - 1, nthreads=threads).reshape(sp.n_adc, -1)
            got = self.own_sig.download(np.complex128, (sp.n_adc, sp.slab))[:, :sp.count]
            return float(np.max(np.abs(got - ref[:, sp.vox0:sp.vox0 + sp.count]))), int(sp.count)
        if nsamp is None and self.own_sig is not None and sp.count >= 4 * 4096:
            worst, block = 0.0, np.zeros((sp.n_adc, 4096), dtype=np.complex128)
            for first in (0, sp.count // 3 + 17, 2 * sp.count // 3 + 501, sp.count - 4096):
                coords = list(np.unravel_index(sp.vox0 + np.arange(first, first + 4096), self.grid))
                ref = epg_c.simulate(oracle_tuples(self.kind, self.params, coords), max_nstate=K_STATES - 1, nthreads=threads)
                self.own_sig.download_2d(block, 0, 4096, sp.n_adc, sp.slab, offset=first)
                worst = max(worst, float(np.max(np.abs(block - ref))))
            return worst, 4 * 4096
        rng = np.random.default_rng(0)
        nsamp = nsamp or 64
        coords = [rng.integers(0, g, nsamp) for g in self.grid]
        flat = np.ravel_multi_index(coords, self.grid) - self.sp.vox0
        keep = (flat >= 0) & (flat < self.sp.count)
        if not keep.any():
            return None, 0
        coords = [c[keep] for c in coords]
        ref = epg_c.simulate(oracle_tuples(self.kind, self.params, coords), max_nstate=K_STATES - 1)
        rows = np.arange(self.sp.n_adc) if self.sig_bytes <= (1 << 30) else np.unique(np.linspace(0, self.sp.n_adc - 1, 16).astype(int))
        got = self.fetch(rows, flat[keep])
        return float(np.max(np.abs(got - ref[rows]))), int(keep.sum())

    def free(self):
        if self.own_sig is not None:
            self.own_sig.free()
        self.state = None


def kernel_names(kind):
    """the kernel a launch of each mode runs at K = 64 (epgx_run): four voxels per wavefront and 4 orders per lane for
    state-resident launches of plain T / E / S(+-1) / ADC sequences, one wavefront per voxel when the state streams"""
    nsp = 1 if kind == "mse" else 2
    # (an echo train from equilibrium walks its records in phases of 1 / 2 / 4 orders per lane while the state matrix grows)
    resident = f"epgx::rows_grow_kernel<{nsp}>" if kind == "mse" else f"epgx::rows_kernel<{nsp}, 4, true>"
    return {"resident": resident, "stream": f"epgx::run_kernel<1, {nsp}, true>"}


def roofline(workload, kind, mode, launch_ms, units_per_launch, live_hash):
    """the roofline object of the dominant kernel of `mode` (see the module docstring)"""
    pmc = pmc_entry(workload, mode) or {}
    stale = bool(pmc) and pmc.get("csrc_hash") != live_hash
    seconds = launch_ms * 1e-3
    hbm_equiv = units_per_launch * B_ALG / seconds / 1e9
    traffic = pmc.get("bytes")
    out = {"kernel": pmc.get("kernel") or kernel_names(kind)[mode], "launch_ms": round(launch_ms, 4),
           "units_per_launch": int(units_per_launch), "traffic": traffic,
           "pmc_source": pmc.get("source"), "pmc_stale": stale if pmc else None}
    if mode == "stream":
        out.update({"bound": "hbm", "achieved": round(hbm_equiv, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(hbm_equiv / HBM_PEAK_GBS, 4), "alg_bytes_per_unit": B_ALG,
                    "note": "read+write launches only; the write-only first launch of a pass is excluded from bytes and time"})
        if traffic:
            out["traffic_over_algorithmic"] = round(traffic / (units_per_launch * B_ALG), 4)
        return out
    executed = pmc.get("fp64_flop_executed")
    achieved = executed / seconds / 1e12 if executed else None
    out.update({"bound": "fp64_valu", "achieved": round(achieved, 2) if achieved else None, "peak": FP64_VALU_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": round(achieved / FP64_VALU_PEAK_TFLOPS, 4) if achieved else None,
                "executed_flop_per_unit": round(executed / units_per_launch, 1) if executed else None,
                "nominal_flop_per_unit": FLOP_PER_UNIT,
                "hbm_equiv": {"alg_bytes_per_unit": B_ALG, "GB/s": round(hbm_equiv, 1), "x_hbm_peak": round(hbm_equiv / HBM_PEAK_GBS, 3),
                              "note": "SURVEY.md 8d's streaming-model bytes; the state never leaves the register file, so this is not a bandwidth"},
                "hbm_measured": ({"GB/s": round(traffic / seconds / 1e9, 1), "frac": round(traffic / seconds / 1e9 / HBM_PEAK_GBS, 4)}
                                 if traffic else None)})
    if achieved is not None:
        assert achieved <= FP64_VALU_PEAK_TFLOPS * 1.02, "executed fp64 rate above the vector peak: stale PMC data?"
    if kind == "mse" and "grow" in str(out["kernel"]):
        # the kernel walks the echo train in phases of 16 / 32 / 64 orders per voxel (the state matrix grows by two orders per echo, as
        # in the reference): it EXECUTES fewer flop for the same result, so `frac` fell while the launch got faster.  The same
        # launch priced with the flop of the fixed-capacity formulation it replaced:
        eq = units_per_launch * FLOP_PER_UNIT_64_ORDERS / seconds / 1e12
        out["fixed_capacity_equiv"] = {"flop_per_unit": FLOP_PER_UNIT_64_ORDERS, "TFLOP/s": round(eq, 2), "frac": round(eq / FP64_VALU_PEAK_TFLOPS, 4),
                                       "note": "executed flop of rows_kernel<1, 4, true> (all 64 orders at every echo; r03) over THIS launch's time: "
                                               "compare instructions and time, not the rate"}
        if pmc.get("valu_instructions") and pmc.get("fp64_instructions"):
            f64 = sum(pmc["fp64_instructions"].values())
            # issue model of profiles/README.md: an fp64 instruction holds a SIMD for 4 cycles, the other vector instructions for 2
            cycles = (4.0 * f64 + 2.0 * (pmc["valu_instructions"] - f64)) / 1024.0
            out["valu_issue"] = {"vector_instructions": int(pmc["valu_instructions"]), "fp64_instructions": int(f64),
                                 "busy_frac_at_2.4GHz": round(cycles / (seconds * 2.4e9), 4),
                                 "note": "issue cycles per SIMD (4 per fp64 instruction, 2 per other vector instruction) over launch time x 2.4 GHz"}
            if pmc.get("grbm_gui_active") and pmc.get("rocprof_avg_launch_ms"):
                # the clock the chip held during the profiled launches (GRBM_GUI_ACTIVE is the sum over the 8 XCDs), and the same
                # issue cycles over THIS launch's time at that clock
                clock = pmc["grbm_gui_active"] / 8.0 / (pmc["rocprof_avg_launch_ms"] * 1e-3)
                out["valu_issue"]["clock_GHz_under_pmc"] = round(clock / 1e9, 3)
                out["valu_issue"]["busy_frac_at_pmc_clock"] = round(cycles / (seconds * clock), 4)
    return out


def crash_guard(json_fd):
    """N > 1 runs, rank 0: a child forked BEFORE anything in this process touches the GPU keeps the line as it stands after
    the headline measurement (messages `P <json>`) and prints it if this process ends without having printed its own
    (message `F`) -- a fault inside the strong-scaling legs (the communicator and the gather have only ever run with one
    rank on hardware) then costs those legs, not the weak-scaling figure.  Hangs are the watchdog's business; this is for
    the process dying.  The child does pipe I/O only.  Returns the write end."""
    rd, wr = os.pipe()
    if os.fork() == 0:
        os.close(wr)
        try:
            os.setsid()          # (the launcher signals its workers' group when one of them fails)
        except OSError:
            pass
        data = b""
        while True:
            chunk = os.read(rd, 1 << 16)
            if not chunk:
                break
            data += chunk
        lines = data.split(b"\n")
        held = [ln[2:] for ln in lines if ln.startswith(b"P ")]
        if b"F" not in lines and held:
            os.write(json_fd, held[-1] + b"\n")
        os._exit(0)
    os.close(rd)
    return wr


def launched_by():
    """who started this process: "self" (bench.py's own launcher below), "torchrun" (torch.distributed.run set the rendezvous
    variables), or "none" (a plain one-process run)"""
    if os.environ.get("EPGX_BENCH_LAUNCHER") == "self":
        return "self"
    return "torchrun" if "WORLD_SIZE" in os.environ else "none"


def self_launch(n, argv):
    """`python bench.py --gpus N` with no launcher around it (the way the driver starts the N = 1 run): THIS process starts the
    N rank processes itself -- the reference's own parallel attempt is a self-contained pool as well
    (epgpy/functions.py:195-248) -- and only waits for them.  It never touches a GPU (no HIP call, no library load), so
    nothing is inherited by the ranks but the environment: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT as
    torch.distributed.run would set them.  Rank 0 inherits this process's stdout (the ONE JSON line goes straight
    through), the other ranks write to stderr.  Returns the exit code: 0 only if every rank returned 0; the first rank that
    fails takes the others down (exact PIDs), so a dead rank costs seconds, not a rendezvous timeout."""
    import signal
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), EPGX_BENCH_LAUNCHER="self")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else sys.stderr))

    def stop(*_):
        for p in procs:
            if p.poll() is None:
                p.terminate()

    for signum in (signal.SIGTERM, signal.SIGINT):
        signal.signal(signum, lambda *_, s=signum: (stop(), sys.exit(128 + s)))
    code, live = 0, set(range(n))
    while live:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0 and code == 0:
                code = rc if rc > 0 else 1
                print(f"bench.py: rank {r} exited with {rc}; stopping the other ranks", file=sys.stderr, flush=True)
                stop()
                deadline = time.time() + 10.0
                while time.time() < deadline and any(p.poll() is None for p in procs):
                    time.sleep(0.05)
                for p in procs:
                    if p.poll() is None:
                        p.kill()
        time.sleep(0.02)
    return code


def stub_leg(args):
    """--stub-leg (CPU tests of the launcher and of the line's bookkeeping): the control flow of a run -- rendezvous over gloo,
    barriers on both sides of the timed region, max over ranks, ONE line from rank 0 -- with a sleep where the kernels would be"""
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if args.stub_fail_rank is not None and rank == args.stub_fail_rank:
        raise SystemExit(3)
    sys.stdout.flush()
    json_fd = os.dup(1)          # as in main(): stdout carries the line only (gloo prints its connection banner from C)
    os.dup2(2, 1)
    dist = None
    if "WORLD_SIZE" in os.environ:
        import datetime

        import torch
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", timeout=datetime.timedelta(minutes=2))
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(1e-3 * (1 + rank))
    wall = time.perf_counter() - t0
    if dist is not None:
        ten = torch.tensor([wall], dtype=torch.float64)
        dist.all_reduce(ten, op=dist.ReduceOp.MAX)
        wall = float(ten.item())
    if rank == 0:
        os.write(json_fd, (json.dumps({"metric": "stub", "value": world * args.steps / wall, "unit": "steps/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": 1e3 * wall / args.steps, "launcher": launched_by(), "rccl_ranks": None,
                          "stub": True}) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    # `--gpus N` with no launcher around this process: start the N ranks here (before anything could touch a GPU) and relay
    pre = argparse.ArgumentParser(add_help=False)
    pre.add_argument("--gpus", type=int, default=1)
    ngpus = pre.parse_known_args()[0].gpus
    if ngpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(ngpus, sys.argv[1:]))

    guard_fd = None
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and int(os.environ.get("RANK", "0")) == 0 and "--stub-leg" not in sys.argv:
        sys.stdout.flush()
        guard_fd = crash_guard(os.dup(1))

    from epgpy_amd import workloads as wl

    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", choices=["resident", "stream"], default="resident")
    ap.add_argument("--workload", choices=sorted(n for n, (k, _) in wl.GRIDS.items() if k != "pgse"), default="mse_1024")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--extra-timeout", type=float, default=240.0,
                    help="N>1 weak runs: seconds the strong_mrf_100 leg (communicator + gather) may take before the line is printed without it")
    ap.add_argument("--one-gpu", action="store_true",
                    help="rehearsal on a one-GPU box: every rank uses GPU 0 (the control flow of an N > 1 run -- barriers, status flags, "
                         "watchdog, the strong legs without a communicator: RCCL refuses two ranks on one GPU; no scaling figure)")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip configs1 / configs3 / configs5 / e2e / strong_mrf_100")
    ap.add_argument("--only", action="store_true", help="measure only --mode of --workload (profiling runs)")
    ap.add_argument("--cpu-side", type=int, default=1024, help="CPU baseline grid side (default: the workload's own)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline time budget (all-thread leg)")
    ap.add_argument("--no-fuse", action="store_true", help="keep E . T . E as three operators (A/B measurements)")
    ap.add_argument("--stub-leg", action="store_true", help=argparse.SUPPRESS)          # (tests/test_host.py: launcher + line bookkeeping, no GPU)
    ap.add_argument("--stub-fail-rank", type=int, default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.stub_leg:
        return stub_leg(args)
    if args.only:
        args.no_extra_legs = args.no_cpu_baseline = True

    # stdout carries ONE JSON line: libraries that print banners from C (RCCL prints its version block on the first
    # communicator) get stderr instead; the line itself goes to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.one_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    from epgpy_amd import epg, _lib
    from epgpy_amd.distributed import SlabGather, torch_id_exchange

    dist = torch = None
    if "WORLD_SIZE" in os.environ:   # launched by torch.distributed.run (also with one rank)
        import datetime

        import torch
        import torch.distributed as dist

        # the control plane (barriers, max-over-ranks, status flags, the 128-byte communicator id) runs over gloo: torch's
        # NCCL backend would bring a second HIP runtime and a second RCCL into the process for the sake of a few scalars
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", timeout=datetime.timedelta(minutes=10))
        dev = torch.device("cpu")

    live_hash = csrc_hash()

    def sync(ctx):
        ctx.synchronize()      # (all device work of this process is on the library's streams)

    def barrier():
        if dist is not None:
            dist.barrier()

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def all_ok(ok):
        """every rank reports whether its local preparation worked: nobody enters a collective alone"""
        if dist is None:
            return ok
        t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item() > 0.5)

    def timed(leg, mode, steps, warmup, after_step=None):
        """W warm-up steps, then EXACTLY `steps` steps between barrier + synchronize on both sides; max over ranks"""
        for _ in range(warmup):
            leg.step(mode)
            if after_step:
                after_step()
        sync(leg.ctx); barrier(); sync(leg.ctx)
        t0 = time.perf_counter()
        for _ in range(steps):
            leg.step(mode)
            if after_step:
                after_step()
        sync(leg.ctx); barrier(); sync(leg.ctx)
        return max_over_ranks(time.perf_counter() - t0)

    # ------------------------------------------------------------------ the main workload
    comm, comm_error, rccl_ranks = None, None, None

    def make_comm():
        """libepgx's own RCCL communicator (torch.distributed carries the 128-byte id).  Only the strong-scaling legs
        gather anything, so a weak-scaling run creates it AFTER its headline measurement (nothing that could go wrong
        here can then cost the driver its weak-scaling line)"""
        nonlocal comm, comm_error, rccl_ranks
        if world == 1 or comm is not None:
            return
        try:
            comm = _lib.Comm(_lib.get_context(local_rank), rank, world, torch_id_exchange())
            rccl_ranks = comm.count()          # ncclCommCount: the ranks RCCL itself sees in this communicator
        except Exception as exc:   # noqa: BLE001  (the kernel-only measurement does not need it)
            comm, comm_error = None, repr(exc)
        if not all_ok(comm is not None):
            comm, comm_error = None, comm_error or "another rank could not create its communicator"

    if args.scaling == "strong":
        make_comm()

    def strong_leg(workload, steps, warmup, fuse=True):
        """BASELINE.json configs[3]: the SAME grid cut into `world` slabs, the signal gathered to rank 0.
        Returns (leg, info dict): the kernel-only rate, the gather alone, and the rate with the gather inside every step --
        serial (all kernels, then the gather) and overlapped (sub-slab k travels while k + 1 computes)"""
        info = {"workload": workload, "scaling": "strong", "n_gpus": world}
        gather = None
        leg = None
        try:
            if comm is not None:
                leg = Leg(epg, _lib, workload, local_rank, rank, world, "strong", fuse, alloc_signal=False)
                gather = SlabGather(leg.sp, comm, root=0, nsub=4)     # the root's slab is produced inside the gathered buffer
                leg.sig_ptr = gather.local_ptr
            else:
                leg = Leg(epg, _lib, workload, local_rank, rank, world, "strong", fuse)
            ok = True
        except Exception as exc:   # noqa: BLE001
            info["error"] = repr(exc)
            ok = False
        if not all_ok(ok):
            info.setdefault("error", "another rank could not set the leg up")
            return None, info
        total_units = leg.n_adc * leg.sp.nvox          # the whole grid, all ranks together
        before_ms, _ = leg.kernel_ms("resident", 20 if leg.kind == "mse" else 3)     # (out of the idle clocks: see the weak leg)
        info["kernel_ms_rank0_before_timed_region"] = round(before_ms, 4)
        wall = timed(leg, "resident", steps, warmup)
        info.update({"value": total_units * steps / wall, "ms_per_step": 1e3 * wall / steps, "steps": steps,
                     "voxels_per_gpu": leg.nvox, "voxels_total": leg.sp.nvox})
        info["kernel_only"] = {"value": info["value"], "ms_per_step": info["ms_per_step"], "steps": steps}
        ms, _ = leg.kernel_ms("resident", max(1, min(steps, 5)))
        info["kernel_ms_rank0"] = round(ms, 4)
        if gather is not None:
            sync(leg.ctx); barrier()
            t0 = time.perf_counter()
            gather()
            sync(leg.ctx); barrier()
            dt = max_over_ranks(time.perf_counter() - t0)
            gb = (world - 1) * gather.block / 1e9
            info["gather"] = {"ms": round(1e3 * dt, 3), "GB_to_rank0": round(gb, 3), "GB_per_s": round(gb / dt, 1) if dt > 0 else None,
                              "how": "epgx_comm_gather: ncclSend / ncclRecv in one group, every peer over its own xGMI link"}
            nin = max(2, steps // 4)

            class _Whole:     # `timed` drives an object with step(): one whole step = kernels + gather
                ctx = leg.ctx

                def __init__(self, fn):
                    self.fn = fn

                def step(self, mode):
                    self.fn()

            for key, fn in (("gather_serial", gather.run_serial), ("gather_overlapped", gather.run_overlapped)):
                wall_in = timed(_Whole(fn), "resident", nin, 1)
                info[key] = {"value": total_units * nin / wall_in, "ms_per_step": 1e3 * wall_in / nin, "steps": nin,
                             "sub_slabs": gather.nsub}
            info["gather_inclusive"] = info["gather_overlapped"]
            if rank == 0:   # gathered blocks of the first / middle / last rank against the oracle (layout of the last run:
                try:        # sub-slabs [n_adc][sub] one after the other inside every rank's block)
                    from oracle import epg_c

                    worst = 0.0
                    rng = np.random.default_rng(1)
                    for src in sorted({0, world // 2, world - 1}):
                        v0 = src * leg.sp.slab
                        cnt = min(leg.sp.slab, leg.sp.nvox - v0)
                        pick = v0 + rng.integers(0, cnt, 8)
                        coords = list(np.unravel_index(pick, leg.grid))
                        ref = epg_c.simulate(oracle_tuples(leg.kind, leg.params, coords), max_nstate=K_STATES - 1)
                        rows = np.unique(np.linspace(0, leg.sp.n_adc - 1, 8).astype(int))
                        one = np.empty(1, dtype=np.complex128)
                        base = gather.gathered.ptr.value + src * gather.block
                        for i, r in enumerate(rows):
                            for c, vx in enumerate(pick - v0):
                                k, col = divmod(int(vx), gather.sub)
                                _lib.check(leg.ctx.lib.epgx_memcpy_d2h(leg.ctx.handle, one.ctypes.data,
                                                                       base + k * gather.sub_bytes + 16 * (int(r) * gather.sub + col), 16))
                                worst = max(worst, abs(one[0] - ref[r, c]))
                    info["gathered_parity_max_abs_err_vs_oracle"] = float(worst)
                except Exception as exc:   # noqa: BLE001
                    info["gathered_parity_error"] = repr(exc)
        elif world > 1:
            info["gather"] = {"error": f"no RCCL communicator: {comm_error}"}
        if world > 1:
            # what a caller with a HOST array waits for on N GPUs: one whole simulate_sharded(out="host") call, plan compilation
            # included -- the result in shared memory that every rank fills over its own PCIe link (via="pcie"; complex128 and
            # complex64 records), against the device-side gather + download through rank 0's one link (via="rccl")
            from epgpy_amd.distributed import simulate_sharded

            seq_h, _, _, opts_h = wl.build(epg, workload)
            routes = [("host_pcie", dict(via="pcie")), ("host_pcie_c64", dict(via="pcie", dtype=np.complex64))]
            if comm is not None:
                routes += [("host_rccl", dict(via="rccl")), ("host_rccl_c64", dict(via="rccl", dtype=np.complex64))]
            for key, kw in routes:
                try:
                    res = simulate_sharded(seq_h, device=local_rank, **kw, **opts_h)       # (first call: mappings, staging ring)
                    del res
                    laps = []
                    for _ in range(2):
                        barrier()
                        t0 = time.perf_counter()
                        res = simulate_sharded(seq_h, device=local_rank, **kw, **opts_h)
                        barrier()
                        laps.append(max_over_ranks(time.perf_counter() - t0))
                        nbytes = res.nbytes if res is not None else 0
                        del res
                    dt = min(laps)
                    info[key] = {"simulate_sharded_ms": round(1e3 * dt, 2), "value": total_units / dt, "result_GB": round(nbytes / 1e9, 3),
                                 "GB_per_s": round(nbytes / 1e9 / dt, 1)}
                except Exception as exc:   # noqa: BLE001
                    info[key] = {"error": repr(exc)}
                if not all_ok(True):
                    break
        info["_gather_obj"] = gather
        return leg, info

    results, extra = {}, {}
    if args.scaling == "strong":
        leg, sinfo = strong_leg(args.workload, args.steps, args.warmup, not args.no_fuse)
        if leg is None:
            raise SystemExit(f"strong-scaling leg failed: {sinfo.get('error')}")
        gobj = sinfo.pop("_gather_obj", None)
        results["resident"] = {"value": sinfo["value"], "wall": sinfo["ms_per_step"] * sinfo["steps"] / 1e3, "steps": sinfo["steps"],
                               "launch_ms": sinfo["kernel_ms_rank0"], "first_ms": None}
        for key in ("kernel_only", "gather", "gather_serial", "gather_overlapped", "gathered_parity_max_abs_err_vs_oracle", "gathered_parity_error"):
            if key in sinfo:
                extra[key] = sinfo[key]
        args.mode = "resident"
    else:
        leg = Leg(epg, _lib, args.workload, local_rank, rank, world, "weak", not args.no_fuse)
        for mode in ("resident", "stream"):
            if args.only and mode != args.mode:
                continue
            steps = args.steps if mode == args.mode else max(3, args.steps // 4)
            # The event-timed launches of the roofline object run on BOTH sides of the wall-clock region.  The first set is
            # also what takes the chip out of its idle clocks: plan compilation and table upload leave the GPU idle for
            # ~0.5 s, after which 3 warm-up launches (3 ms) are not enough to be back at the sustained clock -- 20 launches
            # are (tools/headline_gap_probe.py: 0.980 ms per step with 3 warm-up launches after an idle half second,
            # 0.897 with 20; no idle: 0.895 either way).  Both event times travel in the line.
            before_ms, _ = leg.kernel_ms(mode, 20 if (mode == "resident" and kind_of(args.workload) == "mse") else 2)
            wall = timed(leg, mode, steps, args.warmup)
            launch_ms, first_ms = leg.kernel_ms(mode, max(1, min(steps, 10)))
            results[mode] = {"value": leg.units_per_step * world * steps / wall, "wall": wall, "steps": steps,
                             "launch_ms": launch_ms, "first_ms": first_ms, "before_ms": before_ms}
    kind, grid = leg.kind, leg.grid
    n_launch = {"resident": 1, "stream": leg.n_seg}

    # parity spot check of what was just computed (rank 0, oracle as checker only)
    parity = parity_error = parity_voxels = None
    if rank == 0:
        try:
            leg.step("resident")
            leg.ctx.synchronize()
            parity, parity_voxels = leg.parity()
        except Exception as exc:   # noqa: BLE001  (a failing check must not cost the measurement its JSON line)
            parity_error = repr(exc)

    def roof(mode):
        r = results[mode]
        # resident: one launch = the whole sequence over the rank's voxels; stream: one echo of every voxel per launch
        upl = leg.units_per_step if mode == "resident" else leg.nvox
        out = roofline(args.workload, kind, mode, r["launch_ms"], upl, live_hash)
        if mode == "stream" and r["first_ms"] is not None:
            out["first_launch_ms"] = round(r["first_ms"], 4)
        if r.get("before_ms") is not None:
            out["launch_ms_before_timed_region"] = round(r["before_ms"], 4)
            out["order"] = ("event-timed launches (launch_ms_before_timed_region: they also take the chip out of its idle clocks), "
                            "W warm-up steps, K wall-clock steps, event-timed launches (launch_ms)")
        return out

    # ------------------------------------------------------------------ extra legs (rank 0 of a 1-GPU run)
    single = world == 1 and rank == 0 and not args.no_extra_legs
    if single and kind == "mse" and args.workload != "mse_256":
        # BASELINE.json configs[1] itself (256 x 256): 65 536 voxels do not fill the chip for long enough to be a
        # roofline measurement (a resident launch is ~0.1 ms), but its rate is reported too
        try:
            l1 = Leg(epg, _lib, "mse_256", local_rank, fuse=not args.no_fuse)
            c1 = {"workload": "mse_256 (BASELINE.json configs[1]): the same sequence over 256x256 (T1, T2)"}
            for mode1 in ("resident", "stream"):
                wake(l1.ctx, lambda: l1.step(mode1))
                l1.ctx.timer_start()
                for _ in range(50):
                    l1.step(mode1)
                ms1 = l1.ctx.timer_stop() / 50
                c1[mode1] = {"ms_per_step": round(ms1, 4), "value": l1.units_per_step / (ms1 * 1e-3)}
            l1.step("resident")
            c1["parity_max_abs_err_vs_oracle"], c1["parity_voxels"] = l1.parity()
            l1.free()
            extra["configs1"] = c1
        except Exception as exc:   # noqa: BLE001
            extra["configs1"] = {"error": repr(exc)}
    if single and args.workload != "mrf_100":
        # BASELINE.json configs[2]: 1000-TR MRF over 100^3 (T1, T2, B1) voxels, state-resident (a 16 GB signal)
        try:
            t0 = time.perf_counter()
            l3 = Leg(epg, _lib, "mrf_100", local_rank, fuse=not args.no_fuse)
            build_s = time.perf_counter() - t0
            l3.step("resident"); l3.ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                l3.step("resident")
            l3.ctx.synchronize()
            wall3 = time.perf_counter() - t0
            ms3, _ = l3.kernel_ms("resident", 2)
            c3 = {"workload": "mrf_100 (BASELINE.json configs[2]): 1000-TR variable-FA SSFP over 100x100x100 (T1, T2, B1), max_nstate=63",
                  "mode": "resident", "steps": 3, "ms_per_step": round(1e3 * wall3 / 3, 3), "value": 3 * l3.units_per_step / wall3,
                  "unit": "TR*voxels/s", "plan_build_s": round(build_s, 2), "signal_GB_in_HBM": round(l3.sig_bytes / 1e9, 2),
                  "roofline": roofline("mrf_100", "mrf", "resident", ms3, l3.units_per_step, live_hash)}
            c3["parity_max_abs_err_vs_oracle"], c3["parity_voxels"] = l3.parity()
            l3.free()
            del l3
            # what a caller of this config waits for: one whole epg.simulate() -- the signal left on the device, and downloaded
            # into a 16 GB NumPy array (PCIe-inclusive; only with plenty of free host memory)
            try:
                import psutil

                seq3, _, n3, opts3 = wl.build(epg, "mrf_100")
                epg.simulate(seq3, out="device", **opts3)
                t0 = time.perf_counter()
                epg.simulate(seq3, out="device", **opts3)
                c3["simulate_call_s"] = {"out_device": round(time.perf_counter() - t0, 4)}
                if psutil.virtual_memory().available > 96e9:
                    t0 = time.perf_counter()
                    res3 = epg.simulate(seq3, **opts3)
                    dt3 = time.perf_counter() - t0
                    c3["simulate_call_s"].update({"numpy_result": round(dt3, 4), "result_GB": round(res3.nbytes / 1e9, 2),
                                                  "GB_per_s": round(res3.nbytes / 1e9 / dt3, 1), "pcie_floor_s": round(res3.nbytes / 54e9, 3)})
                    del res3
                    t0 = time.perf_counter()
                    res3 = epg.simulate(seq3, dtype=np.complex64, **opts3)       # complex64 records: half the bytes over PCIe
                    dt3 = time.perf_counter() - t0
                    c3["simulate_call_s"].update({"numpy_result_c64": round(dt3, 4), "result_GB_c64": round(res3.nbytes / 1e9, 2)})
                    del res3
                # (no release of the context's cached device blocks here: after a hipFree of the 16 GB buffer every later
                # copy into page-locked host memory runs at half rate for the rest of the process -- measured,
                # tools/release_probe.py; the blocks stay in the context's pool, HBM is not short)
            except Exception as exc:   # noqa: BLE001
                c3["simulate_call_s"] = {"error": repr(exc)}
            extra["configs3"] = c3
        except Exception as exc:   # noqa: BLE001
            extra["configs3"] = {"error": repr(exc)}
    if single:
        # BASELINE.json configs[4]: PGSE over 512 x 512 (T2, ADC), 3-D gather shifts + D.  262 144 short-lived
        # wavefronts of <= 7 orders: a LATENCY figure, not a roofline one
        try:
            from epgpy_amd import functions

            seq5, _, _, opts5 = wl.build(epg, "pgse_512")
            sig5 = epg.simulate(seq5, **opts5)
            laps5 = []
            for _ in range(7):
                t0 = time.perf_counter()
                sig5 = epg.simulate(seq5, **opts5)
                laps5.append(time.perf_counter() - t0)
            sim5 = sorted(laps5)[len(laps5) // 2]
            ctx5 = _lib.get_context(local_rank)
            ms5 = {}
            for tag5 in ("packed", "one_wavefront_per_voxel"):
                enc5, _, _ = functions.compile_sequence(seq5, None, options=opts5)
                K5 = (enc5.packable_nd() if tag5 == "packed" else 0) or enc5.capacity()    # 16: four voxels per wavefront (what simulate() takes)
                plan5 = enc5.device_plan(ctx5, K5)
                buf5 = _lib.DeviceBuffer(ctx5, 16 * enc5.n_adc * enc5.nvox)
                run5 = lambda: _lib.run(ctx5, plan5, 0, plan5.n_ops, 0, enc5.nvox, None, None, K5, buf5.ptr.value, enc5.nvox, 0)  # noqa: E731
                wake(ctx5, run5); ctx5.timer_start()
                for _ in range(20):
                    run5()
                ms5[tag5] = ctx5.timer_stop() / 20
                buf5.free()
            extra["configs5"] = {"workload": "pgse_512 (BASELINE.json configs[4]): PGSE over 512x512 (T2, ADC), 3-D shift + D, 13 operators",
                                 "label": "latency (13 operators on <= 7 orders per voxel: a wavefront lives for microseconds)",
                                 "kernel_ms": round(ms5["packed"], 4), "voxels_per_s": enc5.nvox / (ms5["packed"] * 1e-3),
                                 "kernel": "rows_kernel<., 1, .>: 16 lanes per voxel, gather shifts by ds_bpermute (K = 16)",
                                 "kernel_ms_one_wavefront_per_voxel": round(ms5["one_wavefront_per_voxel"], 4),
                                 "simulate_call_ms": round(1e3 * sim5, 3), "simulate_call_ms_max_of_7": round(1e3 * max(laps5), 3), "signal_abs_range": [float(np.abs(sig5).min()), float(np.abs(sig5).max())]}
        except Exception as exc:   # noqa: BLE001
            extra["configs5"] = {"error": repr(exc)}
    if single and kind == "mse" and not args.no_extra_legs:
        # long state matrices: echo trains FROM EQUILIBRIUM whose state matrix is never bounded (2 n + 1 orders after n echoes -- the
        # reference's own growth, functions.py:135 / shift.py:86,98), 512 x 512 voxels, one state-resident launch each.  The kernels
        # walk the records in phases while the state matrix is short (run_contig_grow_kernel; 2048 orders: two legs)
        try:
            from epgpy_amd import functions
            from oracle import epg_c

            ctxl = _lib.get_context(local_rank)
            T1l, T2l = np.linspace(200, 3000, 512)[:, None], np.linspace(20, 300, 512)[None, :]
            long_leg = {"workload": "N-echo MSE trains from equilibrium, no max_nstate, 512x512 (T1, T2) voxels, state-resident",
                        "unit": "echo*voxels/s", "trains": []}
            for necho_l in (100, 250, 500, 1000):
                seql = wl.mse_sequence(epg, T1l, T2l, necho=necho_l)
                encl, _, _ = functions.compile_sequence(seql, None, options={})
                Kl = encl.capacity(resident=True)
                planl = encl.device_plan(ctxl, Kl)
                bufl = _lib.DeviceBuffer(ctxl, 16 * encl.n_adc * encl.nvox)
                runl = lambda: _lib.run(ctxl, planl, 0, planl.n_ops, 0, encl.nvox, None, None, Kl, bufl.ptr.value, encl.nvox, 0)  # noqa: E731
                runl(); ctxl.synchronize(); ctxl.timer_start()
                for _ in range(2):
                    runl()
                msl = ctxl.timer_stop() / 2
                entry = {"necho": necho_l, "orders": 2 * necho_l + 1, "K": Kl, "kernel": _lib.kernel_for(ctxl, planl, Kl), "ms_per_launch": round(msl, 3),
                         "value": necho_l * encl.nvox / (msl * 1e-3),
                         "populated_order_echo_voxels_per_s": sum(2 * e + 1 for e in range(1, necho_l + 1)) * encl.nvox / (msl * 1e-3)}
                if necho_l == 250:      # a few voxels of the buffer the timed launch wrote, against the C oracle
                    pick = np.linspace(0, encl.nvox - 1, 16).astype(np.int64)
                    got = np.stack([_column(bufl, encl, int(v)) for v in pick], axis=1)
                    i1, i2 = np.unravel_index(pick, (512, 512))
                    ref = epg_c.simulate([("T", 90, 90)] + [("S", 1), ("E", 5.0, T1l[i1, 0], T2l[0, i2], 0), ("T", 120, 0), ("S", 1),
                                                            ("E", 5.0, T1l[i1, 0], T2l[0, i2], 0), ("ADC",)] * necho_l)
                    entry["parity_max_abs_err_vs_oracle"] = float(np.abs(got - ref).max())
                long_leg["trains"].append(entry)
                bufl.free()
                del planl
            extra["long_state_matrices"] = long_leg
        except Exception as exc:   # noqa: BLE001
            extra["long_state_matrices"] = {"error": repr(exc)}
    if single and kind == "mse":
        # what a caller waits for: one whole epg.simulate() on host buffers (plan compilation, table upload, kernel,
        # D2H of the signal into a NumPy array) -- PCIe-inclusive, never `value`
        try:
            seq_e, _, necho, opts_e = wl.build(epg, args.workload)
            res = epg.simulate(seq_e, **opts_e)          # first calls: library warm-up, page cache, result blocks pinned
            res = epg.simulate(seq_e, **opts_e)
            laps = []
            for _ in range(5):
                t0 = time.perf_counter()
                res = epg.simulate(seq_e, **opts_e)      # a loop that rebinds its result (the previous array is released
                laps.append(time.perf_counter() - t0)    # AFTER the call returns: two result blocks alternate)
            ms_e = 1e3 * sorted(laps)[2]
            held = [res]
            t0 = time.perf_counter()
            held += [epg.simulate(seq_e, **opts_e) for _ in range(3)]     # a caller that keeps every result
            ms_keep = 1e3 * (time.perf_counter() - t0) / 3
            extra["e2e"] = {"what": f"one epg.simulate() call of {args.workload}, operators prebuilt, result = NumPy array on the host "
                                    "(third and later calls of a loop that rebinds the result; median of 5)",
                            "simulate_ms": round(ms_e, 3), "value": necho * leg.sp.nvox / (ms_e * 1e-3), "unit": "echo*voxels/s",
                            "simulate_ms_results_kept": round(ms_keep, 3), "result_MB": round(res.nbytes / 1e6, 1),
                            "pcie_floor_ms": round(res.nbytes / 54e9 * 1e3, 2)}
            del held, res
            # the same call with complex64 records (dtype=np.complex64: float64 arithmetic, one rounding per record on the device,
            # half the bytes over PCIe)
            for _ in range(2):
                res = epg.simulate(seq_e, dtype=np.complex64, **opts_e)
            laps = []
            for _ in range(5):
                t0 = time.perf_counter()
                res = epg.simulate(seq_e, dtype=np.complex64, **opts_e)
                laps.append(time.perf_counter() - t0)
            ms_c = 1e3 * sorted(laps)[2]
            extra["e2e"].update({"simulate_ms_c64": round(ms_c, 3), "value_c64": necho * leg.sp.nvox / (ms_c * 1e-3),
                                 "result_MB_c64": round(res.nbytes / 1e6, 1), "pcie_floor_ms_c64": round(res.nbytes / 54e9 * 1e3, 2)})
            del res
        except Exception as exc:   # noqa: BLE001
            extra.setdefault("e2e", {})["error"] = repr(exc)
    if single and kind == "mse":
        # first-order derivatives: the same train with d/dT2 -- and d/dT1, d/dB1 -- propagated next to
        # the state (diff.py:264-288); kernel time of one state-resident launch per number of variables, checked against
        # the NumPy oracle on a few voxels
        try:
            from epgpy_amd import functions
            from oracle import epg_numpy as onp

            n1, n2 = grid
            T1j, T2j = np.linspace(200, 3000, n1)[:, None], np.linspace(20, 300, n2)[None, :]
            excj = epg.T(90, 90, order1={"B1": {"alpha": 90}})
            rfcj = epg.T(120, 0, order1={"B1": {"alpha": 120}})
            rlxj = epg.E(5.0, T1j, T2j, order1=["T1", "T2"])
            seqj = [excj] + [epg.S(1), rlxj, rfcj, epg.S(1), rlxj, epg.ADC] * leg.n_adc
            ctxj = _lib.get_context(local_rank)
            jac = {"workload": f"the {leg.n_adc}-echo train of {args.workload} with derivative states (order1: T2, T1, B1), K = 64, state-resident",
                   "unit": "echo*voxels/s", "ms_per_launch": {}, "value": {}}
            for names in (["T2"], ["T2", "T1"], ["T2", "T1", "B1"]):
                encj, _, _ = functions.compile_sequence(seqj, None, options={"max_nstate": 63}, variables=names, fuse=not args.no_fuse)
                planj = encj.device_plan(ctxj, 64)
                bufj = _lib.DeviceBuffer(ctxj, 16 * encj.n_adc * encj.nvox)
                runj = lambda: _lib.run(ctxj, planj, 0, planj.n_ops, 0, encj.nvox, None, None, 64, bufj.ptr.value, encj.nvox, 0)  # noqa: E731
                wake(ctxj, runj); ctxj.timer_start()
                for _ in range(5):
                    runj()
                msj = ctxj.timer_stop() / 5
                key = f"{len(names)}_variable" + ("s" if len(names) > 1 else "")
                jac["ms_per_launch"][key] = round(msj, 3)
                jac["value"][key] = leg.n_adc * encj.nvox / (msj * 1e-3)
                if len(names) == 1:       # rows [echo][1 + V][voxel]: a few voxels against the oracle's recurrence
                    rows = np.empty((encj.n_adc, 8), dtype=np.complex128)
                    pick = np.linspace(0, encj.nvox - 1, 8).astype(np.int64)
                    for j, v in enumerate(pick):
                        rows[:, j] = _column(bufj, encj, int(v))
                    i1, i2 = np.unravel_index(pick, (n1, n2))
                    o1 = {"order1": {"T2": {"T2": 1}}}
                    tup = [("T", 90, 90)] + [("S", 1), ("E", 5.0, T1j[i1, 0], T2j[0, i2], 0, o1), ("T", 120, 0), ("S", 1),
                                             ("E", 5.0, T1j[i1, 0], T2j[0, i2], 0, o1), ("ADC",)] * leg.n_adc
                    ref = onp.simulate_jacobian(tup, ["magnitude", "T2"])          # [echo, voxel, 2]
                    got = rows.reshape(leg.n_adc, 2, 8)
                    jac["parity_max_abs_err_vs_oracle"] = float(max(np.abs(got[:, 0] - ref[..., 0]).max(), np.abs(got[:, 1] - ref[..., 1]).max()))
                    jac["fused_partials"] = bool(len(encj.fuse_partials))
                bufj.free()
            # and what a caller waits for: epg.simulate(probe=Jacobian) with the [echo, T1, T2, 1 + V] result in NumPy
            pj = epg.Jacobian(["magnitude", "T2"])
            for _ in range(3):
                rj = epg.simulate(seqj, probe=pj, max_nstate=63)
            t0 = time.perf_counter()
            rj = epg.simulate(seqj, probe=pj, max_nstate=63)
            jac["simulate_call_ms_1_variable"] = round(1e3 * (time.perf_counter() - t0), 2)
            jac["result_MB"] = round(rj.nbytes / 1e6)
            jac["pcie_floor_ms"] = round(rj.nbytes / 54e9 * 1e3, 1)
            del rj
            extra["jacobian"] = jac
        except Exception as exc:   # noqa: BLE001
            extra["jacobian"] = {"error": repr(exc)}
        # ---- the same question for the MRF train of config 3 (a dictionary WITH gradients): the
        # repetitions cannot be fused on the host (rotation over B1 between relaxations over (T1, T2)); the library folds them
        # at run time and carries the relaxation partials in logarithmic form (drun_kernel, DRUN_FOLD)
        try:
            from epgpy_amd import functions
            from oracle import epg_numpy as onp

            ntr_j = 250
            nm = wl.GRIDS["mrf_100"][1]
            T1m = np.linspace(300, 3000, nm[0])[:, None, None]
            T2m = np.linspace(20, 300, nm[1])[None, :, None]
            B1m = np.linspace(0.7, 1.3, nm[2])[None, None, :]
            alpha_j, TR_j = wl.mrf_trains(ntr_j)

            def mrf_train(make_T, make_E, shift, adc, t1, t2, b1):
                seq = [make_T(180 * b1, 90, 180.0), make_E(20.0, t1, t2)]
                e_te = make_E(3.0, t1, t2)
                for a_, tr_ in zip(alpha_j, TR_j):
                    seq += [make_T(a_ * b1, 90, float(a_)), e_te, adc, make_E(tr_ - 3.0, t1, t2), shift]
                return seq

            seqm = mrf_train(lambda a_, ph, c: epg.T(a_, ph, order1={"B1": {"alpha": c}}),
                             lambda tau, t1, t2: epg.E(tau, t1, t2, order1=["T1", "T2"]), epg.S(1), epg.ADC, T1m, T2m, B1m)
            ctxm = _lib.get_context(local_rank)
            jm = {"workload": f"the first {ntr_j} repetitions of mrf_100's train over its 100 x 100 x 100 (T1, T2, B1) grid with derivative "
                              "states (order1: T2, T1, B1), K = 64, state-resident",
                  "unit": "TR*voxels/s", "ms_per_launch": {}, "value": {}}
            for names in (["T2"], ["T2", "T1"], ["T2", "T1", "B1"]):
                encm, _, _ = functions.compile_sequence(seqm, None, options={"max_nstate": 63}, variables=names)
                planm = encm.device_plan(ctxm, 64)
                bufm = _lib.DeviceBuffer(ctxm, 16 * encm.n_adc * encm.nvox)
                runm = lambda: _lib.run(ctxm, planm, 0, planm.n_ops, 0, encm.nvox, None, None, 64, bufm.ptr.value, encm.nvox, 0)  # noqa: E731
                runm(); ctxm.synchronize(); ctxm.timer_start()
                for _ in range(3):
                    runm()
                msm = ctxm.timer_stop() / 3
                key = f"{len(names)}_variable" + ("s" if len(names) > 1 else "")
                jm["ms_per_launch"][key] = round(msm, 3)
                jm["value"][key] = ntr_j * encm.nvox / (msm * 1e-3)
                if len(names) == 3:       # rows [TR][1 + V][voxel]: a few voxels against the oracle's recurrence
                    pick = np.linspace(0, encm.nvox - 1, 6).astype(np.int64)
                    rows = np.empty((encm.n_adc, 6), dtype=np.complex128)
                    for j, v in enumerate(pick):
                        rows[:, j] = _column(bufm, encm, int(v))
                    i1, i2, i3 = np.unravel_index(pick, nm)
                    o1 = {"order1": {"T1": {"T1": 1}, "T2": {"T2": 1}}}
                    tup = mrf_train(lambda a_, ph, c: ("T", a_, ph, {"order1": {"B1": {"alpha": c}}}),
                                    lambda tau, t1, t2: ("E", tau, t1, t2, 0, o1), ("S", 1), ("ADC",),
                                    T1m[i1, 0, 0], T2m[0, i2, 0], B1m[0, 0, i3])
                    ref = onp.simulate_jacobian(tup, ["magnitude"] + names, max_nstate=63)          # [TR, voxel, 4]
                    got = np.moveaxis(rows.reshape(ntr_j, 4, 6), 1, 2)
                    jm["parity_max_abs_err_vs_oracle"] = float(np.abs(got - ref).max())
                    jm["parity_max_abs_reference"] = float(np.abs(ref[..., 1:]).max())
                bufm.free()
            extra["jacobian_mrf"] = jm
        except Exception as exc:   # noqa: BLE001
            extra["jacobian_mrf"] = {"error": repr(exc)}
    # ------------------------------------------------------------------ the JSON line
    emitted = threading.Lock()

    def emit_line():
        """rank 0 prints the ONE line (once: the watchdog below may get here before the main thread does)"""
        if not emitted.acquire(blocking=False):
            return
        if rank == 0:
            sys.stdout.flush()
            os.write(json_fd, (json.dumps(build_line()) + "\n").encode())
            if guard_fd is not None:
                os.write(guard_fd, b"F\n")

    def build_line():
        """the line as rank 0 prints it (cpu_baseline: N = 1 only)"""
        main_r = results[args.mode]
        other = "stream" if args.mode == "resident" else "resident"
        n1 = grid[0]
        scale_note = "*N" if args.scaling == "weak" else ""
        if kind == "mse":
            wtxt = (f"{args.workload}: 20-echo MSE (FA=120, ESP=10 ms), T1=linspace(200,3000,{n1}{scale_note}) x "
                    f"T2=linspace(20,300,{grid[1]}), max_nstate=63 (K=64)")
        else:
            wtxt = (f"{args.workload}: MRF {wl.MRF_NTR}-TR variable-FA SSFP (SURVEY.md 8d C3), T1=linspace(300,3000,{n1}{scale_note}) x "
                    f"T2=linspace(20,300,{grid[1]}) x B1=linspace(0.7,1.3,{grid[2]}), max_nstate=63 (K=64)")
        out = {
            "metric": ("echo-points x voxels / sec (MSE, 20 echoes, 64 k-states)" if kind == "mse" else
                       f"echo-points x voxels / sec (MRF, {wl.MRF_NTR} TR, 64 k-states)"),
            "value": main_r["value"], "unit": "echo*voxels/s",
            "n_gpus": world, "steps": main_r["steps"], "warmup": args.warmup,
            "ms_per_step": 1e3 * main_r["wall"] / main_r["steps"],
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic", "launcher": launched_by(), "rccl_ranks": rccl_ranks,
            "config": {"workload": wtxt, "mode": args.mode, "voxels_per_gpu": leg.nvox, "echoes": leg.n_adc, "k_states": K_STATES,
                       "launches_per_step": n_launch[args.mode], "parallelism": f"voxel-slabs x{world}",
                       "collective": "none in the timed region (voxel slabs are independent; every GPU's signal slab stays in its HBM)"},
            "roofline": roof(args.mode),
        }
        if world > 1 and comm is None:
            out["rccl_error"] = comm_error
        if other in results:
            out[other] = {"value": results[other]["value"], "ms_per_step": 1e3 * results[other]["wall"] / results[other]["steps"],
                          "steps": results[other]["steps"], "launches_per_step": n_launch[other], "roofline": roof(other)}
        out["parity_max_abs_err_vs_oracle"] = parity
        out["parity_voxels"] = parity_voxels
        if parity_error:
            out["parity_error"] = parity_error
        out.update(extra)
        if not args.no_cpu_baseline and world == 1:
            try:
                from oracle import epg_c

                native = epg_c.use_native_build()     # -O3 -march=native build for THIS host (portable build if it fails)
                threads = usable_cpus()
                info = host_info()
                if kind == "mse":
                    side1 = max(64, args.cpu_side // 4)
                    v1, t1, p1 = cpu_baseline_mse(side1, 1, args.cpu_seconds / 3, leg.n_adc)
                    vn, tn, pn = cpu_baseline_mse(args.cpu_side, threads, args.cpu_seconds, leg.n_adc)
                    sample = (f"the same 20-echo MSE on a {args.cpu_side}x{args.cpu_side} (T1, T2) grid, {pn} passes = "
                              f"{pn * leg.n_adc * args.cpu_side ** 2} echo*voxels in {tn:.1f} s, C oracle + OpenMP ({threads} threads); "
                              f"1-thread leg: {side1}x{side1}, {p1} passes in {t1:.1f} s")
                else:
                    vn, tn = cpu_baseline_mrf(16, threads, wl.MRF_NTR)
                    v1 = None
                    sample = (f"the same {wl.MRF_NTR}-TR MRF train on a 16x16x16 (T1, T2, B1) sub-grid of the same ranges, 1 pass = "
                              f"{wl.MRF_NTR * 4096} TR*voxels in {tn:.1f} s, C oracle + OpenMP ({threads} threads)")
                out["cpu_baseline"] = {"value": vn, "unit": "echo*voxels/s", "cores": threads, "kind": "port", "sample": sample,
                                       "value_1core": v1, "nproc": info["nproc"], "cpus_usable": info["cpus_usable"], "cpu_model": info["cpu_model"],
                                       "oracle_build": "gcc -O3 -march=native" if native else "gcc -O2 (portable)",
                                       "reference_as_shipped": {"value": REFERENCE_AS_SHIPPED, "cores": 1,
                                                                "where": "BASELINE.md section 2: the reference's NumPy path, 256x256 MSE with max_nstate=63, "
                                                                         "survey container (Xeon 2.1 GHz); the reference cannot travel to the GPU box"}}
            except Exception as exc:   # noqa: BLE001
                out["cpu_baseline"] = {"error": repr(exc)}
        return out

    if world > 1 and args.scaling == "weak":
        # BASELINE.json configs[3] next to the weak-scaling headline: mrf_100 cut into N slabs + ONE gather.  The
        # headline is already measured: if the communicator or this leg stalls on some rank, the watchdog prints
        # the line without it and every rank leaves, instead of the whole run being lost to the driver's time limit.
        def bail():
            if rank == 0:
                for name4 in (() if args.no_extra_legs else ("mse_1024", "mrf_100")):
                    extra.setdefault(f"strong_{name4}", {"error": f"did not finish within {args.extra_timeout:.0f} s (communicator or gather stalled); skipped"})
                emit_line()
            else:
                time.sleep(5.0)
            os._exit(0)

        if guard_fd is not None:        # the line as it stands now, for the case that this process does not survive the legs
            held = build_line()
            for name4 in (() if args.no_extra_legs else ("mse_1024", "mrf_100")):
                held[f"strong_{name4}"] = {"error": "the process ended inside the strong-scaling legs; the headline above was measured before them"}
            os.write(guard_fd, b"P " + json.dumps(held).encode() + b"\n")
        watchdog = threading.Timer(args.extra_timeout, bail)
        watchdog.daemon = True
        watchdog.start()
        try:
            make_comm()
            for name4, steps4 in (() if args.no_extra_legs else (("mse_1024", 8), ("mrf_100", 3))):
                l4, s4 = strong_leg(name4, steps4, 1, not args.no_fuse)
                g4 = s4.pop("_gather_obj", None)
                if rank == 0:
                    extra[f"strong_{name4}"] = s4
                if g4 is not None:
                    g4.free()
                if l4 is not None:
                    l4.free()
                del l4, g4
        except Exception as exc:   # noqa: BLE001
            if rank == 0:
                extra.setdefault("strong_mrf_100", {"error": repr(exc)})
        barrier()
        watchdog.cancel()
    emit_line()
    if dist is not None:
        dist.barrier()
        if comm is not None:
            comm.destroy()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
