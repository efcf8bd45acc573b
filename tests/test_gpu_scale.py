"""GPU: the full-size BASELINE configurations that do not fit the per-operator tests, and the multi-GPU plumbing
that can be exercised on ONE device (the RCCL gather with a one-rank communicator, slab assembly during the
download, two host threads on one context).

  * config 2-L with first-order derivatives: the 1024 x 1024 Jacobian, fused (one variable) and three-stage records.
  * config 3: 1000-TR MRF over 100 x 100 x 100 (T1, T2, B1) voxels, max_nstate = 63 -- the shape
    of the reference's examples/differentiation/optim_mrf.py:78-82 at dictionary size.  The 16 GB signal stays in
    HBM; the test fetches columns, not the array.
"""
import ctypes
import os
import socket
import threading

import numpy as np
import pytest

from epgpy_amd import epg, _lib, workloads as wl
from epgpy_amd.distributed import ShardedPlan, simulate_sharded
from oracle import epg_c, workloads as ow

pytestmark = pytest.mark.gpu
TOL = 1e-12


def _columns(buf, n_adc, ld, voxels):
    """signal[:, voxels] of a device buffer [n_adc][ld] (one strided 2-D copy per voxel)"""
    out = np.zeros((n_adc, len(voxels)), dtype=np.complex128)
    one = np.zeros((n_adc, 1), dtype=np.complex128)
    for c, vx in enumerate(voxels):
        buf.download_2d(one, 0, 1, n_adc, ld, offset=int(vx))
        out[:, c] = one[:, 0]
    return out


@pytest.mark.timeout(900)
def test_full_size_mrf_100x100x100_x_1000TR():
    seq, (T1, T2, B1), n_adc, opts = wl.build(epg, "mrf_100")
    grid = wl.GRIDS["mrf_100"][1]
    sp = ShardedPlan(seq, rank=0, world_size=1, **opts).bind()
    assert sp.nvox == 10 ** 6 and sp.n_adc == n_adc == 1000 and sp.K == sp.K_resident == 64
    buf = _lib.DeviceBuffer(sp._ctx, 16 * sp.n_adc * sp.slab)        # 16 GB, stays in HBM
    sp.run(buf.ptr.value)                                             # state-resident: rows_kernel, runs of record pairs
    sp._ctx.synchronize()
    # (a) 64 random voxels x all 1000 repetitions against the C oracle
    rng = np.random.default_rng(7)
    coords = [rng.integers(0, g, 64) for g in grid]
    flat = np.ravel_multi_index(coords, grid)
    alpha, TR = ow.mrf_trains()
    ref = epg_c.simulate(ow.mrf_tuples(T1[coords[0], 0, 0], T2[0, coords[1], 0], B1[0, 0, coords[2]], alpha, TR),
                         max_nstate=63, nthreads=4)
    got = _columns(buf, sp.n_adc, sp.slab, flat)
    assert float(np.max(np.abs(got - ref))) < TOL
    # first repetition, closed form: both pulses rotate about y, so the magnetisation stays in the x-z plane --
    # (Mx, Mz) = (sin t1, cos t1) after T(180 B1), relaxes for 20 ms, is rotated by t2 = alpha_0 B1, decays for TE
    t1, t2 = np.pi * B1[0, 0, coords[2]], np.deg2rad(alpha[0] * B1[0, 0, coords[2]])
    T1v, T2v = T1[coords[0], 0, 0], T2[0, coords[1], 0]
    mx, mz = np.sin(t1) * np.exp(-20.0 / T2v), 1 - (1 - np.cos(t1)) * np.exp(-20.0 / T1v)
    f0 = (mx * np.cos(t2) + mz * np.sin(t2)) * np.exp(-3.0 / T2v)
    assert np.allclose(np.abs(got[0]), np.abs(f0), rtol=0, atol=1e-12)
    # (b) per-timestep launches over a 4096-voxel slab give the same BITS as the resident run of the whole grid
    off, cnt = 481 * 1024 + 3, 4096
    slab_res = np.zeros((sp.n_adc, cnt), dtype=np.complex128)
    buf.download_2d(slab_res, 0, cnt, sp.n_adc, sp.slab, offset=off)
    small = _lib.DeviceBuffer(sp._ctx, 16 * sp.n_adc * cnt)
    sp.run(small.ptr.value, mode="stream", state=sp.new_state(cnt), part=(off, cnt), signal_ld=cnt)
    slab_str = small.download(np.complex128, (sp.n_adc, cnt))
    assert np.array_equal(slab_str, slab_res)
    # ... and so does a resident run of just that slab (voxel ranges do not change arithmetic)
    sp.run(small.ptr.value, part=(off, cnt), signal_ld=cnt)
    assert np.array_equal(small.download(np.complex128, (sp.n_adc, cnt)), slab_res)
    # (c) the signal is linear in the proton density
    sp2 = ShardedPlan([epg.PD(2.5)] + seq, rank=0, world_size=1, **opts).bind()
    sp2.run(small.ptr.value, part=(off, cnt), signal_ld=cnt)
    scaled = small.download(np.complex128, (sp.n_adc, cnt))
    assert np.allclose(scaled, 2.5 * slab_res, rtol=1e-13, atol=1e-15)
    small.free()
    buf.free()
    sp._ctx.release_cache()


@pytest.mark.timeout(600)
def test_full_size_jacobian_1024x1024_fused_records():
    """config 2-L with derivative states (diff.py:264-288): the 20-echo train over 1024 x 1024 (T1, T2)
    through `epg.simulate(probe=Jacobian)`.  One variable per plan: fused E . T . E records with library-generated partials
    (rows_deriv_kernel); three variables: the three-stage records of deriv_kernel.  Random voxels against the oracle's
    recurrence, the two forms against each other, the state column against the plain simulation bit for bit."""
    from oracle import epg_numpy as onp
    n = 1024
    T1, T2 = np.linspace(200, 3000, n)[:, None], np.linspace(20, 300, n)[None, :]

    def train(epg_, t1, t2, differentiated=True):
        o = (lambda **kw: kw) if differentiated else (lambda **kw: {})
        exc = epg_.T(90, 90, **o(order1={"B1": {"alpha": 90}}))
        rfc = epg_.T(120, 0, **o(order1={"B1": {"alpha": 120}}))
        rlx = epg_.E(5.0, t1, t2, **o(order1=["T1", "T2"]))
        return [exc] + [epg_.S(1), rlx, rfc, epg_.S(1), rlx, epg_.ADC] * 20

    seq = train(epg, T1, T2)
    one = epg.simulate(seq, probe=epg.Jacobian(["magnitude", "T2"]), max_nstate=63)              # fused records
    three = epg.simulate(seq, probe=epg.Jacobian(["magnitude", "T1", "T2", "B1"]), max_nstate=63)   # three stages
    assert one.shape == (20, n, n, 2) and three.shape == (20, n, n, 4)
    plain = epg.simulate(train(epg, T1, T2, differentiated=False), max_nstate=63)
    assert np.array_equal(one[..., 0], plain)
    assert np.abs(one[..., 1] - three[..., 2]).max() < 1e-11 and np.abs(three[..., 0] - plain).max() < TOL
    rng = np.random.default_rng(7)
    i1, i2 = rng.integers(0, n, 24), rng.integers(0, n, 24)
    o1 = {"order1": {"T1": {"T1": 1}, "T2": {"T2": 1}}}
    tup = ([("T", 90, 90, {"order1": {"B1": {"alpha": 90}}})]
           + [("S", 1), ("E", 5.0, T1[i1, 0], T2[0, i2], 0, o1), ("T", 120, 0, {"order1": {"B1": {"alpha": 120}}}), ("S", 1),
              ("E", 5.0, T1[i1, 0], T2[0, i2], 0, o1), ("ADC",)] * 20)
    ref = onp.simulate_jacobian(tup, ["magnitude", "T1", "T2", "B1"], max_nstate=63)     # [20, 24, 4]
    assert np.abs(three[:, i1, i2] - ref).max() < TOL
    assert np.abs(one[:, i1, i2, 1] - ref[..., 2]).max() < TOL
    assert np.abs(ref[..., 2]).max() > 1e-4                                              # (a derivative worth the name)


@pytest.mark.timeout(600)
def test_grids_beyond_the_fused_table_budget():
    """a 2048 x 2048 (T1, T2) grid: the four E . T . E tables of the train would take 1.7 GB of the library's coefficient pool
    (32-bit byte offsets; a 4096 x 4096 grid would not fit at all), so the planner leaves the sequence unfused and the
    library folds the relaxations into the rotations at run time -- same signal, no tables"""
    from epgpy_amd import functions
    n = 2048
    T1, T2 = np.linspace(200, 3000, n)[:, None], np.linspace(20, 300, n)[None, :]
    seq = wl.mse_sequence(epg, T1, T2)
    enc, _, _ = functions.compile_sequence(seq, None, options={"max_nstate": 63})
    assert not enc.fuses and enc.generated_size * 8 < 2e8                 # only the assembled relaxation table
    small, _, _ = functions.compile_sequence(wl.mse_sequence(epg, T1[:1024], T2[:, :1024]), None, options={"max_nstate": 63})
    assert len(small.fuses) == 4                                          # (below the budget the tables are generated)
    sig = epg.simulate(seq, max_nstate=63, out="device")
    assert sig.shape == (20, n, n)
    rng = np.random.default_rng(11)
    i1, i2 = rng.integers(0, n, 32), rng.integers(0, n, 32)
    got = _columns(sig._buf, 20, n * n, i1 * n + i2)
    ref = epg_c.simulate(ow.mse_tuples(T1[i1, 0], T2[0, i2]), max_nstate=63)
    assert np.abs(got - ref).max() < TOL


def test_strided_download_assembles_slabs():
    T1 = np.linspace(300, 2500, 9)[:, None]
    T2 = np.linspace(30, 150, 5)[None, :]
    seq = wl.mse_sequence(epg, T1, T2, necho=6)
    full = epg.simulate(seq, max_nstate=63).reshape(6, -1)
    out = np.zeros((6, 45), dtype=np.complex128)
    for r in range(4):    # 45 voxels over 4 ranks: slabs of 12, the last one ragged (9)
        sp = ShardedPlan(seq, rank=r, world_size=4, max_nstate=63).bind()
        buf = _lib.DeviceBuffer(sp._ctx, 16 * sp.n_adc * sp.slab)
        sp.run(buf.ptr.value)
        buf.download_2d(out, sp.vox0, sp.count, sp.n_adc, sp.slab)
        with pytest.raises(ValueError):
            buf.download_2d(out, 40, sp.slab, sp.n_adc, sp.slab)
    assert np.array_equal(out, full)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(600)
def test_rccl_gather_with_one_rank():
    """the one-process-per-GPU path end to end on one device: communicator id over torch.distributed (gloo), RCCL
    communicator of one rank, slab produced inside the gathered buffer, stream-ordered gather, strided download"""
    import torch.distributed as dist

    T1 = np.linspace(300, 2500, 33)[:, None]
    T2 = np.linspace(30, 150, 7)[None, :]
    seq = wl.mse_sequence(epg, T1, T2, necho=8)
    ref = epg.simulate(seq, max_nstate=63)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        for mode in ("resident", "stream"):
            got = simulate_sharded(seq, max_nstate=63, mode=mode)
            assert got.shape == ref.shape and np.array_equal(got, ref)
        # the communicator object on its own: gather of a block that is NOT in place
        ctx = _lib.get_context()
        comm = _lib.Comm(ctx, 0, 1, lambda raw: raw)
        src, dst = _lib.DeviceBuffer(ctx, 4096), _lib.DeviceBuffer(ctx, 4096)
        data = np.arange(512, dtype=np.float64)
        src.upload(data)
        comm.gather(src.ptr.value, dst.ptr.value, 4096, 0)
        assert np.array_equal(dst.download(np.float64, (512,)), data)
        with pytest.raises(_lib.EpgxError):
            comm.gather(src.ptr.value, dst.ptr.value, 4095, 0)      # not a multiple of 8
        with pytest.raises(_lib.EpgxError):
            comm.gather(src.ptr.value, dst.ptr.value, 4096, 1)      # no such root
        comm.destroy()
    finally:
        dist.destroy_process_group()


def test_sharded_c_entry_through_rccl(monkeypatch):
    """epgx_simulate_sharded_f64 with its device-side RCCL gather forced on for a single device"""
    from epgpy_amd import functions

    T1 = np.linspace(300, 2500, 21)[:, None]
    T2 = np.linspace(30, 150, 3)[None, :]
    seq = wl.mse_sequence(epg, T1, T2, necho=5)
    ref = epg.simulate(seq, max_nstate=63).reshape(5, -1)
    enc, _, _ = functions.compile_sequence(seq, options={"max_nstate": 63})
    ops, grid, spaces, coef, _ = enc.arrays()
    fuses = enc.fuse_array()
    strides = np.zeros((max(len(spaces), 1), _lib.MAX_DIMS), dtype=np.int64)
    for s, st in enumerate(spaces):
        strides[s, : len(st)] = st
    desc = _lib.PlanDesc(len(ops), ops.ctypes.data, len(grid), grid.ctypes.data, len(spaces), strides.ctypes.data,
                         coef.size, coef.ctypes.data, enc.n_adc, 0, None, 0, len(fuses),
                         fuses.ctypes.data if len(fuses) else None, enc.generated_size)
    lib = _lib.load()
    for force in ("0", "1"):
        monkeypatch.setenv("EPGX_FORCE_RCCL", force)
        out = np.zeros((5, 63), dtype=np.complex128)
        rc = lib.epgx_simulate_sharded_f64(ctypes.byref(desc), 64, 1, None, out.ctypes.data)
        assert rc == 0, lib.epgx_last_error()
        assert np.array_equal(out, ref)
    bad = _lib.PlanDesc(len(ops), ops.ctypes.data, 0, grid.ctypes.data, len(spaces), strides.ctypes.data, coef.size,
                        coef.ctypes.data, enc.n_adc)
    assert lib.epgx_simulate_sharded_f64(ctypes.byref(bad), 64, 1, None, out.ctypes.data) == -1     # ndim checked first
    assert lib.epgx_simulate_sharded_f64(ctypes.byref(desc), 64, 99, None, out.ctypes.data) == -1   # more GPUs than visible


def test_two_host_threads_share_one_context():
    """ctypes releases the GIL: two threads running simulate() on the same context / stream must not corrupt the
    caching allocator or the plan caches (mutexes in epgx_ctx / epgx_plan)"""
    T1 = np.linspace(300, 2500, 64)[:, None]
    T2 = np.linspace(30, 150, 48)[None, :]
    seqs = [wl.mse_sequence(epg, T1, T2 * (1 + 0.1 * i), necho=10) for i in range(2)]
    refs = [epg.simulate(s, max_nstate=63) for s in seqs]
    errors = []

    def work(i):
        try:
            for _ in range(25):
                if not np.array_equal(epg.simulate(seqs[i], max_nstate=63), refs[i]):
                    errors.append(f"thread {i}: result differs")
                    return
        except Exception as exc:   # noqa: BLE001
            errors.append(repr(exc))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    # the same with results large enough for the slab pipeline (copy stream, events and page-locked blocks are shared)
    T1b, T2b = np.linspace(300, 2500, 600)[:, None], np.linspace(30, 150, 300)[None, :]
    big = [wl.mse_sequence(epg, T1b, T2b * (1 + 0.1 * i), necho=12) for i in range(2)]
    big_ref = [epg.simulate(s, max_nstate=63).copy() for s in big]

    def work_big(i):
        try:
            for _ in range(6):
                if not np.array_equal(epg.simulate(big[i], max_nstate=63), big_ref[i]):
                    errors.append(f"thread {i}: large result differs")
                    return
        except Exception as exc:   # noqa: BLE001
            errors.append(repr(exc))

    threads = [threading.Thread(target=work_big, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_assembled_tables_pinned_pipeline_and_device_output(monkeypatch):
    """what a whole `simulate()` call adds around the kernel: relaxation tables assembled on the device from per-axis
    columns (same bits as the uploaded table), the signal leaving in voxel slabs into a recycled page-locked block
    while the next slab computes (same bits as one launch + one copy), results that stay valid while later calls
    recycle blocks, and `out="device"`"""
    from epgpy_amd import plan as _plan, functions

    T1 = np.linspace(200, 3000, 700)[:, None]
    T2 = np.linspace(20, 300, 260)[None, :]
    seq = wl.mse_sequence(epg, T1, T2, necho=12)                     # 182 000 voxels, 35 MB of signal
    enc, _, _ = functions.compile_sequence(seq, options={"max_nstate": 63})
    enc.arrays()
    assert len(enc.assemble_array()) == 1
    first = epg.simulate(seq, max_nstate=63)                         # assembled tables + slab pipeline + pinned block
    assert first.shape == (12, 700, 260) and not first.flags.owndata
    ii, jj = np.arange(0, 700, 97), np.arange(0, 260, 37)
    ref = epg_c.simulate(ow.mse_tuples(T1[ii, 0], T2[0, jj], necho=12), max_nstate=63)      # voxel v = (ii[v], jj[v])
    assert float(np.max(np.abs(first[:, ii, jj] - ref))) < TOL
    keep = first.copy()
    # (a) uploaded tables instead of assembled ones: the same bits
    monkeypatch.setattr(_plan.Encoder, "ASSEMBLE_MIN_ENTRIES", 1 << 40)
    enc2, _, _ = functions.compile_sequence(seq, options={"max_nstate": 63})
    enc2.arrays()
    assert len(enc2.assemble_array()) == 0
    assert np.array_equal(epg.simulate(seq, max_nstate=63), keep)
    monkeypatch.undo()
    # (b) one launch + one copy (the pageable path) instead of the slab pipeline: the same bits
    monkeypatch.setattr(_lib, "PINNED_MAX_BYTES", 0)
    plain = epg.simulate(seq, max_nstate=63)
    assert plain.flags.owndata or plain.base is not None
    assert np.array_equal(plain, keep)
    monkeypatch.undo()
    # (c) several results alive at once never share memory (only the first two get page-locked blocks); dropped
    # blocks are recycled and recycling never touches a result that is still alive
    others = [epg.simulate(seq, max_nstate=63) for _ in range(3)]
    assert all(np.array_equal(o, keep) for o in others) and np.array_equal(first, keep)
    assert len({o.ctypes.data for o in others} | {first.ctypes.data}) == 4
    del others
    for _ in range(3):
        again = epg.simulate(seq, max_nstate=63)
        assert np.array_equal(again, keep) and np.array_equal(first, keep)
    # (d) the signal left on the device
    dev = epg.simulate(seq, max_nstate=63, out="device")
    assert isinstance(dev, functions.DeviceSignal) and dev.shape == (12, 700, 260) and dev.dtype == np.complex128
    assert np.array_equal(np.asarray(dev), keep)
    f0, z0 = epg.simulate(seq, max_nstate=63, out="device", probe=["F0", "Z0"])
    assert f0.shape == z0.shape == (12, 700, 260) and f0.ptr != z0.ptr and np.array_equal(f0.download(), keep)
    with pytest.raises(NotImplementedError):
        epg.simulate(seq[:-1] + [epg.Adc("F0", phase=30.0)], max_nstate=63, out="device")
    with pytest.raises(ValueError):
        epg.simulate(seq, max_nstate=63, out="elsewhere")


def test_c_entry_pipelines_large_signals_and_simulate_options(capsys):
    """epgx_simulate_f64 with a signal above 32 MB takes the slab pipeline (epgx_run_to_host) -- same bits as the Python
    path; `squeeze=True` (the reference's unimplemented hook) and `disp=True` are accepted and change nothing"""
    from epgpy_amd import functions

    T1 = np.linspace(200, 3000, 600)[:, None]
    T2 = np.linspace(20, 300, 300)[None, :]
    seq = wl.mse_sequence(epg, T1, T2, necho=12)                 # 180 000 voxels: 34.6 MB of signal
    ref = epg.simulate(seq, max_nstate=63)
    enc, _, _ = functions.compile_sequence(seq, options={"max_nstate": 63})
    ops, grid, spaces, coef, _ = enc.arrays()
    fuses, asm = enc.fuse_array(), enc.assemble_array()
    strides = np.zeros((max(len(spaces), 1), _lib.MAX_DIMS), dtype=np.int64)
    for s, st in enumerate(spaces):
        strides[s, : len(st)] = st
    desc = _lib.PlanDesc(len(ops), ops.ctypes.data, len(grid), grid.ctypes.data, len(spaces), strides.ctypes.data,
                         coef.size, coef.ctypes.data, enc.n_adc, 0, None, 0, len(fuses), fuses.ctypes.data, enc.generated_size,
                         len(asm), 0, asm.ctypes.data)
    ctx = _lib.get_context()
    out = np.zeros((12, 600 * 300), dtype=np.complex128)
    rc = ctx.lib.epgx_simulate_f64(ctx.handle, ctypes.byref(desc), 64, None, None, out.ctypes.data, None)
    assert rc == 0, ctx.lib.epgx_last_error()
    assert np.array_equal(out.reshape(ref.shape), ref)
    assert np.array_equal(epg.simulate(seq, max_nstate=63, squeeze=True), ref)
    for mode in ("resident", "stream"):
        got = epg.simulate(wl.mse_sequence(epg, T1[:40], T2[:, :30], necho=5), max_nstate=63, disp=True, mode=mode)
        assert got.shape == (5, 40, 30)
        assert "Simulating: [" in capsys.readouterr().out
