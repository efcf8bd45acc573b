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
from oracle import epg_c, epg_numpy as onp, workloads as ow
from tests import sequences as sq

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
    # (a) 16 384 voxels x all 1000 repetitions against the C oracle: the first and the last 4096 voxels of the grid (both
    # ends of the launch, every offset inside a wave group and a workgroup) and two ranges across the middle
    alpha, TR = ow.mrf_trains()
    threads = max(1, len(os.sched_getaffinity(0)))
    block = np.zeros((sp.n_adc, 4096), dtype=np.complex128)
    for first in (0, 333 * 1000 + 17, 707 * 1000 + 501, 10 ** 6 - 4096):
        flat = np.arange(first, first + 4096)
        coords = np.unravel_index(flat, grid)
        ref = epg_c.simulate(ow.mrf_tuples(T1[coords[0], 0, 0], T2[0, coords[1], 0], B1[0, 0, coords[2]], alpha, TR),
                             max_nstate=63, nthreads=threads)
        buf.download_2d(block, 0, 4096, sp.n_adc, sp.slab, offset=first)
        assert float(np.max(np.abs(block - ref))) < TOL, first
    rng = np.random.default_rng(7)
    coords = [rng.integers(0, g, 64) for g in grid]
    flat = np.ravel_multi_index(coords, grid)
    got = _columns(buf, sp.n_adc, sp.slab, flat)
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
    assert sq.same_bits(one[..., 0], plain, x64=True)      # (state column of the derivative kernel / the 64-order rows kernel)
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
    """a 3300 x 3300 (T1, T2) grid: the E . T . E table of the train and its intermediate would take 2.1 GB of the library's
    coefficient pool (the four-voxels-per-wavefront kernels reach 2 GiB of it; a 4096 x 4096 grid would not fit its 32-bit
    byte offsets at all), so the planner leaves the sequence unfused and the library folds the relaxations into the rotations
    at run time -- same signal, no tables"""
    from epgpy_amd import functions
    n = 3300
    T1, T2 = np.linspace(200, 3000, n)[:, None], np.linspace(20, 300, n)[None, :]
    seq = wl.mse_sequence(epg, T1, T2)
    enc, _, _ = functions.compile_sequence(seq, None, options={"max_nstate": 63})
    assert not enc.fuses and enc.generated_size * 8 < 4e8                 # only the assembled relaxation table
    small, _, _ = functions.compile_sequence(wl.mse_sequence(epg, T1[:1024], T2[:, :1024]), None, options={"max_nstate": 63})
    assert len(small.fuses) == 2                                          # (below the budget the tables are generated)
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
            for via in ("rccl", "pcie", "auto"):      # gathered on the device / written into the shared-memory result
                got = simulate_sharded(seq, max_nstate=63, mode=mode, via=via)
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
    """epgx_simulate_sharded_f64: per-device slab pipelines into the caller's array (default), and its device-side RCCL
    gather (EPGX_SHARDED_GATHER=rccl) on a single device, twice (the communicator set is created once and kept)"""
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
    desc = _lib.PlanDesc(ctypes.sizeof(_lib.PlanDesc), len(ops), ops.ctypes.data, len(grid), grid.ctypes.data, len(spaces), strides.ctypes.data,
                         coef.size, coef.ctypes.data, enc.n_adc, 0, None, 0, len(fuses),
                         fuses.ctypes.data if len(fuses) else None, enc.generated_size)
    lib = _lib.load()
    for how in ("direct", "rccl", "rccl"):
        monkeypatch.setenv("EPGX_SHARDED_GATHER", how)
        out = np.zeros((5, 63), dtype=np.complex128)
        rc = lib.epgx_simulate_sharded_f64(ctypes.byref(desc), 64, 1, None, out.ctypes.data, 0)
        assert rc == 0, lib.epgx_last_error()
        assert sq.same_bits(out, ref, x64=True)      # (the C entry runs the capacity it is given, 64; simulate() packs short trains)
    bad = _lib.PlanDesc(ctypes.sizeof(_lib.PlanDesc), len(ops), ops.ctypes.data, 0, grid.ctypes.data, len(spaces), strides.ctypes.data, coef.size,
                        coef.ctypes.data, enc.n_adc)
    assert lib.epgx_simulate_sharded_f64(ctypes.byref(bad), 64, 1, None, out.ctypes.data, 0) == -1     # ndim checked first
    assert lib.epgx_simulate_sharded_f64(ctypes.byref(desc), 64, 99, None, out.ctypes.data, 0) == -1   # more GPUs than visible
    short = _lib.PlanDesc(ctypes.sizeof(_lib.PlanDesc) - 8, len(ops), ops.ctypes.data, len(grid), grid.ctypes.data, len(spaces),
                          strides.ctypes.data, coef.size, coef.ctypes.data, enc.n_adc)
    assert lib.epgx_simulate_sharded_f64(ctypes.byref(short), 64, 1, None, out.ctypes.data, 0) == -1 and b"struct_size" in lib.epgx_last_error()


def test_plan_desc_struct_size_is_checked():
    """a caller built against another epgx_plan_desc (a shorter struct, an older ABI whose first member was n_ops) is
    refused with EPGX_ERR_INVALID before anything else of the struct is read"""
    enc, _, _ = epg.compile_sequence(wl.mse_sequence(epg, 1000.0, [50.0, 80.0], necho=3), options={"max_nstate": 63}, fuse=False)
    ops, grid, spaces, coef, _ = enc.arrays()
    strides = np.zeros((max(len(spaces), 1), _lib.MAX_DIMS), dtype=np.int64)
    for s_, st in enumerate(spaces):
        strides[s_, : len(st)] = st
    ctx = _lib.get_context()
    for size in (0, ctypes.sizeof(_lib.PlanDesc) - 8, ctypes.sizeof(_lib.PlanDesc) + 8, len(ops)):
        desc = _lib.PlanDesc(size, len(ops), ops.ctypes.data, len(grid), grid.ctypes.data, len(spaces), strides.ctypes.data,
                             coef.size, coef.ctypes.data, enc.n_adc)
        handle = ctypes.c_void_p()
        assert ctx.lib.epgx_plan_create(ctx.handle, ctypes.byref(desc), ctypes.byref(handle)) == -1
        assert b"struct_size" in ctx.lib.epgx_last_error() and not handle.value
    desc = _lib.PlanDesc(ctypes.sizeof(_lib.PlanDesc), len(ops), ops.ctypes.data, len(grid), grid.ctypes.data, len(spaces),
                         strides.ctypes.data, coef.size, coef.ctypes.data, enc.n_adc)
    handle = ctypes.c_void_p()
    assert ctx.lib.epgx_plan_create(ctx.handle, ctypes.byref(desc), ctypes.byref(handle)) == 0
    assert ctx.lib.epgx_plan_destroy(handle) == 0


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
    # (b) an ordinary (pageable) result array, filled through the library's staging ring and host copy threads: the same bits
    monkeypatch.setattr(_lib, "PINNED_MAX_BYTES", 0)
    live = {k: list(v) for k, v in _lib._PinnedBlock.live.items()}
    plain = epg.simulate(seq, max_nstate=63)
    assert {k: list(v) for k, v in _lib._PinnedBlock.live.items()} == live          # (no block of the page-locked pool behind this one)
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


@pytest.mark.parametrize("fuse", [True, False])
def test_assembled_partial_tables(monkeypatch, fuse):
    """d(relaxation)/d(T1 | T2) over a (T1, T2) grid varies along one axis per column pair: the partial tables are
    assembled on the device from those columns (Encoder.partial_table), same bits as the uploaded tables, for the fused
    S.E.T.S.E records and the record-by-record form alike"""
    from epgpy_amd import plan as _plan, functions

    T1 = np.linspace(200, 3000, 96)[:, None]
    T2 = np.linspace(20, 300, 80)[None, :]
    _, ops, variables = sq.jac_mse(T1, T2, 1.0, necho=8)
    seq, pj = ops(epg), epg.Jacobian(variables)
    options = {"max_nstate": 15, "fuse": fuse}
    enc, _, _ = functions.compile_sequence(seq, [pj], options=options, variables=variables[1:])
    arrays = enc.arrays(16)
    assert len(enc.assemble_array()) >= 3                     # the value table and the two relaxation partials
    assert arrays[3].nbytes < 96 * 80 * 4 * 8                 # ... none of which travels whole
    got = epg.simulate(seq, probe=pj, **options)
    assert got.shape == (8, 96, 80, 4)
    ii, jj = np.arange(0, 96, 19), np.arange(0, 80, 13)       # voxel v = (ii[v], jj[v]) against the NumPy oracle
    tuples, _, _ = sq.jac_mse(T1[ii[:5], 0], T2[0, jj[:5]], 1.0, necho=8)
    assert float(np.max(np.abs(got[:, ii[:5], jj[:5]] - onp.simulate_jacobian(tuples, variables, max_nstate=15)))) < TOL
    monkeypatch.setattr(_plan.Encoder, "ASSEMBLE_MIN_ENTRIES", 1 << 40)
    enc2, _, _ = functions.compile_sequence(seq, [pj], options=options, variables=variables[1:])
    enc2.arrays(16)
    assert len(enc2.assemble_array()) == 0
    assert np.array_equal(epg.simulate(seq, probe=pj, **options), got)


def test_jacobian_left_on_the_device():
    """`simulate(probe=Jacobian(...), out="device")`: the state and its derivative rows stay in HBM (a dictionary WITH gradients
    for device-side matching / CRLB code); the handle describes them, `np.asarray` gives the host path's array bit for bit"""
    from epgpy_amd import functions

    T1 = np.linspace(300, 2500, 24)[:, None]
    T2 = np.linspace(20, 300, 20)[None, :]
    _, ops, variables = sq.jac_mse(T1, T2, 1.0, necho=6)
    seq = ops(epg)
    host = epg.simulate(seq, probe=epg.Jacobian(variables), max_nstate=15)
    dev = epg.simulate(seq, probe=epg.Jacobian(variables), max_nstate=15, out="device")
    assert isinstance(dev, functions.DeviceJacobian) and dev.shape == host.shape == (6, 24, 20, 4)
    assert dev.variables == variables and dev.rows == [0, 1, 2, 3] and dev.record_stride == 4 * 480 and dev.row_stride == 480
    assert np.array_equal(np.asarray(dev), host)
    col = dev.column("T2")
    assert isinstance(col, functions.DeviceSignal) and col.shape == (6, 24, 20) and np.array_equal(col.download(), host[..., 2])
    # a column subset in another order, next to a plain probe
    pair = [epg.Jacobian(["T2", "magnitude"]), "F0"]
    jac, f0 = epg.simulate(seq, probe=pair, max_nstate=15, out="device")
    jac_h, f0_h = epg.simulate(seq, probe=pair, max_nstate=15)
    assert jac.rows == [1, 0] and np.array_equal(np.asarray(jac), jac_h) and np.abs(jac_h - host[..., [2, 0]]).max() < 1e-12
    assert isinstance(f0, functions.DeviceSignal) and np.array_equal(np.asarray(f0), f0_h)
    with pytest.raises(NotImplementedError):
        epg.simulate(seq, probe=epg.Jacobian(["T2", "nobody"]), max_nstate=15, out="device")
    with pytest.raises(NotImplementedError):
        epg.simulate(seq, probe=epg.Jacobian(variables), max_nstate=15, out="device", ngpu=2)


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
    desc = _lib.PlanDesc(ctypes.sizeof(_lib.PlanDesc), len(ops), ops.ctypes.data, len(grid), grid.ctypes.data, len(spaces), strides.ctypes.data,
                         coef.size, coef.ctypes.data, enc.n_adc, 0, None, 0, len(fuses), fuses.ctypes.data, enc.generated_size,
                         len(asm), 0, asm.ctypes.data)
    ctx = _lib.get_context()
    out = np.zeros((12, 600 * 300), dtype=np.complex128)
    rc = ctx.lib.epgx_simulate_f64(ctx.handle, ctypes.byref(desc), 64, None, None, out.ctypes.data, None, 0)
    assert rc == 0, ctx.lib.epgx_last_error()
    assert sq.same_bits(out.reshape(ref.shape), ref, x64=True)      # (capacity 64 as given / simulate() at 32 orders per voxel)
    assert np.array_equal(epg.simulate(seq, max_nstate=63, squeeze=True), ref)
    for mode in ("resident", "stream"):
        got = epg.simulate(wl.mse_sequence(epg, T1[:40], T2[:, :30], necho=5), max_nstate=63, disp=True, mode=mode)
        assert got.shape == (5, 40, 30)
        assert "Simulating: [" in capsys.readouterr().out


# ------------------------------------------------------------------ round 3: multi-GPU paths, staged downloads
def _probe_sequences():
    """(name, sequence, simulate keywords) covering every probe flavour the device path records"""
    T1 = np.linspace(300, 2500, 40)[:, None]
    T2 = np.linspace(30, 150, 25)[None, :]
    necho = 6
    phases = 58.5 * np.arange(necho) ** 2
    w_full = (np.arange(1000).reshape(40, 25) + 1.0) * (1 + 0.5j)
    exc, rfc, rlx, sh = epg.T(90, 90), epg.T(120, 0), epg.E(5, T1, T2, 0.01), epg.S(1)

    def seq_with(adcs):
        return [exc] + [op for n in range(necho) for op in (sh, rlx, rfc, sh, rlx, adcs[n])]

    red0, wsum = epg.Adc("F0", reduce=0), epg.Adc("F0", weights=w_full)
    return [("plain", seq_with([epg.ADC] * necho), {}),
            ("phase", seq_with([epg.Adc("F0", phase=p) for p in phases]), {}),
            ("F0_Z0", seq_with([epg.ADC] * necho), {"probe": ["F0", "Z0"]}),
            ("Z0_times", seq_with([epg.ADC] * necho), {"probe": "Z0", "adc_time": True}),
            ("reduce0", seq_with([red0] * necho), {}),
            ("reduce_last", seq_with([epg.Adc("F0", reduce=1)] * necho), {}),
            ("weights", seq_with([wsum] * necho), {}),
            ("mixed", seq_with([epg.ADC, red0] * (necho // 2)), {"asarray": False})]


def _same(a, b, exact=True):
    if isinstance(a, (tuple, list)):
        assert len(a) == len(b)
        for x, y in zip(a, b):
            _same(x, y, exact)
        return
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape
    if exact:
        assert np.array_equal(a, b)
    else:       # sums over grid axes: the order of summation follows the slabs
        assert np.allclose(a, b, rtol=1e-13, atol=1e-12)


def test_ngpu_one_is_the_plain_call(monkeypatch):
    """simulate(ngpu=1) == simulate() bit for bit (the same code path: functions._Fleet with one device); the slab / thread /
    per-device-download machinery of ngpu > 1 is driven on this one-GPU box by handing the same device out three times"""
    from epgpy_amd import functions

    for name, seq, kw in _probe_sequences():
        _same(epg.simulate(seq, max_nstate=63, ngpu=1, **kw), epg.simulate(seq, max_nstate=63, **kw))
    tuples, ops, variables = sq.jac_mse(np.linspace(500, 2000, 30)[:, None], np.linspace(40, 120, 20)[None, :], 1.0, necho=5)
    jac = epg.Jacobian(variables)
    ref_j = epg.simulate(ops(epg), probe=jac, max_nstate=63)
    _same(epg.simulate(ops(epg), probe=jac, max_nstate=63, ngpu=1), ref_j)
    # three "GPUs" = three voxel slabs on device 0, each driven by its own host thread into ONE result array
    monkeypatch.setattr(functions, "_device_list", lambda device, ngpu: [0] * (ngpu or 1))
    for name, seq, kw in _probe_sequences():
        exact = not any(word in name for word in ("reduce", "weights", "mixed"))
        _same(epg.simulate(seq, max_nstate=63, ngpu=3, **kw), epg.simulate(seq, max_nstate=63, **kw), exact)
    _same(epg.simulate(ops(epg), probe=jac, max_nstate=63, ngpu=3), ref_j)
    # a start state (init=): cut into per-device slab states (functions.py:149: the caller's init is copied, never mutated)
    name, seq, kw = _probe_sequences()[0]
    init = epg.StateMatrix([0, 0, 1], shape=(40, 25), max_nstate=63)
    for op in seq[:6]:                                   # the state matrix after the excitation and most of the first echo
        init = op(init)
    before = init.states.copy()
    _same(epg.simulate(seq[7:], init=init, ngpu=3, **kw), epg.simulate(seq[7:], init=init, **kw))
    assert np.array_equal(init.states, before)
    # ... also under a Jacobian probe (the derivative states start from zero on every device)
    jseq = ops(epg)
    jinit = epg.S(1)(epg.T(35, 20)(epg.StateMatrix([0, 0, 1], shape=(30, 20), max_nstate=63)))
    _same(epg.simulate(jseq, init=jinit, probe=jac, ngpu=3), epg.simulate(jseq, init=jinit, probe=jac))
    # a result large enough for sub-slabs inside every device's slab; and the signal left on the devices
    T1 = np.linspace(200, 3000, 600)[:, None]
    T2 = np.linspace(20, 300, 300)[None, :]
    big = wl.mse_sequence(epg, T1, T2, necho=12)
    ref = epg.simulate(big, max_nstate=63)
    _same(epg.simulate(big, max_nstate=63, ngpu=3), ref)
    dev = epg.simulate(big, max_nstate=63, ngpu=3, out="device")
    assert isinstance(dev, functions.ShardedDeviceSignal) and dev.shape == ref.shape and len(dev.parts) == 3
    assert [p.vox0 for p in dev.parts] == [0, 60000, 120000] and all(p.count == 60000 for p in dev.parts)
    _same(np.asarray(dev), ref)


def test_staged_download_into_pageable_memory():
    """epgx_download_2d / epgx_run_to_host into ordinary host memory: tiles of the staging ring (1 MiB here, so that a few
    MB exercise many tiles), rows narrower and wider than a tile, odd pitches -- against the page-locked route"""
    os.environ["EPGX_STAGE_MB"] = "1"
    try:
        ctx = _lib.Context(0)           # a fresh context: its staging ring is created with the small blocks
        rng = np.random.default_rng(11)
        for rows, width, dev_ld in ((700, 1000, 1003), (3, 200001, 200001), (1, 400000, 400000), (5000, 7, 16)):
            data = (rng.standard_normal((rows, dev_ld)) + 1j * rng.standard_normal((rows, dev_ld)))
            buf = _lib.DeviceBuffer(ctx, data.nbytes)
            buf.upload(data)
            out = np.full((rows, width + 5), -1.0 + 0j)
            _lib.check(ctx.lib.epgx_download_2d(ctx.handle, out.ctypes.data + 32, 16 * out.shape[1], buf.ptr, 16 * dev_ld, 16 * width, rows), "download")
            assert np.array_equal(out[:, 2: 2 + width], data[:, :width])
            assert (out[:, :2] == -1).all() and (out[:, 2 + width:] == -1).all()      # nothing written outside the block
            pinned = _lib.pinned_empty(ctx, (rows, width), np.complex128)
            _lib.check(ctx.lib.epgx_download_2d(ctx.handle, pinned.ctypes.data, 16 * width, buf.ptr, 16 * dev_ld, 16 * width, rows), "download")
            assert np.array_equal(pinned, data[:, :width])
            buf.free()
        # a whole pipelined run into a plain array == into a page-locked one
        T1 = np.linspace(200, 3000, 500)[:, None]
        T2 = np.linspace(20, 300, 280)[None, :]
        enc, _, _ = epg.compile_sequence(wl.mse_sequence(epg, T1, T2, necho=10), options={"max_nstate": 63})
        plan = enc.device_plan(ctx, 64)
        sig = _lib.DeviceBuffer(ctx, 16 * enc.n_adc * enc.nvox)
        plain = np.zeros((enc.n_adc, enc.nvox), dtype=np.complex128)
        locked = _lib.pinned_empty(ctx, (enc.n_adc, enc.nvox), np.complex128)
        _lib.run_to_host(ctx, plan, 64, sig.ptr.value, plain)
        _lib.run_to_host(ctx, plan, 64, sig.ptr.value, locked)
        assert np.array_equal(plain, locked) and np.abs(plain).max() > 0.1
        # a voxel range into its columns of a larger host array
        part = np.zeros((enc.n_adc, enc.nvox), dtype=np.complex128)
        small = _lib.DeviceBuffer(ctx, 16 * enc.n_adc * 70001)
        _lib.run_to_host(ctx, plan, 64, small.ptr.value, part, vox0=33333, nvox=70001)
        assert np.array_equal(part[:, 33333: 33333 + 70001], plain[:, 33333: 33333 + 70001])
        assert not part[:, :33333].any() and not part[:, 33333 + 70001:].any()
        with pytest.raises(_lib.EpgxError):
            _lib.run_to_host(ctx, plan, 64, small.ptr.value, part, vox0=enc.nvox - 10, nvox=11)
    finally:
        del os.environ["EPGX_STAGE_MB"]


@pytest.mark.timeout(600)
def test_sharded_probes_and_pipelined_gather_with_one_rank():
    """the one-process-per-GPU path with a one-rank RCCL communicator (gloo side channel): every probe flavour returns
    what epg.simulate returns; the communicator is created once and reused; epgx_comm_reduce; the gather in sub-slabs on the
    communicator's stream (compute k + 1 while k travels) lands the same bits as the serial one"""
    import torch.distributed as dist

    from epgpy_amd.distributed import SlabGather, group_key

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        for name, seq, kw in _probe_sequences():
            _same(simulate_sharded(seq, max_nstate=63, via="rccl", **kw), epg.simulate(seq, max_nstate=63, **kw))
            _same(simulate_sharded(seq, max_nstate=63, **kw), epg.simulate(seq, max_nstate=63, **kw))     # (one node: shared-memory result)
        assert len(_lib._COMMS) == 1                                  # one communicator for all of these calls
        tuples, ops, variables = sq.jac_mse(np.linspace(500, 2000, 30)[:, None], np.linspace(40, 120, 20)[None, :], 1.0, necho=5)
        jac = epg.Jacobian(variables)
        _same(simulate_sharded(ops(epg), probe=jac, max_nstate=63, via="rccl"), epg.simulate(ops(epg), probe=jac, max_nstate=63))
        _same(simulate_sharded(ops(epg), probe=jac, max_nstate=63), epg.simulate(ops(epg), probe=jac, max_nstate=63))
        name, seq, kw = _probe_sequences()[0]
        dev = simulate_sharded(seq, max_nstate=63, out="device")
        assert dev.vox0 == 0 and dev.count == 1000 and np.array_equal(np.asarray(dev).reshape(6, 40, 25), epg.simulate(seq, max_nstate=63))
        with pytest.raises(NotImplementedError):
            simulate_sharded(_probe_sequences()[1][1], max_nstate=63, out="device")
        # pipelined gather: 200 000 voxels in 4 sub-slabs of 50 048 / 50 048 / 50 048 / 49 856
        T1 = np.linspace(200, 3000, 500)[:, None]
        T2 = np.linspace(20, 300, 400)[None, :]
        big = wl.mse_sequence(epg, T1, T2, necho=6)
        ref = epg.simulate(big, max_nstate=63)
        sp = ShardedPlan(big, rank=0, world_size=1, max_nstate=63).bind()
        comm = _lib.get_comm(sp._ctx, 0, 1, lambda raw: raw, key=group_key(None))
        assert comm is next(iter(_lib._COMMS.values()))
        piped = SlabGather(sp, comm, nsub=4)
        assert piped.nsub == 4 and piped.sub == 50048 and [c for _, c, _ in piped.parts()] == [50048, 50048, 50048, 49856]
        piped.run_overlapped()
        serial = SlabGather(sp, comm, nsub=1)
        serial.run_serial()
        assert np.array_equal(piped.download(), ref) and np.array_equal(serial.download(), ref)
        assert np.array_equal(simulate_sharded(big, max_nstate=63, subslabs=4, via="rccl"), ref)
        assert np.array_equal(simulate_sharded(big, max_nstate=63), ref)
        piped.free()
        serial.free()
        # epgx_comm_reduce: sum over one rank = the data, in place and out of place; argument errors
        ctx = sp._ctx
        src, dst = _lib.DeviceBuffer(ctx, 4096), _lib.DeviceBuffer(ctx, 4096)
        data = np.arange(512, dtype=np.float64) * 0.5
        src.upload(data)
        comm.reduce(src.ptr.value, dst.ptr.value, 512, 0)
        assert np.array_equal(dst.download(np.float64, (512,)), data)
        comm.reduce(src.ptr.value, src.ptr.value, 512, 0)
        assert np.array_equal(src.download(np.float64, (512,)), data)
        with pytest.raises(_lib.EpgxError):
            comm.reduce(src.ptr.value, dst.ptr.value, 512, 3)
        with pytest.raises(_lib.EpgxError):
            comm.gather_part(src.ptr.value, dst.ptr.value, 4096, 1024, 0)      # stride below the block size
    finally:
        _lib.drop_comms()
        dist.destroy_process_group()
