"""CPU, world_size 2, gloo: the sharding + single-gather plumbing of epgpy_amd.distributed.

The GPU launch is replaced by the oracle (test double, `compute=` hook) so that what is
tested here is exactly the multi-process part: slab bounds, padding of the ragged last slab,
the gather to rank 0 and the re-assembly into (n_adc, *grid)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import sequences as sq


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from epgpy_amd import epg
        from epgpy_amd.distributed import simulate_sharded
        from oracle import epg_numpy as onp

        T1 = np.linspace(300, 2500, 7)[:, None]       # 7 x 3 = 21 voxels: ragged over 2 ranks
        T2 = np.linspace(30, 150, 3)[None, :]
        seq = sq.mse_ops(epg, T1, T2, necho=5)
        full = onp.simulate(sq.mse_tuples(T1, T2, necho=5), max_nstate=63).reshape(5, -1)

        def compute(sp):  # what the GPU would produce for this rank's slab
            block = np.zeros((sp.n_adc, sp.slab), dtype=np.complex128)
            block[:, : sp.count] = full[:, sp.vox0: sp.vox0 + sp.count]
            return torch.from_numpy(block)

        got = simulate_sharded(seq, compute=compute, max_nstate=63)
        if rank == 0:
            assert got.shape == (5, 7, 3)
            np.save(out_path, got)
        else:
            assert got is None
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_gather_world_size_2(tmp_path):
    from oracle import epg_numpy as onp

    out = str(tmp_path / "signal.npy")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    T1 = np.linspace(300, 2500, 7)[:, None]
    T2 = np.linspace(30, 150, 3)[None, :]
    ref = onp.simulate(sq.mse_tuples(T1, T2, necho=5), max_nstate=63)
    assert np.array_equal(np.load(out), ref)


def _worker_ragged(rank, world, port, out_path):
    """three ranks, 10 voxels (slabs 4 / 4 / 2: ragged), MRF-shaped train; then the same over the SUB-GROUP of
    global ranks (1, 2) with the group's rank 0 (= global rank 1) as destination"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from epgpy_amd import epg
        from epgpy_amd.distributed import simulate_sharded
        from oracle import epg_numpy as onp

        T1 = np.linspace(400, 2000, 5)[:, None, None]
        T2 = np.linspace(30, 150, 2)[None, :, None]
        B1 = np.array([1.0])[None, None, :]
        alpha, TR = sq.mrf_trains(6)
        seq = sq.mrf_ops(epg, T1, T2, B1, alpha, TR)
        full = onp.simulate(sq.mrf_tuples(T1, T2, B1, alpha, TR), max_nstate=63).reshape(6, -1)

        def compute(sp):
            assert sp.slab == -(-10 // sp.world_size) and sp.count == max(0, min(sp.slab, 10 - sp.vox0))
            block = np.zeros((sp.n_adc, sp.slab), dtype=np.complex128)
            block[:, : sp.count] = full[:, sp.vox0: sp.vox0 + sp.count]
            return torch.from_numpy(block)

        got = simulate_sharded(seq, compute=compute, max_nstate=63)
        if rank == 0:
            assert got.shape == (6, 5, 2, 1)
            np.save(out_path, got)
        else:
            assert got is None
        sub = dist.new_group([1, 2])
        if rank in (1, 2):
            got = simulate_sharded(seq, compute=compute, group=sub, dst=0, max_nstate=63)   # dst = rank 0 OF THE GROUP
            if rank == 1:
                assert got is not None and np.array_equal(got.reshape(6, -1), full)
                np.save(out_path + ".sub.npy", got)
            else:
                assert got is None
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_ragged_three_ranks_and_subgroup(tmp_path):
    from oracle import epg_numpy as onp

    out = str(tmp_path / "signal.npy")
    mp.spawn(_worker_ragged, args=(3, _free_port(), out), nprocs=3, join=True)
    T1 = np.linspace(400, 2000, 5)[:, None, None]
    T2 = np.linspace(30, 150, 2)[None, :, None]
    B1 = np.array([1.0])[None, None, :]
    alpha, TR = sq.mrf_trains(6)
    ref = onp.simulate(sq.mrf_tuples(T1, T2, B1, alpha, TR), max_nstate=63)
    assert np.array_equal(np.load(out), ref)
    assert np.array_equal(np.load(out + ".sub.npy"), ref)


# ------------------------------------------------------------------ probe semantics across ranks
def _probe_cases(epg):
    """(name, product sequence builder, simulate keywords, expected(F0 rows, Z0 rows) -> result) on a 6 x 3 grid"""
    T1 = np.linspace(300, 2500, 6)[:, None]
    T2 = np.linspace(30, 150, 3)[None, :]
    necho = 4
    phases = 58.5 * np.arange(necho) ** 2
    w_full = (np.arange(18).reshape(6, 3) + 1.0) * (1 + 0.5j)
    w_row = np.linspace(0.5, 2.0, 6)

    def seq_with(adcs):
        exc, rfc, rlx, sh = epg.T(90, 90), epg.T(120, 0), epg.E(5, T1, T2), epg.S(1)
        return [exc] + [op for n in range(necho) for op in (sh, rlx, rfc, sh, rlx, adcs[n])]

    tuples = lambda what: [("T", 90, 90)] + [("S", 1), ("E", 5, T1, T2), ("T", 120, 0), ("S", 1), ("E", 5, T1, T2), ("ADC", what)] * necho
    phasor = np.exp(1j * phases / 180 * np.pi)
    red0 = epg.Adc("F0", reduce=0)
    wsum = epg.Adc("F0", weights=w_full)
    wrow = epg.Adc("F0", weights=w_row, phase=15.0)
    cases = [
        ("plain", seq_with([epg.ADC] * necho), {}, lambda f, z: f),
        ("phase", seq_with([epg.Adc("F0", phase=p) for p in phases]), {}, lambda f, z: f * phasor[:, None, None]),
        ("F0_Z0", seq_with([epg.ADC] * necho), {"probe": ["F0", "Z0"]}, lambda f, z: (f, z)),
        ("Z0", seq_with([epg.ADC] * necho), {"probe": "Z0"}, lambda f, z: z),
        ("reduce0", seq_with([red0] * necho), {}, lambda f, z: f.sum(axis=1)),
        ("reduce_all", seq_with([epg.Adc("F0", reduce=True)] * necho), {}, lambda f, z: f.sum(axis=(1, 2))),
        ("weights", seq_with([wsum] * necho), {}, lambda f, z: (f * w_full).sum(axis=(1, 2))),
        ("weights_row_phase", seq_with([wrow] * necho), {}, lambda f, z: (f * w_row[:, None]).sum(axis=1) * np.exp(1j * 15.0 / 180 * np.pi)),
        ("mixed", seq_with([epg.ADC, red0, epg.ADC, red0]), {"asarray": False}, None),
        ("adc_time", seq_with([epg.ADC] * necho), {"adc_time": True}, lambda f, z: f),
    ]
    return cases, tuples


def _device_model(full_f, full_z):
    """test doubles of the kernels: the rank's rows from oracle records, and the slab-restricted weighted sums"""
    def compute(sp):
        block = np.zeros((sp.n_adc, sp.slab), dtype=np.complex128)
        for i, (_, slots) in enumerate(sp.records):
            for pb, slot in slots:
                src = full_f if pb._device_kind() == 0 else full_z
                block[slot, : sp.count] = src[i].reshape(-1)[sp.vox0: sp.vox0 + sp.count]
        return torch.from_numpy(block)

    def reduce_local(sp, rows, mask, weights, row0, step, count):
        grid = sp.enc.grid
        full = np.zeros((count, sp.nvox), dtype=np.complex128)
        full[:, sp.vox0: sp.vox0 + sp.count] = rows[row0: row0 + step * count: step][:, : sp.count]
        full = full.reshape((count,) + grid)
        if weights is not None:
            w = np.asarray(weights)
            full = full * w.reshape(w.shape + (1,) * (len(grid) - w.ndim))
        return full.sum(axis=tuple(1 + d for d, m in enumerate(mask) if m))

    return compute, reduce_local


def _worker_probes(rank, world, port, out_path):
    import pickle

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from epgpy_amd import epg
        from epgpy_amd.distributed import simulate_sharded
        from oracle import epg_numpy as onp

        cases, tuples = _probe_cases(epg)
        full_f = onp.simulate(tuples("F0"), max_nstate=63)
        full_z = onp.simulate(tuples("Z0"), max_nstate=63)
        compute, reduce_local = _device_model(full_f, full_z)
        solo = dist.new_group([0])        # a "world" of one rank: the one-GPU result through the same code
        results = {}
        for name, seq, kw, _ in cases:
            got = simulate_sharded(seq, compute=compute, reduce_local=reduce_local, max_nstate=63, **kw)
            one = simulate_sharded(seq, compute=compute, reduce_local=reduce_local, group=solo, max_nstate=63, **kw) if rank == 0 else None
            if rank == 0:
                results[name] = (got, one)
            else:
                assert got is None
        if rank == 0:
            with open(out_path, "wb") as fh:
                pickle.dump(results, fh)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_probe_semantics_across_ranks(tmp_path, world):
    """Adc(phase), ['F0', 'Z0'], Z0, reduce=, weights=, mixed probes, adc_time: N ranks == one rank == the reference's
    probe arithmetic (probe.py:141-165) on the oracle's records.  18 voxels over 2 (9 / 9) and 3 (6 / 6 / 6) ranks -- and
    slabs that cut grid rows for 2 ranks (6 x 3 grid: slab 9 = 3 rows; with 4 ranks it would not) are covered by the 7 x 3 test above"""
    import pickle

    from epgpy_amd import epg
    from oracle import epg_numpy as onp

    out = str(tmp_path / "probes.pkl")
    mp.spawn(_worker_probes, args=(world, _free_port(), out), nprocs=world, join=True)
    with open(out, "rb") as fh:
        results = pickle.load(fh)
    cases, tuples = _probe_cases(epg)
    f = onp.simulate(tuples("F0"), max_nstate=63)
    z = onp.simulate(tuples("Z0"), max_nstate=63)
    for name, _, kw, expected in cases:
        got, one = results[name]
        if name == "mixed":           # asarray=False: a tuple of per-ADC records, raw and reduced ones alternating
            assert len(got) == len(one) == 4
            for n in range(4):
                want = f[n] if n % 2 == 0 else f[n].sum(axis=0)
                assert got[n].shape == want.shape and np.allclose(got[n], want, rtol=0, atol=1e-13)
                assert np.allclose(one[n], got[n], rtol=0, atol=1e-13)
            continue
        if kw.get("adc_time"):
            (t_got, got), (t_one, one) = got, one
            assert np.array_equal(t_got, t_one) and len(t_got) == 4
        want = expected(f, z)
        pairs = zip(got, one, want) if isinstance(want, tuple) else [(got, one, want)]
        for g, o, w in pairs:
            assert g.shape == w.shape, name
            if "reduce" in name or "weights" in name:      # sums: the order of summation follows the slabs
                assert np.allclose(g, w, rtol=0, atol=1e-12), name
                assert np.allclose(g, o, rtol=0, atol=1e-12), name
            else:
                assert np.array_equal(g, w), name
                assert np.array_equal(g, o), name


# ------------------------------------------------------------------ host results over every rank's own link (shared memory)
def _worker_routes(rank, world, port, out_path):
    """the same sequence through both routes to a host result -- via="pcie" (every rank writes its columns of ONE shared-memory
    array) and via="rccl" (gather on the destination) -- in complex128 and complex64, raw + reduced probes mixed, on a ragged
    grid (23 voxels), with a destination other than rank 0"""
    import pickle

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from epgpy_amd import epg
        from epgpy_amd.distributed import simulate_sharded, same_node
        from oracle import epg_numpy as onp

        assert same_node()
        T1 = np.linspace(300, 2500, 23)[:, None]
        T2 = np.array([80.0])[None, :]
        necho = 4
        red = epg.Adc("F0", reduce=0)
        exc, rfc, rlx, sh = epg.T(90, 90), epg.T(120, 0), epg.E(5, T1, T2), epg.S(1)
        plain = [exc] + [op for n in range(necho) for op in (sh, rlx, rfc, sh, rlx, epg.ADC)]
        mixed = [exc] + [op for n in range(necho) for op in (sh, rlx, rfc, sh, rlx, epg.ADC if n % 2 == 0 else red)]
        tup = [("T", 90, 90)] + [("S", 1), ("E", 5, T1, T2), ("T", 120, 0), ("S", 1), ("E", 5, T1, T2), ("ADC", "F0")] * necho
        full_f = onp.simulate(tup, max_nstate=63)
        compute, reduce_local = _device_model(full_f, full_f)
        dst = world - 1
        results = {}
        for via in ("pcie", "rccl"):
            for dtype in (None, np.complex64):
                got = simulate_sharded(plain, compute=compute, reduce_local=reduce_local, max_nstate=63, via=via, dtype=dtype, dst=dst)
                mix = simulate_sharded(mixed, compute=compute, reduce_local=reduce_local, max_nstate=63, via=via, dtype=dtype, dst=dst,
                                       asarray=False)
                if rank == dst:
                    assert got.dtype == (np.complex64 if dtype else np.complex128)
                    results[via, "c64" if dtype else "c128"] = (np.array(got), [np.array(m) for m in mix])
                else:
                    assert got is None and mix is None
        # nothing of the shared results stays behind in /dev/shm
        assert not [name for name in os.listdir("/dev/shm") if name.startswith(f"epgx_result_{os.getpid()}_")]
        if rank == dst:
            with open(out_path, "wb") as fh:
                pickle.dump(results, fh)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3, 8])
def test_host_result_through_shared_memory_equals_the_gather(tmp_path, world):
    """simulate_sharded(out="host") on one node: via="pcie" (the default there) lets every rank fill its own columns of one
    shared-memory result -- N PCIe links instead of a funnel through the destination's GPU.  2 / 3 / 8 ranks (ragged: 23
    voxels; the last rank is the destination) give the one-rank result bit for bit, through both routes; complex64 results are the
    complex128 ones rounded once; reducing probes next to raw ones work on both routes"""
    import pickle

    from oracle import epg_numpy as onp

    out = str(tmp_path / "routes.pkl")
    mp.spawn(_worker_routes, args=(world, _free_port(), out), nprocs=world, join=True)
    with open(out, "rb") as fh:
        results = pickle.load(fh)
    T1 = np.linspace(300, 2500, 23)[:, None]
    T2 = np.array([80.0])[None, :]
    tup = [("T", 90, 90)] + [("S", 1), ("E", 5, T1, T2), ("T", 120, 0), ("S", 1), ("E", 5, T1, T2), ("ADC", "F0")] * 4
    ref = onp.simulate(tup, max_nstate=63)
    for via in ("pcie", "rccl"):
        got, mix = results[via, "c128"]
        assert got.shape == (4, 23, 1) and np.array_equal(got, ref), via
        got32, mix32 = results[via, "c64"]
        assert got32.dtype == np.complex64 and np.array_equal(got32, ref.astype(np.complex64)), via
        for n in range(4):
            want = ref[n] if n % 2 == 0 else ref[n].sum(axis=0)
            assert mix[n].shape == want.shape and np.allclose(mix[n], want, rtol=0, atol=1e-13), (via, n)
            assert mix32[n].dtype == np.complex64 and np.allclose(mix32[n], want, rtol=2e-7, atol=1e-7), (via, n)


def _worker_pool(rank, world, port, out_path):
    """the destination recycles the shared-memory segment of a result that was dropped, creates a second one while a result is
    alive, and keeps nothing in /dev/shm; the other rank's cache of mappings follows what the destination retires"""
    import gc
    import pickle

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from epgpy_amd import epg, distributed as dd
        from oracle import epg_numpy as onp

        T1 = np.linspace(300, 2500, 11)[:, None]
        T2 = np.array([60.0, 90.0])[None, :]
        seq = sq.mse_ops(epg, T1, T2, necho=3)
        full = onp.simulate(sq.mse_tuples(T1, T2, necho=3), max_nstate=63)
        compute, reduce_local = _device_model(full, full)
        run = lambda: dd.simulate_sharded(seq, compute=compute, reduce_local=reduce_local, max_nstate=63, via="pcie")   # noqa: E731
        log = {}
        a = run()
        if rank == 0:
            first = a.ctypes.data
            assert np.array_equal(a, full) and len(dd._SEGMENTS) == 1 and not dd._FREE
        del a
        gc.collect()
        b = run()                                   # the dropped result's segment again
        if rank == 0:
            log["reused"] = b.ctypes.data == first and len(dd._SEGMENTS) == 1
        c = run()                                   # b is alive: a second segment
        if rank == 0:
            log["second"] = c.ctypes.data != b.ctypes.data and len(dd._SEGMENTS) == 2 and np.array_equal(b, full) and np.array_equal(c, full)
        held = [b, c, run()]                        # a third while two are alive
        del b, c
        if rank == 0:
            log["third"] = len(dd._SEGMENTS) == 3
        del held
        gc.collect()
        if rank == 0:
            log["pooled_two_retired_one"] = len(dd._FREE) == dd.SHARED_POOL_PER_CLASS and len(dd._RETIRED) == 1 and len(dd._SEGMENTS) == 2
        d = run()                                   # tells the other rank which segment the destination gave up
        log_other = len(dd._SEGMENTS)
        del d
        gc.collect()
        dd.release_shared()
        assert not dd._SEGMENTS
        e = run()                                   # and everything starts over
        if rank == 0:
            log["after_release"] = np.array_equal(e, full)
        assert not [n for n in os.listdir("/dev/shm") if n.startswith("epgx_result_")]
        if rank == 0:
            with open(out_path, "wb") as fh:
                pickle.dump(log, fh)
        else:
            assert log_other == 2, log_other        # three mapped, one retired by the destination
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_shared_results_are_recycled(tmp_path):
    import pickle

    out = str(tmp_path / "pool.pkl")
    mp.spawn(_worker_pool, args=(2, _free_port(), out), nprocs=2, join=True)
    with open(out, "rb") as fh:
        log = pickle.load(fh)
    assert all(log.values()) and len(log) == 5, log
