"""GPU: what a caller with HOST arrays waits for.

  * complex64 records (`simulate(..., dtype=np.complex64)`, enum epgx_signal_dtype): the arithmetic stays float64, every record
    is rounded once on the device, half the bytes cross PCIe.  Reference surface: Probe.acquire returns host copies
    (epgpy/probe.py:63-66), simulate stacks them (epgpy/functions.py:157-165).
  * `simulate_sharded(out="host")` with several ranks on ONE node: every rank downloads its slab over its own PCIe link into
    its columns of a shared-memory result.  Two rank processes share the box's one GPU here (the route needs no RCCL, which
    refuses two ranks on one device): the N > 1 control flow, the shared result and the downloads are the real ones.
"""
import ctypes
import os
import socket

import numpy as np
import pytest

from epgpy_amd import epg, _lib, functions, workloads as wl
from tests import sequences as sq

pytestmark = pytest.mark.gpu
C64_REL = 1.2e-7      # one rounding to float32 per component: 2^-24 = 6e-8 per component


def _rel(a, b):
    return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))


def test_complex64_records_small_large_and_pinned(golden):
    """dtype=complex64 on the three download routes of simulate(): small results (one download of the narrowed buffer), large
    ones (slab pipeline into a recycled page-locked block, then -- results kept -- through the staging ring into plain
    memory).  Every value is the complex128 value rounded ONCE: bit-equal to astype(complex64) of the complex128 run"""
    g1 = golden("g1_readme_mse")
    sig = epg.simulate(wl.mse_sequence(epg, 150.0, [30.0, 40.0, 50.0]), dtype=np.complex64)
    assert sig.dtype == np.complex64 and sig.shape == (20, 3)
    assert _rel(sig, g1["signal"]) < 1e-6                       # the north_star bar against the reference's own output
    assert np.array_equal(sig, g1["signal"].astype(np.complex64)) or _rel(sig, g1["signal"]) < C64_REL
    T1 = np.linspace(200, 3000, 512)[:, None]
    T2 = np.linspace(20, 300, 384)[None, :]
    seq = wl.mse_sequence(epg, T1, T2)                          # 20 x 512 x 384 x 16 B = 63 MB: the slab pipeline
    ref = epg.simulate(seq, max_nstate=63)
    held = []
    for _ in range(4):                                          # first results in pool blocks, later ones in plain arrays
        got = epg.simulate(seq, max_nstate=63, dtype=np.complex64)
        assert got.dtype == np.complex64 and got.shape == ref.shape
        assert np.array_equal(got, ref.astype(np.complex64))
        held.append(got)
    assert _rel(held[0], ref) < C64_REL
    for mode in ("stream", "stepwise"):
        got = epg.simulate(seq[:13], max_nstate=63, dtype=np.complex64, mode=mode)
        assert got.dtype == np.complex64 and np.array_equal(got, epg.simulate(seq[:13], max_nstate=63, mode=mode).astype(np.complex64))
    with pytest.raises(ValueError):
        epg.simulate(seq, max_nstate=63, dtype=np.float32)
    with pytest.raises(NotImplementedError):
        epg.simulate(seq, max_nstate=63, dtype=np.complex64, out="device")


def test_complex64_with_probe_flavours_jacobian_and_ngpu():
    """probes that are finished on the host (phases, weights, reductions), probe lists, Jacobians and the ngpu= path return
    complex64 as well, equal to the complex128 result rounded once (device-reduced sums: to float32 rounding)"""
    T1 = np.linspace(300, 2500, 40)[:, None]
    T2 = np.linspace(30, 150, 25)[None, :]
    necho = 6
    exc, rfc, rlx, sh = epg.T(90, 90), epg.T(120, 0), epg.E(5, T1, T2), epg.S(1)
    def train(adc):
        return [exc] + [op for n in range(necho) for op in (sh, rlx, rfc, sh, rlx, adc(n))]
    cases = [(train(lambda n: epg.ADC), {"probe": ["F0", "Z0"]}),
             (train(lambda n: epg.Adc("F0", phase=58.5 * n * n)), {}),
             (train(lambda n: epg.Adc("F0", reduce=0)), {}),
             (train(lambda n: epg.ADC), {"adc_time": True}),
             (train(lambda n: epg.ADC), {"ngpu": 1}),
             (train(lambda n: epg.ADC), {"asarray": False})]
    def flat(x):
        return [x] if isinstance(x, np.ndarray) else [y for item in x for y in flat(item)]
    for seq, kw in cases:
        want = epg.simulate(seq, max_nstate=63, **kw)
        got = epg.simulate(seq, max_nstate=63, dtype=np.complex64, **kw)
        for w, g in zip(flat(want), flat(got)):
            if w.dtype.kind != "c":
                assert np.array_equal(w, g)          # the ADC times
                continue
            assert g.dtype == np.complex64 and g.shape == w.shape
            assert np.allclose(g, w, rtol=C64_REL, atol=1e-7 * np.abs(w).max())
    tuples, ops, variables = sq.jac_mse(np.linspace(500, 2000, 30)[:, None], np.linspace(40, 120, 20)[None, :], 1.0, necho=5)
    jac = epg.Jacobian(variables)
    want = epg.simulate(ops(epg), probe=jac, max_nstate=63)
    got = epg.simulate(ops(epg), probe=jac, max_nstate=63, dtype=np.complex64)
    assert got.dtype == np.complex64 and got.shape == want.shape and np.array_equal(got, want.astype(np.complex64))


def test_signal_narrow_and_the_c_entries():
    """epgx_signal_narrow on a strided block; epgx_simulate_f64 / epgx_simulate_sharded_f64 with EPGX_SIGNAL_C64 through raw
    ctypes (small: one download; large: the slab pipeline; EPGX_SHARDED_GATHER=rccl: narrowed before the gather); a
    dtype that is no epgx_signal_dtype is rejected"""
    ctx = _lib.get_context()
    rng = np.random.default_rng(3)
    rows, ld, cols = 7, 1000, 777
    data = (rng.standard_normal((rows, ld)) + 1j * rng.standard_normal((rows, ld))) * 10.0 ** rng.integers(-30, 30, (rows, ld))
    src = _lib.DeviceBuffer(ctx, data.nbytes)
    src.upload(data)
    small = _lib.signal_narrow(ctx, src.ptr.value, ld, rows, cols)
    out = np.zeros((rows, cols), dtype=np.complex64)
    small.download_2d(out, 0, cols, rows, cols)
    with np.errstate(over="ignore"):
        assert np.array_equal(out, data[:, :cols].astype(np.complex64))
    assert ctx.lib.epgx_signal_narrow(ctx.handle, src.ptr, 10, small.ptr, cols, rows, cols) == -1      # src_ld < cols
    for side in (24, 640):          # 20 x 24 x 24 x 16 B = 184 kB; 20 x 640 x 640 x 16 B = 131 MB
        T1 = np.linspace(200, 3000, side)[:, None]
        T2 = np.linspace(20, 300, side)[None, :]
        seq = wl.mse_sequence(epg, T1, T2)
        ref = epg.simulate(seq, max_nstate=63)
        enc, _, _ = functions.compile_sequence(seq, None, options={"max_nstate": 63})
        desc, keep = _lib.plan_desc(**enc.plan_arrays(64))
        for code, dtype in ((0, np.complex128), (1, np.complex64)):
            got = np.zeros(ref.shape, dtype=dtype)
            assert ctx.lib.epgx_simulate_f64(ctx.handle, ctypes.byref(desc), 64, None, None, got.ctypes.data, None, code) == 0, ctx.lib.epgx_last_error()
            assert np.array_equal(got, ref.astype(dtype))
            for how in ("", "rccl"):
                os.environ["EPGX_SHARDED_GATHER"] = how
                try:
                    got = np.zeros(ref.shape, dtype=dtype)
                    assert ctx.lib.epgx_simulate_sharded_f64(ctypes.byref(desc), 64, 1, None, got.ctypes.data, code) == 0, ctx.lib.epgx_last_error()
                    assert np.array_equal(got, ref.astype(dtype))
                finally:
                    del os.environ["EPGX_SHARDED_GATHER"]
        got = np.zeros(ref.shape, dtype=np.complex128)
        assert ctx.lib.epgx_simulate_f64(ctx.handle, ctypes.byref(desc), 64, None, None, got.ctypes.data, None, 7) == -1
        assert ctx.lib.epgx_simulate_sharded_f64(ctypes.byref(desc), 64, 1, None, got.ctypes.data, 7) == -1


# ------------------------------------------------------------------ two rank processes, one GPU, a shared-memory result
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, out_path):
    import pickle

    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), EPGX_DEVICE="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from epgpy_amd.distributed import simulate_sharded

        results = {}
        T1 = np.linspace(200, 3000, 301)[:, None]           # 301 x 127 = 38 227 voxels: ragged over 2 and 3 ranks
        T2 = np.linspace(20, 300, 127)[None, :]
        seq = wl.mse_sequence(epg, T1, T2, necho=8)
        for dtype in (None, np.complex64):
            for mode in ("resident", "stream"):
                got = simulate_sharded(seq, max_nstate=63, mode=mode, dtype=dtype, dst=world - 1)
                if rank == world - 1:
                    results["mse", mode, "c64" if dtype else "c128"] = np.array(got)
                else:
                    assert got is None
        red = epg.Adc("F0", reduce=1)
        exc, rfc, rlx, sh = epg.T(90, 90), epg.T(120, 0), epg.E(5, T1, T2), epg.S(1)
        mixed = [exc] + [op for n in range(4) for op in (sh, rlx, rfc, sh, rlx, epg.ADC if n % 2 == 0 else red)]
        try:
            got = simulate_sharded(mixed, max_nstate=63, asarray=False, dst=world - 1)
            if rank == world - 1:
                results["mixed"] = [np.array(g) for g in got]
        except _lib.EpgxError as exc:      # (the reduction crosses RCCL, which refuses two ranks on one device: recorded, not fatal)
            if rank == world - 1:
                results["mixed_error"] = str(exc)
        tuples, ops, variables = sq.jac_mse(np.linspace(500, 2000, 61)[:, None], np.linspace(40, 120, 37)[None, :], 1.0, necho=5)
        got = simulate_sharded(ops(epg), probe=epg.Jacobian(variables), max_nstate=63, dst=world - 1)
        if rank == world - 1:
            results["jacobian"] = np.array(got)
            with open(out_path, "wb") as fh:
                pickle.dump(results, fh)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world", [2, 3])
def test_ranks_fill_one_shared_result_over_their_own_links(tmp_path, world):
    """simulate_sharded(out="host") with 2 / 3 rank processes (all on this box's one GPU): the destination's result -- a
    NumPy array in shared memory that every rank wrote its columns of through epgx_run_to_host -- equals the one-process
    simulate() bit for bit: state-resident and per-timestep mode, complex128 and complex64, a Jacobian probe"""
    import pickle

    import torch.multiprocessing as mp

    out = str(tmp_path / "ranks.pkl")
    mp.spawn(_rank_main, args=(world, _free_port(), out), nprocs=world, join=True)
    with open(out, "rb") as fh:
        results = pickle.load(fh)
    T1 = np.linspace(200, 3000, 301)[:, None]
    T2 = np.linspace(20, 300, 127)[None, :]
    seq = wl.mse_sequence(epg, T1, T2, necho=8)
    for mode in ("resident", "stream"):
        ref = epg.simulate(seq, max_nstate=63, mode=mode)
        assert np.array_equal(results["mse", mode, "c128"], ref)
        assert np.array_equal(results["mse", mode, "c64"], ref.astype(np.complex64))
    tuples, ops, variables = sq.jac_mse(np.linspace(500, 2000, 61)[:, None], np.linspace(40, 120, 37)[None, :], 1.0, necho=5)
    assert np.array_equal(results["jacobian"], epg.simulate(ops(epg), probe=epg.Jacobian(variables), max_nstate=63))
    if "mixed" in results:
        exc, rfc, rlx, sh = epg.T(90, 90), epg.T(120, 0), epg.E(5, T1, T2), epg.S(1)
        mixed = [exc] + [op for n in range(4) for op in (sh, rlx, rfc, sh, rlx, epg.ADC if n % 2 == 0 else epg.Adc("F0", reduce=1))]
        want = epg.simulate(mixed, max_nstate=63, asarray=False)
        for g, w in zip(results["mixed"], want):
            assert g.shape == w.shape and np.allclose(g, w, rtol=0, atol=1e-12)
    else:
        assert "mixed_error" in results


@pytest.mark.timeout(600)
def test_rccl_gather_moves_complex64_records():
    """via="rccl" with dtype=complex64: every sub-slab is narrowed where it was computed, the gather (one-rank communicator on this
    box) moves 8-byte records, the root downloads complex64 blocks; reducing probes next to raw ones still sum the complex128
    records.  Equal to the complex128 result rounded once, for the plain and the sub-slab layout, resident and per-timestep mode"""
    import torch.distributed as dist

    from epgpy_amd.distributed import simulate_sharded

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        T1 = np.linspace(200, 3000, 500)[:, None]
        T2 = np.linspace(20, 300, 400)[None, :]
        big = wl.mse_sequence(epg, T1, T2, necho=6)             # 200 000 voxels: four sub-slabs
        ref = epg.simulate(big, max_nstate=63)
        for kw in (dict(subslabs=4), dict(subslabs=1), dict(mode="stream")):
            got = simulate_sharded(big, max_nstate=63, via="rccl", dtype=np.complex64, **kw)
            assert got.dtype == np.complex64 and np.array_equal(got, epg.simulate(big, max_nstate=63, mode=kw.get("mode", "resident")).astype(np.complex64))
        assert np.array_equal(simulate_sharded(big, max_nstate=63, via="rccl"), ref)
        red = epg.Adc("F0", reduce=1)
        exc, rfc, rlx, sh = epg.T(90, 90), epg.T(120, 0), epg.E(5, T1[:60], T2[:, :50]), epg.S(1)
        mixed = [exc] + [op for n in range(4) for op in (sh, rlx, rfc, sh, rlx, epg.ADC if n % 2 == 0 else red)]
        want = epg.simulate(mixed, max_nstate=63, asarray=False)
        got = simulate_sharded(mixed, max_nstate=63, via="rccl", dtype=np.complex64, asarray=False)
        for g, w in zip(got, want):
            assert g.dtype == np.complex64 and g.shape == w.shape and np.allclose(g, w, rtol=2e-7, atol=1e-6)
    finally:
        _lib.drop_comms()
        dist.destroy_process_group()
