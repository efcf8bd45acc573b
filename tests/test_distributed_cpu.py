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
