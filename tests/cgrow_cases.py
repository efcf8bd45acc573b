"""Sequences for the growing launches at 128 .. 512 orders per voxel (run_contig_grow_kernel): run as a script it writes their
signals to an .npz -- tests/test_gpu_parity.py runs it in a child process with EPGX_CGROW=0 EPGX_SPLIT_GROW=0 (the library reads the
variables once per process: the fixed-capacity kernels then take the launches) and compares with its own results."""
import sys

import numpy as np


def cases(epg):
    from tests import sequences as sq
    from epgpy_amd import workloads as wl

    out = {}
    T1 = np.linspace(200, 3000, 9)[:, None]
    T2 = np.linspace(20, 300, 7)[None, :]                        # 63 voxels: the last wavefront group holds 3
    # echo trains whose state matrix is never bounded: 2 n + 1 orders after n echoes -- capacities 128 (rows kernel unless
    # EPGX_CGROW=2), 256 and 512; phase boundaries inside, at the end of and beyond the train
    for necho in (40, 64, 65, 100, 127, 128, 150, 250):
        out[f"mse_{necho}"] = (wl.mse_sequence(epg, T1, T2, necho=necho), {})
    out["mse_90_unfused"] = (wl.mse_sequence(epg, T1, T2, necho=90), {"fuse": False})
    out["mse_200_cap150"] = (wl.mse_sequence(epg, T1, T2, necho=200), {"max_nstate": 150})         # truncation at K = 256
    out["mse_300_cap300"] = (wl.mse_sequence(epg, T1, T2, necho=300), {"max_nstate": 300})         # truncation at K = 512
    out["mse_400"] = (wl.mse_sequence(epg, T1[:3], T2, necho=400), {})                             # K = 1024: 16 orders per lane
    out["mse_900"] = (wl.mse_sequence(epg, T1[:2], T2[:, :5], necho=900), {})                      # K = 2048: four wavefronts per voxel,
    out["mse_900_cap1100"] = (wl.mse_sequence(epg, T1[:2], T2[:, :5], necho=900), {"max_nstate": 1100})   # joining at 512, 1024, 1536 orders
    B1 = np.linspace(0.8, 1.2, 5)[None, None, :]
    alpha, TR = sq.mrf_trains(300)
    out["mrf_300"] = (sq.mrf_ops(epg, T1[:5, :, None], T2[:, :3, None], B1, alpha, TR), {})       # three index spaces, folded records
    alpha_l, TR_l = sq.mrf_trains(1100)                          # 1101 orders -> K = 2048: two legs, three index spaces (table indices per voxel slab)
    out["mrf_1100"] = (sq.mrf_ops(epg, T1[:3, :, None], T2[:, :3, None], B1, alpha_l, TR_l), {})
    out["mrf_200_cap130"] = (sq.mrf_ops(epg, T1[:5, :, None], T2[:, :3, None], B1, alpha[:200], TR[:200]), {"max_nstate": 130})
    rng = np.random.default_rng(23)
    for seed in range(6):                                        # random echo trains: repeated blocks, S(-1), spoilers, resets, PD, Z0
        grid = (int(rng.integers(2, 9)), int(rng.integers(2, 7)))
        seq = []
        for _ in range(4):
            for blk, rep in sq.random_train_blocks(rng, grid, nblocks=5):
                seq += sq.to_ops(epg, blk) * (3 * rep)
        seq.append(epg.ADC)
        out[f"train_{seed}"] = (seq, {})
    return out


if __name__ == "__main__":
    import os

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from epgpy_amd import epg

    res = {}
    for name, (seq, kw) in cases(epg).items():
        if len(sys.argv) > 2 and name not in sys.argv[2:]:
            continue
        res[name] = epg.simulate(seq, **kw)
    np.savez(sys.argv[1], **res)
