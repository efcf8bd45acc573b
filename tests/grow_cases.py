"""Sequences for the growing-state-matrix launches (rows_grow_kernel): run as a script it writes their signals to an .npz --
tests/test_gpu_parity.py runs it in a child process with EPGX_GROW=0 (the library reads the variable once per process) and
compares with its own results."""
import sys

import numpy as np


def cases(epg):
    from tests import sequences as sq
    from epgpy_amd import workloads as wl

    out = {}
    T1 = np.linspace(200, 3000, 37)[:, None]
    T2 = np.linspace(20, 300, 23)[None, :]                       # 851 voxels: the last wavefront holds 3
    for necho in (3, 8, 9, 16, 17, 20, 33, 40):                  # phase boundaries inside / at the end of / beyond the echo train
        out[f"mse_{necho}"] = (wl.mse_sequence(epg, T1, T2, necho=necho), {"max_nstate": 63})
    out["mse_20_unfused"] = (wl.mse_sequence(epg, T1, T2), {"max_nstate": 63, "fuse": False})       # pair runs [T E S ADC][E S]
    out["mse_20_cap40"] = (wl.mse_sequence(epg, T1, T2, necho=30), {"max_nstate": 40})             # truncation at K = 64
    B1 = np.linspace(0.8, 1.2, 5)[None, None, :]
    alpha, TR = sq.mrf_trains(90)
    out["mrf_90"] = (sq.mrf_ops(epg, T1[:9, :, None], T2[:, :7, None], B1, alpha, TR), {"max_nstate": 63})   # folded single runs
    out["mrf_40_cap35"] = (sq.mrf_ops(epg, T1[:9, :, None], T2[:, :7, None], B1, alpha[:40], TR[:40]), {"max_nstate": 35})
    rng = np.random.default_rng(11)
    for seed in range(6):                                        # random echo trains: repeated blocks, S(-1), spoilers, resets, PD, Z0
        grid = (int(rng.integers(2, 9)), int(rng.integers(2, 7)))
        seq = []
        for blk, rep in sq.random_train_blocks(rng, grid, nblocks=5):
            ops = sq.to_ops(epg, blk)
            seq += ops * rep
        seq.append(epg.ADC)
        out[f"train_{seed}"] = (seq, {"max_nstate": 63})
    return out


if __name__ == "__main__":
    import os

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from epgpy_amd import epg

    res = {}
    for name, (seq, kw) in cases(epg).items():
        res[name] = epg.simulate(seq, **kw)
    np.savez(sys.argv[1], **res)
