"""One process per GPU: an MRF dictionary over a (T1, T2, B1) grid, every rank simulating its contiguous voxel slab.

    python -m torch.distributed.run --standalone --local-addr 127.0.0.1 --nproc-per-node <GPUs> examples/mrf_sharded.py [m] [ntr]

The result arrives on rank 0 as ONE NumPy array that lives in shared memory: every rank downloads its slab over its own PCIe link
into its columns (`via="pcie"`, the default on one node).  `out="device"` instead keeps every slab in the
HBM of the GPU that computed it (dictionary matching on the device), `Adc(reduce=...)` probes send only their sums.
torch.distributed (gloo) is the side channel for a few flags; nothing of the simulation passes through it.
"""
import os
import sys
import time

import numpy as np
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from a checkout
from epgpy_amd import epg  # noqa: E402
from epgpy_amd.distributed import simulate_sharded  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 48
ntr = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dist.init_process_group("gloo")                      # RANK / WORLD_SIZE / MASTER_* from the launcher; GPU = LOCAL_RANK
rank, world = dist.get_rank(), dist.get_world_size()

T1 = np.linspace(300, 3000, m)[:, None, None]
T2 = np.linspace(20, 300, m)[None, :, None]
B1 = np.linspace(0.7, 1.3, m)[None, None, :]
rng = np.random.default_rng(0)
alpha = 10 + 50 * np.abs(np.sin(np.pi * np.arange(ntr) / 250)) * (0.6 + 0.4 * rng.random(ntr))
TR = 11 + 5 * rng.random(ntr)
seq = [epg.T(180 * B1, 90), epg.E(20.0, T1, T2)]
for a, tr in zip(alpha, TR):
    seq += [epg.T(a * B1, 90), epg.E(3.0, T1, T2), epg.ADC, epg.E(tr - 3.0, T1, T2), epg.S(1)]

simulate_sharded(seq, max_nstate=63)                 # first call: library, kernels, the shared result and its page-locking
t0 = time.perf_counter()
signal = simulate_sharded(seq, max_nstate=63, dtype=np.complex64)     # (complex64 records: half the bytes per link)
dt = time.perf_counter() - t0
if rank == 0:
    print(f"{world} rank(s), grid {m}^3, {ntr} TR: {signal.shape} {signal.dtype} in {1e3 * dt:.1f} ms = "
          f"{ntr * m ** 3 / dt:.3e} TR*voxels/s (host result included)")
else:
    assert signal is None
dist.destroy_process_group()
