"""Multi-GPU execution: one process per GPU, voxels sharded in contiguous slabs.

Voxels never interact (every operator is local to a voxel: SURVEY.md section 8e), so the
parameter grid is cut into `world_size` contiguous slabs of the flattened voxel index; every
rank compiles the same plan and runs its slab state-resident on its own GPU.  The only
communication is ONE gather of the per-rank signal slab `[n_adc][slab]` at the end
(`torch.distributed.gather`, i.e. RCCL send/recv over xGMI with backend "nccl"; each peer
uses its own point-to-point link to the root, so the gather is link-parallel, not ring-bound).
Slabs are padded to equal size so the gather needs no size exchange.
"""
import numpy as np

from . import _lib, functions


def slab_bounds(nvox, world_size):
    """equal slabs (last one ragged): returns (slab, [(vox0, count), ...])"""
    slab = -(-int(nvox) // int(world_size))
    out = []
    for r in range(world_size):
        v0 = min(r * slab, nvox)
        out.append((v0, min((r + 1) * slab, nvox) - v0))
    return slab, out


class ShardedPlan:
    """a compiled sequence bound to this rank's slab of the grid"""

    def __init__(self, sequence, *, rank, world_size, device=None, probes=None, fuse=True, **options):
        self.sequence = functions.flatten_sequence(sequence)
        self.enc, self.records, self.bounds = functions.compile_sequence(self.sequence, probes, options=options,
                                                                         fuse=fuse)
        self.rank, self.world_size = int(rank), int(world_size)
        self.nvox = self.enc.nvox
        self.slab, bounds = slab_bounds(self.nvox, world_size)
        self.vox0, self.count = bounds[rank]
        self.K = self.enc.capacity()
        self.K_resident = self.enc.packable() or self.K   # 4 / 2 voxels per wave when <= 16 / 32 orders
        self.n_adc = self.enc.n_adc
        self.device = device
        self._ctx = None
        self._plan = None

    # device objects are created lazily so that the host logic is testable without a GPU
    def bind(self, stream_ptr=None):
        self._ctx = _lib.get_context(self.device)
        if stream_ptr is not None:
            self._ctx.set_stream(stream_ptr)
        self._plan = self.enc.device_plan(self._ctx)
        return self

    def run(self, signal_ptr, mode="resident", state=None, part=None, signal_ld=None):
        """enqueue this rank's slab (or the sub-range `part` = (offset, count) of it);
        signal_ptr -> complex128 [n_adc][signal_ld] device buffer (signal_ld defaults to the slab)"""
        off, count = (0, self.count) if part is None else part
        count = max(0, min(count, self.count - off))
        if count == 0:
            return
        ld = self.slab if signal_ld is None else int(signal_ld)
        ctx, plan = self._ctx, self._plan
        vox0 = self.vox0 + off
        if mode == "resident":
            _lib.run(ctx, plan, 0, plan.n_ops, vox0, count, None, None, self.K_resident, signal_ptr, ld, 0)
            return
        begin = 0
        ends = self.bounds + ([plan.n_ops] if (not self.bounds or self.bounds[-1] < plan.n_ops) else [])
        first = True
        for end in ends:
            if end > begin:
                _lib.run(ctx, plan, begin, end, vox0, count, None if first else state, state,
                         self.K, signal_ptr, ld, 0)
                first = False
            begin = end

    def new_state(self, count=None):
        return _lib.DeviceState(self._ctx, max(self.count if count is None else count, 1), self.K)

    def assemble(self, gathered):
        """[world][n_adc][slab] -> (n_adc, *grid)"""
        gathered = np.asarray(gathered)
        full = np.moveaxis(gathered, 0, 1).reshape(self.n_adc, self.world_size * self.slab)
        return full[:, : self.nvox].reshape((self.n_adc,) + self.enc.grid)


def simulate_sharded(sequence, *, group=None, dst=0, compute=None, mode="resident", **options):
    """`simulate` over all ranks of a torch.distributed process group.

    Every rank must call it with the same sequence.  Returns the full signal
    `(n_adc, *grid)` (complex128 NumPy) on rank `dst`, None elsewhere.
    `compute(sharded_plan) -> torch tensor [n_adc, slab] complex128` replaces the GPU launch
    (used by the CPU/gloo tests to exercise the sharding and gather plumbing).
    """
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    sp = ShardedPlan(sequence, rank=rank, world_size=world, **options)
    if compute is not None:
        local = compute(sp)
    else:
        dev = torch.device("cuda", _lib.default_device())
        torch.cuda.set_device(dev)
        sp.bind(torch.cuda.current_stream().cuda_stream)
        local = torch.zeros((sp.n_adc, sp.slab), dtype=torch.complex128, device=dev)
        state = sp.new_state() if mode == "stream" else None
        sp.run(local.data_ptr(), mode=mode, state=state)
        sp._ctx.synchronize()   # (the null stream means "library's own stream": RCCL must not read early)
    real = torch.view_as_real(local).contiguous()
    bucket = [torch.empty_like(real) for _ in range(world)] if rank == dst else None
    dist.gather(real, bucket, dst=dst, group=group)
    if rank != dst:
        return None
    stacked = torch.stack([torch.view_as_complex(b) for b in bucket]).cpu().numpy()
    return sp.assemble(stacked)
