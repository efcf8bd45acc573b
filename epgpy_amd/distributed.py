"""Multi-GPU execution: one process per GPU, voxels sharded in contiguous slabs.

Voxels never interact (every operator is local to a voxel: SURVEY.md section 8e), so the
parameter grid is cut into `world_size` contiguous slabs of the flattened voxel index; every
rank compiles the same plan and runs its slab state-resident on its own GPU.  The only
communication is ONE gather of the per-rank signal slab `[n_adc][slab]` at the end, on the device
side: libepgx's own RCCL gather (`epgx_comm_gather`: ncclSend / ncclRecv in one group, each peer
over its own point-to-point xGMI link to the root, so the gather is link-parallel, not
ring-bound), stream-ordered behind the kernels.  torch.distributed is only the side channel that
carries the communicator id (and the gather of the CPU tests, backend gloo).
Slabs are padded to equal size so the gather needs no size exchange.
"""
import ctypes

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

    def segments(self):
        """end indices of the per-timestep launches of mode="stream" (one per ADC-to-ADC segment + the tail)"""
        n_ops = len(self.enc.records)
        return self.bounds + ([n_ops] if (not self.bounds or self.bounds[-1] < n_ops) else [])

    def run(self, signal_ptr, mode="resident", state=None, part=None, signal_ld=None, segments=None):
        """enqueue this rank's slab (or the sub-range `part` = (offset, count) of it);
        signal_ptr -> complex128 [n_adc][signal_ld] device buffer (signal_ld defaults to the slab).
        mode="stream": `segments` = (first, last) restricts the call to those per-timestep launches
        (default all; launch 0 starts from equilibrium and only WRITES `state`)"""
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
        ends = self.segments()
        first, last = (0, len(ends)) if segments is None else segments
        for j in range(first, min(last, len(ends))):
            begin = ends[j - 1] if j else 0
            if ends[j] > begin:
                _lib.run(ctx, plan, begin, ends[j], vox0, count, None if j == 0 else state, state,
                         self.K, signal_ptr, ld, 0)

    def new_state(self, count=None):
        return _lib.DeviceState(self._ctx, max(self.count if count is None else count, 1), self.K)

    def assemble(self, gathered):
        """[world][n_adc][slab] -> (n_adc, *grid)"""
        gathered = np.asarray(gathered)
        full = np.moveaxis(gathered, 0, 1).reshape(self.n_adc, self.world_size * self.slab)
        return full[:, : self.nvox].reshape((self.n_adc,) + self.enc.grid)


def torch_id_exchange(group=None, src=0):
    """`exchange` callable for _lib.Comm: rank `src` of the torch.distributed group hands its communicator id out"""
    import torch.distributed as dist

    def exchange(raw):
        box = [raw]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, src) if group is not None else src, group=group)
        return box[0]

    return exchange


class SlabGather:
    """ONE gather of per-rank signal slabs [n_adc][slab] to `root` on the device side, over RCCL send / recv
    (epgx_comm_gather: every peer over its own xGMI link), stream-ordered behind the kernels on the library's
    stream.  The root's own slab is produced in place (its block of the gathered buffer)."""

    def __init__(self, sp, comm, root=0):
        self.sp, self.comm, self.root = sp, comm, int(root)
        ctx = sp._ctx
        self.block = 16 * sp.n_adc * sp.slab
        self.is_root = comm.rank == self.root
        if self.is_root:
            self.gathered = _lib.DeviceBuffer(ctx, max(self.block * comm.world_size, 16))
            self.local_ptr = self.gathered.ptr.value + comm.rank * self.block
            self.local = None
        else:
            self.gathered = None
            self.local = _lib.DeviceBuffer(ctx, max(self.block, 16))
            self.local_ptr = self.local.ptr.value
        if sp.count < sp.slab:   # ragged last slab: the padding columns are never written by the kernel
            _lib.check(ctx.lib.epgx_memset(ctx.handle, ctypes.c_void_p(self.local_ptr), 0, self.block), "epgx_memset")

    def __call__(self):
        self.comm.gather(self.local_ptr, self.gathered.ptr.value if self.is_root else 0, self.block, self.root)

    def download(self):
        """(root) the gathered blocks -> NumPy (n_adc, *grid): every block is copied straight to its columns of
        the result (strided D2H, no host-side re-assembly)"""
        sp = self.sp
        out = _lib.host_empty((sp.n_adc, sp.nvox), np.complex128)
        for r, (v0, count) in enumerate(slab_bounds(sp.nvox, sp.world_size)[1]):
            if count:
                self.gathered.download_2d(out, v0, count, sp.n_adc, sp.slab, offset=r * sp.n_adc * sp.slab)
        return out.reshape((sp.n_adc,) + sp.enc.grid)

    def free(self):
        for buf in (self.gathered, self.local):
            if buf is not None:
                buf.free()


def simulate_sharded(sequence, *, group=None, dst=0, compute=None, mode="resident", exchange=None, **options):
    """`simulate` over all ranks of a process group: every rank simulates its contiguous voxel slab, ONE gather.

    Every rank must call it with the same sequence.  Returns the full signal `(n_adc, *grid)` (complex128
    NumPy) on rank `dst` OF THE GROUP, None elsewhere.
    GPU path: the slabs meet on the device over RCCL (`_lib.Comm` / epgx_comm_gather); torch.distributed (or
    the `exchange` callable, see _lib.Comm) only carries the 128-byte communicator id.
    `compute(sharded_plan) -> torch tensor [n_adc, slab] complex128` replaces the GPU launch and the device-side
    gather by torch.distributed.gather (used by the CPU/gloo tests to exercise the sharding plumbing).
    """
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)    # ranks of the GROUP
    sp = ShardedPlan(sequence, rank=rank, world_size=world, **options)
    if compute is not None:
        import torch

        local = compute(sp)
        real = torch.view_as_real(local).contiguous()
        bucket = [torch.empty_like(real) for _ in range(world)] if rank == dst else None
        # torch addresses the destination by its GLOBAL rank
        dist.gather(real, bucket, dst=dist.get_global_rank(group, dst) if group is not None else dst, group=group)
        if rank != dst:
            return None
        stacked = torch.stack([torch.view_as_complex(b) for b in bucket]).cpu().numpy()
        return sp.assemble(stacked)
    sp.bind()            # the library's own stream: allocation, kernels and the gather are ordered on it
    comm = _lib.Comm(sp._ctx, rank, world, exchange or torch_id_exchange(group))
    gather = SlabGather(sp, comm, root=dst)
    state = sp.new_state() if mode == "stream" else None
    sp.run(gather.local_ptr, mode=mode, state=state)
    gather()
    out = gather.download() if rank == dst else None
    sp._ctx.synchronize()
    gather.free()
    comm.destroy()
    return out
