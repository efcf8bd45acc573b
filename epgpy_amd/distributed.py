"""Multi-GPU execution, one process per GPU: voxels sharded in contiguous slabs.

Voxels never interact (every operator is local to a voxel: SURVEY.md section 8e), so the
parameter grid is cut into `world_size` contiguous slabs of the flattened voxel index; every
rank compiles the same plan and runs its slab state-resident on its own GPU.  What travels
between GPUs is decided by the probes, exactly as on one GPU (`functions._simulate_device`):

* records of plain / phase-compensated probes (`ADC`, `Adc(phase=)`, `probe="Z0"`, lists): ONE gather of the
  per-rank signal slab, on the device side -- libepgx's own RCCL gather (`epgx_comm_gather_part`: ncclSend /
  ncclRecv in one group, each peer over its own point-to-point xGMI link to the root, so the gather is
  link-parallel, not ring-bound).  The slab is cut into sub-slabs: sub-slab k is on the wire (the
  communicator's stream) while sub-slab k + 1 computes;
* records of probes that sum over grid axes (`Adc(weights=, reduce=)`, epgpy/probe.py:141-165): every rank
  sums over its own voxels (`epgx_signal_reduce` on its slab) and the partial sums meet in ONE `ncclReduce`
  (`epgx_comm_reduce`) -- only the reduced records travel;
* `out="device"`: nothing travels, every rank keeps a `DeviceSignal` of its slab.

The root then finishes the records with the code of the one-GPU path (`functions._finish_records`: weights,
phase, `post`, Jacobian assembly), so N ranks return what one GPU returns.
torch.distributed is only the side channel: it carries the 128-byte communicator id and the ranks' "ready"
flags (and, in the CPU tests, stands in for RCCL: backend gloo).  The communicator is created once per
(context, group) and kept.
"""
import ctypes
import os

import numpy as np

from . import _lib, functions
from .functions import slab_bounds   # noqa: F401  (the slab arithmetic is shared with simulate(ngpu=N))


class ShardedPlan:
    """a compiled sequence bound to this rank's slab of the grid"""

    def __init__(self, sequence, *, rank, world_size, device=None, probes=None, fuse=True, variables=(), **options):
        self.sequence = functions.flatten_sequence(sequence)
        self.enc, self.records, self.bounds = functions.compile_sequence(self.sequence, probes, options=options,
                                                                         fuse=fuse, variables=variables)
        self.rank, self.world_size = int(rank), int(world_size)
        self.nvox = self.enc.nvox
        self.slab, bounds = slab_bounds(self.nvox, world_size)
        self.vox0, self.count = bounds[rank]
        self.K = self.enc.capacity()
        self.K_resident = self.enc.packable(derivatives=bool(variables)) or self.K   # 4 / 2 voxels per wave when <= 16 / 32 orders
        self.n_adc = self.enc.n_adc
        self.device = device
        self._ctx = None
        self._plan = None

    # device objects are created lazily so that the host logic is testable without a GPU
    def bind(self, stream_ptr=None):
        self._ctx = _lib.get_context(self.device)
        if stream_ptr is not None:
            self._ctx.set_stream(stream_ptr)
        self._plan = self.enc.device_plan(self._ctx, self.K)
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


def all_agree(ok, group=None):
    """True if `ok` on EVERY rank of the group (any backend: objects, not tensors).  Called before a rank enters a
    collective of the data path, so that a rank whose local preparation failed does not leave the others waiting"""
    import torch.distributed as dist

    flags = [None] * dist.get_world_size(group)
    dist.all_gather_object(flags, bool(ok), group=group)
    return all(flags)


def group_key(group=None):
    """hashable description of a process group (its global ranks): the key of the cached communicator"""
    import torch.distributed as dist

    if group is None:
        return ("world", dist.get_world_size())
    return tuple(dist.get_process_group_ranks(group))


class SlabGather:
    """The gather of per-rank signal slabs to `root` on the device side, over RCCL send / recv (every peer over its own
    xGMI link), on the communicator's stream.  The rank's slab is cut into `nsub` sub-slabs, each a contiguous block
    [n_adc][sub] of the rank's buffer, so that sub-slab k can leave while k + 1 computes (`run_overlapped`); the root's
    own sub-slabs are produced in place (its block of the gathered buffer).  nsub = 1: the plain layout [n_adc][slab]."""

    def __init__(self, sp, comm, root=0, nsub=1, dtype=np.complex128):
        self.sp, self.comm, self.root = sp, comm, int(root)
        ctx = sp._ctx
        nsub = max(1, min(int(nsub), max(sp.slab // 4096, 1)))     # (a sub-slab should still fill the chip)
        self.sub = -(-sp.slab // nsub)
        self.sub += -self.sub % 64 if nsub > 1 else 0               # whole wavefront groups
        self.nsub = -(-sp.slab // self.sub) if sp.slab else 1
        # complex64 records on the wire: every sub-slab is narrowed where it was computed (epgx_signal_narrow, behind its kernel on
        # the context's stream) and the gather moves half the bytes; the complex128 sub-slabs stay in `work` (reducing probes)
        self.narrow = np.dtype(dtype) == np.complex64
        self.isz = 8 if self.narrow else 16
        self.sub_bytes = self.isz * sp.n_adc * self.sub              # one sub-slab on the wire
        self.block = self.sub_bytes * self.nsub                      # one rank's block of the gathered buffer
        self.is_root = comm.rank == self.root
        if self.is_root:
            self.gathered = _lib.DeviceBuffer(ctx, max(self.block * comm.world_size, 16), itemsize=self.isz)
            self.local_ptr = self.gathered.ptr.value + comm.rank * self.block
            self.local = None
        else:
            self.gathered = None
            self.local = _lib.DeviceBuffer(ctx, max(self.block, 16), itemsize=self.isz)
            self.local_ptr = self.local.ptr.value
        self.work = _lib.DeviceBuffer(ctx, max(16 * sp.n_adc * self.sub * self.nsub, 16)) if self.narrow else None
        self.compute_ptr = self.work.ptr.value if self.narrow else self.local_ptr      # where the kernels write (complex128)
        self.compute_bytes = 16 * sp.n_adc * self.sub                                  # ... one sub-slab of it
        if sp.count < self.nsub * self.sub:   # padding columns are never written by the kernel
            _lib.check(ctx.lib.epgx_memset(ctx.handle, ctypes.c_void_p(self.compute_ptr), 0, self.compute_bytes * self.nsub), "epgx_memset")

    def parts(self):
        """(offset into the slab, voxels, sub-slab index) per sub-slab"""
        return [(k * self.sub, max(0, min(self.sub, self.sp.count - k * self.sub)), k) for k in range(self.nsub)]

    def _compute(self, k, off, cnt, mode, state):
        """sub-slab k into the compute buffer and, for complex64 wire records, narrowed into the rank's block"""
        self.sp.run(self.compute_ptr + k * self.compute_bytes, mode=mode, state=state, part=(off, cnt), signal_ld=self.sub)
        if self.narrow:
            _lib.signal_narrow_into(self.sp._ctx, self.compute_ptr + k * self.compute_bytes, self.sub, self.local_ptr + k * self.sub_bytes,
                                    self.sub, self.sp.n_adc, self.sub)

    def run_overlapped(self, mode="resident", state=None):
        """compute sub-slab k + 1 while sub-slab k travels: the kernels go to the context's stream, every gather part to
        the communicator's stream behind its kernel (epgx_comm_gather_part); one join at the end"""
        gathered = self.gathered.ptr.value if self.is_root else 0
        for off, cnt, k in self.parts():
            byte0 = k * self.sub_bytes
            self._compute(k, off, cnt, mode, state)
            self.comm.gather_part(self.local_ptr + byte0, gathered + byte0 if gathered else 0, self.sub_bytes, self.block, self.root)
        self.comm.join()

    def run_serial(self, mode="resident", state=None):
        """all kernels first, then the gather of the whole buffer (what `run_overlapped` is measured against)"""
        for off, cnt, k in self.parts():
            self._compute(k, off, cnt, mode, state)
        self()

    def __call__(self):
        """the gather alone: every rank's whole buffer in one part"""
        self.comm.gather(self.local_ptr, self.gathered.ptr.value if self.is_root else 0, self.block, self.root)

    def download(self, out=None):
        """(root) the gathered blocks -> NumPy (n_adc, *grid) of the wire's record type: every block is copied straight to its
        columns of the result (strided D2H, no host-side re-assembly)"""
        sp = self.sp
        dtype = np.complex64 if self.narrow else np.complex128
        if out is None:
            out = _lib.result_empty(sp._ctx, (sp.n_adc, sp.nvox), dtype)
        flat = out.reshape(sp.n_adc, sp.nvox)
        for r, (v0, count) in enumerate(slab_bounds(sp.nvox, sp.world_size)[1]):
            for k in range(self.nsub):
                cnt = max(0, min(self.sub, count - k * self.sub))
                if cnt:
                    self.gathered.download_2d(flat, v0 + k * self.sub, cnt, sp.n_adc, self.sub,
                                              offset=(r * self.block + k * self.sub_bytes) // self.isz)
        return out.reshape((sp.n_adc,) + sp.enc.grid)

    def free(self):
        for buf in (self.gathered, self.local, self.work):
            if buf is not None:
                buf.free()


# ---------------------------------------------------------------------------------- host results over N PCIe links
SHARED_POOL_PER_CLASS = 2          # recycled segments of about one size the destination keeps (a loop that rebinds its result alternates)
SHARED_POOL_MAX_BYTES = 64 << 30   # ... and in total
SHARED_REGISTER_MAX = int(os.environ.get("EPGX_SHARED_REGISTER_MAX_GB", 64)) << 30   # segments up to this size are page-locked on every rank
                                   # (hipHostRegister: ~0.2 ms per MB, once per segment and rank; a refusal leaves the staged route).  Large
                                   # results matter most: N ranks staging through host copy threads move every byte twice through host
                                   # memory, N direct DMA streams once


class _Segment:
    """this rank's mapping of one /dev/shm file (the file itself is unlinked as soon as every rank has it open: the memory
    lives as long as some mapping does)"""

    def __init__(self, name, nbytes, create):
        import mmap

        fd = os.open(name, (os.O_CREAT | os.O_EXCL | os.O_RDWR) if create else os.O_RDWR, 0o600)
        try:
            if create:
                os.ftruncate(fd, nbytes)
            self.map = mmap.mmap(fd, nbytes)
        finally:
            os.close(fd)
        self.name, self.nbytes = name, int(nbytes)
        self.addr = ctypes.addressof(ctypes.c_char.from_buffer(self.map))
        self.ctx = None                 # the context the mapping is registered with (page-locked), or None

    def register(self, ctx):
        """page-lock the mapping for `ctx`'s device (once): this rank's copies into it are then direct DMA"""
        if self.ctx is None and self.nbytes <= SHARED_REGISTER_MAX:
            if ctx.lib.epgx_host_register(ctx.handle, ctypes.c_void_p(self.addr), self.nbytes) == 0:
                self.ctx = ctx          # (a refusal -- locked-memory limit -- leaves the staged route: slower, not wrong)

    def array(self, shape, dtype, owner=None):
        """ndarray on the first bytes of the mapping; `owner` is kept alive by the array and all its views"""
        count = int(np.prod(shape))
        raw = (ctypes.c_char * max(count * np.dtype(dtype).itemsize, 1)).from_address(self.addr)
        raw._epgx_owner = (owner, self)
        return np.frombuffer(raw, dtype=dtype, count=count).reshape(shape)

    def close(self):
        if self.map is None:
            return
        if self.ctx is not None and _lib._alive() and self.ctx.handle:
            self.ctx.lib.epgx_host_unregister(self.ctx.handle, ctypes.c_void_p(self.addr))
        self.ctx = None
        try:
            self.map.close()
        except BufferError:       # (ctypes.from_buffer exports are alive: the mapping goes with the last of them)
            pass
        self.map = None


_SEGMENTS = {}        # name -> _Segment: this rank's open mappings (its own segments and other destinations')
_FREE = []            # (destination side) names of pooled segments no result array refers to any more
_GROUP_OF = {}        # (destination side) segment name -> key of the group whose ranks have it mapped (only they can reuse it)
_RETIRED = []         # (destination side) names dropped from the pool: told to the other ranks with the next call
_LEASED = set()       # (destination side) names of segments that a live result array refers to
_counter = [0]


class _Lease:
    """what a result array of the destination keeps alive: when the last view of the array dies the segment goes back to
    the destination's pool (or is retired when the pool is full)"""

    def __init__(self, name):
        self.name = name
        _LEASED.add(name)

    def __del__(self):
        try:
            _LEASED.discard(self.name)
            seg = _SEGMENTS.get(self.name)
            if seg is None:
                return
            alike = sum(1 for n in _FREE if seg.nbytes // 2 <= _SEGMENTS[n].nbytes <= 2 * seg.nbytes)
            pooled = sum(_SEGMENTS[n].nbytes for n in _FREE)
            if alike < SHARED_POOL_PER_CLASS and pooled + seg.nbytes <= SHARED_POOL_MAX_BYTES:
                _FREE.append(self.name)
            else:
                _RETIRED.append(self.name)
                _SEGMENTS.pop(self.name).close()
        except Exception:   # noqa: BLE001  (interpreter shutdown)
            pass


def release_shared():
    """drop this rank's cached mappings of shared results (and, on a destination, its pool of recycled segments); results
    that are still referenced stay valid.  Call it on EVERY rank of a group or on none: a destination that keeps its pool
    would hand out a segment whose file is gone to a rank that has dropped its mapping (the call then fails on all ranks)"""
    for name in list(_FREE):
        _RETIRED.append(name)
    del _FREE[:]
    for name in list(_SEGMENTS):
        if name not in _LEASED:         # (a live result keeps its mapping; its lease retires the segment later)
            _SEGMENTS.pop(name).close()


class SharedResult:
    """A result array [n_adc, *grid] in POSIX shared memory (/dev/shm) that every rank of ONE node maps, so that each
    rank's GPU downloads its voxel slab over ITS OWN PCIe link straight into its columns -- no device-side gather, no
    funnel through the destination's GPU and its one link (the reference returns host copies: epgpy/probe.py:63-66;
    its result is one array: epgpy/functions.py:157-165).  The destination rank owns the segments: it recycles the one
    of a result that has been dropped (a loop that rebinds its result alternates between two, like the page-locked blocks
    of the one-process path) or creates a file that the others map and that is unlinked as soon as everyone has it open.
    Mappings are cached per rank and page-locked on first use (`_Segment.register`), so from the second call on a rank's
    copy is a direct DMA into pages that already exist.  The memory of a segment lives as long as some mapping of it
    does; on the destination the returned ndarray (and every view of it) keeps its segment out of the pool.
    COLLECTIVE: every rank of the group constructs it with the same arguments."""

    def __init__(self, shape, dtype, group, rank, dst):
        import torch.distributed as dist

        self.shape, self.dtype = tuple(int(d) for d in shape), np.dtype(dtype)
        self.nbytes = max(int(np.prod(self.shape)) * self.dtype.itemsize, 1)
        self.segment = self.array = None
        failure, msg = None, None
        if rank == dst:
            try:
                gkey = group_key(group)
                fits = [n for n in _FREE if _GROUP_OF.get(n) == gkey and
                        self.nbytes <= _SEGMENTS[n].nbytes <= self.nbytes + max(self.nbytes // 4, 1 << 20)]
                if fits:
                    name = min(fits, key=lambda n: _SEGMENTS[n].nbytes)
                    _FREE.remove(name)
                    msg = (name, _SEGMENTS[name].nbytes, False)
                else:
                    _counter[0] += 1
                    name = f"/dev/shm/epgx_result_{os.getpid()}_{_counter[0]}"
                    free = os.statvfs("/dev/shm")
                    if free.f_bavail * free.f_frsize < self.nbytes + (64 << 20):    # (a tmpfs that overflows ends in SIGBUS, not in an error)
                        raise OSError(f"/dev/shm has {free.f_bavail * free.f_frsize >> 20} MiB free, the result needs {self.nbytes >> 20}")
                    _SEGMENTS[name] = _Segment(name, self.nbytes, create=True)
                    _GROUP_OF[name] = gkey
                    msg = (name, self.nbytes, True)
            except Exception as exc:   # noqa: BLE001   (the other ranks are waiting for the name: fail together)
                failure, msg = exc, None
        box = [(msg, list(_RETIRED) if rank == dst else None)]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, dst) if group is not None else dst, group=group)
        msg, retired = box[0]
        if rank == dst:
            del _RETIRED[:len(retired)]
        else:
            for name in retired:               # segments the destination has given up: drop this rank's mappings of them
                if name in _SEGMENTS:
                    _SEGMENTS.pop(name).close()
        if msg is not None and rank != dst and msg[0] not in _SEGMENTS:
            try:
                _SEGMENTS[msg[0]] = _Segment(msg[0], msg[1], create=False)
            except Exception as exc:   # noqa: BLE001
                failure = exc
        ok = all_agree(msg is not None and msg[0] in _SEGMENTS, group)      # (also the barrier: everyone has the file open, or gave up)
        if rank == dst and msg is not None and msg[2]:
            try:
                os.unlink(msg[0])
            except OSError:
                pass
        if not ok:
            if msg is not None and msg[0] in _SEGMENTS:
                _SEGMENTS.pop(msg[0]).close()
            raise failure or _lib.EpgxError("another rank could not map the shared result")
        self.segment = _SEGMENTS[msg[0]]
        self.array = self.segment.array(self.shape, self.dtype, owner=_Lease(msg[0]) if rank == dst else None)

    def close(self):
        """(ranks other than the destination, after their columns are written) forget the array; the mapping stays cached"""
        self.array = None


_SAME_NODE = {}       # group key -> bool (a group does not move between machines)


def same_node(group=None):
    """True if every rank of the group runs on this machine (same boot id and hostname): shared memory reaches them all.
    COLLECTIVE on first use per group, remembered afterwards"""
    import socket
    import torch.distributed as dist

    key = group_key(group)
    if key in _SAME_NODE:
        return _SAME_NODE[key]

    try:
        boot = open("/proc/sys/kernel/random/boot_id").read().strip()
    except OSError:
        boot = ""
    mine = (socket.gethostname(), boot)
    seen = [None] * dist.get_world_size(group)
    dist.all_gather_object(seen, mine, group=group)
    _SAME_NODE[key] = all(other == mine for other in seen)
    return _SAME_NODE[key]


class _RcclBackend:
    """this rank's GPU + libepgx's communicator"""

    def __init__(self, sp, group, rank, world, dst, mode, exchange, nsub, need_comm=True, dtype=np.complex128):
        import torch.distributed as dist   # noqa: F401  (the side channel)

        self.sp, self.group, self.rank, self.dst, self.mode = sp, group, rank, dst, mode
        self.comm = self.gather = self.local = None
        self.dtype = np.dtype(dtype)
        failure = None
        try:
            sp.bind()            # the library's own stream: allocation and kernels are ordered on it, transfers behind them
        except Exception as exc:   # noqa: BLE001
            failure = exc
        if not all_agree(failure is None, group):
            raise failure or _lib.EpgxError("another rank could not bind its GPU")
        self.nsub = nsub
        if not need_comm:
            return
        try:
            self.comm = _lib.get_comm(sp._ctx, rank, world, exchange or torch_id_exchange(group), key=group_key(group),
                                       agree=lambda flag: all_agree(flag, group))
        except Exception as exc:   # noqa: BLE001   (_lib.Comm hands an all-zero id around when rank 0 fails: nobody waits)
            failure = exc
        if not all_agree(failure is None, group):
            raise failure or _lib.EpgxError("another rank could not create its communicator")

    def run(self, gather):
        """simulate this rank's slab; gather = the raw rows are wanted at the root"""
        sp = self.sp
        state = sp.new_state() if self.mode == "stream" else None
        if gather:
            self.gather = SlabGather(sp, self.comm, root=self.dst, nsub=self.nsub if self.mode == "resident" else 1, dtype=self.dtype)
            if self.mode == "resident":
                self.gather.run_overlapped()
            else:
                self.gather.run_serial(mode="stream", state=state)
            self.local_ptr, self.ld = self.gather.compute_ptr, self.gather.sub       # (complex128: what reducing probes sum over)
            self.one_block = self.gather.nsub == 1
        else:
            self.local = _lib.DeviceBuffer(sp._ctx, 16 * max(sp.n_adc, 1) * max(sp.count, 1))
            self.local_ptr, self.ld, self.one_block = self.local.ptr.value, max(sp.count, 1), True
            sp.run(self.local_ptr, mode=self.mode, state=state, signal_ld=self.ld)

    def run_to_shared(self, shared):
        """simulate this rank's slab and download it over THIS GPU's PCIe link into its columns of the shared result
        (state-resident: in sub-slabs whose columns leave while the next sub-slab computes, epgx_run_to_host).  The whole
        slab also stays in the rank's device buffer [n_adc][count] for reducing probes next to the raw ones"""
        sp, ctx = self.sp, self.sp._ctx
        flat = shared.array.reshape(max(sp.n_adc, 1), -1)
        self.local = _lib.DeviceBuffer(ctx, 16 * max(sp.n_adc, 1) * max(sp.count, 1))
        self.local_ptr, self.ld, self.one_block = self.local.ptr.value, max(sp.count, 1), True
        if not sp.count or not sp.n_adc:
            return
        shared.segment.register(ctx)      # (once per segment and rank: page-locked from here on)
        if self.mode == "resident":
            _lib.run_to_host(ctx, sp._plan, sp.K_resident, self.local_ptr, flat, vox0=sp.vox0, nvox=sp.count)
        else:
            sp.run(self.local_ptr, mode="stream", state=sp.new_state(), signal_ld=self.ld)
            if flat.dtype == np.complex64:       # narrowed on the device: half the bytes over this rank's link
                small = _lib.signal_narrow(ctx, self.local_ptr, self.ld, sp.n_adc, sp.count)
                small.download_2d(flat, sp.vox0, sp.count, sp.n_adc, sp.count)
                small.free()
            else:
                self.local.download_2d(flat, sp.vox0, sp.count, sp.n_adc, self.ld)

    def reduce(self, mask, weights, row0, step, count):
        """weighted sums over the masked grid axes: over this rank's voxels on its GPU, then ONE ncclReduce"""
        sp, ctx = self.sp, self.sp._ctx
        if not self.one_block:
            raise NotImplementedError("reducing probes next to raw ones: simulate_sharded(..., subslabs=1)")
        part = _lib.signal_reduce(ctx, self.local_ptr, self.ld, row0, step, count, sp.enc.grid, mask, weights,
                                  vox0=sp.vox0, nvox=sp.count, to_host=False)
        self.comm.reduce(part.ptr.value, part.ptr.value, 2 * int(np.prod(part.shape)), self.dst)    # in place at the root
        res = part.download(np.complex128, part.shape) if self.rank == self.dst else None
        if self.rank != self.dst:
            ctx.synchronize()
        part.free()
        return res

    def raw(self):
        return self.gather.download() if self.rank == self.dst else None

    def device_signals(self, nprobe):
        sp = self.sp
        return [functions.DeviceSignal(self.local, sp.n_adc, sp.enc.grid, j, max(nprobe, 1), vox0=sp.vox0, count=max(sp.count, 1))
                for j in range(nprobe)]

    def finish(self, keep_local=False):
        self.sp._ctx.synchronize()
        if self.gather is not None:
            self.gather.free()
        if self.local is not None and not keep_local:
            self.local.free()


class _HookBackend:
    """the CPU tests' stand-in: `compute(sp)` produces the rank's rows, `reduce_local(sp, rows, mask, weights, row0, step,
    count)` its partial sums (test doubles of the kernels), torch.distributed (gloo) moves them"""

    def __init__(self, sp, group, rank, world, dst, compute, reduce_local):
        self.sp, self.group, self.rank, self.world, self.dst = sp, group, rank, world, dst
        self.compute, self.reduce_local = compute, reduce_local
        self.rows = self.bucket = None

    def _global(self, r):
        import torch.distributed as dist

        return dist.get_global_rank(self.group, r) if self.group is not None else r

    def run(self, gather):
        import torch
        import torch.distributed as dist

        self.rows = self.compute(self.sp)
        if not gather:
            return
        real = torch.view_as_real(self.rows).contiguous()
        self.bucket = [torch.empty_like(real) for _ in range(self.world)] if self.rank == self.dst else None
        # torch addresses the destination by its GLOBAL rank
        dist.gather(real, self.bucket, dst=self._global(self.dst), group=self.group)

    def run_to_shared(self, shared):
        self.rows = self.compute(self.sp)
        sp = self.sp
        if sp.count and sp.n_adc:
            flat = shared.array.reshape(sp.n_adc, -1)
            flat[:, sp.vox0:sp.vox0 + sp.count] = self.rows.numpy()[:, :sp.count]

    def reduce(self, mask, weights, row0, step, count):
        import torch
        import torch.distributed as dist

        if self.reduce_local is None:
            raise NotImplementedError("this sequence reduces on the device: pass reduce_local= next to compute=")
        part = np.ascontiguousarray(self.reduce_local(self.sp, self.rows.numpy(), mask, weights, row0, step, count), dtype=np.complex128)
        ten = torch.view_as_real(torch.from_numpy(part)).contiguous()
        dist.reduce(ten, dst=self._global(self.dst), op=dist.ReduceOp.SUM, group=self.group)
        return torch.view_as_complex(ten).numpy() if self.rank == self.dst else None

    def raw(self):
        import torch

        if self.rank != self.dst:
            return None
        return self.sp.assemble(torch.stack([torch.view_as_complex(b) for b in self.bucket]).numpy())

    def device_signals(self, nprobe):
        raise NotImplementedError('out="device" needs the GPU path')

    def finish(self, keep_local=False):
        pass


def simulate_sharded(sequence, *, group=None, dst=0, probe=None, adc_time=False, asarray=True, out="host", mode="resident",
                     subslabs=4, exact_partials=False, compute=None, reduce_local=None, exchange=None, via="auto", dtype=None,
                     **options):
    """`simulate` over all ranks of a process group: every rank simulates its contiguous voxel slab.

    Every rank must call it with the same sequence and arguments.  Returns what `epg.simulate` returns for the same
    sequence / `probe` / `adc_time` / `asarray` on rank `dst` OF THE GROUP, None elsewhere -- identical to the one-GPU
    result bit for bit (sums of `Adc(reduce=)` to rounding: the order of summation follows the slabs).
    `out="device"`: nothing is gathered; every rank receives `DeviceSignal` handles on its own slab (`.vox0`, `.count`).
    `via`: how the raw records reach the destination's HOST array.  "pcie": the result lives in shared memory that every rank of
    the node maps, and every GPU downloads its slab over its own PCIe link into its columns (N links in parallel, nothing
    crosses xGMI, no funnel through the destination's GPU); "rccl": the slabs are gathered on the destination's GPU (RCCL
    send / recv over xGMI) and leave through its one link; "auto" (default): "pcie" when all ranks run on one machine.
    `dtype`: np.complex64 halves the bytes of the result (the arithmetic stays float64; every value is rounded once, 6e-8
    relative); default complex128 as the reference.
    `subslabs` (via="rccl"): pieces a rank's slab is cut into so that piece k travels while k + 1 computes.
    GPU path: the slabs / partial sums meet on the device over RCCL (`_lib.Comm`); torch.distributed (or the `exchange`
    callable, see _lib.Comm) only carries the 128-byte communicator id and the ranks' "ready" flags.
    `compute(sharded_plan) -> torch tensor [n_adc, slab] complex128` (and `reduce_local`, see _HookBackend) replace the
    GPU launch and the device-side collectives by torch.distributed ones: the CPU / gloo tests exercise the sharding
    plumbing and the probe semantics with them.
    """
    import torch.distributed as dist

    if out not in ("host", "device"):
        raise ValueError(f'out={out!r}: expected "host" or "device"')
    if mode not in ("resident", "stream"):
        raise ValueError(f"mode={mode!r}: sharded runs are device runs (resident or stream)")
    if via not in ("auto", "pcie", "rccl"):
        raise ValueError(f'via={via!r}: expected "auto", "pcie" or "rccl"')
    dtype = functions.signal_dtype(dtype)
    if dtype != np.complex128 and out == "device":
        raise NotImplementedError('out="device" keeps complex128 records')
    rank, world = dist.get_rank(group), dist.get_world_size(group)    # ranks of the GROUP
    flat = functions.flatten_sequence(sequence)
    probes = []
    if probe:
        probes = probe if isinstance(probe, (tuple, list)) else [probe]
        probes = [pb if isinstance(pb, (functions.Probe, type(None))) else functions.Probe(pb) for pb in probes]
    if not any(isinstance(op, functions.Probe) for op in flat):
        raise ValueError("Cannot simulate sequence without at least one Probe/ADC operator")
    variables = functions._jacobian_variables(flat, probes)
    if len(variables) > _lib.MAX_VARS:
        raise NotImplementedError(f"sharded runs carry at most {_lib.MAX_VARS} derivative variables per call")
    if variables and (mode == "stream" or out == "device"):
        raise NotImplementedError("derivatives run state-resident and return host arrays")
    # (probes the kernel cannot record raise NotImplementedError here, on every rank alike)
    sp = ShardedPlan(flat, rank=rank, world_size=world, probes=probes, variables=variables, **options)
    if variables and exact_partials:
        sp.enc.deriv_flags |= _lib.DERIV_THROUGH_PLAIN_OPS
    if variables and sp.K > _lib.MAX_DERIV_K:
        raise NotImplementedError(f"derivatives with {sp.enc.peak + 1} phase states per voxel: the device path keeps at most "
                                  f"{_lib.MAX_DERIV_K}; bound the state matrix with max_nstate=...")
    if variables and len(variables) > _lib.max_vars(sp.K):
        raise NotImplementedError(f"sharded runs at {sp.K} orders per voxel carry at most {_lib.max_vars(sp.K)} derivative variable(s) per call")
    records = sp.records
    groups = functions._reduction_groups(records, sp.enc.grid) if not variables else {}
    need_raw = any(id(pb) not in groups for _, slots in records for pb, _ in slots)
    plain = bool(records) and all(op._is_plain() and pb._is_plain() for op, slots in records for pb, _ in slots)
    if out == "device" and not plain:
        raise NotImplementedError('out="device" returns raw F0 / Z0 records: no weights / reduce / phase / post on the probes')
    shared_route = out == "host" and need_raw and via != "rccl" and (same_node(group) or via == "pcie")
    shared = None
    if shared_route:
        try:
            shared = SharedResult((sp.n_adc,) + sp.enc.grid, dtype, group, rank, dst)       # (collective: fails on every rank or on none)
        except (OSError, _lib.EpgxError):
            if via == "pcie":
                raise
            shared_route = False        # ("auto": e.g. a /dev/shm too small for the result -- gather on the device instead)
    if compute is not None:
        be = _HookBackend(sp, group, rank, world, dst, compute, reduce_local)
    else:    # (the communicator is only created when something will cross it)
        be = _RcclBackend(sp, group, rank, world, dst, mode, exchange, 1 if groups else subslabs,
                          need_comm=bool(groups) or (need_raw and out == "host" and not shared_route), dtype=dtype)
    keep_local = False
    try:      # (whatever fails below, the rank's device buffers go back to the context's allocator)
        if shared_route:
            failure = None
            try:
                be.run_to_shared(shared)
            except Exception as exc:   # noqa: BLE001
                failure = exc
            if not all_agree(failure is None, group):      # (also the barrier: every column is written)
                raise failure or _lib.EpgxError("another rank could not write its slab of the result")
        else:
            be.run(gather=need_raw and out == "host")
        reduced = functions._reduce_groups(groups, be.reduce)          # (collective: every rank takes part)
        times = functions._probe_times(flat)
        if out == "device":
            nprobe = len(records[0][1]) if records else 0
            values = functions._Stacked(be.device_signals(nprobe))
            keep_local = True
            return functions._pack_values(values, times, asarray=asarray, adc_time=adc_time, stacked_as_is=True)
        if shared is not None:
            raw = shared.array if rank == dst else None
        else:
            raw = be.raw() if need_raw else None
            if raw is not None and raw.dtype != dtype:
                raw = raw.astype(dtype)
    finally:
        be.finish(keep_local=keep_local)
        if shared is not None and rank != dst:
            shared.close()
    if rank != dst:
        return None
    if variables:
        views = functions._jacobian_views(flat, records, raw, variables, sp.enc.grid)
        if views is not None:
            values = functions._Stacked(views)
        else:
            base, partials = {}, {}
            functions._collect_jacobian(records, raw, variables, base, partials)
            values, times = functions._finish_jacobian(flat, records, base, partials)
    else:
        values, times = functions._finish_records(flat, records, raw, reduced)
    return functions._pack_values(values, times, asarray=asarray, adc_time=adc_time, dtype=dtype)
