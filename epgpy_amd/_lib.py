"""ctypes binding of libepgx.so (the C ABI of include/epgx.h).

This is the only place the product touches native code.  If the library is missing, or no
MI355X-class GPU is visible, the failure is loud (`EpgxError`): there is no NumPy or other
CPU execution path behind the operators of this package.
"""
import ctypes
import os
import sys
import threading

import numpy as np

from . import _build

MAX_DIMS = 8
MAX_SPACES = 4
SUPPORTED_K = (64, 128, 256, 512, 1024)
RESIDENT_ONLY_K = 2048   # state-resident launches from equilibrium only: four wavefronts per voxel (csrc/epgx_split.hip)
PACKED_K = (16, 32)  # state-resident only: four / two voxels per wavefront (csrc/epgx_packed_kernels.hip.h)
MAX_DERIV_K = 1024  # derivative plans: the state and its derivative states of a voxel live in registers (one wavefront per voxel)


def max_vars(K):
    """derivative states one launch carries at capacity K: 3 up to 512 orders (at 512 the registers of a whole SIMD), 1 at 1024"""
    return MAX_VARS if K <= 512 else 1

OP_NOP, OP_T, OP_MAT, OP_E, OP_S, OP_ADC, OP_SPOIL, OP_RESET, OP_PD, OP_D, OP_GS, OP_MAT0, OP_T0 = range(13)
NCOEF = {OP_T: 8, OP_MAT: 10, OP_E: 4, OP_PD: 1, OP_MAT0: 14, OP_T0: 12}   # OP_D: 3*K, OP_GS: 3*K/2 (depend on the capacity)
GS_ZERO, GS_CONJ = -1, 1 << 30

c_void_pp = ctypes.POINTER(ctypes.c_void_p)


class EpgxError(RuntimeError):
    """a libepgx call failed (or the library / GPU is not available)"""


class Op(ctypes.Structure):
    _fields_ = [("opcode", ctypes.c_int32), ("space", ctypes.c_int32), ("ia", ctypes.c_int32),
                ("ib", ctypes.c_int32), ("coef_off", ctypes.c_int64), ("ncoef", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]


OP_DTYPE = np.dtype([("opcode", "<i4"), ("space", "<i4"), ("ia", "<i4"), ("ib", "<i4"),
                     ("coef_off", "<i8"), ("ncoef", "<i4"), ("reserved", "<i4")])
assert OP_DTYPE.itemsize == ctypes.sizeof(Op) == 32


class PlanDesc(ctypes.Structure):
    """epgx_plan_desc (include/epgx.h; tests/test_host.py compares the field list with the header)"""
    _fields_ = [("struct_size", ctypes.c_uint32), ("n_ops", ctypes.c_int32), ("ops", ctypes.c_void_p), ("ndim", ctypes.c_int32),
                ("grid_shape", ctypes.c_void_p), ("n_spaces", ctypes.c_int32),
                ("space_strides", ctypes.c_void_p), ("n_coef", ctypes.c_int64),
                ("coef", ctypes.c_void_p), ("n_adc", ctypes.c_int32), ("n_vars", ctypes.c_int32),
                ("dops", ctypes.c_void_p), ("deriv_flags", ctypes.c_int32), ("n_fuse", ctypes.c_int32),
                ("fuse", ctypes.c_void_p), ("n_coef_generated", ctypes.c_int64),
                ("n_assemble", ctypes.c_int32), ("n_fuse_partial", ctypes.c_int32), ("assemble", ctypes.c_void_p),
                ("fuse_partial", ctypes.c_void_p)]


MAX_VARS = 3
DERIV_THROUGH_PLAIN_OPS = 1
PLAN_NO_FOLD = 2      # keep every relaxation a stage of its own (include/epgx.h EPGX_PLAN_NO_FOLD)
FUSE_DTYPE = np.dtype([("dst_off", "<i8"), ("src_off", "<i8"), ("e_off", "<i8"), ("dst_space", "<i4"),
                       ("src_space", "<i4"), ("e_space", "<i4"), ("src_ncoef", "<i4"), ("after", "<i4"),
                       ("reserved", "<i4")])
assert FUSE_DTYPE.itemsize == 48
FUSE_PARTIAL_DTYPE = np.dtype([("dst_off", "<i8"), ("src_off", "<i8"), ("dsrc_off", "<i8"), ("e_off", "<i8"), ("de_off", "<i8"),
                               ("dst_space", "<i4"), ("src_space", "<i4"), ("dsrc_space", "<i4"), ("e_space", "<i4"),
                               ("de_space", "<i4"), ("src_ncoef", "<i4"), ("dsrc_ncoef", "<i4"), ("after", "<i4")])
assert FUSE_PARTIAL_DTYPE.itemsize == 72
MAX_ASM_SRC, MAX_ASM_COLS = 4, 16
ASM_SRC_DTYPE = np.dtype([("off", "<i8"), ("ncol", "<i4"), ("reserved", "<i4"), ("strides", "<i8", (MAX_DIMS,))])
ASSEMBLE_DTYPE = np.dtype([("dst_off", "<i8"), ("dst_space", "<i4"), ("ncoef", "<i4"), ("n_src", "<i4"), ("reserved", "<i4"),
                           ("src", ASM_SRC_DTYPE, (MAX_ASM_SRC,)), ("col_src", "u1", (MAX_ASM_COLS,)), ("col_idx", "u1", (MAX_ASM_COLS,))])
assert ASM_SRC_DTYPE.itemsize == 80 and ASSEMBLE_DTYPE.itemsize == 376
DOP_DTYPE = np.dtype([("space", "<i4", (MAX_VARS,)), ("reserved", "<i4"), ("coef_off", "<i8", (MAX_VARS,))])
assert DOP_DTYPE.itemsize == 40


class DeviceInfo(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 128), ("arch", ctypes.c_char * 64),
                ("compute_units", ctypes.c_int32), ("wavefront_size", ctypes.c_int32),
                ("clock_khz", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("hbm_bytes", ctypes.c_int64)]


# every symbol include/epgx.h declares: (restype, argtypes)
_i, _i32, _i64, _p, _d = ctypes.c_int, ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p, ctypes.c_double
SYMBOLS = {
    "epgx_abi_version": (_i, []),
    "epgx_last_error": (ctypes.c_char_p, []),
    "epgx_device_count": (_i, []),
    "epgx_ctx_create": (_i, [_i, c_void_pp]),
    "epgx_ctx_destroy": (_i, [_p]),
    "epgx_ctx_set_stream": (_i, [_p, _p]),
    "epgx_ctx_synchronize": (_i, [_p]),
    "epgx_ctx_info": (_i, [_p, ctypes.POINTER(DeviceInfo)]),
    "epgx_ctx_release_cache": (_i, [_p]),
    "epgx_malloc": (_i, [_p, _i64, c_void_pp]),
    "epgx_free": (_i, [_p, _p]),
    "epgx_memset": (_i, [_p, _p, _i, _i64]),
    "epgx_memcpy_h2d": (_i, [_p, _p, _p, _i64]),
    "epgx_memcpy_d2h": (_i, [_p, _p, _p, _i64]),
    "epgx_memcpy_d2d": (_i, [_p, _p, _p, _i64]),
    "epgx_timer_start": (_i, [_p]),
    "epgx_timer_stop": (_i, [_p, ctypes.POINTER(ctypes.c_float)]),
    "epgx_plan_create": (_i, [_p, ctypes.POINTER(PlanDesc), c_void_pp]),
    "epgx_plan_destroy": (_i, [_p]),
    "epgx_state_create": (_i, [_p, _i64, _i32, c_void_pp]),
    "epgx_state_destroy": (_i, [_p]),
    "epgx_state_upload": (_i, [_p, _p, _p]),
    "epgx_state_download": (_i, [_p, _p, _p]),
    "epgx_state_copy": (_i, [_p, _p]),
    "epgx_state_broadcast": (_i, [_p, _p, _p]),
    "epgx_state_info": (_i, [_p, ctypes.POINTER(_i64), ctypes.POINTER(_i32), c_void_pp, c_void_pp]),
    "epgx_state_axpy": (_i, [_p, _p, ctypes.c_double, _i32]),
    "epgx_run": (_i, [_p, _p, _i32, _i32, _i64, _i64, _p, _p, _i32, _p, _i64, _i64]),
    "epgx_kernel_for": (_i, [_p, _p, _i32, _i32, _i32, _p, _p, ctypes.c_char_p, _i64]),
    "epgx_signal_reduce": (_i, [_p, _p, _i64, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _p, _i64, _i64]),
    "epgx_simulate_f64": (_i, [_p, ctypes.POINTER(PlanDesc), _i32, _p, _p, _p, _p, _i32]),
    "epgx_simulate_sharded_f64": (_i, [ctypes.POINTER(PlanDesc), _i32, _i32, _p, _p, _i32]),
    "epgx_comm_unique_id": (_i, [_p]),
    "epgx_comm_create": (_i, [_p, _p, _i32, _i32, c_void_pp]),
    "epgx_comm_destroy": (_i, [_p]),
    "epgx_comm_gather": (_i, [_p, _p, _p, _i64, _i32]),
    "epgx_comm_gather_part": (_i, [_p, _p, _p, _i64, _i64, _i32]),
    "epgx_comm_join": (_i, [_p]),
    "epgx_comm_reduce": (_i, [_p, _p, _p, _i64, _i32]),
    "epgx_comm_count": (_i, [_p, ctypes.POINTER(_i32)]),
    "epgx_host_register": (_i, [_p, _p, _i64]),
    "epgx_host_unregister": (_i, [_p, _p]),
    "epgx_signal_narrow": (_i, [_p, _p, _i64, _p, _i64, _i64, _i64]),
    "epgx_memcpy_d2h_2d": (_i, [_p, _p, _i64, _p, _i64, _i64, _i64]),
    "epgx_host_alloc": (_i, [_p, _i64, _i32, c_void_pp]),
    "epgx_host_free": (_i, [_p, _p]),
    "epgx_run_to_host": (_i, [_p, _p, _i32, _i64, _i64, _p, _i64, _p, _i64, _i64, _i64, _i32]),
    "epgx_download_2d": (_i, [_p, _p, _i64, _p, _i64, _i64, _i64]),
}
ABI_VERSION = 6
COMM_ID_BYTES = 128

_lock = threading.Lock()
_cdll = None
_contexts = {}
# At interpreter exit Python tears objects down in no particular order (a context can go before the
# plans / buffers that point into it) while the HIP runtime is winding down as well: native
# destructors are skipped from then on, the process exit reclaims everything.
_exiting = False


def _mark_exit():
    global _exiting
    _exiting = True


import atexit  # noqa: E402
atexit.register(_mark_exit)


def _alive():
    return not _exiting and not sys.is_finalizing()


def library_path():
    """in-tree libepgx.so; EPGX_LIBRARY points at another build (A/B measurements of kernel versions)"""
    return os.environ.get("EPGX_LIBRARY") or _build.LIBPATH


def load():
    """dlopen libepgx.so (never builds implicitly on import paths that may lack hipcc)"""
    global _cdll
    with _lock:
        if _cdll is not None:
            return _cdll
        path = library_path()
        if not os.path.exists(path):
            raise EpgxError(
                f"libepgx.so not found at {path}: build it with `python -m epgpy_amd._build` "
                "(or __graft_entry__.build()). epgpy_amd has no CPU fallback.")
        try:
            cdll = ctypes.CDLL(path)
        except OSError as exc:
            raise EpgxError(f"cannot load {path}: {exc}") from exc
        for name, (restype, argtypes) in SYMBOLS.items():
            try:
                fn = getattr(cdll, name)
            except AttributeError as exc:
                raise EpgxError(f"{path} does not export {name}") from exc
            fn.restype, fn.argtypes = restype, argtypes
        if cdll.epgx_abi_version() != ABI_VERSION:
            raise EpgxError(f"{path}: ABI version {cdll.epgx_abi_version()}, expected {ABI_VERSION}")
        _cdll = cdll
        return cdll


def check(rc, what=""):
    if rc != 0:
        msg = load().epgx_last_error().decode(errors="replace")
        raise EpgxError(f"{what or 'libepgx'} failed ({rc}): {msg}")


class Context:
    """one device context (device id + HIP stream); objects created from it keep it alive"""

    def __init__(self, device=0):
        self.lib = load()
        handle = ctypes.c_void_p()
        check(self.lib.epgx_ctx_create(int(device), ctypes.byref(handle)), "epgx_ctx_create")
        self.handle = handle
        self.device = int(device)

    def info(self):
        inf = DeviceInfo()
        check(self.lib.epgx_ctx_info(self.handle, ctypes.byref(inf)), "epgx_ctx_info")
        return {"name": inf.name.decode(), "arch": inf.arch.decode(),
                "compute_units": inf.compute_units, "wavefront_size": inf.wavefront_size,
                "clock_khz": inf.clock_khz, "hbm_bytes": inf.hbm_bytes}

    def release_cache(self):
        """hand the context's cached (freed) device blocks back to HIP"""
        check(self.lib.epgx_ctx_release_cache(self.handle), "epgx_ctx_release_cache")

    def synchronize(self):
        check(self.lib.epgx_ctx_synchronize(self.handle), "epgx_ctx_synchronize")

    def set_stream(self, stream_ptr):
        check(self.lib.epgx_ctx_set_stream(self.handle, ctypes.c_void_p(stream_ptr or 0)),
              "epgx_ctx_set_stream")

    def timer_start(self):
        check(self.lib.epgx_timer_start(self.handle), "epgx_timer_start")

    def timer_stop(self):
        ms = ctypes.c_float()
        check(self.lib.epgx_timer_stop(self.handle, ctypes.byref(ms)), "epgx_timer_stop")
        return float(ms.value)

    def __del__(self):
        try:
            if getattr(self, "handle", None) and _alive():
                self.lib.epgx_ctx_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def default_device():
    """device index for this process: EPGX_DEVICE, else LOCAL_RANK (one process per GPU), else 0"""
    for var in ("EPGX_DEVICE", "LOCAL_RANK"):
        if os.environ.get(var, "") != "":
            return int(os.environ[var])
    return 0


def get_context(device=None):
    device = default_device() if device is None else int(device)
    with _lock:
        ctx = _contexts.get(device)
    if ctx is None:
        ctx = Context(device)
        with _lock:
            _contexts.setdefault(device, ctx)
            ctx = _contexts[device]
    return ctx


def host_empty(shape, dtype):
    """plain (pageable) result array.  Its pages are touched for the first time by the library's copy threads, a dozen
    at once, while the next block of the result crosses PCIe (epgx_run_to_host / epgx_download_2d) -- no prefault pass here"""
    return np.empty(shape, dtype=dtype)


PINNED_MAX_BYTES = 1 << 31      # results up to 2 GiB may live in page-locked blocks of the context's pool (the 1024 x 1024 three-variable Jacobian: 1.34 GB)
PINNED_MIN_BYTES = 8 << 20      # below this a plain array is as good (a 4 MB result crosses PCIe in 0.1 ms either way)
PINNED_NEW_BLOCKS = 2           # page-locked blocks OF ONE SIZE CLASS a context pins for results that are alive at the same time


class _PinnedBlock:
    """a page-locked host block of the context's pool (epgx_host_alloc); goes back to the pool when the last NumPy
    view of it dies"""
    live = {}      # context handle -> sizes of the blocks currently handed out

    def __init__(self, ctx, ptr, nbytes):
        self.ctx, self.ptr, self.nbytes = ctx, ptr, int(nbytes)
        _PinnedBlock.live.setdefault(ctx.handle.value, []).append(self.nbytes)

    def __del__(self):
        try:
            if getattr(self, "ptr", None):
                _PinnedBlock.live[self.ctx.handle.value].remove(self.nbytes)
                if _alive() and self.ctx.handle:
                    self.ctx.lib.epgx_host_free(self.ctx.handle, self.ptr)
                self.ptr = None
        except Exception:
            pass


def pinned_empty(ctx, shape, dtype):
    """ndarray whose memory is a page-locked block of the context's pool, or None.  A D2H copy into such a block needs
    no staging and no page faults, and dropping the array costs nothing (a plain 336 MB array costs the kernel ~15 ms of
    page-table work when it is freed); the block is recycled when the array (and every view of it) is gone.  Pinning
    itself is expensive (~0.2 ms per MB), so NEW blocks are only pinned while fewer than PINNED_NEW_BLOCKS of about this
    size are handed out: a loop that rebinds its result (`sig = simulate(...)`) alternates between two blocks for ever; a
    caller that keeps every result gets None after the second and takes a plain array, which the library fills through
    its staging ring at nearly the same rate (result_empty)"""
    dtype = np.dtype(dtype)
    nbytes = int(np.prod(shape)) * dtype.itemsize
    alike = sum(1 for n in _PinnedBlock.live.get(ctx.handle.value, ()) if nbytes // 2 <= n <= 2 * nbytes)
    cached_only = 1 if alike >= PINNED_NEW_BLOCKS else 0
    ptr = ctypes.c_void_p()
    check(ctx.lib.epgx_host_alloc(ctx.handle, nbytes, cached_only, ctypes.byref(ptr)), "epgx_host_alloc")
    if not ptr.value:
        return None
    block = _PinnedBlock(ctx, ptr, nbytes)
    raw = (ctypes.c_char * nbytes).from_address(ptr.value)
    raw._epgx_block = block            # the buffer exporter keeps the block alive as long as any array refers to it
    return np.frombuffer(raw, dtype=dtype).reshape(shape)


def result_empty(ctx, shape, dtype):
    """where a downloaded result goes: a recycled page-locked block of the context's pool when one is to be had (see
    pinned_empty), else a plain array.  Either way the array is an ordinary ndarray for the caller; one that lives in a
    pool block does not own its data (`arr.flags.owndata` is False, `arr.base` keeps the block alive) and the block
    returns to the pool when the last view of it is garbage-collected"""
    nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
    out = pinned_empty(ctx, shape, dtype) if PINNED_MIN_BYTES <= nbytes <= PINNED_MAX_BYTES else None
    return out if out is not None else host_empty(shape, dtype)


SIGNAL_C128, SIGNAL_C64 = 0, 1      # enum epgx_signal_dtype


def signal_code(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.complex128:
        return SIGNAL_C128
    if dtype == np.complex64:
        return SIGNAL_C64
    raise ValueError(f"signal dtype {dtype}: complex128 or complex64")


def run_to_host(ctx, plan, K, signal_ptr, out, slab=0, vox0=0, nvox=None, dev_ld=None, host_col0=None):
    """epgx_run_to_host: voxels [vox0, vox0 + nvox) of the plan, state-resident, in slabs whose signal columns are copied
    to columns host_col0 .. of `out` ([n_adc, grid voxels]-shaped, C-contiguous complex128 or complex64; page-locked or
    plain) while the next slab computes.  complex64: every slab is narrowed on the device before it leaves.  Defaults: the
    whole grid"""
    nvox = plan.nvox - vox0 if nvox is None else int(nvox)
    if out.dtype not in (np.complex128, np.complex64) or not out.flags.c_contiguous or out.size % max(plan.n_adc, 1):
        raise ValueError("run_to_host: `out` must be C-contiguous complex128 / complex64 with n_adc rows")
    host_ld = out.size // max(plan.n_adc, 1)
    host_col0 = vox0 if host_col0 is None else int(host_col0)
    check(ctx.lib.epgx_run_to_host(ctx.handle, plan.handle, int(K), int(vox0), nvox, ctypes.c_void_p(signal_ptr),
                                   int(nvox if dev_ld is None else dev_ld), ctypes.c_void_p(out.ctypes.data), host_ld, host_col0,
                                   int(slab), signal_code(out.dtype)), "epgx_run_to_host")


def signal_narrow_into(ctx, src_ptr, src_ld, dst_ptr, dst_ld, rows, cols):
    """records [rows][cols] complex128 at src_ptr (row pitch src_ld) -> complex64 at dst_ptr (row pitch dst_ld), on the device"""
    check(ctx.lib.epgx_signal_narrow(ctx.handle, ctypes.c_void_p(src_ptr), int(src_ld), ctypes.c_void_p(dst_ptr), int(dst_ld), int(rows),
                                     int(cols)), "epgx_signal_narrow")


def signal_narrow(ctx, src_ptr, src_ld, rows, cols):
    """records [rows][cols] complex128 (row pitch src_ld elements) -> a new DeviceBuffer [rows][cols] complex64, on the
    device (epgx_signal_narrow: one rounding per value, stream-ordered)"""
    dst = DeviceBuffer(ctx, 8 * max(int(rows) * int(cols), 1), itemsize=8)
    check(ctx.lib.epgx_signal_narrow(ctx.handle, ctypes.c_void_p(src_ptr), int(src_ld), dst.ptr, int(cols), int(rows), int(cols)),
          "epgx_signal_narrow")
    return dst


DOWNLOAD_ENGINE_MIN = 1 << 20    # copies from here on go through the library's download engine (epgx_download_2d)


class DeviceBuffer:
    """raw device allocation (epgx_malloc)"""

    def __init__(self, ctx, nbytes, itemsize=16):
        self.ctx, self.nbytes, self.itemsize = ctx, int(nbytes), int(itemsize)     # itemsize: bytes per record (download_2d)
        ptr = ctypes.c_void_p()
        check(ctx.lib.epgx_malloc(ctx.handle, self.nbytes, ctypes.byref(ptr)), "epgx_malloc")
        self.ptr = ptr

    def download(self, dtype, shape, out=None):
        if out is None:
            out = np.empty(shape, dtype=dtype)
        elif out.shape != tuple(shape) or out.dtype != np.dtype(dtype) or not out.flags.c_contiguous:
            raise ValueError("download: `out` does not match")
        if out.nbytes > self.nbytes:
            raise ValueError("download larger than the buffer")
        if out.nbytes >= DOWNLOAD_ENGINE_MIN:    # direct into page-locked memory, staged + host copy threads otherwise
            check(self.ctx.lib.epgx_download_2d(self.ctx.handle, out.ctypes.data, out.nbytes, self.ptr, out.nbytes, out.nbytes, 1),
                  "epgx_download_2d")
        else:
            check(self.ctx.lib.epgx_memcpy_d2h(self.ctx.handle, out.ctypes.data, self.ptr, out.nbytes),
                  "epgx_memcpy_d2h")
        return out

    def download_2d(self, out, col0, width, rows, dev_ld, offset=0):
        """rows x width records starting `offset` elements into the buffer (row pitch dev_ld elements) ->
        out[:rows, col0:col0 + width] of a C-contiguous 2-D host array of the buffer's record type -- complex128, or
        complex64 for a narrowed buffer (epgx_memcpy_d2h_2d)"""
        isz = self.itemsize
        if out.ndim != 2 or out.dtype.itemsize != isz or out.dtype.kind != "c" or not out.flags.c_contiguous:
            raise ValueError(f"download_2d: `out` must be a C-contiguous 2-D complex array of {isz}-byte records")
        if rows > out.shape[0] or col0 < 0 or col0 + width > out.shape[1] or isz * (offset + (rows - 1) * dev_ld + width) > self.nbytes:
            raise ValueError("download_2d: block out of range")
        fn = self.ctx.lib.epgx_download_2d if isz * width * rows >= DOWNLOAD_ENGINE_MIN else self.ctx.lib.epgx_memcpy_d2h_2d
        check(fn(self.ctx.handle, out.ctypes.data + isz * col0, isz * out.shape[1],
                 ctypes.c_void_p(self.ptr.value + isz * offset), isz * dev_ld, isz * width, rows), "download_2d")
        return out

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        if arr.nbytes > self.nbytes:
            raise ValueError("upload larger than the buffer")
        check(self.ctx.lib.epgx_memcpy_h2d(self.ctx.handle, self.ptr, arr.ctypes.data, arr.nbytes),
              "epgx_memcpy_h2d")

    def free(self):
        if getattr(self, "ptr", None):
            self.ctx.lib.epgx_free(self.ctx.handle, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            if _alive():
                self.free()
        except Exception:
            pass


class Comm:
    """RCCL communicator over the contexts of all ranks (epgx_comm_*): one process per GPU.  `exchange(id_bytes)`
    must hand rank 0's 128-byte id to every rank (any side channel: torch.distributed, MPI, a file)"""

    def __init__(self, ctx, rank, world_size, exchange):
        self.ctx, self.rank, self.world_size = ctx, int(rank), int(world_size)
        ident = ctypes.create_string_buffer(COMM_ID_BYTES)
        failure = None
        if self.rank == 0:
            try:
                check(ctx.lib.epgx_comm_unique_id(ident), "epgx_comm_unique_id")
            except EpgxError as exc:      # the other ranks are waiting in `exchange`: hand them an all-zero id, fail together
                failure, ident = exc, ctypes.create_string_buffer(COMM_ID_BYTES)
        raw = exchange(bytes(ident.raw))
        if len(raw) != COMM_ID_BYTES:
            raise ValueError(f"communicator id must have {COMM_ID_BYTES} bytes")
        if failure is not None or not any(raw):
            raise EpgxError(f"no communicator id from rank 0: {failure or 'rank 0 failed'}")
        handle = ctypes.c_void_p()
        check(ctx.lib.epgx_comm_create(ctx.handle, ctypes.create_string_buffer(raw, COMM_ID_BYTES), self.rank,
                                       self.world_size, ctypes.byref(handle)), "epgx_comm_create")
        self.handle = handle

    def gather(self, send_ptr, gathered_ptr, nbytes, root=0):
        """every rank's `nbytes` at send_ptr -> block `rank` of the root's buffer (stream-ordered, asynchronous)"""
        check(self.ctx.lib.epgx_comm_gather(self.handle, ctypes.c_void_p(send_ptr), ctypes.c_void_p(gathered_ptr or 0),
                                            int(nbytes), int(root)), "epgx_comm_gather")

    def gather_part(self, send_ptr, gathered_ptr, nbytes, block_stride, root=0):
        """one part of a pipelined gather: rank r's `nbytes` land at gathered_ptr + r * block_stride; runs on the
        communicator's stream behind what the context's stream holds now -- the context's stream does not wait (join)"""
        check(self.ctx.lib.epgx_comm_gather_part(self.handle, ctypes.c_void_p(send_ptr), ctypes.c_void_p(gathered_ptr or 0),
                                                 int(nbytes), int(block_stride), int(root)), "epgx_comm_gather_part")

    def join(self):
        """the context's stream waits for every transfer enqueued so far"""
        check(self.ctx.lib.epgx_comm_join(self.handle), "epgx_comm_join")

    def reduce(self, send_ptr, recv_ptr, count, root=0):
        """sum of `count` doubles over the ranks, at the root (ncclReduce); stream-ordered"""
        check(self.ctx.lib.epgx_comm_reduce(self.handle, ctypes.c_void_p(send_ptr), ctypes.c_void_p(recv_ptr or 0), int(count),
                                            int(root)), "epgx_comm_reduce")

    def count(self):
        """ranks RCCL itself counts in this communicator (ncclCommCount)"""
        n = ctypes.c_int32(0)
        check(self.ctx.lib.epgx_comm_count(self.handle, ctypes.byref(n)), "epgx_comm_count")
        return int(n.value)

    def destroy(self):
        if getattr(self, "handle", None):
            self.ctx.lib.epgx_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            if _alive() and self.ctx.handle:
                self.destroy()
        except Exception:
            pass


_COMMS = {}     # (context handle, group key) -> Comm: a communicator costs 0.1 - 1 s to create and is kept


def get_comm(ctx, rank, world_size, exchange, key=None, agree=None):
    """the communicator of (context, `key`), created on first use.  COLLECTIVE: every rank of the group must get here with
    the same key (a hashable description of the group, e.g. the tuple of its global ranks).  The cache is local to a rank
    and may differ between ranks (an earlier failure after only some had stored theirs, drop_comms() on a subset): with
    `agree(flag) -> True iff the flag is set on EVERY rank` (collective, any side channel) the ranks first settle whether
    ALL of them hold the communicator, and otherwise ALL create a fresh one -- nobody enters the id exchange alone"""
    slot = (ctx.handle.value, key, int(rank), int(world_size))
    with _lock:
        comm = _COMMS.get(slot)
    have = comm is not None and bool(comm.handle)
    if agree is not None and not agree(have):
        if have:          # some rank lost its communicator: this one is of no use any more
            with _lock:
                _COMMS.pop(slot, None)
            comm.destroy()
        have = False
    if not have:
        comm = Comm(ctx, rank, world_size, exchange)
        with _lock:
            _COMMS[slot] = comm
    return comm


def drop_comms():
    """destroy the cached communicators (collective: every rank of every cached group must call it)"""
    with _lock:
        comms = list(_COMMS.values())
        _COMMS.clear()
    for comm in comms:
        comm.destroy()


def plan_desc(ops, grid_shape, space_strides, coef, n_adc, dops=None, n_vars=0, deriv_flags=0,
              fuse=None, n_coef_generated=0, assemble=None, fuse_partial=None):
    """(epgx_plan_desc, the arrays it points into -- keep them alive as long as the struct is used) from the host arrays of
    plan.Encoder.plan_arrays"""
    ops = np.ascontiguousarray(ops, dtype=OP_DTYPE)
    if dops is not None:
        dops = np.ascontiguousarray(dops, dtype=DOP_DTYPE)
    if fuse is not None:
        fuse = np.ascontiguousarray(fuse, dtype=FUSE_DTYPE)
    if assemble is not None:
        assemble = np.ascontiguousarray(assemble, dtype=ASSEMBLE_DTYPE)
    if fuse_partial is not None:
        fuse_partial = np.ascontiguousarray(fuse_partial, dtype=FUSE_PARTIAL_DTYPE)
    grid = np.ascontiguousarray(grid_shape, dtype=np.int64)
    strides = np.zeros((max(len(space_strides), 1), MAX_DIMS), dtype=np.int64)
    for s, st in enumerate(space_strides):
        strides[s, : len(st)] = st
    coef = np.ascontiguousarray(coef, dtype=np.float64)
    desc = PlanDesc(ctypes.sizeof(PlanDesc), len(ops), ops.ctypes.data, len(grid), grid.ctypes.data, len(space_strides),
                    strides.ctypes.data, coef.size, coef.ctypes.data if coef.size else None,
                    int(n_adc), int(n_vars), dops.ctypes.data if dops is not None else None,
                    int(deriv_flags), 0 if fuse is None else len(fuse),
                    None if fuse is None or not len(fuse) else fuse.ctypes.data, int(n_coef_generated),
                    0 if assemble is None else len(assemble), 0 if fuse_partial is None else len(fuse_partial),
                    None if assemble is None or not len(assemble) else assemble.ctypes.data,
                    None if fuse_partial is None or not len(fuse_partial) else fuse_partial.ctypes.data)
    return desc, (ops, dops, fuse, assemble, fuse_partial, grid, strides, coef)


class DevicePlan:
    """epgx_plan handle built from host arrays (see plan.py)"""

    def __init__(self, ctx, ops, grid_shape, space_strides, coef, n_adc, **more):
        self.ctx = ctx
        desc, keep = plan_desc(ops, grid_shape, space_strides, coef, n_adc, **more)
        ops, grid = keep[0], keep[5]
        handle = ctypes.c_void_p()
        check(ctx.lib.epgx_plan_create(ctx.handle, ctypes.byref(desc), ctypes.byref(handle)),
              "epgx_plan_create")
        self.handle = handle
        self.n_ops, self.n_adc = len(ops), int(n_adc)
        self.nvox = int(np.prod(grid))

    def __del__(self):
        try:
            if getattr(self, "handle", None) and _alive() and self.ctx.handle:
                self.ctx.lib.epgx_plan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class DeviceState:
    """epgx_state handle: [nvox][3][K] complex128 + density[nvox] in HBM"""

    def __init__(self, ctx, nvox, K):
        self.ctx, self.nvox, self.K = ctx, int(nvox), int(K)
        handle = ctypes.c_void_p()
        check(ctx.lib.epgx_state_create(ctx.handle, self.nvox, self.K, ctypes.byref(handle)),
              "epgx_state_create")
        self.handle = handle

    def upload(self, half, density=None):
        half = np.ascontiguousarray(half, dtype=np.complex128)
        if half.shape != (self.nvox, 3, self.K):
            raise ValueError(f"expected half-state of shape {(self.nvox, 3, self.K)}, got {half.shape}")
        dptr = None
        if density is not None:
            density = np.ascontiguousarray(density, dtype=np.float64)
            if density.shape != (self.nvox,):
                raise ValueError("density must have one value per voxel")
            dptr = density.ctypes.data
        check(self.ctx.lib.epgx_state_upload(self.handle, half.ctypes.data, dptr), "epgx_state_upload")

    def download(self):
        half = np.empty((self.nvox, 3, self.K), dtype=np.complex128)
        dens = np.empty(self.nvox, dtype=np.float64)
        check(self.ctx.lib.epgx_state_download(self.handle, half.ctypes.data, dens.ctypes.data),
              "epgx_state_download")
        return half, dens

    def copy(self, K=None):
        new = DeviceState(self.ctx, self.nvox, K or self.K)
        check(self.ctx.lib.epgx_state_copy(new.handle, self.handle), "epgx_state_copy")
        return new

    def axpy(self, src, alpha=1.0, zero_density=False):
        """self += alpha * src on the device; zero_density: self becomes a derivative state"""
        check(self.ctx.lib.epgx_state_axpy(self.handle, src.handle, float(alpha), 1 if zero_density else 0),
              "epgx_state_axpy")

    def zero_density(self):
        """no equilibrium term any more: the state becomes a derivative state"""
        _, dens = self.pointers()
        check(self.ctx.lib.epgx_memset(self.ctx.handle, ctypes.c_void_p(dens), 0, 8 * self.nvox), "epgx_memset")

    def broadcast(self, src_index):
        src_index = np.ascontiguousarray(src_index, dtype=np.int32)
        new = DeviceState(self.ctx, src_index.size, self.K)
        check(self.ctx.lib.epgx_state_broadcast(new.handle, self.handle, src_index.ctypes.data),
              "epgx_state_broadcast")
        return new

    def pointers(self):
        nvox, K = ctypes.c_int64(), ctypes.c_int32()
        data, dens = ctypes.c_void_p(), ctypes.c_void_p()
        check(self.ctx.lib.epgx_state_info(self.handle, ctypes.byref(nvox), ctypes.byref(K),
                                           ctypes.byref(data), ctypes.byref(dens)), "epgx_state_info")
        return data.value, dens.value

    def __del__(self):
        try:
            if getattr(self, "handle", None) and _alive() and self.ctx.handle:
                self.ctx.lib.epgx_state_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def signal_reduce(ctx, signal_ptr, signal_ld, row0, row_step, n_rows, grid, reduce_mask, weights=None, vox0=0, nvox=None,
                  to_host=True):
    """device-side  sum over the masked grid axes of  weights * signal[row]  for n_rows rows
    (epgx_signal_reduce); weights: array broadcastable to `grid` (leading-axis aligned) or None.
    The buffer may hold the voxel slab [vox0, vox0 + nvox) of the grid only (the other voxels count as zero: the partial
    sums of a multi-GPU run).  Returns complex128 [n_rows, *kept axes] -- on the host, or with to_host=False the
    DeviceBuffer that holds it"""
    grid = tuple(int(g) for g in grid)
    shape = np.ascontiguousarray(grid, dtype=np.int64)
    mask = np.ascontiguousarray(reduce_mask, dtype=np.uint8)
    kept = tuple(g for g, m in zip(grid, mask) if not m)
    n_out = int(np.prod(kept)) if kept else 1
    wbuf, wstrides = None, None
    if weights is not None:
        w = np.asarray(weights, dtype=np.complex128)
        w = np.ascontiguousarray(w.reshape(w.shape + (1,) * (len(grid) - w.ndim)))
        wstrides = np.ascontiguousarray([0 if w.shape[d] == 1 else w.strides[d] // 16 for d in range(len(grid))],
                                        dtype=np.int64)
        wbuf = DeviceBuffer(ctx, max(w.nbytes, 16))
        wbuf.upload(w)
    out = DeviceBuffer(ctx, 16 * max(n_rows * n_out, 1))
    check(ctx.lib.epgx_signal_reduce(ctx.handle, ctypes.c_void_p(signal_ptr), int(signal_ld), int(row0), int(row_step),
                                     int(n_rows), len(grid), shape.ctypes.data, mask.ctypes.data,
                                     wbuf.ptr if wbuf is not None else None,
                                     wstrides.ctypes.data if wstrides is not None else None, out.ptr, int(vox0),
                                     int(np.prod(grid)) - int(vox0) if nvox is None else int(nvox)),
          "epgx_signal_reduce")
    if wbuf is not None:
        wbuf.free()        # (stream-ordered: the block is recycled behind the kernel that reads it)
    if not to_host:
        out.shape = (int(n_rows),) + kept
        return out
    res = out.download(np.complex128, (int(n_rows),) + kept)
    out.free()
    return res


def kernel_for(ctx, plan, K, op_begin=0, op_end=None, state_in=None, state_out=None):
    """name (with template arguments) of the kernel epgx_run would launch for this range of the plan at capacity K"""
    buf = ctypes.create_string_buffer(160)
    check(ctx.lib.epgx_kernel_for(ctx.handle, plan.handle, int(op_begin), int(plan.n_ops if op_end is None else op_end), int(K),
                                  state_in.handle if state_in is not None else None,
                                  state_out.handle if state_out is not None else None, buf, len(buf)), "epgx_kernel_for")
    return buf.value.decode()


def run(ctx, plan, op_begin, op_end, vox0, nvox, state_in, state_out, K, signal_ptr, signal_ld,
        signal_col0):
    check(ctx.lib.epgx_run(ctx.handle, plan.handle, int(op_begin), int(op_end), int(vox0), int(nvox),
                           state_in.handle if state_in is not None else None,
                           state_out.handle if state_out is not None else None, int(K),
                           ctypes.c_void_p(signal_ptr) if signal_ptr else None, int(signal_ld),
                           int(signal_col0)), "epgx_run")
