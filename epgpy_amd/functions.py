"""simulate() and sequence helpers (mirrors epgpy/functions.py:14-192, :355-369).

`simulate` keeps the reference signature and return conventions, but instead of looping
over operators in Python it compiles the flattened sequence into one plan and runs it in a
single launch of the fused HIP kernel with every voxel's state resident in registers
("state-resident" mode).  `mode="stream"` runs the same plan one ADC-to-ADC segment per
launch with the state streamed through HBM (the per-timestep mode whose HBM roofline
BASELINE.md quotes).  Both modes run the same arithmetic chains and give identical bits -- with ONE exception:
at 64 orders per voxel the state-resident kernels evaluate rotations about x in a sum / difference form (16 instead
of 18 instructions per order: the same products in another association order), so there a state-resident result and
the per-timestep result of the same plan may differ in the last bits (<= 1e-13 on O(1) signals; against the oracle
both stay within 1e-12).
Sequences with a `callback`, or with probes the kernel cannot record itself, run
segment-wise and evaluate the probes on a host view of the device state.
"""
import os

import numpy as np

from . import common, operator as _operator, probe as _probe, statematrix, plan as _plan, shift as _shift, _lib

LOGGER = common.LOGGER
Probe = _probe.Probe


def flatten_sequence(seq, flatten_multi=True):
    """flat list of operators from nested lists / MultiOperators (functions.py:355-369)"""
    seq = [seq] if isinstance(seq, _operator.Operator) else seq
    flat = []
    for item in seq:
        if isinstance(item, list):
            flat.extend(flatten_sequence(item))
        elif flatten_multi and isinstance(item, _operator.MultiOperator):
            flat.extend(flatten_sequence(item.operators))
        elif isinstance(item, _operator.Operator):
            flat.append(item)
        else:
            raise ValueError(f"Invalid operator: {item}")
    return flat


def getshape(sequence):
    """overall parameter-grid shape of a sequence (functions.py:14-17)"""
    sequence = flatten_sequence(sequence)
    return common.broadcast_shapes(*[op.shape for op in sequence], append=True)


def getnshift(sequence):
    """total number of (absolute) phase-state shifts (functions.py:20-26)"""
    return sum(op.nshift for op in flatten_sequence(sequence))


def getkdim(sequence):
    return max([getattr(op, "kdim", 1) for op in flatten_sequence(sequence)] + [1])


def get_adc_times(sequence):
    """ADC opening times from the operators' durations (functions.py:38-47)"""
    tim, times = 0, []
    for op in flatten_sequence(sequence):
        tim = tim + op.duration
        if isinstance(op, Probe):
            times.append(tim)
    return times


def modify(sequence, modifier=None, *, expand=True, **params):
    """attach tissue / system parameters to a timing-only sequence (functions.py:251-313):
    every operator with a non-zero duration gets an E (or P) of that duration appended, T
    operators are scaled by `att`; non-scalar parameters become new grid axes when `expand`"""
    shape = getshape(sequence)
    values = common.expand_arrays(*params.values(), append=True)
    if expand and (len(shape) > 1 or shape[0] > 1):
        dims = tuple(range(len(shape)))
        values = common.map_arrays(values, lambda arr: np.expand_dims(arr, dims))
    params = dict(zip(params, values))
    if not modifier:
        modifier = default_modifier
        if not params:
            return sequence
    elif not callable(modifier):
        raise TypeError("`modifier` must be a callable")
    newseq, done = [], {}
    for op in flatten_sequence(sequence):
        if op not in done:
            done[op] = modifier(op, **params)
        newseq.append(done[op])
    LOGGER.info(f"Modify sequence: {shape}->{getshape(newseq)}")
    if isinstance(sequence, _operator.MultiOperator):
        return _operator.MultiOperator(newseq, name=sequence.name)
    return newseq


def default_modifier(op, **kwargs):
    """handles 'T1', 'T2', 'g' and 'att' (B1 attenuation)  (functions.py:316-347)"""
    from . import transition, evolution

    if isinstance(op, transition.T):
        att = kwargs.get("att")
        if att is not None and not np.allclose(att, 1):
            op = transition.T(op.alpha * att, op.phi, name=op.name, duration=op.duration)
            op.name += "#"
    if np.any(op.duration > 0):
        T1, T2, g = kwargs.get("T1"), kwargs.get("T2"), kwargs.get("g")
        if T1 is None and T2 is None and g is None:
            pass
        elif T1 is None and T2 is None:
            op = op * evolution.P(op.duration, g, duration=0)
            op.name = op[0].name + "*"
        else:
            T1 = 1e10 if T1 is None else T1
            T2 = 1e10 if T2 is None else T2
            g = 0 if g is None else g
            op = op * evolution.E(op.duration, T1, T2, g, duration=0)
            op.name = op[0].name + "*"
    return op


def squeeze_sequence(seq):
    """combine operators where that is exact and cheap: runs  E . T . E  of precession-free relaxations around a
    rotation collapse into single operators (fusion.py).  The reference declares this function and raises
    NotImplementedError (functions.py:350-352); `simulate` applies the same pass by default (`fuse=True`), within the
    same budget for device-generated tables (past it the sequence is returned as it is: the library folds the
    relaxations into the rotations at run time instead)."""
    from . import fusion

    flat = flatten_sequence(seq)
    if not fusion.fusable(flat):
        return flat
    fused = fusion.fuse_sequence(flat)
    return fused if fusion.generated_bytes(fused, 0) <= FUSED_TABLE_BUDGET else flat


def signal_dtype(dtype):
    """the dtype of returned records: complex128 (None, the reference's: epgpy/statematrix.py:392) or complex64"""
    dtype = np.dtype(np.complex128 if dtype is None else dtype)
    if dtype not in (np.dtype(np.complex128), np.dtype(np.complex64)):
        raise ValueError(f"dtype={dtype}: records are complex128 (default) or complex64")
    return dtype


def _segments(sequence):
    """[begin, end) operator ranges ending right after each probe (+ the tail)"""
    out, begin = [], 0
    for i, op in enumerate(sequence):
        if isinstance(op, Probe):
            out.append((begin, i + 1))
            begin = i + 1
    if begin < len(sequence):
        out.append((begin, len(sequence)))
    return out


FUSED_TABLE_BUDGET = float(os.environ.get("EPGX_FUSED_TABLE_BUDGET", 2.0e9))     # bytes of device-generated fused tables per plan (compile_sequence)
FUSE_DERIVATIVES = {64: 3, 32: 2}     # orders per voxel -> most variables per plan for which differentiated E . T . E runs are fused


def _fusion_pays(sequence, variables, nstate0, options, from_state=False):
    """Plain plans: always.  Differentiated plans: a fused record reads its table entry AND one partial entry per
    variable (96 + 112 V bytes per voxel and echo).  The four-voxels-per-wavefront kernels fetch those as prefetched
    lines and have straight-line bodies for them.  At 64 orders an echo train of fused records runs on rotating order
    slots with up to three derivative states (drun_kernel, csrc/epgx_drun_kernels.hip.h; relaxation-only partials in
    logarithmic form): 20-echo 1024 x 1024 train 2.6 / 3.9 / 5.6 ms with 1 / 2 / 3 variables against 4.3 / 6.9 / 10.2 ms
    three-stage.  (Trains that cannot be fused here -- a rotation over one grid axis between relaxations over others -- are
    folded by the library at run time instead: DESIGN.md 4.3.)  packed_deriv_kernel (up to 32
    orders): one / two variables 3.1 -> 2.7, 4.8 -> 4.2 ms (16 orders: 1.56 -> 1.35, 2.36 -> 2.09 ms); with three
    variables it is at its register budget (the fused records take its flag-tested body, 6.6 -> 8.1 ms): those plans keep
    their three stages.  A run from a given state matrix takes deriv_kernel, whose dependent scalar loads of per-voxel
    entries lose more than the shorter arithmetic gains: three stages as well."""
    if not variables:
        return True
    if from_state:                   # (a run from a given state matrix takes deriv_kernel)
        return False
    if any(isinstance(op, _shift.S) and (not isinstance(op.k, int) or abs(op.k) != 1) for op in sequence):
        return False                 # (neither of the two kernel families takes the plan then)
    peak = nstate0 + int(getnshift(sequence))
    cap = (options or {}).get("max_nstate")
    orders = (min(peak, int(cap)) if cap else peak) + 1
    fits = [k for k in FUSE_DERIVATIVES if orders <= k]       # (the entry of the capacity class the plan runs at)
    return bool(fits) and len(variables) <= FUSE_DERIVATIVES[min(fits)]


def compile_sequence(sequence, probes=None, *, shape=None, options=None, nstate0=0, kspace0=None,
                     dense_start=False, variables=(), fuse=True):
    """flatten + encode; returns (encoder, records) with records = [(op, [(probe, slot)...])]

    variables: names of the (at most 3) order1 variables whose derivative states the plan
    propagates; every probe then owns 1 + len(variables) consecutive signal rows from `slot`"""
    sequence = flatten_sequence(sequence)
    grid = getshape(sequence)
    if shape is not None:
        grid = common.broadcast_shapes(grid, tuple(shape), append=True)
    if kspace0 is None and any(isinstance(op, _shift.S) and not isinstance(op.k, int) for op in sequence):
        # the sequence uses integer n-D shifts: plan the k-space coordinate set from the start
        from . import kspace
        kdim = getkdim(sequence)
        kspace0 = (kspace.KSpace.from_orders(nstate0, kdim) if (nstate0 > 0 or dense_start)
                   else kspace.KSpace.equilibrium(kdim))
    enc = _plan.Encoder(grid, options=options, nstate0=nstate0, kspace0=kspace0)
    enc.variables = list(variables)
    if not fuse:     # operator-by-operator arithmetic: no host-side E.T.E tables, no run-time fold in the library either
        enc.deriv_flags |= _lib.PLAN_NO_FOLD
    records, bounds = [], []
    if fuse and kspace0 is None and _fusion_pays(sequence, variables, nstate0, options, dense_start):
        from . import fusion
        if fusion.fusable(sequence):                    # probes keep their place: records / bounds are unaffected
            fused = fusion.fuse_sequence(sequence, variables=list(variables) if variables else None)
            # every fused operator is a table of 12 (+ 14 per variable) doubles per entry that the library generates in its
            # coefficient pool (32-bit byte offsets: 4 GB; the four-voxels-per-wavefront kernels reach 2 GB).  Past the budget
            # the sequence stays as it is: the library then folds the relaxations into the rotations at run time (no
            # tables: include/epgx.h EPGX_PLAN_NO_FOLD), or runs them as stages
            if fusion.generated_bytes(fused, len(variables)) <= FUSED_TABLE_BUDGET:
                sequence = fused
    for op in sequence:
        if isinstance(op, Probe):
            slots = []
            for pb in (probes or [op]):
                kind = (pb or op)._device_kind()
                if kind is None:
                    raise NotImplementedError(f"probe {pb or op!r} cannot be recorded on the device")
                slots.append((pb or op, enc.add_adc(kind)))
            records.append((op, slots))
            bounds.append(len(enc.records))
        else:
            op._encode(enc)
    return enc, records, bounds


def simulate(sequence, *, adc_time=False, init=None, squeeze=False, probe=None, callback=None,
             asarray=True, disp=False, device=None, ngpu=None, mode="auto", exact_partials=False, fuse=True, packed=True,
             out="host", dtype=None, **options):
    """simulate a sequence; values are returned for every Probe/ADC (functions.py:50-170)

    Extra keywords (not in the reference): `device` (GPU index, or a list of indices); `ngpu=N`: cut the parameter grid
    into N contiguous voxel slabs and simulate them on GPUs 0 .. N-1 of this process at the same time (results identical
    to one GPU, bit for bit; sums of `Adc(reduce=)` to rounding) -- every GPU downloads its slab over its own PCIe link
    into its columns of ONE result array (state-resident mode from equilibrium; the one-process-per-GPU form is
    `epgpy_amd.distributed.simulate_sharded`); `mode` in {"auto", "resident", "stream", "stepwise"};
    `exact_partials`: with Jacobian probes, let SPOILER / RESET / PD / D act on the derivative states too.  The
    reference applies them to the state only (they are plain Operators, operator.py:95-104), so its Jacobian after e.g.
    a spoiler is not the derivative of the spoiled signal; the default reproduces the reference's numbers.
    `fuse`: collapse E . T . E runs into single operators (fusion.py; rounding-level differences); `squeeze=True` (the
    reference's keyword, functions.py:350-352) asks for the same pass.
    `packed`: state matrices of at most 16 / 32 orders run with one / two orders per lane, four voxels per wavefront, at that capacity (the bits of the
    per-timestep kernels; a 64-order launch of the same plan may differ in the last bits on rotations about x, see above).
    `out`: "host" (NumPy arrays, as the reference) or "device": the records of every probe stay in HBM and a
    `DeviceSignal` handle is returned per probe (plain F0 / Z0 probes, device modes only; with `ngpu` a
    `ShardedDeviceSignal` whose `.parts` are the per-GPU handles) -- for consumers that reduce or match the dictionary
    on the GPU and never need the 16 GB of a 10^6-voxel MRF signal on the host.  A Jacobian probe (at most three
    variables, one GPU) leaves a `DeviceJacobian`: per ADC the probed state and its derivative rows, `.column(var)` a
    `DeviceSignal` on one of them.

    `dtype`: np.complex64 returns single-precision records: the simulation itself stays float64 (complex128 states, as the
    reference), every record is rounded ONCE when it leaves the device (<= 6e-8 relative), and half the bytes cross PCIe --
    which is what a caller of a large grid waits for.  Default: complex128, the reference's.

    Result arrays.  Large results are views of page-locked blocks that the device context recycles (`arr.flags.owndata`
    is False): they behave like any ndarray and stay valid as long as they, or any view of them, are referenced; the
    block returns to the pool when the last reference is dropped.  A caller that holds more than two large results at a
    time receives ordinary arrays for the further ones (filled by the library's copy threads at nearly the same rate).
    """
    sequence = flatten_sequence(sequence)
    nshift, shape = getnshift(sequence), getshape(sequence)
    if options.get("shape") is not None:      # (extension: a grid larger than the operators span -- the reference's simulate passes its own
        options = dict(options)               #  `shape` to the state matrix and would reject the keyword)
        shape = tuple(common.broadcast_shapes(shape, tuple(options.pop("shape")), append=True))
    LOGGER.info(f"Simulate sequence: num. operators: {len(sequence)}, num. shifts: {nshift}, shape: {shape}")
    if squeeze:     # the fusion pass of compile_sequence IS the squeeze (with its table budget; derivative plans as _fusion_pays decides)
        fuse = True
    if not any(isinstance(op, Probe) for op in sequence):
        raise ValueError("Cannot simulate sequence without at least one Probe/ADC operator")

    probes = []
    if probe:
        probes = probe if isinstance(probe, (tuple, list)) else [probe]
        probes = [pb if isinstance(pb, (Probe, type(None))) else Probe(pb) for pb in probes]

    dtype = signal_dtype(dtype)
    if dtype != np.complex128 and out == "device":
        raise NotImplementedError('out="device" keeps complex128 records')
    devices = _device_list(device, ngpu)
    device = devices[0]
    if init is None and options.get("equilibrium") is not None:     # (an equilibrium other than [0, 0, 1]: the start state)
        init = statematrix.StateMatrix(None, shape=shape, device=device, **options)
        options = {k: v for k, v in options.items() if k != "equilibrium"}
    if init is not None and not isinstance(init, statematrix.StateMatrix):
        init = statematrix.StateMatrix(init, shape=shape, device=device, **options)
    elif init is not None:
        if not common.broadcastable(init.shape, shape, append=True):
            raise ValueError(f"Incompatible StateMatrix and operator shapes: {init.shape}, {shape}")
        options = {**init.options, **options}

    on_device = all((pb or op)._device_kind() is not None
                    for op in sequence if isinstance(op, Probe) for pb in (probes or [op]))
    on_device = on_device and not any(part._on_host() for op in sequence for part in op._parts())   # (user-written operators)
    if init is not None and getattr(init, "_eq", None) is not None:
        # a general equilibrium (a second state matrix per voxel): operator by operator (plan._apply_with_equilibrium)
        if mode not in ("auto", "stepwise"):
            raise NotImplementedError(f"mode={mode!r} with a general equilibrium: only the operator-by-operator path carries one")
        mode = "stepwise"
    if mode == "auto":
        mode = "resident" if (on_device and not callback) else "stepwise"
    if mode in ("resident", "stream") and (callback or not on_device):
        raise ValueError(f"mode={mode!r} needs device-recordable probes (F0/Z0), library operators only and no callback")
    if out not in ("host", "device"):
        raise ValueError(f'out={out!r}: expected "host" or "device"')
    if out == "device" and mode == "stepwise":
        raise ValueError('out="device" needs device-recordable probes (F0/Z0) and no callback')
    if len(devices) > 1 and mode != "resident":
        raise NotImplementedError("ngpu > 1 runs state-resident (no mode='stream' / 'stepwise', no callback)")

    progress = None
    if disp:
        from . import utils
        progress = utils.Progress(sum(isinstance(op, Probe) for op in sequence) if mode == "stream" else
                                  (len(sequence) if mode == "stepwise" else 1))
    if mode == "stepwise":
        values, times = _simulate_stepwise(sequence, probes, init, shape, callback, device, options, progress)
    else:
        values, times = _simulate_device(sequence, probes, init, mode, devices, options, exact_partials, fuse, packed, progress,
                                         to_host=(out != "device"), dtype=dtype, shape=shape)
    if progress is not None:
        progress.close()
    return _pack_values(values, times, asarray=asarray, adc_time=adc_time, stacked_as_is=(out == "device"), dtype=dtype)


def _pack_values(values, times, *, asarray=True, adc_time=False, stacked_as_is=False, dtype=None):
    """the return conventions of the reference's simulate (functions.py:157-170): per probe a stacked array (or a tuple of
    records), a single probe unwrapped from its 1-tuple, optionally preceded by the ADC times.  `dtype` complex64: records
    that were finished on the host (weights, phases, callable probes: complex128 arithmetic) are rounded here; records
    that left the device as complex64 pass through"""
    narrow = dtype is not None and np.dtype(dtype) == np.complex64
    cast = (lambda arr: arr.astype(np.complex64, copy=False) if isinstance(arr, np.ndarray) and arr.dtype == np.complex128 else arr)
    if isinstance(values, _Stacked):
        values = tuple(values) if (asarray or stacked_as_is) else tuple(tuple(arr) for arr in values)
        if narrow and asarray and not stacked_as_is:
            values = tuple(cast(arr) for arr in values)
    else:
        values = tuple(zip(*values))
        if asarray:
            values = tuple(np.asarray(arr) for arr in values)
            if narrow:
                values = tuple(cast(arr) for arr in values)
        elif narrow:
            values = tuple(tuple(cast(np.asarray(rec)) for rec in arr) for arr in values)
    if asarray:
        times = np.asarray(times)
    if len(values) == 1:
        values = values[0]
    if adc_time:
        return times, values
    return values


def _device_list(device, ngpu):
    """GPU indices of a simulate() call: `device` (an index, a list of indices, or None = the process default) and `ngpu`
    (GPUs 0 .. ngpu-1, or the first ngpu entries counted from `device`)"""
    if isinstance(device, (list, tuple)):
        devices = [int(d) for d in device]
        if ngpu is not None and int(ngpu) != len(devices):
            raise ValueError(f"ngpu={ngpu} but device={device}")
    else:
        first = _lib.default_device() if device is None else int(device)
        n = 1 if ngpu is None else int(ngpu)
        if n < 1:
            raise ValueError(f"ngpu={ngpu}: expected a positive number of GPUs")
        devices = [first + g for g in range(n)]
    if len(set(devices)) != len(devices):
        raise ValueError(f"device={device}: every GPU once")
    return devices


def _jacobian_variables(sequence, probes):
    """order1 variables that a Jacobian probe asks for AND some operator differentiates against,
    in first-use order"""
    wanted = []
    for op in sequence:
        if isinstance(op, Probe):
            for pb in (probes or [op]):
                for var in getattr(pb or op, "_device_variables", list)():
                    if var not in wanted:
                        wanted.append(var)
    known = {var for op in sequence for var in (getattr(op, "order1", None) or {})}
    return [var for var in wanted if var in known]


def slab_bounds(nvox, parts):
    """equal contiguous voxel slabs (the last ones ragged or empty): (slab, [(vox0, count), ...])"""
    slab = -(-int(nvox) // int(parts))
    out = []
    for r in range(int(parts)):
        v0 = min(r * slab, int(nvox))
        out.append((v0, min((r + 1) * slab, int(nvox)) - v0))
    return slab, out


def _probe_times(sequence):
    times, tic = [], 0
    for op in sequence:
        tic = tic + op.duration
        if isinstance(op, Probe):
            times.append(tic)
    return times


class _Fleet:
    """the GPUs of one simulate() call: a context, the plan and a signal buffer [n_adc][slab voxels] per device, each
    with its contiguous voxel slab of the grid.  One device: the whole grid (the same code path)"""

    def __init__(self, enc, K, devices, ctx0=None):
        self.enc, self.n = enc, len(devices)
        self.ctxs = [ctx0 if (g == 0 and ctx0 is not None) else _lib.get_context(d) for g, d in enumerate(devices)]
        self.slab, self.bounds = slab_bounds(enc.nvox, self.n)
        # the host arrays are built ONCE, here on the calling thread (building mutates the encoder: deferred tables join the
        # pool); the per-device threads only upload them (side by side)
        arrays = enc.plan_arrays(K)
        self.plans = self.each(lambda g: _lib.DevicePlan(self.ctxs[g], **arrays))
        self.sigs = [_lib.DeviceBuffer(ctx, 16 * max(enc.n_adc, 1) * max(cnt, 1)) for ctx, (_, cnt) in zip(self.ctxs, self.bounds)]

    def each(self, fn):
        """fn(g) for every device -- from one host thread per device when there are several (the library releases the GIL)"""
        if self.n == 1:
            return [fn(0)]
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(self.n) as pool:
            return list(pool.map(fn, range(self.n)))

    def slabs_of(self, state):
        """a start state [nvox][3][K] that lives on the first device, cut into per-device slab states (one device: the state
        itself).  The other devices' slabs travel through host memory -- a copy, once per call: the reference's `init=` is a
        host array to begin with (functions.py:149)"""
        if state is None or self.n == 1:
            return [state] * self.n
        half, dens = state.download()
        out = []
        for g, (v0, cnt) in enumerate(self.bounds):
            if not cnt:
                out.append(None)
                continue
            slab = _lib.DeviceState(self.ctxs[g], cnt, state.K)
            slab.upload(np.ascontiguousarray(half[v0:v0 + cnt]), np.ascontiguousarray(dens[v0:v0 + cnt]))
            out.append(slab)
        return out

    def run(self, K_run, state_in=None):
        """one state-resident launch per device (asynchronous); `state_in`: the start state of the whole grid (on the first device)"""
        slabs = self.slabs_of(state_in)
        for g, (v0, cnt) in enumerate(self.bounds):
            if cnt:
                _lib.run(self.ctxs[g], self.plans[g], 0, self.plans[g].n_ops, v0, cnt, slabs[g], None, K_run,
                         self.sigs[g].ptr.value, cnt, 0)

    def run_to_host(self, K_run, out):
        """every device: its slab in sub-slabs whose rows leave over ITS PCIe link while the next sub-slab computes, into
        its columns of `out` [n_adc, *grid]"""
        def drive(g):
            v0, cnt = self.bounds[g]
            if cnt:
                _lib.run_to_host(self.ctxs[g], self.plans[g], K_run, self.sigs[g].ptr.value, out, vox0=v0, nvox=cnt)
        self.each(drive)

    def download(self, out):
        """the signal buffers -> their columns of `out` [n_adc, *grid] (after run); a complex64 `out` receives records that
        were narrowed on the device (half the bytes over PCIe)"""
        flat = out.reshape(self.enc.n_adc, self.enc.nvox)

        def fetch(g):
            v0, cnt = self.bounds[g]
            if not cnt:
                return
            if out.dtype == np.complex64:
                small = _lib.signal_narrow(self.ctxs[g], self.sigs[g].ptr.value, cnt, self.enc.n_adc, cnt)
                small.download_2d(flat, v0, cnt, self.enc.n_adc, cnt)
                small.free()
            else:
                self.sigs[g].download_2d(flat, v0, cnt, self.enc.n_adc, cnt)
        self.each(fetch)
        return out

    def reduce(self, mask, weights, row0, step, count):
        """sum over the masked grid axes of rows row0, row0 + step, ...: every device sums over its own voxels
        (epgx_signal_reduce on its slab); the partial sums of several devices are added up on the first one"""
        grid = self.enc.grid
        if self.n == 1:
            return _lib.signal_reduce(self.ctxs[0], self.sigs[0].ptr.value, self.bounds[0][1], row0, step, count, grid, mask, weights)
        parts = self.each(lambda g: None if not self.bounds[g][1] else _lib.signal_reduce(
            self.ctxs[g], self.sigs[g].ptr.value, self.bounds[g][1], row0, step, count, grid, mask, weights,
            vox0=self.bounds[g][0], nvox=self.bounds[g][1]))
        parts = [part for part in parts if part is not None]
        kept = parts[0].shape[1:]
        stack = _lib.DeviceBuffer(self.ctxs[0], 16 * len(parts) * parts[0].size)
        stack.upload(np.stack(parts))
        total = _lib.signal_reduce(self.ctxs[0], stack.ptr.value, len(parts) * parts[0].size, 0, 1, 1,
                                   (len(parts), parts[0].size), [1, 0])
        stack.free()
        return total.reshape((count,) + kept)

    def free(self):
        for sig in self.sigs:
            sig.free()


def _reduction_groups(records, grid):
    """probes whose weighted sums over grid axes run on the device: id(probe) -> ((mask, weights), [(i, j, slot), ...])"""
    groups = {}
    for i, (_, slots) in enumerate(records):
        for j, (pb, slot) in enumerate(slots):
            spec = pb._device_reduction(grid) if hasattr(pb, "_device_reduction") else None
            if spec is not None:
                groups.setdefault(id(pb), (spec, []))[1].append((i, j, slot))
    return groups


def _group_runs(members):
    """[(first row, row step, number of rows, members)]: an echo train's records are equidistant rows -- one call"""
    rows = [slot for _, _, slot in members]
    steps = {b - a for a, b in zip(rows, rows[1:])}
    if len(steps) <= 1 and (not steps or min(steps) > 0):
        return [(rows[0], steps.pop() if steps else 1, len(rows), members)]
    return [(slot, 1, 1, [m]) for m, slot in zip(members, rows)]


def _reduce_groups(groups, reduce_rows):
    """run the device reductions; reduce_rows(mask, weights, row0, step, count) -> [count, *kept] on the host (None on a
    rank of a sharded run that is not the destination: it takes part in the collective and keeps nothing)"""
    reduced = {}
    for (mask, weights), members in groups.values():
        for row0, step, count, part in _group_runs(members):
            for c0 in range(0, count, 32768):
                c1 = min(count, c0 + 32768)
                res = reduce_rows(mask, weights, row0 + c0 * step, step, c1 - c0)
                for r, (i, j, _) in enumerate(part[c0:c1] if res is not None else ()):
                    reduced[i, j] = res[r]
    return reduced


def _finish_records(sequence, records, raw, reduced):
    """what the probes hand back, from the downloaded rows `raw` [n_adc, *grid] and the device-reduced records:
    (values, times) as `_pack_values` takes them.  The signal buffer already is the stacked result: when no probe
    post-processes its record (no weights / reduce / phase / post), strided views of it are handed out instead of
    copying every record and stacking the copies again (2 x 336 MB at 1024 x 1024 x 20)"""
    nprobe = len(records[0][1]) if records else 0
    times = _probe_times(sequence)
    if bool(records) and all(op._is_plain() and pb._is_plain() for op, slots in records for pb, _ in slots):
        return _Stacked(raw[j::nprobe] for j in range(nprobe)), times
    values = []
    for i, op in enumerate(op for op in sequence if isinstance(op, Probe)):
        row = []
        for j, (pb, slot) in enumerate(records[i][1]):
            if (i, j) in reduced:
                row.append(op.post(reduced[i, j]))
            elif hasattr(pb, "_assemble"):
                row.append(op.post(pb._assemble(raw[slot], {})))
            else:
                row.append(op.post(np.asarray(pb._finish(raw[slot]))))
        values.append(row)
    return values, times


PIPELINE_MIN_BYTES = 32 << 20      # results from here on leave in voxel slabs while the next slab computes (epgx_run_to_host)


def _jacobian_views(sequence, records, raw, chunk, grid):
    """the usual case -- one pass, Jacobian probes only, nothing post-processes the records: every result is a strided
    VIEW [n_adc, *grid, nvar] of the downloaded rows (variable axis moved last), not a stack of stacks (2 x 6.7 GB of
    copies for 400 TR x 64^3 voxels x 4 columns).  None when some probe needs the general path"""
    if not records:
        return None
    nprobe, nrow = len(records[0][1]), 1 + len(chunk)
    untouched = all(op._is_plain() or (hasattr(op, "_assemble") and not op._post) for op, _ in records)
    views = []
    for j in range(nprobe):
        pbs = {id(slots[j][0]) for _, slots in records}
        pb = records[0][1][j][0]
        block = raw.reshape((len(records), nprobe, nrow) + tuple(grid))[:, j]
        if len(pbs) != 1 or not untouched:
            return None
        if not hasattr(pb, "_assemble"):          # a plain F0 / Z0 probe next to the Jacobians
            if not pb._is_plain():
                return None
            views.append(block[:, 0])
            continue
        cols = [0 if var == "magnitude" else (1 + chunk.index(var) if var in chunk else None) for var in pb.variables]
        if not cols or None in cols:              # unknown variables are zeros: general path
            return None
        block = block if cols == list(range(nrow)) else block[:, cols]
        views.append(np.moveaxis(block, 1, -1))
    return views


def _simulate_jacobian(sequence, probes, variables, init, devices, options, exact_partials=False, packed=True, fuse=True,
                       to_host=True, dtype=np.complex128, shape=None):
    """derivative passes: the state and up to 3 derivative states per launch (diff.py:119-139);
    the derivative states start from zero (an `init` state matrix carries no partials here).
    `to_host=False` (simulate(out="device")): one pass on one GPU, the rows stay in HBM -- per probe a DeviceJacobian"""
    if not to_host and (len(variables) > _lib.MAX_VARS or init is not None or len(devices) > 1):
        raise NotImplementedError(f'out="device" with Jacobian probes: one pass (at most {_lib.MAX_VARS} variables) on one GPU, from equilibrium')
    ctx0 = init._ctx if init is not None else None
    options = dict(options)
    if init is not None:
        options.setdefault("kvalue", init.kvalue)
    base, partials = {}, {}      # (probe index in the sequence, probe index) -> arrays
    first, per_pass = 0, _lib.MAX_VARS
    while first < len(variables):
        chunk = variables[first:first + per_pass]
        enc, records, _ = compile_sequence(sequence, probes, options=options, variables=chunk,
                                           shape=init.shape if init is not None else shape,
                                           nstate0=init.nstate if init is not None else 0,
                                           kspace0=init._kspace if init is not None else None,
                                           dense_start=init is not None, fuse=fuse)
        enc.deriv_flags |= _lib.DERIV_THROUGH_PLAIN_OPS if exact_partials else 0
        K = enc.capacity(at_least=(init.nstate + 1) if init is not None else 0)
        state_in = None
        if init is not None:
            work = init.copy()      # never mutate the caller's init (functions.py:149)
            work._broadcast_to(enc.grid)
            work._reserve(max(K, init._state.K))
            state_in, K = work._state, work._state.K
        if K > _lib.MAX_DERIV_K:
            raise NotImplementedError(
                f"derivatives with {enc.peak + 1} phase states per voxel: the device path keeps at most "
                f"{_lib.MAX_DERIV_K}; bound the state matrix with max_nstate=...")
        if len(chunk) > _lib.max_vars(K):      # long state matrices: fewer derivative states per pass (1024 orders: one)
            per_pass = _lib.max_vars(K)
            continue
        if not to_host and len(chunk) < len(variables):
            raise NotImplementedError(f'out="device" with Jacobian probes: one pass -- at {K} orders per voxel {_lib.max_vars(K)} variable(s)')
        first += len(chunk)
        fleet = _Fleet(enc, K, devices, ctx0)
        if state_in is None and packed and enc.packable(derivatives=True):
            K = enc.packable(derivatives=True)     # at most 16 / 32 orders: four / two voxels per wavefront
        nbytes = 16 * enc.n_adc * enc.nvox
        if not to_host:
            handles = _jacobian_handles(sequence, records, fleet.sigs[0], chunk, enc)
            fleet.run(K, None)
            return _Stacked(handles), _probe_times(sequence)
        raw = _lib.result_empty(fleet.ctxs[0], (enc.n_adc,) + enc.grid, dtype)
        if state_in is None and (nbytes >= PIPELINE_MIN_BYTES or fleet.n > 1):
            # as in the plain path: voxel slabs whose rows leave over PCIe while the next slab computes
            fleet.run_to_host(K, raw)
        else:
            fleet.run(K, state_in)
            fleet.download(raw)
        fleet.free()
        if len(variables) == len(chunk):
            views = _jacobian_views(sequence, records, raw, chunk, enc.grid)
            if views is not None:
                return _Stacked(views), _probe_times(sequence)
        _collect_jacobian(records, raw, chunk, base, partials)
    return _finish_jacobian(sequence, records, base, partials)


def _jacobian_handles(sequence, records, buf, chunk, enc):
    """out="device": per probe of the ADCs a handle on its rows of the signal buffer [record][probe][1 + V][voxel] -- a
    DeviceJacobian for a Jacobian probe, a DeviceSignal for a plain F0 / Z0 probe next to it.  Raises where the host path
    would have to post-process (weights, phases, callable probes) or fill in zeros (unknown variables)"""
    if not records:
        return []
    nprobe, nrow = len(records[0][1]), 1 + len(chunk)
    if not all(op._is_plain() or (hasattr(op, "_assemble") and not op._post) for op, _ in records):
        raise NotImplementedError('out="device" returns raw records: no weights / reduce / phase / post on the probes')
    handles = []
    for j in range(nprobe):
        pb = records[0][1][j][0]
        if len({id(slots[j][0]) for _, slots in records}) != 1:
            raise NotImplementedError('out="device": one probe object per position of the probe list')
        if not hasattr(pb, "_assemble"):
            if not pb._is_plain():
                raise NotImplementedError('out="device" returns raw records: no weights / reduce / phase / post on the probes')
            handles.append(DeviceSignal(buf, enc.n_adc, enc.grid, j * nrow, nprobe * nrow))
            continue
        rows = [0 if var == "magnitude" else (1 + chunk.index(var) if var in chunk else None) for var in pb.variables]
        if not rows or None in rows:
            raise NotImplementedError('out="device": a Jacobian variable that no operator of the sequence declares')
        handles.append(DeviceJacobian(buf, len(records), enc.grid, nprobe, nrow, j, rows, list(pb.variables)))
    return handles


def _collect_jacobian(records, raw, chunk, base, partials):
    """rows of one derivative pass -> per probe record its state row and {variable: derivative row}"""
    for i, (_, slots) in enumerate(records):
        for j, (_, slot) in enumerate(slots):
            base[i, j] = raw[slot]
            partials.setdefault((i, j), {}).update({var: raw[slot + 1 + v] for v, var in enumerate(chunk)})


def _finish_jacobian(sequence, records, base, partials):
    values, i = [], 0
    for op in sequence:
        if isinstance(op, Probe):
            row = []
            for j, (pb, _) in enumerate(records[i][1]):
                if hasattr(pb, "_assemble"):
                    row.append(op.post(pb._assemble(base[i, j], partials[i, j])))
                else:
                    row.append(op.post(np.array(pb._finish(base[i, j]))))
            values.append(row)
            i += 1
    return values, _probe_times(sequence)


def _simulate_device(sequence, probes, init, mode, devices, options, exact_partials=False, fuse=True, packed=True,
                     progress=None, to_host=True, dtype=np.complex128, shape=None):
    variables = _jacobian_variables(sequence, probes)
    if variables:
        if mode == "stream":
            raise NotImplementedError("derivatives run state-resident (no mode='stream')")
        return _simulate_jacobian(sequence, probes, variables, init, devices, options, exact_partials, packed, fuse, to_host, dtype, shape)
    grid0 = init.shape if init is not None else shape
    options = dict(options)
    if init is not None:
        options.setdefault("kvalue", init.kvalue)
    enc, records, bounds = compile_sequence(sequence, probes, shape=grid0, options=options,
                                            nstate0=init.nstate if init is not None else 0,
                                            kspace0=init._kspace if init is not None else None,
                                            dense_start=init is not None, fuse=fuse)
    K = enc.capacity(at_least=(init.nstate + 1) if init is not None else 0, resident=(init is None and mode == "resident"))
    if init is not None:
        K = max(K, init._state.K)
    elif mode == "resident" and packed and enc.packable_nd():
        # integer n-D shifts / diffusion with at most 16 orders (BASELINE config 5: 7), state-resident from equilibrium: the
        # plan's gather / diffusion tables are built [3][16] and four voxels share a wavefront (rows_kernel, R = 1)
        K = enc.packable_nd()
    fleet = _Fleet(enc, K, devices, init._ctx if init is not None else None)
    ctx, plan, sig = fleet.ctxs[0], fleet.plans[0], fleet.sigs[0]
    nvox = enc.nvox
    state_in = None
    if init is not None:
        work = init.copy()  # never mutate the caller's init (functions.py:149)
        work._broadcast_to(enc.grid)
        work._reserve(K)
        state_in = work._state
        K = state_in.K
    if mode == "stream":
        state = state_in if state_in is not None else _lib.DeviceState(ctx, nvox, K)
        begin = 0
        for end in bounds + ([plan.n_ops] if (not bounds or bounds[-1] < plan.n_ops) else []):
            if end > begin:
                _lib.run(ctx, plan, begin, end, 0, nvox, state, state, K, sig.ptr.value, nvox, 0)
                if progress is not None:
                    ctx.synchronize()
                    progress.step()
            begin = end
    # Adc(weights=..., reduce=...): the weighted sums over grid axes run on the device
    # (epgx_signal_reduce), only the reduced records travel to the host
    groups = _reduction_groups(records, enc.grid)
    need_raw = any(id(pb) not in groups for _, slots in records for pb, _ in slots)
    plain = bool(records) and all(op._is_plain() and pb._is_plain() for op, slots in records for pb, _ in slots)
    if not to_host and not plain:
        raise NotImplementedError('out="device" returns raw F0 / Z0 records: no weights / reduce / phase / post on the probes')
    raw = None
    if mode != "stream":
        # short state matrices (max_nstate <= 15, the reference's usual MRF setting): 4 voxels per wave
        K_run = (enc.packable() if (state_in is None and packed) else 0) or K
        nbytes = 16 * enc.n_adc * nvox
        if state_in is None and not groups and records and to_host and (nbytes >= PIPELINE_MIN_BYTES or fleet.n > 1):
            # the whole signal goes to the host: run in voxel slabs whose columns leave over PCIe while the next slab
            # computes -- the call then lasts as long as the copy (epgx_run_to_host); into a recycled page-locked block of
            # the context's pool, or through the library's staging ring into a plain array
            raw = _lib.result_empty(ctx, (enc.n_adc,) + enc.grid, dtype)      # (complex64: narrowed on the device, slab by slab)
            fleet.run_to_host(K_run, raw)
        else:
            fleet.run(K_run, state_in)
    reduced = _reduce_groups(groups, fleet.reduce)
    if not to_host:
        # out="device": the signal stays in HBM (dictionary matching, further reductions ...): per probe a handle on
        # its rows of the buffer(s) instead of a NumPy array
        nprobe = len(records[0][1]) if records else 0
        step = max(nprobe, 1)
        if fleet.n == 1:
            return _Stacked(DeviceSignal(sig, enc.n_adc, enc.grid, j, step) for j in range(nprobe)), _probe_times(sequence)
        return _Stacked(ShardedDeviceSignal([DeviceSignal(buf, enc.n_adc, enc.grid, j, step, vox0=v0, count=cnt)
                                             for buf, (v0, cnt) in zip(fleet.sigs, fleet.bounds)], enc.grid)
                        for j in range(nprobe)), _probe_times(sequence)
    if need_raw and raw is None:
        raw = fleet.download(_lib.result_empty(ctx, (enc.n_adc,) + enc.grid, dtype))
    fleet.free()
    return _finish_records(sequence, records, raw, reduced)


class DeviceSignal:
    """the records of one probe left in HBM (`simulate(..., out="device")`): rows row0, row0 + step, ... of the
    signal buffer [n_adc][count] complex128, which holds voxels [vox0, vox0 + count) of the grid (the whole grid unless
    the run was sharded over GPUs).  `np.asarray(sig)` / `sig.download()` copies them to the host;
    `sig.ptr`, `sig.shape`, `sig.row_stride` describe them to other device code; the buffer goes back to the context's
    pool when the last handle on it is dropped"""

    def __init__(self, buf, n_adc, grid, row0, step, vox0=0, count=None):
        self._buf, self._n_adc, self.grid = buf, int(n_adc), tuple(grid)
        self.row0, self.step = int(row0), int(step)
        self.nvox = int(np.prod(grid))
        self.vox0, self.count = int(vox0), self.nvox if count is None else int(count)
        self.whole = self.count == self.nvox
        nrec = len(range(self.row0, self._n_adc, self.step))
        self.shape = (nrec,) + (self.grid if self.whole else (self.count,))
        self.dtype = np.dtype(np.complex128)
        self.device = buf.ctx.device
        self.ptr = buf.ptr.value + 16 * self.row0 * self.count
        self.row_stride = self.step * self.count          # elements between consecutive records

    def download(self):
        shape = (self._n_adc,) + (self.grid if self.whole else (self.count,))
        full = self._buf.download(np.complex128, shape, out=_lib.result_empty(self._buf.ctx, shape, np.complex128))
        return full[self.row0::self.step]

    def __array__(self, dtype=None, copy=None):
        out = self.download()
        return out if dtype is None else out.astype(dtype)

    def __len__(self):
        return self.shape[0]


class DeviceJacobian:
    """the records of one Jacobian probe left in HBM (`simulate(..., probe=Jacobian(...), out="device")`): the signal buffer holds,
    per ADC and probe, 1 + V rows of `nvox` complex128 -- the probed state, then its derivative w.r.t. every variable of the
    plan.  `sig.ptr` points at row 0 of the first record of this probe, `sig.record_stride` (elements) leads from one ADC to
    the next, `sig.rows[c]` is the row (inside a record) of column c of the probe's `variables`, `sig.row_stride` = nvox.
    `np.asarray(sig)` / `sig.download()` gives what the host path returns: [n_adc, *grid, len(variables)].
    `sig.column(var)` is a DeviceSignal on one column.  The buffer goes back to the context's pool with its last handle."""

    def __init__(self, buf, nrec, grid, nprobe, nrow, j, rows, variables):
        self._buf, self.nrec, self.grid = buf, int(nrec), tuple(grid)
        self.nvox = int(np.prod(grid))
        self._nprobe, self._nrow, self._j = int(nprobe), int(nrow), int(j)
        self.rows, self.variables = list(rows), list(variables)
        self.shape = (self.nrec,) + self.grid + (len(self.variables),)
        self.dtype = np.dtype(np.complex128)
        self.device = buf.ctx.device
        self.ptr = buf.ptr.value + 16 * self.nvox * self._j * self._nrow
        self.row_stride = self.nvox
        self.record_stride = self._nprobe * self._nrow * self.nvox

    def column(self, variable):
        """DeviceSignal [n_adc, *grid] of one column (`variable` as in the probe's list)"""
        row = self.rows[self.variables.index(variable)]
        return DeviceSignal(self._buf, self.nrec * self._nprobe * self._nrow, self.grid, self._j * self._nrow + row,
                            self._nprobe * self._nrow)

    def download(self):
        shape = (self.nrec * self._nprobe * self._nrow,) + self.grid
        full = self._buf.download(np.complex128, shape, out=_lib.result_empty(self._buf.ctx, shape, np.complex128))
        block = full.reshape((self.nrec, self._nprobe, self._nrow) + self.grid)[:, self._j]
        block = block if self.rows == list(range(self._nrow)) else block[:, self.rows]
        return np.moveaxis(block, 1, -1)

    def __array__(self, dtype=None, copy=None):
        out = self.download()
        return out if dtype is None else out.astype(dtype)

    def __len__(self):
        return self.nrec


class ShardedDeviceSignal:
    """the records of one probe of a run over several GPUs (`simulate(..., ngpu=N, out="device")`): `.parts[g]` is the
    DeviceSignal of GPU g's voxel slab (`.vox0`, `.count`: its range of the flattened grid).  `np.asarray(sig)` /
    `sig.download()` assembles the slabs on the host, every GPU over its own PCIe link"""

    def __init__(self, parts, grid):
        self.parts, self.grid = list(parts), tuple(grid)
        self.shape = (self.parts[0].shape[0],) + self.grid
        self.dtype = np.dtype(np.complex128)

    def download(self):
        nrec, nvox = self.shape[0], int(np.prod(self.grid))
        out = _lib.result_empty(self.parts[0]._buf.ctx, (nrec, nvox), np.complex128)

        def fetch(part):
            if part.count:
                part._buf.download_2d(out, part.vox0, part.count, nrec, part.row_stride, offset=part.row0 * part.count)
        if len(self.parts) > 1:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(len(self.parts)) as pool:
                list(pool.map(fetch, self.parts))
        else:
            fetch(self.parts[0])
        return out.reshape(self.shape)

    def __array__(self, dtype=None, copy=None):
        out = self.download()
        return out if dtype is None else out.astype(dtype)

    def __len__(self):
        return self.shape[0]


class _Stacked(tuple):
    """per-probe arrays [n_adc, *grid] that are already stacked (views of the downloaded signal)"""


def _simulate_stepwise(sequence, probes, init, shape, callback, device, options, progress=None):
    """reference-shaped loop (functions.py:173-192) over device launches; used for callbacks
    and for probes that need the full state on the host"""
    if init is None:
        sm = statematrix.StateMatrix([0, 0, 1], shape=shape, device=device, **options)
    else:
        sm = init.copy()
        sm.options.update(options)
    values, times, tic, pending = [], [], 0, []
    # sequences that carry derivatives go through op(sm) one by one so that sm.order1 follows
    # (DiffOperator.__call__); everything else is batched between probes / callbacks
    one_by_one = any(getattr(op, "order1", None) or getattr(op, "order2", None) for op in sequence)

    def flush():
        nonlocal sm
        if pending:
            common_shape = common.broadcast_shapes(sm.shape, *[o.shape for o in pending], append=True)
            if len(common_shape) > sm.ndim:
                sm.expand(len(common_shape))
            sm = _plan.apply_operators(sm, [part for o in pending for part in o._parts()])
            pending.clear()

    for op in sequence:
        tic = tic + op.duration
        if progress is not None:
            progress.step()
        if isinstance(op, Probe):
            flush()
            values.append([(pb or op).acquire(sm, post=op.post) for pb in (probes or [op])])
            times.append(tic)
            continue
        if not common.broadcastable(sm.shape, op.shape, append=True):
            raise ValueError(f"Incompatible StateMatrix and operator shapes: {sm.shape}, {op.shape}")
        if one_by_one:
            if len(common.broadcast_shapes(sm.shape, op.shape, append=True)) > sm.ndim:
                sm.expand(len(common.broadcast_shapes(sm.shape, op.shape, append=True)))
            sm = op(sm, inplace=True)
        else:
            pending.append(op)
        if callback:
            flush()
            callback(sm)
    flush()
    return values, times
