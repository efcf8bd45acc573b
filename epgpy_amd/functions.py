"""simulate() and sequence helpers (mirrors epgpy/functions.py:14-192, :355-369).

`simulate` keeps the reference signature and return conventions, but instead of looping
over operators in Python it compiles the flattened sequence into one plan and runs it in a
single launch of the fused HIP kernel with every voxel's state resident in registers
("state-resident" mode).  `mode="stream"` runs the same plan one ADC-to-ADC segment per
launch with the state streamed through HBM (the per-timestep mode whose HBM roofline
BASELINE.md quotes); both modes execute the same device code and give identical bits.
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
    NotImplementedError (functions.py:350-352); `simulate` applies the same pass by default (`fuse=True`)."""
    from . import fusion

    flat = flatten_sequence(seq)
    return fusion.fuse_sequence(flat) if fusion.fusable(flat) else flat


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


FUSED_TABLE_BUDGET = float(os.environ.get("EPGX_FUSED_TABLE_BUDGET", 1.25e9))     # bytes of device-generated fused tables per plan (compile_sequence)
FUSE_DERIVATIVES = {64: 1, 32: 2}     # orders per voxel -> most variables per plan for which differentiated E . T . E runs are fused


def _fusion_pays(sequence, variables, nstate0, options, from_state=False):
    """Plain plans: always.  Differentiated plans: a fused record reads its table entry AND one partial entry per
    variable (96 + 112 V bytes per voxel and echo).  The four-voxels-per-wavefront kernels fetch those as prefetched
    lines and have straight-line bodies for them -- rows_deriv_kernel (one variable, up to 64 orders): 20-echo
    1024 x 1024 train 4.4 -> 3.5 ms; packed_deriv_kernel (up to 32 orders, one / two variables): 3.1 -> 2.7, 4.8 -> 4.2 ms
    (16 orders: 1.56 -> 1.35, 2.36 -> 2.09 ms).  deriv_kernel (more variables or more orders) reads them as dependent
    scalar loads in front of every record and loses more than the shorter arithmetic gains (two / three variables at
    64 orders: 8.1 -> 10.2, 9.5 -> 13.6 ms), and so does the packed kernel with three variables (at its register budget:
    the fused records take its flag-tested body, 6.6 -> 8.1 ms): those plans keep their three stages."""
    if not variables:
        return True
    if from_state:                   # (a run from a given state matrix takes deriv_kernel)
        return False
    if any(isinstance(op, _shift.S) and (not isinstance(op.k, int) or abs(op.k) != 1) for op in sequence):
        return False                 # (neither of the two kernel families takes the plan then)
    peak = nstate0 + int(getnshift(sequence))
    cap = (options or {}).get("max_nstate")
    orders = (min(peak, int(cap)) if cap else peak) + 1
    return any(orders <= k and len(variables) <= most for k, most in FUSE_DERIVATIVES.items())


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
             asarray=True, disp=False, device=None, mode="auto", exact_partials=False, fuse=True, packed=True,
             out="host", **options):
    """simulate a sequence; values are returned for every Probe/ADC (functions.py:50-170)

    Extra keywords (not in the reference): `device` (GPU index), `mode` in
    {"auto", "resident", "stream", "stepwise"}; `exact_partials`: with Jacobian probes, let
    SPOILER / RESET / PD / D act on the derivative states too.  The reference applies them to the
    state only (they are plain Operators, operator.py:95-104), so its Jacobian after e.g. a spoiler
    is not the derivative of the spoiled signal; the default reproduces the reference's numbers.
    `fuse`: collapse E . T . E runs into single operators (fusion.py; rounding-level differences).
    `packed`: state matrices of at most 16 orders run four voxels per wavefront (identical bits).
    `out`: "host" (NumPy arrays, as the reference) or "device": the records of every probe stay in HBM and a
    `DeviceSignal` handle is returned per probe (plain F0 / Z0 probes, device modes only) -- for consumers that
    reduce or match the dictionary on the GPU and never need the 16 GB of a 10^6-voxel MRF signal on the host.
    """
    sequence = flatten_sequence(sequence)
    nshift, shape = getnshift(sequence), getshape(sequence)
    LOGGER.info(f"Simulate sequence: num. operators: {len(sequence)}, num. shifts: {nshift}, shape: {shape}")
    if squeeze and not any(getattr(op, "order1", None) or getattr(op, "order2", None) for op in sequence):
        sequence = squeeze_sequence(sequence)     # (derivative plans keep their operators apart: partials are per operator)
    if not any(isinstance(op, Probe) for op in sequence):
        raise ValueError("Cannot simulate sequence without at least one Probe/ADC operator")

    probes = []
    if probe:
        probes = probe if isinstance(probe, (tuple, list)) else [probe]
        probes = [pb if isinstance(pb, (Probe, type(None))) else Probe(pb) for pb in probes]

    if init is not None and not isinstance(init, statematrix.StateMatrix):
        init = statematrix.StateMatrix(init, shape=shape, device=device, **options)
    elif init is not None:
        if not common.broadcastable(init.shape, shape, append=True):
            raise ValueError(f"Incompatible StateMatrix and operator shapes: {init.shape}, {shape}")
        options = {**init.options, **options}

    on_device = all((pb or op)._device_kind() is not None
                    for op in sequence if isinstance(op, Probe) for pb in (probes or [op]))
    if mode == "auto":
        mode = "resident" if (on_device and not callback) else "stepwise"
    if mode in ("resident", "stream") and (callback or not on_device):
        raise ValueError(f"mode={mode!r} needs device-recordable probes (F0/Z0) and no callback")
    if out not in ("host", "device"):
        raise ValueError(f'out={out!r}: expected "host" or "device"')
    if out == "device" and mode == "stepwise":
        raise ValueError('out="device" needs device-recordable probes (F0/Z0) and no callback')

    progress = None
    if disp:
        from . import utils
        progress = utils.Progress(sum(isinstance(op, Probe) for op in sequence) if mode == "stream" else
                                  (len(sequence) if mode == "stepwise" else 1))
    if mode == "stepwise":
        values, times = _simulate_stepwise(sequence, probes, init, shape, callback, device, options, progress)
    else:
        values, times = _simulate_device(sequence, probes, init, mode, device, options, exact_partials, fuse, packed, progress,
                                         to_host=(out != "device"))
    if progress is not None:
        progress.close()

    if isinstance(values, _Stacked):
        values = tuple(values) if (asarray or out == "device") else tuple(tuple(arr) for arr in values)
    else:
        values = tuple(zip(*values))
        if asarray:
            values = tuple(np.asarray(arr) for arr in values)
    if asarray:
        times = np.asarray(times)
    if len(values) == 1:
        values = values[0]
    if adc_time:
        return times, values
    return values


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


def _simulate_jacobian(sequence, probes, variables, init, device, options, exact_partials=False, packed=True, fuse=True):
    """derivative passes: the state and up to 3 derivative states per launch (diff.py:119-139);
    the derivative states start from zero (an `init` state matrix carries no partials here)"""
    ctx = init._ctx if init is not None else _lib.get_context(device)
    options = dict(options)
    if init is not None:
        options.setdefault("kvalue", init.kvalue)
    base, partials = {}, {}      # (probe index in the sequence, probe index) -> arrays
    for first in range(0, len(variables), _lib.MAX_VARS):
        chunk = variables[first:first + _lib.MAX_VARS]
        enc, records, _ = compile_sequence(sequence, probes, options=options, variables=chunk,
                                           shape=init.shape if init is not None else None,
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
        plan = enc.device_plan(ctx, K)
        sig = _lib.DeviceBuffer(ctx, 16 * enc.n_adc * enc.nvox)
        if state_in is None and packed and enc.packable(derivatives=True):
            K = enc.packable(derivatives=True)     # at most 16 / 32 orders: four / two voxels per wavefront
        raw, nbytes = None, 16 * enc.n_adc * enc.nvox
        if state_in is None and (32 << 20) <= nbytes <= _lib.PINNED_MAX_BYTES:
            # as in the plain path: voxel slabs whose rows leave over PCIe while the next slab computes, into a recycled
            # page-locked block (a pageable 671 MB result of the 1024 x 1024 one-variable Jacobian took 95 ms to download)
            raw = _lib.pinned_empty(ctx, (enc.n_adc,) + enc.grid, np.complex128)
        if raw is not None:
            _lib.run_to_host(ctx, plan, K, sig.ptr.value, raw)
        else:
            _lib.run(ctx, plan, 0, plan.n_ops, 0, enc.nvox, state_in, None, K, sig.ptr.value, enc.nvox, 0)
            raw = sig.download(np.complex128, (enc.n_adc,) + enc.grid,
                               out=_lib.host_empty((enc.n_adc,) + enc.grid, np.complex128))
        sig.free()
        # the usual case -- one pass, Jacobian probes only, nothing post-processes the records: every
        # result is a strided VIEW [n_adc, *grid, nvar] of the downloaded rows (variable axis moved last),
        # not a stack of stacks (2 x 6.7 GB of copies for 400 TR x 64^3 voxels x 4 columns)
        if len(variables) <= _lib.MAX_VARS and records:
            nprobe, nrow = len(records[0][1]), 1 + len(chunk)
            views = []
            untouched = all(op._is_plain() or (hasattr(op, "_assemble") and not op._post) for op, _ in records)
            for j in range(nprobe):
                pbs = {id(slots[j][0]) for _, slots in records}
                pb = records[0][1][j][0]
                block = raw.reshape((len(records), nprobe, nrow) + enc.grid)[:, j]
                if len(pbs) != 1 or not untouched:
                    views = None
                    break
                if not hasattr(pb, "_assemble"):          # a plain F0 / Z0 probe next to the Jacobians
                    if not pb._is_plain():
                        views = None
                        break
                    views.append(block[:, 0])
                    continue
                cols = [0 if var == "magnitude" else (1 + chunk.index(var) if var in chunk else None)
                        for var in pb.variables]
                if not cols or None in cols:              # unknown variables are zeros: general path
                    views = None
                    break
                block = block if cols == list(range(nrow)) else block[:, cols]
                views.append(np.moveaxis(block, 1, -1))
            if views is not None:
                times, tic = [], 0
                for op in sequence:
                    tic = tic + op.duration
                    if isinstance(op, Probe):
                        times.append(tic)
                return _Stacked(views), times
        for i, (_, slots) in enumerate(records):
            for j, (_, slot) in enumerate(slots):
                base[i, j] = raw[slot]
                partials.setdefault((i, j), {}).update(
                    {var: raw[slot + 1 + v] for v, var in enumerate(chunk)})
    values, times, tic, i = [], [], 0, 0
    for op in sequence:
        tic = tic + op.duration
        if isinstance(op, Probe):
            row = []
            for j, (pb, _) in enumerate(records[i][1]):
                if hasattr(pb, "_assemble"):
                    row.append(op.post(pb._assemble(base[i, j], partials[i, j])))
                else:
                    row.append(op.post(np.array(pb._finish(base[i, j]))))
            values.append(row)
            times.append(tic)
            i += 1
    return values, times


def _simulate_device(sequence, probes, init, mode, device, options, exact_partials=False, fuse=True, packed=True,
                     progress=None, to_host=True):
    variables = _jacobian_variables(sequence, probes)
    if variables:
        if mode == "stream":
            raise NotImplementedError("derivatives run state-resident (no mode='stream')")
        if not to_host:
            raise NotImplementedError('out="device" is not available for Jacobian probes')
        return _simulate_jacobian(sequence, probes, variables, init, device, options, exact_partials, packed, fuse)
    grid0 = init.shape if init is not None else None
    options = dict(options)
    if init is not None:
        options.setdefault("kvalue", init.kvalue)
    enc, records, bounds = compile_sequence(sequence, probes, shape=grid0, options=options,
                                            nstate0=init.nstate if init is not None else 0,
                                            kspace0=init._kspace if init is not None else None,
                                            dense_start=init is not None, fuse=fuse)
    ctx = init._ctx if init is not None else _lib.get_context(device)
    K = enc.capacity(at_least=(init.nstate + 1) if init is not None else 0)
    if init is not None:
        K = max(K, init._state.K)
    plan = enc.device_plan(ctx, K)
    nvox = enc.nvox
    state_in = None
    if init is not None:
        work = init.copy()  # never mutate the caller's init (functions.py:149)
        work._broadcast_to(enc.grid)
        work._reserve(K)
        state_in = work._state
        K = state_in.K
    sig = _lib.DeviceBuffer(ctx, 16 * max(enc.n_adc, 1) * nvox)
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
    reduced, groups = {}, {}
    for i, (_, slots) in enumerate(records):
        for j, (pb, slot) in enumerate(slots):
            spec = pb._device_reduction(enc.grid) if hasattr(pb, "_device_reduction") else None
            if spec is not None:
                groups.setdefault(id(pb), (spec, []))[1].append((i, j, slot))
    raw = None
    if mode != "stream":
        # short state matrices (max_nstate <= 15, the reference's usual MRF setting): 4 voxels per wave
        K_run = (enc.packable() if (state_in is None and packed) else 0) or K
        nbytes = 16 * enc.n_adc * nvox
        if state_in is None and not groups and records and to_host and (32 << 20) <= nbytes <= _lib.PINNED_MAX_BYTES:
            # the whole signal goes to the host: run in voxel slabs whose columns leave over PCIe while the next slab
            # computes, into a recycled page-locked block -- the call then lasts as long as the copy (epgx_run_to_host)
            raw = _lib.pinned_empty(ctx, (enc.n_adc,) + enc.grid, np.complex128)
        if raw is not None:
            _lib.run_to_host(ctx, plan, K_run, sig.ptr.value, raw)
        else:
            _lib.run(ctx, plan, 0, plan.n_ops, 0, nvox, state_in, None, K_run, sig.ptr.value, nvox, 0)
    for (mask, weights), members in groups.values():
        rows = [slot for _, _, slot in members]
        steps = {b - a for a, b in zip(rows, rows[1:])}
        if len(steps) <= 1 and (not steps or min(steps) > 0):
            runs = [(rows[0], steps.pop() if steps else 1, len(rows), members)]
        else:
            runs = [(slot, 1, 1, [m]) for m, slot in zip(members, rows)]
        for row0, step, count, part in runs:
            for c0 in range(0, count, 32768):
                c1 = min(count, c0 + 32768)
                res = _lib.signal_reduce(ctx, sig.ptr.value, nvox, row0 + c0 * step, step, c1 - c0, enc.grid, mask, weights)
                for r, (i, j, _) in enumerate(part[c0:c1]):
                    reduced[i, j] = res[r]
    need_raw = any((i, j) not in reduced for i, (_, slots) in enumerate(records) for j in range(len(slots)))
    if not to_host:
        # out="device": the signal stays in HBM (dictionary matching, further reductions ...): per probe a handle on
        # its rows of the buffer instead of a NumPy array
        if reduced or not all(op._is_plain() and pb._is_plain() for op, slots in records for pb, _ in slots):
            raise NotImplementedError('out="device" returns raw F0 / Z0 records: no weights / reduce / phase / post on the probes')
        times, tic = [], 0
        for op in sequence:
            tic = tic + op.duration
            if isinstance(op, Probe):
                times.append(tic)
        nprobe = len(records[0][1]) if records else 0
        return _Stacked(DeviceSignal(sig, enc.n_adc, enc.grid, j, max(nprobe, 1)) for j in range(nprobe)), times
    if need_raw and raw is None:
        # launches are asynchronous: map the pages of the result array while the kernel runs
        raw = sig.download(np.complex128, (enc.n_adc,) + enc.grid, out=_lib.host_empty((enc.n_adc,) + enc.grid, np.complex128))
    sig.free()

    # the signal buffer already is the stacked result [n_adc, *grid]: when no probe post-processes
    # its record (no weights / reduce / phase / post), hand out strided views of it instead of
    # copying every record and stacking the copies again (2 x 336 MB at 1024 x 1024 x 20)
    nprobe = len(records[0][1]) if records else 0
    plain = bool(records) and all(op._is_plain() and pb._is_plain() for op, slots in records for pb, _ in slots)
    values, times, tic = [], [], 0
    for i, op in enumerate(op for op in sequence if isinstance(op, Probe)):
        _, slots = records[i]
        if not plain:
            row = []
            for j, (pb, slot) in enumerate(slots):
                if (i, j) in reduced:
                    row.append(op.post(reduced[i, j]))
                elif hasattr(pb, "_assemble"):
                    row.append(op.post(pb._assemble(raw[slot], {})))
                else:
                    row.append(op.post(np.asarray(pb._finish(raw[slot]))))
            values.append(row)
    for op in sequence:
        tic = tic + op.duration
        if isinstance(op, Probe):
            times.append(tic)
    if plain:
        return _Stacked(raw[j::nprobe] for j in range(nprobe)), times
    return values, times


class DeviceSignal:
    """the records of one probe left in HBM (`simulate(..., out="device")`): rows row0, row0 + step, ... of the
    signal buffer [n_adc][nvox] complex128.  `np.asarray(sig)` / `sig.download()` copies them to the host;
    `sig.ptr`, `sig.shape`, `sig.row_stride` describe them to other device code; the buffer goes back to the context's
    pool when the last handle on it is dropped"""

    def __init__(self, buf, n_adc, grid, row0, step):
        self._buf, self._n_adc, self.grid = buf, int(n_adc), tuple(grid)
        self.row0, self.step = int(row0), int(step)
        self.nvox = int(np.prod(grid))
        self.shape = (len(range(self.row0, self._n_adc, self.step)),) + self.grid
        self.dtype = np.dtype(np.complex128)
        self.ptr = buf.ptr.value + 16 * self.row0 * self.nvox
        self.row_stride = self.step * self.nvox          # elements between consecutive records

    def download(self):
        full = self._buf.download(np.complex128, (self._n_adc,) + self.grid,
                                  out=_lib.host_empty((self._n_adc,) + self.grid, np.complex128))
        return full[self.row0::self.step]

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
        if pending:
            common_shape = common.broadcast_shapes(sm.shape, *[o.shape for o in pending], append=True)
            if len(common_shape) > sm.ndim:
                sm.expand(len(common_shape))
            _plan.apply_operators(sm, list(pending))
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
