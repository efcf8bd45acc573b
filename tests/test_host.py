"""CPU: host-side logic of the product package (no GPU compute): shape rules, sequence
helpers, plan encoding, operator coefficient tables vs the reference's golden values, and
the C-ABI library (loads, exports every symbol of include/epgx.h, refuses to run w/o GPU)."""
import os
import re

import numpy as np
import pytest

from epgpy_amd import epg, _lib, common, plan as _plan, transition, evolution
from epgpy_amd.distributed import slab_bounds, ShardedPlan
from tests import sequences as sq

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------ C ABI
def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "epgx.h")).read()
    declared = set(re.findall(r"\b(epgx_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 25
    lib = _lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"libepgx.so does not export {name}"
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    assert lib.epgx_abi_version() == _lib.ABI_VERSION == int(re.search(r"#define EPGX_ABI_VERSION (\d+)", header).group(1))


def test_no_cpu_fallback():
    lib = _lib.load()
    if lib.epgx_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_lib.EpgxError, match="no HIP device"):
        _lib.Context(0)
    with pytest.raises(_lib.EpgxError):
        epg.simulate([epg.T(90, 90), epg.ADC])
    with pytest.raises(_lib.EpgxError):
        epg.StateMatrix()


def test_op_record_layout():
    assert _lib.OP_DTYPE.itemsize == 32
    header = open(os.path.join(ROOT, "include", "epgx.h")).read()
    for name, val in [("EPGX_OP_T", _lib.OP_T), ("EPGX_OP_MAT", _lib.OP_MAT), ("EPGX_OP_E", _lib.OP_E),
                      ("EPGX_OP_S", _lib.OP_S), ("EPGX_OP_ADC", _lib.OP_ADC), ("EPGX_OP_SPOIL", _lib.OP_SPOIL),
                      ("EPGX_OP_RESET", _lib.OP_RESET), ("EPGX_OP_PD", _lib.OP_PD)]:
        assert re.search(rf"{name} = {val},", header), name
    assert f"#define EPGX_MAX_SPACES {_lib.MAX_SPACES}" in header
    assert f"#define EPGX_MAX_DIMS {_lib.MAX_DIMS}" in header


# ------------------------------------------------------------------ broadcasting rules
def test_broadcast_shapes_append():
    assert common.broadcast_shapes((3,), (3, 2), append=True) == (3, 2)
    assert common.broadcast_shapes((1,), (2, 3), (2, 1), append=True) == (2, 3)
    assert common.broadcast_shapes((2,), (1, 3), append=True) == (2, 3)
    with pytest.raises(ValueError):
        common.broadcast_shapes((2,), (3,), append=True)
    assert common.broadcastable((4,), (4, 5), append=True)
    assert not common.broadcastable((4,), (3,), append=True)


def test_operator_shapes_follow_reference_rules():
    # test/test_transition.py:19-51
    assert epg.T(90, [90, 0]).shape == (2,)
    assert epg.T(90, [[90, 0]]).shape == (1, 2)
    assert epg.T([[90, 90]], [0, 90]).shape == (2, 2)
    with pytest.raises(ValueError):
        epg.T([90] * 2, [90] * 3)
    # test/test_evolution.py:25-58
    assert epg.E(10, [[1e10, 1e-10]], [[1e10], [1e-10]]).shape == (2, 2)
    assert epg.E([[0, 10]], 1e-10, [1e-10] * 3).shape == (3, 2)
    with pytest.raises(ValueError):
        epg.E(10, [1000] * 3, [100] * 2)
    with pytest.raises(ValueError):
        epg.E([10] * 2, [1000] * 3, 100)
    # test/test_shift.py:169-184
    s = epg.S(-2)
    assert s.nshift == 2 and s.k == -2 and s.shape == (1,)
    with pytest.raises(TypeError):
        epg.S(0)
    with pytest.raises(ValueError):
        epg.T(90, 0, duration=-1)


def test_axes_keyword():
    ax = epg.Axes("FA", "T2")
    refoc = epg.T([180, 150], 0, axes=ax.FA)
    relax = epg.E(10, 1e3, [30, 40, 50], axes=ax.T2)
    assert refoc.shape == (2,) and relax.shape == (1, 3)
    seq = [epg.T(90, 90)] + [epg.S(1), relax, refoc, epg.S(1), relax, epg.ADC] * 2
    assert epg.getshape(seq) == (2, 3)
    assert epg.getnshift(seq) == 4
    with pytest.raises(ValueError):
        epg.getshape(seq + [epg.T([90] * 3, 180)])


# ------------------------------------------------------------------ coefficient tables
def test_T_and_E_tables_match_reference(golden):
    g = golden("g4_operators")
    assert np.array_equal(epg.T(120, 0).mat, g["T_120_0_mat"])
    assert np.array_equal(epg.T(90, 90).mat, g["T_90_90_mat"])
    assert np.array_equal(transition.rotation_operator(g["T_alpha"], g["T_phi"]), g["T_mat"])
    op = epg.E(5, 150, 30, 0.01)
    assert np.array_equal(op.arr, g["E_5_150_30_001_arr"])
    assert np.array_equal(op.arr0, g["E_5_150_30_001_arr0"])
    arr, arr0 = evolution.relaxation_operator(g["E_tau"], g["E_T1"], g["E_T2"], g["E_g"])
    assert np.array_equal(arr, g["E_arr"]) and np.array_equal(arr0, g["E_arr0"])
    parr, p0 = evolution.precession_operator(g["E_tau"], g["E_g"])
    assert np.array_equal(parr, g["P_arr"]) and p0 is None


def test_packed_tables():
    from epgpy_amd import opmatrix, opscalar
    opcode, tab = opmatrix.pack_matrix(epg.T([30.0, 77.0], 15.0).mat)
    assert opcode == _lib.OP_T and tab.shape == (2, 8)
    m = epg.T([30.0, 77.0], 15.0).mat
    assert np.allclose(tab[:, 0], m[:, 0, 0].real) and np.allclose(tab[:, 7], m[:, 2, 2].real)
    assert np.allclose(tab[:, 1] + 1j * tab[:, 2], m[:, 0, 1])
    assert np.allclose(tab[:, 5] + 1j * tab[:, 6], m[:, 2, 0])
    opcode, tab = opmatrix.pack_matrix(epg.Phi(33.0).mat)
    assert opcode == _lib.OP_MAT and tab.shape == (1, 10)
    opcode, tab = opscalar.pack_scalar(*evolution.relaxation_operator(5, 150, 30, 0.01))
    assert opcode == _lib.OP_E and tab.shape == (1, 4)
    assert np.isclose(tab[0, 0] + 1j * tab[0, 1], 0.80505196038198 + 0.2615772384190187j)
    assert np.isclose(tab[0, 3], 0.0327838995179941)


# ------------------------------------------------------------------ sequence helpers
def test_sequence_helpers():
    # test/test_functions.py:6-39
    excit, refoc = epg.T(90, 90), epg.T(180, 0)
    grad, relax = epg.S(1, duration=10), epg.E(10, 1000, 30)
    seq1 = [excit, grad, relax, refoc, grad, relax, epg.ADC]
    seq2 = excit * grad * relax * refoc * grad * relax * epg.ADC
    assert all(a is b for a, b in zip(seq1, seq2))
    assert epg.getnshift(seq1) == epg.getnshift(seq2) == seq2.nshift == 2
    assert epg.getshape(seq1) == epg.getshape(seq2) == (1,)
    assert epg.get_adc_times(seq1) == epg.get_adc_times(seq2) == [20]
    with pytest.raises(ValueError):
        epg.simulate([epg.T(90, 90)])  # no ADC (checked before any device work)
    with pytest.raises(ValueError):
        epg.flatten_sequence([excit, "nope"])
    with pytest.raises(TypeError):
        excit * 3
    # squeeze=True / squeeze_sequence: declared but unimplemented in the reference (functions.py:350-352); here the
    # E . T . E fusion pass -- fewer operators, probes and shifts untouched
    from epgpy_amd import functions, fusion
    mse = sq.mse_ops(epg, 900.0, [40.0, 60.0], necho=3)
    squeezed = functions.squeeze_sequence(mse)
    assert len(squeezed) < len(epg.flatten_sequence(mse)) and any(isinstance(op, fusion.FusedTE) for op in squeezed)
    assert epg.getnshift(squeezed) == epg.getnshift(mse) and epg.get_adc_times(squeezed) == epg.get_adc_times(mse)
    # the pair cache on an operator is bounded (a fitting loop rebuilds its relaxations every iteration)
    rot = epg.T(120, 0)
    for i in range(3 * fusion.FUSE_CACHE):
        fusion.fuse_sequence([rot, epg.E(5.0, 900.0, 40.0 + i), epg.ADC])
    assert len(rot.__dict__["_fused_with"]) <= fusion.FUSE_CACHE


def test_readme_sequence_metadata():
    seq = sq.mse_ops(epg, 150.0, [30.0, 40.0, 50.0])
    assert epg.getshape(seq) == (3,)
    assert epg.getnshift(seq) == 40
    assert epg.get_adc_times(seq) == [10.0 * (i + 1) for i in range(20)]


# ------------------------------------------------------------------ plan encoding
def test_plan_encoding_dedupes_tables_and_tracks_nstate():
    T1 = np.linspace(200, 3000, 8)[:, None]
    T2 = np.linspace(20, 300, 4)[None, :]
    seq = sq.mse_ops(epg, T1, T2)
    enc, records, bounds = epg.compile_sequence(seq, options={"max_nstate": 63}, fuse=False)
    ops, grid, spaces, coef, _ = enc.arrays()
    assert tuple(grid) == (8, 4)
    assert len(ops) == 1 + 6 * 20 and enc.n_adc == 20 and len(records) == 20
    assert bounds[0] == 7 and bounds[-1] == 121
    # one table per distinct operator object: exc (8), rfc (8), rlx (8*4*4)
    assert coef.size == 8 + 8 + 8 * 4 * 4
    assert spaces == [(4, 1)]
    assert enc.peak == 40 and enc.capacity() == 64
    shifts = ops[ops["opcode"] == _lib.OP_S]
    assert np.all(shifts["ia"] == 1) and np.all(shifts["ib"] == 63)
    adcs = ops[ops["opcode"] == _lib.OP_ADC]
    assert list(adcs["ia"]) == list(range(20))
    e_ops = ops[ops["opcode"] == _lib.OP_E]
    assert len(set(e_ops["coef_off"])) == 1 and np.all(e_ops["space"] == 0)


def test_plan_truncation_and_capacity():
    seq = [epg.T(30, 0), epg.S(1)] * 100 + [epg.ADC]
    enc, _, _ = epg.compile_sequence(seq)
    assert enc.peak == 100 and enc.capacity() == 128
    ops = enc.arrays()[0]
    assert np.all(ops[ops["opcode"] == _lib.OP_S]["ib"] == _plan.NO_TRUNCATION)
    enc, _, _ = epg.compile_sequence(seq, options={"max_nstate": 10})
    assert enc.peak == 10 and enc.nstate == 10 and enc.capacity() == 64
    seq = [epg.T(30, 0), epg.S(1, nmax=7)] * 20 + [epg.ADC]
    enc, _, _ = epg.compile_sequence(seq)
    assert enc.nstate == 7
    enc, _, _ = epg.compile_sequence(seq, options={"max_nstate": 12})  # shift.py:86: option wins
    assert enc.nstate == 12
    with pytest.raises(NotImplementedError):
        epg.compile_sequence([epg.S(1)] * 2000 + [epg.ADC])[0].capacity()


def test_plan_index_spaces_follow_axes():
    T1 = np.linspace(500, 1500, 5)[:, None, None]
    T2 = np.linspace(40, 120, 3)[None, :, None]
    B1 = np.linspace(0.7, 1.3, 4)[None, None, :]
    alpha, TR = sq.mrf_trains(4)
    enc, _, _ = epg.compile_sequence(sq.mrf_ops(epg, T1, T2, B1, alpha, TR))
    assert enc.grid == (5, 3, 4)
    assert set(enc.spaces) == {(0, 0, 1), (3, 1, 0)}
    with pytest.raises(ValueError):
        epg.compile_sequence([epg.T([1, 2, 3], 0), epg.E(1, [1, 2], 3), epg.ADC])


def test_probe_classification():
    assert epg.ADC._device_kind() == 0
    assert epg.Adc("Z0")._device_kind() == 1
    assert epg.Adc("F")._device_kind() is None
    assert epg.Probe("F0")._device_kind() == 0
    assert epg.Probe("abs(F0)")._device_kind() is None
    with pytest.raises(ValueError):
        epg.Adc("nope")
    adc = epg.Adc(reduce=1, weights=[[1, 2, 3, 4, 5]])
    raw = np.arange(15.0).reshape(3, 5)
    assert np.allclose(adc._finish(raw), raw @ [1, 2, 3, 4, 5])
    assert np.allclose(epg.Adc(phase=15).post(np.ones(2)), np.exp(1j * np.pi / 12))


# ------------------------------------------------------------------ sharding
def test_slab_bounds():
    slab, b = slab_bounds(10, 4)
    assert slab == 3 and b == [(0, 3), (3, 3), (6, 3), (9, 1)]
    slab, b = slab_bounds(2, 4)
    assert slab == 1 and b == [(0, 1), (1, 1), (2, 0), (2, 0)]
    for n, w in [(1048576, 8), (65536, 3), (7, 7)]:
        slab, b = slab_bounds(n, w)
        assert sum(c for _, c in b) == n and all(c <= slab for _, c in b)
        assert all(b[i][0] + b[i][1] == b[i + 1][0] for i in range(w - 1))


def test_sharded_plan_assemble():
    seq = sq.mse_ops(epg, 150.0, np.linspace(20, 100, 7), necho=3)
    parts = []
    for r in range(3):
        sp = ShardedPlan(seq, rank=r, world_size=3)
        assert sp.nvox == 7 and sp.slab == 3
        block = np.zeros((sp.n_adc, sp.slab), complex)
        block[:, : sp.count] = (np.arange(sp.count) + sp.vox0)[None, :] + 10j * np.arange(sp.n_adc)[:, None]
        parts.append(block)
    full = sp.assemble(np.stack(parts))
    assert full.shape == (3, 7)
    assert np.array_equal(full.real, np.tile(np.arange(7.0), (3, 1)))
    assert np.array_equal(full.imag, 10 * np.arange(3.0)[:, None] * np.ones(7))


def test_stream_segments_and_strong_slabs():
    """per-timestep launches = one per ADC-to-ADC segment; a strong split gives every rank the same plan and its own slab"""
    from epgpy_amd import workloads as wl

    T1, T2 = np.linspace(300, 2000, 6)[:, None], np.linspace(30, 100, 5)[None, :]
    sp = ShardedPlan(wl.mse_sequence(epg, T1, T2, necho=4), rank=0, world_size=1, max_nstate=63)
    assert sp.n_adc == 4 and len(sp.segments()) == 4          # every echo ends with its ADC
    sp2 = ShardedPlan(wl.mse_sequence(epg, T1, T2, necho=4) + [epg.S(1)], rank=0, world_size=1, max_nstate=63)
    assert len(sp2.segments()) == 5                             # + the tail after the last probe
    parts = [ShardedPlan(wl.mse_sequence(epg, T1, T2, necho=4), rank=r, world_size=4, max_nstate=63) for r in range(4)]
    assert [p.slab for p in parts] == [8] * 4 and [(p.vox0, p.count) for p in parts] == [(0, 8), (8, 8), (16, 8), (24, 6)]
    assert all(p.nvox == 30 and p.n_adc == 4 for p in parts)


def test_workload_definitions_agree():
    """product-side (epgpy_amd/workloads.py) and oracle-side (oracle/workloads.py) descriptions are the same workloads"""
    from epgpy_amd import workloads as wl
    from oracle import workloads as ow

    for a, b in zip(wl.mrf_trains(), ow.mrf_trains()):
        assert np.array_equal(a, b) and len(a) == wl.MRF_NTR == ow.MRF_NTR
    assert wl.PGSE_KVALUE == ow.PGSE_KVALUE
    for name, (kind, grid) in wl.GRIDS.items():
        params = wl.grid_parameters(name)
        assert common.broadcast_shapes(*[p.shape for p in params], append=True) == grid
    T1, T2 = wl.grid_parameters("mse_256")
    from epgpy_amd import functions

    seq = functions.flatten_sequence(wl.mse_sequence(epg, T1, T2))
    tup = ow.mse_tuples(T1, T2)
    assert len(seq) == len(tup) == 121 and [type(o).__name__[0] for o in seq] == [t[0][0] for t in tup]
    # weak scaling: rank r of N owns rows [r n1, (r+1) n1) of an N n1-row axis over the same range
    rows = wl.grid_parameters("mse_256", rows=(256, 256, 512))[0]
    assert np.array_equal(rows[:, 0], np.linspace(200, 3000, 512)[256:])
    seq3, params3, nadc3, opts3 = wl.build(epg, "mrf_32")
    assert nadc3 == 1000 and functions.getshape(seq3) == (32, 32, 32) and opts3 == {"max_nstate": 63}
    assert len(functions.flatten_sequence(seq3)) == len(ow.mrf_tuples(*params3, *ow.mrf_trains())) == 5002
    seq5, _, nadc5, opts5 = wl.build(epg, "pgse_512")
    assert nadc5 == 1 and functions.getshape(seq5) == (512, 512) and len(seq5) == len(ow.pgse_tuples(50.0, 1e-3)) == 13


def test_modify_host_logic():
    """test/test_functions.py:110-199 (the parts that need no device)"""
    pulse = epg.T(90, 0, duration=1)
    grad = epg.S(1, duration=5)
    seq = [pulse, grad, pulse, epg.ADC]
    assert epg.modify(seq, lambda op: op) == seq
    newseq = epg.modify(seq, T2=100)
    assert len(newseq) == len(seq) and newseq[0] is newseq[2]
    assert epg.get_adc_times(seq) == epg.get_adc_times(newseq)
    assert all(o1.duration == o2.duration for o1, o2 in zip(seq, newseq))
    flat = epg.flatten_sequence(newseq)
    assert isinstance(flat[0], epg.T) and flat[0].alpha == 90 and flat[0].duration == seq[0].duration
    assert isinstance(flat[1], epg.E) and flat[1].T2 == 100 and flat[1].duration == 0
    mod = epg.flatten_sequence(epg.modify(seq, T2=30))
    assert len(mod) == 7
    assert mod[1].tau == mod[0].duration and mod[3].tau == mod[2].duration and mod[5].tau == mod[4].duration
    seq = [epg.T(90, 90), epg.Wait(1), epg.T(90, 90), epg.ADC]
    newseq = epg.modify(seq, g=[[0, 0.25, 0.5]], att=[1, 0.5])
    assert epg.getshape(newseq) == (2, 3)
    seq2 = epg.modify(seq, T2=[30, 40])
    assert epg.getshape(seq2) == (2,)
    assert epg.getshape(epg.modify(seq2, att=[1, 0.9, 0.7])) == (2, 3)
    assert epg.getshape(epg.modify(seq2, T2=[50, 60], expand=False)) == (2,)
    with pytest.raises(ValueError):
        epg.modify(seq2, att=[1, 0.9, 0.7], expand=False)

    def modifier(op, x):
        return op if not isinstance(op, epg.T) else epg.T(op.alpha, op.phi * np.asarray(x))
    seqc = epg.modify(seq, modifier, x=[0.1, 0.2])
    assert np.allclose(seqc[0].phi, seq[0].phi * np.r_[0.1, 0.2])
    with pytest.raises(TypeError):
        epg.modify(seq, modifier="nope", x=1)


def test_kspace_planner_matches_reference_shiftnd_coordinates():
    """coordinates of test/test_shift.py:34-52 and structural pruning"""
    from epgpy_amd import kspace
    ks = kspace.KSpace(np.zeros((1, 3), int), [True], [True])
    ks1, tab = ks.shifted([1, 0, 0])
    assert np.array_equal(ks1.coords, [[-1, 0, 0], [0, 0, 0], [1, 0, 0]])
    assert tab.shape == (3, 2)
    assert tab[0, 1] == 0 and tab[0, 0] == _lib.GS_ZERO          # A_1 <- A_0, nothing wraps into A_0
    assert tab[1, 0] == _lib.GS_ZERO and tab[2, 0] == 0          # B_0 <- (empty), Z_0 stays
    ks2, tab = ks1.shifted([-1, 0, 0])
    assert np.array_equal(ks2.half[0], [0, 0, 0]) and ks2.nrow in (1, 3, 5)
    # equilibrium start: a shift moves nothing, the set stays {0}
    ks0, tab = kspace.KSpace.equilibrium(3).shifted([1, 1, 1])
    assert ks0.nrow == 1 and tab.shape == (3, 1) and tab[2, 0] == 0
    # PGSE: T, S, T(180), S -> orders 0, d, 2d
    ks = kspace.KSpace.equilibrium(3).after_mixing().shifted([1, 1, 1])[0].after_mixing().shifted([1, 1, 1])[0]
    assert np.array_equal(ks.half, [[0, 0, 0], [1, 1, 1], [2, 2, 2]])
    # nmax crops coordinates (shift.py:330-341)
    ks = kspace.KSpace.from_orders(2, 1).shifted([1], nmax=2)[0]
    assert ks.nstate == 2
    # b-matrices of diffusion.py:86-123 (test/test_diffusion.py:8-20)
    ks = kspace.KSpace.from_orders(1, 1)
    bL, bT, bM = ks.bmatrices(2e3)
    assert np.allclose(bL[1], 4e-3) and np.allclose(bT[1], 4e-3)
    bL, bT, bM = kspace.KSpace.from_orders(3, 1).bmatrices(1e3, shift=[1])
    assert np.allclose(bT[3], (4 + 2 + 1 / 3) * 1e-3)            # k1 = 2e3, k2 = 3e3 rad/m


def test_kspace_planner_with_one_shift_vector_per_voxel(golden):
    """a vectorised k (shift.py:38-41, test_shift.py:196-203): ONE row structure, coordinates per voxel; every row the
    reference ends with (G14) is in the planned set, at the same place in the sort order; D tables follow the voxel"""
    from epgpy_amd import kspace
    from tests import sequences as sq
    sm1 = kspace.KSpace.from_orders(0, 3).shifted([[1, 0, 0], [2, 0, 0]])[0]        # test_shift.py:196-203
    assert sm1.lead == (2,) and sm1.nstate == 1 and sm1.coords.shape == (2, 3, 3)
    assert np.array_equal(sm1.coords[0] * 2, sm1.coords[1])
    assert sm1.half.shape == (2, 2, 3) and sm1.bmatrices(1.0)[0].shape == (2, 2, 3, 3)
    again = kspace.KSpace.from_coords(sm1.coords, sm1.nz_f, sm1.nz_z)               # round trip of `StateMatrix.coords`
    assert again.lead == (2,) and np.array_equal(again.points, sm1.points)
    with pytest.raises(ValueError):
        kspace.KSpace.from_coords(sm1.coords[:, ::-1])                              # not sorted
    with pytest.raises(ValueError):
        kspace.KSpace.from_coords(np.abs(sm1.coords))                               # not symmetric

    g = golden("g14_nd_vector")
    for name, tuples, opts in sq.nd_vector_cases():
        ops = sq.nd_to_ops(epg, tuples)
        enc, _, _ = epg.compile_sequence(ops, options=opts)
        ks = enc.kspace
        ref = g[name + "_coords"]                                                   # [*lead.., R, kdim]
        ref_rows = np.moveaxis(ref, -2, 0).reshape(ref.shape[-2], -1)
        assert ks.lead == (4,) and ks.points.shape[1:] == (4, ref.shape[-1])
        mine = ks.points.reshape(ks.nrow, -1)
        where = {tuple(r): i for i, r in enumerate(mine.tolist())}
        at = [where[tuple(r)] for r in ref_rows.tolist()]                           # KeyError: a row the reference keeps is missing
        assert at == sorted(at) and at[(len(at) - 1) // 2] == ks.centre


def test_combine_host_algebra():
    """`@` combination: test/test_opscalar.py:50-79, test/test_opmatrix.py:50-109 (algebra only)"""
    from epgpy_amd.opscalar import ScalarOp
    from epgpy_amd.opmatrix import MatrixOp
    rs = np.random.RandomState(0)

    def sym_arr():
        a = rs.uniform(-1, 1, (3, 2)).dot([1, 1j])
        return a + a[..., (1, 0, 2)].conj()

    def sym_mat():
        m = rs.uniform(-1, 1, (3, 3, 2)).dot([1, 1j])
        return m + m[..., (1, 0, 2), :][..., (1, 0, 2)].conj()

    arr, arr0, arr_, arr0_ = sym_arr(), sym_arr(), sym_arr(), sym_arr()
    op = ScalarOp(arr) @ ScalarOp(arr_)
    assert np.allclose(op.arr, arr_ * arr) and op.arr0 is None
    op = ScalarOp(arr, arr0, name="a") @ ScalarOp(arr_, name="b")
    assert np.allclose(op.arr0, arr_ * arr0) and op.name == "a|b"
    op = ScalarOp(arr) @ ScalarOp(arr_, arr0_)
    assert np.allclose(op.arr0, arr0_)
    op = ScalarOp(arr, arr0) @ ScalarOp(arr_, arr0_)
    assert np.allclose(op.arr, arr_ * arr) and np.allclose(op.arr0, arr_ * arr0 + arr0_)
    assert np.allclose(ScalarOp(arr).mat[0], np.diag(arr))

    mat, mat0, mat_, mat0_ = sym_mat(), sym_mat(), sym_mat(), sym_mat()
    op = MatrixOp(mat) @ MatrixOp(mat_)
    assert np.allclose(op.mat, mat_ @ mat) and op.mat0 is None
    op = MatrixOp(mat, mat0) @ MatrixOp(mat_, mat0_)
    assert np.allclose(op.mat, mat_ @ mat) and np.allclose(op.mat0, mat_ @ mat0 + mat0_)
    op = MatrixOp(mat_, mat0_) @ MatrixOp(mat, mat0)
    assert np.allclose(op.mat, mat @ mat_) and np.allclose(op.mat0, mat @ mat0_ + mat0)
    # matrix with scalar, both orders (ScalarOp defers to MatrixOp.__rmatmul__)
    op = MatrixOp(mat, mat0) @ ScalarOp(arr, arr0)
    assert np.allclose(op.mat, np.diag(arr) @ mat) and np.allclose(op.mat0, np.diag(arr) @ mat0 + np.diag(arr0))
    op = ScalarOp(arr, arr0) @ MatrixOp(mat, mat0)
    assert isinstance(op, MatrixOp)
    assert np.allclose(op.mat, mat @ np.diag(arr)) and np.allclose(op.mat0, mat @ np.diag(arr0) + mat0)
    # E @ T on operators with different shapes, durations add up
    e = epg.E(5, 800, [40, 80], duration=True)
    t = epg.T([[30, 60, 90]], 10)
    et = e @ t
    assert et.shape == (2, 3) and et.duration == 5 and et.name == f"{e.name}|{t.name}"
    with pytest.raises(TypeError):
        e @ epg.S(1)
    with pytest.raises(TypeError):
        epg.E(1, 2, 3) @ epg.P(1, 0.1)      # opscalar.py:90-91: E only combines with E


# ------------------------------------------------------------------ first-order derivatives (host side)
def test_order1_parsing_matches_reference_forms():
    """the normalised {variable: {parameter: coeff}} forms of diff.py:153-198"""
    assert epg.T(30, 0, order1=True).order1 == {"alpha": {"alpha": 1}, "phi": {"phi": 1}}
    assert epg.T(30, 0, order1="alpha").order1 == {"alpha": {"alpha": 1}}
    assert epg.E(5, 1000, 100, order1=["T1", "T2"]).order1 == {"T1": {"T1": 1}, "T2": {"T2": 1}}
    assert epg.E(5, 1000, 100, order1={"R2": "T2"}).order1 == {"R2": {"T2": 1}}
    assert epg.T(30, 0, order1={"B1": {"alpha": 30}}).order1 == {"B1": {"alpha": 30}}
    assert epg.T(30, 0).order1 == {}
    with pytest.raises(ValueError):
        epg.T(30, 0, order1="T2")
    with pytest.raises(ValueError):
        epg.P(5, 0.1, order1=3)
    # second order (diff.py:201-262)
    e = epg.E(5, 1000, 100, order2="T2")
    assert e.order1 == {"T2": {"T2": 1}} and e.order2 == {("T2", "T2"): {}} and e.auto_cross_derivatives
    e = epg.E(5, 1000, 100, order1=["T1", "T2"], order2=[("T2", "T1")])
    assert e.order2 == {("T1", "T2"): {}} and e.parameters_order2 == set() and not e.auto_cross_derivatives
    assert epg.T(30, 0, order2=True).parameters_order2 == {("alpha", "alpha"), ("alpha", "phi"), ("phi", "phi")}
    with pytest.raises(ValueError):
        epg.E(5, 1000, 100, order1="T1", order2=[("T2", "g")])       # no variable of the pair in order1
    assert epg.Hessian("T2").variables1 == epg.Hessian("T2").variables2 == ["T2"]


def test_partial_tables_against_finite_differences():
    """dOp/dparam tables shipped to the device = central differences of the operator tables"""
    from epgpy_amd import opscalar, diff
    h = 1e-5
    op = epg.E(5.0, np.array([900.0, 1200.0]), 70.0, 0.013, order1=True)
    tabs = op._variable_tables()
    for name, idx in (("tau", 0), ("T1", 1), ("T2", 2), ("g", 3)):
        args = [5.0, np.array([900.0, 1200.0]), 70.0, 0.013]
        up, dn = list(args), list(args)
        up[idx], dn[idx] = args[idx] + h, args[idx] - h
        fd = (opscalar.pack_scalar(epg.E(*up).arr, epg.E(*up).arr0)[1]
              - opscalar.pack_scalar(epg.E(*dn).arr, epg.E(*dn).arr0)[1]) / (2 * h)
        np.testing.assert_allclose(tabs[name], fd, rtol=1e-6, atol=1e-9)
    t = epg.T(np.array([30.0, 75.0]), 20.0, order1=True)
    tabs = t._variable_tables()
    for name, idx in (("alpha", 0), ("phi", 1)):
        args = [np.array([30.0, 75.0]), 20.0]
        up, dn = list(args), list(args)
        up[idx], dn[idx] = args[idx] + h, args[idx] - h
        fd = (diff.pack_matrix_partial(epg.T(*up).mat) - diff.pack_matrix_partial(epg.T(*dn).mat)) / (2 * h)
        np.testing.assert_allclose(tabs[name], fd, rtol=1e-6, atol=1e-9)
    # combination over parameters and a per-voxel coefficient (append-axes broadcasting)
    c = np.array([2.0, 3.0])
    t2 = epg.T(np.array([30.0, 75.0]), 20.0, order1={"x": {"alpha": c, "phi": -1.0}})
    np.testing.assert_allclose(t2._variable_tables()["x"], tabs["alpha"] * c[:, None] - tabs["phi"], rtol=0, atol=1e-15)


def test_derivative_plan_arrays():
    """dops stay index-aligned with ops; every ADC owns 1 + n_vars signal rows"""
    from epgpy_amd import functions
    seq = [epg.T(np.array([20.0, 30.0]), 0, order1={"fa": "alpha"}), epg.E(5, 1000, 80, order1=["T2"]),
           epg.ADC, epg.S(1), epg.T(15, 0), epg.ADC]
    enc, records, _ = functions.compile_sequence(seq, [epg.Jacobian(["fa", "T2"])], variables=["T2", "fa"], fuse=False)
    ops, grid, spaces, coef, dops = enc.arrays()
    assert enc.n_adc == 6 and [slot for _, slots in records for _, slot in slots] == [0, 3]
    assert dops.shape == ops.shape
    assert dops["coef_off"][0].tolist() == [-1, dops["coef_off"][0][1], -1] and dops["coef_off"][0][1] >= 0
    assert dops["space"][0].tolist() == [-1, 0, -1]          # per-voxel flip angle -> index space 0
    assert dops["coef_off"][1][0] >= 0 and dops["space"][1][0] == -1   # scalar E: broadcast entry
    assert (dops["coef_off"][2:] == -1).all()
    assert functions._jacobian_variables(seq, [epg.Jacobian(["magnitude", "T2", "nope", "fa"])]) == ["T2", "fa"]
    # fused (the default): E . T is ONE operator whose table AND partials the library generates (epgx_fuse / epgx_fuse_partial);
    # the partials of the parts stay in the host part of the pool as the recipes' sources
    enc, records, _ = functions.compile_sequence(seq, [epg.Jacobian(["fa", "T2"])], variables=["T2", "fa"])
    ops, grid, spaces, coef, dops = enc.arrays()
    fuse, fpart = enc.fuse_array(), enc.fuse_partial_array()
    assert ops["opcode"][0] == _lib.OP_T0 and len(ops) == 5 and len(fuse) == 1 and len(fpart) == 2
    assert ops["coef_off"][0] == fuse["dst_off"][0] >= coef.size
    assert sorted(dops["coef_off"][0][:2].tolist()) == sorted(fpart["dst_off"].tolist()) and dops["coef_off"][0][2] == -1
    assert (fpart["dst_off"] >= coef.size).all() and (fpart["src_off"] == fuse["src_off"][0]).all() and (fpart["e_off"] == fuse["e_off"][0]).all()
    by_var = {int(d): (int(a), int(b)) for d, a, b in zip(fpart["dst_off"], fpart["dsrc_off"], fpart["de_off"])}
    t2, fa = by_var[int(dops["coef_off"][0][0])], by_var[int(dops["coef_off"][0][1])]
    assert t2[0] == -1 and 0 <= t2[1] < coef.size            # d/dT2: only the relaxation depends on it
    assert 0 <= fa[0] < coef.size and fa[1] == -1            # d/dfa: only the rotation
    assert enc.generated_size == 2 * 12 + 2 * 2 * 14         # two voxels: one 12-coefficient table, two 14-coefficient partials
    assert (dops["coef_off"][1:] == -1).all()


def test_differential_operator_guards():
    with pytest.raises(NotImplementedError):
        epg.T(30, 0, order1=True) @ epg.T(20, 0)


def test_more_broadcast_patterns_than_index_spaces():
    """a 3-D grid offers 7 broadcast patterns, the kernel 4 index spaces: later patterns borrow a
    space that varies along a superset of their axes (materialised table), last resort = dense grid"""
    from epgpy_amd import plan
    grid = (2, 3, 4)
    enc = plan.Encoder(grid)
    rng = np.random.default_rng(0)
    shapes = [(2,), (1, 3), (1, 1, 4), (2, 3), (2, 1, 4), (1, 3, 4), (2, 3, 4)]
    tabs = [rng.random(sh + (4,)) for sh in shapes]
    entries = [enc._table(t, None) for t in tabs]
    assert len(enc.spaces) <= _lib.MAX_SPACES
    pool = np.concatenate(enc.pool)
    vox = np.indices(grid).reshape(3, -1).T
    for tab, (space, off, ncoef) in zip(tabs, entries):
        strides = enc.spaces[space]
        full = np.broadcast_to(tab.reshape(tab.shape[:-1] + (1,) * (3 - (tab.ndim - 1)) + (4,)), grid + (4,))
        for v in vox:
            idx = int(np.dot(v, strides))
            np.testing.assert_array_equal(pool[off + idx * ncoef: off + (idx + 1) * ncoef], full[tuple(v)])


# ------------------------------------------------------------------ E . T . E fusion (host side)
def _t0_as_affine(table):
    """[..., 12] EPGX_OP_T0 table -> (3x3 matrix, constant term) in the (F, conj F-, Z) basis"""
    m00, m01, m02 = table[..., 0], table[..., 1] + 1j * table[..., 2], table[..., 3] + 1j * table[..., 4]
    m20, m22 = table[..., 5] + 1j * table[..., 6], table[..., 7]
    mat = np.empty(table.shape[:-1] + (3, 3), complex)
    mat[..., 0, 0], mat[..., 0, 1], mat[..., 0, 2] = m00, m01, m02
    mat[..., 1, 0], mat[..., 1, 1], mat[..., 1, 2] = m01.conj(), m00, m02.conj()
    mat[..., 2, 0], mat[..., 2, 1], mat[..., 2, 2] = m20, m20.conj(), m22
    o0 = table[..., 8] + 1j * table[..., 9]
    off = np.stack([o0, o0.conj(), table[..., 10] + 0j], axis=-1)
    return mat, off


def test_fusion_algebra_and_structure():
    from epgpy_amd import fusion, functions
    rng = np.random.default_rng(7)
    T1, T2 = rng.uniform(300, 2000, (4, 1)), rng.uniform(30, 200, (1, 5))
    e1, e2 = epg.E(5.0, T1, T2), epg.E(3.0, T1, T2)
    rf = epg.T(rng.uniform(20, 160, (4, 1)), 35.0)
    sh = epg.S(1)
    out = fusion.fuse_sequence([e1, rf, sh, sh, e2, epg.ADC, sh, e1, rf, sh, sh, e2, epg.ADC])
    kinds = [type(o).__name__ for o in out]
    assert kinds == ["FusedTE", "S", "S", "Adc", "S", "FusedTE", "S", "S", "Adc"]
    assert out[0] is out[5]                                  # one fused object (one table) for the repeated triple
    mat, off = _t0_as_affine(out[0].host_table())
    d1 = np.broadcast_to(e1.arr, (4, 5, 3))[..., None] * np.eye(3)
    d2 = np.broadcast_to(e2.arr, (4, 5, 3))[..., None] * np.eye(3)
    tm = np.broadcast_to(rf.mat, (4, 5, 3, 3))
    np.testing.assert_allclose(mat, d2 @ tm @ d1, rtol=0, atol=1e-15)
    z = np.array([0, 0, 1.0])
    r1 = np.broadcast_to(e1.arr0, (4, 5, 3)) * z
    r2 = np.broadcast_to(e2.arr0, (4, 5, 3)) * z
    expect = np.einsum("...ij,...j->...i", d2 @ tm, r1) + r2
    np.testing.assert_allclose(off, expect, rtol=0, atol=1e-15)
    # no fusion: precession, probes / spoilers in between, index spaces that do not nest, derivatives
    assert [type(o).__name__ for o in fusion.fuse_sequence([epg.E(5, 1000, 50, 0.1), rf])] == ["E", "T"]
    assert [type(o).__name__ for o in fusion.fuse_sequence([rf, epg.ADC, e2])] == ["T", "Adc", "E"]
    assert [type(o).__name__ for o in fusion.fuse_sequence([rf, sh, epg.SPOILER, e2])] == ["T", "S", "Spoiler", "E"]
    b1 = epg.T(np.array([60.0, 70.0, 80.0])[None, None, :], 0)
    assert [type(o).__name__ for o in fusion.fuse_sequence([e1, b1, e1])] == ["E", "T", "E"]
    assert [type(o).__name__ for o in fusion.fuse_sequence([epg.E(5, 1000, 50, order1="T2"), rf])] == ["E", "T"]
    # the compiled plan: one T0 record per fused triple, probes and bounds unchanged
    seq = [epg.T(90, 90)] + [sh, e1, rf, sh, e1, epg.ADC] * 3
    enc, records, bounds = functions.compile_sequence(seq)
    codes = [r[0] for r in enc.records]
    # (the relaxation behind the excitation pulse belongs to the first echo: E . T . E like all the others -- one table, one
    # repeated record -- and the pulse stays a plain rotation)
    assert codes.count(_lib.OP_T0) == 3 and codes.count(_lib.OP_T) == 1 and codes.count(_lib.OP_E) == 0 and codes.count(_lib.OP_ADC) == 3
    fused = [o for o in fusion.fuse_sequence(functions.flatten_sequence(seq)) if isinstance(o, fusion.FusedTE)]
    assert len(fused) == 3 and fused[0] is fused[1] is fused[2]
    # ... only where that gives the table of a later echo: another relaxation behind the pulse stays with the pulse
    other = [epg.T(90, 90), sh, epg.E(2.5, 1000, 50), rf, sh, e1, epg.ADC] + [sh, e1, rf, sh, e1, epg.ADC] * 2
    kinds = [type(o).__name__ for o in fusion.fuse_sequence(functions.flatten_sequence(other))]
    assert kinds[:3] == ["FusedTE", "S", "FusedTE"] and kinds.count("FusedTE") == 4
    assert len({r[4] for r in enc.records if r[0] == _lib.OP_T0}) == 1      # E.T.E (shared by all echoes)
    # the tables are generated on the device: the host pool holds the sources only, the recipes travel
    ops, grid, spaces, coef, _ = enc.arrays()
    fuse = enc.fuse_array()
    assert len(fuse) == 2 and enc.generated_size == 2 * 20 * 12      # the triple and its intermediate T.E
    assert coef.size == 8 + 4 * 8 + 20 * 4                           # exc, rf (one entry per T1 row), E
    assert (fuse["dst_off"] >= coef.size).all() and (ops["coef_off"][ops["opcode"] == _lib.OP_T0] >= coef.size).all()
    enc2, _, bounds2 = functions.compile_sequence(seq, fuse=False)
    assert [r[0] for r in enc2.records].count(_lib.OP_E) == 6 and len(bounds) == len(bounds2) == 3


def test_generated_tables_cover_their_sources():
    """a fused table must vary along every grid axis its sources vary along -- also when the plan ran
    out of index spaces and a source table was materialised into a borrowed, larger space (found by
    tools/stress_fuzz.py: epgx_plan_create rejected such a plan)"""
    from epgpy_amd import functions
    from tests import sequences as sq
    checked = 0
    for seed in range(1000, 1400):
        rng = np.random.default_rng(7000 + seed)
        grid = tuple(int(x) for x in rng.integers(1, 6, rng.integers(1, 4)))
        rng.integers(0, 5)
        tuples = sq.random_sequence(rng, grid, nops=int(rng.integers(20, 80)), precession=False)
        enc, _, _ = functions.compile_sequence(sq.to_ops(epg, tuples), shape=grid)
        enc.arrays()
        for f in enc.fuse_array():
            dst = enc.spaces[f["dst_space"]] if f["dst_space"] >= 0 else (0,) * len(enc.grid)
            for sp in (f["src_space"], f["e_space"]):
                if sp >= 0:
                    assert all(b != 0 or a == 0 or g == 1 for a, b, g in zip(enc.spaces[sp], dst, enc.grid)), (seed, grid)
                    checked += 1
    assert checked > 1000


def test_assembled_tables_reproduce_the_uploaded_ones():
    """relaxation tables over several grid axes are shipped as per-axis columns and assembled on the device
    (epgx_assemble): a NumPy restatement of the recipe gives the dense packed table bit for bit, the host part of the
    pool shrinks from the table to its columns, and fused E.T.E recipes take the assembled table as their source"""
    from epgpy_amd import functions, opscalar

    T1 = np.linspace(200, 3000, 96)[:, None]
    T2 = np.linspace(20, 300, 80)[None, :]
    g = np.linspace(-0.01, 0.01, 80)[None, :]
    ops = {"E": epg.E(5.0, T1, T2), "Eg": epg.E(5.0, T1, T2, g), "P": epg.P(3.0, g * np.ones((96, 1))),
           "R": epg.R(0.01 + 0.1j * T2 / 300, 5.0 / T1, r0=5.0 / T1)}
    for name, op in ops.items():
        seq = [epg.T(30, 10), op, epg.ADC]
        enc, _, _ = functions.compile_sequence(seq, fuse=False)
        ops_arr, grid, spaces, coef, _ = enc.arrays()
        asm = enc.assemble_array()
        dense = opscalar.pack_scalar(op.arr, op.arr0)[1]
        if name == "P":          # P over a full-grid g: its only group is as large as the table -> uploaded
            assert len(asm) == 0 and coef.size >= dense.size
            continue
        assert len(asm) == 1 and coef.size < dense.size // 10, (name, coef.size)
        rec = asm[0]
        assert rec["ncoef"] == 4 and rec["dst_off"] == coef.size and enc.generated_size == dense.size
        table = np.empty(tuple(grid) + (4,))
        coords = np.indices(tuple(grid))
        for c in range(4):
            src = rec["src"][rec["col_src"][c]]
            index = sum(coords[d] * src["strides"][d] for d in range(len(grid)))
            table[..., c] = coef[src["off"] + index * src["ncol"] + rec["col_idx"][c]]
        assert np.array_equal(table, np.broadcast_to(dense, table.shape)), name
        e_rec = ops_arr[1]
        assert e_rec["opcode"] == _lib.OP_E and e_rec["coef_off"] == rec["dst_off"] and e_rec["space"] == rec["dst_space"]
    # fused: the E.T.E recipe reads the assembled table
    seq = sq.mse_ops(epg, T1, T2, necho=3)
    enc, _, _ = functions.compile_sequence(seq, options={"max_nstate": 63})
    _, _, _, coef, _ = enc.arrays()
    asm, fuses = enc.assemble_array(), enc.fuse_array()
    assert len(asm) == 1 and len(fuses) >= 1 and coef.size < 2000
    assert all(f["e_off"] == asm[0]["dst_off"] for f in fuses)
    # small grids keep the plain upload
    small = functions.compile_sequence(sq.mse_ops(epg, T1[:8], T2[:, :8], necho=2), options={"max_nstate": 63})[0]
    small.arrays()
    assert len(small.assemble_array()) == 0


# ------------------------------------------------------------------ boundary: struct layout in three places
_C_TYPES = {"uint32_t": "c_uint32", "int32_t": "c_int32", "int64_t": "c_int64"}


def _header_plan_desc_fields():
    """[(name, ctypes type name)] of `struct epgx_plan_desc`, parsed from include/epgx.h"""
    text = open(os.path.join(ROOT, "include", "epgx.h")).read()
    body = re.search(r"typedef struct epgx_plan_desc \{(.*?)\} epgx_plan_desc;", text, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        m = re.match(r"(const )?(\w+) (\*?)(\w+)$", decl)
        assert m, decl
        fields.append((m.group(4), "c_void_p" if m.group(3) else _C_TYPES[m.group(2)]))
    return fields


def test_plan_desc_matches_header_and_docs():
    """epgx_plan_desc as the header declares it == _lib.PlanDesc == the stub INTEGRATION.md shows a maintainer
    (round 2 shipped a documented stub that was 8 bytes short of the header)"""
    import ctypes

    header = _header_plan_desc_fields()
    assert header[0] == ("struct_size", "c_uint32")
    # (ctypes.c_uint32 is an alias of c_uint on this platform: compare the type objects, not their names)
    assert [(name, typ) for name, typ in _lib.PlanDesc._fields_] == [(name, getattr(ctypes, typ)) for name, typ in header]
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    stub = re.search(r"class PlanDesc\(ctypes\.Structure\):.*?_fields_ = \[(.*?)\]\n", doc, re.S).group(1)
    stub = re.sub(r"#[^\n]*", "", stub)
    assert re.findall(r'\("(\w+)", ctypes\.(\w+)\)', stub) == header
    # natural alignment, as the C compiler lays the struct out
    names = [n for n, _ in header]
    assert ctypes.sizeof(_lib.PlanDesc) == 128 and _lib.PlanDesc.fuse_partial.offset == 120, (ctypes.sizeof(_lib.PlanDesc), names)


def test_every_declared_entry_point_is_bound_with_the_declared_arity():
    """include/epgx.h prototypes vs _lib.SYMBOLS: same names, same number of arguments"""
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "epgx.h")).read(), flags=re.S)
    protos = re.findall(r"^(?:int|const char \*)\s*(epgx_\w+)\(([^;]*?)\);", text, re.M | re.S)
    assert len(protos) >= 43
    for name, args in protos:
        nargs = 0 if args.strip() == "void" else len(args.split(","))
        assert name in _lib.SYMBOLS, name
        assert len(_lib.SYMBOLS[name][1]) == nargs, (name, nargs, len(_lib.SYMBOLS[name][1]))
    assert set(_lib.SYMBOLS) == {name for name, _ in protos}


# ------------------------------------------------------------------ simulate(ngpu=N): slab and offset arithmetic
@pytest.mark.parametrize("ngpu", [1, 2, 3, 8])
def test_ngpu_slabs_cover_the_grid_once(ngpu):
    from epgpy_amd import functions

    for nvox in (1, 7, 64, 1000, 1024 * 1024, 10 ** 6):
        slab, bounds = functions.slab_bounds(nvox, ngpu)
        assert len(bounds) == ngpu and slab * ngpu >= nvox
        covered = np.zeros(nvox, dtype=np.int32)
        for v0, cnt in bounds:
            assert 0 <= cnt <= slab and 0 <= v0 <= nvox
            covered[v0: v0 + cnt] += 1
        assert (covered == 1).all()                                    # every voxel on exactly one GPU
        assert [v0 for v0, _ in bounds] == sorted(v0 for v0, _ in bounds)
        # host array [n_adc][nvox]: GPU g's columns are [v0, v0 + cnt) -- byte offsets of its first row
        assert all(16 * v0 + 16 * cnt <= 16 * nvox for v0, cnt in bounds)


def test_device_list_of_simulate():
    from epgpy_amd import functions

    assert functions._device_list(None, None) == [_lib.default_device()]
    assert functions._device_list(2, None) == [2]
    assert functions._device_list(None, 3) == [_lib.default_device() + g for g in range(3)]
    assert functions._device_list(1, 2) == [1, 2]
    assert functions._device_list([3, 1], None) == [3, 1]
    assert functions._device_list([3, 1], 2) == [3, 1]
    for bad in (dict(device=[0, 0], ngpu=None), dict(device=[0, 1], ngpu=3), dict(device=None, ngpu=0)):
        with pytest.raises(ValueError):
            functions._device_list(**bad)


def test_ngpu_needs_the_resident_path():
    seq = sq.mse_ops(epg, 1000.0, [50.0, 80.0], necho=2)
    for kw in (dict(mode="stream"), dict(callback=lambda sm: None)):
        with pytest.raises((NotImplementedError, ValueError, _lib.EpgxError)):
            epg.simulate(seq, ngpu=2, **kw)


def test_bench_crash_guard_prints_the_held_line_only_when_the_parent_ends_silent():
    """bench.py, N > 1, rank 0: the child forked before the GPU is touched prints the provisional line (headline measured,
    strong-scaling legs marked unfinished) if the rank dies inside those legs, and nothing if the rank printed its own"""
    import bench

    for printed_own in (False, True):
        rd, wr = os.pipe()
        guard = bench.crash_guard(wr)            # (the child keeps `wr` as its stdout until it exits)
        os.close(wr)
        os.write(guard, b'P {"value": 1}\n')
        os.write(guard, b'P {"value": 2, "strong_mrf_100": {"error": "unfinished"}}\n')
        if printed_own:
            os.write(guard, b"F\n")
        os.close(guard)                          # the rank is gone
        seen = b""
        while True:
            chunk = os.read(rd, 4096)
            if not chunk:
                break
            seen += chunk
        os.close(rd)
        assert seen == (b"" if printed_own else b'{"value": 2, "strong_mrf_100": {"error": "unfinished"}}\n')


def _run_bench(args, env_extra=None, launcher=()):
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "EPGX_BENCH_LAUNCHER")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, *launcher, os.path.join(root, "bench.py"), *args], env=env, cwd=root, capture_output=True,
                          text=True, timeout=180)


@pytest.mark.parametrize("n", [2, 3])
def test_bench_starts_its_own_ranks_when_no_launcher_did(n):
    """`python bench.py --gpus N` started plainly (the way the driver starts the N = 1 run): the process launches N rank
    processes itself (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment), relays rank 0's ONE line and returns 0.
    The stub leg stands in for the GPU work: rendezvous, barriers, max over ranks and the line are the real ones"""
    import json

    done = _run_bench(["--gpus", str(n), "--steps", "3", "--stub-leg"])
    assert done.returncode == 0, done.stderr[-2000:]
    lines = [ln for ln in done.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, done.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == n and line["launcher"] == "self" and line["steps"] == 3
    # the slowest rank sets the time (rank r sleeps (1 + r) ms per step)
    assert line["ms_per_step"] >= n * 0.9


def test_bench_self_launcher_fails_when_a_rank_fails():
    """a rank that dies takes the run down with a non-zero exit code within seconds (the other ranks are stopped, nobody waits
    for a rendezvous that cannot complete) and no line is printed"""
    import time

    t0 = time.time()
    done = _run_bench(["--gpus", "3", "--steps", "3", "--stub-leg", "--stub-fail-rank", "1"])
    assert done.returncode == 3 and not done.stdout.strip()
    assert "rank 1 exited with 3" in done.stderr
    assert time.time() - t0 < 60


def test_bench_under_torchrun_is_unchanged():
    """the documented N > 1 form: torch.distributed.run provides the ranks, bench.py must not launch a second set"""
    import json
    import socket

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    done = _run_bench(["--gpus", "2", "--steps", "3", "--stub-leg"],
                      launcher=("-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                                "--master-port", str(port)))
    assert done.returncode == 0, done.stderr[-2000:]
    lines = [ln for ln in done.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["launcher"] == "torchrun"


def test_plan_arrays_are_built_once_under_concurrent_callers():
    """Encoder.plan_arrays mutates the encoder (the deferred n-D shift / diffusion tables join the pool, records are
    rewritten): threads that ask at the same time -- one per GPU in simulate(ngpu=N) -- must all receive the ONE set of arrays
    a single-threaded build produces.  A sleep inside _table forces the interleaving that used to pool the tables three times"""
    import threading
    import time

    from epgpy_amd import functions, workloads as wl

    def encoder():
        seq, _, _, opts = wl.build(epg, "pgse_512")
        enc, _, _ = functions.compile_sequence(seq, None, options=opts)
        return enc

    ref = encoder()
    want = ref.plan_arrays(ref.packable_nd() or ref.capacity())
    enc = encoder()
    K = enc.packable_nd() or enc.capacity()
    slow = enc._table

    def slow_table(table, key):
        time.sleep(0.01)
        return slow(table, key)

    enc._table = slow_table
    got = [None] * 3

    def work(i):
        got[i] = enc.plan_arrays(K)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert got[0] is got[1] is got[2]
    assert got[0]["coef"].size == want["coef"].size and got[0]["coef"].tobytes() == want["coef"].tobytes()   # (gather tables are int32 pairs: compare bits)
    assert np.array_equal(got[0]["ops"], want["ops"])


def test_relaxation_partial_columns_equal_the_sliced_table():
    """E._partial_column_groups evaluates the partial's column groups directly on the parameters they depend on; the full
    [*grid, 4] partial table -- what the device would otherwise be sent -- has exactly these values in those places"""
    rng = np.random.default_rng(5)
    T1, T2 = rng.uniform(300, 3000, 7)[:, None, None], rng.uniform(20, 300, 5)[None, :, None]
    g = rng.uniform(-0.02, 0.02, 3)[None, None, :]
    tau3 = rng.uniform(2, 9, 3)[None, None, :]
    cases = [epg.E(5.0, T1, T2, order1=["T1", "T2"]),
             epg.E(5.0, T1, T2, g, order1={"a": {"T2": 2.0, "T1": -0.5}, "b": {"g": 1.0}, "c": {"tau": 3.0}}),
             epg.E(tau3, T1, T2, order1=["tau", "T1", "T2"]),
             epg.E(5.0, T1[:, 0, 0], 80.0, order1=["T1", "T2"]),
             epg.E(4.0, 900.0, T2[0, :, 0], 0.01, order1=["T2", "g"])]
    for op in cases:
        for var in op.order1:
            cols = op._partial_column_groups(var)
            assert cols is not None, (op.name, var)
            groups, columns = cols
            full = op._variable_tables()[var]
            rebuilt = np.empty_like(full)
            for c, (gi, j) in enumerate(columns):
                rebuilt[..., c] = np.broadcast_to(groups[gi][..., j], full.shape[:-1])
            assert np.array_equal(rebuilt, full), (op.name, var)
            lead, imaginary = op._partial_shape_facts(var)
            assert lead == full.shape[:-1] and (imaginary or not np.any(full[..., 1] != 0.0))
