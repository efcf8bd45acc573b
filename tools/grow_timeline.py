"""Where a wavefront of rows_grow_kernel<1> spends its time on the C2-L launch: shader-clock stamps per voxel group and phase
(measurement build: tools/build_timing_variant.sh) -> mean cycles per segment, voxel groups in flight per SIMD, gaps between
the groups of a wave slot.  The stamps cost about 4 % of the launch.
    tools/build_timing_variant.sh && EPGX_LIBRARY=$PWD/epgpy_amd/csrc/variants/libepgx_timing.so python tools/grow_timeline.py    (GPU box)
"""
import ctypes, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epgpy_amd import epg, _lib, functions, workloads as wl

T1, T2 = wl.grid_parameters("mse_1024")
ctx = _lib.get_context(None)
seq = wl.mse_sequence(epg, T1, T2, necho=20)
enc, _, _ = functions.compile_sequence(seq, None, options={"max_nstate": 63})
plan = enc.device_plan(ctx, 64)
sig = _lib.DeviceBuffer(ctx, 16 * enc.n_adc * enc.nvox)
run = lambda: _lib.run(ctx, plan, 0, plan.n_ops, 0, enc.nvox, None, None, 64, sig.ptr.value, enc.nvox, 0)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.05:
    run(); ctx.synchronize()
ctx.timer_start()
for _ in range(10): run()
ms = ctx.timer_stop() / 10
run(); ctx.synchronize()
ngroups = enc.nvox // 4
raw = np.zeros(ngroups * 8, dtype=np.uint64)
lib = ctypes.CDLL(_lib.library_path())
lib.epgx_dbg_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
rc = lib.epgx_dbg_stamps(raw.ctypes.data, raw.size)
assert rc == 0, rc
st = raw.reshape(ngroups, 8).astype(np.int64)
t = st[:, :7]
hw = (st[:, 7] >> 32) & 0xffffffff
blk = st[:, 7] & 0xffffffff
names = ["prologue (indices, record 0, first line)", "phase R=1", "widen 1->2", "phase R=2", "widen 2->4", "phase R=4 (+ stamp)"]
seg = np.diff(t, axis=1)
total = t[:, 6] - t[:, 0]
span = t[:, 6].max() - t[:, 0].min()
out = {"ms_per_launch": round(ms, 4), "groups": int(ngroups), "kernel_span_cycles": int(span),
       "clock_GHz_if_span_is_launch": round(span / (ms * 1e6), 3),
       "group_cycles_mean": round(float(total.mean()), 1), "group_cycles_p10_p50_p90": [int(x) for x in np.percentile(total, [10, 50, 90])]}
for i, n in enumerate(names):
    out[n] = {"mean": round(float(seg[:, i].mean()), 1), "p10_p50_p90": [int(x) for x in np.percentile(seg[:, i], [10, 50, 90])],
              "share": round(float(seg[:, i].mean() / total.mean()), 3)}
# residency: per SIMD (xcc, se, cu, simd), how many groups are in flight over time
wave = hw & 0xf
simd = (hw >> 4) & 3
cu = (hw >> 8) & 0xf
sh = (hw >> 12) & 1
se = (hw >> 13) & 7
key = (se * 2 + sh) * 16 * 4 + cu * 4 + simd      # (per XCC the ids repeat: XCC_ID is another register; blocks go round-robin over 8 XCCs)
xcc = blk % 8
key = xcc * 4096 + key
uniq, inv = np.unique(key, return_inverse=True)
out["simds_seen"] = int(uniq.size)
out["groups_per_simd_mean"] = round(ngroups / uniq.size, 1)
# busy time per SIMD = union of [t0, t6] of its groups; occupancy = sum of durations / union
t_begin, t_end = t[:, 0].min(), t[:, 6].max()
occ, cover = [], []
for s in range(min(uniq.size, 256)):
    idx = np.nonzero(inv == s)[0]
    ev = np.concatenate([np.stack([t[idx, 0], np.ones(idx.size, np.int64)], 1), np.stack([t[idx, 6], -np.ones(idx.size, np.int64)], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    lvl = np.cumsum(ev[:, 1])
    dt = np.diff(ev[:, 0])
    hist = np.bincount(lvl[:-1].clip(0, 8), weights=dt, minlength=9)
    occ.append(hist / max(dt.sum(), 1))
    cover.append((ev[-1, 0] - ev[0, 0]) / (t_end - t_begin))
occ = np.array(occ).mean(0)
out["share_of_time_with_n_groups_in_flight_per_simd"] = {str(n): round(float(v), 3) for n, v in enumerate(occ)}
out["simd_active_span_over_kernel_span"] = round(float(np.mean(cover)), 3)
# start-to-start gap between consecutive groups in the same wave slot (same simd + wave id): launch gap when the block changes
slot = key * 16 + wave
order = np.lexsort((t[:, 0], slot))
s_sorted, t0s, t6s, b_sorted = slot[order], t[order, 0], t[order, 6], blk[order]
same = s_sorted[1:] == s_sorted[:-1]
gap = (t0s[1:] - t6s[:-1])[same]
newblk = (b_sorted[1:] != b_sorted[:-1])[same]
out["gap_cycles_between_groups_same_wave"] = {"mean": round(float(gap[~newblk].mean()), 1) if (~newblk).any() else None, "n": int((~newblk).sum())}
out["gap_cycles_between_blocks_same_slot"] = {"mean": round(float(gap[newblk].mean()), 1) if newblk.any() else None,
                                              "p10_p50_p90": [int(x) for x in np.percentile(gap[newblk], [10, 50, 90])] if newblk.any() else None, "n": int(newblk.sum())}
print(json.dumps(out, indent=1))
