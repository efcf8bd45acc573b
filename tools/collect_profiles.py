#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/prof_<tag>[_<workload>]_<mode>/...) into the small, tracked files under
profiles/:  <tag>[_<workload>]_<mode>_kernel_stats.csv (verbatim --stats summary), ..._pmc.csv (per-counter mean over
the dispatches of the mode's dominant kernel: epgx::rows_kernel for state-resident launches, epgx::run_kernel<..., true>
for the read+write per-timestep launches) and the entry of profiles/traffic.json that bench.py reads -- with the kernel
name and the hash of the device sources it was measured on (bench.py marks the figures `pmc_stale` when they differ).

    python3 tools/collect_profiles.py <tag> [out_tag] [workload] [mode ...]
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import csrc_hash  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
out_tag = sys.argv[2] if len(sys.argv) > 2 else tag
workload = sys.argv[3] if len(sys.argv) > 3 else "mse_1024"
modes = sys.argv[4:] or ["resident", "stream"]
suffix = "" if workload == "mse_1024" else f"_{workload}"
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)


def dominant(mode, name):
    if mode == "resident":
        return "rows_kernel" in name or "rows_grow_kernel" in name
    return re.search(r"run_kernel<\d+, \d+, true>", name) is not None   # HAS_IN = true: the read+write launches


for mode in modes:
    base = os.path.join(ROOT, "gpurun_out", f"prof_{tag}{suffix}_{mode}")
    stats = glob.glob(os.path.join(base, "trace", "*", "*_kernel_stats.csv"))
    if not stats:
        continue
    shutil.copy(stats[0], os.path.join(ROOT, "profiles", f"{out_tag}{suffix}_{mode}_kernel_stats.csv"))
    avg_ns = None
    for r in csv.DictReader(open(stats[0])):
        if dominant(mode, r["Name"]):
            avg_ns = float(r["AverageNs"])
            break
    rows_out, kernel = [], None
    for sub in ("pmc_sq1", "pmc_sq2", "pmc_fetch", "pmc_write", "pmc_mix"):
        for f in glob.glob(os.path.join(base, sub, "*", "*_counter_collection.csv")):
            agg = collections.defaultdict(list)
            meta = {}
            for r in csv.DictReader(open(f)):
                if dominant(mode, r["Kernel_Name"]):
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                    meta = {k: r[k] for k in ("Kernel_Name", "Grid_Size", "Workgroup_Size", "VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size")}
            for name, vals in agg.items():
                kernel = meta["Kernel_Name"]
                rows_out.append({"pass": sub, "counter": name, "dispatches": len(vals), "mean": sum(vals) / len(vals),
                                 "min": min(vals), "max": max(vals), **meta})
    if not rows_out:
        continue
    with open(os.path.join(ROOT, "profiles", f"{out_tag}{suffix}_{mode}_pmc.csv"), "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows_out[0].keys()))
        w.writeheader()
        w.writerows(rows_out)
    print(mode, "->", len(rows_out), "counters", kernel)
    # HBM bytes per launch from the two separate PMC passes.  MI355X_MICROARCH.md (HBM section):
    # FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes
    # of a wide coalesced streaming read, WRITE_SIZE is exact for 16-B-per-lane streaming stores.
    vals = {r["counter"]: r["mean"] for r in rows_out}
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
        entry = {
            "read_bytes": 2.0 * vals["FETCH_SIZE"] * 1024, "write_bytes": vals["WRITE_SIZE"] * 1024,
            "bytes": 2.0 * vals["FETCH_SIZE"] * 1024 + vals["WRITE_SIZE"] * 1024,
            "kernel": kernel, "csrc_hash": csrc_hash(), "rocprof_avg_launch_ms": None if avg_ns is None else avg_ns * 1e-6,
            "source": f"profiles/{out_tag}{suffix}_{mode}_pmc.csv (FETCH_SIZE x2 x1024 + WRITE_SIZE x1024, mean per launch)"}
        if "SQ_INSTS_VALU_FMA_F64" in vals:   # executed fp64 operations: wave instructions x 64 lanes, an FMA counts 2
            entry["fp64_flop_executed"] = 64.0 * (2 * vals["SQ_INSTS_VALU_FMA_F64"] + vals["SQ_INSTS_VALU_MUL_F64"]
                                                  + vals.get("SQ_INSTS_VALU_ADD_F64", 0.0))
            entry["fp64_instructions"] = {k: vals.get("SQ_INSTS_VALU_" + k + "_F64") for k in ("FMA", "MUL", "ADD")}
            entry["valu_instructions"] = vals.get("SQ_INSTS_VALU")
        if "GRBM_GUI_ACTIVE" in vals:
            entry["grbm_gui_active"] = vals["GRBM_GUI_ACTIVE"]
        traffic.setdefault(workload, {})[mode] = entry
        json.dump(traffic, open(tpath, "w"), indent=1, sort_keys=True)
