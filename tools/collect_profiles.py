#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/prof_<tag>_<mode>/...) into the small, tracked files
under profiles/:  <tag>_<mode>_kernel_stats.csv (verbatim --stats summary) and
<tag>_<mode>_pmc.csv (per-counter mean over the dispatches of the mode's kernel: epgx::rows_kernel for state-resident launches,
epgx::run_kernel for per-timestep launches)."""
import collections
import csv
import glob
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
out_tag = sys.argv[2] if len(sys.argv) > 2 else tag
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
for mode in ("resident", "stream"):
    base = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_{mode}")
    stats = glob.glob(os.path.join(base, "trace", "*", "*_kernel_stats.csv"))
    if not stats:
        continue
    shutil.copy(stats[0], os.path.join(ROOT, "profiles", f"{out_tag}_{mode}_kernel_stats.csv"))
    rows_out = []
    for sub in ("pmc_sq1", "pmc_sq2", "pmc_fetch", "pmc_write", "pmc_mix"):
        for f in glob.glob(os.path.join(base, sub, "*", "*_counter_collection.csv")):
            agg = collections.defaultdict(list)
            meta = {}
            for r in csv.DictReader(open(f)):
                if ("rows_kernel" if mode == "resident" else "run_kernel") in r["Kernel_Name"]:
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                    meta = {k: r[k] for k in ("Kernel_Name", "Grid_Size", "Workgroup_Size", "VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size")}
            for name, vals in agg.items():
                rows_out.append({"pass": sub, "counter": name, "dispatches": len(vals), "mean": sum(vals) / len(vals),
                                 "min": min(vals), "max": max(vals), **meta})
    with open(os.path.join(ROOT, "profiles", f"{out_tag}_{mode}_pmc.csv"), "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows_out[0].keys()))
        w.writeheader()
        w.writerows(rows_out)
    print(mode, "->", len(rows_out), "counters")
    # HBM bytes per launch from the two separate PMC passes.  MI355X_MICROARCH.md (HBM section):
    # FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes
    # of a wide coalesced streaming read, WRITE_SIZE is exact for 16-B-per-lane streaming stores.
    vals = {r["counter"]: r["mean"] for r in rows_out}
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        import json
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
        traffic.setdefault("mse_1024", {})[mode] = {
            "read_bytes": 2.0 * vals["FETCH_SIZE"] * 1024, "write_bytes": vals["WRITE_SIZE"] * 1024,
            "bytes": 2.0 * vals["FETCH_SIZE"] * 1024 + vals["WRITE_SIZE"] * 1024,
            "source": f"profiles/{out_tag}_{mode}_pmc.csv (FETCH_SIZE x2 x1024 + WRITE_SIZE x1024, mean per launch)"}
        if "SQ_INSTS_VALU_FMA_F64" in vals:   # executed fp64 operations: wave instructions x 64 lanes, an FMA counts 2
            traffic["mse_1024"][mode]["fp64_flop_executed"] = 64.0 * (2 * vals["SQ_INSTS_VALU_FMA_F64"] + vals["SQ_INSTS_VALU_MUL_F64"]
                                                                      + vals.get("SQ_INSTS_VALU_ADD_F64", 0.0))
            traffic["mse_1024"][mode]["valu_instructions"] = vals.get("SQ_INSTS_VALU")
        json.dump(traffic, open(tpath, "w"), indent=1, sort_keys=True)
