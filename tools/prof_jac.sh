#!/bin/bash
# PMC passes over tools/bench_jacobian.py (derivative kernels): instruction mix, fp64 instruction counts, busy cycles
#   tools/prof_jac.sh [tag]   -> profiles/<tag>_jacobian_pmc.csv, profiles/<tag>_jacobian_kernel_stats.csv
export TMPDIR=/tmp
TAG=${1:-r03}
OUT=gpurun_out/prof_jac_$TAG; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/bench_jacobian.py > $OUT/log0.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc1 -- python3 tools/bench_jacobian.py > $OUT/log1.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc2 -- python3 tools/bench_jacobian.py > $OUT/log2.txt 2>&1
python3 - <<PY
import csv, glob, collections, shutil, os
os.makedirs("profiles", exist_ok=True)
for f in glob.glob("$OUT/trace/*/*_kernel_stats.csv"):
    shutil.copy(f, "profiles/${TAG}_jacobian_kernel_stats.csv")
rows = []
for sub in ("pmc1", "pmc2"):
    for f in glob.glob("$OUT/%s/*/*_counter_collection.csv" % sub):
        agg, meta = collections.defaultdict(list), {}
        for r in csv.DictReader(open(f)):
            if "deriv" in r["Kernel_Name"] or "rows_kernel" in r["Kernel_Name"] or "rows_grow" in r["Kernel_Name"] or "drun" in r["Kernel_Name"]:
                key = (r["Kernel_Name"].split("(")[0], r["Counter_Name"])
                agg[key].append(float(r["Counter_Value"]))
                meta[key] = (r["VGPR_Count"], r["SGPR_Count"], r["Scratch_Size"], r["Grid_Size"])
        for (k, c), v in sorted(agg.items()):
            rows.append({"pass": sub, "kernel": k, "counter": c, "dispatches": len(v), "mean": sum(v) / len(v), "vgpr": meta[(k, c)][0],
                         "sgpr": meta[(k, c)][1], "scratch": meta[(k, c)][2], "grid": meta[(k, c)][3]})
with open("profiles/${TAG}_jacobian_pmc.csv", "w", newline="") as fh:
    w = csv.DictWriter(fh, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
for r in rows: print(r["kernel"][-40:], r["counter"], r["dispatches"], r["mean"])
PY
