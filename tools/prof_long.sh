#!/bin/bash
# profiling of the long-state-matrix kernels on the GPU box: $1 = tag.  One --kernel-trace --stats run and one --pmc pass (fp64
# instruction mix) of tools/bench_long_trains.py; the program itself follows `--`.  Condensed into profiles/<tag>_long_trains_pmc.csv
# (per kernel: launches, mean time, executed fp64 flop, fraction of the 78.6 TFLOP/s vector peak).
set -e
export TMPDIR=/tmp
TAG=${1:-r04}
OUT=gpurun_out/prof_${TAG}_long; rm -rf $OUT; mkdir -p $OUT
ARGS="tools/bench_long_trains.py --nechos 100 255 511 1023 --steps 2"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT.trace.log 2>&1
echo "long trace done"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mix -- python3 $ARGS > $OUT.pmc.log 2>&1
echo "long pmc done"
python3 - $TAG $OUT <<'PY'
import collections, csv, glob, os, sys
tag, out = sys.argv[1], sys.argv[2]
stats = glob.glob(os.path.join(out, "trace", "*", "*_kernel_stats.csv"))
want = ("run_contig_grow_kernel", "run_contig_kernel", "run_split_kernel", "run_kernel<8")
times = {}
if stats:
    import shutil
    shutil.copy(stats[0], f"profiles/{tag}_long_trains_kernel_stats.csv")
    for r in csv.DictReader(open(stats[0])):
        if any(w in r["Name"] for w in want):
            times[r["Name"]] = (int(r["Calls"]), float(r["AverageNs"]) * 1e-6)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for f in glob.glob(os.path.join(out, "pmc_mix", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if any(w in r["Kernel_Name"] for w in want):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[r["Kernel_Name"]] = (r["VGPR_Count"], r["Scratch_Size"], r["Grid_Size"], r["Workgroup_Size"])
with open(f"profiles/{tag}_long_trains_pmc.csv", "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "launches_traced", "mean_ms_traced", "SQ_INSTS_VALU", "FMA_F64", "MUL_F64", "ADD_F64", "fp64_flop_executed_per_launch",
                "TFLOP_per_s", "frac_of_78.6", "fp64_share_of_valu", "VGPR", "Scratch", "grid", "workgroup"])
    for k, c in sorted(agg.items()):
        m = {n: sum(v) / len(v) for n, v in c.items()}
        flop = 64.0 * (2 * m.get("SQ_INSTS_VALU_FMA_F64", 0) + m.get("SQ_INSTS_VALU_MUL_F64", 0) + m.get("SQ_INSTS_VALU_ADD_F64", 0))
        calls, ms = times.get(k, (0, float("nan")))
        tf = flop / (ms * 1e-3) / 1e12 if ms == ms and ms > 0 else float("nan")
        f64 = m.get("SQ_INSTS_VALU_FMA_F64", 0) + m.get("SQ_INSTS_VALU_MUL_F64", 0) + m.get("SQ_INSTS_VALU_ADD_F64", 0)
        w.writerow([k[:90], calls, round(ms, 3), int(m.get("SQ_INSTS_VALU", 0)), int(m.get("SQ_INSTS_VALU_FMA_F64", 0)), int(m.get("SQ_INSTS_VALU_MUL_F64", 0)),
                    int(m.get("SQ_INSTS_VALU_ADD_F64", 0)), int(flop), round(tf, 2), round(tf / 78.6, 3), round(f64 / max(m.get("SQ_INSTS_VALU", 1), 1), 3), *meta[k]])
print(open(f"profiles/{tag}_long_trains_pmc.csv").read())
PY
