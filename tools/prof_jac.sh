#!/bin/bash
# PMC pass over tools/bench_jacobian.py (derivative kernel): instruction mix and busy cycles per dispatch
export TMPDIR=/tmp
OUT=gpurun_out/prof_jac; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc -- python3 tools/bench_jacobian.py > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/pmc/*/*_counter_collection.csv"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "deriv_kernel" in r["Kernel_Name"] or "rows_kernel" in r["Kernel_Name"] or "run_kernel" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][:48], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items()):
        print(k, c, len(v), sum(v) / len(v))
PY
