#!/bin/bash
# PMC pass over the state-resident MRF C3 run (rows_kernel without run folding): instruction mix and busy cycles
export TMPDIR=/tmp
OUT=gpurun_out/prof_mrf; rm -rf $OUT; mkdir -p $OUT
ARGS="bench.py --only --workload mrf_100 --steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc1 -- python3 $ARGS > $OUT/log1.txt 2>&1
rocprofv3 --pmc SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU --output-format csv -d $OUT/pmc2 -- python3 $ARGS > $OUT/log2.txt 2>&1
python3 - <<PY
import csv, glob, collections
for sub in ("pmc1", "pmc2"):
    for f in glob.glob("$OUT/%s/*/*_counter_collection.csv" % sub):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "rows_kernel" in r["Kernel_Name"]:
                agg[(r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(agg.items()):
            print(k, c, len(v), sum(v) / len(v))
PY
