#!/bin/bash
# extra PMC passes for the state-resident kernel: instruction cache / fetch, fp64 instruction mix   $1 = tag
set -e
export TMPDIR=/tmp
TAG=${1:-r01}
OUT=gpurun_out/profx_${TAG}; rm -rf $OUT; mkdir -p $OUT
ARGS="bench.py --only --mode resident --steps 10 --warmup 2 --no-cpu-baseline"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/icache -- python3 $ARGS > $OUT/icache.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INST_CYCLES_SALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_LEVEL_WAVES SQ_CYCLES --output-format csv -d $OUT/mix -- python3 $ARGS > $OUT/mix.log 2>&1
python3 - <<PY
import csv, glob, collections
for sub in ("icache", "mix"):
    for f in glob.glob("$OUT/%s/*/*_counter_collection.csv" % sub):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(agg.items()):
            print(sub, k, c, len(v), sum(v) / len(v))
PY
