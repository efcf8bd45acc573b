#!/bin/bash
# scalar-cache counters of the resident bench kernel (separate pass, counters only)
export TMPDIR=/tmp
OUT=gpurun_out/prof_sqc; rm -rf $OUT
rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQC_TC_DATA_READ_REQ SQC_ICACHE_REQ SQC_ICACHE_MISSES --output-format csv -d $OUT -- python3 bench.py --only --mode resident --steps 3 --warmup 1 --no-cpu-baseline > $OUT.log 2>&1
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/prof_sqc/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "run_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k, sum(v) / len(v))
PY
tail -3 $OUT.log
