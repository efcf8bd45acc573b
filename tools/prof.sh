#!/bin/bash
# profiling helper run on the GPU box: $1 = mode (resident|stream), $2 = tag, $3 = workload (default mse_1024)
# one --kernel-trace --stats run and separate --pmc passes of the SAME bench command (FETCH_SIZE and WRITE_SIZE
# cannot share a pass: MI355X_MICROARCH.md, rocprofv3 PMC slots); the program itself follows `--`
set -e
export TMPDIR=/tmp
MODE=${1:-resident}; TAG=${2:-r02}; WL=${3:-mse_1024}
SUF=""; [ "$WL" != "mse_1024" ] && SUF="_$WL"
OUT=gpurun_out/prof_${TAG}${SUF}_${MODE}; rm -rf $OUT; mkdir -p $OUT
STEPS=20; WARM=3; [ "$WL" = "mrf_100" ] && STEPS=3 && WARM=1
ARGS="bench.py --only --mode $MODE --workload $WL --steps $STEPS --warmup $WARM"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT.trace.log 2>&1
echo "trace done"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq1 -- python3 $ARGS > $OUT.pmc1.log 2>&1
echo "pmc1 done"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- python3 $ARGS > $OUT.pmc2.log 2>&1
echo "pmc2 done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT.pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT.pmc4.log 2>&1
echo "pmc3/4 done"
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_BRANCH SQC_ICACHE_REQ SQC_ICACHE_MISSES --output-format csv -d $OUT/pmc_mix -- python3 $ARGS > $OUT.pmc5.log 2>&1
echo "pmc5 done"
python3 tools/collect_profiles.py $TAG $TAG $WL $MODE
