#!/bin/bash
# round 6: counters of the lane-per-slot kernels inside a step (one step at a time): issue, waits, LDS, instruction cache, HBM bytes.
# usage: tools/experiments/prof_slot.sh TAG [slot mode]
TAG=${1:-slot}
MODE=${2:-2}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python $PWD/bench.py --steps 12 --warmup 6 --windows 2 --depth 1 --slot-mode $MODE --no-cpu-baseline --no-secondary --no-e2e --no-sustained --no-latency-form"
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS --kernel-trace -d $OUT/pmc_sq -o pmc -- $BENCH > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc_sq2 -o pmc -- $BENCH > $OUT/pmc_sq2.log 2>&1
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES SQC_TC_INST_REQ SQ_IFETCH SQ_IFETCH_LEVEL --kernel-trace -d $OUT/pmc_ic -o pmc -- $BENCH > $OUT/pmc_ic.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o pmc -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o pmc -- $BENCH > $OUT/pmc_write.log 2>&1
cd - >/dev/null
python $PWD/tools/prof_summary.py $OUT > $OUT/summary.md 2>&1
cat $OUT/summary.md | head -60
