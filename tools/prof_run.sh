#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun).  Outputs under gpurun_out/prof_$TAG/.
#   kernel trace + stats, then PMC passes (each in its own run, as the counters do not fit one pass)
TAG=${1:-r2}
WL=${2:-mix}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python $PWD/bench.py --workload $WL --steps 12 --warmup 6 --windows 2 --depth ${3:-1} --no-cpu-baseline --no-secondary --no-e2e --no-sustained --no-latency-form"
echo "$BENCH" > $OUT/cmd.txt
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS --kernel-trace -d $OUT/pmc_sq -o pmc -- $BENCH > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc_sq2 -o pmc -- $BENCH > $OUT/pmc_sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o pmc -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o pmc -- $BENCH > $OUT/pmc_write.log 2>&1
cd - >/dev/null
python $PWD/tools/prof_summary.py $OUT > $OUT/summary.md 2>&1
du -sh $OUT
