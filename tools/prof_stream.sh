#!/bin/bash
# rocprofv3 passes over tools/experiments/exp_stream_front.py (stream front end only).  Outputs under gpurun_out/prof_$TAG/.
TAG=${1:-sf}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $PWD/tools/experiments/exp_stream_front.py"
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
if [ "$2" != "trace" ]; then
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS --kernel-trace -d $OUT/pmc_sq -o pmc -- $CMD > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc_sq2 -o pmc -- $CMD > $OUT/pmc_sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o pmc -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o pmc -- $CMD > $OUT/pmc_write.log 2>&1
fi
cd - >/dev/null
python $PWD/tools/prof_summary.py $OUT > $OUT/summary.md 2>&1
head -30 $OUT/summary.md
