#!/bin/bash
# PMC passes over k_front_stream alone (tools/experiments/front_ablate.py): instruction fetch / issue / LDS / VMEM queue counters.
# Outputs under gpurun_out/prof_$TAG/.  usage: tools/prof_front_issue.sh TAG
TAG=${1:-fi}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $PWD/tools/experiments/front_ablate.py"
cd /tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_INPUT_VALID_READYB SQC_TC_INST_REQ GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc_ic -o pmc -- $CMD > $OUT/pmc_ic.log 2>&1
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES --kernel-trace -d $OUT/pmc_if -o pmc -- $CMD > $OUT/pmc_if.log 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_LDS_ADDR_CONFLICT --kernel-trace -d $OUT/pmc_q -o pmc -- $CMD > $OUT/pmc_q.log 2>&1
cd - >/dev/null
python $PWD/tools/prof_summary.py $OUT > $OUT/summary.md 2>&1
grep "k_front_stream |" $OUT/summary.md
