#!/bin/bash
# kernel trace only: tools/prof_quick.sh TAG <bench args...>
TAG=$1; shift
OUT=$PWD/gpurun_out/q_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $R/bench.py "$@" > $OUT/trace.log 2>&1
cd - > /dev/null
python $R/tools/prof_summary.py $OUT | head -22
