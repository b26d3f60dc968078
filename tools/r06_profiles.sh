#!/bin/bash
# round-6 profile set (run on the GPU box): rocprofv3 kernel trace + PMC passes of the metric workload one step at a time and
# pipelined, config 2, config 5.  Only the summaries come back (gpurun merges at most 64 MiB): the databases are deleted.
for spec in "r06mix mix 1" "r06d8 mix 8" "r06c2 config2 1" "r06c5 config5 1"; do
  set -- $spec
  timeout 900 bash tools/prof_run.sh $1 $2 $3 > gpurun_out/prof_$1.log 2>&1
  find gpurun_out/prof_$1 -name "*.db" -delete
  find gpurun_out/prof_$1 -name "*.csv" -size +1M -delete
  du -sh gpurun_out/prof_$1
done
