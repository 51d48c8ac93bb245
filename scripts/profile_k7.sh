#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats of the device-resident
# expansion loop (K7, expand_kernel) on BASELINE configs 3 (one threshold, and 15 thresholds in one launch)
# and 4 (scripts/run_configs.py 3 3t 4).
# Output: gpurun_out/prof_k7_$1/ ; copy kernel_stats.csv + run.log to profiles/ afterwards.
TAG=${1:-r02}
OUT=$PWD/gpurun_out/prof_k7_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
RUN="python3 $PWD/scripts/run_configs.py 3 3t 4"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $RUN > $OUT/run.log 2> $OUT/trace.err
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; done
find $OUT/trace -name "*kernel_trace.csv" -size +2M -delete
cat $OUT/run.log
head -12 $OUT/kernel_stats.csv
tail -3 $OUT/trace.err
