#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats of the float32-route self distances (K8 masked sweep against
# K8-tri + tri_rescore_kernel, scripts/gpu_f32_selfdist.py 100000).
# Output: gpurun_out/prof_f32sd_$1/ ; copy kernel_stats.csv + run.log to profiles/ afterwards.
TAG=${1:-r06}
OUT=$PWD/gpurun_out/prof_f32sd_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
RUN="python3 $PWD/scripts/gpu_f32_selfdist.py 100000"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $RUN > $OUT/run.log 2> $OUT/trace.err
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; done
find $OUT/trace -name "*kernel_trace.csv" -size +2M -delete
cat $OUT/run.log
head -14 $OUT/kernel_stats.csv
tail -3 $OUT/trace.err
