#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + SQ counter passes (each in its own run) of the float32-route
# self distances: K8 masked sweep against K8-tri + tri_rescore_kernel (scripts/gpu_f32_selfdist.py 100000).
# Output: gpurun_out/prof_f32sd_$1/ ; copy kernel_stats.csv, pmc_*_summary.csv and run.log to profiles/ afterwards.
TAG=${1:-r06}
OUT=$PWD/gpurun_out/prof_f32sd_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python3 $PWD/scripts/gpu_f32_selfdist.py 100000"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/run.log 2> $OUT/trace.err
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $OUT/pmc_sq1 -- $CMD > /dev/null 2> $OUT/pmc_sq1.err
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- $CMD > /dev/null 2> $OUT/pmc_sq2.err
cd $OUT
python3 - <<'PY'
import csv, glob, os, collections
for d in sorted(glob.glob("pmc_*")):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            k = (row["Kernel_Name"][:70], row["Counter_Name"])
            agg[k][0] += float(row["Counter_Value"]); agg[k][1] += 1
        with open(d + "_summary.csv", "w") as w:
            w.write("kernel,counter,mean_per_dispatch,dispatches\n")
            for (k, c), (s, n) in sorted(agg.items()):
                if "filter_kernel" in k or "rescore" in k or "rescan" in k:
                    w.write('"%s",%s,%.1f,%d\n' % (k, c, s / n, n))
for f in glob.glob("trace/**/*kernel_stats.csv", recursive=True):
    os.system("cp %s kernel_stats.csv" % f)
os.system("rm -rf trace pmc_sq1 pmc_sq2")
PY
cat $OUT/run.log; head -8 $OUT/kernel_stats.csv; cat $OUT/pmc_sq1_summary.csv $OUT/pmc_sq2_summary.csv | grep filter_kernel
