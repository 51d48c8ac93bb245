#!/bin/bash
# Counters of the 12-pair K1 launch under each workgroup order (option k1_order): duration (kernel trace),
# FETCH_SIZE / WRITE_SIZE, TCC hit / miss / EA read requests, MFMA busy + GRBM_GUI_ACTIVE -- one rocprofv3 pass each.
# Outputs: gpurun_out/prof_k1xcd_$1/order<N>_<pass>_summary.csv, copied to profiles/ by hand.
TAG=${1:-r04}
OUT=$PWD/gpurun_out/prof_k1xcd_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
RUN="$PWD/scripts/gpu_k1_order_run.py"
cd /tmp
I=0
: > $OUT/variants.txt
for V in ${ORDERS:-0 1 2}; do      # a bare number = k1_order, or an option list like bound_every=4,k1_order=0
  O=v$I; I=$((I+1)); echo "$O $V" >> $OUT/variants.txt
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/o${O}_trace -- python3 $RUN $V 6 > $OUT/o${O}_trace.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/o${O}_fetch -- python3 $RUN $V 3 > /dev/null 2> $OUT/o${O}_fetch.err
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/o${O}_write -- python3 $RUN $V 3 > /dev/null 2> $OUT/o${O}_write.err
  timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/o${O}_tcc -- python3 $RUN $V 3 > /dev/null 2> $OUT/o${O}_tcc.err
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/o${O}_sq -- python3 $RUN $V 3 > /dev/null 2> $OUT/o${O}_sq.err
done
cd $OUT
python3 - <<'PY'
import csv, glob, os, collections
rows = []
names = dict(l.split() for l in open("variants.txt") if l.strip())
for d in sorted(glob.glob("ov*_*")):
    if not os.path.isdir(d): continue
    order, what = names.get(d.split("_")[0][1:], d.split("_")[0][1:]), d.split("_", 1)[1]
    if what == "trace":
        for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
            dur = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
                         for r in csv.DictReader(open(f)) if "rowreduce_batch_kernel" in r["Kernel_Name"] and "Lb1" not in r["Kernel_Name"] and "true>" not in r["Kernel_Name"])
            dur = [x[1] for x in dur][1:]          # first launch: clock ramp, cold caches
            if dur: rows.append((order, "duration_ms_mean_steady", sum(dur) / len(dur) / 1e6, len(dur)))
        continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if "rowreduce_batch_kernel" not in r["Kernel_Name"] or "true>" in r["Kernel_Name"]: continue
            k = r["Counter_Name"]
            agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
        for k, (s, n) in sorted(agg.items()):
            rows.append((order, k, s / n, n))
with open("k1_xcd_summary.csv", "w") as w:
    w.write("variant,counter,mean_per_12_pair_launch,launches\n")
    for r in rows: w.write("%s,%s,%.1f,%d\n" % r)
print(open("k1_xcd_summary.csv").read())
PY
rm -rf $OUT/o*_fetch $OUT/o*_write $OUT/o*_tcc $OUT/o*_sq
find $OUT -name "*kernel_trace.csv" -size +2M -delete
for f in $OUT/*.err; do tail -n 1 $f; done | head
