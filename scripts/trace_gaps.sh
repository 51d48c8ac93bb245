#!/bin/bash
# Device timeline of the bench step: gaps between consecutive distance-kernel launches (rocprofv3 kernel trace).
OUT=$PWD/gpurun_out/trace_gaps
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-legs > $OUT/bench.json 2> $OUT/err.log
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
k1 = [r for r in rows if "rowreduce_batch_kernel" in r["Kernel_Name"]]
print("batch launches:", len(k1))
prev = None
for r in k1[-12:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    # small kernels that START inside the gap
    print("dur %8.1f us   gap before %7.1f us   grid %s" % ((e - s) / 1e3, gap, r.get("Grid_Size", r.get("Grid_Size_X", "?"))))
    prev = e
PY
