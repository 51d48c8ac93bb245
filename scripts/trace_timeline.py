"""Print the device timeline (kernels + memory copies) of the last ~40 operations from a rocprofv3 trace dir."""
import csv, glob, sys
d = sys.argv[1]
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
ev.sort()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
t0 = ev[-n][0] if len(ev) >= n else ev[0][0]
prev = None
for s, e, name in ev[-n:]:
    gap = (s - prev) / 1e3 if prev else 0.0
    print("%10.1f us  +%8.1f us  (gap %7.1f)  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, name))
    prev = e
