#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/park_trace
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/scripts/gpu_c3_clustered.py 1 > $OUT/run.log 2>&1
cat $OUT/run.log | tail -3
python3 - <<'P'
import csv, glob, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/park_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# runs of the chunked expand kernel with delegation: find the expand kernels; the delegated get() is the one with ~1595 expand launches in a row
ex = [i for i, e in enumerate(ev) if "expand_kernel" in e[2]]
print("expand launches", len(ex), "kernels", len(ev))
# window: the longest stretch of events where consecutive expand launches are < 5 ms apart
best = None
s = 0
for a in range(1, len(ex) + 1):
    if a == len(ex) or ev[ex[a]][0] - ev[ex[a - 1]][1] > 5_000_000:
        if best is None or a - s > best[1] - best[0]: best = (s, a)
        s = a
i0, i1 = ex[best[0]], ex[best[1] - 1]
win = ev[i0:i1 + 1]
t0, t1 = win[0][0], win[-1][1]
busy = 0; cur_s, cur_e = win[0][0], win[0][1]
for s_, e_, _ in win[1:]:
    if s_ > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s_, e_
    else: cur_e = max(cur_e, e_)
busy += cur_e - cur_s
by = {}
for s_, e_, n in win:
    k = n.split("(")[0][:60]
    by[k] = by.get(k, [0, 0]); by[k][0] += 1; by[k][1] += e_ - s_
print("window %.4f s, %d expand launches, GPU busy %.4f s, idle %.4f s (%.1f us per park)" % ((t1 - t0) / 1e9, best[1] - best[0], busy / 1e9, (t1 - t0 - busy) / 1e9, (t1 - t0 - busy) / 1e3 / (best[1] - best[0])))
for k, v in sorted(by.items(), key=lambda kv: -kv[1][1]): print("  %-62s %6d launches %9.4f s  avg %.1f us" % (k, v[0], v[1] / 1e9, v[1] / 1e3 / v[0]))
P
