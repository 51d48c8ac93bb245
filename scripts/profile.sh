#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes of
# the default bench workload.  Outputs under gpurun_out/prof_$1/ ; summaries are copied to
# profiles/ by hand afterwards.
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export FM_REPO=$PWD FM_PROFILE_TAG=$TAG
BENCH="python3 $PWD/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-legs"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/bench_trace.json 2> $OUT/trace.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $OUT/pmc_sq1 -- $BENCH > /dev/null 2> $OUT/pmc_sq1.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- $BENCH > /dev/null 2> $OUT/pmc_sq2.err
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > /dev/null 2> $OUT/pmc_fetch.err
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > /dev/null 2> $OUT/pmc_write.err
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/pmc_tcc -- $BENCH > /dev/null 2> $OUT/pmc_tcc.err
cd $OUT
# keep the outputs small: per-kernel stats + aggregated counters only
python3 - <<'PY'
import csv, glob, os, collections
out = os.getcwd()
for d in sorted(glob.glob("pmc_*")):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            k = (row["Kernel_Name"][:60], row["Counter_Name"])
            agg[k][0] += float(row["Counter_Value"]); agg[k][1] += 1
        with open(os.path.join(out, d + "_summary.csv"), "w") as w:
            w.write("kernel,counter,mean_per_dispatch,dispatches\n")
            for (k, c), (s, n) in sorted(agg.items()):
                w.write('"%s",%s,%.1f,%d\n' % (k, c, s / n, n))
# HBM bytes per launch of the dominant kernel: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
# FETCH_SIZE counts 128-B requests as 64 B for wide coalesced reads -> x2 (MI355X_MICROARCH.md, HBM)
import json, re
def mean_of(fn, counter):
    # the headline's distance kernel: the batched launch (several image pairs per dispatch) if the bench
    # used it, else the single-pair top-1 kernel
    try:
        rows = [r for r in csv.DictReader(open(fn)) if r["counter"] == counter]
        for want in ("rowreduce_batch_kernel", "rowreduce_kernel<4, 1,"):
            for row in rows:
                if want in row["kernel"]:
                    return float(row["mean_per_dispatch"])
    except Exception:
        return None
fs, ws = mean_of("pmc_fetch_summary.csv", "FETCH_SIZE"), mean_of("pmc_write_summary.csv", "WRITE_SIZE")
if fs is not None and ws is not None:
    d = {"hbm_bytes_per_launch": (2.0 * fs + ws) * 1024.0, "fetch_size_kib_raw": fs, "write_size_kib": ws,
         "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py (scripts/profile.sh), FETCH_SIZE x2 gfx950 correction"}
    mb, ga = mean_of("pmc_sq1_summary.csv", "SQ_VALU_MFMA_BUSY_CYCLES"), mean_of("pmc_sq2_summary.csv", "GRBM_GUI_ACTIVE")
    if mb and ga:       # MFMA busy cycles are summed over 1024 SIMDs, GPU-active cycles over 8 XCDs
        d.update({"mfma_busy_cycles_per_simd": mb / 1024, "gpu_cycles_per_launch": ga / 8,
                  "mfma_pipe_busy_frac": (mb / 1024) / (ga / 8),
                  "pmc_source": "SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over GRBM_GUI_ACTIVE / 8 XCDs (separate rocprofv3 --pmc passes)"})
    # tie the counters to the kernel source they were collected from (bench.py attaches them
    # only when this hash equals the hash of the source it runs)
    import hashlib
    h = hashlib.sha256()
    for rel in ("fast-match_amd/csrc/rowreduce.hip", "fast-match_amd/csrc/tile_ops.h"):
        h.update(open(os.path.join(os.environ["FM_REPO"], rel), "rb").read())
    d.update({"k1_source_sha256": h.hexdigest(), "tag": os.environ.get("FM_PROFILE_TAG", "")})
    # image pairs per dispatch of that kernel in the profiled runs (the counters above are per dispatch)
    try:
        line = [l for l in open("bench_trace.json") if l.startswith("{")][-1]
        d["image_pairs_per_launch"] = json.loads(line)["roofline"]["image_pairs_per_launch"]
    except Exception:
        d["image_pairs_per_launch"] = None
    json.dump(d, open("latest_pmc.json", "w"))
for f in glob.glob("trace/**/*kernel_stats.csv", recursive=True):
    os.system("cp %s %s/kernel_stats.csv" % (f, out))
# The tool's own statistics average every dispatch, the bench's warm-up step included (its first launch
# pays clock ramp-up and cold caches: 11.7 vs 9.9-10.4 ms in r02).  kernel_stats_steady.csv leaves the
# first FM_PROFILE_WARMUP (default 1) dispatches of every kernel out, so that the profile and the bench's
# HIP-event figure describe the same launches.
skip = int(os.environ.get("FM_PROFILE_WARMUP", "1"))
for f in glob.glob("trace/**/*kernel_trace.csv", recursive=True):
    per = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        try:
            per[row["Kernel_Name"]].append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
        except (KeyError, ValueError):
            continue
    with open(os.path.join(out, "kernel_stats_steady.csv"), "w") as w:
        w.write('"Name","Calls","CallsCounted","AverageNs_all","AverageNs_steady","MinNs","MaxNs_steady","skipped_first"\n')
        for name, d in sorted(per.items(), key=lambda kv: -sum(x[1] for x in kv[1])):
            d.sort()
            dur = [x[1] for x in d]
            st = dur[skip:] if len(dur) > skip else dur
            w.write('"%s",%d,%d,%.1f,%.1f,%d,%d,%d\n' % (name[:120], len(dur), len(st), sum(dur) / len(dur), sum(st) / len(st),
                                                       min(dur), max(st), len(dur) - len(st)))
    break
PY
rm -rf $OUT/pmc_sq1 $OUT/pmc_sq2 $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_tcc
find $OUT/trace -name "*kernel_trace.csv" -size +2M -delete
ls -la $OUT
cat $OUT/kernel_stats.csv | head -20
head -6 $OUT/kernel_stats_steady.csv
cat $OUT/*_summary.csv | grep -i rowreduce
for f in $OUT/*.err; do tail -n 2 $f; done | head -30
