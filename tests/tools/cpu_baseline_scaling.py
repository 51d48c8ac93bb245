"""The blocked VNNI CPU baseline (oracle.bf_xcheck1_blocked, bench.py's cpu_baseline_blocked) at 1 ... 128 threads on this host,
with the container's CPU quota beside it.  Lives in tests/tools/ because it runs the oracle (test infrastructure).
Usage: python tests/tools/cpu_baseline_scaling.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import oracle as orc
from fastmatch_amd import synth
print("nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "omp max", orc.max_threads())
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "-")
rng = np.random.default_rng(3)
Q = synth.synth_sift(20000, rng); T = synth.synth_sift(100000, rng)
for th in (1, 2, 4, 8, 16, 32, 64, 128):
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); orc.bf_xcheck1_blocked(Q, T, threads=th); best = min(best, time.perf_counter() - t0)
    print("threads %3d: %.3f s  %.3e pairs/s  (%.3e per thread)" % (th, best, 2e9 / best, 2e9 / best / th), flush=True)
