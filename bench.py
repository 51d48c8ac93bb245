#!/usr/bin/env python3
"""bench.py -- throughput of the descriptor-matching hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL; a bare
   `python bench.py --gpus N` starts those N ranks itself as a child process)

Workload (BASELINE.json configs[1]): 100k x 100k synthetic 128-D uint8 SIFT descriptors,
brute-force cross-checked 1-NN + ratio test.  Every GPU holds a batch of PAIRS_PER_STEP
independent image pairs (distinct banks, all resident in HBM before the timed region; weak
scaling: the batch per GPU is fixed as N grows; SURVEY.md 8(d) C2, seed 20250002 + rank).
A step = one pass of the hot path over that batch: per pair, cross-checked 1-NN (OpenCV
BFMatcher crossCheck semantics) + float64 ratio test at tau = 0.7 against the query bank's
self distances + ordered compaction of the accepted matches on the device.  N = 1: the
accepted matches land in caller-owned page-locked host buffers, the pairs of a step enqueued back
to back with one synchronisation per step (fm_match_accepted_async); N > 1:
they stay on the device as packed 12-byte rows (fm_match_accepted_dev) and go straight into
the RCCL all-gather, overlapped with the next pair's kernels.
Prints ONE JSON line (rank 0).  Beside the headline it carries legs for the other BASELINE
configs: `classic_ratio_match` and `self_2nn` (configs[1] read literally / Metric_Cache build),
`expand_c3` (configs[2]), `expand_c4` (configs[3]), `float32_route` (configs[4]).
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

NQ = int(os.environ.get("FM_BENCH_NQ", "100000"))     # (the knobs exist for tests/test_bench_multirank_gpu.py: a small workload
NT = int(os.environ.get("FM_BENCH_NT", "100000"))     # through the N > 1 code path; the driver runs the defaults)
TAU = 0.7
SEED = 20250002 + int(os.environ.get("FM_BENCH_SEED_OFFSET", "0"))
PAIRS_PER_STEP = int(os.environ.get("FM_BENCH_PAIRS", "12"))   # independent 100k x 100k pairs per GPU and step (timed region >= 0.2 s at 20 steps; at most 16 share a launch)
PROFILE_JSON = os.path.join(ROOT, "profiles", "latest_pmc.json")   # written by scripts/profile.sh
K1_SOURCES = ("fast-match_amd/csrc/rowreduce.hip", "fast-match_amd/csrc/tile_ops.h")
INT8_DENSE_PEAK_TOPS = 5000.0     # MI355X dense int8 MFMA (2x bf16's ~2.5 PF), MI355X_MICROARCH.md
OPS_PER_PAIR = 256                # 128 MACs per 128-D descriptor pair


def k1_source_hash():
    """SHA-256 over the sources of the dominant kernel; scripts/profile.sh stores the same hash
    next to the PMC counters so that stale counters are never attached to a changed kernel."""
    h = hashlib.sha256()
    for rel in K1_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def host_cpu_allowance():
    """(threads to use, CPU quota or None): the CPUs this process may run on -- its affinity mask, cut to the cgroup's CPU
    quota where one is set (cpu.max, or cfs_quota_us / cfs_period_us under cgroup v1).  r06: the GPU boxes of the pool show 256
    logical CPUs and grant 16 (cpu.max 1600000 100000); 128 threads under that quota are throttled, not faster, and "cores: 128"
    in the line was not what the figure was measured on."""
    import math
    import oracle
    n = min(len(os.sched_getaffinity(0)), oracle.max_threads())
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota is not None:
        n = max(1, min(n, int(math.floor(quota + 1e-9))))
    return n, quota


def cpu_baseline(Q, T, budget_s=float(os.environ.get("FM_BENCH_CPU_BUDGET_S", "12")), keep=None):
    """Oracle (kind 'port': the reference's .so cannot run here) on a bounded row sample of
    the same workload, all host cores.  keep (a dict): receives the sample size and the oracle's
    (tidx, dist) of the sample, for the self-check of the timed batch."""
    import oracle
    threads, quota = host_cpu_allowance()
    s0 = 256
    t0 = time.perf_counter()
    oracle.bf_xcheck1(Q[:s0], T, threads=threads)
    dt = time.perf_counter() - t0
    rate = s0 * len(T) / dt
    s = int(min(len(Q), max(s0, budget_s * rate / len(T))))
    t0 = time.perf_counter()
    res = oracle.bf_xcheck1(Q[:s], T, threads=threads)
    dt = time.perf_counter() - t0
    if keep is not None:
        keep["rows"], keep["tidx"], keep["dist"] = s, res[0], res[1]
    return {"value": s * len(T) / dt, "unit": "pairs/s", "cores": threads, "kind": "port", "host_cpu_quota": quota,
            "sample": "oracle bf_xcheck1 (C, OpenMP, %d threads%s) on the first %d of %d query rows x all %d target rows of pair 0, %.1f s"
                      % (threads, "" if quota is None else " = this container's CPU quota of %.4g" % quota, s, len(Q), len(T), dt)}


def cpu_baseline_simd(Q, T, budget_s=float(os.environ.get("FM_BENCH_CPU_SIMD_BUDGET_S", "6"))):
    """The same oracle semantics with the inner loop a CPU programmer would write (int16 differences, vpmaddwd, eight
    output rows per pass over the other bank; oracle.bf_xcheck1_simd, results identical): the faithful restatement runs
    at ~1 multiply-add per cycle and thread, and a GPU / CPU ratio against THAT flatters the GPU."""
    import oracle
    threads, quota = host_cpu_allowance()
    s0 = 1024
    t0 = time.perf_counter()
    oracle.bf_xcheck1_simd(Q[:s0], T, threads=threads)
    dt = time.perf_counter() - t0
    s = int(min(len(Q), max(s0, budget_s * s0 / dt)))
    t0 = time.perf_counter()
    oracle.bf_xcheck1_simd(Q[:s], T, threads=threads)
    dt = time.perf_counter() - t0
    value = s * len(T) / dt
    # what the host could do at best with this instruction: 2 vpmaddwd (16 int16 products each) per cycle and PHYSICAL core
    # = 32 multiply-adds of the 128 a descriptor pair takes, i.e. 0.25 pairs per cycle and core
    phys, ghz = host_cores_and_ghz(threads)
    peak = phys * ghz * 1e9 * 32.0 / 128.0
    return {"value": value, "unit": "pairs/s", "cores": threads, "kind": "port", "host_cpu_quota": quota,
            "physical_cores": phys, "ghz": ghz, "host_vpmaddwd_peak_pairs_per_s": peak, "frac_of_host_peak": value / peak,
            "sample": "oracle bf_xcheck1_simd (C, AVX2 vpmaddwd, OpenMP) on the first %d of %d query rows x all %d target rows of pair 0, %.1f s; "
                      "= %.1f %% of this host's vpmaddwd peak (%d physical cores x %.2f GHz x 32 multiply-adds per cycle / 128 per pair): "
                      "one target row streams past eight query rows per pass, a horizontal sum and a scalar compare per pair -- a "
                      "cache-blocked kernel with vector compares runs many times faster (cpu_baseline_blocked, where the host has AVX-512 VNNI): "
                      "the GPU / CPU ratio from THIS figure flatters the GPU; the fraction of the MFMA roof is the number that counts"
                      % (s, len(Q), len(T), dt, 100.0 * value / peak, phys, ghz)}


def cpu_baseline_blocked(Q, T, budget_s=float(os.environ.get("FM_BENCH_CPU_BLOCKED_BUDGET_S", "6"))):
    """The same oracle semantics programmed for the host's dot-product units and caches (AVX-512 VNNI vpdpbusd, 64 output rows
    x 6 candidates per register block, vector compares; oracle.bf_xcheck1_blocked, results identical): the honest denominator
    of a GPU / CPU ratio.  Raises on a host without VNNI."""
    import oracle
    threads, quota = host_cpu_allowance()
    s0 = 4096
    oracle.bf_xcheck1_blocked(Q[:256], T, threads=threads)          # (thread team, page faults)
    t0 = time.perf_counter()
    oracle.bf_xcheck1_blocked(Q[:s0], T, threads=threads)
    dt = time.perf_counter() - t0
    s = int(min(len(Q), max(s0, budget_s * s0 / dt)))
    t0 = time.perf_counter()
    oracle.bf_xcheck1_blocked(Q[:s], T, threads=threads)
    dt = time.perf_counter() - t0
    value = s * len(T) / dt
    # host peak with this instruction, assuming two 512-bit vpdpbusd (64 products each) per cycle and PHYSICAL core = 128 of the
    # 128 multiply-adds a descriptor pair takes, i.e. one pair per cycle and core (half of that where 512-bit ops issue once per cycle)
    phys, ghz = host_cores_and_ghz(threads)
    peak = phys * ghz * 1e9
    return {"value": value, "unit": "pairs/s", "cores": threads, "kind": "port", "host_cpu_quota": quota,
            "pairs_per_s_per_core": value / threads, "whole_host_physical_cores": host_cores_and_ghz(1 << 20)[0],
            "physical_cores": phys, "ghz": ghz, "host_vpdpbusd_peak_pairs_per_s": peak, "frac_of_host_peak": value / peak,
            "sample": "oracle bf_xcheck1_blocked (C, AVX-512 VNNI vpdpbusd, register-blocked 64 x 6, OpenMP) on the first %d of %d query "
                      "rows x all %d target rows of pair 0, %.2f s; = %.0f %% of this host's vpdpbusd peak (%d physical cores x %.2f GHz x "
                      "128 multiply-adds per cycle / 128 per pair); the GPU / CPU ratio to quote is against THIS figure"
                      % (s, len(Q), len(T), dt, 100.0 * value / peak, phys, ghz)}


def host_cores_and_ghz(threads):
    """(physical cores, nominal GHz) of this host from /proc/cpuinfo; (threads, 2.0) where that cannot be read."""
    try:
        txt = open("/proc/cpuinfo").read()
        cores = set()
        phys_id = core_id = None
        mhz = []
        for line in txt.splitlines():
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "physical id":
                phys_id = v
            elif k == "core id":
                core_id = v
            elif k == "cpu MHz":
                mhz.append(float(v))
            elif k == "" and phys_id is not None and core_id is not None:
                cores.add((phys_id, core_id))
                phys_id = core_id = None
        if phys_id is not None and core_id is not None:
            cores.add((phys_id, core_id))
        m = __import__("re").search(r"@\s*([0-9.]+)\s*GHz", txt)
        ghz = float(m.group(1)) if m else (max(mhz) / 1000.0 if mhz else 2.0)
        return (min(len(cores), threads) if cores else threads), ghz
    except Exception:
        return threads, 2.0


def verify_against_oracle(ctx, Q, T, selfdist, keep, timed_rows, pair0):
    """The oracle result cpu_baseline computed anyway, against the device: if the sample covered every query
    row, against pair 0's accepted rows AS THE TIMED BATCH LEFT THEM (timed_rows = qidx, tidx, dist, ratio);
    otherwise (few host cores) against a fresh device call on the same sub-problem (the election is over all
    query rows, so a row sample of the full problem is a different problem)."""
    import oracle
    s, otidx, odist = keep["rows"], keep["tidx"], keep["dist"]
    m = otidx >= 0
    rows = np.nonzero(m)[0].astype(np.int32)
    oratio, opass = oracle.ratio_filter(odist[m], selfdist[:s] if s < len(Q) else selfdist, TAU, qrows=rows)
    want = (rows[opass], otidx[m][opass], odist[m][opass], oratio[opass])
    if s == len(Q) and timed_rows is not None:
        got, scope = timed_rows, "pair 0 of the timed batch, all %d query rows" % s
    elif s == len(Q):
        got, scope = ctx.match_accepted(pair0[0], pair0[1], TAU), "fresh call on pair 0, all %d query rows" % s
    else:
        qb, tb = ctx.bank(Q[:s]), ctx.bank(T)
        qb.set_selfdist(selfdist[:s])
        got, scope = ctx.match_accepted(qb, tb, TAU), "fresh call on the first %d query rows of pair 0 x all target rows" % s
    ok = all(len(a) == len(b) and np.array_equal(np.asarray(a).view(np.uint8), np.asarray(b).view(np.uint8)) for a, b in zip(got, want))
    return bool(ok), scope, int(len(want[0]))


def derived_pair(Q, T, j, rng):
    """Pair j of the batch: rows of pair 0 permuted and the 128 dimensions rolled by 8 j --
    different banks in memory, the same distribution of distances and accepted matches.  The LAST pair of a batch is
    drawn independently from ANOTHER distribution (a batch of twelve twins cannot show a cost tuned to
    one distribution): 55 % planted rows instead of 30 %, noise sigma 10 instead of 6."""
    if j == 0:
        return Q, T
    if j == PAIRS_PER_STEP - 1:
        from fastmatch_amd import synth
        Qi, Ti, _ = synth.planted_pair(len(Q), len(T), seed=SEED + 7777 + j, p=0.55, sigma=10.0)
        return Qi, Ti
    pq, pt = rng.permutation(len(Q)), rng.permutation(len(T))
    return np.ascontiguousarray(np.roll(Q[pq], 8 * j, axis=1)), np.ascontiguousarray(np.roll(T[pt], 8 * j, axis=1))


def build_image_pair(ctx, size, n, seed, n_thumb, **kw):
    from fastmatch_amd import synth, cache
    q, t = synth.image_pair(size, n, seed, n_thumb=n_thumb, **kw)
    mc = cache.Metric_Cache.from_arrays(q["descriptors"], q["positions"], q["size"], q["thumb_descriptors"],
                                        q["thumb_positions"], q["thumb_size"], options={"context": ctx})
    fi = cache.Feature_Image(t["size"], t["positions"], t["descriptors"], t["thumb_positions"],
                             t["thumb_descriptors"], t["thumb_size"])
    return mc, fi


def leg_expand_c3(ctx, reps=3):
    """BASELINE.json configs[2]: one 24-MP image pair (6000 x 4000, 300k keypoints per side,
    default options), fastmatch.match()(0.7) through the device-resident expansion loop."""
    from fastmatch_amd import fastmatch
    t0 = time.perf_counter()
    mc, fi = build_image_pair(ctx, (6000, 4000), 300000, 20250003, 2000)
    setup_s = time.perf_counter() - t0
    stats = {}
    get = fastmatch.match(mc, fi, {"context": ctx, "stats": stats, "return_arrays": True})
    t0 = time.perf_counter()
    get.expander()                                      # cell packing, position index, upload, first run state
    build_s = time.perf_counter() - t0
    get(TAU)                                            # warms the module
    best = None
    for _ in range(reps):
        stats.clear()
        ctx.reset_stats()
        t0 = time.perf_counter()
        index, pos, ratio = get(TAU)
        wall = time.perf_counter() - t0
        k_ms = ctx.stats()["kernel_ms"]
        if best is None or wall < best[0]:
            best = (wall, k_ms, stats.get("rounds", 0), stats.get("pairs", 0), len(index))
    wall, k_ms, rounds, pairs, nm = best
    return {"workload": "BASELINE configs[2]: 6000x4000 pair, 300k keypoints/side, grid 50, margin 25, radius 100, tau 0.7",
            "wall_s": wall, "rounds": rounds, "descriptor_pairs": pairs, "matches": nm,
            "rounds_per_s": rounds / wall, "pairs_per_s": pairs / wall, "matches_per_s": nm / wall,
            "kernel_ms": k_ms, "frac_wall_in_kernels": k_ms * 1e-3 / wall, "device_loop": rounds > 0,
            "setup_s": setup_s, "expander_build_s": build_s,
            "note": "fm_expand_run (K7): exact depth-first replay of do_iter on the device, results fetched to the host; "
                    "pairs = sum over rounds of nq_i x nt_i; setup_s = synthetic pair + Metric_Cache (self 2-NN of 300k rows), "
                    "expander_build_s = cell packing + position index + upload (once per pair, reused by every threshold)"}, get


def leg_expand_c3_clustered(ctx, uniform_rounds_per_s, reps=2):
    """configs[2] on CLUSTERED keypoints (real SIFT crowds on texture; the uniform synthetic pair hides every capacity of
    the device loop): the same 6000 x 4000 image and 300k keypoints per side, half of them in 12 Gaussian blobs
    (sigma 60 px) -- radius subsets of ~10 000 rows and cells of thousands where r03 gave the whole pair back to the
    host loop at 4096.  The device loop takes such rounds in chunks (expand.hip, HUGE tier)."""
    from fastmatch_amd import fastmatch
    mc, fi = build_image_pair(ctx, (6000, 4000), 300000, 20250004, 2000, p=0.15, clusters=12, cluster_sigma=60.0, cluster_frac=0.5)
    qpos = mc.original["positions"]
    dens = max(int((((qpos - c) ** 2).sum(axis=1) <= 100.0 ** 2).sum()) for c in qpos[::2503])
    stats = {}
    get = fastmatch.match(mc, fi, {"context": ctx, "stats": stats, "return_arrays": True})
    get.expander()
    t0 = time.perf_counter()
    get(TAU)                                            # first run: climbs the capacity tiers (2048 -> 4096 -> chunked)
    first = time.perf_counter() - t0
    best = None
    for _ in range(reps):
        stats.clear()
        t0 = time.perf_counter()
        index, pos, ratio = get(TAU)
        wall = time.perf_counter() - t0
        if best is None or wall < best[0]:
            best = (wall, stats.get("rounds", 0), stats.get("pairs", 0), len(index), stats.get("device_fallbacks", 0))
    wall, rounds, pairs, nm, fb = best
    # the same run with every round's cross-check done by the run's own workgroup (no delegation to the dense kernels)
    dmin = ctx.get_option("expand_delegate")
    ctx.set_option("expand_delegate", 0)
    try:
        t0 = time.perf_counter()
        get(TAU)
        own_wall = time.perf_counter() - t0
    finally:
        ctx.set_option("expand_delegate", dmin)
    return {"workload": "configs[2] geometry, 300k keypoints/side, half of them in 12 Gaussian blobs (sigma 60 px), p 0.15, tau 0.7",
            "delegate_min_descriptor_pairs": dmin, "wall_s_without_delegation": own_wall,
            "largest_radius_subset_sampled": dens, "wall_s": wall, "first_run_wall_s": first, "rounds": rounds,
            "descriptor_pairs": pairs, "matches": nm, "rounds_per_s": rounds / wall, "pairs_per_s": pairs / wall,
            "device_fallbacks": fb, "uniform_rounds_per_s": uniform_rounds_per_s,
            "note": "first_run_wall_s includes the runs in the 2048- and 4096-row kernels that end at the first oversize subset; "
                    "later runs of the pair start in the chunked variant (tier hint).  Rounds of >= delegate_min descriptor pairs park "
                    "the run: their cross-check is K1 + the election on the whole GPU (fm_expand_run), the run resumes at steps 4 / 5; "
                    "wall_s_without_delegation = the round's own workgroup does every cross-check in chunks"}


def leg_expand_c3_taus(ctx, get, single_wall):
    """The reference's driver asks a pair for 15 thresholds (turntable.py:59-60; Evaluate Turntable.ipynb uses
    numpy.linspace(0.5, 1.2, 15)): 15 runs of the configs[2] pair in ONE launch (one workgroup and one run
    state each) against the same 15 runs one launch after the other.  Thresholds 0.5 .. 1.0 here: above 1.0
    this synthetic pair accepts every cross-checked pair and the expansion heads for all 9801^2 (cell, query
    cell) combinations -- minutes of rounds in the reference's own semantics, not a measurement of anything."""
    from fastmatch_amd import fastmatch
    taus = [float(t) for t in np.linspace(0.5, 1.0, 15)]
    ex = get.expander()
    seeds = [get.seeds_for(t) for t in taus]
    res = fastmatch.run_device_loops(ctx, [ex] * len(taus), seeds, taus, as_arrays=True)      # warm: allocates the run states
    best = None
    for _ in range(2):
        stats = {}
        ctx.reset_stats()
        t0 = time.perf_counter()
        res = fastmatch.run_device_loops(ctx, [ex] * len(taus), seeds, taus, stats=stats, as_arrays=True)
        wall = time.perf_counter() - t0
        k_ms = ctx.stats()["kernel_ms"]
        if best is None or wall < best[0]:
            best = (wall, k_ms, stats.get("rounds", 0), stats.get("pairs", 0), stats.get("device_fallbacks", 0))
    wall, k_ms, rounds, pairs, failed = best
    t0 = time.perf_counter()
    seq_rounds = []
    for t, sd in zip(taus, seeds):
        st1 = {}
        fastmatch.run_device_loops(ctx, [ex], [sd], [t], stats=st1, as_arrays=True)
        seq_rounds.append(st1.get("rounds", 0))
    seq_wall = time.perf_counter() - t0
    return {"workload": "BASELINE configs[2] pair at 15 thresholds linspace(0.5, 1.0, 15), one launch",
            "thresholds": taus, "wall_s": wall, "kernel_ms": k_ms, "rounds": rounds, "descriptor_pairs": pairs,
            "matches": int(sum(len(r[0]) for r in res if r is not None)), "runs_given_up_by_the_device": failed,
            "rounds_per_s": rounds / wall, "pairs_per_s": pairs / wall,
            "rounds_per_run": seq_rounds, "sequential_wall_s": seq_wall, "speedup_vs_sequential": seq_wall / wall,
            "wall_vs_single_tau_0.7": wall / single_wall,
            "note": "each run is a sequential chain of rounds on one workgroup, so the launch lasts as long as its LONGEST run "
                    "(the largest threshold); the other 14 run beside it"}


def leg_train_sharded(ctx, rank, world, dev, backend, reps=5):
    """N > 1 only, not part of the headline: BASELINE config 2 as ONE problem (second half; the reference
    call is fastmatch.pyx:161-162 on one image pair too large or too urgent for one GPU).  The query bank is replicated,
    the train rows are split by row range; every rank runs the reverse-NN + election of its shard
    (fm_xcheck1_keys_dev) and ONE all-reduce(min) of NQ packed 64-bit keys (RCCL) leaves the cross-checked 1-NN of the
    whole problem on every rank.  Reported: wall per problem (max over ranks), the time of the shard kernels alone, the
    collective's bytes, and whether the result is the single-GPU fm_xcheck1's bit for bit (checked on rank 0)."""
    import torch
    import torch.distributed as dist
    from fastmatch_amd import synth, sharding
    Q, T, _ = synth.planted_pair(NQ, NT, seed=SEED)                     # the same problem on every rank
    lo, hi = sharding.shard_rows(NT, rank, world)
    qb, tb = ctx.bank(Q), ctx.bank(T[lo:hi])
    on_gpu = backend == "nccl"
    keys_t = torch.empty(NQ, dtype=torch.int64, device=dev) if on_gpu else None

    def barrier():
        dist.barrier()

    def shard_only():                                                   # the kernels of this rank's shard, keys in HBM
        if on_gpu:
            ctx.xcheck1_keys_dev(qb, tb, lo, keys_t.data_ptr())
            ctx.sync()
        else:
            ctx.xcheck1_keys(qb, tb, lo)

    tidx, d = sharding.xcheck1_sharded(ctx, qb, tb, lo, device=dev)     # warm-up (communicator, workspaces)
    shard_only()
    t_all, t_shard = [], []
    for _ in range(reps):
        barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        tidx, d = sharding.xcheck1_sharded(ctx, qb, tb, lo, device=dev)
        torch.cuda.synchronize(); barrier()
        t_all.append(time.perf_counter() - t0)
        barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        shard_only()
        torch.cuda.synchronize(); barrier()
        t_shard.append(time.perf_counter() - t0)
    tt = torch.tensor([min(t_all), min(t_shard)], dtype=torch.float64, device=dev if on_gpu else "cpu")
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    same = None
    if rank == 0:
        fb = ctx.bank(T)
        ft, fd = ctx.xcheck1(qb, fb)                                    # the unsharded call
        same = bool(np.array_equal(ft, tidx) and np.array_equal(fd.view(np.uint32), d.view(np.uint32)))
        fb.close()
    qb.close(); tb.close()
    wall, shard = float(tt[0].item()), float(tt[1].item())
    return {"mode": "one %d x %d problem, train rows sharded over %d ranks" % (NQ, NT, world),
            "wall_ms": 1e3 * wall, "shard_kernels_ms": 1e3 * shard, "exchange_exposed_ms": 1e3 * (wall - shard),
            "pairs_per_s": float(NQ) * NT / wall, "identical_to_single_gpu": same,
            "collective": {"op": "all_reduce(min)", "backend": ("rccl (torch.distributed nccl)" if on_gpu else backend),
                           "elements": NQ, "bytes_per_rank": NQ * 8, "collectives_per_problem": 1},
            "note": "un-timed leg; wall = barrier-bracketed best of %d, max over ranks, results on the host of every rank; "
                    "exposed = wall - the shard's kernels alone (the collective, the key decode and the copy to the host)" % reps}


def leg_expand_c4(ctx, rank, world, dev, backend, n_pairs=int(os.environ.get("FM_BENCH_C4_PAIRS", "64")), reps=3):
    """BASELINE.json configs[3]: 64 independent 1-MP pairs (1000 x 1000, 12.5k keypoints per
    side) sharded over the ranks (pair i -> rank i mod N), one launch per rank, one
    variable-length all-gather of the match rows."""
    import torch
    import torch.distributed as dist
    from fastmatch_amd import fastmatch, sharding
    mine = sharding.shard_items(n_pairs, rank, world)
    t0 = time.perf_counter()
    pairs = [build_image_pair(ctx, (1000, 1000), 12500, 20250100 + i, 600) for i in mine]
    setup_s = time.perf_counter() - t0
    prepared, stats = [], {}
    t0 = time.perf_counter()
    fastmatch.match_many(pairs, TAU, {"context": ctx, "prepared_out": prepared, "return_arrays": True})
    build_s = time.perf_counter() - t0                      # grids + seeding + expanders + the first (warming) launch
    tdev = dev if backend == "nccl" else "cpu"
    best = None
    for _ in range(reps):
        stats.clear()
        ctx.reset_stats()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = fastmatch.match_many(pairs, TAU, {"context": ctx, "prepared": prepared, "stats": stats, "return_arrays": True})
        rows = [sharding.pack_matches(np.full(len(r[0]), i, np.int32), r[0], r[2].astype(np.float32))
                for i, r in zip(mine, res)]
        packed = np.concatenate(rows) if rows else np.zeros((0, 3), np.int32)
        gathered = sharding.all_gather_matches(packed, device=tdev)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        wall = time.perf_counter() - t0
        k_ms = ctx.stats()["kernel_ms"]
        tot = torch.tensor([float(stats.get("rounds", 0)), float(stats.get("pairs", 0)), float(len(packed))],
                           dtype=torch.float64, device=tdev)
        tm = torch.tensor([wall, k_ms], dtype=torch.float64, device=tdev)
        if world > 1:
            dist.all_reduce(tot)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        wall, k_ms = float(tm[0].item()), float(tm[1].item())
        assert sum(len(g) for g in gathered) == int(tot[2].item())
        if best is None or wall < best[0]:
            best = (wall, k_ms, int(tot[0].item()), int(tot[1].item()), int(tot[2].item()))
    wall, k_ms, rounds, npairs, nm = best
    # The 64-pair figure cannot scale: a run is ONE workgroup walking a dependent chain of rounds, one MI355X holds 256 of
    # them at once (one per CU: the round's tables fill a CU's LDS), so 64 runs leave 3/4 of ONE GPU idle and 8 GPUs with
    # 8 runs each finish in the time of the longest run (r03: 8 pairs 21.3 ms, 64 pairs 27.1 ms on one GPU).  What the
    # reference's driver asks for is pairs x thresholds (turntable.py:59-60, 15 thresholds): 64 x 15 = 960 independent
    # runs in one launch per rank -- a batch that does fill the GPUs, reported beside the literal configs[3] number.
    sat = None
    if os.environ.get("FM_BENCH_C4_SATURATE", "1") != "0":
        taus = [float(t) for t in np.linspace(0.5, 1.0, 15)]
        exs, sds, tts = [], [], []
        for p in prepared:
            if p["expander"] in (None, False):
                continue
            for t in taus:
                exs.append(p["expander"]); sds.append(p["seeds"][p["ratios"] < t]); tts.append(t)
        sbest = None
        for _ in range(2):
            sst = {}
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sres = fastmatch.run_device_loops(ctx, exs, sds, tts, stats=sst, as_arrays=True)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            swall = time.perf_counter() - t0
            stot = torch.tensor([float(sst.get("rounds", 0)), float(sst.get("pairs", 0)), float(sum(len(r[0]) for r in sres if r is not None)),
                                 float(len(exs)), float(sst.get("device_fallbacks", 0))], dtype=torch.float64, device=tdev)
            stm = torch.tensor([swall], dtype=torch.float64, device=tdev)
            if world > 1:
                dist.all_reduce(stot)
                dist.all_reduce(stm, op=dist.ReduceOp.MAX)
            if sbest is None or float(stm[0].item()) < sbest[0]:
                sbest = (float(stm[0].item()),) + tuple(float(x) for x in stot.tolist())
        swall, srounds, spairs, smatches, sruns, sfb = sbest
        sat = {"workload": "the same %d pairs x 15 thresholds linspace(0.5, 1.0, 15) = %d independent runs, one launch per rank" % (n_pairs, int(sruns)),
               "runs": int(sruns), "runs_per_gpu": int(sruns) // world, "runs_in_flight_per_gpu": 256, "wall_s": swall,
               "rounds": int(srounds), "descriptor_pairs": int(spairs), "matches": int(smatches), "runs_given_up_by_the_device": int(sfb),
               "rounds_per_s": srounds / swall, "pairs_per_s": spairs / swall, "runs_per_s": sruns / swall,
               "note": "strong scaling over the ranks (run k of the global list -> the rank that holds its pair); no gather in this "
                       "figure (the match rows stay on the rank: 44 bytes each)"}
    return {"workload": "BASELINE configs[3]: %d x (1000x1000 pair, 12.5k keypoints/side), pair i -> rank i mod %d, tau 0.7"
                        % (n_pairs, world),
            "runs_in_flight_per_gpu": 256, "saturating_batch": sat,
            "n_gpus": world, "scaling": "strong", "wall_s": wall, "rounds": rounds, "descriptor_pairs": npairs,
            "matches": nm, "rounds_per_s": rounds / wall, "pairs_per_s": npairs / wall, "matches_per_s": nm / wall,
            "image_pairs_per_s": n_pairs / wall, "kernel_ms": k_ms, "frac_wall_in_kernels": k_ms * 1e-3 / wall,
            "setup_s": setup_s, "expander_build_s": build_s,
            "note": "fastmatch.match_many: one fm_expand_run launch per rank (one workgroup per image pair) + "
                    "all-gather of (pair, query index, ratio) rows; wall = max over ranks, best of %d" % reps}


def leg_fresh_pair(ctx, Q, T, rank, resident_counts, steps=8):
    """What a NEW image pair costs end to end (the headline times resident banks with precomputed self
    distances): per step 12 new pairs go upload -> Metric_Cache self distances -> X1 + R1 + compaction, pipelined:
      fm_bank_refill_u8_async  24 banks of 100k rows from page-locked host memory into one of two resident bank sets,
                               on the upload stream, beside the previous step's kernels (no allocation, no host sync)
      fm_upload_fence + fm_self_dist_batch   the 12 query banks' self distances in TWO launches of the triangular top-1
                               sweep (diagonal blocks, then the rest: every distance once), attached on the device
      fm_match_accepted_batch  the headline's step on the refilled banks
    two steps in flight (fm_mark / fm_wait).  Checked: every pair's accepted count equals the resident run's.  Beside it
    the naive flow a caller without the pipeline runs per pair (two fm_bank_create_u8, fm_self_dist,
    fm_bank_set_selfdist, fm_match_accepted)."""
    rng = np.random.default_rng(SEED + 1000 + rank)
    n = PAIRS_PER_STEP
    src = []
    for j in range(n):
        Qj, Tj = derived_pair(Q, T, j, rng)
        pq, pt = ctx.pinned_empty((NQ, 128), np.uint8), ctx.pinned_empty((NT, 128), np.uint8)
        pq[:] = Qj
        pt[:] = Tj
        src.append((pq, pt))
    sets = []
    for s in range(2):
        banks = [(ctx.bank(src[j][0]), ctx.bank(src[j][1])) for j in range(n)]
        outs = [(ctx.pinned_empty(NQ, np.int32), ctx.pinned_empty(NQ, np.int32), ctx.pinned_empty(NQ, np.float32),
                 ctx.pinned_empty(NQ, np.float64)) for _ in range(n)]
        cnts = [ctx.pinned_empty(1, np.int64) for _ in range(n)]
        ctx.self_dist_batch([q for q, _ in banks], want_host=False)
        ctx.sync()
        sets.append({"banks": banks, "outs": outs, "cnts": cnts, "batch": ctx.prepare_batch(banks, outs, cnts)})

    do_upload, do_self = os.environ.get("FM_FRESH_UPLOAD", "1") != "0", os.environ.get("FM_FRESH_SELF", "1") != "0"   # (experiments)

    def enqueue(s):
        st = sets[s]
        if do_upload:
            for j, (qb, tb) in enumerate(st["banks"]):
                qb.refill_async(src[j][0])
                tb.refill_async(src[j][1])
            ctx.upload_fence()
        if do_self:
            ctx.self_dist_batch([q for q, _ in st["banks"]], want_host=False)
        ctx.match_accepted_batch(st["batch"], TAU)
        return ctx.mark()

    def run(k, mark_at=-1):
        """k steps through the pipeline; returns the host time at which step mark_at - 1 was complete (steps
        mark_at .. k - 1 then run in a full pipeline: the first upload of a cold pipeline overlaps nothing)."""
        prev, t_mark = None, None
        for i in range(k):
            tk = enqueue(i % 2)          # (set i % 2 was last used by step i - 2, whose ticket has been waited for)
            if prev is not None:
                ctx.wait(prev)
            if i == mark_at:
                t_mark = time.perf_counter()
            prev = tk
        ctx.wait(prev)
        ctx.sync()
        return t_mark

    run(2)
    ctx.reset_stats()
    warm = 2
    t0 = run(warm + steps, mark_at=warm)
    dt = time.perf_counter() - t0
    st = ctx.stats()
    st["kernel_ms"] *= steps / float(warm + steps)          # (the distance-kernel time of the timed steps' share)
    same = resident_counts is None or all(int(c[0]) == int(r) for stt in sets for c, r in zip(stt["cnts"], resident_counts))
    ms_pair = 1e3 * dt / steps / n
    # upload alone: 24 refills + a sync, nothing else on the device
    ctx.sync()
    t0 = time.perf_counter()
    for j, (qb, tb) in enumerate(sets[0]["banks"]):
        qb.refill_async(src[j][0])
        tb.refill_async(src[j][1])
    ctx.sync()
    up_ms = 1e3 * (time.perf_counter() - t0) / n
    # the naive flow, one pair at a time
    reps = 4
    t0 = time.perf_counter()
    for j in range(reps):
        qb, tb = ctx.bank(src[j][0]), ctx.bank(src[j][1])
        qb.set_selfdist(ctx.self_dist(qb))
        nq_acc = len(ctx.match_accepted(qb, tb, TAU, out=sets[0]["outs"][0])[0])
        qb.close()
        tb.close()
    naive_ms = 1e3 * (time.perf_counter() - t0) / reps
    for stt in sets:
        for qb, tb in stt["banks"]:
            qb.close()
            tb.close()
    ops = 2.0 * float(NQ) * NT * OPS_PER_PAIR                     # self sweep + cross-check sweep per image pair
    return {"workload": "%d NEW 100k x 100k uint8 image pairs per step: upload (2 x 12.8 MB per pair, page-locked source) -> "
                        "self distances (Metric_Cache build) -> X1 + R1 at 0.7 + compaction; two steps in flight" % n,
            "steps": steps, "ms_per_image_pair": ms_pair, "image_pairs_per_s": 1e3 / ms_pair,
            "descriptor_pairs_per_s": 2.0 * float(NQ) * NT / (ms_pair * 1e-3),
            "frac_int8_mfma_peak_over_2e10_pairs": ops / (ms_pair * 1e-3) / (INT8_DENSE_PEAK_TOPS * 1e12),
            "frac_note": "r05: the self sweep is triangular and EVALUATES ~0.51e10 of its 1e10 pairs, so this figure (2e10 pairs a new image "
                         "pair gets, per second, over the peak) is a rate of useful results, not an MFMA utilisation: see self_2nn and roofline",
            "distance_kernel_ms_per_image_pair": st["kernel_ms"] / (steps * n),
            "upload_alone_ms_per_image_pair": up_ms, "upload_gb_per_s": 2 * NQ * 128 / (up_ms * 1e-3) / 1e9,
            "naive_flow_ms_per_image_pair": naive_ms, "accepted_counts_equal_resident_run": bool(same),
            "note": "pipelined: fm_bank_refill_u8_async (upload stream) + fm_upload_fence + fm_self_dist_batch (two launches for the "
                    "12 query banks: triangular top-1 sweep) + fm_match_accepted_batch; naive: fm_bank_create_u8 x 2, fm_self_dist, "
                    "fm_bank_set_selfdist, fm_match_accepted per pair, each synchronous; distance_kernel_ms = both sweeps"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="headline only (profiling runs)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
            # started bare: launch the N ranks as a CHILD process (nothing here has touched the
            # GPU yet, and a process that has must never exec) and pass its exit code on
            port = 29500 + os.getpid() % 2000
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
                   "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
            sys.exit(subprocess.call(cmd))
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d; launch with torch.distributed.run "
                         "--nproc-per-node %d\n" % (args.gpus, world, args.gpus))
        sys.exit(2)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # stdout carries the ONE JSON line and nothing else: libraries that print banners to fd 1 (RCCL's
    # version block) are sent to stderr, the line itself goes to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    # FM_BENCH_BACKEND=gloo + FM_BENCH_SINGLE_DEVICE=1: dry-run of the N > 1 code path on a
    # box with one GPU (all ranks share device 0, collectives on CPU tensors); never used by
    # the driver.
    backend = os.environ.get("FM_BENCH_BACKEND", "nccl")
    if os.environ.get("FM_BENCH_SINGLE_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    dev = torch.device("cuda", local_rank) if backend == "nccl" else "cpu"
    legs = not args.no_legs

    # Steps are pipelined two deep: step i + 1 is enqueued before the host waits for step i (fm_mark / fm_wait, two
    # output sets), so a step's twelve pairs share ONE distance-kernel launch -- its small kernels run beside the next
    # step's launch.  (r05: the per-pair, blocking and unpipelined step variants of r02 - r04 are gone; their A/Bs are
    # in DESIGN.md section 6.)
    import fastmatch_amd
    from fastmatch_amd import synth, sharding
    ctx = fastmatch_amd.Context(local_rank)
    if "FM_BATCH_GROUP" not in os.environ:                  # the whole step in one launch (fm_ctx_set_option)
        ctx.set_option("batch_group", 16)
    if "FM_BATCH_TAIL" not in os.environ:
        ctx.set_option("batch_tail", 0)

    # the batch of independent pairs of this rank, resident in HBM before the timed region
    Q, T, planted = synth.planted_pair(NQ, NT, seed=SEED + rank)
    rng = np.random.default_rng(SEED + 1000 + rank)
    banks = []
    self_s = self_kernel_ms = None
    for j in range(PAIRS_PER_STEP):
        Qj, Tj = derived_pair(Q, T, j, rng)
        qb, tb = ctx.bank(Qj), ctx.bank(Tj)
        selfdist = ctx.self_dist(qb)                    # Metric_Cache build (once per query image)
        if j == 0:
            selfdist0 = selfdist
        if j == 1 or (j == 0 and PAIRS_PER_STEP == 1):  # timed on a later run (the first pays module load)
            ctx.reset_stats()
            t0 = time.perf_counter()
            ctx.self_dist(qb)
            self_s = time.perf_counter() - t0
            self_kernel_ms = ctx.stats()["kernel_ms"]
        qb.set_selfdist(selfdist)
        banks.append((qb, tb))
        del Qj, Tj

    # N = 1: caller-owned output buffers in page-locked memory, one set per pair of the batch and pipeline slot: the
    # compaction kernel writes them directly (fm_match_accepted_batch), the host reads step i while step i + 1 runs
    use_async = world == 1
    outbufs_sets, counts_sets, batch_sets = [], [], []
    if world == 1:
        for _ in range(2):
            ob = [(ctx.pinned_empty(NQ, np.int32), ctx.pinned_empty(NQ, np.int32),
                   ctx.pinned_empty(NQ, np.float32), ctx.pinned_empty(NQ, np.float64)) for _ in range(PAIRS_PER_STEP)]
            cn = [ctx.pinned_empty(1, np.int64) for _ in range(PAIRS_PER_STEP)]
            outbufs_sets.append(ob)
            counts_sets.append(cn)
            batch_sets.append(ctx.prepare_batch(banks, ob, cn))
    outbufs = outbufs_sets[0] if world == 1 else [(ctx.pinned_empty(NQ, np.int32), ctx.pinned_empty(NQ, np.int32),
                                                  ctx.pinned_empty(NQ, np.float32), ctx.pinned_empty(NQ, np.float64))]
    outbuf = outbufs[0]

    # N > 1 (BASELINE.json: "RCCL over xGMI only for result gather"): the rows never leave HBM -- the step's pairs share
    # distance-kernel launches (fm_match_accepted_dev_batch writes every pair's accepted rows into the gatherer's send
    # buffer) and ONE all-gather per step ships them on a stream of its own, beside the next step's kernels.
    #   default               padded: every rank ships PAIRS_PER_STEP x NQ row slots, no host wait
    #   FM_BENCH_GATHER=counted   counts first, then only the rows that are there (a third of the bytes, one host wait per
    #                         step; sharding.MatchGatherer two_phase) -- the tested alternative, for the day the padded
    #                         all-gather shows in the 8-GPU curve
    gather_mode = "counted" if os.environ.get("FM_BENCH_GATHER") == "counted" else "padded"
    gatherer = sharding.MatchGatherer(dev, capacity=NQ, fill_device=torch.device("cuda", local_rank), pairs_per_step=PAIRS_PER_STEP,
                                      two_phase=gather_mode == "counted") if world > 1 else None
    pair_args = ctx.prepare_pairs(banks) if world > 1 else None
    h_counts_sets = [ctx.pinned_empty(PAIRS_PER_STEP, np.int64), ctx.pinned_empty(PAIRS_PER_STEP, np.int64)]

    pipe = {"i": 0, "prev": None, "last": 0}

    def consume(prev):                                      # the host reads step i's results while step i + 1 runs
        s, ticket = prev
        ctx.wait(ticket)
        if world == 1:
            return int(sum(int(c[0]) for c in counts_sets[s]))
        return int(h_counts_sets[s].sum())

    def finish_steps():
        if pipe["prev"] is not None:
            pipe["last"] = consume(pipe["prev"])
            pipe["prev"] = None
        return pipe["last"]

    def step(gather=True):
        """One pass of the hot path over this rank's batch; gather=False (N > 1, measurement only): the same kernels
        without the all-gather, for `collective.gather_exposed_ms`."""
        s = pipe["i"] % 2
        pipe["i"] += 1
        if world == 1:
            ctx.match_accepted_batch(batch_sets[s], TAU)
        else:
            rows, cnts = gatherer.send_buffers()
            ctx.match_accepted_dev_batch(pair_args, TAU, rows.data_ptr(), cnts.data_ptr(), NQ, h_counts=h_counts_sets[s],
                                         consumer_stream=gatherer.consumer_stream())
            if gather:
                gatherer.submit_device()
        ticket = ctx.mark()
        if pipe["prev"] is not None:
            pipe["last"] = consume(pipe["prev"])
        pipe["prev"] = (s, ticket)
        return pipe["last"]

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    finish_steps()
    if gatherer is not None:
        gatherer.finish()
    ctx.sync()
    ctx.reset_stats()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    npass = 0
    for _ in range(args.steps):
        npass = step()
    npass = finish_steps()                                  # the last step's results, still inside the timed region
    if gatherer is not None:
        gatherer.finish()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        tot_pass = torch.tensor([npass], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tot_pass, op=dist.ReduceOp.SUM)
        npass_all = int(tot_pass.item())
    else:
        npass_all = npass
    st = ctx.stats()
    # N > 1: what the collective is and what it costs, so that a 1 -> 8 curve explains itself.  Measured once, OUTSIDE
    # the timed region: the same steps with and without the all-gather (same kernels, same pipeline).
    collective = None
    if world > 1:
        ones = torch.ones(1, dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        ksteps = max(2, min(args.steps, 6))

        def timed_steps(gather):
            barrier()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(ksteps):
                step(gather)
            finish_steps()
            if gather:
                gatherer.finish()
            torch.cuda.synchronize()
            barrier()
            return (time.perf_counter() - t) / ksteps

        with_g, without_g = timed_steps(True), timed_steps(False)
        tt = torch.tensor([with_g, without_g], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        collective = {"backend": ("rccl (torch.distributed nccl)" if backend == "nccl" else backend), "world_size": world,
                      "ranks_seen": int(ones.item()), "gather": gather_mode, "collectives_per_step": 1 if gather_mode == "padded" else 2,
                      "rows_per_rank_per_step": int(gatherer.rows_shipped),
                      "bytes_per_rank_per_step": int(gatherer.rows_shipped) * 12 + PAIRS_PER_STEP * 8,
                      "bytes_received_per_rank_per_step": (world - 1) * (int(gatherer.rows_shipped) * 12 + PAIRS_PER_STEP * 8),
                      "step_ms_with_gather": 1e3 * float(tt[0].item()), "step_ms_without_gather": 1e3 * float(tt[1].item()),
                      "gather_exposed_ms": 1e3 * float(tt[0].item() - tt[1].item()),
                      "note": "one all-gather of every rank's accepted rows (int32 query, train, distance bits) + counts per step, on "
                              "a stream of its own beside the next step's kernels; exposed = step wall with - without it, max over "
                              "ranks, %d steps each, measured after the timed region" % ksteps}
    per_pair_counts = ([int(c[0]) for c in counts_sets[0]] if world == 1 else [int(c) for c in h_counts_sets[0]])
    qb, tb = banks[0]
    # pair 0's accepted rows as the timed batch left them (output set 0), kept for the self-check below
    timed_rows0 = None
    if world == 1 and use_async:
        m0 = int(counts_sets[0][0][0])
        timed_rows0 = tuple(a[:m0].copy() for a in outbufs[0])

    # Classic Ratio-Match (2-NN + d1/d2 < 0.7, the literal "2-NN + ratio" of configs[1]), reported
    # beside the headline; not part of the timed region above.
    crm = None
    if rank == 0 and legs:
        ctx.knn2_ratio(qb, tb, TAU, out=outbuf)
        ctx.reset_stats()
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            cq, _, _, _ = ctx.knn2_ratio(qb, tb, TAU, out=outbuf)
        dt = (time.perf_counter() - t0) / reps
        crm = {"pairs_per_s": float(NQ) * NT / dt, "matches_per_s": len(cq) / dt, "accepted": int(len(cq)),
               "ms_per_call": 1e3 * dt, "kernel_ms": ctx.stats()["kernel_ms"] / reps,
               "frac_int8_mfma_peak": float(NQ) * NT * OPS_PER_PAIR / (ctx.stats()["kernel_ms"] / reps * 1e-3) / (INT8_DENSE_PEAK_TOPS * 1e12),
               "note": "fm_knn2_ratio: K2 top-2 + Lowe ratio + compaction, same banks"}

    # Metric_Cache build (fm_self_dist): the single call timed during set-up (between bank uploads, clock still ramping)
    # and, for the kernel's own rate, the mean of ten back-to-back calls.  r05: the sweep is TRIANGULAR (d(i, j) = d(j, i):
    # every tile above the diagonal is computed once and serves both of its rows), so the matrix cores evaluate about half
    # of the n x n pairs the caller gets -- both figures are reported, the fraction of the MFMA roof over the EVALUATED ones.
    from fastmatch_amd import _ffi
    tri_table, _, tri_stages = _ffi.self_dist_plan(((NQ + 127) // 128) * 128)
    evaluated = float((tri_table[:, 2] - tri_table[:, 1]).sum()) * 128.0 * 512.0       # stages x rows per stage x output rows per workgroup
    tri_on = ctx.get_option("self_tri") == 1 and NQ >= 32768 - 127
    if not tri_on:
        evaluated = float(NQ) * NQ
    self2 = {"pairs_per_s": float(NQ) * NQ / (self_kernel_ms * 1e-3), "kernel_ms": self_kernel_ms, "wall_s": self_s,
             "kernel": ("fm::rowreduce_tri_kernel<1> (K1's top-1 body, triangular: launch A = the masked diagonal blocks, launch B = the "
                        "rest, both directions of every tile; %d stages per workgroup)" % tri_stages) if tri_on else
                       "fm::rowreduce_kernel<4,1,true,8,3,1,true> (K1's top-1 kernel, diagonal masked)",
             "distance_evaluations": evaluated, "evaluated_fraction_of_n_squared": evaluated / (float(NQ) * NQ),
             "note": "Metric_Cache build, 100k x 100k self distances = min over j != i (the second entry of the self 2-NN), outside the "
                     "timed region; kernel_ms = one call during set-up, kernel_ms_steady = mean of 10 back-to-back calls; pairs_per_s "
                     "counts the n x n pairs the caller asked for, frac_int8_mfma_peak_steady the distance evaluations the kernel made"}
    if rank == 0 and legs:
        ctx.self_dist(qb)
        ctx.reset_stats()
        for _ in range(10):
            ctx.self_dist(qb)
        s2 = ctx.stats()
        self2["kernel_ms_steady"] = s2["kernel_ms"] / max(s2["kernel_launches"], 1)
        self2["pairs_per_s_steady"] = float(NQ) * NQ / (self2["kernel_ms_steady"] * 1e-3)
        self2["frac_int8_mfma_peak_steady"] = evaluated * OPS_PER_PAIR / (self2["kernel_ms_steady"] * 1e-3) / (INT8_DENSE_PEAK_TOPS * 1e12)
        if tri_on:                                           # the masked full sweep of r04 beside it, same box, same process
            ctx.set_option("self_tri", 0)
            ctx.self_dist(qb)
            ctx.reset_stats()
            for _ in range(10):
                ctx.self_dist(qb)
            s3 = ctx.stats()
            ctx.set_option("self_tri", 1)
            self2["kernel_ms_steady_full_sweep"] = s3["kernel_ms"] / max(s3["kernel_launches"], 1)
        if PAIRS_PER_STEP > 1:                               # the way a dataset's Metric_Caches are built: all query banks in ONE call
            qbs = [q for q, _ in banks]                      # (two launches for all of them; the same values are attached again)
            ctx.self_dist_batch(qbs, want_host=False)
            ctx.sync()                                       # (the call is enqueue-only: its events are read at the next sync)
            ctx.reset_stats()
            t0 = time.perf_counter()
            for _ in range(5):
                ctx.self_dist_batch(qbs, want_host=False)
            ctx.sync()
            self2["ms_per_bank_in_a_batch"] = (time.perf_counter() - t0) * 1e3 / (5.0 * len(qbs))
            s4 = ctx.stats()
            self2["kernel_ms_steady_per_bank_in_a_batch"] = s4["kernel_ms"] / max(s4["kernel_launches"], 1) / len(qbs)
            self2["banks_per_batch_call"] = len(qbs)

    # ONE configs[1] call, the way a caller without a batch makes it: fm_match_accepted (K1 + election +
    # ratio test + compaction into page-locked buffers) and its synchronisation, 20 repetitions.
    single = None
    if rank == 0 and legs:
        ctx.match_accepted(qb, tb, TAU, out=outbuf)
        ctx.reset_stats()
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            sq, _, _, _ = ctx.match_accepted(qb, tb, TAU, out=outbuf)
        dt = (time.perf_counter() - t0) / reps
        sst = ctx.stats()
        skms = sst["kernel_ms"] / max(sst["kernel_launches"], 1)
        single = {"ms_per_call": 1e3 * dt, "kernel_ms": skms, "pairs_per_s": float(NQ) * NT / dt, "accepted": int(len(sq)),
                  "frac_int8_mfma_peak": float(NQ) * NT * OPS_PER_PAIR / (skms * 1e-3) / (INT8_DENSE_PEAK_TOPS * 1e12),
                  "frac_int8_mfma_peak_whole_call": float(NQ) * NT * OPS_PER_PAIR / dt / (INT8_DENSE_PEAK_TOPS * 1e12),
                  "note": "fm_match_accepted, one 100k x 100k pair per call, synchronous (rowreduce_kernel<4,1,true,8,3,1>)"}

    # Float32 route (BASELINE.json configs[4]: 1M-row float32 target bank vs 10k-row query
    # batches): fm_knn2 on non-integer descriptors = K8 (fp16-MFMA filter + exact float32
    # rescoring).  Reported beside the headline at N = 1; FM_BENCH_F32=0 skips it.
    f32 = None
    if rank == 0 and world == 1 and legs and os.environ.get("FM_BENCH_F32", "1") != "0":
        rng5 = np.random.default_rng(20250005)
        n_bank, n_query = 1000000, 10000
        Tf = synth.synth_sift(n_bank, rng5).astype(np.float32) + rng5.uniform(-0.5, 0.5, (n_bank, 128)).astype(np.float32)
        Qf = synth.synth_sift(n_query, rng5).astype(np.float32) + rng5.uniform(-0.5, 0.5, (n_query, 128)).astype(np.float32)
        tbf, qbf = ctx.bank(Tf), ctx.bank(Qf)
        Sf = Tf[:100000].copy()               # (the self-distance entry below)
        del Tf
        f32 = {}
        for name, fn in (("knn2", lambda: ctx.knn2(qbf, tbf)), ("xcheck1", lambda: ctx.xcheck1(qbf, tbf))):
            fn()
            ctx.reset_stats()
            t0 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                fn()
            dt = (time.perf_counter() - t0) / reps
            stf = ctx.stats()
            kms = stf["kernel_ms"] / max(stf["kernel_launches"], 1)
            f32[name] = {"pairs_per_s": float(n_bank) * n_query / (kms * 1e-3), "kernel_ms": kms, "ms_per_call": 1e3 * dt,
                         "frac_fp16_mfma_peak": float(n_bank) * n_query * 256 / (kms * 1e-3) / 2.5e15}
        launches, redone = ctx.f32_filter_stats()
        f32.update({"filtered_calls": launches, "redone_by_all_pairs_kernel": redone,
                    "note": "fm_knn2 / fm_xcheck1, 10k x 1M non-integer float32 descriptors: fp16 MFMA filter (256 flop/pair) "
                            "+ exact float32 rescoring, results bit-identical to the all-pairs float32 chain"})
        # backward-compatible flat keys = the knn2 call (BASELINE configs[4] as benchmarked in r01)
        f32.update({k: f32["knn2"][k] for k in ("pairs_per_s", "kernel_ms", "ms_per_call", "frac_fp16_mfma_peak")})
        qbf.close()
        # configs[4] asks for ten query batches: 2-NN rows are independent, so the ten batches can go
        # through ONE call (100k output rows: fewer splits per output chunk, longer sweeps per workgroup)
        n_batches = 10
        Qall = np.concatenate([Qf] + [synth.synth_sift(n_query, rng5).astype(np.float32)
                                      + rng5.uniform(-0.5, 0.5, (n_query, 128)).astype(np.float32) for _ in range(n_batches - 1)])
        qall = ctx.bank(Qall)
        ctx.knn2(qall, tbf)
        ctx.reset_stats()
        t0 = time.perf_counter()
        ctx.knn2(qall, tbf)
        ctx.knn2(qall, tbf)
        dt = (time.perf_counter() - t0) / 2
        stf = ctx.stats()
        kms = stf["kernel_ms"] / max(stf["kernel_launches"], 1)
        f32["knn2_ten_batches_one_call"] = {
            "pairs_per_s": float(n_bank) * n_query * n_batches / (kms * 1e-3), "kernel_ms": kms, "ms_per_call": 1e3 * dt,
            "ms_per_batch": 1e3 * dt / n_batches,
            "frac_fp16_mfma_peak": float(n_bank) * n_query * n_batches * 256 / (kms * 1e-3) / 2.5e15}
        del Qall
        qall.close()
        tbf.close()
        # Metric_Cache build of a RootSIFT-style query image (cache.pyx:250-252, 271-273): the self distances of a 100k-row
        # float32-route bank -- the masked full sweep against the triangular one (r06: every distance once)
        sb = ctx.bank(Sf)
        keep_tri = ctx.get_option("self_tri")
        sd_ms, sd_out = {}, {}
        for name, opt in (("masked_full_sweep", 0), ("triangular_sweep", 2)):
            ctx.set_option("self_tri", opt)
            sd_out[name] = ctx.self_dist(sb)
            ctx.reset_stats()
            for _ in range(3):
                ctx.self_dist(sb)
            sst = ctx.stats()
            sd_ms[name] = sst["kernel_ms"] / max(sst["kernel_launches"], 1)
        ctx.set_option("self_tri", keep_tri)
        f32["self_dist_100k"] = {
            "kernel_ms_masked_full_sweep": sd_ms["masked_full_sweep"], "kernel_ms_triangular_sweep": sd_ms["triangular_sweep"],
            "ratio": sd_ms["triangular_sweep"] / sd_ms["masked_full_sweep"],
            "identical": bool(np.array_equal(sd_out["masked_full_sweep"].view(np.uint64), sd_out["triangular_sweep"].view(np.uint64))),
            "frac_fp16_mfma_peak_over_n_squared": 1e10 * 256 / (sd_ms["triangular_sweep"] * 1e-3) / 2.5e15,
            "note": "fm_self_dist of a 100k-row float32-route bank (the default rule takes the triangular sweep from 65536 padded rows)"}
        sb.close()

    fresh = None
    if rank == 0 and world == 1 and legs and use_async and os.environ.get("FM_BENCH_FRESH", "1") != "0":
        fresh = leg_fresh_pair(ctx, Q, T, rank, [int(c[0]) for c in counts_sets[0]])

    c3 = c3t = None
    if rank == 0 and world == 1 and legs and os.environ.get("FM_BENCH_C3", "1") != "0":
        c3, c3_get = leg_expand_c3(ctx)
        c3t = leg_expand_c3_taus(ctx, c3_get, c3["wall_s"])
        del c3_get
        if os.environ.get("FM_BENCH_C3_CLUSTERED", "1") != "0":
            c3["clustered"] = leg_expand_c3_clustered(ctx, c3["rounds_per_s"])
    c4 = None
    if legs and os.environ.get("FM_BENCH_C4", "1") != "0":
        c4 = leg_expand_c4(ctx, rank, world, dev, backend)
    # N > 1: the OTHER multi-GPU mode of SURVEY.md 8(e), un-timed, so that a multi-GPU record covers both halves
    tsh = leg_train_sharded(ctx, rank, world, dev, backend) if world > 1 and os.environ.get("FM_BENCH_TRAIN_SHARDED", "1") != "0" else None

    if rank == 0:
        pairs_per_step = float(NQ) * NT * PAIRS_PER_STEP
        value = world * pairs_per_step * args.steps / elapsed
        # time of the distance kernel per image pair: a launch may hold several pairs (fm_match_accepted_batch)
        timed_image_pairs = st["pairs"] / (float(NQ) * NT)
        k_ms = st["kernel_ms"] / max(timed_image_pairs, 1e-9)
        pairs_per_launch = timed_image_pairs / max(st["kernel_launches"], 1)
        call_ms = st["total_ms"] / max(st["calls"], 1)
        achieved = float(NQ) * NT * OPS_PER_PAIR / (k_ms * 1e-3) / 1e12
        # HBM bytes per K1 launch come from the committed rocprofv3 PMC passes (not live): they are
        # attached only if they were collected from exactly the kernel source that is built here
        traffic = traffic_src = traffic_tag = busy = None
        traffic_note = "no profiles/latest_pmc.json"
        try:
            with open(PROFILE_JSON) as f:
                pmc = json.load(f)
            if pmc.get("k1_source_sha256") == k1_source_hash():
                traffic, traffic_src, traffic_tag = pmc["hbm_bytes_per_launch"], pmc["source"], pmc.get("tag")
                if pmc.get("image_pairs_per_launch"):       # counters are per dispatch: rescale to this run's launches
                    traffic *= pairs_per_launch / float(pmc["image_pairs_per_launch"])
                busy = pmc.get("mfma_pipe_busy_frac")
                traffic_note = "counters collected from this kernel source (sha256 match)"
            else:
                traffic_note = ("profiles/latest_pmc.json (tag %s) was collected from a different rowreduce.hip/tile_ops.h; "
                                "rerun scripts/profile.sh" % pmc.get("tag"))
        except Exception:
            pass
        out = {
            "metric": "descriptor-pair distances/sec (cross-checked 1-NN + ratio test at 0.7)",
            "value": value,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "ms_per_image_pair": 1e3 * elapsed / args.steps / PAIRS_PER_STEP,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int8",
            "data": "synthetic",
            "config": {"workload": "batch of %d independent image pairs per GPU and step, each %dk x %dk synthetic 128-D uint8 "
                                   "SIFT descriptors, brute-force cross-checked 1-NN + ratio 0.7 (BASELINE.json configs[1])"
                                   % (PAIRS_PER_STEP, NQ // 1000, NT // 1000),
                       "nq": NQ, "nt": NT, "dim": 128, "tau": TAU, "pairs_per_gpu": PAIRS_PER_STEP,
                       "parallelism": "independent image pairs sharded over GPUs; RCCL all-gather of accepted matches from device buffers"},
            "matches_per_s": npass_all * args.steps / elapsed,
            "accepted_matches_per_step": npass_all,
            "accepted_matches_pair0": per_pair_counts[0],
            "accepted_matches_independent_pair": per_pair_counts[-1] if PAIRS_PER_STEP > 1 else None,
            "batch_note": "pairs 1 .. n-2 of a rank's batch are row permutations + dimension rolls of pair 0 (same distance "
                          "distribution, other addresses); the last pair is drawn independently with 55 % planted rows and noise "
                          "sigma 10 (pair 0: 30 %, sigma 6): its accepted count stands beside pair 0's",
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": INT8_DENSE_PEAK_TOPS,
                         "unit": "TOP/s", "frac": achieved / INT8_DENSE_PEAK_TOPS, "traffic": traffic,
                         "traffic_source": traffic_src, "traffic_tag": traffic_tag, "traffic_note": traffic_note,
                         "kernel": "fm::rowreduce_batch_kernel<4,1,8,3,1> (v_mfma_i32_16x16x64_i8)",
                         "kernel_ms": k_ms, "kernel_ms_per_launch": k_ms * pairs_per_launch,
                         "image_pairs_per_launch": pairs_per_launch,
                         "kernel_launches_timed": st["kernel_launches"],
                         "algorithmic_bytes_per_launch": st.get("bytes_moved", 0) / max(st["kernel_launches"], 1),
                         "algorithmic_gbps": st.get("bytes_moved", 0) / max(st["kernel_ms"], 1e-9) / 1e6,
                         "hbm_gbps": (traffic / (k_ms * pairs_per_launch * 1e-3) / 1e9) if traffic else None,
                         "hbm_frac_of_8tbps": (traffic / (k_ms * pairs_per_launch * 1e-3) / 8e12) if traffic else None,
                         "mfma_pipe_busy_frac": busy,
                         "note": "int8 ops: 256 per descriptor pair x 1e10 descriptor pairs per image pair x image_pairs_per_launch; "
                                 "algorithmic_bytes_per_launch = fm_stats_ex.bytes_moved / launches (every bank row of a launch read once: "
                                 "2 x 100k x 128 B per image pair) -- 0.03 TB/s, the path is MFMA-bound by a factor of 270; "
                                 "kernel_ms = HIP-event time of the distance-kernel launches of the timed region (events on the "
                                 "library's own stream) per image pair, kernel_ms_per_launch = per launch; traffic / hbm_gbps = PMC HBM bytes "
                                 "per launch / that time; mfma_pipe_busy_frac = rocprofv3 SQ_VALU_MFMA_BUSY_CYCLES per SIMD / GPU "
                                 "cycles of the launch (profiles/)"},
            "collective": collective,
            "train_sharded": tsh,
            "self_2nn": self2,
            "classic_ratio_match": crm,
            "single_pair": single,
            "fresh_pair": fresh,
            "expand_c3": c3,
            "expand_c3_taus": c3t,
            "expand_c4": c4,
            "float32_route": f32,
            "call_ms": call_ms,
            "device": ctx.device_name(),
        }
        verified = None
        if world == 1 and not args.no_cpu_baseline:
            keep = {}
            out["cpu_baseline"] = cpu_baseline(Q, T, keep=keep)
            try:
                out["cpu_baseline_simd"] = cpu_baseline_simd(Q, T)
            except Exception as e:                                   # (a host without AVX2)
                out["cpu_baseline_simd"] = {"value": None, "note": str(e)}
            try:
                out["cpu_baseline_blocked"] = cpu_baseline_blocked(Q, T)
            except Exception as e:                                   # (a host without AVX-512 VNNI)
                out["cpu_baseline_blocked"] = {"value": None, "note": str(e)}
            # self-check: the oracle result just computed against what the timed batch left for pair 0
            verified, scope, n_want = verify_against_oracle(ctx, Q, T, selfdist0, keep, timed_rows0, banks[0])
            out["verified_vs_oracle"] = verified
            out["verified_scope"] = scope + " (%d accepted rows: query index, train index, distance bits, ratio bits)" % n_want
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
        if verified is False:
            sys.stderr.write("bench.py: the device result differs from the oracle\n")
            if world == 1:
                sys.exit(1)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
