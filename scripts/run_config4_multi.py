"""BASELINE.json config 4: a batch of 64 independent 1-MP image pairs sharded over the GPUs of
one node (pair i -> rank i mod N), matching with no communication, results returned with one
variable-length all-gather (RCCL over xGMI).

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port 29500 scripts/run_config4_multi.py [--pairs 64]

Dry run on a one-GPU box (all ranks share device 0, gloo): FM_BENCH_BACKEND=gloo FM_BENCH_SINGLE_DEVICE=1.
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=64)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = 0 if os.environ.get("FM_BENCH_SINGLE_DEVICE") else int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("FM_BENCH_BACKEND", "nccl")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    dev = torch.device("cuda", local_rank) if backend == "nccl" else "cpu"

    import fastmatch_amd
    from fastmatch_amd import synth, cache, fastmatch, sharding
    ctx = fastmatch_amd.Context(local_rank)
    mine = sharding.shard_items(args.pairs, rank, world)
    pairs = []
    for i in mine:
        q, t = synth.image_pair((1000, 1000), 12500, 20250100 + i)
        mc = cache.Metric_Cache.from_arrays(q["descriptors"], q["positions"], q["size"], q["thumb_descriptors"],
                                            q["thumb_positions"], q["thumb_size"], options={"context": ctx})
        fi = cache.Feature_Image(t["size"], t["positions"], t["descriptors"], t["thumb_positions"],
                                 t["thumb_descriptors"], t["thumb_size"])
        pairs.append((mc, fi))
    prepared, stats = [], {}
    fastmatch.match_many(pairs, 0.7, {"context": ctx, "prepared_out": prepared, "return_arrays": True})   # warm-up + setup

    def barrier():
        if world > 1:
            dist.barrier()

    times = []
    for _ in range(args.reps):
        stats.clear()
        barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = fastmatch.match_many(pairs, 0.7, {"context": ctx, "prepared": prepared, "stats": stats, "return_arrays": True})
        # (pair id, query index, distance bits are not needed here: gather index + ratio bits) -> 12-byte rows
        rows = [sharding.pack_matches(np.full(len(r[0]), i, np.int32), r[0], r[2].astype(np.float32))
                for i, r in zip(mine, res)]
        packed = np.concatenate(rows) if rows else np.zeros((0, 3), np.int32)
        gathered = sharding.all_gather_matches(packed, device=dev)
        torch.cuda.synchronize(); barrier()
        times.append(time.perf_counter() - t0)
    tot = torch.tensor([float(stats.get("rounds", 0)), float(stats.get("pairs", 0)), float(len(packed))], dtype=torch.float64,
                       device=dev if backend == "nccl" else "cpu")
    tmax = torch.tensor([min(times)], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(tot)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    if rank == 0:
        t = float(tmax.item())
        assert sum(len(g) for g in gathered) == int(tot[2].item())
        print(json.dumps({"config": 4, "n_gpus": world, "pairs": args.pairs, "wall_s": t, "rounds_per_s": float(tot[0]) / t,
                          "descriptor_pairs_per_s": float(tot[1]) / t, "matches_per_s": float(tot[2]) / t,
                          "matches": int(tot[2].item()), "gather": "all_gather_matches (%s)" % backend}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
