"""BASELINE.json config 2 as ONE problem split over the GPUs of a node (strong scaling, SURVEY.md
8(e)): the 100k-row query bank is replicated, the 100k train rows are sharded by row range; every
rank runs the reverse-NN + election of its shard (fm_xcheck1_keys) and ONE all-reduce(min) of
100k packed keys (RCCL over xGMI) yields the cross-checked 1-NN of the whole problem on every
rank, bit-identical to the single-GPU fm_xcheck1.  Rank 0 checks that and prints one JSON line.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port 29500 scripts/run_config2_sharded.py

Dry run on a one-GPU box (all ranks share device 0, gloo): FM_BENCH_BACKEND=gloo FM_BENCH_SINGLE_DEVICE=1.
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nq", type=int, default=100000)
    ap.add_argument("--nt", type=int, default=100000)
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = 0 if os.environ.get("FM_BENCH_SINGLE_DEVICE") else int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("FM_BENCH_BACKEND", "nccl")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    dev = torch.device("cuda", local_rank) if backend == "nccl" else "cpu"

    import fastmatch_amd
    from fastmatch_amd import synth, sharding
    ctx = fastmatch_amd.Context(local_rank)
    Q, T, _ = synth.planted_pair(args.nq, args.nt, seed=20250002)        # same problem on every rank
    lo, hi = sharding.shard_rows(args.nt, rank, world)
    qb, tb = ctx.bank(Q), ctx.bank(T[lo:hi])
    selfdist = ctx.self_dist(qb)

    def barrier():
        if world > 1:
            dist.barrier()

    sharding.xcheck1_sharded(ctx, qb, tb, lo, device=dev)
    times = []
    for _ in range(args.reps):
        barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        tidx, d = sharding.xcheck1_sharded(ctx, qb, tb, lo, device=dev)
        ratio = d.astype(np.float64) / selfdist                         # R1 on the host (fastmatch.pyx:124)
        accepted = int(np.count_nonzero((ratio < 0.7) & (tidx >= 0)))
        torch.cuda.synchronize(); barrier()
        times.append(time.perf_counter() - t0)
    tmax = torch.tensor([min(times)], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    if rank == 0:
        ft, fd = ctx.xcheck1(qb, ctx.bank(T))                           # the unsharded call, for the check
        same = bool(np.array_equal(ft, tidx) and np.array_equal(fd.view(np.uint32), d.view(np.uint32)))
        t = float(tmax.item())
        print(json.dumps({"config": 2, "mode": "one problem, train rows sharded", "n_gpus": world, "wall_s": t,
                          "pairs_per_s": float(args.nq) * args.nt / t, "accepted": accepted,
                          "identical_to_single_gpu": same, "exchange": "all_reduce(min) of %d uint64 keys (%s)" % (args.nq, backend)}),
              flush=True)
        assert same
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
