"""GPU: the N > 1 branch of bench.py (the path the driver's 2 / 4 / 8-GPU runs take) exercised on ONE GPU: two ranks
share device 0, the collectives run over gloo on host tensors (FM_BENCH_BACKEND=gloo FM_BENCH_SINGLE_DEVICE=1), a small
workload (FM_BENCH_NQ / FM_BENCH_NT / FM_BENCH_PAIRS; 33 000 rows keep the batched 8-wave kernel).  The JSON line of the
two-rank run must carry the SUM of what the two ranks' batches give when each is run alone at N = 1 -- rank r's batch is
seeded SEED + r, which FM_BENCH_SEED_OFFSET reproduces on a single rank."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra_env, *args):
    env = dict(os.environ)
    env.update({"FM_BENCH_NQ": "33000", "FM_BENCH_NT": "33000", "FM_BENCH_PAIRS": "3", "FM_BENCH_C3": "0", "FM_BENCH_F32": "0",
                "FM_BENCH_FRESH": "0", "FM_BENCH_C4_PAIRS": "4", "MASTER_ADDR": "127.0.0.1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    env.update(extra_env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"] + list(args),
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout.decode()[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("gather", ["padded", "counted"])
def test_two_rank_bench_equals_the_sum_of_its_ranks(gather):
    two = _bench({"FM_BENCH_BACKEND": "gloo", "FM_BENCH_SINGLE_DEVICE": "1",
                  "FM_BENCH_GATHER": "counted" if gather == "counted" else ""}, "--gpus", "2")
    assert two["n_gpus"] == 2 and two["steps"] == 2 and two["scaling"] == "weak" and two["unit"] == "pairs/s"
    assert two["roofline"]["kernel"].startswith("fm::rowreduce_batch_kernel") and two["roofline"]["image_pairs_per_launch"] == 3
    solo = [_bench({"FM_BENCH_SEED_OFFSET": str(r)}, "--gpus", "1") for r in range(2)]
    assert all(s["n_gpus"] == 1 for s in solo)
    assert two["accepted_matches_per_step"] == sum(s["accepted_matches_per_step"] for s in solo) > 1000
    # the line describes its collective: what was gathered, by whom, how many bytes, what it cost
    c = two["collective"]
    assert c["backend"] == "gloo" and c["world_size"] == 2 and c["ranks_seen"] == 2 and c["gather"] == gather
    assert c["collectives_per_step"] == (1 if gather == "padded" else 2)
    if gather == "padded":
        assert c["rows_per_rank_per_step"] == 3 * 33000 and c["bytes_per_rank_per_step"] == 3 * 33000 * 12 + 3 * 8
    else:
        assert 0 < c["rows_per_rank_per_step"] < 3 * 33000
    assert c["bytes_received_per_rank_per_step"] == c["bytes_per_rank_per_step"]
    assert c["step_ms_with_gather"] > 0 and c["step_ms_without_gather"] > 0
    assert abs(c["gather_exposed_ms"] - (c["step_ms_with_gather"] - c["step_ms_without_gather"])) < 1e-9
    assert all(s["collective"] is None for s in solo)
    # the other multi-GPU mode (second half): ONE problem, train rows sharded, one
    # all-reduce(min) of the election keys -- in the line whenever N > 1, checked against the single-GPU call on rank 0
    t = two["train_sharded"]
    assert t["identical_to_single_gpu"] is True and t["collective"]["op"] == "all_reduce(min)"
    assert t["collective"]["elements"] == 33000 and t["collective"]["bytes_per_rank"] == 33000 * 8 and t["collective"]["backend"] == "gloo"
    assert t["wall_ms"] > 0 and t["shard_kernels_ms"] > 0 and abs(t["exchange_exposed_ms"] - (t["wall_ms"] - t["shard_kernels_ms"])) < 1e-9
    assert t["pairs_per_s"] == pytest.approx(33000.0 * 33000.0 / (t["wall_ms"] * 1e-3))
    assert all(s["train_sharded"] is None for s in solo)
    # the batch's last pair comes from another distribution: its accepted count differs from pair 0's
    assert solo[0]["accepted_matches_independent_pair"] not in (None, solo[0]["accepted_matches_pair0"])
    # configs[3] leg: the same four pairs, sharded over the ranks or not -- same rounds, pairs and matches
    c2, c1 = two["expand_c4"], solo[0]["expand_c4"]
    assert c2["n_gpus"] == 2 and (c2["rounds"], c2["descriptor_pairs"], c2["matches"]) == (c1["rounds"], c1["descriptor_pairs"], c1["matches"])
    s2, s1 = c2["saturating_batch"], c1["saturating_batch"]
    assert s2["runs"] == s1["runs"] == 60 and (s2["rounds"], s2["matches"]) == (s1["rounds"], s1["matches"])
    assert s2["runs_given_up_by_the_device"] == 0
