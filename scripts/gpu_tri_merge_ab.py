"""Triangular self sweep: the two launches (diagonal blocks, then the rest) against both phases in ONE grid (FM_TRI_MERGE,
read per launch), interleaved in one process; results must not change.  python scripts/gpu_tri_merge_ab.py [rows]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastmatch_amd as fm
from fastmatch_amd import synth

ctx = fm.Context(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
bank = ctx.bank(synth.synth_sift(N, np.random.default_rng(7)))
banks = [ctx.bank(synth.synth_sift(N, np.random.default_rng(100 + i))) for i in range(12)]
ctx.set_option("self_tri", 1)
ctx.set_option("batch_group", 16)
ref, res = None, {}
for rep in range(5):
    for name in ("two_launches", "merged"):
        os.environ.pop("FM_TRI_MERGE", None)
        if name == "merged":
            os.environ["FM_TRI_MERGE"] = "1"
        ctx.self_dist(bank)
        ctx.reset_stats()
        for _ in range(8):
            sd = ctx.self_dist(bank)
        s = ctx.stats()
        res.setdefault(name, []).append(round(s["kernel_ms"] / 8, 4))
        if ref is None:
            ref = sd
        assert np.array_equal(ref.view(np.uint64), sd.view(np.uint64)), name
        ctx.self_dist_batch(banks, want_host=False); ctx.sync()
        ctx.reset_stats()
        for _ in range(4):
            ctx.self_dist_batch(banks, want_host=False)
        ctx.sync()
        res.setdefault(name + "_batch12_per_bank", []).append(round(ctx.stats()["kernel_ms"] / 48, 4))
os.environ.pop("FM_TRI_MERGE", None)
print(json.dumps(res))
