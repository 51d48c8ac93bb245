"""Float32-route self distances (Metric_Cache build of a RootSIFT-style bank): the masked full sweep against the triangular
one (option self_tri 0 / 2), kernel ms, bit-equality, a row sample against the oracle is in tests/test_selfdist_gpu.py."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastmatch_amd
from fastmatch_amd import synth

sizes = [int(a) for a in sys.argv[1:]] or [100000]
c = fastmatch_amd.Context(0)
rng = np.random.default_rng(5)
for n in sizes:
    T = (synth.synth_sift(n, rng) + rng.uniform(-0.5, 0.5, (n, 128))).astype(np.float32)
    b = c.bank(T)
    res = {}
    for tri in (0, 2, 0, 2):
        c.set_option("self_tri", tri)
        sd = c.self_dist(b)
        c.reset_stats()
        for _ in range(5):
            c.self_dist(b)
        st = c.stats()
        res[tri] = (sd, st["kernel_ms"] / max(st["kernel_launches"], 1))
        print("n %7d self_tri %d: kernel %.3f ms" % (n, tri, res[tri][1]), flush=True)
    print("n %7d: masked %.3f ms, triangular %.3f ms (x %.2f), same bits %s, filter %s"
          % (n, res[0][1], res[2][1], res[2][1] / res[0][1], np.array_equal(res[0][0].view(np.uint64), res[2][0].view(np.uint64)), c.f32_filter_stats()), flush=True)
