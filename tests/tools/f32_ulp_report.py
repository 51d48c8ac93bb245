"""How far is the float32 route's fixed accumulation order (order 1 = fma chain, what the device
computes bit for bit: tests/test_parity_gpu.py) from the orders OpenCV builds use?

  order 0  generic C++ loop unrolled by 4          (OpenCV any version, no SIMD)
  order 2  SSE2 2 x 4 lanes, mul + add             (OpenCV 2.4.x, the reference's era)
  order 3  128-bit universal intrinsics 4 x 4 lanes (OpenCV 3.4 / 4.x baseline)

CPU only (the oracle): BASELINE config 5 data (synthetic SIFT + uniform(-0.5, 0.5), seed
20250005), a sample of the 10k queries against the full 1M-row bank (or a smaller bank with
--bank).  Prints one JSON object: per order the ulp histogram of the 1st/2nd neighbour distances
and the index agreement of the 2-NN lists and of the cross-checked match.
"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import oracle
from fastmatch_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--bank", type=int, default=1000000)
ap.add_argument("--queries", type=int, default=1000)
args = ap.parse_args()
rng = np.random.default_rng(20250005)
T = synth.synth_sift(args.bank, rng).astype(np.float32) + rng.uniform(-0.5, 0.5, (args.bank, 128)).astype(np.float32)
Q = synth.synth_sift(10000, rng).astype(np.float32) + rng.uniform(-0.5, 0.5, (10000, 128)).astype(np.float32)
Q = Q[:args.queries]
ref_i, ref_d = oracle.bf_knn(Q, T, 2, order=1)
xq = min(args.queries, 1000)
ref_x = oracle.bf_xcheck1(Q[:xq], T[:20000], order=1)
out = {"data": "BASELINE config 5 (non-integer float32), %d queries x %d bank rows" % (args.queries, args.bank),
       "reference": "order 1 (fma chain) = the device result, bit for bit", "orders": {}}
names = {0: "generic unrolled-by-4", 2: "SSE2 2x4 lanes (OpenCV 2.4.x)", 3: "128-bit SIMD 4x4 lanes (OpenCV 4.x)"}
for order in (0, 2, 3):
    i, d = oracle.bf_knn(Q, T, 2, order=order)
    ulp = np.abs(d.view(np.int32).astype(np.int64) - ref_d.view(np.int32).astype(np.int64))
    same = i == ref_i
    hist = np.bincount(np.minimum(ulp[same], 8), minlength=9)
    x = oracle.bf_xcheck1(Q[:xq], T[:20000], order=order)
    out["orders"][str(order)] = {
        "name": names[order],
        "knn2_index_agreement": float(same.mean()),
        "knn2_rows_with_any_index_difference": int((~same).any(axis=1).sum()),
        "ulp_histogram_same_index_0_to_8plus": hist.tolist(),
        "max_ulp_same_index": int(ulp[same].max()),
        "max_rel_err_same_index": float(np.max(np.abs(d[same] - ref_d[same]) / ref_d[same])),
        "xcheck_match_agreement_%dx20000" % xq: float((x[0] == ref_x[0]).mean()),
    }
print(json.dumps(out, indent=1))
