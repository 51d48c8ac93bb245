"""The reference's README example (README.md:34-55 / Example.ipynb) on the MI355X path, with a
synthetic pre-extracted image pair standing in for SIFT (cv2 is not required):

    python examples/example.py            # needs an MI355X

With OpenCV installed, replace the two `from_arrays` / `Feature_Image` lines by
`cache.Metric_Cache(path_query)` and `imaging.open_img(path_target)` as in the reference.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastmatch_amd                                   # noqa: F401
from fastmatch_amd import fastmatch, cache, synth

# "images": 800 x 640 like images/graf, ~3000 keypoints each
q, t = synth.image_pair((800, 640), 3000, seed=1)
query_cache = cache.Metric_Cache.from_arrays(q["descriptors"], q["positions"], q["size"],
                                             q["thumb_descriptors"], q["thumb_positions"], q["thumb_size"])
target_img = cache.Feature_Image(t["size"], t["positions"], t["descriptors"],
                                 t["thumb_positions"], t["thumb_descriptors"], t["thumb_size"])

log = []                                               # per-round log as figures.visualize_log consumes it
options = {"log": log}                                 # a log forces the host-driven loop; drop it for the device loop
match_fun = fastmatch.match(query_cache, target_img, options)
matches = match_fun(0.7)

print("%d matches in %d rounds" % (len(matches), len(log)))
index, info = matches[0]
print("first match: query keypoint %d at %s -> target %s, ratio %.3f"
      % (index, info["positions"][0], info["positions"][1], info["ratio"]))
planted = q["planted"]
good = sum(1 for i, m in matches
           if planted[i] >= 0 and np.allclose(m["positions"][1], t["positions"][planted[i]], atol=1e-9))
print("%d of them are planted correspondences" % good)
fast = fastmatch.match(query_cache, target_img, {})(0.7)          # same result from the device-resident loop
assert [m[0] for m in fast] == [m[0] for m in matches]

# The reference's evaluation asks one pair for a whole list of thresholds (turntable.py:59-60): a list goes
# through ONE launch of the device loop, one workgroup per threshold.
taus = [0.5, 0.6, 0.7, 0.8, 0.9]
per_tau = fastmatch.match(query_cache, target_img, {})(taus)
print("matches per threshold:", dict(zip(taus, (len(m) for m in per_tau))))
assert [m[0] for m in per_tau[2]] == [m[0] for m in matches]
