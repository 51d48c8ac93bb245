"""The bench's fresh_pair leg alone (for rocprofv3 --kernel-trace): 12 new 100k x 100k pairs per step through
upload -> self distances -> X1 + R1, two steps in flight.  FM_FRESH_UPLOAD=0 leaves the refills out (kernels only),
FM_FRESH_SELF=0 the self distances."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import fastmatch_amd as fm
from fastmatch_amd import synth

ctx = fm.Context(0)
ctx.set_option("batch_group", 16)
ctx.set_option("batch_tail", 0)
Q, T, _ = synth.planted_pair(bench.NQ, bench.NT, seed=bench.SEED)
print(json.dumps(bench.leg_fresh_pair(ctx, Q, T, 0, None, steps=int(sys.argv[1]) if len(sys.argv) > 1 else 8), indent=1))
