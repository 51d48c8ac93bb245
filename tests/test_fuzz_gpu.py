"""GPU: a fixed slice of the differential fuzz (tests/tools/gpu_fuzz.py) as a regression test -- random banks through
every operator and random image pairs through fastmatch.match(), each against the oracle.  The seeds are fixed, so a
failure names a reproducible problem; the tool itself runs open-ended with fresh seeds."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed0", [1000, 777000, 20261003])
def test_fixed_fuzz_slice_equals_oracle(ctx, seed0):
    import gpu_fuzz
    # (no wall-clock budget: only the fixed seed slice decides, a loaded host cannot end it early)
    # (r06: 12 problems per slice instead of 20 -- the suite's time budget; the open-ended tool runs hundreds per round)
    n, counts = gpu_fuzz.run(budget=float("inf"), seed0=seed0, max_problems=12, context=ctx)
    assert n == 12 and counts.get("match", 0) == 3 and len(counts) >= 3
