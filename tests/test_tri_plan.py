"""CPU: the workgroup table of the triangular self-distance sweep (fm_self_dist_plan, host code of the library;
rowreduce.hip "TRI").  Metric_Cache's self distances (cache.pyx:250-252, 271-273) need every pair (i, j), i != j,
once: output chunk k (512 rows = 4 stages of 128) must meet every stage >= 4 k exactly once -- its own four stages
in launch A (the masked block on the diagonal), the rest in launch B."""
import numpy as np
import pytest

import fastmatch_amd
from fastmatch_amd import _ffi


@pytest.mark.parametrize("n_pad", [128, 256, 512, 640, 1024, 4224, 33024, 100096, 300032])
@pytest.mark.parametrize("stages", [0, 4, 7, 32, 61, 5000])
def test_every_chunk_meets_every_later_stage_once(n_pad, stages):
    table, n_diag, used = _ffi.self_dist_plan(n_pad, stages)
    nstages = n_pad // 128
    nchunks = (nstages + 3) // 4
    assert n_diag == nchunks
    assert used >= 4 and (stages == 0 or used == max(stages, 4))
    cover = np.zeros((nchunks, nstages), dtype=np.int32)
    for i, (k, s0, s1, _) in enumerate(table):
        assert 0 <= k < nchunks and 4 * k <= s0 < s1 <= nstages
        if i < n_diag:
            assert (k, s0, s1) == (i, 4 * i, min(nstages, 4 * i + 4))
        else:
            assert s0 >= 4 * k + 4 and s1 - s0 <= used
        cover[k, s0:s1] += 1
    want = np.arange(nstages)[None, :] >= 4 * np.arange(nchunks)[:, None]
    assert np.array_equal(cover, want.astype(np.int32))


def _entry(e, nchunks, nstages, S):
    """rowreduce.hip tri_entry, statement by statement: the kernel derives its workgroup from these numbers (no table)."""
    if e < nchunks:
        return e, 4 * e, min(nstages, 4 * e + 4)
    r, i = e - nchunks, 0
    while True:
        cnt = (nstages - 4 - i * S + 3) >> 2
        if r < cnt or cnt <= 0:
            break
        r -= cnt
        i += 1
    st0 = 4 * r + 4 + i * S
    return r, st0, min(nstages, st0 + S)


@pytest.mark.parametrize("n_pad", [128, 256, 512, 640, 1024, 3072, 12544, 33024, 100096, 250112])
@pytest.mark.parametrize("stages", [0, 4, 5, 17, 39, 72, 400])
def test_the_kernels_arithmetic_names_the_planners_workgroups(n_pad, stages):
    """The device has no table (r05: a table per bank size made a NEW size cost a hipMalloc and an upload): entry e of the
    host planner's list == what the kernel computes from (chunks, stages, piece length)."""
    table, n_diag, used = _ffi.self_dist_plan(n_pad, stages)
    nstages = n_pad // 128
    nchunks = (nstages + 3) // 4
    got = np.array([_entry(e, nchunks, nstages, used) for e in range(len(table))], dtype=np.int32).reshape(-1, 3)
    assert np.array_equal(got, table[:, :3])


def test_plan_rejects_bad_sizes():
    for n_pad in (0, 100, -128):
        with pytest.raises(fastmatch_amd.FastMatchHipError):
            _ffi.self_dist_plan(n_pad)


@pytest.mark.parametrize("n_pad", [40064, 65536, 100096, 131072, 160000, 250112])
def test_chosen_piece_length_fills_the_chip(n_pad):
    """stages = 0: launch B's workgroups, list-scheduled on the 512 resident slots, leave no long tail -- and the piece
    length comes from arithmetic over the chunks (r05: the first planner simulated 53 candidates on a heap, 2.3 ms of host
    time per NEW bank size at 100k rows against a 0.6 ms kernel), within a few per cent of the best of all candidates."""
    import heapq

    def makespan(table, n_diag):
        lens = (table[n_diag:, 2] - table[n_diag:, 1]).astype(np.float64) + 2.0
        slots = [0.0] * 512
        for L in lens:
            heapq.heapreplace(slots, slots[0] + L)
        return max(slots), lens.sum() / 512

    table, n_diag, used = _ffi.self_dist_plan(n_pad, 0)
    assert 20 <= used <= 72
    chosen, mean = makespan(table, n_diag)
    assert chosen <= 1.12 * mean
    best = min(makespan(*_ffi.self_dist_plan(n_pad, S)[:2])[0] for S in range(20, 73, 4))
    assert chosen <= 1.04 * best


def test_plan_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    """plan_tri (host code, api_grid.hip) built for the HOST with -fsanitize=address,undefined and driven by
    tests/tools/tri_plan_sanitize.hip over ~1900 (size, piece length) combinations: no report, coverage holds."""
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    exe = str(tmp_path / "tri_plan_sanitize")
    csrc = os.path.join(root, "fast-match_amd", "csrc")
    subprocess.check_call([hipcc, "-std=c++17", "-O1", "-g", "--cuda-host-only", "--offload-arch=gfx950",
                           "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I", csrc, "-I", os.path.join(root, "include"),
                           os.path.join(csrc, "api_grid.hip"), os.path.join(root, "tests", "tools", "tri_plan_sanitize.hip"), "-o", exe],
                          stderr=subprocess.DEVNULL)
    p = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and p.stdout.startswith("ok:"), p.stdout + p.stderr


def test_count_only_call_equals_the_table_and_large_sizes_are_refused():
    """The count-only form (cap 0) is arithmetic in 64 bits and agrees with the walked table; a bank whose workgroup count
    would leave int32 is refused with FM_EINVAL instead of a negative count."""
    import ctypes
    lib = _ffi.load_library()
    for n_pad in (128, 512, 640, 33792, 100096, 1000064):
        for stages in (0, 4, 17, 72):
            nwg, nd, su = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
            assert lib.fm_self_dist_plan(n_pad, stages, None, 0, ctypes.byref(nwg), ctypes.byref(nd), ctypes.byref(su)) == 0
            table, n_diag, used = _ffi.self_dist_plan(n_pad, stages)
            assert (nwg.value, nd.value, su.value) == (table.shape[0], n_diag, used)
    nwg = ctypes.c_int32()
    assert lib.fm_self_dist_plan((1 << 26) + 128, 0, None, 0, ctypes.byref(nwg), None, None) != 0
    assert lib.fm_self_dist_plan(1 << 26, 0, None, 0, ctypes.byref(nwg), None, None) == 0 and nwg.value > 0
